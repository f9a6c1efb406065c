// Flash-style scaled-dot-product attention, head_dim 64, bf16 in/out, fp32 softmax state (gfx950).
//
// Workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries.  Per 64-key tile:
//   S^T[key][q] = K . Q^T      (mfma_f32_32x32x16_bf16, K fragment = A operand, Q^T = B operand held in registers)
//   online softmax in registers: a lane holds 32 of the 64 scores of ONE query, the lane+32 partner the rest
//   O^T[d][q]  += V^T . P^T    (the S accumulator, converted to bf16, is directly the B operand; V^T comes from an
//                               LDS image written transposed with the matching key permutation)
// K/V tiles are prefetched into registers while the previous tile computes (issue-early / write-late).
#include "aptp_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // (HIP's uint4 struct makes value selects go through scratch)
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct AttnK {
  const __bf16* q; int64_t qsb, qsl;
  const __bf16* k; int64_t ksb, ksl;
  const __bf16* v; int64_t vsb, vsl;
  __bf16* o; int64_t osb, osl;
  int B, H, Lq, Lk;
  float c;   // scale * log2(e)
  float* lse;   // optional [B, H, Lq]: log2-domain log-sum-exp of the scaled scores (for the backward pass)
};

// position of key offset o (0..15) inside its 16-key group of the V^T image: o = 8a + 4h + c  ->  8h + 4a + c
__device__ __forceinline__ int vt_pos(int key) {
  const int o = key & 15;
  return (key & ~15) | ((o & 4) << 1) | ((o & 8) >> 1) | (o & 3);
}

// NG = number of key-range groups per workgroup (1 or 2).  With NG = 2 the workgroup has 8 waves: waves 0-3 sweep the
// first half of the key tiles, waves 4-7 the second half, for the SAME 128 queries, and the two partial softmax states
// are merged through LDS at the end.  This doubles the waves per SIMD when B*heads*Lq/32 alone cannot fill the chip
// (SD-2.1 level 0: 1024 waves for 1024 SIMDs), so one wave's softmax VALU work overlaps the other's MFMAs.
template <int NG>
__global__ __launch_bounds__(256 * NG) void attn_fwd_kernel(const AttnK p) {
  // one LDS array: per group a [key][d] K image (chunks XOR-swizzled by (key>>1)&7) and a [d][vt_pos(key)] V^T image
  // (chunks XOR-swizzled by (d>>1)&7); reused at the end for the group merge (NG == 2)
  constexpr int KV_ELEMS = 2 * 64 * 64;
  constexpr int MERGE_FLOATS = (NG >= 2) ? (NG / 2) * 256 * 34 : 0;     // one merge round: the upper half publishes
  constexpr int LDS_BYTES = (NG * KV_ELEMS * 2 > MERGE_FLOATS * 4) ? NG * KV_ELEMS * 2 : MERGE_FLOATS * 4;
  __shared__ __attribute__((aligned(16))) char lds_raw[LDS_BYTES];
  const int grp = (NG >= 2) ? (int)(threadIdx.x >> 8) : 0;
  __bf16* Ks = reinterpret_cast<__bf16*>(lds_raw) + grp * KV_ELEMS;
  __bf16* Vt = Ks + 64 * 64;

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;

  const __bf16* qp = p.q + (int64_t)b * p.qsb + (int64_t)h * 64;
  const __bf16* kp = p.k + (int64_t)b * p.ksb + (int64_t)h * 64;
  const __bf16* vp = p.v + (int64_t)b * p.vsb + (int64_t)h * 64;

  // ---- Q^T fragments (B operand: lane (q, hh) element j = Q[q][16s + 8hh + j]) -----------------------------------
  bf16x8 qf[4];
  {
    const int qrow = q0 + lq;
    const bool ok = qrow < p.Lq;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int qc = ok ? qrow : p.Lq - 1;
      u32x4 t = *reinterpret_cast<const u32x4*>(qp + (int64_t)qc * p.qsl + 16 * s + 8 * hh);
      t = ok ? t : (u32x4){0u, 0u, 0u, 0u};
      qf[s] = __builtin_bit_cast(bf16x8, t);
    }
    // Retire the Q loads HERE.  Otherwise their first use is the first MFMA of the key loop, behind the conditionally
    // issued K/V prefetch of that iteration: the compiler cannot count those loads and emits s_waitcnt vmcnt(0) there,
    // in EVERY iteration, which drains the prefetch it was meant to overlap (one exposed L2 round trip per key tile).
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" :: "v"(qf[s]));
  }

  // ---- staging coordinates ------------------------------------------------------------------------------------
  const int chunk = tid & 7;
  const int krow = tid >> 3;        // K: rows krow, krow+32
  const int kpair = tid >> 3;       // V: keys 2*kpair, 2*kpair+1
  u32x4 kreg[2], vreg[2];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // per-thread row pointers of the tile being prefetched, advanced by 64 keys per tile (the per-tile 64-bit address
  // arithmetic was 34 of the loop's ~280 vector instructions, and the loop is bound by vector issue)
  const __bf16* kptr[2];
  const __bf16* vptr[2];
  auto load_kv = [&](int tile) {
    const int key0 = tile * 64;
    if (key0 + 64 <= p.Lk) {                   // full tile (wave-uniform)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        kreg[i] = *reinterpret_cast<const u32x4*>(kptr[i]);
        vreg[i] = *reinterpret_cast<const u32x4*>(vptr[i]);
        kptr[i] += 64 * p.ksl;
        vptr[i] += 64 * p.vsl;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // rows past Lk: load a clamped (valid) row unconditionally; store_kv zeroes the VALUE (a pointer select against a
      // local zero would put the staging registers in scratch, and a value select HERE makes the compiler wait for the
      // prefetch -- s_waitcnt vmcnt(0) -- right after issuing it, ahead of the MFMAs it is meant to overlap)
      const int key = key0 + krow + 32 * i;
      const int keyc = key < p.Lk ? key : p.Lk - 1;
      kreg[i] = *reinterpret_cast<const u32x4*>(kp + (int64_t)keyc * p.ksl + chunk * 8);
      const int vkey = key0 + 2 * kpair + i;
      const int vkeyc = vkey < p.Lk ? vkey : p.Lk - 1;
      vreg[i] = *reinterpret_cast<const u32x4*>(vp + (int64_t)vkeyc * p.vsl + chunk * 8);
    }
  };
  auto store_kv = [&](int tile) {
    if (tile * 64 + 64 > p.Lk) {                 // ragged last tile (wave-uniform): zero the rows past Lk
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        kreg[i] = tile * 64 + krow + 32 * i < p.Lk ? kreg[i] : zero4;
        vreg[i] = tile * 64 + 2 * kpair + i < p.Lk ? vreg[i] : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = krow + 32 * i;
      const int sw = chunk ^ ((r >> 1) & 7);
      *reinterpret_cast<u32x4*>(Ks + r * 64 + sw * 8) = kreg[i];
    }
    // transposed V: dword (V[2kp][d], V[2kp+1][d]) -> Vt[d][vt_pos(2kp) .. +1]
    const int pos = vt_pos(2 * kpair);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = chunk * 8 + e;
      const int sw = (pos >> 3) ^ ((d >> 1) & 7);
      const uint32_t a = vreg[0][e >> 1], b = vreg[1][e >> 1];
      const uint32_t w = (e & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
      *reinterpret_cast<uint32_t*>(Vt + d * 64 + sw * 8 + (pos & 7)) = w;
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[u][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles_all = (p.Lk + 63) / 64;
  const int per_grp = (ntiles_all + NG - 1) / NG;
  const int t_begin = grp * per_grp;
  const int t_end = (t_begin + per_grp < ntiles_all) ? t_begin + per_grp : ntiles_all;   // may be empty for grp 1
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    kptr[i] = kp + (int64_t)(t_begin * 64 + krow + 32 * i) * p.ksl + chunk * 8;
    vptr[i] = vp + (int64_t)(t_begin * 64 + 2 * kpair + i) * p.vsl + chunk * 8;
  }
  if (t_begin < t_end) load_kv(t_begin);
  for (int it = 0; it < per_grp; ++it) {       // uniform trip count: both groups meet at every barrier
    const int tile = t_begin + it;
    const bool active = tile < t_end;
    __syncthreads();   // everyone finished reading the previous tile
    if (active) store_kv(tile);
    __syncthreads();
    if (!active) continue;
    if (tile + 1 < t_end) load_kv(tile + 1);

    // ---- S^T = K . Q^T -------------------------------------------------------------------------------------------
    f32x16 sacc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r_ = 32 * t + lq;
        const int sw = (2 * s + hh) ^ ((r_ >> 1) & 7);
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + r_ * 64 + sw * 8);
        sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[t], 0, 0, 0);
      }
    }
    // sacc[t][r] = score(key = tile*64 + 32t + (r&3) + 8(r>>2) + 4hh, query = q0 + lq)
    const int key_base = tile * 64 + 4 * hh;
    // VALU budget (the level-64 self-attention is exp/VALU-bound, not MFMA-bound): the softmax scale is folded into the
    // exp2 argument (one fma per score instead of a multiply pass + a subtract), the running maximum is taken over the
    // raw scores (c > 0), and the accumulator is only rescaled when some lane's maximum moved.
    float mx = -INFINITY;
    if (tile * 64 + 64 <= p.Lk) {              // full tile (wave-uniform): no key masking
      // v_max3_f32 directly: fmaxf() on MFMA results makes hipcc canonicalise every operand first (v_max_f32 x, x),
      // 56 instructions for 32 scores instead of 16
      // (hipcc's hazard recognizer does not look into inline asm: one compiler-visible read of each accumulator first, so that
      //  the wait states an MFMA result needs are inserted in front of it -- see attn_fwd_sp_kernel, where their absence was a race)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        mx = fmaxf(mx, sacc[t][0]);
#pragma unroll
        for (int r = 0; r < 16; r += 2) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(sacc[t][r]), "v"(sacc[t][r + 1]));
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key_base + 32 * t + (r & 3) + 8 * (r >> 2);
          const float sv = key < p.Lk ? sacc[t][r] : -INFINITY;
          sacc[t][r] = sv;
          mx = fmaxf(mx, sv);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * p.c;
    const float m_new = fmaxf(m_run, mx);       // finite: every tile holds >= 1 valid key
    // (raw v_exp_f32: arguments are <= 0, so the range scaling / denormal fix-up exp2f() wraps around the instruction --
    // compare, two selects, an add and an ldexp per score -- buys nothing; results below 2^-126 flush to zero)
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // m_run = -inf on the first tile -> 0
    // scale + shift and the row sum on packed fp32 pairs (v_pk_fma_f32 / v_pk_add_f32: half the instructions)
    f32x2 rs2 = {0.f, 0.f};
    const f32x2 c2 = {p.c, p.c}, nm2 = {-m_new, -m_new};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 x2 = {sacc[t][r], sacc[t][r + 1]};
        x2 = __builtin_elementwise_fma(x2, c2, nm2);               // masked keys: -inf * c - m = -inf -> 0
        f32x2 pv2;
        pv2[0] = __builtin_amdgcn_exp2f(x2[0]);
        pv2[1] = __builtin_amdgcn_exp2f(x2[1]);
        sacc[t][r] = pv2[0];
        sacc[t][r + 1] = pv2[1];
        rs2 += pv2;
      }
    float rs = rs2[0] + rs2[1];
    rs += __shfl_xor(rs, 32);
    l_run = l_run * alpha + rs;
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {                // wave-uniform: after the first tiles the maximum rarely moves
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[u][r] *= alpha;
    }

    // ---- O^T += V^T . P^T ----------------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)sacc[t][8 * s2 + j];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int d = 32 * u + lq;
          const int sw = (4 * t + 2 * s2 + hh) ^ ((d >> 1) & 7);
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vt + d * 64 + sw * 8);
          oacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[u], 0, 0, 0);
        }
      }
    }
  }

  if (NG >= 2) {
    // merge the key-range groups pairwise (a tree: NG/2 groups publish (m, l, O) per lane, their partners combine, ...)
    float* mg = reinterpret_cast<float*>(lds_raw);
    __syncthreads();                            // all K/V tile reads are done; the LDS array is free
#pragma unroll
    for (int half = NG / 2; half >= 1; half >>= 1) {
      if (grp >= half && grp < 2 * half) {
        float* dst = mg + ((grp - half) * 256 + tid) * 34;
        dst[0] = m_run; dst[1] = l_run;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[2 + u * 16 + r] = oacc[u][r];
      }
      __syncthreads();
      if (grp < half) {
        const float* src = mg + (grp * 256 + tid) * 34;
        const float m1 = src[0], l1 = src[1];
        const float m = fmaxf(m_run, m1);
        // a group that owns no key tile carries m = -inf, l = 0, O = 0 and must weigh 0 (never exp2(-inf + inf))
        const float a0 = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_run - m);
        const float a1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m1 - m);
        l_run = l_run * a0 + l1 * a1;
        m_run = m;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[u][r] = oacc[u][r] * a0 + src[2 + u * 16 + r] * a1;
      }
      __syncthreads();
    }
    if (grp != 0) return;
  }

  // ---- epilogue: O[q][d] = oacc / l -----------------------------------------------------------------------------------
  const int qrow = q0 + lq;
  if (qrow < p.Lq) {
    if (p.lse && hh == 0) p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = m_run + log2f(l_run);
    const float inv = 1.0f / l_run;
    __bf16* op = p.o + (int64_t)b * p.osb + (int64_t)qrow * p.osl + (int64_t)h * 64;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * u + 8 * g + 4 * hh;
        uint2 w;
        w.x = pack_bf16x2(oacc[u][4 * g + 0] * inv, oacc[u][4 * g + 1] * inv);
        w.y = pack_bf16x2(oacc[u][4 * g + 2] * inv, oacc[u][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + d) = w;
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Staggered form of the two-group kernel (AptpAttentionParams.variant = 1; a measured negative result, see the launcher) (8 waves: waves w and w+4 share a SIMD and sweep the two halves of the key range
// for the same 128 queries).  In attn_fwd_kernel<2> both groups run the same phase at the same time, so on every SIMD
// the two waves' MFMAs meet in the matrix pipe and their softmax VALU work meets in the vector pipe: the per-tile cost
// is the SUM of both (the level-64 self-attention of SD-2.1 ran at ~380 TFLOP/s).  Here every key tile is two phases
// separated by workgroup barriers,
//   A: S^T = K.Q^T (8 MFMAs), running maximum, exp2, row sum                      (vector-heavy)
//   B: P -> bf16, O^T += V^T.P^T (8 MFMAs), stage the next K/V tile into the other LDS buffer, request the tile after it
// and group 1 runs ONE phase behind group 0 (one extra barrier before its loop, one after group 0's), so a SIMD always
// pairs a phase-A wave with a phase-B wave: the matrix work of one hides under the vector work of the other.  K/V are
// double-buffered per group, which also removes the second barrier of the lock-step loop (a tile is staged while the
// previous one is still being read).
// ---------------------------------------------------------------------------------------------------------------------
template <bool STAGGER>
__global__ __launch_bounds__(512) void attn_fwd_pp_kernel(const AttnK p) {
  constexpr int KV_ELEMS = 2 * 64 * 64;                 // one K image + one V^T image
  constexpr int MERGE_FLOATS = 256 * 34;
  constexpr int LDS_BYTES = (4 * KV_ELEMS * 2 > MERGE_FLOATS * 4) ? 4 * KV_ELEMS * 2 : MERGE_FLOATS * 4;   // 64 KiB
  __shared__ __attribute__((aligned(16))) char lds_raw[LDS_BYTES];
  const int grp = (int)(threadIdx.x >> 8);
  __bf16* const kv_base = reinterpret_cast<__bf16*>(lds_raw) + grp * 2 * KV_ELEMS;    // [buffer][K | V^T]

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;

  const __bf16* qp = p.q + (int64_t)b * p.qsb + (int64_t)h * 64;
  const __bf16* kp = p.k + (int64_t)b * p.ksb + (int64_t)h * 64;
  const __bf16* vp = p.v + (int64_t)b * p.vsb + (int64_t)h * 64;

  bf16x8 qf[4];
  {
    const int qrow = q0 + lq;
    const bool ok = qrow < p.Lq;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int qc = ok ? qrow : p.Lq - 1;
      u32x4 t = *reinterpret_cast<const u32x4*>(qp + (int64_t)qc * p.qsl + 16 * s + 8 * hh);
      t = ok ? t : (u32x4){0u, 0u, 0u, 0u};
      qf[s] = __builtin_bit_cast(bf16x8, t);
    }
    // Retire the Q loads HERE.  Otherwise their first use is the first MFMA of the key loop, behind the conditionally
    // issued K/V prefetch of that iteration: the compiler cannot count those loads and emits s_waitcnt vmcnt(0) there,
    // in EVERY iteration, which drains the prefetch it was meant to overlap (one exposed L2 round trip per key tile).
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" :: "v"(qf[s]));
  }

  const int chunk = tid & 7;
  const int krow = tid >> 3;
  const int kpair = tid >> 3;
  u32x4 kreg[2], vreg[2];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto load_kv = [&](int tile) {
    const int key0 = tile * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = key0 + krow + 32 * i;
      const int keyc = key < p.Lk ? key : p.Lk - 1;
      kreg[i] = *reinterpret_cast<const u32x4*>(kp + (int64_t)keyc * p.ksl + chunk * 8);
      const int vkey = key0 + 2 * kpair + i;
      const int vkeyc = vkey < p.Lk ? vkey : p.Lk - 1;
      vreg[i] = *reinterpret_cast<const u32x4*>(vp + (int64_t)vkeyc * p.vsl + chunk * 8);
    }
  };
  auto store_kv = [&](int buf, int tile) {
    __bf16* Ks = kv_base + buf * KV_ELEMS;
    __bf16* Vt = Ks + 64 * 64;
    if (tile * 64 + 64 > p.Lk) {                 // ragged last tile (wave-uniform): zero the rows past Lk
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        kreg[i] = tile * 64 + krow + 32 * i < p.Lk ? kreg[i] : zero4;
        vreg[i] = tile * 64 + 2 * kpair + i < p.Lk ? vreg[i] : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = krow + 32 * i;
      const int sw = chunk ^ ((r >> 1) & 7);
      *reinterpret_cast<u32x4*>(Ks + r * 64 + sw * 8) = kreg[i];
    }
    const int pos = vt_pos(2 * kpair);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = chunk * 8 + e;
      const int sw = (pos >> 3) ^ ((d >> 1) & 7);
      const uint32_t a = vreg[0][e >> 1], bb = vreg[1][e >> 1];
      const uint32_t w = (e & 1) ? ((a >> 16) | (bb & 0xffff0000u)) : ((a & 0xffffu) | (bb << 16));
      *reinterpret_cast<uint32_t*>(Vt + d * 64 + sw * 8 + (pos & 7)) = w;
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[u][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles_all = (p.Lk + 63) / 64;
  const int per_grp = (ntiles_all + 1) / 2;
  const int t_begin = grp * per_grp;
  const int t_end = (t_begin + per_grp < ntiles_all) ? t_begin + per_grp : ntiles_all;   // may be empty for group 1

  // prologue: tile t_begin staged in buffer 0, tile t_begin + 1 requested
  if (t_begin < t_end) { load_kv(t_begin); store_kv(0, t_begin); }
  if (t_begin + 1 < t_end) load_kv(t_begin + 1);
  __syncthreads();
  if (STAGGER && grp == 1) __syncthreads();      // group 1 runs one phase behind

  f32x16 sacc[2];
  for (int it = 0; it < per_grp; ++it) {         // uniform trip count: both groups meet at every barrier
    const int tile = t_begin + it;
    const bool active = tile < t_end;
    const __bf16* Ks = kv_base + (it & 1) * KV_ELEMS;
    const __bf16* Vt = Ks + 64 * 64;
    // ---- phase A: scores and softmax ---------------------------------------------------------------------------------
    if (active) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[t][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r_ = 32 * t + lq;
          const int sw = (2 * s + hh) ^ ((r_ >> 1) & 7);
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + r_ * 64 + sw * 8);
          sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[t], 0, 0, 0);
        }
      }
      const int key_base = tile * 64 + 4 * hh;
      float mx = -INFINITY;
      if (tile * 64 + 64 <= p.Lk) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; r += 2) mx = fmaxf(mx, fmaxf(sacc[t][r], sacc[t][r + 1]));
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key_base + 32 * t + (r & 3) + 8 * (r >> 2);
            const float sv = key < p.Lk ? sacc[t][r] : -INFINITY;
            sacc[t][r] = sv;
            mx = fmaxf(mx, sv);
          }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32)) * p.c;
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float rs = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[t][r], p.c, -m_new));
          sacc[t][r] = pv;
          rs += pv;
        }
      rs += __shfl_xor(rs, 32);
      l_run = l_run * alpha + rs;
      m_run = m_new;
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[u][r] *= alpha;
      }
    }
    if (STAGGER) __syncthreads();                // (!STAGGER: double-buffered lock-step form, ONE barrier per key tile)
    // ---- phase B: O^T += V^T . P^T, stage the next tile, request the one after -----------------------------------------
    if (active) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = (__bf16)sacc[t][8 * s2 + j];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int d = 32 * u + lq;
            const int sw = (4 * t + 2 * s2 + hh) ^ ((d >> 1) & 7);
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vt + d * 64 + sw * 8);
            oacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[u], 0, 0, 0);
          }
        }
      }
      // the other buffer was last read in phases A/B of the previous tile: at least one barrier ago
      if (tile + 1 < t_end) store_kv((it + 1) & 1, tile + 1);
      if (tile + 2 < t_end) load_kv(tile + 2);
    }
    __syncthreads();
  }
  if (STAGGER && grp == 0) __syncthreads();      // pairs group 1's last barrier

  // merge the two key-range groups: group 1 publishes (m, l, O) per lane, group 0 combines
  float* mg = reinterpret_cast<float*>(lds_raw);
  __syncthreads();
  if (grp == 1) {
    float* dst = mg + tid * 34;
    dst[0] = m_run; dst[1] = l_run;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[2 + u * 16 + r] = oacc[u][r];
  }
  __syncthreads();
  if (grp == 1) return;
  {
    const float* src = mg + tid * 34;
    const float m1 = src[0], l1 = src[1];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);   // m_run is finite (group 0 owns >= 1 tile)
    l_run = l_run * a0 + l1 * a1;
    m_run = m;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[u][r] = oacc[u][r] * a0 + src[2 + u * 16 + r] * a1;
  }

  const int qrow = q0 + lq;
  if (qrow < p.Lq) {
    if (p.lse && hh == 0) p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = m_run + log2f(l_run);
    const float inv = 1.0f / l_run;
    __bf16* op = p.o + (int64_t)b * p.osb + (int64_t)qrow * p.osl + (int64_t)h * 64;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * u + 8 * g + 4 * hh;
        uint2 w;
        w.x = pack_bf16x2(oacc[u][4 * g + 0] * inv, oacc[u][4 * g + 1] * inv);
        w.y = pack_bf16x2(oacc[u][4 * g + 2] * inv, oacc[u][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + d) = w;
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Software-pipelined form of the double-buffered two-group kernel (AptpAttentionParams.variant = 6; long key ranges with
// Lk % 128 == 0, Lq % 128 == 0).  In attn_fwd_pp_kernel<false> a wave runs QK^T -> softmax -> PV strictly in sequence and the
// two waves of a SIMD do so in lock step (one barrier per key tile), so the matrix pipe idles during both waves' softmax
// (~950 issue cycles of VALU per wave and tile against 16 MFMAs of 32) and the vector pipe during both waves' MFMAs.  Here a
// wave computes the scores of tile i+1 (8 MFMAs that depend on nothing the vector unit is working on) WHILE it runs the
// softmax of tile i, and P.V of tile i while it transposes / stages the following K and V tiles: every MFMA has independent
// vector work next to it in the same basic block.  K is staged two tiles ahead, V one (separate rings, still one barrier
// per tile); the loop is unrolled by two so that the two score accumulators swap roles without moves; tile indices are
// clamped instead of branched on, so that the body stays one scheduling region per phase.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void attn_fwd_sp_kernel(const AttnK p) {
  constexpr int KV_ELEMS = 2 * 64 * 64;                 // one K image + one V^T image
  constexpr int MERGE_FLOATS = 256 * 34;
  constexpr int LDS_BYTES = (4 * KV_ELEMS * 2 > MERGE_FLOATS * 4) ? 4 * KV_ELEMS * 2 : MERGE_FLOATS * 4;   // 64 KiB
  __shared__ __attribute__((aligned(16))) char lds_raw[LDS_BYTES];
  const int grp = (int)(threadIdx.x >> 8);
  __bf16* const kv_base = reinterpret_cast<__bf16*>(lds_raw) + grp * 2 * KV_ELEMS;    // [buffer][K | V^T]

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;

  const __bf16* qp = p.q + (int64_t)b * p.qsb + (int64_t)h * 64;
  const __bf16* kp = p.k + (int64_t)b * p.ksb + (int64_t)h * 64;
  const __bf16* vp = p.v + (int64_t)b * p.vsb + (int64_t)h * 64;

  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qp + (int64_t)(q0 + lq) * p.qsl + 16 * s + 8 * hh));
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" :: "v"(qf[s]));       // retire the Q loads here (see attn_fwd_pp_kernel)

  const int chunk = tid & 7;
  const int krow = tid >> 3;
  u32x4 kra[2], vra[2], krb[2], vrb[2];        // two register sets: a tile's K / V are requested ~1.75 tile periods before their LDS store

  const int n = (p.Lk / 64) / 2;                   // tiles per group (host: Lk % 128 == 0, n >= 2)
  const int t_begin = grp * n;
  const int t_last = t_begin + n - 1;

  // wave-uniform tile base + 32-bit per-thread byte offset (host: 64 rows of K / V span < 2 GiB): no 64-bit vector address math
  uint32_t koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    koff[i] = (uint32_t)(((krow + 32 * i) * p.ksl + chunk * 8) * 2);
    voff[i] = (uint32_t)(((2 * krow + i) * p.vsl + chunk * 8) * 2);
  }
  auto load_k = [&](u32x4 (&kreg)[2], int tile) {
    tile = __builtin_amdgcn_readfirstlane(tile < t_last ? tile : t_last);
    const char* base = reinterpret_cast<const char*>(kp + (int64_t)tile * 64 * p.ksl);
#pragma unroll
    for (int i = 0; i < 2; ++i) kreg[i] = *reinterpret_cast<const u32x4*>(base + koff[i]);
  };
  auto load_v = [&](u32x4 (&vreg)[2], int tile) {
    tile = __builtin_amdgcn_readfirstlane(tile < t_last ? tile : t_last);
    const char* base = reinterpret_cast<const char*>(vp + (int64_t)tile * 64 * p.vsl);
#pragma unroll
    for (int i = 0; i < 2; ++i) vreg[i] = *reinterpret_cast<const u32x4*>(base + voff[i]);
  };
  // max(a, b, c) without the canonicalising v_max hipcc puts in front of fmaxf on MFMA outputs; the other half-wave's value
  // (one asm statement per 8 scores: between separate statements hipcc puts an s_nop)
  auto max8 = [](float m, const float* v) {
    asm("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %0, %0, %7, %8"
        : "+v"(m) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
    return m;
  };
  // (x of lanes 0-31 in both halves, x of lanes 32-63 in both halves): v_permlane32_swap, no LDS round trip
  auto halves = [](float x, float& lo, float& hi) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    lo = __builtin_bit_cast(float, (unsigned)r[0]);
    hi = __builtin_bit_cast(float, (unsigned)r[1]);
  };
  auto store_k = [&](const u32x4 (&kreg)[2], int buf) {
    __bf16* Ks = kv_base + buf * KV_ELEMS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = krow + 32 * i;
      const int sw = chunk ^ ((r >> 1) & 7);
      *reinterpret_cast<u32x4*>(Ks + r * 64 + sw * 8) = kreg[i];
    }
  };
  // V is stored ROW-major like K (two 16-byte stores per thread, keys 2*krow and 2*krow + 1) and read TRANSPOSED by the hardware
  // (ds_read_b64_tr_b16: a 16-lane group gets a 4-key x 16-d block column-major): no software transpose, no scattered 4-byte
  // stores (they were 8 per thread and tile, several-way bank conflicts: rows of the V^T image are bank-aligned).  Chunk c of key
  // r sits at chunk c ^ 4*((r >> 1) & 1): the four keys of a block then cover disjoint 64-byte spans of the 256-byte bank line.
  auto store_v = [&](const u32x4 (&vreg)[2], int buf) {
    __bf16* Vs = kv_base + buf * KV_ELEMS + 64 * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(Vs + (2 * krow + i) * 64 + ((chunk ^ ((krow & 1) << 2)) * 8)) = vreg[i];
  };
  f32x16 oacc[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[u][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // hipcc on its own emits each phase as [fragment reads + MFMAs][all the vector work]: the in-order wave then sits through
  // its 8 MFMAs (256 cycles) before it issues a single softmax instruction.  The body below is written slot by slot -- one
  // MFMA plus >= 32 issue cycles of independent vector work -- and sched_barrier(0) between the slots pins that order.
#define APTP_SLOT_END() __builtin_amdgcn_sched_barrier(0)
#ifndef APTP_ATTN_ABL
#define APTP_ATTN_ABL 0      // timing experiments only (wrong results): 1 no exp2, 2 no loop barrier, 4 no LDS staging stores, 8 no MFMAs,
#endif                       // 16 fragments from registers (no LDS fragment reads), 64 per-phase cycle totals of block 0 into p.lse
  unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define APTP_ATTN_STAMP(i) do { if constexpr ((APTP_ATTN_ABL & 64) != 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
    if ((i) > 0) st_acc[(i) - 1] += t_ - st_prev; st_prev = t_; } } while (0)
  // K fragment j = (s, t) = (j >> 1, j & 1) of the tile in K buffer `buf`; V fragment j = (t, s2, u) = (j >> 2, (j >> 1) & 1, j & 1)
  auto read_k = [&](bf16x8 (&kf)[8], int buf) {
    const __bf16* Ks = kv_base + buf * KV_ELEMS;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r_ = 32 * (j & 1) + lq;
      const int sw = (2 * (j >> 1) + hh) ^ ((r_ >> 1) & 7);
      kf[j] = *reinterpret_cast<const bf16x8*>(Ks + r_ * 64 + sw * 8);
    }
  };
  // V^T fragment j = (t, s2, u): lane (hh, gb = (lane >> 4) & 1, q_ = (lane >> 2) & 3, p_ = lane & 3) addresses key
  // 32t + 16s2 + 4hh + q_ (+ 8 for the second half), d = 32u + 16gb + 4p_ and receives V[32t + 16s2 + 4hh + {0..3} (+8)][32u + 16gb + (lane & 15)]
  // -- element j of the operand <-> key (j & 3) + 8 (j >> 2) + 4hh of the 16-key slot, the order the P fragments are in
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const int tr_q = (lane >> 2) & 3, tr_p = lane & 3, tr_gb = (lane >> 4) & 1;
  int tr_off[2];                                    // element offset inside a V buffer for u = 0, 1 at (t, s2) = (0, 0), first half
#pragma unroll
  for (int u = 0; u < 2; ++u)
    tr_off[u] = (4 * hh + tr_q) * 64 + (4 * (u ^ (tr_q >> 1)) + 2 * tr_gb + (tr_p >> 1)) * 8 + 4 * (tr_p & 1);
  auto read_v = [&](bf16x8 (&vf)[8], int buf) {
    const __bf16* Vs = kv_base + buf * KV_ELEMS + 64 * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16* a = Vs + tr_off[j & 1] + (32 * (j >> 2) + 16 * ((j >> 1) & 1)) * 64;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 8 * 64));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      vf[j] = __builtin_bit_cast(bf16x8, both);
    }
  };
  // one tile: scores of the NEXT tile (K buffer kbuf) into nxt while the softmax of cur runs; P.V of cur (V buffer vbuf) while
  // K(i+2) / V(i+1) are written to LDS.  WITH_NEXT = false: the last tile of the group (no scores, no staging).
  // kreg / vreg: K(i+2) / V(i+1), requested during the previous tile, stored here; kreq / vreq: K(i+3) / V(i+2), requested here
  auto tile_step = [&](auto with_next, f32x16 (&cur)[2], f32x16 (&nxt)[2], int kbuf, int vbuf, int kst, int vst, int tk, int tv,
                       const u32x4 (&kreg)[2], const u32x4 (&vreg)[2], u32x4 (&kreq)[2], u32x4 (&vreq)[2]) {
    constexpr bool WITH_NEXT = decltype(with_next)::value;
    bf16x8 kf[8], vf[8], pf[4];
    APTP_ATTN_STAMP(0);
    if constexpr (WITH_NEXT) {
      load_k(kreq, tk);
      load_v(vreq, tv);
      if constexpr ((APTP_ATTN_ABL & 16) != 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] = qf[j & 3];
      } else read_k(kf, kbuf);
    }
    // score MFMA j of the next tile (accumulator t = j & 1, Q fragment s = j >> 1)
    auto qk = [&](int j) {
      if constexpr (WITH_NEXT && !(APTP_ATTN_ABL & 8)) nxt[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j], qf[j >> 1], j < 2 ? zero16 : nxt[j & 1], 0, 0, 0);
    };
    auto pv = [&](int j) { if constexpr (!(APTP_ATTN_ABL & 8)) oacc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[j], pf[j >> 1], oacc[j & 1], 0, 0, 0); };
    float rs = 0.f, m_new;
    // half a P fragment: exp2 / row sum / bf16 pack of 4 scores (fragment q = (t, s2) = (q >> 1, q & 1), half hf)
    auto exp4 = [&](int q, int hf) {
#pragma unroll
      for (int j = 4 * hf; j < 4 * hf + 4; ++j) {
        const float x = __builtin_fmaf(cur[q >> 1][8 * (q & 1) + j], p.c, -m_new);
        const float e = (APTP_ATTN_ABL & 1) ? x : __builtin_amdgcn_exp2f(x);
        rs += e;
        pf[q][j] = (__bf16)e;
      }
    };
    auto stage_v = [&]() {
      if constexpr (WITH_NEXT && !(APTP_ATTN_ABL & 4)) store_v(vreg, vst);      // V(i+1) over V(i-1)
    };
    // A: running maximum (2 slots of 16 scores) next to score MFMAs 0, 1
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      qk(j);
      // The asm below reads MFMA results, and hipcc's hazard recognizer does not look into inline asm: cur[1]'s last writer (score
      // MFMA 7, which the scheduler sinks to the top of this tile) was 3 MFMAs + ~20 instructions upstream of the first v_max3 on
      // its registers and no wait states were inserted -- outputs changed from run to run.  One compiler-visible read of the
      // accumulator first: the required wait states are inserted in front of it.
      mx = fmaxf(mx, cur[j][0]);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float v8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v8[r] = cur[j][8 * hf + r];
        mx = max8(mx, v8);
      }
      APTP_SLOT_END();
    }
    APTP_ATTN_STAMP(1);
    if constexpr ((APTP_ATTN_ABL & 16) != 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) vf[j] = qf[j & 3];
    } else read_v(vf, vbuf);
    float mlo, mhi;
    halves(mx, mlo, mhi);
    mx = fmaxf(mlo, mhi) * p.c;
    // (A deferred reference maximum -- moved only on jumps above 2^6, so that O is rescaled in the first tiles of a row only -- was
    //  3.5 % faster and is NOT used: the largest P of a tile is then no longer exactly 1.0, its bf16 rounding error is not cancelled
    //  by the unrounded row sum, and the full-size gate-gradient test's error rose from 1.1e-2 to 3.5e-2.)
    m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[u][r] *= alpha;
    }
    APTP_ATTN_STAMP(2);
    // C0: P fragment 0 next to score MFMAs 2, 3
    qk(2); exp4(0, 0); APTP_SLOT_END();
    qk(3); exp4(0, 1); APTP_SLOT_END();
    APTP_ATTN_STAMP(3);
    // C1-C3: P fragment q next to the P.V MFMAs of fragment q - 1 and one more score MFMA
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      pv(2 * q - 2); qk(3 + q); exp4(q, 0); APTP_SLOT_END();
      pv(2 * q - 1); exp4(q, 1); APTP_SLOT_END();
    }
    APTP_ATTN_STAMP(4);
    float slo, shi;
    halves(rs, slo, shi);
    l_run = l_run * alpha + (slo + shi);
    m_run = m_new;
    // D: the last two P.V MFMAs and the last score MFMA next to the LDS stores of V(i+1) and K(i+2)
    pv(6); stage_v(); APTP_SLOT_END();
    pv(7); qk(7); if constexpr (WITH_NEXT && !(APTP_ATTN_ABL & 4)) store_k(kreg, kst);      // K(i+2) over K(i) (past the end: a clamped copy nobody reads)
    APTP_ATTN_STAMP(5);
    if constexpr (WITH_NEXT && !(APTP_ATTN_ABL & 2)) __syncthreads();
    APTP_ATTN_STAMP(6);
  };

  // prologue: K(0), V(0), K(1) staged; S(0) computed; K(2), V(1) requested
  load_k(kra, t_begin); load_v(vra, t_begin);
  store_k(kra, 0); store_v(vra, 0);
  load_k(kra, t_begin + 1);
  store_k(kra, 1);
  __syncthreads();
  f32x16 sa[2], sb[2];
  {
    bf16x8 kf[8];
    read_k(kf, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) sa[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j], qf[j >> 1], j < 2 ? zero16 : sa[j & 1], 0, 0, 0);
  }
  load_k(kra, t_begin + 2); load_v(vra, t_begin + 1);
  __syncthreads();                                  // every wave has read K(0): its buffer may be overwritten from here on

  // step i: S(i+1) next to softmax(i); P.V(i) next to staging K(i+2) -> K buffer i&1, V(i+1) -> V buffer (i+1)&1, then K(i+3), V(i+2) requested
  auto body = [&](int i, f32x16 (&cur)[2], f32x16 (&nxt)[2], u32x4 (&k0)[2], u32x4 (&v0)[2], u32x4 (&k1)[2], u32x4 (&v1)[2]) {
    tile_step(std::true_type{}, cur, nxt, (i + 1) & 1, i & 1, i & 1, (i + 1) & 1, t_begin + i + 3, t_begin + i + 2, k0, v0, k1, v1);
  };
  int i = 0;
  for (; i + 2 < n; i += 2) {
    body(i, sa, sb, kra, vra, krb, vrb);
    body(i + 1, sb, sa, krb, vrb, kra, vra);
  }
  if (i + 1 < n) {                                  // n even: one more full step, the last tile's scores are in sb
    body(i, sa, sb, kra, vra, krb, vrb);
    tile_step(std::false_type{}, sb, sa, 0, (i + 1) & 1, 0, 0, 0, 0, kra, vra, krb, vrb);
  } else {
    tile_step(std::false_type{}, sa, sb, 0, i & 1, 0, 0, 0, 0, kra, vra, krb, vrb);
  }

  // merge the two key-range groups: group 1 publishes (m, l, O) per lane, group 0 combines
  float* mg = reinterpret_cast<float*>(lds_raw);
  __syncthreads();
  if (grp == 1) {
    float* dst = mg + tid * 34;
    dst[0] = m_run; dst[1] = l_run;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[2 + u * 16 + r] = oacc[u][r];
  }
  __syncthreads();
  if (grp == 1) return;
  {
    const float* src = mg + tid * 34;
    const float m1 = src[0], l1 = src[1];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
    l_run = l_run * a0 + l1 * a1;
    m_run = m;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[u][r] = oacc[u][r] * a0 + src[2 + u * 16 + r] * a1;
  }
  const int qrow = q0 + lq;
  if constexpr ((APTP_ATTN_ABL & 64) != 0) {
    if (p.lse && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0)
      for (int j = 0; j < 6; ++j) p.lse[wave * 8 + j] = (float)st_acc[j] / (float)n;
  } else if (p.lse && hh == 0) p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = m_run + log2f(l_run);
  const float inv = 1.0f / l_run;
  __bf16* op = p.o + (int64_t)b * p.osb + (int64_t)qrow * p.osl + (int64_t)h * 64;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * u + 8 * g + 4 * hh;
      uint2 w;
      w.x = pack_bf16x2(oacc[u][4 * g + 0] * inv, oacc[u][4 * g + 1] * inv);
      w.y = pack_bf16x2(oacc[u][4 * g + 2] * inv, oacc[u][4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(op + d) = w;
    }
}

}  // namespace

extern "C" int aptp_attention(const AptpAttentionParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->q && p->k && p->v && p->o, "attention: null pointer");
  APTP_CHECK(p->B > 0 && p->heads > 0 && p->Lq > 0 && p->Lk > 0, "attention: bad extents");
  if (p->io_f32) return aptp_attention_f32(p, stream);
  APTP_CHECK(p->q_stride_l % 8 == 0 && p->k_stride_l % 8 == 0 && p->v_stride_l % 8 == 0 && p->o_stride_l % 4 == 0, "attention: row strides must be multiples of 8 elements");
  APTP_CHECK(p->q_stride_b % 8 == 0 && p->k_stride_b % 8 == 0 && p->v_stride_b % 8 == 0 && p->o_stride_b % 4 == 0, "attention: batch strides must be multiples of 8 elements");
  APTP_CHECK(((uintptr_t)p->q % 16) == 0 && ((uintptr_t)p->k % 16) == 0 && ((uintptr_t)p->v % 16) == 0 && ((uintptr_t)p->o % 8) == 0, "attention: pointer alignment");
  APTP_CHECK(p->q_stride_l >= (int64_t)p->heads * 64 && p->k_stride_l >= (int64_t)p->heads * 64 && p->v_stride_l >= (int64_t)p->heads * 64 && p->o_stride_l >= (int64_t)p->heads * 64, "attention: row stride < heads*64");
  APTP_CHECK(p->heads <= 65535 && p->B <= 65535, "attention: grid limits");
  AttnK k;
  k.q = (const __bf16*)p->q; k.qsb = p->q_stride_b; k.qsl = p->q_stride_l;
  k.k = (const __bf16*)p->k; k.ksb = p->k_stride_b; k.ksl = p->k_stride_l;
  k.v = (const __bf16*)p->v; k.vsb = p->v_stride_b; k.vsl = p->v_stride_l;
  k.o = (__bf16*)p->o; k.osb = p->o_stride_b; k.osl = p->o_stride_l;
  k.B = p->B; k.H = p->heads; k.Lq = p->Lq; k.Lk = p->Lk;
  k.c = p->scale * 1.44269504088896340736f;
  k.lse = p->lse;
  dim3 grid((p->Lq + 127) / 128, p->heads, p->B);
  // Two key-range wave groups (8 waves) when the query-parallel grid alone gives < 2 waves per SIMD, and -- whatever the
  // occupancy -- on long key ranges, where the double-buffered form (one barrier per key tile) wins even on a full chip
  // (dense SD-2.1 level 64, B=4, 5 heads: 168 vs 183 us; masked, 2 heads: 64-68 vs 91 us; tools/bench_attn.py).
  // An explicit variant forces its kernel.
  const int64_t waves = (int64_t)grid.x * grid.y * grid.z * 4;
  const int ntiles = (p->Lk + 63) / 64;
  const bool two_groups = ntiles >= 2 && p->variant != 5 && (waves < 2 * 1024 || ntiles >= 32 || p->variant != 0);
  // Software-pipelined two-group form (round 4): whole 128-row query blocks, an even number >= 4 of whole key tiles, K / V rows of a
  // tile within 2 GiB.  Faster than every other form wherever it applies (tools/bench_attn.py, MI355X: level-64 self-attention
  // 51-53 us against 64-68 masked and 130-135 against 168 dense; 1024 keys 16.3 against 18.8; 256 keys 7.5 against 8.0).
  if ((p->variant == 0 || p->variant == 6) && p->Lk % 128 == 0 && p->Lk >= 256 && p->Lq % 128 == 0 &&
      p->k_stride_l * 64 * 2 < (1ll << 31) && p->v_stride_l * 64 * 2 < (1ll << 31)) {
    hipLaunchKernelGGL(attn_fwd_sp_kernel, grid, dim3(512), 0, (hipStream_t)stream, k);
    APTP_LAUNCH_CHECK();
    return APTP_OK;
  }
  if (two_groups) {
    // The staggered form measured 5-10 % SLOWER than the lock-step one on MI355X (level-64 self-attention of SD-2.1:
    // 75.9 vs 69.3 us; tools/bench_attn.py), so it is opt-in only.
    if (p->variant == 1) hipLaunchKernelGGL(attn_fwd_pp_kernel<true>, grid, dim3(512), 0, (hipStream_t)stream, k);     // staggered groups
    // double-buffered K/V, ONE barrier per key tile: -10 % on long key ranges (65.7 vs 73.2 us at level 64), no gain below
    else if (p->variant == 4 || (p->variant == 0 && ntiles >= 32))
      hipLaunchKernelGGL(attn_fwd_pp_kernel<false>, grid, dim3(512), 0, (hipStream_t)stream, k);
    else if (p->variant == 2)      // four key-range groups (16 waves): measured slower (90 vs 71 us at level 64), opt-in only
      hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, dim3(1024), 0, (hipStream_t)stream, k);
    else hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, dim3(512), 0, (hipStream_t)stream, k);                     // two lock-step groups (also variant 3)
  }
  else hipLaunchKernelGGL(attn_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, k);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}
