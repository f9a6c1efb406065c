// GroupNorm(+SiLU) and LayerNorm over channels-last bf16 (gfx950).  HBM-bound: 16-byte vector loads/stores,
// fp32 statistics, deterministic two-stage reduction (no float atomics).
#include "aptp_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ------------------------------------------------------------------------------------------------------------
// GroupNorm
//   stage 1 (gn_stats): grid (nchunk, B); each workgroup reduces a slab of rows to per-group (sum, sumsq)
//   stage 2 (gn_apply): grid (nchunk2, B); folds the nchunk partials, then y = act(x*scale + shift)
// Thread -> channel-octet mapping is fixed for the whole kernel (octet o = tid % TPR (+ 256*page)), so the
// per-channel accumulators / affine coefficients live in registers.
// ------------------------------------------------------------------------------------------------------------
struct GnK {
  const __bf16* x; int64_t ldx; __bf16* y; int64_t ldy;
  int B, HW, C, G, cg, CO;   // CO = ceil(C/8) octets per row
  const float* gamma; const float* beta; float eps; int silu;
  float* ws; int nchunk;
  float* fin;                // finalised [B, G, 2] (mean, rstd), written by gn_finalize_kernel
  int fold_in_apply;         // 1: <= 16 coarse chunks and no finalize launch: every apply workgroup folds the partials
  int* counters;             // optional [B] arrival counters (zero on entry, left zero): the last stage-1 workgroup of a
                             // sample folds the partials itself and gn_finalize_kernel is not launched
  int TPR, RPAR;             // threads per row (= min(CO,256)), rows processed in parallel (256/TPR)
  // round 4: per-(sample, channel unit) fixed-point sums left by the producer(s) (AptpGroupNormColStats.ustats): the apply pass
  // finishes (mean, rstd) itself -- no finalise launch.  Up to two channel segments (skip-concat).
  const long long* us[2]; int us_units[2]; int us_segC0; int us_unit, us_nrep;
};

// stage 1.5: one workgroup per sample folds the nchunk partials ONCE (fixed order => deterministic).  Without it every
// apply workgroup re-read all partials (32 KB each): more bytes than the activation tile it normalises.
__device__ __forceinline__ void gn_finalize_sample(const GnK& p, int b, int tid) {
  const int g = tid >> 3, sub = tid & 7;
  float a = 0.f, a2 = 0.f;
  if (g < p.G) {
    // 8 partials in flight per thread (the chunk loop is a chain of L2 round trips otherwise: 16 of them at HW = 4096);
    // the adds keep the order c = sub, sub + 8, ... => deterministic
    const float2* w = reinterpret_cast<const float2*>(p.ws) + (int64_t)b * p.nchunk * p.G + g;
    for (int c0 = sub; c0 < p.nchunk; c0 += 64) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + 8 * u;
        v[u] = w[(int64_t)(c < p.nchunk ? c : sub) * p.G];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c0 + 8 * u < p.nchunk) { a += v[u].x; a2 += v[u].y; }
    }
  }
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off);
    a2 += __shfl_xor(a2, off);
  }
  if (g < p.G && sub == 0) {
    const float inv = 1.0f / ((float)p.cg * (float)p.HW);
    const float mean = a * inv;
    float var = a2 * inv - mean * mean;
    var = var < 0.f ? 0.f : var;
    p.fin[((int64_t)b * p.G + g) * 2] = mean;
    p.fin[((int64_t)b * p.G + g) * 2 + 1] = rsqrtf(var + p.eps);
  }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const GnK p) { gn_finalize_sample(p, blockIdx.x, threadIdx.x); }

// Finalise from PRODUCER-emitted column statistics (aptp_conv_gemm colstat_out): the tensor is one or two channel
// segments (a skip-concat has two producers), each with per-(row block, channel) (sum, sumsq) partials.  Grid (G, B):
// one workgroup folds the cg channels x row blocks of its (sample, group) in a fixed order and writes (mean, rstd).
struct GnSeg { const float2* st; int ld, rows, C; };
struct GnCols { GnSeg seg[2]; int B, HW, G, cg; float eps; float* fin; };

__global__ __launch_bounds__(256) void gn_finalize_cols_kernel(const GnCols p) {
  __shared__ float red[2][256];
  const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y;
  const int c_lo = g * p.cg, c_hi = c_lo + p.cg;
  float a = 0.f, a2 = 0.f;
  int cbase = 0;
#pragma unroll
  for (int sI = 0; sI < 2; ++sI) {
    const GnSeg sg = p.seg[sI];
    if (sg.st) {
      const int lo = c_lo > cbase ? c_lo : cbase, hi = c_hi < cbase + sg.C ? c_hi : cbase + sg.C;   // channels of this group here
      const int nch = hi - lo;
      if (nch > 0) {
        const int nrb = p.HW / sg.rows;                       // row blocks per sample
        const float2* base = sg.st + (int64_t)b * nrb * sg.ld + (lo - cbase);
        const int cnt = nrb * nch;
        for (int i0 = tid; i0 < cnt; i0 += 1024) {            // 4 partials in flight per thread
          float2 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u < cnt ? i0 + 256 * u : i0;
            const int rb = i / nch, c = i - rb * nch;
            v[u] = base[(int64_t)rb * sg.ld + c];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (i0 + 256 * u < cnt) { a += v[u].x; a2 += v[u].y; }
        }
      }
      cbase += sg.C;
    }
  }
  red[0][tid] = a; red[1][tid] = a2;
  __syncthreads();
#pragma unroll
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) { red[0][tid] += red[0][tid + off]; red[1][tid] += red[1][tid + off]; }
    __syncthreads();
  }
  if (tid == 0) {
    const float inv = 1.0f / ((float)p.cg * (float)p.HW);
    const float mean = red[0][0] * inv;
    float var = red[1][0] * inv - mean * mean;
    var = var < 0.f ? 0.f : var;
    p.fin[((int64_t)b * p.G + g) * 2] = mean;
    p.fin[((int64_t)b * p.G + g) * 2 + 1] = rsqrtf(var + p.eps);
  }
}

template <int NP>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnK p) {
  __shared__ float red[2][2048 * NP];   // [sum|sumsq][RPAR * TPR*8*NP]  (RPAR*TPR <= 256)
  const int tid = threadIdx.x;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = (int)(((int64_t)p.HW * chunk) / p.nchunk), r1 = (int)(((int64_t)p.HW * (chunk + 1)) / p.nchunk);
  const int rl = tid / p.TPR, ot = tid - rl * p.TPR;
  const bool active = rl < p.RPAR;
  float s[NP][8], ss[NP][8];
#pragma unroll
  for (int pg = 0; pg < NP; ++pg)
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[pg][e] = 0.f; ss[pg][e] = 0.f; }
  if (active) {
    const __bf16* base = p.x + ((int64_t)b * p.HW) * p.ldx;
    constexpr int U = 4;   // rows in flight per thread: the loop is latency-bound without it
    for (int r = r0 + rl; r < r1; r += p.RPAR * U) {
      uint4 q[U][NP];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = r + u * p.RPAR;
#pragma unroll
        for (int pg = 0; pg < NP; ++pg) {
          const int o = ot + pg * 256;
          q[u][pg] = make_uint4(0u, 0u, 0u, 0u);
          if (rr < r1 && o < p.CO) q[u][pg] = *reinterpret_cast<const uint4*>(base + (int64_t)rr * p.ldx + o * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int pg = 0; pg < NP; ++pg) {
          float f[8];
          unpack_bf16x8(q[u][pg], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { s[pg][e] += f[e]; ss[pg][e] += f[e] * f[e]; }
        }
    }
  }
  // per-channel partials -> LDS [rl][channel]
  const int CP = p.TPR * 8 * NP;   // padded channel count in LDS rows
  if (active) {
#pragma unroll
    for (int pg = 0; pg < NP; ++pg) {
      const int o = ot + pg * 256;
      if (o < p.CO) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[0][rl * CP + o * 8 + e] = s[pg][e];
          red[1][rl * CP + o * 8 + e] = ss[pg][e];
        }
      }
    }
  }
  __syncthreads();
  // 8 threads per group fold channels x RPAR
  const int g = tid >> 3, sub = tid & 7;
  float a = 0.f, a2 = 0.f;
  if (g < p.G) {
    for (int rr = 0; rr < p.RPAR; ++rr)
      for (int i = sub; i < p.cg; i += 8) {
        const int c = g * p.cg + i;
        a += red[0][rr * CP + c];
        a2 += red[1][rr * CP + c];
      }
  }
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) {
    a += __shfl_xor(a, off);
    a2 += __shfl_xor(a2, off);
  }
  if (g < p.G && sub == 0) {
    float* o = p.ws + (((int64_t)b * p.nchunk + chunk) * p.G + g) * 2;
    if (p.counters) {
      // fused finalize: the partial is one naturally aligned 8-byte agent-scope (write-through) store
      union { float f[2]; unsigned long long u; } pk; pk.f[0] = a; pk.f[1] = a2;
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(o), pk.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      o[0] = a;
      o[1] = a2;
    }
  }
  if (!p.counters) return;
  // Last-arriver finalize (replaces the gn_finalize_kernel launch, 5.5 us + a kernel boundary per GroupNorm): every
  // workgroup drains its write-through partials and takes a ticket on the sample's counter; the one that draws
  // nchunk-1 acquires, folds all partials in the fixed order of gn_finalize_sample (deterministic) and resets the counter.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* flag = reinterpret_cast<int*>(&red[0][0]);
  if (tid == 0) *flag = __hip_atomic_fetch_add(p.counters + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*reinterpret_cast<volatile int*>(flag) != p.nchunk - 1) return;
  if (tid == 0) {
    __hip_atomic_store(p.counters + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  gn_finalize_sample(p, b, tid);
}

// gamma / beta of 8 consecutive channels starting at c0 (a multiple of 8): two 16-byte loads each when the whole octet is
// inside [0, C) -- 4 wave instructions instead of 16 four-byte ones; these kernels stream a few KB per workgroup, so the
// coefficient loads were the majority of their vector-memory instructions -- element-wise at a ragged tail
__device__ __forceinline__ void load_affine8(const float* gamma, const float* beta, int c0, int C, bool on, float ga[8], float be[8]) {
  if (on && c0 + 8 <= C) {
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + c0), b1 = *reinterpret_cast<const float4*>(beta + c0 + 4);
    ga[0] = g0.x; ga[1] = g0.y; ga[2] = g0.z; ga[3] = g0.w; ga[4] = g1.x; ga[5] = g1.y; ga[6] = g1.z; ga[7] = g1.w;
    be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w; be[4] = b1.x; be[5] = b1.y; be[6] = b1.z; be[7] = b1.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = on && c0 + e < C;
      ga[e] = ok ? gamma[c0 + e] : 0.f;
      be[e] = ok ? beta[c0 + e] : 0.f;
    }
  }
}

template <int NP>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnK p) {
  __shared__ float mean_s[32], rstd_s[32];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int rl = tid / p.TPR, ot = tid - rl * p.TPR;
  const bool active = rl < p.RPAR;
  // Issue every independent global load up front (gamma/beta of this thread's channels, the first rows of x, the
  // stage-1 partials) so the workgroup pays ONE memory latency before it starts streaming, not three in a row.
  float ga[NP][8], be[NP][8];
#pragma unroll
  for (int pg = 0; pg < NP; ++pg) {
    const int o = ot + pg * 256;
    load_affine8(p.gamma, p.beta, o * 8, p.C, active && o < p.CO, ga[pg], be[pg]);
  }
  const int r0 = (int)(((int64_t)p.HW * chunk) / gridDim.x), r1 = (int)(((int64_t)p.HW * (chunk + 1)) / gridDim.x);
  const __bf16* xb = p.x + ((int64_t)b * p.HW) * p.ldx;
  __bf16* yb = p.y + ((int64_t)b * p.HW) * p.ldy;
  constexpr int U = (NP == 1) ? 4 : 2;   // rows in flight per thread (x2 with the software pipeline below)
  u32x4 q[U][NP];
  auto load_rows = [&](int r) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = r + u * p.RPAR;
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {
        const int o = ot + pg * 256;
        const bool ok = active && rr < r1 && o < p.CO;
        const int rc = ok ? rr : r0, oc = ok ? o : 0;          // clamped, always-valid address; value unused if !ok
        q[u][pg] = *reinterpret_cast<const u32x4*>(xb + (int64_t)rc * p.ldx + oc * 8);
      }
    }
  };
  load_rows(r0 + rl);
  if (p.us[0]) {
    // 8 threads per group: each takes every 8th (unit, replica) pair of its group, 64-bit integer sums (order-independent),
    // folded over the 8 lanes; then mean / rstd in double (the fixed-point totals carry ~40 significant bits)
    const int g = tid >> 3, sub = tid & 7;
    long long S = 0, S2 = 0;
    if (g < p.G) {
      const int upg = p.cg / p.us_unit, u_lo = g * upg, useg0 = p.us_segC0 / p.us_unit;
      const int n = upg * p.us_nrep;
      for (int i = sub; i < n; i += 8) {
        const int u = u_lo + i / p.us_nrep, r = i % p.us_nrep;
        const int sg = u < useg0 ? 0 : 1, lu = sg ? u - useg0 : u;
        const long long* q = p.us[sg] + (((int64_t)r * p.B + b) * p.us_units[sg] + lu) * 2;
        S += q[0]; S2 += q[1];
      }
    }
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) { S += __shfl_xor(S, off); S2 += __shfl_xor(S2, off); }
    if (g < p.G && sub == 0) {
      const double inv = 1.0 / ((double)p.cg * (double)p.HW);
      const double mean = (double)S * (1.0 / 1048576.0) * inv;
      double var = (double)S2 * (1.0 / 4096.0) * inv - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      mean_s[g] = (float)mean;
      rstd_s[g] = rsqrtf((float)var + p.eps);
    }
  } else
  if (tid < p.G) {
    if (p.fold_in_apply) {
      // two-launch form: at most 16 coarse chunks, all requested at once (one L2 round trip, in flight together with the
      // first rows of x), folded in chunk order => deterministic; 4 KB per workgroup instead of a finalize launch
      const float2* w = reinterpret_cast<const float2*>(p.ws) + (int64_t)b * p.nchunk * p.G + tid;
      float2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = w[(int64_t)(u < p.nchunk ? u : 0) * p.G];
      float a = 0.f, a2 = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (u < p.nchunk) { a += v[u].x; a2 += v[u].y; }
      const float inv = 1.0f / ((float)p.cg * (float)p.HW);
      const float mean = a * inv;
      float var = a2 * inv - mean * mean;
      var = var < 0.f ? 0.f : var;
      mean_s[tid] = mean;
      rstd_s[tid] = rsqrtf(var + p.eps);
    } else {
      mean_s[tid] = p.fin[((int64_t)b * p.G + tid) * 2];
      rstd_s[tid] = p.fin[((int64_t)b * p.G + tid) * 2 + 1];
    }
  }
  __syncthreads();
  if (!active) return;
  float sc[NP][8], sh[NP][8];
  unsigned valid[NP];
#pragma unroll
  for (int pg = 0; pg < NP; ++pg) {
    const int o = ot + pg * 256;
    valid[pg] = 0u;
    const int c0 = o * 8;
    int g = c0 / p.cg, rem = c0 - g * p.cg;   // one division per octet, then walk
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      if (o < p.CO && c < p.C) {
        const float k = rstd_s[g] * ga[pg][e];
        sc[pg][e] = k;
        sh[pg][e] = be[pg][e] - mean_s[g] * k;
        valid[pg] |= 1u << e;
      } else {
        sc[pg][e] = 0.f; sh[pg][e] = 0.f;
      }
      if (++rem == p.cg) { rem = 0; ++g; }
    }
  }
  for (int r = r0 + rl; r < r1; r += p.RPAR * U) {
    u32x4 cur[U][NP];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) cur[u][pg] = q[u][pg];
    if (r + p.RPAR * U < r1) load_rows(r + p.RPAR * U);      // next rows in flight while these are transformed
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = r + u * p.RPAR;
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {
        const int o = ot + pg * 256;
        if (rr < r1 && o < p.CO) {
          float f[8];
          union { u32x4 v; uint4 s; } cv; cv.v = cur[u][pg];
          unpack_bf16x8(cv.s, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = f[e] * sc[pg][e] + sh[pg][e];
            if (p.silu) v = silu_f(v);
            // channels >= C inside the last octet are padding: always written as exact zero
            f[e] = ((valid[pg] >> e) & 1u) ? v : 0.f;
          }
          *reinterpret_cast<uint4*>(yb + (int64_t)rr * p.ldy + o * 8) = pack_bf16x8(f);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Single-launch GroupNorm: one workgroup OWNS gpb whole groups of one sample (a [HW, gpb*cg] column slab; gpb*cg is a
// multiple of 8, so the slab is made of 16-byte chunks) and keeps the slab in registers (U chunks per thread), so the
// statistics never leave the workgroup and x is read once: load -> reduce -> normalise -> store.  No inter-workgroup
// hand-off, deterministic.  Used where the slab fits (HW * gpb*cg/8 <= U * NT chunks); three dependent launches cost
// 13-30 us on those maps, this one 6-12.
// ------------------------------------------------------------------------------------------------------------
struct GnG {
  const __bf16* x; int64_t ldx; __bf16* y; int64_t ldy;
  int B, HW, C, G, cg;
  const float* gamma; const float* beta; float eps; int silu;
  int gpb;        // groups per workgroup (1, 2, 4 or 8)
  int TPR, RPAR;  // chunks per slab row (= gpb*cg/8 <= 32), rows in parallel (NT / TPR)
  int R256;       // rows folded per LDS round (256 / TPR)
  float* stats_out; int nchunk;   // optional [B, nchunk, G, 2] (sum, sumsq) partials for the backward: chunk 0 = the sums, rest 0
};

template <int U, int NT>
__global__ __launch_bounds__(NT) void gn_group_kernel(const GnG p) {
  __shared__ float red[2][2048];
  __shared__ float chs[2][256];
  __shared__ float mean_s[8], rstd_s[8];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, g0 = blockIdx.x * p.gpb;
  const int ng = (p.G - g0 < p.gpb) ? p.G - g0 : p.gpb;
  const int c0 = g0 * p.cg;                         // multiple of 8 by construction
  const int c1 = c0 + ng * p.cg;                    // <= C
  const int w16 = ((c1 + 7) / 8) - c0 / 8;          // chunks of this slab (the last one may hold zero padding)
  const int rl = tid / p.TPR, ot = tid - rl * p.TPR;
  const bool active = rl < p.RPAR && ot < w16;
  __bf16* yb = p.y + ((int64_t)b * p.HW) * p.ldy + c0 + ot * 8;
  // the slab: every load of the thread is issued before the first use.  Buffer loads: rows past the sample (and the
  // lanes without a chunk) read as zero from the out-of-range offset, so no select doubles the registers.
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(p.x + ((int64_t)b * p.HW) * p.ldx), 0, (int)(((int64_t)(p.HW - 1) * p.ldx + ((p.C + 7) / 8) * 8) * 2), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const unsigned col_b = (unsigned)(c0 + ot * 8) * 2u, ld_b = (unsigned)p.ldx * 2u;
  // affine coefficients of this thread's channels, requested together with the slab (they used to be fetched element by
  // element AFTER the reduction: a second dependent memory round trip in a kernel that is one latency chain)
  float gaf[8], bef[8];
  load_affine8(p.gamma, p.beta, c0 + ot * 8, c1, active, gaf, bef);
  u32x4 q[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rr = rl + u * p.RPAR;
    q[u] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, (active && rr < p.HW) ? (unsigned)rr * ld_b + col_b : OOB, 0, 0);
  }
  {
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // opaque touch: the chunk is unpacked HERE and again in the normalise loop, so the slab stays packed in between
      // (4 registers per chunk instead of 8 unpacked floats held across the reduction)
      asm volatile("" : "+v"(q[u].x), "+v"(q[u].y), "+v"(q[u].z), "+v"(q[u].w));
      float f[8];
      union { u32x4 v; uint4 s4; } cv; cv.v = q[u];
      unpack_bf16x8(cv.s4, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
    }
    // per-channel partials of the rows-in-parallel -> LDS, R256 row slots per round (fixed order => deterministic)
    const int CP = p.TPR * 8;
    const int rounds = (p.RPAR + p.R256 - 1) / p.R256;
    const int myround = rl / p.R256, slot = rl - myround * p.R256;
    for (int rd = 0; rd < rounds; ++rd) {
      if (rl < p.RPAR && myround == rd) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int i = slot * CP + ot * 8 + e;
          if (rd == 0) { red[0][i] = s[e]; red[1][i] = ss[e]; }
          else { red[0][i] += s[e]; red[1][i] += ss[e]; }
        }
      }
      __syncthreads();
    }
    // stage 1: thread c folds the row slots of slab channel c (conflict-free LDS walk)
    const int nslot = p.RPAR < p.R256 ? p.RPAR : p.R256;
    if (tid < CP) {
      float a = 0.f, a2 = 0.f;
      for (int rr = 0; rr < nslot; ++rr) { a += red[0][rr * CP + tid]; a2 += red[1][rr * CP + tid]; }
      chs[0][tid] = a;
      chs[1][tid] = a2;
    }
    __syncthreads();
    // stage 2: 32 lanes per owned group fold its cg channels
    const int lg = tid >> 5, l32 = tid & 31;
    float a = 0.f, a2 = 0.f;
    if (lg < ng) {
      for (int i = l32; i < p.cg; i += 32) {
        a += chs[0][lg * p.cg + i];
        a2 += chs[1][lg * p.cg + i];
      }
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
      a += __shfl_xor(a, off);
      a2 += __shfl_xor(a2, off);
    }
    if (lg < ng && l32 == 0) {
      const float inv = 1.0f / ((float)p.cg * (float)p.HW);
      const float mean = a * inv;
      float var = a2 * inv - mean * mean;
      var = var < 0.f ? 0.f : var;
      mean_s[lg] = mean;
      rstd_s[lg] = rsqrtf(var + p.eps);
      if (p.stats_out) {                       // what the three-launch form leaves for groupnorm_bwd, in its layout
        float2* st = reinterpret_cast<float2*>(p.stats_out) + (int64_t)b * p.nchunk * p.G + (g0 + lg);
        st[0] = make_float2(a, a2);
        for (int c = 1; c < p.nchunk; ++c) st[(int64_t)c * p.G] = make_float2(0.f, 0.f);
      }
    }
  }
  __syncthreads();
  if (!active) return;
  float sc[8], sh[8];
  unsigned valid = 0u;
  {
    int lg = (ot * 8) / p.cg, rem = ot * 8 - lg * p.cg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + ot * 8 + e;
      if (c < c1) {
        const float k = rstd_s[lg] * gaf[e];
        sc[e] = k;
        sh[e] = bef[e] - mean_s[lg] * k;
        valid |= 1u << e;
      } else {
        sc[e] = 0.f; sh[e] = 0.f;
      }
      if (++rem == p.cg) { rem = 0; ++lg; }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rr = rl + u * p.RPAR;
    asm volatile("" : "+v"(q[u].x), "+v"(q[u].y), "+v"(q[u].z), "+v"(q[u].w));
    if (rr < p.HW) {
      float f[8];
      union { u32x4 v; uint4 s4; } cv; cv.v = q[u];
      unpack_bf16x8(cv.s4, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = f[e] * sc[e] + sh[e];
        if (p.silu) v = silu_f(v);
        f[e] = ((valid >> e) & 1u) ? v : 0.f;   // padding channels of the last chunk: exact zero
      }
      *reinterpret_cast<uint4*>(yb + (int64_t)rr * p.ldy) = pack_bf16x8(f);
    }
  }
}

template <int NT>
bool launch_group(const GnG& q, int rows, dim3 grid, hipStream_t s) {
  if (rows <= 4) hipLaunchKernelGGL((gn_group_kernel<4, NT>), grid, dim3(NT), 0, s, q);
  else if (rows <= 8) hipLaunchKernelGGL((gn_group_kernel<8, NT>), grid, dim3(NT), 0, s, q);
  else if (rows <= 16) hipLaunchKernelGGL((gn_group_kernel<16, NT>), grid, dim3(NT), 0, s, q);
  else if (rows <= 24 && NT == 256) hipLaunchKernelGGL((gn_group_kernel<24, 256>), grid, dim3(256), 0, s, q);
  else return false;
  return true;
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, lane handles octets lane, lane+64, ...; exact two-pass statistics in registers.
// ------------------------------------------------------------------------------------------------------------
struct LnK {
  const __bf16* x; int64_t ldx; __bf16* y; int64_t ldy; int rows, C, CO;
  const float* gamma; const float* beta; float eps;
};

template <int NO>
__global__ __launch_bounds__(256) void ln_kernel(const LnK p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= p.rows) return;
  float f[NO][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int o = lane + 64 * i;
    if (o < p.CO) {
      const uint4 q = *reinterpret_cast<const uint4*>(p.x + (int64_t)row * p.ldx + o * 8);
      unpack_bf16x8(q, f[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += f[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[i][e] = 0.f;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  const float mean = sum / (float)p.C;
  float vs = 0.f;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int o = lane + 64 * i;
    if (o < p.CO) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; vs += d * d; }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) vs += __shfl_xor(vs, off);
  const float rstd = rsqrtf(vs / (float)p.C + p.eps);
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int o = lane + 64 * i;
    if (o < p.CO) {
      const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + o * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(p.gamma + o * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.beta + o * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(p.beta + o * 8 + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = (f[i][e] - mean) * rstd * gg[e] + bb[e];
      *reinterpret_cast<uint4*>(p.y + (int64_t)row * p.ldy + o * 8) = pack_bf16x8(o8);
    }
  }
}

}  // namespace

#ifndef APTP_GN_GROUP_MAX_HW
#define APTP_GN_GROUP_MAX_HW 256
#endif

extern "C" int aptp_groupnorm_nchunk(int HW) {
  // >= 16 rows per stage-1 workgroup on the large maps; on the small ones (HW <= 256: U-Net levels 16 and 8, where a row is
  // 1280-2560 channels) 4 rows, or the row-chunked kernels of the training path -- GroupNorm statistics / backward, gate and
  // depth-gate backward, column sums -- run on 16-64 workgroups (GroupNorm backward of a 0.7 MB map: 30 us)
  int n = HW <= 256 ? HW / 4 : HW / 16;
  if (n < 1) n = 1;
  if (n > 128) n = 128;
  return n;
}

// Row chunks of the kernels that have no batch dimension in their grid (layernorm_pgrad; colsum with batch == 1): the cap of the
// per-sample rule (128) would put 128 workgroups on 16384 rows -- half the chip, 128 rows each
extern "C" int aptp_rows_nchunk(int rows) {
  int n = rows <= 1024 ? rows / 4 : rows / 16;
  if (n < 1) n = 1;
  if (n > 1024) n = 1024;
  return n;
}

extern "C" int64_t aptp_groupnorm_workspace_bytes(const AptpGroupNormParams* p) {
  if (!p) return 0;
  // [B, nchunk, G, 2] partials followed by [B, G, 2] finalised (mean, rstd)
  return (int64_t)p->B * (aptp_groupnorm_nchunk(p->HW) + 1) * p->groups * 2 * (int64_t)sizeof(float);
}

extern "C" int aptp_groupnorm(const AptpGroupNormParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->x && p->y && p->gamma && p->beta && p->workspace, "groupnorm: null pointer");
  APTP_CHECK(p->B > 0 && p->HW > 0 && p->C > 0 && p->groups > 0 && p->groups <= 32, "groupnorm: bad extents (groups <= 32)");
  APTP_CHECK(p->C % p->groups == 0, "groupnorm: C (%d) not divisible by groups (%d)", p->C, p->groups);
  if (p->io_f32) return aptp_groupnorm_f32(p, stream);
  const int CO = (p->C + 7) / 8;
  APTP_CHECK(p->ldx % 8 == 0 && p->ldy % 8 == 0 && p->ldx >= CO * 8 && p->ldy >= CO * 8, "groupnorm: ld must be a multiple of 8 and >= roundup8(C)");
  APTP_CHECK(((uintptr_t)p->x % 16) == 0 && ((uintptr_t)p->y % 16) == 0, "groupnorm: pointer alignment");
  APTP_CHECK(CO <= 512, "groupnorm: C too large (max 4096)");
  GnK k;
  k.x = (const __bf16*)p->x; k.ldx = p->ldx; k.y = (__bf16*)p->y; k.ldy = p->ldy;
  k.B = p->B; k.HW = p->HW; k.C = p->C; k.G = p->groups; k.cg = p->C / p->groups; k.CO = CO;
  k.gamma = p->gamma; k.beta = p->beta; k.eps = p->eps; k.silu = p->silu;
  k.ws = (float*)p->workspace; k.nchunk = aptp_groupnorm_nchunk(p->HW);
  k.fin = k.ws + (int64_t)p->B * k.nchunk * p->groups * 2;
  k.counters = p->counters;
  k.fold_in_apply = 0;
  k.TPR = CO < 256 ? CO : 256;
  k.RPAR = 256 / k.TPR;
  hipStream_t s = (hipStream_t)stream;
  APTP_CHECK(p->variant >= 0 && p->variant <= 4, "groupnorm: variant %d", p->variant);
  // group-owner single launch: gpb = fewest groups whose channels fill whole 16-byte chunks
  int gpb = 1;
  while ((gpb * k.cg) % 8 != 0) gpb *= 2;   // cg * 8 is always a multiple of 8, so gpb <= 8
  const int tpr = gpb * k.cg / 8;
  if (p->variant != 1 && tpr <= 32 && p->colstats[0].stats == nullptr) {
    GnG q;
    q.x = k.x; q.ldx = k.ldx; q.y = k.y; q.ldy = k.ldy; q.B = k.B; q.HW = k.HW; q.C = k.C; q.G = k.G; q.cg = k.cg;
    q.gamma = k.gamma; q.beta = k.beta; q.eps = k.eps; q.silu = k.silu;
    q.gpb = gpb; q.TPR = tpr; q.R256 = 256 / tpr;
    q.stats_out = p->variant == 4 ? k.ws : nullptr; q.nchunk = k.nchunk;
    const dim3 grid((k.G + gpb - 1) / gpb, k.B);
    bool done = false;
    q.RPAR = 256 / tpr;
    int rows = (k.HW + q.RPAR - 1) / q.RPAR;
    // variant 0: only the small maps.  One workgroup streams its whole slab through one CU, and a CU's outstanding-miss
    // budget makes that slower than three chip-wide launches from HW = 1024 up (tools/bench_gn.py on MI355X: 6-12 us vs
    // 11-30 us at HW <= 256; 14-32 vs 13-24 at HW = 1024; 40-50 vs 21-35 at HW = 4096).
    if (p->variant == 2 || k.HW <= APTP_GN_GROUP_MAX_HW) {
      if (rows <= 24) {
        done = launch_group<256>(q, rows, grid, s);
      } else {
        q.RPAR = 1024 / tpr;
        rows = (k.HW + q.RPAR - 1) / q.RPAR;
        if (rows <= 16) done = launch_group<1024>(q, rows, grid, s);
      }
    }
    if (done) {
      APTP_LAUNCH_CHECK();
      return APTP_OK;
    }
  }
  APTP_CHECK(p->variant != 2, "groupnorm: variant 2 does not fit (HW %d, cg %d: slab too large for one workgroup's registers)", p->HW, k.cg);
  if (p->variant == 3) {
    // two launches: coarse statistics chunks (>= 256 rows each, at most 16) that every apply workgroup folds itself
    int nc = p->HW / 256;
    nc = nc < 1 ? 1 : (nc > 16 ? 16 : nc);
    k.nchunk = nc;
    k.fold_in_apply = 1;
    k.counters = nullptr;
  }
  dim3 grid1(k.nchunk, p->B);
  const bool from_cols = p->colstats[0].stats != nullptr;
  k.us[0] = k.us[1] = nullptr; k.us_units[0] = k.us_units[1] = 0; k.us_segC0 = 0; k.us_unit = 1; k.us_nrep = 1;
  bool from_units = from_cols && p->colstats[0].ustats != nullptr;
  if (from_units) {
    // unit statistics: every segment must have them, with one unit size / replica count that divides the group size and the segments
    const int unit = p->colstats[0].unit, nrep = p->colstats[0].nrep;
    int ctot = 0;
    for (int i = 0; i < 2 && from_units; ++i) {
      const AptpGroupNormColStats& d = p->colstats[i];
      if (!d.stats) continue;
      from_units = d.ustats && d.unit == unit && d.nrep == nrep && unit > 0 && nrep >= 1 && d.C % unit == 0 && d.units * unit >= d.C &&
                   ((uintptr_t)d.ustats % 8) == 0;
      ctot += d.C;
    }
    from_units = from_units && ctot == p->C && (p->C / p->groups) % unit == 0;
    if (from_units) {
      for (int i = 0; i < 2; ++i) { k.us[i] = (const long long*)p->colstats[i].ustats; k.us_units[i] = p->colstats[i].units; }
      k.us_segC0 = p->colstats[0].C; k.us_unit = unit; k.us_nrep = nrep;
    }
  }
  if (from_cols && !from_units) {
    GnCols c;
    int ctot = 0;
    for (int i = 0; i < 2; ++i) {
      const AptpGroupNormColStats& d = p->colstats[i];
      c.seg[i].st = reinterpret_cast<const float2*>(d.stats); c.seg[i].ld = d.ld; c.seg[i].rows = d.rows_per_block; c.seg[i].C = d.C;
      if (d.stats) {
        APTP_CHECK(d.rows_per_block > 0 && p->HW % d.rows_per_block == 0 && d.C > 0 && d.ld >= d.C && ((uintptr_t)d.stats % 8) == 0,
                   "groupnorm: bad colstats segment %d (rows_per_block must divide HW, ld >= C)", i);
        ctot += d.C;
      }
    }
    APTP_CHECK(ctot == p->C, "groupnorm: colstats segments cover %d channels, tensor has %d", ctot, p->C);
    c.B = p->B; c.HW = p->HW; c.G = p->groups; c.cg = p->C / p->groups; c.eps = p->eps; c.fin = k.fin;
    k.counters = nullptr; k.fold_in_apply = 0;
    hipLaunchKernelGGL(gn_finalize_cols_kernel, dim3(p->groups, p->B), dim3(256), 0, s, c);
  }
  int nchunk2 = p->HW / 16;   // >= 16 rows per apply workgroup, up to 8 resident workgroups per CU
  if (nchunk2 < 1) nchunk2 = 1;
  if (nchunk2 > 256) nchunk2 = 256;
  dim3 grid2(nchunk2, p->B);
  if (CO <= 256) {
    if (!from_cols) hipLaunchKernelGGL(gn_stats_kernel<1>, grid1, dim3(256), 0, s, k);
    if (!from_cols && !k.counters && !k.fold_in_apply) hipLaunchKernelGGL(gn_finalize_kernel, dim3(p->B), dim3(256), 0, s, k);
    hipLaunchKernelGGL(gn_apply_kernel<1>, grid2, dim3(256), 0, s, k);
  } else {
    if (!from_cols) hipLaunchKernelGGL(gn_stats_kernel<2>, grid1, dim3(256), 0, s, k);
    if (!from_cols && !k.counters && !k.fold_in_apply) hipLaunchKernelGGL(gn_finalize_kernel, dim3(p->B), dim3(256), 0, s, k);
    hipLaunchKernelGGL(gn_apply_kernel<2>, grid2, dim3(256), 0, s, k);
  }
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}

extern "C" int aptp_layernorm(const AptpLayerNormParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->x && p->y && p->gamma && p->beta, "layernorm: null pointer");
  if (p->io_f32) {
    APTP_CHECK(p->rows > 0 && p->C > 0 && p->ldx >= p->C && p->ldy >= p->C, "layernorm(io_f32): bad extents");
    return aptp_layernorm_f32(p, stream);
  }
  APTP_CHECK(p->rows > 0 && p->C > 0 && p->C % 8 == 0 && p->C <= 2048, "layernorm: C (%d) must be a multiple of 8, <= 2048", p->C);
  APTP_CHECK(p->ldx % 8 == 0 && p->ldy % 8 == 0 && p->ldx >= p->C && p->ldy >= p->C, "layernorm: ld");
  APTP_CHECK(((uintptr_t)p->x % 16) == 0 && ((uintptr_t)p->y % 16) == 0 && ((uintptr_t)p->gamma % 16) == 0 && ((uintptr_t)p->beta % 16) == 0, "layernorm: pointer alignment");
  LnK k;
  k.x = (const __bf16*)p->x; k.ldx = p->ldx; k.y = (__bf16*)p->y; k.ldy = p->ldy;
  k.rows = p->rows; k.C = p->C; k.CO = p->C / 8; k.gamma = p->gamma; k.beta = p->beta; k.eps = p->eps;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((p->rows + 3) / 4);
  const int no = (k.CO + 63) / 64;
  switch (no) {
    case 1: hipLaunchKernelGGL(ln_kernel<1>, grid, dim3(256), 0, s, k); break;
    case 2: hipLaunchKernelGGL(ln_kernel<2>, grid, dim3(256), 0, s, k); break;
    case 3: hipLaunchKernelGGL(ln_kernel<3>, grid, dim3(256), 0, s, k); break;
    default: hipLaunchKernelGGL(ln_kernel<4>, grid, dim3(256), 0, s, k); break;
  }
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}
