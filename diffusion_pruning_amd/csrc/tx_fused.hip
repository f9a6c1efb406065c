// Fused tail of a transformer block for the large-M levels (SD-2.1 level 64: M = B*4096 tokens, C = 320):
//     LN3 -> GEGLU projection -> ff.net[2] + residual -> proj_out + residual            (blocks.py:41-50,121-129,821-823; the
//     Transformer2DModel tail proj_out / "+ residual", blocks.py:1294-1308)
// as ONE kernel per 64-token row tile.  Everything after the cross-attention is row-local, and at this level the per-launch
// route is bound by activation bytes (each projection moves 21-42 MB for 3-13 GFLOP): here the residual
// stream h stays in LDS, the [M, 4C*keep] GEGLU intermediate -- the largest activation of the block -- never exists outside a
// 64 x 128 LDS tile, and HBM sees h and x once on the way in and the block output once on the way out (3 launches -> 1).
//
// Workgroup = 64 rows, 8 waves as 2 (rows) x 4 (columns).  LDS: the residual-stream tile H [C/64][64][64] bf16 in the
// XOR-swizzled 128-byte-row image of conv_gemm (read as the MFMA activation operand, updated in place h2 -> h3), the GEGLU tile
// F [2][64][64], a two-stage ring of weight tiles [rows][64], per-row LayerNorm statistics, the epilogue constants (bias',
// column sums, biases).  The weight stream (1.4 MB per tile pass at level 64, re-read from L2 by every workgroup) is what bounds
// this kernel: with ONE 32-40 KB tile in flight per CU and ~1.2 us from request to LDS it ran at 70 us per launch (= the three
// launches it replaces); so the tiles are register-staged TWO ahead (two sets of 16-byte loads per thread: tile t+2 is requested
// while tile t is multiplied and tile t+1 is written to its ring stage), which doubles the bytes in flight per CU.  Per hidden chunk of 128 units: K = C steps of the LN-folded, (h,g)-interleaved projection into a 64 x 256 fp32
// tile (epilogue: rstd*(acc - mean*colsum) + bias, h*gelu(g) -> F), then two K-steps of ff.net[2] accumulate F x W2 into the
// 64 x C fp32 tile that survives all chunks.  Then h3 = acc + b2 + h2 goes back into H, proj_out runs over it, and the tile
// leaves through an fp32 staging image in the (now idle) ring: 16-byte coalesced x loads and y stores, column statistics of
// the rounded outputs for the next GroupNorm (one (sum, sumsq) per (tile, channel): rows_per_block = 64).
// Rounding points are those of the three separate launches (h3, f and the output are bf16; LayerNorm statistics come from the
// bf16 values of h2), so results agree with them up to fp32 accumulation order.
#include "aptp_common.h"

namespace {

struct FfK {
  const __bf16* h; int64_t ldh; const __bf16* x; int64_t ldx; __bf16* y; int64_t ldy;
  const __bf16* w1; const float* b1; const float* cs1; int n1, ld1;      // [n1][ld1]: LN-folded, interleaved 16 h | 16 g
  const __bf16* w2; const float* b2; int ld2;                              // [C][ld2]  (ld2 = padded hidden width)
  const __bf16* w3; const float* b3; int ld3;                              // [C][ld3]
  float* colstat; int colstat_ld;
  int M, nchunks; float eps;
};

template <int NF3>
__global__ __launch_bounds__(512) void ff_tail_kernel(const FfK p) {
  constexpr int C = 64 * NF3, KC = NF3, NC = C / 4;        // channels, K-chunks of the residual stream, columns per wave
  constexpr int WROWS = C > 256 ? C : 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* H = reinterpret_cast<__bf16*>(smem_raw);          // [KC][64][64]
  __bf16* F = H + KC * 64 * 64;                             // [2][64][64]
  __bf16* ring = F + 2 * 64 * 64;                           // [2][WROWS][64]
  float* stat = reinterpret_cast<float*>(ring + 2 * WROWS * 64);   // [64][2] mean, rstd

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int frow = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * 64;

  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  const int drow = tid >> 3;                                 // row of a 64-row DMA pass this lane fills
  const int schunk = (tid & 7) ^ ((drow >> 1) & 7);          // source chunk that lands in this lane's slot

  // ---- residual-stream tile -> H ---------------------------------------------------------------------------------------------
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const char* src = reinterpret_cast<const char*>(p.h + (int64_t)(m0 + drow) * p.ldh + kc * 64 + schunk * 8);
    __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(H + (kc * 64 + wave * 8) * 64), 16, 0, 0);
  }

  const int T1 = p.nchunks * (KC + 2), T = T1 + KC;
  constexpr int NP = WROWS / 64;                             // 64-row passes of the largest tile = 16-byte loads per thread
  struct TileSet { uint4 v[NP]; };
  auto tile_geom = [&](int t, const __bf16*& W, int& row0, int& k0, int& ld, int& nrows, int& kcols, int& R) {
    if (t < T1) {
      const int ch = t / (KC + 2), u = t - ch * (KC + 2);
      if (u < KC) { W = p.w1; row0 = ch * 256; k0 = u * 64; ld = p.ld1; nrows = p.n1; kcols = p.ld1; R = 256; }
      else { W = p.w2; row0 = 0; k0 = ch * 128 + (u - KC) * 64; ld = p.ld2; nrows = C; kcols = p.ld2; R = C; }
    } else { W = p.w3; row0 = 0; k0 = (t - T1) * 64; ld = p.ld3; nrows = C; kcols = p.ld3; R = C; }
  };
  auto load_tile = [&](int t, TileSet& S) {                  // request: global -> registers (zeros outside the matrix)
    const __bf16* W; int row0, k0, ld, nrows, kcols, R;
    tile_geom(t, W, row0, k0, ld, nrows, kcols, R);
    const bool kok = k0 + schunk * 8 < kcols;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int r = row0 + i * 64 + drow;
      S.v[i] = make_uint4(0u, 0u, 0u, 0u);
      if (i * 64 < R && kok && r < nrows) S.v[i] = *reinterpret_cast<const uint4*>(W + (int64_t)r * ld + k0 + schunk * 8);
    }
  };
  auto store_tile = [&](const TileSet& S, int t, int buf) {  // registers -> ring stage, in the swizzled image of the fragment reads
    const int R = (t < T1 && (t % (KC + 2)) < KC) ? 256 : C;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (i * 64 < R) *reinterpret_cast<uint4*>(ring + (buf * WROWS + i * 64 + drow) * 64 + (tid & 7) * 8) = S.v[i];
  };

  // ---- epilogue constants -> LDS (read there by the epilogues: no vector-memory wait inside the loop) --------------------------
  float* c_cs1 = stat + 128;                                 // [n1]
  float* c_b1 = c_cs1 + p.n1;                                // [n1]
  float* c_b2 = c_b1 + p.n1;                                 // [C]
  float* c_b3 = c_b2 + C;                                    // [C]
  for (int i = tid; i < p.n1; i += 512) { c_cs1[i] = p.cs1[i]; c_b1[i] = p.b1[i]; }
  for (int i = tid; i < C; i += 512) { c_b2[i] = p.b2[i]; c_b3[i] = p.b3[i]; }

  TileSet S0, S1;
  load_tile(0, S0);
  if (T > 1) load_tile(1, S1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the H tile's LDS-DMA and tile 0 / 1)
  store_tile(S0, 0, 0);
  __syncthreads();

  // ---- LayerNorm statistics of the bf16 rows (8 threads per row) -------------------------------------------------------------
  {
    const int m = tid >> 3, q = tid & 7;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(H + (kc * 64 + m) * 64 + q * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s += f; ss += f * f; }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) { s += __shfl_xor(s, off); ss += __shfl_xor(ss, off); }
    if (q == 0) {
      const float mean = s * (1.0f / C);
      float var = ss * (1.0f / C) - mean * mean;
      var = var < 0.f ? 0.f : var;
      stat[m * 2] = mean;
      stat[m * 2 + 1] = rsqrtf(var + p.eps);
    }
  }

  auto act_frag = [&](const __bf16* img, int kc, int i, int s) -> bf16x8 {      // activation operand: rows of the tile
    const int r = wm * 32 + i * 16 + frow;
    const int sw = (s * 4 + fq) ^ ((r >> 1) & 7);
    return *reinterpret_cast<const bf16x8*>(img + (kc * 64 + r) * 64 + sw * 8);
  };
  auto w_frag = [&](int buf, int r, int s) -> bf16x8 {                            // weight operand: rows of the ring tile
    const int sw = (s * 4 + fq) ^ ((r >> 1) & 7);
    return *reinterpret_cast<const bf16x8*>(ring + (buf * WROWS + r) * 64 + sw * 8);
  };
  auto img_ptr = [&](__bf16* img, int m, int c) -> __bf16* {                      // 4 consecutive columns c..c+3 of row m
    const int kc = c >> 6, k = c & 63;
    return img + (kc * 64 + m) * 64 + (((k >> 3) ^ ((m >> 1) & 7)) << 3) + (k & 7);
  };

  f32x4 acc1[2][4], acc3[2][NF3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NF3; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // iteration t: everyone is done with tile t-1 and sees tile t -> request tile t+2 into the set tile t came from -> multiply
  // tile t -> write tile t+1 (requested during iteration t-1) into the stage tile t-1 occupied
  auto iteration = [&](int t, TileSet& Sa, const TileSet& Sb) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < T) load_tile(t + 2, Sa);
    const int buf = t & 1;
    int ch = 0, u = 0;
    bool g1 = false, g2 = false;
    if (t < T1) { ch = t / (KC + 2); u = t - ch * (KC + 2); g1 = u < KC; g2 = !g1; }
    if (g1) {
      // ---- GEGLU projection: 64 x 256 += H[:, 64u..] x W1[chunk rows, 64u..] ------------------------------------------------
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[2], wf[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = act_frag(H, u, i, s);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = w_frag(buf, wn * 64 + j * 16 + frow, s);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc1[i][j], 0, 0, 0);
      }
      if (u == KC - 1) {
        // epilogue: folded LayerNorm, bias, h * gelu(g) -> F (bf16)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = wm * 32 + i * 16 + frow;
          const float mean = stat[m * 2], rstd = stat[m * 2 + 1];
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const int nh = ch * 256 + wn * 64 + t2 * 32 + fq * 4, ng = nh + 16;
            float4 csh = {0, 0, 0, 0}, csg = {0, 0, 0, 0}, bh = {0, 0, 0, 0}, bg = {0, 0, 0, 0};
            if (ng + 3 < p.n1) {
              csh = *reinterpret_cast<const float4*>(c_cs1 + nh); csg = *reinterpret_cast<const float4*>(c_cs1 + ng);
              bh = *reinterpret_cast<const float4*>(c_b1 + nh); bg = *reinterpret_cast<const float4*>(c_b1 + ng);
            }
            const float ch4[4] = {csh.x, csh.y, csh.z, csh.w}, cg4[4] = {csg.x, csg.y, csg.z, csg.w};
            const float bh4[4] = {bh.x, bh.y, bh.z, bh.w}, bg4[4] = {bg.x, bg.y, bg.z, bg.w};
            float f[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float hv = rstd * (acc1[i][2 * t2][r] - mean * ch4[r]) + bh4[r];
              const float gv = rstd * (acc1[i][2 * t2 + 1][r] - mean * cg4[r]) + bg4[r];
              f[r] = hv * gelu_erf_f(gv);
            }
            uint2 pk; pk.x = pack_bf16x2(f[0], f[1]); pk.y = pack_bf16x2(f[2], f[3]);
            *reinterpret_cast<uint2*>(img_ptr(F, m, wn * 32 + t2 * 16 + fq * 4)) = pk;
            acc1[i][2 * t2] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[i][2 * t2 + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    } else {
      // ---- ff.net[2] (g2) or proj_out: 64 x C += A[:, 64-wide K-step] x W[:, same] ----------------------------------------------
      const __bf16* A = g2 ? F : H;
      const int kc = g2 ? (u - KC) : (t - T1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[2], wf[NF3];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = act_frag(A, kc, i, s);
#pragma unroll
        for (int j = 0; j < NF3; ++j) wf[j] = w_frag(buf, wn * NC + j * 16 + frow, s);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NF3; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc3[i][j], 0, 0, 0);
      }
      if (g2 && t == T1 - 1) {
        // h3 = acc + b2 + h2, rounded to bf16, back into H (every element has exactly one owner; nobody reads H in this phase)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = wm * 32 + i * 16 + frow;
#pragma unroll
          for (int j = 0; j < NF3; ++j) {
            const int n = wn * NC + j * 16 + fq * 4;
            const float4 b = *reinterpret_cast<const float4*>(c_b2 + n);
            __bf16* hp = img_ptr(H, m, n);
            union { uint2 u; __bf16 e[4]; } old;
            old.u = *reinterpret_cast<const uint2*>(hp);
            uint2 pk;
            pk.x = pack_bf16x2(acc3[i][j][0] + b.x + (float)old.e[0], acc3[i][j][1] + b.y + (float)old.e[1]);
            pk.y = pack_bf16x2(acc3[i][j][2] + b.z + (float)old.e[2], acc3[i][j][3] + b.w + (float)old.e[3]);
            *reinterpret_cast<uint2*>(hp) = pk;
            acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    }
    if (t + 1 < T) store_tile(Sb, t + 1, (t + 1) & 1);
  };
  for (int t = 0; t < T; t += 2) {
    iteration(t, S0, S1);
    if (t + 1 < T) iteration(t + 1, S1, S0);
  }

  // ---- proj_out accumulators + bias -> fp32 staging image [64][C] in the idle ring ------------------------------------------------
  __syncthreads();
  float* stage = reinterpret_cast<float*>(ring);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = wm * 32 + i * 16 + frow;
#pragma unroll
    for (int j = 0; j < NF3; ++j) {
      const int n = wn * NC + j * 16 + fq * 4;
      const float4 b = *reinterpret_cast<const float4*>(c_b3 + n);
      float4 o; o.x = acc3[i][j][0] + b.x; o.y = acc3[i][j][1] + b.y; o.z = acc3[i][j][2] + b.z; o.w = acc3[i][j][3] + b.w;
      *reinterpret_cast<float4*>(stage + m * C + n) = o;
    }
  }
  __syncthreads();

  // ---- y = bf16(stage + x): 16-byte coalesced loads / stores; column statistics of the rounded values --------------------------
  constexpr int CPR = C / 8, RP = (512 / CPR) < 16 ? (512 / CPR) : 16;   // 16-byte chunks per row, rows handled in parallel
  float* red = reinterpret_cast<float*>(H);                // [RP][C][2] <= 128*C bytes: H and F (contiguous, both dead now)
  const int cc = tid % CPR, rr = tid / CPR;
  float cs[8], cq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cs[e] = 0.f; cq[e] = 0.f; }
  if (rr < RP) {
    for (int m = rr; m < 64; m += RP) {
      const float4 a0 = *reinterpret_cast<const float4*>(stage + m * C + cc * 8);
      const float4 a1 = *reinterpret_cast<const float4*>(stage + m * C + cc * 8 + 4);
      const uint4 xq = *reinterpret_cast<const uint4*>(p.x + (int64_t)(m0 + m) * p.ldx + cc * 8);
      float xv[8];
      unpack_bf16x8(xq, xv);
      float v[8] = {a0.x + xv[0], a0.y + xv[1], a0.z + xv[2], a0.w + xv[3], a1.x + xv[4], a1.y + xv[5], a1.z + xv[6], a1.w + xv[7]};
      const uint4 oq = pack_bf16x8(v);
      *reinterpret_cast<uint4*>(p.y + (int64_t)(m0 + m) * p.ldy + cc * 8) = oq;
      if (p.colstat) {
        float rv[8];
        unpack_bf16x8(oq, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { cs[e] += rv[e]; cq[e] += rv[e] * rv[e]; }
      }
    }
    if (p.colstat) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[(rr * C + cc * 8 + e) * 2] = cs[e]; red[(rr * C + cc * 8 + e) * 2 + 1] = cq[e]; }
    }
  }
  if (p.colstat) {
    __syncthreads();
    if (tid < C) {
      float a = 0.f, b = 0.f;
      for (int r = 0; r < RP; ++r) { a += red[(r * C + tid) * 2]; b += red[(r * C + tid) * 2 + 1]; }
      float* dst = p.colstat + ((int64_t)blockIdx.x * p.colstat_ld + tid) * 2;
      dst[0] = a; dst[1] = b;
    }
  }
}

template <int NF3>
int launch_ff(const FfK& k, hipStream_t s) {
  constexpr int C = 64 * NF3, WROWS = C > 256 ? C : 256;
  const size_t lds = (size_t)(NF3 * 64 * 64 + 2 * 64 * 64 + 2 * WROWS * 64) * 2 + 64 * 2 * 4 + ((size_t)2 * k.n1 + 2 * C) * 4;
  if (lds > 160 * 1024) {
    aptp_set_error("ff_tail: %zu bytes of LDS needed (n1 = %d too wide for C = %d)", lds, k.n1, C);
    return APTP_EINVAL;
  }
  static size_t reserved = 0;
  if (lds > reserved) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ff_tail_kernel<NF3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      aptp_set_error("ff_tail: cannot reserve %zu bytes of LDS", lds);
      return APTP_ELAUNCH;
    }
    reserved = lds;
  }
  hipLaunchKernelGGL(ff_tail_kernel<NF3>, dim3(k.M / 64), dim3(512), lds, s, k);
  return APTP_OK;
}

}  // namespace

#define ALIGN16(p) (((uintptr_t)(p) % 16) == 0)

extern "C" int aptp_ff_tail_supported(int M, int C, int n1, int ld1, int ld2, int ld3) {
  if (!(M > 0 && M % 64 == 0 && C % 64 == 0 && C >= 64 && C <= 320 && n1 > 0 && n1 % 32 == 0 && ld1 == C && ld3 == C && ld2 % 64 == 0 && ld2 * 2 >= n1))
    return 0;
  const int wrows = C > 256 ? C : 256;
  const long lds = (long)(C * 64 + 2 * 64 * 64 + 2 * wrows * 64) * 2 + 512 + (2L * n1 + 2 * C) * 4;
  return lds <= 160 * 1024;
}

extern "C" int aptp_ff_tail(const AptpFfTailParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->h && p->x && p->y && p->w1 && p->b1 && p->cs1 && p->w2 && p->b2 && p->w3 && p->b3, "ff_tail: null pointer");
  APTP_CHECK(aptp_ff_tail_supported(p->M, p->C, p->n1, p->ld1, p->ld2, p->ld3), "ff_tail: unsupported geometry (M=%d C=%d n1=%d ld1=%d ld2=%d ld3=%d)",
             p->M, p->C, p->n1, p->ld1, p->ld2, p->ld3);
  APTP_CHECK(p->ldh % 8 == 0 && p->ldx % 8 == 0 && p->ldy % 8 == 0 && ALIGN16(p->h) && ALIGN16(p->x) && ALIGN16(p->y) && ALIGN16(p->w1)
             && ALIGN16(p->w2) && ALIGN16(p->w3) && ALIGN16(p->b1) && ALIGN16(p->cs1) && ALIGN16(p->b2) && ALIGN16(p->b3), "ff_tail: alignment");
  APTP_CHECK(!p->colstat || p->colstat_ld >= p->C, "ff_tail: colstat_ld");
  FfK k;
  k.h = (const __bf16*)p->h; k.ldh = p->ldh; k.x = (const __bf16*)p->x; k.ldx = p->ldx; k.y = (__bf16*)p->y; k.ldy = p->ldy;
  k.w1 = (const __bf16*)p->w1; k.b1 = p->b1; k.cs1 = p->cs1; k.n1 = p->n1; k.ld1 = p->ld1;
  k.w2 = (const __bf16*)p->w2; k.b2 = p->b2; k.ld2 = p->ld2;
  k.w3 = (const __bf16*)p->w3; k.b3 = p->b3; k.ld3 = p->ld3;
  k.colstat = p->colstat; k.colstat_ld = p->colstat_ld;
  k.M = p->M; k.nchunks = (p->n1 + 255) / 256; k.eps = p->eps;
  int rc;
  switch (p->C / 64) {
    case 1: rc = launch_ff<1>(k, (hipStream_t)stream); break;
    case 2: rc = launch_ff<2>(k, (hipStream_t)stream); break;
    case 3: rc = launch_ff<3>(k, (hipStream_t)stream); break;
    case 4: rc = launch_ff<4>(k, (hipStream_t)stream); break;
    default: rc = launch_ff<5>(k, (hipStream_t)stream); break;
  }
  if (rc) return rc;
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}
