// Lean contraction kernel for the LINEAR layers of the masked U-Net (1x1 convolutions / nn.Linear: proj_in, to_q / to_k / to_v,
// to_out, ff.net[2], proj_out -- reference call sites pdm/models/unet/blocks.py:228-268 (F.linear inside the attention
// processor), :776-849 (transformer block), diffusers' Transformer2DModel proj_in / proj_out).  113 of the 174 contraction
// launches of the headline step are such layers with K = 128 .. 2560, and round 4's SQ counters (profiles/r4_tile_pmc.txt)
// and launch timeline (profiles/r4_phase_timeline_before.txt) show what they cost in the general implicit-GEMM kernel: a wave
// of a K = 320 projection executes 750 VALU + 480 SALU instructions around 40 MFMAs, spends 1.5 us before its first operand
// request (tap / pixel decode, next-launch prefetch with an integer division, LayerNorm statistics) and 2.7-3.0 us in an
// epilogue that handles one 16-row fragment at a time, each with its own dependent round trip for bias and residual.
//
// Same tiles, LDS image (XOR-swizzled 128-byte rows, LDS-DMA with the swizzle applied to the source chunk), MFMA mapping
// (weights = A operand, activations = B operand) and ring schedule as conv_gemm_dma_kernel, so the statistics slots, the
// tuning table's tile ids and the consumers' expectations carry over.  What differs:
//   * no filter geometry at all: a row of the activation tile IS a row of x (pointer + 128 B per K-tile);
//   * every global read of the workgroup is requested up front: operand tiles first, then -- before anything waits -- the
//     residual rows, bias, LayerNorm column sums and row statistics in the layout the epilogue will use them in;
//   * the epilogue runs in the TRANSPOSED domain only: all MF x NF accumulator fragments of a wave go through its LDS slice
//     once (raw fp32), then a lane owns 8 consecutive columns of one row per pass and applies folded LayerNorm, bias,
//     activation and residual there, stores whole 128-byte lines and emits the row / column statistics -- one LDS round
//     trip and no global-memory wait inside the epilogue;
//   * the next-launch weight prefetch takes its slice bounds from the host (no division on the device).
// GEGLU projections (ff.net[0]) take the same path with value and gate columns read side by side in the transposed domain.
// Anything else (split-K, gates, depth lerp, fp32 I/O, second operand) stays with conv_gemm_dma_kernel.
#include "conv_gemm_core.h"

namespace {
using namespace aptp_cg;

__device__ __attribute__((aligned(256))) unsigned char g_lin_zero_page[8192];

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

template <int BM, int BN, int WM, int WN, int STAGES, int KU, bool GEGLU = false>
__global__ __launch_bounds__(WM * WN * 64) void lin_gemm_kernel(const KParams p) {
  constexpr int NW = WM * WN, NT = NW * 64, RPP = NT / 8;      // RPP: tile rows one LDS-DMA pass of the workgroup covers
  constexpr int WTM = BM / WM, WTN = BN / WN, MF = WTM / 16, NF = WTN / 16;
  constexpr int A_PASS = BM / RPP, B_PASS = BN / RPP, NLD = A_PASS + B_PASS;
  constexpr int D = STAGES - KU;                                // K-tiles in flight ahead of the one being multiplied
  static_assert(BM % RPP == 0 && BN % RPP == 0 && WTM % 16 == 0 && WTN % 32 == 0, "tile shape");
  static_assert(STAGES % KU == 0 && D >= KU && NLD * (D - KU) <= 63, "ring");
  // transposed epilogue domain: LPR lanes share a row (8 columns each), RPW rows per pass, NPASS passes over the wave's WTM rows
  // (GEGLU: the packed weight rows interleave value / gate columns in blocks of 16, so a wave's WTN packed columns are WTN / 2
  //  logical ones; a lane reads its 8 value columns and the 8 gate columns 16 packed columns further on)
  constexpr int WL = GEGLU ? WTN / 2 : WTN;                     // logical (stored) columns of a wave
  constexpr int LPR = WL / 8, RPW = (64 / LPR) < WTM ? (64 / LPR) : WTM, NPASS = WTM / RPW, PITCH = WTN + 4;
  static_assert((LPR & (LPR - 1)) == 0 && WTM % RPW == 0, "transposed layout");
  static_assert(NW * WTM * PITCH * 4 <= STAGES * (BM + BN) * BK * 2, "epilogue transpose buffer");

  __shared__ __attribute__((aligned(16))) __bf16 smem[STAGES * (BM + BN) * BK];
  __shared__ unsigned pf_scratch[64];
  __bf16* As = smem;
  __bf16* Bs = smem + STAGES * BM * BK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // ---- tile decode (split_k == 1): the XCD-aware orders of decode_block ----
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  int tm, tn;
  {
    const int L = blockIdx.x;
    if (p.order == 0) {
      tm = p.fd_tn.div(L); tn = L - tm * tiles_n;
    } else {
      const int T = gridDim.x, xcd = L & 7, j = L >> 3, qq = T >> 3, r = T & 7;
      const int q = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + j;
      if (p.order == 1) { tn = p.fd_tm.div(q); tm = q - tn * tiles_m; }
      else { tm = p.fd_tn.div(q); tn = q - tm * tiles_n; }
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = p.ncc;                                         // K-tiles (taps == 1, no second operand, no K split)

  // ---- operand row pointers ----
  const int rowbase = tid >> 3;
  const int schunk = (tid & 7) ^ ((rowbase >> 1) & 7);          // source chunk that lands in this lane's LDS slot
  const char* const zpage = reinterpret_cast<const char*>(g_lin_zero_page) + schunk * 16;
  const bool tail_bad = ((nk - 1) * BK + schunk * 8) >= p.Cin;  // this lane's chunk of the last K-tile is channel padding
  const char* a_ptr[A_PASS];
  const char* b_ptr[B_PASS];
#pragma unroll
  for (int i = 0; i < A_PASS; ++i) {
    const int m = m0 + rowbase + RPP * i;
    const char* va = reinterpret_cast<const char*>(p.x) + ((int64_t)m * p.ldx + schunk * 8) * 2;
    a_ptr[i] = m < p.M ? va : zpage;                            // rows past M multiply zeros (never stored)
  }
#pragma unroll
  for (int i = 0; i < B_PASS; ++i) {
    int n = n0 + rowbase + RPP * i;
    n = n < p.N ? n : p.N - 1;                                  // columns past N accumulate garbage that is never stored
    b_ptr[i] = reinterpret_cast<const char*>(p.w) + ((int64_t)n * p.Ktot + schunk * 8) * 2;
  }
  auto issue_tile = [&](int t, int stage) {
    const bool pad = (t == nk - 1) && tail_bad;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const char* src = pad ? zpage : a_ptr[i];
      __bf16* dst = As + (stage * BM + wave * 8 + RPP * i) * BK;     // wave-uniform; lane l lands at dst + l * 16 B
      __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)dst, 16, 0, 0);
      a_ptr[i] += BK * 2;
    }
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      __bf16* dst = Bs + (stage * BN + wave * 8 + RPP * i) * BK;
      __builtin_amdgcn_global_load_lds((gbl_ptr)b_ptr[i], (lds_ptr)dst, 16, 0, 0);
      b_ptr[i] += BK * 2;
    }
  };
  // next launch's weights towards this XCD's L2 (slice bounds from the host, launch_lin).  Requested BEFORE the operand tiles: an
  // LDS-DMA into pf_scratch that is still pending when the K loop starts makes the compiler's wait-count pass put a vmcnt(0) in
  // front of the loop's fragment reads (it cannot tell the scratch row from the operand stages), which serialises the ring.
  if (p.pf_ptr) {
    const int xcd = blockIdx.x & 7;
    const int x0 = (int)(((int64_t)p.pf_lines * xcd) >> 3), x1 = (int)(((int64_t)p.pf_lines * (xcd + 1)) >> 3);
    const int l0 = x0 + (int)(blockIdx.x >> 3) * p.pf_per;
    const int l1 = l0 + p.pf_per < x1 ? l0 + p.pf_per : x1;
    for (int l = l0 + tid; l < l1; l += NT)
      __builtin_amdgcn_global_load_lds((gbl_ptr)(p.pf_ptr + (int64_t)l * 64), (lds_ptr)pf_scratch, 4, 0, 0);
  }

  const int pre = nk < D ? nk : D;
#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < pre) issue_tile(t, t);

  // ---- everything the epilogue will read, requested now (transposed domain: lane -> row lrow of a pass, columns c0 .. c0+7) ----
  const int lrow = lane / LPR, lc8 = lane - lrow * LPR;
  const int c0 = (GEGLU ? ((n0 + wn * WTN) >> 1) : (n0 + wn * WTN)) + lc8 * 8;          // first logical column of this lane
  const bool col_on = c0 < p.Nout && lrow < RPW;
  const int c0c = c0 < p.Nout ? c0 : p.Nout - 8;
  // packed column of this lane's 8 value columns (GEGLU: block of 32 = 16 value + 16 gate columns)
  const int pc = GEGLU ? ((c0c >> 4) * 32 + (c0c & 15)) : c0c;
  const int bofs = GEGLU ? ((lc8 >> 1) * 32 + (lc8 & 1) * 8) : lc8 * 8;                 // same, relative to the wave's LDS columns
  int mrow[NPASS];
  u32x4 rres[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int m2 = m0 + wm * WTM + ps * RPW + lrow;
    mrow[ps] = m2 < p.M ? m2 : p.M - 1;
    rres[ps] = (u32x4){0u, 0u, 0u, 0u};
  }
  if (p.residual) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) rres[ps] = *reinterpret_cast<const u32x4*>(p.residual + (int64_t)mrow[ps] * p.ldres + c0c);
  }
  float4 bia[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  float4 csm[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  float4 biag[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  float4 csmg[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if (p.bias) {
    bia[0] = *reinterpret_cast<const float4*>(p.bias + pc);
    bia[1] = *reinterpret_cast<const float4*>(p.bias + pc + 4);
    if constexpr (GEGLU) {
      biag[0] = *reinterpret_cast<const float4*>(p.bias + pc + 16);
      biag[1] = *reinterpret_cast<const float4*>(p.bias + pc + 20);
    }
  }
  float ln_mean[NPASS], ln_rstd[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) { ln_mean[ps] = 0.f; ln_rstd[ps] = 1.f; }
  if (p.ln_stats) {
    csm[0] = *reinterpret_cast<const float4*>(p.ln_colsum + pc);
    csm[1] = *reinterpret_cast<const float4*>(p.ln_colsum + pc + 4);
    if constexpr (GEGLU) {
      csmg[0] = *reinterpret_cast<const float4*>(p.ln_colsum + pc + 16);
      csmg[1] = *reinterpret_cast<const float4*>(p.ln_colsum + pc + 20);
    }
    // producer partials [npair][M] x (sum, sumsq, sum, sumsq): the LPR lanes of a row take pairs lc8, lc8 + LPR, ...
    const float4* sp = reinterpret_cast<const float4*>(p.ln_stats);
    const int npair = p.ln_slots >> 1;
    float a[NPASS], a2[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) { a[ps] = 0.f; a2[ps] = 0.f; }
    constexpr int UNR = 3;                    // rounds requested together (wave-uniform guards skip the empty ones)
    for (int base = 0; base < npair; base += UNR * LPR) {
      float4 t[UNR][NPASS];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (base + u * LPR < npair) {
          const int pr = base + u * LPR + lc8;
          const int64_t row0 = (int64_t)(pr < npair ? pr : 0) * p.M;
#pragma unroll
          for (int ps = 0; ps < NPASS; ++ps) t[u][ps] = sp[row0 + mrow[ps]];
        } else {
#pragma unroll
          for (int ps = 0; ps < NPASS; ++ps) t[u][ps] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const bool ok = base + u * LPR + lc8 < npair;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          a[ps] += ok ? t[u][ps].x + t[u][ps].z : 0.f;
          a2[ps] += ok ? t[u][ps].y + t[u][ps].w : 0.f;
        }
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
      for (int off = 1; off < LPR; off <<= 1) { a[ps] += __shfl_xor(a[ps], off); a2[ps] += __shfl_xor(a2[ps], off); }
      const float mean = a[ps] * p.ln_invC;
      float var = a2[ps] * p.ln_invC - mean * mean;
      var = var < 0.f ? 0.f : var;
      ln_mean[ps] = mean;
      ln_rstd[ps] = rsqrtf(var + p.ln_eps);
    }
  }
  f32x4 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  auto compute = [&](int stage) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[MF], wf[NF];
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int r = wm * WTM + i * 16 + frow;
        af[i] = *reinterpret_cast<const bf16x8*>(As + (stage * BM + r) * BK + ((s * 4 + fq) ^ ((r >> 1) & 7)) * 8);
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int r = wn * WTN + j * 16 + frow;
        wf[j] = *reinterpret_cast<const bf16x8*>(Bs + (stage * BN + r) * BK + ((s * 4 + fq) ^ ((r >> 1) & 7)) * 8);
      }
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  // ---- K loop: ring of STAGES stages, KU tiles per barrier, D tiles in flight ahead (counted vmcnt, raw s_barrier) ----
  // Loads issued after the prologue tiles (residual, statistics, prefetch) only make a counted wait stricter for the older
  // operand tiles, never weaker: vmcnt(X) completes all but the youngest X requests.
  if (pre == D && D > KU) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD * (D - KU)) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    int cur = 0, nxt = D;
    const int nfull = nk / KU;                  // iterations that multiply KU whole tiles (KU == 2: an odd last tile follows the loop,
    for (int it = 0; it < nfull; ++it) {        //  so the loop body has no conditional MFMA block -- the accumulators stay put)
      const int kt = it * KU;
#pragma unroll
      for (int u = 0; u < KU; ++u)
        if (kt + D + u < nk) issue_tile(kt + D + u, nxt + u);
      compute(cur);
      if constexpr (KU == 2) compute(cur + 1);
      asm volatile("" ::: "memory");
      if (kt + D + KU <= nk && D > KU) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD * (D - KU)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tail: everything still in flight is needed soon
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wave's fragment reads are done
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      cur = cur + KU == STAGES ? 0 : cur + KU;
      nxt = nxt + KU == STAGES ? 0 : nxt + KU;
    }
    if constexpr (KU == 2) {
      if (nk & 1) {                             // (wave-uniform) the odd last tile: landed with the last wait above (vmcnt(0))
        compute(cur);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
  }

  // ---- epilogue (the last barrier above: every wave is done with the stages, no LDS-DMA in flight) ----
  float* const buf = reinterpret_cast<float*>(smem) + wave * (WTM * PITCH);
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
      *reinterpret_cast<f32x4*>(buf + (i * 16 + frow) * PITCH + j * 16 + fq * 4) = acc[i][j];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // cross-lane exchange through LDS inside one wave (in-order LDS)
  __builtin_amdgcn_wave_barrier();
  __bf16* const yb = reinterpret_cast<__bf16*>(p.y);
  float cs[8], cs2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cs[e] = 0.f; cs2[e] = 0.f; }
  const int slot = tn * WN + wn;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int r = (ps * RPW + lrow) < WTM ? (ps * RPW + lrow) : 0, m2 = m0 + wm * WTM + r;
    const bool on = col_on && m2 < p.M;
    float v[8], gt[8];
    {
      const float4 a = *reinterpret_cast<const float4*>(buf + r * PITCH + bofs);
      const float4 b = *reinterpret_cast<const float4*>(buf + r * PITCH + bofs + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    if constexpr (GEGLU) {
      const float4 a = *reinterpret_cast<const float4*>(buf + r * PITCH + bofs + 16);
      const float4 b = *reinterpret_cast<const float4*>(buf + r * PITCH + bofs + 20);
      gt[0] = a.x; gt[1] = a.y; gt[2] = a.z; gt[3] = a.w; gt[4] = b.x; gt[5] = b.y; gt[6] = b.z; gt[7] = b.w;
    }
    const float cb[8] = {bia[0].x, bia[0].y, bia[0].z, bia[0].w, bia[1].x, bia[1].y, bia[1].z, bia[1].w};
    if (p.ln_stats) {
      // y = LN(x) W^T with gamma folded into W: rstd * (x W'^T - mean * colsum(W')); the beta term sits in the bias
      const float cc[8] = {csm[0].x, csm[0].y, csm[0].z, csm[0].w, csm[1].x, csm[1].y, csm[1].z, csm[1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ln_rstd[ps] * (v[e] - ln_mean[ps] * cc[e]);
      if constexpr (GEGLU) {
        const float cg[8] = {csmg[0].x, csmg[0].y, csmg[0].z, csmg[0].w, csmg[1].x, csmg[1].y, csmg[1].z, csmg[1].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) gt[e] = ln_rstd[ps] * (gt[e] - ln_mean[ps] * cg[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += cb[e];
    if constexpr (GEGLU) {
      // ff.net[0] (GEGLU, blocks.py:776-849 / diffusers GEGLU): value * gelu(gate), both halves with their own bias
      const float cbg[8] = {biag[0].x, biag[0].y, biag[0].z, biag[0].w, biag[1].x, biag[1].y, biag[1].z, biag[1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * gelu_erf_f(gt[e] + cbg[e]);
    } else
    if (p.act == APTP_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    }
    if (p.residual) {
      float f[8];
      union { u32x4 v; uint4 s; } cv; cv.v = rres[ps];
      unpack_bf16x8(cv.s, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += f[e];
    }
    const uint4 o = pack_bf16x8(v);
    if (on) *reinterpret_cast<uint4*>(yb + (int64_t)m2 * p.ldy + c0) = o;
    if (p.cstat_out || p.rstat_out) {
      float f[8];                      // statistics of the values as stored (bf16-rounded): what the consumer will read
      unpack_bf16x8(o, f);
      if (p.cstat_out && on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { cs[e] += f[e]; cs2[e] += f[e] * f[e]; }
      }
      if (p.rstat_out) {
        float s0 = 0.f, s1 = 0.f;
        if (col_on) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { s0 += f[e]; s1 += f[e] * f[e]; }
        }
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
        if (m2 < p.M && lc8 == 0) {      // (also for a wave past the last column: its slot reads 0)
          float2 q; q.x = s0; q.y = s1;
          reinterpret_cast<float2*>(p.rstat_out)[((int64_t)(slot >> 1) * p.M + m2) * 2 + (slot & 1)] = q;
        }
      }
    }
  }
  if (!GEGLU && p.cstat_out) {
    // GroupNorm statistics for the consumer of y: per channel, (sum, sumsq) over the WTM rows this wave stored; the RPW lanes
    // that hold the same 8-column chunk are folded through the wave's LDS slice (fixed order: deterministic)
    static_assert(RPW * WTN <= WTM * PITCH, "column-statistics staging");
    float tot[2][(WTN + 63) / 64];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float* dst = buf + lrow * WTN + lc8 * 8;
      if (ph == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(cs2[0], cs2[1], cs2[2], cs2[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(cs2[4], cs2[5], cs2[6], cs2[7]);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < (WTN + 63) / 64; ++q) {
        const int t = lane + 64 * q;
        float a = 0.f;
        if (t < WTN) {
#pragma unroll
          for (int r = 0; r < RPW; ++r) a += buf[r * WTN + t];
        }
        tot[ph][q] = a;
      }
    }
    const int cw0 = n0 + wn * WTN;
    float2* dstg = reinterpret_cast<float2*>(p.cstat_out) + (int64_t)(m0 / WTM + wm) * p.cstat_ld;
#pragma unroll
    for (int q = 0; q < (WTN + 63) / 64; ++q) {
      const int t = lane + 64 * q;
      if (t < WTN && cw0 + t < p.Nout) dstg[cw0 + t] = make_float2(tot[0][q], tot[1][q]);
    }
    if constexpr (!GEGLU) {
      if (p.ustat_out) emit_unit_stats<WTN>(p, tot, buf, lane, cw0, m0 + wm * WTM);
    }
  }
}

template <int BM, int BN, int WM, int WN, int STAGES, int KU>
void launch_lin(KParams k, hipStream_t s) {
  const bool geglu = k.act == APTP_ACT_GEGLU;
  const int tiles = ((k.M + BM - 1) / BM) * ((k.N + BN - 1) / BN);
  // next-launch prefetch: slice bounds on the host (lines of the xcd-th eighth per workgroup of that XCD; aptp_prefetch_slice)
  k.pf_lines = 0; k.pf_per = 0;
  if (k.pf_ptr) {
    const int nper = (tiles + 7) >> 3;                // workgroups per XCD under round-robin dispatch (upper bound)
    k.pf_lines = (int)(k.pf_bytes >> 6);
    const int per_xcd = (k.pf_lines + 7) >> 3;
    k.pf_per = (per_xcd + nper - 1) / nper;
    if (k.pf_per < 1) k.pf_per = 1;
  }
  if (geglu) hipLaunchKernelGGL((lin_gemm_kernel<BM, BN, WM, WN, STAGES, KU, true>), dim3(tiles), dim3(WM * WN * 64), 0, s, k);
  else hipLaunchKernelGGL((lin_gemm_kernel<BM, BN, WM, WN, STAGES, KU, false>), dim3(tiles), dim3(WM * WN * 64), 0, s, k);
}

}  // namespace

namespace aptp_cg {

// Which launches the lean kernel takes: plain linear layers with the coalesced bf16 epilogue.  Everything else stays with
// conv_gemm_dma_kernel (returns false).
bool aptp_lin_eligible(const KParams& k, int tile) {
  if (k.KH != 1 || k.KW != 1 || k.stride != 1 || k.pad != 0 || k.ups != 0 || k.zins || k.x2 || k.ncc2) return false;
  if (k.split_k != 1 || k.io_f32 || k.out_f32 || !k.epi16 || k.gn_gamma) return false;
  if (k.colgate || k.corr || k.rowbias || k.depth) return false;
  if (k.act == APTP_ACT_GEGLU && (k.residual || k.rstat_out || k.cstat_out || k.N % 32 != 0)) return false;
  if (k.Nout < 8 || k.ncc < 1) return false;
  switch (tile) {
    case APTP_TILE_DMA_64x64: case APTP_TILE_DMA3_64x64: case APTP_TILE_DMA4_64x64: case APTP_TILE_DMA6_64x64: case APTP_TILE_DMA8S_64x64:
    case APTP_TILE_KU2S4_64x64: case APTP_TILE_KU2S6_64x64:
    case APTP_TILE_DMA_64x128: case APTP_TILE_DMA3_64x128: case APTP_TILE_DMA4_64x128: case APTP_TILE_DMA6_64x128:
    case APTP_TILE_KU2S4_64x128: case APTP_TILE_KU2S6_64x128:
    case APTP_TILE_DMA_128x64: case APTP_TILE_DMA3_128x64: case APTP_TILE_DMA4_128x64: case APTP_TILE_DMA6_128x64: case APTP_TILE_KU2S4_128x64:
      return true;
    default: return false;
  }
}

int aptp_launch_lin(const KParams& k, int tile, hipStream_t s) {
  switch (tile) {
    case APTP_TILE_DMA_64x64: launch_lin<64, 64, 2, 2, 2, 1>(k, s); break;
    case APTP_TILE_DMA3_64x64: launch_lin<64, 64, 2, 2, 3, 1>(k, s); break;
    case APTP_TILE_DMA4_64x64: case APTP_TILE_DMA6_64x64: case APTP_TILE_DMA8S_64x64: launch_lin<64, 64, 2, 2, 4, 1>(k, s); break;
    case APTP_TILE_KU2S4_64x64: case APTP_TILE_KU2S6_64x64: launch_lin<64, 64, 2, 2, 4, 2>(k, s); break;
    case APTP_TILE_DMA_64x128: launch_lin<64, 128, 2, 2, 2, 1>(k, s); break;
    case APTP_TILE_DMA3_64x128: launch_lin<64, 128, 2, 2, 3, 1>(k, s); break;
    case APTP_TILE_DMA4_64x128: case APTP_TILE_DMA6_64x128: launch_lin<64, 128, 2, 2, 4, 1>(k, s); break;
    case APTP_TILE_KU2S4_64x128: case APTP_TILE_KU2S6_64x128: launch_lin<64, 128, 2, 2, 4, 2>(k, s); break;
    case APTP_TILE_DMA_128x64: launch_lin<128, 64, 2, 2, 2, 1>(k, s); break;
    case APTP_TILE_DMA3_128x64: launch_lin<128, 64, 2, 2, 3, 1>(k, s); break;
    case APTP_TILE_DMA4_128x64: case APTP_TILE_DMA6_128x64: launch_lin<128, 64, 2, 2, 4, 1>(k, s); break;
    case APTP_TILE_KU2S4_128x64: launch_lin<128, 64, 2, 2, 4, 2>(k, s); break;
    default: aptp_set_error("lin_gemm: tile %d has no lean instantiation", tile); return APTP_EINVAL;
  }
  return APTP_OK;
}

}  // namespace aptp_cg
