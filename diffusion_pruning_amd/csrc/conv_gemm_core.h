// Shared core of the implicit-GEMM kernels (conv_gemm.hip, conv_gemm_sk.hip): launch parameters, the XCD-aware
// workgroup -> tile order, the fused epilogues (accumulator-layout and coalesced), folded-LayerNorm row statistics and
// the in-kernel split-K combine.  Device code only; every function is inlined into the kernels of the including file.
#pragma once
#include "aptp_common.h"
// In-kernel stamps (timing experiments only, -DAPTP_STAMPS): per-wave cycle totals of the K-loop segments
// [0->1 DMA issue, 1->2 LDS reads + MFMAs, 2->3 waits + barrier], dumped by lane 0 of every wave into g_stamps.
#if defined(APTP_STAMPS) && defined(APTP_CG_MAIN)
__device__ unsigned long long g_stamps[4096 * 4];
// launch timeline of a wave (cycles since its first instruction): [0] prologue DMA issued, [1] first tile landed (first
// barrier passed), [2] K loop done, [3] epilogue done and its stores drained
__device__ unsigned long long g_phase[4096 * 8];   // + [4] tile decoded, [5] prefetch / statistics requested, [6] addresses ready
#define APTP_PHASE(i) do { st_phase[i] = __builtin_readcyclecounter() - st_entry; } while (0)
// epilogue timeline (absolute cycles): [0] entry, [1] stages free (barrier passed), [2] first fragment's per-column half in LDS,
// [4] all stores issued, [5] stores drained
__device__ unsigned long long g_epi[4096 * 8];
#define APTP_EPI(k) do { if ((threadIdx.x & 63) == 0) g_epi[((blockIdx.x * 8 + (threadIdx.x >> 6)) & 4095) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#define APTP_STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); \
    if ((i) > 0) st_acc[(i) - 1] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define APTP_STAMP(i) do { } while (0)
#define APTP_PHASE(i) do { } while (0)
#define APTP_EPI(k) do { } while (0)
#endif
#ifndef APTP_ABLATE
#define APTP_ABLATE 0   // timing experiments only (tools/ablate_conv.py): 1 no LDS-DMA in the loop, 2 no MFMA, 4 no ds_read, 8 no barrier,
                        // 32 no epilogue stores, 64 no residual / depth_in loads, 128 no LayerNorm row-statistics fetch, 256 no ln_colsum loads
#endif

namespace aptp_cg {


constexpr int BK = 64;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// Division by a launch constant as multiply-high + shift (the host computes the magic pair): the prologue / epilogue of a
// launch decode tile and pixel coordinates with ~10 runtime divisions per lane, 25-40 instructions each on this ISA, and
// a cycle-stamp timeline showed 1.6-2.2 us of address arithmetic before the first operand request of every launch.
// Exact for 0 <= n < 2^31:  p = 31 + ceil(log2 d), mul = ceil(2^p / d) in [2^31, 2^32), error mul*d - 2^p < d <= 2^(p-31).
struct FastDiv {
  unsigned mul, shr, one;      // one = ~0u for d == 1 (mul = 0, shr = 0: q = n)
  __device__ __forceinline__ int div(int n) const { return (int)((__umulhi((unsigned)n, mul) + ((unsigned)n & one)) >> shr); }
};

struct KParams {
  const __bf16* x; int64_t ldx;
  int B, Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad, ups;   // ups: right-shift applied to gather coordinates (0/1)
  int zins;                 // 1: zero-insertion upsample (odd coordinates read zero) instead of nearest
  int HinE, WinE;           // effective (upsampled) input extent
  const __bf16* w; int N; int ncc; int nK; int64_t Ktot;   // ncc = cin_pad/64, nK = taps*ncc + ncc2
  // optional second operand: one more K-segment after the filter taps, a 1x1 "tap" at the output pixel over x2's channels
  // (the resnet's conv_shortcut fused into conv2: K = 9*C_mid + C_in)
  const __bf16* x2; int64_t ldx2; int Cin2; int ncc2; int x2_bytes;
  const float* bias; const float* rowbias; int ld_rowbias;
  const float* colgate; int gate_group, gate_B;
  int act;
  const float* corr; int corr_B;
  const __bf16* residual; int64_t ldres;
  const float* depth; int depth_B; const __bf16* depth_in; int64_t lddin;
  void* y; int64_t ldy; int out_f32;
  int split_k; float* ws;
  int x_bytes, w_bytes;     // buffer-resource extents (bytes)
  int M, HW, Nout;          // Nout = logical output columns (N/2 for GEGLU)
  int64_t ws_ld;            // workspace row stride (floats)
  int order;                // workgroup -> tile order (decode_block): 0 legacy, 1 weight-major, 2 activation-major
  const char* pf_ptr; int64_t pf_bytes;   // operand of the NEXT launch to pull towards the Infinity Cache (prefetch_next)
  int pf_lines, pf_per;                   // lin_gemm_kernel: 64-byte lines of pf_ptr, lines per workgroup slice (host-computed)
  int* counters;            // in-kernel split-K: one arrival counter per output tile (zero on entry, left zero)
  float* cstat_out; int cstat_ld;   // per-(row block, channel) (sum, sumsq) of the stored outputs: GroupNorm statistics
  int epi16;                // 1: bf16 output (and residual / depth_in) rows are 16-byte aligned -> coalesced epilogue
  int io_f32;               // 1: fp32 parity instantiation (AptpConvGemmParams.io_f32): x, w, x2, residual, depth_in, y are fp32
  float* rstat_out; int rstat_slots;                       // per-row (sum, sumsq) partials of the stored outputs
  const float* ln_stats; int ln_slots; const float* ln_colsum; float ln_eps; float ln_invC;   // folded LayerNorm
  const float* gn_gamma; const float* gn_beta; int gn_groups, gn_C, gn_silu; float gn_eps;   // reduce launch applies a GroupNorm
  long long* ustat_out; int ustat_unit, ustat_units, ustat_nrep;   // per-(sample, channel unit) fixed-point statistics (atomics)
  FastDiv fd_hw, fd_wout;                    // / HW, / Wout (per lane)
  FastDiv fd_tm, fd_tn, fd_sk, fd_perm, fd_tmn;   // decode_block: / tiles_m, / tiles_n, / split_k, / (tiles_n * split_k), / (tiles_m * tiles_n)
};

// ---------------------------------------------------------------------------------------------------------------
// XCD-aware workgroup -> (M-tile, N-tile, K-slice) mapping.  Workgroups are dealt round-robin over the 8 XCDs
// (linear id % 8) and every XCD has its own 4 MiB L2, so the legacy order (N-tile fastest) makes each XCD stream the
// WHOLE weight matrix and the whole activation tensor: with 8 L2s that is up to 8x the operand bytes on the fabric
// (measured 10.7 GB beyond L2 per step against 2.9 GB algorithmic).  Here each XCD gets one contiguous range of a
// work order in which neighbours share an operand:
//   order 1 (weight-major, weights larger than the activation tensor: levels 16/8): q -> (weight slice = (N-tile,
//            K-slice), M-tile fastest): the weight matrix is PARTITIONED over the XCDs, the small activation tensor
//            is what gets replicated;
//   order 2 (activation-major, levels 64/32): q -> (M-tile, then N-tile / K-slice fastest): each XCD owns a contiguous
//            band of output rows (3x3 halos of neighbouring tiles hit its L2), the small weight matrix is replicated.
// The (xcd, j) -> q map is the bijective remap for any workgroup count (cdna_hip_programming.md, 256^2 template).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_block(const KParams& p, int tiles_m, int tiles_n, int& tm, int& tn, int& kz) {
  const int L = blockIdx.x;
  if (p.order == 0) {
    kz = p.fd_tmn.div(L);
    const int t = L - kz * (tiles_m * tiles_n);
    tm = p.fd_tn.div(t);
    tn = t - tm * tiles_n;
    return;
  }
  const int T = gridDim.x;
  const int xcd = L & 7, j = L >> 3;
  const int qq = T >> 3, r = T & 7;
  const int q = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + j;
  if (p.order == 1) {
    const int w = p.fd_tm.div(q);       // weight slice: K-slice fastest, so an XCD's slices of one N-tile are adjacent
    tm = q - w * tiles_m;
    tn = p.fd_sk.div(w);
    kz = w - tn * p.split_k;
  } else {
    const int per_m = tiles_n * p.split_k;
    tm = p.fd_perm.div(q);
    const int rest = q - tm * per_m;
    kz = p.fd_tn.div(rest);
    tn = rest - kz * tiles_n;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// epilogue for 4 consecutive (packed) columns of one output row.  Shared by the GEMM kernel and the split-K reducer.
// For GEGLU `h` are the value columns at packed col n, `g` the gate columns at packed col n+16, and the logical
// output column is (n/32)*16 + n%16.
// ---------------------------------------------------------------------------------------------------------------
// per-row context of the epilogue: sample index, border class (GroupNorm-beta correction) and, for a folded LayerNorm,
// the row's mean / rstd finished from the producer's per-tile (sum, sumsq) partials
struct RowCtx { int b, cls; float mean, rstd; };

// first half of the epilogue of 4 consecutive packed columns: everything that needs only per-column / per-sample vectors
// (folded LayerNorm, bias, time-embedding bias, width gate, activation, GroupNorm-beta correction) -> v[4]
// column vectors of one quad of packed columns (4 values each; GEGLU: also the gate half at n + 16)
struct ColVecs { float4 cs, cg, bb, bg; };

template <bool GEGLU>
__device__ __forceinline__ void load_colvecs(const KParams& p, int n, ColVecs& c) {
  c.cs = c.cg = c.bb = c.bg = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.ln_stats) {
#if APTP_ABLATE & 256
    c.cs = c.cg = make_float4(1.f, 1.f, 1.f, 1.f);
#else
    c.cs = *reinterpret_cast<const float4*>(p.ln_colsum + n);
    if (GEGLU) c.cg = *reinterpret_cast<const float4*>(p.ln_colsum + n + 16);
#endif
  }
  if (p.bias) {
    c.bb = *reinterpret_cast<const float4*>(p.bias + n);
    if (GEGLU) c.bg = *reinterpret_cast<const float4*>(p.bias + n + 16);
  }
}

// first half of the epilogue of 4 consecutive packed columns: everything that needs only per-column / per-sample vectors
// (folded LayerNorm, bias, time-embedding bias, width gate, activation, GroupNorm-beta correction) -> v[4]
template <bool GEGLU>
__device__ __forceinline__ void epilogue_pre(const KParams& p, const RowCtx& rc, int n, const ColVecs& cv, float h[4], float g[4], float v[4]) {
  const int b = rc.b, cls = rc.cls;
  if (p.ln_stats) {
    // y = LN(x) W^T with gamma folded into W:  rstd * (x W'^T - mean * colsum(W')) ; the beta term sits in `bias`
    h[0] = rc.rstd * (h[0] - rc.mean * cv.cs.x); h[1] = rc.rstd * (h[1] - rc.mean * cv.cs.y);
    h[2] = rc.rstd * (h[2] - rc.mean * cv.cs.z); h[3] = rc.rstd * (h[3] - rc.mean * cv.cs.w);
    if (GEGLU) {
      g[0] = rc.rstd * (g[0] - rc.mean * cv.cg.x); g[1] = rc.rstd * (g[1] - rc.mean * cv.cg.y);
      g[2] = rc.rstd * (g[2] - rc.mean * cv.cg.z); g[3] = rc.rstd * (g[3] - rc.mean * cv.cg.w);
    }
  }
  if (p.bias) {
    h[0] += cv.bb.x; h[1] += cv.bb.y; h[2] += cv.bb.z; h[3] += cv.bb.w;
    if (GEGLU) { g[0] += cv.bg.x; g[1] += cv.bg.y; g[2] += cv.bg.z; g[3] += cv.bg.w; }
  }
  if (p.rowbias) {
    const float4 rb = *reinterpret_cast<const float4*>(p.rowbias + (int64_t)b * p.ld_rowbias + n);
    h[0] += rb.x; h[1] += rb.y; h[2] += rb.z; h[3] += rb.w;
  }
  const int c = GEGLU ? ((n >> 5) * 16 + (n & 15)) : n;   // logical output column of element 0
  if (p.colgate) {
    const float* gr = p.colgate + (int64_t)(b % p.gate_B) * ((GEGLU ? p.Nout : p.N) / p.gate_group);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gm = gr[(c + r) / p.gate_group];
      h[r] *= gm;
      if (GEGLU) g[r] *= gm;
    }
  }
  if (GEGLU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = h[r] * gelu_erf_f(g[r]);
  } else if (p.act == APTP_ACT_SILU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = silu_f(h[r]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = h[r];
  }
  if (p.corr) {
    const float4 cc = *reinterpret_cast<const float4*>(p.corr + ((int64_t)(b % p.corr_B) * 9 + cls) * p.Nout + c);
    v[0] += cc.x; v[1] += cc.y; v[2] += cc.z; v[3] += cc.w;
  }
}

template <bool GEGLU>
__device__ __forceinline__ void epilogue_quad(const KParams& p, int m, const RowCtx& rc, int n, float h[4], float g[4], float st[2]) {
  const int b = rc.b;
  float v[4];
  ColVecs cv;
  load_colvecs<GEGLU>(p, n, cv);
  epilogue_pre<GEGLU>(p, rc, n, cv, h, g, v);
  const int c = GEGLU ? ((n >> 5) * 16 + (n & 15)) : n;   // logical output column of element 0
  if (p.residual && !(APTP_ABLATE & 64)) {
    if (p.io_f32) {           // (wave-uniform) fp32 parity instantiation: residual / depth_in / y are fp32 tensors
      const float4 rr = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.residual) + (int64_t)m * p.ldres + c);
      v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
    } else {
      const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + (int64_t)m * p.ldres + c);
      union { uint2 u; __bf16 e[4]; } ru; ru.u = rr;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += (float)ru.e[r];
    }
  }
  if (p.depth && !(APTP_ABLATE & 64)) {
    const float d = p.depth[b % p.depth_B];
    float din[4];
    if (p.io_f32) {
      const float4 rr = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.depth_in) + (int64_t)m * p.lddin + c);
      din[0] = rr.x; din[1] = rr.y; din[2] = rr.z; din[3] = rr.w;
    } else {
      const uint2 rr = *reinterpret_cast<const uint2*>(p.depth_in + (int64_t)m * p.lddin + c);
      union { uint2 u; __bf16 e[4]; } ru; ru.u = rr;
#pragma unroll
      for (int r = 0; r < 4; ++r) din[r] = (float)ru.e[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (1.0f - d) * din[r] + d * v[r];
  }
#if APTP_ABLATE & 32
  asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
  return;
#endif
  if (p.out_f32) {
    float4 o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (int64_t)m * p.ldy + c) = o;
    if (p.io_f32 && p.rstat_out) {          // fp32 storage: the statistics are those of the stored values themselves
#pragma unroll
      for (int r = 0; r < 4; ++r) { st[0] += v[r]; st[1] += v[r] * v[r]; }
    }
  } else {
    union { uint2 u; __bf16 e[4]; } o;
    o.u.x = pack_bf16x2(v[0], v[1]); o.u.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(p.y) + (int64_t)m * p.ldy + c) = o.u;
    if (p.rstat_out) {          // statistics of the values as stored (bf16-rounded): what the consumer will read
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float f = (float)o.e[r]; st[0] += f; st[1] += f * f; }
    }
  }
}

__device__ __forceinline__ void row_info(const KParams& p, int m, RowCtx& rc) {
  rc.b = p.fd_hw.div(m);
  rc.cls = 4;
  rc.mean = 0.f; rc.rstd = 1.f;
  if (p.corr) {
    const int rem = m - rc.b * p.HW;
    const int oy = p.fd_wout.div(rem), ox = rem - oy * p.Wout;
    const int rr = oy == 0 ? 0 : (oy == p.Hout - 1 ? 2 : 1);
    const int cc = ox == 0 ? 0 : (ox == p.Wout - 1 ? 2 : 1);
    rc.cls = rr * 3 + cc;
  }
}

__device__ __forceinline__ void ln_finish(const KParams& p, float a, float a2, RowCtx& rc) {
  const float mean = a * p.ln_invC;
  float var = a2 * p.ln_invC - mean * mean;
  var = var < 0.f ? 0.f : var;
  rc.mean = mean;
  rc.rstd = rsqrtf(var + p.ln_eps);
}

// Epilogue of one workgroup tile from the MFMA accumulators:
//   acc[i][j][r] = out[m = m0 + wm*WTM + i*16 + (lane&15)][n = n0 + wn*WTN + j*16 + (lane>>4)*4 + r]
// With rstat_out, each wave also emits its rows' (sum, sumsq) over the columns it owns into slot tn*WN + wn: the four
// lanes that share a row (lane>>4 = 0..3) fold their partial sums with two cross-lane adds.
// Folded LayerNorm: mean / rstd of this lane's MF rows from the producer's partials [slots][M] (sum, sumsq).
// Two halves, both in the kernel PROLOGUE: ln_rows_issue() at the very top requests the first 32 slots (the four lanes that
// share a row split the slot PAIRS, pair = fq, fq+4, ...; all MF rows per lane => up to 4*MF independent 16-byte loads), and
// ln_rows_finish(), called after the first operand tiles have been requested, folds them.  Its wait coincides with the
// wait for operand tile 0, so the statistics cost no extra memory round trip (in the epilogue, or requested after the
// operand tiles, they cost every workgroup one loaded L2 round trip: +3..6 us per launch on the GEGLU projections).
// More than 32 slots (a 1280-wide producer on 64-wide tiles) take further rounds inside ln_rows_finish.  Fixed fold
// order: deterministic.
template <int MF>
struct LnRaw { float4 v[4][MF]; int mr[MF]; };

template <int MF, int WTM>
__device__ __forceinline__ void ln_rows_issue(const KParams& p, int m0, int wm, int lane, LnRaw<MF>& raw, bool split_ok = false) {
  // (split-K: only the workgroup that combines the slices needs the rows, and asks for them then)
  if (!p.ln_stats || (p.split_k > 1 && !split_ok) || (APTP_ABLATE & 128)) return;
  const int frow = lane & 15, fq = lane >> 4;
  const float4* sp = reinterpret_cast<const float4*>(p.ln_stats);     // [slots/2][M]: two slots per 16-byte element
  const int npair = p.ln_slots >> 1;
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int m = m0 + wm * WTM + i * 16 + frow;
    raw.mr[i] = m < p.M ? m : p.M - 1;
  }
  // (these short-K launches are bound by the vector-memory instruction rate, so rounds with no live slot are skipped
  // by a wave-uniform branch: MF * ceil(slots/8) load instructions per wave, every lane a distinct (row, slot pair))
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u * 4 < npair) {
      const int pr = u * 4 + fq;
      const int64_t row0 = (int64_t)(pr < npair ? pr : 0) * p.M;
#pragma unroll
      for (int i = 0; i < MF; ++i) raw.v[u][i] = sp[row0 + raw.mr[i]];
    } else {
#pragma unroll
      for (int i = 0; i < MF; ++i) raw.v[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  asm volatile("" ::: "memory");      // keep the requests here, ahead of the address generation and the operand DMA
}

template <int MF>
__device__ __forceinline__ void ln_rows_finish(const KParams& p, int lane, const LnRaw<MF>& raw, float (&ln_mean)[MF], float (&ln_rstd)[MF],
                                               bool split_ok = false) {
  if (!p.ln_stats || (p.split_k > 1 && !split_ok) || (APTP_ABLATE & 128)) return;
  const int fq = lane >> 4;
  const float4* sp = reinterpret_cast<const float4*>(p.ln_stats);
  const int npair = p.ln_slots >> 1;
  float lna[MF], lna2[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) { lna[i] = 0.f; lna2[i] = 0.f; }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const bool ok = u * 4 + fq < npair;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      lna[i] += ok ? raw.v[u][i].x + raw.v[u][i].z : 0.f;
      lna2[i] += ok ? raw.v[u][i].y + raw.v[u][i].w : 0.f;
    }
  }
  for (int base = 16; base < npair; base += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (base + u * 4 < npair) {
        const int pr = base + u * 4 + fq;
        const bool ok = pr < npair;
        const int64_t row0 = (int64_t)(ok ? pr : 0) * p.M;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          const float4 t = sp[row0 + raw.mr[i]];
          lna[i] += ok ? t.x + t.z : 0.f; lna2[i] += ok ? t.y + t.w : 0.f;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    lna[i] += __shfl_xor(lna[i], 16); lna2[i] += __shfl_xor(lna2[i], 16);
    lna[i] += __shfl_xor(lna[i], 32); lna2[i] += __shfl_xor(lna2[i], 32);
    RowCtx rc;
    ln_finish(p, lna[i], lna2[i], rc);
    ln_mean[i] = rc.mean; ln_rstd[i] = rc.rstd;
  }
}

template <int MF, int NF, int WTM, int WTN, int WN>
__device__ __forceinline__ void tile_epilogue(const KParams& p, f32x4 (&acc)[MF][NF], int m0, int n0, int tn, int wm, int wn, int lane,
                                              const float (&ln_mean)[MF], const float (&ln_rstd)[MF]) {
  const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int m = m0 + wm * WTM + i * 16 + frow;
    if (m >= p.M) continue;
    RowCtx rc;
    row_info(p, m, rc);
    rc.mean = ln_mean[i]; rc.rstd = ln_rstd[i];
    float st[2] = {0.f, 0.f};
    if (p.act == APTP_ACT_GEGLU) {
      if constexpr (NF % 2 == 0) {
#pragma unroll
        for (int j = 0; j < NF; j += 2) {
          const int n = n0 + wn * WTN + j * 16 + fq * 4;
          if (n >= p.N) continue;
          float h[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          float g[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
          epilogue_quad<true>(p, m, rc, n, h, g, st);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int n = n0 + wn * WTN + j * 16 + fq * 4;
        if (n >= p.N) continue;
        float h[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        epilogue_quad<false>(p, m, rc, n, h, h, st);
      }
    }
    if (p.rstat_out) {
      // lanes l, l^16, l^32, l^48 hold the same row (and took the same branches above)
      st[0] += __shfl_xor(st[0], 16); st[1] += __shfl_xor(st[1], 16);
      st[0] += __shfl_xor(st[0], 32); st[1] += __shfl_xor(st[1], 32);
      if (fq == 0) {
        float2 o; o.x = st[0]; o.y = st[1];
        const int slot = tn * WN + wn;     // [slots/2][M][2 slots x (sum, sumsq)]: the consumer reads two slots per 16-byte load
        reinterpret_cast<float2*>(p.rstat_out)[((int64_t)(slot >> 1) * p.M + m) * 2 + (slot & 1)] = o;
      }
    }
  }
}

// Unit statistics (AptpConvGemmParams.ustat_out): a wave that has the per-channel (sum, sumsq) of its WTM x WL block in `tot`
// (lane t + 64 q <-> channel cw0 + t + 64 q) stages them in its LDS slice, then lane u folds the channels of the u-th unit its
// columns touch and adds the two sums to the (replica, sample, unit) slot as 64-bit fixed point.  Integer addition is
// associative: the totals are independent of arrival order, i.e. deterministic, unlike float atomics.
constexpr float USTAT_SUM_SCALE = 1048576.0f;     // 2^20
constexpr float USTAT_SQ_SCALE = 4096.0f;         // 2^12
template <int WL>
__device__ __forceinline__ void emit_unit_stats(const KParams& p, const float (&tot)[2][(WL + 63) / 64], float* buf, int lane, int cw0,
                                                int row0) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < (WL + 63) / 64; ++q) {
    const int t = lane + 64 * q;
    if (t < WL) { buf[t] = tot[0][q]; buf[WL + t] = tot[1][q]; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int cend = cw0 + WL < p.Nout ? cw0 + WL : p.Nout;
  if (cw0 >= cend || row0 >= p.M) return;
  const int unit = p.ustat_unit;
  const int u0 = cw0 / unit, u1 = (cend - 1) / unit;
  const int u = u0 + lane;
  if (u > u1) return;
  const int lo = (u * unit > cw0 ? u * unit : cw0) - cw0;
  const int hi = ((u + 1) * unit < cend ? (u + 1) * unit : cend) - cw0;
  float s = 0.f, s2 = 0.f;
  for (int c = lo; c < hi; ++c) { s += buf[c]; s2 += buf[WL + c]; }
  const int b = p.fd_hw.div(row0);
  const int rep = (int)blockIdx.x & (p.ustat_nrep - 1);
  unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.ustat_out) + (((int64_t)rep * p.B + b) * p.ustat_units + u) * 2;
  __hip_atomic_fetch_add(dst, (unsigned long long)__float2ll_rn(s * USTAT_SUM_SCALE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(dst + 1, (unsigned long long)__float2ll_rn(s2 * USTAT_SQ_SCALE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Coalesced epilogue (bf16 outputs whose rows are 16-byte aligned).  The MFMA accumulator layout gives a lane 4 columns
// of 16 different rows, so storing from it costs one 8-byte access per (row, quad): 16 x 32-byte pieces per wave
// instruction, and as many again for the residual / depth-gate operands.  The short-K launches of the masked U-Net
// (1x1 projections with K = 128..1280) issue more vector-memory instructions in such an epilogue than in their whole
// K-loop.  Here every wave transposes its 16 x WL fp32 row fragment through its own slice of the (now idle) staging LDS:
// the per-column half of the epilogue runs in accumulator layout, then a lane owns 8 consecutive columns of one row, so
// residual / depth_in are read and y is written with 16-byte accesses covering whole 128-byte lines, and the
// residual loads are issued before the transpose so their latency overlaps it.  No workgroup barrier is needed after
// the first one: a wave's LDS instructions execute in order.
template <bool GEGLU, int MF, int NF, int WTM, int WTN, int WN>
__device__ __forceinline__ void tile_epilogue_lds(const KParams& p, f32x4 (&acc)[MF][NF], int m0, int n0, int tn, int wm, int wn,
                                                  int lane, const float (&ln_mean)[MF], const float (&ln_rstd)[MF], float* buf) {
  constexpr int WL = GEGLU ? WTN / 2 : WTN;       // logical (stored) columns of this wave
  constexpr int PITCH = WL + 4;                   // floats; +4 keeps the 16-byte LDS writes and reads conflict-free
  constexpr int LPR = WL / 8;                     // lanes per row in the transposed layout
  constexpr int RPP = 64 / LPR;                   // rows per pass
  constexpr int NPASS = (16 + RPP - 1) / RPP;
  constexpr bool POW2 = (LPR & (LPR - 1)) == 0;
  const int frow = lane & 15, fq = lane >> 4;
  const int lrow = lane / LPR, lc8 = lane - lrow * LPR;
  const int c0 = (GEGLU ? ((n0 + wn * WTN) >> 1) : (n0 + wn * WTN)) + lc8 * 8;   // this lane's first logical column
  const bool lane_on = lrow < RPP && c0 < p.Nout;
  __bf16* const yb = reinterpret_cast<__bf16*>(p.y);
  float cs[8], cs2[8];            // per-channel (sum, sumsq) over this lane's rows (p.cstat_out)
#pragma unroll
  for (int e = 0; e < 8; ++e) { cs[e] = 0.f; cs2[e] = 0.f; }
  // (bias / LayerNorm column sums are loaded per (row fragment, quad): hoisting them above the row loop measured
  // neutral-to-slightly-slower -- 173.3-173.7 vs 173.9 steps/s on one box -- for NF x 8 more live registers)
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int mbase = m0 + wm * WTM + i * 16;
    // transposed domain: request the residual / depth-gate operands first
    u32x4 rres[NPASS], rdin[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int r = ps * RPP + lrow, m2 = mbase + r;
      const bool on = lane_on && r < 16 && m2 < p.M;
      rres[ps] = (u32x4){0u, 0u, 0u, 0u}; rdin[ps] = (u32x4){0u, 0u, 0u, 0u};
      if (on && p.residual && !(APTP_ABLATE & 64)) rres[ps] = *reinterpret_cast<const u32x4*>(p.residual + (int64_t)m2 * p.ldres + c0);
      if (on && p.depth && !(APTP_ABLATE & 64)) rdin[ps] = *reinterpret_cast<const u32x4*>(p.depth_in + (int64_t)m2 * p.lddin + c0);
    }
    // accumulator domain: per-column half of the epilogue, fp32 row fragment -> LDS
    {
      const int m = mbase + frow;
      RowCtx rc;
      row_info(p, m < p.M ? m : p.M - 1, rc);
      rc.mean = ln_mean[i]; rc.rstd = ln_rstd[i];
      if constexpr (GEGLU) {
        if constexpr (NF % 2 == 0) {
#pragma unroll
          for (int j = 0; j < NF; j += 2) {
            const int n = n0 + wn * WTN + j * 16 + fq * 4;
            float h[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            float g[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (n < p.N) {
              ColVecs cv;
              load_colvecs<true>(p, n, cv);
              epilogue_pre<true>(p, rc, n, cv, h, g, v);
            }
            *reinterpret_cast<float4*>(buf + frow * PITCH + (j / 2) * 16 + fq * 4) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int n = n0 + wn * WTN + j * 16 + fq * 4;
          float h[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          if (n < p.N) {
            ColVecs cv;
            load_colvecs<false>(p, n, cv);
            epilogue_pre<false>(p, rc, n, cv, h, h, v);
          }
          *reinterpret_cast<float4*>(buf + frow * PITCH + j * 16 + fq * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    // (cross-lane exchange through LDS inside one wave: tell the compiler, see the column-statistics staging below)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (i == 0) APTP_EPI(2);
    // transposed domain: + residual, depth lerp, round, 16-byte store, row statistics
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int r = ps * RPP + lrow, m2 = mbase + r;
      const bool on = lane_on && r < 16 && m2 < p.M;
      float v[8];
      {
        const int rr = r < 16 ? r : 15;
        const float4 a = *reinterpret_cast<const float4*>(buf + rr * PITCH + lc8 * 8);
        const float4 b4 = *reinterpret_cast<const float4*>(buf + rr * PITCH + lc8 * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
      }
      if (p.residual && !(APTP_ABLATE & 64)) {
        float f[8];
        union { u32x4 v; uint4 s; } cv; cv.v = rres[ps];
        unpack_bf16x8(cv.s, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += f[e];
      }
      if (p.depth && !(APTP_ABLATE & 64)) {
        const float d = p.depth[p.fd_hw.div(m2 < p.M ? m2 : p.M - 1) % p.depth_B];
        float f[8];
        union { u32x4 v; uint4 s; } cv; cv.v = rdin[ps];
        unpack_bf16x8(cv.s, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (1.0f - d) * f[e] + d * v[e];
      }
      const uint4 o = pack_bf16x8(v);
#if APTP_ABLATE & 32
      asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
#else
      if (on) *reinterpret_cast<uint4*>(yb + (int64_t)m2 * p.ldy + c0) = o;
#endif
      if constexpr (!GEGLU) {
        if (p.cstat_out && on) {   // statistics of the values as stored (bf16-rounded): what the GroupNorm will read
          float f[8];
          unpack_bf16x8(o, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { cs[e] += f[e]; cs2[e] += f[e] * f[e]; }
        }
      }
      if constexpr (POW2) {
        if (p.rstat_out) {       // statistics of the values as stored (bf16-rounded): what the consumer will read
          float f[8], s0 = 0.f, s1 = 0.f;
          unpack_bf16x8(o, f);
          if (c0 < p.Nout) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s0 += f[e]; s1 += f[e] * f[e]; }
          }
#pragma unroll
          for (int off = 1; off < LPR; off <<= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
          if (lrow < RPP && r < 16 && m2 < p.M && lc8 == 0) {   // (also for a wave past the last column: its slot reads 0)
            float2 q; q.x = s0; q.y = s1;
            const int slot = tn * WN + wn;
            reinterpret_cast<float2*>(p.rstat_out)[((int64_t)(slot >> 1) * p.M + m2) * 2 + (slot & 1)] = q;
          }
        }
      }
    }
    if (i == 0) APTP_EPI(3);
  }
  APTP_EPI(4);
  if constexpr (!GEGLU && WL >= 32) {
    if (p.cstat_out) {
      // GroupNorm statistics for the consumer of y: per channel, the sum / sum of squares over the WTM rows this wave
      // stored.  Lanes with the same 8-column chunk (lrow = 0..RPP-1) are folded through the wave's LDS slice, sums first,
      // then squares (RPP*WL <= 512 floats fit the 16 x (WL+4) slice); one (sum, sumsq) per channel goes to row block
      // m0/WTM + wm of cstat_out.  Fixed order: deterministic.
      static_assert(RPP * WL <= 16 * PITCH, "column-statistics staging");
      float tot[2][(WL + 63) / 64];
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        if (lrow < RPP) {
          float* dst = buf + lrow * WL + lc8 * 8;
          if (ph == 0) {
            *reinterpret_cast<float4*>(dst) = make_float4(cs[0], cs[1], cs[2], cs[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
          } else {
            *reinterpret_cast<float4*>(dst) = make_float4(cs2[0], cs2[1], cs2[2], cs2[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(cs2[4], cs2[5], cs2[6], cs2[7]);
          }
        }
        // Lanes exchange data through LDS here without a workgroup barrier (one wave, in-order LDS): the compiler must
        // still be told.  Without the fence it reasons per thread -- "a lane that skipped the store reads what it read in
        // the previous phase" -- and reuses the phase-0 loads for the lanes with lrow >= RPP (seen: sumsq == sum there).
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < (WL + 63) / 64; ++q) {
          const int t = lane + 64 * q;
          float a = 0.f;
          if (t < WL) {
#pragma unroll
            for (int r = 0; r < RPP; ++r) a += buf[r * WL + t];
          }
          tot[ph][q] = a;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      const int cw0 = n0 + wn * WTN;
      float2* dstg = reinterpret_cast<float2*>(p.cstat_out) + (int64_t)(m0 / WTM + wm) * p.cstat_ld;
#pragma unroll
      for (int q = 0; q < (WL + 63) / 64; ++q) {
        const int t = lane + 64 * q;
        if (t < WL && cw0 + t < p.Nout) dstg[cw0 + t] = make_float2(tot[0][q], tot[1][q]);
      }
      if (p.ustat_out) emit_unit_stats<WL>(p, tot, buf, lane, cw0, m0 + wm * WTM);
    }
  }
}

// In-kernel split-K reduction (p.counters != nullptr).  Every K-slice workgroup of an output tile stores its fp32
// accumulators to its slab in ACCUMULATOR order (float4 (i,j) of thread t at [(i*NF+j)*NT + t]: 1 KiB per wave
// instruction, written and read by the same thread index, no transposition), publishes it (write-through stores, drained)
// and draws a ticket from the tile's arrival counter; the workgroup that draws split_k-1 acquires, re-reads ALL slabs in
// slice order (its own included, so the sum does not depend on which slice arrived last: deterministic), resets the
// counter for the next launch and goes on to the ordinary epilogue.  Placement-independent: any distribution of a
// tile's slices over CUs / XCDs is correct (cdna_hip_programming.md, projection GEMM item 2).  Replaces the
// splitk_reduce_kernel launch (5.9-6.6 us + a kernel boundary each, 56-62 per forward).
template <int NT, int MF, int NF>
__device__ __forceinline__ bool splitk_combine(const KParams& p, f32x4 (&acc)[MF][NF], int tile, int kz, int tid, int* lds_word) {
  constexpr int PER = MF * NF;
  float4* const slab = reinterpret_cast<float4*>(p.ws) + (int64_t)tile * p.split_k * (PER * NT);
  float4* const mine = slab + (int64_t)kz * (PER * NT);
  // The slab is stored WRITE-THROUGH (sc1: 16-byte buffer stores with aux = 16), which publishes it without an
  // agent-scope release fence: a release writes back the whole XCD L2's dirty lines and costs every slice workgroup
  // 2-6 us on its tail (measured: the fence form ran 1-4 us per launch behind the two-launch form even at split_k = 2).
  {
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(mine, 0, PER * NT * 16, 0x00020000);
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), srsrc, ((i * NF + j) * NT + tid) * 16, 0, 16);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
  __syncthreads();
  if (tid == 0) *lds_word = __hip_atomic_fetch_add(p.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int ticket = *reinterpret_cast<volatile int*>(lds_word);
  if (ticket != p.split_k - 1) return false;
  if (tid == 0) {
    __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every slice has arrived
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.split_k; ++z) {
    const float4* src = slab + (int64_t)z * (PER * NT);
    float4 t[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) t[i][j] = src[(i * NF + j) * NT + tid];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        acc[i][j][0] += t[i][j].x; acc[i][j][1] += t[i][j].y; acc[i][j][2] += t[i][j].z; acc[i][j][3] += t[i][j].w;
      }
  }
  return true;
}

// Cold-weight prefetch (LDS-DMA kernels).  Every weight is read once per forward and the forward's working set exceeds
// the 256 MB Infinity Cache, so each launch starts on cold weights (4.7 ms of conv_gemm per forward
// against 4.0 ms with warm weights).  With pf_ptr set, every workgroup touches its share of the NEXT launch's weights at
// kernel entry, one dword per 64-byte line, XCD-aware (aptp_prefetch_slice).  Measured on the headline forward: +1.6 %
// (186.3 vs 183.4 steps/s) for weights up to 12 MB -- what fits the eight 4 MB L2s next to the running launch's own
// traffic; +0.5 % without the size cap; a slice that ignores the XCDs (lines land in the Infinity Cache and in random L2s)
// was neutral, and issuing the touches after the epilogue, where the wave's end waits for them, cost 4 %.
__device__ __forceinline__ void prefetch_next(const KParams& p, int tid, int nt, unsigned* scratch) {
  if (!p.pf_ptr) return;
  // LDS-DMA loads into a 256-byte scratch row: no VGPRs, nothing waits on them but the counted vmcnt of the main loop
  // (in-order retirement: they were issued before the first operand stage).
  // XCD-aware split: workgroup L runs on XCD L % 8 (round-robin dispatch, see decode_block) and touches lines of the
  // (L % 8)-th eighth of the buffer -- the rows of a [N][K] weight matrix that the weight-major order of the next launch
  // hands to the same XCD, so the lines land in the L2 that will be asked for them, not only in the Infinity Cache.
  const int xcd = blockIdx.x & 7;
  aptp_prefetch_slice(p.pf_ptr, p.pf_bytes, xcd, blockIdx.x >> 3, ((int)gridDim.x + 7 - xcd) >> 3, tid, nt, scratch);
}

// picks the epilogue form (wave-uniform): the coalesced one needs 16-byte aligned bf16 rows (p.epi16, set on the host)
template <int NW, int MF, int NF, int WTM, int WTN, int WN>
__device__ __forceinline__ void run_epilogue(const KParams& p, f32x4 (&acc)[MF][NF], int m0, int n0, int tn, int wm, int wn, int lane,
                                             int wave, const float (&ln_mean)[MF], const float (&ln_rstd)[MF], __bf16* smem) {
  constexpr int PITCH_MAX = WTN + 4;
  constexpr bool POW2 = ((WTN / 8) & (WTN / 8 - 1)) == 0;
  if (p.epi16 && (POW2 || !p.rstat_out)) {
    APTP_EPI(0);
    __syncthreads();                        // every wave is done reading the operand stages
    APTP_EPI(1);
    float* buf = reinterpret_cast<float*>(smem) + wave * 16 * PITCH_MAX;
    if (p.act == APTP_ACT_GEGLU) {
      if constexpr (NF % 2 == 0) tile_epilogue_lds<true, MF, NF, WTM, WTN, WN>(p, acc, m0, n0, tn, wm, wn, lane, ln_mean, ln_rstd, buf);
    } else {
      tile_epilogue_lds<false, MF, NF, WTM, WTN, WN>(p, acc, m0, n0, tn, wm, wn, lane, ln_mean, ln_rstd, buf);
    }
  } else {
    tile_epilogue<MF, NF, WTM, WTN, WN>(p, acc, m0, n0, tn, wm, wn, lane, ln_mean, ln_rstd);
  }
}

// host side of FastDiv
inline FastDiv make_fastdiv(int d) {
  FastDiv f;
  if (d <= 1) { f.mul = 0; f.shr = 0; f.one = ~0u; return f; }
  int l = 0;
  while ((1ll << l) < d) ++l;                       // ceil(log2 d)
  const int p = 31 + l;
  f.mul = (unsigned)(((1ull << p) + (unsigned)d - 1) / (unsigned)d);
  f.shr = (unsigned)(p - 32);
  f.one = 0;
  return f;
}

// conv_gemm_sk.hip: persistent stream-K macro-tile kernels (APTP_TILE_SK_*)
int aptp_sk_cus();                                              // CUs of the current device = workgroups of a launch
int aptp_launch_sk(const KParams& k, int tile, hipStream_t s);
// lin_gemm.hip: lean kernels for plain linear layers (1x1, stride 1, one K-slice, coalesced bf16 epilogue)
bool aptp_lin_eligible(const KParams& k, int tile);
int aptp_launch_lin(const KParams& k, int tile, hipStream_t s);

}  // namespace aptp_cg
