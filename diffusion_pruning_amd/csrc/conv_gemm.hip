// Implicit-GEMM convolution / linear on MFMA (gfx950), channels-last bf16, fp32 accumulate.
//
// One kernel family serves every contraction of the masked U-Net: 3x3 / 1x1 convs (stride 1/2, nearest-x2
// upsample folded into the gather), and all linear layers (1x1 "conv" over tokens).  GEMM view:
//   M = B*Hout*Wout (pixels / tokens),  N = output channels,  K = taps * Cin_pad.
// Tile: BM x BN x 64 per 256-thread workgroup (4 waves), double-buffered LDS with an XOR-swizzled
// [row][64] bf16 image (conflict-free ds_read_b128 for the 16x16x32 MFMA fragments), register-staged
// global->LDS prefetch one K-step ahead (loads issued before the MFMA phase, written after it).
// The MFMA is issued with the WEIGHT fragment as the A operand and the activation fragment as the B operand, so
// each lane ends up with 4 consecutive output channels of one pixel => 8-byte epilogue loads/stores.
// The architecture-code gate, time-embedding bias, GEGLU, GroupNorm-beta correction, residual and depth lerp are
// fused into the epilogue (see include/aptp_hip.h for the reference call sites each one replaces).
#define APTP_CG_MAIN 1   // this file owns the timing-experiment stamp buffers
#include "conv_gemm_core.h"

using namespace aptp_cg;

namespace {

// ---------------------------------------------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------------------------------------------
// Element type of the operands.  T = __bf16 is the product path.  T = float is the fp32 PARITY instantiation (SURVEY section 8:
// "fp32 path kept for parity"; AptpConvGemmParams.io_f32): the same gather, tap walk, zero padding, K-slice bounds, split-K and
// epilogue code on fp32 tensors and exact-fp32 MFMAs (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain), so addressing, border
// classes and epilogue order are pinned on the GPU against the fp32 oracle at 1e-5 instead of bf16's 3e-3.  A 16-byte chunk is 8
// bf16 or 4 fp32 values and an LDS row stays 128 bytes, so a K-step is 64 bf16 / 32 fp32 channels (the host passes ncc, nK and
// Ktot in units of that step) and every index computation below is shared.  Never used by bench.py.
template <typename T> struct Elem;
template <> struct Elem<__bf16> { static constexpr int EPC = 8, KS = 64; };
template <> struct Elem<float> { static constexpr int EPC = 4, KS = 32; };

template <int BM, int BN, int WM, int WN, typename T = __bf16>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const KParams p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int EPC = Elem<T>::EPC, KS = Elem<T>::KS, ES = (int)sizeof(T);   // elements per 16-byte chunk, per K-step; element size
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MF = WTM / 16, NF = WTN / 16;
  constexpr int A_PASS = BM / 32, B_PASS = BN / 32;
  static_assert(BM % 32 == 0 && BN % 32 == 0 && WTM % 16 == 0 && WTN % 16 == 0, "tile shape");

  __shared__ __attribute__((aligned(16))) T smem[2 * (BM + BN) * KS];
  T* As = smem;
  T* Bs = smem + 2 * BM * KS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  int tm, tn, kz;
  decode_block(p, tiles_m, tiles_n, tm, tn, kz);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kt_begin = p.fd_sk.div(p.nK * kz);            // (nK * split_k < 2^31: checked on the host)
  const int kt_end = p.fd_sk.div(p.nK * (kz + 1));
  LnRaw<MF> ln_raw;
  ln_rows_issue<MF, WTM>(p, m0, wm, lane, ln_raw);

  // ---- per-thread staging coordinates -------------------------------------------------------------------------
  // Operands are fetched with raw buffer loads: an out-of-range offset returns zeros, which implements the conv
  // zero padding, the M/N/Cin tails and the predicated-off prefetch without any divergent control flow.
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x2src = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x2 ? p.x2 : p.x), 0, p.x2 ? p.x2_bytes : 0, 0x00020000);
  // (p.x / p.w / p.x2 are byte addresses typed __bf16* in KParams whatever T is: all offsets below are in BYTES)
  constexpr unsigned OOB = 0x80000000u;
  const int chunk = tid & 7, rowbase = tid >> 3;
  int a_iy0[A_PASS], a_ix0[A_PASS], a_pix0[A_PASS], a_m[A_PASS];
#pragma unroll
  for (int i = 0; i < A_PASS; ++i) {
    const int m = m0 + rowbase + 32 * i;
    if (m < p.M) {
      const int b = p.fd_hw.div(m), rem = m - b * p.HW;
      const int oy = p.fd_wout.div(rem), ox = rem - oy * p.Wout;
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad;
      a_pix0[i] = b * p.Hin * p.Win;
      a_m[i] = m;
    } else {
      a_iy0[i] = -100000; a_ix0[i] = -100000; a_pix0[i] = 0; a_m[i] = -1;   // always out of range => zero rows
    }
  }
  unsigned b_off[B_PASS];
#pragma unroll
  for (int i = 0; i < B_PASS; ++i) {
    const int n = n0 + rowbase + 32 * i;
    b_off[i] = n < p.N ? (unsigned)(((int64_t)n * p.Ktot + chunk * EPC) * ES) : OOB;
  }

  // K-iteration state of the NEXT tile to load (l_ky == KH: the x2 segment after the filter taps)
  const int nk_taps = p.KH * p.KW * p.ncc;
  int l_kt = kt_begin;
  int l_tap = 0, l_cc = 0, l_ky = 0, l_kx = 0;
  if (kt_begin != 0) {                   // (workgroup-uniform; only K-slices past the first pay the divisions)
    l_tap = kt_begin / p.ncc;
    l_cc = kt_begin - l_tap * p.ncc;
    l_ky = l_tap / p.KW;
    l_kx = l_tap - l_ky * p.KW;
  }
  if (kt_begin >= nk_taps) { l_ky = p.KH; l_kx = 0; l_cc = kt_begin - nk_taps; }

  u32x4 ra[A_PASS], rb[B_PASS];

  auto load_tile = [&](bool pred) {
    const int c = l_cc * KS + chunk * EPC;
    if (l_ky >= p.KH) {                       // wave-uniform: second operand, the output pixel itself
      const bool c_ok = pred && c < p.Cin2;
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        const unsigned off = ((unsigned)a_m[i] * (unsigned)p.ldx2 + (unsigned)c) * (unsigned)ES;
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(x2src, (c_ok && a_m[i] >= 0) ? off : OOB, 0, 0);
      }
    } else {
      const bool c_ok = pred && c < p.Cin;
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        int iy = a_iy0[i] + l_ky, ix = a_ix0[i] + l_kx;
        const bool ok = c_ok && (unsigned)iy < (unsigned)p.HinE && (unsigned)ix < (unsigned)p.WinE && !(p.zins & (iy | ix));
        iy >>= p.ups; ix >>= p.ups;
        const unsigned off = ((unsigned)(a_pix0[i] + iy * p.Win + ix) * (unsigned)p.ldx + (unsigned)c) * (unsigned)ES;   // < 2^31 (checked on the host)
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, ok ? off : OOB, 0, 0);
      }
    }
    const unsigned koff = (unsigned)l_kt * (KS * ES);
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, (b_off[i] == OOB || !pred) ? OOB : b_off[i] + koff, 0, 0);
    }
    // advance
    ++l_kt;
    if (++l_cc == (l_ky >= p.KH ? p.ncc2 : p.ncc)) {
      l_cc = 0;
      if (++l_kx == p.KW || l_ky >= p.KH) { l_kx = 0; ++l_ky; }
    }
  };

  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int r = rowbase + 32 * i;
      const int sw = chunk ^ ((r >> 1) & 7);
      *reinterpret_cast<u32x4*>(As + (buf * BM + r) * KS + sw * EPC) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      const int r = rowbase + 32 * i;
      const int sw = chunk ^ ((r >> 1) & 7);
      *reinterpret_cast<u32x4*>(Bs + (buf * BN + r) * KS + sw * EPC) = rb[i];
    }
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
#if APTP_ABLATE & 4
  bf16x8 abl_frag;
  for (int e = 0; e < 8; ++e) abl_frag[e] = (__bf16)(float)(lane + e);
  asm volatile("" : "+v"(abl_frag));
#endif
  auto compute = [&](int buf) {
    if constexpr (F32) {
      // exact-fp32 MFMA 16x16x4: a lane supplies ONE value per operand, k = lane >> 4.  The two 16-byte chunks 2*fq and 2*fq+1
      // of a row hold channels (2*fq + h)*4 + e of the 32-wide K-step: chunk h, element e feed the (h, e)-th of eight MFMAs,
      // the same channel for both operands -- the sum over k is complete, in an order fixed by the layout.
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 af[MF], wf[NF];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          const int r = wm * WTM + i * 16 + frow;
          const int sw = (fq * 2 + h) ^ ((r >> 1) & 7);
          af[i] = *reinterpret_cast<const f32x4*>(As + (buf * BM + r) * KS + sw * EPC);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int r = wn * WTN + j * 16 + frow;
          const int sw = (fq * 2 + h) ^ ((r >> 1) & 7);
          wf[j] = *reinterpret_cast<const f32x4*>(Bs + (buf * BN + r) * KS + sw * EPC);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], af[i][e], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[MF], wf[NF];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          const int r = wm * WTM + i * 16 + frow;
          const int sw = (s * 4 + fq) ^ ((r >> 1) & 7);
          af[i] = *reinterpret_cast<const bf16x8*>(As + (buf * BM + r) * KS + sw * EPC);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int r = wn * WTN + j * 16 + frow;
          const int sw = (s * 4 + fq) ^ ((r >> 1) & 7);
          wf[j] = *reinterpret_cast<const bf16x8*>(Bs + (buf * BN + r) * KS + sw * EPC);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
          for (int j = 0; j < NF; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- main loop: one barrier per K-step, loads for step t+1 in flight during the MFMAs of step t ----------------
  // (the prefetch of the step past the end is predicated off and its zero tile is stored but never read, which keeps
  // the loop body branch-free so the staging registers stay in VGPRs)
  float ln_mean[MF], ln_rstd[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) { ln_mean[i] = 0.f; ln_rstd[i] = 1.f; }
  {
    load_tile(kt_begin < kt_end);
    ln_rows_finish<MF>(p, lane, ln_raw, ln_mean, ln_rstd);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      load_tile((kt + 1) < kt_end);
      compute(buf);
      store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue --------------------------------------------------------------------------------------------------
  // acc[i][j][r] = out[m = m0 + wm*WTM + i*16 + (lane&15)][n = n0 + wn*WTN + j*16 + (lane>>4)*4 + r]
  if (p.split_k > 1 && p.counters) {
    if (!splitk_combine<256, MF, NF>(p, acc, tm * tiles_n + tn, kz, tid, reinterpret_cast<int*>(smem))) return;
    ln_rows_issue<MF, WTM>(p, m0, wm, lane, ln_raw, true);
    ln_rows_finish<MF>(p, lane, ln_raw, ln_mean, ln_rstd, true);
  } else
  if (p.split_k > 1) {
    float* ws = p.ws + (int64_t)kz * p.M * p.ws_ld;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int m = m0 + wm * WTM + i * 16 + frow;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int n = n0 + wn * WTN + j * 16 + fq * 4;
        if (n >= p.N) continue;
        float4 o; o.x = acc[i][j][0]; o.y = acc[i][j][1]; o.z = acc[i][j][2]; o.w = acc[i][j][3];
        *reinterpret_cast<float4*>(ws + (int64_t)m * p.ws_ld + n) = o;
      }
    }
    return;
  }
  run_epilogue<4, MF, NF, WTM, WTN, WN>(p, acc, m0, n0, tn, wm, wn, lane, wave, ln_mean, ln_rstd, reinterpret_cast<__bf16*>(smem));
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant: same tiling / LDS image / MFMA schedule / epilogue, but the operand tiles are copied global -> LDS
// directly (global_load_lds_dwordx4, 16 B per lane, destination = wave-uniform base + lane*16) instead of being staged
// through VGPRs and ds_write_b128.  The LDS image stays lane-linear per wave-instruction (8 rows x 128 B), so the XOR
// swizzle is applied to the per-lane SOURCE chunk (slot s of row r receives chunk s ^ ((r>>1)&7)) and the same XOR is
// used by the fragment reads.  Zero padding / ragged tails: an invalid lane streams zeros from g_zero_page.
// ---------------------------------------------------------------------------------------------------------------
// 8 KiB of zeros: an invalid (padding / out-of-range) lane streams from here; its pointer is advanced together with the
// real ones inside a tap, so it must cover one whole channel row (cin_pad * 2 B <= 8064 B, checked on the host).
__device__ uint4 g_zero_page[512];

// STAGES = 2: double buffer, one __syncthreads() per K-step (the LDS-DMA of step t+1 overlaps the MFMAs of step t).
// STAGES = 3: ring of three buffers, the DMA runs TWO K-steps ahead; a counted s_waitcnt vmcnt(N) (N = the DMA
//             instructions this thread issued for the newest tile) retires only the older tile, and a raw s_barrier
//             (no implicit vmcnt(0)) publishes it, so one tile stays in flight across every barrier.
// Address generation is incremental: the per-row source pointer is recomputed only when the filter tap changes (every
// cin_pad/64 K-steps) and otherwise advanced by 128 B per K-step; PMC counters showed the previous per-step
// recomputation (~115 VALU instructions per K-step per wave) made the loop VALU-issue-bound at 24 % MFMA utilisation.
// KS = 2: the workgroup carries TWO copies of the WM x WN wave grid; copy ks multiplies the ks-th 32-wide half of every
//         64-wide K-tile (intra-workgroup split-K).  All 2*WM*WN waves share the LDS-DMA of the tile; after the K loop copy 1
//         hands its accumulators to copy 0 through LDS and retires (s_barrier counts surviving waves only), copy 0 runs the
//         epilogue.  Twice the waves per SIMD on the same operand traffic, for the small tiles whose waves otherwise sit
//         alone on their SIMD between DMA issue, LDS reads and MFMAs.
#ifndef APTP_IL
#define APTP_IL 0
#endif
template <int BM, int BN, int WM, int WN, int STAGES, bool PP = false, int KU = 1, int KS = 1>
__global__ __launch_bounds__(WM * WN * KS * 64) void conv_gemm_dma_kernel(const KParams p) {
  constexpr int NW = WM * WN;                 // waves of one compute grid
  constexpr int NWA = NW * KS;                // waves per workgroup: 4 (256 threads) or 8 (512 threads)
  constexpr int NT = NWA * 64;
  constexpr int RPP = NT / 8;                 // tile rows covered by one LDS-DMA pass of the whole workgroup
  static_assert(NWA == 4 || NWA == 8, "4 or 8 waves per workgroup");
  static_assert(KS == 1 || (KS == 2 && !PP), "intra-workgroup K split: two copies, ring schedules only");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MF = WTM / 16, NF = WTN / 16;
  constexpr int A_PASS = (BM + RPP - 1) / RPP, B_PASS = (BN + RPP - 1) / RPP;
  static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BM % 8 == 0 && BN % 8 == 0, "tile shape");

  __shared__ __attribute__((aligned(16))) __bf16 smem[STAGES * (BM + BN) * BK];
  __bf16* As = smem;
  __bf16* Bs = smem + STAGES * BM * BK;
#ifdef APTP_STAMPS
  const unsigned long long st_entry = __builtin_readcyclecounter();
  unsigned long long st_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: LDS-DMA bases stay scalar
  const int ks = KS == 2 ? wave / NW : 0, cw = KS == 2 ? wave % NW : wave;   // compute-grid copy, wave within it
  const int wm = cw / WN, wn = cw % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  int tm, tn, kz;
  decode_block(p, tiles_m, tiles_n, tm, tn, kz);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kt_begin = p.fd_sk.div(p.nK * kz);            // (nK * split_k < 2^31: checked on the host)
  const int kt_end = p.fd_sk.div(p.nK * (kz + 1));
  __shared__ unsigned pf_scratch[64];
  APTP_PHASE(4);          // tile decoded (the first kernel-argument batches have arrived)
  prefetch_next(p, tid, NT, pf_scratch);
  LnRaw<MF> ln_raw;
  ln_rows_issue<MF, WTM>(p, m0, wm, lane, ln_raw);
  APTP_PHASE(5);          // next-launch prefetch + LayerNorm statistics requested

  const int rowbase = tid >> 3;
  const int schunk = (tid & 7) ^ ((rowbase >> 1) & 7);      // source chunk that lands in this lane's LDS slot
  const char* zpage = reinterpret_cast<const char*>(g_zero_page) + schunk * 16;
  const char* xbase = reinterpret_cast<const char*>(p.x) + schunk * 16;
  const bool tail_bad = ((p.ncc - 1) * BK + schunk * 8) >= p.Cin;   // this lane's chunk of the last channel step is padding
  const bool tail_bad2 = ((p.ncc2 - 1) * BK + schunk * 8) >= p.Cin2;  // same for the x2 segment
  const char* x2base = reinterpret_cast<const char*>(p.x2) + schunk * 16;
  int a_iy0[A_PASS], a_ix0[A_PASS], a_pix0[A_PASS], a_m[A_PASS];
#pragma unroll
  for (int i = 0; i < A_PASS; ++i) {
    const int m = m0 + rowbase + RPP * i;
    if (m < p.M && rowbase + RPP * i < BM) {
      const int b = p.fd_hw.div(m), rem = m - b * p.HW;
      const int oy = p.fd_wout.div(rem), ox = rem - oy * p.Wout;
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad;
      a_pix0[i] = b * p.Hin * p.Win;
      a_m[i] = m;
    } else {
      a_iy0[i] = -100000; a_ix0[i] = -100000; a_pix0[i] = 0; a_m[i] = -1;   // rows past M: every tap is "padding"
    }
  }

  // (l_ky == KH: the x2 segment after the filter taps, see KParams)
  const int nk_taps = p.KH * p.KW * p.ncc;
  int l_tap = 0, l_cc = 0, l_ky = 0, l_kx = 0;
  if (kt_begin != 0) {                   // (workgroup-uniform; only K-slices past the first pay the divisions)
    l_tap = kt_begin / p.ncc;
    l_cc = kt_begin - l_tap * p.ncc;
    l_ky = l_tap / p.KW;
    l_kx = l_tap - l_ky * p.KW;
  }
  if (kt_begin >= nk_taps) { l_ky = p.KH; l_kx = 0; l_cc = kt_begin - nk_taps; }

  const char* a_ptr[A_PASS];
  auto set_tap = [&](int cc0) {          // (re)compute the row pointers of the current tap, positioned at channel step cc0
    if (l_ky >= p.KH) {                  // wave-uniform: the second operand, read at the output pixel itself
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        const uint64_t va = reinterpret_cast<uint64_t>(x2base) + ((unsigned)a_m[i] * (unsigned)p.ldx2) * 2u + (unsigned)cc0 * (BK * 2);   // < 2^31
        const uint64_t vz = reinterpret_cast<uint64_t>(zpage) + (unsigned)cc0 * (BK * 2);
        a_ptr[i] = reinterpret_cast<const char*>((a_m[i] >= 0 && p.x2) ? va : vz);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      int iy = a_iy0[i] + l_ky, ix = a_ix0[i] + l_kx;
      const bool ok = (unsigned)iy < (unsigned)p.HinE && (unsigned)ix < (unsigned)p.WinE && !(p.zins & (iy | ix));
      iy >>= p.ups; ix >>= p.ups;
      const unsigned off = ((unsigned)(a_pix0[i] + iy * p.Win + ix) * (unsigned)p.ldx) * 2u + (unsigned)cc0 * (BK * 2);   // < 2^31
      const uint64_t va = reinterpret_cast<uint64_t>(xbase) + off;
      const uint64_t vz = reinterpret_cast<uint64_t>(zpage) + (unsigned)cc0 * (BK * 2);
      a_ptr[i] = reinterpret_cast<const char*>(ok ? va : vz);
    }
  };
  const char* b_ptr[B_PASS];
#pragma unroll
  for (int i = 0; i < B_PASS; ++i) {
    int n = n0 + rowbase + RPP * i;
    n = n < p.N ? n : p.N - 1;            // columns past N accumulate garbage that is never stored
    b_ptr[i] = reinterpret_cast<const char*>(p.w) + ((int64_t)n * p.Ktot + (int64_t)kt_begin * BK + schunk * 8) * 2;
  }
  set_tap(l_cc);
  APTP_PHASE(6);          // operand addresses ready

  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;

#if APTP_ABLATE & 1
  bool abl_first = true;
#endif
  auto issue_tile = [&](int buf) {
    const bool seg2 = l_ky >= p.KH;                                 // wave-uniform
    const bool last_cc = l_cc == (seg2 ? p.ncc2 : p.ncc) - 1;
    const bool tb = seg2 ? tail_bad2 : tail_bad;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      if (wave * 8 + RPP * i < BM) {                               // wave-uniform: the last pass may be partial
        const char* src = (last_cc && tb) ? zpage : a_ptr[i];
        __bf16* dst = As + (buf * BM + wave * 8 + RPP * i) * BK;   // wave-uniform; lane l lands at dst + l*16 B
#if APTP_ABLATE & 1
        if (abl_first)
#endif
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)dst, 16, 0, 0);
      }
      a_ptr[i] += BK * 2;
    }
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      if (wave * 8 + RPP * i < BN) {
        __bf16* dst = Bs + (buf * BN + wave * 8 + RPP * i) * BK;
#if APTP_ABLATE & 1
        if (abl_first)
#endif
        __builtin_amdgcn_global_load_lds((gbl_ptr)b_ptr[i], (lds_ptr)dst, 16, 0, 0);
      }
      b_ptr[i] += BK * 2;
    }
    if (++l_cc == (seg2 ? p.ncc2 : p.ncc)) {   // next tap (wave-uniform branch)
      l_cc = 0;
      if (++l_kx == p.KW || seg2) { l_kx = 0; ++l_ky; }
      set_tap(0);
    }
#if APTP_ABLATE & 1
    abl_first = false;
#endif
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef APTP_STAMPS
  unsigned long long st_acc[3] = {0, 0, 0}, st_prev = 0;
  const unsigned long long st_begin = __builtin_readcyclecounter();
#endif
  const int frow = lane & 15, fq = lane >> 4;
#if APTP_ABLATE & 4
  bf16x8 abl_frag;
  for (int e = 0; e < 8; ++e) abl_frag[e] = (__bf16)(float)(lane + e);
  asm volatile("" : "+v"(abl_frag));
#endif
  auto compute = [&](int buf) {
#pragma unroll
    for (int s0 = 0; s0 < 2 / KS; ++s0) {
      const int s = KS == 2 ? ks : s0;          // 32-wide half of the K-tile
      bf16x8 af[MF], wf[NF];
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int r = wm * WTM + i * 16 + frow;
        const int sw = (s * 4 + fq) ^ ((r >> 1) & 7);
#if APTP_ABLATE & 4
        af[i] = abl_frag; (void)r; (void)sw;
#else
        af[i] = *reinterpret_cast<const bf16x8*>(As + (buf * BM + r) * BK + sw * 8);
#endif
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int r = wn * WTN + j * 16 + frow;
        const int sw = (s * 4 + fq) ^ ((r >> 1) & 7);
#if APTP_ABLATE & 4
        wf[j] = abl_frag; (void)r; (void)sw;
#else
        wf[j] = *reinterpret_cast<const bf16x8*>(Bs + (buf * BN + r) * BK + sw * 8);
#endif
      }
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
#if APTP_ABLATE & 2
          asm volatile("" :: "v"(wf[j]), "v"(af[i]));
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
#endif
        }
    }
  };

  float ln_mean[MF], ln_rstd[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) { ln_mean[i] = 0.f; ln_rstd[i] = 1.f; }
  if constexpr (PP) {
    // Ping-pong schedule (8 waves = two groups of four; waves w and w+4 share a SIMD).  Every K-step is two slots
    // separated by workgroup barriers: in slot L a wave issues its share of the LDS-DMA for tile k+D, in slot C it
    // reads the fragments of tile k from LDS and runs the MFMAs.  Group 1 runs one slot behind group 0, so on every
    // SIMD one wave drives the matrix pipe (and the LDS read port) while the other drives the load path.  In the
    // lockstep loops the three engines take turns (in-kernel stamps, 128x160 tile: DMA issue 490-550, LDS reads + MFMA
    // 470-525, waits 400-550 cycles per K-step).
    //   RAW: a wave waits for its own DMA rows of tile k+1 at the end of C(k); group 0 reads tile k+1 two barriers
    //        later, group 1 three.
    //   WAR: tile k+D lands in the stage of tile k+D-S = k-2 (S = D+2), last read by group 1 in its C(k-2), which ends
    //        one barrier before group 0's L(k) starts.  (S = D+1 would let group 0's DMA overwrite the stage group 1 is
    //        still reading in the same slot.)
    static_assert(NWA == 8 && KS == 1 && STAGES >= 3, "ping-pong needs 8 waves and a ring of >= 3 stages");
    static_assert(BM % RPP == 0, "ring: the activation tile must be whole row passes");
    constexpr int D = STAGES - 2;
    constexpr int A_FULL = BM / RPP, B_FULL = BN / RPP;
    constexpr int NLD_LO = A_FULL + B_FULL, NLD_HI = A_PASS + B_PASS;
    static_assert(NLD_HI * (D - 1) <= 63, "vmcnt immediate");
    const bool hi = (NLD_HI != NLD_LO) && (wave * 8 + RPP * (B_PASS - 1) < BN);   // wave-uniform
    const int grp = wave >> 2;
    const int n = kt_end - kt_begin;
    if (n > 0) {
      const int pre = n < D ? n : D;
      for (int t = 0; t < pre; ++t) issue_tile(t);
      ln_rows_finish<MF>(p, lane, ln_raw, ln_mean, ln_rstd);
      if (pre == D) {
        if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_HI * (D - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_LO * (D - 1)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (grp == 1) {                       // stagger: group 1 starts one slot late
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      int cur = 0, nxt = D % STAGES;
      for (int kt = 0; kt < n; ++kt) {
        // ---- slot L(kt): the load path ----
        APTP_STAMP(0);
        const bool issue = kt + D < n;
        if (issue) issue_tile(nxt);
        APTP_STAMP(1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        // ---- slot C(kt): LDS reads + MFMAs ----
        __builtin_amdgcn_s_setprio(1);
        compute(cur);
        __builtin_amdgcn_s_setprio(0);
        APTP_STAMP(2);
        asm volatile("" ::: "memory");
#if !(APTP_ABLATE & 16)
        if (issue) {
          if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_HI * (D - 1)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_LO * (D - 1)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        APTP_STAMP(3);
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
      }
      if (grp == 0) {                       // group 0 pairs group 1's last barrier
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
  } else
  if constexpr (KU == 2) {
    // Two K-tiles per synchronisation point (an effective K-step of 128): iteration i issues tiles 2i+D and 2i+D+1 into the
    // two stages read in iteration i-1, runs the MFMAs of tiles 2i and 2i+1, waits until tiles 2i+2 and 2i+3 have landed
    // (D-2 newer tiles stay in flight) and meets ONE barrier: half the s_waitcnt / s_barrier stops of the ring above.
    static_assert(STAGES >= 4 && STAGES % 2 == 0, "pairs of stages");
    constexpr int D = STAGES - 2;
    constexpr int A_FULL = BM / RPP, B_FULL = BN / RPP;
    constexpr int NLD_LO = A_FULL + B_FULL, NLD_HI = A_PASS + B_PASS;
    static_assert(BM % RPP == 0, "ring: the activation tile must be whole row passes");
    static_assert(NLD_HI * (D - 2) <= 63, "vmcnt immediate");
    const bool hi = (NLD_HI != NLD_LO) && (wave * 8 + RPP * (B_PASS - 1) < BN);   // wave-uniform
    const int n = kt_end - kt_begin;
    if (n > 0) {
      const int pre = n < D ? n : D;
      for (int t = 0; t < pre; ++t) issue_tile(t);
      APTP_PHASE(0);
      ln_rows_finish<MF>(p, lane, ln_raw, ln_mean, ln_rstd);
      if (pre == D && D > 2) {   // tiles 0 and 1 landed, D-2 tiles still in flight
        if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_HI * (D - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_LO * (D - 2)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      APTP_PHASE(1);
      int cur = 0, nxt = D;
      for (int kt = 0; kt < n; kt += 2) {
        const int ni = n - (kt + D);          // tiles left to request: two per iteration while >= 2
        if (ni >= 1) issue_tile(nxt);
        if (ni >= 2) issue_tile(nxt + 1);
        compute(cur);
        if (kt + 1 < n) compute(cur + 1);
        asm volatile("" ::: "memory");
        if (ni >= 2 && D > 2) {
          if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_HI * (D - 2)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_LO * (D - 2)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (D == 2: plain double buffering of 128-wide steps) / tail
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur = cur + 2 == STAGES ? 0 : cur + 2;
        nxt = nxt + 2 == STAGES ? 0 : nxt + 2;
      }
    }
  } else
  if constexpr (STAGES == 2) {
    if (kt_begin < kt_end) {
      issue_tile(0);
      APTP_PHASE(0);
      ln_rows_finish<MF>(p, lane, ln_raw, ln_mean, ln_rstd);
      __syncthreads();                       // (the compiler drains vmcnt(0) for the LDS-DMA before the barrier)
      APTP_PHASE(1);
      int buf = 0;
      for (int kt = kt_begin; kt < kt_end; ++kt) {
        APTP_STAMP(0);
        if (kt + 1 < kt_end) issue_tile(buf ^ 1);   // in flight during the MFMAs below
        APTP_STAMP(1);
        compute(buf);
        APTP_STAMP(2);
#if !(APTP_ABLATE & 8)
        __syncthreads();
#endif
        APTP_STAMP(3);
        buf ^= 1;
      }
    }
  } else {
    // STAGES-deep ring, D = STAGES-1 tiles of LDS-DMA in flight across the barriers (counted vmcnt, raw s_barrier): the
    // operand round trip (~1-2k cycles) is several K-steps long at these tile sizes, so one tile ahead is not enough.
    //   iteration kt: issue tile kt+D into stage (kt+D)%S == (kt-1)%S (last read before the previous barrier),
    //                 compute stage kt%S, wait until tile kt+1 has landed (D-1 newer tiles stay in flight), barrier.
    // Waves may issue different numbers of DMAs per tile (partial last row pass): each waits on its own count.
    constexpr int D = STAGES - 1;
    constexpr int A_FULL = BM / RPP, B_FULL = BN / RPP;            // passes every wave takes part in
    constexpr int NLD_LO = A_FULL + B_FULL;
    constexpr int NLD_HI = A_PASS + B_PASS;
    static_assert(BM % RPP == 0, "ring: the activation tile must be whole row passes");
    static_assert(NLD_HI * (D - 1) <= 63, "vmcnt immediate");
    const bool hi = (NLD_HI != NLD_LO) && (wave * 8 + RPP * (B_PASS - 1) < BN);   // wave-uniform
    const int n = kt_end - kt_begin;
    if (n > 0) {
      const int pre = n < D ? n : D;
      for (int t = 0; t < pre; ++t) issue_tile(t);
      APTP_PHASE(0);
      ln_rows_finish<MF>(p, lane, ln_raw, ln_mean, ln_rstd);
      if (pre == D) {   // tile 0 landed, D-1 tiles still in flight
        if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_HI * (D - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_LO * (D - 1)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      APTP_PHASE(1);
      int cur = 0, nxt = D;          // stage of tile kt, stage of tile kt+D
#if APTP_IL
      // Interleaved K-step (experiment, -DAPTP_IL=1): hipcc emits a K-step as [all LDS-DMA instructions][fragment reads][MFMAs], and
      // a DMA instruction blocks the in-order wave while the CU's 64 B/clk load path is busy -- the load path, the LDS read port and
      // the matrix pipe take turns.  Here the wave's DMA pieces of tile kt+D and the fragment reads of the second 32-wide half are
      // placed BETWEEN the MFMAs, by hand, pinned with sched_barrier(0).
      if constexpr (KS == 1) {
        constexpr int NP = A_PASS + B_PASS, NM = 2 * MF * NF;
        auto piece = [&](int q, int buf) {
          const bool seg2 = l_ky >= p.KH;
          const bool last_cc = l_cc == (seg2 ? p.ncc2 : p.ncc) - 1;
          const bool tb = seg2 ? tail_bad2 : tail_bad;
          if (q < A_PASS) {
            const int i = q;
            if (wave * 8 + RPP * i < BM) {
              const char* src = (last_cc && tb) ? zpage : a_ptr[i];
              __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(As + (buf * BM + wave * 8 + RPP * i) * BK), 16, 0, 0);
            }
            a_ptr[i] += BK * 2;
          } else {
            const int i = q - A_PASS;
            if (wave * 8 + RPP * i < BN)
              __builtin_amdgcn_global_load_lds((gbl_ptr)b_ptr[i], (lds_ptr)(Bs + (buf * BN + wave * 8 + RPP * i) * BK), 16, 0, 0);
            b_ptr[i] += BK * 2;
          }
        };
        auto next_tap = [&]() {
          const bool seg2 = l_ky >= p.KH;
          if (++l_cc == (seg2 ? p.ncc2 : p.ncc)) {
            l_cc = 0;
            if (++l_kx == p.KW || seg2) { l_kx = 0; ++l_ky; }
            set_tap(0);
          }
        };
        auto frags = [&](int buf, int sh, bf16x8 (&af)[MF], bf16x8 (&wf)[NF]) {
#pragma unroll
          for (int i = 0; i < MF; ++i) {
            const int r = wm * WTM + i * 16 + frow;
            af[i] = *reinterpret_cast<const bf16x8*>(As + (buf * BM + r) * BK + ((sh * 4 + fq) ^ ((r >> 1) & 7)) * 8);
          }
#pragma unroll
          for (int j = 0; j < NF; ++j) {
            const int r = wn * WTN + j * 16 + frow;
            wf[j] = *reinterpret_cast<const bf16x8*>(Bs + (buf * BN + r) * BK + ((sh * 4 + fq) ^ ((r >> 1) & 7)) * 8);
          }
        };
        for (int kt = 0; kt < n; ++kt) {
          const bool issue = kt + D < n;
          bf16x8 af0[MF], wf0[NF], af1[MF], wf1[NF];
          frags(cur, 0, af0, wf0);
          if (issue) {
            int m = 0;
#pragma unroll
            for (int sh = 0; sh < 2; ++sh)
#pragma unroll
              for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j, ++m) {
                  if (sh == 0 && m == 1) frags(cur, 1, af1, wf1);
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh ? wf1[j] : wf0[j], sh ? af1[i] : af0[i], acc[i][j], 0, 0, 0);
                  if ((m + 1) * NP / NM > m * NP / NM) piece(m * NP / NM, nxt);
                  __builtin_amdgcn_sched_barrier(0);
                }
            next_tap();
          } else {
            frags(cur, 1, af1, wf1);
#pragma unroll
            for (int sh = 0; sh < 2; ++sh)
#pragma unroll
              for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh ? wf1[j] : wf0[j], sh ? af1[i] : af0[i], acc[i][j], 0, 0, 0);
          }
          asm volatile("" ::: "memory");
          if (issue) {
            if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_HI * (D - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_LO * (D - 1)) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          cur = cur + 1 == STAGES ? 0 : cur + 1;
          nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
        }
      } else
#endif
      for (int kt = 0; kt < n; ++kt) {
        const bool issue = kt + D < n;
        APTP_STAMP(0);
        if (issue) issue_tile(nxt);
        APTP_STAMP(1);
        compute(cur);
        APTP_STAMP(2);
        asm volatile("" ::: "memory");
        if (issue) {
          if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_HI * (D - 1)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_LO * (D - 1)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tail: everything still in flight is needed soon
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's LDS reads of `cur` are done
#if !(APTP_ABLATE & 8)
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        APTP_STAMP(3);
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
      }
    }
  }

  APTP_PHASE(2);
#ifdef APTP_STAMPS
  if (lane == 0 && kz == 0) {
    const int slot = (blockIdx.x * NW + wave) & 4095;
    g_stamps[slot * 4 + 0] = st_acc[0]; g_stamps[slot * 4 + 1] = st_acc[1]; g_stamps[slot * 4 + 2] = st_acc[2];
    g_stamps[slot * 4 + 3] = __builtin_readcyclecounter() - st_begin;
  }
#endif
  if constexpr (KS == 2) {
    // every path above ends on a barrier with the DMA drained and the fragment reads done: the stages are free
    f32x4* xch = reinterpret_cast<f32x4*>(smem) + cw * (MF * NF * 64) + lane;
    if (ks == 1) {
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) xch[(i * NF + j) * 64] = acc[i][j];
    }
    __syncthreads();
    if (ks == 1) return;                   // (wave-uniform; later barriers count the surviving waves)
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] += xch[(i * NF + j) * 64];
  }
  if (p.split_k > 1 && p.counters) {
    if (!splitk_combine<NW * 64, MF, NF>(p, acc, tm * tiles_n + tn, kz, tid, reinterpret_cast<int*>(smem))) return;
    ln_rows_issue<MF, WTM>(p, m0, wm, lane, ln_raw, true);
    ln_rows_finish<MF>(p, lane, ln_raw, ln_mean, ln_rstd, true);
  } else
  if (p.split_k > 1) {
    float* ws = p.ws + (int64_t)kz * p.M * p.ws_ld;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int m = m0 + wm * WTM + i * 16 + frow;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int n = n0 + wn * WTN + j * 16 + fq * 4;
        if (n >= p.N) continue;
        float4 o; o.x = acc[i][j][0]; o.y = acc[i][j][1]; o.z = acc[i][j][2]; o.w = acc[i][j][3];
        *reinterpret_cast<float4*>(ws + (int64_t)m * p.ws_ld + n) = o;
      }
    }
    return;
  }
  static_assert(NW * 16 * (WTN + 4) * 4 <= (int)sizeof(smem), "epilogue transpose buffer");
  static_assert(KS == 1 || NW * MF * NF * 64 * 16 <= (int)sizeof(smem), "accumulator hand-over buffer");
  run_epilogue<NW, MF, NF, WTM, WTN, WN>(p, acc, m0, n0, tn, wm, wn, lane, cw, ln_mean, ln_rstd, smem);
#ifdef APTP_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  APTP_EPI(5);
  APTP_PHASE(3);
  if (lane == 0 && kz == 0) {
    const int slot = (blockIdx.x * NW + wave) & 4095;
    for (int i = 0; i < 8; ++i) g_phase[slot * 8 + i] = st_phase[i];
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride-1 / pad-1 convolution with the INPUT HALO KEPT IN LDS (tiles APTP_TILE_HALO_*).
// The implicit-GEMM kernels above fetch a [BM x 64] activation tile per (tap, channel step): nine shifted copies of almost
// the same pixels, 16 of the 36 KB a 128x160 tile moves per K-step.  With one workgroup per CU those launches run at the
// rate their operand bytes in flight allow (Little's law), so fewer bytes per FLOP is the lever.
// Here the workgroup's 128 output pixels are whole image rows (R = 128 / W of them) and, per 64-channel step, the
// (R+2) x (W+2) input patch is copied to LDS ONCE (LDS-DMA, zero page outside the image); the nine taps are then nine
// fragment reads of the same patch at shifted row offsets.  K-loop order is channel-step-major: step kt -> (cc = kt / 9,
// tap = kt % 9); the weights of one (tap, cc) stream through a STAGES-deep ring exactly as in conv_gemm_dma_kernel, the
// patch is double-buffered and requested one channel step (nine K-steps) ahead.  Per K-step 20 + 33/9 = 23.7 KB instead of
// 36 KB.  8 waves (4 x 2), BM = 128, same LDS images (XOR-swizzled 128-byte rows), same epilogues.
// ---------------------------------------------------------------------------------------------------------------
template <int BN, int STAGES>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const KParams p) {
  constexpr int BM = 128, WM = 4, WN = 2, NW = 8, NT = 512, RPP = 64;
  constexpr int WTM = BM / WM, WTN = BN / WN, MF = WTM / 16, NF = WTN / 16;
  constexpr int PROWS = 264;                          // (R+2)*(W+2) <= 264 for W = 64 (R = 2), 32 (R = 4), 16 (R = 8)
  constexpr int P_PASS = (PROWS + RPP - 1) / RPP, B_PASS = (BN + RPP - 1) / RPP;
  constexpr int D = STAGES - 1;
  static_assert(WTN % 16 == 0 && STAGES >= 3, "tile shape");

  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * PROWS * BK + STAGES * BN * BK];
  __bf16* Ps = smem;
  __bf16* Bs = smem + 2 * PROWS * BK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  int tm, tn, kz;
  decode_block(p, tiles_m, tiles_n, tm, tn, kz);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kt_begin = p.fd_sk.div(p.nK * kz);            // (nK * split_k < 2^31: checked on the host)
  const int kt_end = p.fd_sk.div(p.nK * (kz + 1));
  LnRaw<MF> ln_raw;
  ln_rows_issue<MF, WTM>(p, m0, wm, lane, ln_raw);

  const int W = p.Wout, PW = W + 2, R = BM / W;
  const int prows = (R + 2) * PW;
  const int b_img = p.fd_hw.div(m0), oy0 = (m0 - b_img * p.HW) / W;

  // ---- patch source pointers: row prow = (pr, pc) of the patch <-> input pixel (oy0 - 1 + pr, pc - 1) ---------------------
  const int rowbase = tid >> 3;
  const int schunk = (tid & 7) ^ ((rowbase >> 1) & 7);      // (rows of one pass differ by 64: same swizzle term)
  const char* zpage = reinterpret_cast<const char*>(g_zero_page) + schunk * 16;
  const bool tail_bad = ((p.ncc - 1) * BK + schunk * 8) >= p.Cin;
  const char* p_ptr[P_PASS];
#pragma unroll
  for (int i = 0; i < P_PASS; ++i) {
    const int pr_ = rowbase + RPP * i;
    const int pr = pr_ / PW, pc = pr_ - pr * PW;
    const int iy = oy0 - 1 + pr, ix = pc - 1;
    const bool ok = pr_ < prows && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
    const unsigned off = ((unsigned)((b_img * p.Hin + iy) * p.Win + ix) * (unsigned)p.ldx) * 2u;   // < 2^31
    p_ptr[i] = ok ? reinterpret_cast<const char*>(p.x) + off + schunk * 16 : nullptr;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  auto issue_patch = [&](int cc, int pbuf) {
    const bool bad = tail_bad && cc == p.ncc - 1;
#pragma unroll
    for (int i = 0; i < P_PASS; ++i) {
      if (wave * 8 + RPP * i < prows) {                            // wave-uniform
        const char* src = (p_ptr[i] && !bad) ? p_ptr[i] + cc * (BK * 2) : zpage;
        __bf16* dst = Ps + (pbuf * PROWS + wave * 8 + RPP * i) * BK;
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)dst, 16, 0, 0);
      }
    }
  };
  // ---- weights: step kt -> (cc, tap); packed [N][tap][cin_pad] ---------------------------------------------------------
  const char* b_row[B_PASS];
#pragma unroll
  for (int i = 0; i < B_PASS; ++i) {
    int n = n0 + rowbase + RPP * i;
    n = n < p.N ? n : p.N - 1;
    b_row[i] = reinterpret_cast<const char*>(p.w) + ((int64_t)n * p.Ktot + schunk * 8) * 2;
  }
  const int cin_pad = p.ncc * BK;
  auto issue_w = [&](int kt, int stage) {
    const int cc = kt / 9, tap = kt - cc * 9;
    const int koff = (tap * cin_pad + cc * BK) * 2;
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      if (wave * 8 + RPP * i < BN) {
        __bf16* dst = Bs + (stage * BN + wave * 8 + RPP * i) * BK;
        __builtin_amdgcn_global_load_lds((gbl_ptr)(b_row[i] + koff), (lds_ptr)dst, 16, 0, 0);
      }
    }
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  int prow0[MF];                                   // patch row of this lane's output pixel at tap (0, 0)
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int r = wm * WTM + i * 16 + frow;
    const int lr = r / W;
    prow0[i] = lr * PW + (r - lr * W);
  }
  auto compute = [&](int pbuf, int tap, int stage) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int toff = ky * PW + kx;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 af[MF], wf[NF];
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int pr_ = prow0[i] + toff;
        const int sw = (s2 * 4 + fq) ^ ((pr_ >> 1) & 7);
        af[i] = *reinterpret_cast<const bf16x8*>(Ps + (pbuf * PROWS + pr_) * BK + sw * 8);
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int r = wn * WTN + j * 16 + frow;
        const int sw = (s2 * 4 + fq) ^ ((r >> 1) & 7);
        wf[j] = *reinterpret_cast<const bf16x8*>(Bs + (stage * BN + r) * BK + sw * 8);
      }
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  float ln_mean[MF], ln_rstd[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) { ln_mean[i] = 0.f; ln_rstd[i] = 1.f; }

  // ---- main loop: weights ring with D tiles in flight; patch of the next channel step requested at tap 0 --------------------
  // vmcnt is in order, so the wait that retires weight tile kt+1 must allow exactly the requests issued after it: D-1 weight
  // tiles plus the patch requests of the last D steps (a plain (D-1)*NLD would retire the newest weight tile too whenever
  // a patch was just requested: one exposed round trip per channel step).
  constexpr int B_FULL = BN / RPP;
  constexpr int NLD_LO = B_FULL, NLD_HI = B_PASS;
  const bool hi = (NLD_HI != NLD_LO) && (wave * 8 + RPP * (B_PASS - 1) < BN);   // wave-uniform
  const int nld = hi ? NLD_HI : NLD_LO;
  int pw_cnt = 0;                                   // patch requests this wave issues per patch (wave-uniform)
#pragma unroll
  for (int i = 0; i < P_PASS; ++i) pw_cnt += (wave * 8 + RPP * i < prows) ? 1 : 0;
  auto wait_vm = [&](int allowed) {                 // s_waitcnt needs an immediate: wave-uniform switch over the few values
    switch (allowed) {
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  const int n = kt_end - kt_begin;
  if (n > 0) {
    const int cc_first = kt_begin / 9;
    issue_patch(cc_first, cc_first & 1);
    const int pre = n < D ? n : D;
    for (int t = 0; t < pre; ++t) issue_w(kt_begin + t, t);
    ln_rows_finish<MF>(p, lane, ln_raw, ln_mean, ln_rstd);
    if (pre == D) {
      if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_HI * (D - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD_LO * (D - 1)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int cur = 0, nxt = D % STAGES;
    int cc = cc_first, tap = kt_begin - cc_first * 9;
    int since_patch = D;                                // steps since the last patch request (>= D: none in the window)
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const bool issue = kt + D < kt_end;
      if (issue) issue_w(kt + D, nxt);
      // the patch of the next channel step: requested when this step's first tap starts (or at the slice start)
      if ((tap == 0 || kt == kt_begin) && (cc + 1) * 9 < kt_end) { issue_patch(cc + 1, (cc + 1) & 1); since_patch = 0; }
      compute(cc & 1, tap, cur);
      asm volatile("" ::: "memory");
      if (issue) wait_vm(nld * (D - 1) + (since_patch < D ? pw_cnt : 0));
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ++since_patch;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      cur = cur + 1 == STAGES ? 0 : cur + 1;
      nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
      if (++tap == 9) { tap = 0; ++cc; }
    }
  }

  if (p.split_k > 1 && p.counters) {
    if (!splitk_combine<NT, MF, NF>(p, acc, tm * tiles_n + tn, kz, tid, reinterpret_cast<int*>(smem))) return;
    ln_rows_issue<MF, WTM>(p, m0, wm, lane, ln_raw, true);
    ln_rows_finish<MF>(p, lane, ln_raw, ln_mean, ln_rstd, true);
  } else
  if (p.split_k > 1) {
    float* ws = p.ws + (int64_t)kz * p.M * p.ws_ld;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int m = m0 + wm * WTM + i * 16 + frow;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int nn = n0 + wn * WTN + j * 16 + fq * 4;
        if (nn >= p.N) continue;
        float4 o; o.x = acc[i][j][0]; o.y = acc[i][j][1]; o.z = acc[i][j][2]; o.w = acc[i][j][3];
        *reinterpret_cast<float4*>(ws + (int64_t)m * p.ws_ld + nn) = o;
      }
    }
    return;
  }
  static_assert(NW * 16 * (WTN + 4) * 4 <= (int)sizeof(smem), "epilogue transpose buffer");
  run_epilogue<NW, MF, NF, WTM, WTN, WN>(p, acc, m0, n0, tn, wm, wn, lane, wave, ln_mean, ln_rstd, smem);
}

// split-K reducer + epilogue: one thread per (row, 4 packed columns)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const KParams p) {
  const int quads = (p.act == APTP_ACT_GEGLU) ? p.N / 8 : p.N / 4;   // GEGLU: one thread per h-quad (+ its g-quad)
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)p.M * quads) return;
  const int m = (int)(idx / quads);
  const int q = (int)(idx - (int64_t)m * quads);
  int n;
  if (p.act == APTP_ACT_GEGLU) n = (q >> 2) * 32 + (q & 3) * 4;
  else n = q * 4;
  float h[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
  // four slabs in flight per thread (one slab per round trip otherwise); the adds stay in slice order => deterministic
  for (int z0 = 0; z0 < p.split_k; z0 += 4) {
    float4 a[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int z = z0 + u < p.split_k ? z0 + u : z0;
      const float* row = p.ws + ((int64_t)z * p.M + m) * p.ws_ld;
      a[u] = *reinterpret_cast<const float4*>(row + n);
      if (p.act == APTP_ACT_GEGLU) c[u] = *reinterpret_cast<const float4*>(row + n + 16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (z0 + u < p.split_k) {
        h[0] += a[u].x; h[1] += a[u].y; h[2] += a[u].z; h[3] += a[u].w;
        if (p.act == APTP_ACT_GEGLU) { g[0] += c[u].x; g[1] += c[u].y; g[2] += c[u].z; g[3] += c[u].w; }
      }
    }
  }
  RowCtx rc;
  row_info(p, m, rc);
  if (p.ln_stats) {
    const float4* sp = reinterpret_cast<const float4*>(p.ln_stats);
    float a = 0.f, a2 = 0.f;
    for (int pr = 0; pr < (p.ln_slots >> 1); ++pr) { const float4 t = sp[(int64_t)pr * p.M + m]; a += t.x + t.z; a2 += t.y + t.w; }
    ln_finish(p, a, a2, rc);
  }
  float st[2] = {0.f, 0.f};     // (row statistics are only emitted by the un-split kernel: checked on the host)
  if (p.act == APTP_ACT_GEGLU) epilogue_quad<true>(p, m, rc, n, h, g, st);
  else epilogue_quad<false>(p, m, rc, n, h, h, st);
}

// Split-K reduce + the GroupNorm(+SiLU) that follows the convolution, for the small maps (AptpConvGemmParams.gn_gamma):
// grid (groups, B); the workgroup owns the HW x cg column slab of its (sample, group): it sums the K-slices in slice order
// (bit-identical to splitk_reduce_kernel), adds bias / rowbias, rounds to bf16, reduces (sum, sumsq) of the rounded values in a
// fixed order, and writes the normalised activation.  Replaces splitk_reduce_kernel + gn_group_kernel (6 + 8 us on a level-16
// map) and the round trip of the convolution output in between.
constexpr int GN_RED_NT = 1024;            // 16 waves: only groups x B workgroups exist, so each must keep many loads in flight
constexpr int GN_RED_QPT = 6;               // column quads per thread: HW * cg / 4 <= 6 * 1024
__global__ __launch_bounds__(GN_RED_NT) void splitk_reduce_gn_kernel(const KParams p) {
  __shared__ float red[2][GN_RED_NT];
  __shared__ float stat[2];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cg = p.gn_C / p.gn_groups, qpr = cg >> 2;         // channels per group, quads per row
  const int nquads = p.HW * qpr;
  float v[GN_RED_QPT][4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < GN_RED_QPT; ++i) {
    const int q = tid + GN_RED_NT * i;
    if (q < nquads) {
      const int r = q / qpr, n = g * cg + (q - r * qpr) * 4;
      const int m = b * p.HW + r;
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      // four slabs in flight per round trip; the adds stay in slice order (bit-identical to splitk_reduce_kernel)
      for (int z0 = 0; z0 < p.split_k; z0 += 4) {
        float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int z = z0 + u < p.split_k ? z0 + u : z0;
          t[u] = *reinterpret_cast<const float4*>(p.ws + ((int64_t)z * p.M + m) * p.ws_ld + n);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (z0 + u < p.split_k) { a[0] += t[u].x; a[1] += t[u].y; a[2] += t[u].z; a[3] += t[u].w; }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = a[e];
        if (p.bias) x += p.bias[n + e];
        if (p.rowbias) x += p.rowbias[(int64_t)b * p.ld_rowbias + n + e];
        x = (float)(__bf16)x;                               // the value the separate launches would have stored
        v[i][e] = x;
        s1 += x; s2 += x * x;
      }
    }
  }
  red[0][tid] = s1; red[1][tid] = s2;
  __syncthreads();
  if (tid < 64) {                                             // fixed-order fold: 16 strided sums, then a 64-lane tree
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int w = 0; w < GN_RED_NT / 64; ++w) { a += red[0][tid + 64 * w]; c += red[1][tid + 64 * w]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); c += __shfl_xor(c, off); }
    if (tid == 0) {
      const float inv = 1.0f / ((float)cg * (float)p.HW);
      const float mean = a * inv;
      float var = c * inv - mean * mean;
      var = var < 0.f ? 0.f : var;
      stat[0] = mean; stat[1] = rsqrtf(var + p.gn_eps);
    }
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
  __bf16* y = reinterpret_cast<__bf16*>(p.y);
#pragma unroll
  for (int i = 0; i < GN_RED_QPT; ++i) {
    const int q = tid + GN_RED_NT * i;
    if (q < nquads) {
      const int r = q / qpr, n = g * cg + (q - r * qpr) * 4;
      const int m = b * p.HW + r;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = (v[i][e] - mean) * rstd * p.gn_gamma[n + e] + p.gn_beta[n + e];
        o[e] = p.gn_silu ? silu_f(z) : z;
      }
      uint2 pk; pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]);
      *reinterpret_cast<uint2*>(y + (int64_t)m * p.ldy + n) = pk;
    }
  }
  // pad columns [gn_C, N) of this sample (zero weights upstream, zero here): the last group's workgroup clears them
  if (g == p.gn_groups - 1 && p.N > p.gn_C) {
    const int padw = p.N - p.gn_C;
    for (int e = tid; e < p.HW * padw; e += GN_RED_NT) {
      const int r = e / padw, n = p.gn_C + (e - r * padw);
      y[(int64_t)(b * p.HW + r) * p.ldy + n] = (__bf16)0.0f;
    }
  }
}

struct TileCfg { int bm, bn, wn, nw; };   // tile extents, the wave grid's N extent and the wave count (launch table below)
const TileCfg kTiles[] = {
    {0, 0, 0, 0}, {128, 128, 2, 4}, {128, 160, 2, 4}, {64, 128, 2, 4}, {64, 160, 2, 4}, {128, 64, 2, 4},
    {64, 64, 2, 4}, {128, 128, 2, 4}, {128, 160, 2, 4}, {64, 128, 2, 4}, {64, 160, 2, 4}, {128, 64, 2, 4},
    {64, 64, 2, 4}, {128, 128, 2, 4}, {128, 160, 2, 4}, {64, 128, 2, 4}, {64, 160, 2, 4}, {128, 64, 2, 4},
    {64, 64, 2, 4}, {128, 160, 2, 8}, {256, 160, 2, 8}, {128, 128, 4, 8}, {256, 128, 2, 8}, {64, 160, 2, 4},
    {64, 128, 2, 4}, {64, 64, 2, 4}, {128, 64, 2, 4}, {128, 128, 2, 4}, {128, 160, 2, 8}, {128, 160, 2, 8},
    {128, 128, 4, 8}, {128, 128, 4, 8}, {256, 128, 2, 8}, {128, 160, 2, 8}, {128, 160, 2, 8}, {128, 128, 4, 8},
    {128, 128, 4, 8}, {64, 160, 2, 8}, {64, 128, 4, 8}, {64, 160, 2, 8}, {128, 64, 2, 8}, {256, 128, 2, 8},
    {128, 256, 4, 8},
    {128, 160, 2, 8}, {128, 128, 2, 8},       // 43, 44: 3x3 halo-in-LDS kernel (conv3x3_halo_kernel)
    {64, 64, 2, 4}, {64, 64, 2, 4}, {64, 128, 2, 4}, {128, 64, 2, 4},   // 45-48: deep rings for latency-bound small-M launches
    {64, 64, 2, 4}, {64, 64, 2, 4}, {64, 128, 2, 4}, {64, 128, 2, 4}, {128, 64, 2, 4}, {128, 128, 2, 4}, {64, 160, 2, 4},
    {128, 160, 2, 8}, {128, 128, 4, 8},                                  // 49-57: two K-tiles per barrier
    {64, 64, 2, 4}, {64, 64, 2, 4}, {64, 128, 2, 4}, {64, 128, 2, 4}, {128, 64, 2, 4}, {128, 128, 2, 4},   // 58-63: intra-workgroup K split (compute grid 2 x 2)
    {256, 160, 2, 8}, {256, 128, 2, 8}, {128, 256, 4, 8},               // 64-66: persistent stream-K macro-tiles (conv_gemm_sk.hip)
    {256, 160, 2, 8}, {256, 128, 2, 8}, {128, 256, 4, 8},               // 67-69: the same, fragment reads in the load slot
    {128, 160, 2, 8}, {128, 160, 2, 8}, {128, 128, 2, 8}, {128, 128, 2, 8}};   // 70-73: stream-K on 128-row tiles (round 4)
constexpr int kNumTiles = 74;
inline bool is_sk_tile(int t) { return t >= APTP_TILE_SK_256x160 && t <= APTP_TILE_SKL_128x128; }
static_assert(sizeof(kTiles) / sizeof(kTiles[0]) == kNumTiles, "tile table");

int pick_tile(const AptpConvGemmParams* p, int M) {
  if (p->tile != APTP_TILE_AUTO) return p->tile;
  const bool geglu = p->act == APTP_ACT_GEGLU;
  // N-tile: least padding waste; 160 not usable with GEGLU (needs an even number of 16-col fragments per wave)
  auto waste = [&](int bn) { return ((p->N + bn - 1) / bn) * bn - p->N; };
  int bn = 128;
  if (!geglu && waste(160) < waste(128)) bn = 160;
  if (p->N <= 64 || (waste(64) < waste(bn) && p->N < 128)) bn = 64;
  auto blocks = [&](int bm, int bn_) { return (int64_t)((M + bm - 1) / bm) * ((p->N + bn_ - 1) / bn_); };
  int bm = blocks(128, bn) >= 512 ? 128 : 64;
  if (bn == 160) return bm == 128 ? APTP_TILE_128x160 : APTP_TILE_64x160;
  if (bn == 128) return bm == 128 ? APTP_TILE_128x128 : APTP_TILE_64x128;
  return bm == 128 ? APTP_TILE_128x64 : APTP_TILE_64x64;
}

int fill_kparams(const AptpConvGemmParams* p, KParams& k) {
  APTP_CHECK(p && p->x && p->w && p->y, "conv_gemm: null pointer");
  APTP_CHECK(p->B > 0 && p->Hin > 0 && p->Win > 0 && p->Hout > 0 && p->Wout > 0, "conv_gemm: bad extent");
  APTP_CHECK(p->Cin > 0 && p->Cin % 8 == 0, "conv_gemm: Cin (%d) must be a positive multiple of 8", p->Cin);
  APTP_CHECK(p->ldx % 8 == 0 && p->ldx >= p->Cin, "conv_gemm: ldx (%lld) must be a multiple of 8 and >= Cin", (long long)p->ldx);
  APTP_CHECK(p->N > 0 && p->N % 8 == 0, "conv_gemm: N (%d) must be a positive multiple of 8", p->N);
  APTP_CHECK(p->cin_pad % BK == 0 && p->cin_pad >= p->Cin && p->cin_pad < p->Cin + BK, "conv_gemm: cin_pad (%d) != ceil(Cin/64)*64", p->cin_pad);
  APTP_CHECK(p->KH >= 1 && p->KW >= 1 && p->stride >= 1 && p->pad >= 0 && p->ups >= 0 && p->ups <= 2, "conv_gemm: bad filter geometry");
  APTP_CHECK(((uintptr_t)p->x % 16) == 0 && ((uintptr_t)p->w % 16) == 0 && ((uintptr_t)p->y % 8) == 0, "conv_gemm: pointer alignment");
  const int geglu = p->act == APTP_ACT_GEGLU;
  APTP_CHECK(!geglu || p->N % 32 == 0, "conv_gemm: GEGLU needs N %% 32 == 0");
  const int nout = geglu ? p->N / 2 : p->N;
  APTP_CHECK(p->ldy >= nout && p->ldy % 4 == 0, "conv_gemm: ldy (%lld) must be >= %d and a multiple of 4", (long long)p->ldy, nout);
  APTP_CHECK(!p->colgate || (p->gate_group > 0 && p->gate_B > 0 && nout % p->gate_group == 0), "conv_gemm: bad gate geometry");
  APTP_CHECK(!p->corr || (p->corr_B > 0 && p->Hout >= 2 && p->Wout >= 2), "conv_gemm: corr needs Hout,Wout >= 2");
  APTP_CHECK(!p->residual || (p->ldres % 4 == 0 && ((uintptr_t)p->residual % 8) == 0), "conv_gemm: residual alignment");
  APTP_CHECK(!p->depth || (p->depth_in && p->depth_B > 0 && p->lddin % 4 == 0), "conv_gemm: depth gate needs depth_in");
  APTP_CHECK(!p->rowbias || p->ld_rowbias % 4 == 0, "conv_gemm: ld_rowbias must be a multiple of 4");
  APTP_CHECK(p->split_k >= 1, "conv_gemm: split_k >= 1");
  const int64_t M64 = (int64_t)p->B * p->Hout * p->Wout;
  APTP_CHECK(M64 < (1ll << 31), "conv_gemm: M too large");
  // geometry consistency: every output pixel's centre tap must map inside the (upsampled) input
  const int sh = p->ups ? 1 : 0;   // ups 1 = nearest x2, ups 2 = zero-insertion x2 (both double the gather extent)
  const int HinE = p->Hin << sh, WinE = p->Win << sh;
  APTP_CHECK(p->Hout == (HinE + 2 * p->pad - p->KH) / p->stride + 1 && p->Wout == (WinE + 2 * p->pad - p->KW) / p->stride + 1,
             "conv_gemm: Hout/Wout inconsistent with input extent, filter, stride and padding");
  k.x = (const __bf16*)p->x; k.ldx = p->ldx;
  k.B = p->B; k.Hin = p->Hin; k.Win = p->Win; k.Cin = p->Cin; k.Hout = p->Hout; k.Wout = p->Wout;
  k.KH = p->KH; k.KW = p->KW; k.stride = p->stride; k.pad = p->pad; k.ups = sh; k.zins = p->ups == 2 ? 1 : 0;
  k.HinE = HinE; k.WinE = WinE;
  // fp32 parity instantiation (io_f32): 4-byte elements, a K-step is 32 channels (one 128-byte LDS row either way)
  const int f32 = p->io_f32 ? 1 : 0, es = f32 ? 4 : 2, ks = f32 ? 32 : BK;
  APTP_CHECK(!f32 || (p->out_f32 && !p->colstat_out && !p->gn_gamma && !p->prefetch),
             "conv_gemm: io_f32 (fp32 parity path) writes fp32 and has no column statistics / fused GroupNorm / prefetch");
  k.io_f32 = f32;
  k.x2 = (const __bf16*)p->x2; k.ldx2 = p->ldx2; k.Cin2 = p->x2 ? p->Cin2 : 0; k.ncc2 = p->x2 ? p->cin2_pad / ks : 0; k.x2_bytes = 0;
  if (p->x2) {
    APTP_CHECK(p->Cin2 > 0 && p->Cin2 % 8 == 0 && p->cin2_pad % BK == 0 && p->cin2_pad >= p->Cin2 && p->cin2_pad < p->Cin2 + BK,
               "conv_gemm: x2 needs Cin2 (%d) a positive multiple of 8 and cin2_pad (%d) == ceil(Cin2/64)*64", p->Cin2, p->cin2_pad);
    APTP_CHECK(p->ldx2 % 8 == 0 && p->ldx2 >= p->Cin2 && ((uintptr_t)p->x2 % 16) == 0, "conv_gemm: x2 row stride / alignment");
    APTP_CHECK(p->stride == 1 && p->ups == 0 && p->Hout == p->Hin && p->Wout == p->Win,
               "conv_gemm: x2 (second operand read at the output pixel) needs a stride-1, same-size convolution");
    const int64_t x2b = (((int64_t)p->B * p->Hout * p->Wout - 1) * p->ldx2 + p->Cin2) * es;
    APTP_CHECK(x2b < (1ll << 31), "conv_gemm: x2 larger than 2 GiB");
    k.x2_bytes = (int)x2b;
  }
  k.w = (const __bf16*)p->w; k.N = p->N; k.ncc = p->cin_pad / ks; k.nK = p->KH * p->KW * k.ncc + k.ncc2;
  k.Ktot = (int64_t)k.nK * ks;
  k.bias = p->bias; k.rowbias = p->rowbias; k.ld_rowbias = p->ld_rowbias;
  k.colgate = p->colgate; k.gate_group = p->gate_group; k.gate_B = p->gate_B;
  k.act = p->act; k.corr = p->corr; k.corr_B = p->corr_B;
  k.residual = (const __bf16*)p->residual; k.ldres = p->ldres;
  k.depth = p->depth; k.depth_B = p->depth_B; k.depth_in = (const __bf16*)p->depth_in; k.lddin = p->lddin;
  k.y = p->y; k.ldy = p->ldy; k.out_f32 = p->out_f32;
  k.split_k = p->split_k < k.nK ? p->split_k : k.nK;
  k.ws = (float*)p->workspace;
  k.M = (int)M64; k.HW = p->Hout * p->Wout; k.Nout = nout;
  k.ws_ld = p->N;
  APTP_CHECK(p->order >= 0 && p->order <= 3, "conv_gemm: order %d", p->order);
  if (p->order == 0) {
    // auto: partition over the XCDs whichever operand is larger; the other one is replicated in every XCD's L2
    const int64_t w_bytes = (int64_t)p->N * k.Ktot * 2;
    const int64_t a_bytes = (int64_t)p->B * p->Hin * p->Win * p->Cin * 2;
    k.order = w_bytes > a_bytes ? 1 : 2;
  } else {
    k.order = p->order - 1;   // 1 legacy, 2 weight-major, 3 activation-major
  }
  k.rstat_out = p->rowstat_out; k.rstat_slots = p->rowstat_slots;
  k.ln_stats = p->ln_stats; k.ln_slots = p->ln_slots; k.ln_colsum = p->ln_colsum; k.ln_eps = p->ln_eps;
  k.ln_invC = p->ln_C > 0 ? 1.0f / (float)p->ln_C : 0.f;
  APTP_CHECK(!p->rowstat_out || (!geglu && (!p->out_f32 || f32) && ((uintptr_t)p->rowstat_out % 16) == 0),
             "conv_gemm: rowstat_out needs a bf16, non-GEGLU output and an 8-byte aligned buffer");
  APTP_CHECK(!p->ln_stats || (p->ln_colsum && p->ln_slots > 0 && p->ln_C > 0 && p->KH == 1 && p->KW == 1 &&
                              p->ln_slots % 2 == 0 && ((uintptr_t)p->ln_stats % 16) == 0 && ((uintptr_t)p->ln_colsum % 16) == 0),
             "conv_gemm: folded LayerNorm needs ln_colsum [N], ln_slots > 0, ln_C > 0 and a 1x1 filter");
  k.counters = p->tile_counters;
  k.pf_ptr = ((uintptr_t)p->prefetch & 3) ? nullptr : (const char*)p->prefetch;   // dword loads
  k.pf_bytes = k.pf_ptr ? (p->prefetch_bytes < (1ll << 36) ? p->prefetch_bytes : (1ll << 36)) : 0;   // (line index fits an int)
  k.cstat_out = p->colstat_out; k.cstat_ld = p->colstat_ld;
  k.ustat_out = (long long*)p->ustat_out; k.ustat_unit = p->ustat_unit; k.ustat_units = p->ustat_units; k.ustat_nrep = p->ustat_nrep;
  APTP_CHECK(!p->ustat_out || (p->colstat_out && p->ustat_unit > 0 && p->ustat_units * p->ustat_unit >= nout && p->ustat_nrep >= 1 &&
                               (p->ustat_nrep & (p->ustat_nrep - 1)) == 0 && ((uintptr_t)p->ustat_out % 8) == 0),
             "conv_gemm: ustat_out goes with colstat_out; ustat_unit > 0, ustat_units >= ceil(N_out / unit), ustat_nrep a power of two");
  k.epi16 = !p->out_f32 && p->ldy % 8 == 0 && ((uintptr_t)p->y % 16) == 0 && nout % 8 == 0 &&
            (!p->residual || (p->ldres % 8 == 0 && ((uintptr_t)p->residual % 16) == 0)) &&
            (!p->depth || (p->lddin % 8 == 0 && ((uintptr_t)p->depth_in % 16) == 0));
  if (p->epilogue == 1) k.epi16 = 0;          // testing / tuning: force the accumulator-layout epilogue
  const int64_t xb = (((int64_t)p->B * p->Hin * p->Win - 1) * p->ldx + p->Cin) * es;
  const int64_t wb = (int64_t)p->N * k.Ktot * es;
  APTP_CHECK(xb < (1ll << 31) && wb < (1ll << 31), "conv_gemm: operand larger than 2 GiB");
  k.x_bytes = (int)xb; k.w_bytes = (int)wb;
  return APTP_OK;
}

template <int BM, int BN>
void launch_tile(const KParams& k, hipStream_t s) {
  const int tiles = ((k.M + BM - 1) / BM) * ((k.N + BN - 1) / BN);
  dim3 grid(tiles * k.split_k, 1, 1);
  if (k.io_f32) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, 2, 2, float>), grid, dim3(256), 0, s, k);
  else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, 2, 2>), grid, dim3(256), 0, s, k);
}

template <int BM, int BN, int STAGES>
void launch_tile_dma(const KParams& k, hipStream_t s) {
  const int tiles = ((k.M + BM - 1) / BM) * ((k.N + BN - 1) / BN);
  dim3 grid(tiles * k.split_k, 1, 1);
  hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 2, 2, STAGES>), grid, dim3(256), 0, s, k);
}

// 8-wave workgroups (512 threads, wave grid WM x WN): one weight tile shared by twice the rows
template <int BM, int BN, int WM, int WN, int STAGES = 2>
void launch_tile_dma8(const KParams& k, hipStream_t s) {
  const int tiles = ((k.M + BM - 1) / BM) * ((k.N + BN - 1) / BN);
  dim3 grid(tiles * k.split_k, 1, 1);
  hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, WM, WN, STAGES>), grid, dim3(512), 0, s, k);
}

// two K-tiles per barrier (KU = 2, see the kernel): 4 waves (2 x 2) or 8 waves
template <int BM, int BN, int WM, int WN, int STAGES>
void launch_tile_ku2(const KParams& k, hipStream_t s) {
  const int tiles = ((k.M + BM - 1) / BM) * ((k.N + BN - 1) / BN);
  dim3 grid(tiles * k.split_k, 1, 1);
  hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, WM, WN, STAGES, false, 2>), grid, dim3(WM * WN * 64), 0, s, k);
}

// intra-workgroup K split (KS = 2, see the kernel): 2 x (2 x 2) waves
template <int BM, int BN, int STAGES>
void launch_tile_ks2(const KParams& k, hipStream_t s) {
  const int tiles = ((k.M + BM - 1) / BM) * ((k.N + BN - 1) / BN);
  dim3 grid(tiles * k.split_k, 1, 1);
  hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, 2, 2, STAGES, false, 1, 2>), grid, dim3(512), 0, s, k);
}

// 8-wave ping-pong schedule (see the kernel): wave grid WM x WN, STAGES-deep ring
template <int BM, int BN, int WM, int WN, int STAGES>
void launch_tile_pp(const KParams& k, hipStream_t s) {
  const int tiles = ((k.M + BM - 1) / BM) * ((k.N + BN - 1) / BN);
  dim3 grid(tiles * k.split_k, 1, 1);
  hipLaunchKernelGGL((conv_gemm_dma_kernel<BM, BN, WM, WN, STAGES, true>), grid, dim3(512), 0, s, k);
}

}  // namespace

#ifdef APTP_STAMPS
extern "C" int aptp_debug_read_stamps(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int aptp_debug_read_epi(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_epi), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int aptp_debug_read_phases(unsigned long long* dst, int n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_phase), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int64_t aptp_conv_gemm_workspace_bytes(const AptpConvGemmParams* p) {
  if (!p || p->split_k <= 1) return 0;
  // covers both slab forms: [split_k][M][N] rows (separate reduce launch) and whole padded tiles in accumulator order
  // (in-kernel reduction)
  const int64_t M = (int64_t)p->B * p->Hout * p->Wout;
  const int t = pick_tile(p, (int)M);
  if (is_sk_tile(t))    // two fp32 slabs (head / tail partial tile) per workgroup
    return (int64_t)aptp_sk_cus() * 2 * kTiles[t].bm * kTiles[t].bn * (int64_t)sizeof(float);
  int64_t mp = M, np = p->N;
  if (t > 0 && t < kNumTiles) {
    mp = (M + kTiles[t].bm - 1) / kTiles[t].bm * kTiles[t].bm;
    np = ((int64_t)p->N + kTiles[t].bn - 1) / kTiles[t].bn * kTiles[t].bn;
  }
  return (int64_t)p->split_k * mp * np * (int64_t)sizeof(float);
}

extern "C" int aptp_conv_gemm_tiles(const AptpConvGemmParams* p) {
  if (!p) return 0;
  const int64_t M = (int64_t)p->B * p->Hout * p->Wout;
  const int t = pick_tile(p, (int)M);
  if (t <= 0 || t >= kNumTiles) return 0;
  return (int)(((M + kTiles[t].bm - 1) / kTiles[t].bm) * ((p->N + kTiles[t].bn - 1) / kTiles[t].bn));
}

extern "C" int aptp_conv_gemm_colstat_rows(const AptpConvGemmParams* p) {
  if (!p) return 0;
  const int t = pick_tile(p, p->B * p->Hout * p->Wout);
  if (t <= 0 || t >= kNumTiles) return 0;
  return kTiles[t].bm / (kTiles[t].nw / kTiles[t].wn);      // WTM: rows of one wave's block
}

extern "C" int aptp_conv_gemm_rowstat_slots(const AptpConvGemmParams* p) {
  if (!p) return 0;
  const int t = pick_tile(p, p->B * p->Hout * p->Wout);
  if (t <= 0 || t >= kNumTiles) return 0;
  return ((p->N + kTiles[t].bn - 1) / kTiles[t].bn) * kTiles[t].wn;
}

extern "C" int aptp_conv_gemm_suggest_split_k(const AptpConvGemmParams* p) {
  if (!p) return 1;
  const int M = p->B * p->Hout * p->Wout;
  const int t = pick_tile(p, M);
  const int64_t blocks = (int64_t)((M + kTiles[t].bm - 1) / kTiles[t].bm) * ((p->N + kTiles[t].bn - 1) / kTiles[t].bn);
  const int nK = p->KH * p->KW * (p->cin_pad / BK);
  if (blocks >= 256) return 1;
  int s = (int)((512 + blocks - 1) / blocks);
  const int max_s = nK / 8 > 0 ? nK / 8 : 1;   // keep >= 8 K-steps per slice
  if (s > max_s) s = max_s;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}

extern "C" int aptp_conv_gemm(const AptpConvGemmParams* p, aptp_stream_t stream) {
  KParams k;
  const int rc = fill_kparams(p, k);
  if (rc != APTP_OK) return rc;
  if (k.split_k > 1) APTP_CHECK(k.ws != nullptr && ((uintptr_t)k.ws % 16) == 0, "conv_gemm: split_k > 1 needs a 16B-aligned workspace");
  if (k.split_k == 1) k.counters = nullptr;
  k.gn_gamma = p->gn_gamma; k.gn_beta = p->gn_beta; k.gn_groups = p->gn_groups; k.gn_C = p->gn_C; k.gn_silu = p->gn_silu; k.gn_eps = p->gn_eps;
  if (k.gn_gamma) {
    APTP_CHECK(k.gn_beta && k.split_k > 1 && !k.counters && !k.out_f32 && k.act == APTP_ACT_NONE && !k.colgate && !k.corr && !k.residual && !k.depth
               && !k.rstat_out && !k.cstat_out && !k.ln_stats, "conv_gemm: gn_gamma needs a plain bf16 convolution split along K with a reduce launch");
    APTP_CHECK(k.gn_groups > 0 && k.gn_C > 0 && k.gn_C <= k.N && k.gn_C % k.gn_groups == 0 && k.gn_C % 8 == 0 && (k.gn_C / k.gn_groups) % 4 == 0
               && (int64_t)k.HW * (k.gn_C / k.gn_groups) <= 4LL * GN_RED_QPT * GN_RED_NT && k.ldy % 4 == 0,
               "conv_gemm: gn_gamma geometry (C=%d groups=%d HW=%d)", k.gn_C, k.gn_groups, k.HW);
  }
  hipStream_t s = (hipStream_t)stream;
  int t = pick_tile(p, k.M);
  if (t < 0 || t >= kNumTiles) { aptp_set_error("conv_gemm: unknown tile %d", t); return APTP_EINVAL; }
  if (k.io_f32 && t >= APTP_TILE_DMA_128x128) {
    aptp_set_error("conv_gemm: io_f32 (fp32 parity path) runs on the register-staged tiles 1..6 only (got tile %d)", t);
    return APTP_EINVAL;
  }
  if (t >= APTP_TILE_DMA_128x128 && !is_sk_tile(t) && (p->cin_pad * 2 > 8064 || (p->x2 && p->cin2_pad * 2 > 8064))) {
    // the LDS-DMA variants stream padding lanes from an 8 KiB zero page that must cover one channel row
    aptp_set_error("conv_gemm: LDS-DMA tiles need Cin <= 4032 (got cin_pad %d)", p->cin_pad);
    return APTP_EINVAL;
  }
  if (k.rstat_out) {
    const int slots = ((k.N + kTiles[t].bn - 1) / kTiles[t].bn) * kTiles[t].wn;
    if ((k.split_k != 1 && !k.counters) || k.rstat_slots != slots) {
      aptp_set_error("conv_gemm: rowstat_out needs split_k == 1 (or the in-kernel reduction) and rowstat_slots == aptp_conv_gemm_rowstat_slots() (%d), got split_k %d, slots %d",
                     slots, k.split_k, k.rstat_slots);
      return APTP_EINVAL;
    }
  }
  if (k.cstat_out) {
    const int wtm = kTiles[t].bm / (kTiles[t].nw / kTiles[t].wn), wtn = kTiles[t].bn / kTiles[t].wn;
    if (!k.epi16 || k.act == APTP_ACT_GEGLU || (k.split_k != 1 && !k.counters) || k.HW % wtm != 0 || wtn < 32 ||
        k.cstat_ld < k.Nout || ((uintptr_t)k.cstat_out % 8) != 0) {
      aptp_set_error("conv_gemm: colstat_out needs the coalesced bf16 epilogue, no GEGLU, split_k == 1 (or the in-kernel reduction), "
                     "Hout*Wout %% %d == 0 (aptp_conv_gemm_colstat_rows) and colstat_ld >= N", wtm);
      return APTP_EINVAL;
    }
  }
  if (k.act == APTP_ACT_GEGLU && (kTiles[t].bn == 160)) {
    aptp_set_error("conv_gemm: GEGLU cannot use a 160-wide tile");
    return APTP_EINVAL;
  }
  {
    const int tiles_m = (k.M + kTiles[t].bm - 1) / kTiles[t].bm, tiles_n = (k.N + kTiles[t].bn - 1) / kTiles[t].bn;
    APTP_CHECK((int64_t)tiles_m * tiles_n * k.split_k < (1ll << 31) && (int64_t)k.nK * (k.split_k + 1) < (1ll << 31), "conv_gemm: grid too large");
    k.fd_hw = make_fastdiv(k.HW); k.fd_wout = make_fastdiv(k.Wout);
    k.fd_tm = make_fastdiv(tiles_m); k.fd_tn = make_fastdiv(tiles_n); k.fd_sk = make_fastdiv(k.split_k);
    k.fd_perm = make_fastdiv(tiles_n * k.split_k); k.fd_tmn = make_fastdiv(tiles_m * tiles_n);
  }
  // plain linear layers take the lean kernel of the same tile shape (lin_gemm.hip); APTP_LIN=0 keeps them here (A/B timing, tests)
  static const bool lin_on = !(getenv("APTP_LIN") && getenv("APTP_LIN")[0] == '0');
  if (lin_on && p->epilogue != 2 && aptp_lin_eligible(k, t)) {
    const int rc2 = aptp_launch_lin(k, t, s);
    if (rc2 != APTP_OK) return rc2;
    APTP_LAUNCH_CHECK();
    return APTP_OK;
  }
  switch (t) {
    case APTP_TILE_128x128: launch_tile<128, 128>(k, s); break;
    case APTP_TILE_128x160: launch_tile<128, 160>(k, s); break;
    case APTP_TILE_64x128: launch_tile<64, 128>(k, s); break;
    case APTP_TILE_64x160: launch_tile<64, 160>(k, s); break;
    case APTP_TILE_128x64: launch_tile<128, 64>(k, s); break;
    case APTP_TILE_64x64: launch_tile<64, 64>(k, s); break;
    case APTP_TILE_DMA_128x128: launch_tile_dma<128, 128, 2>(k, s); break;
    case APTP_TILE_DMA_128x160: launch_tile_dma<128, 160, 2>(k, s); break;
    case APTP_TILE_DMA_64x128: launch_tile_dma<64, 128, 2>(k, s); break;
    case APTP_TILE_DMA_64x160: launch_tile_dma<64, 160, 2>(k, s); break;
    case APTP_TILE_DMA_128x64: launch_tile_dma<128, 64, 2>(k, s); break;
    case APTP_TILE_DMA_64x64: launch_tile_dma<64, 64, 2>(k, s); break;
    case APTP_TILE_DMA3_128x128: launch_tile_dma<128, 128, 3>(k, s); break;
    case APTP_TILE_DMA3_128x160: launch_tile_dma<128, 160, 3>(k, s); break;
    case APTP_TILE_DMA3_64x128: launch_tile_dma<64, 128, 3>(k, s); break;
    case APTP_TILE_DMA3_64x160: launch_tile_dma<64, 160, 3>(k, s); break;
    case APTP_TILE_DMA3_128x64: launch_tile_dma<128, 64, 3>(k, s); break;
    case APTP_TILE_DMA3_64x64: launch_tile_dma<64, 64, 3>(k, s); break;
    case APTP_TILE_DMA8_128x160: launch_tile_dma8<128, 160, 4, 2>(k, s); break;
    case APTP_TILE_DMA8_256x160: launch_tile_dma8<256, 160, 4, 2>(k, s); break;
    case APTP_TILE_DMA8_128x128: launch_tile_dma8<128, 128, 2, 4>(k, s); break;
    case APTP_TILE_DMA8_256x128: launch_tile_dma8<256, 128, 4, 2>(k, s); break;
    case APTP_TILE_DMA4_64x160: launch_tile_dma<64, 160, 4>(k, s); break;
    case APTP_TILE_DMA4_64x128: launch_tile_dma<64, 128, 4>(k, s); break;
    case APTP_TILE_DMA4_64x64: launch_tile_dma<64, 64, 4>(k, s); break;
    case APTP_TILE_DMA4_128x64: launch_tile_dma<128, 64, 4>(k, s); break;
    case APTP_TILE_DMA4_128x128: launch_tile_dma<128, 128, 4>(k, s); break;
    case APTP_TILE_DMA8R3_128x160: launch_tile_dma8<128, 160, 4, 2, 3>(k, s); break;
    case APTP_TILE_DMA8R4_128x160: launch_tile_dma8<128, 160, 4, 2, 4>(k, s); break;
    case APTP_TILE_DMA8R3_128x128: launch_tile_dma8<128, 128, 2, 4, 3>(k, s); break;
    case APTP_TILE_DMA8R4_128x128: launch_tile_dma8<128, 128, 2, 4, 4>(k, s); break;
    case APTP_TILE_DMA8R3_256x128: launch_tile_dma8<256, 128, 4, 2, 3>(k, s); break;
    case APTP_TILE_PP3_128x160: launch_tile_pp<128, 160, 4, 2, 3>(k, s); break;
    case APTP_TILE_PP4_128x160: launch_tile_pp<128, 160, 4, 2, 4>(k, s); break;
    case APTP_TILE_PP3_128x128: launch_tile_pp<128, 128, 2, 4, 3>(k, s); break;
    case APTP_TILE_PP4_128x128: launch_tile_pp<128, 128, 2, 4, 4>(k, s); break;
    case APTP_TILE_PP4_64x160: launch_tile_pp<64, 160, 4, 2, 4>(k, s); break;
    case APTP_TILE_PP4_64x128: launch_tile_pp<64, 128, 2, 4, 4>(k, s); break;
    case APTP_TILE_PP5_64x160: launch_tile_pp<64, 160, 4, 2, 5>(k, s); break;
    case APTP_TILE_PP4_128x64: launch_tile_pp<128, 64, 4, 2, 4>(k, s); break;
    case APTP_TILE_PP3_256x128: launch_tile_pp<256, 128, 4, 2, 3>(k, s); break;
    case APTP_TILE_PP3_128x256: launch_tile_pp<128, 256, 2, 4, 3>(k, s); break;
    case APTP_TILE_DMA6_64x64: launch_tile_dma<64, 64, 6>(k, s); break;
    case APTP_TILE_DMA8S_64x64: launch_tile_dma<64, 64, 8>(k, s); break;
    case APTP_TILE_DMA6_64x128: launch_tile_dma<64, 128, 6>(k, s); break;
    case APTP_TILE_DMA6_128x64: launch_tile_dma<128, 64, 6>(k, s); break;
    case APTP_TILE_KU2S4_64x64: launch_tile_ku2<64, 64, 2, 2, 4>(k, s); break;
    case APTP_TILE_KU2S6_64x64: launch_tile_ku2<64, 64, 2, 2, 6>(k, s); break;
    case APTP_TILE_KU2S4_64x128: launch_tile_ku2<64, 128, 2, 2, 4>(k, s); break;
    case APTP_TILE_KU2S6_64x128: launch_tile_ku2<64, 128, 2, 2, 6>(k, s); break;
    case APTP_TILE_KU2S4_128x64: launch_tile_ku2<128, 64, 2, 2, 4>(k, s); break;
    case APTP_TILE_KU2S4_128x128: launch_tile_ku2<128, 128, 2, 2, 4>(k, s); break;
    case APTP_TILE_KU2S4_64x160: launch_tile_ku2<64, 160, 2, 2, 4>(k, s); break;
    case APTP_TILE_KU2S4_128x160: launch_tile_ku2<128, 160, 4, 2, 4>(k, s); break;
    case APTP_TILE_KU2S4_128x128W8: launch_tile_ku2<128, 128, 2, 4, 4>(k, s); break;
    case APTP_TILE_KS2S3_64x64: launch_tile_ks2<64, 64, 3>(k, s); break;
    case APTP_TILE_KS2S4_64x64: launch_tile_ks2<64, 64, 4>(k, s); break;
    case APTP_TILE_KS2S3_64x128: launch_tile_ks2<64, 128, 3>(k, s); break;
    case APTP_TILE_KS2S4_64x128: launch_tile_ks2<64, 128, 4>(k, s); break;
    case APTP_TILE_KS2S3_128x64: launch_tile_ks2<128, 64, 3>(k, s); break;
    case APTP_TILE_KS2S3_128x128: launch_tile_ks2<128, 128, 3>(k, s); break;
    case APTP_TILE_SK_256x160: case APTP_TILE_SK_256x128: case APTP_TILE_SK_128x256:
    case APTP_TILE_SKL_256x160: case APTP_TILE_SKL_256x128: case APTP_TILE_SKL_128x256:
    case APTP_TILE_SK_128x160: case APTP_TILE_SKL_128x160: case APTP_TILE_SK_128x128: case APTP_TILE_SKL_128x128: {
      if (k.split_k > 1 && !k.counters) { aptp_set_error("conv_gemm: the stream-K tiles combine in-kernel: split_k > 1 needs tile_counters"); return APTP_EINVAL; }
      const int rc2 = aptp_launch_sk(k, t, s);
      if (rc2 != APTP_OK) return rc2;
      break;
    }
    case APTP_TILE_HALO_128x160:
    case APTP_TILE_HALO_128x128: {
      // whole image rows per tile, the patch (R+2) x (W+2) must fit the 264-row LDS image, no second operand
      const int W = p->Wout;
      if (!(p->KH == 3 && p->KW == 3 && p->stride == 1 && p->pad == 1 && p->ups == 0 && !p->x2 && W > 0 && 128 % W == 0 &&
            (p->Hout * W) % 128 == 0 && (128 / W + 2) * (W + 2) <= 264)) {
        aptp_set_error("conv_gemm: the halo tiles need a 3x3 / stride-1 / pad-1 convolution without x2 whose width divides 128 (16, 32 or 64)");
        return APTP_EINVAL;
      }
      const int tiles = ((k.M + 127) / 128) * ((k.N + kTiles[t].bn - 1) / kTiles[t].bn);
      if (t == APTP_TILE_HALO_128x160) hipLaunchKernelGGL((conv3x3_halo_kernel<160, 4>), dim3(tiles * k.split_k), dim3(512), 0, s, k);
      else hipLaunchKernelGGL((conv3x3_halo_kernel<128, 4>), dim3(tiles * k.split_k), dim3(512), 0, s, k);
      break;
    }
    default: aptp_set_error("conv_gemm: unknown tile %d", t); return APTP_EINVAL;
  }
  APTP_LAUNCH_CHECK();
  if (k.split_k > 1 && !k.counters) {
    if (k.gn_gamma) {
      hipLaunchKernelGGL(splitk_reduce_gn_kernel, dim3(k.gn_groups, k.B), dim3(GN_RED_NT), 0, s, k);
    } else {
      const int quads = (k.act == APTP_ACT_GEGLU) ? k.N / 8 : k.N / 4;
      const int64_t total = (int64_t)k.M * quads;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k);
    }
    APTP_LAUNCH_CHECK();
  }
  return APTP_OK;
}
