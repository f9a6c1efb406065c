// Persistent stream-K implicit GEMM on 256-row macro-tiles (gfx950): the tile engine's second outer loop.
//
// conv_gemm.hip launches one workgroup per (output tile, K-slice): a launch of the masked U-Net is 16-512 tiles of 64-128
// rows, every workgroup pays its own fill (address set-up, first operand round trip) and drain (epilogue), the last
// "wave" of tiles leaves most CUs idle, and 64 x 64 ... 128 x 160 tiles read 0.7-1.0 LDS cycles per MFMA cycle.  Here:
//   * ONE workgroup per CU (grid = CU count), 8 waves, a BM x BN = 256 x 160 / 256 x 128 / 128 x 256 macro-tile: a wave owns
//     64 x 80 (64 x 64) outputs, 0.45 LDS-read cycles per MFMA cycle;
//   * stream-K: the launch's work is the sequence of (tile, K-step) units in tile order; workgroup g takes the g-th of G
//     equal contiguous ranges, so every CU runs the same number of K-steps whatever the tile count.  A range covers the
//     tail of one tile, whole tiles, and the head of another; partial tiles are combined through fp32 slabs by the
//     workgroup that arrives LAST at the tile's counter (no waiting anywhere: any residency / dispatch order is
//     correct), slabs summed in range order: deterministic.  split_k == 1 turns the K split off (whole tiles only);
//   * two wave groups (waves 0-3 / 4-7, partners on each SIMD) run one barrier apart: a phase is an L slot (fragment
//     reads of this phase's 32-deep K half from LDS into registers + this wave's share of the LDS-DMA of a later tile)
//     and a C slot (the MF x NF MFMAs of that half, operands already in registers), so on every SIMD the matrix pipe
//     of one group runs beside the LDS / load path of the other (cdna_hip_programming.md, 256^2 8-phase template).
// LDS: 3 stages of (BM + BN) x 64 bf16 (XOR-swizzled 128-byte rows, as conv_gemm_dma_kernel), filled by LDS-DMA in
// half-tiles.  Gather (3x3 taps, stride, nearest / zero-insertion upsampling, second operand), zero padding through the
// zero page and the fused epilogues are the ones of conv_gemm.hip (conv_gemm_core.h).
//
// Schedule (S = 3 stages; slot j lies between barriers j and j+1; group 0 runs L(q) in slot 2q and C(q) in slot 2q+1, group 1
// one slot later; K-tile t = phases 2t, 2t+1, read from stage t % 3):
//   DMA   group 0 issues half-tile q+3 in L(q), group 1 half-tile q+4 (its slots are one later, the data is needed at
//         the same time): 1.5-2 K-tiles (78-104 KB) in flight per CU.
//   WAR   half-tiles of tile t overwrite tile t-3, whose last fragment reads (group 1, L(2t-5) in slot 4t-9) are retired
//         by the lgkmcnt(0) that opens its C slot 4t-8; the earliest DMA of tile t is issued in slot 4t-7 (group 1) /
//         4t-6 (group 0), i.e. behind the barrier that closes slot 4t-8.
//   RAW   tile t is first read by group 0 in slot 4t.  In slot 4t-1 (group 0: end of C(2t-1); group 1: end of L(2t-1))
//         every wave waits, with a counted vmcnt, for its own DMA of tile t -- the requests it has issued since (half 2t+2
//         for group 0; halves 2t+2, 2t+3 for group 1) stay in flight -- and the barrier that closes the slot publishes it.
#include "conv_gemm_core.h"

using namespace aptp_cg;

namespace {


struct SkParams {
  int G, T, tiles_m, tiles_n, nK, U;       // grid, output tiles, K-steps per tile, U = T * nK work units
  int whole;                               // 1: whole tiles only (no K split, no workspace)
  FastDiv fd_nk, fd_G, fd_U, fd_tm, fd_tn;
};

// counted wait on the vector-memory queue: at most `allowed` of this wave's newest requests may stay in flight.  The count
// is an immediate, so the value is rounded DOWN to one of the few that occur in steady state (waiting for more is always
// safe): four wave-uniform branches instead of a 17-way switch in every phase.
__device__ __forceinline__ void wait_vm(int allowed) {
  if (allowed >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (allowed >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (allowed >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// wait until this wave's DMA of K-tile t has landed; the requests issued since (half-tiles 2t+2 .. hidx-1, even ones cnt_h0
// instructions, odd ones cnt_h1) stay in flight.
__device__ __forceinline__ void wait_tile(int t, int hidx, int cnt_h0, int cnt_h1) {
  const int newer = hidx - (2 * t + 2);
  wait_vm(newer <= 0 ? 0 : ((newer + 1) >> 1) * cnt_h0 + (newer >> 1) * cnt_h1);
}

__device__ __forceinline__ void slot_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
}

// Partial tile: store this range's accumulators to its slab, draw a ticket; the last arriver sums all slabs of the tile
// in range order (its own included) and goes on to the epilogue.  Same publish / acquire protocol as splitk_combine
// (conv_gemm_core.h): write-through slab stores, drained by every wave, one relaxed agent-scope ticket, one acquire.
// Workgroup gg keeps two slabs: slot 0 for the tile its range begins in, slot 1 for a tile its range began before (then the
// range ends inside that tile: it is the range's last, partial one).  Writer and reader derive the slot from the range bounds.
template <int NT, int MF, int NF>
__device__ __forceinline__ bool sk_combine(const KParams& p, const SkParams& sk, f32x4 (&acc)[MF][NF], int tile, int g, int tid,
                                           int* lds_word) {
  constexpr int PER = MF * NF;
  const int x = tile * sk.nK, y = x + sk.nK;
  // contributors = ranges that intersect [x, y); range gg = [floor(U gg / G), floor(U (gg+1) / G))
  const int g_first = sk.fd_U.div((x + 1) * sk.G + sk.U - 1) - 1;      // max gg with begin(gg) <= x
  const int g_last = sk.fd_U.div(y * sk.G + sk.U - 1) - 1;              // max gg with begin(gg) < y
  float4* const ws = reinterpret_cast<float4*>(p.ws);
  auto slab_of = [&](int gg) {
    const int bg = sk.fd_G.div(sk.U * gg);
    return ws + ((int64_t)gg * 2 + (bg < x ? 1 : 0)) * (int64_t)(PER * NT);
  };
  {
    float4* const mine = slab_of(g);
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(mine, 0, PER * NT * 16, 0x00020000);
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), srsrc, ((i * NF + j) * NT + tid) * 16, 0, 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
  __syncthreads();
  if (tid == 0) *lds_word = __hip_atomic_fetch_add(p.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int ticket = *reinterpret_cast<volatile int*>(lds_word);
  if (ticket != g_last - g_first) return false;
  if (tid == 0) {
    __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every range has arrived
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int gg = g_first; gg <= g_last; ++gg) {
    const float4* src = slab_of(gg);
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

// RDC = false: a phase's fragments are read in its own L slot (beside the other group's MFMAs);
// RDC = true:  they are read one phase EARLIER, in the previous C slot, beside this wave's own MFMAs (two register sets): the
//              L slot then carries only the DMA issue, and no MFMA ever waits for an LDS round trip.
template <int BM, int BN, int WM, int WN, bool RDC>
__global__ __launch_bounds__(512) void conv_gemm_sk_kernel(const KParams p, const SkParams sk) {
  constexpr int NW = 8, NT = 512, RPP = 64, S = 3;
  static_assert(WM * WN == NW, "8 waves");
  constexpr int WTM = BM / WM, WTN = BN / WN, MF = WTM / 16, NF = WTN / 16;
  constexpr int A_PASS = BM / RPP, B_PASS = (BN + RPP - 1) / RPP;
  constexpr int A_H0 = A_PASS / 2, B_H0 = B_PASS / 2;          // row passes of a K-tile issued with its first half
  static_assert(BM % RPP == 0 && A_PASS >= 2 && B_PASS >= 2 && WTM % 16 == 0 && WTN % 16 == 0, "tile shape");
  constexpr int STAGE = (BM + BN) * BK;                          // bf16 elements per stage: [BM rows | BN rows] x 64
  static_assert(S * STAGE * 2 + 256 <= 160 * 1024, "LDS");
  static_assert(NW * 16 * (WTN + 4) * 4 <= S * STAGE * 2, "epilogue transpose buffer");
  constexpr unsigned OOB = 0x80000000u;                          // buffer offset past every extent: the load returns zeros

  __shared__ __attribute__((aligned(16))) __bf16 smem[S * STAGE + 128];     // (+ one 256-byte scratch row for prefetch_next)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: LDS-DMA bases stay scalar
  const int grp = wave >> 2;                                     // waves w and w + 4 share a SIMD
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 15, fq = lane >> 4;

  prefetch_next(p, tid, NT, reinterpret_cast<unsigned*>(smem + S * STAGE));

  // ---- this workgroup's range of (tile, K-step) units -----------------------------------------------------------------
  // rank in the work order: the workgroups of one XCD (linear id % 8, round-robin dispatch) take adjacent ranges, so an XCD's
  // L2 sees a contiguous band of tiles (neighbours share the activation rows or the weight rows, by p.order)
  int g;
  {
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3, qq = sk.G >> 3, r = sk.G & 7;
    g = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + j;
  }
  int u_begin, u_end;
  if (sk.whole) { u_begin = sk.fd_G.div(sk.T * g) * sk.nK; u_end = sk.fd_G.div(sk.T * (g + 1)) * sk.nK; }
  else { u_begin = sk.fd_G.div(sk.U * g); u_end = sk.fd_G.div(sk.U * (g + 1)); }

  // ---- per-thread constants of the LDS-DMA stream -----------------------------------------------------------------------
  // Operands go global -> LDS by buffer loads with the LDS flag: a lane's address is a 32-bit offset register (recomputed
  // only when the filter tap changes) plus a SCALAR offset that carries the K position, so a K-step costs the stream one
  // scalar add per operand; an out-of-range offset returns zeros (conv padding, rows past M, the ragged last channel
  // chunk) -- no zero page, no 64-bit pointer arithmetic, no selects on the common path.
  const int rowbase = tid >> 3;
  const int schunk = (tid & 7) ^ ((rowbase >> 1) & 7);      // source chunk that lands in this lane's LDS slot
  const bool tail_bad = ((p.ncc - 1) * BK + schunk * 8) >= p.Cin;     // this lane's chunk of the last channel step is padding
  const bool tail_bad2 = ((p.ncc2 - 1) * BK + schunk * 8) >= p.Cin2;
  const int nk_taps = p.KH * p.KW * p.ncc;
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x2src = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x2 ? p.x2 : p.x), 0, p.x2 ? p.x2_bytes : 0, 0x00020000);
  // DMA instructions this wave issues per half-tile (the last weight pass may be partial: rows [RPP*(B_PASS-1), BN))
  int cnt_h0 = A_H0, cnt_h1 = A_PASS - A_H0;
#pragma unroll
  for (int i = 0; i < B_PASS; ++i) {
    const int on = (wave * 8 + RPP * i < BN) ? 1 : 0;
    if (i < B_H0) cnt_h0 += on; else cnt_h1 += on;
  }
  // half-tiles the DMA stream runs ahead of the phase (group 1's slots are one later; the data is needed at the same time)
  const int ahead = (RDC ? 4 : 3) + grp;

  typedef __attribute__((address_space(3))) void* lds_ptr;

  for (int u = u_begin; u < u_end;) {
    const int tile = sk.fd_nk.div(u);
    const int k0 = u - tile * sk.nK;
    const int n = (sk.nK - k0) < (u_end - u) ? (sk.nK - k0) : (u_end - u);       // K-tiles of this segment
    int tm, tn;
    if (p.order == 1) { tn = sk.fd_tm.div(tile); tm = tile - tn * sk.tiles_m; }     // weight-major: M-tile fastest
    else { tm = sk.fd_tn.div(tile); tn = tile - tm * sk.tiles_n; }                  // activation-major: N-tile fastest
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- gather coordinates of this thread's rows ----------------------------------------------------------------------
    int a_iy0[A_PASS], a_ix0[A_PASS], a_pix0[A_PASS], a_m[A_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int m = m0 + rowbase + RPP * i;
      if (m < p.M) {
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
    int l_cc = 0, l_ky = 0, l_kx = 0;          // (l_ky == KH: the x2 segment after the filter taps)
    if (k0 != 0) {
      const int l_tap = k0 / p.ncc;
      l_cc = k0 - l_tap * p.ncc;
      l_ky = l_tap / p.KW;
      l_kx = l_tap - l_ky * p.KW;
    }
    if (k0 >= nk_taps) { l_ky = p.KH; l_kx = 0; l_cc = k0 - nk_taps; }

    unsigned a_off[A_PASS];                // this lane's byte offsets of the current tap (channel step 0), OOB = padding
    auto set_tap = [&]() {
      if (l_ky >= p.KH) {                  // wave-uniform: the second operand, read at the output pixel itself
#pragma unroll
        for (int i = 0; i < A_PASS; ++i)
          a_off[i] = a_m[i] >= 0 ? ((unsigned)a_m[i] * (unsigned)p.ldx2) * 2u + (unsigned)schunk * 16u : OOB;
        return;
      }
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        int iy = a_iy0[i] + l_ky, ix = a_ix0[i] + l_kx;
        const bool ok = (unsigned)iy < (unsigned)p.HinE && (unsigned)ix < (unsigned)p.WinE && !(p.zins & (iy | ix));
        iy >>= p.ups; ix >>= p.ups;
        a_off[i] = ok ? ((unsigned)(a_pix0[i] + iy * p.Win + ix) * (unsigned)p.ldx) * 2u + (unsigned)schunk * 16u : OOB;   // < 2^31
      }
    };
    unsigned b_off[B_PASS];
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      int nn = n0 + rowbase + RPP * i;
      nn = nn < p.N ? nn : p.N - 1;            // columns past N accumulate garbage that is never stored
      b_off[i] = (unsigned)(((int64_t)nn * p.Ktot + schunk * 8) * 2);     // < 2^31 (w_bytes, checked on the host)
    }
    set_tap();

    // ---- the DMA stream: half-tiles 0 .. 2n-1 of this segment, in order ------------------------------------------------------
    int hidx = 0, ist = 0;                 // next half-tile to request, stage of its K-tile
    int s_b = k0 * (BK * 2);               // scalar byte offsets: K position in the weight rows / channel step in the pixel rows
    auto issue_half = [&]() {
      if (hidx >= 2 * n) return;           // (wave-uniform)
      __bf16* const st = smem + ist * STAGE;
      const bool seg2 = l_ky >= p.KH;
      const bool mask_tail = (l_cc == (seg2 ? p.ncc2 : p.ncc) - 1) && (seg2 ? tail_bad2 : tail_bad);
      const int s_a = l_cc * (BK * 2);
      const bool h0 = (hidx & 1) == 0;
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        if ((i < A_H0) == h0) {
          const unsigned off = mask_tail ? OOB : a_off[i];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(seg2 ? x2src : xsrc, (lds_ptr)(st + (wave * 8 + RPP * i) * BK), 16, off, s_a, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) {
        // (a local copy: with the captured array element passed straight to the builtin, hipcc's HOST pass silently drops the
        // kernel's launch stub -- the library then fails to load with an undefined __device_stub__ symbol)
        const unsigned boff = b_off[i];
        if ((i < B_H0) == h0 && wave * 8 + RPP * i < BN)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_ptr)(st + (BM + wave * 8 + RPP * i) * BK), 16, boff, s_b, 0, 0);
      }
      if (!h0) {                           // the K-tile is complete: next stage, next K position
        ist = ist + 1 == S ? 0 : ist + 1;
        s_b += BK * 2;
        if (++l_cc == (seg2 ? p.ncc2 : p.ncc)) {   // next tap (wave-uniform branch)
          l_cc = 0;
          if (++l_kx == p.KW || seg2) { l_kx = 0; ++l_ky; }
          set_tap();
        }
      }
      ++hidx;
    };
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 af0[MF], wf0[NF], af1[MF], wf1[NF];
    auto read_frags = [&](int stage, int s, bf16x8 (&af)[MF], bf16x8 (&wf)[NF]) {
      const __bf16* const st = smem + stage * STAGE;
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int r = wm * WTM + i * 16 + frow;
        const int sw = (s * 4 + fq) ^ ((r >> 1) & 7);
        af[i] = *reinterpret_cast<const bf16x8*>(st + r * BK + sw * 8);
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int r = wn * WTN + j * 16 + frow;
        const int sw = (s * 4 + fq) ^ ((r >> 1) & 7);
        wf[j] = *reinterpret_cast<const bf16x8*>(st + (BM + r) * BK + sw * 8);
      }
    };
    auto mfmas = [&](const bf16x8 (&af)[MF], const bf16x8 (&wf)[NF]) {
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: the stream's head start, tile 0 landed ---------------------------------------------------------------------
    for (int c = 0; c < ahead; ++c) issue_half();
    wait_tile(0, hidx, cnt_h0, cnt_h1);
    slot_barrier();
    if (grp == 1) slot_barrier();               // stagger: group 1 runs one slot behind

    int cur = 0;
    if constexpr (RDC) {
      // reads one phase ahead.  Tile t is first read in C(2t-1) (group 0: slot 4t-1), so it is published in slot 4t-2: group 0
      // waits at the end of L(2t-1), group 1 at the end of C(2t-2); tile t+3 overwrites tile t from slot 4t+3 (group 1's L(2t+1))
      // on, behind the lgkmcnt(0) that closes group 1's C(2t) in slot 4t+2.
      read_frags(0, 0, af0, wf0);
      for (int tt = 0; tt < n; ++tt) {
        const int nxt = cur + 1 == S ? 0 : cur + 1;
        // ---- phase 0
        issue_half();
        slot_barrier();
        __builtin_amdgcn_s_setprio(1);
        read_frags(cur, 1, af1, wf1);
        mfmas(af0, wf0);
        __builtin_amdgcn_s_setprio(0);
        if (grp == 1) wait_tile(tt + 1, hidx, cnt_h0, cnt_h1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot_barrier();
        // ---- phase 1
        issue_half();
        if (grp == 0) wait_tile(tt + 1, hidx, cnt_h0, cnt_h1);
        slot_barrier();
        __builtin_amdgcn_s_setprio(1);
        if (tt + 1 < n) read_frags(nxt, 0, af0, wf0);
        mfmas(af1, wf1);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot_barrier();
        cur = nxt;
      }
    } else {
      for (int tt = 0; tt < n; ++tt) {
        // ---- phase 0: first 32-deep half of K-tile tt
        read_frags(cur, 0, af0, wf0);
        issue_half();
        slot_barrier();
        __builtin_amdgcn_s_setprio(1);
        mfmas(af0, wf0);
        __builtin_amdgcn_s_setprio(0);
        slot_barrier();
        // ---- phase 1: second half; its last slot before the barrier that opens tile tt + 1 retires this wave's DMA of that tile
        read_frags(cur, 1, af0, wf0);
        issue_half();
        if (grp == 1) wait_tile(tt + 1, hidx, cnt_h0, cnt_h1);
        slot_barrier();
        __builtin_amdgcn_s_setprio(1);
        mfmas(af0, wf0);
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) wait_tile(tt + 1, hidx, cnt_h0, cnt_h1);
        slot_barrier();
        cur = cur + 1 == S ? 0 : cur + 1;
      }
    }
    if (grp == 0) slot_barrier();               // group 0 pairs group 1's last barrier: both groups leave together

    // ---- partial tile -> slab + ticket; the last arriver (or the owner of a whole tile) runs the epilogue -------------------
    bool emit = true;
    if (n != sk.nK) emit = sk_combine<NT, MF, NF>(p, sk, acc, tile, g, tid, reinterpret_cast<int*>(smem));
    if (emit) {
      float ln_mean[MF], ln_rstd[MF];
#pragma unroll
      for (int i = 0; i < MF; ++i) { ln_mean[i] = 0.f; ln_rstd[i] = 1.f; }
      LnRaw<MF> ln_raw;
      ln_rows_issue<MF, WTM>(p, m0, wm, lane, ln_raw, true);
      ln_rows_finish<MF>(p, lane, ln_raw, ln_mean, ln_rstd, true);
      run_epilogue<NW, MF, NF, WTM, WTN, WN>(p, acc, m0, n0, tn, wm, wn, lane, wave, ln_mean, ln_rstd, smem);
    }
    __syncthreads();                             // the stages are free again (epilogue transposes, ticket word)
    u += n;
  }
}

int g_sk_cus = 0;

template <int BM, int BN, int WM, int WN, bool RDC>
int launch_sk(const KParams& k, hipStream_t s) {
  SkParams sk;
  sk.tiles_m = (k.M + BM - 1) / BM; sk.tiles_n = (k.N + BN - 1) / BN;
  sk.T = sk.tiles_m * sk.tiles_n; sk.nK = k.nK;
  const int64_t U = (int64_t)sk.T * sk.nK;
  const int cus = aptp_sk_cus();
  sk.whole = k.split_k > 1 ? 0 : 1;
  const int64_t units = sk.whole ? sk.T : U;
  sk.G = (int)(units < cus ? units : cus);
  APTP_CHECK(U * (sk.G + 1) < (1ll << 31) && sk.T <= 65536, "conv_gemm: stream-K work list too large (%d tiles x %d K-steps)", sk.T, sk.nK);
  sk.U = (int)U;
  APTP_CHECK(sk.whole || (k.ws && k.counters), "conv_gemm: the stream-K tiles with split_k > 1 need workspace and tile_counters");
  sk.fd_nk = make_fastdiv(sk.nK); sk.fd_G = make_fastdiv(sk.G); sk.fd_U = make_fastdiv(sk.U);
  sk.fd_tm = make_fastdiv(sk.tiles_m); sk.fd_tn = make_fastdiv(sk.tiles_n);
  hipLaunchKernelGGL((conv_gemm_sk_kernel<BM, BN, WM, WN, RDC>), dim3(sk.G), dim3(512), 0, s, k, sk);
  return APTP_OK;
}

}  // namespace

namespace aptp_cg {

int aptp_sk_cus() {
  if (g_sk_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_sk_cus = n;
  }
  return g_sk_cus;
}

int aptp_launch_sk(const KParams& k, int tile, hipStream_t s) {
  switch (tile) {
    case APTP_TILE_SK_256x160: return launch_sk<256, 160, 4, 2, true>(k, s);
    case APTP_TILE_SK_256x128: return launch_sk<256, 128, 4, 2, true>(k, s);
    case APTP_TILE_SK_128x256: return launch_sk<128, 256, 2, 4, true>(k, s);
    case APTP_TILE_SKL_256x160: return launch_sk<256, 160, 4, 2, false>(k, s);
    case APTP_TILE_SKL_256x128: return launch_sk<256, 128, 4, 2, false>(k, s);
    case APTP_TILE_SKL_128x256: return launch_sk<128, 256, 2, 4, false>(k, s);
    case APTP_TILE_SK_128x160: return launch_sk<128, 160, 4, 2, true>(k, s);
    case APTP_TILE_SKL_128x160: return launch_sk<128, 160, 4, 2, false>(k, s);
    case APTP_TILE_SK_128x128: return launch_sk<128, 128, 4, 2, true>(k, s);
    case APTP_TILE_SKL_128x128: return launch_sk<128, 128, 4, 2, false>(k, s);
    default: aptp_set_error("conv_gemm: unknown stream-K tile %d", tile); return APTP_EINVAL;
  }
}

}  // namespace aptp_cg
