// Weight gradient of a "same" convolution (3x3 or 1x1 / linear; 3x3 also with stride 2 -- the down-samplers -- or a folded
// nearest-x2 up-sample -- the up-samplers) for the expert fine-tune step (trainer.py:1616 loss.backward through
// F.conv2d / F.linear): dW[n][tap][c] = sum_m dy[m][n] * x[pix(m, tap)][c].
//
// The contraction runs over PIXELS, and both operands are stored pixel-major (channels contiguous), so the MFMA's K
// dimension is the strided one.  gfx950's transposed LDS read (ds_read_b64_tr_b16: a 4-row x 16-column block delivered
// column-major) makes that free: the tiles are copied to LDS exactly as they lie in memory ([pixel][channel], 16-byte
// chunks) and read back K-major.  No im2col, no transposed copies of dy or x in HBM.
//
// Workgroup = one (64 output channels n) x (64 input channels c) block of dW for ALL taps and one slice of the pixel range.
// Per 32-pixel K-step it stages the dy tile [32][64] and, for a 3x3 filter, the input HALO of those 32 pixels
// ((R+2) x (Wt+2) pixels, R = rows covered, Wt = min(W, 32)) once; the nine taps are nine shifted row offsets into that
// one LDS image (the shift is in the pixel dimension, where a transposed read takes any row address), so x is read 1.06x
// instead of 9x and dy once.  4 waves as 2(n) x 2(c), wave tile 32 x 32, mfma_f32_16x16x32_bf16 with A = x^T fragment
// (rows = c), B = dy^T fragment (columns = n): a lane ends up with 4 consecutive c of one n -> 16-byte stores into the
// packed-weight order [n][tap][c].  Global loads of step s+1 are in flight during the MFMAs of step s (register staging,
// two LDS buffers, one barrier per step).  The pixel range is split over `split_m` workgroups (fp32 slabs, summed by the
// caller in a fixed order: deterministic).
//
// Resampling convolutions (AptpWgradParams.stride / ups; H, W are always the OUTPUT map, the one dy lives on).  Stride 2:
// the halo of a step's R x Wt output pixels is (2R+1) x (2Wt+1) INPUT pixels (up to 195 rows: a template instantiation with a
// larger x image), output pixel (r, c) sits at halo (2r, 2c), so its tap (ky, kx) is halo row (2r + ky) * hw2 + 2c + kx and the
// second 4-pixel block of a fragment lies 8 halo rows on instead of 4.  Nearest-x2: the halo lives on the up-sampled grid
// ((R+2) x (Wt+2), exactly the stride-1 geometry) and each halo entry is FETCHED from input pixel (iy >> 1, ix >> 1), zero
// outside the up-sampled extent -- the gather of the forward kernel, nothing materialised.  (Round 2 ran these six layers as
// GEMMs on transposed torch copies, or as four stride-1 calls on parity planes that compute 36 tap results to use 9.)
#include "aptp_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct WgK {
  const __bf16* x; int64_t ldx; const __bf16* dy; int64_t lddy; float* dw;
  int B, H, W, C, N, M, HW, split, nsteps, tiles_n, tiles_c;
  int Wt, R, hw2;      // halo geometry (3x3): tile width, rows per step, halo width (Wt + 2; stride 2: 2 Wt + 1)
  int Hin, Win, ups;   // input map extent; ups = 1: halo coordinates are on the nearest-x2 up-sampled grid
  int ldw;             // row length of dw's c dimension
  float* db;           // optional [split][N]: column sums of dy over the slice (written by the workgroups of c-block 0)
  int64_t slab_stride, db_stride;
  int first_block;     // conv_wgrad_many_kernel: the item's first workgroup in the batch's grid
  int pad_;
};

#define APTP_WGRAD_SUB_DEFAULT 2      // measured on the fine-tune step: 38.3 / 37.7 / 38.0 ms with 1 / 2 / 4
constexpr int PITCH = 72;          // bf16 elements per LDS row (64 + 8: spreads the 4-row transposed blocks over the banks)
constexpr int XROWS = 104;         // halo rows per buffer: (R+2) * (Wt+2) <= 102
constexpr int XROWS_S2 = 200;      // stride 2: (2R+1) * (2Wt+1) <= 195

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* lds_row_q_col_4p, int plus4_rows_elems) {
  // two 4-row x 16-column transposed blocks -> the 8 consecutive k of one MFMA operand lane
  typedef __attribute__((address_space(3))) s16x4* lp;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds_row_q_col_4p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds_row_q_col_4p + plus4_rows_elems));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// SUB: 32-pixel sub-steps per barrier.  The 3x3 form stages one halo per 32 pixels (SUB = 1); a 1x1 / linear layer has no halo
// and only 4 MFMAs per wave and sub-step, so it takes 64 pixels per barrier (SUB = 2; 4 fits too -- 74 KB of LDS -- and measures the same).
template <int TAPS, int SUB, int STRIDE>
__device__ __forceinline__ void wgrad_body(const WgK& p, int bid) {
  static_assert(TAPS == 1 || SUB == 1, "sub-steps only without a halo");
  static_assert(STRIDE == 1 || TAPS == 9, "stride 2: the 3x3 down-samplers");
  constexpr int XR = STRIDE == 2 ? XROWS_S2 : XROWS;
  __shared__ __attribute__((aligned(16))) __bf16 dys[2][32 * SUB * PITCH];
  __shared__ __attribute__((aligned(16))) __bf16 xs[2][(TAPS == 9 ? XR : 32 * SUB) * PITCH];
  constexpr int XCH = TAPS == 9 ? (XR * 8 + 255) / 256 : SUB;         // 16-byte x chunks per thread and step

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1;
  const int slice = bid % p.split; bid /= p.split;
  const int tc = bid % p.tiles_c, tn = bid / p.tiles_c;
  const int n0 = tn * 64, c0 = tc * 64;
  const int s_begin = (int)(((int64_t)p.nsteps * slice) / p.split), s_end = (int)(((int64_t)p.nsteps * (slice + 1)) / p.split);

  // ---- staging registers -------------------------------------------------------------------------------------------------
  // bias gradient: the thread's dy chunk is 8 output channels of one pixel; it stages the same chunk column every step, so it
  // keeps their running sums (c-block 0 only: every c-block stages the same dy tiles)
  const bool want_db = p.db != nullptr && tc == 0;
  float dbacc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dbacc[e] = 0.f;
  u32x4 rdy[SUB], rx[XCH];
  const int dj = tid >> 3, dch = tid & 7;          // dy tile: row (pixel of the sub-step), 16-byte chunk
  const int xrows = TAPS == 9 ? (STRIDE * (p.R - 1) + 3) * p.hw2 : 32;
  auto load_step = [&](int step) {
    const int m0 = step * 32 * SUB;
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      const int m = m0 + sb * 32 + dj, n = n0 + dch * 8;
      rdy[sb] = (u32x4){0u, 0u, 0u, 0u};
      if (m < p.M && n < p.N) rdy[sb] = *reinterpret_cast<const u32x4*>(p.dy + (int64_t)m * p.lddy + n);
    }
    if (TAPS == 9) {
      const int b = m0 / p.HW, rem = m0 - b * p.HW;
      const int y0 = rem / p.W, x0 = rem - y0 * p.W;
#pragma unroll
      for (int i = 0; i < XCH; ++i) {
        const int id = tid + 256 * i, hr = id >> 3, ch = id & 7;
        rx[i] = (u32x4){0u, 0u, 0u, 0u};
        if (hr < xrows) {
          const int hy = hr / p.hw2, hx = hr - hy * p.hw2;
          // halo coordinates on the (up-sampled) input grid; with ups the source pixel is (iy >> 1, ix >> 1)
          const int iy = STRIDE * y0 - 1 + hy, ix = STRIDE * x0 - 1 + hx, c = c0 + ch * 8;
          if ((unsigned)iy < (unsigned)(p.Hin << p.ups) && (unsigned)ix < (unsigned)(p.Win << p.ups) && c < p.C)
            rx[i] = *reinterpret_cast<const u32x4*>(p.x + ((int64_t)(b * p.Hin + (iy >> p.ups)) * p.Win + (ix >> p.ups)) * p.ldx + c);
        }
      }
    } else {
#pragma unroll
      for (int sb = 0; sb < SUB; ++sb) {
        const int m = m0 + sb * 32 + dj, c = c0 + dch * 8;
        rx[sb] = (u32x4){0u, 0u, 0u, 0u};
        if (m < p.M && c < p.C) rx[sb] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)m * p.ldx + c);
      }
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      *reinterpret_cast<u32x4*>(&dys[buf][(sb * 32 + dj) * PITCH + dch * 8]) = rdy[sb];
      if (want_db) {
        float f[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(&rdy[sb]), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) dbacc[e] += f[e];
      }
    }
    if (TAPS == 9) {
#pragma unroll
      for (int i = 0; i < XCH; ++i) {
        const int id = tid + 256 * i, hr = id >> 3, ch = id & 7;
        if (hr < xrows) *reinterpret_cast<u32x4*>(&xs[buf][hr * PITCH + ch * 8]) = rx[i];
      }
    } else {
#pragma unroll
      for (int sb = 0; sb < SUB; ++sb) *reinterpret_cast<u32x4*>(&xs[buf][(sb * 32 + dj) * PITCH + dch * 8]) = rx[sb];
    }
  };

  // ---- fragment addressing (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3) -------------
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int j = 8 * g + q;                                   // pixel of the step this lane addresses (and j + 4)
  int xrow = j;                                              // its row in the x image for tap (0, 0)
  if (TAPS == 9) xrow = STRIDE * ((j / p.Wt) * p.hw2 + (j % p.Wt));    // (halo coordinates of the pixel's top-left neighbour)
  const int x_off = xrow * PITCH + wc * 32 + 4 * pp;         // + cf*16 + tap shift * PITCH
  const int d_off = j * PITCH + wn * 32 + 4 * pp;            // + nf*16

  f32x4 acc[TAPS][2][2];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) acc[t][a][b2] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (s_begin < s_end) {
    load_step(s_begin);
    int buf = 0;
    for (int s = s_begin; s < s_end; ++s) {
      store_step(buf);
      __syncthreads();
      if (s + 1 < s_end) load_step(s + 1);
#pragma unroll
      for (int sb = 0; sb < SUB; ++sb) {
        const int sub_off = sb * 32 * PITCH;
        bf16x8 dfr[2];
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) dfr[nf] = tr_frag(&dys[buf][sub_off + d_off + nf * 16], 4 * PITCH);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int shift = TAPS == 9 ? ((t / 3) * p.hw2 + (t % 3)) * PITCH : 0;
#pragma unroll
          for (int cf = 0; cf < 2; ++cf) {
            const bf16x8 xf = tr_frag(&xs[buf][sub_off + x_off + shift + cf * 16], 4 * STRIDE * PITCH);
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
              acc[t][cf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, dfr[nf], acc[t][cf][nf], 0, 0, 0);
          }
        }
      }
      buf ^= 1;
    }
  }

  // ---- bias-gradient slab: fold the 32 threads that share a chunk column (fixed order), one value per output channel ---------
  if (want_db) {
    __syncthreads();                                             // the last step's fragment reads are done: reuse dys as fp32 scratch
    float* red = reinterpret_cast<float*>(&dys[0][0]);           // 32 x 64 floats = 8 KiB <= sizeof(dys)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[dj * 64 + dch * 8 + e] = dbacc[e];
    __syncthreads();
    if (tid < 64 && n0 + tid < p.N) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) v += red[r * 64 + tid];
      p.db[(int64_t)slice * p.db_stride + n0 + tid] = v;
    }
  }

  // ---- store: lane owns c = cbase + 4*(lane/16) .. +3 of n = nbase + lane%16 -------------------------------------------------
  float* out = p.dw + (int64_t)slice * p.slab_stride;
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    const int n = n0 + wn * 32 + nf * 16 + (lane & 15);
    if (n >= p.N) continue;
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
      const int c = c0 + wc * 32 + cf * 16 + 4 * (lane >> 4);
      if (c >= p.C) continue;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        float4 o; o.x = acc[t][cf][nf][0]; o.y = acc[t][cf][nf][1]; o.z = acc[t][cf][nf][2]; o.w = acc[t][cf][nf][3];
        *reinterpret_cast<float4*>(out + ((int64_t)n * TAPS + t) * p.ldw + c) = o;
      }
    }
  }
}

template <int TAPS, int SUB, int STRIDE = 1>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgK p) {
  wgrad_body<TAPS, SUB, STRIDE>(p, (int)blockIdx.x);
}

// Every stride-1 weight gradient of a backward pass as ONE launch (per filter size).  Issued layer by layer, each launch has to
// fill the chip on its own, so a 176 x 176 projection at 16,384 pixels was cut into 57 pixel slices (9 dW tiles x 57 = 513
// workgroups of 9 steps, 57 fp32 slabs to fold afterwards); batched, the grid is every layer's tiles and a slice is as long
// as a workgroup is worth starting for (~2,048 pixels): 8 slices there, none for the large weights, whose gradients are
// then written straight into the optimizer's buffer.  Descriptors (WgK, filled by aptp_conv_wgrad_many_fill) and the
// workgroup -> item map live in device memory; both loads are wave-uniform.
template <int TAPS, int SUB>
__global__ __launch_bounds__(256) void conv_wgrad_many_kernel(const WgK* __restrict__ items, const int* __restrict__ block_item) {
  const int it = block_item[blockIdx.x];
  const WgK p = items[it];
  wgrad_body<TAPS, SUB, 1>(p, (int)blockIdx.x - p.first_block);
}

}  // namespace

#define ALIGN16(p) (((uintptr_t)(p) % 16) == 0)

extern "C" int aptp_conv_wgrad_supported(const AptpWgradParams* p) {
  if (!p || (p->KH != 1 && p->KH != 3) || p->KH != p->KW) return 0;
  if (p->C % 8 || p->N % 8 || p->ldx % 8 || p->lddy % 8) return 0;
  if (p->stride != 0 && p->stride != 1 && p->stride != 2) return 0;
  if ((p->stride == 2 || p->ups) && (p->KH != 3 || (p->stride == 2 && p->ups) || (p->ups != 0 && p->ups != 1))) return 0;
  if (p->ups && ((p->H | p->W) & 1)) return 0;
  if (p->KH == 3) {
    const int W = p->W, HW = p->H * p->W;
    if (HW % 32) return 0;
    if (!((W >= 32 && W % 32 == 0) || (W >= 8 && 32 % W == 0))) return 0;
  }
  return 1;
}

extern "C" int aptp_conv_wgrad_suggest_split(const AptpWgradParams* p) {
  const int M = p->B * p->H * p->W, nsteps = (M + 31) / 32;
  const int tiles = ((p->N + 63) / 64) * ((p->C + 63) / 64);
  int split = (512 + tiles - 1) / tiles;
  const int cap = nsteps / 4 > 0 ? nsteps / 4 : 1;
  split = split < 1 ? 1 : (split > cap ? cap : split);
  return split > 64 ? 64 : split;
}

static int sub_1x1() {
  // sub-steps per barrier of the 1x1 form (APTP_WGRAD_SUB = 1 / 2 / 4 for A/B timing)
  static const int sub_env = [] { const char* e = getenv("APTP_WGRAD_SUB"); return e ? atoi(e) : 0; }();
  return (sub_env == 1 || sub_env == 2 || sub_env == 4) ? sub_env : APTP_WGRAD_SUB_DEFAULT;
}

static int fill_wgk(const AptpWgradParams* p, WgK& k, int SUB1, int& stride_out) {
  APTP_CHECK(p && p->x && p->dy && p->dw, "conv_wgrad: null pointer");
  APTP_CHECK(aptp_conv_wgrad_supported(p), "conv_wgrad: unsupported geometry (KH=%d KW=%d H=%d W=%d C=%d N=%d)", p->KH, p->KW, p->H, p->W, p->C, p->N);
  APTP_CHECK(ALIGN16(p->x) && ALIGN16(p->dy) && ALIGN16(p->dw) && p->split_m >= 1, "conv_wgrad: alignment / split");
  k.x = (const __bf16*)p->x; k.ldx = p->ldx; k.dy = (const __bf16*)p->dy; k.lddy = p->lddy; k.dw = p->dw;
  k.B = p->B; k.H = p->H; k.W = p->W; k.C = p->C; k.N = p->N; k.HW = p->H * p->W; k.M = p->B * k.HW;
  const int stride = p->stride == 2 ? 2 : 1;
  k.ups = p->ups ? 1 : 0;
  k.Hin = stride == 2 ? 2 * p->H : (k.ups ? p->H / 2 : p->H);
  k.Win = stride == 2 ? 2 * p->W : (k.ups ? p->W / 2 : p->W);
  k.nsteps = p->KH == 3 ? (k.M + 31) / 32 : (k.M + 32 * SUB1 - 1) / (32 * SUB1);
  k.ldw = p->ld_dw ? p->ld_dw : p->C;
  k.db = p->db;
  k.slab_stride = p->slab_stride ? p->slab_stride : (int64_t)p->N * p->KH * p->KW * k.ldw;
  k.db_stride = p->db_stride ? p->db_stride : p->N;
  APTP_CHECK(k.slab_stride >= (int64_t)p->N * p->KH * p->KW * k.ldw && k.slab_stride % 4 == 0 && k.db_stride >= p->N, "conv_wgrad: slab strides");
  APTP_CHECK(k.ldw >= p->C && k.ldw % 4 == 0, "conv_wgrad: ld_dw");
  k.split = p->split_m;                 // (a slice without a barrier step of its own writes zeros: the caller folds split_m slabs)
  APTP_CHECK(p->split_m <= (k.M + 31) / 32, "conv_wgrad: split_m %d exceeds the %d 32-pixel steps", p->split_m, (k.M + 31) / 32);
  k.tiles_n = (p->N + 63) / 64; k.tiles_c = (p->C + 63) / 64;
  k.Wt = p->W < 32 ? p->W : 32; k.R = p->W < 32 ? 32 / p->W : 1; k.hw2 = stride * (k.Wt - 1) + 3;
  APTP_CHECK(p->KH == 1 || (stride * (k.R - 1) + 3) * k.hw2 <= (stride == 2 ? XROWS_S2 : XROWS), "conv_wgrad: halo does not fit");
  k.first_block = 0; k.pad_ = 0;
  stride_out = stride;
  return APTP_OK;
}

extern "C" int aptp_conv_wgrad(const AptpWgradParams* p, aptp_stream_t stream) {
  WgK k;
  int stride = 1;
  const int SUB1 = sub_1x1();
  const int rc = fill_wgk(p, k, SUB1, stride);
  if (rc != APTP_OK) return rc;
  const int64_t nblk = (int64_t)k.tiles_n * k.tiles_c * k.split;
  APTP_CHECK(nblk < (1LL << 31), "conv_wgrad: grid too large");
  if (p->KH == 3 && stride == 2) hipLaunchKernelGGL((conv_wgrad_kernel<9, 1, 2>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, k);
  else if (p->KH == 3) hipLaunchKernelGGL((conv_wgrad_kernel<9, 1>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, k);
  else if (SUB1 == 4) hipLaunchKernelGGL((conv_wgrad_kernel<1, 4>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, k);
  else if (SUB1 == 2) hipLaunchKernelGGL((conv_wgrad_kernel<1, 2>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, k);
  else hipLaunchKernelGGL((conv_wgrad_kernel<1, 1>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, k);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}

// ---- batched form ----------------------------------------------------------------------------------------------------------
extern "C" int64_t aptp_conv_wgrad_many_item_bytes() { return (int64_t)sizeof(WgK); }

extern "C" int aptp_conv_wgrad_many_blocks(const AptpWgradParams* p) {
  if (!p || p->split_m < 1) return -1;
  const int64_t n = (int64_t)((p->N + 63) / 64) * ((p->C + 63) / 64) * p->split_m;
  return n < (1LL << 30) ? (int)n : -1;
}

extern "C" int aptp_conv_wgrad_many_fill(const AptpWgradParams* p, void* item_out, int32_t first_block) {
  APTP_CHECK(item_out, "conv_wgrad_many_fill: null item");
  APTP_CHECK(p && (p->stride == 0 || p->stride == 1), "conv_wgrad_many: stride-1 layers only (incl. the folded nearest-x2 up-sample; the stride-2 layers take aptp_conv_wgrad)");
  WgK k;
  int stride = 1;
  const int rc = fill_wgk(p, k, APTP_WGRAD_SUB_DEFAULT, stride);
  if (rc != APTP_OK) return rc;
  k.first_block = first_block;
  memcpy(item_out, &k, sizeof(WgK));
  return APTP_OK;
}

extern "C" int aptp_conv_wgrad_many(const void* items_dev, const int32_t* block_item_dev, int32_t n_items, int32_t total_blocks,
                                    int32_t taps, aptp_stream_t stream) {
  APTP_CHECK(items_dev && block_item_dev && n_items > 0 && total_blocks > 0, "conv_wgrad_many: empty batch");
  APTP_CHECK(taps == 1 || taps == 9, "conv_wgrad_many: taps %d", taps);
  APTP_CHECK(ALIGN16(items_dev), "conv_wgrad_many: descriptor table alignment");
  const WgK* items = (const WgK*)items_dev;
  if (taps == 9) hipLaunchKernelGGL((conv_wgrad_many_kernel<9, 1>), dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items, block_item_dev);
  else hipLaunchKernelGGL((conv_wgrad_many_kernel<1, APTP_WGRAD_SUB_DEFAULT>), dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items, block_item_dev);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}
