// Small data-movement kernels of the packed-parameter fine-tune step (packed_train.py), each replacing a chain of torch
// kernels inside the captured graph:
//   aptp_fold_rows   out[row][c] = sum_r in[r][row][c]   (fixed order: deterministic) -- the slab sum of a split weight
//                    gradient written straight into the padded packed layout, and the chunk sums of bias / norm-affine
//                    gradients (was: torch reduce kernels, 526 per step at 8 us);
//   aptp_pack_dgrad  the data-gradient operand of a contraction from its forward operand: dst[c][taps-1-t][n] = src[n][t][c]
//                    (180-degree rotated transpose, bf16 -> bf16) as one tiled transpose through LDS (was: flip + strided
//                    copy, 6-9 ms per step over 350 weights).
#include "aptp_common.h"

namespace {

struct FoldK { const float* in; float* out; int R, n_rows, C, ld_out; float* tail_out; int rows1; int pair_split; int tail_n; };

// the four sums of column quad e .. e+3 to their destination (C % 4 == 0)
__device__ __forceinline__ void fold_store4(const FoldK& p, int64_t e, const float4 t) {
  const int64_t row = e / p.C, c = e - row * p.C;
  if (p.pair_split) {          // columns are (a, b) pairs: a_k -> out[k], b_k -> tail_out[k] (or out[ld_out + k])
    float* const ob = p.tail_out ? p.tail_out : p.out + p.ld_out;
    const int k = (int)(c >> 1);
    p.out[k] = t.x; ob[k] = t.y; p.out[k + 1] = t.z; ob[k + 1] = t.w;
  } else if (row < p.rows1) {
    float* o = p.out + row * p.ld_out + c;
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  } else {
    const int64_t i = (row - p.rows1) * p.C + c;      // tail: only its first tail_n values exist at the destination
    float* o = p.tail_out + i;
    if (i + 3 < p.tail_n) { o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
    else { if (i < p.tail_n) o[0] = t.x; if (i + 1 < p.tail_n) o[1] = t.y; if (i + 2 < p.tail_n) o[2] = t.z; }
  }
}

// block = 32 column quads x 8 row slices: slice s sums rows s, s+8, ... (independent 16-byte loads in flight), the slices are
// folded 0..7 through LDS: the order is fixed, the result does not depend on scheduling
__device__ __forceinline__ void fold_rows_block(const FoldK& p, int64_t blk) {
  __shared__ float4 red[8][32];
  const int64_t total = (int64_t)p.n_rows * p.C;
  const int cq = threadIdx.x & 31, rs = threadIdx.x >> 5;
  if ((p.C & 3) == 0) {
    const int64_t e = ((int64_t)blk * 32 + cq) << 2;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < total) {
      int r = rs;
      for (; r + 24 < p.R; r += 32) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.in + (int64_t)r * total + e);
        const float4 b1 = *reinterpret_cast<const float4*>(p.in + (int64_t)(r + 8) * total + e);
        const float4 b2 = *reinterpret_cast<const float4*>(p.in + (int64_t)(r + 16) * total + e);
        const float4 b3 = *reinterpret_cast<const float4*>(p.in + (int64_t)(r + 24) * total + e);
        a.x += (b0.x + b1.x) + (b2.x + b3.x); a.y += (b0.y + b1.y) + (b2.y + b3.y);
        a.z += (b0.z + b1.z) + (b2.z + b3.z); a.w += (b0.w + b1.w) + (b2.w + b3.w);
      }
      for (; r < p.R; r += 8) {
        const float4 b = *reinterpret_cast<const float4*>(p.in + (int64_t)r * total + e);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
    }
    red[rs][cq] = a;
    __syncthreads();
    if (rs == 0 && e < total) {
      float4 t = red[0][cq];
#pragma unroll
      for (int q = 1; q < 8; ++q) { const float4 b = red[q][cq]; t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
      fold_store4(p, e, t);
    }
  } else {
    const int64_t e = (int64_t)blk * 32 + cq;
    float a = 0.f;
    if (e < total)
      for (int r = rs; r < p.R; r += 8) a += p.in[(int64_t)r * total + e];
    reinterpret_cast<float*>(red)[rs * 32 + cq] = a;
    __syncthreads();
    if (rs == 0 && e < total) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += reinterpret_cast<float*>(red)[q * 32 + cq];
      const int64_t row = e / p.C;
      if (p.pair_split) ((e & 1) ? (p.tail_out ? p.tail_out : p.out + p.ld_out) : p.out)[e >> 1] = t;
      else if (row < p.rows1) p.out[row * p.ld_out + (e - row * p.C)] = t;
      else if ((row - p.rows1) * p.C + (e - row * p.C) < p.tail_n) p.tail_out[(row - p.rows1) * p.C + (e - row * p.C)] = t;
    }
  }
}

__global__ __launch_bounds__(256) void fold_rows_kernel(const FoldK p) { fold_rows_block(p, blockIdx.x); }

// every slab fold of a backward pass in ONE launch (the expert fine-tune step issued 357 of them, 7.8 us each): items[] in
// device memory, starts[i] = first 32-quad unit of item i.  A workgroup takes FOLD_UNITS consecutive units: one bisection of
// the table per workgroup (nine dependent loads: per unit they cost more than the fold itself), then a walk.
constexpr int FOLD_UNITS = 16;
__device__ __forceinline__ FoldK fold_item(const AptpFoldRowsParams& it) {
  return FoldK{it.partials, it.out, it.R, it.n_rows, it.C, it.ld_out, it.tail_out, it.n_rows - it.tail_rows, it.pair_split,
               it.tail_n > 0 ? it.tail_n : it.tail_rows * it.C};
}
__global__ __launch_bounds__(256) void fold_rows_many_kernel(const AptpFoldRowsParams* __restrict__ items,
                                                             const int32_t* __restrict__ starts, int n_items, int total_units) {
  int u = blockIdx.x * FOLD_UNITS;
  const int u_end = u + FOLD_UNITS < total_units ? u + FOLD_UNITS : total_units;
  int lo = 0, hi = n_items;                     // invariant: starts[lo] <= u < starts[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (starts[mid] <= u) lo = mid; else hi = mid;
  }
  int s0 = starts[lo], s1 = starts[lo + 1];
  FoldK k = fold_item(items[lo]);
  while (u < u_end) {
    while (u >= s1) { ++lo; s0 = s1; s1 = starts[lo + 1]; k = fold_item(items[lo]); }
    const int run_end = u_end < s1 ? u_end : s1;        // this workgroup's units inside item `lo`
    if (k.R <= 8 && (k.C & 3) == 0) {
      // few slabs (the batched weight gradients leave 2..8): one column quad per thread, all R loads of a thread in flight
      // at once, no staging.  Same summation order as the staged form (slab 0, 1, ..., R-1): bitwise the same result.
      const int64_t total = (int64_t)k.n_rows * k.C;
      for (int64_t q = (int64_t)(u - s0) * 32 + threadIdx.x; q < (int64_t)(run_end - s0) * 32; q += 256) {
        const int64_t e = q << 2;
        if (e >= total) break;
        float4 b[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
          b[r] = r < k.R ? *reinterpret_cast<const float4*>(k.in + (int64_t)r * total + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 t = b[0];
#pragma unroll
        for (int r = 1; r < 8; ++r)
          if (r < k.R) { t.x += b[r].x; t.y += b[r].y; t.z += b[r].z; t.w += b[r].w; }
        fold_store4(k, e, t);
      }
      u = run_end;
    } else {
      for (; u < run_end; ++u) {
        __syncthreads();                          // (the staging array of the previous unit is free)
        fold_rows_block(k, u - s0);
      }
    }
  }
}

struct PackK { const __bf16* src; __bf16* dst; int N, C, taps, src_ld, dst_ld, dst_rows; };

// tile: src rows n0..n0+63 (64 channels c0..c0+63 each) -> dst rows c0.., columns n0..
__device__ __forceinline__ void pack_dgrad_tile(const PackK& p, int n0, int c0, int t) {
  __shared__ __bf16 tile[64][72];
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i, r = id >> 3, ch = id & 7;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    const int n = n0 + r, c = c0 + ch * 8;
    if (n < p.N && c < p.src_ld) v = *reinterpret_cast<const uint4*>(p.src + ((int64_t)n * p.taps + t) * p.src_ld + c);
    *reinterpret_cast<uint4*>(&tile[r][ch * 8]) = v;
  }
  __syncthreads();
  const int td = p.taps - 1 - t;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i, r = id >> 3, ch = id & 7;      // output row c0 + r, columns n0 + 8*ch ..
    const int c = c0 + r, n = n0 + ch * 8;
    if (c < p.dst_rows && n < p.dst_ld) {
      union { uint4 q; __bf16 e[8]; } o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o.e[k] = (c < p.C && n + k < p.N) ? tile[ch * 8 + k][r] : (__bf16)0.0f;
      *reinterpret_cast<uint4*>(p.dst + ((int64_t)c * p.taps + td) * p.dst_ld + n) = o.q;
    }
  }
}

// grid (ceil(dst_ld/64), ceil(dst_rows/64), taps)
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const PackK p) {
  pack_dgrad_tile(p, blockIdx.x * 64, blockIdx.y * 64, blockIdx.z);
}

// every data-gradient operand of a model in ONE launch: items[] (device memory) describes the weights, starts[i] is the first
// workgroup of item i (starts[n_items] = grid size); a workgroup finds its item by bisection
__global__ __launch_bounds__(256) void pack_dgrad_many_kernel(const AptpPackDgradParams* __restrict__ items,
                                                              const int32_t* __restrict__ starts, int n_items) {
  const int b = blockIdx.x;
  int lo = 0, hi = n_items;                     // invariant: starts[lo] <= b < starts[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (starts[mid] <= b) lo = mid; else hi = mid;
  }
  const AptpPackDgradParams it = items[lo];
  PackK p{(const __bf16*)it.src, (__bf16*)it.dst, it.N, it.C, it.taps, it.src_ld, it.dst_ld, it.dst_rows};
  int local = b - starts[lo];
  const int gx = (it.dst_ld + 63) / 64, gy = (it.dst_rows + 63) / 64;
  const int bx = local % gx; local /= gx;
  const int by = local % gy, t = local / gy;
  pack_dgrad_tile(p, bx * 64, by * 64, t);
}

}  // namespace

#define ALIGN16(p) (((uintptr_t)(p) % 16) == 0)

extern "C" int aptp_fold_rows(const AptpFoldRowsParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->partials && p->out && p->R >= 1 && p->n_rows >= 1 && p->C >= 1 && (p->ld_out >= p->C || (p->pair_split && 2 * p->ld_out >= p->C)),
             "fold_rows: bad arguments");
  APTP_CHECK((p->C & 3) != 0 || (ALIGN16(p->partials) && (((int64_t)p->n_rows * p->C) & 3) == 0), "fold_rows: alignment");
  APTP_CHECK(!p->pair_split || (p->n_rows == 1 && p->tail_rows == 0 && (p->C & 1) == 0), "fold_rows: pair_split folds ONE row of (a, b) pairs");
  APTP_CHECK(p->tail_rows >= 0 && (p->pair_split || p->tail_rows < p->n_rows) && (p->tail_rows == 0 || p->tail_out), "fold_rows: tail");
  APTP_CHECK(p->tail_n >= 0 && p->tail_n <= p->tail_rows * p->C, "fold_rows: tail_n");
  FoldK k{p->partials, p->out, p->R, p->n_rows, p->C, p->ld_out, p->tail_out, p->n_rows - p->tail_rows, p->pair_split,
          p->tail_n > 0 ? p->tail_n : p->tail_rows * p->C};
  const int64_t total = (int64_t)p->n_rows * p->C;
  const int64_t work = (p->C & 3) == 0 ? total / 4 : total;          // column quads (or single columns), 32 per block
  const int64_t blocks = (work + 31) / 32;
  APTP_CHECK(blocks < (1LL << 31), "fold_rows: grid too large");
  hipLaunchKernelGGL(fold_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}

extern "C" int aptp_fold_rows_blocks(const AptpFoldRowsParams* p) {
  if (!p || p->n_rows < 1 || p->C < 1) return 0;
  const int64_t total = (int64_t)p->n_rows * p->C;
  const int64_t work = (p->C & 3) == 0 ? total / 4 : total;
  const int64_t blocks = (work + 31) / 32;
  return blocks < (1LL << 31) ? (int)blocks : 0;
}

extern "C" int aptp_fold_rows_many(const AptpFoldRowsParams* items_dev, const int32_t* starts_dev, int32_t n_items,
                                   int32_t total_blocks, aptp_stream_t stream) {
  APTP_CHECK(items_dev && starts_dev && n_items >= 1 && total_blocks >= 1, "fold_rows_many: bad arguments");
  hipLaunchKernelGGL(fold_rows_many_kernel, dim3((unsigned)((total_blocks + FOLD_UNITS - 1) / FOLD_UNITS)), dim3(256), 0, (hipStream_t)stream,
                     items_dev, starts_dev, n_items, total_blocks);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}

extern "C" int aptp_pack_dgrad_blocks(const AptpPackDgradParams* p) {
  if (!p || p->dst_ld < 1 || p->dst_rows < 1 || p->taps < 1) return 0;
  return ((p->dst_ld + 63) / 64) * ((p->dst_rows + 63) / 64) * p->taps;
}

extern "C" int aptp_pack_dgrad_many(const AptpPackDgradParams* items_dev, const int32_t* starts_dev, int32_t n_items,
                                    int32_t total_blocks, aptp_stream_t stream) {
  APTP_CHECK(items_dev && starts_dev && n_items >= 1 && total_blocks >= 1, "pack_dgrad_many: bad arguments");
  hipLaunchKernelGGL(pack_dgrad_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, starts_dev, n_items);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}

extern "C" int aptp_pack_dgrad(const AptpPackDgradParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->src && p->dst && p->N >= 1 && p->C >= 1 && p->taps >= 1, "pack_dgrad: bad arguments");
  APTP_CHECK(p->src_ld % 8 == 0 && p->dst_ld % 8 == 0 && ALIGN16(p->src) && ALIGN16(p->dst) && p->dst_rows >= p->C && p->src_ld >= p->C
             && p->dst_ld >= p->N, "pack_dgrad: layout");
  PackK k{(const __bf16*)p->src, (__bf16*)p->dst, p->N, p->C, p->taps, p->src_ld, p->dst_ld, p->dst_rows};
  dim3 grid((p->dst_ld + 63) / 64, (p->dst_rows + 63) / 64, p->taps);
  hipLaunchKernelGGL(pack_dgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, k);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}
