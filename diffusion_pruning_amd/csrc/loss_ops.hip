// Mean-squared-error terms of the pruning / fine-tune steps (pdm/training/trainer.py:1197-1225, 1729-1752): the diffusion,
// output-distillation and nine block-distillation losses, each mean((a - b)^2) over a whole activation.
//
//   aptp_mse(backward = 0)   out[0] = sum((a - b)^2) * inv_n    two launches: <= 1024 workgroup partials (every slot is
//                            written, nothing needs zeroing), then one workgroup folds them in a fixed order
//   aptp_mse(backward = 1)   da = (a - b) * g[0] * scale         (scale = 2 / n: the gradient of the mean w.r.t. a)
//
// Operands are read where they lie (bf16 activations or fp32 predictions, row-strided views such as a channel slice of a
// skip-concat buffer); no fp32 copies are made.  Why not torch's mse_loss: its reduction over a large tensor is a multi-block
// kernel with semaphores, and replayed from a HIP graph next to other graphs it returned wrong sums on MI355X (values only;
// tools/diag_overlap.py) -- these two launches have no cross-workgroup protocol at all.  HBM-bound: 2 reads (+1 write).
#include "aptp_common.h"

namespace {

struct MseK {
  const void* a; int64_t lda;
  const void* b; int64_t ldb;
  void* da; int64_t ldda;
  int64_t nvec;          // rows * CO
  int CO;                // C / 8
  float* partial; float* out; float inv_n;
  const float* g; float scale;
  int nblk;
};

template <bool F32>
__device__ __forceinline__ void load8(const void* base, int64_t row, int64_t ld, int o, float* f) {
  if (F32) {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + row * ld + (int64_t)o * 8);
    const float4 x = p[0], y = p[1];
    f[0] = x.x; f[1] = x.y; f[2] = x.z; f[3] = x.w; f[4] = y.x; f[5] = y.y; f[6] = y.z; f[7] = y.w;
  } else {
    const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(base) + row * ld + (int64_t)o * 8);
    unpack_bf16x8(q, f);
  }
}

template <bool F32>
__global__ __launch_bounds__(256) void mse_partial_kernel(const MseK p) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < p.nvec; v += (int64_t)gridDim.x * 256) {
    const int64_t row = v / p.CO;
    const int o = (int)(v - row * p.CO);
    float x[8], y[8];
    load8<F32>(p.a, row, p.lda, o, x);
    load8<F32>(p.b, row, p.ldb, o, y);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = x[e] - y[e]; s += d * d; }
    acc += s;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) p.partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup: thread t sums partials t, t + 256, ...; then a fixed tree.  nblk <= 1024.
__global__ __launch_bounds__(256) void mse_finish_kernel(const MseK p) {
  __shared__ float red[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < p.nblk; i += 256) a += p.partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.out[0] = red[0] * p.inv_n;
}

template <bool F32>
__global__ __launch_bounds__(256) void mse_bwd_kernel(const MseK p) {
  const float k = p.g[0] * p.scale;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < p.nvec; v += (int64_t)gridDim.x * 256) {
    const int64_t row = v / p.CO;
    const int o = (int)(v - row * p.CO);
    float x[8], y[8];
    load8<F32>(p.a, row, p.lda, o, x);
    load8<F32>(p.b, row, p.ldb, o, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (x[e] - y[e]) * k;
    if (F32) {
      float4* d = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.da) + row * p.ldda + (int64_t)o * 8);
      d[0] = make_float4(x[0], x[1], x[2], x[3]);
      d[1] = make_float4(x[4], x[5], x[6], x[7]);
    } else {
      *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(p.da) + row * p.ldda + (int64_t)o * 8) = pack_bf16x8(x);
    }
  }
}

}  // namespace

extern "C" int aptp_mse_nblocks(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0 || C % 8) return 0;
  const int64_t nvec = rows * (C / 8);
  const int64_t n = (nvec + 511) / 512;            // >= 2 vectors per thread before another workgroup is added
  return (int)(n < 1 ? 1 : (n > 1024 ? 1024 : n));
}

extern "C" int aptp_mse(const AptpMseParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->a && p->b, "mse: null operand");
  APTP_CHECK(p->rows > 0 && p->C > 0 && p->C % 8 == 0, "mse: C must be a positive multiple of 8 (got %d)", p ? p->C : 0);
  APTP_CHECK(p->lda >= p->C && p->ldb >= p->C && p->lda % 8 == 0 && p->ldb % 8 == 0, "mse: leading dimensions (multiples of 8, >= C)");
  APTP_CHECK(((uintptr_t)p->a & 15) == 0 && ((uintptr_t)p->b & 15) == 0, "mse: operands must be 16-byte aligned");
  MseK k;
  k.a = p->a; k.lda = p->lda; k.b = p->b; k.ldb = p->ldb;
  k.CO = p->C / 8; k.nvec = p->rows * k.CO;
  k.nblk = aptp_mse_nblocks(p->rows, p->C);
  hipStream_t s = (hipStream_t)stream;
  if (!p->backward) {
    APTP_CHECK(p->partial && p->out, "mse: partial / out");
    k.partial = p->partial; k.out = p->out; k.inv_n = 1.0f / ((float)p->rows * (float)p->C);
    if (p->f32) hipLaunchKernelGGL(mse_partial_kernel<true>, dim3(k.nblk), dim3(256), 0, s, k);
    else hipLaunchKernelGGL(mse_partial_kernel<false>, dim3(k.nblk), dim3(256), 0, s, k);
    hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, s, k);
  } else {
    APTP_CHECK(p->g && p->da && p->ldda >= p->C && p->ldda % 8 == 0 && ((uintptr_t)p->da & 15) == 0, "mse: backward operands");
    k.g = p->g; k.scale = 2.0f / ((float)p->rows * (float)p->C); k.da = p->da; k.ldda = p->ldda;
    const int64_t nb = (k.nvec + 255) / 256;
    const int grid = (int)(nb > 2048 ? 2048 : nb);
    if (p->f32) hipLaunchKernelGGL(mse_bwd_kernel<true>, dim3(grid), dim3(256), 0, s, k);
    else hipLaunchKernelGGL(mse_bwd_kernel<false>, dim3(grid), dim3(256), 0, s, k);
  }
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}
