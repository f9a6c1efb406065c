// PROTOTYPE (not part of libaptp_hip.so; see tools/proto/bench_proj_stream.py): how fast can a short-K projection of the masked
// step run when the whole weight matrix is resident?  y[M][N] = x[M][K] . W[N][K]^T + bias (+ residual), bf16 in / out, fp32
// accumulate, K <= 128, N <= 320 (level-64 to_out: M = 16384, K = 128, N = 320: 11.4-12.6 us with the tiled kernel, whose
// 768 workgroups each pay a pipeline fill, two K-steps and an epilogue for 8 KB of output).
// One workgroup (4 waves) = 64 rows x ALL N columns: W (N x K, 80 KB) and the x tile (64 x K, 16 KB) are copied to LDS once,
// wave w owns columns [NW*w, NW*(w+1)) of all 64 rows (4 x NF accumulator tiles), the result goes back through LDS so that
// residual reads and y writes are 16-byte row pieces.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int KMAX = 128, PITCH = KMAX + 8;      // bf16 elements per LDS row of an operand (272 B: conflict-free 16-byte fragment reads)

__device__ __forceinline__ unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

template <int NF>   // 16-column accumulator tiles per wave: N = 4 * 16 * NF
__global__ __launch_bounds__(256) void proj_stream_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w,
                                                          const float* __restrict__ bias, const __bf16* __restrict__ res,
                                                          __bf16* __restrict__ y, int M, int K, int N) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* ws = reinterpret_cast<__bf16*>(smem);                 // [N][PITCH]
  __bf16* xs = ws + (size_t)N * PITCH;                          // [64][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 64;
  const int kc = K >> 3;                                        // 16-byte chunks per row
  // ---- copy W and the x tile (all loads of a batch in flight before the first LDS write) ----------------------------------
  const int wchunks = N * kc, xchunks = 64 * kc;
  for (int base = 0; base < wchunks; base += 256 * 8) {
    u32x4 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = base + tid + 256 * i;
      r[i] = c < wchunks ? *reinterpret_cast<const u32x4*>(w + (size_t)(c / kc) * K + (c % kc) * 8) : (u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = base + tid + 256 * i;
      if (c < wchunks) *reinterpret_cast<u32x4*>(ws + (size_t)(c / kc) * PITCH + (c % kc) * 8) = r[i];
    }
  }
  {
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      const int row = c / kc;
      r[i] = (c < xchunks && m0 + row < M) ? *reinterpret_cast<const u32x4*>(x + (size_t)(m0 + row) * K + (c % kc) * 8) : (u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      if (c < xchunks) *reinterpret_cast<u32x4*>(xs + (size_t)(c / kc) * PITCH + (c % kc) * 8) = r[i];
    }
  }
  __syncthreads();
  // ---- MFMA: wave owns columns n0 .. n0 + 16 NF of all 64 rows -------------------------------------------------------------
  const int n0 = wave * 16 * NF;
  const int lr = lane & 15, lk = lane >> 4;
  f32x4 acc[4][NF];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {
    bf16x8 a[4], b[NF];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) a[mf] = *reinterpret_cast<const bf16x8*>(xs + (size_t)(mf * 16 + lr) * PITCH + k0 + lk * 8);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) b[nf] = *reinterpret_cast<const bf16x8*>(ws + (size_t)(n0 + nf * 16 + lr) * PITCH + k0 + lk * 8);
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        // A = W fragment (rows = n), B = x fragment (columns = m): a lane ends up with 4 consecutive n of one row m
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nf], a[mf], acc[mf][nf], 0, 0, 0);
  }
  __syncthreads();                                              // all fragment reads done: W's space becomes the output tile
  // ---- epilogue: + bias, to bf16, through LDS [64][N + 8], then 16-byte row pieces (+ residual) ----------------------------
  unsigned short* os = reinterpret_cast<unsigned short*>(smem);
  const int OP = N + 8;
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int m = mf * 16 + lr, n = n0 + nf * 16 + 4 * lk;      // lane: row m, columns n .. n + 3
      const float4 bb = *reinterpret_cast<const float4*>(bias + n);
      ushort4 o;
      o.x = f2bf(acc[mf][nf][0] + bb.x); o.y = f2bf(acc[mf][nf][1] + bb.y);
      o.z = f2bf(acc[mf][nf][2] + bb.z); o.w = f2bf(acc[mf][nf][3] + bb.w);
      *reinterpret_cast<ushort4*>(os + (size_t)m * OP + n) = o;
    }
  __syncthreads();
  const int nc = N >> 3;
  for (int c = tid; c < 64 * nc; c += 256) {
    const int row = c / nc, col = (c % nc) * 8;
    if (m0 + row >= M) continue;
    u32x4 v = *reinterpret_cast<const u32x4*>(os + (size_t)row * OP + col);
    if (res) {
      const u32x4 r = *reinterpret_cast<const u32x4*>(res + (size_t)(m0 + row) * N + col);
      unsigned int* pv = reinterpret_cast<unsigned int*>(&v);
      const unsigned int* pr = reinterpret_cast<const unsigned int*>(&r);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf2f((unsigned short)(pv[e] & 0xffffu)) + bf2f((unsigned short)(pr[e] & 0xffffu));
        const float hi = bf2f((unsigned short)(pv[e] >> 16)) + bf2f((unsigned short)(pr[e] >> 16));
        pv[e] = (unsigned int)f2bf(lo) | ((unsigned int)f2bf(hi) << 16);
      }
    }
    *reinterpret_cast<u32x4*>(y + (size_t)(m0 + row) * N + col) = v;
  }
}

extern "C" int proj_stream(const void* x, const void* w, const float* bias, const void* res, void* y, int M, int K, int N, void* stream) {
  if (K % 32 || K > KMAX || N % 64 || N > 320) return -1;
  const int NF = N / 64;
  const size_t lds = (size_t)(N + 64) * PITCH * 2;
  const size_t lds_out = (size_t)64 * (N + 8) * 2;
  const size_t bytes = lds > lds_out ? lds : lds_out;
  dim3 grid((M + 63) / 64), block(256);
#define LAUNCH(NFV)                                                                                                      \
  do {                                                                                                                   \
    hipFuncSetAttribute(reinterpret_cast<const void*>(proj_stream_kernel<NFV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
    hipLaunchKernelGGL(proj_stream_kernel<NFV>, grid, block, bytes, (hipStream_t)stream, (const __bf16*)x, (const __bf16*)w, bias, \
                       (const __bf16*)res, (__bf16*)y, M, K, N);                                                        \
  } while (0)
  switch (NF) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    default: LAUNCH(5); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
