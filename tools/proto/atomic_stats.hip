// Micro-benchmark (experiment, not product code): what do deterministic fixed-point atomics cost at the END of a producer launch?
// G workgroups of 256 threads stream `bytes_per_wg` (to give the launch a realistic duration), then each of its 4 waves issues
// `n_atomics` 64-bit integer atomic adds (no return) onto `n_addr` distinct addresses -- the pattern of per-(sample, 10-channel
// unit) GroupNorm statistics emitted by a GEMM epilogue.  Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC atomic_stats.hip -o libatomic_stats.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void atomic_tail_kernel(const uint4* src, uint4* dst, int vec_per_wg, unsigned long long* stats, int n_addr,
                                                          int n_atomics, int mode) {
  const int tid = threadIdx.x;
  const uint4* s = src + (size_t)blockIdx.x * vec_per_wg;
  uint4* d = dst + (size_t)blockIdx.x * vec_per_wg;
  unsigned acc = 0;
  for (int i = tid; i < vec_per_wg; i += 256) {
    uint4 v = s[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
    d[i] = v;
  }
  if (mode == 0) return;
  const int lane = tid & 63, wave = tid >> 6;
  if (lane < n_atomics) {
    // address pattern: the units a wave's column range touches, for the sample its rows belong to
    const int b = (blockIdx.x * 4 / gridDim.x) & 3;
    const int a = (b * (n_addr / 4) + ((blockIdx.x * 7 + wave * 3 + lane) % (n_addr / 4)));
    const unsigned long long val = (unsigned long long)(acc & 0xffff) + 1ull;
    if (mode == 1) __hip_atomic_fetch_add(stats + a, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else atomicAdd(reinterpret_cast<float*>(stats) + a, (float)val);
  }
}

extern "C" int atomic_tail(const void* src, void* dst, int grid, int vec_per_wg, void* stats, int n_addr, int n_atomics, int mode, void* stream) {
  hipLaunchKernelGGL(atomic_tail_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, vec_per_wg,
                     (unsigned long long*)stats, n_addr, n_atomics, mode);
  return (int)hipGetLastError();
}
