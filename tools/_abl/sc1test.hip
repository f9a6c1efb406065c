#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int AUX>
__global__ void k(float4* out) {
  const int tid = threadIdx.x;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + blockIdx.x * 256, 0, 256 * 16, 0x00020000);
  u32x4 v; v[0] = tid * 4 + 0; v[1] = tid * 4 + 1; v[2] = tid * 4 + 2; v[3] = tid * 4 + 3;
  __builtin_amdgcn_raw_buffer_store_b128(v, r, tid * 16, 0, AUX);
}
int main() {
  float4* d; hipMalloc(&d, 4 * 256 * 16);
  unsigned h[4 * 256 * 4];
  for (int aux : {0, 16}) {
    hipMemset(d, 0xff, 4 * 256 * 16);
    if (aux == 0) hipLaunchKernelGGL(k<0>, dim3(4), dim3(256), 0, 0, d); else hipLaunchKernelGGL(k<16>, dim3(4), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4 * 256 * 4; ++i) bad += h[i] != (unsigned)(i % 1024);
    printf("aux %d bad %d first %u %u %u %u %u %u %u %u\n", aux, bad, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
  return 0;
}
