// What does ds_read_b64_tr_b16 hand each lane?  LDS holds a [16 rows][64 cols] tile of 16-bit values row*256 + col; lane 4q+p of a
// 16-lane group addresses row r0 + q, columns c0 + 4p .. +3 (r0 = 4 * (lane >> 5), c0 = 16 * ((lane >> 4) & 1)).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 tools/proto/tr_read_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out) {
  __shared__ short lds[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = (short)((i / 64) * 256 + (i % 64));
  __syncthreads();
  const int lane = threadIdx.x, li = lane & 15, q = li >> 2, p = li & 3, gb = (lane >> 4) & 1, hh = lane >> 5;
  const short* a = lds + (4 * hh + q) * 64 + 16 * gb + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf("  (row %d, col %2d)", h[l * 4 + e] / 256, h[l * 4 + e] % 256);
    printf("\n");
  }
  return 0;
}
