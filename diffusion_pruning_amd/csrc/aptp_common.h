// Shared device/host helpers for libaptp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "aptp_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define APTP_WAVE 64

void aptp_set_error(const char* fmt, ...);

// fp32 parity instantiations (parity_f32.hip), reached through io_f32 of the public parameter blocks
int aptp_groupnorm_f32(const AptpGroupNormParams* p, aptp_stream_t stream);
int aptp_layernorm_f32(const AptpLayerNormParams* p, aptp_stream_t stream);
int aptp_attention_f32(const AptpAttentionParams* p, aptp_stream_t stream);

#define APTP_CHECK(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      aptp_set_error(__VA_ARGS__);       \
      return APTP_EINVAL;                \
    }                                    \
  } while (0)

#define APTP_LAUNCH_CHECK()                                              \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      aptp_set_error("launch failed: %s", hipGetErrorString(e__));       \
      return APTP_ELAUNCH;                                               \
    }                                                                    \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(__bf16 v) { return (float)v; }

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  union { __bf16 h[2]; uint32_t u; } r;
  r.h[0] = (__bf16)a;
  r.h[1] = (__bf16)b;
  return r.u;
}

__device__ __forceinline__ void unpack_bf16x8(const uint4& q, float* f) {
  union { uint4 q; __bf16 h[8]; } u;
  u.q = q;
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)u.h[i];
}

__device__ __forceinline__ uint4 pack_bf16x8(const float* f) {
  uint4 q;
  q.x = pack_bf16x2(f[0], f[1]);
  q.y = pack_bf16x2(f[2], f[3]);
  q.z = pack_bf16x2(f[4], f[5]);
  q.w = pack_bf16x2(f[6], f[7]);
  return q;
}

// (v_rcp_f32, 1 ulp, instead of an IEEE division -- ~10 instructions per element in kernels that apply it to every element)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// Standard normal CDF Phi(x) = 0.5 * erfc(-x / sqrt 2) with erfc by Abramowitz & Stegun 7.1.26 (erfc(z) = poly(t) * t * exp(-z^2),
// t = 1 / (1 + 0.3275911 z), z >= 0; |error| <= 1.5e-7 in exact arithmetic, ~6e-7 in fp32): one reciprocal, one raw v_exp_f32
// (argument <= 0) and six FMAs, against the ~30 instructions of libm's erff in loops that are VALU-bound (the GEGLU epilogue of
// every feed-forward GEMM, the training GEGLU kernels).  The negative side is 0.5 * erfc(|z|) itself -- no 1 - (1 - tiny)
// cancellation in GELU's tail.  Also hands back exp(-x^2 / 2) for the density in GELU's derivative.
__device__ __forceinline__ float norm_cdf_f(float x, float* exp_mhx2) {
  const float z = __builtin_fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
  float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  const float half_erfc = 0.5f * poly * t * e;
  if (exp_mhx2) *exp_mhx2 = e;
  return x >= 0.f ? 1.0f - half_erfc : half_erfc;
}
__device__ __forceinline__ float gelu_erf_f(float x) { return x * norm_cdf_f(x, nullptr); }

// Next-launch weight prefetch (AptpConvGemmParams.prefetch / AptpGroupNormParams.prefetch): touch one dword per 64-byte line
// of slice j of nper of the xcd-th eighth of [ptr, ptr + bytes) by LDS-DMA into a 256-byte scratch row (no VGPRs; results
// discarded).  A workgroup with linear id L runs on XCD L % 8 (round-robin dispatch), and the weight-major workgroup order
// of the next conv_gemm launch hands the rows of that eighth of a [N][K] weight matrix to the same XCD: the lines land
// in the L2 that will be asked for them.
__device__ __forceinline__ void aptp_prefetch_slice(const char* ptr, int64_t bytes, int xcd, int j, int nper, int tid, int nt,
                                                    unsigned* scratch) {
  const int lines = (int)(bytes >> 6);
  const int x0 = (int)(((int64_t)lines * xcd) >> 3), x1 = (int)(((int64_t)lines * (xcd + 1)) >> 3);
  const int per = ((x1 - x0) + nper - 1) / nper;
  const int l0 = x0 + j * per;
  const int l1 = l0 + per < x1 ? l0 + per : x1;
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  for (int l = l0 + tid; l < l1; l += nt)
    __builtin_amdgcn_global_load_lds((gptr)(ptr + (int64_t)l * 64), (lptr)scratch, 4, 0, 0);
}
