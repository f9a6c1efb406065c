// The two ends of UNet2DConditionModel.forward that are not GEMMs (unet_2d_conditional.py:1497-1519,1614,1721-1726;
// diffusers Timesteps / get_timestep_embedding with flip_sin_to_cos = True, downscale_freq_shift = 0):
//   prologue: sample NCHW (fp32 or bf16) -> channels-last bf16 [B, H, W, cin_pad] with zeroed padding channels, and the
//             sinusoidal timestep embedding t_emb[b] = [cos(t_b * f_k) | sin(t_b * f_k)] as bf16 [B, 2 * half];
//   epilogue: conv_out's fp32 [B, H, W, ld] -> NCHW [B, C, H, W] in the caller's dtype.
// One launch each instead of the ~10 elementwise torch kernels (4.7 us apiece in the HIP graph) they replace.
#include "aptp_common.h"

namespace {

struct IoK {
  const void* sample; int sample_bf16; void* x; int B, C, HW, cpad;
  const float* t; const float* freqs; int half; void* temb;
  int pix_blocks;
};

__global__ __launch_bounds__(256) void unet_prologue_kernel(const IoK p) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < p.pix_blocks) {
    // one thread per pixel: gather its C channels (NCHW: stride HW), write cpad bf16 (cpad is a multiple of 8)
    const int64_t pix = (int64_t)blockIdx.x * 256 + tid;
    if (pix >= (int64_t)p.B * p.HW) return;
    const int b = (int)(pix / p.HW), r = (int)(pix - (int64_t)b * p.HW);
    __bf16* dst = reinterpret_cast<__bf16*>(p.x) + pix * p.cpad;
    for (int c0 = 0; c0 < p.cpad; c0 += 8) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        float v = 0.f;
        if (c < p.C) {
          const int64_t i = ((int64_t)b * p.C + c) * p.HW + r;
          v = p.sample_bf16 ? (float)reinterpret_cast<const __bf16*>(p.sample)[i] : reinterpret_cast<const float*>(p.sample)[i];
        }
        f[e] = v;
      }
      *reinterpret_cast<uint4*>(dst + c0) = pack_bf16x8(f);
    }
    return;
  }
  // timestep embedding: one thread per (b, k)
  const int i = ((int)blockIdx.x - p.pix_blocks) * 256 + tid;
  if (i >= p.B * p.half) return;
  const int b = i / p.half, k = i - b * p.half;
  const float a = p.t[b] * p.freqs[k];
  __bf16* e = reinterpret_cast<__bf16*>(p.temb) + (int64_t)b * 2 * p.half;
  e[k] = (__bf16)cosf(a);
  e[p.half + k] = (__bf16)sinf(a);
}

struct OutK { const float* y; int64_t ld; void* out; int out_bf16; int B, C, HW; };

__global__ __launch_bounds__(256) void unet_epilogue_kernel(const OutK p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // output element (b, c, r): coalesced stores
  if (i >= (int64_t)p.B * p.C * p.HW) return;
  const int r = (int)(i % p.HW);
  const int64_t bc = i / p.HW;
  const int c = (int)(bc % p.C), b = (int)(bc / p.C);
  const float v = p.y[((int64_t)b * p.HW + r) * p.ld + c];
  if (p.out_bf16) reinterpret_cast<__bf16*>(p.out)[i] = (__bf16)v;
  else reinterpret_cast<float*>(p.out)[i] = v;
}

}  // namespace

extern "C" int aptp_unet_prologue(const AptpUnetPrologueParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->sample && p->x && p->timesteps && p->freqs && p->t_emb, "unet_prologue: null pointer");
  APTP_CHECK(p->B > 0 && p->C > 0 && p->H > 0 && p->W > 0 && p->cin_pad >= p->C && p->cin_pad % 8 == 0 && p->half > 0,
             "unet_prologue: bad extents (cin_pad must be a multiple of 8 and >= C)");
  APTP_CHECK(((uintptr_t)p->x % 16) == 0, "unet_prologue: x alignment");
  IoK k;
  k.sample = p->sample; k.sample_bf16 = p->sample_bf16; k.x = p->x; k.B = p->B; k.C = p->C; k.HW = p->H * p->W; k.cpad = p->cin_pad;
  k.t = p->timesteps; k.freqs = p->freqs; k.half = p->half; k.temb = p->t_emb;
  const int64_t pix = (int64_t)p->B * k.HW;
  k.pix_blocks = (int)((pix + 255) / 256);
  const int tblocks = (p->B * p->half + 255) / 256;
  hipLaunchKernelGGL(unet_prologue_kernel, dim3(k.pix_blocks + tblocks), dim3(256), 0, (hipStream_t)stream, k);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}

extern "C" int aptp_unet_epilogue(const AptpUnetEpilogueParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->y && p->out, "unet_epilogue: null pointer");
  APTP_CHECK(p->B > 0 && p->C > 0 && p->H > 0 && p->W > 0 && p->ldy >= p->C, "unet_epilogue: bad extents");
  OutK k;
  k.y = p->y; k.ld = p->ldy; k.out = p->out; k.out_bf16 = p->out_bf16; k.B = p->B; k.C = p->C; k.HW = p->H * p->W;
  const int64_t n = (int64_t)k.B * k.C * k.HW;
  hipLaunchKernelGGL(unet_epilogue_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}
