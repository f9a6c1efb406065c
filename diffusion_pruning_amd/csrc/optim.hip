// AdamW over the packed trainable state of an expert (packed_train.PackedTrainer) as ONE launch, with the bf16 operand the
// GEMM kernels read ("shadow") written by the same pass.  Replaces torch's multi-tensor fused AdamW (34 launches, 3.4 ms at
// 396 M parameters) plus the multi-tensor fp32 -> bf16 cast of the shadows (25 launches, 0.8 ms): per parameter 16 B read
// (p, g, m, v) and 14 B written (p, m, v, shadow) in one pass -- HBM-bound, 11.9 GB per step for this expert.
//
// Same arithmetic as torch.optim.AdamW (decoupled weight decay, bias correction, fp32 throughout; torch/optim/adamw.py,
// the `capturable` fused form: step count read from device memory):
//   p <- p * (1 - lr * wd);  m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2
//   p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The table of tensors lives in device memory (built once; the gradient addresses are those of the captured backward).
#include "aptp_common.h"

namespace {

constexpr int ELEMS_PER_BLOCK = 4096;          // 256 threads x 4 float4

struct AdamK {
  const AptpAdamWItem* items; const int32_t* starts; int n_items;
  float lr, beta1, beta2, eps, wd;
  const float* step;                            // device scalar: number of steps taken BEFORE this one
  const float* gate;                            // optional device scalar: the step is applied only when it is finite
  float gscale;                                 // gradients are multiplied by this first (1 / world size of a summed exchange)
};

__global__ __launch_bounds__(256) void adamw_many_kernel(const AdamK k) {
  const int b = blockIdx.x;
  int lo = 0, hi = k.n_items;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (k.starts[mid] <= b) lo = mid; else hi = mid;
  }
  // NaN guard of a replayed graph (reference: pdm/training/trainer.py:921-929 skips the batch whose backward produced
  // NaNs): a non-finite loss leaves parameters, moments and operands untouched; the caller advances the step count by
  // the same predicate.  Uniform over the grid: one scalar load.
  if (k.gate) {
    const float gv = k.gate[0];
    if (!(fabsf(gv) <= 3.0e38f)) return;
  }
  const AptpAdamWItem it = k.items[lo];
  const float t = k.step[0] + 1.0f;
  const float bc1 = 1.0f - powf(k.beta1, t), bc2 = 1.0f - powf(k.beta2, t);
  const float step_size = k.lr / bc1, bc2_sqrt = sqrtf(bc2), decay = 1.0f - k.lr * k.wd;
  const int64_t base = (int64_t)(b - k.starts[lo]) * ELEMS_PER_BLOCK;
  __bf16* sh = (__bf16*)it.shadow;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t e = base + ((int64_t)i * 256 + threadIdx.x) * 4;
    if (e + 3 < it.n) {
      float4 p = *reinterpret_cast<const float4*>(it.p + e);
      const float4 g = *reinterpret_cast<const float4*>(it.g + e);
      float4 m = *reinterpret_cast<const float4*>(it.m + e);
      float4 v = *reinterpret_cast<const float4*>(it.v + e);
      float* pp = &p.x; const float* g0 = &g.x; float* mm = &m.x; float* vv = &v.x;
      const float gg[4] = {g0[0] * k.gscale, g0[1] * k.gscale, g0[2] * k.gscale, g0[3] * k.gscale};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pp[j] *= decay;
        mm[j] = k.beta1 * mm[j] + (1.0f - k.beta1) * gg[j];
        vv[j] = k.beta2 * vv[j] + (1.0f - k.beta2) * gg[j] * gg[j];
        pp[j] -= step_size * mm[j] / (sqrtf(vv[j]) / bc2_sqrt + k.eps);
      }
      *reinterpret_cast<float4*>(it.p + e) = p;
      *reinterpret_cast<float4*>(it.m + e) = m;
      *reinterpret_cast<float4*>(it.v + e) = v;
      if (sh) {
        uint2 q; q.x = pack_bf16x2(p.x, p.y); q.y = pack_bf16x2(p.z, p.w);
        *reinterpret_cast<uint2*>(sh + e) = q;
      }
    } else {
      for (int64_t x = e; x < it.n && x < e + 4; ++x) {
        float p = it.p[x] * decay;
        const float g = it.g[x] * k.gscale;
        const float m = k.beta1 * it.m[x] + (1.0f - k.beta1) * g;
        const float v = k.beta2 * it.v[x] + (1.0f - k.beta2) * g * g;
        p -= step_size * m / (sqrtf(v) / bc2_sqrt + k.eps);
        it.p[x] = p; it.m[x] = m; it.v[x] = v;
        if (sh) sh[x] = (__bf16)p;
      }
    }
  }
}

}  // namespace

extern "C" int aptp_adamw_blocks(int64_t n) { return n <= 0 ? 0 : (int)((n + ELEMS_PER_BLOCK - 1) / ELEMS_PER_BLOCK); }

extern "C" int aptp_adamw_many(const AptpAdamWParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->items_dev && p->starts_dev && p->step_dev && p->n_items >= 1 && p->total_blocks >= 1, "adamw_many: bad arguments");
  APTP_CHECK(p->lr >= 0.f && p->beta1 >= 0.f && p->beta1 < 1.f && p->beta2 >= 0.f && p->beta2 < 1.f && p->eps > 0.f, "adamw_many: hyper-parameters");
  APTP_CHECK(p->grad_scale >= 0.f, "adamw_many: grad_scale");
  AdamK k{p->items_dev, p->starts_dev, p->n_items, p->lr, p->beta1, p->beta2, p->eps, p->weight_decay, p->step_dev,
          p->gate_dev, p->grad_scale == 0.f ? 1.0f : p->grad_scale};
  hipLaunchKernelGGL(adamw_many_kernel, dim3((unsigned)p->total_blocks), dim3(256), 0, (hipStream_t)stream, k);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}
