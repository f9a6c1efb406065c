// Backward-path kernels that are not contractions (gfx950): gate multiply + gate gradient, GEGLU forward/backward,
// GroupNorm(+SiLU) backward, LayerNorm backward.  All are HBM-bound streaming kernels over channels-last bf16 with
// the same fixed thread -> channel-octet mapping as norm.hip; per-group reductions are two-stage and deterministic
// (stage 1 writes per-(sample, row-chunk, group) partials, the consumer folds them in a fixed order).
#include "aptp_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void unpack8(const u32x4& q, float* f) {
  union { u32x4 v; __bf16 h[8]; } u;
  u.v = q;
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)u.h[i];
}

__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 q;
  q[0] = pack_bf16x2(f[0], f[1]);
  q[1] = pack_bf16x2(f[2], f[3]);
  q[2] = pack_bf16x2(f[4], f[5]);
  q[3] = pack_bf16x2(f[6], f[7]);
  return q;
}

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // v_rcp_f32, no IEEE division

// ------------------------------------------------------------------------------------------------------------
// Row-chunk layout shared by the streaming kernels below: grid (nchunk, B); block 256 threads; thread -> octet
// o = tid % TPR (+256 per page), row lane rl = tid / TPR (RPAR rows in parallel).
// ------------------------------------------------------------------------------------------------------------
struct RowK {
  int B, HW, C, CO, G, cg, nchunk, TPR, RPAR;
};

__device__ __forceinline__ void fold_channels_to_groups(const RowK& k, float (*red)[6144], int nacc, int CP, int tid,
                                                        float* out /* [G][nacc] */) {
  // TPG (8/4/2 for G <= 32/64/128) threads per group fold channels x RPAR for `nacc` accumulators held in
  // red[a][rl*CP + c]; fixed order => deterministic
  const int tpg = k.G <= 32 ? 8 : (k.G <= 64 ? 4 : 2);
  const int g = tid / tpg, sub = tid % tpg;
  for (int a = 0; a < nacc; ++a) {
    float v = 0.f;
    if (g < k.G) {
      for (int rr = 0; rr < k.RPAR; ++rr)
        for (int i = sub; i < k.cg; i += tpg) v += red[a][rr * CP + g * k.cg + i];
    }
    for (int off = tpg >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (g < k.G && sub == 0) out[g * nacc + a] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------
// gate backward: dx = dy * gate[b % Bg, c / cg];  dgate_partial[b, chunk, g] = sum_{rows in chunk, c in g} dy * y0
// ------------------------------------------------------------------------------------------------------------
struct GateBwdK {
  RowK r;
  const __bf16* dy; int64_t lddy; const __bf16* y0; int64_t ldy0; __bf16* dx; int64_t lddx;
  const float* gate; int gate_B; float* partial;
};

template <int NP>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const GateBwdK p) {
  __shared__ float red[1][6144];
  const RowK& k = p.r;
  const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = (int)(((int64_t)k.HW * chunk) / k.nchunk), r1 = (int)(((int64_t)k.HW * (chunk + 1)) / k.nchunk);
  const int rl = tid / k.TPR, ot = tid - rl * k.TPR;
  const bool active = rl < k.RPAR;
  float acc[NP][8], gm[NP][8];
#pragma unroll
  for (int pg = 0; pg < NP; ++pg) {
    const int o = ot + pg * 256;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      acc[pg][e] = 0.f;
      const int c = o * 8 + e;
      gm[pg][e] = (active && o < k.CO && c < k.C) ? p.gate[(int64_t)(b % p.gate_B) * k.G + c / k.cg] : 0.f;
    }
  }
  if (active) {
    const int64_t row0 = (int64_t)b * k.HW;
    // two rows per trip: all four 16-byte loads of a (page, row pair) are in flight before the arithmetic
    for (int r = r0 + rl; r < r1; r += 2 * k.RPAR) {
      const int rb = r + k.RPAR;
      const bool two = rb < r1;
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {
        const int o = ot + pg * 256;
        if (o < k.CO) {
          u32x4 qd[2], qy[2];
          qd[0] = *reinterpret_cast<const u32x4*>(p.dy + (row0 + r) * p.lddy + o * 8);
          qy[0] = *reinterpret_cast<const u32x4*>(p.y0 + (row0 + r) * p.ldy0 + o * 8);
          qd[1] = two ? *reinterpret_cast<const u32x4*>(p.dy + (row0 + rb) * p.lddy + o * 8) : (u32x4){0u, 0u, 0u, 0u};
          qy[1] = two ? *reinterpret_cast<const u32x4*>(p.y0 + (row0 + rb) * p.ldy0 + o * 8) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            float d[8], y[8], dxv[8];
            unpack8(qd[h], d); unpack8(qy[h], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc[pg][e] += d[e] * y[e]; dxv[e] = d[e] * gm[pg][e]; }
            *reinterpret_cast<u32x4*>(p.dx + (row0 + (h ? rb : r)) * p.lddx + o * 8) = pack8(dxv);
          }
        }
      }
    }
  }
  const int CP = k.TPR * 8 * NP;
  if (active) {
#pragma unroll
    for (int pg = 0; pg < NP; ++pg) {
      const int o = ot + pg * 256;
      if (o < k.CO) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[0][rl * CP + o * 8 + e] = acc[pg][e];
      }
    }
  }
  __syncthreads();
  fold_channels_to_groups(k, red, 1, CP, tid, p.partial + ((int64_t)b * k.nchunk + chunk) * k.G);
}

// ------------------------------------------------------------------------------------------------------------
// GEGLU (training form, un-interleaved): hg = [h | g] of width 2*C each row;  out = (h*m) * gelu_erf(g*m)
//   backward: dh = dout*gelu(gm)*m, dg = dout*hm*gelu'(gm)*m, dm_partial[b,chunk,grp] = sum dout*(gelu(gm)*h + hm*gelu'(gm)*g)
// ------------------------------------------------------------------------------------------------------------
struct GegluK {
  RowK r;                 // C = hidden width (4*dim), G = gate width (32)
  const __bf16* hg; int64_t ldhg; __bf16* out; int64_t ldout;       // forward
  const __bf16* dout; int64_t lddout; __bf16* dhg; int64_t lddhg;   // backward
  const float* gate; int gate_B; float* partial;
};

template <int NP, bool BWD>
__global__ __launch_bounds__(256) void geglu_kernel(const GegluK p) {
  __shared__ float red[1][6144];
  const RowK& k = p.r;
  const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = (int)(((int64_t)k.HW * chunk) / k.nchunk), r1 = (int)(((int64_t)k.HW * (chunk + 1)) / k.nchunk);
  const int rl = tid / k.TPR, ot = tid - rl * k.TPR;
  const bool active = rl < k.RPAR;
  float acc[NP][8], gm[NP][8];
#pragma unroll
  for (int pg = 0; pg < NP; ++pg) {
    const int o = ot + pg * 256;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      acc[pg][e] = 0.f;
      const int c = o * 8 + e;
      gm[pg][e] = (active && o < k.CO && c < k.C) ? (p.gate ? p.gate[(int64_t)(b % p.gate_B) * k.G + c / k.cg] : 1.0f) : 0.f;
    }
  }
  if (active) {
    const int64_t row0 = (int64_t)b * k.HW;
    for (int r = r0 + rl; r < r1; r += k.RPAR) {
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {
        const int o = ot + pg * 256;
        if (o < k.CO) {
          const __bf16* rowp = p.hg + (row0 + r) * p.ldhg;
          const u32x4 qh = *reinterpret_cast<const u32x4*>(rowp + o * 8);
          const u32x4 qg = *reinterpret_cast<const u32x4*>(rowp + k.C + o * 8);
          float h[8], g[8];
          unpack8(qh, h); unpack8(qg, g);
          if (!BWD) {
            float ov[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (h[e] * gm[pg][e]) * gelu_erf_f(g[e] * gm[pg][e]);
            *reinterpret_cast<u32x4*>(p.out + (row0 + r) * p.ldout + o * 8) = pack8(ov);
          } else {
            const u32x4 qd = *reinterpret_cast<const u32x4*>(p.dout + (row0 + r) * p.lddout + o * 8);
            float d[8], dh[8], dg[8];
            unpack8(qd, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float m = gm[pg][e];
              const float gmv = g[e] * m, hmv = h[e] * m;
              float e_half;                                        // exp(-gmv^2 / 2): the density's exponential
              const float cdf = norm_cdf_f(gmv, &e_half);
              const float pdf = 0.39894228040143267794f * e_half;
              const float gelu = gmv * cdf, dgelu = cdf + gmv * pdf;
              const float dhm = d[e] * gelu, dgm = d[e] * hmv * dgelu;
              dh[e] = dhm * m;
              dg[e] = dgm * m;
              acc[pg][e] += dhm * h[e] + dgm * g[e];
            }
            __bf16* drow = p.dhg + (row0 + r) * p.lddhg;
            *reinterpret_cast<u32x4*>(drow + o * 8) = pack8(dh);
            *reinterpret_cast<u32x4*>(drow + k.C + o * 8) = pack8(dg);
          }
        }
      }
    }
  }
  if (BWD) {
    const int CP = k.TPR * 8 * NP;
    if (active) {
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {
        const int o = ot + pg * 256;
        if (o < k.CO) {
#pragma unroll
          for (int e = 0; e < 8; ++e) red[0][rl * CP + o * 8 + e] = acc[pg][e];
        }
      }
    }
    __syncthreads();
    fold_channels_to_groups(k, red, 1, CP, tid, p.partial + ((int64_t)b * k.nchunk + chunk) * k.G);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Second stage of the per-gate reductions: out[bg, g] = sum over the batch rows that share gate row bg (CFG tiling:
// b = rep*gate_B + bg) and over the row chunks of partial[b, chunk, g].  One workgroup per gate row, fixed summation
// order (slice s of S = 256 / Gs threads takes chunks s, s+S, ...; slices are folded 0..S-1 by one thread per g), so the
// result does not depend on scheduling.  Replaces two torch reductions per gate (a strided sum over the chunk axis and
// a sum over the batch repetitions) by one 2-3 us launch inside the same ABI call.
// ------------------------------------------------------------------------------------------------------------
struct FoldK { const float* partial; float* out; int B, nchunk, G, gate_B; };

__global__ __launch_bounds__(256) void fold_partials_kernel(const FoldK p) {
  __shared__ float red[256];
  const int tid = threadIdx.x, bg = blockIdx.x;
  const int Gs = p.G <= 256 ? p.G : 256;            // (G <= 128 by fill_rowk)
  const int S = 256 / Gs;
  const int g = tid % Gs, s = tid / Gs;
  float v = 0.f;
  if (s < S) {
    const int reps = p.B / p.gate_B;
    for (int rep = 0; rep < reps; ++rep) {
      const float* w = p.partial + (int64_t)(rep * p.gate_B + bg) * p.nchunk * p.G + g;
      for (int c = s; c < p.nchunk; c += S) v += w[(int64_t)c * p.G];
    }
  }
  red[tid] = v;
  __syncthreads();
  if (tid < Gs) {
    float a = 0.f;
    for (int q = 0; q < S; ++q) a += red[q * Gs + tid];
    p.out[(int64_t)bg * p.G + tid] = a;
  }
}

void launch_fold(const float* partial, float* out, int B, int nchunk, int G, int gate_B, hipStream_t s) {
  FoldK f{partial, out, B, nchunk, G, gate_B};
  hipLaunchKernelGGL(fold_partials_kernel, dim3(gate_B), dim3(256), 0, s, f);
}

// ------------------------------------------------------------------------------------------------------------
// DepthGate (gates.py:36-42) in its training form: y = (1 - d[b % dB]) * x_in + d[b % dB] * x_out.
//   backward: d_out = dy * d,  d_in = dy * (1 - d),  dd_partial[b, chunk] = sum dy * (x_out - x_in)
// Same row-chunk streaming layout as the gate kernels (one "group" = all channels).
// ------------------------------------------------------------------------------------------------------------
struct LerpK {
  RowK r;
  const __bf16* xin; int64_t ldin; const __bf16* xout; int64_t ldout; __bf16* y; int64_t ldy;
  const __bf16* dy; int64_t lddy; __bf16* din; int64_t lddin; __bf16* dout; int64_t lddout;
  const float* d; int dB; float* partial;
};

template <int NP, bool BWD>
__global__ __launch_bounds__(256) void depth_lerp_kernel(const LerpK p) {
  __shared__ float red[1][6144];
  const RowK& k = p.r;
  const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = (int)(((int64_t)k.HW * chunk) / k.nchunk), r1 = (int)(((int64_t)k.HW * (chunk + 1)) / k.nchunk);
  const int rl = tid / k.TPR, ot = tid - rl * k.TPR;
  const bool active = rl < k.RPAR;
  const float dv = p.d[b % p.dB], iv = 1.0f - dv;
  float acc[NP][8];
#pragma unroll
  for (int pg = 0; pg < NP; ++pg)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[pg][e] = 0.f;
  if (active) {
    const int64_t row0 = (int64_t)b * k.HW;
    for (int r = r0 + rl; r < r1; r += k.RPAR) {
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {
        const int o = ot + pg * 256;
        if (o < k.CO) {
          const u32x4 qi = *reinterpret_cast<const u32x4*>(p.xin + (row0 + r) * p.ldin + o * 8);
          const u32x4 qo = *reinterpret_cast<const u32x4*>(p.xout + (row0 + r) * p.ldout + o * 8);
          float xi[8], xo[8], v[8];
          unpack8(qi, xi); unpack8(qo, xo);
          if (!BWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = iv * xi[e] + dv * xo[e];
            *reinterpret_cast<u32x4*>(p.y + (row0 + r) * p.ldy + o * 8) = pack8(v);
          } else {
            const u32x4 qd = *reinterpret_cast<const u32x4*>(p.dy + (row0 + r) * p.lddy + o * 8);
            float g[8], w[8];
            unpack8(qd, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc[pg][e] += g[e] * (xo[e] - xi[e]); v[e] = g[e] * dv; w[e] = g[e] * iv; }
            *reinterpret_cast<u32x4*>(p.dout + (row0 + r) * p.lddout + o * 8) = pack8(v);
            *reinterpret_cast<u32x4*>(p.din + (row0 + r) * p.lddin + o * 8) = pack8(w);
          }
        }
      }
    }
  }
  if (BWD) {
    const int CP = k.TPR * 8 * NP;
    if (active) {
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {
        const int o = ot + pg * 256;
        if (o < k.CO) {
#pragma unroll
          for (int e = 0; e < 8; ++e) red[0][rl * CP + o * 8 + e] = acc[pg][e];
        }
      }
    }
    __syncthreads();
    fold_channels_to_groups(k, red, 1, CP, tid, p.partial + ((int64_t)b * k.nchunk + chunk));
  }
}

// ------------------------------------------------------------------------------------------------------------
// GroupNorm(+SiLU) backward.  With xh = (x - mean)*rstd, z = xh*gamma + beta, y = act(z):
//   dxh = dy*act'(z)*gamma;  per (b, group): S1 = sum dxh, S2 = sum dxh*xh;  dx = rstd*(dxh - S1/n - xh*S2/n)
// stage 1 (gn_bwd_stats): folds the forward statistics partials, writes (S1, S2) partials per row chunk
// stage 2 (gn_bwd_apply): folds both, writes dx
// ------------------------------------------------------------------------------------------------------------
struct GnBwdK {
  RowK r;
  const __bf16* x; int64_t ldx; const __bf16* dy; int64_t lddy; __bf16* dx; int64_t lddx;
  const float* gamma; const float* beta; float eps; int silu;
  const float* fstats;   // forward partials [B, nchunk, G, 2] (sum, sumsq)
  float* bpart;          // backward partials [B, nchunk, G, 2] (S1, S2)
  float* pgrad;          // optional per-channel partials [B, nchunk, C, 2] (sum dz, sum dz*xh) -> dbeta, dgamma
  const __bf16* add; int64_t ldadd;   // optional: dx += add (the gradient arriving over the residual path that forked at x)
};

__device__ __forceinline__ void fold_forward_stats(const GnBwdK& p, int b, int tid, float* mean_s, float* rstd_s) {
  const RowK& k = p.r;
  const int g = tid >> 3, sub = tid & 7;
  float a = 0.f, a2 = 0.f;
  if (g < k.G) {
    const float* w = p.fstats + ((int64_t)b * k.nchunk * k.G + g) * 2;
    for (int c = sub; c < k.nchunk; c += 8) { a += w[(int64_t)c * k.G * 2]; a2 += w[(int64_t)c * k.G * 2 + 1]; }
  }
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) { a += __shfl_xor(a, off); a2 += __shfl_xor(a2, off); }
  if (g < k.G && sub == 0) {
    const float inv = 1.0f / ((float)k.cg * (float)k.HW);
    const float mean = a * inv;
    float var = a2 * inv - mean * mean;
    var = var < 0.f ? 0.f : var;
    mean_s[g] = mean;
    rstd_s[g] = rsqrtf(var + p.eps);
  }
}

template <int NP, bool APPLY>
__global__ __launch_bounds__(256) void gn_bwd_kernel(const GnBwdK p) {
  __shared__ float red[2][6144];
  __shared__ float mean_s[32], rstd_s[32], s1_s[32], s2_s[32];
  const RowK& k = p.r;
  const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = (int)(((int64_t)k.HW * chunk) / gridDim.x), r1 = (int)(((int64_t)k.HW * (chunk + 1)) / gridDim.x);
  const int rl = tid / k.TPR, ot = tid - rl * k.TPR;
  const bool active = rl < k.RPAR;
  fold_forward_stats(p, b, tid, mean_s, rstd_s);
  if (APPLY) {
    const int g = tid >> 3, sub = tid & 7;
    float a = 0.f, a2 = 0.f;
    if (g < k.G) {
      const float* w = p.bpart + ((int64_t)b * k.nchunk * k.G + g) * 2;
      for (int c = sub; c < k.nchunk; c += 8) { a += w[(int64_t)c * k.G * 2]; a2 += w[(int64_t)c * k.G * 2 + 1]; }
    }
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) { a += __shfl_xor(a, off); a2 += __shfl_xor(a2, off); }
    if (g < k.G && sub == 0) {
      const float inv = 1.0f / ((float)k.cg * (float)k.HW);
      s1_s[g] = a * inv;
      s2_s[g] = a2 * inv;
    }
  }
  __syncthreads();
  float ga[NP][8], be[NP][8], mu[NP][8], rs[NP][8], m1[NP][8], m2[NP][8], a1[NP][8], a2v[NP][8];
#pragma unroll
  for (int pg = 0; pg < NP; ++pg) {
    const int o = ot + pg * 256;
    const int c0 = o * 8;
    int g = c0 / k.cg, rem = c0 - g * k.cg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      const bool ok = active && o < k.CO && c < k.C;
      ga[pg][e] = ok ? p.gamma[c] : 0.f;
      be[pg][e] = ok ? p.beta[c] : 0.f;
      mu[pg][e] = ok ? mean_s[g] : 0.f;
      rs[pg][e] = ok ? rstd_s[g] : 0.f;
      m1[pg][e] = (APPLY && ok) ? s1_s[g] : 0.f;
      m2[pg][e] = (APPLY && ok) ? s2_s[g] : 0.f;
      a1[pg][e] = 0.f; a2v[pg][e] = 0.f;
      if (++rem == k.cg) { rem = 0; ++g; }
    }
  }
  if (active) {
    const int64_t row0 = (int64_t)b * k.HW;
    // two rows per trip: the four 16-byte loads of a (page, row pair) are issued before any of the arithmetic
    for (int r = r0 + rl; r < r1; r += 2 * k.RPAR) {
      const int rb = r + k.RPAR;
      const bool two = rb < r1;
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {
        const int o = ot + pg * 256;
        if (o < k.CO) {
          u32x4 qx[2], qd[2];
          qx[0] = *reinterpret_cast<const u32x4*>(p.x + (row0 + r) * p.ldx + o * 8);
          qd[0] = *reinterpret_cast<const u32x4*>(p.dy + (row0 + r) * p.lddy + o * 8);
          qx[1] = two ? *reinterpret_cast<const u32x4*>(p.x + (row0 + rb) * p.ldx + o * 8) : (u32x4){0u, 0u, 0u, 0u};
          qd[1] = two ? *reinterpret_cast<const u32x4*>(p.dy + (row0 + rb) * p.lddy + o * 8) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            float x[8], d[8], dxv[8];
            unpack8(qx[h], x); unpack8(qd[h], d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float xh = (x[e] - mu[pg][e]) * rs[pg][e];
              float dz = d[e];
              if (p.silu) {
                const float z = xh * ga[pg][e] + be[pg][e];
                const float sg = sigmoid_f(z);
                dz *= sg * (1.0f + z * (1.0f - sg));
              }
              if (APPLY) dxv[e] = rs[pg][e] * (dz * ga[pg][e] - m1[pg][e] - xh * m2[pg][e]);
              else { a1[pg][e] += dz; a2v[pg][e] += dz * xh; }
            }
            if (APPLY) {
              if (p.add) {
                float ad[8];
                unpack8(*reinterpret_cast<const u32x4*>(p.add + (row0 + (h ? rb : r)) * p.ldadd + o * 8), ad);
#pragma unroll
                for (int e = 0; e < 8; ++e) dxv[e] += ad[e];
              }
              *reinterpret_cast<u32x4*>(p.dx + (row0 + (h ? rb : r)) * p.lddx + o * 8) = pack8(dxv);
            }
          }
        }
      }
    }
  }
  if (!APPLY) {
    const int CP = k.TPR * 8 * NP;
    if (active) {
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {
        const int o = ot + pg * 256;
        if (o < k.CO) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            red[0][rl * CP + o * 8 + e] = a1[pg][e];
            red[1][rl * CP + o * 8 + e] = a2v[pg][e];
          }
        }
      }
    }
    __syncthreads();
    // fold the RPAR row lanes per channel; the per-channel sums are the affine-parameter gradients' partials, and,
    // weighted by gamma, the per-group sums S1 = sum dxh, S2 = sum dxh*xh
    for (int c = tid; c < k.C; c += 256) {
      float v0 = 0.f, v1 = 0.f;
      for (int rr = 0; rr < k.RPAR; ++rr) { v0 += red[0][rr * CP + c]; v1 += red[1][rr * CP + c]; }
      if (p.pgrad) {
        float* o = p.pgrad + (((int64_t)b * k.nchunk + chunk) * k.C + c) * 2;
        o[0] = v0; o[1] = v1;
      }
      const float gmm = p.gamma[c];
      red[0][c] = v0 * gmm;
      red[1][c] = v1 * gmm;
    }
    __syncthreads();
    RowK k1 = k;
    k1.RPAR = 1;
    fold_channels_to_groups(k1, red, 2, CP, tid, p.bpart + ((int64_t)b * k.nchunk + chunk) * k.G * 2);
  }
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm backward (data gradient): one wave per row.
// ------------------------------------------------------------------------------------------------------------
struct LnBwdK {
  const __bf16* x; int64_t ldx; const __bf16* dy; int64_t lddy; __bf16* dx; int64_t lddx;
  int rows, C, CO; const float* gamma; float eps;
  const __bf16* add; int64_t ldadd;   // optional: dx += add (residual-path gradient of the fork at x)
};

template <int NO>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdK p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= p.rows) return;
  float x[NO][8], d[NO][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int o = lane + 64 * i;
    if (o < p.CO) {
      unpack8(*reinterpret_cast<const u32x4*>(p.x + (int64_t)row * p.ldx + o * 8), x[i]);
      unpack8(*reinterpret_cast<const u32x4*>(p.dy + (int64_t)row * p.lddy + o * 8), d[i]);
      float g8[8];
      const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + o * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(p.gamma + o * 8 + 4);
      g8[0] = g0.x; g8[1] = g0.y; g8[2] = g0.z; g8[3] = g0.w; g8[4] = g1.x; g8[5] = g1.y; g8[6] = g1.z; g8[7] = g1.w;
#pragma unroll
      for (int e = 0; e < 8; ++e) { sum += x[i][e]; d[i][e] *= g8[e]; }   // d <- dxh = dy*gamma
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[i][e] = 0.f; d[i][e] = 0.f; }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  const float mean = sum / (float)p.C;
  float vs = 0.f;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int o = lane + 64 * i;
    if (o < p.CO) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float c = x[i][e] - mean; vs += c * c; }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) vs += __shfl_xor(vs, off);
  const float rstd = rsqrtf(vs / (float)p.C + p.eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int o = lane + 64 * i;
    if (o < p.CO) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (x[i][e] - mean) * rstd;
        x[i][e] = xh;
        s1 += d[i][e];
        s2 += d[i][e] * xh;
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
  s1 /= (float)p.C; s2 /= (float)p.C;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int o = lane + 64 * i;
    if (o < p.CO) {
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = rstd * (d[i][e] - s1 - x[i][e] * s2);
      if (p.add) {
        float ad[8];
        unpack8(*reinterpret_cast<const u32x4*>(p.add + (int64_t)row * p.ldadd + o * 8), ad);
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] += ad[e];
      }
      *reinterpret_cast<u32x4*>(p.dx + (int64_t)row * p.lddx + o * 8) = pack8(o8);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Column sums of `batch` consecutive [rows, C] bf16 matrices (bias gradients; per-sample sums for the time-embedding bias):
// partial[chunk, b, c] = sum over the rows of chunk `chunk` of matrix b.
// ------------------------------------------------------------------------------------------------------------
struct ColsumK { const __bf16* x; int64_t ldx; int rows, C, CO, nchunk, TPR, RPAR; float* partial; int batch; };

template <int NP>
__global__ __launch_bounds__(256) void colsum_kernel(const ColsumK p) {
  __shared__ float red[6144];
  const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;     // b: sample (rows = rows PER sample)
  const int r0 = (int)(((int64_t)p.rows * chunk) / p.nchunk), r1 = (int)(((int64_t)p.rows * (chunk + 1)) / p.nchunk);
  const int rl = tid / p.TPR, ot = tid - rl * p.TPR;
  const bool active = rl < p.RPAR;
  const __bf16* xb = p.x + (int64_t)b * p.rows * p.ldx;
  float acc[NP][8];
#pragma unroll
  for (int pg = 0; pg < NP; ++pg)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[pg][e] = 0.f;
  if (active) {
    for (int r = r0 + rl; r < r1; r += p.RPAR) {
#pragma unroll
      for (int pg = 0; pg < NP; ++pg) {
        const int o = ot + pg * 256;
        if (o < p.CO) {
          float f[8];
          unpack8(*reinterpret_cast<const u32x4*>(xb + (int64_t)r * p.ldx + o * 8), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[pg][e] += f[e];
        }
      }
    }
  }
  const int CP = p.TPR * 8 * NP;
  if (active) {
#pragma unroll
    for (int pg = 0; pg < NP; ++pg) {
      const int o = ot + pg * 256;
      if (o < p.CO) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rl * CP + o * 8 + e] = acc[pg][e];
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < p.C; c += 256) {
    float v = 0.f;
    for (int rr = 0; rr < p.RPAR; ++rr) v += red[rr * CP + c];
    p.partial[((int64_t)chunk * p.batch + b) * p.C + c] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm affine-parameter gradients: partial[chunk, c, 0] = sum_rows dy (dbeta), [.., 1] = sum_rows dy*xh (dgamma).
// One wave per row (statistics by wave reduction), per-lane channel accumulators, 4 waves folded through LDS.
// ------------------------------------------------------------------------------------------------------------
struct LnPgK { const __bf16* x; int64_t ldx; const __bf16* dy; int64_t lddy; int rows, C, CO, nchunk; float eps; float* partial; };

template <int NO>
__global__ __launch_bounds__(256) void ln_pgrad_kernel(const LnPgK p) {
  __shared__ float red[2][4][2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, chunk = blockIdx.x;
  const int r0 = (int)(((int64_t)p.rows * chunk) / p.nchunk), r1 = (int)(((int64_t)p.rows * (chunk + 1)) / p.nchunk);
  float ab[NO][8], ag[NO][8];
#pragma unroll
  for (int i = 0; i < NO; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ab[i][e] = 0.f; ag[i][e] = 0.f; }
  for (int row = r0 + wave; row < r1; row += 4) {
    float x[NO][8], d[NO][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NO; ++i) {
      const int o = lane + 64 * i;
      if (o < p.CO) {
        unpack8(*reinterpret_cast<const u32x4*>(p.x + (int64_t)row * p.ldx + o * 8), x[i]);
        unpack8(*reinterpret_cast<const u32x4*>(p.dy + (int64_t)row * p.lddy + o * 8), d[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += x[i][e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[i][e] = 0.f; d[i][e] = 0.f; }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)p.C;
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < NO; ++i) {
      const int o = lane + 64 * i;
      if (o < p.CO) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float c = x[i][e] - mean; vs += c * c; }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vs += __shfl_xor(vs, off);
    const float rstd = rsqrtf(vs / (float)p.C + p.eps);
#pragma unroll
    for (int i = 0; i < NO; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) { ab[i][e] += d[i][e]; ag[i][e] += d[i][e] * (x[i][e] - mean) * rstd; }
  }
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int o = lane + 64 * i;
    if (o < p.CO) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[0][wave][o * 8 + e] = ab[i][e]; red[1][wave][o * 8 + e] = ag[i][e]; }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float* o = p.partial + ((int64_t)chunk * p.C + c) * 2;
    o[0] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    o[1] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
  }
}

int fill_rowk(RowK& k, int B, int HW, int C, int G, const char* what) {
  APTP_CHECK(B > 0 && HW > 0 && C > 0 && G > 0 && G <= 128 && C % G == 0, "%s: bad extents (B=%d HW=%d C=%d G=%d)", what, B, HW, C, G);
  k.B = B; k.HW = HW; k.C = C; k.G = G; k.cg = C / G; k.CO = (C + 7) / 8;
  APTP_CHECK(k.CO <= 768, "%s: C too large (max 6144)", what);
  k.nchunk = aptp_groupnorm_nchunk(HW);
  k.TPR = k.CO < 256 ? k.CO : 256;
  k.RPAR = 256 / k.TPR;
  return APTP_OK;
}

}  // namespace

#define ALIGN16(p) (((uintptr_t)(p) % 16) == 0)

extern "C" int aptp_gate_bwd(const AptpGateBwdParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->dy && p->y0 && p->dx && p->gate && p->dgate_partial, "gate_bwd: null pointer");
  GateBwdK k;
  const int rc = fill_rowk(k.r, p->B, p->HW, p->C, p->groups, "gate_bwd");
  if (rc) return rc;
  APTP_CHECK(p->C % 8 == 0 && p->lddy % 8 == 0 && p->ldy0 % 8 == 0 && p->lddx % 8 == 0, "gate_bwd: C and ld must be multiples of 8");
  APTP_CHECK(ALIGN16(p->dy) && ALIGN16(p->y0) && ALIGN16(p->dx) && p->gate_B > 0, "gate_bwd: alignment");
  k.dy = (const __bf16*)p->dy; k.lddy = p->lddy; k.y0 = (const __bf16*)p->y0; k.ldy0 = p->ldy0;
  k.dx = (__bf16*)p->dx; k.lddx = p->lddx; k.gate = p->gate; k.gate_B = p->gate_B; k.partial = p->dgate_partial;
  dim3 grid(k.r.nchunk, p->B);
  if (k.r.CO <= 256) hipLaunchKernelGGL(gate_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, k);
  else if (k.r.CO <= 512) hipLaunchKernelGGL(gate_bwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, k);
  else hipLaunchKernelGGL(gate_bwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, k);
  APTP_LAUNCH_CHECK();
  if (p->dgate) {
    APTP_CHECK(p->B % p->gate_B == 0, "gate_bwd: B must be a multiple of gate_B");
    launch_fold(p->dgate_partial, p->dgate, p->B, k.r.nchunk, p->groups, p->gate_B, (hipStream_t)stream);
    APTP_LAUNCH_CHECK();
  }
  return APTP_OK;
}

extern "C" int aptp_geglu(const AptpGegluParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->hg, "geglu: null pointer");
  GegluK k;
  const int rc = fill_rowk(k.r, p->B, p->HW, p->C, p->groups, "geglu");
  if (rc) return rc;
  APTP_CHECK(p->C % 8 == 0 && p->ldhg % 8 == 0 && p->ldhg >= 2 * p->C && ALIGN16(p->hg), "geglu: C, ld multiples of 8, ldhg >= 2C");
  APTP_CHECK(!p->gate || p->gate_B > 0, "geglu: gate_B");
  k.hg = (const __bf16*)p->hg; k.ldhg = p->ldhg; k.gate = p->gate; k.gate_B = p->gate_B;
  k.out = (__bf16*)p->out; k.ldout = p->ldout;
  k.dout = (const __bf16*)p->dout; k.lddout = p->lddout; k.dhg = (__bf16*)p->dhg; k.lddhg = p->lddhg;
  k.partial = p->dgate_partial;
  dim3 grid(k.r.nchunk, p->B);
  hipStream_t s = (hipStream_t)stream;
  if (!p->backward) {
    APTP_CHECK(p->out && p->ldout % 8 == 0 && p->ldout >= p->C && ALIGN16(p->out), "geglu: out");
    if (k.r.CO <= 256) hipLaunchKernelGGL((geglu_kernel<1, false>), grid, dim3(256), 0, s, k);
    else if (k.r.CO <= 512) hipLaunchKernelGGL((geglu_kernel<2, false>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((geglu_kernel<3, false>), grid, dim3(256), 0, s, k);
  } else {
    APTP_CHECK(p->dout && p->dhg && p->dgate_partial && p->lddout % 8 == 0 && p->lddhg % 8 == 0 && p->lddhg >= 2 * p->C
               && ALIGN16(p->dout) && ALIGN16(p->dhg), "geglu: backward operands");
    if (k.r.CO <= 256) hipLaunchKernelGGL((geglu_kernel<1, true>), grid, dim3(256), 0, s, k);
    else if (k.r.CO <= 512) hipLaunchKernelGGL((geglu_kernel<2, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((geglu_kernel<3, true>), grid, dim3(256), 0, s, k);
  }
  APTP_LAUNCH_CHECK();
  if (p->backward && p->dgate) {
    APTP_CHECK(p->gate && p->B % p->gate_B == 0, "geglu: dgate needs a gate and B a multiple of gate_B");
    launch_fold(p->dgate_partial, p->dgate, p->B, k.r.nchunk, p->groups, p->gate_B, s);
    APTP_LAUNCH_CHECK();
  }
  return APTP_OK;
}

extern "C" int aptp_depth_lerp(const AptpDepthLerpParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->x_in && p->x_out && p->d && p->d_B > 0, "depth_lerp: null pointer");
  LerpK k;
  const int rc = fill_rowk(k.r, p->B, p->HW, p->C, 1, "depth_lerp");
  if (rc) return rc;
  APTP_CHECK(p->C % 8 == 0 && p->ld_in % 8 == 0 && p->ld_out % 8 == 0 && ALIGN16(p->x_in) && ALIGN16(p->x_out) && p->B % p->d_B == 0,
             "depth_lerp: C and ld must be multiples of 8, bases 16-byte aligned, B a multiple of d_B");
  k.xin = (const __bf16*)p->x_in; k.ldin = p->ld_in; k.xout = (const __bf16*)p->x_out; k.ldout = p->ld_out;
  k.y = (__bf16*)p->y; k.ldy = p->ld_y;
  k.dy = (const __bf16*)p->dy; k.lddy = p->ld_dy; k.din = (__bf16*)p->d_in; k.lddin = p->ld_d_in;
  k.dout = (__bf16*)p->d_out; k.lddout = p->ld_d_out;
  k.d = p->d; k.dB = p->d_B; k.partial = p->dd_partial;
  dim3 grid(k.r.nchunk, p->B);
  hipStream_t s = (hipStream_t)stream;
  if (!p->backward) {
    APTP_CHECK(p->y && p->ld_y % 8 == 0 && ALIGN16(p->y), "depth_lerp: y");
    if (k.r.CO <= 256) hipLaunchKernelGGL((depth_lerp_kernel<1, false>), grid, dim3(256), 0, s, k);
    else if (k.r.CO <= 512) hipLaunchKernelGGL((depth_lerp_kernel<2, false>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((depth_lerp_kernel<3, false>), grid, dim3(256), 0, s, k);
    APTP_LAUNCH_CHECK();
  } else {
    APTP_CHECK(p->dy && p->d_in && p->d_out && p->dd_partial && p->dd && p->ld_dy % 8 == 0 && p->ld_d_in % 8 == 0 && p->ld_d_out % 8 == 0
               && ALIGN16(p->dy) && ALIGN16(p->d_in) && ALIGN16(p->d_out), "depth_lerp: backward operands");
    if (k.r.CO <= 256) hipLaunchKernelGGL((depth_lerp_kernel<1, true>), grid, dim3(256), 0, s, k);
    else if (k.r.CO <= 512) hipLaunchKernelGGL((depth_lerp_kernel<2, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((depth_lerp_kernel<3, true>), grid, dim3(256), 0, s, k);
    APTP_LAUNCH_CHECK();
    launch_fold(p->dd_partial, p->dd, p->B, k.r.nchunk, 1, p->d_B, s);
    APTP_LAUNCH_CHECK();
  }
  return APTP_OK;
}

extern "C" int aptp_groupnorm_bwd(const AptpGroupNormBwdParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->x && p->dy && p->dx && p->gamma && p->beta && p->fwd_stats && p->workspace, "groupnorm_bwd: null pointer");
  GnBwdK k;
  const int rc = fill_rowk(k.r, p->B, p->HW, p->C, p->groups, "groupnorm_bwd");
  if (rc) return rc;
  APTP_CHECK(p->groups <= 32 && k.r.CO <= 512, "groupnorm_bwd: groups <= 32, C <= 4096");
  APTP_CHECK(p->ldx % 8 == 0 && p->lddy % 8 == 0 && p->lddx % 8 == 0 && ALIGN16(p->x) && ALIGN16(p->dy) && ALIGN16(p->dx), "groupnorm_bwd: alignment");
  k.x = (const __bf16*)p->x; k.ldx = p->ldx; k.dy = (const __bf16*)p->dy; k.lddy = p->lddy; k.dx = (__bf16*)p->dx; k.lddx = p->lddx;
  k.gamma = p->gamma; k.beta = p->beta; k.eps = p->eps; k.silu = p->silu;
  k.fstats = p->fwd_stats; k.bpart = (float*)p->workspace; k.pgrad = p->pgrad_partial;
  k.add = (const __bf16*)p->add; k.ldadd = p->ldadd;
  APTP_CHECK(!p->add || (p->ldadd % 8 == 0 && ALIGN16(p->add)), "groupnorm_bwd: add alignment");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(k.r.nchunk, p->B);
  if (k.r.CO <= 256) {
    hipLaunchKernelGGL((gn_bwd_kernel<1, false>), grid, dim3(256), 0, s, k);
    hipLaunchKernelGGL((gn_bwd_kernel<1, true>), grid, dim3(256), 0, s, k);
  } else {
    hipLaunchKernelGGL((gn_bwd_kernel<2, false>), grid, dim3(256), 0, s, k);
    hipLaunchKernelGGL((gn_bwd_kernel<2, true>), grid, dim3(256), 0, s, k);
  }
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}

extern "C" int aptp_layernorm_bwd(const AptpLayerNormBwdParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->x && p->dy && p->dx && p->gamma, "layernorm_bwd: null pointer");
  APTP_CHECK(p->rows > 0 && p->C > 0 && p->C % 8 == 0 && p->C <= 2048, "layernorm_bwd: C");
  APTP_CHECK(p->ldx % 8 == 0 && p->lddy % 8 == 0 && p->lddx % 8 == 0 && ALIGN16(p->x) && ALIGN16(p->dy) && ALIGN16(p->dx) && ALIGN16(p->gamma), "layernorm_bwd: alignment");
  LnBwdK k;
  k.x = (const __bf16*)p->x; k.ldx = p->ldx; k.dy = (const __bf16*)p->dy; k.lddy = p->lddy; k.dx = (__bf16*)p->dx; k.lddx = p->lddx;
  k.rows = p->rows; k.C = p->C; k.CO = p->C / 8; k.gamma = p->gamma; k.eps = p->eps;
  k.add = (const __bf16*)p->add; k.ldadd = p->ldadd;
  APTP_CHECK(!p->add || (p->ldadd % 8 == 0 && ALIGN16(p->add)), "layernorm_bwd: add alignment");
  dim3 grid((p->rows + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  switch ((k.CO + 63) / 64) {
    case 1: hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, dim3(256), 0, s, k); break;
    case 2: hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, dim3(256), 0, s, k); break;
    case 3: hipLaunchKernelGGL(ln_bwd_kernel<3>, grid, dim3(256), 0, s, k); break;
    default: hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, dim3(256), 0, s, k); break;
  }
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}

extern "C" int aptp_colsum(const AptpColsumParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->x && p->partial, "colsum: null pointer");
  APTP_CHECK(p->rows > 0 && p->C > 0 && p->C % 8 == 0 && p->C <= 6144 && p->ldx % 8 == 0 && ALIGN16(p->x), "colsum: C multiple of 8 (<= 6144), ld multiple of 8");
  ColsumK k;
  k.x = (const __bf16*)p->x; k.ldx = p->ldx; k.rows = p->rows; k.C = p->C; k.CO = p->C / 8;
  k.batch = p->batch > 0 ? p->batch : 1;
  k.nchunk = k.batch > 1 ? aptp_groupnorm_nchunk(p->rows) : aptp_rows_nchunk(p->rows);
  k.TPR = k.CO < 256 ? k.CO : 256; k.RPAR = 256 / k.TPR; k.partial = p->partial;
  APTP_CHECK(k.batch <= 65535, "colsum: batch");
  dim3 grid(k.nchunk, k.batch);
  hipStream_t s = (hipStream_t)stream;
  if (k.CO <= 256) hipLaunchKernelGGL(colsum_kernel<1>, grid, dim3(256), 0, s, k);
  else if (k.CO <= 512) hipLaunchKernelGGL(colsum_kernel<2>, grid, dim3(256), 0, s, k);
  else hipLaunchKernelGGL(colsum_kernel<3>, grid, dim3(256), 0, s, k);
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}

extern "C" int aptp_layernorm_pgrad(const AptpLayerNormPgradParams* p, aptp_stream_t stream) {
  APTP_CHECK(p && p->x && p->dy && p->partial, "layernorm_pgrad: null pointer");
  APTP_CHECK(p->rows > 0 && p->C > 0 && p->C % 8 == 0 && p->C <= 2048 && p->ldx % 8 == 0 && p->lddy % 8 == 0 && ALIGN16(p->x) && ALIGN16(p->dy), "layernorm_pgrad: C multiple of 8 (<= 2048)");
  LnPgK k;
  k.x = (const __bf16*)p->x; k.ldx = p->ldx; k.dy = (const __bf16*)p->dy; k.lddy = p->lddy;
  k.rows = p->rows; k.C = p->C; k.CO = p->C / 8; k.nchunk = aptp_rows_nchunk(p->rows); k.eps = p->eps; k.partial = p->partial;
  dim3 grid(k.nchunk);
  hipStream_t s = (hipStream_t)stream;
  switch ((k.CO + 63) / 64) {
    case 1: hipLaunchKernelGGL(ln_pgrad_kernel<1>, grid, dim3(256), 0, s, k); break;
    case 2: hipLaunchKernelGGL(ln_pgrad_kernel<2>, grid, dim3(256), 0, s, k); break;
    case 3: hipLaunchKernelGGL(ln_pgrad_kernel<3>, grid, dim3(256), 0, s, k); break;
    default: hipLaunchKernelGGL(ln_pgrad_kernel<4>, grid, dim3(256), 0, s, k); break;
  }
  APTP_LAUNCH_CHECK();
  return APTP_OK;
}
