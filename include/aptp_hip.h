/*
 * aptp_hip.h — C ABI of libaptp_hip.so: the MI355X (gfx950) kernels behind APTP's masked-U-Net denoising path.
 *
 * The reference (rezashkv/diffusion_pruning) is pure Python and has no FFI boundary of its own; every entry point
 * below replaces a *call site into torch/diffusers* inside the reference's gated blocks.  Citations are
 * file:line relative to the reference tree.
 *
 * Conventions
 *   - All pointers are DEVICE pointers owned by the caller (PyTorch allocates everything, including workspaces).
 *   - Activations are bf16, channels-last: a conv input is [B, H, W, C] with a row (pixel) stride `ld*` in
 *     ELEMENTS, so strided channel-slices of wider buffers can be read/written in place (no torch.cat copies).
 *     A linear layer is the KH=KW=1 case with H = tokens per sample, W = 1.
 *   - Weights are pre-packed bf16 [N][KH*KW][Cin_pad] (K-contiguous), Cin_pad = ceil(Cin/64)*64, zero padded.
 *   - Epilogue vectors (bias, gates, corrections, depth) are fp32.
 *   - Every launch is asynchronous on `stream`; no hidden sync, no global state => hipGraph-capturable.
 *   - Return 0 on success, a negative APTP_E* code otherwise; aptp_last_error() gives a thread-local message.
 */
#ifndef APTP_HIP_H
#define APTP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* aptp_stream_t; /* hipStream_t */

enum {
  APTP_OK = 0,
  APTP_EINVAL = -1,   /* bad shape / alignment / null pointer */
  APTP_ELAUNCH = -2,  /* hipLaunch failed */
  APTP_EWORKSPACE = -3
};

enum { APTP_ACT_NONE = 0, APTP_ACT_SILU = 1, APTP_ACT_GEGLU = 2 };

/*
 * Implicit-GEMM convolution / linear:  y[m, n] = epilogue( sum_{tap,c} x[pix(m,tap), c] * w[n, tap, c] )
 *   m = (b, oy, ox) over B*Hout*Wout,  pix = (b, (oy*stride - pad + ky) >> u, (ox*stride - pad + kx) >> u)
 *   ups = 0: u = 0;  ups = 1: u = 1 (nearest x2 upsample folded into the gather);  ups = 2: u = 1 and odd coordinates read
 *   zero (zero-insertion x2 = the data-gradient of a stride-2 convolution, with flipped/transposed packed weights)
 * Replaces: F.conv2d in ResnetBlock2D*.conv1/conv2/conv_shortcut (pdm/models/unet/blocks.py:331,362,364-367,
 * 537,568,570-573), Downsample2D/Upsample2D convs (inherited diffusers; nearest-x2 folded via `ups`),
 * conv_in/conv_out (pdm/models/unet/unet_2d_conditional.py:1614,1721), and every F.linear of the transformer
 * path: to_q/to_k/to_v/to_out (blocks.py:228-240,266-268), GEGLU proj (blocks.py:43), ff.net[2], proj_in/proj_out
 * (blocks.py:1239-1243,1301-1305), time_emb_proj (blocks.py:336-340), time_embedding
 * (unet_2d_conditional.py:1519).
 * Epilogue, in this order (each optional):
 *   v += bias[n]; v += rowbias[b, n]                      (conv bias; + time_emb_proj(SiLU(temb)), blocks.py:342-343)
 *   v *= colgate[b % gate_B, n / gate_group]              (WidthGate, gates.py:15-21; blocks.py:345-348, 250-255)
 *   act: SiLU, or GEGLU on (h,g) column pairs: h * gelu_erf(g)   (blocks.py:41-50; gate applied to both halves)
 *   v += corr[(b % corr_B), border_class(oy,ox), n]       (gated-vs-pruned GroupNorm-beta term)
 *   v += residual[m, n]                                   (blocks.py:369, 799, 818, 849, 1308)
 *   v = (1-d[b % depth_B]) * depth_in[m, n] + d * v       (DepthGate, gates.py:36-42; blocks.py:577-582,1345-1348)
 */
typedef struct {
  const void* x;        /* bf16 [B, Hin, Win, >=Cin], row stride ldx */
  int64_t ldx;
  int32_t B, Hin, Win, Cin;
  int32_t Hout, Wout;
  int32_t KH, KW, stride, pad, ups;
  const void* w;        /* bf16 packed [N][KH*KW][Cin_pad] */
  int32_t N;            /* GEMM N (for GEGLU: 2*hidden, packed in [16 h | 16 g] column blocks) */
  int32_t cin_pad;      /* ceil(Cin/64)*64 */
  const float* bias;    /* [N] or NULL */
  const float* rowbias; /* [B, ld_rowbias] or NULL */
  int32_t ld_rowbias;
  const float* colgate; /* [gate_B, Nlogical/gate_group] or NULL */
  int32_t gate_group, gate_B;
  int32_t act;          /* APTP_ACT_* */
  const float* corr;    /* [corr_B, 9, N] or NULL; class = 3*rowclass + colclass (0 first / 1 interior / 2 last row or column), class 4 = interior */
  int32_t corr_B;
  const void* residual; /* bf16 [M, ldres] or NULL */
  int64_t ldres;
  const float* depth;   /* [depth_B] or NULL */
  int32_t depth_B;
  const void* depth_in; /* bf16 [M, lddin] */
  int64_t lddin;
  void* y;              /* bf16 (or fp32 if out_f32) [M, ldy]; GEGLU writes N/2 columns */
  int64_t ldy;
  int32_t out_f32;
  int32_t split_k;      /* >=1; >1 needs workspace of aptp_conv_gemm_workspace_bytes() */
  void* workspace;
  int32_t tile;         /* 0 = auto; otherwise APTP_TILE_* (testing / tuning) */
  int32_t order;        /* workgroup -> tile order over the 8 XCDs: 0 = auto (partition the larger operand), 1 = legacy
                         * (N-tile fastest), 2 = weight-major, 3 = activation-major (testing / tuning) */
  /* LayerNorm folded into the neighbouring GEMMs (nn.LayerNorm norm1/norm2/norm3 of BasicTransformerBlock,
   * blocks.py:782,808-810,821; inference path).  The GEMM that PRODUCES the normalised tensor (proj_in, to_out with its
   * residual) also emits, per output row, fp32 (sum, sumsq) partials of the bf16 values it stores, one per (N-tile, wave
   * column): rowstat_out [rowstat_slots / 2, M, 2, 2] (two slots per 16-byte element; the slot count is always even),
   * rowstat_slots = aptp_conv_gemm_rowstat_slots(); needs split_k == 1 or tile_counters, a bf16
   * non-GEGLU output.  The GEMM that CONSUMES LayerNorm(x) reads x itself with gamma folded into its packed weights
   * (w' = w * gamma) and finishes the normalisation in its epilogue, before bias:
   *     v = rstd[m] * (acc[m, n] - mean[m] * ln_colsum[n]),   mean/rstd from ln_stats [ln_slots / 2, M, 2, 2] over ln_C channels,
   * ln_colsum[n] = sum_k bf16(w'[n, k]) (fp32), and the caller adds sum_k beta[k] w[n, k] to bias[n]. */
  float* rowstat_out;
  int32_t rowstat_slots;
  const float* ln_stats;
  int32_t ln_slots;
  const float* ln_colsum;
  float ln_eps;
  int32_t ln_C;
  /* GroupNorm statistics emitted by the producer (F.group_norm of ResnetBlock2D.norm1 / norm2, Transformer2DModel.norm,
   * conv_norm_out: blocks.py:296-301,350-359,1227; unet_2d_conditional.py:1718): per channel, fp32 (sum, sumsq) of the
   * bf16 values stored for each block of R = aptp_conv_gemm_colstat_rows() consecutive output rows, colstat_out
   * [ceil(M/R) rounded up to whole tiles, colstat_ld, 2].  Needs Hout*Wout % R == 0 (a block never straddles samples),
   * a bf16 non-GEGLU output with 16-byte aligned rows, split_k == 1 or the in-kernel reduction.  aptp_groupnorm consumes
   * them through AptpGroupNormParams.colstats and skips its own statistics pass. */
  float* colstat_out;
  int32_t colstat_ld;
  int32_t* tile_counters; /* optional, split_k > 1: >= aptp_conv_gemm_tiles() int32 words, ZERO on entry and left zero: the
                           * K-slices are then combined inside the launch by the last-arriving workgroup of each output tile
                           * (one agent-scope release / acquire per tile, slabs re-read in slice order: deterministic) and no
                           * reduce kernel is launched; words may be shared by launches of one stream */
  /* optional second operand: one more K-segment after the KH*KW filter taps, read AT the output pixel (a 1x1 "tap") over
   * the Cin2 channels of x2 [B, Hout, Wout, >=Cin2] (row stride ldx2); the packed weights are then
   * [N][KH*KW*cin_pad + cin2_pad].  Fuses ResnetBlock2D.conv_shortcut (blocks.py:364-367,570-573) into conv2:
   * conv2(h) + conv_shortcut(x) is one GEMM with K = 9*C_mid + C_in, the shortcut never round-trips through memory and
   * its bias is added to conv2's.  Needs stride 1, no upsampling, Hout == Hin, Wout == Win. */
  const void* x2;
  int64_t ldx2;
  int32_t Cin2, cin2_pad;
  /* optional: [prefetch, prefetch + prefetch_bytes) = an operand of the NEXT launch (its packed weights).  The workgroups
   * touch it at kernel entry (one dword per 64-byte line, results discarded), each XCD its eighth of the buffer -- the rows
   * the weight-major order of the next launch gives that XCD -- so that the weights are in the right L2 when the next launch
   * starts: every weight of the U-Net is read once per forward and would otherwise be fetched cold.  Pays for buffers that
   * fit the L2s (<= ~12 MB); 4-byte aligned. */
  const void* prefetch;
  int64_t prefetch_bytes;
  int32_t epilogue;     /* 0 = auto (coalesced 16-byte stores through an LDS transpose when y / residual / depth_in rows are
                         * 16-byte aligned), 1 = force the accumulator-layout epilogue (testing / tuning), 2 = keep a plain linear layer on the
                         * general implicit-GEMM kernel instead of the lean one of csrc/lin_gemm.hip (testing: the two must agree) */
  /* optional, small maps (Hout*Wout <= 256), split_k > 1 without tile_counters: the reduce launch of the split also applies
   * the GroupNorm(+SiLU) that follows this convolution (ResnetBlock2D: conv1 + time_emb_proj -> norm2 -> SiLU,
   * blocks.py:331-359): one workgroup per (sample, group) sums the K-slices of its Hout*Wout x (gn_C / gn_groups) columns,
   * adds bias / rowbias, rounds to bf16 (the value the separate launches would have stored), takes the group's statistics,
   * and writes y = act(gamma * (h - mean) * rstd + beta) -- y is then the NORMALISED tensor; the convolution output itself
   * is never stored.  Needs act == NONE and no colgate / corr / residual / depth / statistics outputs, gn_C % 8 == 0,
   * (gn_C / gn_groups) % 4 == 0; columns >= gn_C of y are written as zeros. */
  const float* gn_gamma;  /* [gn_C] or NULL (= feature off) */
  const float* gn_beta;
  int32_t gn_groups, gn_C, gn_silu;
  float gn_eps;
  /* fp32 PARITY instantiation (never benchmarked): x, w (packed fp32 [N][KH*KW][Cin_pad]), x2, residual, depth_in and y are fp32
   * tensors (out_f32 must be 1; row strides in elements as always), the contraction runs on exact-fp32 MFMAs
   * (v_mfma_f32_16x16x4_f32) through the SAME gather / tap walk / zero padding / split-K / epilogue code as the bf16
   * register-staged kernel (tiles 1..6 only).  Pins addressing and epilogue order on the GPU against the fp32 oracle at 1e-5
   * per op; the reference computes in fp32 (configs/pruning/sd-2-1_cc3m.yaml:79). */
  int32_t io_f32;
  /* round 4, optional, together with colstat_out: GroupNorm statistics the consumer can use WITHOUT a finalise launch.  Every
   * wave that emits column statistics also adds, per channel unit of `ustat_unit` consecutive output channels it touches, the
   * (sum, sum of squares) of its stored values to ustat_out[rep][sample][unit][2] -- int64 fixed point (sum * 2^20, squares *
   * 2^12), 64-bit integer atomics: the totals do not depend on the order of arrival (deterministic), `ustat_nrep` (a power of
   * two) replicas selected by the workgroup index spread the contention.  The buffer must be ZERO on entry
   * (nrep * B * ustat_units * 2 words, ustat_units = ceil(N_out / ustat_unit)); nobody resets it. */
  void* ustat_out;
  int32_t ustat_unit, ustat_units, ustat_nrep;
} AptpConvGemmParams;

enum { APTP_TILE_AUTO = 0, APTP_TILE_128x128 = 1, APTP_TILE_128x160 = 2, APTP_TILE_64x128 = 3, APTP_TILE_64x160 = 4,
       APTP_TILE_128x64 = 5, APTP_TILE_64x64 = 6,
       /* same tiles, operands copied global->LDS by LDS-DMA instead of through registers */
       APTP_TILE_DMA_128x128 = 7, APTP_TILE_DMA_128x160 = 8, APTP_TILE_DMA_64x128 = 9, APTP_TILE_DMA_64x160 = 10,
       APTP_TILE_DMA_128x64 = 11, APTP_TILE_DMA_64x64 = 12,
       /* LDS-DMA with a 3-deep ring (DMA two K-steps ahead, counted vmcnt + raw barrier) */
       APTP_TILE_DMA3_128x128 = 13, APTP_TILE_DMA3_128x160 = 14, APTP_TILE_DMA3_64x128 = 15, APTP_TILE_DMA3_64x160 = 16,
       APTP_TILE_DMA3_128x64 = 17, APTP_TILE_DMA3_64x64 = 18,
       /* LDS-DMA, 8-wave (512-thread) workgroups: one weight tile shared by twice the rows */
       APTP_TILE_DMA8_128x160 = 19, APTP_TILE_DMA8_256x160 = 20, APTP_TILE_DMA8_128x128 = 21, APTP_TILE_DMA8_256x128 = 22,
       /* LDS-DMA, 4 waves, 4-stage ring (three operand tiles in flight) */
       APTP_TILE_DMA4_64x160 = 23, APTP_TILE_DMA4_64x128 = 24, APTP_TILE_DMA4_64x64 = 25, APTP_TILE_DMA4_128x64 = 26,
       APTP_TILE_DMA4_128x128 = 27,
       /* LDS-DMA, 8 waves, 3- or 4-stage ring */
       APTP_TILE_DMA8R3_128x160 = 28, APTP_TILE_DMA8R4_128x160 = 29, APTP_TILE_DMA8R3_128x128 = 30,
       APTP_TILE_DMA8R4_128x128 = 31, APTP_TILE_DMA8R3_256x128 = 32,
       /* LDS-DMA, 8 waves in two groups that alternate load and MFMA slots (ping-pong), 3- or 4-stage ring */
       APTP_TILE_PP3_128x160 = 33, APTP_TILE_PP4_128x160 = 34, APTP_TILE_PP3_128x128 = 35, APTP_TILE_PP4_128x128 = 36,
       APTP_TILE_PP4_64x160 = 37, APTP_TILE_PP4_64x128 = 38, APTP_TILE_PP5_64x160 = 39, APTP_TILE_PP4_128x64 = 40,
       APTP_TILE_PP3_256x128 = 41, APTP_TILE_PP3_128x256 = 42,
       /* 3x3 / stride-1 / pad-1 only, image width 16 / 32 / 64: the input halo of 128 output pixels (whole image rows) is
        * kept in LDS across the nine taps, weights stream through a 4-stage ring (8 waves) */
       APTP_TILE_HALO_128x160 = 43, APTP_TILE_HALO_128x128 = 44,
       /* LDS-DMA, 4 waves, 6- or 8-stage ring (96-144 KB of LDS, one workgroup per CU): launches with few workgroups and a
        * long K whose time is K-steps x operand latency / tiles in flight */
       APTP_TILE_DMA6_64x64 = 45, APTP_TILE_DMA8S_64x64 = 46, APTP_TILE_DMA6_64x128 = 47, APTP_TILE_DMA6_128x64 = 48,
       /* LDS-DMA ring with TWO 64-wide K-tiles per barrier (an effective K-step of 128), 4- or 6-stage ring; 4 waves, the
        * last two 8 waves */
       APTP_TILE_KU2S4_64x64 = 49, APTP_TILE_KU2S6_64x64 = 50, APTP_TILE_KU2S4_64x128 = 51, APTP_TILE_KU2S6_64x128 = 52,
       APTP_TILE_KU2S4_128x64 = 53, APTP_TILE_KU2S4_128x128 = 54, APTP_TILE_KU2S4_64x160 = 55, APTP_TILE_KU2S4_128x160 = 56,
       APTP_TILE_KU2S4_128x128W8 = 57,
       /* LDS-DMA ring, 8 waves as TWO copies of a 2 x 2 wave grid: copy 0 / 1 multiplies the first / second 32-wide half
        * of every K-tile (intra-workgroup split-K, summed through LDS before the epilogue); 3- or 4-stage ring */
       APTP_TILE_KS2S3_64x64 = 58, APTP_TILE_KS2S4_64x64 = 59, APTP_TILE_KS2S3_64x128 = 60, APTP_TILE_KS2S4_64x128 = 61,
       APTP_TILE_KS2S3_128x64 = 62, APTP_TILE_KS2S3_128x128 = 63,
       /* persistent stream-K macro-tiles (conv_gemm_sk.hip): one 8-wave workgroup per CU walks an equal share of the
        * launch's (tile, K-step) units; split_k > 1 = K split on (needs workspace + tile_counters: partial tiles are
        * combined in-kernel by the last arriver), split_k == 1 = whole tiles only.  Two wave groups alternate an LDS /
        * load slot and a register-only MFMA slot per 32-deep K half; 3-stage LDS-DMA ring filled in half-tiles */
       APTP_TILE_SK_256x160 = 64, APTP_TILE_SK_256x128 = 65, APTP_TILE_SK_128x256 = 66,
       /* the same with a phase's fragment reads in its own load slot (beside the OTHER group's MFMAs) instead of one phase
        * ahead beside the wave's own MFMAs: one register set, shorter DMA lead (tuner candidates) */
       APTP_TILE_SKL_256x160 = 67, APTP_TILE_SKL_256x128 = 68, APTP_TILE_SKL_128x256 = 69,
       /* round 4: the same persistent stream-K kernels on 128-row tiles -- the tile sizes the batch-4 launches fill the chip with
        * (128 tiles of 128 x 160 at level 64 = one two-way split per tile, like the ping-pong tile's own split-K), with the
        * macro-tile kernel's per-lane offset tables and two-slot schedule instead of the per-step tap logic */
       APTP_TILE_SK_128x160 = 70, APTP_TILE_SKL_128x160 = 71, APTP_TILE_SK_128x128 = 72, APTP_TILE_SKL_128x128 = 73 };

int aptp_conv_gemm(const AptpConvGemmParams* p, aptp_stream_t stream);
int64_t aptp_conv_gemm_workspace_bytes(const AptpConvGemmParams* p);
/* rows per statistics block of colstat_out for the tile the launch will use */
int aptp_conv_gemm_colstat_rows(const AptpConvGemmParams* p);
/* output tiles of the launch (size of tile_counters) */
int aptp_conv_gemm_tiles(const AptpConvGemmParams* p);
/* number of row-statistics slots a launch of this problem writes per row (N-tiles x wave columns of the tile it will use) */
int aptp_conv_gemm_rowstat_slots(const AptpConvGemmParams* p);
/* heuristic split-K the library would choose for this problem (host helper, no launch) */
int aptp_conv_gemm_suggest_split_k(const AptpConvGemmParams* p);

/*
 * GroupNorm (+ optional SiLU) over channels-last bf16:  y = act((x - mean_g) * rstd_g * gamma + beta)
 * Replaces F.group_norm + F.silu of ResnetBlock2D.norm1/norm2 + nonlinearity (blocks.py:296-301,350-359),
 * Transformer2DModel.norm (blocks.py:1227, eps 1e-6, no SiLU) and conv_norm_out + conv_act
 * (unet_2d_conditional.py:1718-1720).  `groups` groups of C/groups consecutive channels; statistics over
 * (C/groups)*HW elements per (sample, group), biased variance, fp32 accumulation.
 * C need not be a multiple of 8 (compacted channel counts such as 17 groups x 10): rows are processed in 8-channel
 * octets, ld >= roundup8(C), and channels [C, roundup8(C)) of y are written as exact zeros.
 * workspace: aptp_groupnorm_workspace_bytes() = fp32 [B, nchunk, groups, 2] (sum, sumsq) partials, nchunk =
 * aptp_groupnorm_nchunk(HW), followed by the finalised [B, groups, 2] (mean, rstd).
 */
/* one channel segment of producer-emitted statistics (AptpConvGemmParams.colstat_out) */
typedef struct {
  const float* stats;       /* [B*HW/rows_per_block (+ padding blocks), ld, 2] or NULL */
  int32_t ld, rows_per_block, C;
  /* round 4, optional: the same producer also left per-(sample, channel UNIT) sums (AptpConvGemmParams.ustat_out): int64
   * fixed point [nrep][B][units][2] (sum * 2^20, sum of squares * 2^12), `unit` channels per unit.  When every segment has
   * them, C of every segment and C / groups are multiples of `unit`, the apply pass finishes mean / rstd itself and NO
   * finalise launch runs (one launch per GroupNorm). */
  const void* ustats;
  int32_t unit, units, nrep;
} AptpGroupNormColStats;

typedef struct {
  const void* x; int64_t ldx;   /* bf16 [B*HW, >=C] */
  void* y; int64_t ldy;         /* bf16 [B*HW, >=C] */
  int32_t B, HW, C, groups;
  const float* gamma; const float* beta; /* [C] */
  float eps;
  int32_t silu;
  void* workspace;
  /* 0 = auto; 1 = three launches (stats, finalise, apply), which fill the workspace partials aptp_groupnorm_bwd consumes;
   * 2 = one launch, each workgroup owning whole groups of one sample (small maps);
   * 3 = two launches: at most 16 coarse statistics chunks, folded by every apply workgroup (no finalise launch);
   * 4 = auto WITH those partials: the small maps still take one launch (the owning workgroup writes its groups' sums into
   *     chunk 0 and zeros into the other chunks), everything else the three-launch form. */
  int32_t variant;
  /* optional, three-launch form only: >= B int32 words, ZERO on entry and left zero.  The last statistics workgroup of
   * each sample then folds the partials itself (write-through partial stores, one agent-scope acquire) and the
   * finalise launch is skipped: two launches instead of three. */
  int32_t* counters;
  /* optional: the statistics were emitted by the GEMM(s) that produced x -- one segment, or two for a skip-concat
   * (segment 0 = channels [0, C0), segment 1 = [C0, C)).  The statistics pass over x is skipped: a (groups x B)-workgroup
   * finalise from the partials, then the apply pass (two launches, x read once). */
  AptpGroupNormColStats colstats[2];
  /* fp32 PARITY path (csrc/parity_f32.hip; never benchmarked): x and y are fp32, three plain launches with the product's
   * statistics layout and variance formula; no producer statistics, no fused finalize */
  int32_t io_f32;
} AptpGroupNormParams;

int aptp_groupnorm(const AptpGroupNormParams* p, aptp_stream_t stream);
int aptp_groupnorm_nchunk(int HW);
int aptp_rows_nchunk(int rows);   /* row chunks of the kernels without a batch dimension (aptp_layernorm_pgrad, aptp_colsum with batch <= 1) */
int64_t aptp_groupnorm_workspace_bytes(const AptpGroupNormParams* p);

/*
 * LayerNorm over the last dim of bf16 [rows, C]: replaces nn.LayerNorm norm1/norm2/norm3 of
 * BasicTransformerBlock (blocks.py:782,808-810,821); eps 1e-5, affine, two-pass fp32 statistics.
 */
typedef struct {
  const void* x; int64_t ldx;
  void* y; int64_t ldy;
  int32_t rows, C;
  const float* gamma; const float* beta;
  float eps;
  int32_t io_f32;   /* fp32 PARITY path: x and y are fp32 (csrc/parity_f32.hip) */
} AptpLayerNormParams;

int aptp_layernorm(const AptpLayerNormParams* p, aptp_stream_t stream);

/*
 * Scaled-dot-product attention, head_dim 64, no mask, no dropout: o = softmax(q k^T * scale) v per (batch, head).
 * Replaces F.scaled_dot_product_attention in HeadGatedAttnProcessor2 (blocks.py:258-260).
 * q/k/v/o are bf16 with layout [B, L, heads, 64] expressed through strides (elements): element (b, l, h, d) at
 * ptr + b*stride_b + l*stride_l + h*64 + d, so fused QKV / KV GEMM outputs are consumed in place.
 */
typedef struct {
  const void* q; int64_t q_stride_b, q_stride_l;
  const void* k; int64_t k_stride_b, k_stride_l;
  const void* v; int64_t v_stride_b, v_stride_l;
  void* o; int64_t o_stride_b, o_stride_l;
  int32_t B, heads, Lq, Lk;
  float scale;
  float* lse;   /* optional fp32 [B, heads, Lq]: log2-domain log-sum-exp of the scaled scores, consumed by aptp_attention_bwd */
  int32_t variant; /* 0 = auto; 1 = run the two key-range wave groups of the 8-wave kernel one phase apart (double-buffered
                    * K/V; measured slower than the lock-step form, kept for testing / A-B timing); 2 = force four key-range
                    * groups (16 waves); 3 = force two lock-step groups with single-buffered K/V (two barriers per key tile);
                    * 4 = force two lock-step groups with double-buffered K/V (one barrier per key tile; auto from 32 key
                    * tiles); 5 = force the 4-wave kernel (one group); 6 = the software-pipelined two-group kernel (round 4:
                    * scores of key tile i+1 next to the softmax of tile i, V read transposed in hardware), which is also
                    * what auto takes whenever Lq % 128 == 0, Lk % 128 == 0, Lk >= 256 -- bit-identical to variant 4.
                    * An explicit variant overrides the occupancy rule that otherwise chooses between one and two groups */
  int32_t io_f32;  /* fp32 PARITY path: q, k, v, o are fp32 (strides in elements), no lse (csrc/parity_f32.hip) */
} AptpAttentionParams;

int aptp_attention(const AptpAttentionParams* p, aptp_stream_t stream);

/*
 * The two non-GEMM ends of UNet2DConditionModel.forward (unet_2d_conditional.py:1497-1519 time embedding input,
 * :1614 conv_in input layout, :1721-1726 output): one launch each instead of ~10 elementwise kernels.
 *   prologue: sample [B, C, H, W] (fp32, or bf16 when sample_bf16) -> x bf16 [B, H, W, cin_pad] (padding channels zero);
 *             t_emb[b] = [cos(timesteps[b] * freqs[k]) | sin(timesteps[b] * freqs[k])], k < half, as bf16 [B, 2*half]
 *             (diffusers get_timestep_embedding with flip_sin_to_cos = True).
 *   epilogue: y fp32 [B, H, W, ldy] (conv_out) -> out [B, C, H, W] (fp32, or bf16 when out_bf16).
 */
typedef struct {
  const void* sample; int32_t sample_bf16;
  void* x;
  int32_t B, C, H, W, cin_pad;
  const float* timesteps;   /* fp32 [B] */
  const float* freqs;       /* fp32 [half] */
  int32_t half;
  void* t_emb;              /* bf16 [B, 2*half] */
} AptpUnetPrologueParams;

typedef struct {
  const float* y; int64_t ldy;
  void* out; int32_t out_bf16;
  int32_t B, C, H, W;
} AptpUnetEpilogueParams;

int aptp_unet_prologue(const AptpUnetPrologueParams* p, aptp_stream_t stream);
int aptp_unet_epilogue(const AptpUnetEpilogueParams* p, aptp_stream_t stream);

/* Fused tail of a transformer block on the large-M levels (diffusers BasicTransformerBlock.norm3 -> ff (GEGLUGated +
 * Linear, pdm/models/unet/blocks.py:41-50,121-129,821-823) -> "+ hidden_states", then Transformer2DModel.proj_out and its
 * "+ residual", blocks.py:1294-1308) as ONE kernel per 64-token tile:
 *   y = proj_out(h + ff2(GEGLU(LN3(h) W1'))) + x,   optionally with the column statistics of y for the next GroupNorm
 * (colstat [M/64][colstat_ld][2] = per (64-row block, channel) (sum, sumsq) of the bf16 outputs: AptpGroupNormParams.colstats
 * with rows_per_block = 64).  w1 = the GEGLU projection packed with the LayerNorm folded in and (value | gate) rows interleaved
 * in blocks of 16 (ops.pack_weight(geglu=True, ln_gamma=, ln_beta=)), b1 its bias', cs1 its column sums; w2 = ff.net[2] packed
 * [C][ld2]; w3 = proj_out packed [C][ld3].  Requires aptp_ff_tail_supported(...) != 0 (M % 64 == 0, C in {64..320} a multiple
 * of 64, ld1 == ld3 == C, ld2 % 64 == 0, n1 % 32 == 0). */
typedef struct {
  const void* h; int64_t ldh;      /* bf16 [M, C]: the residual stream after the cross-attention */
  const void* x; int64_t ldx;      /* bf16 [M, C]: the transformer's input (residual of proj_out) */
  void* y; int64_t ldy;            /* bf16 [M, C] */
  const void* w1; const float* b1; const float* cs1; int32_t n1, ld1;
  const void* w2; const float* b2; int32_t ld2;
  const void* w3; const float* b3; int32_t ld3;
  float* colstat; int32_t colstat_ld;
  int32_t M, C;
  float eps;
} AptpFfTailParams;
int aptp_ff_tail_supported(int M, int C, int n1, int ld1, int ld2, int ld3);
int aptp_ff_tail(const AptpFfTailParams* p, aptp_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Backward path (data gradients + gate gradients; the U-Net weights are frozen in APTP's pruning step,
 * pdm/training/trainer.py:742,827-829).  Data gradients of convolutions / linears reuse aptp_conv_gemm with
 * flipped-transposed packed weights (ups = 2 for the stride-2 downsamplers).  What autograd derives in the reference
 * from F.group_norm / F.silu / F.layer_norm / F.gelu / SDPA / the gate multiplies (blocks.py:41-50,250-260,296-359,
 * 782-821; gates.py:15-21) is implemented by the entry points below.
 * ------------------------------------------------------------------------------------------------------------------- */

/* Width-gate backward: dx = dy * gate[b % gate_B, c / (C/groups)];  dgate_partial[b, chunk, g] = sum over the rows of the
 * chunk and the channels of group g of dy * y0 (y0 = the pre-gate activation).  nchunk = aptp_groupnorm_nchunk(HW); the caller
 * sums the chunk axis (and the CFG-tiled batch rows that share a gate row). */
typedef struct {
  const void* dy; int64_t lddy;   /* bf16 [B*HW, C] */
  const void* y0; int64_t ldy0;   /* bf16 [B*HW, C] */
  void* dx; int64_t lddx;         /* bf16 [B*HW, C] */
  int32_t B, HW, C, groups;
  const float* gate; int32_t gate_B;   /* fp32 [gate_B, groups] */
  float* dgate_partial;                /* fp32 [B, nchunk, groups] */
  float* dgate;                        /* optional fp32 [gate_B, groups]: the partials folded over the chunk axis and over
                                          the batch rows that share a gate row (b = rep*gate_B + bg), fixed order */
} AptpGateBwdParams;
int aptp_gate_bwd(const AptpGateBwdParams* p, aptp_stream_t stream);

/* GEGLU in its training form (value | gate halves side by side, not interleaved): forward out = (h*m) * gelu_erf(g*m);
 * backward (backward = 1): dhg = [dh | dg], dgate_partial[b, chunk, grp] (blocks.py:41-50 through autograd). */
typedef struct {
  const void* hg; int64_t ldhg;        /* bf16 [B*HW, 2C] : h = cols [0,C), g = cols [C,2C) */
  void* out; int64_t ldout;            /* bf16 [B*HW, C]  (forward) */
  const void* dout; int64_t lddout;    /* bf16 [B*HW, C]  (backward) */
  void* dhg; int64_t lddhg;            /* bf16 [B*HW, 2C] (backward) */
  int32_t B, HW, C, groups;
  const float* gate; int32_t gate_B;   /* fp32 [gate_B, groups] or NULL (mask == 1) */
  float* dgate_partial;                /* fp32 [B, nchunk, groups] (backward) */
  int32_t backward;
  float* dgate;                        /* optional fp32 [gate_B, groups] (backward): folded partials, as in AptpGateBwdParams */
} AptpGegluParams;
int aptp_geglu(const AptpGegluParams* p, aptp_stream_t stream);

/* DepthGate in its training form (pdm/models/unet/gates.py:36-42): forward y = (1 - d[b % d_B]) * x_in + d[b % d_B] * x_out;
 * backward (backward = 1): d_out = dy * d, d_in = dy * (1 - d), dd[bg] = sum over the samples that share gate row bg and over
 * all elements of dy * (x_out - x_in) (two-stage, deterministic; dd_partial: fp32 [B, aptp_groupnorm_nchunk(HW)] scratch).
 * In the reference this is five elementwise passes forward and autograd's chain backward; here one launch each way. */
typedef struct {
  const void* x_in; int64_t ld_in;     /* bf16 [B*HW, C] */
  const void* x_out; int64_t ld_out;   /* bf16 [B*HW, C] */
  void* y; int64_t ld_y;               /* bf16 [B*HW, C]  (forward) */
  const void* dy; int64_t ld_dy;       /* bf16 [B*HW, C]  (backward) */
  void* d_in; int64_t ld_d_in;         /* bf16 [B*HW, C]  (backward) */
  void* d_out; int64_t ld_d_out;       /* bf16 [B*HW, C]  (backward) */
  int32_t B, HW, C;
  const float* d; int32_t d_B;         /* fp32 [d_B] */
  float* dd_partial;                   /* fp32 [B, nchunk] (backward) */
  float* dd;                           /* fp32 [d_B] (backward) */
  int32_t backward;
} AptpDepthLerpParams;
int aptp_depth_lerp(const AptpDepthLerpParams* p, aptp_stream_t stream);

/* Weight gradient of a stride-1 "same" convolution (3x3, pad 1) or of a 1x1 convolution / linear layer, for the expert
 * fine-tune step (pdm/training/trainer.py:1616, loss.backward through F.conv2d / F.linear of the pruned expert):
 *   dw[s][n][tap][c] = sum over the pixels m of slice s of dy[m][n] * x[pix(m, tap)][c]      (fp32, packed-weight order)
 * The caller sums the split_m slabs (fixed order: deterministic).  Operands stay in their forward layout: no im2col and no
 * transposed copies; the kernel reads its LDS tiles K-major with gfx950's transposed LDS read.
 * aptp_conv_wgrad_supported: 1 when the geometry is handled (KH == KW in {1, 3}; C, N, ld multiples of 8; for 3x3: H*W a
 * multiple of 32 and W a multiple of 32 or a divisor of 32 that is >= 8), else 0 (use the GEMM on transposed copies). */
typedef struct {
  const void* x; int64_t ldx;      /* bf16 [B*H*W, C]: the convolution's input */
  const void* dy; int64_t lddy;    /* bf16 [B*H*W, N]: gradient of its output */
  float* dw;                       /* fp32 [split_m, N, KH*KW, C] */
  int32_t B, H, W, C, N, KH, KW, split_m;
  int32_t ld_dw;                   /* row length of dw's innermost (c) dimension; 0 = C.  With split_m == 1 and ld_dw = cin_pad the
                                      kernel writes straight into a packed-layout gradient (pad columns are not touched) */
  float* db;                       /* optional fp32 [split_m, N]: per-slice column sums of dy = the bias gradient's slabs (the
                                      workgroups of input-channel block 0 sum the dy tiles they stage anyway) */
  int64_t slab_stride;             /* elements between consecutive slices of dw; 0 = N*KH*KW*ld_dw */
  int64_t db_stride;               /* elements between consecutive slices of db; 0 = N (both non-zero: db rows appended to dw's slabs) */
  /* resampling 3x3 convolutions (Downsample2D / Upsample2D of the U-Net).  H, W above are always the OUTPUT map (the one dy
   * lives on).  stride 2 (0 and 1 both mean 1): x is [B, 2H, 2W, C], pad 1.  ups 1: the convolution reads the nearest-x2
   * up-sampled input: x is [B, H/2, W/2, C].  Not both. */
  int32_t stride, ups;
} AptpWgradParams;
int aptp_conv_wgrad_supported(const AptpWgradParams* p);
int aptp_conv_wgrad_suggest_split(const AptpWgradParams* p);
int aptp_conv_wgrad(const AptpWgradParams* p, aptp_stream_t stream);

/* Every stride-1 weight gradient of a backward pass as ONE launch per filter size (round 3; replaces 225 + 43 per-layer launches
 * of the expert fine-tune step, i.e. F.conv2d / F.linear weight gradients of trainer.py:1616 issued when the backward is
 * complete).  Per item the caller chooses split_m for the BATCH (a pixel slice only has to be worth a workgroup, not to fill the
 * chip: ~2,048 pixels), fills one descriptor of aptp_conv_wgrad_many_item_bytes() bytes with aptp_conv_wgrad_many_fill (host
 * memory; first_block = sum of aptp_conv_wgrad_many_blocks() of the items before it), uploads the descriptor table (16-byte
 * aligned) and an int32 map block -> item of total_blocks entries, and launches.  All items of one call share `taps` (1 or 9);
 * results, slabs and bias-gradient slabs exactly as aptp_conv_wgrad with the same split_m (bit-identical). */
int64_t aptp_conv_wgrad_many_item_bytes(void);
int aptp_conv_wgrad_many_blocks(const AptpWgradParams* p);
int aptp_conv_wgrad_many_fill(const AptpWgradParams* p, void* item_out, int32_t first_block);
int aptp_conv_wgrad_many(const void* items_dev, const int32_t* block_item_dev, int32_t n_items, int32_t total_blocks,
                         int32_t taps, aptp_stream_t stream);

/* out[row][c] = sum over r < R of partials[r][row][c] (partials: fp32 [R][n_rows][C] contiguous; out: fp32 [n_rows][ld_out]); fixed order.
 * The slab sum of a split weight gradient (written into the padded packed layout) and the chunk sums of bias / affine gradients. */
typedef struct {
  const float* partials; float* out; int32_t R, n_rows, C, ld_out;
  float* tail_out; int32_t tail_rows;   /* optional: the LAST tail_rows of the n_rows go to tail_out (fp32 [tail_rows][C], contiguous)
                                           instead of out -- a weight gradient's slabs and its bias gradient's slabs in one fold */
  int32_t pair_split;                   /* 1 (n_rows == 1, C even): the columns are (a, b) pairs -- the (dbeta, dgamma) partials of the norm
                                           backward kernels -- and out receives a_k at [k], b_k at [ld_out + k]: two contiguous
                                           parameter gradients without a strided copy each; b goes to tail_out when that is set (two separate gradient
                                           tensors), else to out + ld_out (>= C/2) */
  int32_t tail_n;                       /* 0 = every tail value; otherwise only the first tail_n of the tail_rows * C tail values are written
                                           (a bias gradient of N values whose slabs were padded to whole rows of C) */
} AptpFoldRowsParams;
int aptp_fold_rows(const AptpFoldRowsParams* p, aptp_stream_t stream);
/* many folds as ONE launch: items_dev = n_items parameter blocks in DEVICE memory (each valid for aptp_fold_rows), starts_dev =
 * n_items + 1 int32 prefix sums of aptp_fold_rows_blocks() (starts[n_items] = total_blocks).  The expert fine-tune step folds the
 * slabs of every split weight gradient this way after its backward (FineTuner.step, trainer.py:1616). */
int aptp_fold_rows_blocks(const AptpFoldRowsParams* p);
int aptp_fold_rows_many(const AptpFoldRowsParams* items_dev, const int32_t* starts_dev, int32_t n_items, int32_t total_blocks,
                        aptp_stream_t stream);

/* AdamW (torch.optim.AdamW arithmetic: decoupled weight decay, bias correction, fp32) over many tensors in ONE launch, writing
 * the bf16 operand the GEMM kernels read in the same pass (the optimizer step of FineTuner.step, pdm/training/trainer.py:1529-1540,
 * 1616-1619).  items_dev: n_items descriptors in DEVICE memory; p, g, m, v 16-byte aligned fp32 [n]; shadow optional bf16 [n]
 * (8-byte aligned).  starts_dev: int32 [n_items + 1] prefix sums of aptp_adamw_blocks(n) in device memory; total_blocks =
 * starts[n_items].  step_dev: fp32 device scalar = number of steps taken before this one (the caller increments it). */
typedef struct { float* p; const float* g; float* m; float* v; void* shadow; int64_t n; } AptpAdamWItem;
typedef struct {
  const AptpAdamWItem* items_dev; const int32_t* starts_dev; int32_t n_items, total_blocks;
  float lr, beta1, beta2, eps, weight_decay;
  const float* step_dev;
  const float* gate_dev;  /* optional fp32 device scalar (e.g. the step's total loss): the launch is a no-op unless it is finite --
                           * the batch-skip of pdm/training/trainer.py:921-929 for a replayed graph; the caller advances step_dev
                           * by the same predicate */
  float grad_scale;       /* gradients are multiplied by this before use (0 = 1): 1 / world size folds the mean of a summed
                           * data-parallel exchange (DDP inside accelerator.backward, trainer.py:1616) into the optimizer pass */
} AptpAdamWParams;
int aptp_adamw_blocks(int64_t n);
int aptp_adamw_many(const AptpAdamWParams* p, aptp_stream_t stream);

/* Mean squared error of two equally shaped activations, read where they lie (the loss terms of Pruner.step / FineTuner.step:
 * pdm/training/trainer.py:1197-1225 diffusion MSE, output distillation, block distillation; :1729-1752 for the fine-tune step).
 * backward = 0: out[0] = mean((a - b)^2); partial: fp32 [aptp_mse_nblocks(rows, C)] scratch (every slot is written).
 * backward = 1: da = (a - b) * g[0] * 2 / (rows * C)   (g: the fp32 device scalar dL/d(out); da has a's element type).
 * f32 = 0: bf16 operands; f32 = 1: fp32 operands.  C multiple of 8; leading dimensions in elements, multiples of 8. */
typedef struct {
  const void* a; int64_t lda;
  const void* b; int64_t ldb;
  int64_t rows; int32_t C; int32_t f32;
  float* partial; float* out;
  const float* g; void* da; int64_t ldda;
  int32_t backward;
} AptpMseParams;
int aptp_mse_nblocks(int64_t rows, int32_t C);
int aptp_mse(const AptpMseParams* p, aptp_stream_t stream);

/* Data-gradient operand from the forward operand of the same contraction (both in packed bf16 layout):
 * dst[c][taps-1-t][n] = src[n][t][c] for n < N, c < C, zero in the rest of dst's [dst_rows][taps][dst_ld] image.
 * (ops.pack_weight_dgrad of the same weights, without going through the diffusers layout.) */
typedef struct { const void* src; void* dst; int32_t N, C, taps, src_ld, dst_ld, dst_rows; } AptpPackDgradParams;
int aptp_pack_dgrad(const AptpPackDgradParams* p, aptp_stream_t stream);
/* The same for many weights in ONE launch (the operand refresh after an optimizer step): items_dev = n_items descriptors in
 * DEVICE memory (each validated by the caller as aptp_pack_dgrad would), starts_dev = int32 [n_items + 1] prefix sums of
 * aptp_pack_dgrad_blocks(item) (device memory), total_blocks = starts[n_items]. */
int aptp_pack_dgrad_blocks(const AptpPackDgradParams* p);
int aptp_pack_dgrad_many(const AptpPackDgradParams* items_dev, const int32_t* starts_dev, int32_t n_items, int32_t total_blocks,
                         aptp_stream_t stream);

/* GroupNorm(+SiLU) data gradient.  fwd_stats = the [B, nchunk, groups, 2] (sum, sumsq) partials the forward wrote into its
 * workspace (keep that buffer alive); workspace: fp32 [B, nchunk, groups, 2]. */
typedef struct {
  const void* x; int64_t ldx;
  const void* dy; int64_t lddy;
  void* dx; int64_t lddx;
  int32_t B, HW, C, groups;
  const float* gamma; const float* beta;
  float eps; int32_t silu;
  const float* fwd_stats;
  void* workspace;
  float* pgrad_partial;   /* optional fp32 [B, nchunk, C, 2]: per-channel (sum dz, sum dz*xhat) -> dbeta, dgamma (fine-tuning) */
  const void* add; int64_t ldadd;   /* optional bf16 [B*HW, C]: dx = (norm gradient) + add in fp32, rounded once -- the gradient that
                                     * arrives over the residual path when x forks into norm(x) and `+ x` (round 3) */
} AptpGroupNormBwdParams;
int aptp_groupnorm_bwd(const AptpGroupNormBwdParams* p, aptp_stream_t stream);

/* LayerNorm data gradient (statistics recomputed from x in registers). */
typedef struct {
  const void* x; int64_t ldx;
  const void* dy; int64_t lddy;
  void* dx; int64_t lddx;
  int32_t rows, C;
  const float* gamma;
  float eps;
  const void* add; int64_t ldadd;   /* optional bf16 [rows, C]: dx += add (as AptpGroupNormBwdParams.add) */
} AptpLayerNormBwdParams;
int aptp_layernorm_bwd(const AptpLayerNormBwdParams* p, aptp_stream_t stream);

/* Column sums of `batch` (0 = 1) consecutive bf16 [rows, C] matrices (bias gradients; batch > 1: per-sample sums for the
 * gradient of a per-sample output bias): partial fp32 [nchunk, batch, C], nchunk = aptp_groupnorm_nchunk(rows) for batch > 1,
 * aptp_rows_nchunk(rows) otherwise. */
typedef struct { const void* x; int64_t ldx; int32_t rows, C; float* partial; int32_t batch; } AptpColsumParams;
int aptp_colsum(const AptpColsumParams* p, aptp_stream_t stream);

/* LayerNorm affine-parameter gradient partials: fp32 [aptp_rows_nchunk(rows), C, 2] = (sum dy, sum dy*xhat). */
typedef struct {
  const void* x; int64_t ldx;
  const void* dy; int64_t lddy;
  int32_t rows, C;
  float eps;
  float* partial;
} AptpLayerNormPgradParams;
int aptp_layernorm_pgrad(const AptpLayerNormPgradParams* p, aptp_stream_t stream);

/* Attention backward: dq, dk, dv from q, k, v, o, dout and the forward's lse; delta is fp32 scratch [B, heads, Lq].
 * Same strided [B, L, heads, 64] layout convention as aptp_attention. */
typedef struct {
  const void* q; int64_t q_stride_b, q_stride_l;
  const void* k; int64_t k_stride_b, k_stride_l;
  const void* v; int64_t v_stride_b, v_stride_l;
  const void* o; int64_t o_stride_b, o_stride_l;
  const void* dout; int64_t dout_stride_b, dout_stride_l;
  void* dq; int64_t dq_stride_b, dq_stride_l;
  void* dk; int64_t dk_stride_b, dk_stride_l;
  void* dv; int64_t dv_stride_b, dv_stride_l;
  const float* lse; float* delta;
  int32_t B, heads, Lq, Lk;
  float scale;
  int32_t q_split;   /* dK/dV: split the query range over this many workgroups (0 / 1 = off; aptp_attention_bwd_q_split suggests
                        it: few keys -- cross-attention's 77 -- leave one key tile per (b, head)), fp32 partials folded in slice order */
  void* workspace;   /* aptp_attention_bwd_workspace_bytes(p, q_split) bytes, 16-byte aligned, when q_split > 1 */
} AptpAttentionBwdParams;
int aptp_attention_bwd_q_split(const AptpAttentionBwdParams* p);
size_t aptp_attention_bwd_workspace_bytes(const AptpAttentionBwdParams* p, int32_t q_split);
int aptp_attention_bwd(const AptpAttentionBwdParams* p, aptp_stream_t stream);

const char* aptp_last_error(void);
int aptp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* APTP_HIP_H */
