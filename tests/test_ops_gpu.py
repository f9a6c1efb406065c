"""GPU parity tests of the individual HIP kernels (through the C ABI) against plain PyTorch fp32 references.

Tolerances (bf16 operands, fp32 accumulate): relative L2 error <= 4e-3 per op against an fp32
reference evaluated on the SAME bf16-rounded inputs (so only accumulation order and the bf16 output rounding
differ), max-abs error bounded by a few bf16 ulps of the output scale.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

REL_L2_TOL = 4e-3


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def _rand(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale)


@pytest.fixture(scope="module")
def ops(cuda):
    from diffusion_pruning_amd import ops as o
    o._lib.load()
    return o


@pytest.fixture(params=[True, False], ids=["splitk-in-kernel", "splitk-reduce-launch"])
def both_splitk(ops, request):
    """run a test with the in-kernel split-K reduction and with the separate reduce launch"""
    ops.SPLITK_IN_KERNEL = request.param
    yield request.param
    ops.SPLITK_IN_KERNEL = True


@pytest.fixture(params=[True, False], ids=["gn-fused-finalize", "gn-finalize-launch"])
def both_gn_forms(ops, request):
    """GroupNorm with the last-arriver finalise inside the statistics kernel and with the separate finalise launch"""
    ops.GN_FUSED_FINALIZE = request.param
    yield request.param
    ops.GN_FUSED_FINALIZE = False


@pytest.fixture(params=[0, 1, 2, 3, 4, 6], ids=["attn-auto", "attn-staggered", "attn-4groups", "attn-2groups", "attn-dbuf", "attn-swpipe"])
def both_attn_forms(ops, request):
    """key-split attention in its forms: auto (= the software-pipelined kernel where whole 128-blocks allow it), two groups one phase
    apart, four groups, two lock-step groups, double-buffered, software-pipelined"""
    ops.ATTN_VARIANT = request.param
    yield request.param
    ops.ATTN_VARIANT = 0


@pytest.fixture(params=[0, 1], ids=["epi-auto", "epi-acc-layout"])
def both_epilogues(ops, request):
    """run a test with the coalesced (LDS-transposed) epilogue and with the accumulator-layout one"""
    ops.EPILOGUE = request.param
    yield request.param
    ops.EPILOGUE = 0


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, ups, tile, split_k
    (2, 16, 16, 64, 64, 3, 1, 0, 0, 1),
    (1, 8, 8, 320, 320, 3, 1, 0, 0, None),
    (2, 32, 32, 160, 320, 3, 1, 0, 2, 1),      # Cin not multiple of 64 (zero-padded K chunk), 128x160 tile
    (2, 32, 32, 160, 320, 3, 1, 0, 4, 1),      # 64x160 tile
    (2, 16, 16, 128, 128, 3, 2, 0, 0, 1),      # stride-2 downsample
    (2, 8, 8, 128, 128, 3, 1, 1, 0, 1),        # nearest-x2 upsample folded in
    (2, 16, 16, 192, 128, 1, 1, 0, 0, 1),      # 1x1 shortcut
    (3, 7, 5, 72, 40, 3, 1, 0, 0, 1),          # ragged M / N / Cin tails
    (1, 8, 8, 1280, 1280, 3, 1, 0, 3, 4),      # split-K
    (1, 8, 8, 2560, 1280, 3, 1, 0, 0, None),   # auto split-K, K = 23040
    (4, 8, 8, 64, 64, 3, 1, 0, 1, 1),          # 128x128 tile
    (4, 8, 8, 64, 64, 3, 1, 0, 5, 1),          # 128x64 tile
    (4, 8, 8, 64, 64, 3, 1, 0, 6, 2),          # 64x64 tile + split-K
    # LDS-DMA variants (tiles 7..12): zero padding / ragged tails come from the zero page, swizzle on the source chunk
    (2, 16, 16, 64, 64, 3, 1, 0, 7, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 8, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 10, 1),
    (2, 16, 16, 128, 128, 3, 2, 0, 9, 1),
    (2, 8, 8, 128, 128, 3, 1, 1, 9, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 12, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 11, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 9, 4),
    (2, 16, 16, 192, 128, 1, 1, 0, 7, 1),
    (4, 64, 64, 320, 320, 3, 1, 0, 10, 1),     # full level-64 shape
    # 3x3 halo-in-LDS kernel (tiles 43, 44): image widths 64 / 32 / 16, ragged N and Cin, split-K slices that start mid channel step
    (2, 64, 64, 64, 160, 3, 1, 0, 43, 1),
    (4, 64, 64, 320, 320, 3, 1, 0, 43, 1),
    (4, 64, 64, 320, 160, 3, 1, 0, 43, 2),
    (2, 32, 32, 160, 320, 3, 1, 0, 43, 1),
    (2, 32, 32, 320, 136, 3, 1, 0, 44, 3),
    (4, 16, 16, 72, 200, 3, 1, 0, 43, 1),
    (1, 64, 64, 640, 128, 3, 1, 0, 44, 4),
    (3, 32, 32, 1280, 320, 3, 1, 0, 43, 7),
    # 6- / 8-stage rings (tiles 45..48): K shorter than, equal to and longer than the ring; split-K slices shorter than the ring
    (2, 16, 16, 128, 128, 1, 1, 0, 45, 1),
    (2, 16, 16, 384, 200, 1, 1, 0, 46, 1),
    (1, 16, 16, 1280, 1280, 1, 1, 0, 46, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 46, 5),
    (3, 7, 5, 72, 40, 3, 1, 0, 45, 1),
    (2, 16, 16, 640, 320, 1, 1, 0, 47, 1),
    (1, 8, 8, 1280, 640, 3, 1, 0, 47, 4),
    (2, 32, 32, 320, 72, 3, 1, 0, 48, 1),
    (2, 16, 16, 512, 128, 1, 1, 0, 48, 2),
    # two K-tiles per barrier (tiles 49..57): odd and even tile counts, K shorter than the ring, split-K slices of 1..3 tiles, tails
    (2, 16, 16, 64, 64, 1, 1, 0, 49, 1),       # one K-tile
    (2, 16, 16, 128, 128, 1, 1, 0, 49, 1),     # two
    (2, 16, 16, 192, 128, 1, 1, 0, 50, 1),     # three (odd) on the 6-stage ring
    (2, 16, 16, 320, 200, 1, 1, 0, 50, 1),     # five = D + 1
    (1, 16, 16, 1280, 1280, 1, 1, 0, 50, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 49, 5),
    (3, 7, 5, 72, 40, 3, 1, 0, 49, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 52, 4),
    (2, 16, 16, 640, 320, 1, 1, 0, 51, 1),
    (1, 8, 8, 1280, 640, 3, 1, 0, 52, 4),
    (2, 32, 32, 320, 72, 3, 1, 0, 53, 1),
    (2, 16, 16, 128, 128, 3, 2, 0, 54, 1),
    (2, 8, 8, 128, 128, 3, 1, 1, 54, 3),
    (2, 32, 32, 160, 320, 3, 1, 0, 55, 1),
    (4, 64, 64, 320, 320, 3, 1, 0, 56, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 56, 2),
    (4, 32, 32, 1280, 640, 3, 1, 0, 57, 3),
    (3, 7, 5, 72, 40, 3, 1, 0, 57, 1),
    # intra-workgroup K split (tiles 58..63): copy 1 hands its accumulators over through LDS and retires before the epilogue
    (2, 16, 16, 64, 64, 1, 1, 0, 58, 1),
    (2, 16, 16, 192, 128, 1, 1, 0, 59, 1),
    (1, 16, 16, 1280, 1280, 1, 1, 0, 58, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 59, 5),
    (3, 7, 5, 72, 40, 3, 1, 0, 58, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 61, 4),
    (2, 16, 16, 640, 320, 1, 1, 0, 60, 1),
    (2, 16, 16, 128, 128, 3, 2, 0, 61, 1),
    (2, 32, 32, 320, 72, 3, 1, 0, 62, 1),
    (2, 8, 8, 128, 128, 3, 1, 1, 63, 3),
    (4, 32, 32, 320, 640, 3, 1, 0, 63, 1),
    # 3-stage LDS-DMA ring (tiles 13..18): counted vmcnt + raw barrier; short and long K, split-K slices of 1-2 steps
    (2, 16, 16, 64, 64, 3, 1, 0, 13, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 14, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 16, 1),
    (2, 16, 16, 128, 128, 3, 2, 0, 15, 1),
    (2, 8, 8, 128, 128, 3, 1, 1, 15, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 18, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 17, 4),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 15, 4),
    (2, 16, 16, 64, 128, 1, 1, 0, 13, 1),      # a single K-step
    (2, 16, 16, 128, 128, 1, 1, 0, 18, 1),     # two K-steps
    (4, 64, 64, 320, 320, 3, 1, 0, 16, 1),
    (4, 64, 64, 320, 320, 3, 1, 0, 14, 1),
    (4, 32, 32, 1280, 640, 3, 1, 0, 14, 3),
    # 8-wave LDS-DMA workgroups (tiles 19..22)
    (2, 32, 32, 160, 320, 3, 1, 0, 19, 1),
    (4, 64, 64, 160, 320, 3, 1, 0, 19, 1),
    (4, 64, 64, 320, 320, 3, 1, 0, 20, 1),
    (2, 16, 16, 128, 128, 3, 2, 0, 21, 1),
    (2, 8, 8, 128, 128, 3, 1, 1, 22, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 19, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 21, 2),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 22, 4),
    (2, 16, 16, 192, 128, 1, 1, 0, 20, 1),
    # 4-stage rings (tiles 23..27), 8-wave 3/4-stage rings (28..32) and the 8-wave ping-pong schedule (33..40): several
    # tiles in flight across the barriers, waves with different DMA counts (160-wide tile on 512 threads), K shorter than
    # the ring, split-K slices of 1-3 steps, staggered wave groups
    (2, 32, 32, 160, 320, 3, 1, 0, 23, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 23, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 23, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 23, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 23, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 23, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 23, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 24, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 24, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 24, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 24, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 24, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 24, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 24, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 25, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 25, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 25, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 25, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 25, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 25, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 25, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 26, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 26, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 26, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 26, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 26, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 26, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 26, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 27, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 27, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 27, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 27, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 27, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 27, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 27, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 28, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 28, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 28, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 28, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 28, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 28, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 28, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 29, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 29, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 29, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 29, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 29, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 29, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 29, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 30, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 30, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 30, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 30, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 30, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 30, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 30, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 31, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 31, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 31, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 31, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 31, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 31, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 31, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 32, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 32, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 32, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 32, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 32, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 32, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 32, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 33, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 33, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 33, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 33, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 33, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 33, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 33, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 34, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 34, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 34, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 34, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 34, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 34, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 34, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 35, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 35, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 35, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 35, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 35, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 35, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 35, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 36, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 36, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 36, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 36, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 36, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 36, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 36, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 37, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 37, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 37, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 37, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 37, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 37, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 37, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 38, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 38, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 38, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 38, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 38, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 38, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 38, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 39, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 39, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 39, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 39, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 39, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 39, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 39, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 40, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 40, 1),
    (2, 16, 16, 64, 128, 1, 1, 0, 40, 1),
    (2, 16, 16, 128, 128, 1, 1, 0, 40, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 40, 4),
    (2, 8, 8, 128, 128, 3, 1, 1, 40, 3),
    (2, 16, 16, 128, 128, 3, 2, 0, 40, 1),
    (4, 64, 64, 320, 320, 3, 1, 0, 23, 1),
    (4, 64, 64, 320, 320, 3, 1, 0, 29, 1),
    (4, 64, 64, 320, 320, 3, 1, 0, 28, 1),
    (4, 32, 32, 1280, 640, 3, 1, 0, 31, 3),
    (4, 32, 32, 1280, 640, 3, 1, 0, 32, 2),
    (4, 64, 64, 320, 320, 3, 1, 0, 33, 1),
    (4, 64, 64, 320, 320, 3, 1, 0, 34, 1),
    (4, 64, 64, 320, 160, 3, 1, 0, 37, 1),
    (4, 64, 64, 320, 160, 3, 1, 0, 39, 1),
    (4, 32, 32, 1280, 640, 3, 1, 0, 36, 3),
    (4, 32, 32, 640, 640, 3, 1, 0, 38, 1),
    (4, 32, 32, 640, 640, 3, 1, 0, 40, 1),
    # persistent stream-K macro-tiles (64: 256x160, 65: 256x128, 66: 128x256; split_k 1 = whole tiles, 2 = K split on):
    # fewer units than CUs, ragged tails, 1 / 2 / 3 K-steps (shorter than the DMA stream's head start), stride 2, folded
    # upsample, tiles shared by 30+ workgroups (K = 180 / 360 steps on 8 / 5 tiles), full-size level-64 / 32 / 16 shapes
    (2, 16, 16, 64, 64, 3, 1, 0, 64, 1),
    (2, 16, 16, 64, 64, 3, 1, 0, 64, 2),
    (3, 7, 5, 72, 40, 3, 1, 0, 64, 2),
    (3, 7, 5, 72, 40, 3, 1, 0, 65, 1),
    (3, 7, 5, 72, 40, 3, 1, 0, 66, 2),
    (2, 16, 16, 64, 128, 1, 1, 0, 64, 2),
    (2, 16, 16, 128, 128, 1, 1, 0, 65, 2),
    (2, 16, 16, 192, 128, 1, 1, 0, 66, 2),
    (2, 16, 16, 192, 200, 1, 1, 0, 64, 1),
    (2, 16, 16, 128, 128, 3, 2, 0, 64, 2),
    (2, 8, 8, 128, 128, 3, 1, 1, 65, 2),
    (2, 8, 8, 128, 128, 3, 1, 1, 66, 1),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 64, 2),
    (1, 8, 8, 2560, 1280, 3, 1, 0, 66, 2),
    (4, 8, 8, 1280, 640, 3, 1, 0, 65, 2),
    (2, 32, 32, 160, 320, 3, 1, 0, 64, 1),
    (2, 32, 32, 160, 320, 3, 1, 0, 64, 2),
    (4, 64, 64, 320, 320, 3, 1, 0, 64, 2),
    (4, 64, 64, 320, 160, 3, 1, 0, 64, 2),
    (4, 64, 64, 160, 320, 3, 1, 0, 65, 1),
    (4, 32, 32, 1280, 640, 3, 1, 0, 64, 2),
    (4, 32, 32, 640, 1280, 3, 1, 0, 66, 2),
    (4, 16, 16, 1280, 1280, 1, 1, 0, 66, 2),
    (4, 16, 16, 2560, 1280, 3, 1, 0, 64, 2),
    (1, 32, 32, 4160, 200, 1, 1, 0, 64, 2),      # Cin beyond the 8 KiB zero page of the other LDS-DMA tiles
    # the same macro-tiles with the fragment reads in the load slot (67..69)
    (3, 7, 5, 72, 40, 3, 1, 0, 67, 2),
    (2, 16, 16, 64, 128, 1, 1, 0, 68, 2),
    (2, 16, 16, 128, 128, 1, 1, 0, 69, 1),
    (2, 16, 16, 192, 128, 1, 1, 0, 67, 2),
    (2, 16, 16, 128, 128, 3, 2, 0, 68, 2),
    (2, 8, 8, 128, 128, 3, 1, 1, 69, 2),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 67, 2),
    (4, 64, 64, 320, 320, 3, 1, 0, 67, 2),
    (4, 32, 32, 1280, 640, 3, 1, 0, 68, 2),
    (4, 16, 16, 2560, 1280, 3, 1, 0, 69, 2),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_plain(ops, cuda, case, both_splitk):
    B, H, W, Cin, Cout, k, stride, ups, tile, split_k = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = _rand((B, Cin, H, W), g).bfloat16()
    w = _rand((Cout, Cin, k, k), g, 1.0 / math.sqrt(Cin * k * k)).bfloat16()
    b = _rand((Cout,), g, 0.1)
    pw = ops.pack_weight(w.float(), b, device=cuda)
    y = ops.conv_gemm(nhwc(x).to(cuda), pw, stride=stride, ups=ups, tile=tile, split_k=split_k)
    xr = x.float()
    if ups:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xr, w.float(), b, stride=stride, padding=k // 2)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    e = rel_l2(got, ref)
    assert e <= REL_L2_TOL, f"rel-L2 {e:.3e}"


@pytest.mark.parametrize("B,H,Cin,Cout,k,tile,split_k", [(4, 16, 2560, 1280, 1, 30, 3), (4, 16, 640, 1280, 3, 34, 4),
                                                         (4, 8, 1280, 640, 3, 40, 12), (4, 32, 320, 640, 3, 34, 2),
                                                         (4, 32, 1920, 320, 3, 19, 8), (1, 24, 64, 200, 3, 12, 3),
                                                         (4, 16, 1280, 1280, 1, 58, 3), (4, 8, 640, 1280, 3, 63, 6)])
def test_splitk_in_kernel_matches_reduce_launch_bitwise_and_is_stable(ops, cuda, B, H, Cin, Cout, k, tile, split_k):
    """The in-kernel reduction re-reads every slab in slice order, so it must equal the reduce-launch form BIT FOR BIT
    whichever workgroup arrives last, on every one of many back-to-back launches (stale slab lines in a CU's L1 / another
    XCD's L2, a lost counter reset or a torn ticket would show up as a differing or missing tile), with other split-K
    launches of different tile counts interleaved on the same counters."""
    g = torch.Generator().manual_seed(B * H + Cin + Cout)
    x = nhwc(_rand((B, Cin, H, H), g)).bfloat16().to(cuda)
    res = nhwc(_rand((B, Cout, H, H), g)).bfloat16().to(cuda)
    pw = ops.pack_weight(_rand((Cout, Cin, k, k), g, 1.0 / math.sqrt(Cin * k * k)), _rand((Cout,), g, 0.1), device=cuda)
    x2 = nhwc(_rand((2, 256, 8, 8), g)).bfloat16().to(cuda)
    pw2 = ops.pack_weight(_rand((136, 256, 3, 3), g, 0.02), None, device=cuda)
    ops.SPLITK_IN_KERNEL = False
    try:
        ref = ops.conv_gemm(x, pw, residual=res, tile=tile, split_k=split_k).clone()
        ref2 = ops.conv_gemm(x2, pw2, tile=18, split_k=5).clone()
    finally:
        ops.SPLITK_IN_KERNEL = True
    out = torch.empty_like(ref)
    bad = 0
    for it in range(60):
        ops.conv_gemm(x, pw, residual=res, tile=tile, split_k=split_k, out=out)
        y2 = ops.conv_gemm(x2, pw2, tile=18, split_k=5)
        if it % 6 == 5:
            bad += int(not torch.equal(out, ref)) + int(not torch.equal(y2, ref2))
    torch.cuda.synchronize()
    assert bad == 0
    assert int(ops._tile_counters(cuda).abs().sum()) == 0          # every launch left its counters at zero


@pytest.mark.parametrize("B,H,Cin,Cout,k,tile", [(4, 16, 2560, 1280, 1, 64), (4, 8, 1280, 640, 3, 65), (4, 32, 320, 640, 3, 64),
                                                 (1, 24, 64, 200, 3, 66), (4, 16, 640, 1280, 3, 66),
                                                 (4, 32, 320, 160, 3, 70), (4, 16, 640, 1280, 3, 71), (1, 24, 64, 200, 3, 72), (4, 32, 640, 320, 3, 73)])
def test_streamk_is_stable_and_close_to_whole_tiles(ops, cuda, B, H, Cin, Cout, k, tile):
    """The persistent stream-K tiles: a partial tile's ranges are summed in range order by whichever workgroup arrives last,
    so many back-to-back launches -- interleaved with stream-K launches of another geometry on the same counters -- must be
    BIT-identical (a stale slab line, a lost counter reset or a wrong contributor count would show as a differing or missing
    tile); against the whole-tile form only the fp32 summation order differs."""
    g = torch.Generator().manual_seed(B * H + Cin + Cout + tile)
    x = nhwc(_rand((B, Cin, H, H), g)).bfloat16().to(cuda)
    res = nhwc(_rand((B, Cout, H, H), g)).bfloat16().to(cuda)
    pw = ops.pack_weight(_rand((Cout, Cin, k, k), g, 1.0 / math.sqrt(Cin * k * k)), _rand((Cout,), g, 0.1), device=cuda)
    x2 = nhwc(_rand((2, 256, 8, 8), g)).bfloat16().to(cuda)
    pw2 = ops.pack_weight(_rand((136, 256, 3, 3), g, 0.02), None, device=cuda)
    whole = ops.conv_gemm(x, pw, residual=res, tile=tile, split_k=1)
    ref = ops.conv_gemm(x, pw, residual=res, tile=tile, split_k=2).clone()
    ref2 = ops.conv_gemm(x2, pw2, tile=64, split_k=2).clone()
    assert rel_l2(ref.float(), whole.float()) <= 2e-3
    out = torch.empty_like(ref)
    bad = 0
    for it in range(60):
        ops.conv_gemm(x, pw, residual=res, tile=tile, split_k=2, out=out)
        y2 = ops.conv_gemm(x2, pw2, tile=64, split_k=2)
        if it % 6 == 5:
            bad += int(not torch.equal(out, ref)) + int(not torch.equal(y2, ref2))
    torch.cuda.synchronize()
    assert bad == 0
    assert int(ops._tile_counters(cuda).abs().sum()) == 0          # every launch left its counters at zero


ORDER_CASES = [
    # B, H, W, Cin, Cout, k, tile, split_k : workgroup counts that are / are not multiples of 8, with and without split-K
    (2, 32, 32, 160, 320, 3, 10, 1),     # 32 x 2 tiles
    (3, 7, 5, 72, 40, 3, 12, 1),         # 2 x 1 tiles (< 8 workgroups)
    (1, 24, 24, 64, 200, 3, 12, 3),      # 9 x 4 tiles x 3 slices = 108
    (1, 8, 8, 1280, 1280, 3, 18, 5),     # 1 x 20 x 5
    (4, 16, 16, 320, 136, 3, 34, 2),     # ping-pong tile, 8 x 1 x 2
    (2, 40, 40, 64, 64, 1, 6, 1),        # 50 x 1 (register-staged kernel)
    (4, 16, 16, 320, 136, 3, 64, 1),     # persistent macro-tile, whole tiles (with the K split the summation order follows the tile order)
    (2, 32, 32, 128, 520, 1, 66, 1),
]


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("case", ORDER_CASES)
def test_conv_workgroup_orders_agree(ops, cuda, case, order):
    """every XCD-aware workgroup -> tile order computes the same tiles: bit-identical to the legacy order"""
    B, H, W, Cin, Cout, k, tile, split_k = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = nhwc(_rand((B, Cin, H, W), g).bfloat16()).to(cuda)
    pw = ops.pack_weight(_rand((Cout, Cin, k, k), g, 1.0 / math.sqrt(Cin * k * k)), _rand((Cout,), g, 0.1), device=cuda)
    ref = ops.conv_gemm(x, pw, tile=tile, split_k=split_k, order=1)
    y = ops.conv_gemm(x, pw, tile=tile, split_k=split_k, order=order)
    assert torch.equal(y, ref)
    auto = ops.conv_gemm(x, pw, tile=tile, split_k=split_k)
    assert torch.equal(auto, ref)


@pytest.mark.parametrize("nbytes", [4, 64, 200, 65536 + 12, 3 << 20])
@pytest.mark.parametrize("tile", [0, 9, 19, 34])
def test_conv_next_weight_prefetch_changes_nothing(ops, cuda, nbytes, tile):
    """AptpConvGemmParams.prefetch: the workgroups only touch [prefetch, prefetch + bytes) (sizes below one line, ragged, more lines
    than threads); the result is bitwise the one without it and the touched buffer is unchanged"""
    g = torch.Generator().manual_seed(11)
    x = nhwc(_rand((2, 128, 16, 16), g).bfloat16()).to(cuda)
    pw = ops.pack_weight(_rand((192, 128, 3, 3), g, 0.03), _rand((192,), g, 0.1), device=cuda)
    nxt = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, generator=g).to(cuda)
    keep = nxt.clone()
    y0 = ops.conv_gemm(x, pw, tile=tile)
    y1 = ops.conv_gemm(x, pw, tile=tile, prefetch=nxt)
    assert torch.equal(y0, y1) and torch.equal(nxt, keep)


@pytest.mark.parametrize("tile,split_k", [(0, 1), (9, 1), (18, 1), (19, 1), (34, 1), (34, 2), (38, 1), (42, 1), (50, 1), (56, 1),
                                          (58, 1), (60, 2), (63, 1), (64, 1), (64, 2), (65, 2), (66, 1)])
@pytest.mark.parametrize("Cout", [320, 200])
def test_groupnorm_takes_the_producer_column_statistics(ops, cuda, tile, split_k, Cout):
    """conv_gemm(colstats=True) leaves per-(row block, channel) sum / sum of squares of what it stored with the tensor
    (AptpConvGemmParams.colstat_out); GroupNorm finds them and skips its statistics pass.  Same result as the
    statistics pass over the stored bf16 tensor, to bf16 rounding of the statistics' inputs."""
    g = torch.Generator().manual_seed(5 + tile)
    B, H, W, Cin = 2, 32, 32, 136
    x = nhwc(_rand((B, Cin, H, W), g).bfloat16()).to(cuda)
    pw = ops.pack_weight(_rand((Cout, Cin, 3, 3), g, 0.05), _rand((Cout,), g, 0.3), device=cuda)
    res = nhwc(_rand((B, Cout, H, W), g).bfloat16()).to(cuda)
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).to(cuda), _rand((Cout,), g, 0.2).to(cuda)
    y = ops.conv_gemm(x, pw, residual=res, colstats=True, tile=tile, split_k=split_k)
    segs = ops._colstats_get(y, Cout)
    assert segs is not None and len(segs) == 1
    st, rpb, cseg = segs[0][:3]
    assert cseg == Cout and (H * W) % rpb == 0
    nblk = B * H * W // rpb
    want = y.float().reshape(nblk, rpb, Cout)
    assert torch.allclose(st[:nblk, :, 0], want.sum(1), rtol=2e-2, atol=0.3)
    assert torch.allclose(st[:nblk, :, 1], (want * want).sum(1), rtol=2e-2, atol=0.3)
    with_stats = ops.groupnorm(y, gamma, beta, 8 if Cout == 200 else 32, 1e-5, True)
    ops.COLSTATS = False
    try:
        plain = ops.groupnorm(y, gamma, beta, 8 if Cout == 200 else 32, 1e-5, True)
    finally:
        ops.COLSTATS = True
    assert rel_l2(with_stats.float().cpu(), plain.float().cpu()) <= 3e-3
    # overwriting the tensor without statistics forgets them
    ops.conv_gemm(x, pw, out=y, tile=tile, split_k=split_k)
    assert ops._colstats_get(y, Cout) is None


@pytest.mark.parametrize("tile,split_k", [(0, 1), (9, 1), (19, 1), (34, 3), (25, 1), (50, 2), (56, 1), (59, 1), (63, 2), (64, 2), (65, 1), (66, 2)])
@pytest.mark.parametrize("Cin2", [64, 200, 960])
def test_conv_second_operand_segment(ops, cuda, tile, split_k, Cin2, both_splitk):
    """x2: conv3x3(x) + conv1x1(x2) in one launch (the resnet's conv_shortcut as extra K-steps of conv2)"""
    g = torch.Generator().manual_seed(17 + Cin2)
    B, H, W, Cin, Cout = 2, 16, 16, 72, 136
    x, x2 = _rand((B, Cin, H, W), g).bfloat16(), _rand((B, Cin2, H, W), g).bfloat16()
    w = _rand((Cout, Cin, 3, 3), g, 1.0 / math.sqrt(9 * Cin)).bfloat16()
    w2 = _rand((Cout, Cin2, 1, 1), g, 1.0 / math.sqrt(Cin2)).bfloat16()
    b, b2 = _rand((Cout,), g, 0.1), _rand((Cout,), g, 0.1)
    pw = ops.pack_weight_cat(ops.pack_weight(w.float(), b, device=cuda), w2.float(), b2)
    y = ops.conv_gemm(nhwc(x).to(cuda), pw, x2=nhwc(x2).to(cuda), tile=tile, split_k=split_k)
    ref = F.conv2d(x.float(), w.float(), b, padding=1) + F.conv2d(x2.float(), w2.float(), b2)
    assert rel_l2(y.float().cpu().permute(0, 3, 1, 2), ref) <= REL_L2_TOL


def test_conv_strided_views(ops, cuda, both_epilogues):
    """input is a channel slice of a wider buffer, output written into a slice of a wider buffer"""
    g = torch.Generator().manual_seed(7)
    B, H, W, Cin, Cout = 2, 8, 8, 64, 96
    x = _rand((B, Cin, H, W), g).bfloat16()
    w = _rand((Cout, Cin, 3, 3), g, 0.05).bfloat16()
    pw = ops.pack_weight(w.float(), None, device=cuda)
    wide_in = torch.full((B, H, W, 160), float("nan"), dtype=torch.bfloat16, device=cuda)
    wide_in[..., 32:96] = nhwc(x).to(cuda)
    wide_out = torch.zeros(B, H, W, 256, dtype=torch.bfloat16, device=cuda)
    ops.conv_gemm(wide_in[..., 32:96], pw, out=wide_out[..., 64:160])
    ref = F.conv2d(x.float(), w.float(), None, padding=1)
    got = wide_out[..., 64:160].float().cpu().permute(0, 3, 1, 2)
    assert rel_l2(got, ref) <= REL_L2_TOL
    assert float(wide_out[..., :64].abs().max()) == 0.0 and float(wide_out[..., 160:].abs().max()) == 0.0


@pytest.mark.parametrize("split_k,tile", [(1, 0), (3, 0), (2, 64), (1, 65), (2, 66)])
def test_conv_full_epilogue(ops, cuda, split_k, tile, both_epilogues, both_splitk):
    """bias + temb rowbias + per-sample width gate, then (separately) corr + residual + depth lerp"""
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout, G = 4, 8, 8, 64, 64, 32
    x = _rand((B, Cin, H, W), g).bfloat16()
    w = _rand((Cout, Cin, 3, 3), g, 0.05).bfloat16()
    b = _rand((Cout,), g, 0.1)
    pw = ops.pack_weight(w.float(), b, device=cuda)
    temb = _rand((B, Cout), g, 0.5)
    gate = torch.rand((2, G), generator=g)                     # Bg=2 tiled over B=4 (CFG layout)
    y = ops.conv_gemm(nhwc(x).to(cuda), pw, rowbias=temb.to(cuda), colgate=gate.to(cuda).contiguous(),
                      gate_group=Cout // G, split_k=split_k, tile=tile)
    ref = F.conv2d(x.float(), w.float(), b, padding=1) + temb[:, :, None, None]
    mask = gate.repeat_interleave(Cout // G, dim=1).repeat(2, 1)[:, :, None, None]
    ref = ref * mask
    assert rel_l2(y.float().cpu().permute(0, 3, 1, 2), ref) <= REL_L2_TOL

    res = _rand((B, Cout, H, W), g).bfloat16()
    din = _rand((B, Cout, H, W), g).bfloat16()
    d = torch.rand((2,), generator=g)
    corr = _rand((1, 9, Cout), g, 0.3)
    y = ops.conv_gemm(nhwc(x).to(cuda), pw, corr=corr.to(cuda).contiguous(), residual=nhwc(res).to(cuda),
                      depth=d.to(cuda), depth_in=nhwc(din).to(cuda), split_k=split_k, tile=tile)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    cls = torch.ones(H, dtype=torch.long); cls[0] = 0; cls[-1] = 2
    cmap = cls[:, None] * 3 + cls[None, :]                      # [H, W]
    ref = ref + corr[0][cmap].permute(2, 0, 1)[None]
    ref = ref + res.float()
    dm = d.repeat(2)[:, None, None, None]
    ref = (1 - dm) * din.float() + dm * ref
    assert rel_l2(y.float().cpu().permute(0, 3, 1, 2), ref) <= REL_L2_TOL


@pytest.mark.parametrize("split_k,tile", [(1, 0), (2, 0), (1, 6), (1, 21), (2, 22), (1, 9), (1, 15), (1, 58), (2, 60), (1, 50), (2, 65), (1, 66), (2, 66)])
def test_linear_geglu(ops, cuda, split_k, tile, both_epilogues, both_splitk):
    g = torch.Generator().manual_seed(13)
    B, L, C, inner = 2, 96, 64, 256
    x = _rand((B, L, C), g).bfloat16()
    w = _rand((2 * inner, C), g, 0.125).bfloat16()
    b = _rand((2 * inner,), g, 0.1)
    gate = (torch.rand((B, 32), generator=g) > 0.3).float()
    pw = ops.pack_weight(w.float(), b, geglu=True, device=cuda)
    y = ops.linear(x.to(cuda), pw, colgate=gate.to(cuda).contiguous(), gate_group=inner // 32, split_k=split_k, tile=tile)
    hcat = F.linear(x.float(), w.float(), b)
    h, gg = hcat.chunk(2, dim=-1)
    m = gate.repeat_interleave(inner // 32, dim=1)[:, None, :]
    ref = (h * m) * F.gelu(gg * m)
    assert y.shape == ref.shape
    assert rel_l2(y.float().cpu(), ref) <= REL_L2_TOL


def test_linear_compact_geglu_and_silu_f32(ops, cuda, both_epilogues):
    g = torch.Generator().manual_seed(17)
    B, L, C, inner = 1, 50, 128, 512
    x = _rand((B, L, C), g).bfloat16()
    w = _rand((2 * inner, C), g, 0.09).bfloat16()
    b = _rand((2 * inner,), g, 0.1)
    keep = torch.arange(inner)[(torch.arange(inner) // 16) % 3 != 1]     # arbitrary live hidden units
    pw = ops.pack_weight(w.float(), b, geglu=True, out_idx=keep, device=cuda)
    y = ops.linear(x.to(cuda), pw)
    hcat = F.linear(x.float(), w.float(), b)
    h, gg = hcat.chunk(2, dim=-1)
    ref = (h * F.gelu(gg))[..., keep]
    n = keep.numel()
    assert rel_l2(y.float().cpu()[..., :n], ref) <= REL_L2_TOL
    assert float(y[..., n:].abs().max()) == 0.0 if y.shape[-1] > n else True
    # SiLU epilogue with fp32 output (time embedding MLP)
    w2 = _rand((96, C), g, 0.09).bfloat16()
    pw2 = ops.pack_weight(w2.float(), None, device=cuda)
    y2 = ops.linear(x.to(cuda), pw2, act=ops.ACT_SILU, out_f32=True)
    assert y2.dtype == torch.float32
    assert rel_l2(y2.cpu(), F.silu(F.linear(x.float(), w2.float()))) <= 1e-5


GN_CASES = [
    # B, H, W, C, groups, silu, eps
    (2, 16, 16, 64, 32, True, 1e-5),
    (2, 8, 8, 320, 32, True, 1e-5),
    (1, 32, 32, 320, 32, False, 1e-6),
    (2, 8, 8, 2560, 32, True, 1e-5),     # two octet pages
    (2, 8, 8, 1920, 32, True, 1e-5),
    (3, 4, 4, 170, 17, True, 1e-5),      # compacted: 17 live groups of 10, padded to 176 columns
    (2, 8, 8, 960, 32, True, 1e-5),
    (2, 16, 16, 1280, 32, True, 1e-5),   # level-16 map: 16 slab rows per thread
    (1, 16, 16, 2560, 32, True, 1e-5),
    (2, 16, 16, 85, 17, True, 1e-5),     # 17 live groups of 5: 8 groups per workgroup, ragged last one, padded to 88
    (2, 16, 16, 340, 17, False, 1e-5),   # groups of 20: 2 per workgroup, ragged last one
    (1, 16, 16, 2048, 8, True, 1e-5),    # groups of 256 = 32 chunks per slab row
    (1, 32, 32, 640, 32, True, 1e-5),
    (1, 64, 64, 320, 32, True, 1e-5),
]


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("case", GN_CASES)
def test_groupnorm(ops, cuda, case, variant, both_gn_forms):
    """variant 1 = three launches (the form the backward consumes), 2 = one launch with group-owning workgroups,
    0 = the library's choice; all against F.group_norm in fp32"""
    B, H, W, C, G, silu, eps = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = (_rand((B, C, H, W), g) * 2.0 + 0.7).bfloat16()
    gamma = 1.0 + 0.2 * _rand((C,), g)
    beta = 0.3 * _rand((C,), g)
    Cp = ops.round_up(C, 8)
    xin = torch.full((B, H, W, Cp), 0.0, dtype=torch.bfloat16)
    xin[..., :C] = nhwc(x)
    try:
        y = ops.groupnorm(xin.to(cuda), gamma.to(cuda), beta.to(cuda), G, eps, silu, C=C, variant=variant)
    except Exception as e:
        if variant == 2 and "does not fit" in str(e):
            pytest.skip("slab too large for the single-launch form")
        raise
    ref = F.group_norm(x.float(), G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    got = y.float().cpu()
    assert rel_l2(got[..., :C].permute(0, 3, 1, 2), ref) <= REL_L2_TOL
    if Cp > C:
        assert float(got[..., C:].abs().max()) == 0.0


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_groupnorm_strided_view_and_determinism(ops, cuda, variant, both_gn_forms):
    """input = channel slice of a wider (concat) buffer, output into a slice too; two runs are bit-identical"""
    B, H, W, C, G = 2, 16, 16, 640, 32
    g = torch.Generator().manual_seed(11)
    wide = (_rand((B, H, W, C + 320), g) * 3 + 1).bfloat16().to(cuda)
    x = wide[..., 320:]
    gamma = (1.0 + 0.2 * _rand((C,), g)).to(cuda)
    beta = (0.3 * _rand((C,), g)).to(cuda)
    outw = torch.zeros(B, H, W, C + 64, dtype=torch.bfloat16, device=cuda)
    y = ops.groupnorm(x, gamma, beta, G, 1e-5, True, out=outw[..., :C], variant=variant)
    y2 = ops.groupnorm(x.contiguous(), gamma, beta, G, 1e-5, True, variant=variant)
    assert torch.equal(y, y2)
    assert float(outw[..., C:].abs().max()) == 0.0
    ref = F.silu(F.group_norm(x.float().permute(0, 3, 1, 2), G, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
    assert rel_l2(y.float().cpu(), ref.cpu()) <= REL_L2_TOL


def test_groupnorm_variants_agree(ops, cuda):
    """the single-launch and three-launch forms see the same statistics (different fold order only)"""
    g = torch.Generator().manual_seed(12)
    x = (_rand((4, 8, 8, 1280), g) * 2 - 0.3).bfloat16().to(cuda)
    gamma = (1.0 + 0.2 * _rand((1280,), g)).to(cuda)
    beta = (0.3 * _rand((1280,), g)).to(cuda)
    a = ops.groupnorm(x, gamma, beta, 32, 1e-5, True, variant=1).float()
    b = ops.groupnorm(x, gamma, beta, 32, 1e-5, True, variant=2).float()
    assert float((a - b).abs().max()) <= 2 ** -6 * float(a.abs().max())


def test_groupnorm_zero_group_gives_beta(ops, cuda):
    """a fully zeroed group normalises to exactly beta"""
    B, H, W, C, G = 1, 8, 8, 64, 32
    g = torch.Generator().manual_seed(3)
    x = _rand((B, H, W, C), g).bfloat16()
    x[..., 10:12] = 0
    gamma = torch.ones(C)
    beta = 0.25 * torch.arange(C, dtype=torch.float32) / C
    y = ops.groupnorm(x.to(cuda), gamma.to(cuda), beta.to(cuda), G, 1e-5, False)
    assert torch.equal(y[..., 10:12].float().cpu(), beta[10:12].bfloat16().float().expand(B, H, W, 2))


@pytest.mark.parametrize("rows,C", [(7, 64), (128, 320), (130, 640), (33, 1280)])
def test_layernorm(ops, cuda, rows, C):
    g = torch.Generator().manual_seed(rows * C)
    x = (_rand((1, rows, C), g) * 1.5 + 0.4).bfloat16()
    gamma = 1.0 + 0.2 * _rand((C,), g)
    beta = 0.3 * _rand((C,), g)
    y = ops.layernorm(x.to(cuda), gamma.to(cuda), beta.to(cuda), 1e-5)
    ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    assert rel_l2(y.float().cpu(), ref) <= REL_L2_TOL


@pytest.mark.parametrize("rows,C,N,tile", [(96, 64, 64, 0), (200, 192, 320, 0), (2048, 128, 320, 9), (300, 640, 640, 21),
                                            (128, 320, 320, 34), (130, 640, 1280, 38), (256, 1280, 1280, 42),
                                            (300, 640, 640, 58), (1024, 1280, 1280, 60), (300, 640, 640, 64), (1024, 1280, 1280, 66),
                                            (2048, 320, 320, 65)])
def test_linear_emits_row_statistics(ops, cuda, rows, C, N, tile, both_epilogues):
    """the per-row (sum, sumsq) partials a producer GEMM emits (one slot per N-tile x wave column) add up to the
    statistics of the bf16 values it stored, for 4- and 8-wave tiles, ragged rows and a residual in the epilogue"""
    g = torch.Generator().manual_seed(rows + C + N)
    x = _rand((1, rows, C), g).bfloat16()
    w = _rand((N, C), g, 0.1).bfloat16()
    b = _rand((N,), g, 0.5)
    res = (_rand((1, rows, N), g) + 0.7).bfloat16()
    pw = ops.pack_weight(w.float(), b, device=cuda)
    y, st = ops.linear(x.to(cuda), pw, residual=res.to(cuda), rowstats=True, tile=tile, split_k=1)
    ref = F.linear(x.float(), w.float(), b) + res.float()
    assert rel_l2(y.float().cpu(), ref) <= REL_L2_TOL
    assert st.shape[1] == rows and st.shape[2] == 4            # two (sum, sumsq) slots per 16-byte element
    tot = st.sum(0).cpu().double()
    yf = y[0].double().cpu()
    assert torch.allclose(tot[:, 0] + tot[:, 2], yf.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(tot[:, 1] + tot[:, 3], (yf * yf).sum(1), rtol=1e-5, atol=1e-3)
    # split along K: the workgroup that combines the slices emits the statistics (in-kernel reduction) ...
    y2, st2 = ops.linear(x.to(cuda), pw, residual=res.to(cuda), rowstats=True, tile=tile, split_k=2)
    assert rel_l2(y2.float().cpu(), ref) <= REL_L2_TOL
    if C >= 128:      # (a single K-step cannot be split)
        y2f = y2[0].double().cpu()
        t2 = st2.sum(0).cpu().double()
        assert torch.allclose(t2[:, 0] + t2[:, 2], y2f.sum(1), rtol=1e-5, atol=1e-3)
    # ... and with the separate reduce launch there are none: the caller falls back to the LayerNorm kernel
    ops.SPLITK_IN_KERNEL = False
    try:
        y3, st3 = ops.linear(x.to(cuda), pw, residual=res.to(cuda), rowstats=True, split_k=2)
    finally:
        ops.SPLITK_IN_KERNEL = True
    assert (st3 is None or C < 128) and rel_l2(y3.float().cpu(), ref) <= REL_L2_TOL


@pytest.mark.parametrize("rows,C,N,tile,split_k,geglu", [(96, 64, 192, 0, 1, False), (200, 320, 384, 0, 1, False),
                                                         (2048, 320, 128, 18, 1, False), (256, 1280, 640, 25, 3, False),
                                                         (130, 640, 2560, 21, 1, True), (64, 1280, 5120, 40, 1, True),
                                                         (300, 320, 1280, 9, 2, True), (1024, 1280, 1920, 59, 1, False), (300, 640, 2560, 61, 2, True),
                                                         (300, 640, 1280, 64, 2, False), (1024, 1280, 1920, 64, 1, False), (300, 640, 2560, 65, 2, True),
                                                         (2048, 320, 2560, 66, 2, True)])
def test_linear_with_folded_layernorm(ops, cuda, rows, C, N, tile, split_k, geglu, both_epilogues, both_splitk):
    """linear(LayerNorm(x)) in one launch (gamma folded into the weights, mean / rstd from the producer's row
    statistics) against F.layer_norm + F.linear in fp32 on the same bf16 x: blocks.py:782-785,808-813,821-823,41-50"""
    g = torch.Generator().manual_seed(rows * 7 + C + N)
    gamma = 1.0 + 0.2 * _rand((C,), g)
    beta = 0.3 * _rand((C,), g)
    # the producer: an identity-free GEMM whose stored output IS the LayerNorm input (mean 0.4, spread 1.5, like test_layernorm)
    src = _rand((1, rows, 64), g).bfloat16()
    wp = _rand((C, 64), g, 0.19).bfloat16()
    bp = torch.full((C,), 0.4)
    x, st = ops.linear(src.to(cuda), ops.pack_weight(wp.float(), bp, device=cuda), rowstats=True, split_k=1)
    w = _rand((N, C), g, 0.06).bfloat16()
    b = _rand((N,), g, 0.2)
    pw = ops.pack_weight(w.float(), b, geglu=geglu, device=cuda, ln_gamma=gamma, ln_beta=beta)
    kw = {}
    if geglu:
        gate = (torch.rand((1, 32), generator=g) > 0.3).float()
        kw = dict(colgate=gate.to(cuda).contiguous(), gate_group=N // 2 // 32)
    y = ops.linear(x, pw, ln=(st, 1e-5), tile=tile, split_k=split_k, **kw)
    n = F.layer_norm(x.float().cpu(), (C,), gamma, beta, 1e-5)
    ref = F.linear(n, w.float(), b)
    if geglu:
        h, gg = ref.chunk(2, dim=-1)
        m = gate.repeat_interleave(N // 2 // 32, dim=1)[:, None, :]
        ref = (h * m) * F.gelu(gg * m)
    assert y.shape == ref.shape
    assert rel_l2(y.float().cpu(), ref) <= REL_L2_TOL
    with pytest.raises(ValueError):
        ops.linear(x, pw)                       # folded weights without statistics


ATTN_CASES = [
    # B, heads, Lq, Lk
    (1, 1, 64, 64),
    (1, 1, 128, 128),      # two key tiles: one per wave group
    (1, 2, 200, 192),      # three key tiles: group 1 owns a single tile
    (4, 2, 4096, 4096),    # SD-2.1 level-64 self-attention of the masked model
    (2, 2, 256, 256),
    (2, 5, 1024, 1024),
    (2, 3, 256, 77),       # cross attention, ragged key tile
    (1, 2, 100, 77),       # ragged queries too
    (1, 4, 64, 200),
]


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention(ops, cuda, case, both_attn_forms):
    B, h, Lq, Lk = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    q = _rand((B, Lq, h * 64), g).bfloat16()
    k = _rand((B, Lk, h * 64), g).bfloat16()
    v = _rand((B, Lk, h * 64), g).bfloat16()
    o = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), h)

    def heads(t, L):
        return t.float().view(B, L, h, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(heads(q, Lq), heads(k, Lk), heads(v, Lk))
    ref = ref.transpose(1, 2).reshape(B, Lq, h * 64)
    e = rel_l2(o.float().cpu(), ref)
    assert e <= 6e-3, f"rel-L2 {e:.3e}"   # P is rounded to bf16 before P.V (as any bf16 flash kernel does)


@pytest.mark.parametrize("case", [(1, 5, 4096), (2, 10, 1024), (2, 20, 256), (4, 2, 4096)])
def test_attention_software_pipelined_kernel_is_race_free_and_equals_the_double_buffered_kernel(ops, cuda, case):
    """attn_fwd_sp_kernel (what `auto` takes on whole 128-blocks) hand-places MFMAs between the softmax instructions; its running
    maximum is inline asm (v_max3_f32) and hipcc's hazard recognizer does not look into inline asm -- a first version read MFMA
    results too early and its outputs changed from run to run.  Twenty repeats on the same operands must agree bit for bit, output
    and log-sum-exp, and equal the double-buffered kernel's (same arithmetic in the same order); backward: repeats agree bit for bit."""
    B, h, L = case
    g = torch.Generator().manual_seed(B * 1000 + h)
    q, k, v, do = ((_rand((B, L, h * 64), g) * a).bfloat16().to(cuda) for a in (2.0, 2.0, 1.0, 1.0))
    ops.ATTN_VARIANT = 4
    try:
        lse4 = torch.zeros(B, h, L, device=cuda)
        o4 = ops.attention(q, k, v, h, lse=lse4)
    finally:
        ops.ATTN_VARIANT = 0
    dq0 = dk0 = dv0 = None
    for it in range(20):
        lse = torch.zeros(B, h, L, device=cuda)
        o = ops.attention(q, k, v, h, lse=lse)
        assert torch.equal(o, o4) and torch.equal(lse, lse4), f"repeat {it}: {int((o != o4).sum())} outputs, {int((lse != lse4).sum())} lse differ"
        if it % 5 == 0:
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            ops.attention_bwd(q, k, v, o4, do, lse4, h, dq, dk, dv)
            if dq0 is None:
                dq0, dk0, dv0 = dq, dk, dv
            assert torch.equal(dq, dq0) and torch.equal(dk, dk0) and torch.equal(dv, dv0)


@pytest.mark.parametrize("L,spike", [(320, 6.0), (512, 6.0), (512, 1.6), (1024, 30.0)])
def test_attention_fused_qkv_views_and_spike(ops, cuda, both_attn_forms, L, spike):
    """q/k/v as column slices of one fused buffer; one key spiked so the running max jumps mid-sequence (L = 512 / 1024: shapes the
    software-pipelined kernel takes -- its reference maximum moves only on jumps above 2^6, so a small and two large jumps)"""
    g = torch.Generator().manual_seed(5)
    B, h = 2, 2
    qkv = _rand((B, L, 3 * h * 64), g).bfloat16()
    qkv[:, 200, h * 64:2 * h * 64] *= spike    # large-norm key in the 4th key tile
    dq = qkv.to(cuda)
    o = ops.attention(dq[..., :h * 64], dq[..., h * 64:2 * h * 64], dq[..., 2 * h * 64:], h)

    def heads(t):
        return t.float().reshape(B, L, h, 64).transpose(1, 2)
    q, k, v = qkv[..., :h * 64], qkv[..., h * 64:2 * h * 64], qkv[..., 2 * h * 64:]
    ref = F.scaled_dot_product_attention(heads(q), heads(k), heads(v)).transpose(1, 2).reshape(B, L, h * 64)
    assert rel_l2(o.float().cpu(), ref) <= 6e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unet_prologue_and_epilogue(ops, cuda, dtype):
    """the non-GEMM ends of UNet2DConditionModel.forward (unet_2d_conditional.py:1497-1519,1614,1721-1726): channel-padded
    channels-last staging of the sample + the [cos|sin] timestep embedding, and the NCHW output in the caller's dtype"""
    g = torch.Generator().manual_seed(5)
    B, C, H, W, half = 3, 4, 24, 40, 160
    sample = _rand((B, C, H, W), g).to(dtype)
    t = torch.tensor([0, 500, 999])
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    x, temb = ops.unet_prologue(sample.to(cuda), t.to(cuda), freqs.to(cuda), 8)
    assert x.shape == (B, H, W, 8) and float(x[..., C:].abs().max()) == 0.0
    assert torch.equal(x[..., :C].cpu(), sample.permute(0, 2, 3, 1).to(torch.bfloat16))
    ang = t.float()[:, None] * freqs[None, :]
    ref = torch.cat([torch.cos(ang), torch.sin(ang)], -1)
    assert float((temb.float().cpu() - ref).abs().max()) <= 2 ** -8 + 2e-4      # bf16 rounding + sin/cos at |x| <= 1000
    y = _rand((B, H, W, 8), g).to(cuda)
    out = ops.unet_epilogue(y, C, dtype)
    assert out.dtype == dtype and torch.equal(out.cpu(), y[..., :C].permute(0, 3, 1, 2).to(dtype).cpu())


def test_bad_arguments_fail_loudly(ops, cuda):
    from diffusion_pruning_amd._lib import AptpError
    x = torch.zeros(1, 4, 4, 12, dtype=torch.bfloat16, device=cuda)   # Cin not a multiple of 8
    pw = ops.pack_weight(torch.zeros(8, 12, 3, 3), None, cin_pad_to=1, device=cuda)
    with pytest.raises((AptpError, ValueError)):
        ops.conv_gemm(x, pw)


@pytest.mark.parametrize("B,L,C,inner,x_wide", [(2, 1024, 320, 640, 0), (1, 4096, 320, 1280, 320), (2, 256, 64, 256, 0),
                                               (1, 1024, 128, 200, 0), (4, 64, 256, 1024, 0)])
def test_ff_tail_fused_kernel(cuda, B, L, C, inner, x_wide):
    """aptp_ff_tail: LN3 -> GEGLU projection -> ff.net[2] + residual -> proj_out + residual in one kernel per 64-token tile,
    against (a) fp32 PyTorch on the same bf16 operands with the same bf16 rounding points and (b) the four separate launches;
    inner = live hidden width (640 = the 50 % mask at level 64, 200 = an irregular expert: not a multiple of 64)."""
    from diffusion_pruning_amd import ops
    g = torch.Generator().manual_seed(C * 7 + inner)
    M = B * L
    hbuf = (torch.randn(B, L, C, generator=g) * 1.5).bfloat16()
    xbuf = torch.randn(B, L, C + x_wide, generator=g).bfloat16()
    ln_g, ln_b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    w1 = torch.randn(2 * inner, C, generator=g) / C ** 0.5
    b1 = torch.randn(2 * inner, generator=g) * 0.1
    w2 = torch.randn(C, inner, generator=g) / inner ** 0.5
    b2 = torch.randn(C, generator=g) * 0.1
    w3 = torch.randn(C, C, generator=g) / C ** 0.5
    b3 = torch.randn(C, generator=g) * 0.1
    # full-width GEGLU module with `inner` live units = what a compacted plan packs: rows [0, inner) of each half
    pw1 = ops.pack_weight(w1, b1, geglu=True, device=cuda, ln_gamma=ln_g, ln_beta=ln_b)
    pw1_plain = ops.pack_weight(w1, b1, geglu=True, device=cuda)
    pw2 = ops.pack_weight(w2, b2, cin_pad_to=16, device=cuda)
    pw3 = ops.pack_weight(w3, b3, device=cuda)
    h, x = hbuf.to(cuda), xbuf.to(cuda)[..., :C]
    old_min, old_fuse = ops.FUSE_TAIL_MIN_ROWS, ops.FUSE_TAIL
    ops.FUSE_TAIL_MIN_ROWS, ops.FUSE_TAIL = 64, True          # (the fused tail is off by default since round 4)
    try:
        assert ops.ff_tail_supported(h, pw1, pw2, pw3)
        y = ops.ff_tail(h, x, pw1, pw2, pw3, 1e-5, colstats=True)
    finally:
        ops.FUSE_TAIL_MIN_ROWS, ops.FUSE_TAIL = old_min, old_fuse
    torch.cuda.synchronize()
    # (a) fp32 reference with the kernel's rounding points (f and h3 are bf16; weights are bf16)
    hf, xf = hbuf.float(), xbuf[..., :C].float()
    bfw = lambda t: t.bfloat16().float()
    n = F.layer_norm(hf, (C,), ln_g, ln_b, 1e-5)
    w1f = bfw(w1 * ln_g[None, :])                                   # the fold rounds w * gamma to bf16
    pre = (n - ln_b) / ln_g                                          # = (h - mean) * rstd
    hg = pre @ w1f.t() + (b1 + w1 @ ln_b)
    f = bfw(hg[..., :inner] * F.gelu(hg[..., inner:]))
    h3 = bfw(f @ bfw(w2).t() + b2 + hf)
    ref = h3 @ bfw(w3).t() + b3 + xf
    assert rel_l2(y.float().cpu(), ref) <= 4e-3
    # (b) the separate launches
    nn_ = ops.layernorm(h, ln_g.to(cuda), ln_b.to(cuda), 1e-5)
    ff = ops.linear(nn_, pw1_plain)
    h3u = ops.linear(ff, pw2, residual=h)
    yu = ops.linear(h3u, pw3, residual=x.contiguous())
    assert rel_l2(y.float(), yu.float()) <= 4e-3
    # column statistics of the stored bf16 values, one (sum, sumsq) per (64-row block, channel)
    if L >= ops.COLSTATS_MIN_HW:
        rec = ops._colstats_get(y.unsqueeze(2), C)
        assert rec is not None and rec[0][1] == 64
        st = rec[0][0].float().cpu()
        yb = y.float().cpu().reshape(M // 64, 64, C)
        assert torch.allclose(st[..., 0], yb.sum(1), rtol=1e-4, atol=1e-2)
        assert torch.allclose(st[..., 1], (yb * yb).sum(1), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("B,H,Cin,N,groups,live", [(4, 16, 1280, 640, 16, 640), (2, 8, 2560, 640, 16, 640), (1, 16, 640, 1280, 32, 1280),
                                                   (2, 8, 320, 88, 11, 88), (2, 16, 128, 64, 16, 64)])
def test_conv_with_groupnorm_in_the_splitk_reduce(cuda, B, H, Cin, N, groups, live):
    """conv_gemm(gn=...): on the small maps the reduce launch of a split-K convolution also applies the GroupNorm(+SiLU) that
    follows it (ResnetBlock2D conv1 + temb -> norm2 -> SiLU): same values as the two separate launches up to the group
    statistics' summation order, and within the per-op tolerance of an fp32 reference."""
    from diffusion_pruning_amd import ops
    g = torch.Generator().manual_seed(Cin + N)
    x = torch.randn(B, H, H, Cin, generator=g).bfloat16()
    w = torch.randn(N, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    rb = torch.randn(B, N, generator=g) * 0.2
    gamma, beta = torch.rand(live, generator=g) + 0.5, torch.randn(live, generator=g) * 0.1
    pw = ops.pack_weight(w, b, device=cuda)
    xd, rbd, gd, bd = x.to(cuda), rb.to(cuda).contiguous(), gamma.to(cuda), beta.to(cuda)
    spec = (gd, bd, groups, 1e-5, True, live)
    old = ops.FUSE_GN_REDUCE
    try:
        ops.FUSE_GN_REDUCE = True
        ops.LAUNCH_LOG = []
        fused = ops.conv_gemm(xd, pw, rowbias=rbd, gn=spec, split_k=4)
        assert ops.LAUNCH_LOG[-1]["params"].gn_gamma                 # the reduce launch really carried the normalisation
        ops.FUSE_GN_REDUCE = False
        sep = ops.conv_gemm(xd, pw, rowbias=rbd, gn=spec, split_k=4)
    finally:
        ops.FUSE_GN_REDUCE, ops.LAUNCH_LOG = old, None
    torch.cuda.synchronize()
    assert fused.shape == sep.shape == (B, H, H, pw.N)
    assert rel_l2(fused.float(), sep.float()) <= 2e-3
    assert bool((fused[..., live:] == 0).all())
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w.bfloat16().float(), b, padding=1) + rb[:, :, None, None]
    ref = F.silu(F.group_norm(y[:, :live].bfloat16().float(), groups, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
    assert rel_l2(fused[..., :live].float().cpu(), ref) <= 6e-3


def test_capture_time_scratch_buffers_are_never_released(ops, cuda):
    """A scratch buffer handed out while a stream is capturing lives in that graph's private pool and its address is baked into
    every launch captured with it -- also into graphs captured later on the same stream.  When a larger request replaced it in
    the cache after its owning graph had died, torch.cuda.graph.__enter__'s empty_cache() returned the block to the device and
    the younger graph's replays wrote through a stale address ("write access to a read-only page", for some test orders only).
    The policy that prevents it: such buffers are kept for the life of the process (ops._ws_capture_keep)."""
    import gc
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        w1 = ops._workspace(1 << 20, cuda)
        w1.zero_()
        p1, n1 = w1.data_ptr(), w1.numel()
    del w1, g1
    gc.collect()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        small = ops._workspace(1 << 10, cuda)              # served from the cached buffer of the dead graph
        small.fill_(7)
        assert small.data_ptr() == p1
        big = ops._workspace(n1 * 4, cuda)                 # replaces it in the cache
        big.zero_()
    del small, big
    gc.collect()
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):                             # __enter__ empties the caching allocator
        torch.zeros(8, device=cuda)
    kept = [b for b in ops._ws_capture_keep if b.data_ptr() == p1]
    assert kept and kept[0].numel() == n1
    g2.replay()                                            # writes through p1: must still be this process's memory
    torch.cuda.synchronize()
    assert bool((kept[0][:1 << 10] == 7).all())


LIN_TILES = [12, 18, 25, 49, 9, 15, 24, 51, 11, 17, 26, 53]


@pytest.mark.parametrize("tile", LIN_TILES)
@pytest.mark.parametrize("M,Cin,N,res,ln,silu", [
    (4096, 320, 640, True, False, False),      # a1_out at level 32: + residual, row statistics out
    (1000, 160, 328, True, True, False),       # ragged M and N tails, channel padding in the last K-tile, folded LayerNorm
    (16384, 128, 320, False, False, False),    # short K (2 K-tiles), column statistics for a GroupNorm
    (300, 1280, 1288, True, True, True),       # long K, 20 producer slot pairs, SiLU
    (4, 320, 1280, False, False, True),        # time-embedding MLP: four rows
])
def test_lean_linear_kernel_equals_the_general_kernel(ops, cuda, tile, M, Cin, N, res, ln, silu):
    """csrc/lin_gemm.hip (plain linear layers: blocks.py:228-268,776-849) against conv_gemm_dma_kernel on the same tile shape
    (AptpConvGemmParams.epilogue = 2 keeps the launch on the general kernel) and against fp32 PyTorch: the accumulation order is
    the same, so y must be bit-identical; row / column statistics may differ by their summation order."""
    g = torch.Generator().manual_seed(M + Cin + N + tile)
    H = M
    x = (_rand((1, H, 1, Cin), g)).bfloat16().to(cuda)
    w = _rand((N, Cin, 1, 1), g, 1.0 / math.sqrt(Cin))
    b = _rand((N,), g, 0.1)
    r = _rand((1, H, 1, N), g).bfloat16().to(cuda) if res else None
    kw = dict(tile=tile, residual=r, act=ops.ACT_SILU if silu else ops.ACT_NONE, pad=0)
    if ln:
        gamma, beta = torch.rand(Cin, generator=g) + 0.5, _rand((Cin,), g, 0.1)
        pw = ops.pack_weight(w, b, device=cuda, ln_gamma=gamma, ln_beta=beta)
        xs = x.float().reshape(M, Cin)
        npair = 3 if Cin < 1000 else 20            # statistics split over several producer slots (zeros in the rest of a pair)
        st = torch.zeros(npair, M, 4, dtype=torch.float32, device=cuda)
        cuts = torch.linspace(0, Cin, 2 * npair + 1).long().tolist()
        for s in range(2 * npair):
            seg = xs[:, cuts[s]:cuts[s + 1]]
            st[s // 2, :, 2 * (s % 2)] = seg.sum(1)
            st[s // 2, :, 2 * (s % 2) + 1] = (seg * seg).sum(1)
        kw["ln"] = (st, 1e-5)
    else:
        pw = ops.pack_weight(w, b, device=cuda)
    ops.EPILOGUE = 2
    try:
        y_ref, st_ref = ops.conv_gemm(x, pw, rowstats=True, **kw)
    finally:
        ops.EPILOGUE = 0
    y, st_new = ops.conv_gemm(x, pw, rowstats=True, colstats=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref), float((y.float() - y_ref.float()).abs().max())
    assert st_new is not None and st_new.shape == st_ref.shape
    yf = y.float().reshape(M, N)
    tot = st_new.sum(0)                                # [M, 4] -> (sum, sumsq) pairs of all slots
    assert torch.allclose(tot[:, 0] + tot[:, 2], yf.sum(1), rtol=1e-4, atol=2e-2)
    assert torch.allclose(tot[:, 1] + tot[:, 3], (yf * yf).sum(1), rtol=1e-4, atol=2e-2)
    assert torch.allclose(st_new, st_ref, rtol=1e-4, atol=2e-2)
    # fp32 reference on the same bf16 operands
    xf = x.float().reshape(M, Cin).cpu()
    wf = (w.reshape(N, Cin) * (gamma[None, :] if ln else 1.0)).bfloat16().float()
    if ln:
        pre = (F.layer_norm(xf, (Cin,), gamma, beta, 1e-5) - beta) / gamma
        ref = pre @ wf.t() + (b + w.reshape(N, Cin) @ beta)
    else:
        ref = xf @ wf.t() + b
    if silu:
        ref = F.silu(ref)
    if res:
        ref = ref + r.float().reshape(M, N).cpu()
    assert rel_l2(yf.cpu(), ref) <= REL_L2_TOL
    # column statistics (GroupNorm producer side) where the launch could emit them
    rec = ops._colstats_get(y, N) if (H >= ops.COLSTATS_MIN_HW) else None
    if rec is not None:
        cst, rpb = rec[0][0].float(), rec[0][1]
        nb = M // rpb
        yb = yf[:nb * rpb].reshape(nb, rpb, N)
        assert torch.allclose(cst[:nb, :, 0], yb.sum(1), rtol=1e-4, atol=2e-2)
        assert torch.allclose(cst[:nb, :, 1], (yb * yb).sum(1), rtol=1e-4, atol=2e-2)


@pytest.mark.parametrize("tile", LIN_TILES)
@pytest.mark.parametrize("M,C,inner", [(4096, 640, 1280), (1000, 160, 200), (256, 1280, 2560)])
def test_lean_geglu_projection_equals_the_general_kernel(ops, cuda, tile, M, C, inner):
    """ff.net[0] (GEGLU with the LayerNorm folded in, blocks.py:776-849) through csrc/lin_gemm.hip: bit-identical to the general
    kernel on the same tile shape up to the FMA contraction of the activation chain (<= 1 bf16 ulp on <= 0.2 % of the outputs), and
    within the per-op tolerance of fp32 PyTorch on the same bf16 operands."""
    g = torch.Generator().manual_seed(M + C + inner + tile)
    x = (_rand((1, M, 1, C), g) * 1.5).bfloat16().to(cuda)
    w1 = _rand((2 * inner, C), g, 1.0 / math.sqrt(C))
    b1 = _rand((2 * inner,), g, 0.1)
    gamma, beta = torch.rand(C, generator=g) + 0.5, _rand((C,), g, 0.1)
    pw = ops.pack_weight(w1, b1, geglu=True, device=cuda, ln_gamma=gamma, ln_beta=beta)
    xs = x.float().reshape(M, C)
    st = torch.zeros(2, M, 4, dtype=torch.float32, device=cuda)
    h = C // 3 // 8 * 8
    for s_, (a, b) in enumerate(((0, h), (h, 2 * h), (2 * h, C))):
        st[s_ // 2, :, 2 * (s_ % 2)] = xs[:, a:b].sum(1)
        st[s_ // 2, :, 2 * (s_ % 2) + 1] = (xs[:, a:b] * xs[:, a:b]).sum(1)
    ops.EPILOGUE = 2
    try:
        y_ref = ops.conv_gemm(x, pw, tile=tile, pad=0, ln=(st, 1e-5))
    finally:
        ops.EPILOGUE = 0
    y = ops.conv_gemm(x, pw, tile=tile, pad=0, ln=(st, 1e-5))
    torch.cuda.synchronize()
    assert y.shape[-1] == pw.N // 2
    # same accumulators; the h * gelu(g) chain may be contracted into FMAs differently by the compiler in the two epilogues, so a
    # few outputs land on the other side of a bf16 rounding boundary: at most one ulp, and rarely
    a, b = y.float(), y_ref.float()
    diff = (a - b).abs()
    assert bool((diff <= torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + 1e-6).all()), float(diff.max())
    assert float((diff > 0).float().mean()) <= 2e-3
    xf = xs.cpu()
    pre = (F.layer_norm(xf, (C,), gamma, beta, 1e-5) - beta) / gamma
    w1f = (w1 * gamma[None, :]).bfloat16().float()
    hg = pre @ w1f.t() + (b1 + w1 @ beta)
    ref = hg[:, :inner] * F.gelu(hg[:, inner:])
    assert rel_l2(y.float().reshape(M, -1)[:, :inner].cpu(), ref) <= REL_L2_TOL


@pytest.mark.parametrize("tile", [0, 18, 11, 34, 19])
@pytest.mark.parametrize("B,H,C0,C1,k,groups", [(4, 32, 320, 0, 1, 32), (2, 64, 160, 160, 3, 32), (4, 32, 640, 320, 3, 32), (2, 32, 170, 0, 3, 17)])
def test_groupnorm_from_unit_statistics_needs_no_finalise_launch(ops, cuda, tile, B, H, C0, C1, k, groups):
    """Round 4: a producer inside an ops.ustat_begin() bracket also leaves per-(sample, 10-channel unit) fixed-point sums (64-bit
    integer atomics: AptpConvGemmParams.ustat_out); F.group_norm of its output (blocks.py:296-301,350-359) then runs as ONE launch.
    One producer or two (skip-concat, blocks.py:485-495), lean and general kernels, compacted 17 x 10 channels: the unit sums
    equal the sums of the stored bf16 values, the GroupNorm equals the finalise-launch form, and repeated runs are bit-identical
    (integer accumulation does not depend on the order of arrival)."""
    g = torch.Generator().manual_seed(B * H + C0 + C1 + k + tile)
    unit = 10
    Ctot = C0 + C1
    cat = torch.empty(B, H, H, (Ctot + 7) // 8 * 8, dtype=torch.bfloat16, device=cuda)
    cat.zero_()
    segs = [(0, C0)] + ([(C0, C1)] if C1 else [])
    xs, pws = [], []
    for (off, Cn) in segs:
        Cin = 128 if k == 1 else 64
        xs.append(nhwc(_rand((B, Cin, H, H), g)).bfloat16().to(cuda))
        pws.append(ops.pack_weight(_rand((Cn, Cin, k, k), g, 1.0 / math.sqrt(Cin * k * k)), _rand((Cn,), g, 0.3), device=cuda))
    gamma, beta = (torch.rand(Ctot, generator=g) + 0.5).to(cuda), (_rand((Ctot,), g, 0.1)).to(cuda)

    def produce(with_units):
        if with_units:
            ops.ustat_begin(cuda, unit)
        try:
            for (off, Cn), x, pw in zip(segs, xs, pws):
                Cp = pw.N
                ops.conv_gemm(x, pw, out=cat[..., off:off + Cp] if (off + Cp <= cat.shape[3]) else None, colstats=True, tile=tile)
        finally:
            ops.ustat_end()
    if any(o + pw.N > cat.shape[3] or (o % 8) for (o, _), pw in zip(segs, pws)):
        pytest.skip("segment layout not expressible as channel slices of one buffer")
    produce(True)
    rec = ops._colstats_get(cat, Ctot)
    assert rec is not None and all(r[3] is not None for r in rec), "every producer must have left unit statistics"
    # the unit sums themselves
    yf = cat.float()
    for (off, Cn), r in zip(segs, rec):
        us, u_unit, units, nrep = r[3]
        tot = us.view(nrep, B, units, 2).sum(0).double()
        ref = yf[..., off:off + Cn].reshape(B, H * H, Cn)
        nu = Cn // unit
        rs = ref[..., :nu * unit].reshape(B, H * H, nu, unit).double()
        assert torch.allclose(tot[:, :nu, 0] / 2 ** 20, rs.sum((1, 3)), rtol=1e-5, atol=2e-2)
        assert torch.allclose(tot[:, :nu, 1] / 2 ** 12, (rs * rs).sum((1, 3)), rtol=1e-5, atol=5e-1)
    ops.GN_LAUNCH_LOG = []
    try:
        y_units = ops.groupnorm(cat, gamma, beta, groups, 1e-5, True, C=Ctot)
        used = ops.GN_LAUNCH_LOG[-1]["params"].colstats[0].ustats
    finally:
        ops.GN_LAUNCH_LOG = None
    assert used, "the GroupNorm must have taken the unit statistics"
    snapshot = [r[3][0].clone() for r in rec]
    produce(False)                                   # same producers, column statistics only: the finalise-launch form
    y_fin = ops.groupnorm(cat, gamma, beta, groups, 1e-5, True, C=Ctot)
    ref = F.silu(F.group_norm(yf[..., :Ctot].permute(0, 3, 1, 2), groups, gamma.float(), beta.float(), 1e-5)).permute(0, 2, 3, 1)
    assert rel_l2(y_units[..., :Ctot].float(), ref) <= REL_L2_TOL
    assert rel_l2(y_units.float(), y_fin.float()) <= 2e-4
    for _ in range(5):                               # deterministic: integer sums, whatever the arrival order
        produce(True)
        rec2 = ops._colstats_get(cat, Ctot)
        for a, r in zip(snapshot, rec2):
            assert torch.equal(a, r[3][0])
        assert torch.equal(ops.groupnorm(cat, gamma, beta, groups, 1e-5, True, C=Ctot), y_units)
