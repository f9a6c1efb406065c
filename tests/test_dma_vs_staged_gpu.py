"""The fp32 parity path (tests/test_fp32_parity_gpu.py: <= 8e-7 per op, <= 5e-6 whole U-Net against the oracle) instantiates the
REGISTER-STAGED kernel (conv_gemm_kernel<BM, BN, 2, 2, T>, tiles 1..6); every launch of the benchmarked forward is an LDS-DMA
instantiation (conv_gemm_dma_kernel / the fused transformer tail) whose gather, swizzled-source addressing, ring schedule
and split-K combine are separate code.  This file transfers the pin: for every distinct (shape, tile, split-K, epilogue
form) the HEADLINE forward launches (bs=4, fixed 50 % mask, SD-2.1 size; recorded through ops.LAUNCH_LOG), the launch is
repeated on the SAME bf16 operands by its DMA tile and by a register-staged tile, and the two must agree
  (1) with fp32 outputs: to fp32 re-association (different K order of partial sums: <= 2e-6 relative to the output's RMS),
  (2) with the launch's own bf16 epilogue (coalesced, through LDS): every element within one bf16 ulp, and only the few
      elements whose fp32 value sits on a rounding boundary differ at all.
Reference call sites of the ops: blocks.py:228-268,297-369,776-849 (F.conv2d / F.linear)."""
import ctypes

import pytest
import torch

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
from tests.margins import check  # noqa: E402

def _clone(p):
    return type(p).from_buffer_copy(p)


@pytest.fixture(scope="module")
def headline_log(cuda):
    from diffusion_pruning_amd import ops
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    model = UNet2DConditionModelGated().init_synthetic(seed=0).to(cuda)
    cfg = O.SD21
    model.set_structure({k: [v.to(cuda) for v in vs] for k, vs in O.fixed_half_mask(cfg).items()})
    sample, t, ehs = O.synthetic_inputs(cfg, 4, 64, seed=5)
    with torch.no_grad():
        model(sample.to(cuda), t.to(cuda), ehs.to(cuda))          # plans, prefetch order
        ops.LAUNCH_LOG = []
        try:
            model(sample.to(cuda), t.to(cuda), ehs.to(cuda))
            torch.cuda.synchronize()
            log = ops.LAUNCH_LOG
        finally:
            ops.LAUNCH_LOG = None
    return model, log


def _launch(lib, p, what):
    from diffusion_pruning_amd import _lib
    _lib.check(lib.aptp_conv_gemm(ctypes.byref(p), torch.cuda.current_stream().cuda_stream), what)


def test_every_headline_launch_dma_tile_equals_register_staged_tile(headline_log, cuda):
    from diffusion_pruning_amd import _lib
    lib = _lib.load()
    model, log = headline_log
    recs = [r for r in log if "fn" not in r]
    assert len(recs) >= 160
    seen, worst32, worst_frac, worst_ulp, n_dma = set(), 0.0, 0.0, 0.0, 0
    for r in recs:
        p = r["params"]
        key = (p.B * p.Hout * p.Wout, p.N, p.Cin, p.KH, p.stride, p.ups, p.act, p.Cin2, p.tile, p.split_k, bool(p.ln_stats),
               bool(p.residual), bool(p.rowbias), bool(p.corr), bool(p.tile_counters))
        if key in seen or p.tile < 7:
            continue
        seen.add(key)
        n_dma += 1
        M = p.B * p.Hout * p.Wout
        nout = p.N // 2 if p.act == 2 else p.N
        what = f"M{M} N{p.N} Cin{p.Cin} taps{p.KH * p.KW} s{p.stride} u{p.ups} act{p.act} x2 {p.Cin2} tile {p.tile} split {p.split_k}"
        # the register-staged launch: same operands, whole K in one slice, no statistics side outputs
        def staged(q):
            q.tile, q.split_k, q.order = 6, 1, 0
            q.tile_counters = q.workspace = q.prefetch = None
            q.prefetch_bytes = 0
            return q
        outs = {}
        for form in ("f32", "bf16"):
            for side in ("dma", "staged"):
                q = _clone(p)
                q.rowstat_out = q.colstat_out = q.prefetch = q.ustat_out = None
                q.rowstat_slots = q.colstat_ld = 0
                q.prefetch_bytes = 0
                if side == "staged":
                    staged(q)
                y = torch.empty(M, nout, dtype=torch.float32 if form == "f32" else torch.bfloat16, device=cuda)
                q.y, q.ldy, q.out_f32 = y.data_ptr(), nout, int(form == "f32")
                _launch(lib, q, what + f" [{form}/{side}]")
                outs[form, side] = y
        torch.cuda.synchronize()
        a, b = outs["f32", "dma"].double(), outs["f32", "staged"].double()
        rms = float(b.pow(2).mean().sqrt())
        e32 = float((a - b).abs().max()) / rms
        # max-norm over up to 10 M elements (the L2 figure is recorded below); fp32 re-association error grows with the depth of the
        # contraction (K = 23,040 under an 8-way split sums in another order than the single slice): sqrt(K) scaling
        Ktot = p.KH * p.KW * p.Cin + p.Cin2
        assert e32 <= 2e-6 * 8 * max(1.0, (Ktot / 2048.0) ** 0.5), (what, e32)
        l2 = float((a - b).norm() / b.norm())
        worst32 = max(worst32, l2)
        assert l2 <= 2e-6, (what, l2)
        c, d = outs["bf16", "dma"].float(), outs["bf16", "staged"].float()
        diff = (c - d).abs()
        # one bf16 ulp of the larger value, plus the fp32 re-association slack for outputs that cancel to (almost) nothing
        ulp = torch.maximum(c.abs(), d.abs()) * 2.0 ** -7 + 8e-6 * rms
        assert bool((diff <= ulp).all()), (what, float((diff / ulp).max()))
        worst_ulp = max(worst_ulp, float((diff / ulp).max()))
        frac = float((diff > 0).float().mean())
        worst_frac = max(worst_frac, frac)
    assert n_dma >= 40, n_dma
    check(worst32, 2e-6, f"DMA tile vs register-staged tile, fp32 outputs, worst of {n_dma} distinct headline launches (rel-L2)")
    check(worst_frac, 5e-3, "fraction of bf16 outputs that differ at all (each by <= 1 ulp)")


def test_fused_transformer_tail_equals_its_three_register_staged_launches(headline_log, cuda):
    """aptp_ff_tail (LN3 -> GEGLU projection -> ff.net[2] + residual -> proj_out + residual, blocks.py:799-818) on the headline
    forward's own operands against the same chain through register-staged launches with the same bf16 rounding points."""
    from diffusion_pruning_amd import ops
    model, _ = headline_log
    cfg = O.SD21
    sample, t, ehs = O.synthetic_inputs(cfg, 4, 64, seed=5)
    old_fuse = ops.FUSE_TAIL
    ops.FUSE_TAIL = True                    # (off by default since round 4: the kernel stays a tested option)
    model.invalidate_plans()
    try:
        with torch.no_grad():
            model(sample.to(cuda), t.to(cuda), ehs.to(cuda))
            ops.LAUNCH_LOG = []
            model(sample.to(cuda), t.to(cuda), ehs.to(cuda))
            torch.cuda.synchronize()
            log = ops.LAUNCH_LOG
    finally:
        ops.LAUNCH_LOG = None
        ops.FUSE_TAIL = old_fuse
        model.invalidate_plans()
    tails = [r for r in log if r.get("fn") == "aptp_ff_tail"]
    assert len(tails) == 5
    worst = 0.0
    for r in tails[:2] + tails[-1:]:
        h, x, out, pw1, pw2, pw3, _ = r["keep"]
        y = ops.ff_tail(h, x, pw1, pw2, pw3, 1e-5)
        # the un-fused chain on register-staged tiles; the folded LayerNorm of the projection takes row statistics computed here
        B, L, C = h.shape
        hs = h.float()
        st = torch.zeros(1, B * L, 4, dtype=torch.float32, device=cuda)
        st[0, :, 0] = hs.sum(-1).reshape(-1)
        st[0, :, 1] = (hs * hs).sum(-1).reshape(-1)
        ff = ops.linear(h, pw1, ln=(st, 1e-5), tile=3)
        h3 = ops.linear(ff, pw2, residual=h, tile=6)
        yu = ops.linear(h3, pw3, residual=x, tile=6)
        torch.cuda.synchronize()
        c, d = y.float(), yu.float()
        e = float((c - d).norm() / d.norm())
        worst = max(worst, e)
        # (three chained contractions with two bf16 rounding points in between: an intermediate whose rounding flips moves the
        # outputs it feeds by a fraction of an ulp, so the two forms agree to a bf16 rounding of the output, not bit for bit)
    check(worst, 2e-3, "fused transformer tail vs three register-staged launches (rel-L2, bf16 outputs)")
