"""The fp32 PARITY path on the GPU (SURVEY section 8 preamble: "fp32 path kept for parity"; the reference computes in fp32,
configs/pruning/sd-2-1_cc3m.yaml:79).

``aptp_conv_gemm(io_f32)`` is an fp32 instantiation of the register-staged bf16 kernel -- the same gather, tap walk, zero
padding, K-slice bounds, split-K (both forms) and epilogue code on exact-fp32 MFMAs -- and ``aptp_groupnorm / aptp_layernorm /
aptp_attention(io_f32)`` are fp32 kernels with the product's statistics layout and formulas.  With ``ops.ACT_DTYPE = float32``
the unchanged model code runs the whole SD-2.1 U-Net through them.  Tolerances: rel-L2 <= 1e-5 per op against an fp64
PyTorch reference of the same op, <= 1e-4 for the whole U-Net against the fp32 CPU oracle.  Two orders below what bf16 storage
allows (3e-3 per block): an addressing, border-class or epilogue-order defect cannot hide here.  Never used by bench.py."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
from tests.margins import check  # noqa: E402

OP_TOL = 1e-5
NET_TOL = 1e-4


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.fixture()
def ops32(cuda, monkeypatch):
    from diffusion_pruning_amd import ops
    ops._lib.load()
    monkeypatch.setattr(ops, "ACT_DTYPE", torch.float32)
    return ops


@pytest.fixture(params=[True, False], ids=["splitk-in-kernel", "splitk-reduce-launch"])
def both_splitk(ops32, request):
    ops32.SPLITK_IN_KERNEL = request.param
    yield request.param
    ops32.SPLITK_IN_KERNEL = True


CASES = [
    # B, H, W, Cin, Cout, k, stride, ups, tile, split_k
    (2, 16, 16, 64, 64, 3, 1, 0, 0, 1),
    (1, 8, 8, 320, 320, 3, 1, 0, 0, None),
    (2, 32, 32, 160, 320, 3, 1, 0, 2, 1),      # Cin not a multiple of the K-step, 128x160 tile
    (2, 32, 32, 160, 320, 3, 1, 0, 4, 1),
    (2, 16, 16, 128, 128, 3, 2, 0, 0, 1),      # stride-2 downsample
    (2, 8, 8, 128, 128, 3, 1, 1, 0, 1),        # nearest-x2 upsample folded into the gather
    (2, 16, 16, 192, 128, 1, 1, 0, 0, 1),      # 1x1
    (3, 7, 5, 72, 40, 3, 1, 0, 0, 1),          # ragged M / N / Cin tails
    (3, 7, 5, 72, 40, 3, 1, 0, 6, 3),
    (1, 8, 8, 1280, 1280, 3, 1, 0, 3, 4),      # split-K, K = 11520
    (1, 8, 8, 2560, 1280, 3, 1, 0, 0, None),   # auto split-K, K = 23040
    (4, 8, 8, 64, 64, 3, 1, 0, 1, 1),
    (4, 8, 8, 64, 64, 3, 1, 0, 5, 2),
    (2, 64, 64, 320, 160, 3, 1, 0, 0, 1),      # level-64 shape
]


@pytest.mark.parametrize("case", CASES)
def test_conv_fp32(ops32, cuda, case, both_splitk):
    B, H, W, Cin, Cout, k, stride, ups, tile, split_k = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g) * 0.1
    pw = ops32.pack_weight(w, b, device=cuda)
    assert pw.w.dtype == torch.float32
    y = ops32.conv_gemm(nhwc(x).to(cuda), pw, stride=stride, ups=ups, tile=tile, split_k=split_k)
    assert y.dtype == torch.float32
    xr = x.double()
    if ups:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xr, w.double(), b.double(), stride=stride, padding=k // 2)
    check(rel_l2(y.cpu().permute(0, 3, 1, 2), ref), OP_TOL, "fp32 conv %s" % (case,))


@pytest.mark.parametrize("split_k,tile", [(1, 0), (3, 0), (2, 6)])
def test_conv_fp32_full_epilogue_and_second_operand(ops32, cuda, split_k, tile, both_splitk):
    """bias + temb rowbias + per-sample width gate; corr + residual + depth lerp; conv3x3(x) + conv1x1(x2) as one contraction"""
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout, G = 4, 8, 8, 64, 64, 32
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) * 0.1
    pw = ops32.pack_weight(w, b, device=cuda)
    temb = torch.randn(B, Cout, generator=g) * 0.5
    gate = torch.rand(2, G, generator=g)
    y = ops32.conv_gemm(nhwc(x).to(cuda), pw, rowbias=temb.to(cuda), colgate=gate.to(cuda).contiguous(), gate_group=Cout // G,
                        split_k=split_k, tile=tile, act=ops32.ACT_SILU)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + temb.double()[:, :, None, None]
    ref = F.silu(ref * gate.double().repeat_interleave(Cout // G, dim=1).repeat(2, 1)[:, :, None, None])
    check(rel_l2(y.cpu().permute(0, 3, 1, 2), ref), OP_TOL, "fp32 epilogue: bias, rowbias, gate, SiLU")

    res, din = torch.randn(B, Cout, H, W, generator=g), torch.randn(B, Cout, H, W, generator=g)
    d = torch.rand(2, generator=g)
    corr = torch.randn(1, 9, Cout, generator=g) * 0.3
    y = ops32.conv_gemm(nhwc(x).to(cuda), pw, corr=corr.to(cuda).contiguous(), residual=nhwc(res).to(cuda), depth=d.to(cuda),
                        depth_in=nhwc(din).to(cuda), split_k=split_k, tile=tile)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    cls = torch.ones(H, dtype=torch.long); cls[0] = 0; cls[-1] = 2
    ref = ref + corr.double()[0][cls[:, None] * 3 + cls[None, :]].permute(2, 0, 1)[None] + res.double()
    dm = d.double().repeat(2)[:, None, None, None]
    ref = (1 - dm) * din.double() + dm * ref
    check(rel_l2(y.cpu().permute(0, 3, 1, 2), ref), OP_TOL, "fp32 epilogue: corr, residual, depth lerp")

    Cin2 = 200
    x2 = torch.randn(B, Cin2, H, W, generator=g)
    w2 = torch.randn(Cout, Cin2, 1, 1, generator=g) / math.sqrt(Cin2)
    b2 = torch.randn(Cout, generator=g) * 0.1
    pwc = ops32.pack_weight_cat(ops32.pack_weight(w, b, device=cuda), w2, b2)
    y = ops32.conv_gemm(nhwc(x).to(cuda), pwc, x2=nhwc(x2).to(cuda), split_k=split_k, tile=tile)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + F.conv2d(x2.double(), w2.double(), b2.double())
    check(rel_l2(y.cpu().permute(0, 3, 1, 2), ref), OP_TOL, "fp32 second-operand K-segment")


@pytest.mark.parametrize("split_k", [1, 2])
def test_linear_fp32_geglu_rowstats_and_folded_layernorm(ops32, cuda, split_k, both_splitk):
    g = torch.Generator().manual_seed(13)
    B, L, C, inner = 2, 96, 128, 256
    src = torch.randn(B, L, 64, generator=g)
    wp, bp = torch.randn(C, 64, generator=g) * 0.19, torch.full((C,), 0.4)
    h, st = ops32.linear(src.to(cuda), ops32.pack_weight(wp, bp, device=cuda), rowstats=True, split_k=1)
    href = F.linear(src.double(), wp.double(), bp.double())
    check(rel_l2(h.cpu(), href), OP_TOL, "fp32 linear (producer)")
    assert st is not None
    tot = st.sum(0).cpu().double()
    assert torch.allclose(tot[:, 0] + tot[:, 2], h.cpu().double().reshape(-1, C).sum(1), rtol=1e-5, atol=1e-4)
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
    w, b = torch.randn(2 * inner, C, generator=g) * 0.1, torch.randn(2 * inner, generator=g) * 0.1
    gate = (torch.rand(B, 32, generator=g) > 0.3).float()
    pw = ops32.pack_weight(w, b, geglu=True, device=cuda, ln_gamma=gamma, ln_beta=beta)
    y = ops32.linear(h, pw, ln=(st, 1e-5), colgate=gate.to(cuda).contiguous(), gate_group=inner // 32, split_k=split_k)
    n = F.layer_norm(href, (C,), gamma.double(), beta.double(), 1e-5)
    hh, gg = F.linear(n, w.double(), b.double()).chunk(2, dim=-1)
    m = gate.double().repeat_interleave(inner // 32, dim=1)[:, None, :]
    ref = (hh * m) * F.gelu(gg * m)
    check(rel_l2(y.cpu(), ref), 3e-5, "fp32 folded LayerNorm + GEGLU")      # (the kernels' GELU uses a 6e-7-accurate normal CDF)


@pytest.mark.parametrize("B,H,C,groups,silu", [(2, 32, 320, 32, True), (2, 8, 1280, 32, True), (1, 16, 170, 17, False), (3, 7, 40, 4, True)])
def test_groupnorm_fp32(ops32, cuda, B, H, C, groups, silu):
    g = torch.Generator().manual_seed(C)
    Cp = (C + 7) // 8 * 8
    x = torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3
    xp = torch.zeros(B, H, H, Cp)
    xp[..., :C] = nhwc(x)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    y = ops32.groupnorm(xp.to(cuda), gamma.to(cuda), beta.to(cuda), groups, 1e-5, silu, C=C)
    ref = F.group_norm(x.double(), groups, gamma.double(), beta.double(), 1e-5)
    ref = F.silu(ref) if silu else ref
    assert y.dtype == torch.float32 and float(y[..., C:].abs().max() if Cp > C else 0.0) == 0.0
    check(rel_l2(y[..., :C].cpu().permute(0, 3, 1, 2), ref), OP_TOL, "fp32 GroupNorm C=%d" % C)


def test_layernorm_and_attention_fp32(ops32, cuda):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 300, 640, generator=g) * 1.5 + 0.4
    gamma, beta = 1.0 + 0.2 * torch.randn(640, generator=g), 0.3 * torch.randn(640, generator=g)
    y = ops32.layernorm(x.to(cuda), gamma.to(cuda), beta.to(cuda), 1e-5)
    check(rel_l2(y.cpu(), F.layer_norm(x.double(), (640,), gamma.double(), beta.double(), 1e-5)), OP_TOL, "fp32 LayerNorm")
    for (B, h, Lq, Lk) in ((2, 3, 256, 77), (1, 2, 200, 192), (1, 5, 1024, 1024)):
        q, k, v = (torch.randn(B, L, h * 64, generator=g) for L in (Lq, Lk, Lk))
        o = ops32.attention(q.to(cuda), k.to(cuda), v.to(cuda), h)
        qd, kd, vd = (t.double().view(B, -1, h, 64).transpose(1, 2) for t in (q, k, v))
        ref = F.scaled_dot_product_attention(qd, kd, vd).transpose(1, 2).reshape(B, Lq, h * 64)
        check(rel_l2(o.cpu(), ref), OP_TOL, "fp32 attention %s" % ((B, h, Lq, Lk),))


def _clone_mask(mask):
    return {k: [v.clone() for v in vs] for k, vs in mask.items()}


@pytest.mark.parametrize("case", ["dense", "half_gated", "random_depth", "soft_per_sample"])
def test_whole_sd21_unet_fp32(ops32, cuda, case):
    """the WHOLE SD-2.1 U-Net in fp32 on the GPU (unchanged model code; every contraction through the fp32 instantiation of the
    bf16 kernel's code) against the fp32 oracle: dense, fixed 50 % mask (compacted weights + GroupNorm-beta border correction),
    random hard mask with depth gates off, soft per-sample masks with CFG batch doubling"""
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    cfg = O.SD21
    torch.set_num_threads(min(32, torch.get_num_threads()))
    model = UNet2DConditionModelGated().init_synthetic(seed=0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    if case == "dense":
        B, mask = 1, O.ones_mask(cfg)
    elif case == "half_gated":
        B, mask = 1, O.fixed_half_mask(cfg)
    elif case == "random_depth":
        B, mask = 1, O.random_mask(cfg, 0.4, 1, n_depth_off=3)
    else:
        g = torch.Generator().manual_seed(4)
        st = O.get_structure(cfg)
        B = 2
        mask = {"width": [torch.rand(1, w, generator=g) * 0.9 + 0.1 for sub in st["width"] for w in sub],
                "depth": [torch.rand(1, generator=g) for sub in st["depth"] for d in sub if d == 1]}
    sample, t, ehs = O.synthetic_inputs(cfg, B, 64, seed=3)
    with torch.no_grad():
        ref, blocks = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, _clone_mask(mask)), "gated", return_blocks=True)
    model.to(cuda)
    model.set_structure({k: [v.to(cuda) for v in vs] for k, vs in _clone_mask(mask).items()})
    seen = {}
    hooks = [model.mid_block.register_forward_hook(lambda m, i, o: seen.__setitem__("mid", o)),
             model.down_blocks[0].register_forward_hook(lambda m, i, o: seen.__setitem__("down0", o[0]))]
    with torch.no_grad():
        out = model(sample.to(cuda), t.to(cuda), ehs.to(cuda)).sample
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    assert out.dtype == torch.float32 and seen["mid"].dtype == torch.float32
    check(rel_l2(seen["down0"].float().cpu(), blocks[0]), NET_TOL, "fp32 U-Net down_blocks[0] (%s)" % case)
    check(rel_l2(seen["mid"].float().cpu(), blocks[4]), NET_TOL, "fp32 U-Net mid_block (%s)" % case)
    check(rel_l2(out.cpu(), ref), NET_TOL, "fp32 U-Net output (%s)" % case)
