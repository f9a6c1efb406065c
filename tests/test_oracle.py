"""Pins the CPU oracle (oracle/unet_oracle.py) with what is available without diffusers: exact SD-2.1
parameter / MAC anchors and the invariants of SURVEY App. B.6."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as O


def test_sd21_anchors():
    shapes = O.param_shapes(O.SD21)
    assert sum(math.prod(s) for s in shapes.values()) == 865_910_724
    st = O.get_structure(O.SD21)
    assert len(st["width"]) == 38 and sum(len(s) for s in st["width"]) == 70
    assert sum(w for s in st["width"] for w in s) == 1606 and sum(d for s in st["depth"] for d in s) == 14
    assert abs(O.count_macs(O.SD21, 64) / 1e9 - 402.13) < 0.01
    assert abs(O.count_macs(O.SD21, 64, masked=True) / 1e9 - 223.67) < 0.01
    assert abs(O.count_macs(O.SD21, 32) / 1e9 - 90.55) < 0.01
    assert abs(O.count_macs(O.SD21, 96) / 1e9 - 1074.55) < 0.01
    # arch-vector layout of SURVEY App. A.2 (per container: resnets first, then attentions)
    assert st["width"][:4] == [[32], [32], [5, 5, 32], [5, 5, 32]] and st["depth"][:4] == [[0], [1], [0], [1]]
    arch = torch.arange(1620.0)[None]
    sep = O.split_arch_vector(O.SD21, arch)
    assert len(sep["width"]) == 70 and len(sep["depth"]) == 14 and float(sep["depth"][0]) == 1606.0
    gates = O.assign_gates(O.SD21, sep)
    assert float(gates["down_blocks.0.resnets.1.depth_gate"]) == 1606.0
    assert gates["down_blocks.0.attentions.0.attn1.gate"].shape == (1, 5)


@pytest.fixture(scope="module")
def tiny():
    cfg = O.TINY
    p = O.init_params(cfg, seed=3)
    s, t, e = O.synthetic_inputs(cfg, 2, 16)
    return cfg, p, s, t, e


def test_ones_mask_equals_ungated(tiny):
    """x*1 and (1-1)*x_in + 1*out are exact; what remains is torch picking different CPU kernels for differently
    strided (gated vs ungated) operands, i.e. fp32 reassociation noise."""
    cfg, p, s, t, e = tiny
    y0 = O.unet_forward(p, cfg, s, t, e)
    y1 = O.unet_forward(p, cfg, s, t, e, O.assign_gates(cfg, O.ones_mask(cfg)), "gated")
    assert float((y0 - y1).abs().max()) < 1e-5


def test_attention_and_ff_gated_equals_pruned(tiny):
    cfg, p, s, t, e = tiny
    spec = O.build_specs(cfg)[1].attns[0]                      # a 2-head transformer
    x = torch.randn(2, spec.ch, 8, 8)
    gates = {spec.name + ".attn1.gate": torch.tensor([[1.0, 0.0]]), spec.name + ".attn2.gate": torch.tensor([[0.0, 1.0]]),
             spec.name + ".ff.gate": (torch.arange(32) % 3 != 0).float()[None]}
    yg = O.transformer_forward(p, spec, cfg, x, e, gates, "gated")
    yp = O.transformer_forward(p, spec, cfg, x, e, gates, "pruned")
    assert torch.allclose(yg, yp, atol=2e-5, rtol=1e-5)


def test_resnet_gated_minus_pruned_is_the_beta_plane(tiny):
    """App. B.1: gated - pruned == conv2_zero-pad(constant plane SiLU(beta_dead), W2[:, dead])"""
    cfg, p, s, t, e = tiny
    p = {k: v.double() for k, v in p.items()}
    r = O.build_specs(cfg)[0].resnets[0]
    x = torch.randn(2, r.cin, 8, 8, dtype=torch.float64)
    temb = torch.randn(2, cfg.temb_dim, dtype=torch.float64)
    mask = (torch.arange(32) % 2 == 0).double()[None]
    g = {r.name + ".gate": mask}
    yg = O.resnet_forward(p, r, cfg, x, temb, g, "gated")
    yp = O.resnet_forward(p, r, cfg, x, temb, g, "pruned")
    dead = ~mask[0].bool().repeat_interleave(r.cout // 32)
    plane = F.silu(p[r.name + ".norm2.bias"][dead])[None, :, None, None].expand(2, -1, 8, 8)
    expect = F.conv2d(plane, p[r.name + ".conv2.weight"][:, dead], None, padding=1)
    assert float((yg - yp - expect).abs().max()) < 1e-12
    # with beta == 0 the two semantics coincide
    p0 = dict(p); p0[r.name + ".norm2.bias"] = torch.zeros_like(p[r.name + ".norm2.bias"])
    assert float((O.resnet_forward(p0, r, cfg, x, temb, g, "gated") - O.resnet_forward(p0, r, cfg, x, temb, g, "pruned")).abs().max()) < 1e-12


def test_depth_gate_skip_keep_and_up_block_slicing(tiny):
    cfg, p, s, t, e = tiny
    r = O.build_specs(cfg)[5].resnets[-1]                     # up_blocks.0 last resnet: depth gated, concatenated input
    assert r.depth_gated and r.skip_dim > 0
    x = torch.randn(2, r.cin, 4, 4)
    temb = torch.randn(2, cfg.temb_dim)
    on = O.resnet_forward(p, r, cfg, x, temb, {r.name + ".depth_gate": torch.ones(1)}, "gated")
    off = O.resnet_forward(p, r, cfg, x, temb, {r.name + ".depth_gate": torch.zeros(1)}, "gated")
    assert torch.equal(off, x[:, :r.cin - r.skip_dim])
    assert torch.equal(O.resnet_forward(p, r, cfg, x, temb, {r.name + ".depth_gate": torch.zeros(1)}, "pruned"), off)
    half = O.resnet_forward(p, r, cfg, x, temb, {r.name + ".depth_gate": torch.full((1,), 0.5)}, "gated")
    assert torch.allclose(half, 0.5 * on + 0.5 * off, atol=1e-6)


def test_cfg_tiling_and_per_sample_independence(tiny):
    cfg, p, s, t, e = tiny
    m = O.random_mask(cfg, 0.5, 3, n_depth_off=1, batch=2)
    s4, t4, e4 = torch.cat([s, s]), torch.cat([t, t]), torch.cat([e, e])
    y4 = O.unet_forward(p, cfg, s4, t4, e4, O.assign_gates(cfg, m), "gated")       # gate batch 2 tiled over batch 4
    y2 = O.unet_forward(p, cfg, s, t, e, O.assign_gates(cfg, m), "gated")
    assert torch.allclose(y4[:2], y2, atol=1e-5) and torch.allclose(y4[2:], y2, atol=1e-5)
    # row 0 of a per-sample mask gives the same result as running sample 0 alone with that row
    m0 = {k: [v[:1] for v in vs] for k, vs in m.items()}
    y0 = O.unet_forward(p, cfg, s[:1], t[:1], e[:1], O.assign_gates(cfg, m0), "gated")
    assert torch.allclose(y2[:1], y0, atol=1e-5)


def test_fp64_fp32_self_consistency(tiny):
    cfg, p, s, t, e = tiny
    y32 = O.unet_forward(p, cfg, s, t, e)
    y64 = O.unet_forward({k: v.double() for k, v in p.items()}, cfg, s.double(), t, e.double())
    rel = float((y32.double() - y64).norm() / y64.norm())
    assert rel < 1e-5, rel
