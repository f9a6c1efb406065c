"""GPU parity of the differentiable (pruning-step) path: gradients of a scalar loss w.r.t. every width / depth gate,
HIP forward + HIP backward, against PyTorch autograd through the fp32 CPU oracle on the same weights and inputs.

Tolerance: activations AND gradients travel in bf16 through ~60 layers; the gate-gradient vector (1620-dim per sample
at SD-2.1, smaller here) must agree to rel-L2 <= 6e-2 overall, and the forward output to 2e-2."""
import pytest
import torch

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
from tests.margins import check  # noqa: E402


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.fixture(scope="module")
def tiny(cuda):
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    cfg = O.TINY
    model = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                      cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(cuda)
    model.freeze()
    return cfg, model, params


def soft_gates(cfg, batch, seed, device=None):
    g = torch.Generator().manual_seed(seed)
    st = O.get_structure(cfg)
    width = [(torch.rand(batch, w, generator=g) * 0.8 + 0.2) for sub in st["width"] for w in sub]
    depth = [(torch.rand(batch, generator=g) * 0.8 + 0.2) for sub in st["depth"] for d in sub if d == 1]
    if device is not None:
        width, depth = [w.to(device) for w in width], [d.to(device) for d in depth]
    for t in width + depth:
        t.requires_grad_()
    return width, depth


@pytest.mark.parametrize("B,Bg", [(2, 2), (4, 2)])
def test_gate_gradients_match_oracle_autograd(tiny, cuda, B, Bg):
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, B, 16, seed=21)
    R = torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    # oracle
    w_ref, d_ref = soft_gates(cfg, Bg, 77)
    out_ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, {"width": list(w_ref), "depth": list(d_ref)}), "gated")
    (out_ref * R).sum().backward()
    # HIP path
    w_dev, d_dev = soft_gates(cfg, Bg, 77, cuda)
    model.set_structure({"width": list(w_dev), "depth": list(d_dev)})
    out = model(sample.to(cuda), t.to(cuda), ehs.to(cuda)).sample
    assert out.requires_grad
    check(rel_l2(out.detach().float().cpu(), out_ref.detach()), 2e-2, "forward")
    (out.float() * R.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    got = torch.cat([g.grad.float().cpu().flatten() for g in w_dev + d_dev])
    ref = torch.cat([g.grad.flatten() for g in w_ref + d_ref])
    assert torch.isfinite(got).all()
    e = rel_l2(got, ref)
    per = [rel_l2(a.grad.float().cpu(), b.grad) for a, b in zip(w_dev + d_dev, w_ref + d_ref)]
    check(e, 6e-2, "gate gradients")
    assert sorted(per)[len(per) // 2] <= 8e-2, sorted(per)[-5:]


def test_block_activation_hooks_carry_gradients(tiny, cuda):
    """block-distillation loss (trainer.py:1220-1225): MSE on hooked block outputs must back-propagate into the gates"""
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16, seed=3)
    w_dev, d_dev = soft_gates(cfg, 2, 9, cuda)
    acts = {}
    hooks = [model.mid_block.register_forward_hook(lambda m, i, o: acts.__setitem__("m", o)),
             model.down_blocks[1].register_forward_hook(lambda m, i, o: acts.__setitem__("d1", o[0]))]
    model.set_structure({"width": list(w_dev), "depth": list(d_dev)})
    model(sample.to(cuda), t.to(cuda), ehs.to(cuda))
    for h in hooks:
        h.remove()
    loss = acts["m"].float().pow(2).mean() + acts["d1"].float().pow(2).mean()
    loss.backward()
    # gates of down_blocks.0/1 feed both activations; up-block gates feed none of them
    assert float(w_dev[0].grad.abs().sum()) > 0 and float(d_dev[0].grad.abs().sum()) > 0
    assert w_dev[-1].grad is None or float(w_dev[-1].grad.abs().sum()) == 0.0
