"""GPU parity of the fine-tuning path: parameter gradients of a pruned expert (HIP forward + HIP backward
incl. weight gradients) vs PyTorch autograd through the fp32 CPU oracle in *pruned* semantics."""
import pytest
import torch

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
from tests.margins import check  # noqa: E402


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def build(cuda, mask):
    from diffusion_pruning_amd.unet import UNet2DConditionModelPruned
    cfg = O.TINY
    pm = UNet2DConditionModelPruned(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                    cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=0)
    params = {k: v.detach().clone().requires_grad_() for k, v in pm.state_dict().items()}
    pm.to(cuda)
    pm.prune({k: [v.clone().to(cuda) for v in vs] for k, vs in mask.items()})
    return cfg, pm, params


@pytest.mark.parametrize("seed,keep,ndoff", [(4, 0.5, 1), (5, 1.0, 0)])
def test_parameter_gradients_match_oracle(cuda, seed, keep, ndoff):
    cfg = O.TINY
    mask = O.random_mask(cfg, keep, seed, n_depth_off=ndoff) if keep < 1.0 else O.ones_mask(cfg)
    cfg, pm, params = build(cuda, mask)
    B = 2
    sample, t, ehs = O.synthetic_inputs(cfg, B, 16, seed=31)
    R = torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(9))
    out_ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, {k: [v.clone() for v in vs] for k, vs in mask.items()}), "pruned")
    (out_ref * R).sum().backward()
    out = pm(sample.to(cuda), t.to(cuda), ehs.to(cuda)).sample
    check(rel_l2(out.detach().float().cpu(), out_ref.detach()), 2e-2, "forward")
    (out.float() * R.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    errs = {}
    worst = []
    got_all, ref_all = [], []
    for name, p in pm.named_parameters():
        ref = params[name].grad
        if ref is None or float(ref.abs().sum()) == 0.0:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, name     # dropped / dead module
            continue
        assert p.grad is not None, name
        g = p.grad.float().cpu()
        assert torch.isfinite(g).all(), name
        # structurally dead rows / columns of a pruned expert must receive exactly zero gradient
        if ref.dim() >= 2:
            dead_rows = ref.flatten(1).abs().sum(1) == 0
            dead_cols = ref.transpose(0, 1).flatten(1).abs().sum(1) == 0
            if bool(dead_rows.any()):
                assert float(g[dead_rows].abs().max()) == 0.0, name
            if bool(dead_cols.any()):
                assert float(g[:, dead_cols].abs().max()) == 0.0, name
        errs[name] = rel_l2(g, ref)
        got_all.append(g.flatten()); ref_all.append(ref.flatten())
    e_all = rel_l2(torch.cat(got_all), torch.cat(ref_all))
    med = sorted(errs.values())[len(errs) // 2]
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    check(e_all, 6e-2, "all parameter gradients")
    check(med, 6e-2, "median parameter")
    check(worst[0][1], 0.25, "worst parameter " + worst[0][0])


def test_finetune_step_updates_only_live_parameters(cuda):
    from diffusion_pruning_amd.train_step import FineTunerStep, synthetic_batch
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    cfg = O.TINY
    mask = O.random_mask(cfg, 0.5, 8, n_depth_off=1)
    cfg, student, params = build(cuda, mask)
    teacher = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                        cross_attention_dim=cfg.cross_attention_dim)
    teacher.load_state_dict({k: v.detach() for k, v in params.items()})
    teacher.to(cuda).freeze()
    teacher.set_structure({k: [v.to(cuda) for v in vs] for k, vs in O.ones_mask(cfg).items()})
    step = FineTunerStep(student, teacher)
    opt = torch.optim.AdamW([p for p in student.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.0)
    batch = synthetic_batch(2, 16, cuda, seed=4, cross_dim=cfg.cross_attention_dim)
    before = {n: p.detach().clone() for n, p in student.named_parameters()}
    l0 = float(step.train_step(opt, batch)["loss"].detach())
    l1 = float(step.train_step(opt, batch)["loss"].detach())
    assert l0 == l0 and l1 == l1
    w = dict(student.named_parameters())["down_blocks.0.resnets.0.conv1.weight"]
    delta = (w.detach() - before["down_blocks.0.resnets.0.conv1.weight"]).abs().sum(dim=(1, 2, 3)).cpu()
    keep = mask["width"][0][0].bool().repeat_interleave(w.shape[0] // 32)
    assert float(delta[keep].min()) > 0 and float(delta[~keep].max()) == 0.0
    assert l1 < l0 * 1.5
