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


def _two_students(cuda, mask):
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    cfg, a, params = build(cuda, mask)
    _, b, _ = build(cuda, mask)
    teacher = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                        cross_attention_dim=cfg.cross_attention_dim)
    teacher.load_state_dict({k: v.detach() for k, v in params.items()})
    teacher.to(cuda).freeze()
    teacher.set_structure({k: [v.to(cuda) for v in vs] for k, vs in O.ones_mask(cfg).items()})
    return cfg, a, b, teacher


@pytest.mark.parametrize("fuse_qkv", [False, True])
def test_packed_masters_train_like_diffusers_layout_masters(cuda, monkeypatch, fuse_qkv):
    """packed_train.PackedTrainer: the optimizer owns compact fp32 tensors in the kernels' order; one SGD step on them, written
    back with export_(), equals one SGD step on the diffusers-layout masters (same kernels produce the gradients; only the
    scatter into full-shape tensors and the re-pack disappear).  With to_q | to_k | to_v as ONE packed contraction
    (unet.FT_FUSE_QKV) the launches differ in shape, so the comparison is to accumulation-order accuracy instead of bitwise."""
    from diffusion_pruning_amd import unet as unet_mod
    monkeypatch.setattr(unet_mod, "FT_FUSE_QKV", fuse_qkv)
    from diffusion_pruning_amd.packed_train import PackedTrainer
    from diffusion_pruning_amd.train_step import FineTunerStep, synthetic_batch
    mask = O.random_mask(O.TINY, 0.5, 8, n_depth_off=1)
    cfg, sa, sb, teacher = _two_students(cuda, mask)
    batch = synthetic_batch(2, 16, cuda, seed=4, cross_dim=cfg.cross_attention_dim)
    step_a = FineTunerStep(sa, teacher)
    opt_a = torch.optim.SGD([p for p in sa.parameters() if p.requires_grad], lr=1e-2)
    la = float(step_a.train_step(opt_a, batch)["loss"].detach())
    step_a.remove_hooks() if hasattr(step_a, "remove_hooks") else None
    step_b = FineTunerStep(sb, teacher)
    pk = PackedTrainer(sb).attach().materialize(batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"])
    n_pk, n_full = pk.n_trainable(), sum(p.numel() for p in sb.parameters())
    assert n_pk < 0.8 * n_full                        # dead channels / heads / chunks / the dropped block carry no state
    opt_b = torch.optim.SGD(pk.parameters(), lr=1e-2)
    lb = float(step_b.train_step(opt_b, batch)["loss"].detach())
    assert abs(la - lb) <= (2e-3 if fuse_qkv else 1e-5) * abs(la)
    pk.export_()
    worst = 0.0
    for (na, pa), (nb, pb) in zip(sa.named_parameters(), sb.named_parameters()):
        assert na == nb
        d = float((pa.detach() - pb.detach()).abs().max())
        worst = max(worst, d / (float(pa.detach().abs().max()) + 1e-12))
    check(worst, 2e-3 if fuse_qkv else 1e-5, "max relative parameter difference after one SGD step (packed vs diffusers-layout masters%s)"
          % (", fused qkv" if fuse_qkv else ""))
    # the second forward runs on the refreshed shadows: same loss as the reference path's second step
    la2 = float(step_a.train_step(opt_a, batch)["loss"].detach())
    lb2 = float(step_b.train_step(opt_b, batch)["loss"].detach())
    assert abs(la2 - lb2) <= 2e-3 * abs(la2) and la2 != la


def test_no_grad_forward_and_state_dict_follow_the_packed_state(cuda):
    """While a PackedTrainer is attached, the packed tensors ARE the model: a no-grad forward in the middle of training (the
    reference FineTuner's validation / sample generation, trainer.py:1766-1830) and state_dict() must see the trained values,
    not inference packs built from the diffusers-layout masters, which keep their initial values until export_()."""
    from diffusion_pruning_amd.packed_train import PackedTrainer
    from diffusion_pruning_amd.train_step import FineTunerStep, synthetic_batch
    mask = O.random_mask(O.TINY, 0.5, 8, n_depth_off=1)
    cfg, sa, sb, teacher = _two_students(cuda, mask)
    batch = synthetic_batch(2, 16, cuda, seed=4, cross_dim=cfg.cross_attention_dim)
    args = (batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"])
    with torch.no_grad():
        y_init = sb(*args).sample.float().clone()
    step = FineTunerStep(sb, teacher)
    pk = PackedTrainer(sb).attach().materialize(*args)
    opt = torch.optim.SGD(pk.parameters(), lr=5e-2)
    for _ in range(2):
        step.train_step(opt, batch)
    y_grad = sb(*args).sample.detach().float()                     # the training forward: reads the packed state
    with torch.no_grad():
        y_eval = sb(*args).sample.float()
    assert rel_l2(y_eval, y_grad) <= 1e-6                          # same packs, same kernels
    assert rel_l2(y_eval, y_init) > 1e-3                           # and they did move
    # the checkpoint interface exports first
    w0 = dict(sa.named_parameters())["down_blocks.0.resnets.0.conv1.weight"].detach()
    sd = sb.state_dict()
    assert float((sd["down_blocks.0.resnets.0.conv1.weight"].to(cuda) - w0).abs().max()) > 0
    # after detach() the inference path runs on packs rebuilt from the exported masters: same function again
    pk.detach()
    sb.invalidate_plans()
    with torch.no_grad():
        y_inf = sb(*args).sample.float()
    check(rel_l2(y_inf, y_eval), 2e-2, "inference forward on exported masters vs no-grad forward on the packed state")


def test_graphed_finetune_step_equals_the_eager_packed_step(cuda, monkeypatch):
    """GraphedFineTunerStep (teacher forward; student forward + losses + backward + fused AdamW + operand refresh as HIP
    graphs) reproduces the eager packed step: losses of three consecutive steps and the parameters after them"""
    from diffusion_pruning_amd.packed_train import PackedTrainer
    from diffusion_pruning_amd.train_step import FineTunerStep, GraphedFineTunerStep, synthetic_batch
    mask = O.random_mask(O.TINY, 0.6, 9, n_depth_off=1)
    cfg, sa, sb, teacher = _two_students(cuda, mask)
    batches = [synthetic_batch(2, 16, cuda, seed=s, cross_dim=cfg.cross_attention_dim) for s in (4, 5, 6)]
    kw = dict(lr=1e-4, weight_decay=1e-2)
    from diffusion_pruning_amd.packed_train import PackedAdamW
    eager = FineTunerStep(sa, teacher)
    pk = PackedTrainer(sa).attach().materialize(batches[0]["noisy_latents"], batches[0]["timesteps"], batches[0]["encoder_hidden_states"])
    # the eager reference runs the same one-launch AdamW (its table re-bound to each step's freshly allocated gradients): bf16
    # activations make training chaotic at the last-ulp level, so only identical optimizer arithmetic can be compared over
    # several steps; the arithmetic itself is checked against torch.optim.AdamW in the first step below and in
    # test_one_launch_adamw_matches_torch_adamw
    ref, opt = [], None
    p0 = [p.detach().clone() for p in pk.parameters()]
    # (the graphed step runs its weight gradients as one batched launch with the batch's pixel split: the eager step takes the
    #  same split here, so the two stay bitwise comparable)
    from diffusion_pruning_amd import ops as _ops
    monkeypatch.setattr(_ops, "WGRAD_SPLIT_RULE", "batch")
    for i, b in enumerate(batches):
        for p in pk.parameters():
            p.grad = None
        out = eager.step(b["noisy_latents"], b["timesteps"], b["encoder_hidden_states"], b["target"])
        out["loss"].backward()
        if i == 0:                       # what torch.optim.AdamW would make of this first step
            tp = [torch.nn.Parameter(p.detach().clone()) for p in pk.parameters()]
            for t, p in zip(tp, pk.parameters()):
                t.grad = None if p.grad is None else p.grad.clone()
            torch.optim.AdamW(tp, **kw).step()
        opt = PackedAdamW(pk, **kw) if opt is None else opt
        opt.rebind_().step()
        if i == 0:
            a = torch.cat([p.detach().flatten() for p in pk.parameters()])
            t = torch.cat([p.detach().flatten() for p in tp])
            check(rel_l2((a - torch.cat([p.flatten() for p in p0])).cpu(), (t - torch.cat([p.flatten() for p in p0])).cpu()), 3e-5,
                  "first AdamW update: one-launch kernel vs torch.optim.AdamW")   # the update is ~1e-4 of an fp32 parameter: its own rounding is ~1e-5
        ref.append(float(out["loss"].detach()))
    graphed = GraphedFineTunerStep(sb, teacher, **kw)
    graphed.capture(batches[0], offload_masters=True)     # the diffusers-layout masters wait on the host meanwhile
    got = []
    for b in batches:
        got.append(float(graphed.train_step(None, b)["loss"]))
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert abs(a - b) <= 2e-3 * abs(a), (ref, got)
    pa = torch.cat([p.detach().flatten() for p in pk.parameters()])
    pb = torch.cat([p.detach().flatten() for p in graphed.trainer.parameters()])
    assert pa.shape == pb.shape
    check(rel_l2(pb.cpu(), pa.cpu()), 1e-4, "parameters after three AdamW steps (graphed vs eager)")
    # checkpoint interface: export_() writes the trained values back into the diffusers-named masters (on the host here)
    before = sb.down_blocks[0].resnets[0].conv1.weight.detach().clone()
    graphed.trainer.export_()
    after = sb.down_blocks[0].resnets[0].conv1.weight.detach()
    assert after.device.type == "cpu" and float((after - before).abs().max()) > 0
    pk.export_()
    assert torch.equal(sa.down_blocks[0].resnets[0].conv1.weight.detach().cpu(), after)


def test_one_launch_adamw_matches_torch_adamw(cuda):
    """packed_train.PackedAdamW (csrc/optim.hip: every tensor in one launch, bf16 shadows written in the same pass) vs
    torch.optim.AdamW on the same tensors for four steps: parameters to 1e-6, shadows = bf16(parameters) exactly; odd sizes
    exercise the scalar tail."""
    from types import SimpleNamespace
    from diffusion_pruning_amd.packed_train import PackedAdamW
    g = torch.Generator().manual_seed(21)
    shapes = [(64, 9, 72), (8, 1, 8), (1283,), (5,), (4096 * 3 + 7,)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(cuda)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    shadows = [torch.zeros(s, dtype=torch.bfloat16, device=cuda) for s in shapes[:2]]
    gemms = {0: SimpleNamespace(P=ps[0], Pb=ps[2], pw=SimpleNamespace(w=shadows[0])),
             1: SimpleNamespace(P=ps[1], Pb=None, pw=SimpleNamespace(w=shadows[1]))}
    affines = {0: SimpleNamespace(Pg=ps[3], Pb=ps[4])}
    trainer = SimpleNamespace(gemms=gemms, affines=affines, refresh_=lambda shadows_done=False: None)
    order = [ps[0], ps[2], ps[1], ps[3], ps[4]]
    for p in order:
        p.grad = torch.zeros_like(p)
    opt = PackedAdamW(trainer, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    topt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    for it in range(4):
        for p, r in zip(ps, ref):
            gr = torch.randn(p.shape, generator=g).to(cuda)
            p.grad.copy_(gr)                     # in place: the addresses are part of the kernel's table
            r.grad = gr.clone()
        opt.step()
        topt.step()
        for p, r in zip(ps, ref):
            assert float((p - r).abs().max()) <= 1e-6 * float(r.abs().max()) + 1e-7, it
    assert torch.equal(shadows[0], ps[0].detach().to(torch.bfloat16)) and torch.equal(shadows[1], ps[1].detach().to(torch.bfloat16))
    assert float(opt.step_t) == 4.0
    # resume: a fresh optimizer over the same tensors, loaded with the saved state, takes the same fifth step
    sd = opt.state_dict()
    snap = [p.detach().clone() for p in ps]
    for p in order:
        p.grad.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(99)).to(cuda))
    opt.step()
    after = [p.detach().clone() for p in ps]
    with torch.no_grad():
        for p, s0 in zip(ps, snap):
            p.copy_(s0)
    opt2 = PackedAdamW(trainer, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    opt2.load_state_dict(sd)
    opt2.step()
    assert all(torch.equal(p.detach(), a) for p, a in zip(ps, after)) and float(opt2.step_t) == 5.0
    # entries are matched by NAME: a state saved in another order loads onto the right tensors; a moment of another shape
    # (either list) or a state of another tensor list is refused
    perm = [4, 2, 0, 3, 1]
    sd_perm = {"step": sd["step"], "names": [sd["names"][i] for i in perm], "exp_avg": [sd["exp_avg"][i] for i in perm],
               "exp_avg_sq": [sd["exp_avg_sq"][i] for i in perm]}
    with torch.no_grad():
        for p, s0 in zip(ps, snap):
            p.copy_(s0)
    opt3 = PackedAdamW(trainer, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    opt3.load_state_dict(sd_perm)
    opt3.step()
    assert all(torch.equal(p.detach(), a) for p, a in zip(ps, after))
    bad = dict(sd)
    bad["exp_avg_sq"] = [v.flatten()[:-1].clone() if i == 1 else v for i, v in enumerate(sd["exp_avg_sq"])]
    with pytest.raises(AssertionError):
        opt3.load_state_dict(bad)
    other = dict(sd)
    other["names"] = ["x" + n for n in sd["names"]]
    with pytest.raises(AssertionError):
        opt3.load_state_dict(other)


def test_adamw_gate_grad_scale_and_groups(cuda):
    """AptpAdamWParams.gate_dev / grad_scale and the per-group tables of PackedAdamW: a non-finite gate leaves parameters,
    moments, operands and the step count untouched (the batch skip of pdm/training/trainer.py:921-929 for a replayed graph);
    grad_scale = 1 / w on w-times-summed gradients is the step on the mean; stepping group by group equals one launch."""
    from types import SimpleNamespace
    from diffusion_pruning_amd.packed_train import PackedAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 9, 72), (8, 1, 8), (1283,), (5,), (4096 * 3 + 7,)]

    def make():
        ps = [torch.nn.Parameter(torch.randn(*s, generator=torch.Generator().manual_seed(11 + i)).to(cuda)) for i, s in enumerate(shapes)]
        sh = [torch.zeros(s, dtype=torch.bfloat16, device=cuda) for s in shapes[:2]]
        gemms = {0: SimpleNamespace(P=ps[0], Pb=ps[2], pw=SimpleNamespace(w=sh[0])), 1: SimpleNamespace(P=ps[1], Pb=None, pw=SimpleNamespace(w=sh[1]))}
        tr = SimpleNamespace(gemms=gemms, affines={0: SimpleNamespace(Pg=ps[3], Pb=ps[4])}, refresh_=lambda shadows_done=False: None)
        for p in ps:
            p.grad = torch.zeros_like(p)
        return ps, sh, tr
    grads = [torch.randn(*s, generator=g).to(cuda) for s in shapes]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    # reference: one plain step on the mean gradient
    ps0, sh0, tr0 = make()
    for p, gr in zip(ps0, grads):
        p.grad.copy_(gr)
    o0 = PackedAdamW(tr0, **kw)
    o0.step()
    # (a) gate = NaN / inf: nothing moves
    ps1, sh1, tr1 = make()
    before = [p.detach().clone() for p in ps1]
    for p, gr in zip(ps1, grads):
        p.grad.copy_(gr * 3.0)
    o1 = PackedAdamW(tr1, **kw)
    for bad in (float("nan"), float("inf")):
        o1.step(gate=torch.tensor(bad, device=cuda), grad_scale=1.0 / 3.0)
        assert all(torch.equal(p.detach(), b) for p, b in zip(ps1, before)) and float(o1.step_t) == 0.0
        assert all(float(m.abs().max()) == 0.0 for m in o1.m) and float(sh1[0].float().abs().max()) == 0.0
    # (b) finite gate, gradients summed over 3 "ranks", scale 1/3: the reference step
    o1.step(gate=torch.tensor(0.25, device=cuda), grad_scale=1.0 / 3.0)
    assert float(o1.step_t) == 1.0
    for p, r in zip(ps1, ps0):
        assert float((p - r).abs().max()) <= 2e-6 * float(r.abs().max())
    assert torch.equal(sh1[0], ps1[0].detach().to(torch.bfloat16))
    # (c) two groups, stepped one after the other: bitwise the single launch
    ps2, sh2, tr2 = make()
    for p, gr in zip(ps2, grads):
        p.grad.copy_(gr)
    ids = {id(p): i for i, p in enumerate(ps2)}
    o2 = PackedAdamW(tr2, group_of=lambda p: ids[id(p)] % 2, **kw)
    assert sorted(o2.groups) == [0, 1]
    o2.step_group(1)
    o2.step_group(0)
    o2.finish_step()
    assert float(o2.step_t) == 1.0 and all(torch.equal(p.detach(), r.detach()) for p, r in zip(ps2, ps0))
    assert torch.equal(sh2[1], sh0[1])


def test_graphed_finetune_step_skips_a_batch_with_nan_loss(cuda):
    """GraphedFineTunerStep(nan_guard=True): a batch that makes the loss NaN (here: a NaN in the target) is skipped on the
    device -- parameters, AdamW moments and step count keep their values -- and the following clean batch trains normally
    (reference: Pruner's "NaNs detected in the loss. Skipping batch.", trainer.py:921-929)."""
    from diffusion_pruning_amd.train_step import GraphedFineTunerStep, synthetic_batch
    mask = O.random_mask(O.TINY, 0.6, 9, n_depth_off=1)
    cfg, sa, sb, teacher = _two_students(cuda, mask)
    batches = [synthetic_batch(2, 16, cuda, seed=s, cross_dim=cfg.cross_attention_dim) for s in (4, 5, 6)]
    step = GraphedFineTunerStep(sb, teacher, lr=1e-3, weight_decay=1e-2)
    step.capture(batches[0])
    step.train_step(None, batches[0])
    torch.cuda.synchronize()
    snap = [p.detach().clone() for p in step.trainer.parameters()]
    m_snap = [m.clone() for m in step.optimizer.m]
    assert float(step.optimizer.step_t) == 1.0
    bad = {k: v.clone() for k, v in batches[1].items()}
    bad["target"][0, 0, 0, 0] = float("nan")
    out = step.train_step(None, bad)
    torch.cuda.synchronize()
    assert not torch.isfinite(out["loss"])
    assert float(step.optimizer.step_t) == 1.0
    assert all(torch.equal(p.detach(), s) for p, s in zip(step.trainer.parameters(), snap))
    assert all(torch.equal(m, s) for m, s in zip(step.optimizer.m, m_snap))
    out = step.train_step(None, batches[2])
    torch.cuda.synchronize()
    assert torch.isfinite(out["loss"]) and float(step.optimizer.step_t) == 2.0
    moved = sum(float((p.detach() - s).abs().max()) > 0 for p, s in zip(step.trainer.parameters(), snap))
    assert moved >= len(snap) // 2
    assert all(torch.isfinite(p).all() for p in step.trainer.parameters())


def test_graphed_finetune_data_parallel_schedule_on_one_rank(cuda):
    """GraphedFineTunerStep(data_parallel=True) with a process group of ONE rank on the GPU backend (RCCL): the gradient arena is
    exchanged in place (2 collectives per bucket and step, no per-tensor operation) and the step equals the plain graphed step
    bit for bit (sum over one rank = the gradient, grad_scale = 1)."""
    import os
    import socket
    import torch.distributed as dist
    from diffusion_pruning_amd.train_step import GraphedFineTunerStep, synthetic_batch
    mask = O.random_mask(O.TINY, 0.6, 9, n_depth_off=1)
    cfg, sa, sb, teacher = _two_students(cuda, mask)
    batches = [synthetic_batch(2, 16, cuda, seed=s, cross_dim=cfg.cross_attention_dim) for s in (4, 5, 6)]
    plain = GraphedFineTunerStep(sa, teacher, lr=1e-3, weight_decay=1e-2)
    plain.capture(batches[0])
    la = [float(plain.train_step(None, b)["loss"]) for b in batches]
    torch.cuda.synchronize()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        dp = GraphedFineTunerStep(sb, teacher, lr=1e-3, weight_decay=1e-2, data_parallel=True, bucket_bytes=1 << 20)
        dp.capture(batches[0])
        lb = [float(dp.train_step(None, b)["loss"]) for b in batches]
        torch.cuda.synchronize()
        red = dp.reducer
        assert len(red.buckets) >= 2 and red.stats["steps"] == 3
        assert red.stats["collectives"] == 2 * len(red.buckets) * 3 and red.stats["tensor_ops"] == 0
        # every packed gradient is a view into the arena
        a0, a1 = dp.trainer.grad_arena.data_ptr(), dp.trainer.grad_arena.data_ptr() + dp.trainer.grad_arena.numel() * 4
        assert all(a0 <= p.grad.data_ptr() < a1 for p in dp.trainer.parameters())
    finally:
        if created:
            dist.destroy_process_group()
    assert la == lb, (la, lb)
    for pa, pb in zip(plain.trainer.parameters(), dp.trainer.parameters()):
        assert torch.equal(pa.detach(), pb.detach())
