"""CPU tests of the product's host logic (no GPU): the model code runs unchanged, with the HIP ops replaced by the
test-only PyTorch emulator (tests/hip_emulator.py), and must agree with the oracle."""
import pytest
import torch

from oracle import unet_oracle as O
from tests import hip_emulator

TOL = 2e-2


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def clone_mask(m):
    return {k: [v.clone() for v in vs] for k, vs in m.items()}


@pytest.fixture()
def tiny(monkeypatch):
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    hip_emulator.install(monkeypatch)
    cfg = O.TINY
    model = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                      cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return cfg, model, params


def run(model, mask, sample, t, ehs):
    model.set_structure(clone_mask(mask))
    with torch.no_grad():
        return model(sample, t, ehs).sample.float()


def test_state_dict_matches_diffusers_names_and_sd21_param_count():
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    m = UNet2DConditionModelGated()
    sd = m.state_dict()
    shapes = O.param_shapes(O.SD21)
    assert set(sd) == set(shapes)
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    assert sum(p.numel() for p in m.parameters()) == 865_910_724
    st = m.get_structure()
    assert st == O.get_structure(O.SD21)
    assert sum(len(s) for s in st["width"]) == 70 and sum(d for s in st["depth"] for d in s) == 14
    mask = O.fixed_half_mask(O.SD21)
    m.set_structure(mask)
    assert mask["width"] == [] and mask["depth"] == []      # set_structure consumes the caller's lists


def test_dense(tiny):
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16)
    ref = O.unet_forward(params, cfg, sample, t, ehs)
    assert rel_l2(run(model, O.ones_mask(cfg), sample, t, ehs), ref) <= TOL


def test_skip_concats_are_views_without_autograd(tiny):
    """the 12 torch.cat([hidden, skip]) of the up path (diffusers CrossAttnUpBlock2D.forward) are written in place by
    their producers; a hard depth gate of 0 makes the skipped block's consumer fall back to the copying concat"""
    from diffusion_pruning_amd import unet as U
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16)
    n_cat = sum(len(b.resnets) for b in model.up_blocks)
    U.CAT_STATS.update(views=0, copies=0)
    ref = O.unet_forward(params, cfg, sample, t, ehs)
    assert rel_l2(run(model, O.ones_mask(cfg), sample, t, ehs), ref) <= TOL
    assert U.CAT_STATS == {"views": n_cat, "copies": 0}
    mask = O.ones_mask(cfg)
    for d in mask["depth"]:
        d.zero_()
    U.CAT_STATS.update(views=0, copies=0)
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    assert rel_l2(run(model, mask, sample, t, ehs), ref) <= TOL
    assert U.CAT_STATS["views"] + U.CAT_STATS["copies"] == n_cat and U.CAT_STATS["copies"] > 0


def test_half_mask_matches_gated_not_pruned(tiny):
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16)
    mask = O.fixed_half_mask(cfg)
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    refp = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "pruned")
    out = run(model, mask, sample, t, ehs)
    e, ep = rel_l2(out, ref), rel_l2(out, refp)
    assert e <= TOL and ep > 2 * e, (e, ep)


@pytest.mark.parametrize("seed,keep,ndoff", [(1, 0.4, 2), (2, 0.75, 4)])
def test_random_masks_with_depth(tiny, seed, keep, ndoff):
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16, seed=seed)
    mask = O.random_mask(cfg, keep, seed, n_depth_off=ndoff)
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    assert rel_l2(run(model, mask, sample, t, ehs), ref) <= TOL


def test_soft_per_sample_cfg(tiny):
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 4, 16, seed=9)
    g = torch.Generator().manual_seed(4)
    st = O.get_structure(cfg)
    mask = {"width": [torch.rand(2, w, generator=g) * 0.9 + 0.1 for sub in st["width"] for w in sub],
            "depth": [torch.rand(2, generator=g) for sub in st["depth"] for d in sub if d == 1]}
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    assert rel_l2(run(model, mask, sample, t, ehs), ref) <= TOL


def test_hard_per_sample(tiny):
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16, seed=5)
    mask = O.random_mask(cfg, 0.5, 7, n_depth_off=1, batch=2)
    mask["depth"][0] = torch.tensor([1.0, 0.0])
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    assert rel_l2(run(model, mask, sample, t, ehs), ref) <= TOL


def test_pruned_semantics(tiny, monkeypatch):
    from diffusion_pruning_amd.unet import UNet2DConditionModelPruned
    cfg, model, params = tiny
    pm = UNet2DConditionModelPruned(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                    cross_attention_dim=cfg.cross_attention_dim)
    pm.load_state_dict(params)
    mask = O.random_mask(cfg, 0.5, 11, n_depth_off=2)
    soft = {"width": [w * 0.9 for w in mask["width"]], "depth": [d * 0.9 for d in mask["depth"]]}
    pm.prune(clone_mask(soft))
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16, seed=6)
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(soft)), "pruned")
    with torch.no_grad():
        out = pm(sample, t, ehs, return_dict=False)[0]
    assert rel_l2(out.float(), ref) <= TOL


def test_product_rejects_cpu_tensors_without_the_emulator():
    """no CPU fallback: with the real ops, a CPU forward must fail loudly"""
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    cfg = O.TINY
    model = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                      cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=0)
    sample, t, ehs = O.synthetic_inputs(cfg, 1, 16)
    with pytest.raises(Exception):
        model(sample, t, ehs)


def test_vectorised_relaxation_equals_the_per_segment_loop():
    """quantizer.gumbel_sigmoid_trick / width_depth_normalize are evaluated as whole-vector passes; the reference loops
    over the 70 width segments / 14 depth-gated blocks (quantizer.py:196-261).  Same host-RNG stream, same values, same
    gradients, including the non-zero-width rule on segments whose relaxation is dead."""
    import torch
    from diffusion_pruning_amd.estimation_utils import (gumbel_softmax_sample, hard_concrete,
                                                        importance_gumbel_softmax_sample)
    from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    from oracle import unet_oracle as O
    cfg = O.TINY
    st = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                   cross_attention_dim=cfg.cross_attention_dim).get_structure()
    q = StructureVectorQuantizer(n_e=4, structure=st, temperature=0.4, base=3, resource_aware_normalization=False)
    q.train()
    nw = sum(q.width_list)
    torch.manual_seed(0)
    z = torch.randn(3, q.vq_embed_dim)
    s0 = 0
    for j, w in enumerate(q.width_list):          # kill a few segments (row 1) so the +0.5 rule fires
        if j % 9 == 0:
            z[1, s0:s0 + w] = -40.0
        s0 += w
    z1 = z.clone().requires_grad_(True)
    z2 = z.clone().requires_grad_(True)

    torch.manual_seed(77)
    got = q.gumbel_sigmoid_trick(z1)

    torch.manual_seed(77)                         # the reference's loop, verbatim order of host draws
    zw, zd = z2[:, :nw], z2[:, nw:]
    d_sorted = importance_gumbel_softmax_sample(zd, temperature=q.temperature, offset=q.base, fixed_seed=False)
    d = torch.zeros_like(d_sorted)
    d[:, q.depth_order] = d_sorted
    parts, s0 = [], 0
    for w in q.width_list:
        parts.append(gumbel_softmax_sample(zw[:, s0:s0 + w], temperature=q.temperature, offset=q.base,
                                           force_width_non_zero=True, fixed_seed=False))
        s0 += w
    ref = torch.cat([torch.cat(parts, dim=1), d], dim=1)
    assert torch.equal(got, ref)
    assert float((hard_concrete(got)[1, :nw].reshape(1, -1) @ q._seg_maps(got.device)[0]).min()) >= 1.0   # no dead segment left
    wgt = torch.randn_like(got)
    (got * wgt).sum().backward()
    (ref * wgt).sum().backward()
    assert torch.allclose(z1.grad, z2.grad, rtol=0, atol=0)

    # width_depth_normalize against the slice-assign loop
    a = got.detach().clone().requires_grad_(True)
    b = got.detach().clone().requires_grad_(True)
    out = q.width_depth_normalize(a)
    tmp = hard_concrete(b.clone())
    for i, has_depth in enumerate(q.depth_list):
        if has_depth != 0:
            lo, hi = q.width_intervals[i]
            di = q.depth_indices[i]
            tmp[:, lo:hi] = b[:, lo:hi] * b[:, di:di + 1]
    ref2 = tmp * torch.sqrt(q.template).detach()
    assert torch.equal(out, ref2)
    (out * wgt).sum().backward()
    (ref2 * wgt).sum().backward()
    assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6)    # the depth columns sum many products: order of summation only


def test_plan_cache_pins_plans_seen_during_capture_and_drops_stale_ones(monkeypatch):
    """a captured HIP graph bakes pointers to a plan's packs, so a plan created or used during capture must
    survive any number of other masks; and packs made from older parameter values must never be served again."""
    from diffusion_pruning_amd import unet as U
    c = U._PlanCache(cap=4)
    capturing = {"on": False}
    monkeypatch.setattr(U, "_capturing", lambda: capturing["on"])
    c.put("m0", (0,), "plan0")
    capturing["on"] = True
    assert c.get("m0", (0,)) == "plan0"             # used while capturing -> pinned
    c.put("mcap", (0,), "plancap")                  # created while capturing -> pinned
    capturing["on"] = False
    for i in range(1, 9):                           # eight more masks pass through a cache of four
        c.put(f"m{i}", (0,), f"plan{i}")
    assert c.get("m0", (0,)) == "plan0" and c.get("mcap", (0,)) == "plancap"
    assert c.get("m1", (0,)) is None and c.get("m8", (0,)) == "plan8"
    assert sum(1 for e in c.entries.values() if not e[2]) <= 4
    # an in-place parameter update bumps the version: stale entries are misses; a pinned one stays allocated (parked)
    assert c.get("m8", (1,)) is None and "m8" not in c.entries
    assert c.get("m0", (1,)) is None and "plan0" in c.parked
    c.clear()
    assert len(c) == 0 and not c.parked


def test_resnet_plan_follows_in_place_parameter_updates(tiny):
    """a fine-tuning loop that never calls invalidate_plans() must still compute with the current weights"""
    cfg, model, params = tiny
    model.set_structure(clone_mask(O.fixed_half_mask(cfg)))
    r = model.down_blocks[0].resnets[0]
    p0 = r.plan(torch.device("cpu"))
    assert r.plan(torch.device("cpu")) is p0
    with torch.no_grad():
        r.conv1.weight.mul_(2.0)                    # what optimizer.step() does: an in-place update
    p1 = r.plan(torch.device("cpu"))
    assert p1 is not p0
    assert torch.allclose(p1["w1"].w.float(), (p0["w1"].w.float() * 2), rtol=1e-2)
    assert len(r._plans) == 1                       # the stale plan is gone, not merely shadowed
    t = model.down_blocks[0].attentions[0]
    q0 = t.plan(torch.device("cpu"))
    with torch.no_grad():
        t.proj_in.bias.add_(1.0)
    assert t.plan(torch.device("cpu")) is not q0


def test_prefetch_plan_is_per_thread_and_scoped_to_the_forward(tiny):
    import threading
    from diffusion_pruning_amd import ops
    cfg, model, params = tiny
    seen = {}

    def probe(x, pw, **kw):
        seen.setdefault(threading.get_ident(), []).append(ops._current_prefetch_plan())
        return hip_emulator.conv_gemm(x, pw, **kw)
    ops.conv_gemm = probe                           # (restored by the fixture's monkeypatch teardown)
    sample, t, ehs = O.synthetic_inputs(cfg, 1, 16)
    run(model, O.ones_mask(cfg), sample, t, ehs)
    main_plans = set(id(p) for p in seen[threading.get_ident()])
    assert len(main_plans) == 1 and None not in seen[threading.get_ident()]
    assert ops._current_prefetch_plan() is None     # cleared when the forward returns
    other = {}
    th = threading.Thread(target=lambda: other.setdefault("plan", ops._current_prefetch_plan()))
    th.start(); th.join()
    assert other["plan"] is None                    # another thread never sees this thread's plan


def test_tuning_lookup_falls_back_to_the_nearest_tuned_shape_of_the_same_class():
    """shapes of architecture codes that were never tuned (config 5's experts) take the tile of the closest tuned shape"""
    from diffusion_pruning_amd import ops
    exact = ops.tuning_lookup(16384, 320, 320, 1, 1, 0, False)
    assert exact is ops.TUNING[ops.tuning_key(16384, 320, 320, 1, 1, 0, False)]
    near = ops.tuning_lookup(16384, 328, 320, 1, 1, 0, False)               # a 41-group expert width: not in the table
    assert near is not None and near["tile"] == exact["tile"]
    assert ops.tuning_lookup(16384, 328, 320, 1, 1, 0, False) is near       # cached
    assert ops.tuning_lookup(24, 40, 72, 1, 1, 0, False) is None            # nothing within reach: the library heuristic decides
    far3x3 = ops.tuning_lookup(16384, 176, 352, 9, 1, 0, False)
    assert far3x3 is not None and far3x3["tile"] not in ops._HALO_TILES
    # a neighbour's split-K never leaves a slice with fewer than four K-steps
    deep = ops.tuning_lookup(256, 1280, 1200, 9, 1, 0, False)
    assert deep is not None and deep["split_k"] <= max(1, (9 * 1200 // 64) // 4)
    # the LDS-DMA tiles (every id past the six register-staged ones) take channel steps up to 4032: a shape with more channels per tap
    # never inherits one from a neighbour (round 4: the batch-64 training shapes put 128x160 DMA tiles next to K = 5120 linears)
    for M in (16384, 12288, 4096 * 3):
        wide = ops.tuning_lookup(M, 1280, 5100, 1, 1, 0, False)
        assert wide is None or wide["tile"] <= 6, wide


# ---- the same host logic without any rounding: fp32 storage end to end (emulator exact mode) vs the fp32 oracle -------------
EXACT_TOL = 1e-5


@pytest.fixture()
def tiny_exact(monkeypatch):
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    hip_emulator.install(monkeypatch, exact=True)
    cfg = O.TINY
    model = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                      cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return cfg, model, params


@pytest.mark.parametrize("case", ["dense", "half_gated", "random_depth", "soft_cfg", "hard_per_sample"])
def test_host_logic_is_exact_in_fp32(tiny_exact, case):
    """SURVEY §8 preamble asks for an fp32 path at <= 1e-5 per op / <= 1e-4 whole U-Net.  The MFMA kernels are bf16 by
    construction, so what an fp32 run can pin is everything AROUND them: weight compaction by architecture code, the
    GroupNorm-beta border-class correction that separates gated from pruned semantics, gate / depth-lerp epilogue
    semantics, LayerNorm folding into packed weights, the fused shortcut K-segment, batched time-embedding and text K/V
    projections, in-place skip-concats.  Here that logic runs with fp32 storage end to end (no rounding point left) and
    must reproduce the fp32 oracle to 1e-5 -- two orders below the 1e-2 level at which bf16 storage would mask a defect."""
    cfg, model, params = tiny_exact
    if case == "dense":
        B, mask, sem, seed = 2, O.ones_mask(cfg), "gated", 1
    elif case == "half_gated":
        B, mask, sem, seed = 2, O.fixed_half_mask(cfg), "gated", 2
    elif case == "random_depth":
        B, mask, sem, seed = 2, O.random_mask(cfg, 0.4, 1, n_depth_off=2), "gated", 3
    elif case == "soft_cfg":
        g = torch.Generator().manual_seed(4)
        st = O.get_structure(cfg)
        B, sem, seed = 4, "gated", 9
        mask = {"width": [torch.rand(2, w, generator=g) * 0.9 + 0.1 for sub in st["width"] for w in sub],
                "depth": [torch.rand(2, generator=g) for sub in st["depth"] for d in sub if d == 1]}
    else:
        B, sem, seed = 2, "gated", 5
        mask = O.random_mask(cfg, 0.5, 7, n_depth_off=1, batch=2)
        mask["depth"][0] = torch.tensor([1.0, 0.0])
    sample, t, ehs = O.synthetic_inputs(cfg, B, 16, seed=seed)
    ref, blocks = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), sem, return_blocks=True)
    seen = {}
    hooks = [model.mid_block.register_forward_hook(lambda m, i, o: seen.__setitem__("mid", o)),
             model.down_blocks[0].register_forward_hook(lambda m, i, o: seen.__setitem__("down0", o[0]))]
    out = run(model, mask, sample, t, ehs)
    for h in hooks:
        h.remove()
    assert out.dtype == torch.float32
    assert rel_l2(seen["down0"].float(), blocks[0]) <= EXACT_TOL
    assert rel_l2(seen["mid"].float(), blocks[4]) <= EXACT_TOL
    e = rel_l2(out, ref)
    assert e <= EXACT_TOL, e


def test_pruned_semantics_is_exact_in_fp32(tiny_exact):
    from diffusion_pruning_amd.unet import UNet2DConditionModelPruned
    cfg, _, params = tiny_exact
    pm = UNet2DConditionModelPruned(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                    cross_attention_dim=cfg.cross_attention_dim)
    pm.load_state_dict(params)
    mask = O.random_mask(cfg, 0.5, 3, n_depth_off=2)
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16, seed=8)
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "pruned")
    pm.prune(clone_mask(mask))
    with torch.no_grad():
        out = pm(sample, t, ehs).sample.float()
    assert rel_l2(out, ref) <= EXACT_TOL
    gated = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    assert rel_l2(out, gated) > 100 * EXACT_TOL        # ... and the two semantics really differ (the beta term)


@pytest.mark.parametrize("wn,bias", [(True, False), (False, True), (True, True), (False, False)])
def test_hypernet_heads_as_one_gemm_equal_the_per_head_loop(wn, bias):
    """HyperStructure._forward_fused (parameters re-homed into flat storage, one weight-normalised GEMM) vs the reference's
    per-head loop (hypernet.py:62-72): same outputs, same per-parameter gradients, same state_dict keys; the aliasing
    survives optimizer steps, load_state_dict and deepcopy."""
    import copy
    from diffusion_pruning_amd.hypernet import HyperStructure
    st = {"width": [[32, 32], [5, 5, 20]], "depth": [[1, 1], [1]]}
    torch.manual_seed(0)
    h = HyperStructure(st, wn_flag=wn, linear_bias=bias)
    if bias:
        with torch.no_grad():
            for hd in h.mh_fc:
                hd.bias.normal_()
    loop = copy.deepcopy(h)
    loop.fuse_heads = False
    x = torch.randn(4, 768)
    a, b = h(x), loop(x)
    assert float((a - b).abs().max()) <= 1e-6
    a.square().sum().backward()
    b.square().sum().backward()
    for (n, p), (n2, p2) in zip(h.named_parameters(), loop.named_parameters()):
        assert n == n2 and p.grad is not None
        assert float((p.grad - p2.grad).abs().max()) <= 1e-5 * float(p2.grad.abs().max()) + 1e-12, n
    assert list(h.state_dict().keys()) == list(loop.state_dict().keys())
    o1, o2 = torch.optim.AdamW(h.parameters(), 1e-2), torch.optim.AdamW(loop.parameters(), 1e-2)
    o1.step(); o2.step()
    assert float((h(x) - loop(x)).abs().max()) < 1e-5
    assert h._flat["v"].data_ptr() == h._head_params()["v"][0].data_ptr()
    loop2 = HyperStructure(st, wn_flag=wn, linear_bias=bias)
    h.load_state_dict(loop2.state_dict())
    loop2.fuse_heads = False
    assert float((h(x) - loop2(x)).abs().max()) <= 1e-6
    assert float((copy.deepcopy(h)(x) - loop2(x)).abs().max()) <= 1e-6
    h.double()                                       # re-allocates the parameters: the flat buffers are rebuilt
    assert h(x.double()).dtype == torch.float64 and h._flat["v"].dtype == torch.float64


@pytest.mark.parametrize("fixed", [False, True])
def test_one_host_draw_equals_the_per_block_gumbel_draws(fixed):
    """estimation_utils.sample_gumbel_blocks vs the reference's order of draws (quantizer.py:196-213: depth block, then one
    torch.rand per width segment on the host generator): identical values AND the generator ends in the same state."""
    from diffusion_pruning_amd.estimation_utils import sample_gumbel, sample_gumbel_blocks
    widths = [14, 320, 320, 5, 1280, 17, 1]
    for B in (1, 4):
        torch.manual_seed(77)
        ref = torch.cat([sample_gumbel((B, w), fixed_seed=fixed) for w in widths], dim=1)
        after_ref = torch.rand(3)
        torch.manual_seed(77)
        got = sample_gumbel_blocks(B, widths, fixed_seed=fixed)
        after_got = torch.rand(3)
        assert torch.equal(ref, got)
        assert torch.equal(after_ref, after_got)


def test_scratch_domain_is_thread_local_and_restores():
    """ops.scratch_domain: the key under which split-K counters / scratch are looked up (graphs that replay concurrently are
    captured under different domains); nesting restores the outer name, other threads are not affected"""
    import threading
    from diffusion_pruning_amd import ops
    assert ops._domain() is None
    seen = {}
    with ops.scratch_domain("teacher"):
        assert ops._domain() == "teacher"
        t = threading.Thread(target=lambda: seen.setdefault("other", ops._domain()))
        t.start(); t.join()
        with ops.scratch_domain("inner"):
            assert ops._domain() == "inner"
        assert ops._domain() == "teacher"
    assert ops._domain() is None and seen["other"] is None


def test_single_arch_param_baseline_repeats_one_code_over_the_batch():
    """trainer.py:1134-1136: with ``HyperStructure(single_arch_param=True)`` the one learned architecture vector is repeated over
    the batch AFTER the Gumbel-sigmoid relaxation (every sample sees the same noise draw), kept as ``hyper_net.arch_gs``, and
    its gradient reaches the single parameter"""
    from diffusion_pruning_amd.hypernet import HyperStructure
    from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
    from diffusion_pruning_amd.train_step import PrunerStep, synthetic_batch
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    from tests.test_distributed_cpu import DEPTH_ORDER, StubUNet
    cfg = O.TINY
    real = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                     cross_attention_dim=cfg.cross_attention_dim)
    torch.manual_seed(1)
    hn = HyperStructure(structure=real.get_structure(), input_dim=16, single_arch_param=True)
    qz = StructureVectorQuantizer(n_e=4, structure=real.get_structure(), temperature=0.4, base=3, depth_order=DEPTH_ORDER,
                                  resource_aware_normalization=False, optimal_transport=False)
    step = PrunerStep(StubUNet(real), hn, qz)
    hn.train(); qz.train()
    step.count_macs(8)
    batch = synthetic_batch(3, 8, "cpu", seed=5, cross_dim=cfg.cross_attention_dim, text_dim=16)
    torch.manual_seed(3)
    out = step.step(batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"], batch["mpnet_embeddings"],
                    batch["target"], pretrain=True)
    E = sum(hn.width_list) + sum(hn.depth_list)
    assert hn.arch_gs.shape == (3, E) and torch.equal(hn.arch_gs[0], hn.arch_gs[1]) and torch.equal(hn.arch_gs[0], hn.arch_gs[2])
    out["loss"].backward()
    assert hn.arch.grad is not None and float(hn.arch.grad.abs().sum()) > 0 and torch.isfinite(out["loss"])
