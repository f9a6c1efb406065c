"""GPU parity of the whole masked U-Net forward (HIP path through the C ABI) against the CPU oracle.

Tolerance: bf16 storage / fp32 accumulate end to end => whole-U-Net output rel-L2 <= 2e-2 against the
fp32 oracle on the same seeded weights and inputs.
"""
import pytest
import torch

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
from tests.margins import check  # noqa: E402
TOL = 2e-2


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def clone_mask(m, dev=None):
    return {k: [v.clone() if dev is None else v.clone().to(dev) for v in vs] for k, vs in m.items()}


@pytest.fixture(scope="module")
def tiny(cuda):
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    cfg = O.TINY
    model = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                      cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(cuda)
    return cfg, model, params


def run(model, cuda, mask, sample, t, ehs):
    model.set_structure(clone_mask(mask, cuda))
    with torch.no_grad():
        out = model(sample.to(cuda), t.to(cuda), ehs.to(cuda)).sample
    torch.cuda.synchronize()
    return out.float().cpu()


def test_dense_ones_mask(tiny, cuda):
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16)
    ref = O.unet_forward(params, cfg, sample, t, ehs)
    out = run(model, cuda, O.ones_mask(cfg), sample, t, ehs)
    assert out.shape == ref.shape
    e = rel_l2(out, ref)
    check(e, TOL)


def test_fixed_half_mask_gated_semantics(tiny, cuda):
    """hard batch-shared mask -> compacted weights + GroupNorm-beta border correction == reference gated model"""
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16)
    mask = O.fixed_half_mask(cfg)
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    refp = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "pruned")
    out = run(model, cuda, mask, sample, t, ehs)
    e, ep = rel_l2(out, ref), rel_l2(out, refp)
    check(e, TOL)
    assert ep > 2 * e, (e, ep)   # the beta term is really there: we match gated, not pruned, semantics


@pytest.mark.parametrize("batch,latent,masked", [(3, 40, True), (1, 48, False), (2, 32, True), (1, 64, True)])
def test_other_resolutions_and_batches(tiny, cuda, batch, latent, masked):
    """latent sizes whose first levels are large maps (HW >= 1024: producer-emitted GroupNorm statistics, in-place
    skip-concats) with widths that are not powers of two (40, 48: statistics row blocks / tiles straddle image rows and
    samples), odd batch sizes, gated and dense: SURVEY 8d 'other resolutions'"""
    from diffusion_pruning_amd import unet as U
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, batch, latent)
    mask = O.fixed_half_mask(cfg) if masked else O.ones_mask(cfg)
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    U.CAT_STATS.update(views=0, copies=0)
    out = run(model, cuda, mask, sample, t, ehs)
    assert out.shape == ref.shape
    e = rel_l2(out, ref)
    check(e, TOL)
    assert U.CAT_STATS["copies"] == 0 and U.CAT_STATS["views"] == sum(len(b.resnets) for b in model.up_blocks)


@pytest.mark.parametrize("seed,keep,ndoff", [(1, 0.4, 2), (2, 0.75, 4), (3, 0.55, 0)])
def test_random_hard_masks_with_depth(tiny, cuda, seed, keep, ndoff):
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16, seed=seed)
    mask = O.random_mask(cfg, keep, seed, n_depth_off=ndoff)
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    out = run(model, cuda, mask, sample, t, ehs)
    e = rel_l2(out, ref)
    check(e, TOL)


def test_soft_per_sample_masks_and_cfg_tiling(tiny, cuda):
    """training-style soft gates, one row per prompt, activation batch = 2 x gate batch (CFG layout)"""
    cfg, model, params = tiny
    B = 4
    sample, t, ehs = O.synthetic_inputs(cfg, B, 16, seed=9)
    g = torch.Generator().manual_seed(4)
    st = O.get_structure(cfg)
    width = [torch.rand(2, w, generator=g) * 0.9 + 0.1 for sub in st["width"] for w in sub]
    depth = [torch.rand(2, generator=g) for sub in st["depth"] for d in sub if d == 1]
    mask = {"width": width, "depth": depth}
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    out = run(model, cuda, mask, sample, t, ehs)
    e = rel_l2(out, ref)
    check(e, TOL)


def test_hard_per_sample_masks(tiny, cuda):
    """different hard mask per sample -> dense path with 0/1 gates in the epilogue"""
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16, seed=5)
    mask = O.random_mask(cfg, 0.5, 7, n_depth_off=1, batch=2)
    mask["depth"][0] = torch.tensor([1.0, 0.0])
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(mask)), "gated")
    out = run(model, cuda, mask, sample, t, ehs)
    e = rel_l2(out, ref)
    check(e, TOL)


def test_pruned_model_semantics(tiny, cuda):
    from diffusion_pruning_amd.unet import UNet2DConditionModelPruned
    cfg, model, params = tiny
    pm = UNet2DConditionModelPruned(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                    cross_attention_dim=cfg.cross_attention_dim)
    pm.load_state_dict(params)
    pm.to(cuda)
    mask = O.random_mask(cfg, 0.5, 11, n_depth_off=2)
    soft = {"width": [w * 0.9 for w in mask["width"]], "depth": [d * 0.9 for d in mask["depth"]]}  # 0.9 / 0 like the reference
    pm.prune(clone_mask(soft, cuda))
    sample, t, ehs = O.synthetic_inputs(cfg, 2, 16, seed=6)
    ref = O.unet_forward(params, cfg, sample, t, ehs, O.assign_gates(cfg, clone_mask(soft)), "pruned")
    with torch.no_grad():
        out = pm(sample.to(cuda), t.to(cuda), ehs.to(cuda), return_dict=False)[0]
    e = rel_l2(out.float().cpu(), ref)
    check(e, TOL)


def test_forward_hooks_see_block_outputs(tiny, cuda):
    """trainer.py:496-511 registers forward hooks on down/mid/up blocks; shapes follow the diffusers convention"""
    cfg, model, params = tiny
    sample, t, ehs = O.synthetic_inputs(cfg, 1, 16)
    seen = {}
    hooks = []
    for name, blk in [("down0", model.down_blocks[0]), ("mid", model.mid_block), ("up3", model.up_blocks[3])]:
        hooks.append(blk.register_forward_hook(lambda m, i, o, name=name: seen.__setitem__(name, o)))
    ref_out, ref_blocks = O.unet_forward(params, cfg, sample, t, ehs, return_blocks=True)
    run(model, cuda, O.ones_mask(cfg), sample, t, ehs)
    for h in hooks:
        h.remove()
    assert isinstance(seen["down0"], tuple) and len(seen["down0"]) == 2          # (hidden, res_tuple)
    assert seen["mid"].shape == ref_blocks[4].shape
    check(rel_l2(seen["mid"].float().cpu(), ref_blocks[4]), TOL, "ref_blocks[4]")
    check(rel_l2(seen["up3"].float().cpu(), ref_blocks[8]), TOL, "ref_blocks[8]")
    check(rel_l2(seen["down0"][0].float().cpu(), ref_blocks[0]), TOL, "ref_blocks[0]")


def test_denoise_loop_hip_graph_matches_eager_and_oracle(tiny, cuda):
    """pipeline glue: CFG doubling + U-Net + guidance + DDIM update; one captured HIP graph replayed per
    step with the cross-attention context computed once == eager loop == oracle loop"""
    from diffusion_pruning_amd.pipeline import DDIMSchedulerLite, PruningDenoiseLoop
    cfg, model, params = tiny
    mask = O.fixed_half_mask(cfg)
    model.set_structure(clone_mask(mask, cuda))
    g = torch.Generator().manual_seed(2)
    B, steps, s = 2, 4, 3.0
    lat = torch.randn(B, 4, 16, 16, generator=g)
    cond = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    uncond = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    loop = PruningDenoiseLoop(model)
    out_g = loop(cond.to(cuda), lat.to(cuda), steps, s, negative_prompt_embeds=uncond.to(cuda), use_graph=True).latents
    out_e = loop(cond.to(cuda), lat.to(cuda), steps, s, negative_prompt_embeds=uncond.to(cuda), use_graph=False).latents
    torch.cuda.synchronize()
    assert rel_l2(out_g.float().cpu(), out_e.float().cpu()) < 1e-6          # same kernels, same order
    sch = DDIMSchedulerLite()
    ts = sch.set_timesteps(steps)
    gates = O.assign_gates(cfg, clone_mask(mask))
    x = lat.clone()
    ehs = torch.cat([uncond, cond])
    for i in range(steps):
        noise = O.unet_forward(params, cfg, torch.cat([x, x]), ts[i].expand(2 * B), ehs, gates, "gated")
        u, c = noise.chunk(2)
        x = sch.step_coef(u + s * (c - u), sch.coef[i], x)
    assert rel_l2(out_g.float().cpu(), x) < 6e-2      # guidance amplifies the per-forward bf16 error; 4 steps accumulate


def test_pndm_loop_and_graph_reuse_across_prompt_batches(tiny, cuda):
    """PNDM / PLMS scheduler (the reference's image-generation default) in the captured-step loop: graph == eager; a second
    prompt batch through the same expert re-uses the captured step (FID-generation use case) and still equals eager"""
    from diffusion_pruning_amd.pipeline import PNDMSchedulerLite, PruningDenoiseLoop
    cfg, model, params = tiny
    model.set_structure(clone_mask(O.fixed_half_mask(cfg), cuda))
    g = torch.Generator().manual_seed(6)
    B, steps, s = 2, 5, 2.0
    loop = PruningDenoiseLoop(model, scheduler=PNDMSchedulerLite())
    graphs = []
    for call in range(2):
        lat = torch.randn(B, 4, 16, 16, generator=g).to(cuda)
        cond = torch.randn(B, 77, cfg.cross_attention_dim, generator=g).to(cuda)
        uncond = torch.randn(B, 77, cfg.cross_attention_dim, generator=g).to(cuda)
        out_g = loop(cond, lat, steps, s, negative_prompt_embeds=uncond, use_graph=True).latents
        graphs.append(loop._graph["graph"])
        out_e = loop(cond, lat, steps, s, negative_prompt_embeds=uncond, use_graph=False).latents
        torch.cuda.synchronize()
        assert torch.isfinite(out_g).all()
        assert rel_l2(out_g.float().cpu(), out_e.float().cpu()) < 1e-5
    assert graphs[0] is graphs[1]                          # captured once
    model.set_structure(clone_mask(O.ones_mask(cfg), cuda))      # another architecture code: the key changes, a new capture
    loop(cond, lat, steps, s, negative_prompt_embeds=uncond, use_graph=True)
    assert loop._graph["graph"] is not graphs[0]
