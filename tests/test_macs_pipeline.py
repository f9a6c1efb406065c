"""CPU tests of the MAC accounting and of the denoise-loop glue (a21) with the emulated ops."""
import pytest
import torch

from oracle import unet_oracle as O
from tests import hip_emulator


def _val(v):
    return float(torch.as_tensor(v).flatten()[0])


def test_mac_accounting_conventions():
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    m = UNet2DConditionModelGated()
    m.set_structure(O.ones_mask(O.SD21))
    d = m.count_macs(64)
    # independent recomputation of the hook conventions: matmul-only MACs (App. D) + quirk Q4 (cross-attention counted
    # with L_query^2) + bias / norm / activation terms
    q4 = 0.0
    for b in O.build_specs(O.SD21):
        lvl = {"down": lambda n: int(n.split(".")[1]), "mid": lambda n: 3, "up": lambda n: 3 - int(n.split(".")[1])}[b.kind](b.name)
        P = (64 >> lvl) ** 2
        for a in b.attns:
            q4 += a.heads * (2.0 * P * P * 64 + P * P) - 2.0 * a.heads * P * 77 * 64      # cross attention as counted - as is
            q4 += a.heads * P * P                                                           # softmax term of self attention
    matmul = O.count_macs(O.SD21, 64)
    assert 0 < _val(d["total_macs"]) - (matmul + q4) < 0.004 * matmul       # what is left: biases, norms, SiLUs
    assert _val(d["prunable_macs"]) < _val(d["total_macs"])
    # with all gates on, the depth-gated modules add their non-prunable part (blocks.py:630-631)
    assert _val(d["cur_prunable_macs"]) > _val(d["prunable_macs"])
    assert abs(_val(d["cur_total_macs"]) - _val(d["total_macs"])) < 1e-3 * _val(d["total_macs"])
    pm = m.get_prunable_macs()
    assert len(pm) == 38 and sum(len(x) for x in pm) == 70
    assert abs(sum(v for sub in m.prunable_macs_list for v in sub) - 1.0) < 1e-5
    # fixed 50 % mask: every prunable part halves except 2/5 heads at the first level
    m.set_structure(O.fixed_half_mask(O.SD21))
    r = m.calc_macs()["cur_prunable_macs"] / m.resource_info_dict["cur_prunable_macs"]
    assert 0.45 < _val(r) < 0.5
    util = m.get_block_utilization()
    assert len(util) == 9 and abs(_val(util[0][0]) - 0.5) < 1e-6 and abs(_val(util[0][1]) - (0.4 * 2 + 0.5) / 3) < 0.08
    # depth gate off removes the whole module from cur_prunable
    mask = O.fixed_half_mask(O.SD21)
    mask["depth"][0] = torch.zeros(1)
    m.set_structure(mask)
    r2 = m.calc_macs()["cur_prunable_macs"] / m.resource_info_dict["cur_prunable_macs"]
    assert _val(r2) < _val(r)


def test_resource_ratio_is_differentiable_through_the_gates():
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    from diffusion_pruning_amd.losses import ResourceLoss
    cfg = O.TINY
    m = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                  cross_attention_dim=cfg.cross_attention_dim)
    m.set_structure(O.ones_mask(cfg))
    m.count_macs(16)
    g = torch.Generator().manual_seed(0)
    st = O.get_structure(cfg)
    width = [torch.rand(3, w, generator=g).requires_grad_() for sub in st["width"] for w in sub]
    depth = [torch.rand(3, generator=g).requires_grad_() for sub in st["depth"] for d in sub if d == 1]
    m.set_structure({"width": list(width), "depth": list(depth)})
    ratios = m.calc_macs()["cur_prunable_macs"] / m.resource_info_dict["cur_prunable_macs"].squeeze()
    assert ratios.shape == (3, 1)
    loss = ResourceLoss(p=0.6)(ratios.mean())
    loss.backward()
    assert all(w.grad is not None for w in width) and all(d.grad is not None for d in depth)


def test_denoise_loop_matches_oracle_loop(monkeypatch):
    """CFG doubling + U-Net + guidance + DDIM update over 3 steps, K/V context computed once, vs the same loop around
    the oracle U-Net."""
    from diffusion_pruning_amd.pipeline import DDIMSchedulerLite, PruningDenoiseLoop
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    hip_emulator.install(monkeypatch)
    cfg = O.TINY
    model = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                      cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    mask = O.fixed_half_mask(cfg)
    model.set_structure({k: [v.clone() for v in vs] for k, vs in mask.items()})
    g = torch.Generator().manual_seed(2)
    B = 2
    lat = torch.randn(B, 4, 16, 16, generator=g)
    cond = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    uncond = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    loop = PruningDenoiseLoop(model)
    out = loop(cond, lat.clone(), num_inference_steps=3, guidance_scale=3.0, negative_prompt_embeds=uncond, use_graph=False)
    # reference loop
    sch = DDIMSchedulerLite()
    ts = sch.set_timesteps(3)
    gates = O.assign_gates(cfg, mask)
    x = lat.clone()
    ehs = torch.cat([uncond, cond])
    for i in range(3):
        noise = O.unet_forward(params, cfg, torch.cat([x, x]), ts[i].expand(2 * B), ehs, gates, "gated")
        u, c = noise.chunk(2)
        x = sch.step_coef(u + 3.0 * (c - u), sch.coef[i], x)
    rel = float((out.latents - x).norm() / x.norm())
    # guidance amplifies the per-forward bf16 error (<= 2e-2) by up to (2s - 1); three steps accumulate
    assert rel < 5e-2, rel
    assert ts.tolist() == [667, 334, 1]


def test_scheduler_reconstructs_x0_for_exact_v():
    from diffusion_pruning_amd.pipeline import DDIMSchedulerLite
    sch = DDIMSchedulerLite()
    sch.set_timesteps(10)
    g = torch.Generator().manual_seed(0)
    x0, eps = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    c = sch.coef[3]
    xt = c[0] * x0 + c[1] * eps
    v = c[0] * eps - c[1] * x0
    prev = sch.step_coef(v, c, xt)
    assert torch.allclose(prev, c[2] * x0 + c[3] * eps, atol=1e-5)


# ---- PNDM / PLMS (configs/img_generation/sd-2-1_cc3m.yaml:50; pruning_pipelines.py:810-814) --------------------------------
def _plms_reference(model, x, N, prediction_type):
    """PLMS with explicit history lists, written the way the published algorithm / diffusers' step_plms is structured
    (counter, ets, cur_sample), in float64: the table-driven tensor form of PNDMSchedulerLite must agree with it."""
    import numpy as np
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas)
    final = ac[0]
    ratio = 1000 // N
    base = (np.arange(0, N) * ratio).round().astype(np.int64) + 1
    ts = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1]

    def transfer(sample, t, t_prev, out):
        a_t = ac[t]
        a_p = ac[t_prev] if t_prev >= 0 else final
        if prediction_type == "v_prediction":
            out = a_t ** 0.5 * out + (1 - a_t) ** 0.5 * sample
        return (a_p / a_t) ** 0.5 * sample - (a_p - a_t) * out / (a_t * (1 - a_p) ** 0.5 + (a_t * (1 - a_t) * a_p) ** 0.5)

    ets, counter, cur_sample = [], 0, None
    x = x.double()
    for t in ts.tolist():
        out = model(x, t).double()
        prev_t = t - ratio
        if counter != 1:
            ets = ets[-3:]
            ets.append(out)
        else:
            prev_t, t = t, t + ratio
        if len(ets) == 1 and counter == 0:
            cur_sample = x
        elif len(ets) == 1 and counter == 1:
            out = (out + ets[-1]) / 2
            x, cur_sample = cur_sample, None
        elif len(ets) == 2:
            out = (3 * ets[-1] - ets[-2]) / 2
        elif len(ets) == 3:
            out = (23 * ets[-1] - 16 * ets[-2] + 5 * ets[-3]) / 12
        else:
            out = (55 * ets[-1] - 59 * ets[-2] + 37 * ets[-3] - 9 * ets[-4]) / 24
        x = transfer(x, t, prev_t, out)
        counter += 1
    return x


@pytest.mark.parametrize("prediction_type", ["epsilon", "v_prediction"])
@pytest.mark.parametrize("N", [4, 20, 50])
def test_pndm_tables_equal_the_list_form_of_plms(prediction_type, N):
    from diffusion_pruning_amd.pipeline import PNDMSchedulerLite
    g = torch.Generator().manual_seed(N)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    Wm = torch.randn(4, 4, generator=g) * 0.3

    def model(x, t):                                       # any deterministic function of (x, t)
        return torch.tanh(torch.einsum("oc,bchw->bohw", Wm.to(x.dtype), x)) * 0.5 + 0.1 * (t / 1000.0)

    ref = _plms_reference(model, x0, N, prediction_type)
    sch = PNDMSchedulerLite(prediction_type=prediction_type)
    ts = sch.set_timesteps(N)
    assert len(ts) == N + 1 and int(ts[1]) == int(ts[2]) and int(ts[-1]) == 1
    state = sch.make_state(x0)
    x = x0.clone()
    for i in range(sch.n_model_calls()):
        sch.load_step(state, i)
        x = sch.step(model(x, int(ts[i])), x, state)
    assert float((x.double() - ref).norm() / ref.norm()) <= 2e-5


def test_pndm_with_a_constant_model_output_is_ddim():
    """every Adams-Bashforth combination of a constant is that constant, and PNDM's transfer is DDIM's deterministic update"""
    from diffusion_pruning_amd.pipeline import DDIMSchedulerLite, PNDMSchedulerLite
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(1, 4, 8, 8, generator=g)
    eps = torch.randn(1, 4, 8, 8, generator=g)
    N = 25
    pn, dd = PNDMSchedulerLite(prediction_type="epsilon"), DDIMSchedulerLite(prediction_type="epsilon")
    pn.set_timesteps(N); dd.set_timesteps(N)
    sp, sd = pn.make_state(x0), dd.make_state(x0)
    xp = xd = x0
    for i in range(pn.n_model_calls()):
        pn.load_step(sp, i)
        xp = pn.step(eps, xp, sp)
    for i in range(dd.n_model_calls()):
        dd.load_step(sd, i)
        xd = dd.step(eps, xd, sd)
    assert float((xp - xd).norm() / xd.norm()) <= 1e-5
