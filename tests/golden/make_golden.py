"""Generates tests/golden/reference_vectors.npz by RUNNING the reference's own importable code.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):
    python tests/golden/make_golden.py
Imports, by file path, exactly these reference modules (nothing is copied into this repository):
    pdm/models/unet/gates.py, pdm/utils/estimation_utils.py, pdm/losses/{contrastive,resource}_loss.py,
    pdm/utils/metric_utils.py, pdm/models/hypernet/hypernet.py, pdm/models/vq/quantizer.py, pdm/utils/op_counter.py
hypernet.py / quantizer.py use diffusers only for ModelMixin / ConfigMixin / register_to_config (serialisation);
an inert in-process shim stands in for those three names and touches no arithmetic.
op_counter.py (the MAC-counting forward hooks) imports six diffusers / pdm classes only to use them as KEYS of its
hook table; inert placeholder classes stand in for them, and the reference's own hook functions are then called on
plain torch.nn modules (conv / linear / GroupNorm / LayerNorm / SiLU) and on an attribute-only stand-in of a gated
attention module: the numbers recorded are what the reference's code computes for those shapes.
The committed .npz holds inputs and the reference's outputs only (data, no source).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_shims():
    d = types.ModuleType("diffusers")
    d.ModelMixin = nn.Module
    d.ConfigMixin = object
    cu = types.ModuleType("diffusers.configuration_utils")
    cu.register_to_config = lambda f: f
    cu.ConfigMixin = object
    d.configuration_utils = cu
    sys.modules["diffusers"] = d
    sys.modules["diffusers.configuration_utils"] = cu
    # make `pdm.utils.estimation_utils` importable without executing pdm/models/__init__.py
    pdm = types.ModuleType("pdm"); pdm.__path__ = []
    utils = types.ModuleType("pdm.utils"); utils.__path__ = []
    sys.modules["pdm"], sys.modules["pdm.utils"] = pdm, utils
    eu = load("pdm.utils.estimation_utils", "pdm/utils/estimation_utils.py")
    utils.estimation_utils = eu
    return eu


def macs_hook_vectors():
    """Run the reference's MAC-counting hooks (pdm/utils/op_counter.py:44-116,259-306) on concrete modules / shapes.
    Each case is recorded as an int64 row of its defining numbers followed by the MACs the hook reported."""
    dm = types.ModuleType("diffusers.models"); dm.__path__ = []
    ap = types.ModuleType("diffusers.models.attention_processor")
    ap.SpatialNorm = type("SpatialNorm", (), {})
    ap.Attention = type("Attention", (), {})
    nz = types.ModuleType("diffusers.models.normalization")
    nz.AdaGroupNorm = type("AdaGroupNorm", (), {})
    lo = types.ModuleType("diffusers.models.lora")
    lo.LoRACompatibleConv = type("LoRACompatibleConv", (), {})
    lo.LoRACompatibleLinear = type("LoRACompatibleLinear", (), {})
    sys.modules.update({"diffusers.models": dm, "diffusers.models.attention_processor": ap,
                        "diffusers.models.normalization": nz, "diffusers.models.lora": lo})
    pm = types.ModuleType("pdm.models"); pm.__path__ = []
    pu = types.ModuleType("pdm.models.unet"); pu.__path__ = []
    pb = types.ModuleType("pdm.models.unet.blocks")
    pb.GatedAttention = type("GatedAttention", (), {})
    sys.modules.update({"pdm.models": pm, "pdm.models.unet": pu, "pdm.models.unet.blocks": pb})
    oc = load("ref_op_counter", "pdm/utils/op_counter.py")
    out = {}

    def run(hook, mod, *inputs):
        mod.__macs__ = 0
        with torch.no_grad():
            y = mod(*inputs)
        hook(mod, inputs, y)
        return int(mod.__macs__)

    rows = []          # [cin, cout, k, stride, pad, bias, H, W] -> macs
    for cin, cout, k, stride, pad, bias, H, W in [(4, 320, 3, 1, 1, 1, 16, 16), (320, 320, 3, 2, 1, 1, 16, 16),
                                                  (640, 320, 1, 1, 0, 1, 8, 8), (96, 64, 3, 1, 1, 0, 12, 20),
                                                  (320, 4, 3, 1, 1, 1, 64, 64)]:
        m = nn.Conv2d(cin, cout, k, stride, pad, bias=bool(bias))
        rows.append([cin, cout, k, stride, pad, bias, H, W, run(oc.conv_macs_counter_hook, m, torch.zeros(1, cin, H, W))])
    out["macs_hook_conv"] = np.asarray(rows, dtype=np.int64)
    rows = []          # [rows of input, cin, cout, bias] -> macs
    for L, cin, cout, bias in [(1, 320, 1280, 1), (256, 320, 320, 1), (77, 1024, 640, 0), (4096, 320, 2560, 1), (64, 5120, 1280, 1)]:
        m = nn.Linear(cin, cout, bias=bool(bias))
        rows.append([L, cin, cout, bias, run(oc.linear_macs_counter_hook, m, torch.zeros(1, L, cin))])
    out["macs_hook_linear"] = np.asarray(rows, dtype=np.int64)
    rows = []          # [C, H, W] -> GroupNorm(32, C) macs, SiLU macs on the same tensor
    for C, H, W in [(320, 64, 64), (2560, 8, 8), (960, 32, 32)]:
        x = torch.zeros(1, C, H, W)
        rows.append([C, H, W, run(oc.bn_macs_counter_hook, nn.GroupNorm(32, C), x), run(oc.silu_macs_counter_hook, nn.SiLU(), x)])
    out["macs_hook_groupnorm_silu"] = np.asarray(rows, dtype=np.int64)
    rows = []          # [L, C] -> LayerNorm macs
    for L, C in [(4096, 320), (64, 1280)]:
        rows.append([L, C, run(oc.layer_norm_macs_counter_hook, nn.LayerNorm(C), torch.zeros(1, L, C))])
    out["macs_hook_layernorm"] = np.asarray(rows, dtype=np.int64)

    class AttnStandIn(nn.Module):
        """the attributes gated_attention_counter_hook reads (op_counter.py:259-306); leaf MACs come from the linear hook"""

        def __init__(self, C, heads, kv_dim):
            super().__init__()
            self.to_q, self.to_k, self.to_v = nn.Linear(C, C, bias=False), nn.Linear(kv_dim, C, bias=False), nn.Linear(kv_dim, C, bias=False)
            self.to_out = nn.ModuleList([nn.Linear(C, C), nn.Identity()])
            self.heads, self.spatial_norm, self.group_norm, self.norm_cross = heads, None, None, None
            self.total_macs, self.prunable_macs, self.pruned = 0., 0., False
    rows = []          # [Lq, Lkv, C, heads, kv_dim] -> total_macs, prunable_macs
    for Lq, Lkv, C, heads, kv in [(4096, 4096, 320, 5, 320), (4096, 77, 320, 5, 1024), (64, 77, 1280, 20, 1024), (256, 256, 128, 2, 128)]:
        a = AttnStandIn(C, heads, kv)
        xq, xkv = torch.zeros(1, Lq, C), torch.zeros(1, Lkv, kv)
        run(oc.linear_macs_counter_hook, a.to_q, xq)
        run(oc.linear_macs_counter_hook, a.to_k, xkv)
        run(oc.linear_macs_counter_hook, a.to_v, xkv)
        run(oc.linear_macs_counter_hook, a.to_out[0], xq)
        a.__macs__ = 0
        oc.gated_attention_counter_hook(a, (xq,), torch.zeros(1, Lq, C))
        assert a.__macs__ == a.total_macs
        rows.append([Lq, Lkv, C, heads, kv, int(a.total_macs), int(a.prunable_macs)])
    out["macs_hook_gated_attention"] = np.asarray(rows, dtype=np.int64)
    return out


def main():
    from oracle import unet_oracle as O
    out = {}
    eu = install_shims()
    gates = load("ref_gates", "pdm/models/unet/gates.py")
    closs = load("ref_closs", "pdm/losses/contrastive_loss.py")
    rloss = load("ref_rloss", "pdm/losses/resource_loss.py")
    metric = load("ref_metric", "pdm/utils/metric_utils.py")
    hyper = load("ref_hyper", "pdm/models/hypernet/hypernet.py")
    quant = load("ref_quant", "pdm/models/vq/quantizer.py")
    g = torch.Generator().manual_seed(20240612)

    # ---- 1. gates (incl. CFG batch doubling) -------------------------------------------------------------------
    x = torch.randn(4, 64, 3, 5, generator=g)
    gw = torch.rand(2, 32, generator=g)
    wg = gates.WidthGate(32); wg.set_structure_value(gw)
    out["gate_width_x"], out["gate_width_g"], out["gate_width_y"] = x, gw, wg(x)
    xh = torch.randn(4, 5, 7, 64, generator=g)             # [B, heads, L, d]
    gh = torch.rand(2, 5, generator=g)
    hg = gates.WidthGate(5); hg.set_structure_value(gh)
    out["gate_head_x"], out["gate_head_g"], out["gate_head_y"] = xh, gh, hg(xh)
    xl = torch.randn(4, 6, 128, generator=g)
    gl = torch.rand(2, 32, generator=g)
    lg = gates.LinearWidthGate(32); lg.set_structure_value(gl)
    out["gate_linear_x"], out["gate_linear_g"], out["gate_linear_y"] = xl, gl, lg(xl)
    xi, xo = torch.randn(4, 8, 3, 3, generator=g), torch.randn(4, 8, 3, 3, generator=g)
    gd = torch.rand(2, generator=g)
    dg = gates.DepthGate(1); dg.set_structure_value(gd)
    out["gate_depth_in"], out["gate_depth_out"], out["gate_depth_g"], out["gate_depth_y"] = xi, xo, gd, dg((xi, xo))

    # ---- 2. estimation utils ---------------------------------------------------------------------------------------
    v = torch.tensor([[0.5, 0.49999, 0.0, 1.0, 0.75, -0.2, float("nan")]])
    out["hc_in"], out["hc_out"] = v, eu.hard_concrete(v)
    logits = torch.randn(3, 32, generator=g) * 2
    out["gs_logits"] = logits
    out["gs_fixed"] = eu.gumbel_softmax_sample(logits, temperature=0.4, offset=3, fixed_seed=True)
    out["gs_fixed_force"] = eu.gumbel_softmax_sample(logits - 12.0, temperature=0.4, offset=3, force_width_non_zero=True, fixed_seed=True)
    out["igs_logits"] = torch.randn(3, 14, generator=g)
    out["igs_fixed"] = eu.importance_gumbel_softmax_sample(out["igs_logits"], temperature=0.4, offset=3, fixed_seed=True)
    out["gumbel_fixed_5x7"] = eu.sample_gumbel((5, 7), fixed_seed=True)
    torch.manual_seed(77)
    out["gs_global_seed77"] = eu.gumbel_softmax_sample(logits, temperature=0.4, offset=3, force_width_non_zero=True)

    # ---- 3. losses / snr ---------------------------------------------------------------------------------------------
    pe = torch.randn(6, 24, generator=g) * 0.05
    av = torch.rand(6, 40, generator=g)
    out["cl_prompt"], out["cl_arch"] = pe, av
    out["cl_loss"] = closs.ContrastiveLoss(0.03, 0.03)(pe, av)
    out["cl_loss_t1"] = closs.ContrastiveLoss()(pe, av)
    for i, lt in enumerate(["log", "mae", "mse"]):
        rl = rloss.ResourceLoss(p=0.6, loss_type=lt)
        out[f"rl_{lt}"] = torch.stack([rl(torch.tensor(0.45)), rl(torch.tensor(0.8))])

    class Sched:
        pass
    s = Sched()
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2
    s.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    ts = torch.tensor([0, 1, 17, 500, 998, 999])
    out["snr_alphas_cumprod"], out["snr_t"], out["snr"] = s.alphas_cumprod, ts, metric.compute_snr(s, ts)

    # ---- 4. hypernet (SD-2.1 structure, small input dim to keep the fixture small) --------------------------------------
    structure = O.get_structure(O.SD21)
    torch.manual_seed(5)
    hn = hyper.HyperStructure(structure=structure, input_dim=16, wn_flag=False, linear_bias=True)
    with torch.no_grad():
        for p in hn.parameters():
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    z = torch.randn(3, 16, generator=g)
    a = hn(z)
    out["hn_z"], out["hn_out"] = z, a
    for k, t in hn.state_dict().items():
        out["hn_sd/" + k] = t
    sep = hn.transform_structure_vector(a)
    out["hn_n_width"], out["hn_n_depth"] = torch.tensor(len(sep["width"])), torch.tensor(len(sep["depth"]))
    out["hn_sep_w3"], out["hn_sep_w69"], out["hn_sep_d13"] = sep["width"][3], sep["width"][69], sep["depth"][13]
    av2 = torch.rand(2, 1620, generator=g)
    av2[0, 32:64] = 0.1                        # an all-dead width segment to exercise force_width_non_zero
    tav = hyper.HyperStructure.transform_arch_vector(av2, structure, force_width_non_zero=True)
    out["hn_tav_in"], out["hn_tav_w1"] = av2, tav["width"][1]
    torch.manual_seed(123)
    out["hn_random_arch_0p6"] = hyper.HyperStructure.get_random_arch_vector(0.6, structure)
    hn_wn = hyper.HyperStructure(structure={"width": [[4], [3, 3, 8]], "depth": [[0], [1]]}, input_dim=6, wn_flag=True, linear_bias=False)
    zz = torch.randn(2, 6, generator=g)
    out["hnwn_z"], out["hnwn_out"] = zz, hn_wn(zz)
    for k, t in hn_wn.state_dict().items():
        out["hnwn_sd/" + k] = t

    # ---- 5. quantizer -------------------------------------------------------------------------------------------------
    depth_order = [-1, -2, 0, 1, -3, -4, 2, 3, -5, -6, 4, 5, -7, 6]
    torch.manual_seed(9)
    q = quant.StructureVectorQuantizer(n_e=8, structure=structure, temperature=0.4, base=3, depth_order=depth_order,
                                       non_zero_width=True, resource_aware_normalization=False, optimal_transport=True)
    out["q_embedding"] = q.embedding.weight.detach().clone()
    out["q_embedding_gs_init"] = q.embedding_gs.detach().clone()
    out["q_depth_order"] = torch.tensor(q.depth_order)
    out["q_template"] = q.template.clone()
    zq_in = torch.randn(5, 1620, generator=g)
    q.eval()
    out["q_in"] = zq_in
    out["q_gst_eval"] = q.gumbel_sigmoid_trick(zq_in)
    out["q_wdn"] = q.width_depth_normalize(out["q_gst_eval"])
    zq, (_, _, idx) = q(zq_in)
    out["q_eval_zq"], out["q_eval_idx"] = zq, idx
    out["q_cos_idx"] = q.get_cosine_sim_min_encoding_indices(zq_in)
    out["q_codebook_hard"] = q.get_codebook_entry_gumbel_sigmoid(torch.arange(8), hard=True)
    # training mode: Sinkhorn OT with the global host RNG seeded
    q.train()
    torch.manual_seed(31)
    zq_t, (_, _, idx_t) = q(zq_in)
    out["q_train_zq"], out["q_train_idx"] = zq_t.detach(), idx_t
    out["q_train_embedding_gs"] = q.embedding_gs.detach().clone()
    # resource-aware normalisation
    q2 = quant.StructureVectorQuantizer(n_e=4, structure=structure, temperature=0.4, base=3, depth_order=depth_order,
                                        resource_aware_normalization=True)
    pm = [[float(10 + 3 * i + j) for j in range(len(sub))] for i, sub in enumerate(structure["width"])]
    out["q2_prunable_macs_flat"] = torch.tensor([v_ for sub in pm for v_ in sub])
    q2.set_prunable_macs_template([list(s_) for s_ in pm])
    out["q2_macs_template"] = q2.prunable_macs_template.clone()
    q2.eval()
    out["q2_wdn"] = q2.width_depth_normalize(out["q_gst_eval"])

    out.update(macs_hook_vectors())

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()})
    print("wrote", os.path.join(HERE, "reference_vectors.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
