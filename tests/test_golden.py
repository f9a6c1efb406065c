"""Pins the oracle's gate semantics and the product's router / loss modules against vectors produced by the
reference's own code (tests/golden/make_golden.py; data only).  Bit-exact where the arithmetic is elementwise
fp32, 1e-6 where reductions may reassociate."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz"))


def T(k):
    return torch.from_numpy(G[k])


def eq(a, b, tol=0.0):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if tol == 0.0:
        assert torch.equal(torch.nan_to_num(a.float(), nan=-7.0), torch.nan_to_num(b.float(), nan=-7.0))
    else:
        assert torch.allclose(a.float(), b.float(), rtol=tol, atol=tol), float((a.float() - b.float()).abs().max())


# ---- oracle gates vs reference gates.py -------------------------------------------------------------------------------
def test_oracle_gates_match_reference():
    eq(O.width_gate(T("gate_width_x"), T("gate_width_g")), T("gate_width_y"))
    eq(O.width_gate(T("gate_head_x"), T("gate_head_g")), T("gate_head_y"))
    eq(O.linear_width_gate(T("gate_linear_x"), T("gate_linear_g")), T("gate_linear_y"))
    eq(O.depth_gate(T("gate_depth_in"), T("gate_depth_out"), T("gate_depth_g")), T("gate_depth_y"))
    finite = ~torch.isnan(T("hc_in"))      # (the reference propagates NaN through its straight-through arithmetic)
    eq(O.hard(T("hc_in"))[finite], T("hc_out")[finite])


# ---- product estimation utils ------------------------------------------------------------------------------------------
def test_estimation_utils():
    from diffusion_pruning_amd import estimation_utils as eu
    eq(eu.hard_concrete(T("hc_in")), T("hc_out"))
    eq(eu.sample_gumbel((5, 7), fixed_seed=True), T("gumbel_fixed_5x7"))
    lg = T("gs_logits")
    eq(eu.gumbel_softmax_sample(lg, temperature=0.4, offset=3, fixed_seed=True), T("gs_fixed"))
    forced = eu.gumbel_softmax_sample(lg - 12.0, temperature=0.4, offset=3, force_width_non_zero=True, fixed_seed=True)
    eq(forced, T("gs_fixed_force"))
    assert bool((forced[:, 0] >= 0.5).all())          # the all-dead rows were bumped on entry 0
    eq(eu.importance_gumbel_softmax_sample(T("igs_logits"), temperature=0.4, offset=3, fixed_seed=True), T("igs_fixed"), 1e-6)
    torch.manual_seed(77)
    eq(eu.gumbel_softmax_sample(lg, temperature=0.4, offset=3, force_width_non_zero=True), T("gs_global_seed77"))
    x = torch.tensor([0.2, 0.7], requires_grad=True)
    eu.hard_concrete(x).sum().backward()
    eq(x.grad, torch.ones(2))                           # straight-through


def test_losses_and_snr():
    from diffusion_pruning_amd.losses import ContrastiveLoss, ResourceLoss, compute_snr
    eq(ContrastiveLoss(0.03, 0.03)(T("cl_prompt"), T("cl_arch")), T("cl_loss"), 1e-6)
    eq(ContrastiveLoss()(T("cl_prompt"), T("cl_arch")), T("cl_loss_t1"), 1e-6)
    for lt in ["log", "mae", "mse"]:
        rl = ResourceLoss(p=0.6, loss_type=lt)
        eq(torch.stack([rl(torch.tensor(0.45)), rl(torch.tensor(0.8))]), T(f"rl_{lt}"), 1e-7)

    class S:
        alphas_cumprod = T("snr_alphas_cumprod")
    eq(compute_snr(S(), T("snr_t")), T("snr"))


def _sd(prefix):
    return {k[len(prefix):]: T(k) for k in G.files if k.startswith(prefix)}


def test_hyperstructure():
    from diffusion_pruning_amd.hypernet import HyperStructure
    structure = O.get_structure(O.SD21)
    hn = HyperStructure(structure=structure, input_dim=16, wn_flag=False, linear_bias=True)
    assert sum(p.numel() for p in hn.parameters()) == 1620 * 16 + 1620
    hn.load_state_dict(_sd("hn_sd/"))
    out = hn(T("hn_z"))
    eq(out, T("hn_out"), 1e-6)
    sep = hn.transform_structure_vector(out)
    assert len(sep["width"]) == int(G["hn_n_width"]) == 70 and len(sep["depth"]) == int(G["hn_n_depth"]) == 14
    eq(sep["width"][3], T("hn_sep_w3"), 1e-6)
    eq(sep["width"][69], T("hn_sep_w69"), 1e-6)
    eq(sep["depth"][13], T("hn_sep_d13"), 1e-6)
    tav = HyperStructure.transform_arch_vector(T("hn_tav_in"), structure, force_width_non_zero=True)
    eq(tav["width"][1], T("hn_tav_w1"))
    torch.manual_seed(123)
    eq(HyperStructure.get_random_arch_vector(0.6, structure), T("hn_random_arch_0p6"))
    # weight-norm variant: same parameter names, same outputs
    hw = HyperStructure(structure={"width": [[4], [3, 3, 8]], "depth": [[0], [1]]}, input_dim=6, wn_flag=True, linear_bias=False)
    hw.load_state_dict(_sd("hnwn_sd/"))
    eq(hw(T("hnwn_z")), T("hnwn_out"), 1e-6)
    # SD-2.1 production shape: 1,245,780 parameters at input_dim 768 with bias
    full = HyperStructure(structure=structure, input_dim=768, wn_flag=False, linear_bias=True)
    assert sum(p.numel() for p in full.parameters()) == 1_245_780
    w0 = full.mh_fc[0].weight
    eq(w0 @ w0.T, torch.eye(32), 1e-4)                  # orthogonal init (hypernet.py:58-63)


DEPTH_ORDER = [-1, -2, 0, 1, -3, -4, 2, 3, -5, -6, 4, 5, -7, 6]


def _quantizer(**kw):
    from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
    q = StructureVectorQuantizer(n_e=8, structure=O.get_structure(O.SD21), temperature=0.4, base=3,
                                 depth_order=DEPTH_ORDER, non_zero_width=True, resource_aware_normalization=False,
                                 optimal_transport=True, **kw)
    with torch.no_grad():
        q.embedding.weight.copy_(T("q_embedding"))
        q.embedding_gs.copy_(T("q_embedding_gs_init"))
    return q


def test_quantizer_eval_path():
    q = _quantizer()
    assert q.vq_embed_dim == 1620 and q.n_e == 8
    eq(torch.tensor(q.depth_order), T("q_depth_order"))
    eq(q.template, T("q_template"))
    q.eval()
    gst = q.gumbel_sigmoid_trick(T("q_in"))
    eq(gst, T("q_gst_eval"), 1e-6)
    eq(q.width_depth_normalize(T("q_gst_eval")), T("q_wdn"), 1e-6)
    zq, (_, _, idx) = q(T("q_in"))
    eq(idx, T("q_eval_idx"))
    eq(zq, T("q_eval_zq"))
    assert set(zq.unique().tolist()) <= {0.0, 1.0}     # eval returns hard codes (quantizer.py:166-167)
    eq(q.get_cosine_sim_min_encoding_indices(T("q_in")), T("q_cos_idx"))
    eq(q.get_codebook_entry_gumbel_sigmoid(torch.arange(8), hard=True), T("q_codebook_hard"))


def test_quantizer_train_path_sinkhorn():
    q = _quantizer()
    q.train()
    torch.manual_seed(31)
    zq, (_, _, idx) = q(T("q_in"))
    eq(idx, T("q_train_idx"))
    eq(zq.detach(), T("q_train_zq"), 1e-6)
    eq(q.embedding_gs.detach(), T("q_train_embedding_gs"), 1e-6)
    assert zq.requires_grad                            # gradient reaches the codebook through the soft code
    zq.sum().backward()
    assert q.embedding.weight.grad is not None and float(q.embedding.weight.grad.abs().sum()) > 0


def test_quantizer_resource_aware_normalisation():
    from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
    structure = O.get_structure(O.SD21)
    q2 = StructureVectorQuantizer(n_e=4, structure=structure, temperature=0.4, base=3, depth_order=DEPTH_ORDER,
                                  resource_aware_normalization=True)
    flat = T("q2_prunable_macs_flat").tolist()
    pm, i = [], 0
    for sub in structure["width"]:
        pm.append(flat[i:i + len(sub)])
        i += len(sub)
    q2.set_prunable_macs_template(pm)
    eq(q2.prunable_macs_template, T("q2_macs_template"))
    q2.eval()
    eq(q2.width_depth_normalize(T("q_gst_eval")), T("q2_wdn"), 1e-6)
