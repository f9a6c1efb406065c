"""Checkpoint I/O in the reference's on-disk layout: diffusers-named safetensors, pruned experts stored
at their sliced shapes with arch_vector.pt beside unet/, router checkpoints (trainer.py:253-313,
unet_2d_conditional.py:2409-2447).  CPU only: no kernel is involved (the oracle checks that the slicing / scatter indices
mean what the pruned forward reads)."""
import os

import pytest
import torch

from oracle import unet_oracle as O
from diffusion_pruning_amd import checkpoint as C
from diffusion_pruning_amd.hypernet import HyperStructure
from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
from diffusion_pruning_amd.unet import UNet2DConditionModelGated, UNet2DConditionModelPruned


def _tiny(cls=UNet2DConditionModelGated, seed=0):
    cfg = O.TINY
    return cfg, cls(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                    cross_attention_dim=cfg.cross_attention_dim).init_synthetic(seed=seed)


def _soft_mask(cfg, keep, seed, ndoff):
    m = O.random_mask(cfg, keep, seed, n_depth_off=ndoff)
    return {"width": [w * 0.9 for w in m["width"]], "depth": [d * 0.9 for d in m["depth"]]}   # 0.9 / 0 like the reference


def _clone(m):
    return {k: [v.clone() for v in vs] for k, vs in m.items()}


def test_gated_roundtrip_diffusers_layout(tmp_path):
    cfg, model = _tiny()
    d = C.save_pretrained(model, str(tmp_path))
    assert sorted(os.listdir(d)) == ["config.json", "diffusion_pytorch_model.safetensors"]
    assert not os.path.exists(tmp_path / "arch_vector.pt")
    back = C.from_pretrained(str(tmp_path))
    assert type(back) is UNet2DConditionModelGated
    a, b = model.state_dict(), back.state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_pruned_state_dict_has_the_reference_sliced_shapes():
    cfg, pm = _tiny(UNet2DConditionModelPruned)
    soft = _soft_mask(cfg, 0.5, 11, 3)
    hard_w = [(w >= 0.5).float()[0] for w in soft["width"]]
    hard_d = [float(d[0] >= 0.5) for d in soft["depth"]]
    pm.prune(_clone(soft))
    sd, full = C.pruned_state_dict(pm), pm.state_dict()
    # walk the modules in structure order and recompute every shape from the gates, as blocks.py's prune() methods do
    wi, di, n_dropped = 0, 0, 0
    for c in pm._containers():
        for b in list(c.resnets) + list(c.attentions):
            name = next(n for n, m in pm.named_modules() if m is b)
            s = b.get_gate_structure()
            gates = hard_w[wi:wi + len(s["width"])]
            wi += len(s["width"])
            depth = 1.0
            if s["depth"] == [1]:
                depth = hard_d[di]
                di += 1
            keys = [k for k in full if k.startswith(name + ".")]
            if depth == 0.0:
                n_dropped += 1
                assert not any(k in sd for k in keys), name           # every sub-module is nn.Identity in the reference
                continue
            if len(gates) == 1:     # resnet: conv1 rows / temb rows / norm2 / conv2 columns
                cg = b.out_channels // 32
                live = int(gates[0].sum()) * cg
                assert sd[name + ".conv1.weight"].shape == (live, full[name + ".conv1.weight"].shape[1], 3, 3)
                assert sd[name + ".conv1.bias"].shape == (live,)
                assert sd[name + ".time_emb_proj.weight"].shape == (live, full[name + ".time_emb_proj.weight"].shape[1])
                assert sd[name + ".norm2.weight"].shape == sd[name + ".norm2.bias"].shape == (live,)
                assert sd[name + ".conv2.weight"].shape == (b.out_channels, live, 3, 3)
                assert sd[name + ".conv2.bias"].shape == (b.out_channels,)
                assert sd[name + ".norm1.weight"].shape == full[name + ".norm1.weight"].shape
            else:                   # transformer: heads of attn1 / attn2, GEGLU chunks
                tb = name + ".transformer_blocks.0."
                C_ = full[name + ".proj_in.weight"].shape[0]
                for an, g in (("attn1", gates[0]), ("attn2", gates[1])):
                    hl = int(g.sum()) * 64
                    for p in ("to_q", "to_k", "to_v"):
                        assert sd[tb + an + f".{p}.weight"].shape == (hl, full[tb + an + f".{p}.weight"].shape[1])
                    assert sd[tb + an + ".to_out.0.weight"].shape == (C_, hl)
                    assert sd[tb + an + ".to_out.0.bias"].shape == (C_,)
                inner = 4 * C_
                kept = int(gates[2].sum()) * (inner // 32)
                assert sd[tb + "ff.net.0.proj.weight"].shape == (2 * kept, C_)
                assert sd[tb + "ff.net.0.proj.bias"].shape == (2 * kept,)
                assert sd[tb + "ff.net.2.weight"].shape == (C_, kept)
    assert wi == len(hard_w) and di == len(hard_d) and n_dropped == 3
    assert sum(v.numel() for v in sd.values()) < sum(v.numel() for v in full.values())


def test_pruned_roundtrip_and_scatter_indices_against_the_oracle(tmp_path):
    """Save a pruned expert, load it into a model with DIFFERENT random masters: only the live entries come from the file,
    and the oracle's pruned forward (which reads exactly those) must give the same output for both parameter sets."""
    cfg, pm = _tiny(UNet2DConditionModelPruned, seed=0)
    soft = _soft_mask(cfg, 0.5, 5, 2)
    pm.prune(_clone(soft))
    C.save_pretrained(pm, str(tmp_path))
    assert os.path.exists(tmp_path / "arch_vector.pt")
    av = torch.load(tmp_path / "arch_vector.pt")
    assert av.shape == (1, sum(w.shape[1] for w in soft["width"]) + len(soft["depth"]))
    # prune() installs the binarised code (hard_concrete's 0.5 threshold), which is what gets recorded
    assert torch.equal(av, (torch.cat([t.reshape(1, -1) for t in soft["width"] + soft["depth"]], dim=1) >= 0.5).float())

    back = C.from_pretrained(str(tmp_path))
    assert type(back) is UNet2DConditionModelPruned and back.semantics == "pruned"
    a, b = C.pruned_state_dict(pm), C.pruned_state_dict(back)
    assert list(a.keys()) == list(b.keys()) and all(torch.equal(a[k], b[k]) for k in a)

    # the masters of `back` are a fresh random init everywhere the file has no data
    other = _tiny(UNet2DConditionModelPruned, seed=123)[1]
    other.prune(_clone(soft))
    from safetensors.torch import load_file
    C.load_pruned_state_dict(other, load_file(str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors")))
    p0 = {k: v.detach().clone() for k, v in pm.state_dict().items()}
    p1 = {k: v.detach().clone() for k, v in other.state_dict().items()}
    assert any(not torch.equal(p0[k], p1[k]) for k in p0)           # dead entries differ ...
    sample, t, ehs = O.synthetic_inputs(cfg, 1, 16, seed=3)
    gates = lambda: O.assign_gates(cfg, _clone(soft))
    y0 = O.unet_forward(p0, cfg, sample, t, ehs, gates(), "pruned")
    y1 = O.unet_forward(p1, cfg, sample, t, ehs, gates(), "pruned")
    assert float((y0 - y1).abs().max()) <= 1e-5 * float(y0.abs().max())   # ... and are never read


def test_unpruned_weights_into_a_pruned_model_and_shape_mismatch(tmp_path):
    cfg, model = _tiny()
    C.save_pretrained(model, str(tmp_path))                            # full-shape weights, no arch_vector.pt
    soft = _soft_mask(cfg, 0.6, 9, 1)
    av = torch.cat([t.reshape(1, -1) for t in soft["width"] + soft["depth"]], dim=1)
    with pytest.raises(FileNotFoundError):
        C.from_pretrained(str(tmp_path), cls=UNet2DConditionModelPruned)
    pm = C.from_pretrained(str(tmp_path), cls=UNet2DConditionModelPruned, arch_vector=av)
    assert all(torch.equal(v, model.state_dict()[k]) for k, v in pm.state_dict().items())
    # a pruned file does not fit another architecture vector: loud error (the reference keeps random weights, quirk Q3)
    C.save_pretrained(pm, str(tmp_path / "expert"))
    other = _soft_mask(cfg, 0.3, 10, 0)
    av2 = torch.cat([t.reshape(1, -1) for t in other["width"] + other["depth"]], dim=1)
    with pytest.raises((ValueError, KeyError)):
        C.from_pretrained(str(tmp_path / "expert"), arch_vector=av2)
    rnd = C.from_pretrained(str(tmp_path), cls=UNet2DConditionModelPruned, random_pruning_ratio=0.5)
    assert rnd.semantics == "pruned"


def test_router_checkpoint_layout(tmp_path):
    cfg, model = _tiny()
    st = model.get_structure()
    hn = HyperStructure(structure=st, input_dim=32)
    q = StructureVectorQuantizer(n_e=4, structure=st)
    C.save_router(str(tmp_path), hn, q)
    for sub in ("hypernet", "quantizer"):
        assert sorted(os.listdir(tmp_path / sub)) == ["config.json", "diffusion_pytorch_model.safetensors"]
    assert torch.equal(torch.load(tmp_path / "quantizer_embeddings.pt"), q.embedding_gs.detach().cpu())
    hn2 = HyperStructure(structure=st, input_dim=32)
    q2 = StructureVectorQuantizer(n_e=4, structure=st)
    C.load_router(str(tmp_path), hn2, q2)
    assert all(torch.equal(v, hn2.state_dict()[k]) for k, v in hn.state_dict().items())
    assert all(torch.equal(v, q2.state_dict()[k]) for k, v in q.state_dict().items())
    hn3 = HyperStructure.from_pretrained(str(tmp_path / "hypernet"))
    assert hn3.config["input_dim"] == 32


def test_modelmixin_style_api_as_the_reference_scripts_use_it(tmp_path):
    """trainer.py:262-265 (model.save_pretrained(<dir>/unet)) and generate_fid_images.py:88-101
    (from_pretrained(sd21, subfolder="unet", arch_vector=...) then load_state_dict(pruned safetensors))."""
    from safetensors.torch import load_file
    cfg, dense = _tiny()
    dense.save_pretrained(str(tmp_path / "sd21" / "unet"))                       # stands in for the SD-2.1 folder
    soft = _soft_mask(cfg, 0.5, 21, 2)
    av = torch.cat([t.reshape(1, -1) for t in soft["width"] + soft["depth"]], dim=1)
    # a fine-tuned expert: different live weights
    expert = UNet2DConditionModelPruned.from_pretrained(str(tmp_path / "sd21"), subfolder="unet", arch_vector=av,
                                                        revision=None, down_block_types=None)
    with torch.no_grad():
        for p_ in expert.parameters():
            p_.add_(0.01)
    expert.save_pretrained(str(tmp_path / "ft" / "unet"))
    assert os.path.exists(tmp_path / "ft" / "arch_vector.pt")
    arch_v = torch.load(tmp_path / "ft" / "arch_vector.pt", map_location="cpu")
    unet = UNet2DConditionModelPruned.from_pretrained(str(tmp_path / "sd21"), subfolder="unet", arch_vector=arch_v)
    sd = load_file(str(tmp_path / "ft" / "unet" / "diffusion_pytorch_model.safetensors"))
    assert any(tuple(v.shape) != tuple(unet.state_dict()[k].shape) for k, v in sd.items())   # sliced shapes on disk
    unet.load_state_dict(sd)
    a, b = C.pruned_state_dict(expert), C.pruned_state_dict(unet)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
