"""Checkpoint I/O in the reference's on-disk layout.

The reference serialises through diffusers' ``ModelMixin.save_pretrained`` (``pdm/training/trainer.py:253-313``): one directory
per model (``unet/``, ``hypernet/``, ``quantizer/``) holding ``config.json`` + ``diffusion_pytorch_model.safetensors`` with
diffusers parameter names, plus ``quantizer_embeddings.pt`` (``trainer.py:273``) and, for a pruned expert, ``arch_vector.pt`` next
to ``unet/`` (``unet_2d_conditional.py:2412-2421``).  A pruned expert's tensors are stored at their **sliced** shapes, because
``UNet2DConditionModelPruned.from_pretrained`` prunes the freshly built model and then loads the file into it
(``unet_2d_conditional.py:2421-2447``; quirk Q3: un-pruned weights therefore do not fit a pruned model).

This package keeps full-shape fp32 masters and selects live rows / columns when it packs the bf16 kernel operands, so:

* ``pruned_state_dict(model)`` slices the masters exactly as the reference's ``prune()`` methods do
  (``blocks.py:53-67,122-129,154-187,425-465,641-697,1428-1438``) and drops the parameters of depth-0 blocks;
* ``load_pruned_state_dict(model, sd)`` scatters such tensors back into the masters at the live indices (the dead entries are
  never read by a pruned model);
* ``save_pretrained`` / ``from_pretrained`` wrap both in the directory layout above.

Host-side only (safetensors + torch); no kernel is involved.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

from .hypernet import HyperStructure
from .unet import (ResnetBlock2DWidthGated, Transformer2DModelWidthGated, UNet2DConditionModelGated,
                   UNet2DConditionModelPruned, _live_index)

WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
CONFIG_NAME = "config.json"
ARCH_VECTOR_NAME = "arch_vector.pt"
QUANTIZER_EMBEDDINGS_NAME = "quantizer_embeddings.pt"

_DROP = "drop"


def _slicers(model: UNet2DConditionModelGated) -> Dict[str, object]:
    """state-dict key -> (row index | None, column index | None), or _DROP for parameters of depth-0 blocks.
    Keys that are absent keep their full shape."""
    out: Dict[str, object] = {}
    for name, m in model.named_modules():
        if isinstance(m, ResnetBlock2DWidthGated):
            if getattr(m, "dropped", False):
                for k, _ in m.named_parameters():
                    out[f"{name}.{k}"] = _DROP          # blocks.py:645-658: every sub-module becomes nn.Identity
                continue
            mask = m.gate.hard_uniform()
            if mask is None:
                raise ValueError(f"{name}: a pruned checkpoint needs a hard single-row width gate (blocks.py:426)")
            if bool((mask == 1).all()):
                continue
            live = _live_index(mask, m.out_channels // m.groups)
            for k in ("conv1.weight", "conv1.bias", "time_emb_proj.weight", "time_emb_proj.bias", "norm2.weight", "norm2.bias"):
                out[f"{name}.{k}"] = (live, None)
            out[f"{name}.conv2.weight"] = (None, live)
        elif isinstance(m, Transformer2DModelWidthGated):
            if getattr(m, "dropped", False):
                for k, _ in m.named_parameters():
                    out[f"{name}.{k}"] = _DROP          # blocks.py:1428-1438
                continue
            tb = m.transformer_blocks[0]
            for an, attn in (("attn1", tb.attn1), ("attn2", tb.attn2)):
                mask = attn.gate.hard_uniform()
                if mask is None:
                    raise ValueError(f"{name}.{an}: a pruned checkpoint needs a hard single-row head gate")
                if bool((mask == 1).all()):
                    continue
                live = _live_index(mask, 64)
                base = f"{name}.transformer_blocks.0.{an}"
                for k in ("to_q.weight", "to_k.weight", "to_v.weight"):
                    out[f"{base}.{k}"] = (live, None)
                out[f"{base}.to_out.0.weight"] = (None, live)
            geglu = tb.ff.net[0]
            mask = geglu.gate.hard_uniform()
            if mask is None:
                raise ValueError(f"{name}.ff: a pruned checkpoint needs a hard single-row FF gate")
            if not bool((mask == 1).all()):
                live = _live_index(mask, geglu.dim_out // geglu.gate.width)
                both = torch.cat([live, live + geglu.dim_out])      # value half and gate half (blocks.py:55-56)
                base = f"{name}.transformer_blocks.0.ff.net"
                out[f"{base}.0.proj.weight"] = (both, None)
                out[f"{base}.0.proj.bias"] = (both, None)
                out[f"{base}.2.weight"] = (None, live)
    return out


def _slice(t: torch.Tensor, rows, cols) -> torch.Tensor:
    if rows is not None:
        t = t.index_select(0, rows.to(t.device))
    if cols is not None:
        t = t.index_select(1, cols.to(t.device))
    return t.contiguous()


def pruned_state_dict(model: UNet2DConditionModelGated) -> "OrderedDict[str, torch.Tensor]":
    """The state dict the reference's pruned model would hold: sliced shapes, no entries for dropped blocks."""
    sl = _slicers(model)
    out = OrderedDict()
    for k, v in model.state_dict().items():
        s = sl.get(k)
        if s == _DROP:
            continue
        out[k] = v.detach() if s is None else _slice(v.detach(), *s)
    return out


def pruned_shapes(model: UNet2DConditionModelGated) -> Dict[str, Tuple[int, ...]]:
    sl = _slicers(model)
    shapes = {}
    for k, v in model.state_dict().items():
        s = sl.get(k)
        if s == _DROP:
            continue
        shp = list(v.shape)
        if s is not None:
            if s[0] is not None:
                shp[0] = int(s[0].numel())
            if s[1] is not None:
                shp[1] = int(s[1].numel())
        shapes[k] = tuple(shp)
    return shapes


@torch.no_grad()
def load_pruned_state_dict(model: UNet2DConditionModelGated, sd: Dict[str, torch.Tensor], strict: bool = True):
    """Scatter a pruned-shape state dict into the full-shape masters of a model whose structure is already installed
    (``model.prune(arch)``).  Shape mismatches raise (the reference would silently keep random weights, quirk Q3)."""
    sl = _slicers(model)
    own = model.state_dict()
    expected = pruned_shapes(model)
    missing = [k for k in expected if k not in sd]
    unexpected = [k for k in sd if k not in expected]
    if strict and (missing or unexpected):
        raise KeyError(f"pruned checkpoint does not match the installed structure: missing {missing[:4]}… ({len(missing)}), "
                       f"unexpected {unexpected[:4]}… ({len(unexpected)})")
    for k, shp in expected.items():
        if k not in sd:
            continue
        v = sd[k]
        if tuple(v.shape) != shp:
            raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != pruned shape {shp} of the installed architecture vector")
        dst = own[k]
        v = v.to(device=dst.device, dtype=dst.dtype)
        s = sl.get(k)
        if s is None:
            dst.copy_(v)
        else:
            rows, cols = s
            if rows is not None and cols is None:
                dst.index_copy_(0, rows.to(dst.device), v)
            elif rows is None:
                dst.index_copy_(1, cols.to(dst.device), v)
            else:  # pragma: no cover - no parameter is sliced on both axes in this architecture
                tmp = dst.index_select(0, rows.to(dst.device))
                tmp.index_copy_(1, cols.to(dst.device), v)
                dst.index_copy_(0, rows.to(dst.device), tmp)
    model.invalidate_plans()
    return missing, unexpected


def _config_json(model, class_name: str) -> str:
    cfg = {"_class_name": class_name, "_aptp_format": 1}
    for k, v in model.config.items():
        cfg[k] = list(v) if isinstance(v, tuple) else v
    return json.dumps(cfg, indent=2, sort_keys=True)


def save_pretrained(model: UNet2DConditionModelGated, root: str, subfolder: str = "unet",
                    arch_vector: Optional[torch.Tensor] = None) -> str:
    """``<root>/<subfolder>/{config.json, diffusion_pytorch_model.safetensors}``; a pruned model is written at its sliced
    shapes with ``<root>/arch_vector.pt`` beside the folder (what the reference's from_pretrained looks for)."""
    from safetensors.torch import save_file
    d = os.path.join(root, subfolder) if subfolder else root
    os.makedirs(d, exist_ok=True)
    pruned = isinstance(model, UNet2DConditionModelPruned)
    sd = pruned_state_dict(model) if pruned else OrderedDict((k, v.detach()) for k, v in model.state_dict().items())
    save_file({k: v.to("cpu").contiguous() for k, v in sd.items()}, os.path.join(d, WEIGHTS_NAME))
    with open(os.path.join(d, CONFIG_NAME), "w") as f:
        f.write(_config_json(model, type(model).__name__))
    if pruned:
        if arch_vector is None:
            arch_vector = arch_vector_of(model)
        torch.save(arch_vector.detach().cpu(), os.path.join(root, ARCH_VECTOR_NAME))
    return d


def arch_vector_of(model: UNet2DConditionModelGated) -> torch.Tensor:
    """Flat [1, 1620]-style architecture vector (all width gates, then all depth gates: hypernet.py:103-124) of the gates
    currently installed in the model."""
    inst = getattr(model, "_installed_structure", None)
    if inst is None:
        raise ValueError("no structure has been installed in this model (set_structure / prune)")
    parts = [t.detach().float().cpu().reshape(t.shape[0], -1) for t in inst["width"] + inst["depth"]]
    return torch.cat(parts, dim=1)


_CONFIG_KEYS = ("sample_size", "in_channels", "out_channels", "down_block_types", "mid_block_type", "up_block_types",
                "block_out_channels", "layers_per_block", "cross_attention_dim", "attention_head_dim", "norm_num_groups",
                "norm_eps", "gated_ff", "ff_gate_width")


def from_pretrained(root: str, subfolder: Optional[str] = "unet", cls=None, arch_vector: Optional[torch.Tensor] = None,
                    random_pruning_ratio: Optional[float] = None, device=None, **overrides):
    """Build the model named in ``config.json`` (or ``cls``), install the architecture vector for a pruned expert
    (``arch_vector`` argument, else ``<root>/arch_vector.pt``, else a random one at ``random_pruning_ratio``:
    unet_2d_conditional.py:2409-2436) and load the weights."""
    from safetensors.torch import load_file
    d = os.path.join(root, subfolder) if subfolder else root
    with open(os.path.join(d, CONFIG_NAME)) as f:
        cfg = json.load(f)
    name = cfg.pop("_class_name", "UNet2DConditionModelGated")
    cfg.pop("_aptp_format", None)
    cfg = {k: v for k, v in cfg.items() if k in _CONFIG_KEYS}
    cfg.update({k: v for k, v in overrides.items() if k in _CONFIG_KEYS and v is not None})
    if cls is None:
        cls = UNet2DConditionModelPruned if name == "UNet2DConditionModelPruned" else UNet2DConditionModelGated
    model = cls.from_config(cfg)
    sd = load_file(os.path.join(d, WEIGHTS_NAME))
    if issubclass(cls, UNet2DConditionModelPruned):
        if arch_vector is None:
            # the reference looks in subfolder.rsplit('/', 1)[0] (unet_2d_conditional.py:2412-2414); scripts keep the file
            # beside unet/ (generate_fid_images.py:88)
            cands = [os.path.join(root, ARCH_VECTOR_NAME)]
            if subfolder:
                cands.insert(0, os.path.join(root, subfolder.rsplit("/", 1)[0], ARCH_VECTOR_NAME))
            for c in cands:
                if os.path.exists(c):
                    arch_vector = torch.load(c, map_location="cpu")
                    break
        if random_pruning_ratio is not None:
            arch_vector = HyperStructure.get_random_arch_vector(random_pruning_ratio, model.get_structure())
        if arch_vector is None:
            raise FileNotFoundError(f"no {ARCH_VECTOR_NAME} next to {d} and no arch_vector / random_pruning_ratio given")
        model.prune(HyperStructure.transform_arch_vector(arch_vector.float(), model.get_structure()))
        full = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        if all(k in sd and tuple(sd[k].shape) == full[k] for k in full):
            model.load_state_dict(sd)          # an un-pruned checkpoint: slice at pack time (what prune() does after loading)
        else:
            load_pruned_state_dict(model, sd)
    else:
        model.load_state_dict(sd)
    if device is not None:
        model = model.to(device)
    model.eval()
    return model


# ---- router (hyper-network + quantizer) ----------------------------------------------------------------------------
def save_router(root: str, hyper_net, quantizer) -> None:
    """``hypernet/``, ``quantizer/`` and ``quantizer_embeddings.pt`` as trainer.py:266-273 writes them."""
    os.makedirs(root, exist_ok=True)
    hyper_net.save_pretrained(os.path.join(root, "hypernet"))
    quantizer.save_pretrained(os.path.join(root, "quantizer"))
    torch.save(quantizer.embedding_gs.detach().cpu(), os.path.join(root, QUANTIZER_EMBEDDINGS_NAME))


def load_router(root: str, hyper_net, quantizer) -> None:
    """Load ``hypernet/`` and ``quantizer/`` into already-constructed modules (trainer.py:296-305)."""
    hyper_net.load_state_dict(type(hyper_net).from_pretrained(os.path.join(root, "hypernet")).state_dict())
    quantizer.load_state_dict(type(quantizer).from_pretrained(os.path.join(root, "quantizer")).state_dict())
