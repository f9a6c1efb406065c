"""HyperStructure — the architecture predictor (pdm/models/hypernet/hypernet.py:27-153).

70 Linear(768 -> w_i) width heads + one Linear(768 -> 14) depth head (1.25 M parameters).  The math is negligible next
to the U-Net, so it stays PyTorch on the device; what matters is the call signature, the parameter
names (``mh_fc.{i}.weight|bias``) and the arch-vector layout that the gated U-Net's ``set_structure`` consumes.
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.parametrizations import weight_norm

from .estimation_utils import hard_concrete


def _flat(list_of_lists):
    return [v for sub in list_of_lists for v in sub]


class _AliasCat(torch.autograd.Function):
    """cat(parts, dim=0) where the parts already ARE consecutive row blocks of ``flat``: forward returns a view of flat,
    backward hands every part its rows of the incoming gradient (views as well)"""

    @staticmethod
    def forward(ctx, flat, *parts):
        ctx.sizes = [p.shape[0] for p in parts]
        return flat.view_as(flat)

    @staticmethod
    def backward(ctx, grad):
        return (None,) + tuple(grad.split(ctx.sizes, dim=0))


class HyperStructure(nn.Module):
    def __init__(self, structure, input_dim: int = 768, wn_flag: bool = True, linear_bias: bool = False,
                 single_arch_param: bool = False):
        super().__init__()
        self.config = dict(structure=structure, input_dim=input_dim, wn_flag=wn_flag, linear_bias=linear_bias,
                           single_arch_param=single_arch_param)
        self.structure = structure
        self.input_dim, self.linear_bias, self.wn_flag = input_dim, linear_bias, wn_flag
        self.width_list = _flat(structure["width"])
        self.depth_list = _flat(structure["depth"])
        self.single_arch_param = single_arch_param
        total = sum(self.width_list) + sum(self.depth_list)
        if single_arch_param:
            self.arch = nn.Parameter(torch.randn(1, total))
            self.arch_gs = torch.zeros(1, total)
        else:
            heads = [nn.Linear(input_dim, w, bias=linear_bias) for w in self.width_list]
            heads.append(nn.Linear(input_dim, sum(self.depth_list), bias=linear_bias))
            if wn_flag:
                heads = [weight_norm(h) for h in heads]
            self.mh_fc = nn.ModuleList(heads)
            self.initialize_weights()

    def initialize_weights(self):
        for name, param in self.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(param)
            elif "bias" in name:
                nn.init.zeros_(param)

    def print_param_stats(self):
        """hypernet.py:81-84: mean / std of every weight tensor"""
        for name, param in self.named_parameters():
            if "weight" in name:
                print(f"{name}: {param.mean()}, {param.std()}")

    def forward(self, x):
        if self.single_arch_param:
            return self.arch          # one shared architecture for the whole batch (hypernet.py:66-68)
        if self.fuse_heads:
            return self._forward_fused(x)
        x = x.to(self.mh_fc[0].weight.device)
        return torch.cat([head(x) for head in self.mh_fc], dim=1)

    # ---- the 71 heads as ONE GEMM ---------------------------------------------------------------------------------------
    # cat_i(head_i(x)) = x @ cat_i(W_i)^T + cat_i(b_i)  (W_i = g_i * v_i / |v_i|_row under weight norm): the reference's loop
    # is 2-5 launch-bound kernels per head forward and as many backward (~600 per step, a few ms of an otherwise
    # graph-replayed pruning step).  The per-head Parameters stay what they are (names, shapes, state_dict:
    # mh_fc.{i}.weight|bias, or mh_fc.{i}.parametrizations.weight.original{0,1}); their STORAGE is re-homed into one flat
    # buffer per kind, and _AliasCat hands autograd the flat view forward / row slices backward, so neither direction copies.
    fuse_heads = True

    def _head_params(self):
        if self.wn_flag:
            kinds = {"v": [h.parametrizations.weight.original1 for h in self.mh_fc],
                     "g": [h.parametrizations.weight.original0 for h in self.mh_fc]}
        else:
            kinds = {"v": [h.weight for h in self.mh_fc]}
        if self.linear_bias:
            kinds["b"] = [h.bias for h in self.mh_fc]
        return kinds

    def _flat_heads(self):
        kinds = self._head_params()
        flat = self.__dict__.get("_flat")
        if flat is not None and flat.keys() == kinds.keys():
            ok = True
            for k, ps in kinds.items():
                f = flat[k]
                # EVERY head must still alias its rows of the flat buffer (a partial load or a manual re-initialisation can
                # re-assign one head's .data in the middle): 71 integer compares per kind
                base, row_bytes, off = f.data_ptr(), f.stride(0) * f.element_size(), 0
                for q in ps:
                    ok = ok and q.data_ptr() == base + off * row_bytes and q.device == f.device and q.dtype == f.dtype
                    off += q.shape[0]
                ok = ok and off == f.shape[0]
            if ok:
                return flat, kinds
        flat = {}
        with torch.no_grad():                      # first use, or the parameters were moved (.to / .cuda re-allocate them)
            for k, ps in kinds.items():
                f = torch.cat([p.data for p in ps], dim=0)
                for p, c in zip(ps, f.split([p.shape[0] for p in ps], dim=0)):
                    p.data = c
                flat[k] = f
        self.__dict__["_flat"] = flat
        return flat, kinds

    def _forward_fused(self, x):
        flat, kinds = self._flat_heads()
        W = _AliasCat.apply(flat["v"], *kinds["v"])
        if self.wn_flag:
            W = torch._weight_norm(W, _AliasCat.apply(flat["g"], *kinds["g"]), 0)   # parametrizations.weight_norm, dim=0
        b = _AliasCat.apply(flat["b"], *kinds["b"]) if self.linear_bias else None
        return F.linear(x.to(W.device), W, b)

    @staticmethod
    def _split(inputs, width_list, n_depth, force_width_non_zero=False):
        nw = sum(width_list)
        assert inputs.shape[1] == nw + n_depth
        widths, start = [], 0
        for w in width_list:
            seg = inputs[:, start:start + w]
            if force_width_non_zero:
                alive = hard_concrete(seg).sum(dim=1)
                if not bool(alive.all()):
                    seg = seg.clone()
                    seg[alive == 0, 0] = seg[alive == 0, 0] + 0.5
            widths.append(seg)
            start += w
        depths = [inputs[:, nw + i] for i in range(n_depth)]
        return {"width": widths, "depth": depths}

    def transform_structure_vector(self, inputs):
        """[B, 1620] -> {"width": 70 x [B, w], "depth": 14 x [B]} (hypernet.py:86-101)"""
        return self._split(inputs, self.width_list, sum(self.depth_list))

    @classmethod
    def transform_arch_vector(cls, inputs, structure, force_width_non_zero: bool = False):
        """hypernet.py:103-129"""
        return cls._split(inputs, _flat(structure["width"]), sum(_flat(structure["depth"])), force_width_non_zero)

    @classmethod
    def get_random_arch_vector(cls, target_ratio, structure):
        """hypernet.py:131-153: per width segment int(ratio*w) random entries = 0.9, every depth entry = 0.9"""
        parts = []
        for w in _flat(structure["width"]):
            seg = torch.zeros(1, w)
            seg[0, torch.randperm(w)[:int(target_ratio * w)]] = 0.9
            parts.append(seg)
        for _ in range(sum(_flat(structure["depth"]))):
            parts.append(torch.tensor([[0.9]]))
        return torch.cat(parts, dim=1)

    # ---- checkpoint helpers in the layout trainer.py:253-313 writes (config.json + weights) ------------------------
    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.config, f)
        # diffusers 0.23.1 ModelMixin.save_pretrained defaults to safetensors
        from safetensors.torch import save_file
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "diffusion_pytorch_model.safetensors"))

    @classmethod
    def from_pretrained(cls, path: str, **kwargs):
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        cfg.update(kwargs)
        m = cls(**cfg)
        st = os.path.join(path, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            m.load_state_dict(load_file(st))
        else:   # torch-pickled weights (safe_serialization=False)
            m.load_state_dict(torch.load(os.path.join(path, "diffusion_pytorch_model.bin"), map_location="cpu"))
        return m
