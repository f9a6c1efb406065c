"""StructureVectorQuantizer — the prompt router / architecture codebook (pdm/models/vq/quantizer.py:12-340).

Routes each prompt's architecture vector to one of ``n_e`` codes (cosine assignment in eval, Sinkhorn optimal
transport in training) and returns the code's gumbel-sigmoid relaxation (soft in training, hard_concrete in eval).
Stays PyTorch on the device.  The two ``dist.all_reduce`` call sites of the distributed Sinkhorn
(quantizer.py:284-285, 290-291) run over RCCL when ``torch.distributed`` is initialised with backend "nccl" on ROCm
(gloo in the CPU tests); ``fused_sinkhorn_allreduce=True`` is the MI355X-first variant: the [K, B_loc] score matrix
is all-gathered ONCE (a single small collective instead of 1 + sinkhorn_iterations latency-bound all-reduces) and
the Sinkhorn iterations run redundantly on every rank — same result up to fp32 summation order.
"""
from __future__ import annotations

import json
import os
from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from .dist_utils import _span
from .estimation_utils import sample_gumbel, sample_gumbel_blocks, gumbel_softmax_sample, hard_concrete, importance_gumbel_softmax_sample


class StructureVectorQuantizer(nn.Module):
    def __init__(self, n_e: int, structure: dict, beta: float = 0.25, remap=None, unknown_index: str = "random",
                 sane_index_shape: bool = True, temperature: float = 0.4, base: int = 2, depth_order: list = None,
                 non_zero_width: bool = True, sinkhorn_epsilon: float = 0.05, sinkhorn_iterations: int = 3,
                 resource_aware_normalization: bool = True, optimal_transport: bool = True,
                 fused_sinkhorn_allreduce: bool = False):
        super().__init__()
        self.config = dict(n_e=n_e, structure=structure, beta=beta, remap=remap, unknown_index=unknown_index,
                           sane_index_shape=sane_index_shape, temperature=temperature, base=base,
                           depth_order=depth_order, non_zero_width=non_zero_width, sinkhorn_epsilon=sinkhorn_epsilon,
                           sinkhorn_iterations=sinkhorn_iterations,
                           resource_aware_normalization=resource_aware_normalization,
                           optimal_transport=optimal_transport)
        if remap is not None:
            raise NotImplementedError("index remapping (taming-transformers legacy) is unused by APTP's configs")
        self.structure = structure
        self.n_e, self.beta = n_e, beta
        self.width_list = [w for sub in structure["width"] for w in sub]
        self.width_list_sum = [sum(sub) for sub in structure["width"]]
        self.depth_list = [d for sub in structure["depth"] for d in sub]
        # one embedding dim per width entry plus one per depth-gated sub-block (quantizer.py:43-48)
        self.vq_embed_dim = sum(self.width_list) + sum(1 for d in structure["depth"] if d == [1])
        edges = [0] + np.cumsum(self.width_list_sum).tolist()
        self.width_intervals = [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)]
        self.depth_indices = (sum(self.width_list) - 1 + np.cumsum(self.depth_list)).tolist()
        n_depth = sum(self.depth_list)
        self.input_depth_order = list(range(n_depth)) if depth_order is None else depth_order
        self.depth_order = [i % n_depth for i in self.input_depth_order]
        seg = torch.tensor(self.width_list + [d for d in self.depth_list if d != 0])
        self.template = (1.0 / torch.repeat_interleave(seg, seg).float()).requires_grad_(False)
        self.prunable_macs_template = None
        self.resource_effect_normalization = True if resource_aware_normalization is None else resource_aware_normalization
        self.embedding = nn.Embedding(n_e, self.vq_embed_dim)
        nn.init.orthogonal_(self.embedding.weight)
        self.embedding_gs = nn.Parameter(self.embedding.weight.detach().clone(), requires_grad=False)
        self.remap = None
        self.re_embed = n_e
        self.sane_index_shape = sane_index_shape
        self.temperature, self.base = temperature, base
        self.non_zero_width = non_zero_width
        self.optimal_transport = optimal_transport
        self.sinkhorn_epsilon, self.sinkhorn_iterations = sinkhorn_epsilon, sinkhorn_iterations
        self.fused_sinkhorn_allreduce = fused_sinkhorn_allreduce

    # ---- forward (quantizer.py:136-169) ---------------------------------------------------------------------------
    def forward(self, z: torch.Tensor) -> Tuple[torch.Tensor, Tuple]:
        z = z.contiguous()
        z_flat = z.view(-1, self.vq_embed_dim)
        if self.training:
            codes = self.gumbel_sigmoid_trick(self.embedding.weight)      # gradient reaches the codebook
            self.embedding_gs.data = codes.detach()
            if self.optimal_transport:
                idx = self.get_optimal_transport_min_encoding_indices(z_flat)
            else:
                idx = self.get_cosine_sim_min_encoding_indices(z_flat)
        else:
            codes = self.embedding_gs.detach()
            idx = self.get_cosine_sim_min_encoding_indices(z_flat)
        z_q = codes[idx].view(z.shape).contiguous()
        if self.sane_index_shape:
            idx = idx.reshape(z_q.shape[0])
        if not self.training:
            z_q = hard_concrete(z_q)
        return z_q, (None, None, idx)

    def get_codebook_entry(self, indices: torch.LongTensor, shape=None) -> torch.Tensor:
        z_q = self.embedding(indices)
        if shape is not None:
            z_q = z_q.view(shape).contiguous()
        return z_q

    def get_codebook_entry_gumbel_sigmoid(self, indices: torch.LongTensor, shape=None, hard: bool = False):
        z_q = self.gumbel_sigmoid_trick(self.get_codebook_entry(indices, shape).contiguous())
        return hard_concrete(z_q) if hard else z_q

    # ---- relaxation (quantizer.py:196-231) ---------------------------------------------------------------------------
    def _transform_width_vector(self, inputs):
        assert inputs.shape[1] == sum(self.width_list)
        out, s = [], 0
        for w in self.width_list:
            out.append(inputs[:, s:s + w])
            s += w
        return out

    def _depth_order_on(self, device):
        cache = self.__dict__.setdefault("_depth_order_cache", {})
        t = cache.get(device)
        if t is None:
            t = cache[device] = torch.as_tensor(list(self.depth_order), dtype=torch.long).to(device)
        return t

    def _seg_maps(self, device):
        """[n_width, n_seg] 0/1 membership of every width entry in its segment and the one-hot of each segment's first
        entry: the per-segment non-zero-width rule (estimation_utils.py:13-31) as two small matmuls instead of 70 slices."""
        m = getattr(self, "_seg_cache", None)
        if m is None or m[0].device != device:
            nw, ns = sum(self.width_list), len(self.width_list)
            member = torch.zeros(nw, ns)
            first = torch.zeros(ns, nw)
            s0 = 0
            for j, w in enumerate(self.width_list):
                member[s0:s0 + w, j] = 1.0
                first[j, s0] = 1.0
                s0 += w
            m = (member.to(device), first.to(device))
            self._seg_cache = m
        return m

    def gumbel_sigmoid_trick(self, z_q: torch.Tensor):
        """quantizer.py:196-213.  The noise is drawn on the HOST generator in the reference's order (depth block first, then
        one draw per width segment: quirk Q6), but moved to the device in ONE copy, and the 70 per-segment relaxations are
        evaluated as one elementwise pass over the whole width block (they are elementwise; only the non-zero-width rule
        looks at a segment as a whole).  Same values as the per-segment loop, ~600 fewer launches per call."""
        nw = sum(self.width_list)
        zw, zd = z_q[:, :nw], z_q[:, nw:]
        fixed = not self.training
        B = z_q.shape[0]
        # depth first, then the width segments: the order fixes the host-RNG stream in training mode
        noise = sample_gumbel_blocks(B, [zd.shape[1]] + list(self.width_list), fixed_seed=fixed, device=z_q.device)
        nd, nwz = noise[:, :zd.shape[1]], noise[:, zd.shape[1]:]
        d_sorted = importance_gumbel_softmax_sample(zd, temperature=self.temperature, offset=self.base, fixed_seed=fixed,
                                                    noise=nd)
        d = torch.zeros_like(d_sorted)
        d[:, self._depth_order_on(d.device)] = d_sorted       # (a device index: a Python list here is a synchronous host->device copy per call)
        w = torch.sigmoid((zw + nwz + self.base) / self.temperature)
        if self.non_zero_width:
            member, first = self._seg_maps(z_q.device)
            alive = (w >= 0.5).to(w.dtype) @ member                    # [B, n_seg] live entries per segment
            w = w + 0.5 * ((alive == 0).to(w.dtype) @ first)          # dead segment: +0.5 on its first entry
        return torch.cat([w, d], dim=1)

    def print_param_stats(self):
        for name, param in self.named_parameters():
            if "weight" in name:
                print(f"{name}: {param.mean()}, {param.std()}")

    # ---- normalisation (quantizer.py:233-261) --------------------------------------------------------------------------
    def width_depth_normalize(self, inputs):
        self.template = self.template.to(inputs.device)
        if self.resource_effect_normalization:
            self.prunable_macs_template = self.prunable_macs_template.to(inputs.device)
        # blocks with a depth gate: width entries times the block's depth entry; everything else: hard_concrete
        # (one gather + one select instead of a slice-assign per block)
        idx, in_block = self._depth_maps(inputs.device)
        out = torch.where(in_block[None, :], inputs * inputs[:, idx], hard_concrete(inputs))
        out = out * torch.sqrt(self.template).detach()
        if self.resource_effect_normalization:
            out = out * self.prunable_macs_template.detach()
        return out

    def _depth_maps(self, device):
        m = getattr(self, "_depth_cache", None)
        if m is None or m[0].device != device:
            idx = torch.zeros(self.vq_embed_dim, dtype=torch.long)
            in_block = torch.zeros(self.vq_embed_dim, dtype=torch.bool)
            for i, has_depth in enumerate(self.depth_list):
                if has_depth != 0:
                    lo, hi = self.width_intervals[i]
                    idx[lo:hi] = int(self.depth_indices[i])
                    in_block[lo:hi] = True
            m = (idx.to(device), in_block.to(device))
            self._depth_cache = m
        return m

    def set_prunable_macs_template(self, prunable_macs_list):
        depth_template = [[sum(prunable_macs_list[i])] for i, d in enumerate(self.depth_list) if d == 1]
        prunable_macs_list += depth_template      # (the reference extends the caller's list in place)
        flat = [v for sub in prunable_macs_list for v in sub]
        self.prunable_macs_template = torch.repeat_interleave(
            torch.tensor(flat), torch.tensor(self.width_list + [1] * len(depth_template)))

    # ---- assignment (quantizer.py:263-340) ----------------------------------------------------------------------------
    def _unit(self, x):
        x = self.width_depth_normalize(x)
        return x / x.norm(dim=-1, keepdim=True)

    @torch.no_grad()
    def get_cosine_sim_min_encoding_indices(self, z: torch.Tensor) -> torch.Tensor:
        u = self._unit(self.gumbel_sigmoid_trick(z))
        v = self._unit(self.embedding_gs)
        return torch.argmax(u @ v.t(), dim=-1)

    @torch.no_grad()
    def _sinkhorn(self, out: torch.Tensor, distributed: bool) -> torch.Tensor:
        Q = torch.exp(out / self.sinkhorn_epsilon).t()        # K x B_local
        world = dist.get_world_size() if distributed else 1
        B, K = Q.shape[1] * world, Q.shape[0]
        total = torch.sum(Q)
        if distributed:
            with _span("all_reduce(sinkhorn)"):
                dist.all_reduce(total)
        Q /= total
        for _ in range(self.sinkhorn_iterations):
            rows = torch.sum(Q, dim=1, keepdim=True)
            if distributed:
                with _span("all_reduce(sinkhorn)"):
                    dist.all_reduce(rows)
            Q /= rows
            Q /= K
            Q /= torch.sum(Q, dim=0, keepdim=True)
            Q /= B
        Q *= B
        return Q.t()

    @torch.no_grad()
    def _sinkhorn_fused(self, out: torch.Tensor) -> torch.Tensor:
        """One all-gather of the local score block, then the global Sinkhorn computed redundantly on every rank."""
        world, rank = dist.get_world_size(), dist.get_rank()
        blocks = [torch.empty_like(out) for _ in range(world)]
        with _span("all_gather(sinkhorn scores)"):
            dist.all_gather(blocks, out.contiguous())
        full = self._sinkhorn(torch.cat(blocks, dim=0), distributed=False)
        n = out.shape[0]
        return full[rank * n:(rank + 1) * n]

    @torch.no_grad()
    def get_optimal_transport_min_encoding_indices(self, a: torch.Tensor) -> torch.Tensor:
        a = self._unit(self.gumbel_sigmoid_trick(a))
        codes = self._unit(self.embedding_gs)
        out = a @ codes.t()
        if dist.is_available() and dist.is_initialized():
            Q = self._sinkhorn_fused(out) if self.fused_sinkhorn_allreduce else self._sinkhorn(out, True)
        else:
            Q = self._sinkhorn(out, False)
        return torch.argmax(Q, dim=-1)

    # ---- checkpoint helpers ----------------------------------------------------------------------------------------------
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
