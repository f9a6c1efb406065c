"""CPU oracle for APTP's MAC accounting / resource ratio -- TEST INFRASTRUCTURE ONLY.

An independent restatement (table-driven over ``unet_oracle.build_specs``, no module objects) of how the reference
turns architecture codes into MAC figures.  Only ``tests/`` may import it.

Pinned by:
  * the reference's own hook functions, run in the build container on concrete shapes
    (tests/golden/make_golden.py -> ``macs_hook_*`` arrays of tests/golden/reference_vectors.npz): the leaf formulas
    below reproduce them exactly (tests/test_macs_pinned.py);
  * literal hand-expanded expectations for small module configurations and SD-2.1 @ 64x64
    (tests/golden/macs_expected.json).

Reference (file:line relative to /root/reference):
  leaf conventions   pdm/utils/op_counter.py  conv :89-116, Linear :60-65, GroupNorm (bn hook) :73-79,
                     LayerNorm :82-86, SiLU :55-57, GatedAttention :259-306 (quirk Q4: the SDPA term uses the QUERY
                     length for both factors, also for cross-attention)
  resnet             pdm/models/unet/blocks.py:384-416 (width gated), :598-633 (width+depth gated)
  attention          blocks.py:144-151;  feed-forward blocks.py:103-119
  transformer block  blocks.py:879-917;  Transformer2D :1024-1055, depth gated :1373-1413
  containers         blocks.py:1863-1891 (down), :2196-2224 / :2514-2537 (up), :2380-2403 (down, no attention),
                     :2700-2718 (mid)
  U-Net              pdm/models/unet/unet_2d_conditional.py:2124-2163 (calc_macs), :2165-2172 (get_prunable_macs)
  trainer            pdm/training/trainer.py:1256-1306 (count_macs: prunable template, actual target p)
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import unet_oracle as O


# ---- leaf conventions (batch 1) -------------------------------------------------------------------------------------
def conv_macs(cin: int, cout: int, k: int, out_pixels: int, bias: bool = True) -> int:
    return k * k * cin * cout * out_pixels + (cout * out_pixels if bias else 0)


def linear_macs(input_numel: int, cout: int, bias: bool = True) -> int:
    return input_numel * cout + (cout if bias else 0)          # the bias is counted once, not once per row


def groupnorm_macs(numel: int) -> int:
    return 2 * numel


def layernorm_macs(numel: int) -> int:
    return numel


def silu_macs(numel: int) -> int:
    return 2 * numel


def gated_attention_macs(Lq: int, Lkv: int, C: int, heads: int, kv_dim: int) -> int:
    d = C // heads
    proj = linear_macs(Lq * C, C, False) + 2 * linear_macs(Lkv * kv_dim, C, False) + linear_macs(Lq * C, C, True)
    return proj + heads * (Lq * Lq * d + Lq * Lq + Lq * Lq * d)


# ---- per-module constants -------------------------------------------------------------------------------------------
def _levels(cfg: O.UNetConfig, latent: int) -> List[int]:
    side, out = latent, []
    for i in range(cfg.n_levels):
        out.append(side)
        if i != cfg.n_levels - 1:
            side = (side - 1) // 2 + 1                      # 3x3 stride-2 pad-1 convolution
    return out


def _block_side(cfg, b: O.BlockSpec, sides: List[int]) -> int:
    if b.kind == "mid":
        return sides[-1]
    li = int(b.name.split(".")[1])
    return sides[li] if b.kind == "down" else sides[cfg.n_levels - 1 - li]


def module_table(cfg: O.UNetConfig, latent: int, text_len: int = 77) -> Dict[str, Dict[str, int]]:
    """name -> {"total": .., "prunable": ..} for every resnet / attention / feed-forward, plus the un-gated rest"""
    sides = _levels(cfg, latent)
    T, X = cfg.temb_dim, cfg.cross_attention_dim
    tab: Dict[str, Dict[str, int]] = {}
    for b in O.build_specs(cfg):
        P = _block_side(cfg, b, sides) ** 2
        for r in b.resnets:
            prunable = conv_macs(r.cin, r.cout, 3, P) + linear_macs(T, r.cout) + groupnorm_macs(r.cout * P) + conv_macs(r.cout, r.cout, 3, P)
            total = groupnorm_macs(r.cin * P) + prunable + (conv_macs(r.cin, r.cout, 1, P) if r.cin != r.cout else 0)
            tab[r.name] = {"total": total, "prunable": prunable}
        for a in b.attns:
            C = a.ch
            s = gated_attention_macs(P, P, C, a.heads, C)
            c = gated_attention_macs(P, text_len, C, a.heads, X)
            f = linear_macs(P * C, 8 * C) + linear_macs(P * 4 * C, C)
            tab[a.name + ".attn1"] = {"total": s, "prunable": s}
            tab[a.name + ".attn2"] = {"total": c, "prunable": c}
            tab[a.name + ".ff"] = {"total": f, "prunable": f}
            rest = groupnorm_macs(C * P) + 2 * linear_macs(P * C, C) + 3 * layernorm_macs(P * C)
            tab[a.name] = {"total": s + c + f + rest, "prunable": s + c + f}
        if b.sampler:
            side = _block_side(cfg, b, sides)
            out_side = (side - 1) // 2 + 1 if b.kind == "down" else side * 2
            tab[b.sampler] = {"total": conv_macs(b.sampler_ch, b.sampler_ch, 3, out_side ** 2), "prunable": 0}
    c0, P0 = cfg.block_out_channels[0], latent * latent
    tab["_head"] = {"total": linear_macs(c0, T) + silu_macs(T) + linear_macs(T, T) + conv_macs(cfg.in_channels, c0, 3, P0), "prunable": 0}
    tab["_tail"] = {"total": groupnorm_macs(c0 * P0) + silu_macs(c0 * P0) + conv_macs(c0, cfg.out_channels, 3, P0), "prunable": 0}
    return tab


def _keep_ratio(gate: torch.Tensor) -> torch.Tensor:
    """hard_concrete(gate).sum(1, keepdim) / width with the straight-through gradient of estimation_utils.py:67-75"""
    g = gate if gate.dim() == 2 else gate.unsqueeze(1)
    hard = (g >= 0.5).to(g.dtype)
    ste = g + (hard - g).detach()
    return ste.sum(dim=1, keepdim=True) / g.shape[1]


def calc_macs(cfg: O.UNetConfig, latent: int, gates: Dict[str, torch.Tensor], text_len: int = 77) -> Dict[str, object]:
    """The four numbers of UNet2DConditionModelGated.calc_macs for the architecture code in ``gates``
    (``unet_oracle.assign_gates`` layout): total / prunable are ints; cur_* are [Bg, 1] tensors, cur_prunable
    differentiable in the gates."""
    tab = module_table(cfg, latent, text_len)
    total = tab["_head"]["total"] + tab["_tail"]["total"]
    prunable = 0
    cur_p, cur_t = 0.0, float(total)
    for b in O.build_specs(cfg):
        for r in b.resnets:
            t, p = tab[r.name]["total"], tab[r.name]["prunable"]
            ratio = _keep_ratio(gates[r.name + ".gate"])
            if r.depth_gated:
                dr = _keep_ratio(gates[r.name + ".depth_gate"])
                cp, ct = (ratio * p + (t - p)) * dr, (ratio.detach() * p + (t - p)) * dr.detach()
            else:
                cp, ct = ratio * p, ratio.detach() * p + (t - p)
            total, prunable, cur_p, cur_t = total + t, prunable + p, cur_p + cp, cur_t + ct
        for a in b.attns:
            t, p = tab[a.name]["total"], tab[a.name]["prunable"]
            inner_p, inner_t = 0.0, float(t - p)
            for sub in ("attn1", "attn2", "ff"):
                ratio = _keep_ratio(gates[f"{a.name}.{sub}.gate"])
                inner_p = inner_p + ratio * tab[f"{a.name}.{sub}"]["prunable"]
                inner_t = inner_t + ratio.detach() * tab[f"{a.name}.{sub}"]["prunable"]
            if a.depth_gated:
                dr = _keep_ratio(gates[a.name + ".depth_gate"])
                inner_p, inner_t = (inner_p + (t - p)) * dr, inner_t * dr.detach()
            total, prunable, cur_p, cur_t = total + t, prunable + p, cur_p + inner_p, cur_t + inner_t
        if b.sampler:
            total += tab[b.sampler]["total"]
            cur_t = cur_t + tab[b.sampler]["total"]
    return {"total_macs": total, "prunable_macs": prunable, "cur_prunable_macs": cur_p, "cur_total_macs": cur_t}


def prunable_macs_list(cfg: O.UNetConfig, latent: int, text_len: int = 77) -> List[List[int]]:
    """UNet2DConditionModelGated.get_prunable_macs: per container the resnets first, then the transformers
    (each [attn1, attn2, ff]) -- the order of the architecture vector's width entries"""
    tab = module_table(cfg, latent, text_len)
    out: List[List[int]] = []
    for b in O.build_specs(cfg):
        for r in b.resnets:
            out.append([tab[r.name]["prunable"]])
        for a in b.attns:
            out.append([tab[f"{a.name}.{s}"]["prunable"] for s in ("attn1", "attn2", "ff")])
    return out


def actual_target(cfg: O.UNetConfig, latent: int, p: float) -> float:
    """trainer.py:1303-1304 with the all-ones structure installed: p_actual = 1 - (1 - p) * total / cur_prunable(ones)"""
    ones = O.assign_gates(cfg, O.ones_mask(cfg))
    m = calc_macs(cfg, latent, ones)
    return float(1 - (1 - p) * m["total_macs"] / float(torch.as_tensor(m["cur_prunable_macs"]).flatten()[0]))
