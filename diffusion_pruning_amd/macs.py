"""MAC accounting of the gated U-Net.

The reference measures per-module MACs with ptflops-style forward hooks on a batch-1 forward
(``count_ops_and_params``, pdm/utils/op_counter.py:354-419) and then combines them in the ``calc_macs`` family
(blocks.py:103-119,144-151,384-416,598-633,879-917,1024-1055,1373-1413,1863-1891,2196-2224,2380-2403,2514-2537,
2700-2718; unet_2d_conditional.py:2124-2181).  Here the per-module constants are computed ANALYTICALLY from the shapes
(``assign_module_macs``), with the hooks' conventions restated below, and the combination rules are kept as they are,
including their differentiability through ``hard_concrete`` (the resource loss back-propagates into the gates).

Hook conventions (batch 1; op_counter.py): conv = k*k*Cin*Cout*Hout*Wout + Cout*Hout*Wout (:89-116);
Linear = numel(input)*out + out (bias) (:60-65); GroupNorm = 2*numel (:73-79); LayerNorm = numel (:82-86);
SiLU = 2*numel (:55-57); GatedAttention = to_q + to_k + to_v + heads*(2*L*L*d + L*L) + to_out with L = the QUERY
length also for cross-attention (quirk Q4, :259-306).
"""
from __future__ import annotations

import torch

from .estimation_utils import hard_concrete


def _conv(cin, cout, k, hw_out):
    return float(k * k * cin * cout * hw_out + cout * hw_out)


def _linear(numel_in, cout, bias=True):
    return float(numel_in * cout + (cout if bias else 0))


def assign_module_macs(model, latent_h: int, latent_w: int = None, text_len: int = 77):
    """Set ``__macs__`` on every leaf the calc_macs family reads, for a batch-1 forward at latent_h x latent_w."""
    from . import unet as U
    latent_w = latent_h if latent_w is None else latent_w
    cfg = model.config
    boc = cfg["block_out_channels"]
    T = boc[0] * 4
    X = cfg["cross_attention_dim"]
    P0 = latent_h * latent_w
    model.conv_in.__macs__ = _conv(cfg["in_channels"], boc[0], 3, P0)
    te = model.time_embedding
    te.linear_1.__macs__ = _linear(boc[0], T)
    te.linear_2.__macs__ = _linear(T, T)
    te.act_macs = 2.0 * T                     # nn.SiLU child of TimestepEmbedding
    model.conv_norm_out.__macs__ = 2.0 * boc[0] * P0
    model.conv_act_macs = 2.0 * boc[0] * P0
    model.conv_out.__macs__ = _conv(boc[0], cfg["out_channels"], 3, P0)

    def resnet(r, P):
        r.norm1.__macs__ = 2.0 * r.in_channels * P
        r.conv1.__macs__ = _conv(r.in_channels, r.out_channels, 3, P)
        r.time_emb_proj.__macs__ = _linear(T, r.out_channels)
        r.norm2.__macs__ = 2.0 * r.out_channels * P
        r.conv2.__macs__ = _conv(r.out_channels, r.out_channels, 3, P)
        if r.conv_shortcut is not None:
            r.conv_shortcut.__macs__ = _conv(r.in_channels, r.out_channels, 1, P)
        r.total_macs, r.prunable_macs = 0.0, 0.0

    def transformer(t, P):
        C = t.in_channels
        t.norm.__macs__ = 2.0 * C * P
        t.proj_in.__macs__ = _linear(P * C, C)
        t.proj_out.__macs__ = _linear(P * C, C)
        tb = t.transformer_blocks[0]
        for n in (tb.norm1, tb.norm2, tb.norm3):
            n.__macs__ = float(P * C)
        for attn in (tb.attn1, tb.attn2):
            kv_numel = text_len * X if attn.is_cross else P * C
            attn.to_q.__macs__ = _linear(P * C, C, bias=False)
            attn.to_k.__macs__ = _linear(kv_numel, C, bias=False)
            attn.to_v.__macs__ = _linear(kv_numel, C, bias=False)
            attn.to_out[0].__macs__ = _linear(P * C, C)
            hd = C // attn.heads
            sdpa = attn.heads * (2.0 * P * P * hd + P * P)
            attn.total_macs = attn.to_q.__macs__ + attn.to_k.__macs__ + attn.to_v.__macs__ + sdpa + attn.to_out[0].__macs__
            attn.prunable_macs = attn.total_macs
        ff = tb.ff
        ff.net[0].proj.__macs__ = _linear(P * C, 8 * C)
        ff.net[2].__macs__ = _linear(P * 4 * C, C)
        ff.total_macs, ff.prunable_macs = 0.0, 0.0
        t.total_macs, t.prunable_macs = 0.0, 0.0
        tb.total_macs, tb.prunable_macs = 0.0, 0.0

    h, w = latent_h, latent_w
    for blk in model.down_blocks:
        for r in blk.resnets:
            resnet(r, h * w)
        for t in blk.attentions:
            transformer(t, h * w)
        if blk.downsamplers is not None:
            h, w = (h + 1) // 2, (w + 1) // 2
            c = blk.downsamplers[0].conv.out_channels
            blk.downsamplers[0].conv.__macs__ = _conv(c, c, 3, h * w)
        blk.total_macs, blk.prunable_macs = 0.0, 0.0
    for r in model.mid_block.resnets:
        resnet(r, h * w)
    for t in model.mid_block.attentions:
        transformer(t, h * w)
    model.mid_block.total_macs, model.mid_block.prunable_macs = 0.0, 0.0
    for blk in model.up_blocks:
        for r in blk.resnets:
            resnet(r, h * w)
        for t in blk.attentions:
            transformer(t, h * w)
        if blk.upsamplers is not None:
            h, w = h * 2, w * 2
            c = blk.upsamplers[0].conv.out_channels
            blk.upsamplers[0].conv.__macs__ = _conv(c, c, 3, h * w)
        blk.total_macs, blk.prunable_macs = 0.0, 0.0
    model._macs_assigned = (latent_h, latent_w, text_len)


def _ratio(gate):
    hard = hard_concrete(gate.gate_f)
    return hard.sum(dim=1, keepdim=True) / hard.shape[1]


def _depth_ratio(depth_gate):
    hard = hard_concrete(depth_gate.gate_f).unsqueeze(1)
    return hard.sum(dim=1, keepdim=True) / hard.shape[1]


def _dict():
    return {"prunable_macs": 0.0, "total_macs": 0.0, "cur_prunable_macs": 0.0, "cur_total_macs": 0.0}


def _acc(out, d):
    for k in out:
        out[k] = out[k] + d[k]


def resnet_calc_macs(r):
    """blocks.py:384-416 / 598-633"""
    if r.total_macs == 0.0 or r.prunable_macs == 0.0:
        r.prunable_macs = r.conv1.__macs__ + r.time_emb_proj.__macs__ + r.norm2.__macs__ + r.conv2.__macs__
        r.total_macs = r.norm1.__macs__ + r.prunable_macs
        if r.conv_shortcut is not None:
            r.total_macs += r.conv_shortcut.__macs__
    ratio = _ratio(r.gate)
    rest = r.total_macs - r.prunable_macs
    if r.depth_gated:
        dr = _depth_ratio(r.depth_gate)
        return {"prunable_macs": r.prunable_macs, "total_macs": r.total_macs,
                "cur_prunable_macs": (ratio * r.prunable_macs + rest) * dr,
                "cur_total_macs": (ratio.detach() * r.prunable_macs + rest) * dr.detach()}
    return {"prunable_macs": r.prunable_macs, "total_macs": r.total_macs,
            "cur_prunable_macs": ratio * r.prunable_macs,
            "cur_total_macs": ratio.detach() * r.prunable_macs + rest}


def attention_calc_macs(a):
    """blocks.py:144-151"""
    assert a.total_macs != 0.0 and a.prunable_macs != 0.0
    ratio = _ratio(a.gate)
    return {"prunable_macs": a.prunable_macs, "total_macs": a.total_macs,
            "cur_prunable_macs": ratio * a.prunable_macs,
            "cur_total_macs": ratio.detach() * a.prunable_macs + (a.total_macs - a.prunable_macs)}


def ff_calc_macs(ff):
    """blocks.py:103-119"""
    if ff.total_macs == 0.0 or ff.prunable_macs == 0.0:
        ff.total_macs = ff.net[0].proj.__macs__ + ff.net[2].__macs__
        ff.prunable_macs = ff.total_macs
    ratio = _ratio(ff.net[0].gate)
    return {"prunable_macs": ff.prunable_macs, "total_macs": ff.total_macs,
            "cur_prunable_macs": ratio * ff.prunable_macs,
            "cur_total_macs": ratio.detach() * ff.prunable_macs + (ff.total_macs - ff.prunable_macs)}


def block_calc_macs(tb):
    """BasicTransformerBlockWidthGated.calc_macs, blocks.py:879-917"""
    out = _dict()
    for n, sub in ((tb.norm1, attention_calc_macs(tb.attn1)), (tb.norm2, attention_calc_macs(tb.attn2))):
        out["total_macs"] += n.__macs__
        out["cur_total_macs"] += n.__macs__
        _acc(out, sub)
    out["total_macs"] += tb.norm3.__macs__
    out["cur_total_macs"] += tb.norm3.__macs__
    if tb.gated_ff:
        _acc(out, ff_calc_macs(tb.ff))
    if tb.total_macs == 0.0:
        tb.total_macs = out["total_macs"]
    if tb.prunable_macs == 0.0:
        tb.prunable_macs = out["prunable_macs"]
    return out


def transformer_calc_macs(t):
    """blocks.py:1024-1055 / 1373-1413"""
    out = _dict()
    for m in (t.norm, t.proj_in):
        out["total_macs"] += m.__macs__
        out["cur_total_macs"] += m.__macs__
    for tb in t.transformer_blocks:
        _acc(out, block_calc_macs(tb))
    out["total_macs"] += t.proj_out.__macs__
    out["cur_total_macs"] += t.proj_out.__macs__
    if t.total_macs == 0.0:
        t.total_macs = out["total_macs"]
    if t.prunable_macs == 0:
        t.prunable_macs = out["prunable_macs"]
    if t.depth_gated:
        dr = _depth_ratio(t.depth_gate)
        out["cur_prunable_macs"] = (out["cur_prunable_macs"] + t.total_macs - t.prunable_macs) * dr
        out["cur_total_macs"] = out["cur_total_macs"] * dr.detach()
    return out


def container_calc_macs(blk):
    """blocks.py:1863-1891, 2196-2224, 2380-2403, 2514-2537, 2700-2718"""
    out = _dict()
    from . import unet as U
    if isinstance(blk, U.UNetMidBlock2DCrossAttnWidthGated):
        for r in blk.resnets:
            _acc(out, resnet_calc_macs(r))
        for t in blk.attentions:
            _acc(out, transformer_calc_macs(t))
    elif len(blk.attentions) > 0:
        for r, t in zip(blk.resnets, blk.attentions):
            _acc(out, resnet_calc_macs(r))
            _acc(out, transformer_calc_macs(t))
    else:
        for r in blk.resnets:
            _acc(out, resnet_calc_macs(r))
    samplers = getattr(blk, "downsamplers", None) or getattr(blk, "upsamplers", None)
    if samplers is not None:
        for s in samplers:
            out["total_macs"] += s.conv.__macs__
            out["cur_total_macs"] += s.conv.__macs__
    if blk.total_macs == 0.0:
        blk.total_macs = out["total_macs"]
    if blk.prunable_macs == 0:
        blk.prunable_macs = out["prunable_macs"]
    return out


def unet_calc_macs(model):
    """unet_2d_conditional.py:2124-2163"""
    assert getattr(model, "_macs_assigned", None), "call count_macs(latent_size) first (trainer.py:1256-1296)"
    out = {"total_macs": 0.0, "prunable_macs": 0.0, "cur_prunable_macs": 0.0, "cur_total_macs": 0.0}
    te = model.time_embedding
    fixed = te.linear_1.__macs__ + te.act_macs + te.linear_2.__macs__ + model.conv_in.__macs__
    out["total_macs"] += fixed
    out["cur_total_macs"] += fixed
    for blk in list(model.down_blocks) + [model.mid_block] + list(model.up_blocks):
        _acc(out, container_calc_macs(blk))
    tail = model.conv_norm_out.__macs__ + model.conv_act_macs + model.conv_out.__macs__
    out["total_macs"] += tail
    out["cur_total_macs"] += tail
    return out


def resnet_prunable(r):
    return [r.prunable_macs]


def transformer_prunable(t):
    out = []
    for tb in t.transformer_blocks:
        out += [tb.attn1.prunable_macs, tb.attn2.prunable_macs]
        if tb.gated_ff:
            out.append(tb.ff.prunable_macs)
    return out


def unet_get_prunable_macs(model):
    """unet_2d_conditional.py:2165-2172: per container, resnets first then attentions (blocks.py:1893-1899)"""
    out = []
    for blk in list(model.down_blocks) + [model.mid_block] + list(model.up_blocks):
        for r in blk.resnets:
            out.append(resnet_prunable(r))
        for t in blk.attentions:
            out.append(transformer_prunable(t))
    return out


def resnet_utilization(r):
    u = hard_concrete(r.gate.gate_f).mean(dim=1)
    return u * hard_concrete(r.depth_gate.gate_f) if r.depth_gated else u


def transformer_utilization(t):
    utils = []
    for tb in t.transformer_blocks:
        a1 = hard_concrete(tb.attn1.gate.gate_f).mean(dim=1)
        a2 = hard_concrete(tb.attn2.gate.gate_f).mean(dim=1)
        tot = tb.attn1.prunable_macs + tb.attn2.prunable_macs + (tb.ff.prunable_macs if tb.gated_ff else 0)
        acc = a1 * tb.attn1.prunable_macs + a2 * tb.attn2.prunable_macs
        if tb.gated_ff:
            acc = acc + hard_concrete(tb.ff.net[0].gate.gate_f).mean(dim=1) * tb.ff.prunable_macs
        utils.append(acc / tot)
    u = torch.stack(utils).mean(dim=0)
    return u * hard_concrete(t.depth_gate.gate_f) if t.depth_gated else u


def unet_get_block_utilization(model):
    """unet_2d_conditional.py:2174-2181 (list of per-container lists; down/up pair resnet,attn; mid: resnets then attns)"""
    from . import unet as U
    out = []
    for blk in list(model.down_blocks) + [model.mid_block] + list(model.up_blocks):
        util = []
        if isinstance(blk, U.UNetMidBlock2DCrossAttnWidthGated) or len(blk.attentions) == 0:
            util += [resnet_utilization(r) for r in blk.resnets]
            util += [transformer_utilization(t) for t in blk.attentions]
        else:
            for r, t in zip(blk.resnets, blk.attentions):
                util.append(resnet_utilization(r))
                util.append(transformer_utilization(t))
        out.append(util)
    return out


class VectorizedMacs:
    """``unet_calc_macs`` as a handful of tensor operations on the ARCHITECTURE VECTOR itself (the [Bg, n_width + n_depth]
    tensor whose column slices are the gates), for the pruning step: the module-walking form above -- the reference's
    structure (blocks.py:384-416 ... unet_2d_conditional.py:2124-2163) -- issues ~15 tiny kernels per resnet and ~40 per
    transformer, forward and again backward, ~2,000 launch-bound kernels per train step.  Same numbers:

        cur_prunable = sum_j P_j r_j D_j + sum_k E_k d_k
        cur_total    = sum_j P_j r_j D_j (detached) + sum over modules of (T - P) D (detached) + ungated rest

    r_j = mean of hard_concrete over width segment j, d_k = hard_concrete of depth entry k, D_j = d_k of the module that owns
    gate j if that module is depth-gated, else 1; P_j the prunable MACs behind gate j, E_k the non-prunable MACs of depth-gated
    module k.  Built from the module constants after ``count_macs`` (same walk order as ``set_structure``)."""

    def __init__(self, model, device=None):
        assert getattr(model, "_macs_assigned", None), "call count_macs(latent_size) first"
        ones = unet_calc_macs_constants(model)
        self.total_macs, self.prunable_macs = ones["total_macs"], ones["prunable_macs"]
        P, dmap, widths, E, rest_nd = [], [], [], [], 0.0
        k = 0
        for blk in list(model.down_blocks) + [model.mid_block] + list(model.up_blocks):
            for b in list(blk.resnets) + list(blk.attentions):
                is_res = hasattr(b, "conv1")
                kk = -1
                if b.depth_gated:
                    kk = k
                    k += 1
                if is_res:
                    subs = [(b.gate.width, b.prunable_macs)]
                    rest = b.total_macs - b.prunable_macs
                else:
                    tb = b.transformer_blocks[0]
                    subs = [(tb.attn1.gate.width, tb.attn1.prunable_macs), (tb.attn2.gate.width, tb.attn2.prunable_macs)]
                    if tb.gated_ff:
                        subs.append((tb.ff.net[0].gate.width, tb.ff.prunable_macs))
                    rest = b.total_macs - b.prunable_macs
                for w, pm in subs:
                    widths.append(w); P.append(pm); dmap.append(kk)
                if kk >= 0:
                    E.append(rest)
                else:
                    rest_nd += rest
        self.n_width, self.n_depth = sum(widths), k
        fixed = self.total_macs - sum(b.total_macs for blk in list(model.down_blocks) + [model.mid_block] + list(model.up_blocks)
                                      for b in list(blk.resnets) + list(blk.attentions))
        self.const_total = float(fixed + rest_nd)            # samplers, head, tail, non-prunable parts of un-depth-gated modules
        member = torch.zeros(self.n_width, len(widths))
        s0 = 0
        for j, w in enumerate(widths):
            member[s0:s0 + w, j] = 1.0 / w
            s0 += w
        dm = torch.tensor(dmap, dtype=torch.long)
        self.member = member
        self.P = torch.tensor(P, dtype=torch.float32)
        self.E = torch.tensor(E, dtype=torch.float32)
        self.has_depth = dm >= 0
        self.didx = dm.clamp(min=0)
        if device is not None:
            self.to(device)

    def to(self, device):
        for n in ("member", "P", "E", "has_depth", "didx"):
            setattr(self, n, getattr(self, n).to(device))
        return self

    def __call__(self, arch: torch.Tensor):
        """arch [Bg, n_width + n_depth] (what HyperStructure.transform_structure_vector splits) -> the dict of unet_calc_macs"""
        assert arch.shape[1] == self.n_width + self.n_depth
        hc = hard_concrete(arch)
        r = hc[:, :self.n_width] @ self.member                                   # [Bg, n_gates]
        d = hc[:, self.n_width:]                                                 # [Bg, n_depth]
        D = torch.where(self.has_depth[None, :], d[:, self.didx], torch.ones_like(r))
        cur_p = ((r * D) @ self.P + d @ self.E).unsqueeze(1)
        Dd = D.detach()
        cur_t = ((r.detach() * Dd) @ self.P + d.detach() @ self.E + self.const_total).unsqueeze(1)
        return {"prunable_macs": self.prunable_macs, "total_macs": self.total_macs,
                "cur_prunable_macs": cur_p, "cur_total_macs": cur_t}


def unet_calc_macs_constants(model):
    """total / prunable MACs of the model (module constants only; makes sure every module's lazily filled totals exist)"""
    with torch.no_grad():
        out = unet_calc_macs(model)
    return {"total_macs": float(out["total_macs"]), "prunable_macs": float(out["prunable_macs"])}
