"""Gated SD-2.1 U-Net on the HIP path, behind the reference's Python API.

Mirrors (same class names, call signatures, structure plumbing, state-dict keys):
  pdm/models/unet/unet_2d_conditional.py  UNet2DConditionModelGated (:628-2181; forward :1415-1726,
                                          get_structure :1332-1363, set_structure :1365-1413, freeze :2118-2122),
                                          UNet2DConditionModelPruned (:2184-2472)
  pdm/models/unet/blocks.py               ResnetBlock2DWidthGated (:283-465), ResnetBlock2DWidthDepthGated (:468-697),
                                          GatedAttention (:132-187), GEGLUGated/FeedForwardWidthGated (:24-129),
                                          BasicTransformerBlockWidthGated (:700-938), Transformer2DModelWidthGated
                                          (:941-1067), Transformer2DModelWidthDepthGated (:1070-1438), containers
                                          (:1677-1909, 2004-2243, 2290-2416, 2419-2550, 2554-2736)

Differences that matter (all deliberate):
  * Compute never touches torch.nn.functional: every block launches the hand-written gfx950 kernels of
    libaptp_hip.so through ``ops`` (and fails loudly if the library or the GPU is missing).
  * Activations travel as logical [B,C,H,W] bf16 tensors in channels_last memory (= NHWC for the kernels), so the
    NCHW<->token permutes of the reference vanish and block outputs keep the diffusers shapes for forward hooks.
  * A gate is not a separate elementwise pass.  Hard masks shared by the batch compact the weights once per
    ``set_structure`` (dead conv1 output groups / conv2 input groups / heads / FF chunks are never computed; the
    GroupNorm-beta term that distinguishes *gated* from *pruned* semantics is restored exactly by a 9-class border
    correction in the conv2 epilogue).  Soft or per-sample masks run dense with the mask fused into the epilogue of
    the producing GEMM, exactly the reference's arithmetic order.
"""
from __future__ import annotations

import os

import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops
from .gates import DepthGate, LinearWidthGate, WidthGate
from .ops import PackedWeight

# ----------------------------------------------------------------------------------------------------------------
# parameter containers (diffusers names / shapes at the state-dict boundary; never called)
# ----------------------------------------------------------------------------------------------------------------


class _Params(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container; the owning block launches the HIP kernels")


class Conv2dP(_Params):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = cin, cout, (k, k)
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))


class LinearP(_Params):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.in_features, self.out_features = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None


class NormP(_Params):
    def __init__(self, c, eps, groups=None):
        super().__init__()
        self.num_channels, self.eps, self.num_groups = c, eps, groups
        self.weight = nn.Parameter(torch.empty(c))
        self.bias = nn.Parameter(torch.empty(c))


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    """logical NCHW (channels_last memory) -> NHWC view"""
    return x.permute(0, 2, 3, 1)


def _nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


@dataclass
class TembBundle:
    """SiLU(emb) plus the batched time_emb_proj outputs of every resnet (one GEMM per forward)."""
    emb_silu: torch.Tensor                   # bf16 [B, T]
    proj: Dict[int, torch.Tensor] = field(default_factory=dict)   # id(resnet) -> fp32 [B, Npad] view


@dataclass
class CtxBundle:
    """Text states plus the batched cross-attention K/V projections of every attn2 (one GEMM per forward)."""
    ehs: torch.Tensor                        # bf16 [B, 77, X]
    kv: Dict[int, torch.Tensor] = field(default_factory=dict)     # id(attn2) -> bf16 [B, 77, 2*hl*64] view
    key: Any = None                          # identity of the packed-weight plans the projections were made with


def _live_index(mask: torch.Tensor, group: int) -> torch.Tensor:
    """channel indices kept by a {0,1} mask over groups of `group` consecutive channels"""
    g = torch.nonzero(mask > 0.5).flatten()
    return (g[:, None] * group + torch.arange(group)[None, :]).flatten()


class _CatSlot:
    """One skip-concat buffer [B,H,W,Ch+Cs] of the up path (diffusers torch.cat([hidden, skip], dim=1)): the module that
    produces `hidden` and the module that produces `skip` each write their result straight into their channel range, so
    the concat costs no copy.  Allocated by whichever producer runs first (the skip, in the down path)."""

    def __init__(self, c_hidden: int, c_skip: int):
        self.ch, self.cs, self.buf = c_hidden, c_skip, None

    def view(self, which: int, B: int, H: int, W: int, C: int, device) -> Optional[torch.Tensor]:
        if C != (self.ch if which == 0 else self.cs):
            return None
        if self.buf is None:
            self.buf = torch.empty(B, H, W, self.ch + self.cs, dtype=ops.ACT_DTYPE, device=device)
        elif tuple(self.buf.shape[:3]) != (B, H, W) or self.buf.device != device:
            return None
        return self.buf[..., :self.ch] if which == 0 else self.buf[..., self.ch:]


def _ft_mode(mod) -> bool:
    """The differentiable / fine-tuning forward is the one to run: autograd is on, or a PackedTrainer is attached.  Under a
    PackedTrainer the packed tensors ARE the model (the diffusers-layout masters keep their initial values until
    ``export_()`` and may sit in host memory), so a no-grad forward in the middle of training -- the reference FineTuner's
    validation and sample generation, trainer.py:1766-1830 -- must read them too instead of inference packs built from
    the stale masters; the autograd Functions of that path simply run their forward when grad is off."""
    return torch.is_grad_enabled() or mod.__dict__.get("_pk") is not None


def _take_dst(mod, B: int, H: int, W: int, C: int, device) -> Optional[torch.Tensor]:
    """destination view registered for this module's output by the model's forward (inference only), or None"""
    d = mod.__dict__.pop("_out_dst", None)
    if d is None or _ft_mode(mod):
        return None
    return d[0].view(d[1], B, H, W, C, device)


def _rowbias_of(tproj: torch.Tensor, n: int) -> torch.Tensor:
    """[B, n_live] fp32 projection -> the [B, N] (pack-padded) per-sample bias conv_gemm reads"""
    tproj = tproj.float()
    if tproj.shape[1] >= n:
        return tproj[:, :n]
    return torch.nn.functional.pad(tproj, (0, n - tproj.shape[1]))


def _cw(mod, x, weight, bias, pw, get_bwd, residual=None, **kw):
    """weight-gradient convolution of the fine-tuning path: diffusers-layout masters (AG.conv_w) or, under a PackedTrainer
    (packed_train.py), the packed masters; `residual` (the ``+ x`` of a sub-block) is added in the GEMM epilogue"""
    from . import autograd as AG
    pk = mod.__dict__.get("_pk")
    if pk is not None:
        return pk.conv(x, weight, bias, pw, get_bwd, residual=residual, **kw)
    return AG.conv_w(x, weight, bias, pw, get_bwd, residual=residual, **kw)


def _gnw(mod, x, gamma_p, beta_p, gamma, beta, groups, eps, silu, C, live, fork=False):
    """fork=True: returns (y, x_res) -- x_res aliases x for the sub-block's `+ x`, and the norm's backward kernel adds the
    gradient that comes back over it (autograd.py, "Residual forks")"""
    from . import autograd as AG
    pk = mod.__dict__.get("_pk")
    if pk is not None:
        return pk.groupnorm(x, gamma_p, beta_p, gamma, beta, groups, eps, silu, C, live, fork)
    return AG.GroupNormWFn.apply(x, gamma_p, beta_p, gamma, beta, groups, eps, silu, C, live, fork)


def _lnw(mod, x, gamma_p, beta_p, gamma, beta, eps, fork=False):
    from . import autograd as AG
    pk = mod.__dict__.get("_pk")
    if pk is not None:
        return pk.layernorm(x, gamma_p, beta_p, gamma, beta, eps, fork)
    return AG.LayerNormWFn.apply(x, gamma_p, beta_p, gamma, beta, eps, fork)


def _capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _versions(mod: nn.Module) -> tuple:
    """in-place update counters of a module's parameters: optimizer steps, ``param.data.copy_`` and ``load_state_dict``
    all bump them, so packs made from older values can be recognised as stale without any hook on the training loop"""
    if mod.__dict__.get("_pk") is not None:
        return ()      # under a PackedTrainer the diffusers-layout masters are not what is trained: the packs ARE the state
    ps = mod.__dict__.get("_vparams")
    if ps is None:
        ps = mod.__dict__["_vparams"] = list(mod.parameters())      # (dropped by invalidate(): .to() may replace parameters)
    return tuple(p._version for p in ps)


class _PlanCache:
    """Packed-weight plans of one module, keyed by (mask, semantics, device).

    * At most ``cap`` UNPINNED entries; the oldest is evicted.  A plan that is created or looked up while a stream is
      capturing is PINNED: a HIP graph bakes raw pointers to its packs, so it must outlive the graph and is only released
      by ``clear()`` (``invalidate_plans``), however many other masks pass through the module in between.
    * Every entry remembers the parameter versions it was packed from.  A lookup with newer versions is a miss and drops
      the stale entry (a pinned one is parked, never returned again, so the memory a graph points to stays allocated):
      fine-tuning forwards always compute with the current weights, whatever loop drives the optimizer."""

    def __init__(self, cap: int = 4):
        self.cap, self.entries, self.parked = cap, {}, []

    def get(self, key, version):
        e = self.entries.get(key)
        if e is None:
            return None
        if e[0] != version:
            del self.entries[key]
            if e[2]:
                self.parked.append(e[1])
            return None
        if not e[2] and _capturing():
            e[2] = True
        return e[1]

    def put(self, key, version, plan):
        unpinned = [k for k, e in self.entries.items() if not e[2]]
        if len(unpinned) >= self.cap:
            del self.entries[unpinned[0]]
        self.entries[key] = [version, plan, _capturing()]
        return plan

    def clear(self):
        self.entries.clear()
        self.parked.clear()

    def __len__(self):
        return len(self.entries)


# ----------------------------------------------------------------------------------------------------------------
# ResNet blocks
# ----------------------------------------------------------------------------------------------------------------
class ResnetBlock2DWidthGated(nn.Module):
    """blocks.py:283-465.  GN32->SiLU->conv3x3->+time_emb_proj(SiLU(temb))->width gate->GN32->SiLU->conv3x3
    ->(+1x1 shortcut)->+x."""
    depth_gated = False

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, groups: int = 32, eps: float = 1e-5,
                 is_input_concatenated: bool = False, skip_connection_dim: Optional[int] = None):
        super().__init__()
        self.in_channels, self.out_channels, self.groups, self.eps = in_channels, out_channels, groups, eps
        self.norm1 = NormP(in_channels, eps, groups)
        self.conv1 = Conv2dP(in_channels, out_channels, 3)
        self.time_emb_proj = LinearP(temb_channels, out_channels)
        self.norm2 = NormP(out_channels, eps, groups)
        self.conv2 = Conv2dP(out_channels, out_channels, 3)
        self.conv_shortcut = Conv2dP(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.gate = WidthGate(groups)
        self.is_input_concatenated = is_input_concatenated
        self.skip_connection_dim = skip_connection_dim
        self.structure = {"width": [], "depth": []}
        self.prunable_macs, self.total_macs = 0.0, 0.0
        self.pruned = False
        self.dropped = False
        self.semantics = "gated"
        self._plans = _PlanCache()

    # ---- structure plumbing (blocks.py:373-382) -----------------------------------------------------------------
    def get_gate_structure(self):
        if not self.structure["width"]:
            self.structure = {"width": [self.gate.width], "depth": [1 if self.depth_gated else 0]}
        return self.structure

    def set_gate_structure(self, arch_vectors):
        assert len(arch_vectors["depth"]) == (1 if self.depth_gated else 0)
        assert len(arch_vectors["width"]) == 1
        assert arch_vectors["width"][0].shape[1] == self.gate.width
        self.gate.set_structure_value(arch_vectors["width"][0])
        if self.depth_gated:
            self.depth_gate.set_structure_value(arch_vectors["depth"][0])

    def invalidate(self):
        self._plans.clear()
        self.__dict__.pop("_vparams", None)

    # ---- execution plan -----------------------------------------------------------------------------------------
    def _mask_key(self):
        m = self.gate.hard_uniform()
        return None if m is None else tuple(int(v) for v in m.tolist())

    def plan(self, device, force_dense: bool = False) -> dict:
        """Packed (and, for a hard batch-shared mask, compacted) weights for the current gate value.
        force_dense: ignore the mask (full weights) — the autograd path needs the gate as an explicit multiply."""
        key = (None if force_dense else self._mask_key(), self.semantics, str(device))
        version = _versions(self)
        pl = self._plans.get(key, version)
        if pl is not None:
            return pl
        mask = None if force_dense else self.gate.hard_uniform()
        cg = self.out_channels // self.groups
        pl = {"compact": mask is not None and not bool((mask == 1).all())}
        dev = device
        if mask is not None and float(mask.sum()) == 0:
            raise ValueError("width gate with no live group (the reference forbids it: non_zero_width)")
        if pl["compact"]:
            live = _live_index(mask, cg)
            dead = _live_index(1 - mask, cg)
            k_live = int(mask.sum())
        else:
            live, dead, k_live = None, None, self.groups
        c_live = k_live * cg
        pl["k_live"], pl["c_live"], pl["c_pad"] = k_live, c_live, ops.round_up(c_live, 8)
        pl["live"] = None if live is None else live.to(dev)
        pl["w1"] = ops.pack_weight(self.conv1.weight.detach(), self.conv1.bias.detach(), out_idx=live, device=dev)
        # time_emb_proj rows (+bias) are gathered the same way and batched by the model into one GEMM
        wt, bt = _f32(self.time_emb_proj.weight), _f32(self.time_emb_proj.bias)
        if live is not None:
            wt, bt = wt[live.to(wt.device)], bt[live.to(bt.device)]
        pl["temb_w"], pl["temb_b"] = wt, bt
        g2, b2 = _f32(self.norm2.weight), _f32(self.norm2.bias)
        if live is not None:
            g2, b2 = g2[live.to(g2.device)], b2[live.to(b2.device)]
        pl["g2"], pl["b2"] = g2.to(dev), b2.to(dev)
        pl["g1"], pl["b1"] = _f32(self.norm1.weight).to(dev), _f32(self.norm1.bias).to(dev)
        pl["w2"] = ops.pack_weight(self.conv2.weight.detach(), self.conv2.bias.detach(), in_idx=live, device=dev)
        pl["wsc"] = None
        if self.conv_shortcut is not None:
            pl["wsc"] = ops.pack_weight(self.conv_shortcut.weight.detach(), self.conv_shortcut.bias.detach(), device=dev)
        pl["corr"] = None
        if pl["compact"] and self.semantics == "gated":
            # a zeroed group leaves GroupNorm as beta, so conv2 of the *gated* model still sees
            # SiLU(beta_c) on dead channels; restore that term exactly per border class (zero padding removes taps).
            w2 = _f32(self.conv2.weight).to(dev)[:, dead.to(dev)]
            sb = torch.nn.functional.silu(_f32(self.norm2.bias).to(dev)[dead.to(dev)])
            T = torch.einsum("ncyx,c->nyx", w2, sb)                       # [Cout, 3, 3]
            valid = {0: (1, 2), 1: (0, 1, 2), 2: (0, 1)}
            corr = torch.zeros(1, 9, self.out_channels, device=dev)
            for rc in range(3):
                for cc in range(3):
                    corr[0, rc * 3 + cc] = T[:, list(valid[rc])][:, :, list(valid[cc])].sum(dim=(1, 2))
            pl["corr"] = corr.contiguous()
        return self._plans.put(key, version, pl)

    # ---- forward ------------------------------------------------------------------------------------------------
    def _depth_state(self):
        return None, None   # (hard value or None, tensor or None)

    def _needs_autograd(self, x) -> bool:
        if not torch.is_grad_enabled():
            return False
        req = x.requires_grad or self.gate.gate_f.requires_grad
        if self.depth_gated:
            req = req or self.depth_gate.gate_f.requires_grad
        return bool(req)

    def _bwd_pack(self, pl, name, param, dev):
        def get():
            key = name + "_bwd"
            if key not in pl:
                pl[key] = ops.pack_weight_dgrad(param.detach(), device=dev)
            return pl[key]
        return get

    def _forward_train(self, x, temb):
        """Differentiable path (pruning step): same op order, gates as explicit multiplies, HIP forward + backward."""
        from . import autograd as AG
        dev = x.device
        pl = self.plan(dev, force_dense=True)
        B, H, W, Cin = x.shape
        x_in = x[..., :Cin - self.skip_connection_dim] if (self.depth_gated and self.is_input_concatenated) else x
        if self.depth_gated and self.dropped:
            return _nchw(x_in)
        x_res = None
        if pl["wsc"] is None:                         # identity shortcut: the `+ x` of conv2 reads the norm's alias of x
            a1, x_res = AG.GroupNormFn.apply(x, pl["g1"], pl["b1"], self.groups, self.eps, True, True)
        else:
            a1 = AG.GroupNormFn.apply(x, pl["g1"], pl["b1"], self.groups, self.eps, True)
        rowbias = self._temb_rowbias(temb, pl, B)
        if rowbias.shape[1] != pl["w1"].N:            # bundle was built for a compacted plan: project locally
            rowbias = self._temb_rowbias(TembBundle(emb_silu=temb.emb_silu), pl, B)
        h = AG.conv(a1, pl["w1"], self._bwd_pack(pl, "w1", self.conv1.weight, dev), rowbias=rowbias)
        gate = self.gate.gate_f
        if gate.requires_grad or self.gate.hard_uniform() is None or not bool((self.gate.hard_uniform() == 1).all()):
            h = AG.GateFn.apply(h, gate.to(device=dev, dtype=torch.float32))
        a2 = AG.GroupNormFn.apply(h, pl["g2"], pl["b2"], self.groups, self.eps, True)
        sc = x_res if pl["wsc"] is None else AG.conv(x, pl["wsc"], self._bwd_pack(pl, "wsc", self.conv_shortcut.weight if self.conv_shortcut is not None else None, dev), pad=0)
        out = AG.conv(a2, pl["w2"], self._bwd_pack(pl, "w2", self.conv2.weight, dev), residual=sc)
        if self.depth_gated:
            d = self.depth_gate.gate_f
            d_hard, _ = self._depth_state()
            if d.requires_grad or d_hard != 1.0:
                out = AG.depth_lerp(x_in, out, d.to(device=dev, dtype=torch.float32))
        return _nchw(out)

    def _sel_bwd_pack(self, pl, name, param, dev, live_out=None, live_in=None):
        def get():
            key = name + "_bwd"
            if key not in pl:
                w = param.detach()
                if live_out is not None:
                    w = w[live_out]
                if live_in is not None:
                    w = w[:, live_in]
                pl[key] = ops.pack_weight_dgrad(w, device=dev)
            return pl[key]
        return get

    def _forward_ft(self, x, temb):
        """Fine-tuning path (FineTuner.step, trainer.py:1683-1765): weights require grad; a pruned expert runs on its
        compacted packs and parameter gradients are scattered back into the live rows / columns of the masters."""
        from . import autograd as AG
        dev = x.device
        pl = self.plan(dev)
        B, H, W, Cin = x.shape
        x_in = x[..., :Cin - self.skip_connection_dim] if (self.depth_gated and self.is_input_concatenated) else x
        if self.depth_gated and (self.dropped or self._depth_state()[0] == 0.0):
            return _nchw(x_in)
        live = pl["live"]
        if self.conv_shortcut is None:                # identity shortcut: the `+ x` of conv2 reads the norm's alias of x
            a1, sc = _gnw(self, x, self.norm1.weight, self.norm1.bias, pl["g1"], pl["b1"], self.groups, self.eps, True, Cin, None, fork=True)
        else:
            a1, sc = _gnw(self, x, self.norm1.weight, self.norm1.bias, pl["g1"], pl["b1"], self.groups, self.eps, True, Cin, None), x
        if "temb_pw" not in pl:
            pl["temb_pw"] = ops.pack_weight(pl["temb_w"], pl["temb_b"], device=dev)
        emb_silu = temb.emb_silu if isinstance(temb, TembBundle) else torch.nn.functional.silu(temb.float()).to(torch.bfloat16)
        tproj = _cw(self, emb_silu[None], self.time_emb_proj.weight, self.time_emb_proj.bias, pl["temb_pw"],
                          self._sel_bwd_pack(pl, "temb", self.time_emb_proj.weight, dev, live), pad=0, out_f32=True, live_out=live)[0]
        # the projection rides in conv1's epilogue as a per-sample output bias (one rounding, as in the inference path);
        # its gradient is the per-sample column sum of conv1's output gradient
        h = _cw(self, a1, self.conv1.weight, self.conv1.bias, pl["w1"], self._sel_bwd_pack(pl, "w1", self.conv1.weight, dev, live),
                      live_out=live, rowbias=_rowbias_of(tproj, pl["w1"].N))
        a2 = _gnw(self, h, self.norm2.weight, self.norm2.bias, pl["g2"], pl["b2"], pl["k_live"], self.eps, True,
                                   pl["c_live"], live)
        if self.conv_shortcut is not None:
            sc = _cw(self, x, self.conv_shortcut.weight, self.conv_shortcut.bias, pl["wsc"],
                           self._sel_bwd_pack(pl, "wsc", self.conv_shortcut.weight, dev), pad=0)
        out = _cw(self, a2, self.conv2.weight, self.conv2.bias, pl["w2"],
                        self._sel_bwd_pack(pl, "w2", self.conv2.weight, dev, None, live), live_in=live, residual=sc)
        return _nchw(out)

    def forward(self, input_tensor: torch.Tensor, temb, scale: float = 1.0):
        x = _nhwc(input_tensor)
        dst = _take_dst(self, x.shape[0], x.shape[1], x.shape[2], self.out_channels, x.device)
        if _ft_mode(self) and self.conv1.weight.requires_grad:
            return self._forward_ft(x, temb)
        if self._needs_autograd(x):
            return self._forward_train(x, temb)
        dev = x.device
        pl = self.plan(dev)
        B, H, W, Cin = x.shape
        assert Cin == self.in_channels
        x_in = x[..., :Cin - self.skip_connection_dim] if (self.depth_gated and self.is_input_concatenated) else x
        d_hard, d_vec = self._depth_state()
        if self.depth_gated and (self.dropped or d_hard == 0.0):
            return _nchw(x_in)                                            # blocks.py:497-498 / (1-0)*x_in + 0*out
        a1 = ops.groupnorm(x, pl["g1"], pl["b1"], self.groups, self.eps, True)
        rowbias = self._temb_rowbias(temb, pl, B)
        gate_kw = {}
        if not pl["compact"] and self.gate.hard_uniform() is None:
            gate_kw = dict(colgate=self._gate_dev(dev), gate_group=self.out_channels // self.groups)
        # (colstats: the GEMM that produces a GroupNorm input also emits its per-channel statistics; ops.groupnorm finds
        # them with the tensor and skips its own statistics pass on the large maps)
        if not gate_kw and H * W <= ops.GN_REDUCE_MAX_HW:
            # small maps: conv1 is split along K, and the reduce launch of the split applies norm2 + SiLU (blocks.py:350-359)
            a2 = ops.conv_gemm(a1, pl["w1"], rowbias=rowbias, gn=(pl["g2"], pl["b2"], pl["k_live"], self.eps, True, pl["c_live"]))
        else:
            h = ops.conv_gemm(a1, pl["w1"], rowbias=rowbias, colstats=True, **gate_kw)
            a2 = ops.groupnorm(h, pl["g2"], pl["b2"], pl["k_live"], self.eps, True, C=pl["c_live"])
        dkw = {}
        if self.depth_gated and d_hard is None:
            dkw = dict(depth=d_vec, depth_in=x_in)
        if pl["wsc"] is not None and FUSE_SHORTCUT:
            # conv2(a2) + conv_shortcut(x) (blocks.py:362-369) as ONE GEMM: the 1x1 shortcut is one more K-segment of
            # conv2 (K = 9*C_mid + C_in, second operand x), so its output never round-trips through memory
            pw = pl.get("w2sc")
            if pw is None:
                pw = pl["w2sc"] = ops.pack_weight_cat(pl["w2"], self.conv_shortcut.weight.detach(), self.conv_shortcut.bias.detach())
            out = ops.conv_gemm(a2, pw, x2=x, corr=pl["corr"], out=dst, colstats=True, **dkw)
            return _nchw(out)
        sc = x if pl["wsc"] is None else ops.conv_gemm(x, pl["wsc"], pad=0)
        out = ops.conv_gemm(a2, pl["w2"], corr=pl["corr"], residual=sc, out=dst, colstats=True, **dkw)
        return _nchw(out)

    def _gate_dev(self, dev):
        g = self.gate.gate_f
        return g.detach().to(device=dev, dtype=torch.float32).contiguous()

    def _temb_rowbias(self, temb, pl, B):
        if isinstance(temb, TembBundle):
            rb = temb.proj.get(id(self))
            if rb is not None:
                return rb
            emb_silu = temb.emb_silu
        else:
            # standalone use with a raw [B, T] embedding: SiLU then this block's own projection
            emb_silu = torch.nn.functional.silu(temb.float()).to(ops.ACT_DTYPE)
        pw = pl.get("temb_pw")
        if pw is None:
            pw = ops.pack_weight(pl["temb_w"], pl["temb_b"], device=emb_silu.device)
            pl["temb_pw"] = pw
        return ops.linear(emb_silu[None], pw, out_f32=True)[0]


class ResnetBlock2DWidthDepthGated(ResnetBlock2DWidthGated):
    """blocks.py:468-697: adds the depth gate (1-d)*x_in + d*out and the skip-concat un-slicing for up blocks."""
    depth_gated = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.depth_gate = DepthGate(1)
        if self.is_input_concatenated:
            assert self.skip_connection_dim is not None

    def _depth_state(self):
        h = self.depth_gate.host_value().flatten()
        if bool(((h == 0) | (h == 1)).all()) and bool((h == h[0]).all()):
            return float(h[0]), None
        return None, self.depth_gate.gate_f.detach().flatten().to(dtype=torch.float32)


# ----------------------------------------------------------------------------------------------------------------
# attention / feed-forward parameter holders with the reference's module names
# ----------------------------------------------------------------------------------------------------------------
class GatedAttention(nn.Module):
    """blocks.py:132-187 (+ HeadGatedAttnProcessor2 :190-280): parameters and the per-head gate."""

    def __init__(self, query_dim: int, heads: int, dim_head: int = 64, cross_attention_dim: Optional[int] = None):
        super().__init__()
        assert dim_head == 64, "the HIP attention kernel is specialised for head_dim 64 (SD-2.1)"
        inner = heads * dim_head
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.inner_dim, self.is_cross = heads, inner, cross_attention_dim is not None
        self.to_q = LinearP(query_dim, inner, bias=False)
        self.to_k = LinearP(kv_dim, inner, bias=False)
        self.to_v = LinearP(kv_dim, inner, bias=False)
        self.to_out = nn.ModuleList([LinearP(inner, query_dim), nn.Identity()])
        self.gate = WidthGate(heads)
        self.prunable_macs, self.total_macs = 0.0, 0.0
        self.pruned = False


class GEGLUGated(nn.Module):
    """blocks.py:24-67"""

    def __init__(self, dim_in: int, dim_out: int, gate_width: int = 32):
        super().__init__()
        self.dim_out = dim_out
        self.proj = LinearP(dim_in, dim_out * 2)
        self.gate = LinearWidthGate(gate_width)
        self.pruned = False


class FeedForwardWidthGated(nn.Module):
    """blocks.py:70-129: net = [GEGLUGated, Dropout, Linear]"""

    def __init__(self, dim: int, mult: int = 4, gate_width: int = 32):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLUGated(dim, inner, gate_width), nn.Identity(), LinearP(inner, dim)])
        self.prunable_macs, self.total_macs = 0.0, 0.0


class BasicTransformerBlockWidthGated(nn.Module):
    """blocks.py:700-938 (parameters + structure plumbing; executed by the owning Transformer2DModel*)."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, cross_attention_dim: int,
                 gated_ff: bool = True, ff_gate_width: int = 32):
        super().__init__()
        self.norm1 = NormP(dim, 1e-5)
        self.attn1 = GatedAttention(dim, num_attention_heads, attention_head_dim, None)
        self.norm2 = NormP(dim, 1e-5)
        self.attn2 = GatedAttention(dim, num_attention_heads, attention_head_dim, cross_attention_dim)
        self.norm3 = NormP(dim, 1e-5)
        self.gated_ff = gated_ff
        self.ff = FeedForwardWidthGated(dim, gate_width=ff_gate_width)
        self.structure = {"width": [], "depth": []}

    def get_gate_structure(self):
        if len(self.structure["width"]) == 0:
            self.structure["width"] = [self.attn1.gate.width, self.attn2.gate.width]
            if self.gated_ff:
                self.structure["width"].append(self.ff.net[0].gate.width)
            self.structure["depth"] = [0]
        return self.structure

    def set_gate_structure(self, arch_vectors):
        assert len(arch_vectors["depth"]) == 0
        assert len(arch_vectors["width"]) >= 2
        assert arch_vectors["width"][0].shape[1] == self.attn1.gate.width
        self.attn1.gate.set_structure_value(arch_vectors["width"][0])
        assert arch_vectors["width"][1].shape[1] == self.attn2.gate.width
        self.attn2.gate.set_structure_value(arch_vectors["width"][1])
        if self.gated_ff:
            assert len(arch_vectors["width"]) == 3
            assert arch_vectors["width"][2].shape[1] == self.ff.net[0].gate.width
            self.ff.net[0].gate.set_structure_value(arch_vectors["width"][2])


class Transformer2DModelWidthGated(nn.Module):
    """blocks.py:941-1067 (forward inherited from diffusers Transformer2DModel, use_linear_projection=True):
    GN(32, eps 1e-6) -> proj_in -> [LN->self-attn->+res; LN->cross-attn->+res; LN->GEGLU FF->+res] -> proj_out -> +x."""
    depth_gated = False

    def __init__(self, num_attention_heads: int, attention_head_dim: int, in_channels: int, cross_attention_dim: int,
                 norm_num_groups: int = 32, gated_ff: bool = True, ff_gate_width: int = 32):
        super().__init__()
        C = num_attention_heads * attention_head_dim
        assert C == in_channels
        self.in_channels, self.heads, self.groups = in_channels, num_attention_heads, norm_num_groups
        self.norm = NormP(in_channels, 1e-6, norm_num_groups)
        self.proj_in = LinearP(in_channels, C)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlockWidthGated(C, num_attention_heads, attention_head_dim, cross_attention_dim,
                                            gated_ff, ff_gate_width)])
        self.proj_out = LinearP(C, in_channels)
        self.structure = {"width": [], "depth": []}
        self.prunable_macs, self.total_macs = 0.0, 0.0
        self.pruned = False
        self.dropped = False
        self._plans = _PlanCache()

    # ---- structure plumbing (blocks.py:1007-1022, 1357-1371) ------------------------------------------------------
    def get_gate_structure(self):
        if len(self.structure["width"]) == 0:
            for tb in self.transformer_blocks:
                self.structure["width"] = self.structure["width"] + tb.get_gate_structure()["width"]
            self.structure["depth"] = [1] if self.depth_gated else [0]
        return self.structure

    def set_gate_structure(self, arch_vectors):
        assert len(arch_vectors["depth"]) == (1 if self.depth_gated else 0)
        if self.depth_gated:
            self.depth_gate.set_structure_value(arch_vectors["depth"][0])
        self.transformer_blocks[0].set_gate_structure({"width": arch_vectors["width"], "depth": []})

    def invalidate(self):
        self._plans.clear()
        self.__dict__.pop("_vparams", None)

    # ---- execution plan -------------------------------------------------------------------------------------------
    def _keys(self):
        tb = self.transformer_blocks[0]
        out = []
        for gate in (tb.attn1.gate, tb.attn2.gate, tb.ff.net[0].gate):
            m = gate.hard_uniform()
            out.append(None if m is None else tuple(int(v) for v in m.tolist()))
        return tuple(out)

    def plan(self, device, force_dense: bool = False) -> dict:
        key = ((None, None, None) if force_dense else self._keys(), str(device))
        version = _versions(self)
        pl = self._plans.get(key, version)
        if pl is not None:
            return pl
        tb = self.transformer_blocks[0]
        dev = device
        pl = {}
        pl["gn_g"], pl["gn_b"] = _f32(self.norm.weight).to(dev), _f32(self.norm.bias).to(dev)
        pl["proj_in"] = ops.pack_weight(self.proj_in.weight.detach(), self.proj_in.bias.detach(), device=dev)
        pl["proj_out"] = ops.pack_weight(self.proj_out.weight.detach(), self.proj_out.bias.detach(), device=dev)
        for i, n in enumerate((tb.norm1, tb.norm2, tb.norm3)):
            pl[f"ln{i + 1}_g"], pl[f"ln{i + 1}_b"] = _f32(n.weight).to(dev), _f32(n.bias).to(dev)
        for name, attn in (("a1", tb.attn1), ("a2", tb.attn2)):
            mask = None if force_dense else attn.gate.hard_uniform()
            compact = mask is not None and not bool((mask == 1).all())
            if mask is not None and float(mask.sum()) == 0:
                raise ValueError("head gate with no live head (the reference forbids it: non_zero_width)")
            live = _live_index(mask, 64) if compact else None
            hl = int(mask.sum()) if compact else attn.heads
            pl[name + "_heads"], pl[name + "_compact"] = hl, compact
            pl[name + "_dense_gate"] = mask is None
            wq, wk, wv = (_f32(t.weight) for t in (attn.to_q, attn.to_k, attn.to_v))
            if live is not None:
                wq, wk, wv = wq[live], wk[live], wv[live]
            if not attn.is_cross:
                pl[name + "_qkv"] = ops.pack_weight(torch.cat([wq, wk, wv], 0), None, device=dev)
                pl["a1_qkv_ln_make"] = (lambda attn=attn, live=live: ops.pack_weight(
                    torch.cat([_f32(t.weight) if live is None else _f32(t.weight)[live] for t in (attn.to_q, attn.to_k, attn.to_v)], 0),
                    None, device=dev, ln_gamma=pl["ln1_g"], ln_beta=pl["ln1_b"]))
            else:
                pl[name + "_q"] = ops.pack_weight(wq, None, device=dev)
                pl["a2_q_ln_make"] = (lambda attn=attn, live=live: ops.pack_weight(
                    _f32(attn.to_q.weight) if live is None else _f32(attn.to_q.weight)[live], None, device=dev,
                    ln_gamma=pl["ln2_g"], ln_beta=pl["ln2_b"]))
                pl[name + "_kv_w"] = torch.cat([wk, wv], 0)          # batched across layers by the model
            pl[name + "_o"] = ops.pack_weight(attn.to_out[0].weight.detach(), attn.to_out[0].bias.detach(),
                                              in_idx=live, device=dev)
        geglu, lin2 = tb.ff.net[0], tb.ff.net[2]
        mask = None if force_dense else geglu.gate.hard_uniform()
        compact = mask is not None and not bool((mask == 1).all())
        if mask is not None and float(mask.sum()) == 0:
            raise ValueError("FF gate with no live chunk (the reference forbids it: non_zero_width)")
        chunk = geglu.dim_out // geglu.gate.width
        live = _live_index(mask, chunk) if compact else None
        pl["ff_compact"], pl["ff_dense_gate"] = compact, mask is None
        pl["ff1"] = ops.pack_weight(geglu.proj.weight.detach(), geglu.proj.bias.detach(), out_idx=live, geglu=True, device=dev)
        pl["ff1_ln_make"] = (lambda geglu=geglu, live=live: ops.pack_weight(
            geglu.proj.weight.detach(), geglu.proj.bias.detach(), out_idx=live, geglu=True, device=dev,
            ln_gamma=pl["ln3_g"], ln_beta=pl["ln3_b"]))
        inner_pad = pl["ff1"].N // 2
        pl["ff2"] = ops.pack_weight(lin2.weight.detach(), lin2.bias.detach(), in_idx=live, cin_pad_to=16, device=dev)
        assert pl["ff2"].Cin == inner_pad, (pl["ff2"].Cin, inner_pad)
        return self._plans.put(key, version, pl)

    def _depth_state(self):
        return None, None

    @staticmethod
    def _gate(gate, dev, rep: int = 1):
        g = gate.gate_f.detach().to(device=dev, dtype=torch.float32)
        if rep > 1:
            g = g.repeat(1, rep)
        return g.contiguous()

    # ---- differentiable path ---------------------------------------------------------------------------------------
    def _needs_autograd(self, x) -> bool:
        if not torch.is_grad_enabled():
            return False
        tb = self.transformer_blocks[0]
        req = x.requires_grad or tb.attn1.gate.gate_f.requires_grad or tb.attn2.gate.gate_f.requires_grad \
            or tb.ff.net[0].gate.gate_f.requires_grad
        if self.depth_gated:
            req = req or self.depth_gate.gate_f.requires_grad
        return bool(req)

    def _forward_train(self, x, encoder_hidden_states):
        from . import autograd as AG
        dev = x.device
        pl = self.plan(dev, force_dense=True)
        tb = self.transformer_blocks[0]
        B, H, W, C = x.shape
        P = H * W

        def bwd(name, make_w):
            def get():
                key = name + "_bwd"
                if key not in pl:
                    pl[key] = ops.pack_weight_dgrad(make_w().detach(), device=dev)
                return pl[key]
            return get

        def gated(y0, gate, rep=1):
            g = gate.gate_f
            hu = gate.hard_uniform()
            if not g.requires_grad and hu is not None and bool((hu == 1).all()):
                return y0
            g = g.to(device=dev, dtype=torch.float32)
            return AG.GateFn.apply(y0, g.repeat(1, rep) if rep > 1 else g)

        # (residual forks: each norm returns an alias of its input for the sub-block's `+ x`; see autograd.py)
        a, x_res = AG.GroupNormFn.apply(x, pl["gn_g"], pl["gn_b"], self.groups, 1e-6, False, True)
        tok = a.reshape(B, P, C)
        x_tok = x_res.reshape(B, P, C)
        h = AG.conv(tok, pl["proj_in"], bwd("proj_in", lambda: self.proj_in.weight), pad=0)
        # self attention
        n, h = AG.LayerNormFn.apply(h, pl["ln1_g"], pl["ln1_b"], 1e-5, True)
        a1 = tb.attn1
        qkv = AG.conv(n, pl["a1_qkv"], bwd("a1_qkv", lambda: torch.cat([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight], 0)), pad=0)
        qkv = gated(qkv, a1.gate, 3)
        o = AG.SelfAttnFn.apply(qkv, a1.heads)
        h = AG.conv(o, pl["a1_o"], bwd("a1_o", lambda: a1.to_out[0].weight), pad=0, residual=h)
        # cross attention
        n, h = AG.LayerNormFn.apply(h, pl["ln2_g"], pl["ln2_b"], 1e-5, True)
        a2 = tb.attn2
        q = gated(AG.conv(n, pl["a2_q"], bwd("a2_q", lambda: a2.to_q.weight), pad=0), a2.gate)
        ehs = encoder_hidden_states.ehs if isinstance(encoder_hidden_states, CtxBundle) else \
            encoder_hidden_states.to(device=dev, dtype=torch.bfloat16)
        if "a2_kv" not in pl:
            pl["a2_kv"] = ops.pack_weight(pl["a2_kv_w"], None, device=dev)
        kv = gated(AG.conv(ehs, pl["a2_kv"], None, pad=0), a2.gate, 2)
        o = AG.CrossAttnFn.apply(q, kv, a2.heads)
        h = AG.conv(o, pl["a2_o"], bwd("a2_o", lambda: a2.to_out[0].weight), pad=0, residual=h)
        # feed-forward (GEGLU in its un-interleaved training form)
        n, h = AG.LayerNormFn.apply(h, pl["ln3_g"], pl["ln3_b"], 1e-5, True)
        geglu, lin2 = tb.ff.net[0], tb.ff.net[2]
        if "ff1_plain" not in pl:
            pl["ff1_plain"] = ops.pack_weight(geglu.proj.weight.detach(), geglu.proj.bias.detach(), device=dev)
        hg = AG.conv(n, pl["ff1_plain"], bwd("ff1_plain", lambda: geglu.proj.weight), pad=0)
        fg = geglu.gate.gate_f
        hu = geglu.gate.hard_uniform()
        ffgate = None if (not fg.requires_grad and hu is not None and bool((hu == 1).all())) else fg.to(device=dev, dtype=torch.float32)
        f = AG.GegluFn.apply(hg, ffgate)
        h = AG.conv(f, pl["ff2"], bwd("ff2", lambda: lin2.weight), pad=0, residual=h)
        out = AG.conv(h, pl["proj_out"], bwd("proj_out", lambda: self.proj_out.weight), pad=0, residual=x_tok)
        if self.depth_gated:
            d = self.depth_gate.gate_f
            d_hard, _ = self._depth_state()
            if d.requires_grad or d_hard != 1.0:
                out = AG.depth_lerp(x_tok, out, d.to(device=dev, dtype=torch.float32))
        return _nchw(out.reshape(B, H, W, C))

    def _forward_ft(self, x, encoder_hidden_states):
        """Fine-tuning path: trainable projections / norms, head- and FF-chunk compaction of a pruned expert."""
        from . import autograd as AG
        dev = x.device
        pl = self.plan(dev)
        tb = self.transformer_blocks[0]
        B, H, W, C = x.shape
        P = H * W

        def bwd(name, param, lo=None, li=None):
            def get():
                key = name + "_ftbwd"
                if key not in pl:
                    w = param.detach()
                    if lo is not None:
                        w = w[lo]
                    if li is not None:
                        w = w[:, li]
                    pl[key] = ops.pack_weight_dgrad(w, device=dev)
                return pl[key]
            return get

        def pack(name, param, bias=None, lo=None, li=None, cin_pad_to=8):
            key = name + "_ft"
            if key not in pl:
                pl[key] = ops.pack_weight(param.detach(), None if bias is None else bias.detach(), out_idx=lo, in_idx=li,
                                          cin_pad_to=cin_pad_to, device=dev)
            return pl[key]

        def lin(xx, name, mod, lo=None, li=None, residual=None):
            b = mod.bias
            return _cw(self, xx, mod.weight, b, pack(name, mod.weight, b, lo, li), bwd(name, mod.weight, lo, li), pad=0,
                             live_out=lo, live_in=li, residual=residual)

        def lin_cat(xx, name, mods, lo):
            """projections that share their input (to_q | to_k | to_v) as ONE contraction under a PackedTrainer: one forward,
            one data-gradient and one weight-gradient launch instead of three each, no concatenation copy, no fan-in adds"""
            ws = tuple(m.weight for m in mods)
            key = name + "_ft"
            if key not in pl:
                pl[key] = ops.pack_weight(torch.cat([w.detach() if lo is None else w.detach()[lo] for w in ws], 0), None, device=dev)
            return _cw(self, xx, ws, None, pl[key], None, pad=0, live_out=lo)

        fuse_qkv = self.__dict__.get("_pk") is not None and FT_FUSE_QKV and all(
            m.bias is None for m in (tb.attn1.to_q, tb.attn1.to_k, tb.attn1.to_v, tb.attn2.to_k, tb.attn2.to_v))

        def live_of(gate, width):
            # (cached in the plan: the index tensor is uploaded once, never while a stream is capturing)
            key = ("live", id(gate))
            ent = pl.get(key)
            if ent is None:
                m = gate.hard_uniform()
                if m is None or bool((m == 1).all()):
                    ent = (None, None)
                else:
                    ent = (_live_index(m, width).to(dev), int(m.sum()))
                pl[key] = ent
            return ent

        # (residual forks: each norm returns an alias of its input for the sub-block's `+ x`; see autograd.py)
        a, x_res = _gnw(self, x, self.norm.weight, self.norm.bias, pl["gn_g"], pl["gn_b"], self.groups, 1e-6, False, C, None, fork=True)
        tok = a.reshape(B, P, C)
        x_tok = x_res.reshape(B, P, C)
        h = lin(tok, "proj_in", self.proj_in)
        # self attention
        n, h = _lnw(self, h, tb.norm1.weight, tb.norm1.bias, pl["ln1_g"], pl["ln1_b"], 1e-5, fork=True)
        a1 = tb.attn1
        l1, h1 = live_of(a1.gate, 64)
        h1 = a1.heads if h1 is None else h1
        if fuse_qkv:
            qkv = lin_cat(n, "a1qkv", (a1.to_q, a1.to_k, a1.to_v), l1)
        else:
            qkv = torch.cat([lin(n, "a1q", a1.to_q, l1), lin(n, "a1k", a1.to_k, l1), lin(n, "a1v", a1.to_v, l1)], dim=-1)
        o = AG.SelfAttnFn.apply(qkv, h1)
        h = lin(o, "a1o", a1.to_out[0], None, l1, residual=h)        # "+ h" in the GEMM epilogue (its gradient is dy itself)
        # cross attention
        n, h = _lnw(self, h, tb.norm2.weight, tb.norm2.bias, pl["ln2_g"], pl["ln2_b"], 1e-5, fork=True)
        a2 = tb.attn2
        l2, h2 = live_of(a2.gate, 64)
        h2 = a2.heads if h2 is None else h2
        ehs = encoder_hidden_states.ehs if isinstance(encoder_hidden_states, CtxBundle) else \
            encoder_hidden_states.to(device=dev, dtype=torch.bfloat16)
        q = lin(n, "a2q", a2.to_q, l2)
        if fuse_qkv:
            kv = lin_cat(ehs, "a2kv", (a2.to_k, a2.to_v), l2)
        else:
            kv = torch.cat([lin(ehs, "a2k", a2.to_k, l2), lin(ehs, "a2v", a2.to_v, l2)], dim=-1)
        o = AG.CrossAttnFn.apply(q, kv, h2)
        h = lin(o, "a2o", a2.to_out[0], None, l2, residual=h)
        # feed-forward
        n, h = _lnw(self, h, tb.norm3.weight, tb.norm3.bias, pl["ln3_g"], pl["ln3_b"], 1e-5, fork=True)
        geglu, lin2 = tb.ff.net[0], tb.ff.net[2]
        lf, _ = live_of(geglu.gate, geglu.dim_out // geglu.gate.width)
        lo2 = None if lf is None else torch.cat([lf, lf + geglu.dim_out])
        hg = lin(n, "ff1", geglu.proj, lo2)
        f = AG.GegluFn.apply(hg, None)
        h = lin(f, "ff2", lin2, None, lf, residual=h)
        out = lin(h, "proj_out", self.proj_out, residual=x_tok)
        return _nchw(out.reshape(B, H, W, C))

    # ---- forward --------------------------------------------------------------------------------------------------
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states=None, timestep=None, added_cond_kwargs=None,
                class_labels=None, cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None,
                return_dict: bool = True):
        if attention_mask is not None or encoder_attention_mask is not None:
            raise NotImplementedError("attention masks are not used on the APTP path (pruning_pipelines.py:796-802)")
        x = _nhwc(hidden_states)
        dst = _take_dst(self, x.shape[0], x.shape[1], x.shape[2], x.shape[3], x.device)
        if _ft_mode(self) and self.proj_in.weight.requires_grad:
            if self.depth_gated and (self.dropped or self._depth_state()[0] == 0.0):
                return self._ret(hidden_states, return_dict)
            return self._ret(self._forward_ft(x, encoder_hidden_states), return_dict)
        if self._needs_autograd(x) and not (self.depth_gated and self.dropped):
            return self._ret(self._forward_train(x, encoder_hidden_states), return_dict)
        d_hard, d_vec = self._depth_state()
        if self.depth_gated and (self.dropped or d_hard == 0.0):
            return self._ret(hidden_states, return_dict)                  # blocks.py:1190-1194
        dev = x.device
        pl = self.plan(dev)
        tb = self.transformer_blocks[0]
        B, H, W, C = x.shape
        P = H * W
        a = ops.groupnorm(x, pl["gn_g"], pl["gn_b"], self.groups, 1e-6, False)
        tok = a.reshape(B, P, C)
        x_tok = x.reshape(B, P, C)      # a view also for a channel slice of a wider buffer (uniform pixel stride)
        # The three LayerNorms are folded into their neighbours: the GEMM that produces the residual stream emits per-row
        # (sum, sumsq) partials, the GEMM that consumes LN(h) reads h with gamma folded into its weights (_ln_linear).
        h, st = ops.linear(tok, pl["proj_in"], rowstats=FOLD_LN)
        # --- self attention
        hl = pl["a1_heads"]
        gkw = {}
        if pl["a1_dense_gate"]:
            gkw = dict(colgate=self._gate(tb.attn1.gate, dev, 3), gate_group=64)
        qkv = self._ln_linear(pl, h, st, 1, "a1_qkv", **gkw)
        w = hl * 64
        o = ops.attention(qkv[..., :w], qkv[..., w:2 * w], qkv[..., 2 * w:3 * w], hl)
        h, st = ops.linear(o, pl["a1_o"], residual=h, rowstats=FOLD_LN)
        # --- cross attention
        hl = pl["a2_heads"]
        w = hl * 64
        gkw = {}
        if pl["a2_dense_gate"]:
            gkw = dict(colgate=self._gate(tb.attn2.gate, dev), gate_group=64)
        q = self._ln_linear(pl, h, st, 2, "a2_q", **gkw)
        kv = self._ctx_kv(encoder_hidden_states, pl, tb, dev)
        o = ops.attention(q, kv[..., :w], kv[..., w:2 * w], hl)
        # --- feed-forward + proj_out: one kernel per 64-token tile where a row tile per CU fills the chip (ops.ff_tail)
        soft_depth = self.depth_gated and d_hard is None
        if FOLD_LN and not pl["ff_dense_gate"] and not soft_depth:
            pw1 = pl.get("ff1_ln")
            if pw1 is None:
                pw1 = pl["ff1_ln"] = pl["ff1_ln_make"]()
            if ops.ff_tail_supported(h, pw1, pl["ff2"], pl["proj_out"]):
                h = ops.linear(o, pl["a2_o"], residual=h)
                out = ops.ff_tail(h, x_tok, pw1, pl["ff2"], pl["proj_out"], 1e-5, out=None if dst is None else dst.reshape(B, P, C),
                                  colstats=True)
                return self._ret(_nchw(out.reshape(B, H, W, C) if dst is None else dst), return_dict)
        h, st = ops.linear(o, pl["a2_o"], residual=h, rowstats=FOLD_LN)
        # --- feed-forward
        gkw = {}
        if pl["ff_dense_gate"]:
            geglu = tb.ff.net[0]
            gkw = dict(colgate=self._gate(geglu.gate, dev), gate_group=geglu.dim_out // geglu.gate.width)
        f = self._ln_linear(pl, h, st, 3, "ff1", **gkw)
        h = ops.linear(f, pl["ff2"], residual=h)
        # --- proj_out + residual (+ depth lerp)
        dkw = {}
        if soft_depth:
            dkw = dict(depth=d_vec, depth_in=x_tok)
        out = ops.linear(h, pl["proj_out"], residual=x_tok, out=None if dst is None else dst.reshape(B, P, C), colstats=True, **dkw)
        return self._ret(_nchw(out.reshape(B, H, W, C) if dst is None else dst), return_dict)

    @staticmethod
    def _ln_linear(pl, h, st, idx, name, **kw):
        """linear(LayerNorm_idx(h)) (blocks.py:782-785,808-813,821-823): folded into ONE launch when the producer of h
        emitted row statistics, otherwise the stand-alone LayerNorm kernel followed by the plain GEMM"""
        if st is None:
            n = ops.layernorm(h, pl[f"ln{idx}_g"], pl[f"ln{idx}_b"], 1e-5)
            return ops.linear(n, pl[name], **kw)
        pw = pl.get(name + "_ln")
        if pw is None:
            pw = pl[name + "_ln"] = pl[name + "_ln_make"]()
        return ops.linear(h, pw, ln=(st, 1e-5), **kw)

    @staticmethod
    def _ret(t, return_dict):
        if not return_dict:
            return (t,)
        return Transformer2DModelOutput(sample=t)

    def _ctx_kv(self, ehs, pl, tb, dev):
        if isinstance(ehs, CtxBundle):
            kv = ehs.kv.get(id(tb.attn2))
            if kv is not None:
                return kv
            ehs_t = ehs.ehs
        else:
            ehs_t = ehs.to(device=dev, dtype=ops.ACT_DTYPE)
        pw = pl.get("a2_kv")
        if pw is None:
            pw = ops.pack_weight(pl["a2_kv_w"], None, device=dev)
            pl["a2_kv"] = pw
        gkw = {}
        if pl["a2_dense_gate"]:
            gkw = dict(colgate=self._gate(tb.attn2.gate, dev, 2), gate_group=64)
        return ops.linear(ehs_t, pw, **gkw)


class Transformer2DModelWidthDepthGated(Transformer2DModelWidthGated):
    """blocks.py:1070-1438"""
    depth_gated = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.depth_gate = DepthGate(1)

    def _depth_state(self):
        h = self.depth_gate.host_value().flatten()
        if bool(((h == 0) | (h == 1)).all()) and bool((h == h[0]).all()):
            return float(h[0]), None
        return None, self.depth_gate.gate_f.detach().flatten().to(dtype=torch.float32)


@dataclass
class Transformer2DModelOutput:
    sample: torch.Tensor


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor


class Downsample2D(nn.Module):
    """diffusers Downsample2D(use_conv=True, padding=1): conv 3x3 stride 2."""

    def __init__(self, channels: int):
        super().__init__()
        self.conv = Conv2dP(channels, channels, 3)
        self._pw, self._pw_ver, self._pinned, self._parked = None, None, False, []

    def invalidate(self):
        self._pw, self._pinned = None, False
        self._parked.clear()
        self.__dict__.pop("_vparams", None)

    def forward(self, hidden_states, scale: float = 1.0):
        x = _nhwc(hidden_states)
        if _ft_mode(self):
            self.__dict__.pop("_out_dst", None)
        ver = _versions(self)
        if self._pw is None or self._pw.w.device != x.device or self._pw_ver != ver:
            if self._pw is not None and self._pinned:
                self._parked.append((self._pw, getattr(self, "_pwb", None)))      # a captured graph still points to it
            self._pw = ops.pack_weight(self.conv.weight.detach(), self.conv.bias.detach(), device=x.device)
            self._pwb, self._pw_ver, self._pinned = None, ver, False
        self._pinned = self._pinned or _capturing()
        if _ft_mode(self) and self.conv.weight.requires_grad:
            from . import autograd as AG
            return _nchw(_cw(self, x, self.conv.weight, self.conv.bias, self._pw, self._get_bwd(x.device), stride=2, pad=1))
        if torch.is_grad_enabled() and x.requires_grad:
            from . import autograd as AG
            return _nchw(AG.conv(x, self._pw, self._get_bwd(x.device), stride=2, pad=1))
        Ho, Wo = (x.shape[1] - 1) // 2 + 1, (x.shape[2] - 1) // 2 + 1
        dst = _take_dst(self, x.shape[0], Ho, Wo, self._pw.N, x.device)
        return _nchw(ops.conv_gemm(x, self._pw, stride=2, pad=1, out=dst, colstats=True))

    def _get_bwd(self, dev):
        def get():
            if getattr(self, "_pwb", None) is None:
                self._pwb = ops.pack_weight_dgrad(self.conv.weight.detach(), device=dev)
            return self._pwb
        return get


class Upsample2D(nn.Module):
    """diffusers Upsample2D(use_conv=True): nearest x2 then conv 3x3 — the upsample is folded into the conv's gather."""

    def __init__(self, channels: int):
        super().__init__()
        self.conv = Conv2dP(channels, channels, 3)
        self._pw, self._pw_ver, self._pinned, self._parked = None, None, False, []

    def invalidate(self):
        self._pw, self._pinned = None, False
        self._parked.clear()
        self.__dict__.pop("_vparams", None)

    def forward(self, hidden_states, output_size=None, scale: float = 1.0):
        x = _nhwc(hidden_states)
        if _ft_mode(self):
            self.__dict__.pop("_out_dst", None)
        ver = _versions(self)
        if self._pw is None or self._pw.w.device != x.device or self._pw_ver != ver:
            if self._pw is not None and self._pinned:
                self._parked.append((self._pw, getattr(self, "_pwb", None)))      # a captured graph still points to it
            self._pw = ops.pack_weight(self.conv.weight.detach(), self.conv.bias.detach(), device=x.device)
            self._pwb, self._pw_ver, self._pinned = None, ver, False
        self._pinned = self._pinned or _capturing()
        if _ft_mode(self) and self.conv.weight.requires_grad:
            from . import autograd as AG
            return _nchw(_cw(self, x, self.conv.weight, self.conv.bias, self._pw, self._get_bwd(x.device), ups=1))
        if torch.is_grad_enabled() and x.requires_grad:
            from . import autograd as AG
            return _nchw(AG.conv(x, self._pw, self._get_bwd(x.device), ups=1))
        dst = _take_dst(self, x.shape[0], 2 * x.shape[1], 2 * x.shape[2], self._pw.N, x.device)
        return _nchw(ops.conv_gemm(x, self._pw, ups=1, out=dst, colstats=True))

    def _get_bwd(self, dev):
        def get():
            if getattr(self, "_pwb", None) is None:
                self._pwb = ops.pack_weight_dgrad(self.conv.weight.detach(), device=dev)
            return self._pwb
        return get


# ----------------------------------------------------------------------------------------------------------------
# block containers (forwards follow diffusers CrossAttnDownBlock2D / DownBlock2D / UNetMidBlock2DCrossAttn /
# CrossAttnUpBlock2D / UpBlock2D; gate placement follows blocks.py:1717-1807, 2554-2736, 2004-2243, 2419-2550)
# ----------------------------------------------------------------------------------------------------------------
class _GatedContainer(nn.Module):
    has_cross_attention = True

    def _init_common(self):
        self.structure = {"width": [], "depth": []}
        self.total_macs, self.prunable_macs = 0.0, 0.0

    def get_gate_structure(self):
        if len(self.structure["width"]) == 0:
            structure = {"width": [], "depth": []}
            for b in list(self.resnets) + list(self.attentions):
                s = b.get_gate_structure()
                structure["width"].append(s["width"])
                structure["depth"].append(s["depth"])
            self.structure = structure
        return self.structure

    def set_gate_structure(self, arch_vectors):
        # resnets first, then attentions — the order of get_gate_structure (blocks.py:1833-1861)
        width_vectors, depth_vectors = arch_vectors["width"], arch_vectors["depth"]
        for b in list(self.resnets) + list(self.attentions):
            s = b.get_gate_structure()
            block_vectors = {"width": [], "depth": []}
            for i in range(len(s["width"])):
                assert s["width"][i] == width_vectors[0].shape[1]
                block_vectors["width"].append(width_vectors.pop(0))
            for i in range(len(s["depth"])):
                if s["depth"][i] == 1:
                    block_vectors["depth"].append(depth_vectors.pop(0))
            b.set_gate_structure(block_vectors)


def _make_resnet(cin, cout, temb, depth_gated, groups, eps, concat=False, skip_dim=None):
    cls = ResnetBlock2DWidthDepthGated if depth_gated else ResnetBlock2DWidthGated
    return cls(in_channels=cin, out_channels=cout, temb_channels=temb, groups=groups, eps=eps,
               is_input_concatenated=concat, skip_connection_dim=skip_dim)


def _make_attn(ch, heads, xdim, depth_gated, groups, gated_ff, ff_gate_width):
    cls = Transformer2DModelWidthDepthGated if depth_gated else Transformer2DModelWidthGated
    return cls(heads, ch // heads, in_channels=ch, cross_attention_dim=xdim, norm_num_groups=groups,
               gated_ff=gated_ff, ff_gate_width=ff_gate_width)


class CrossAttnDownBlock2DWidthHalfDepthGated(_GatedContainer):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, num_attention_heads, cross_attention_dim,
                 add_downsample, resnet_groups=32, resnet_eps=1e-5, gated_ff=True, ff_gate_width=32, with_attention=True):
        super().__init__()
        self.has_cross_attention = with_attention
        resnets, attentions = [], []
        for i in range(num_layers):
            last = i == num_layers - 1
            resnets.append(_make_resnet(in_channels if i == 0 else out_channels, out_channels, temb_channels, last,
                                        resnet_groups, resnet_eps))
            if with_attention:
                attentions.append(_make_attn(out_channels, num_attention_heads, cross_attention_dim, last,
                                             resnet_groups, gated_ff, ff_gate_width))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None
        self._init_common()

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs=None, encoder_attention_mask=None, scale: float = 1.0):
        output_states = ()
        for i, resnet in enumerate(self.resnets):
            hidden_states = resnet(hidden_states, temb)
            if self.has_cross_attention:
                hidden_states = self.attentions[i](hidden_states, encoder_hidden_states=encoder_hidden_states,
                                                   return_dict=False)[0]
            output_states = output_states + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states = output_states + (hidden_states,)
        return hidden_states, output_states


class DownBlock2DWidthHalfDepthGated(CrossAttnDownBlock2DWidthHalfDepthGated):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, add_downsample, resnet_groups=32,
                 resnet_eps=1e-5):
        super().__init__(in_channels, out_channels, temb_channels, num_layers, 1, 0, add_downsample, resnet_groups,
                         resnet_eps, with_attention=False)


class UNetMidBlock2DCrossAttnWidthGated(_GatedContainer):
    def __init__(self, in_channels, temb_channels, num_attention_heads, cross_attention_dim, resnet_groups=32,
                 resnet_eps=1e-5, gated_ff=True, ff_gate_width=32):
        super().__init__()
        self.resnets = nn.ModuleList([
            _make_resnet(in_channels, in_channels, temb_channels, False, resnet_groups, resnet_eps),
            _make_resnet(in_channels, in_channels, temb_channels, False, resnet_groups, resnet_eps)])
        self.attentions = nn.ModuleList([
            _make_attn(in_channels, num_attention_heads, cross_attention_dim, False, resnet_groups, gated_ff,
                       ff_gate_width)])
        self._init_common()

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs=None, encoder_attention_mask=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states, return_dict=False)[0]
            hidden_states = resnet(hidden_states, temb)
        return hidden_states


class CrossAttnUpBlock2DWidthHalfDepthGated(_GatedContainer):
    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, num_attention_heads,
                 cross_attention_dim, add_upsample, resnet_groups=32, resnet_eps=1e-5, gated_ff=True, ff_gate_width=32,
                 with_attention=True):
        super().__init__()
        self.has_cross_attention = with_attention
        resnets, attentions = [], []
        for i in range(num_layers):
            last = i == num_layers - 1
            res_skip = in_channels if last else out_channels
            res_in = prev_output_channel if i == 0 else out_channels
            resnets.append(_make_resnet(res_in + res_skip, out_channels, temb_channels, last, resnet_groups, resnet_eps,
                                        concat=True, skip_dim=res_skip))
            if with_attention:
                attentions.append(_make_attn(out_channels, num_attention_heads, cross_attention_dim, last,
                                             resnet_groups, gated_ff, ff_gate_width))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None
        self._init_common()

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                cross_attention_kwargs=None, upsample_size=None, attention_mask=None, encoder_attention_mask=None,
                scale: float = 1.0):
        for i, resnet in enumerate(self.resnets):
            res_hidden_states = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = _cat_channels(hidden_states, res_hidden_states)
            hidden_states = resnet(hidden_states, temb)
            if self.has_cross_attention:
                hidden_states = self.attentions[i](hidden_states, encoder_hidden_states=encoder_hidden_states,
                                                   return_dict=False)[0]
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states


class UpBlock2DWidthHalfDepthGated(CrossAttnUpBlock2DWidthHalfDepthGated):
    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, add_upsample,
                 resnet_groups=32, resnet_eps=1e-5):
        super().__init__(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, 1, 0, add_upsample,
                         resnet_groups, resnet_eps, with_attention=False)


# APTP_FUSE_SHORTCUT=0 keeps the resnets' 1x1 conv_shortcut as its own launch (A/B timing, debugging)
FUSE_SHORTCUT = os.environ.get("APTP_FUSE_SHORTCUT", "1") != "0"
# APTP_FOLD_LN=0 keeps the three LayerNorms of a transformer block as stand-alone kernels (A/B timing, debugging)
FOLD_LN = os.environ.get("APTP_FOLD_LN", "1") != "0"
FT_FUSE_QKV = os.environ.get("APTP_FT_FUSE_QKV", "1") != "0"     # packed fine-tuning: to_q | to_k | to_v (and to_k | to_v) as one contraction
CAT_STATS = {"views": 0, "copies": 0}     # how the skip-concats of the forwards so far were realised (tests / tools)


def _cat_channels(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], dim=1) for channels_last activations, produced directly in NHWC memory."""
    A, Bn = _nhwc(a), _nhwc(b)
    if torch.is_grad_enabled() and (A.requires_grad or Bn.requires_grad):
        return _nchw(torch.cat([A, Bn], dim=3))
    Ct = A.shape[3] + Bn.shape[3]
    if (A.dtype == Bn.dtype and A.shape[:3] == Bn.shape[:3] and A.stride() == Bn.stride() and A.stride(3) == 1
            and A.untyped_storage().data_ptr() == Bn.untyped_storage().data_ptr()
            and Bn.storage_offset() == A.storage_offset() + A.shape[3] and ops._ld(A) == Ct):
        # both halves were produced in place in one _CatSlot buffer: the concat is a view
        CAT_STATS["views"] += 1
        return _nchw(torch.as_strided(A, (A.shape[0], A.shape[1], A.shape[2], Ct), A.stride(), A.storage_offset()))
    CAT_STATS["copies"] += 1
    out = torch.empty(A.shape[0], A.shape[1], A.shape[2], A.shape[3] + Bn.shape[3], dtype=A.dtype, device=A.device)
    out[..., :A.shape[3]].copy_(A)
    out[..., A.shape[3]:].copy_(Bn)
    return _nchw(out)


# ----------------------------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------------------------
class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = LinearP(in_channels, time_embed_dim)
        self.linear_2 = LinearP(time_embed_dim, time_embed_dim)


class UNet2DConditionModelGated(nn.Module):
    """unet_2d_conditional.py:628-2181 for the SD-2.1 family of configurations."""

    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2DWidthHalfDepthGated",) * 3 + ("DownBlock2DWidthHalfDepthGated",),
                 mid_block_type: str = "UNetMidBlock2DCrossAttnWidthGated",
                 up_block_types: Tuple[str, ...] = ("UpBlock2DWidthHalfDepthGated",) + ("CrossAttnUpBlock2DWidthHalfDepthGated",) * 3,
                 block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280), layers_per_block: int = 2,
                 cross_attention_dim: int = 1024, attention_head_dim: Union[int, Tuple[int, ...]] = (5, 10, 20, 20),
                 norm_num_groups: int = 32, norm_eps: float = 1e-5, gated_ff: bool = True, ff_gate_width: int = 32,
                 **unused):
        super().__init__()
        n = len(block_out_channels)
        heads = (attention_head_dim,) * n if isinstance(attention_head_dim, int) else tuple(attention_head_dim)
        assert len(down_block_types) == n and len(up_block_types) == n
        self.config = dict(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                           down_block_types=tuple(down_block_types), mid_block_type=mid_block_type,
                           up_block_types=tuple(up_block_types), block_out_channels=tuple(block_out_channels),
                           layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
                           attention_head_dim=heads, norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                           gated_ff=gated_ff, ff_gate_width=ff_gate_width)
        boc = tuple(block_out_channels)
        T = boc[0] * 4
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv_in = Conv2dP(in_channels, boc[0], 3)
        self.time_embedding = TimestepEmbedding(boc[0], T)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, t in enumerate(down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            last = i == n - 1
            if t == "CrossAttnDownBlock2DWidthHalfDepthGated":
                blk = CrossAttnDownBlock2DWidthHalfDepthGated(in_ch, out_ch, T, layers_per_block, heads[i],
                                                              cross_attention_dim, not last, norm_num_groups, norm_eps,
                                                              gated_ff, ff_gate_width)
            elif t == "DownBlock2DWidthHalfDepthGated":
                blk = DownBlock2DWidthHalfDepthGated(in_ch, out_ch, T, layers_per_block, not last, norm_num_groups,
                                                     norm_eps)
            else:
                raise ValueError(f"unsupported down block type {t}")
            self.down_blocks.append(blk)
        assert mid_block_type == "UNetMidBlock2DCrossAttnWidthGated"
        self.mid_block = UNetMidBlock2DCrossAttnWidthGated(boc[-1], T, heads[-1], cross_attention_dim, norm_num_groups,
                                                           norm_eps, gated_ff, ff_gate_width)
        self.up_blocks = nn.ModuleList()
        rev, rev_heads = list(reversed(boc)), list(reversed(heads))
        out_ch = rev[0]
        for i, t in enumerate(up_block_types):
            prev_out, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, n - 1)]
            last = i == n - 1
            if t == "CrossAttnUpBlock2DWidthHalfDepthGated":
                blk = CrossAttnUpBlock2DWidthHalfDepthGated(in_ch, out_ch, prev_out, T, layers_per_block + 1,
                                                            rev_heads[i], cross_attention_dim, not last,
                                                            norm_num_groups, norm_eps, gated_ff, ff_gate_width)
            elif t == "UpBlock2DWidthHalfDepthGated":
                blk = UpBlock2DWidthHalfDepthGated(in_ch, out_ch, prev_out, T, layers_per_block + 1, not last,
                                                   norm_num_groups, norm_eps)
            else:
                raise ValueError(f"unsupported up block type {t}")
            self.up_blocks.append(blk)
        self.conv_norm_out = NormP(boc[0], norm_eps, norm_num_groups)
        self.conv_out = Conv2dP(boc[0], out_channels, 3)
        self.structure = {"width": [], "depth": []}
        self.prunable_macs_list = None
        self.resource_info_dict = None
        self.semantics = "gated"
        self._misc = None       # packed conv_in / time MLP / conv_out
        self._misc_parked = []  # superseded _misc packs a captured graph may still point to
        self._batched = _PlanCache()      # structure-key -> batched temb / ctx-kv packs

    # ---- construction helpers -------------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config: Optional[dict] = None, **kwargs):
        cfg = dict(config or {})
        cfg.update(kwargs)
        return cls(**cfg)

    @torch.no_grad()
    def init_synthetic(self, seed: int = 0, w_std: float = 0.02, beta_std: float = 0.1):
        """Seeded random-init weights at the configured shapes (no checkpoint access on the GPU box): N(0, w_std)
        conv/linear weights, N(0, w_std/2) biases, norm gamma = 1, norm beta ~ N(0, beta_std)."""
        g = torch.Generator().manual_seed(seed)
        for name, prm in self.named_parameters():
            owner = self.get_submodule(name.rsplit(".", 1)[0])
            if isinstance(owner, NormP):
                if name.endswith(".weight"):
                    prm.fill_(1.0)
                else:
                    prm.copy_(torch.randn(prm.shape, generator=g) * beta_std)
            elif name.endswith(".weight"):
                prm.copy_(torch.randn(prm.shape, generator=g) * w_std)
            else:
                prm.copy_(torch.randn(prm.shape, generator=g) * (0.5 * w_std))
        self.invalidate_plans()
        return self

    def invalidate_plans(self):
        """Drop every packed-weight cache (call after changing parameters)."""
        self._misc = None
        self._misc_parked = []
        self._batched.clear()
        for m in self.modules():
            if m is not self and hasattr(m, "invalidate"):
                m.invalidate()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate_plans()
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.invalidate_plans()
        return out

    def state_dict(self, *a, **k):
        """(under a PackedTrainer the trained values live in the packed tensors: write them back into the diffusers-named
        parameters first, so a checkpoint taken mid-training -- trainer.py:1767-1778 -- holds what was trained)"""
        pk = self.__dict__.get("_pk")
        if pk is not None:
            pk.export_()
        return super().state_dict(*a, **k)

    # ---- checkpoints in the reference's on-disk layout (diffusers ModelMixin API; checkpoint.py) -------------------
    def save_pretrained(self, save_directory: str, **unused):
        """``model.save_pretrained(os.path.join(output_dir, "unet"))`` as trainer.py:262-265 calls it: config.json +
        diffusion_pytorch_model.safetensors in ``save_directory`` (a pruned expert at its sliced shapes, with
        arch_vector.pt in the parent directory)."""
        from . import checkpoint
        root, sub = os.path.split(os.path.normpath(save_directory))
        checkpoint.save_pretrained(self, root, subfolder=sub)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: Optional[str] = None, **kwargs):
        """unet_2d_conditional.py:2184-2472 for local directories: ``arch_vector=`` / ``random_pruning_ratio=`` select the
        structure of a pruned expert, configuration keys (``down_block_types`` …) override config.json, hub-only keyword
        arguments (``revision``, ``cache_dir`` …) are ignored."""
        from . import checkpoint
        return checkpoint.from_pretrained(pretrained_model_name_or_path, subfolder=subfolder, cls=cls, **kwargs)

    # ---- reference API --------------------------------------------------------------------------------------------
    def freeze(self):
        # unet_2d_conditional.py:2118-2122: gate_f is a plain attribute, so this freezes every parameter (quirk Q2)
        for name, param in self.named_parameters():
            if "gate_f" not in name:
                param.requires_grad = False

    def _containers(self):
        return list(self.down_blocks) + [self.mid_block] + list(self.up_blocks)

    def get_structure(self):
        if len(self.structure["width"]) == 0:
            structure = {"width": [], "depth": []}
            for m in self._containers():
                s = m.get_gate_structure()
                assert len(s) == 2 and len(s["width"]) == len(s["depth"])
                structure["width"] = structure["width"] + s["width"]
                structure["depth"] = structure["depth"] + s["depth"]
            self.structure = structure
        return self.structure

    def set_structure(self, arch_vectors, prefetch_hosts: bool = True):
        """unet_2d_conditional.py:1365-1413 — consumes (pops) the caller's width/depth lists.
        prefetch_hosts=False skips the bulk device->host copy (a stream synchronisation) that classifies the gates as
        hard / soft for the next forward; for callers that only need the gates installed (MAC accounting)."""
        width_vectors, depth_vectors = arch_vectors["width"], arch_vectors["depth"]
        # kept (by reference) so a checkpoint can record the installed architecture vector (checkpoint.arch_vector_of)
        self._installed_structure = {"width": list(width_vectors), "depth": list(depth_vectors)}
        self._structure_epoch = self.__dict__.get("_structure_epoch", 0) + 1      # (identity of the installed code for graph caches)
        # one bulk device->host copy of all gates so per-module mode selection needs no further syncs
        self._host_map = {}
        if prefetch_hosts:
            self._prefetch_hosts(width_vectors, depth_vectors)
        for m in self._containers():
            s = m.get_gate_structure()
            block_vectors = {"width": [], "depth": []}
            for i in range(len(s["width"])):
                for j in range(len(s["width"][i])):
                    assert s["width"][i][j] == width_vectors[0].shape[1]
                    block_vectors["width"].append(width_vectors.pop(0))
            for i in range(len(s["depth"])):
                if s["depth"][i] == [1]:
                    block_vectors["depth"].append(depth_vectors.pop(0))
            m.set_gate_structure(block_vectors)
        self._attach_hosts()

    def _prefetch_hosts(self, width_vectors, depth_vectors):
        self._host_map = {}
        tensors = list(width_vectors) + list(depth_vectors)
        if not tensors:
            return
        if all(t.device.type == "cpu" for t in tensors):
            return
        flat = torch.cat([t.detach().float().reshape(-1) for t in tensors]).cpu()
        off = 0
        for t in tensors:
            n = t.numel()
            self._host_map[id(t)] = flat[off:off + n].reshape(t.shape)
            off += n

    def _attach_hosts(self):
        hm = getattr(self, "_host_map", None)
        if not hm:
            return
        for m in self.modules():
            if isinstance(m, (WidthGate, DepthGate)):
                h = hm.get(id(m.gate_f))
                if h is not None:
                    m.set_host_value(h)
        self._host_map = {}

    # ---- MAC accounting (unet_2d_conditional.py:2124-2181; trainer.py:1256-1306) ----------------------------------------
    def count_macs(self, latent_h: int, latent_w: Optional[int] = None, text_len: int = 77):
        """Analytic stand-in for the reference's hook-based count_ops_and_params on a batch-1 forward: assigns every
        module's MAC constant, then fills prunable_macs_list / resource_info_dict the way Pruner.count_macs does
        (with the all-ones structure installed by the caller, trainer.py:1262-1266)."""
        from . import macs
        macs.assign_module_macs(self, latent_h, latent_w, text_len)
        sanity = self.calc_macs()
        self.prunable_macs_list = [[e / sanity["prunable_macs"] for e in elem] for elem in self.get_prunable_macs()]
        self.resource_info_dict = sanity
        return sanity

    def calc_macs(self):
        from . import macs
        return macs.unet_calc_macs(self)

    def get_prunable_macs(self):
        from . import macs
        return macs.unet_get_prunable_macs(self)

    def get_block_utilization(self):
        from . import macs
        return macs.unet_get_block_utilization(self)

    # ---- model-level packed weights ---------------------------------------------------------------------------------
    def _resnets(self):
        return [m for m in self.modules() if isinstance(m, ResnetBlock2DWidthGated)]

    def _transformers(self):
        return [m for m in self.modules() if isinstance(m, Transformer2DModelWidthGated)]

    def _misc_packs(self, dev):
        ver = () if self.__dict__.get("_pk") is not None else \
            tuple(p._version for mod in (self.conv_in, self.time_embedding, self.conv_norm_out, self.conv_out) for p in mod.parameters())
        if self._misc is not None and self._misc["dev"] == str(dev) and self._misc["ver"] == ver:
            self._misc["pinned"] = self._misc["pinned"] or _capturing()
        if self._misc is None or self._misc["dev"] != str(dev) or self._misc["ver"] != ver:
            if self._misc is not None and self._misc["pinned"]:
                self._misc_parked.append(self._misc)
            cin_pad = ops.round_up(self.in_channels, 8)
            m = {"dev": str(dev), "cin_pad": cin_pad, "ver": ver, "pinned": _capturing()}
            m["conv_in"] = ops.pack_weight(self.conv_in.weight.detach(), self.conv_in.bias.detach(), device=dev)
            te = self.time_embedding
            m["t1"] = ops.pack_weight(te.linear_1.weight.detach(), te.linear_1.bias.detach(), device=dev)
            m["t2"] = ops.pack_weight(te.linear_2.weight.detach(), te.linear_2.bias.detach(), device=dev)
            m["gn_g"], m["gn_b"] = _f32(self.conv_norm_out.weight).to(dev), _f32(self.conv_norm_out.bias).to(dev)
            m["conv_out"] = ops.pack_weight(self.conv_out.weight.detach(), self.conv_out.bias.detach(), device=dev)
            half = self.conv_in.out_channels // 2
            m["freqs"] = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=dev) / half)
            self._misc = m
        return self._misc

    def _batched_packs(self, dev):
        """One GEMM for all 22 time_emb_proj, one for all 16 cross-attention K/V projections."""
        resnets, trans = self._resnets(), self._transformers()
        for r in resnets:
            r.semantics = self.semantics
        rplans = [r.plan(dev) for r in resnets if not (r.depth_gated and (r.dropped or r._depth_state()[0] == 0.0))]
        live_res = [r for r in resnets if not (r.depth_gated and (r.dropped or r._depth_state()[0] == 0.0))]
        live_tr = [t for t in trans if not (t.depth_gated and (t.dropped or t._depth_state()[0] == 0.0))]
        tplans = [t.plan(dev) for t in live_tr]
        key = (tuple(id(p) for p in rplans), tuple(id(p) for p in tplans), str(dev))
        bp = self._batched.get(key, ())
        if bp is not None:
            return bp
        bp = {"key": key, "plans": (rplans, tplans)}      # (the plans stay alive as long as their ids are a cache key)
        ws, bs, slots, off = [], [], [], 0
        for r, pl in zip(live_res, rplans):
            npad = pl["c_pad"]
            w, b = pl["temb_w"], pl["temb_b"]
            if w.shape[0] != npad:
                w = torch.cat([w, w.new_zeros(npad - w.shape[0], w.shape[1])], 0)
                b = torch.cat([b, b.new_zeros(npad - b.shape[0])], 0)
            ws.append(w); bs.append(b)
            slots.append((id(r), off, npad))
            off += npad
        bp["temb_pw"] = ops.pack_weight(torch.cat(ws, 0), torch.cat(bs, 0), device=dev)
        bp["temb_slots"] = slots
        ws, slots, off, gates = [], [], 0, []
        for t, pl in zip(live_tr, tplans):
            w = pl["a2_kv_w"]
            ws.append(w)
            slots.append((id(t.transformer_blocks[0].attn2), off, w.shape[0]))
            off += w.shape[0]
            gates.append((t.transformer_blocks[0].attn2.gate, pl["a2_dense_gate"], pl["a2_heads"]))
        if ws:
            bp["kv_pw"] = ops.pack_weight(torch.cat(ws, 0), None, device=dev)
        bp["kv_slots"] = slots
        bp["kv_gates"] = gates
        return self._batched.put(key, (), bp)

    def _project_context(self, encoder_hidden_states, bp, dev) -> CtxBundle:
        ehs = encoder_hidden_states.to(device=dev, dtype=ops.ACT_DTYPE).contiguous()
        # projections made with value-dependent (soft / per-sample) gates in the epilogue must not be reused
        reusable = not any(dense for (_, dense, _) in bp["kv_gates"])
        ctx = CtxBundle(ehs=ehs, key=bp["key"] if reusable else None)
        if bp["kv_slots"]:
            gkw = {}
            if any(dense for (_, dense, _) in bp["kv_gates"]):
                cols = []
                for gate, dense, hl in bp["kv_gates"]:
                    g = gate.gate_f.detach().to(device=dev, dtype=torch.float32) if dense else None
                    bg = max(gt.gate_f.shape[0] for gt, d_, _ in bp["kv_gates"] if d_)
                    if g is None:
                        g = torch.ones(bg, hl, device=dev)
                    elif g.shape[0] != bg:
                        g = g.repeat(bg // g.shape[0], 1)
                    cols.append(g.repeat(1, 2))
                gkw = dict(colgate=torch.cat(cols, 1).contiguous(), gate_group=64)
            kv_all = ops.linear(ehs, bp["kv_pw"], **gkw)
            for aid, off, n in bp["kv_slots"]:
                ctx.kv[aid] = kv_all[..., off:off + n]
        return ctx

    @torch.no_grad()
    def precompute_context(self, encoder_hidden_states: torch.Tensor) -> CtxBundle:
        """Project the text states through every live cross-attention K/V weight once (valid until the next
        set_structure / weight change); pass the result as ``encoder_hidden_states`` to forward()."""
        dev = encoder_hidden_states.device
        return self._project_context(encoder_hidden_states, self._batched_packs(dev), dev)

    def _register_cat_slots(self, B, H, W, c_in, dev):
        """Inference only: give every producer of a skip connection and every producer of an up-path hidden state the
        channel range of the concat buffer its consumer (an up-block resnet, blocks.py:485-495 / diffusers
        CrossAttnUpBlock2D.forward torch.cat) reads, so the 12 concats of a forward are views instead of copies.  A
        producer that does not run its inference path (depth gate 0, autograd) simply ignores the slot and the consumer
        falls back to the copying concat.  Returns the destination of conv_in's output."""
        if _ft_mode(self):
            return None                                  # (_take_dst drops any stale registration under autograd)
        skips = [None]                                   # conv_in is handled by the caller
        for blk in self.down_blocks:
            for i in range(len(blk.resnets)):
                skips.append(blk.attentions[i] if blk.has_cross_attention else blk.resnets[i])
            if blk.downsamplers is not None:
                skips.append(blk.downsamplers[0])
        prev = self.mid_block.resnets[-1]
        conv_in_slot = None
        for blk in self.up_blocks:
            for i, resnet in enumerate(blk.resnets):
                cs = resnet.skip_connection_dim
                slot = _CatSlot(resnet.in_channels - cs, cs)
                prev.__dict__["_out_dst"] = (slot, 0)
                sk = skips.pop()
                if sk is None:
                    conv_in_slot = slot
                else:
                    sk.__dict__["_out_dst"] = (slot, 1)
                prev = blk.attentions[i] if blk.has_cross_attention else resnet
            if blk.upsamplers is not None:
                prev = blk.upsamplers[0]
        return None if conv_in_slot is None else conv_in_slot.view(1, B, H, W, c_in, dev)

    def _forward_ft(self, sample, timestep, encoder_hidden_states, return_dict):
        """Forward of the fine-tuning step: every parameter trainable (FineTuner, trainer.py:1529-1540, 1729)."""
        from . import autograd as AG
        dev = sample.device
        B = sample.shape[0]
        misc = self._misc_packs(dev)
        for r in self._resnets():
            r.semantics = self.semantics

        def bwd(name, param):
            def get():
                if name not in misc:
                    misc[name] = ops.pack_weight_dgrad(param.detach(), device=dev)
                return misc[name]
            return get
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.int64, device=dev)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None].to(dev)
        timesteps = timesteps.to(dev).expand(B)
        ang = timesteps.float()[:, None] * misc["freqs"][None, :]
        t_emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).to(torch.bfloat16)
        te = self.time_embedding
        e1 = _cw(self, t_emb[None], te.linear_1.weight, te.linear_1.bias, misc["t1"], bwd("t1_bwd", te.linear_1.weight), pad=0, out_f32=True)
        e1 = torch.nn.functional.silu(e1).to(torch.bfloat16)
        e2 = _cw(self, e1, te.linear_2.weight, te.linear_2.bias, misc["t2"], bwd("t2_bwd", te.linear_2.weight), pad=0, out_f32=True)
        temb = TembBundle(emb_silu=torch.nn.functional.silu(e2)[0].to(torch.bfloat16))
        ctx = CtxBundle(ehs=encoder_hidden_states.to(device=dev, dtype=torch.bfloat16).contiguous())
        x = torch.zeros(B, sample.shape[2], sample.shape[3], misc["cin_pad"], dtype=torch.bfloat16, device=dev)
        x[..., :self.in_channels] = sample.permute(0, 2, 3, 1)
        live_in = torch.arange(self.in_channels, device=dev)
        h = _nchw(_cw(self, x, self.conv_in.weight, self.conv_in.bias, misc["conv_in"], bwd("conv_in_bwd", self.conv_in.weight),
                            live_in=live_in))
        res_samples = (h,)
        for blk in self.down_blocks:
            h, res = blk(hidden_states=h, temb=temb, encoder_hidden_states=ctx)
            res_samples += res
        h = self.mid_block(h, temb, encoder_hidden_states=ctx)
        for blk in self.up_blocks:
            n_res = len(blk.resnets)
            res = res_samples[-n_res:]
            res_samples = res_samples[:-n_res]
            h = blk(hidden_states=h, temb=temb, res_hidden_states_tuple=res, encoder_hidden_states=ctx)
        a = _gnw(self, _nhwc(h), self.conv_norm_out.weight, self.conv_norm_out.bias, misc["gn_g"], misc["gn_b"],
                                  self.conv_norm_out.num_groups, self.conv_norm_out.eps, True, self.conv_norm_out.num_channels, None)
        live_out = torch.arange(self.out_channels, device=dev)
        y = _cw(self, a, self.conv_out.weight, self.conv_out.bias, misc["conv_out"], bwd("conv_out_bwd", self.conv_out.weight),
                      out_f32=True, live_out=live_out)
        out = y[..., :self.out_channels].permute(0, 3, 1, 2).to(sample.dtype)
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)

    # ---- forward (unet_2d_conditional.py:1415-1726) -------------------------------------------------------------------
    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, class_labels=None,
                timestep_cond=None, attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None, encoder_attention_mask=None,
                return_dict: bool = True):
        # no CPU path exists: ops.* reject non-CUDA tensors and _lib.load() raises if libaptp_hip.so is missing
        if attention_mask is not None or encoder_attention_mask is not None or class_labels is not None \
                or down_block_additional_residuals is not None or mid_block_additional_residual is not None:
            raise NotImplementedError("only the arguments the APTP trainer/pipeline pass are supported")
        dev = sample.device
        B = sample.shape[0]
        # launch-order plan of the next-launch weight prefetch (ops._PrefetchPlan): one per model -- teacher and student
        # forwards alternate -- and per mode; installed for THIS thread for the duration of the forward only
        plans = self.__dict__.setdefault("_pf_plans", {})
        mode = torch.is_grad_enabled()
        if mode not in plans:
            plans[mode] = ops._PrefetchPlan()
        ops.set_prefetch_plan(plans[mode])
        if not mode and sample.is_cuda:
            # GroupNorm statistics without finalise launches (ops.ustat_begin): channel unit = gcd of the levels' group sizes
            # (SD-2.1: 320 / 32 = 10), which every group of every GroupNorm of this model -- skip-concats included -- is a multiple of
            unit = self.__dict__.get("_ustat_unit")
            if unit is None:
                import math
                g = self.conv_norm_out.num_groups
                unit = 0
                for c in self.config["block_out_channels"]:
                    unit = math.gcd(unit, c // g) if c % g == 0 else 1
                unit = self.__dict__["_ustat_unit"] = unit if unit >= 4 else 0
            ops.ustat_begin(dev, unit)
        try:
            return self._forward_impl(sample, timestep, encoder_hidden_states, return_dict)
        finally:
            ops.set_prefetch_plan(None)
            ops.ustat_end()

    def _forward_impl(self, sample, timestep, encoder_hidden_states, return_dict):
        dev = sample.device
        B = sample.shape[0]
        if _ft_mode(self) and self.conv_in.weight.requires_grad:
            return self._forward_ft(sample, timestep, encoder_hidden_states, return_dict)
        misc = self._misc_packs(dev)
        bp = self._batched_packs(dev)
        out_dtype = sample.dtype

        # 1. time (unet_2d_conditional.py:1497-1519): sinusoid [cos|sin] -> Linear -> SiLU -> Linear; the SiLU that
        # every resnet applies to emb (blocks.py:335) is fused into linear_2's epilogue.
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.int64, device=dev)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None].to(dev)
        timesteps = timesteps.to(dev).expand(B)
        fused_io = sample.is_cuda and sample.dtype in (torch.float32, torch.bfloat16) and not torch.is_grad_enabled() \
            and ops.ACT_DTYPE == torch.bfloat16          # (the one-launch prologue / epilogue stage bf16; the fp32 parity path does not)
        if fused_io:
            # one launch: sinusoid [cos|sin] as bf16 + the channel-padded channels-last copy of the sample
            x, t_emb = ops.unet_prologue(sample, timesteps, misc["freqs"], misc["cin_pad"])
        else:
            ang = timesteps.float()[:, None] * misc["freqs"][None, :]
            t_emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).to(ops.ACT_DTYPE)
            x = None
        e1 = ops.linear(t_emb[None], misc["t1"], act=ops.ACT_SILU)
        emb_silu = ops.linear(e1, misc["t2"], act=ops.ACT_SILU)           # bf16 [1, B, T] = SiLU(emb)
        tproj = ops.linear(emb_silu, bp["temb_pw"], out_f32=True)[0]      # fp32 [B, sum Npad]
        temb = TembBundle(emb_silu=emb_silu[0])
        for rid, off, npad in bp["temb_slots"]:
            temb.proj[rid] = tproj[:, off:off + npad]

        # text states: one batched K/V projection for all cross-attention layers (reused as-is when the caller
        # hands in the CtxBundle of precompute_context: K/V depend only on the prompt, not on the denoise step)
        if isinstance(encoder_hidden_states, CtxBundle) and encoder_hidden_states.key is not None \
                and encoder_hidden_states.key == bp["key"]:
            ctx = encoder_hidden_states
        else:
            ehs_in = encoder_hidden_states.ehs if isinstance(encoder_hidden_states, CtxBundle) else encoder_hidden_states
            ctx = self._project_context(ehs_in, bp, dev)

        # 2. pre-process: conv_in on the channel-padded NHWC input
        if x is None:
            x = torch.zeros(B, sample.shape[2], sample.shape[3], misc["cin_pad"], dtype=ops.ACT_DTYPE, device=dev)
            x[..., :self.in_channels] = sample.permute(0, 2, 3, 1)
        conv_in_dst = self._register_cat_slots(B, x.shape[1], x.shape[2], misc["conv_in"].N, dev)
        h = _nchw(ops.conv_gemm(x, misc["conv_in"], out=conv_in_dst, colstats=True))

        # 3. down
        down_block_res_samples = (h,)
        for blk in self.down_blocks:
            h, res = blk(hidden_states=h, temb=temb, encoder_hidden_states=ctx)
            down_block_res_samples += res
        # 4. mid
        h = self.mid_block(h, temb, encoder_hidden_states=ctx)
        # 5. up
        for blk in self.up_blocks:
            n_res = len(blk.resnets)
            res = down_block_res_samples[-n_res:]
            down_block_res_samples = down_block_res_samples[:-n_res]
            h = blk(hidden_states=h, temb=temb, res_hidden_states_tuple=res, encoder_hidden_states=ctx)
        # 6. post-process
        if torch.is_grad_enabled() and h.requires_grad:
            from . import autograd as AG

            def get_out_bwd():
                if "conv_out_bwd" not in misc:
                    misc["conv_out_bwd"] = ops.pack_weight_dgrad(self.conv_out.weight.detach(), device=dev)
                return misc["conv_out_bwd"]
            a = AG.GroupNormFn.apply(_nhwc(h), misc["gn_g"], misc["gn_b"], self.conv_norm_out.num_groups, self.conv_norm_out.eps, True)
            y = AG.conv(a, misc["conv_out"], get_out_bwd, out_f32=True)
        else:
            a = ops.groupnorm(_nhwc(h), misc["gn_g"], misc["gn_b"], self.conv_norm_out.num_groups, self.conv_norm_out.eps, True)
            y = ops.conv_gemm(a, misc["conv_out"], out_f32=True)            # fp32 [B,H,W,roundup8(out)]
            if fused_io:
                out = ops.unet_epilogue(y, self.out_channels, out_dtype)
                return (out,) if not return_dict else UNet2DConditionOutput(sample=out)
        out = y[..., :self.out_channels].permute(0, 3, 1, 2).to(out_dtype)
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)


class UNet2DConditionModelPruned(UNet2DConditionModelGated):
    """unet_2d_conditional.py:2184-2472: the physically pruned expert.  Same kernels; hard masks compact the weights
    and — unlike the gated model — dead GroupNorm channels are deleted, not kept at beta (blocks.py:451-463), so no
    beta correction is applied; depth-0 blocks are dropped (blocks.py:645-658, 1432-1438)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.semantics = "pruned"

    def load_state_dict(self, state_dict, strict: bool = True, **k):
        """A pruned expert's file holds sliced tensors (the reference's modules are physically sliced, and
        scripts/metrics/generate_fid_images.py:99-101 loads such a file into the pruned model): scatter them into the live
        rows / columns of the full-shape masters.  Full-shape state dicts load as usual."""
        own = self.state_dict()
        full = all(kk in state_dict and tuple(state_dict[kk].shape) == tuple(v.shape) for kk, v in own.items())
        if full or not getattr(self, "_installed_structure", None):
            return super().load_state_dict(state_dict, strict=strict, **k)
        from . import checkpoint
        return checkpoint.load_pruned_state_dict(self, state_dict, strict=strict)

    def prune(self, arch_vectors):
        """Binarise the architecture code with hard_concrete's threshold (estimation_utils.py:67-75), install it and
        mark modules pruned / dropped — what the reference does inside from_pretrained (:2421-2436) through the
        per-module prune() methods (which require a single-row gate, blocks.py:426)."""
        hard = {"width": [(w.detach() >= 0.5).float() for w in arch_vectors["width"]],
                "depth": [(d.detach() >= 0.5).float() for d in arch_vectors["depth"]]}
        for t in hard["width"] + hard["depth"]:
            assert t.shape[0] == 1, "Pruning is only supported for single batch size"
        self.set_structure(hard)
        for r in self._resnets():
            r.pruned = True
            r.dropped = bool(r.depth_gated and r._depth_state()[0] == 0.0)
        for t in self._transformers():
            t.pruned = True
            t.dropped = bool(t.depth_gated and t._depth_state()[0] == 0.0)
        return self
