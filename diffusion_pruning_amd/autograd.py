"""torch.autograd glue for the pruning step: every differentiable op of the gated U-Net as an
``autograd.Function`` whose forward AND backward launch the HIP kernels of libaptp_hip.so.

Only data gradients and gate gradients exist: APTP freezes the U-Net during pruning (``unet.freeze()``,
pdm/training/trainer.py:742; the optimizer's U-Net group is empty, :827-829), so the backward pass is
dgrad through every contraction (the same implicit-GEMM kernel with flipped/transposed packed weights), the
norm / attention / GEGLU backward kernels, and per-(sample, gate-entry) reductions for the width / head / FF gates.
Activations stay bf16 channels-last; gate gradients are accumulated in fp32.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch.autograd import Function

from . import ops
from .ops import PackedWeight


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _rc(t: torch.Tensor) -> torch.Tensor:
    """`t` itself when its rows of channels lie at one constant stride (a contiguous tensor, or a channel slice of a wider
    buffer such as one third of a fused dq|dk|dv gradient or one half of a skip-concat gradient) -- the GEMM, weight-gradient
    and column-sum kernels take a leading dimension -- else a contiguous copy."""
    if t.is_contiguous():
        return t
    if t.dim() >= 2 and t.stride(-1) == 1:
        ld = t.stride(-2)
        ok = ld >= t.shape[-1] and ld % 8 == 0 and t.data_ptr() % 16 == 0
        for d in range(t.dim() - 2, 0, -1):
            ok = ok and (t.shape[d - 1] == 1 or t.stride(d - 1) == t.stride(d) * t.shape[d])
        if ok and all(n > 1 for n in t.shape[1:-1]):       # size-1 middle dimensions have arbitrary strides: keep it simple
            return t
    return t.contiguous()


class ConvFn(Function):
    """y = conv(x, w) + bias (+ rowbias[b]) (+ residual) on [B,H,W,C] (or tokens [B,L,C]); backward = dgrad through the same
    kernel; the residual (the ``+ x`` of a resnet / transformer sub-block, blocks.py:369,791,815,823) is added in the GEMM's
    epilogue and its gradient is dy itself -- no elementwise add kernel in either direction."""

    @staticmethod
    def forward(ctx, x, pw: PackedWeight, get_bwd: Callable[[], PackedWeight], stride: int, pad: int, ups: int,
                rowbias: Optional[torch.Tensor], out_f32: bool, residual: Optional[torch.Tensor]):
        tokens = x.dim() == 3
        xin = x.unsqueeze(2) if tokens else x
        res = None
        if residual is not None:
            res = residual.unsqueeze(2) if tokens else residual
        y = ops.conv_gemm(xin, pw, stride=stride, pad=pad, ups=ups, rowbias=rowbias, out_f32=out_f32, residual=res)
        ctx.meta = (pw, get_bwd, stride, pad, ups, tokens)
        return y.squeeze(2) if tokens else y

    @staticmethod
    def backward(ctx, dy):
        pw, get_bwd, stride, pad, ups, tokens = ctx.meta
        dres = dy if ctx.needs_input_grad[8] else None
        if not ctx.needs_input_grad[0]:
            return (None,) * 8 + (dres,)
        pwb = get_bwd()
        dy = _rc(dy.to(torch.bfloat16))
        if dres is not None:
            dres = dy
        if tokens:
            dy = dy.unsqueeze(2)
        k = pw.KH
        if stride == 2:
            dx = ops.conv_gemm(dy, pwb, stride=1, pad=k - 1 - pad, ups=2)          # zero-insertion transposed conv
        else:
            dx = ops.conv_gemm(dy, pwb, stride=1, pad=k - 1 - pad)
            if ups == 1:                                                            # adjoint of the nearest x2 upsample
                B, H2, W2, C = dx.shape
                dx = dx.view(B, H2 // 2, 2, W2 // 2, 2, C).float().sum(dim=(2, 4)).to(torch.bfloat16)
        return (dx.squeeze(2) if tokens else dx,) + (None,) * 7 + (dres,)


def conv(x, pw, get_bwd, stride=1, pad=None, ups=0, rowbias=None, out_f32=False, residual=None):
    pad = pw.KH // 2 if pad is None else pad
    if residual is not None and not (residual.dtype == torch.bfloat16 and not out_f32):
        return ConvFn.apply(x, pw, get_bwd, stride, pad, ups, rowbias, out_f32, None) + residual
    return ConvFn.apply(x, pw, get_bwd, stride, pad, ups, rowbias, out_f32, residual)


class GateFn(Function):
    """WidthGate / LinearWidthGate (gates.py:15-21, 49-55): y = y0 * gate[b % Bg, c // (C/G)]; dgate in fp32."""

    @staticmethod
    def forward(ctx, y0, gate):
        y0 = _c(y0)
        y1, _ = ops.gate_bwd(y0, y0, gate, want_dgate=False)      # the backward kernel's dx = dy*gate path with dy := y0
        ctx.save_for_backward(y0, gate)
        return y1

    @staticmethod
    def backward(ctx, dy1):
        y0, gate = ctx.saved_tensors
        dx, dgate = ops.gate_bwd(_c(dy1), y0, gate)
        return dx, dgate.to(gate.dtype) if ctx.needs_input_grad[1] else None


# Residual forks.  Every sub-block of the U-Net is `x + f(norm(x))`: x feeds the norm AND the `+ x` at the end, so autograd sums
# two gradients for it with an elementwise launch (~160 per training step).  With fork=True the norm Functions return
# (y, x_res): x_res aliases x and is what the caller hands to the `+ x` epilogue; the Function then receives BOTH gradients and
# the norm's backward kernel adds the residual one while it writes dx (fp32 sum, rounded once) -- no separate add.
def _fork_add(dres, like):
    if dres is None:
        return None
    assert dres.shape == like.shape
    return _rc(dres)                 # (a channel slice of a wider gradient buffer is read in place through its leading dimension)


class GroupNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups: int, eps: float, silu: bool, fork: bool = False):
        y, stats = ops.groupnorm(x, gamma, beta, groups, eps, silu, keep_stats=True)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.meta = (groups, eps, silu)
        if fork:
            ctx.set_materialize_grads(False)
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, beta, stats = ctx.saved_tensors
        groups, eps, silu = ctx.meta
        if dy is None:                                 # (only the residual path carries gradient)
            return dres, None, None, None, None, None, None
        return ops.groupnorm_bwd(x, _c(dy), gamma, beta, groups, eps, silu, stats, add=_fork_add(dres, x)), None, None, None, None, None, None


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float, fork: bool = False):
        x = _c(x)
        y = ops.layernorm(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        if fork:
            ctx.set_materialize_grads(False)
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None
        return ops.layernorm_bwd(x, _c(dy), gamma, ctx.eps, add=_fork_add(dres, x)), None, None, None, None


class SelfAttnFn(Function):
    """qkv [B, L, 3*heads*64] (fused projection output) -> o [B, L, heads*64]; dqkv is written in place by the kernel."""

    @staticmethod
    def forward(ctx, qkv, heads: int):
        qkv = _c(qkv)
        B, L, _ = qkv.shape
        w = heads * 64
        lse = torch.empty(B, heads, L, dtype=torch.float32, device=qkv.device)
        o = ops.attention(qkv[..., :w], qkv[..., w:2 * w], qkv[..., 2 * w:], heads, lse=lse)
        ctx.save_for_backward(qkv, o, lse)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        h, w = ctx.heads, ctx.heads * 64
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv[..., :w], qkv[..., w:2 * w], qkv[..., 2 * w:], o, _c(do), lse, h,
                          dqkv[..., :w], dqkv[..., w:2 * w], dqkv[..., 2 * w:])
        return dqkv, None


class CrossAttnFn(Function):
    """q [B, L, heads*64], kv [B, 77, 2*heads*64] -> o"""

    @staticmethod
    def forward(ctx, q, kv, heads: int):
        q, kv = _c(q), _c(kv)
        B, L, w = q.shape
        lse = torch.empty(B, heads, L, dtype=torch.float32, device=q.device)
        o = ops.attention(q, kv[..., :w], kv[..., w:], heads, lse=lse)
        ctx.save_for_backward(q, kv, o, lse)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse = ctx.saved_tensors
        h, w = ctx.heads, ctx.heads * 64
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ops.attention_bwd(q, kv[..., :w], kv[..., w:], o, _c(do), lse, h, dq, dkv[..., :w], dkv[..., w:])
        return dq, dkv, None


class GegluFn(Function):
    """GEGLUGated.forward (blocks.py:41-50) on the un-interleaved projection output hg = [h | g]."""

    @staticmethod
    def forward(ctx, hg, gate):
        hg = _c(hg)
        out = ops.geglu_fwd(hg, gate)
        ctx.save_for_backward(hg, gate)
        return out

    @staticmethod
    def backward(ctx, dout):
        hg, gate = ctx.saved_tensors
        dhg, dgate = ops.geglu_bwd(hg, _c(dout), gate)
        return dhg, (dgate.to(gate.dtype) if (gate is not None and ctx.needs_input_grad[1]) else None)


class DepthLerpFn(Function):
    """DepthGate.forward (gates.py:36-42): (1-d)*x_in + d*out with d [Bg] tiled over the batch -- one launch forward, one
    (+ its reduction) backward, gradient w.r.t. d in fp32."""

    @staticmethod
    def forward(ctx, x_in, out, d):
        out = _c(out)
        y = ops.depth_lerp(x_in, out, d)
        ctx.save_for_backward(x_in, out, d)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_in, out, d = ctx.saved_tensors
        d_in, d_out, dd = ops.depth_lerp_bwd(_c(dy.to(torch.bfloat16)), x_in, out, d)
        return (d_in if ctx.needs_input_grad[0] else None, d_out if ctx.needs_input_grad[1] else None,
                dd.reshape(d.shape).to(d.dtype) if ctx.needs_input_grad[2] else None)


def depth_lerp(x_in: torch.Tensor, out: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    return DepthLerpFn.apply(x_in, out, d)


class MseFn(Function):
    """F.mse_loss(a.float(), b.float(), reduction="mean") of the training steps' loss terms (trainer.py:1197-1225, 1729-1752)
    on the operands as they are (bf16 activations, strided views, fp32 predictions): fp32 accumulation in a fixed order, the
    gradient w.r.t. ``a`` in one launch, none w.r.t. ``b`` (the teacher side is always detached)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return ops.mse(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ops.mse_bwd(a, b, g), None


def mse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    assert not b.requires_grad, "mse: the second operand is the detached target"
    if a.dtype != b.dtype:
        a, b = a.float(), b.float()
    return MseFn.apply(a, b)


# ------------------------------------------------------------------------------------------------------------------
# Fine-tuning variants: the same ops with WEIGHT gradients.  The trainable tensors are the fp32 masters in
# diffusers' layout; a pruned expert computes on compacted packs, so parameter gradients are produced in the compact
# shape (wgrad = the implicit-GEMM kernel on transposed operands, bias = column sums, norm affine = the backward
# kernels' per-channel partials) and scattered into the live rows / columns of the full-shape ``.grad``.
# ------------------------------------------------------------------------------------------------------------------
def _scatter_rows(full_like: torch.Tensor, vals: torch.Tensor, idx: Optional[torch.Tensor]) -> torch.Tensor:
    if idx is None:
        return vals.to(full_like.dtype).reshape(full_like.shape)
    g = torch.zeros_like(full_like)
    g[idx] = vals.to(full_like.dtype)
    return g


class ConvWFn(Function):
    """y = conv(x, W[live_out][:, live_in]) + b[live_out];  grads: dx, dW (scattered), db (scattered)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pw: PackedWeight, get_bwd, stride: int, pad: int, ups: int, out_f32: bool,
                live_out: Optional[torch.Tensor], live_in: Optional[torch.Tensor], rowbias: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None):
        tokens = x.dim() == 3
        xin = x.unsqueeze(2) if tokens else x
        res = None
        if residual is not None:
            res = residual.unsqueeze(2) if tokens else residual
        y = ops.conv_gemm(xin, pw, stride=stride, pad=pad, ups=ups, out_f32=out_f32, rowbias=rowbias, residual=res)
        ctx.rowbias_shape = None if rowbias is None else tuple(rowbias.shape)
        ctx.save_for_backward(x, weight, bias if bias is not None else weight.new_zeros(0), live_out if live_out is not None else
                              torch.zeros(0, dtype=torch.long, device=x.device),
                              live_in if live_in is not None else torch.zeros(0, dtype=torch.long, device=x.device))
        ctx.meta = (pw, get_bwd, stride, pad, ups, tokens, bias is not None, live_out is not None, live_in is not None)
        return y.squeeze(2) if tokens else y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, live_out, live_in = ctx.saved_tensors
        pw, get_bwd, stride, pad, ups, tokens, has_bias, has_lo, has_li = ctx.meta
        dy = _rc(dy.to(torch.bfloat16))
        dx = dW = db = None
        k = pw.KH
        if ctx.needs_input_grad[0]:
            dy4 = dy.unsqueeze(2) if tokens else dy
            pwb = get_bwd()
            if stride == 2:
                dx = ops.conv_gemm(dy4, pwb, stride=1, pad=k - 1 - pad, ups=2)
            else:
                dx = ops.conv_gemm(dy4, pwb, stride=1, pad=k - 1 - pad)
                if ups == 1:
                    B, H2, W2, C = dx.shape
                    dx = dx.view(B, H2 // 2, 2, W2 // 2, 2, C).float().sum(dim=(2, 4)).to(torch.bfloat16)
            dx = dx.squeeze(2) if tokens else dx
        n_live = live_out.numel() if has_lo else weight.shape[0]
        c_live = live_in.numel() if has_li else weight.shape[1]
        want_db = has_bias and ctx.needs_input_grad[2]
        db_raw = None
        if ctx.needs_input_grad[1]:
            g = ops.conv_wgrad(_c(x), dy, pw.KH, pw.KW, stride, pad, ups, want_db=want_db)   # [Npad, taps, Cpad] fp32 (, [Npad])
            if want_db:
                g, db_raw = g
            g = g[:n_live, :, :c_live].permute(0, 2, 1).reshape(n_live, c_live, pw.KH, pw.KW)
            if weight.dim() == 2:
                g = g.reshape(n_live, c_live)
            if has_lo or has_li:
                full = torch.zeros_like(weight)
                ro = live_out if has_lo else torch.arange(weight.shape[0], device=weight.device)
                ci = live_in if has_li else torch.arange(weight.shape[1], device=weight.device)
                full[ro[:, None], ci[None, :]] = g.to(weight.dtype)
                dW = full
            else:
                dW = g.to(weight.dtype)
        if want_db:
            db = _scatter_rows(bias, (db_raw if db_raw is not None else ops.colsum(dy))[:n_live], live_out if has_lo else None)
        drb = _rowbias_grad(dy, ctx.rowbias_shape) if (ctx.rowbias_shape is not None and ctx.needs_input_grad[11]) else None
        return dx, dW, db, None, None, None, None, None, None, None, None, drb, (dy if ctx.needs_input_grad[12] else None)


def conv_w(x, wparam, bparam, pw, get_bwd, stride=1, pad=None, ups=0, out_f32=False, live_out=None, live_in=None, rowbias=None,
           residual=None):
    pad = pw.KH // 2 if pad is None else pad
    if residual is not None and not (residual.dtype == torch.bfloat16 and not out_f32):
        return ConvWFn.apply(x, wparam, bparam, pw, get_bwd, stride, pad, ups, out_f32, live_out, live_in, rowbias) + residual
    return ConvWFn.apply(x, wparam, bparam, pw, get_bwd, stride, pad, ups, out_f32, live_out, live_in, rowbias, residual)


def _rowbias_grad(dy: torch.Tensor, shape) -> torch.Tensor:
    """gradient of the per-sample output bias (the time-embedding projection added by conv1's epilogue, blocks.py:470-476):
    per-sample column sums of dy, fp32, fixed order (colsum + fold kernels; a torch .sum over 4096 pixels per output may
    take the multi-block path that misbehaves inside replayed graphs, see csrc/loss_ops.hip)"""
    B, N = shape
    g = ops.colsum(dy, per_sample=True)
    return g if g.shape[1] == N else g[:, :N].contiguous()


class ConvPFn(Function):
    """Packed-parameter form (packed_train.py): the trainable weight is the fp32 tensor ``P`` in the kernels' own order
    [N][taps][cin_pad]; ``pw`` is its bf16 shadow.  dW comes out of the weight-gradient kernel in that order: no scatter."""

    @staticmethod
    def forward(ctx, x, P, Pb, pw: PackedWeight, get_bwd, stride: int, pad: int, ups: int, out_f32: bool, residual, rowbias=None):
        tokens = x.dim() == 3
        xin = x.unsqueeze(2) if tokens else x
        res = None
        if residual is not None:
            res = residual.unsqueeze(2) if tokens else residual
        y = ops.conv_gemm(xin, pw, stride=stride, pad=pad, ups=ups, out_f32=out_f32, residual=res, rowbias=rowbias)
        ctx.rowbias_shape = None if rowbias is None else tuple(rowbias.shape)
        ctx.save_for_backward(x)
        ctx.meta = (pw, get_bwd, stride, pad, ups, tokens, tuple(P.shape), Pb is not None)
        ctx.params = (P, Pb)                 # (direct-gradient mode writes into their persistent .grad buffers)
        return y.squeeze(2) if tokens else y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        pw, get_bwd, stride, pad, ups, tokens, pshape, has_bias = ctx.meta
        dy = _rc(dy.to(torch.bfloat16))
        dx = dP = db = None
        k = pw.KH
        if ctx.needs_input_grad[0]:
            dy4 = dy.unsqueeze(2) if tokens else dy
            pwb = get_bwd()
            if stride == 2:
                dx = ops.conv_gemm(dy4, pwb, stride=1, pad=k - 1 - pad, ups=2)
            else:
                dx = ops.conv_gemm(dy4, pwb, stride=1, pad=k - 1 - pad)
                if ups == 1:
                    B, H2, W2, C = dx.shape
                    dx = dx.view(B, H2 // 2, 2, W2 // 2, 2, C).float().sum(dim=(2, 4)).to(torch.bfloat16)
            dx = dx.squeeze(2) if tokens else dx
        want_db = has_bias and ctx.needs_input_grad[2]
        P, Pb = ctx.params
        if ops.GRAD_DIRECT and ctx.needs_input_grad[1] and P.grad is not None and (not want_db or Pb.grad is not None):
            # direct-gradient mode (ops.GRAD_DIRECT): kernel / deferred fold write into the parameters' own persistent,
            # zero-padded gradient buffers; autograd gets None for them
            assert tuple(P.grad.shape) == pshape and P.grad.is_contiguous()
            ops.conv_wgrad(_c(x), dy, pw.KH, pw.KW, stride, pad, ups, out=P.grad, want_db=want_db, db_out=Pb.grad if want_db else None)
        elif ctx.needs_input_grad[1]:
            cx = x.shape[-1]
            # straight into the packed layout [N, taps, cin_pad]: the kernel (or the slab fold) writes the first Cx columns,
            # the pad columns stay zero; the bias gradient is a by-product of the same kernel
            dP = (torch.zeros if pshape[2] != cx else torch.empty)(pshape, dtype=torch.float32, device=dy.device)
            r = ops.conv_wgrad(_c(x), dy, pw.KH, pw.KW, stride, pad, ups, out=dP, want_db=want_db)
            if want_db:
                db = r[1][:pshape[0]]
        elif want_db:
            db = ops.colsum(dy)[:pshape[0]]
        drb = _rowbias_grad(dy, ctx.rowbias_shape) if (ctx.rowbias_shape is not None and ctx.needs_input_grad[10]) else None
        return dx, dP, db, None, None, None, None, None, None, (dy if ctx.needs_input_grad[9] else None), drb


def conv_p(x, P, Pb, pw, get_bwd, stride=1, pad=None, ups=0, out_f32=False, residual=None, rowbias=None):
    pad = pw.KH // 2 if pad is None else pad
    if residual is not None and not (residual.dtype == torch.bfloat16 and not out_f32):
        return ConvPFn.apply(x, P, Pb, pw, get_bwd, stride, pad, ups, out_f32, None, rowbias) + residual
    return ConvPFn.apply(x, P, Pb, pw, get_bwd, stride, pad, ups, out_f32, residual, rowbias)


class GroupNormWFn(Function):
    """GroupNorm(+SiLU) with trainable affine; `live` = channel indices of a compacted tensor (norm2 of a pruned resnet)."""

    @staticmethod
    def forward(ctx, x, gamma_p, beta_p, gamma, beta, groups: int, eps: float, silu: bool, C: int, live: Optional[torch.Tensor],
                fork: bool = False):
        y, stats = ops.groupnorm(x, gamma, beta, groups, eps, silu, C=C, keep_stats=True)
        ctx.save_for_backward(x, gamma, beta, stats, gamma_p, live if live is not None else torch.zeros(0, dtype=torch.long, device=x.device))
        ctx.meta = (groups, eps, silu, C, live is not None)
        ctx.params = (gamma_p, beta_p)
        if fork:                                       # (residual fork: see GroupNormFn)
            ctx.set_materialize_grads(False)
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, beta, stats, gamma_p, live = ctx.saved_tensors
        groups, eps, silu, C, has_live = ctx.meta
        want = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gp, bp = ctx.params
        nones = (None,) * 10
        if dy is None:
            return (dres,) + nones
        add = _fork_add(dres, x)
        if ops.GRAD_DIRECT and want and not has_live and gp.grad is not None and bp.grad is not None and gp.grad.numel() == C:
            dx, _, _ = ops.groupnorm_bwd(x, _c(dy), gamma, beta, groups, eps, silu, stats, C=C, want_pgrad=True, pgrad_out=(gp.grad, bp.grad),
                                         add=add)
            return (dx,) + nones
        res = ops.groupnorm_bwd(x, _c(dy), gamma, beta, groups, eps, silu, stats, C=C, want_pgrad=want, add=add)
        if not want:
            return (res,) + nones
        dx, dgamma, dbeta = res
        idx = live if has_live else None
        return (dx, _scatter_rows(gamma_p, dgamma, idx), _scatter_rows(gamma_p, dbeta, idx)) + nones[:8]


class LayerNormWFn(Function):
    @staticmethod
    def forward(ctx, x, gamma_p, beta_p, gamma, beta, eps: float, fork: bool = False):
        x = _c(x)
        y = ops.layernorm(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, gamma_p)
        ctx.eps = eps
        ctx.params = (gamma_p, beta_p)
        if fork:                                       # (residual fork: see GroupNormFn)
            ctx.set_materialize_grads(False)
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, gamma_p = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None, None, None
        dy = _c(dy)
        dx = ops.layernorm_bwd(x, dy, gamma, ctx.eps, add=_fork_add(dres, x))
        dg = db = None
        gp, bp = ctx.params
        if ops.GRAD_DIRECT and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and gp.grad is not None and bp.grad is not None \
                and gp.dtype == torch.float32:
            ops.layernorm_pgrad(x, dy, ctx.eps, pgrad_out=(gp.grad, bp.grad))
        elif ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dgamma, dbeta = ops.layernorm_pgrad(x, dy, ctx.eps)
            dg, db = dgamma.to(gamma_p.dtype), dbeta.to(gamma_p.dtype)
        return dx, dg, db, None, None, None, None
