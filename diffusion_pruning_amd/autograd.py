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


class ConvFn(Function):
    """y = conv(x, w) + bias (+ rowbias[b]) on [B,H,W,C] (or tokens [B,L,C]); backward = dgrad through the same kernel."""

    @staticmethod
    def forward(ctx, x, pw: PackedWeight, get_bwd: Callable[[], PackedWeight], stride: int, pad: int, ups: int,
                rowbias: Optional[torch.Tensor], out_f32: bool):
        tokens = x.dim() == 3
        xin = x.unsqueeze(2) if tokens else x
        y = ops.conv_gemm(xin, pw, stride=stride, pad=pad, ups=ups, rowbias=rowbias, out_f32=out_f32)
        ctx.meta = (pw, get_bwd, stride, pad, ups, tokens)
        return y.squeeze(2) if tokens else y

    @staticmethod
    def backward(ctx, dy):
        pw, get_bwd, stride, pad, ups, tokens = ctx.meta
        if not ctx.needs_input_grad[0]:
            return (None,) * 8
        pwb = get_bwd()
        dy = _c(dy.to(torch.bfloat16))
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
        return (dx.squeeze(2) if tokens else dx,) + (None,) * 7


def conv(x, pw, get_bwd, stride=1, pad=None, ups=0, rowbias=None, out_f32=False):
    pad = pw.KH // 2 if pad is None else pad
    return ConvFn.apply(x, pw, get_bwd, stride, pad, ups, rowbias, out_f32)


class GateFn(Function):
    """WidthGate / LinearWidthGate (gates.py:15-21, 49-55): y = y0 * gate[b % Bg, c // (C/G)]; dgate in fp32."""

    @staticmethod
    def forward(ctx, y0, gate):
        y0 = _c(y0)
        y1, _ = ops.gate_bwd(y0, y0, gate)      # the backward kernel's dx = dy*gate path with dy := y0
        ctx.save_for_backward(y0, gate)
        return y1

    @staticmethod
    def backward(ctx, dy1):
        y0, gate = ctx.saved_tensors
        dx, dgate = ops.gate_bwd(_c(dy1), y0, gate)
        return dx, dgate.to(gate.dtype) if ctx.needs_input_grad[1] else None


class GroupNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups: int, eps: float, silu: bool):
        y, stats = ops.groupnorm(x, gamma, beta, groups, eps, silu, keep_stats=True)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.meta = (groups, eps, silu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        groups, eps, silu = ctx.meta
        return ops.groupnorm_bwd(x, _c(dy), gamma, beta, groups, eps, silu, stats), None, None, None, None, None


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float):
        x = _c(x)
        y = ops.layernorm(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        return ops.layernorm_bwd(x, _c(dy), gamma, ctx.eps), None, None, None


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


def depth_lerp(x_in: torch.Tensor, out: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """DepthGate.forward (gates.py:36-42): (1-d)*x_in + d*out with d [Bg] tiled over the batch; tiny elementwise glue,
    differentiable w.r.t. d through PyTorch's own elementwise kernels."""
    B = out.shape[0]
    dm = d.reshape(-1)
    if dm.shape[0] != B:
        dm = dm.repeat(B // dm.shape[0])
    dm = dm.view(B, *([1] * (out.dim() - 1)))
    return ((1.0 - dm) * x_in.float() + dm * out.float()).to(torch.bfloat16)
