"""TEST-ONLY stand-in for diffusion_pruning_amd.ops, written in plain PyTorch fp32 on CPU.

Purpose: exercise the *host logic* of the product (weight packing/compaction, GEGLU interleave, batched temb/KV GEMMs,
GroupNorm-beta border correction, gate plumbing, skip-concat handling) in the `-m "not gpu"` suite, where no GPU
exists.  It consumes the very PackedWeight objects the product builds and re-implements the documented semantics of
include/aptp_hip.h.  It is never imported by the package; on a GPU the real kernels run (tests/test_unet_gpu.py).
"""
import torch
import torch.nn.functional as F

from diffusion_pruning_amd import ops as real_ops
from diffusion_pruning_amd._lib import ACT_GEGLU, ACT_NONE, ACT_SILU


# EXACT = True (install(..., exact=True)): no bf16 rounding anywhere -- activations and packed weights stay fp32
# (ops.ACT_DTYPE), so the product's host logic can be compared with the fp32 oracle at 1e-5 instead of the 2e-2 that bf16
# storage costs.
EXACT = False


def _bf(x):
    return x.float() if EXACT else x.to(torch.bfloat16)


def conv_gemm(x, pw, *, stride=1, pad=None, ups=0, out=None, rowbias=None, colgate=None, gate_group=0, act=ACT_NONE,
              corr=None, residual=None, depth=None, depth_in=None, out_f32=False, split_k=None, tile=0,
              rowstats=False, ln=None, colstats=False, x2=None, prefetch=None, gn=None):
    if gn is not None:
        gamma, beta, groups, eps_, silu_, Cn = gn
        h = conv_gemm(x, pw, stride=stride, pad=pad, ups=ups, rowbias=rowbias, split_k=split_k, tile=tile)
        return groupnorm(h, gamma, beta, groups, eps_, silu_, C=Cn)
    B, H, W, C = x.shape
    assert C == pw.Cin and x.dtype == real_ops.ACT_DTYPE
    if pad is None:
        pad = pw.KH // 2
    if pw.geglu:
        act = ACT_GEGLU
    wflat = pw.w.float().reshape(pw.N, -1)
    w = wflat[:, :pw.KH * pw.KW * pw.cin_pad].reshape(pw.N, pw.KH * pw.KW, pw.cin_pad)[:, :, :C].reshape(pw.N, pw.KH, pw.KW, C).permute(0, 3, 1, 2)
    xin = x.float().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xin, w, None, stride=stride, padding=pad)
    assert (x2 is not None) == (pw.Cin2 > 0)
    if x2 is not None:          # second-operand K-segment: a 1x1 convolution of x2 at the output pixel
        w2 = wflat[:, pw.KH * pw.KW * pw.cin_pad:][:, :x2.shape[3]]
        y = y + F.conv2d(x2.float().permute(0, 3, 1, 2), w2[:, :, None, None])
    if ln is not None:          # LayerNorm folded into the packed weights (include/aptp_hip.h, ln_stats)
        stats, eps = ln
        assert pw.ln_colsum is not None and pw.KH == 1
        tot = stats.sum(0)                         # [M, 4]: two (sum, sumsq) slots per element
        mean = (tot[:, 0] + tot[:, 2]) / C
        rstd = torch.rsqrt(((tot[:, 1] + tot[:, 3]) / C - mean * mean).clamp_min(0) + eps)
        mean, rstd = (t.reshape(B, 1, y.shape[2], y.shape[3]) for t in (mean, rstd))
        y = rstd * (y - mean * pw.ln_colsum[None, :, None, None])
    else:
        assert pw.ln_colsum is None
    if pw.bias is not None:
        y = y + pw.bias[None, :, None, None]
    if rowbias is not None:
        y = y + rowbias[:, :pw.N, None, None]
    Ho, Wo = y.shape[2:]
    if act == ACT_GEGLU:
        y = y.reshape(B, pw.N // 32, 2, 16, Ho, Wo)
        h, g = y[:, :, 0].reshape(B, pw.N // 2, Ho, Wo), y[:, :, 1].reshape(B, pw.N // 2, Ho, Wo)
        if colgate is not None:
            m = colgate.repeat_interleave(gate_group, dim=1).repeat(B // colgate.shape[0], 1)[:, :, None, None]
            h, g = h * m, g * m
        y = h * F.gelu(g)
    else:
        if colgate is not None:
            m = colgate.repeat_interleave(gate_group, dim=1).repeat(B // colgate.shape[0], 1)[:, :, None, None]
            y = y * m
        if act == ACT_SILU:
            y = F.silu(y)
    if corr is not None:
        cr = torch.ones(Ho, dtype=torch.long); cr[0] = 0; cr[-1] = 2
        cc = torch.ones(Wo, dtype=torch.long); cc[0] = 0; cc[-1] = 2
        cmap = cr[:, None] * 3 + cc[None, :]
        add = corr[:, cmap]                       # [cB, Ho, Wo, N]
        y = y + add.permute(0, 3, 1, 2).repeat(B // corr.shape[0], 1, 1, 1)
    if residual is not None:
        y = y + residual.float().permute(0, 3, 1, 2)
    if depth is not None:
        d = depth.repeat(B // depth.shape[0])[:, None, None, None]
        y = (1 - d) * depth_in.float().permute(0, 3, 1, 2) + d * y
    y = y.permute(0, 2, 3, 1)
    y = y.contiguous() if out_f32 else _bf(y).contiguous()
    stats = None
    if rowstats:
        yf = y.float().reshape(-1, y.shape[-1])
        zz = torch.zeros_like(yf[:, 0])
        stats = torch.stack([yf.sum(1), (yf * yf).sum(1), zz, zz], 1).reshape(1, -1, 4).contiguous()
    if out is not None:
        out.copy_(y)
        y = out
    return (y, stats) if rowstats else y


def linear(x, pw, **kw):
    out = kw.pop("out", None)
    for k in ("residual", "depth_in"):
        if kw.get(k) is not None:
            kw[k] = kw[k].unsqueeze(2)
    y = conv_gemm(x.unsqueeze(2), pw, pad=0, out=None if out is None else out.unsqueeze(2), **kw)
    if "rowstats" in kw:
        return (y[0].squeeze(2), y[1]) if kw["rowstats"] else (y.squeeze(2), None)
    return y.squeeze(2)


def groupnorm(x, gamma, beta, groups, eps, silu, C=None, out=None):
    B, H, W, Cp = x.shape
    C = Cp if C is None else C
    y = F.group_norm(x.float()[..., :C].permute(0, 3, 1, 2), groups, gamma, beta, eps)
    if silu:
        y = F.silu(y)
    res = torch.zeros(B, H, W, Cp, dtype=real_ops.ACT_DTYPE)
    res[..., :C] = _bf(y.permute(0, 2, 3, 1))
    return res


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    return _bf(F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps))


def attention(q, k, v, heads, scale=None, out=None, lse=None):
    """same rounding points as attn_fwd_kernel: scores and the softmax denominator in fp32, the un-normalised probabilities
    rounded to bf16 before P.V (bf16 rounding is relative, so rounding against the final row maximum instead of the
    kernel's running maximum differs only at fp32 level), output rounded to bf16"""
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    sc = (1.0 / 8.0) if scale is None else scale

    def hd(t, L):
        return t.float().reshape(B, L, heads, 64).transpose(1, 2)
    qh, kh, vh = hd(q, Lq), hd(k, Lk), hd(v, Lk)
    o = torch.empty(B, heads, Lq, 64)
    step = max(1, (1 << 24) // max(1, Lk))               # bound the [rows, Lk] score block (L = 4096 at SD-2.1 level 0)
    for b in range(B):
        for h in range(heads):
            for r0 in range(0, Lq, step):
                s_ = (qh[b, h, r0:r0 + step] @ kh[b, h].t()) * sc
                p = torch.exp(s_ - s_.max(dim=1, keepdim=True)[0])
                o[b, h, r0:r0 + step] = (_bf(p).float() @ vh[b, h]) / p.sum(dim=1, keepdim=True)
    return _bf(o.transpose(1, 2).reshape(B, Lq, heads * 64))


def install(monkeypatch, exact: bool = False):
    """Route the product's ops.* calls to this emulator for the duration of a test.  exact=True: fp32 storage, no rounding."""
    for name in ("conv_gemm", "linear", "groupnorm", "layernorm", "attention"):
        monkeypatch.setattr(real_ops, name, globals()[name])
    monkeypatch.setattr(real_ops, "ACT_DTYPE", torch.float32 if exact else torch.bfloat16)
    monkeypatch.setitem(globals(), "EXACT", bool(exact))
