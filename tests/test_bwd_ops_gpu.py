"""GPU parity of the backward kernels (through the C ABI) against PyTorch fp32 autograd on the same bf16-rounded
operands.  Tolerance: rel-L2 <= 8e-3 per op (bf16 operands and bf16 gradient outputs; attention recomputes P in bf16)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 8e-3


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def ops(cuda):
    from diffusion_pruning_amd import ops as o
    o._lib.load()
    return o


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 96, 3, 1, 0), (2, 16, 16, 64, 64, 3, 2, 0), (2, 8, 8, 64, 64, 3, 1, 1),
                                  (2, 8, 8, 192, 128, 1, 1, 0), (1, 8, 8, 320, 640, 3, 1, 0)])
def test_conv_dgrad(ops, cuda, case):
    B, H, W, Cin, Cout, k, stride, ups = case
    g = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float().requires_grad_()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).bfloat16().float()
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    y = F.conv2d(xin, w, None, stride=stride, padding=k // 2)
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy)
    pwb = ops.pack_weight_dgrad(w, device=cuda)
    dyd = nhwc(dy).bfloat16().to(cuda)
    if stride == 2:
        dx = ops.conv_gemm(dyd, pwb, stride=1, pad=k - 1 - k // 2, ups=2)
    else:
        dx = ops.conv_gemm(dyd, pwb, stride=1, pad=k - 1 - k // 2)
        if ups:
            Bq, H2, W2, C = dx.shape
            dx = dx.float().view(Bq, H2 // 2, 2, W2 // 2, 2, C).sum(dim=(2, 4))
    got = dx.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == x.grad.shape
    assert rel_l2(got, x.grad) <= TOL


@pytest.mark.parametrize("case", [(2, 8, 8, 64, 32, True, 1e-5), (2, 16, 16, 320, 32, True, 1e-5), (1, 8, 8, 2560, 32, True, 1e-5),
                                  (2, 8, 8, 128, 32, False, 1e-6)])
def test_groupnorm_bwd(ops, cuda, case):
    B, H, W, C, G, silu, eps = case
    g = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    x = (torch.randn(B, C, H, W, generator=g) * 1.7 + 0.3).bfloat16().float().requires_grad_()
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.3 * torch.randn(C, generator=g)
    y = F.group_norm(x, G, gamma, beta, eps)
    if silu:
        y = F.silu(y)
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy)
    xd = nhwc(x.detach()).bfloat16().to(cuda)
    yk, stats = ops.groupnorm(xd, gamma.to(cuda), beta.to(cuda), G, eps, silu, keep_stats=True)
    assert rel_l2(yk.float().cpu().permute(0, 3, 1, 2), y.detach()) <= 4e-3
    dx = ops.groupnorm_bwd(xd, nhwc(dy).bfloat16().to(cuda), gamma.to(cuda), beta.to(cuda), G, eps, silu, stats)
    assert rel_l2(dx.float().cpu().permute(0, 3, 1, 2), x.grad) <= TOL
    # keep_stats on a small map = the one-launch kernel writing the sums into chunk 0 of the partials; the three-launch form
    # spreads them over the chunks: same totals, same output, same backward
    ops.GN_STATS_ONE_LAUNCH = False
    try:
        y3, stats3 = ops.groupnorm(xd, gamma.to(cuda), beta.to(cuda), G, eps, silu, keep_stats=True)
    finally:
        ops.GN_STATS_ONE_LAUNCH = True
    assert stats.shape == stats3.shape and rel_l2(stats.sum(1), stats3.sum(1)) <= 1e-5
    assert rel_l2(yk.float(), y3.float()) <= 2e-3
    dx3 = ops.groupnorm_bwd(xd, nhwc(dy).bfloat16().to(cuda), gamma.to(cuda), beta.to(cuda), G, eps, silu, stats3)
    assert rel_l2(dx.float(), dx3.float()) <= 2e-3


@pytest.mark.parametrize("rows,C", [(33, 64), (128, 320), (65, 1280)])
def test_layernorm_bwd(ops, cuda, rows, C):
    g = torch.Generator().manual_seed(rows * C)
    x = (torch.randn(1, rows, C, generator=g) * 1.5 + 0.4).bfloat16().float().requires_grad_()
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.3 * torch.randn(C, generator=g)
    y = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy)
    dx = ops.layernorm_bwd(x.detach().bfloat16().to(cuda), dy.bfloat16().to(cuda), gamma.to(cuda), 1e-5)
    assert rel_l2(dx.float().cpu(), x.grad) <= TOL


def test_residual_fork_gradient_is_added_inside_the_norm_backward(ops, cuda):
    """`x + f(norm(x))`: with fork=True the norm Functions hand out an alias of x for the `+ x`, and the backward kernels add the
    gradient that returns over it while writing dx (AptpGroupNormBwdParams.add / AptpLayerNormBwdParams.add).  Kernel: equals the
    un-fused dx plus the residual gradient to one bf16 rounding; Function: the gradient of x equals torch autograd's on the same
    graph, and an unused alias costs nothing"""
    from diffusion_pruning_amd import autograd as AG
    g = torch.Generator().manual_seed(21)
    # GroupNorm (+SiLU), channels-last [B, H, W, C]
    x = (torch.randn(2, 16, 16, 320, generator=g) * 1.3).to(cuda, torch.bfloat16)
    dy = torch.randn(2, 16, 16, 320, generator=g).to(cuda, torch.bfloat16)
    r = torch.randn(2, 16, 16, 320, generator=g).to(cuda, torch.bfloat16)
    gamma = (1.0 + 0.2 * torch.randn(320, generator=g)).to(cuda)
    beta = (0.3 * torch.randn(320, generator=g)).to(cuda)
    _, stats = ops.groupnorm(x, gamma, beta, 32, 1e-5, True, keep_stats=True)
    plain = ops.groupnorm_bwd(x, dy, gamma, beta, 32, 1e-5, True, stats)
    fused = ops.groupnorm_bwd(x, dy, gamma, beta, 32, 1e-5, True, stats, add=r)
    want = plain.float() + r.float()
    assert float((fused.float() - want).abs().max()) <= 2 ** -7 * float(want.abs().max())       # two roundings vs one
    assert rel_l2(fused.float(), want) <= 4e-3
    # LayerNorm rows
    xl = (torch.randn(2, 77, 640, generator=g) * 1.5 + 0.2).to(cuda, torch.bfloat16)
    dyl = torch.randn(2, 77, 640, generator=g).to(cuda, torch.bfloat16)
    rl = torch.randn(2, 77, 640, generator=g).to(cuda, torch.bfloat16)
    gl = (1.0 + 0.2 * torch.randn(640, generator=g)).to(cuda)
    bl = (0.3 * torch.randn(640, generator=g)).to(cuda)
    wantl = ops.layernorm_bwd(xl, dyl, gl, 1e-5).float() + rl.float()
    assert rel_l2(ops.layernorm_bwd(xl, dyl, gl, 1e-5, add=rl).float(), wantl) <= 4e-3
    # Function level: y = x + 2 * norm(x)  (the residual path carries its own gradient)
    for kind in ("gn", "ln"):
        xa = (x if kind == "gn" else xl).clone().requires_grad_()
        if kind == "gn":
            n, xr = AG.GroupNormFn.apply(xa, gamma, beta, 32, 1e-5, True, True)
        else:
            n, xr = AG.LayerNormFn.apply(xa, gl, bl, 1e-5, True)
        assert xr.data_ptr() == xa.data_ptr()
        up = dy if kind == "gn" else dyl
        (xr * 1.0 + 2.0 * n).backward(up)
        xb = xa.detach().clone().requires_grad_()
        nb = AG.GroupNormFn.apply(xb, gamma, beta, 32, 1e-5, True) if kind == "gn" else AG.LayerNormFn.apply(xb, gl, bl, 1e-5)
        (xb * 1.0 + 2.0 * nb).backward(up)
        assert rel_l2(xa.grad.float(), xb.grad.float()) <= 4e-3
        # alias unused: only the norm's gradient
        xc = xa.detach().clone().requires_grad_()
        nc, _ = (AG.GroupNormFn.apply(xc, gamma, beta, 32, 1e-5, True, True) if kind == "gn" else AG.LayerNormFn.apply(xc, gl, bl, 1e-5, True))
        nc.backward(up)
        ref = (plain if kind == "gn" else ops.layernorm_bwd(xl, dyl, gl, 1e-5))
        assert torch.equal(xc.grad, ref)


def test_gate_bwd_and_apply(ops, cuda):
    g = torch.Generator().manual_seed(3)
    B, H, W, C, G = 4, 8, 8, 64, 32
    y0 = torch.randn(B, H, W, C, generator=g).bfloat16().float()
    gate = torch.rand(2, G, generator=g).requires_grad_()
    mask = gate.repeat_interleave(C // G, dim=1).repeat(2, 1)[:, None, None, :]
    y1 = y0 * mask
    dy1 = torch.randn(y1.shape, generator=g).bfloat16().float()
    y1.backward(dy1)
    dx, dgate = ops.gate_bwd(dy1.bfloat16().to(cuda), y0.bfloat16().to(cuda), gate.detach().to(cuda))
    assert rel_l2(dx.float().cpu(), dy1 * mask.detach()) <= 4e-3
    assert rel_l2(dgate.cpu(), gate.grad) <= 2e-3
    # the same kernel doubles as the forward gate multiply (dy := y0)
    fwd, _ = ops.gate_bwd(y0.bfloat16().to(cuda), y0.bfloat16().to(cuda), gate.detach().to(cuda))
    assert rel_l2(fwd.float().cpu(), y1.detach()) <= 4e-3
    # 60 gate entries (fused QKV of a 20-head layer): 4 threads per group
    y0 = torch.randn(2, 4, 4, 3840, generator=g).bfloat16().float()
    gate = torch.rand(2, 60, generator=g).requires_grad_()
    mask = gate.repeat_interleave(64, dim=1)[:, None, None, :]
    dy = torch.randn(y0.shape, generator=g).bfloat16().float()
    (y0 * mask).backward(dy)
    dx, dgate = ops.gate_bwd(dy.bfloat16().to(cuda), y0.bfloat16().to(cuda), gate.detach().to(cuda))
    assert rel_l2(dx.float().cpu(), dy * mask.detach()) <= 4e-3 and rel_l2(dgate.cpu(), gate.grad) <= 2e-3


@pytest.mark.parametrize("use_gate,C", [(True, 256), (False, 256), (True, 5120)])
def test_geglu_fwd_bwd(ops, cuda, use_gate, C):
    g = torch.Generator().manual_seed(5)
    B, L = 2, 40
    hg = torch.randn(B, L, 2 * C, generator=g).bfloat16().float().requires_grad_()
    gate = torch.rand(2, 32, generator=g).requires_grad_() if use_gate else None
    h, gg = hg.chunk(2, dim=-1)
    if use_gate:
        m = gate.repeat_interleave(C // 32, dim=1)[:, None, :]
        out = (h * m) * F.gelu(gg * m)
    else:
        out = h * F.gelu(gg)
    dout = torch.randn(out.shape, generator=g).bfloat16().float()
    out.backward(dout)
    gd = gate.detach().to(cuda) if use_gate else None
    hgd = hg.detach().bfloat16().to(cuda)
    o = ops.geglu_fwd(hgd, gd)
    assert rel_l2(o.float().cpu(), out.detach()) <= 4e-3
    dhg, dgate = ops.geglu_bwd(hgd, dout.bfloat16().to(cuda), gd)
    assert rel_l2(dhg.float().cpu(), hg.grad) <= TOL
    if use_gate:
        assert rel_l2(dgate.cpu(), gate.grad) <= 3e-3


@pytest.mark.parametrize("case", [(1, 1, 64, 64), (2, 2, 256, 256), (2, 3, 200, 77), (1, 2, 1024, 1024), (1, 2, 100, 130)])
def test_attention_bwd(ops, cuda, case):
    B, h, Lq, Lk = case
    g = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    q = torch.randn(B, Lq, h * 64, generator=g).bfloat16().float().requires_grad_()
    k = torch.randn(B, Lk, h * 64, generator=g).bfloat16().float().requires_grad_()
    v = torch.randn(B, Lk, h * 64, generator=g).bfloat16().float().requires_grad_()

    def heads(t, L):
        return t.view(B, L, h, 64).transpose(1, 2)
    o = F.scaled_dot_product_attention(heads(q, Lq), heads(k, Lk), heads(v, Lk)).transpose(1, 2).reshape(B, Lq, h * 64)
    do = torch.randn(o.shape, generator=g).bfloat16().float()
    o.backward(do)
    qd, kd, vd = (t.detach().bfloat16().to(cuda) for t in (q, k, v))
    lse = torch.empty(B, h, Lq, dtype=torch.float32, device=cuda)
    od = ops.attention(qd, kd, vd, h, lse=lse)
    assert rel_l2(od.float().cpu(), o.detach()) <= 6e-3
    # lse is log2-domain of the scaled scores
    s = (heads(q.detach(), Lq) @ heads(k.detach(), Lk).transpose(-1, -2)) / 8.0
    assert torch.allclose(lse.cpu(), torch.logsumexp(s, dim=-1) * math.log2(math.e), atol=2e-2, rtol=1e-3)
    dq, dk, dv = (torch.empty_like(t) for t in (qd, kd, vd))
    ops.attention_bwd(qd, kd, vd, od, do.bfloat16().to(cuda), lse, h, dq, dk, dv)
    assert rel_l2(dv.float().cpu(), v.grad) <= 1e-2
    assert rel_l2(dq.float().cpu(), q.grad) <= 1.5e-2
    assert rel_l2(dk.float().cpu(), k.grad) <= 1.5e-2
    # dK/dV with the query range split over workgroups (the rule switches it on by itself where few keys leave the chip idle:
    # cross-attention); forced here for every case with enough query tiles: same sums up to the fp32 association, deterministic
    if (Lq + 63) // 64 >= 3:
        dq2, dk2, dv2 = (torch.full_like(t, 7.0) for t in (qd, kd, vd))
        ops.attention_bwd(qd, kd, vd, od, do.bfloat16().to(cuda), lse, h, dq2, dk2, dv2, q_split=3)
        assert torch.equal(dq2, dq)
        assert rel_l2(dk2.float(), dk.float()) <= 4e-3 and rel_l2(dv2.float(), dv.float()) <= 4e-3
        assert rel_l2(dv2.float().cpu(), v.grad) <= 1e-2 and rel_l2(dk2.float().cpu(), k.grad) <= 1.5e-2
        dk3, dv3 = torch.empty_like(dk2), torch.empty_like(dv2)
        ops.attention_bwd(qd, kd, vd, od, do.bfloat16().to(cuda), lse, h, torch.empty_like(dq2), dk3, dv3, q_split=3)
        assert torch.equal(dk3, dk2) and torch.equal(dv3, dv2)


@pytest.mark.parametrize("B,dB,HW,C,wide", [(4, 2, 64, 64, 0), (2, 2, 1024, 320, 320), (4, 1, 256, 1280, 1280), (2, 1, 30, 2560, 0)])
def test_depth_lerp_fwd_bwd(ops, cuda, B, dB, HW, C, wide):
    """DepthGate in its training form (gates.py:36-42) in one launch each way; x_in may be a channel slice of a wider
    buffer (ld > C: the un-sliced skip-concat of an up-block resnet); the gate is tiled over the batch (CFG)."""
    from diffusion_pruning_amd import autograd as AG
    g = torch.Generator().manual_seed(B * 1000 + C)
    buf = torch.randn(B, HW, C + wide, generator=g).bfloat16()
    xo = torch.randn(B, HW, C, generator=g).bfloat16()
    d = torch.rand(dB, generator=g)
    dy = torch.randn(B, HW, C, generator=g).bfloat16()
    xi_ref = buf[..., :C].float().requires_grad_()
    xo_ref = xo.float().requires_grad_()
    d_ref = d.clone().requires_grad_()
    dm = d_ref.repeat(B // dB).view(B, 1, 1)
    y_ref = (1 - dm) * xi_ref + dm * xo_ref
    y_ref.backward(dy.float())
    bufd = buf.to(cuda)
    xi = bufd[..., :C].requires_grad_()
    xod = xo.to(cuda).requires_grad_()
    dd = d.to(cuda).requires_grad_()
    y = AG.depth_lerp(xi, xod, dd)
    assert rel_l2(y.float().cpu(), y_ref.detach()) <= 4e-3
    y.backward(dy.to(cuda))
    torch.cuda.synchronize()
    assert rel_l2(xi.grad.float().cpu(), xi_ref.grad) <= 4e-3
    assert rel_l2(xod.grad.float().cpu(), xo_ref.grad) <= 4e-3
    assert rel_l2(dd.grad.float().cpu(), d_ref.grad) <= 1e-3
    # deterministic two-stage reduction: a second backward gives bit-identical gate gradients
    d2 = d.to(cuda).requires_grad_()
    AG.depth_lerp(bufd[..., :C], xo.to(cuda), d2).backward(dy.to(cuda))
    assert torch.equal(d2.grad, dd.grad)


def test_conv_with_fused_residual_matches_separate_add(ops, cuda):
    """AG.conv(residual=): the residual add runs in the GEMM epilogue, its gradient is dy itself"""
    from diffusion_pruning_amd import autograd as AG
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 320, generator=g).bfloat16().to(cuda).requires_grad_()
    r = torch.randn(2, 64, 640, generator=g).bfloat16().to(cuda).requires_grad_()
    w = (torch.randn(640, 320, generator=g) / 18).to(cuda)
    pw = ops.pack_weight(w, torch.zeros(640), device=cuda)
    get = lambda: ops.pack_weight_dgrad(w, device=cuda)
    dy = torch.randn(2, 64, 640, generator=g).bfloat16().to(cuda)
    y = AG.conv(x, pw, get, pad=0, residual=r)
    y.backward(dy)
    gx, gr = x.grad.clone(), r.grad.clone()
    x.grad = r.grad = None
    y2 = AG.conv(x, pw, get, pad=0) + r
    y2.backward(dy)
    assert rel_l2(y.float(), y2.float()) <= 4e-3                 # one bf16 rounding instead of two
    assert torch.equal(gx, x.grad) and torch.equal(gr, r.grad)


WGRAD_CASES = [  # (B, H, W, C, N, k)
    (2, 64, 64, 64, 64, 3),      # W = 64: two 32-pixel steps per image row
    (1, 32, 32, 320, 160, 3),    # W = 32, ragged tiles (C = 5 x 64, N = 2.5 x 64)
    (2, 16, 16, 128, 200, 3),    # W = 16: a step covers two image rows; N not a multiple of 64
    (4, 8, 8, 72, 64, 3),        # W = 8: four rows per step; C = 72 (a compacted width)
    (2, 32, 32, 320, 320, 1),    # 1x1 convolution
    (1, 308, 1, 1024, 136, 1),   # linear layer over 4 x 77 text tokens: ragged pixel count (308 = 9 x 32 + 20)
    (1, 4, 1, 320, 1280, 1),     # the time-embedding MLP: 4 rows
]


@pytest.mark.parametrize("case", WGRAD_CASES)
@pytest.mark.parametrize("split", [None, 1, 3])
def test_conv_wgrad_kernel(ops, cuda, case, split):
    """aptp_conv_wgrad (transposed LDS reads, one halo per step for all nine taps) vs PyTorch's weight gradient of the same
    convolution on the same bf16-rounded operands; also against the GEMM-on-transposed-copies fallback."""
    B, H, W, C, N, k = case
    g = torch.Generator().manual_seed(B * 7919 + H * 131 + C + N + k)
    x = torch.randn(B, C, H, W, generator=g).bfloat16().float()
    w = torch.zeros(N, C, k, k, requires_grad=True)
    dy = torch.randn(B, N, H, W, generator=g).bfloat16().float()
    F.conv2d(x, w, None, stride=1, padding=k // 2).backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(N, k * k, C)                    # [N, taps, C]: the packed-weight order
    xd, dyd = nhwc(x).to(cuda).bfloat16(), nhwc(dy).to(cuda).bfloat16()
    nsteps = (B * H * W + 31) // 32
    if split is not None and split > nsteps:
        pytest.skip("more slices than K-steps")
    got = ops._wgrad_direct(xd, dyd, k, k, split_m=split)
    assert got is not None, "geometry must be handled by the kernel"
    # the bias gradient as a by-product (column sums of dy by the workgroups of input-channel block 0): same dW bits, db = sum dy
    got2, db = ops._wgrad_direct(xd, dyd, k, k, split_m=split, want_db=True)
    assert torch.equal(got2, got) and tuple(db.shape) == (N,)
    assert rel_l2(db.cpu(), dy.sum(dim=(0, 2, 3))) <= 1e-5
    torch.cuda.synchronize()
    assert tuple(got.shape) == (N, k * k, C)
    assert rel_l2(got.float().cpu(), ref) <= 2e-3                           # fp32 accumulation of bf16 products
    if split is None:
        ops.WGRAD_KERNEL = False
        try:
            old = ops.conv_wgrad(xd, dyd, k, k, 1, k // 2, 0)
        finally:
            ops.WGRAD_KERNEL = True
        assert rel_l2(got.float().cpu(), old.float().cpu()) <= 2e-3
        assert torch.equal(got, ops._wgrad_direct(xd, dyd, k, k))            # deterministic


@pytest.mark.parametrize("kind", ["stride2", "ups"])
@pytest.mark.parametrize("B,H,W,C,N", [(2, 16, 16, 72, 136), (1, 64, 64, 64, 64), (2, 32, 32, 128, 72), (3, 16, 16, 8, 8)])
def test_conv_wgrad_kernel_on_resampling_convs(ops, cuda, kind, B, H, W, C, N):
    """the six down / up-sampler convolutions through aptp_conv_wgrad itself (strided / up-sampled halo: output maps 32, 16 and 8
    pixels wide, i.e. one, two and four image rows per 32-pixel step): weight gradient vs PyTorch's, bias gradient as a
    by-product, in-place into a padded packed-layout buffer, deterministic; and equal to the copies + GEMM fallback"""
    g = torch.Generator().manual_seed(17 + H + C)
    x = torch.randn(B, C, H, W, generator=g).bfloat16().float()
    w = torch.zeros(N, C, 3, 3, requires_grad=True)
    if kind == "stride2":
        dy = torch.randn(B, N, H // 2, W // 2, generator=g).bfloat16().float()
        F.conv2d(x, w, None, stride=2, padding=1).backward(dy)
        args = dict(stride=2, pad=1, ups=0)
    else:
        dy = torch.randn(B, N, 2 * H, 2 * W, generator=g).bfloat16().float()
        F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, stride=1, padding=1).backward(dy)
        args = dict(stride=1, pad=1, ups=1)
    ref = w.grad.permute(0, 2, 3, 1).reshape(N, 9, C)
    xd, dyd = nhwc(x).to(cuda).bfloat16(), nhwc(dy).to(cuda).bfloat16()
    direct = ops._wgrad_direct(xd, dyd, 3, 3, stride=args["stride"], ups=args["ups"])
    assert direct is not None, "must take the kernel path"
    got, db = ops.conv_wgrad(xd, dyd, 3, 3, want_db=True, **args)
    assert rel_l2(got.cpu(), ref) <= 2e-3 and rel_l2(db.cpu(), dy.sum(dim=(0, 2, 3))) <= 1e-5
    assert torch.equal(got, direct)                                       # deterministic, same path
    one = ops._wgrad_direct(xd, dyd, 3, 3, split_m=1, stride=args["stride"], ups=args["ups"])
    assert rel_l2(one.cpu(), ref) <= 2e-3
    out = torch.full((N, 9, C + 8), -1.0, device=cuda)
    ops.conv_wgrad(xd, dyd, 3, 3, out=out, **args)
    assert torch.equal(out[:, :, :C], got) and bool((out[:, :, C:] == -1).all())
    ops.WGRAD_KERNEL = False
    try:
        old = ops.conv_wgrad(xd, dyd, 3, 3, **args)
    finally:
        ops.WGRAD_KERNEL = True
    assert rel_l2(got.cpu(), old.cpu()) <= 2e-3


@pytest.mark.parametrize("kind", ["stride2", "ups"])
def test_conv_wgrad_of_resampling_convs_by_parity_split(ops, cuda, kind):
    """the six down/up-sampler convolutions: weight gradient through four stride-1 correlations on the parity planes
    (ops._wgrad_parity) vs PyTorch's, and vs the GEMM-on-transposed-copies fallback; bias gradient as a by-product"""
    B, H, W, C, N = 2, 16, 16, 72, 136
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, C, H, W, generator=g).bfloat16().float()
    w = torch.zeros(N, C, 3, 3, requires_grad=True)
    if kind == "stride2":
        dy = torch.randn(B, N, H // 2, W // 2, generator=g).bfloat16().float()
        F.conv2d(x, w, None, stride=2, padding=1).backward(dy)
        args = dict(stride=2, pad=1, ups=0)
    else:
        dy = torch.randn(B, N, 2 * H, 2 * W, generator=g).bfloat16().float()
        F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, stride=1, padding=1).backward(dy)
        args = dict(stride=1, pad=1, ups=1)
    ref = w.grad.permute(0, 2, 3, 1).reshape(N, 9, C)
    xd, dyd = nhwc(x).to(cuda).bfloat16(), nhwc(dy).to(cuda).bfloat16()
    assert ops._wgrad_parity(xd, dyd, args["stride"], args["ups"], None, False) is not None, "must take the kernel path"
    got, db = ops._wgrad_parity(xd, dyd, args["stride"], args["ups"], None, True)
    assert rel_l2(got.cpu(), ref) <= 2e-3 and rel_l2(db.cpu(), dy.sum(dim=(0, 2, 3))) <= 1e-5
    out = torch.full((N, 9, C + 8), -1.0, device=cuda)
    ops._wgrad_parity(xd, dyd, args["stride"], args["ups"], out, False)
    assert torch.equal(out[:, :, :C], got) and bool((out[:, :, C:] == -1).all())
    ops.WGRAD_KERNEL = False
    try:
        old = ops.conv_wgrad(xd, dyd, 3, 3, **args)
    finally:
        ops.WGRAD_KERNEL = True
    assert rel_l2(got.cpu(), old.cpu()) <= 2e-3


def test_colsum_per_sample_and_strided(ops, cuda):
    """ops.colsum: whole-tensor and per-sample column sums (the gradient of conv1's per-sample time-embedding bias), also on a
    channel slice of a wider buffer"""
    g = torch.Generator().manual_seed(8)
    wide = torch.randn(3, 16, 16, 384, generator=g).bfloat16().to(cuda)
    for x in (wide, wide[..., 64:384]):
        ref = x.float().sum(dim=(1, 2))
        got = ops.colsum(x, per_sample=True)
        assert tuple(got.shape) == tuple(ref.shape) and rel_l2(got, ref) <= 1e-5
        assert rel_l2(ops.colsum(x), ref.sum(0)) <= 1e-5
        assert torch.equal(ops.colsum(x, per_sample=True), got)


def test_conv_wgrad_reads_channel_slices_in_place(ops, cuda):
    """x is a channel slice of a wider buffer (ld > C), as the skip-concat halves are"""
    g = torch.Generator().manual_seed(5)
    wide = torch.randn(2, 16, 16, 192, generator=g).bfloat16().to(cuda)
    x = wide[..., 64:128]
    dy = torch.randn(2, 16, 16, 64, generator=g).bfloat16().to(cuda)
    got = ops._wgrad_direct(x, dy, 3, 3)
    ref = ops._wgrad_direct(x.contiguous(), dy, 3, 3)
    assert torch.equal(got, ref)


def test_fold_rows_and_pack_dgrad_from_packed(ops, cuda):
    """the two data-movement kernels of the packed fine-tune step: a fixed-order slab fold into a padded layout, and the
    data-gradient operand rebuilt from the forward operand (== ops.pack_weight_dgrad of the same weights, bit for bit)"""
    g = torch.Generator().manual_seed(1)
    part = torch.randn(5, 7, 24, generator=g).to(cuda)
    out = torch.full((7, 40), -1.0, device=cuda)
    ops.fold_rows(part, 7, 24, out=out)
    assert torch.allclose(out[:, :24], part.sum(0), atol=1e-5) and bool((out[:, 24:] == -1).all())
    odd = torch.randn(3, 1, 10, generator=g).to(cuda)                     # C not a multiple of 4: scalar path
    assert torch.allclose(ops.fold_rows(odd, 1, 10), odd.sum(0), atol=1e-6)
    for (N, C, k) in [(72, 200, 3), (320, 136, 1), (8, 64, 3), (1280, 640, 1)]:
        w = torch.randn(N, C, k, k, generator=g)
        pw = ops.pack_weight(w, None, device=cuda)
        want = ops.pack_weight_dgrad(pw.w[:, :, :pw.Cin].float().reshape(pw.N, k, k, pw.Cin).permute(0, 3, 1, 2), device=cuda)
        got = ops.pack_weight_dgrad(torch.zeros(pw.N, pw.Cin, k, k), device=cuda)        # same shapes, zero content
        got.w.fill_(7.0)
        ops.pack_dgrad_from_packed(pw, got)
        torch.cuda.synchronize()
        assert got.w.shape == want.w.shape and torch.equal(got.w, want.w), (N, C, k)
    # all four weights in ONE launch (ops.PackDgradBatch: descriptor table in device memory, workgroups bisect for their item)
    pairs, wants = [], []
    for (N, C, k) in [(72, 200, 3), (320, 136, 1), (8, 64, 3), (1280, 640, 1)]:
        pw = ops.pack_weight(torch.randn(N, C, k, k, generator=g), None, device=cuda)
        wants.append(ops.pack_weight_dgrad(pw.w[:, :, :pw.Cin].float().reshape(pw.N, k, k, pw.Cin).permute(0, 3, 1, 2), device=cuda))
        dst = ops.pack_weight_dgrad(torch.zeros(pw.N, pw.Cin, k, k), device=cuda)
        dst.w.fill_(7.0)
        pairs.append((pw, dst))
    ops.PackDgradBatch(pairs).run()
    torch.cuda.synchronize()
    for (pw, dst), want in zip(pairs, wants):
        assert torch.equal(dst.w, want.w)


def test_deferred_folds_in_one_launch_equal_the_single_folds(ops, cuda):
    """ops.FOLD_DEFER / ops.FoldBatch: the slab folds of a backward pass recorded and run as ONE launch over a device table are
    bitwise the single folds (padded destination, bias tail rows, scalar path); pair_split writes the (dbeta, dgamma) pairs of
    the norm backward kernels as two contiguous, 16-byte aligned gradients (C = 170: the halves are padded apart)"""
    g = torch.Generator().manual_seed(2)
    specs = [(5, 7, 24, 40, 0), (3, 1, 10, 10, 0), (12, 136 * 9, 72, 128, 2), (64, 33, 8, 8, 0), (2, 640, 320, 320, 0)]
    parts = [torch.randn(R, rows + tail, C, generator=g).to(cuda) for (R, rows, C, ld, tail) in specs]
    outs_a = [torch.full((rows, ld), -1.0, device=cuda) for (R, rows, C, ld, tail) in specs]
    outs_b = [o.clone() for o in outs_a]
    tails_a = [torch.zeros(tail * C, device=cuda) if tail else None for (R, rows, C, ld, tail) in specs]
    tails_b = [None if t is None else t.clone() for t in tails_a]
    for pt, o, t, (R, rows, C, ld, tail) in zip(parts, outs_a, tails_a, specs):
        ops.fold_rows(pt, rows, C, out=o, tail_out=t)
    pair = torch.randn(9, 1, 340, generator=g).to(cuda)
    want_pair = ops.fold_rows(pair, 1, 340).view(170, 2)
    ops.FOLD_DEFER = []
    try:
        for pt, o, t, (R, rows, C, ld, tail) in zip(parts, outs_b, tails_b, specs):
            r = ops.fold_rows(pt, rows, C, out=o, tail_out=t, deferrable=True)
            assert r is o
        pg = ops.fold_rows(pair, 1, 340, deferrable=True, pair_split=True)
        not_deferred = ops.fold_rows(parts[0], 7, 24)                   # call sites that read their result at once stay immediate
        recs = ops.FOLD_DEFER
    finally:
        ops.FOLD_DEFER = None
    assert len(recs) == len(specs) + 1 and torch.allclose(not_deferred, parts[0].sum(0), atol=1e-5)
    assert bool((outs_b[0] == -1).all())                               # nothing ran yet
    ops.FoldBatch(recs).run()
    torch.cuda.synchronize()
    for a, b, ta, tb in zip(outs_a, outs_b, tails_a, tails_b):
        assert torch.equal(a, b) and (ta is None or torch.equal(ta, tb))
    assert pg.shape == (2, 172) and pg[1].data_ptr() % 16 == 0
    assert torch.equal(pg[0, :170], want_pair[:, 0]) and torch.equal(pg[1, :170], want_pair[:, 1])
    now = ops.fold_rows(pair, 1, 340, pair_split=True)                  # the immediate form of pair_split
    assert torch.equal(now[:, :170], pg[:, :170])


def test_batched_weight_gradients_equal_the_single_launches(ops, cuda):
    """ops.WGRAD_DEFER / ops.WgradBatch (aptp_conv_wgrad_many): the stride-1 weight gradients of a backward recorded and run as one
    launch per filter size are bitwise the per-layer launches with the same pixel split -- direct destinations with a padded
    leading dimension, slabs + bias-gradient tail folded afterwards, ragged N / C, strided dy (a slice of a fused buffer) -- and
    with the batch's own split rule they equal an fp32 reference to summation accuracy"""
    g = torch.Generator().manual_seed(11)
    # (B, H, W, C, N, K, split_m, want_db)
    specs = [(2, 32, 32, 176, 176, 1, 4, True), (2, 16, 16, 352, 704, 1, 1, True), (1, 8, 8, 72, 40, 1, 2, False),
             (2, 32, 32, 176, 96, 3, 3, True), (2, 16, 16, 136, 352, 3, 1, True), (4, 8, 8, 64, 64, 3, 1, False),
             (2, 64, 64, 24, 8, 1, 8, True)]

    def operands():
        out = []
        for (B, H, W, C, N, K, sm, wdb) in specs:
            x = torch.randn(B, H, W, C, generator=g).to(cuda, torch.bfloat16)
            buf = torch.randn(B, H, W, N + 16, generator=g).to(cuda, torch.bfloat16)
            out.append((x, buf[..., 8:8 + N] if N % 16 == 0 else buf[..., :N].contiguous()))
        return out
    opnds = operands()

    def run(split_rule):
        res = []
        for (x, dy), (B, H, W, C, N, K, sm, wdb) in zip(opnds, specs):
            ld = ops.round_up(C, 64)
            out = torch.full((N, K * K, ld), -3.0, device=cuda)
            db = torch.full((N,), -3.0, device=cuda) if wdb else None
            r = ops._wgrad_direct(x, dy, K, K, split_m=(sm if split_rule == "given" else None), out=out, want_db=wdb, db_out=db)
            assert r is not None
            res.append((out, db))
        return res
    want = run("given")
    torch.cuda.synchronize()
    for rule in ("given", "batch"):
        ops.WGRAD_DEFER, ops.FOLD_DEFER = [], []
        try:
            got = run(rule)
            wrec, frec = ops.WGRAD_DEFER, ops.FOLD_DEFER
        finally:
            ops.WGRAD_DEFER, ops.FOLD_DEFER = None, None
        assert len(wrec) == len(specs) and bool((got[1][0][:, :, :352] == -3).all())          # nothing ran yet
        batch = ops.WgradBatch(wrec)
        assert batch.launches() == 2
        batch.run()
        if frec:
            ops.FoldBatch(frec).run()
        torch.cuda.synchronize()
        for (a, da), (b, db_), (x, dy), (B, H, W, C, N, K, sm, wdb) in zip(want, got, opnds, specs):
            if rule == "given":
                assert torch.equal(a, b) and (da is None or torch.equal(da, db_))
            else:
                xf = x.float().permute(0, 3, 1, 2)
                cols = torch.nn.functional.unfold(xf, K, padding=K // 2).view(B, C, K * K, H * W)
                ref = torch.einsum("bctm,bmn->ntc", cols, dy.float().reshape(B, H * W, N))
                tol = 2e-5 * float(ref.abs().max()) + 1e-4
                assert float((b[:, :, :C] - ref).abs().max()) <= tol
                assert bool((b[:, :, C:] == -3).all())                                         # the padding columns are left alone
                if wdb:
                    assert torch.allclose(db_, dy.float().sum((0, 1, 2)), rtol=1e-5, atol=1e-3)


def test_batched_weight_gradient_of_an_upsampler_conv(ops, cuda):
    """the convolution behind a folded nearest-x2 up-sample rides in the same batch (stride 1 on the up-sampled grid; the halo copy
    fetches source pixel (iy >> 1, ix >> 1)): bitwise the per-layer launch with the same split"""
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 16, 16, 96, generator=g).to(cuda, torch.bfloat16)
    dy = torch.randn(2, 32, 32, 72, generator=g).to(cuda, torch.bfloat16)
    x1 = torch.randn(2, 32, 32, 64, generator=g).to(cuda, torch.bfloat16)          # an ordinary 3x3 layer in the same launch
    dy1 = torch.randn(2, 32, 32, 64, generator=g).to(cuda, torch.bfloat16)

    def run():
        a, b = torch.full((72, 9, 128), 7.0, device=cuda), torch.full((64, 9, 64), 7.0, device=cuda)
        da = torch.zeros(72, device=cuda)
        assert ops._wgrad_direct(x, dy, 3, 3, split_m=2, out=a, want_db=True, ups=1, db_out=da) is not None
        assert ops._wgrad_direct(x1, dy1, 3, 3, split_m=1, out=b) is not None
        return a, b, da
    want = run()
    ops.WGRAD_DEFER, ops.FOLD_DEFER = [], []
    try:
        got = run()
        wrec, frec = ops.WGRAD_DEFER, ops.FOLD_DEFER
    finally:
        ops.WGRAD_DEFER, ops.FOLD_DEFER = None, None
    assert len(wrec) == 2 and len(frec) == 1
    ops.WgradBatch(wrec).run()
    ops.FoldBatch(frec).run()
    torch.cuda.synchronize()
    for w, gt in zip(want, got):
        assert torch.equal(w, gt)
    xu = x.float().repeat_interleave(2, 1).repeat_interleave(2, 2).permute(0, 3, 1, 2)
    cols = torch.nn.functional.unfold(xu, 3, padding=1).view(2, 96, 9, 1024)
    ref = torch.einsum("bctm,bmn->ntc", cols, dy.float().reshape(2, 1024, 72))
    assert float((got[0][:, :, :96] - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-4


@pytest.mark.parametrize("case", ["bf16_nhwc", "bf16_slice", "f32_nchw", "bf16_big", "odd_rows", "nchw_face"])
def test_mse_matches_torch_fp64(cuda, case):
    """ops.mse / autograd.MseFn (csrc/loss_ops.hip) vs F.mse_loss in fp64: value to fp32 summation accuracy, gradient to the
    rounding of its own dtype; row-strided operands (a channel slice of a skip-concat buffer) are read in place."""
    from diffusion_pruning_amd import autograd as AG
    g = torch.Generator().manual_seed(3)
    if case == "bf16_nhwc":
        a = torch.randn(2, 16, 16, 320, generator=g).to(cuda, torch.bfloat16)
        b = torch.randn(2, 16, 16, 320, generator=g).to(cuda, torch.bfloat16)
    elif case == "bf16_slice":
        buf = torch.randn(2, 8, 8, 640 + 320, generator=g).to(cuda, torch.bfloat16)
        a = torch.randn(2, 8, 8, 640, generator=g).to(cuda, torch.bfloat16)
        b = buf[..., :640]
        assert not b.is_contiguous()
    elif case == "f32_nchw":
        a = torch.randn(4, 4, 64, 64, generator=g).to(cuda)
        b = torch.randn(4, 4, 64, 64, generator=g).to(cuda)
    elif case == "bf16_big":
        a = (3 * torch.randn(4, 64, 64, 320, generator=g)).to(cuda, torch.bfloat16)
        b = (3 * torch.randn(4, 64, 64, 320, generator=g)).to(cuda, torch.bfloat16)
    elif case == "nchw_face":
        # what the block hooks hand over: the NCHW face of a channels-last activation; the teacher's is a channel slice of a
        # skip-concat buffer.  Read in place, and the gradient comes back channels-last
        a = torch.randn(2, 8, 8, 640, generator=g).to(cuda, torch.bfloat16).permute(0, 3, 1, 2)
        b = torch.randn(2, 8, 8, 960, generator=g).to(cuda, torch.bfloat16)[..., :640].permute(0, 3, 1, 2)
        assert not a.is_contiguous() and not b.is_contiguous()
    else:
        a = torch.randn(3, 5, 7, 24, generator=g).to(cuda, torch.bfloat16)
        b = torch.randn(3, 5, 7, 24, generator=g).to(cuda, torch.bfloat16)
    a1 = a.detach().requires_grad_()
    a2 = a.clone().double().requires_grad_()
    got = AG.mse(a1, b)
    ref = F.mse_loss(a2, b.double())
    assert got.dtype == torch.float32 and got.dim() == 0
    assert abs(float(got) - float(ref)) <= 2e-6 * float(ref), (float(got), float(ref))
    (got * 3.0).backward()
    (ref * 3.0).backward()
    assert a1.grad.dtype == a.dtype and a1.grad.shape == a.shape
    if case == "nchw_face":
        assert a1.grad.permute(0, 2, 3, 1).is_contiguous()
    assert rel_l2(a1.grad, a2.grad) <= (1e-6 if a.dtype == torch.float32 else 4e-3)
    assert torch.equal(AG.mse(a1.detach(), b), got.detach())          # fixed summation order: same bits every call


def test_mse_inside_replayed_graphs_matches_eager(cuda):
    """the loss terms are captured into the train-step graphs: replaying must reproduce the eager value bit for bit, also
    with another graph running on a second stream (torch's multi-block mse_loss reduction did not: tools/diag_overlap.py)"""
    from diffusion_pruning_amd import autograd as AG
    g = torch.Generator().manual_seed(5)
    a = torch.randn(4, 64, 64, 320, generator=g).to(cuda, torch.bfloat16)
    b = torch.randn(4, 64, 64, 320, generator=g).to(cuda, torch.bfloat16)
    noise = torch.randn(1 << 22, device=cuda)

    def busy():
        y = noise
        for _ in range(50):
            y = y * 1.0001 + 1.0
        return y
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        AG.mse(a, b); busy()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager = AG.mse(a, b).clone()
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        out = AG.mse(a, b)
    with torch.cuda.graph(g2):
        keep = busy()
    vals = []
    for _ in range(16):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g2.replay()
        g1.replay()
        vals.append(out.clone())
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert all(torch.equal(v, eager) for v in vals)


def test_standalone_gate_forward_is_the_reference_forward(ops, cuda):
    """WidthGate / LinearWidthGate / DepthGate as modules of their own (pdm/models/unet/gates.py:15-21,36-42,49-55), incl. the
    classifier-free-guidance batch doubling (gate batch 2, activation batch 4), through the HIP gate kernels, differentiably"""
    from diffusion_pruning_amd.gates import DepthGate, LinearWidthGate, WidthGate
    g = torch.Generator().manual_seed(3)
    B, C, H, W, G = 4, 64, 8, 8, 32
    x = torch.randn(B, C, H, W, generator=g).bfloat16().to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    gate = WidthGate(G)
    gf = torch.rand(2, G, generator=g).to(cuda).requires_grad_()
    gate.set_structure_value(gf)
    y = gate(x)
    mask = gf.detach().float().repeat_interleave(C // G, dim=1)[:, :, None, None].repeat(2, 1, 1, 1)
    ref = mask * x.detach().float()
    assert y.shape == x.shape and float((y.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    R = torch.randn(B, C, H, W, generator=g).to(cuda)
    (y.float() * R).sum().backward()
    dgate_ref = (R * x.detach().float()).view(2, 2, G, C // G, H, W).sum(dim=(0, 3, 4, 5))
    assert float((gf.grad - dgate_ref).abs().max()) <= 2e-2 * float(dgate_ref.abs().max())
    assert float((x.grad.float() - mask * R).abs().max()) <= 2e-2 * float(R.abs().max())
    # tokens
    t = torch.randn(B, 40, C, generator=g).bfloat16().to(cuda)
    lg = LinearWidthGate(G)
    lg.set_structure_value(gf.detach())
    yt = lg(t)
    reft = gf.detach().float().repeat_interleave(C // G, dim=1)[:, None, :].repeat(2, 1, 1) * t.float()
    assert float((yt.float() - reft).abs().max()) <= 2e-2 * float(reft.abs().max())
    # depth gate
    a = torch.randn(B, C, H, W, generator=g).bfloat16().to(cuda)
    b = torch.randn(B, C, H, W, generator=g).bfloat16().to(cuda)
    dg = DepthGate(1)
    d = torch.tensor([0.25, 0.9], device=cuda, requires_grad=True)
    dg.set_structure_value(d)
    yd = dg((a, b))
    dm = d.detach().repeat(2)[:, None, None, None]
    refd = (1 - dm) * a.float() + dm * b.float()
    assert float((yd.float() - refd).abs().max()) <= 2e-2 * float(refd.abs().max())
    yd.float().sum().backward()
    dd_ref = (b.float() - a.float()).view(2, 2, -1).sum(dim=(0, 2))
    assert float((d.grad - dd_ref).abs().max()) <= 2e-2 * float(dd_ref.abs().max()) + 1e-2
    # the default state (all ones) is the identity; tensors the kernels do not take are refused, not silently computed elsewhere
    assert torch.equal(WidthGate(G).forward(x.detach()), x.detach())
    with pytest.raises(TypeError):
        gate(x.detach().float())
    with pytest.raises(TypeError):
        gate(x.detach().cpu())
