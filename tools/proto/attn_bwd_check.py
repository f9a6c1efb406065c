"""attention backward against fp64 autograd on peaked inputs (debugging aid); APTP_LIB selects the library"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, h, L, Lk, amp) in [(2, 2, 512, 512, 1.0), (2, 2, 512, 512, 3.0), (1, 2, 2048, 2048, 3.0), (2, 3, 200, 77, 2.0)]:
    q = (torch.randn(B, L, h * 64, device=dev) * amp).bfloat16()
    k = (torch.randn(B, Lk, h * 64, device=dev) * amp).bfloat16()
    v = torch.randn(B, Lk, h * 64, device=dev).bfloat16()
    do = torch.randn(B, L, h * 64, device=dev).bfloat16()
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    def heads(t, n): return t.view(B, n, h, 64).transpose(1, 2)
    s = heads(qd, L) @ heads(kd, Lk).transpose(-1, -2) / 8.0
    ref = (torch.softmax(s, -1) @ heads(vd, Lk)).transpose(1, 2).reshape(B, L, h * 64)
    ref.backward(do.double())
    lse = torch.zeros(B, h, L, device=dev)
    o = ops.attention(q, k, v, h, lse=lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attention_bwd(q, k, v, o, do, lse, h, dq, dk, dv)
    def rel(a, b): return float((a.double() - b).norm() / b.norm())
    print(f"B{B} h{h} L{L} Lk{Lk} amp{amp}: dq {rel(dq, qd.grad):.3e}  dk {rel(dk, kd.grad):.3e}  dv {rel(dv, vd.grad):.3e}", flush=True)
