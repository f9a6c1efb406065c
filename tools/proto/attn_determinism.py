"""run the attention forward / backward repeatedly on the same operands: any bit that changes is a race (debugging aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, h, L) in [(1, 5, 4096), (1, 10, 1024), (1, 20, 256), (4, 2, 4096), (2, 5, 320), (2, 20, 64)]:
    q = (torch.randn(B, L, h * 64, device=dev) * 2).bfloat16()
    k = (torch.randn(B, L, h * 64, device=dev) * 2).bfloat16()
    v = torch.randn(B, L, h * 64, device=dev).bfloat16()
    do = torch.randn(B, L, h * 64, device=dev).bfloat16()
    lse0 = torch.zeros(B, h, L, device=dev)
    o0 = ops.attention(q, k, v, h, lse=lse0).clone()
    dq0, dk0, dv0 = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attention_bwd(q, k, v, o0, do, lse0, h, dq0, dk0, dv0)
    bad = [0, 0, 0, 0, 0]
    for it in range(30):
        lse = torch.zeros(B, h, L, device=dev)
        o = ops.attention(q, k, v, h, lse=lse)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops.attention_bwd(q, k, v, o0, do, lse0, h, dq, dk, dv)
        for i, (a, b) in enumerate(((o, o0), (lse, lse0), (dq, dq0), (dk, dk0), (dv, dv0))):
            bad[i] += int((a != b).sum())
    print(f"B{B} h{h} L{L}: differing elements over 30 repeats  o {bad[0]}  lse {bad[1]}  dq {bad[2]}  dk {bad[3]}  dv {bad[4]}", flush=True)
