"""Attention backward micro-benchmark (delta + dq + dkdv) on SD-2.1's self-attention shapes.  usage: python tools/bench_attn_bwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


for (B, h, L, Lk) in [(4, 5, 4096, 4096), (4, 10, 1024, 1024), (4, 20, 256, 256), (4, 5, 4096, 77), (4, 10, 1024, 77)]:
    q = torch.randn(B, L, h * 64, device=dev).bfloat16()
    k = torch.randn(B, Lk, h * 64, device=dev).bfloat16()
    v = torch.randn(B, Lk, h * 64, device=dev).bfloat16()
    lse = torch.empty(B, h, L, device=dev)
    o = ops.attention(q, k, v, h, lse=lse)
    do = torch.randn_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    t_f = timeit(lambda: ops.attention(q, k, v, h, out=o, lse=lse))
    t_b = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, h, dq, dk, dv))
    fl = 4.0 * B * h * L * Lk * 64
    print(f"B{B} h{h} L{L} Lk{Lk}: fwd {t_f:7.1f} us {fl / t_f / 1e6:6.1f} TF   bwd {t_b:7.1f} us {2.5 * fl / t_b / 1e6:6.1f} TF (5 products) / {3.5 * fl / t_b / 1e6:6.1f} (7 executed)", flush=True)

print("dK/dV query-range split, cross-attention shapes:")
for (B, h, L, Lk) in [(4, 5, 4096, 77), (4, 10, 1024, 77), (4, 20, 256, 77), (4, 20, 64, 77)]:
    q = torch.randn(B, L, h * 64, device=dev).bfloat16()
    k = torch.randn(B, Lk, h * 64, device=dev).bfloat16()
    v = torch.randn(B, Lk, h * 64, device=dev).bfloat16()
    lse = torch.empty(B, h, L, device=dev)
    o = ops.attention(q, k, v, h, lse=lse)
    do = torch.randn_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    row = []
    for sp in (1, 2, 4, 8, 16, 32):
        if sp > max(1, (L + 63) // 64):
            continue
        t = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, h, dq, dk, dv, q_split=sp))
        row.append(f"split {sp:2d}: {t:6.1f} us")
    print(f"B{B} h{h} L{L} Lk{Lk}: " + "   ".join(row), flush=True)
