"""Attention micro-benchmark on the headline config's shapes, both two-group forms.  Usage: python tools/bench_attn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


for (B, h, Lq, Lk) in [(4, 2, 4096, 4096), (4, 5, 1024, 1024), (4, 10, 256, 256), (4, 2, 4096, 77), (4, 5, 1024, 77), (4, 5, 4096, 4096), (4, 10, 1024, 1024), (4, 20, 256, 256), (4, 5, 4096, 77)]:
    q = torch.randn(B, Lq, h * 64, device=dev).bfloat16()
    k = torch.randn(B, Lk, h * 64, device=dev).bfloat16()
    v = torch.randn(B, Lk, h * 64, device=dev).bfloat16()
    o = ops.attention(q, k, v, h)
    fl = 4.0 * B * h * Lq * Lk * 64
    row = []
    for var, name in ((3, "2 groups"), (2, "4 groups"), (1, "staggered"), (4, "dbuf"), (5, "1 group"), (6, "sw-pipelined"), (0, "auto")):
        ops.ATTN_VARIANT = var
        t = timeit(lambda: ops.attention(q, k, v, h, out=o))
        row.append(f"{name} {t:6.1f} us {fl / t / 1e6:6.1f} TF")
    ops.ATTN_VARIANT = 6
    o6 = ops.attention(q, k, v, h).float()
    ops.ATTN_VARIANT = 4
    o4 = ops.attention(q, k, v, h).float()
    ops.ATTN_VARIANT = 0
    row.append(f"| v6 vs v4 rel-L2 {float((o6 - o4).norm() / o4.norm()):.2e}")
    print(f"B{B} h{h} Lq{Lq} Lk{Lk}: " + "   ".join(row), flush=True)
