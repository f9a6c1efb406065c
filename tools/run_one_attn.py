"""A few launches of ONE attention forward form on one shape (the profiled program of tools/pmc_attn.py).
usage: python3 tools/run_one_attn.py <variant> <B> <heads> <L> [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import ops

variant, B, h, L = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(B, L, h * 64, generator=g).bfloat16().to(dev) for _ in range(3))
ops.ATTN_VARIANT = variant
o = torch.empty_like(q)
for _ in range(reps):
    ops.attention(q, k, v, h, out=o)
torch.cuda.synchronize()
