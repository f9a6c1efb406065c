"""Launch one aptp_conv_gemm shape a few times (for rocprofv3 --pmc passes).
Usage: python3 tools/run_one_conv.py B H Cin Cout k tile split_k [reps] [flags]   flags: r = residual, s = row statistics out, a = SiLU"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd import ops  # noqa: E402

B, H, Cin, Cout, k, tile, sk = (int(a) for a in sys.argv[1:8])
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
flags = sys.argv[9] if len(sys.argv) > 9 else ""
dev = torch.device("cuda:0")
x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
pw = ops.pack_weight(torch.randn(Cout, Cin, k, k) * 0.02, torch.zeros(Cout), device=dev)
kw = {}
if "r" in flags:
    kw["residual"] = torch.randn(B, H, H, Cout, device=dev).bfloat16()
if "s" in flags:
    kw["rowstats"] = True
if "a" in flags:
    kw["act"] = ops.ACT_SILU
for _ in range(reps):
    y = ops.conv_gemm(x, pw, tile=tile, split_k=sk, **kw)
y = y[0] if isinstance(y, tuple) else y
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))
