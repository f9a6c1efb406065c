"""Time one conv shape with an ablated library build (tools: where do the K-loop cycles go).
Usage: python3 tools/ablate_conv.py <lib.so> ; prints us per launch for a few shapes/tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from diffusion_pruning_amd import ops

dev = torch.device("cuda:0")
# B, H, Cin, Cout, k, tile, split_k
cases = [(4, 64, 320, 320, 3, 10, 1), (4, 64, 320, 320, 3, 8, 1), (4, 64, 320, 160, 3, 10, 1), (4, 32, 640, 640, 3, 10, 1),
         (4, 32, 640, 640, 3, 12, 1), (4, 16, 1280, 1280, 3, 12, 2)]
out = []
for (B, H, Cin, Cout, k, tile, sk) in cases:
    x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    pw = ops.pack_weight(torch.randn(Cout, Cin, k, k) * 0.02, torch.zeros(Cout), device=dev)
    y = ops.conv_gemm(x, pw, tile=tile, split_k=sk)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                ops.conv_gemm(x, pw, tile=tile, split_k=sk, out=y)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) * 1e3 / 100)
print(os.path.basename(sys.argv[1]), " ".join(f"{u:7.1f}" for u in out))
