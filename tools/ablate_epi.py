"""Time the headline config's GEMM shapes with an (ablated) library build: where does a small-K launch spend its time.
Usage: python3 tools/ablate_epi.py <lib.so>; prints us per launch (tuned tile / split-K from the table) per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from diffusion_pruning_amd import ops

dev = torch.device("cuda:0")
# M (= 4 * H * H), N, Cin, k, residual
cases = [(64, 320, 128, 1, 1), (64, 320, 320, 1, 1), (64, 320, 640, 1, 1), (64, 128, 320, 1, 0), (64, 384, 320, 1, 0),
         (32, 640, 320, 1, 1), (32, 640, 640, 1, 1), (32, 640, 1280, 1, 1), (16, 1280, 640, 1, 1), (16, 1280, 1280, 1, 1),
         (16, 1280, 2560, 1, 1), (64, 320, 160, 3, 1), (32, 640, 320, 3, 1), (16, 1280, 640, 3, 1), (8, 1280, 640, 3, 1)]
out = []
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for (H, N, Cin, k, res) in cases:
    B = 4
    x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    pw = ops.pack_weight(torch.randn(N, Cin, k, k) * 0.02, torch.zeros(N), device=dev)
    r = torch.randn(B, H, H, N, device=dev).bfloat16() if res else None
    y = ops.conv_gemm(x, pw, residual=r)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                ops.conv_gemm(x, pw, residual=r, out=y)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) * 1e3 / 100)
print(f"{os.path.basename(sys.argv[1]):22s}", " ".join(f"{u:6.1f}" for u in out))
