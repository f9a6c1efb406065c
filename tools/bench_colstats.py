import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)
for (H, Cin, N, k, res) in [(64, 160, 320, 3, 1), (64, 320, 160, 3, 0), (64, 640, 160, 3, 0), (32, 320, 640, 3, 1), (32, 640, 320, 3, 0), (64, 640, 320, 1, 1), (64, 320, 320, 1, 1), (32, 1280, 640, 1, 1)]:
    B = 4
    x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    pw = ops.pack_weight(torch.randn(N, Cin, k, k) * 0.02, torch.zeros(N), device=dev)
    r = torch.randn(B, H, H, N, device=dev).bfloat16() if res else None
    y = ops.conv_gemm(x, pw, residual=r)
    t0 = timeit(lambda: ops.conv_gemm(x, pw, residual=r, out=y))
    t1 = timeit(lambda: ops.conv_gemm(x, pw, residual=r, out=y, colstats=True))
    key = ops.tuning_key(B*H*H, N, Cin, k*k, 1, 0, False)
    print(key, ops.TUNING.get(key), f"plain {t0:6.1f} us   colstats {t1:6.1f} us", flush=True)
