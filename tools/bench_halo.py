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
for (H, Cin, N) in [(64, 160, 320), (64, 320, 160), (64, 640, 160), (64, 960, 160), (32, 320, 640), (32, 640, 320), (32, 1280, 320), (32, 1920, 320), (64, 320, 320), (32, 640, 640)]:
    B = 4
    x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    pw = ops.pack_weight(torch.randn(N, Cin, 3, 3) * 0.02, torch.zeros(N), device=dev)
    y = ops.conv_gemm(x, pw)
    t0 = timeit(lambda: ops.conv_gemm(x, pw, out=y))
    row = []
    for tile in (43, 44):
        for sk in (1, 2, 3, 4):
            for order in (2, 3):
                try:
                    t = timeit(lambda: ops.conv_gemm(x, pw, out=y, tile=tile, split_k=sk, order=order))
                    row.append((t, tile, sk, order))
                except Exception as e:
                    pass
    row.sort()
    fl = 2.0 * B * H * H * N * 9 * Cin
    print(f"M{B*H*H}_N{N}_C{Cin}: tuned {t0:6.1f} us ({fl/t0/1e6:5.0f} TF)   halo best {row[0][0]:6.1f} us (t{row[0][1]} s{row[0][2]} o{row[0][3]}, {fl/row[0][0]/1e6:5.0f} TF)   next {row[1][0]:6.1f} (t{row[1][1]} s{row[1][2]} o{row[1][3]})", flush=True)
