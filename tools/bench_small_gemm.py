"""Yardstick for the short-K GEMMs of the transformer path: the forward's 1x1 / linear shapes through this repo's kernel
(tuned tile) and through hipBLASLt (torch.matmul, and torch.addmm where the layer has a bias), each replayed from a HIP graph.
Same-box A/B; prints us and TFLOP/s per shape."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd import ops

SHAPES = [  # (M, N, K, count per step) -- headline config (bs=4, 50 % mask)
    (16384, 320, 320, 10), (16384, 384, 320, 5), (16384, 320, 128, 10), (16384, 128, 320, 5), (16384, 1280, 320, 5), (16384, 320, 640, 5),
    (4096, 640, 640, 10), (4096, 960, 640, 5), (4096, 640, 320, 10), (4096, 320, 640, 5), (4096, 2560, 640, 5), (4096, 640, 1280, 5),
    (1024, 1280, 1280, 10), (1024, 1920, 1280, 5), (1024, 1280, 640, 10), (1024, 640, 1280, 5), (1024, 5120, 1280, 5), (1024, 1280, 2560, 5),
    (256, 1280, 1280, 2), (256, 1920, 1280, 1), (256, 1280, 640, 2), (256, 640, 1280, 1), (256, 5120, 1280, 1), (256, 1280, 2560, 1),
]


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def main():
    dev = torch.device("cuda:0")
    tot_own = tot_lib = 0.0
    for M, N, K, cnt in SHAPES:
        x = torch.randn(1, M, 1, K, device=dev).to(torch.bfloat16)
        w = torch.randn(N, K, device=dev) * 0.02
        geglu = N in (1280, 2560, 5120) and K in (320, 640, 1280) and N == 4 * K
        pw = ops.pack_weight(w, torch.zeros(N, device=dev), geglu=geglu, device=dev)
        own = timed(lambda: ops.conv_gemm(x, pw, pad=0))
        x2 = x.view(M, K); wt = w.to(torch.bfloat16).t().contiguous(); b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        lib = timed(lambda: torch.addmm(b, x2, wt))
        wn = w.to(torch.bfloat16).contiguous()
        lib2 = timed(lambda: torch.nn.functional.linear(x2, wn, b))
        fl = 2.0 * M * N * K
        print(f"M{M:6d} N{N:5d} K{K:5d} x{cnt:2d}  own {own:6.1f} us {fl/own/1e6:7.1f} TF | hipBLASLt addmm {lib:6.1f} us {fl/lib/1e6:7.1f} TF | F.linear {lib2:6.1f} us"
              + ("  (own includes GEGLU epilogue)" if geglu else ""), flush=True)
        tot_own += own * cnt; tot_lib += min(lib, lib2) * cnt
    print(f"sum over one step: own {tot_own/1e3:.3f} ms, hipBLASLt {tot_lib/1e3:.3f} ms")


if __name__ == "__main__":
    main()
