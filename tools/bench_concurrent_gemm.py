"""How much of a short-K GEMM launch is fill/drain?  The same GEMM twice: back-to-back on one stream vs side by side on two
streams (both captured in one HIP graph).  If the pair runs in about the time of one, a launch leaves the chip half idle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import ops

dev = torch.device("cuda:0")


def graph_time(build, reps=20):
    build(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            build()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


for (M, N, K) in [(4096, 2560, 640), (16384, 320, 640), (1024, 1280, 1280), (1024, 1280, 2560), (4096, 640, 640), (16384, 320, 320), (256, 1280, 1280)]:
    w = [torch.randn(N, K) * 0.02 for _ in range(2)]
    pw = [ops.pack_weight(x, torch.zeros(N), device=dev) for x in w]
    xs = [torch.randn(1, M, K, device=dev).bfloat16() for _ in range(2)]
    outs = [torch.empty(1, M, N, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    side = torch.cuda.Stream()

    def serial():
        ops.linear(xs[0], pw[0], out=outs[0]); ops.linear(xs[1], pw[1], out=outs[1])

    def parallel():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), ops.scratch_domain("side"):
            ops.linear(xs[1], pw[1], out=outs[1])
        ops.linear(xs[0], pw[0], out=outs[0])
        cur.wait_stream(side)

    def single():
        ops.linear(xs[0], pw[0], out=outs[0])
    t1, t2, tp = graph_time(single), graph_time(serial), graph_time(parallel)
    print(f"M{M} N{N} K{K}: one {t1:6.1f} us   two serial {t2:6.1f} us   two side by side {tp:6.1f} us", flush=True)
