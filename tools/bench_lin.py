"""Lean linear kernel (csrc/lin_gemm.hip) against the general implicit-GEMM kernel on the 1x1 shapes of the headline step
(profiles/r3_launch_table.txt), launch by launch from replayed HIP graphs, with the table's tile and with every lean tile.
Usage: python3 tools/bench_lin.py [out.txt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
# (M, N, K, tile, ln, rowstat, colstat, residual, launches per step)
SHAPES = [(4096, 640, 320, 18, 0, 1, 0, 1, 10), (1024, 1280, 640, 49, 0, 1, 0, 1, 10), (1024, 1280, 2560, 49, 0, 0, 0, 1, 5),
          (4096, 640, 1280, 18, 0, 0, 0, 1, 5), (1024, 1920, 1280, 49, 1, 0, 0, 0, 5), (16384, 384, 320, 9, 1, 0, 0, 0, 5),
          (4096, 960, 640, 12, 1, 0, 0, 0, 5), (16384, 320, 320, 56, 0, 1, 0, 0, 5), (1024, 1280, 1280, 49, 0, 1, 0, 0, 5),
          (1024, 1280, 1280, 49, 0, 0, 0, 1, 5), (4096, 640, 640, 18, 0, 0, 1, 1, 5), (16384, 320, 128, 9, 0, 1, 0, 1, 5),
          (4096, 640, 640, 18, 0, 1, 0, 0, 5), (1024, 640, 1280, 59, 1, 0, 0, 0, 5), (16384, 320, 128, 9, 0, 0, 0, 1, 5),
          (4096, 320, 640, 49, 1, 0, 0, 0, 5), (16384, 128, 320, 49, 1, 0, 0, 0, 5), (256, 1280, 640, 59, 0, 1, 0, 1, 2)]
LEAN = [12, 18, 25, 49, 9, 15, 24, 51, 11, 17, 26, 53]
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None


def emit(s):
    print(s, flush=True)
    if out:
        out.write(s + "\n")


def timed(fn, reps=20):
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    fn()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


tot_old = tot_new = tot_best = 0.0
for (M, N, K, tile, ln, rs, cs, res, cnt) in SHAPES:
    gen = torch.Generator().manual_seed(M + N + K)
    hw = M // 4
    x = torch.randn(4, hw, 1, K, generator=gen).bfloat16().to(dev)
    w = torch.randn(N, K, 1, 1, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen) * 0.1
    kw = dict(pad=0)
    if ln:
        pw = ops.pack_weight(w, b, device=dev, ln_gamma=torch.ones(K), ln_beta=torch.zeros(K))
        st = torch.zeros(2, M, 4, device=dev)
        xs = x.float().reshape(M, K)
        st[0, :, 0], st[0, :, 1] = xs.sum(1), (xs * xs).sum(1)
        kw["ln"] = (st, 1e-5)
    else:
        pw = ops.pack_weight(w, b, device=dev)
    if res:
        kw["residual"] = torch.randn(4, hw, 1, N, generator=gen).bfloat16().to(dev)
    y = torch.empty(4, hw, 1, N, dtype=torch.bfloat16, device=dev)

    def run(t, epi):
        def fn():
            ops.EPILOGUE = epi
            try:
                ops.conv_gemm(x, pw, tile=t, out=y, rowstats=bool(rs), colstats=bool(cs), **kw)
            finally:
                ops.EPILOGUE = 0
        return fn
    t_old = timed(run(tile, 2))
    lean_tile = tile if tile in LEAN else {56: 18, 59: 49}.get(tile, 18)
    t_new = timed(run(lean_tile, 0))
    alls = {}
    for t in LEAN:
        try:
            alls[t] = timed(run(t, 0))
        except Exception as e:          # a tile the shape cannot use
            alls[t] = float("nan")
    bt = min((v, t) for t, v in alls.items() if v == v)
    gf = 2.0 * M * N * K / 1e9
    emit(f"M{M:6d} N{N:5d} K{K:5d} ln{ln} rs{rs} cs{cs} res{res} x{cnt}: general tile {tile:2d} {t_old:6.2f} us | lean tile {lean_tile:2d} {t_new:6.2f} us "
         f"({t_old / t_new:4.2f}x) | best lean tile {bt[1]:2d} {bt[0]:6.2f} us ({gf / bt[0] / 1e3:6.1f} TF/s) | all: "
         + " ".join(f"{t}:{v:.1f}" for t, v in alls.items()))
    tot_old += cnt * t_old; tot_new += cnt * t_new; tot_best += cnt * bt[0]
emit(f"sum over the step's launches of these shapes: general {tot_old:.0f} us, lean (same tile) {tot_new:.0f} us, lean (best tile) {tot_best:.0f} us")
