"""Persistent stream-K macro-tiles (csrc/conv_gemm_sk.hip) against the committed tuning table, shape by shape: the large
contractions of the headline forward (profiles/r2_launch_table.txt) plus a plain 8192^3 GEMM, timed with HIP-graph replays
of 10 launches (warm operands).  Prints us / TFLOP/s per (tile, split_k, order); writes gpurun_out/bench_sk.txt.
Usage: python tools/bench_sk.py [--quick]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusion_pruning_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
# (tile, split_k, order): the macro-tiles with whole tiles / stream-K in both tile orders; the read-in-load-slot forms once
CONFIGS = [(t, s, o) for t in (64, 65, 66) for (s, o) in ((1, 3), (2, 2), (2, 3))] + [(t, 2, 3) for t in (67, 68, 69)]

# (B, H, W, Cin, N, k, ups, Cin2, geglu) -- the headline step's launches of >= 25 us and its widest 1x1 projections
SHAPES = [
    (4, 32, 32, 1280, 1280, 3, 1, 0, 0),     # up-sampler conv to 64x64 (M 16384 ... here M = 4*64*64 after ups)
    (4, 16, 16, 1280, 1280, 3, 1, 0, 0),     # up-sampler conv to 32x32
    (4, 64, 64, 640, 160, 3, 0, 0, 0),
    (4, 64, 64, 160, 320, 3, 0, 640, 0),
    (4, 64, 64, 320, 160, 3, 0, 0, 0),
    (4, 64, 64, 160, 320, 3, 0, 0, 0),
    (4, 64, 64, 960, 160, 3, 0, 0, 0),
    (4, 64, 64, 160, 320, 3, 0, 960, 0),
    (4, 32, 32, 1920, 320, 3, 0, 0, 0),
    (4, 32, 32, 1280, 320, 3, 0, 0, 0),
    (4, 32, 32, 320, 640, 3, 0, 1920, 0),
    (4, 32, 32, 320, 640, 3, 0, 1280, 0),
    (4, 32, 32, 640, 320, 3, 0, 0, 0),
    (4, 16, 16, 2560, 640, 3, 0, 0, 0),
    (4, 16, 16, 640, 1280, 3, 0, 2560, 0),
    (4, 16, 16, 1280, 640, 3, 0, 0, 0),
    (4, 8, 8, 2560, 640, 3, 0, 0, 0),
    (4, 32, 32, 640, 2560, 1, 0, 0, 1),      # GEGLU projections
    (4, 16, 16, 1280, 5120, 1, 0, 0, 1),
    (4, 16, 16, 2560, 1280, 1, 0, 0, 0),     # ff.net[2]
    (4, 32, 32, 1280, 640, 1, 0, 0, 0),
    (4, 16, 16, 1280, 1920, 1, 0, 0, 0),     # fused QKV
    (4, 64, 64, 320, 384, 1, 0, 0, 0),
    (4, 32, 32, 640, 960, 1, 0, 0, 0),
]


_stream = None


def time_graph(fn, reps=10):
    # ONE stream for every measurement: split-K counters are handed out per launching stream (ops._tile_counters: 16 slabs)
    global _stream
    if _stream is None:
        _stream = torch.cuda.Stream()
    stream = _stream
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(reps):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        g.replay(); g.replay(); g.replay()
        e1.record(stream)
        stream.synchronize()
        return e0.elapsed_time(e1) / (3 * reps) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--tiles128", action="store_true", help="round 4: the 128-row stream-K tiles (70-73) on the convolution shapes only")
    args = ap.parse_args()
    global CONFIGS
    if args.tiles128:
        CONFIGS = [(t, s_, o) for t in (70, 71, 72, 73) for (s_, o) in ((1, 3), (2, 3), (2, 2))]
    ops._lib.load()
    ops._tile_counters(dev)
    lines = []

    def out(s):
        print(s, flush=True)
        lines.append(s)

    gen = torch.Generator().manual_seed(0)
    shapes = SHAPES[:6] if args.quick else (SHAPES[:17] if args.tiles128 else SHAPES)
    for (B, H, W, Cin, N, k, ups, Cin2, geglu) in shapes:
        x = torch.randn(B, H, W, Cin, generator=gen).bfloat16().to(dev)
        w = torch.randn(N, Cin, k, k, generator=gen) / (Cin * k * k) ** 0.5
        pw = ops.pack_weight(w, torch.randn(N, generator=gen), geglu=bool(geglu), device=dev)
        Ho, Wo = (H * 2, W * 2) if ups else (H, W)
        x2 = None
        if Cin2:
            pw = ops.pack_weight_cat(pw, torch.randn(N, Cin2, 1, 1, generator=gen) / Cin2 ** 0.5, None)
            x2 = torch.randn(B, Ho, Wo, Cin2, generator=gen).bfloat16().to(dev)
        M = B * Ho * Wo
        fl = 2.0 * M * N * (k * k * Cin + Cin2)
        y = torch.empty(B, Ho, Wo, N // 2 if geglu else N, dtype=torch.bfloat16, device=dev)
        base = time_graph(lambda: ops.conv_gemm(x, pw, ups=ups, x2=x2, out=y))
        ref = y.clone()
        tuned = ops.tuning_lookup(M, N, Cin, k * k, 1, ups, bool(geglu), Cin2)
        out(f"M{M} N{N} K{k * k * Cin + Cin2}{' geglu' if geglu else ''}: table {tuned and (tuned['tile'], tuned['split_k'])} "
            f"{base:7.1f} us {fl / base / 1e6:7.1f} TF")
        best = None
        for tile, sk, order in CONFIGS:
            if geglu and tile in (64, 67):
                continue
            if True:
                if True:
                    try:
                        us = time_graph(lambda: ops.conv_gemm(x, pw, ups=ups, x2=x2, out=y, tile=tile, split_k=sk, order=order))
                    except Exception as e:      # noqa: BLE001
                        out(f"    t{tile} s{sk} o{order}: {e}")
                        continue
                    err = float((y.float() - ref.float()).norm() / ref.float().norm())
                    out(f"    t{tile} s{sk} o{order}: {us:7.1f} us {fl / us / 1e6:7.1f} TF   vs table {base / us:5.2f}x   rel diff {err:.1e}")
                    if best is None or us < best[0]:
                        best = (us, tile, sk, order)
        out(f"  best SK {best}  speed-up {base / best[0]:.2f}x")
    # plain GEMM
    for n in (() if args.tiles128 else ((4096, 8192) if not args.quick else (8192,))):
        a = torch.randn(n, n, generator=gen).bfloat16().to(dev)
        bt = torch.randn(n, n, generator=gen).bfloat16()
        pw = ops.pack_weight(bt.float(), None, device=dev)
        xx = a.view(1, n, 1, n)
        fl = 2.0 * n ** 3
        bm = bt.to(dev).t().contiguous()
        lib = time_graph(lambda: torch.matmul(a, bm), reps=3)
        base = time_graph(lambda: ops.conv_gemm(xx, pw, pad=0), reps=3)
        out(f"GEMM {n}^3: hipBLASLt {fl / lib / 1e6:7.1f} TF, table / heuristic tile {fl / base / 1e6:7.1f} TF")
        for tile, sk, order in CONFIGS:
            if True:
                if True:
                    us = time_graph(lambda: ops.conv_gemm(xx, pw, pad=0, tile=tile, split_k=sk, order=order), reps=3)
                    out(f"    t{tile} s{sk} o{order}: {fl / us / 1e6:7.1f} TF")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_sk128.txt" if args.tiles128 else "bench_sk.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
