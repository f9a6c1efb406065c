"""Per-kernel timing sweep on the GPU (tile / split-K choices for the SD-2.1 GEMM shapes, attention, norms).
Writes gpurun_out/bench_ops.json.  Usage: python tools/bench_ops.py [--quick]"""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def conv_case(B, H, Cin, Cout, k, tiles, splits):
    x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    w = torch.randn(Cout, Cin, k, k) * 0.02
    pw = ops.pack_weight(w, torch.zeros(Cout), device=dev)
    flop = 2.0 * B * H * H * Cout * Cin * k * k
    res = []
    for t in tiles:
        for s in splits:
            try:
                us = timeit(lambda: ops.conv_gemm(x, pw, tile=t, split_k=s))
                res.append({"tile": t, "split_k": s, "us": round(us, 1), "tflops": round(flop / us / 1e6, 1)})
            except Exception as ex:  # noqa: BLE001
                res.append({"tile": t, "split_k": s, "err": str(ex)[:80]})
    best = min((r for r in res if "us" in r), key=lambda r: r["us"])
    return {"shape": [B, H, Cin, Cout, k], "M": B * H * H, "N": Cout, "K": Cin * k * k, "best": best, "all": res}


def main():
    quick = "--quick" in sys.argv
    out = {"conv": [], "attn": [], "norm": []}
    B = 4
    shapes = [
        (B, 64, 320, 320, 3), (B, 64, 160, 320, 3), (B, 64, 320, 160, 3), (B, 64, 640, 320, 3), (B, 64, 960, 160, 3),
        (B, 32, 640, 640, 3), (B, 32, 320, 640, 3), (B, 32, 1280, 320, 3), (B, 32, 1920, 320, 3),
        (B, 16, 1280, 1280, 3), (B, 16, 640, 1280, 3), (B, 16, 2560, 640, 3),
        (B, 8, 1280, 1280, 3), (B, 8, 2560, 640, 3), (B, 8, 640, 1280, 3),
        # linears (k=1 over tokens): QKV, GEGLU-width, FF-out
        (B, 64, 320, 384, 1), (B, 64, 320, 960, 1), (B, 64, 320, 1280, 1), (B, 64, 640, 320, 1), (B, 64, 320, 320, 1),
        (B, 32, 640, 960, 1), (B, 32, 640, 2560, 1), (B, 32, 1280, 640, 1),
        (B, 16, 1280, 1920, 1), (B, 16, 1280, 5120, 1), (B, 16, 2560, 1280, 1),
    ]
    if quick:
        shapes = shapes[:3] + shapes[15:17]
    for (b, h, ci, co, k) in shapes:
        M = b * h * h
        tiles = [1, 2, 3, 4] if co % 160 == 0 else [1, 3, 5, 6]
        splits = [1] if M >= 4096 else [1, 2, 4, 8, 16]
        r = conv_case(b, h, ci, co, k, tiles, splits)
        out["conv"].append(r)
        print(r["shape"], "best", r["best"], flush=True)
    for (b, h, L, Lk) in [(4, 2, 4096, 4096), (4, 5, 4096, 4096), (4, 5, 1024, 1024), (4, 10, 256, 256), (4, 5, 4096, 77), (4, 10, 1024, 77)]:
        q = torch.randn(b, L, h * 64, device=dev).bfloat16()
        k = torch.randn(b, Lk, h * 64, device=dev).bfloat16()
        v = torch.randn(b, Lk, h * 64, device=dev).bfloat16()
        us = timeit(lambda: ops.attention(q, k, v, h))
        flop = 4.0 * b * h * L * Lk * 64
        r = {"shape": [b, h, L, Lk], "us": round(us, 1), "tflops": round(flop / us / 1e6, 1)}
        out["attn"].append(r)
        print("attn", r, flush=True)
    for (b, hw, c) in [(4, 64, 320), (4, 64, 640), (4, 32, 640), (4, 32, 1920), (4, 16, 1280), (4, 8, 2560)]:
        x = torch.randn(b, hw, hw, c, device=dev).bfloat16()
        ga, be = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        us = timeit(lambda: ops.groupnorm(x, ga, be, 32, 1e-5, True))
        gb = 3.0 * x.numel() * 2 / 1e9
        r = {"gn": [b, hw, c], "us": round(us, 1), "GBps": round(gb / us * 1e6, 1)}
        out["norm"].append(r)
        print(r, flush=True)
        x3 = x.view(b, hw * hw, c)
        us = timeit(lambda: ops.layernorm(x3, ga, be))
        r = {"ln": [b, hw, c], "us": round(us, 1), "GBps": round(2.0 * x.numel() * 2 / 1e9 / us * 1e6, 1)}
        out["norm"].append(r)
        print(r, flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/bench_ops.json", "w"), indent=1)


if __name__ == "__main__":
    main()
