"""Autotune (tile, split_k) of aptp_conv_gemm for every distinct launch of the SD-2.1 forward (masked and dense, bs=4),
timed on the GPU with HIP-graph replays.  Writes diffusion_pruning_amd/tuning_gfx950.json (merged with the existing
table) and gpurun_out/tune_convs.txt (per-launch table sorted by time).
Usage: python tools/tune_convs.py [--dense] [--batch 4] [--quick]"""
import argparse
import copy
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import fixed_half_mask, ones_mask  # noqa: E402
from diffusion_pruning_amd import ops  # noqa: E402
from diffusion_pruning_amd._lib import ACT_GEGLU, ConvGemmParams  # noqa: E402
from diffusion_pruning_amd.unet import UNet2DConditionModelGated  # noqa: E402

dev = torch.device("cuda:0")


def clone_params(p):
    q = ConvGemmParams()
    ctypes.memmove(ctypes.byref(q), ctypes.byref(p), ctypes.sizeof(ConvGemmParams))
    # the statistics outputs are sized for the tile of the recorded launch (slots / row blocks depend on the tile):
    # candidates are timed without them
    q.rowstat_out, q.rowstat_slots, q.colstat_out, q.colstat_ld = None, 0, None, 0
    q.ustat_out = None          # (unit statistics go with colstat_out: AptpConvGemmParams.ustat_out)
    return q


def time_launch(lib, p, reps=10):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        s = torch.cuda.current_stream().cuda_stream
        rc = lib.aptp_conv_gemm(ctypes.byref(p), s)
        if rc != 0:
            return None
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(reps):
                lib.aptp_conv_gemm(ctypes.byref(p), torch.cuda.current_stream().cuda_stream)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        g.replay(); g.replay(); g.replay()
        e1.record(stream)
        stream.synchronize()
        return e0.elapsed_time(e1) / (3 * reps) * 1e3


_flush = None


def time_launch_cold(lib, p, x_view, reps=7):
    """One launch at a time with the caches in the state the forward leaves them: weights cold (a 768 MB fill evicts L2 and
    the Infinity Cache), the activation operand warm (re-read after the fill), timed with events around the launch."""
    global _flush
    if _flush is None:
        _flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    rc = lib.aptp_conv_gemm(ctypes.byref(p), s)
    if rc != 0:
        return None
    ts = []
    for _ in range(reps):
        _flush.fill_(1)
        x_view.float().sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.aptp_conv_gemm(ctypes.byref(p), s)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cold", action="store_true", help="time every candidate with cold weights / warm activations")
    ap.add_argument("--orders", action="store_true", help="also tune the XCD-aware workgroup order (include/aptp_hip.h)")
    ap.add_argument("--modes", action="store_true", help="also tune the split-K form: in-kernel reduction vs reduce launch")
    ap.add_argument("--refine", action="store_true", help="start from the committed table: a candidate must beat the current entry by 3 %")
    ap.add_argument("--train", action="store_true", help="tune the launches of one eager pruning train step (teacher + student forward, data-gradient GEMMs) instead of the inference forward")
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64, help="--train: latent size of the recorded step (the reference trains at 32 with batch 64)")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--tiles", type=str, default="", help="comma-separated APTP_TILE_* ids: only these are candidates (with --refine: a new tile against the committed table)")
    args = ap.parse_args()
    lib = ops._lib.load()
    if args.train:
        log = record_train_step(args)
        tune(args, lib, log)
        return
    model = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
    st = model.get_structure()
    model.set_structure(ones_mask(st, dev) if args.dense else fixed_half_mask(st, dev))
    B = args.batch
    sample = torch.randn(B, 4, 64, 64, device=dev)
    ehs = torch.randn(B, 77, 1024, device=dev)
    t = torch.full((B,), 500, dtype=torch.int64, device=dev)
    if not args.refine:
        ops.TUNING = {}      # record with heuristics only
    with torch.no_grad():
        model(sample, t, ehs)
        ops.LAUNCH_LOG = []
        model(sample, t, ehs)
        torch.cuda.synchronize()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    tune(args, lib, log)


def record_train_step(args):
    """the conv_gemm launches of one eager APTP pruning step (tools/bench_train.py's set-up)"""
    from diffusion_pruning_amd.hypernet import HyperStructure
    from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
    from diffusion_pruning_amd.train_step import PrunerStep, synthetic_batch
    unet = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
    unet.freeze()
    st = unet.get_structure()
    torch.manual_seed(0)
    hn = HyperStructure(structure=st, input_dim=768, wn_flag=False, linear_bias=True).to(dev)
    qz = StructureVectorQuantizer(n_e=8, structure=st, temperature=0.4, base=3,
                                  depth_order=[-1, -2, 0, 1, -3, -4, 2, 3, -5, -6, 4, 5, -7, 6],
                                  resource_aware_normalization=False, optimal_transport=True).to(dev)
    hn.train(); qz.train()
    step = PrunerStep(unet, hn, qz)
    step.count_macs(args.latent)
    opt = torch.optim.AdamW(step.trainable_parameters(), lr=2e-4)
    batch = synthetic_batch(args.batch, args.latent, dev)
    if not args.refine:
        ops.TUNING = {}
    step.train_step(opt, batch)
    ops.LAUNCH_LOG = []
    step.train_step(opt, batch)
    torch.cuda.synchronize()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    return log


def tune(args, lib, log):
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    counters = ops._tile_counters(dev)
    uniq = {}
    for rec in log:
        if "fn" in rec:                   # aptp_ff_tail etc.: not a conv_gemm launch
            continue
        p = rec["params"]
        key = ops.tuning_key(p.B * p.Hout * p.Wout, p.N, p.Cin, p.KH * p.KW, p.stride, p.ups, p.act == ACT_GEGLU, p.Cin2 if p.x2 else 0)
        u = uniq.setdefault(key, {"rec": rec, "count": 0})
        u["count"] += 1
    print(f"{len(log)} launches, {len(uniq)} distinct", flush=True)
    table, rows = {}, []
    for key, u in uniq.items():
        p0 = u["rec"]["params"]
        M, nK = p0.B * p0.Hout * p0.Wout, p0.KH * p0.KW * (p0.cin_pad // 64) + ((p0.cin2_pad // 64) if p0.x2 else 0)
        base = clone_params(p0)
        if args.cold:
            xk = u["rec"]["keep"][0]
            timer = lambda q: time_launch_cold(lib, q, xk)
        else:
            timer = lambda q: time_launch(lib, q)
        t_base = timer(base)
        tiles = [1, 3, 5, 6, 7, 9, 11, 12, 13, 15, 17, 18, 21, 22, 24, 25, 26, 27, 30, 31, 32, 35, 36, 38, 40] if p0.act == ACT_GEGLU else list(range(1, 70))
        if p0.act == ACT_GEGLU:
            tiles += [45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 57, 58, 59, 60, 61, 62, 63, 65, 66, 68, 69]
        if args.tiles:
            tiles = [int(v) for v in args.tiles.split(",") if int(v) in tiles]
        splits = [1, 2, 3, 4, 6, 8, 12, 16, 24]
        if args.quick:
            splits = [1, 2, 4, 8]
        if not args.refine:
            base.tile_counters = counters.data_ptr() if base.split_k > 1 else None
        if args.refine:
            t_base = min(t_base, timer(base)) * 0.97 if t_base is not None else None
        best = (t_base, base.tile, base.split_k, base.order, 1 if base.tile_counters else 0)
        for tl in tiles:
            for sk in splits:
                if tl >= ops.SK_TILE_FIRST:
                    # persistent stream-K macro-tiles: split_k is a switch (1 = whole tiles, 2 = K split on), always in-kernel
                    if sk > 2 or (sk == 2 and nK < 2):
                        continue
                elif sk > 1 and (nK // sk < 3 or (M + 255) * (p0.N + 255) * sk * 4 > ws.numel() or M >= 8192 and sk > 2):
                    continue
                for order in ((2, 3) if args.orders else (0,)):     # XCD-aware workgroup orders: weight- / activation-major
                    for ink in ((1, 0) if (args.modes and sk > 1 and tl < ops.SK_TILE_FIRST) else (1,)):
                        q = clone_params(p0)
                        q.tile, q.split_k, q.order = tl, sk, order
                        q.workspace = ws.data_ptr() if sk > 1 else None
                        q.tile_counters = counters.data_ptr() if (sk > 1 and ink) else None
                        us = timer(q)
                        if us is not None and us < best[0]:
                            best = (us, tl, sk, order, ink)
        if args.orders and best[3] != 0:
            # keep the legacy order honest: re-time the winner against it
            q = clone_params(p0)
            q.tile, q.split_k, q.order = best[1], best[2], 1
            q.workspace = ws.data_ptr() if best[2] > 1 else None
            q.tile_counters = counters.data_ptr() if (best[2] > 1 and best[4]) else None
            us = min(u for u in (timer(q), timer(q)) if u is not None)
            if us < best[0]:
                best = (us, best[1], best[2], 1, best[4])
        table[key] = {"tile": best[1], "split_k": best[2], "order": best[3], "us": round(best[0], 2)}
        if best[2] > 1:
            table[key]["in_kernel"] = int(best[4])
        fl = u["rec"]["flops"]
        rows.append((best[0] * u["count"], key, u["count"], t_base, best, fl))
        print(f"{key:36s} x{u['count']:2d} heur {t_base:7.1f} us (t{base.tile} s{base.split_k}) -> best {best[0]:7.1f} us "
              f"(t{best[1]} s{best[2]} o{best[3]} k{best[4]})  {fl / best[0] / 1e6:6.1f} TF", flush=True)
    rows.sort(reverse=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "tune_convs" + ("_dense" if args.dense else "") + ("_train" if args.train else "") + ".txt"), "w") as f:
        tot_h = sum(r[3] * r[2] for r in rows)
        tot_b = sum(r[0] for r in rows)
        f.write(f"# total conv_gemm time per forward: heuristic {tot_h / 1e3:.3f} ms -> tuned {tot_b / 1e3:.3f} ms\n")
        for tot, key, cnt, tb, best, fl in rows:
            f.write(f"{tot:8.1f} us  {key:36s} x{cnt:2d}  heur {tb:7.1f}  best {best[0]:7.1f} (t{best[1]} s{best[2]})  {fl / best[0] / 1e6:6.1f} TF\n")
    path = os.path.join(ROOT, "gpurun_out", "tuning_gfx950.json")
    old = {}
    src = os.path.join(ROOT, "diffusion_pruning_amd", "tuning_gfx950.json")
    if os.path.exists(src):
        old = json.load(open(src))
    old.update(table)
    json.dump(old, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
