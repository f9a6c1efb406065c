"""In-situ autotune of the conv_gemm tuning table.

tools/tune_convs.py times one launch at a time (warm, or after a cache flush); at the level the table has reached, that no
longer predicts a launch's cost INSIDE the step: the family measures 3.82 ms as a sum of isolated launches and 4.17 ms in the
replayed forward, and a 60-entry isolated re-tune of the training shapes changed nothing in situ.  Here
the objective is the thing itself: the whole forward, captured into a HIP graph with ONE table entry replaced, replayed
`--replays` times.  Greedy coordinate descent over the shapes in order of their share of the step; candidates are every
distinct (tile, split_k, order, in_kernel) the table already uses for a shape of the same class, plus the neighbours of the
incumbent (same tile with split_k +-, the other XCD order).  A candidate replaces the incumbent when the step gets faster by
more than `--margin` (default 0.25 %: run-to-run noise of a 100-replay measurement is ~0.1 %), confirmed by a second
measurement.  Writes gpurun_out/tuning_gfx950.json (the committed table with the winners merged in) and a log.

Usage: python tools/tune_insitu.py [--dense] [--expert N] [--top 24] [--replays 100] [--budget-s 480]"""
import argparse
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import expert_mask, fixed_half_mask, ones_mask  # noqa: E402
from diffusion_pruning_amd import ops  # noqa: E402
from diffusion_pruning_amd._lib import ACT_GEGLU  # noqa: E402
from diffusion_pruning_amd.unet import UNet2DConditionModelGated, UNet2DConditionModelPruned  # noqa: E402

dev = torch.device("cuda:0")
KEY = re.compile(r"M(\d+)_N(\d+)_C(\d+)_T(\d+)_s(\d+)u(\d+)g(\d+)(?:x(\d+))?$")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--expert", type=int, default=None, help="tune the inference forward of benchmark expert N (pruned model)")
    ap.add_argument("--top", type=int, default=24, help="shapes (by share of the family's time) to visit")
    ap.add_argument("--replays", type=int, default=100)
    ap.add_argument("--margin", type=float, default=0.0025)
    ap.add_argument("--budget-s", type=float, default=480.0)
    ap.add_argument("--wide", type=int, default=0, help="also shortlist the N fastest (isolated, warm) of ALL (tile, split-K, order) "
                    "combinations per shape and evaluate those in situ")
    ap.add_argument("--apply", action="store_true", help="also write the merged table over the package's own (for chained runs in one job)")
    ap.add_argument("--batch", type=int, default=4, help="U-Net batch of the forward being tuned (16 = the reference's evaluation batch)")
    ap.add_argument("--lean", action="store_true", help="round 4: visit only the plain linear layers (1x1, stride 1) and try every tile the lean "
                    "kernel of csrc/lin_gemm.hip instantiates (one K-slice), in both XCD orders for the best one")
    args = ap.parse_args()
    t_start = time.time()
    ops._lib.load()
    if args.expert is not None:
        model = UNet2DConditionModelPruned().init_synthetic(seed=0).to(dev)
        model.prune(expert_mask(model.get_structure(), args.expert, dev))
    else:
        model = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
        st = model.get_structure()
        model.set_structure(fixed_half_mask(st, dev))
    B = args.batch
    g = torch.Generator().manual_seed(1234)
    sample = torch.randn(B, 4, 64, 64, generator=g).to(dev)
    ehs = torch.randn(B, 77, 1024, generator=g).to(dev)
    t = torch.full((B,), 500, dtype=torch.int64, device=dev)
    headline_keys = set()
    if args.dense:
        # the dense forward shares a few shapes with the headline (masked) forward: those keep the entries tuned on the headline
        with torch.no_grad():
            model(sample, t, ehs, return_dict=False)
            ops.LAUNCH_LOG = []
            model(sample, t, ehs, return_dict=False)
            torch.cuda.synchronize()
        hl, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
        for rec in hl:
            if "fn" not in rec:
                p = rec["params"]
                headline_keys.add(ops.tuning_key(p.B * p.Hout * p.Wout, p.N, p.Cin, p.KH * p.KW, p.stride, p.ups, p.act == ACT_GEGLU, p.Cin2 if p.x2 else 0))
        model.set_structure(ones_mask(st, dev))

    def fwd():
        return model(sample, t, ehs, return_dict=False)[0]

    side = torch.cuda.Stream()          # ONE warm-up stream: split-K counter slabs are handed out per launching stream

    def measure(replays):
        """steps/s of the forward captured NOW (with whatever ops.TUNING holds)"""
        with torch.no_grad():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fwd()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fwd()
        for _ in range(10):
            graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        del graph
        return replays / (e0.elapsed_time(e1) * 1e-3)

    # the step's launches and their isolated cost (to order the shapes)
    with torch.no_grad():
        fwd()
        ops.LAUNCH_LOG = []
        fwd()
        torch.cuda.synchronize()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    shapes = {}
    for rec in log:
        if "fn" in rec:
            continue
        p = rec["params"]
        key = ops.tuning_key(p.B * p.Hout * p.Wout, p.N, p.Cin, p.KH * p.KW, p.stride, p.ups, p.act == ACT_GEGLU, p.Cin2 if p.x2 else 0)
        s = shapes.setdefault(key, {"count": 0, "flops": rec["flops"], "params": p})
        s["count"] += 1
    LEAN_TILES = (12, 18, 25, 49, 9, 15, 24, 51, 11, 17, 26, 53)
    order = sorted(shapes, key=lambda k: -shapes[k]["flops"] * shapes[k]["count"] ** 0.5)
    if args.lean:
        order = [k for k in order if int(KEY.match(k).group(4)) == 1 and int(KEY.match(k).group(5)) == 1 and int(KEY.match(k).group(1)) >= 256]
        order.sort(key=lambda k: -shapes[k]["count"] * (shapes[k]["flops"] ** 0.5))
    order = order[:args.top]

    def cls(key):
        m = KEY.match(key)
        M, N, C, T, s_, u, g_, x2 = (int(v) if v else 0 for v in m.groups())
        return (T, s_, u, g_, x2 > 0), M, N, T * C + x2

    by_class = {}
    for k, v in ops.TUNING.items():
        if KEY.match(k):
            by_class.setdefault(cls(k)[0], set()).add((v["tile"], v["split_k"], v.get("order", 1), int(v.get("in_kernel", 0))))

    lines = []

    def out(s):
        print(s, flush=True)
        lines.append(s)

    # a candidate is first launched ONCE eagerly on the recorded operands (statistics outputs cleared: their sizes depend on the
    # tile and the forward re-derives them): a tile that cannot run the shape must fail here, not inside a stream capture
    import ctypes
    from tools.tune_convs import clone_params
    lib = ops._lib.load()
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    counters = ops._tile_counters(dev)

    def runnable(key, cd):
        q = clone_params(shapes[key]["params"])
        q.tile, q.split_k, q.order = cd[0], cd[1], cd[2]
        q.workspace = ws.data_ptr() if cd[1] > 1 else None
        q.tile_counters = counters.data_ptr() if (cd[1] > 1 and (cd[3] or cd[0] >= ops.SK_TILE_FIRST)) else None
        q.prefetch, q.prefetch_bytes = None, 0
        if cd[1] > 1 and lib.aptp_conv_gemm_workspace_bytes(ctypes.byref(q)) > ws.numel():
            return False
        rc = lib.aptp_conv_gemm(ctypes.byref(q), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return rc == 0

    committed = set(ops.TUNING)           # an expert run only ADDS entries: the headline / dense / train shapes keep theirs
    def isolated_us(key, cd, reps=6):
        """warm isolated time of one candidate on the recorded operands (None: cannot run); the recorded output is not written"""
        q = clone_params(shapes[key]["params"])
        q.tile, q.split_k, q.order = cd[0], cd[1], cd[2]
        q.workspace = ws.data_ptr() if cd[1] > 1 else None
        q.tile_counters = counters.data_ptr() if (cd[1] > 1 and (cd[3] or cd[0] >= ops.SK_TILE_FIRST)) else None
        q.prefetch, q.prefetch_bytes = None, 0
        if cd[1] > 1 and lib.aptp_conv_gemm_workspace_bytes(ctypes.byref(q)) > (512 << 20):
            return None
        if q.B * q.Hout * q.Wout * max(q.ldy, q.N) * 4 > (512 << 20):
            return None
        q.y = ws.data_ptr() + (512 << 20)
        q.residual = q.depth_in = None
        s = torch.cuda.current_stream().cuda_stream
        if lib.aptp_conv_gemm(ctypes.byref(q), s) != 0:
            torch.cuda.synchronize()
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.aptp_conv_gemm(ctypes.byref(q), s)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    def wide_shortlist(key, nK, n):
        scored = []
        for tile in range(1, ops.SK_TILE_FIRST):
            for sk in (1, 2, 3, 4, 6, 8):
                if sk > 1 and nK // sk < 3:
                    continue
                for o in (1, 3):
                    us = isolated_us(key, (tile, sk, o, 1 if sk > 1 else 0))
                    if us is None:
                        break
                    scored.append((us, (tile, sk, o, 1 if sk > 1 else 0)))
                    if sk > 1:
                        scored.append((us, (tile, sk, o, 0)))          # (the two-launch form: same GEMM time, decided in situ)
        scored.sort(key=lambda t: t[0])
        return [cd for _, cd in scored[:n]]

    base = max(measure(args.replays), measure(args.replays))
    out(f"baseline {base:.2f} steps/s, {len(shapes)} distinct shapes, visiting {len(order)}")
    changed = {}
    for key in order:
        if time.time() - t_start > args.budget_s:
            out("time budget reached")
            break
        if key in headline_keys:
            out(f"{key:40s} x{shapes[key]['count']:2d}  a shape of the headline forward: kept")
            continue
        if args.expert is not None and key in committed:
            out(f"{key:40s} x{shapes[key]['count']:2d}  shared with the tuned workloads: kept")
            continue
        c, M, N, K = cls(key)
        nK = (K + 63) // 64
        cur = ops.tuning_lookup(*[int(v) if v else 0 for v in KEY.match(key).groups()[:7]], int(KEY.match(key).group(8) or 0))
        if cur is None:
            # no table entry and no neighbour: the launch took ops.conv_gemm's own choice (lean tile by row count, stream-K macro-tile,
            # library heuristic) -- that is the incumbent
            q0 = shapes[key]["params"]
            if q0.tile == 0:
                continue
            cur = {"tile": int(q0.tile), "split_k": int(q0.split_k), "order": int(q0.order) if q0.order else 1, "in_kernel": int(bool(q0.tile_counters))}
        inc = (cur["tile"], cur["split_k"], cur.get("order", 1), int(cur.get("in_kernel", 0)))
        cands = set(by_class.get(c, ()))
        for dk in (-2, -1, 1, 2):                           # neighbours of the incumbent
            sk = inc[1] + dk
            if 1 <= sk <= max(1, nK // 3):
                cands.add((inc[0], sk, inc[2], 1 if sk > 1 else 0))
        for o in (2, 3):
            cands.add((inc[0], inc[1], o, inc[3]))
        if args.wide:
            cands.update(wide_shortlist(key, nK, args.wide))
        if args.batch != 4:
            T_ = int(KEY.match(key).group(4))
            if T_ == 1:
                cands.update({(t_, 1, inc[2], 0) for t_ in (11, 17, 12, 18, 49, 9, 15)})
            else:
                cands.update({(t_, sk_, 3, 1 if sk_ > 1 else 0) for t_ in (64, 65, 67, 19, 34, 56, 20) for sk_ in (1, 2)})
        if args.lean:
            cands = {(t_, 1, inc[2], 0) for t_ in LEAN_TILES}
            cands.update({(t_, 1, 5 - inc[2] if inc[2] in (2, 3) else 3, 0) for t_ in (12, 18, 49, 9, 11)})
        cands.discard(inc)
        cands = [cd for cd in cands if not (cd[1] > 1 and nK // cd[1] < 3)][:28 + args.wide]
        best, best_v = inc, base
        saved = ops.TUNING.get(key)
        for cd in sorted(cands):
            if time.time() - t_start > args.budget_s:
                break
            if not runnable(key, cd):
                continue
            ops.TUNING[key] = {"tile": cd[0], "split_k": cd[1], "order": cd[2], "in_kernel": cd[3]}
            ops._tuning_near_cache.clear()
            try:
                v = measure(args.replays)
            except Exception as e:  # noqa: BLE001    (a tile that cannot run this shape: GEGLU on a 160-wide tile, halo geometry, ...)
                torch.cuda.synchronize()
                continue
            if v > best_v * (1 + args.margin):
                v2 = measure(args.replays)                  # confirm
                if min(v, v2) > best_v * (1 + args.margin):
                    best, best_v = cd, min(v, v2)
        if best != inc:
            ops.TUNING[key] = {"tile": best[0], "split_k": best[1], "order": best[2], "in_kernel": best[3], "insitu": round(best_v, 2)}
            changed[key] = ops.TUNING[key]
            out(f"{key:40s} x{shapes[key]['count']:2d}  {inc} -> {best}   {base:.2f} -> {best_v:.2f} steps/s")
            base = best_v
        else:
            if saved is None:
                ops.TUNING.pop(key, None)
            else:
                ops.TUNING[key] = saved
            out(f"{key:40s} x{shapes[key]['count']:2d}  keeps {inc}")
        ops._tuning_near_cache.clear()
    final = max(measure(args.replays), measure(args.replays))
    out(f"final {final:.2f} steps/s with {len(changed)} entries changed")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    src = os.path.join(ROOT, "diffusion_pruning_amd", "tuning_gfx950.json")
    table = json.load(open(src)) if os.path.exists(src) else {}
    table.update(changed)
    tag = "dense" if args.dense else ("expert%d" % args.expert if args.expert is not None else "masked")
    if args.batch != 4:
        tag += "_bs%d" % args.batch
    json.dump(table, open(os.path.join(ROOT, "gpurun_out", "tuning_gfx950.json"), "w"), indent=0, sort_keys=True)
    if args.apply:
        json.dump(table, open(src, "w"), indent=0, sort_keys=True)
    with open(os.path.join(ROOT, "gpurun_out", "tune_insitu_%s.txt" % tag), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
