"""Ring tiles of the LDS-DMA kernel on the conv / linear launches of the headline forward, launch by launch from replayed HIP graphs:
the table's tile against every ring tile (3-8 stages, plain ring schedule).  Run once with the shipped library and once with an
-DAPTP_IL=1 build (APTP_LIB) to see what the hand-interleaved K-step does per shape.  Usage: python3 tools/bench_il.py [--all]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import fixed_half_mask  # noqa: E402
from diffusion_pruning_amd import ops  # noqa: E402
from diffusion_pruning_amd._lib import ACT_GEGLU  # noqa: E402
from diffusion_pruning_amd.unet import UNet2DConditionModelGated  # noqa: E402
from tools.tune_convs import clone_params, time_launch  # noqa: E402

dev = torch.device("cuda:0")
RING = [13, 14, 15, 16, 17, 18, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 45, 46, 47, 48]
lib = ops._lib.load()
model = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
model.set_structure(fixed_half_mask(model.get_structure(), dev))
g = torch.Generator().manual_seed(1234)
sample, ehs = torch.randn(4, 4, 64, 64, generator=g).to(dev), torch.randn(4, 77, 1024, generator=g).to(dev)
t = torch.full((4,), 500, dtype=torch.int64, device=dev)
with torch.no_grad():
    model(sample, t, ehs)
    ops.LAUNCH_LOG = []
    model(sample, t, ehs)
    torch.cuda.synchronize()
log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
uniq = {}
for rec in log:
    if "fn" in rec:
        continue
    p = rec["params"]
    key = ops.tuning_key(p.B * p.Hout * p.Wout, p.N, p.Cin, p.KH * p.KW, p.stride, p.ups, p.act == ACT_GEGLU, p.Cin2 if p.x2 else 0)
    u = uniq.setdefault(key, {"p": p, "n": 0, "flops": rec["flops"]})
    u["n"] += 1
ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
SPLIT = "--split" in sys.argv      # also: 2-stage / ring-3 tiles with MORE K-slices than the table (two workgroups per CU instead of one)
counters = ops._tile_counters(dev)
tot_inc = tot_best = 0.0
rows = []
for key, u in uniq.items():
    p0 = u["p"]
    if "--all" not in sys.argv and p0.KH * p0.KW == 1:
        continue
    base = clone_params(p0)
    t_inc = min(time_launch(lib, base), time_launch(lib, base))
    best = (t_inc, p0.tile)
    for tl in RING:
        q = clone_params(p0)
        q.tile = tl
        q.workspace = ws.data_ptr() if q.split_k > 1 else None
        us = time_launch(lib, q)
        if us is not None and us < best[0]:
            best = (us, tl)
    if SPLIT:
        nK = p0.KH * p0.KW * (p0.cin_pad // 64) + ((p0.cin2_pad // 64) if p0.x2 else 0)
        for tl in (19, 28, 8, 14, 10, 16, 7, 13, 21, 30):
            for sk in (2, 3, 4, 6):
                if sk <= p0.split_k or nK // sk < 3:
                    continue
                q = clone_params(p0)
                q.tile, q.split_k = tl, sk
                q.workspace = ws.data_ptr()
                q.tile_counters = counters.data_ptr()
                if lib.aptp_conv_gemm_workspace_bytes(ctypes.byref(q)) > ws.numel() or lib.aptp_conv_gemm_tiles(ctypes.byref(q)) > 4096:
                    continue
                us = time_launch(lib, q)
                if us is not None and us < best[0]:
                    best = (us, f"{tl} s{sk} in-kernel")
    rows.append((t_inc * u["n"], key, u["n"], t_inc, p0.tile, p0.split_k, best))
    tot_inc += t_inc * u["n"]
    tot_best += best[0] * u["n"]
rows.sort(reverse=True)
for tot, key, n, t_inc, tile, sk, best in rows:
    print(f"{key:38s} x{n:2d}  table tile {tile:2d} s{sk}: {t_inc:7.1f} us   best of table + ring tiles: {best[0]:7.1f} us (t{best[1]})")
print(f"sum over the step: table {tot_inc:.0f} us, best {tot_best:.0f} us   [library: {os.environ.get('APTP_LIB', 'shipped')}]")
