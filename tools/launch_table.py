"""Every conv_gemm-family launch of the headline forward, grouped by shape, each shape timed ALONE (a HIP graph of 8 copies of
that launch): count, extents, tile, split-K, us, TFLOP/s, time per step.  Finds the shapes whose efficiency is out of line.
usage: python tools/launch_table.py [--dense]"""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from diffusion_pruning_amd import ops
from diffusion_pruning_amd.unet import UNet2DConditionModelGated

dev = torch.device("cuda:0")
model = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
st = model.get_structure()
model.set_structure(bench.ones_mask(st, dev) if "--dense" in sys.argv else bench.fixed_half_mask(st, dev))
g = torch.Generator().manual_seed(1234)
sample = torch.randn(4, 4, 64, 64, generator=g).to(dev)
ehs = torch.randn(4, 77, 1024, generator=g).to(dev)
t = torch.full((4,), 500, dtype=torch.int64, device=dev)
lib = ops._lib.load()
with torch.no_grad():
    model(sample, t, ehs)
    torch.cuda.synchronize()
    ops.LAUNCH_LOG = []
    model(sample, t, ehs)
    torch.cuda.synchronize()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
groups = collections.OrderedDict()
for r in log:
    p = r["params"]
    if "fn" in r:
        key = (r["fn"],)
    else:
        key = (p.B * p.Hout * p.Wout, p.N, p.Cin, p.KH * p.KW, p.stride, p.ups, int(p.act), p.Cin2, int(p.tile), p.split_k,
               bool(p.ln_stats), bool(p.rowstat_out), bool(p.colstat_out), bool(p.residual))
    groups.setdefault(key, []).append(r)
stream = torch.cuda.Stream()
rows = []
for key, recs in groups.items():
    rec = recs[0]

    def fn():
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(8):
            assert getattr(lib, rec.get("fn", "aptp_conv_gemm"))(ctypes.byref(rec["params"]), s) == 0
    us = bench._time_graph(torch, stream, fn, reps=5) * 1e3 / 8
    rows.append((us * len(recs), len(recs), us, rec["flops"] / us / 1e6, key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{len(log)} launches, {len(rows)} distinct shapes, sum of isolated times {tot:.0f} us")
print(" total_us  n     us    TF/s   (M, N, Cin, taps, stride, ups, act, Cin2, tile, split_k, ln, rowstat, colstat, residual)")
for r in rows:
    print(f"{r[0]:8.1f} {r[1]:3d} {r[2]:7.1f} {r[3]:7.1f}   {r[4]}")
