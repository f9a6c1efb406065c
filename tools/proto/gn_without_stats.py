"""which GroupNorm calls of the headline forward find no producer statistics (and run gn_stats + gn_finalize)? (debugging aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from diffusion_pruning_amd import ops
from diffusion_pruning_amd.unet import UNet2DConditionModelGated
dev = torch.device("cuda:0")
model = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
model.set_structure(bench.fixed_half_mask(model.get_structure(), dev))
g = torch.Generator().manual_seed(1234)
sample, ehs = torch.randn(4, 4, 64, 64, generator=g).to(dev), torch.randn(4, 77, 1024, generator=g).to(dev)
t = torch.full((4,), 500, dtype=torch.int64, device=dev)
orig = ops.groupnorm
calls = []
def spy(x, *a, **k):
    C = k.get("C", None)
    B, H, W = x.shape[0], x.shape[1], x.shape[2]
    segs = ops._colstats_get(x, x.shape[3] if C is None else C) if H * W >= ops.COLSTATS_MIN_HW else "small map (one-kernel form)"
    calls.append((tuple(x.shape), C, "stats from the producer" if (segs is not None and not isinstance(segs, str)) else (segs or "NO producer statistics")))
    return orig(x, *a, **k)
ops.groupnorm = spy
import diffusion_pruning_amd.unet as U
with torch.no_grad():
    model(sample, t, ehs)
    calls.clear()
    model(sample, t, ehs)
for i, c in enumerate(calls):
    print(i, c)
