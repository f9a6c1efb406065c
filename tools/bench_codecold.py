"""Is the fixed cost of a cold conv_gemm launch its instruction fetch?  For launches of the headline forward: time one launch
after a 768 MB cache fill (weights, instructions and everything else cold; activations re-warmed) against the same launch
preceded by a launch of the SAME kernel on a copy of the weights (instructions, bias, residual warm; the timed launch's
weights still cold), and against the warm replay time.  Usage: python tools/bench_codecold.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import ops
from diffusion_pruning_amd.unet import UNet2DConditionModelGated
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tune_convs import clone_params, time_launch, fixed_half_mask, dev

lib = ops._lib.load()
model = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
model.set_structure(fixed_half_mask(model.get_structure(), dev))
B = 4
sample, ehs = torch.randn(B, 4, 64, 64, device=dev), torch.randn(B, 77, 1024, device=dev)
t = torch.full((B,), 500, dtype=torch.int64, device=dev)
with torch.no_grad():
    model(sample, t, ehs)
    ops.LAUNCH_LOG = []
    model(sample, t, ehs)
    torch.cuda.synchronize()
log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
seen = {}
for rec in log:
    p = rec["params"]
    key = ops.tuning_key(p.B * p.Hout * p.Wout, p.N, p.Cin, p.KH * p.KW, p.stride, p.ups, p.act == ops.ACT_GEGLU, p.Cin2 if p.x2 else 0)
    seen.setdefault(key, [rec, 0])[1] += 1
tot = [0.0, 0.0, 0.0]
for key, (rec, cnt) in seen.items():
    p = rec["params"]                      # (the recorded launch as it ran, statistics outputs included)
    x, pw = rec["keep"][0], rec["keep"][1]
    w_alias = pw.w.clone()
    q = ops.ConvGemmParams()
    ctypes.memmove(ctypes.byref(q), ctypes.byref(p), ctypes.sizeof(ops.ConvGemmParams))
    q.w = w_alias.data_ptr()
    res = []
    for warm_code in (False, True):
        ts = []
        for _ in range(7):
            flush.fill_(1)
            x.float().sum()
            if warm_code:
                lib.aptp_conv_gemm(ctypes.byref(q), s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.aptp_conv_gemm(ctypes.byref(p), s)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res.append(ts[3])
    warm = time_launch(lib, clone_params(p))
    tot[0] += res[0] * cnt; tot[1] += res[1] * cnt; tot[2] += (warm or 0) * cnt
    print(f"{key:36s} x{cnt:2d}  cold {res[0]:6.1f}  code-warm {res[1]:6.1f}  warm {warm:6.1f} us", flush=True)
print(f"per forward: cold {tot[0] / 1e3:.3f} ms, code-warm/weights-cold {tot[1] / 1e3:.3f} ms, warm {tot[2] / 1e3:.3f} ms")
