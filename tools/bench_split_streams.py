"""Experiment: one bs=4 forward as two bs=2 forwards on two streams inside one HIP graph (independent samples: do the
latency chains of the two halves overlap?).  Usage: python tools/bench_split_streams.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import fixed_half_mask
from diffusion_pruning_amd import ops
from diffusion_pruning_amd.unet import UNet2DConditionModelGated

dev = torch.device("cuda:0")
model = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
model.set_structure(fixed_half_mask(model.get_structure(), dev))
g = torch.Generator(device="cpu").manual_seed(1234)
B = 4
sample = torch.randn(B, 4, 64, 64, generator=g).to(dev)
ehs = torch.randn(B, 77, 1024, generator=g).to(dev)
t = torch.full((B,), 500, dtype=torch.int64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def fwd_whole():
    with torch.no_grad():
        return model(sample, t, ehs).sample


def fwd_split(nsplit):
    cur = torch.cuda.current_stream()
    outs = []
    streams = [s1, s2][:nsplit]
    per = B // nsplit
    with torch.no_grad():
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(model(sample[i * per:(i + 1) * per], t[i * per:(i + 1) * per], ehs[i * per:(i + 1) * per]).sample)
        for s in streams:
            cur.wait_stream(s)
    return torch.cat(outs, 0)


def bench(fn, name):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            out = fn()
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{name}: {ms:.3f} ms/step  {1e3 / ms:.1f} steps/s", flush=True)
    return out.float()


ref = bench(fwd_whole, "whole batch, one stream")
o2 = bench(lambda: fwd_split(2), "2 x bs=2 on two streams")
print("max diff split vs whole:", float((o2 - ref).abs().max()), "scale", float(ref.abs().max()))
