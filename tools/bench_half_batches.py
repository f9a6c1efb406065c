"""Would the headline forward (bs=4) run faster as TWO bs=2 forwards replayed side by side on two streams?  (The fine-tune step
gains 7 % from replaying the teacher's forward next to the student's: two launch-bound streams fill each other's gaps.)
Captures the bs=4 graph and two bs=2 graphs (own pools, own split-K counter / scratch domains) and times 4 latents per step
either way.  The tuning table holds bs=4 shapes: the bs=2 launches take the nearest entries."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import fixed_half_mask  # noqa: E402
from diffusion_pruning_amd import ops  # noqa: E402
from diffusion_pruning_amd.unet import UNet2DConditionModelGated  # noqa: E402

dev = torch.device("cuda:0")
ops._lib.load()
model = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
model.set_structure(fixed_half_mask(model.get_structure(), dev))
g = torch.Generator().manual_seed(1234)
sample = torch.randn(4, 4, 64, 64, generator=g).to(dev)
ehs = torch.randn(4, 77, 1024, generator=g).to(dev)
t = torch.full((4,), 500, dtype=torch.int64, device=dev)


def capture(lo, hi, domain):
    s, e, tt = sample[lo:hi].contiguous(), ehs[lo:hi].contiguous(), t[lo:hi].contiguous()

    def fwd():
        with ops.scratch_domain(domain):
            return model(s, tt, e, return_dict=False)[0]
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fwd()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = fwd()
    return graph, out


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g4, o4 = capture(0, 4, "main")
ga, oa = capture(0, 2, "main")
gb, ob = capture(2, 4, "half_b")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
main = torch.cuda.current_stream()


def both():
    s1.wait_stream(main)
    s2.wait_stream(main)
    with torch.cuda.stream(s1):
        ga.replay()
    with torch.cuda.stream(s2):
        gb.replay()
    main.wait_stream(s1)
    main.wait_stream(s2)


def serial():
    ga.replay()
    gb.replay()


t4 = timed(g4.replay)
ta = timed(ga.replay)
tser = timed(serial)
tpar = timed(both)
torch.cuda.synchronize()
err = float((torch.cat([oa, ob]).float() - o4.float()).abs().max() / o4.float().abs().max())
print(f"bs=4 one graph            {t4:.3f} ms  ({1e3 / t4:.1f} steps/s)")
print(f"bs=2 one graph            {ta:.3f} ms")
print(f"two bs=2 graphs, serial   {tser:.3f} ms")
print(f"two bs=2 graphs, 2 streams {tpar:.3f} ms  ({1e3 / tpar:.1f} steps/s of 4 latents)   max |diff| vs bs=4 / max = {err:.2e}")
