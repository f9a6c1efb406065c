"""Full denoise loop (CFG doubling + masked U-Net + guidance + scheduler update, one captured step replayed
per timestep, cross-attention K/V computed once per prompt batch) at SD-2.1 size: ms per scheduler step and latents/s, against
the bare U-Net step of bench.py.  usage: python tools/bench_pipeline.py [--pndm] [--prompts 2] [--steps 50]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from diffusion_pruning_amd.pipeline import DDIMSchedulerLite, PNDMSchedulerLite, PruningDenoiseLoop
from diffusion_pruning_amd.unet import UNet2DConditionModelGated

ap = argparse.ArgumentParser()
ap.add_argument("--pndm", action="store_true")
ap.add_argument("--prompts", type=int, default=2, help="prompts per batch (the U-Net sees twice as many samples under CFG)")
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
model.set_structure(bench.fixed_half_mask(model.get_structure(), dev))
g = torch.Generator().manual_seed(3)
loop = PruningDenoiseLoop(model, scheduler=PNDMSchedulerLite() if a.pndm else DDIMSchedulerLite())


def batch():
    return (torch.randn(a.prompts, 77, 1024, generator=g).to(dev), torch.randn(a.prompts, 77, 1024, generator=g).to(dev),
            torch.randn(a.prompts, 4, 64, 64, generator=g).to(dev))


cond, uncond, lat = batch()
loop(cond, lat, a.steps, 7.5, negative_prompt_embeds=uncond)          # captures the step
torch.cuda.synchronize()
times = []
for _ in range(3):                                                    # new prompt batches re-use the captured step
    cond, uncond, lat = batch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = loop(cond, lat, a.steps, 7.5, negative_prompt_embeds=uncond).latents
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
assert torch.isfinite(out).all()
t = sorted(times)[1]
n_calls = loop.scheduler.n_model_calls()
print(f"{'PNDM' if a.pndm else 'DDIM'} {a.steps} steps ({n_calls} U-Net calls), {a.prompts} prompts x CFG (U-Net batch {2 * a.prompts}): "
      f"{t * 1e3:.1f} ms per prompt batch = {t / n_calls * 1e3:.3f} ms per U-Net call, {a.prompts / t:.2f} latents/s "
      f"(runs {[round(x * 1e3, 1) for x in times]})")
