"""BASELINE configs[2]: APTP pruning train step (teacher fwd + student fwd/bwd + router) at SD-2.1 size on one MI355X,
synthetic CC3M-shape batch.  Prints one JSON line (secondary metric; bench.py stays on the headline inference config)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd.hypernet import HyperStructure  # noqa: E402
from diffusion_pruning_amd.quantizer import StructureVectorQuantizer  # noqa: E402
from diffusion_pruning_amd.train_step import GraphedPrunerStep, PrunerStep, synthetic_batch  # noqa: E402
from diffusion_pruning_amd.unet import UNet2DConditionModelGated  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--eager", action="store_true", help="no HIP graphs (the host-bound reference point)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    unet = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
    unet.freeze()
    st = unet.get_structure()
    torch.manual_seed(0)
    hn = HyperStructure(structure=st, input_dim=768, wn_flag=False, linear_bias=True).to(dev)
    qz = StructureVectorQuantizer(n_e=8, structure=st, temperature=0.4, base=3,
                                  depth_order=[-1, -2, 0, 1, -3, -4, 2, 3, -5, -6, 4, 5, -7, 6],
                                  resource_aware_normalization=False, optimal_transport=True).to(dev)
    hn.train(); qz.train()
    step = (PrunerStep if args.eager else GraphedPrunerStep)(unet, hn, qz)
    step.count_macs(args.latent)
    opt = torch.optim.AdamW(step.trainable_parameters(), lr=2e-4)
    batch = synthetic_batch(args.batch, args.latent, dev)
    if not args.eager:
        step.capture(batch)
    for _ in range(args.warmup):
        out = step.train_step(opt, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step.train_step(opt, batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": "pruning-train-steps/s (SD-2.1, 64x64 latents, teacher fwd + student fwd/bwd + router)",
                      "value": round(1.0 / dt, 3), "unit": "steps/s", "ms_per_step": round(dt * 1e3, 2),
                      "batch": args.batch, "loss": float(out["loss"].detach()),
                      "resource_ratio": float(out["resource_ratio"]),
                      "max_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "mode": "eager (no HIP graph)" if args.eager else "U-Net passes replayed from HIP graphs, router eager"}))


if __name__ == "__main__":
    main()
