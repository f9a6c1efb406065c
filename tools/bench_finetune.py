"""BASELINE configs[4]: expert fine-tune step (teacher dense fwd + pruned student fwd/bwd incl. weight gradients + AdamW) at
SD-2.1 size, one expert per GPU (rank r fine-tunes expert r: keep ratio 0.4..0.75, 0-4 depth gates off).
Experts never communicate (scripts/aptp/finetune.py:27-28), so N GPUs = N independent processes; this tool runs ONE expert
(--expert k) on cuda:0 and prints one JSON line; under `python -m torch.distributed.run --nproc-per-node 8` every rank takes
expert = RANK on cuda:LOCAL_RANK and prints its own line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd.train_step import FineTunerStep, synthetic_batch  # noqa: E402
from diffusion_pruning_amd.unet import UNet2DConditionModelGated, UNet2DConditionModelPruned  # noqa: E402


from bench import expert_mask  # noqa: E402,F401  (the eight benchmark experts live with the benchmark)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--expert", type=int, default=0)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=("graphed", "packed-eager", "eager"), default="graphed",
                    help="graphed: packed masters + HIP graphs (GraphedFineTunerStep); packed-eager: packed masters, eager; "
                         "eager: diffusers-layout masters, eager (the round-1 path)")
    args = ap.parse_args()
    # under torchrun: one expert per GPU (rank r -> expert r on cuda:LOCAL_RANK); the processes never communicate
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ:
        args.expert = int(os.environ["RANK"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    teacher = UNet2DConditionModelGated().init_synthetic(seed=0)
    student = UNet2DConditionModelPruned()
    student.load_state_dict(teacher.state_dict())
    teacher.to(dev).freeze()
    st = teacher.get_structure()
    teacher.set_structure({"width": [torch.ones(1, w, device=dev) for sub in st["width"] for w in sub],
                           "depth": [torch.ones(1, device=dev) for sub in st["depth"] for d in sub if d == 1]})
    student.to(dev)
    student.prune(expert_mask(st, args.expert, dev))
    batch = synthetic_batch(args.batch, 64, dev)
    n_train = None
    if args.mode == "graphed":
        from diffusion_pruning_amd.train_step import GraphedFineTunerStep
        step = GraphedFineTunerStep(student, teacher, lr=1e-5)
        step.capture(batch, offload_masters=True)
        opt, n_train = None, step.trainer.n_trainable()
    elif args.mode == "packed-eager":
        from diffusion_pruning_amd.packed_train import PackedTrainer
        step = FineTunerStep(student, teacher)
        pk = PackedTrainer(student).attach().materialize(batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"])
        opt, n_train = torch.optim.AdamW(pk.parameters(), lr=1e-5, fused=True), pk.n_trainable()
    else:
        step = FineTunerStep(student, teacher)
        opt = torch.optim.AdamW([p for p in student.parameters() if p.requires_grad], lr=1e-5, fused=True)
    for _ in range(args.warmup):
        out = step.train_step(opt, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step.train_step(opt, batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": "expert-finetune-steps/s (SD-2.1 pruned expert, 64x64 latents, teacher fwd + student fwd/bwd/wgrad + AdamW)",
                      "value": round(1.0 / dt, 3), "unit": "steps/s", "ms_per_step": round(dt * 1e3, 1), "expert": args.expert,
                      "batch": args.batch, "loss": float(out["loss"].detach()),
                      "max_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "mode": args.mode,
                      "trainable_parameters": n_train}))


if __name__ == "__main__":
    main()
