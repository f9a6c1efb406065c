"""Inference throughput of the eight config-5 experts (BASELINE configs[4] masks: keep ratio 0.40 .. 0.75, 0-4 depth gates off;
shapes that are NOT in the tuning table) -- one UNet2DConditionModelPruned forward at bs=4, 64x64 latents, HIP graph replay.
APTP_TUNING_NEAREST=0 gives the library-heuristic baseline for the same experts (same-box A/B)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd.unet import UNet2DConditionModelPruned  # noqa: E402
from tools.bench_finetune import expert_mask  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    experts = [int(a) for a in sys.argv[1:]] or list(range(8))
    model = UNet2DConditionModelPruned().init_synthetic(seed=0).to(dev)
    st = model.get_structure()
    g = torch.Generator().manual_seed(1234)
    sample = torch.randn(4, 4, 64, 64, generator=g).to(dev)
    ehs = torch.randn(4, 77, 1024, generator=g).to(dev)
    t = torch.full((4,), 500, dtype=torch.int64, device=dev)
    res = {}
    with torch.no_grad():
        for e in experts:
            model.invalidate_plans()
            model.prune(expert_mask(st, e, dev))
            out = model(sample, t, ehs).sample
            torch.cuda.synchronize()
            assert torch.isfinite(out).all()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(sample, t, ehs)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                model(sample, t, ehs)
            for _ in range(5):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                graph.replay()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 30 * 1e3
            res[e] = round(ms, 3)
            del graph
    print(json.dumps({"metric": "ms per forward, config-5 experts (bs=4, 64x64 latents, HIP graph replay)",
                      "nearest_tuning": os.environ.get("APTP_TUNING_NEAREST", "1") != "0", "ms": res,
                      "steps_per_s": {k: round(1e3 / v, 1) for k, v in res.items()}}))


if __name__ == "__main__":
    main()
