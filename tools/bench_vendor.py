"""Same-node vendor-library yardstick for the headline workload: the oracle's op sequence -- the
reference's diffusers path in GATED semantics: dense compute + mask multiply, as pdm/models/unet/blocks.py does -- on this GPU in
bf16 through torch-ROCm's own libraries (MIOpen convolutions, hipBLASLt linears, SDPA attention, ATen norms), channels_last,
replayed from a HIP graph when it captures.  A stated baseline, not the target; bench.py embeds the JSON as
`gpu_vendor_baseline`.  Usage: python3 tools/bench_vendor.py [--batch 4] [--dense] [--reps 20]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_oracle as O  # noqa: E402  (test infrastructure, used here as the baseline being timed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = O.SD21
    t_start = time.perf_counter()
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    model = UNet2DConditionModelGated().init_synthetic(seed=0)
    params = {}
    for k, v in model.state_dict().items():
        v = v.detach().to(dev, torch.bfloat16)
        params[k] = v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v
    del model
    mask = O.ones_mask(cfg) if args.dense else O.fixed_half_mask(cfg)
    gates = O.assign_gates(cfg, {k: [v.to(dev, torch.bfloat16) for v in vs] for k, vs in mask.items()})
    sample, t, ehs = O.synthetic_inputs(cfg, args.batch, 64)
    sample = sample.to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t, ehs = t.to(dev), ehs.to(dev, torch.bfloat16)

    def fwd():
        return O.unet_forward(params, cfg, sample, t, ehs, gates, "gated")

    how = "eager"
    with torch.no_grad():
        for _ in range(3):
            out = fwd()
        torch.cuda.synchronize()
        t_warm = time.perf_counter() - t_start
        run = fwd
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fwd()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fwd()
            g.replay()
            torch.cuda.synchronize()
            run, how = g.replay, "HIP graph replay"
        except Exception as e:          # a library call that does not capture: time the eager loop
            how = f"eager (graph capture failed: {type(e).__name__})"
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run(); torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
    assert torch.isfinite(out.float()).all()
    print(json.dumps({"value": round(1e3 / ms, 2), "unit": "steps/s", "ms_per_step": round(ms, 3), "kind": "vendor libraries",
                      "how": how, "dtype": "bf16", "semantics": "gated (dense compute + mask multiply: the reference's own form)",
                      "sample": f"{args.reps} steps of the headline workload (bs={args.batch}, {'dense' if args.dense else 'fixed 50 % mask'}) after 3 warm-up "
                                f"steps ({t_warm:.0f} s incl. library kernel selection); oracle op sequence on cuda through torch "
                                f"{torch.__version__}: MIOpen conv2d, hipBLASLt linear, SDPA attention, ATen group_norm / layer_norm"}))


if __name__ == "__main__":
    main()
