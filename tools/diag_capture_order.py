"""Does the replay time of a captured training step depend on WHEN in the process it was captured?  (tools/tune_insitu_train.py
saw the first capture replay ~7 % slower than later ones.)  Builds GraphedPrunerStep A, B (A alive), C (A deleted) and times
each one's graph replays, A again at the end; prints steps/s and the allocator's segment statistics at each capture."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd import ops  # noqa: E402
from diffusion_pruning_amd.hypernet import HyperStructure  # noqa: E402
from diffusion_pruning_amd.quantizer import StructureVectorQuantizer  # noqa: E402
from diffusion_pruning_amd.train_step import GraphedPrunerStep, synthetic_batch  # noqa: E402
from diffusion_pruning_amd.unet import UNet2DConditionModelGated  # noqa: E402

dev = torch.device("cuda:0")
ops._lib.load()
unet = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
unet.freeze()
st = unet.get_structure()
torch.manual_seed(0)
hn = HyperStructure(structure=st, input_dim=768, wn_flag=False, linear_bias=True).to(dev)
qz = StructureVectorQuantizer(n_e=8, structure=st, temperature=0.4, base=3, resource_aware_normalization=False, optimal_transport=True).to(dev)
batch = synthetic_batch(4, 64, dev, seed=1234)
code = (torch.rand(4, qz.vq_embed_dim, generator=torch.Generator().manual_seed(9)) * 0.6 + 0.4).to(dev)


def build(log=False):
    step = GraphedPrunerStep(unet, hn, qz)
    step.count_macs(64)
    ops.LAUNCH_LOG = [] if log else None
    try:
        step.capture(batch)
    finally:
        ops.LAUNCH_LOG = None
    step._cap["install_code"](code)
    s = torch.cuda.memory_stats()
    print(f"   segments {s['segment.all.current']}, reserved {s['reserved_bytes.all.current'] / 2**30:.1f} GiB, "
          f"allocated {s['allocated_bytes.all.current'] / 2**30:.1f} GiB", flush=True)
    return step


def measure(step, n=15):
    cap = step._cap

    def replay():
        step._stage_batch_and_launch_teacher(batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"], batch["target"])
        cap["g_student"].replay()
        torch.cuda.current_stream().wait_stream(cap["side"])
        cap["g_student_bwd"].replay()
    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        replay()
    e1.record()
    torch.cuda.synchronize()
    return n / (e0.elapsed_time(e1) * 1e-3)


A = build(log=("--log-first" in sys.argv))
print(f"A (first capture)          {measure(A):.2f}  {measure(A):.2f} steps/s", flush=True)
B = build()
print(f"B (A alive)                {measure(B):.2f}  {measure(B):.2f}", flush=True)
print(f"A again                    {measure(A):.2f}", flush=True)
del A
gc.collect()
C = build()
print(f"C (A deleted, B alive)     {measure(C):.2f}  {measure(C):.2f}", flush=True)
print(f"B again                    {measure(B):.2f}", flush=True)
del B, C
gc.collect()
D = build()
print(f"D (only one alive)         {measure(D):.2f}  {measure(D):.2f}", flush=True)
