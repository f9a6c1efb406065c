"""Which torch elementwise / copy kernels with >= 1 M elements does one eager pruning step (student forward + backward) issue,
and from where?  (They are what is left of 'torch glue' inside the captured student graphs.)  usage: python tools/diag_torch_glue.py"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from diffusion_pruning_amd.hypernet import HyperStructure
from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
from diffusion_pruning_amd.train_step import PrunerStep, synthetic_batch
from diffusion_pruning_amd.unet import UNet2DConditionModelGated

FT = "--ft" in sys.argv          # the packed-masters fine-tune step instead of the pruning step
THRESH = (1 << 18) if FT else (1 << 20)
COPIES = "--copies" in sys.argv  # only copy-like ops, from 16 K elements up
if COPIES:
    THRESH = 1 << 14
ALL = "--all" in sys.argv        # every torch op that launches something, ranked by COUNT (the launch diet's view)
GRAPHED = "--graphed" in sys.argv  # the ops torch records INTO the three captured U-Net graphs of GraphedPrunerStep (by op, site, shape)
if ALL or GRAPHED:
    THRESH = 1
dev = torch.device("cuda:0")
unet = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
unet.freeze()
st = unet.get_structure()
torch.manual_seed(0)
hn = HyperStructure(structure=st, input_dim=768, wn_flag=False, linear_bias=True).to(dev)
qz = StructureVectorQuantizer(n_e=8, structure=st, temperature=0.4, base=3, resource_aware_normalization=False,
                              optimal_transport=True).to(dev)
hn.train(); qz.train()
b = synthetic_batch(4, 64, dev)
if FT:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_finetune import expert_mask
    from diffusion_pruning_amd.packed_train import PackedTrainer
    from diffusion_pruning_amd.train_step import FineTunerStep
    from diffusion_pruning_amd.unet import UNet2DConditionModelPruned
    student = UNet2DConditionModelPruned()
    student.load_state_dict(unet.state_dict())
    unet.set_structure({"width": [torch.ones(1, w, device=dev) for sub in st["width"] for w in sub],
                        "depth": [torch.ones(1, device=dev) for sub in st["depth"] for d in sub if d == 1]})
    student.to(dev)
    student.prune(expert_mask(st, 3, dev))
    ft = FineTunerStep(student, unet)
    pk = PackedTrainer(student).attach().materialize(b["noisy_latents"], b["timesteps"], b["encoder_hidden_states"])
    opt = torch.optim.AdamW(pk.parameters(), lr=1e-5, fused=True)
else:
    step = PrunerStep(unet, hn, qz)
    step.count_macs(64)


class Big(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()
        self.bytes = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        t = out if isinstance(out, torch.Tensor) else None
        name = str(func)
        if "empty" in name or "_foreach" in name or "fused_adam" in name:
            return out
        if COPIES and not any(k in name for k in ("clone", "copy", "contiguous", "cat", "pad", "stack", "index", "zeros", "fill")):
            return out
        if GRAPHED and not torch.cuda.is_current_stream_capturing():
            return out
        if t is not None and t.is_cuda and t.numel() >= THRESH and not any(s in name for s in ("view", "permute", "slice", "detach",
                                                                                              "t.default", "alias", "expand", "select", "unsqueeze", "squeeze", "as_strided", "reshape", "transpose", "narrow", "split", "unbind")):
            fr = [f for f in traceback.extract_stack() if "diffusion_pruning_amd" in f.filename]
            where = tuple((f.filename.split("/")[-1], f.lineno) for f in fr[-2:]) if fr else ("autograd engine", 0)
            if GRAPHED:
                where = (where, tuple(t.shape))
            self.c[(name, where)] += 1
            self.bytes[(name, where)] += t.numel() * t.element_size()
        return out


def run():
    if FT:
        ft.train_step(opt, b)
        return
    out = step.step(b["noisy_latents"], b["timesteps"], b["encoder_hidden_states"], b["mpnet_embeddings"], b["target"])
    out["loss"].backward()


if GRAPHED:
    from diffusion_pruning_amd.train_step import GraphedPrunerStep
    gstep = GraphedPrunerStep(unet, hn, qz)
    gstep.count_macs(64)
    with Big() as big:
        gstep.capture(b)
else:
    run()
    torch.cuda.synchronize()
    with Big() as big:
        run()
torch.cuda.synchronize()
tot = 0
for k, n in sorted(big.c.items(), key=lambda kv: (-kv[1] if (ALL or GRAPHED) else -big.bytes[kv[0]]))[:(120 if GRAPHED else 70 if ALL else 40)]:
    print(f"{n:4d} x {big.bytes[k] / n / 1e6:7.1f} MB  {k[0]:36s} {k[1]}")
    tot += big.bytes[k]
print("total output bytes of big torch ops per step: %.1f MB; %d torch ops counted" % (tot / 1e6, sum(big.c.values())))
