"""Host synchronisations inside one graphed pruning train step (torch.cuda.set_sync_debug_mode("warn") prints the call site of
every implicit device->host wait): any of them keeps the host from running ahead of the device, which puts the router's
~600 launch-bound kernels on the step's critical path at host-issue speed."""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd.hypernet import HyperStructure  # noqa: E402
from diffusion_pruning_amd.quantizer import StructureVectorQuantizer  # noqa: E402
from diffusion_pruning_amd.train_step import GraphedPrunerStep, synthetic_batch  # noqa: E402
from diffusion_pruning_amd.unet import UNet2DConditionModelGated  # noqa: E402

dev = torch.device("cuda:0")
if "--finetune" in sys.argv:
    # the graphed expert fine-tune step instead
    from bench import expert_mask  # noqa: E402
    from diffusion_pruning_amd.train_step import GraphedFineTunerStep  # noqa: E402
    from diffusion_pruning_amd.unet import UNet2DConditionModelPruned  # noqa: E402
    teacher = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
    teacher.freeze()
    student = UNet2DConditionModelPruned().init_synthetic(seed=0).to(dev)
    student.prune(expert_mask(student.get_structure(), 3, dev))
    ft = GraphedFineTunerStep(student, teacher, lr=1e-5)
    b = synthetic_batch(4, 64, dev, seed=1234)
    ft.capture(b, offload_masters=True)
    for _ in range(3):
        ft.train_step(None, b)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ft.train_step(None, b)
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    print(f"fine-tune step: {len(w)} synchronising calls")
    for x in w:
        print(f"  {x.filename}:{x.lineno}  {str(x.message)[:100]}")
    sys.exit(0)
unet = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
unet.freeze()
st = unet.get_structure()
torch.manual_seed(0)
hn = HyperStructure(structure=st, input_dim=768, wn_flag=False, linear_bias=True).to(dev)
qz = StructureVectorQuantizer(n_e=8, structure=st, temperature=0.4, base=3, resource_aware_normalization=False, optimal_transport=True,
                              fused_sinkhorn_allreduce=True).to(dev)
hn.train(); qz.train()
step = GraphedPrunerStep(unet, hn, qz)
step.count_macs(64)
batch = synthetic_batch(4, 64, dev, seed=1234)
step.capture(batch)
opt = torch.optim.AdamW(step.trainable_parameters(), lr=2e-4)
for _ in range(3):
    step.train_step(opt, batch)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    import traceback
    orig = warnings.showwarning
    step.train_step(opt, batch)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print(f"{len(w)} synchronising calls in one step")
for x in w:
    print(f"  {x.filename}:{x.lineno}  {str(x.message)[:100]}")
