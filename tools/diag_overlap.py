"""Evidence run behind csrc/loss_ops.hip: the block-distillation terms computed inside the pruning step's captured
loss+backward graph by F.mse_loss (torch's multi-block reduction) and by row-wise two-stage sums of the same operands,
replayed back-to-back with the teacher graph beside the student forward (even phases) or serialised (odd phases).
On MI355X / ROCm 7.2 / torch 2.10 the F.mse_loss values of some blocks come out wrong in replay; the row-wise sums never do
(profiles/r2_graph_mse_reduction_evidence.txt).  usage: python tools/diag_overlap.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F

from diffusion_pruning_amd.hypernet import HyperStructure
from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
from diffusion_pruning_amd.train_step import GraphedPrunerStep, synthetic_batch
from diffusion_pruning_amd.unet import UNet2DConditionModelGated

cuda = torch.device("cuda:0")
unet = UNet2DConditionModelGated().init_synthetic(seed=0).to(cuda)
unet.freeze()
st = unet.get_structure()
torch.manual_seed(0)
hn = HyperStructure(structure=st, input_dim=768, wn_flag=False, linear_bias=True).to(cuda)
qz = StructureVectorQuantizer(n_e=8, structure=st, temperature=0.4, base=3, resource_aware_normalization=False,
                              optimal_transport=True).to(cuda)
step = GraphedPrunerStep(unet, hn, qz)
step.count_macs(32)
batch = synthetic_batch(2, 32, cuda, seed=5)
DBG = {}


def losses_both_ways(model_pred, student_acts, full_pred, teacher_acts, w, target):
    loss = F.mse_loss(model_pred.float(), target.float(), reduction="none")
    loss = (loss.mean(dim=list(range(1, loss.dim()))) * w).mean()
    dist = F.mse_loss(model_pred.float(), full_pred.float(), reduction="mean")
    blk = torch.zeros((), device=model_pred.device)
    terms, alt = [], []
    for k in student_acts:
        a, b = student_acts[k].float(), teacher_acts[k].detach().float()
        m = F.mse_loss(a, b, reduction="mean")
        terms.append(m.detach())
        d = (a.detach() - b).reshape(-1)
        alt.append((d * d).view(1024, -1).sum(1).sum() / d.numel())
        blk = blk + m
    DBG["terms"], DBG["alt"] = terms, alt
    return loss, dist, blk / len(student_acts)


step._unet_losses = losses_both_ways
step.capture(batch)
cap = step._cap
g = torch.Generator().manual_seed(9)
cap["install_code"]((torch.rand(batch["noisy_latents"].shape[0], step.quantizer.vq_embed_dim, generator=g) * 0.6 + 0.4).to(cuda))


def replay(overlap):
    step._stage_batch_and_launch_teacher(batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"], batch["target"])
    if not overlap:
        torch.cuda.current_stream().wait_stream(cap["side"])
    cap["g_student"].replay()
    torch.cuda.current_stream().wait_stream(cap["side"])
    cap["g_student_bwd"].replay()
    return cap["blk"] + 0


for phase in range(6):
    torch.cuda.synchronize()
    rs = [replay(phase % 2 == 0) for _ in range(12)]
    torch.cuda.synchronize()
    print("phase", phase, sorted(set(round(float(r), 4) for r in rs)),
          ([round(float(t), 4) for t in DBG["terms"]], [round(float(t), 4) for t in DBG["alt"]]))
