"""cProfile of the eager router on the GPU box (host time is what bounds it).  usage: python tools/diag_router_host.py"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from diffusion_pruning_amd.hypernet import HyperStructure
from diffusion_pruning_amd.quantizer import StructureVectorQuantizer
from diffusion_pruning_amd.unet import UNet2DConditionModelGated

dev = torch.device("cuda:0")
st = UNet2DConditionModelGated().get_structure()
torch.manual_seed(0)
hn = HyperStructure(structure=st, input_dim=768, wn_flag=False, linear_bias=True).to(dev)
qz = StructureVectorQuantizer(n_e=8, structure=st, temperature=0.4, base=3,
                              depth_order=[-1, -2, 0, 1, -3, -4, 2, 3, -5, -6, 4, 5, -7, 6],
                              resource_aware_normalization=False, optimal_transport=True, fused_sinkhorn_allreduce=True).to(dev)
hn.train(); qz.train()
te = (0.05 * torch.randn(4, 768)).to(dev)


def fwd():
    av = hn(te)
    avq, _ = qz(av)
    return qz.gumbel_sigmoid_trick(av), avq


for _ in range(5):
    fwd()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    fwd()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
