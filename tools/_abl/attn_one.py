import os, sys
sys.path.insert(0, ".")
import torch
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
B, h, L = 4, 2, 4096
q = torch.randn(B, L, h * 64, device=dev).bfloat16(); k = torch.randn(B, L, h * 64, device=dev).bfloat16(); v = torch.randn(B, L, h * 64, device=dev).bfloat16()
ops.ATTN_VARIANT = 3
for _ in range(5):
    o = ops.attention(q, k, v, h)
torch.cuda.synchronize()
