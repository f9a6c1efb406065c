import os, sys
sys.path.insert(0, ".")
import torch
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, Cin, N = 1, 64, 64, 160
x = torch.randn(B, Cin, H, H).bfloat16()
w = torch.zeros(N, Cin, 3, 3)
for n in range(Cin): w[n, n, 1, 1] = 1.0
pw = ops.pack_weight(w, None, device=dev)
xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
for tile in (43, 44, 43):
    out = torch.full((B, H, H, N), float("nan"), dtype=torch.bfloat16, device=dev)
    torch.cuda.synchronize()
    ops.conv_gemm(xd, pw, tile=tile, split_k=1, order=1, out=out)
    torch.cuda.synchronize()
    y = out.float().cpu()
    written = [int(not torch.isnan(y[0, 2 * t:2 * t + 2]).any()) for t in range(32)]
    print("tile", tile, "tiles written:", "".join(map(str, written)))
