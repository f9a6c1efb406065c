import os, sys
sys.path.insert(0, ".")
import torch, torch.nn.functional as F
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, Cin, N = 1, 64, 64, 160
x = torch.randn(B, Cin, H, H).bfloat16()
for tap in range(9):
    w = torch.zeros(N, Cin, 3, 3)
    for n in range(min(N, Cin)):
        w[n, n, tap // 3, tap % 3] = 1.0
    pw = ops.pack_weight(w, None, device=dev)
    y = ops.conv_gemm(x.permute(0, 2, 3, 1).contiguous().to(dev), pw, tile=43, split_k=1).float().cpu().permute(0, 3, 1, 2)
    ref = F.conv2d(x.float(), w, None, padding=1)
    d = (y - ref).abs()
    bad = torch.nonzero(d.amax(1)[0] > 1e-3)
    print("tap", tap, "max err", float(d.max()), "bad pixels", len(bad), bad[:4].tolist(), "bad ch", torch.nonzero(d.amax((0, 2, 3)) > 1e-3).flatten()[:6].tolist())
w = torch.zeros(N, Cin, 3, 3)
for n in range(Cin): w[n, n, 1, 1] = 1.0
pw = ops.pack_weight(w, None, device=dev)
y = ops.conv_gemm(x.permute(0, 2, 3, 1).contiguous().to(dev), pw, tile=43, split_k=1).float().cpu()   # [1,H,W,N]
xs = x.float().permute(0, 2, 3, 1)
for r in (31, 32, 33, 40, 63):
    row = y[0, r, :, :Cin]
    best = None
    for rr in range(64):
        e = float((row - xs[0, rr]).abs().max())
        if best is None or e < best[0]: best = (e, rr)
    print("out row", r, "nonzero", float(row.abs().max()), "closest input row", best)
