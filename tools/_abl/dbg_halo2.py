import os, sys
sys.path.insert(0, ".")
import torch, torch.nn.functional as F
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, H, Cin, N) in [(1, 64, 64, 160), (2, 64, 64, 160), (1, 32, 64, 160), (4, 16, 64, 160), (1, 64, 128, 160)]:
    x = torch.randn(B, Cin, H, H).bfloat16()
    w = torch.zeros(N, Cin, 3, 3)
    for n in range(min(N, Cin)): w[n, n, 1, 1] = 1.0
    pw = ops.pack_weight(w, None, device=dev)
    y = ops.conv_gemm(x.permute(0, 2, 3, 1).contiguous().to(dev), pw, tile=43, split_k=1).float().cpu()
    xs = x.float().permute(0, 2, 3, 1)
    okrows = [(b, r) for b in range(B) for r in range(H) if float((y[b, r, :, :min(N,Cin)] - xs[b, r][:, :min(N,Cin)]).abs().max()) < 1e-3]
    print((B, H, Cin, N), "correct rows:", len(okrows), "of", B * H, "first bad:", next(((b, r) for b in range(B) for r in range(H) if (b, r) not in okrows), None))
