import os, sys
sys.path.insert(0, ".")
import torch
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, H, Cin, N, order) in [(1, 64, 64, 160, 0), (1, 64, 64, 160, 1), (1, 64, 64, 320, 1), (1, 48, 64, 160, 1)]:
    if (H * H) % 128 or 128 % H and H != 48:
        pass
    x = torch.randn(B, Cin, H, H).bfloat16()
    w = torch.zeros(N, Cin, 3, 3)
    for n in range(min(N, Cin)): w[n, n, 1, 1] = 1.0
    pw = ops.pack_weight(w, None, device=dev)
    out = torch.full((B, H, H, N), float("nan"), dtype=torch.bfloat16, device=dev)
    try:
        y = ops.conv_gemm(x.permute(0, 2, 3, 1).contiguous().to(dev), pw, tile=43, split_k=1, order=order, out=out).float().cpu()
    except Exception as e:
        print((B, H, Cin, N, order), "error", str(e)[:80]); continue
    xs = x.float().permute(0, 2, 3, 1)
    rows = []
    for r in range(H):
        row = y[0, r, :, :min(N, Cin)]
        st = "nan" if torch.isnan(row).any() else ("ok" if float((row - xs[0, r][:, :min(N, Cin)]).abs().max()) < 1e-3 else ("zero" if float(row.abs().max()) == 0 else "wrong"))
        rows.append(st)
    import itertools
    print((B, H, Cin, N, order), [(k, len(list(g))) for k, g in itertools.groupby(rows)])
