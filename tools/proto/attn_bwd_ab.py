"""dq / dk / dv of the library under APTP_LIB written to a file, or compared bit for bit with such a file (debugging aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
out = {}
for (B, h, L, Lk) in [(2, 5, 1024, 1024), (1, 2, 4096, 4096), (2, 3, 200, 77), (2, 4, 300, 320)]:
    g = torch.Generator().manual_seed(L + Lk)
    q, k, v, do = ((torch.randn(B, n, h * 64, generator=g) * a).bfloat16().to(dev) for n, a in ((L, 2.0), (Lk, 2.0), (Lk, 1.0), (L, 1.0)))
    lse = torch.zeros(B, h, L, device=dev)
    o = ops.attention(q, k, v, h, lse=lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attention_bwd(q, k, v, o, do, lse, h, dq, dk, dv)
    out[(B, h, L, Lk)] = [t.cpu() for t in (dq, dk, dv)]
path = sys.argv[1]
if os.path.exists(path):
    ref = torch.load(path)
    for key, ts in out.items():
        print(key, ["equal" if torch.equal(a, b) else f"{int((a != b).sum())} differ (max {float((a.float() - b.float()).abs().max()):.2e})" for a, b in zip(ts, ref[key])])
else:
    torch.save(out, path)
    print("saved", path)
