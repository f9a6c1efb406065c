import os, sys
sys.path.insert(0, ".")
import torch
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, H, Cin, N, k, tile, sk) in [(4, 64, 160, 320, 3, 0, None), (1, 64, 320, 320, 3, 0, None), (1, 64, 320, 320, 1, 0, None), (4, 64, 320, 160, 3, 0, None),
                                     (1, 64, 320, 320, 3, 4, 1), (1, 64, 320, 320, 3, 2, 1), (1, 64, 320, 320, 3, 10, 1), (1, 64, 320, 320, 3, 34, 1), (1, 64, 320, 320, 3, 3, 1)]:
    x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    pw = ops.pack_weight(torch.randn(N, Cin, k, k) * 0.05, torch.zeros(N), device=dev)
    y = ops.conv_gemm(x, pw, tile=tile, split_k=sk, colstats=True)
    segs = ops._colstats_get(y, N)
    if segs is None:
        print((B, H, Cin, N, k, tile, sk), "no stats"); continue
    st, rpb, C = segs[0]
    M = B * H * H
    yf = y.float().reshape(M // rpb, rpb, N)
    ref = torch.stack([yf.sum(1), (yf * yf).sum(1)], -1)
    got = st[:M // rpb]
    d = (got - ref).abs()
    badblk = torch.nonzero(d.amax((1, 2)) > 1e-2 * ref.abs().max()).flatten()
    badcol = torch.nonzero(d.amax((0, 2)) > 1e-2 * ref.abs().max()).flatten()
    print((B, H, Cin, N, k, tile, sk), "rpb", rpb, "max err", float(d.max()), "scale", float(ref.abs().max()), "bad blocks", len(badblk), badblk[:8].tolist(), "bad cols", len(badcol), badcol[:10].tolist())
B,H,Cin,N,k,tile,sk = 1,64,320,320,3,4,1
x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
pw = ops.pack_weight(torch.randn(N, Cin, k, k) * 0.05, torch.zeros(N), device=dev)
y = ops.conv_gemm(x, pw, tile=tile, split_k=sk, colstats=True)
st, rpb, C = ops._colstats_get(y, N)[0]
yf = y.float().reshape(-1, rpb, N)
print("got", st[0, 56:68, 0].tolist()); print("ref", yf[0].sum(0)[56:68].tolist())
print("got blk1", st[1, 56:68, 0].tolist()); print("ref blk1", yf[1].sum(0)[56:68].tolist())
# partial sums over rows 0..29 only?
print("ref rows<30", yf[0][:30].sum(0)[56:68].tolist())
print("got2", st[0, 56:68, 1].tolist()); print("ref2", (yf[0]*yf[0]).sum(0)[56:68].tolist())
print("got2 136:146", st[0, 136:146, 1].tolist()); print("ref2", (yf[0]*yf[0]).sum(0)[136:146].tolist())
