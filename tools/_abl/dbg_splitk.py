import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
import torch
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B,H,Cin,Cout,k,tile,sk) in [(4,16,2560,1280,1,30,3),(2,8,256,136,3,18,5),(4,16,640,1280,3,34,4)]:
    x = torch.randn(B,H,H,Cin,device=dev).bfloat16()
    pw = ops.pack_weight(torch.randn(Cout,Cin,k,k)*0.02, torch.zeros(Cout), device=dev)
    ops.SPLITK_IN_KERNEL=False
    ref = ops.conv_gemm(x,pw,tile=tile,split_k=sk).clone()
    ops.SPLITK_IN_KERNEL=True
    for it in range(3):
        y = ops.conv_gemm(x,pw,tile=tile,split_k=sk)
        torch.cuda.synchronize()
        d = (y.float()-ref.float()).abs().reshape(-1, y.shape[-1])
        nbad = int((d>0).sum())
        rows = torch.nonzero(d.max(1).values>0).flatten()
        cols = torch.nonzero(d.max(0).values>0).flatten()
        print(tile, sk, it, 'nbad', nbad, 'max', float(d.max()), 'rows', rows[:6].tolist(), len(rows), 'cols', cols[:6].tolist(), len(cols), 'cnt', int(ops._tile_counters(dev).abs().sum()))
