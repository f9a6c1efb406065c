"""K-loop cycle breakdown per wave from the -DAPTP_STAMPS build (timing experiments).
Usage: python3 tools/stamp_conv.py <lib_st.so>"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diffusion_pruning_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from diffusion_pruning_amd import ops
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
cases = [(4, 64, 320, 320, 3, 33, 1), (4, 64, 320, 320, 3, 34, 1), (4, 64, 320, 320, 3, 29, 1), (4, 64, 320, 160, 3, 37, 1), (4, 64, 320, 160, 3, 39, 1), (4, 32, 640, 640, 3, 36, 1)]
for (B, H, Cin, Cout, k, tile, sk) in cases:
    x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    pw = ops.pack_weight(torch.randn(Cout, Cin, k, k) * 0.02, torch.zeros(Cout), device=dev)
    for _ in range(3):
        y = ops.conv_gemm(x, pw, tile=tile, split_k=sk)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv_gemm(x, pw, tile=tile, split_k=sk, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    buf = np.zeros(4096 * 4, dtype=np.uint64)
    rc = raw.aptp_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    st = buf.reshape(4096, 4).astype(np.float64)
    st = st[st[:, 3] > 0]
    nk = (k * k * ((Cin + 63) // 64)) // sk
    m = st.mean(axis=0)
    print(f"{(B,H,Cin,Cout,k)} tile {tile} sk {sk}: waves {len(st)} K-steps {nk}; per K-step cycles: dma-issue {m[0]/nk:7.1f}  lds+mfma {m[1]/nk:7.1f}  wait+barrier {m[2]/nk:7.1f}  | loop total {m[:3].sum()/nk:7.1f}  kernel-body total {m[3]:9.0f}  eager us/launch {us:6.1f}", flush=True)
