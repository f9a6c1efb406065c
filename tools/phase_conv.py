"""Launch timeline of conv_gemm waves from the -DAPTP_STAMPS build (timing experiments): cycles from a wave's first
instruction to [prologue DMA issued, first tile landed, K loop done, epilogue stores drained], against the warm launch time.
Usage: python3 tools/phase_conv.py tools/_abl/libaptp_st.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from diffusion_pruning_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from diffusion_pruning_amd import ops
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
GHZ = 2.4
# rows, Cin, Cout, tile, residual
cases = [(4, 320, 1280, 18, False), (16384, 320, 128, 18, False), (4096, 320, 640, 18, True), (1024, 640, 1280, 18, True),
         (1024, 1280, 1280, 25, True), (16384, 320, 320, 15, True), (4096, 640, 640, 18, True)]
for (M, Cin, Cout, tile, res) in cases:
    x = torch.randn(1, 1, M, Cin, device=dev).bfloat16()
    pw = ops.pack_weight(torch.randn(Cout, Cin, 1, 1) * 0.02, torch.zeros(Cout), device=dev)
    r = torch.randn(1, 1, M, Cout, device=dev).bfloat16() if res else None
    y = ops.conv_gemm(x, pw, tile=tile, residual=r)
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                ops.conv_gemm(x, pw, tile=tile, residual=r, out=y)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    buf = np.zeros(4096 * 4, dtype=np.uint64)
    raw.aptp_debug_read_phases(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    nw = min(4096, ((M + 63) // 64) * ((Cout + 63) // 64) * 4)
    ph = buf.reshape(4096, 4)[:nw].astype(np.float64) / (GHZ * 1e3)
    m, mx = ph.mean(0), ph.max(0)
    print(f"M{M} K{Cin} N{Cout} tile {tile} res {int(res)}: launch {us:5.1f} us | wave timeline (mean / max over {nw} waves, us): "
          f"prologue issued {m[0]:.2f}/{mx[0]:.2f}  first tile landed {m[1]:.2f}/{mx[1]:.2f}  loop done {m[2]:.2f}/{mx[2]:.2f}  "
          f"epilogue drained {m[3]:.2f}/{mx[3]:.2f}", flush=True)
    eb = np.zeros(4096 * 8, dtype=np.uint64)
    raw.aptp_debug_read_epi(eb.ctypes.data_as(ctypes.c_void_p), eb.size)
    e = eb.reshape(4096, 8).astype(np.float64)
    e = e[(e[:, 0] > 0) & (e[:, 5] > e[:, 0])]
    d = (e[:, 1:6] - e[:, 0:1]) / (GHZ * 1e3)
    dm = d.mean(0)
    print(f"      epilogue (us since its entry, mean over {len(e)} waves): stages free {dm[0]:.2f}  first fragment in LDS {dm[1]:.2f}  "
          f"[3 unused]  all stores issued {dm[3]:.2f}  drained {dm[4]:.2f}", flush=True)
