"""Launch timeline of conv_gemm waves from the -DAPTP_STAMPS build (timing experiments): cycles from a wave's first
instruction to [tile decoded, prefetch / statistics requested, addresses ready, prologue DMA issued, first tile landed,
K loop done, epilogue stores drained], against the warm launch time.
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
TILES = {9: (64, 128, 4), 12: (64, 64, 4), 18: (64, 64, 4), 49: (64, 64, 4), 25: (64, 64, 4), 56: (128, 160, 8), 59: (64, 64, 8)}
# rows, Cin, Cout, tile, residual, rowstat   (the headline step's 1x1 launches, profiles/r3_launch_table.txt)
cases = [(4, 320, 1280, 18, False, False), (16384, 128, 320, 9, True, True), (16384, 320, 320, 56, False, True),
         (16384, 320, 128, 49, False, False), (4096, 320, 640, 18, True, True), (4096, 640, 640, 18, True, False),
         (4096, 1280, 640, 18, True, False), (1024, 640, 1280, 49, True, True), (1024, 1280, 1280, 49, True, False),
         (1024, 2560, 1280, 49, True, False), (256, 640, 1280, 59, True, True)]
for (M, Cin, Cout, tile, res, rs) in cases:
    bm, bn, nwv = TILES[tile]
    x = torch.randn(1, 1, M, Cin, device=dev).bfloat16()
    pw = ops.pack_weight(torch.randn(Cout, Cin, 1, 1) * 0.02, torch.zeros(Cout), device=dev)
    r = torch.randn(1, 1, M, Cout, device=dev).bfloat16() if res else None
    kw = dict(tile=tile, residual=r)
    if rs:
        kw["rowstats"] = True
    try:
        y = ops.conv_gemm(x, pw, **kw)
    except TypeError:
        kw.pop("rowstats", None)
        y = ops.conv_gemm(x, pw, **kw)
    if isinstance(y, tuple):
        y = y[0]
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                ops.conv_gemm(x, pw, out=y, **kw)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    raw.aptp_debug_read_phases(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    nw = min(4096, ((M + bm - 1) // bm) * ((Cout + bn - 1) // bn) * nwv)
    ph = buf.reshape(4096, 8)[:nw].astype(np.float64) / (GHZ * 1e3)
    ph = ph[ph[:, 3] > 0]
    m = ph.mean(0)
    print(f"M{M} K{Cin} N{Cout} tile {tile} res {int(res)} rowstat {int(rs)}: launch {us:5.1f} us | wave timeline, us since the "
          f"wave's first instruction (mean over {len(ph)} waves): decoded {m[4]:.2f}  pf/stats requested {m[5]:.2f}  addresses {m[6]:.2f}  "
          f"DMA issued {m[0]:.2f}  first tile landed {m[1]:.2f}  loop done {m[2]:.2f}  epilogue drained {m[3]:.2f}", flush=True)
    eb = np.zeros(4096 * 8, dtype=np.uint64)
    raw.aptp_debug_read_epi(eb.ctypes.data_as(ctypes.c_void_p), eb.size)
    e = eb.reshape(4096, 8).astype(np.float64)
    e = e[(e[:, 0] > 0) & (e[:, 5] > e[:, 0])]
    if len(e):
        d = (e[:, 1:6] - e[:, 0:1]) / (GHZ * 1e3)
        dm = d.mean(0)
        print(f"      epilogue (us since its entry, mean over {len(e)} waves): stages free {dm[0]:.2f}  first fragment in LDS {dm[1]:.2f}  "
              f"first fragment stored {dm[2]:.2f}  all stores issued {dm[3]:.2f}  drained {dm[4]:.2f}", flush=True)
