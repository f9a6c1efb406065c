"""Micro-benchmark of the folded-LayerNorm epilogue: the GEGLU / QKV consumers of the headline config with and without
ln=, and their producers with and without rowstats.  Usage: python tools/bench_lnfold.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])     # an ablated build (tools/build_ablations.sh)
from diffusion_pruning_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


# rows, C (LN width), producer K, consumer N, geglu
for (M, C, Kp, N, geglu) in [(16384, 320, 128, 1280, True), (16384, 320, 128, 384, False), (4096, 640, 320, 2560, True),
                             (4096, 640, 320, 960, False), (1024, 1280, 640, 5120, True), (1024, 1280, 640, 1920, False)]:
    src = torch.randn(1, M, Kp, device=dev).bfloat16()
    res = torch.randn(1, M, C, device=dev).bfloat16()
    pwp = ops.pack_weight(torch.randn(C, Kp) * 0.05, torch.zeros(C), device=dev)
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    w, b = torch.randn(N, C) * 0.05, torch.zeros(N)
    pw0 = ops.pack_weight(w, b, geglu=geglu, device=dev)
    pw1 = ops.pack_weight(w, b, geglu=geglu, device=dev, ln_gamma=gamma, ln_beta=beta)
    x, st = ops.linear(src, pwp, residual=res, rowstats=True)
    y = ops.linear(x, pw0)
    t_p0 = timeit(lambda: ops.linear(src, pwp, residual=res, out=x))
    t_p1 = timeit(lambda: ops.linear(src, pwp, residual=res, out=x, rowstats=True))
    t_c0 = timeit(lambda: ops.linear(x, pw0, out=y))
    t_c1 = timeit(lambda: ops.linear(x, pw1, out=y, ln=(st, 1e-5)))
    st1 = st[:1].contiguous()
    t_c2 = timeit(lambda: ops.linear(x, pw1, out=y, ln=(st1, 1e-5)))
    print(f"M{M} C{C} N{N} geglu={int(geglu)} slots={0 if st is None else st.shape[0]}: producer {t_p0:6.1f} -> {t_p1:6.1f} us   "
          f"consumer {t_c0:6.1f} -> {t_c1:6.1f} us (1 slot: {t_c2:6.1f})", flush=True)
