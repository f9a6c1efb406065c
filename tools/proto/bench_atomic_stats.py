"""Cost of fixed-point 64-bit atomics at the end of a producer launch (experiment for a GroupNorm-statistics scheme without a
finalize launch).  Usage: python3 tools/proto/bench_atomic_stats.py"""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libatomic_stats.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "atomic_stats.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    fn()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for grid, kb in ((1280, 16), (640, 32), (256, 64), (1280, 4)):
    vec = kb * 1024 // 16
    src = torch.empty(grid * vec * 16, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    for n_addr in (256, 1024):
        stats = torch.zeros(n_addr, dtype=torch.int64, device=dev)
        row = []
        for n_at, mode in ((0, 0), (8, 1), (16, 1), (32, 1), (16, 2)):
            us = timed(lambda: lib.atomic_tail(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), grid, vec,
                                               ctypes.c_void_p(stats.data_ptr()), n_addr, n_at, mode,
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            row.append(f"{'none' if mode == 0 else ('i64 x%d' % n_at if mode == 1 else 'f32 x%d' % n_at)}: {us:6.2f} us")
        print(f"grid {grid:5d} x {kb:3d} KB per WG, {n_addr:5d} addresses | " + "  ".join(row), flush=True)
