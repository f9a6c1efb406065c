"""Prototype measurement (see proj_stream.hip): weights-resident projection kernel against the tiled kernel of the product on the
level-64 short-K shapes of the masked step.  Builds tools/proto/proj_stream.hip with hipcc on the GPU box, checks the result
against an fp32 reference, and times both in replayed graphs of 40 launches."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from diffusion_pruning_amd import ops  # noqa: E402

so = os.path.join(ROOT, "gpurun_out", "libproj_stream.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-o", so,
                       os.path.join(HERE, "proj_stream.hip")])
lib = ctypes.CDLL(so)
lib.proj_stream.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
lib.proj_stream.restype = ctypes.c_int
dev = torch.device("cuda:0")
ops._lib.load()


def graph_time(fn, n=40, reps=30):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / n * 1e3


gen = torch.Generator().manual_seed(0)
for (M, K, N, with_res) in [(16384, 128, 320, True), (16384, 128, 320, False), (16384, 128, 128, True), (16384, 64, 192, True),
                            (4096, 128, 320, True)]:
    x = torch.randn(M, K, generator=gen).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5)
    b = torch.randn(N, generator=gen).to(dev)
    res = torch.randn(M, N, generator=gen).to(dev, torch.bfloat16) if with_res else None
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    wd = w.to(dev, torch.bfloat16).contiguous()

    def proto():
        rc = lib.proj_stream(x.data_ptr(), wd.data_ptr(), b.data_ptr(), res.data_ptr() if res is not None else None, y.data_ptr(),
                             M, K, N, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    proto()
    torch.cuda.synchronize()
    ref = x.float() @ wd.float().t() + b + (res.float() if res is not None else 0.0)
    err = float((y.float() - ref).norm() / ref.norm())
    pw = ops.pack_weight(w[:, :, None, None].contiguous(), b.cpu(), device=dev)
    x4 = x.view(4, M // 4 // 64, 64, K) if M % 256 == 0 else x.view(1, M, 1, K)
    r4 = None if res is None else res.view(x4.shape[0], x4.shape[1], x4.shape[2], N)
    out = ops.conv_gemm(x4, pw, pad=0, residual=r4)
    err2 = float((out.float().reshape(M, -1)[:, :N] - ref).norm() / ref.norm())
    t_proto = graph_time(proto)
    t_tiled = graph_time(lambda: ops.conv_gemm(x4, pw, pad=0, residual=r4))
    mb = (M * K + (2 if with_res else 1) * M * N) * 2 / 1e6
    print(f"M {M:6d} K {K:4d} N {N:4d} res {int(with_res)}: resident-W {t_proto:6.2f} us (rel err {err:.1e})   tiled {t_tiled:6.2f} us "
          f"(rel err {err2:.1e})   {mb:.1f} MB -> streaming bound {mb / 5.6e3 * 1e3:.1f} us + 1.6 us launch", flush=True)
