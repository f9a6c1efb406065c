"""A/B of the two split-K forms (in-kernel reduction vs reduce launch) on every split shape of the tuning table, at the
tuned (tile, order) and a range of split factors.  Usage: python tools/bench_splitk_modes.py"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
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


ops._tile_counters(dev)
for key, t in sorted(ops.TUNING.items()):
    if t["split_k"] <= 1:
        continue
    m = re.match(r"M(\d+)_N(\d+)_C(\d+)_T(\d+)_s(\d)u(\d)g(\d)", key)
    M, N, C, T, st, up, gg = (int(v) for v in m.groups())
    if gg or st != 1 or up or M % 4:
        continue
    k = 3 if T == 9 else 1
    H = int(round((M // 4) ** 0.5))
    if 4 * H * H != M:
        continue
    x = torch.randn(4, H, H, C, device=dev).bfloat16()
    pw = ops.pack_weight(torch.randn(N, C, k, k) * 0.02, torch.zeros(N), device=dev)
    res = torch.randn(4, H, H, N, device=dev).bfloat16()
    y = ops.conv_gemm(x, pw, residual=res)
    row = []
    for sk in sorted({2, 3, 4, 6, 8, t["split_k"]}):
        ts = []
        for ink in (False, True):
            ops.SPLITK_IN_KERNEL = ink
            ts.append(timeit(lambda: ops.conv_gemm(x, pw, residual=res, out=y, tile=t["tile"], split_k=sk, order=t.get("order", 1))))
        row.append(f"s{sk}: {ts[0]:5.1f}/{ts[1]:5.1f}")
    ops.SPLITK_IN_KERNEL = True
    print(f"{key:30s} tuned s{t['split_k']} t{t['tile']}  reduce-launch/in-kernel us  " + "  ".join(row), flush=True)
