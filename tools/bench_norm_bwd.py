"""GroupNorm / gate / depth-lerp forward vs backward at SD-2.1's level shapes (bs=4): us per call and effective GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


for (H, C) in [(64, 320), (64, 640), (32, 640), (32, 1280), (16, 1280), (16, 2560), (8, 1280)]:
    x = torch.randn(4, H, H, C, device=dev).bfloat16()
    dy = torch.randn(4, H, H, C, device=dev).bfloat16()
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mb = x.numel() * 2 / 1e6
    t_f = timeit(lambda: ops.groupnorm(x, gamma, beta, 32, 1e-5, True))
    t_fs = timeit(lambda: ops.groupnorm(x, gamma, beta, 32, 1e-5, True, keep_stats=True))
    _, stats = ops.groupnorm(x, gamma, beta, 32, 1e-5, True, keep_stats=True)
    t_b = timeit(lambda: ops.groupnorm_bwd(x, dy, gamma, beta, 32, 1e-5, True, stats))      # (no autograd inside the captured region)
    line = (f"H{H} C{C} ({mb:5.1f} MB): GN fwd {t_f:6.1f} us / {t_fs:6.1f} keeping stats ({3 * mb / t_fs * 1e-3:5.2f} TB/s at 3 passes)"
            f"   GN bwd {t_b:6.1f} us ({5 * mb / t_b * 1e-3:5.2f} TB/s at 5 passes)")
    gate = torch.rand(4, C // 64 if C >= 64 else 1, device=dev)
    try:
        t_g = timeit(lambda: ops.gate_bwd(dy, x, gate))
        t_gf = timeit(lambda: ops.gate_bwd(dy, x, gate, want_dgate=False))
        line += f"   gate bwd {t_g:6.1f} us ({3 * mb / t_g * 1e-3:5.2f} TB/s)  gate fwd-form {t_gf:6.1f} us"
    except Exception as e:                       # noqa: BLE001
        line += f"   (gate: {type(e).__name__} {e})"
    print(line, flush=True)
