"""Weight gradient of the six resampling convolutions: parity-split kernel path vs the GEMM-on-copies fallback (us per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for keep in (1.0, 0.55):
    for kind, H, C in (("stride2", 64, 320), ("stride2", 32, 640), ("stride2", 16, 1280), ("ups", 8, 1280), ("ups", 16, 1280), ("ups", 32, 640)):
        Ck = int(C * keep) // 8 * 8
        x = torch.randn(4, H, H, Ck, device=dev).bfloat16()
        Ho = H // 2 if kind == "stride2" else 2 * H
        dy = torch.randn(4, Ho, Ho, Ck, device=dev).bfloat16()
        args = dict(stride=2, pad=1, ups=0) if kind == "stride2" else dict(stride=1, pad=1, ups=1)
        out = torch.zeros(Ck, 9, Ck, device=dev)
        t_par = timeit(lambda: ops._wgrad_parity(x, dy, args["stride"], args["ups"], out, True))
        ops.WGRAD_KERNEL = False
        t_old = timeit(lambda: ops.conv_wgrad(x, dy, 3, 3, out=out, want_db=True, **args))
        ops.WGRAD_KERNEL = True
        print(f"{kind:8s} H{H:3d} C{Ck:5d}: parity {t_par:8.1f} us   fallback {t_old:8.1f} us", flush=True)
