"""GroupNorm variants on the SD-2.1 shapes: parity vs torch fp32 and time per launch (HIP-graph of 20 launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pruning_amd import ops

dev = torch.device("cuda:0")
shapes = [(4, 64, 320), (4, 64, 160), (4, 64, 640), (4, 64, 960), (4, 32, 640), (4, 32, 320), (4, 32, 1280), (4, 32, 1920),
          (4, 32, 960), (4, 16, 1280), (4, 16, 640), (4, 16, 2560), (4, 16, 1920), (4, 8, 1280), (4, 8, 640), (4, 8, 2560),
          (8, 64, 320), (8, 32, 640), (1, 64, 320), (2, 24, 200), (4, 32, 160), (4, 16, 320), (4, 8, 328), (2, 16, 136)]
torch.manual_seed(0)
for (B, H, C) in shapes:
    G = 32 if C % 32 == 0 else (25 if C % 25 == 0 else (41 if C % 41 == 0 else 17))
    x = (torch.randn(B, H, H, C, device=dev) * 2 + 0.5).to(torch.bfloat16)
    ga = torch.randn(C, device=dev); be = torch.randn(C, device=dev)
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x.float().permute(0, 3, 1, 2), G, ga, be, 1e-5)).permute(0, 2, 3, 1)
    line = f"B{B} H{H} C{C:5d}: "
    for v in (1, 2, 0):
        try:
            y = ops.groupnorm(x, ga, be, G, 1e-5, True, variant=v)
        except Exception as e:
            line += f" v{v}: n/a     "
            continue
        err = ((y.float() - ref).norm() / ref.norm()).item()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ops.groupnorm(x, ga, be, G, 1e-5, True, variant=v)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(20):
                    ops.groupnorm(x, ga, be, G, 1e-5, True, variant=v)
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 100
        line += f" v{v}: {us:6.1f}us e={err:.1e}"
    print(line, flush=True)
# determinism: same input twice -> identical bits
x = torch.randn(4, 64, 64, 320, device=dev).to(torch.bfloat16); ga = torch.randn(320, device=dev); be = torch.randn(320, device=dev)
a = ops.groupnorm(x, ga, be, 32, 1e-5, True); b = ops.groupnorm(x, ga, be, 32, 1e-5, True)
print("deterministic:", torch.equal(a, b))
