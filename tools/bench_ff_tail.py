"""aptp_ff_tail vs the three launches it replaces (LN-folded GEGLU projection, ff.net[2] + residual, proj_out + residual +
column statistics) at SD-2.1 level 64 (bs=4: M = 16384, C = 320; hidden 640 = the 50 % mask, 1280 = dense).  HIP-graph replays."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd import ops
from tools.bench_small_gemm import timed


def main():
    dev = torch.device("cuda:0")
    ops.FUSE_TAIL_MIN_ROWS = 64
    for B, L, C, inner in ((4, 4096, 320, 640), (4, 4096, 320, 1280), (2, 4096, 320, 640)):
        g = torch.Generator().manual_seed(1)
        h = torch.randn(B, L, C, generator=g).bfloat16().to(dev)
        x = torch.randn(B, L, C, generator=g).bfloat16().to(dev)
        ln_g, ln_b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
        w1 = torch.randn(2 * inner, C, generator=g) / C ** 0.5; b1 = torch.randn(2 * inner, generator=g) * 0.1
        w2 = torch.randn(C, inner, generator=g) / inner ** 0.5; b2 = torch.randn(C, generator=g) * 0.1
        w3 = torch.randn(C, C, generator=g) / C ** 0.5; b3 = torch.randn(C, generator=g) * 0.1
        pw1 = ops.pack_weight(w1, b1, geglu=True, device=dev, ln_gamma=ln_g, ln_beta=ln_b)
        pw2 = ops.pack_weight(w2, b2, cin_pad_to=16, device=dev)
        pw3 = ops.pack_weight(w3, b3, device=dev)
        ident = ops.pack_weight(torch.eye(C), None, device=dev)
        _, st = ops.linear(h, ident, rowstats=True)            # row statistics as the producer of h would emit them
        out = torch.empty(B, L, C, dtype=torch.bfloat16, device=dev)

        def chain():
            f = ops.linear(h, pw1, ln=(st, 1e-5))
            h3 = ops.linear(f, pw2, residual=h)
            return ops.linear(h3, pw3, residual=x, out=out, colstats=True)
        t_chain = timed(chain)
        t_fused = timed(lambda: ops.ff_tail(h, x, pw1, pw2, pw3, 1e-5, out=out, colstats=True))
        fl = 2.0 * B * L * (C * 2 * inner + inner * C + C * C)
        print(f"B{B} L{L} C{C} inner{inner}: three launches {t_chain:6.1f} us ({fl/t_chain/1e6:6.1f} TF)  fused {t_fused:6.1f} us ({fl/t_fused/1e6:6.1f} TF)", flush=True)


if __name__ == "__main__":
    main()
