"""Where does a key tile of the software-pipelined attention kernel go?  Times the level-64 self-attention shapes with the ablated
libraries of tools/build_attn_ablations.sh (one subprocess per library: APTP_LIB).  Usage: python3 tools/bench_attn_ablate.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from diffusion_pruning_amd import ops

    def timeit(fn, reps=10):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                for _ in range(reps):
                    fn()
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (5 * reps)
else:
    names = {0: "full kernel", 1: "no exp2", 2: "no loop barrier", 4: "no LDS staging stores", 8: "no MFMAs", 6: "no barrier, no staging", 16: "fragments from registers (no LDS reads)", 20: "no LDS reads, no staging stores", 64: "stamps"}
    for n, what in names.items():
        env = dict(os.environ)
        if n:
            env["APTP_LIB"] = os.path.join(ROOT, "tools", "_abl", f"libaptp_attn{n}.so")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        print(f"{what:45s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
    sys.exit(0)

dev = torch.device("cuda:0")
out = []
for (B, h, L) in [(4, 2, 4096), (4, 5, 4096)]:
    q = torch.randn(B, L, h * 64, device=dev).bfloat16()
    k = torch.randn(B, L, h * 64, device=dev).bfloat16()
    v = torch.randn(B, L, h * 64, device=dev).bfloat16()
    o = ops.attention(q, k, v, h)
    ops.ATTN_VARIANT = 6
    t = timeit(lambda: ops.attention(q, k, v, h, out=o))
    out.append(f"h{h}: {t:6.1f} us")
    if "attn64" in os.environ.get("APTP_LIB", ""):
        lse = torch.zeros(B, h, L, device=dev)
        ops.attention(q, k, v, h, out=o, lse=lse)
        torch.cuda.synchronize()
        st = lse.flatten()[:32].reshape(4, 8)[:, :6].cpu()
        out.append("cycles per tile, waves 0-3 of block 0 [K reads+max | alpha | P0 | P1-3 | staging+QK | barrier]: " + "  ".join(str([int(x) for x in r]) for r in st))
print("   ".join(out))
