"""Mechanism of the capture-order variance: the same pruning step, captured at different moments of the
process, replays 6-10 % faster or slower.  This builds captures A, B (A alive), C (A freed first -- the slow
one in round 3), D; times each; prints where each capture's buffers landed (static inputs, the gate buffer, the segments of the
graphs' private pools: base addresses modulo 4 KiB ... 1 GiB); then replays the FASTEST and the SLOWEST ten times each between
marker fills of distinct sizes, so that a `rocprofv3 --kernel-trace` of this run can be split into the two phases and diffed
per kernel name (`--analyze <kernel_trace.csv>`).
  run:      rocprofv3 --kernel-trace --output-format csv -d /tmp/capvar -- python3 tools/diag_capture_variance.py
  analyze:  python3 tools/diag_capture_variance.py --analyze /tmp/capvar/.../*_kernel_trace.csv"""
import csv
import gc
import os
import re
import sys
from collections import defaultdict

MARK = {"fast": 3 << 20, "slow": 5 << 20, "end": 7 << 20}


def analyze(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    gkey = "Grid_Size" if "Grid_Size" in rows[0] else ("Grid_Size_X" if "Grid_Size_X" in rows[0] else None)
    marks = {}
    for i, r in enumerate(rows):
        if "FillFunctor" in r["Kernel_Name"] and gkey:
            g = int(r[gkey])
            for name, n in MARK.items():
                if abs(g - n // 4) <= 1024 or abs(g - n) <= 1024 or abs(g * 4 - n) <= 4096:
                    marks.setdefault(name, i)
    if len(marks) < 3:
        # fall back: the three LARGEST fills, in time order
        fills = [(int(r[gkey]) if gkey else 0, i) for i, r in enumerate(rows) if "FillFunctor" in r["Kernel_Name"]]
        top = sorted(sorted(fills)[-3:], key=lambda t: t[1])
        marks = {"fast": top[0][1], "slow": top[1][1], "end": top[2][1]}
    print("marker rows:", marks, "of", len(rows))
    seg = {"fast": rows[marks["fast"] + 1:marks["slow"]], "slow": rows[marks["slow"] + 1:marks["end"]]}
    if not seg["fast"] or not seg["slow"]:
        print("could not split the trace into the two phases")
        return

    def short(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"void ", "", n)
        return n.split("(")[0][:64]
    agg = {k: defaultdict(lambda: [0, 0]) for k in seg}
    tot = {}
    for k, rs in seg.items():
        for r in rs:
            a = agg[k][short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        tot[k] = sum(v[1] for v in agg[k].values())
    print(f"kernel time in the two phases (10 replays each): fast {tot['fast'] / 1e6:.2f} ms, slow {tot['slow'] / 1e6:.2f} ms "
          f"({100.0 * (tot['slow'] - tot['fast']) / tot['fast']:+.1f} %)")
    names = sorted(set(agg["fast"]) | set(agg["slow"]), key=lambda n: -(agg["slow"][n][1] - agg["fast"][n][1]))
    print(f"{'kernel':64s} {'n':>5s} {'fast us':>9s} {'slow us':>9s} {'delta':>8s}  share of the gap")
    gap = tot["slow"] - tot["fast"]
    for n in names[:25]:
        f, s = agg["fast"][n], agg["slow"][n]
        d = s[1] - f[1]
        print(f"{n:64s} {f[0]:5d} {f[1] / 1e3:9.1f} {s[1] / 1e3:9.1f} {100.0 * d / max(1, f[1]):+7.1f}%  {100.0 * d / gap if gap else 0:5.1f} %")
    print("...")
    for n in names[-5:]:
        f, s = agg["fast"][n], agg["slow"][n]
        d = s[1] - f[1]
        print(f"{n:64s} {f[0]:5d} {f[1] / 1e3:9.1f} {s[1] / 1e3:9.1f} {100.0 * d / max(1, f[1]):+7.1f}%  {100.0 * d / gap if gap else 0:5.1f} %")


if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    analyze(sys.argv[2])
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_pruning_amd import ops  # noqa: E402
from diffusion_pruning_amd.hypernet import HyperStructure  # noqa: E402
from diffusion_pruning_amd.quantizer import StructureVectorQuantizer  # noqa: E402
from diffusion_pruning_amd.train_step import GraphedPrunerStep, synthetic_batch  # noqa: E402
from diffusion_pruning_amd.unet import UNet2DConditionModelGated  # noqa: E402

dev = torch.device("cuda:0")
ops._lib.load()
unet = UNet2DConditionModelGated().init_synthetic(seed=0).to(dev)
unet.freeze()
st = unet.get_structure()
torch.manual_seed(0)
hn = HyperStructure(structure=st, input_dim=768, wn_flag=False, linear_bias=True).to(dev)
qz = StructureVectorQuantizer(n_e=8, structure=st, temperature=0.4, base=3, resource_aware_normalization=False, optimal_transport=True).to(dev)
batch = synthetic_batch(4, 64, dev, seed=1234)
code = (torch.rand(4, qz.vq_embed_dim, generator=torch.Generator().manual_seed(9)) * 0.6 + 0.4).to(dev)


def placement(step, tag):
    cap = step._cap
    ptrs = {"noisy_latents": cap["st"]["noisy_latents"].data_ptr(), "ehs": cap["st"]["encoder_hidden_states"].data_ptr(),
            "gates": cap["ga"].data_ptr(), "pred": cap["pred"].data_ptr(), "full_pred": cap["full_pred"].data_ptr(), "grad": cap["grad"].data_ptr()}
    acts = sorted(v.data_ptr() for v in cap["acts"].values())
    snap = torch.cuda.memory_snapshot()
    segs = sorted((s["address"], s["total_size"], s.get("segment_pool_id", (0, 0))) for s in snap)
    pools = defaultdict(list)
    for a, sz, pid in segs:
        pools[tuple(pid)].append((a, sz))
    print(f"   [{tag}] static tensors: " + "  ".join(f"{k} 0x{v:x} (mod 2M {v % (2 << 20):>8d}, mod 1G {v % (1 << 30) >> 20:>4d} MiB)" for k, v in ptrs.items()))
    print(f"   [{tag}] block activations (9): " + " ".join(f"0x{a:x}" for a in acts[:9]))
    for pid, ss in sorted(pools.items()):
        if pid == (0, 0):
            continue
        tot = sum(sz for _, sz in ss)
        print(f"   [{tag}] graph pool {pid}: {len(ss)} segments, {tot / 2**20:.0f} MiB, bases " + " ".join(f"0x{a:x}/{sz >> 20}M" for a, sz in ss[:10]) + (" ..." if len(ss) > 10 else ""))


def build(tag):
    step = GraphedPrunerStep(unet, hn, qz)
    step.count_macs(64)
    step.capture(batch)
    step._cap["install_code"](code)
    s = torch.cuda.memory_stats()
    print(f"{tag}: segments {s['segment.all.current']}, reserved {s['reserved_bytes.all.current'] / 2**30:.1f} GiB, allocated "
          f"{s['allocated_bytes.all.current'] / 2**30:.1f} GiB", flush=True)
    placement(step, tag)
    return step


def replay(step):
    cap = step._cap
    step._stage_batch_and_launch_teacher(batch["noisy_latents"], batch["timesteps"], batch["encoder_hidden_states"], batch["target"])
    cap["g_student"].replay()
    torch.cuda.current_stream().wait_stream(cap["side"])
    cap["g_student_bwd"].replay()


def measure(step, n=15):
    for _ in range(3):
        replay(step)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        replay(step)
    e1.record()
    torch.cuda.synchronize()
    return n / (e0.elapsed_time(e1) * 1e-3)


def parts(step, n=10):
    """device time of the three graphs replayed one after the other (no overlap): which of them carries a slow capture's loss"""
    cap = step._cap
    out = []
    for g in (cap["g_teacher"], cap["g_student"], cap["g_student_bwd"]):
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n)
    return out


def copy_gbs(nbytes=1 << 30, reps=10):
    """streaming copy bandwidth of memory allocated NOW (from whatever the driver hands out at this point of the process)"""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    r = 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    pa, pb = a.data_ptr(), b.data_ptr()
    del a, b
    return r, pa, pb


print("allocator config:", os.environ.get("PYTORCH_HIP_ALLOC_CONF") or os.environ.get("PYTORCH_CUDA_ALLOC_CONF") or "(default)")
r0 = copy_gbs()
print(f"1 GiB copy on fresh memory before any capture: {r0[0]:.0f} GB/s (0x{r0[1]:x}, 0x{r0[2]:x})", flush=True)
torch.cuda.empty_cache()
res = {}
A = build("A first capture")
res["A"] = max(measure(A), measure(A))
B = build("B (A alive)")
res["B"] = max(measure(B), measure(B))
pA = parts(A)
del A
gc.collect()
torch.cuda.empty_cache()
r1 = copy_gbs()
print(f"1 GiB copy on memory re-allocated right after A's 12 GiB went back to the driver: {r1[0]:.0f} GB/s (0x{r1[1]:x}, 0x{r1[2]:x})", flush=True)
torch.cuda.empty_cache()
C = build("C (A freed, B alive)")
res["C"] = max(measure(C), measure(C))
junk = [torch.empty(int(1.3 * 2 ** 30), dtype=torch.uint8, device=dev) for _ in range(3)]
D = build("D (after 3.9 GiB of other allocations)")
res["D"] = max(measure(D), measure(D))
steps = {"B": B, "C": C, "D": D}
print("per-graph ms (teacher, student fwd, student bwd), replayed alone: A " + " ".join(f"{v:.2f}" for v in pA) + " | "
      + " | ".join(f"{k} " + " ".join(f"{v:.2f}" for v in parts(st_)) for k, st_ in steps.items()), flush=True)
print("steps/s: " + "  ".join(f"{k} {v:.2f}" for k, v in res.items()), flush=True)
# the same captured graphs (B's) with OTHER side streams for the teacher: is it the stream the runtime handed out?
from diffusion_pruning_amd.graph_utils import concurrent_stream  # noqa: E402
orig = B._cap["side"]
rows = []
for k in range(10):
    s_ = torch.cuda.Stream()
    probe = []
    # overlap probe of THIS stream against the launching stream: two spin kernels, ratio to one
    cyc = int(0.4e-3 * 2.0e9)

    def t_(side):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                torch.cuda._sleep(cyc)
        torch.cuda._sleep(cyc)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    t_(None); one = min(t_(None), t_(None)); t_(s_); two = min(t_(s_), t_(s_))
    B._cap["side"] = s_
    rows.append((k, two / one, max(measure(B), measure(B))))
B._cap["side"] = orig
print("capture B replayed with ten other side streams (spin-kernel overlap ratio: 1.0 = concurrent, 2.0 = serialised; steps/s):", flush=True)
print("   " + "  ".join(f"#{k}: {r:.2f} / {v:.2f}" for k, r, v in rows), flush=True)
chosen = concurrent_stream(log=(plog := []))
C._cap["side"] = chosen
print(f"capture C (the slow one) with a side stream chosen by graph_utils.concurrent_stream {plog}: {max(measure(C), measure(C)):.2f} steps/s (was {res['C']:.2f})", flush=True)
alive = {k: res[k] for k in steps}
fast, slow = max(alive, key=alive.get), min(alive, key=alive.get)
print(f"fastest alive capture {fast} ({alive[fast]:.2f}), slowest {slow} ({alive[slow]:.2f}): {100 * (alive[fast] / alive[slow] - 1):.1f} % apart", flush=True)
torch.cuda.synchronize()
m = torch.empty(MARK["fast"], dtype=torch.float32, device=dev)
m.fill_(1.0)
for _ in range(10):
    replay(steps[fast])
torch.cuda.synchronize()
m2 = torch.empty(MARK["slow"], dtype=torch.float32, device=dev)
m2.fill_(1.0)
for _ in range(10):
    replay(steps[slow])
torch.cuda.synchronize()
m3 = torch.empty(MARK["end"], dtype=torch.float32, device=dev)
m3.fill_(1.0)
torch.cuda.synchronize()
print("done", flush=True)
