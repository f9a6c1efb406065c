"""Per-forward kernel breakdown from a rocprofv3 --kernel-trace CSV: finds the periodic tail of the trace (the timed
HIP-graph replays), and prints count / total / avg per kernel name for ONE period, plus inter-kernel gap time."""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    m = re.match(r"at::native::(\w+)<.*?(\w+_kernel_cuda|FillFunctor|CatArray\w+|silu_kernel|\w+Functor)", n)
    if n.startswith("at::native"):
        return "torch:" + (m.group(1) + ":" + m.group(2) if m else n[12:60])
    return n.split("(")[0][:70]


def main(path, skip_tail=0):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if skip_tail:
        rows = rows[:-skip_tail]
    names = [r["Kernel_Name"] for r in rows]
    n = len(names)
    best = None
    for p in range(50, min(3000, n // 3)):
        if names[n - p:] == names[n - 2 * p:n - p] == names[n - 3 * p:n - 2 * p]:
            best = p
            break
    # bench.py ends with other work (roofline leg, copies): if the tail is not the forward graph, locate the replays by a
    # kernel that runs once per forward (the U-Net prologue / timestep sinusoid) and take the last three equal-length periods
    marks = [i for i, nm in enumerate(names) if "unet_prologue_kernel" in nm or "sin_kernel" in nm]
    if len(marks) >= 4 and (best is None or best < 100):
        done = False
        for stride in (1, 2):           # 2: a train step runs the U-Net prologue twice (teacher, student)
            for j in range(len(marks) - 1, 3 * stride - 1, -1):
                p = marks[j] - marks[j - stride]
                if p >= 100 and marks[j - stride] - marks[j - 2 * stride] == p and marks[j - 2 * stride] - marks[j - 3 * stride] == p \
                        and names[marks[j - stride]:marks[j]] == names[marks[j - 2 * stride]:marks[j - stride]]:
                    best, n = p, marks[j]
                    rows = rows[:n]
                    done = True
                    break
            if done:
                break
    if best is None:
        print("no periodic tail found")
        return
    p = best
    seg = rows[n - 2 * p:n - p]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for r in seg:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += d
        busy += d
    print(f"period = {p} kernels, wall {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, gaps {(t1 - t0 - busy) / 1e6:.3f} ms")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / 1e3:9.1f} us  {c:4d} x {t / c / 1e3:7.2f} us  {k}")


def per_step(path, marker, steps=8):
    """Training steps replay several graphs on more than one stream and end with eager launches, so the trace has no single
    periodic sequence: take the `steps` consecutive occurrences of a once-per-step kernel (`marker`) whose spacing in time is
    most regular (the timed replays, not the warm-up / capture / roofline legs) and average everything launched between them."""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if marker in r["Kernel_Name"]]
    if len(marks) < steps + 1:
        print(f"marker {marker!r} seen {len(marks)} times: need {steps + 1}")
        return
    best = None
    for j in range(len(marks) - steps):
        iv = [marks[j + i + 1] - marks[j + i] for i in range(steps)]
        spread = (max(iv) - min(iv)) / (sum(iv) / steps)
        if best is None or spread < best[0] - 1e-9 or (abs(spread - best[0]) < 0.02 and sum(iv) < best[2]):
            best = (spread, j, sum(iv))
    spread, j, total = best
    t0, t1 = marks[j], marks[j + steps]
    agg = defaultdict(lambda: [0, 0])
    busy = n = 0
    for r in rows:
        s = int(r["Start_Timestamp"])
        if t0 <= s < t1:
            d = int(r["End_Timestamp"]) - s
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += d
            busy += d
            n += 1
    print(f"per-step averages over {steps} consecutive steps (marker {marker}, spacing spread {spread * 100:.1f} %): "
          f"wall {total / steps / 1e6:.3f} ms, busy kernel time {busy / steps / 1e6:.3f} ms (streams overlap), {n / steps:.1f} launches")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / steps / 1e3:9.1f} us  {c / steps:7.1f} x {t / c / 1e3:7.2f} us  {k}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and not sys.argv[2].isdigit():
        per_step(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
