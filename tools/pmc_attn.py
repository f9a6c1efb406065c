"""Wave-level SQ counters of the attention forward at the headline's level-64 shape: the double-buffered two-group kernel (variant 4)
against the software-pipelined one (variant 6).  Every rocprofv3 run has the program directly after `--`; counters in passes of <= 8
SQ slots; no trace domains beside --kernel-trace.  Usage (on the GPU box): python3 tools/pmc_attn.py gpurun_out/r4_attn_pmc.txt"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1]
os.environ["TMPDIR"] = "/tmp"
avail = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, cwd="/tmp").stdout
def have(n):
    return n in avail
PASSES = [[c for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVES",
                       "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA") if have(c)],
          [c for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                       "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE") if have(c)]]
lines = []
for (label, variant, B, h, L) in [("double-buffered two-group kernel (variant 4)", 4, 4, 2, 4096), ("software-pipelined kernel (variant 6)", 6, 4, 2, 4096),
                                  ("double-buffered, dense heads (variant 4)", 4, 4, 5, 4096), ("software-pipelined, dense heads (variant 6)", 6, 4, 5, 4096)]:
    agg, kname = {}, ""
    for counters in PASSES:
        d = "/tmp/pmc_attn"
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
               "python3", os.path.join(ROOT, "tools", "run_one_attn.py"), str(variant), str(B), str(h), str(L), "6"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp")
        fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not fs:
            lines.append(f"{label}: rocprofv3 produced no counters ({r.stderr[-300:]})")
            continue
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(fs[0])):
            if "attn_fwd" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                kname = row["Kernel_Name"]
        for kk, v in acc.items():
            v = v[1:] if len(v) > 2 else v
            agg[kk] = sum(v) / len(v)
    wc = agg.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    lines.append(f"== B{B} h{h} L{L}: {label}\n   kernel {kname[:90]}")
    lines.append("   per launch: " + "  ".join(f"{kk} {v:.4g}" for kk, v in sorted(agg.items())))
    pct = {kk: 100.0 * agg[kk] / wc for kk in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") if kk in agg}
    lines.append("   share of wave cycles: " + "  ".join(f"{kk[3:]} {v:.1f}%" for kk, v in pct.items()))
    if agg.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in agg:
        lines.append(f"   MFMA busy / SQ busy cycles: {100.0 * agg['SQ_VALU_MFMA_BUSY_CYCLES'] / agg['SQ_BUSY_CYCLES']:.1f}%")
    if agg.get("SQ_LDS_IDX_ACTIVE"):
        lines.append(f"   LDS bank-conflict cycles / LDS active cycles: {100.0 * agg.get('SQ_LDS_BANK_CONFLICT', 0.0) / agg['SQ_LDS_IDX_ACTIVE']:.1f}%")
    if agg.get("SQ_WAVES"):
        w = agg["SQ_WAVES"]
        lines.append(f"   per wave: wave quad-cycles {wc / w:.0f}  " + "  ".join(f"{kk[9:]} {agg[kk] / w:.1f}" for kk in sorted(agg) if kk.startswith("SQ_INSTS_")))
txt = "\n".join(lines)
print(txt)
open(out_path, "w").write(txt + "\n")
