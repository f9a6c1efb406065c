"""Wave-level SQ counters of the headline step's dominant conv_gemm instantiations on their top shapes.
Every rocprofv3 run has the program directly after `--`; counters in two passes of <= 8 SQ slots; no trace domains beside
--kernel-trace.  Usage (on the GPU box): python3 tools/pmc_tiles.py gpurun_out/r4_tile_pmc.txt"""
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
PASS_A = [c for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVES",
                      "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA") if have(c)]
PASS_B = [c for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM",
                      "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") if have(c)]
# (label, B, H, Cin, Cout, k, tile, split_k, flags): profiles/r3_launch_table.txt
CASES = [("tile18 M4096 N640 K320 1x1 +res +rowstat (10 launches/step)", 4, 32, 320, 640, 1, 18, 1, "rs"),
         ("tile18 M4096 N640 K1280 1x1 +res (5/step)", 4, 32, 1280, 640, 1, 18, 1, "r"),
         ("tile49 M1024 N1280 K640 1x1 +res +rowstat (10/step)", 4, 16, 640, 1280, 1, 49, 1, "rs"),
         ("tile49 M1024 N1280 K2560 1x1 +res (5/step)", 4, 16, 2560, 1280, 1, 49, 1, "r"),
         ("tile34 M16384 N160 K5760 3x3 split2 (2/step)", 4, 64, 640, 160, 3, 34, 2, ""),
         ("tile34 M1024 N640 K23040 3x3 split8 (2/step)", 4, 16, 2560, 640, 3, 34, 8, ""),
         ("tile18 zero-work M4 N1280 K320 1x1 SiLU", 4, 1, 320, 1280, 1, 18, 1, "a")]
lines = []
for (label, B, H, Cin, Cout, k, tile, sk, flags) in CASES:
    agg = {}
    kname = ""
    for counters in (PASS_A, PASS_B):
        d = "/tmp/pmc_tiles"
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
               "python3", os.path.join(ROOT, "tools", "run_one_conv.py"), *(str(v) for v in (B, H, Cin, Cout, k, tile, sk, 6)), flags or "-"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp")
        fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not fs:
            lines.append(f"{label}: rocprofv3 produced no counters ({r.stderr[-300:]})")
            continue
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(fs[0])):
            if "conv_gemm" in row["Kernel_Name"] and "reduce" not in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                kname = row["Kernel_Name"]
        for kk, v in acc.items():
            v = v[1:] if len(v) > 2 else v          # drop the first (cold) launch
            agg[kk] = sum(v) / len(v)
    wc = agg.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    lines.append(f"== {label}\n   kernel {kname[:90]}")
    lines.append("   per launch: " + "  ".join(f"{kk} {v:.4g}" for kk, v in sorted(agg.items())))
    pct = {kk: 100.0 * agg[kk] / wc for kk in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") if kk in agg}
    lines.append("   share of wave cycles: " + "  ".join(f"{kk[3:]} {v:.1f}%" for kk, v in pct.items()))
    if "SQ_WAVES" in agg and agg["SQ_WAVES"]:
        w = agg["SQ_WAVES"]
        lines.append(f"   per wave: wave quad-cycles {wc / w:.0f} (= {4 * wc / w / 2.4e3:.2f} us at 2.4 GHz)  " + "  ".join(
            f"{kk[9:]} {agg[kk] / w:.1f}" for kk in sorted(agg) if kk.startswith("SQ_INSTS_")))
    print("\n".join(lines[-4:]), flush=True)
open(out_path, "w").write("\n".join(lines) + "\n")
