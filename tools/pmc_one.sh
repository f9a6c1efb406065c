#!/bin/bash
# SQ wait / LDS counters of one conv_gemm launch shape: bash tools/pmc_one.sh B H Cin Cout k [tile split order]
# (averages the counters over the launches of tools/run_one_conv.py; units: SQ_* = quad-cycles summed over the chip)
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf /tmp/pone
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pone -- python3 tools/run_one_conv.py "$@" > /tmp/pone.log 2>&1
f=$(find /tmp/pone -name "*counter_collection.csv" | head -1)
python3 - "$f" "$*" <<'PY'
import csv, collections, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_gemm" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
a = {k: sum(v) / len(v) for k, v in agg.items()}
wc = a.get("SQ_WAVE_CYCLES", 1)
print(sys.argv[2], "|", rows[0]["Kernel_Name"][:70] if rows else "no conv_gemm rows")
print("  " + "  ".join(f"{k[3:]} {v / wc * 100:5.1f}%" for k, v in sorted(a.items()) if k != "SQ_WAVE_CYCLES"), f" wave-quadcycles {wc:.3g}")
PY
