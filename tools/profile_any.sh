#!/bin/bash
# rocprofv3 kernel statistics of an arbitrary python tool on the GPU box -> gpurun_out/<tag>_kernel_stats.csv (+ the top rows).
# Usage (inside a gpurun command): bash tools/profile_any.sh <tag> <script.py> [args ...]
set -euo pipefail
tag=$1; script=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/prof_$tag"
rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$tag" -o b -- python3 "$R/$script" "$@" > "$R/gpurun_out/${tag}_prof.log" 2>&1
cd "$R"
s=$(find "/tmp/prof_$tag" -name '*kernel_stats.csv' | head -1)
[ -n "$s" ] || { echo "profile_any: no stats (see gpurun_out/${tag}_prof.log)"; exit 1; }
cp "$s" "gpurun_out/${tag}_kernel_stats.csv"
python - "$s" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:90]
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.1f} us  {n}")
PY
tail -2 "gpurun_out/${tag}_prof.log"
