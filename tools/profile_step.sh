#!/bin/bash
# Per-step kernel breakdown of bench.py on the GPU box: rocprofv3 kernel trace -> gpurun_out/<tag>_breakdown.txt (one period
# of the timed HIP-graph replays) and gpurun_out/<tag>_kernel_stats.csv (rocprofv3's own per-kernel summary of the run).
# Usage (inside a gpurun command): bash tools/profile_step.sh <tag> [lines to print] [extra bench.py arguments ...]
set -euo pipefail
tag=${1:-prof}
lines=${2:-45}
shift $(( $# > 2 ? 2 : $# ))
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/prof_$tag"
# (the program itself after `--`: no env / bash -c hop between the profiler's preloaded library and python)
rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$tag" -o b -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline "$@" > "$R/gpurun_out/${tag}_prof.log" 2>&1
cd "$R"
f=$(find "/tmp/prof_$tag" -name '*kernel_trace.csv' | head -1)
s=$(find "/tmp/prof_$tag" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && [ -n "$s" ] || { echo "profile_step: rocprofv3 wrote no trace (see gpurun_out/${tag}_prof.log)"; exit 1; }
# PROFILE_MARKER=<kernel that runs once per step>: per-step averages between its occurrences (training steps: several graphs on
# two streams + eager launches have no single periodic sequence); default: one period of the replayed forward graph
if [ -n "${PROFILE_MARKER:-}" ]; then
  python tools/trace_breakdown.py "$f" "$PROFILE_MARKER" 8 > "gpurun_out/${tag}_breakdown.txt"
else
  python tools/trace_breakdown.py "$f" > "gpurun_out/${tag}_breakdown.txt" || echo "(no periodic tail found: see the stats csv)" > "gpurun_out/${tag}_breakdown.txt"
fi
cp "$s" "gpurun_out/${tag}_kernel_stats.csv"
head -"$lines" "gpurun_out/${tag}_breakdown.txt"
