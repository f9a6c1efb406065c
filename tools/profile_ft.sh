#!/bin/bash
# per-step kernel breakdown of the graphed expert fine-tune step (tools/bench_finetune.py) -> gpurun_out/<tag>_ft_breakdown.txt
set -euo pipefail
tag=${1:-ft}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/prof_$tag"
rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$tag" -o b -- python3 "$R/tools/bench_finetune.py" --expert 3 --steps 8 --warmup 2 --mode graphed > "$R/gpurun_out/${tag}_prof.log" 2>&1
cd "$R"
f=$(find "/tmp/prof_$tag" -name '*kernel_trace.csv' | head -1)
python tools/trace_breakdown.py "$f" > "gpurun_out/${tag}_ft_breakdown.txt"
head -${2:-60} "gpurun_out/${tag}_ft_breakdown.txt"
