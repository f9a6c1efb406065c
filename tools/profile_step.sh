#!/bin/bash
# Per-step kernel breakdown of bench.py on the GPU box: rocprofv3 kernel trace -> gpurun_out/<tag>_breakdown.txt and
# gpurun_out/<tag>_kernel_stats.csv.  Usage (inside a gpurun command): bash tools/profile_step.sh <tag>
tag=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- python3 $R/bench.py --steps 10 --warmup 3 > $R/gpurun_out/${tag}_prof.log 2>&1
cd $R
f=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
python tools/trace_breakdown.py $f > gpurun_out/${tag}_breakdown.txt
cp $(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv
head -${2:-45} gpurun_out/${tag}_breakdown.txt
