#!/bin/bash
# Two separate rocprofv3 PMC passes over bench.py (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains) and
# the per-launch HBM-side traffic of the conv_gemm family -> gpurun_out/<tag>_pmc_traffic.json
set -euo pipefail
tag=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_fetch /tmp/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-extras --no-graph > "$R/gpurun_out/${tag}_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-extras --no-graph > "$R/gpurun_out/${tag}_write.log" 2>&1
cd "$R"
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write "gpurun_out/${tag}_pmc_traffic.json"
cat "gpurun_out/${tag}_pmc_traffic.json"
