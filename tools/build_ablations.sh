#!/bin/bash
# Build ablated copies of libaptp_hip.so (timing experiments only): tools/_abl/libaptp_abl<N>.so for every N given.
# Usage: tools/build_ablations.sh 0 32 64 96
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/diffusion_pruning_amd/csrc
mkdir -p $ROOT/tools/_abl
make -C $CS -j4 >/dev/null
PIDS=""
for n in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -Wno-unused-function -DAPTP_ABLATE=$n \
    -c $CS/conv_gemm.hip -o $ROOT/tools/_abl/conv_gemm_$n.o && \
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/_abl/libaptp_abl$n.so $ROOT/tools/_abl/conv_gemm_$n.o \
    $CS/norm.o $CS/attention.o $CS/attention_bwd.o $CS/train_ops.o $CS/unet_io.o $CS/abi.o ) &
done
wait
