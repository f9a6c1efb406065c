#!/bin/bash
# Ablated copies of libaptp_hip.so for the software-pipelined attention kernel (timing experiments only; -DAPTP_ATTN_ABL=<bits>,
# see csrc/attention.hip): tools/_abl/libaptp_attn<N>.so for every N given.  Usage: tools/build_attn_ablations.sh 1 2 4 8
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/diffusion_pruning_amd/csrc
mkdir -p $ROOT/tools/_abl
make -C $CS -j4 >/dev/null
for n in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -Wno-unused-function -DAPTP_ATTN_ABL=$n \
      -c $CS/attention.hip -o $ROOT/tools/_abl/attention_$n.o && \
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/_abl/libaptp_attn$n.so $ROOT/tools/_abl/attention_$n.o \
      $(ls $CS/*.o | grep -v "/attention.o") ) &
done
wait
