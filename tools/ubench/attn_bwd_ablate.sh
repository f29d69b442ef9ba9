#!/bin/bash
# Timing-only ablation builds of the hand-scheduled attention backward (results are wrong by construction; run from the repo root
# on the GPU box).  One library per ablation mask, each timed by tools/attn_bwd_ab.py through CM3P_HIP_LIB.
#   bash tools/ubench/attn_bwd_ablate.sh "1 2 4 8 3 7"
set -e
R=$(pwd)
C=$R/cm3p_amd/csrc
O=$R/gpurun_out/ablate
mkdir -p $O
OBJS=$(ls $C/*.o | grep -v attention_bwd.o)
for m in ${1:-1 2 4 8}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DCM3P_ABL=$m -c $C/attention_bwd.hip -o $O/attention_bwd_$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$m.so $OBJS $O/attention_bwd_$m.o
  echo "== ablation mask $m"
  CM3P_HIP_LIB=$O/lib_$m.so timeout -k 10 120 python3 tools/attn_bwd_ab.py --rounds 3 2>&1 | grep -E "global .*dkv|global .*dq"
done
