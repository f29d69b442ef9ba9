#!/bin/bash
# Variant builds of the hand-scheduled attention forward for debugging (run from the repo root on the GPU box):
#   bash tools/ubench/fwdvar.sh "-DCM3P_FWD3_DEFER=1e30f" "-DCM3P_FWD3_DEFER=-1e30f"
set -e
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/fwdvar; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v attention_fwd.o)
i=0
for flags in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c $C/attention_fwd.hip -o $O/fwd_$i.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$i.so $OBJS $O/fwd_$i.o
  echo "== $flags"
  CM3P_HIP_LIB=$O/lib_$i.so python3 tools/ubench/attn_fwd_debug.py 2>&1 | grep -E "max err|lse"
done
