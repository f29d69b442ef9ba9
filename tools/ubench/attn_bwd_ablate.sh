#!/bin/bash
# Timing-only ablation builds of the hand-scheduled attention backward kernels (results are wrong by construction; run from the
# repo root on the GPU box).  One library per ablation mask, each timed by tools/attn_bwd_ab.py through CM3P_HIP_LIB.
#   bash tools/ubench/attn_bwd_ablate.sh fused "1 2 4 8 16 32"     (attention_bwd_fused.hip, -DCM3P_FABL=mask)
#   bash tools/ubench/attn_bwd_ablate.sh pair "1 2 4 8 3 7"        (attention_bwd.hip, -DCM3P_ABL=mask)
set -e
R=$(pwd)
C=$R/cm3p_amd/csrc
O=$R/gpurun_out/ablate
mkdir -p $O
ARM=${1:-fused}
if [ "$ARM" = fused ]; then SRC=attention_bwd_fused; DEF=CM3P_FABL; else SRC=attention_bwd; DEF=CM3P_ABL; fi
OBJS=$(ls $C/*.o | grep -v "/$SRC.o")
for m in ${2:-1 2 4 8}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -D$DEF=$m -c $C/$SRC.hip -o $O/${SRC}_$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$m.so $OBJS $O/${SRC}_$m.o
  echo "== $ARM ablation mask $m"
  CM3P_ALLOW_ABLATED_LIB=1 CM3P_HIP_LIB=$O/lib_$m.so timeout -k 10 120 python3 tools/attn_bwd_ab.py --rounds 3 --arms $ARM 2>&1 | grep -E "^$ARM "
done
