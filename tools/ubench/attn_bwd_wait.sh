#!/bin/bash
# Wait trace of the fused attention backward (tools/attn_bwd_trace.py) for the working tree's source and, if given, a second source file
# (e.g. an older placement with the same trace hooks; the file must know -DCM3P_FTRACE).
#   bash tools/ubench/attn_bwd_wait.sh [other_source.hip]
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/bwd_wait; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "audit" | grep -v "/attention_bwd_fused.o")
export CM3P_ALLOW_ABLATED_LIB=1
i=0
for src in $C/attention_bwd_fused.hip "$@"; do
  i=$((i+1))
  cp $src $C/_trace_tmp.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DCM3P_FTRACE=1 -c $C/_trace_tmp.hip -o $O/bwd_tr_$i.o 2>/dev/null || { echo "build $src failed"; rm -f $C/_trace_tmp.hip; continue; }
  rm -f $C/_trace_tmp.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_tr_$i.so $OBJS $O/bwd_tr_$i.o
  echo "== $src"
  for sh in c2 c4; do CM3P_HIP_LIB=$O/lib_tr_$i.so timeout -k 10 120 python3 tools/attn_bwd_trace.py $sh 2>&1 | grep -E "waves|per wave|Error|error"; done
done
