#!/bin/bash
# Is the pipelined global forward held by issue or by the clock the chip gives a busy matrix pipe?  Builds the timing-only ablations of
# csrc/attention_fwd.hip (CM3P_GABL, see attn_fwd_ablate.sh; results wrong by construction), and for each takes ONE counter pass
# (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE ...: matrix-busy share and the clock during the kernel) at the C4 shape.
#   bash tools/ubench/attn_fwd_power.sh "0 1 7 31 159 511"        (CLOCK_ONLY=1: only the in-kernel clock leg at the end)
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/fwd_power; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "audit" | grep -v "/attention_fwd.o")
export CM3P_ALLOW_ABLATED_LIB=1
cd /tmp && export TMPDIR=/tmp
for m in $([ -z "$CLOCK_ONLY" ] && echo ${1:-0 1 7 31 159 511}); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DCM3P_GABL=$m -c $C/attention_fwd.hip -o $O/fwd_$m.o 2>/dev/null || { echo "build $m failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$m.so $OBJS $O/fwd_$m.o
  export CM3P_HIP_LIB=$O/lib_$m.so
  echo "== CM3P_GABL=$m"
  (cd $R && timeout -k 10 120 python3 tools/attn_fwd_ab.py time --iters 10 2>&1 | grep -E "^impl.*8192")
  d=$O/pmc_$m
  timeout -k 10 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $d -o out -- python3 $R/tools/attn_probe.py fwd -1 c4 > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_sq.py "$f" attn_fwd_g; else tail -5 $d.log; fi
done
# in-kernel clock of each variant (a build with three stamps per wave, none inside the sweep: -DCM3P_GTRACE=3)
for m in ${1:-0 1 7 31 159 511}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DCM3P_GABL=$m -DCM3P_GTRACE=3 -c $C/attention_fwd.hip -o $O/fwdc_$m.o 2>/dev/null || { echo "build $m failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libc_$m.so $OBJS $O/fwdc_$m.o
  export CM3P_HIP_LIB=$O/libc_$m.so
  echo "== CM3P_GABL=$m (clock build)"
  (cd $R && timeout -k 10 120 python3 tools/attn_fwd_ab.py time --iters 10 2>&1 | grep -E "^impl.*8192"; timeout -k 10 120 python3 tools/attn_fwd_trace.py c4 clock 2>&1 | grep -E "clock|Error|error")
done
