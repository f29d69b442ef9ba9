#!/bin/bash
# Matrix-busy share, cycles and clock of the attention kernels for TWO builds of the library (a one-call A/B of a re-placement: did the cycles
# move, or only the clock?).   bash tools/ubench/lib_pmc_mfma.sh <libA.so> <libB.so> [fwd|bwd] [c2|c4]
R=$(pwd); O=$R/gpurun_out/lib_pmc; mkdir -p $O
W=${3:-bwd}; SH=${4:-c4}
export CM3P_ALLOW_ABLATED_LIB=1
cd /tmp && export TMPDIR=/tmp
i=0
for lib in "$1" "$2"; do
  i=$((i+1))
  export CM3P_HIP_LIB=$lib
  d=$O/run_$i
  rm -rf $d
  timeout -k 10 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $d -o out -- python3 $R/tools/attn_probe.py $W -1 $SH > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== $lib"
  if [ -n "$f" ]; then python3 $R/tools/pmc_mfma.py "$f" $d.json | grep attn_; python3 $R/tools/pmc_sq.py "$f" attn_${W} | grep -E "attn_|WAIT|ACTIVE|WAVE_CYCLES|GUI"; else tail -5 $d.log; fi
done
