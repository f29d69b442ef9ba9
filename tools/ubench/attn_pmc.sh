# SQ counters of the attention kernels at the C4 shape (counter-only passes, MI355X_MICROARCH.md "rocprofv3 PMC slots"): issue / wait / busy
# breakdown per kernel, written as JSON-ish text under gpurun_out/pmc_attn/ and copied to profiles/ by hand.
#   bash tools/ubench/attn_pmc.sh fwd            (pipelined global forward)      CM3P_ATTN_FWD_IMPL=wave3 bash ... fwd   (the r01-r04 kernel)
#   bash tools/ubench/attn_pmc.sh bwd            (fused global backward)
# (the program goes directly after `--`: never env / bash -c under the profiler)
R=$(pwd); W=${1:-fwd}; TAG=${2:-$W}; O=$R/gpurun_out/pmc_attn; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  d=$O/${TAG}_$i
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $d -o out -- python3 $R/tools/attn_probe.py $W -1 c4 > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== $TAG pass $i: $set"
  if [ -n "$f" ]; then python3 $R/tools/pmc_sq.py "$f" attn_; else tail -5 $d.log; fi
done
