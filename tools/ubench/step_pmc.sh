# SQ issue / wait / busy counters of EVERY kernel of the real step (counter-only passes over bench.py, three counter sets; MI355X_MICROARCH.md
# "rocprofv3 PMC slots").  Output: gpurun_out/sq_<workload>.txt (copied to profiles/<round>_sq_<workload>.txt), per kernel each counter's mean
# per launch and its share of SQ_WAVE_CYCLES.      bash tools/ubench/step_pmc.sh c4
R=$(pwd); WL=${1:-c4}; O=$R/gpurun_out/step_pmc_$WL; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer --no-secondary --no-profile"
: > $R/gpurun_out/sq_$WL.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC" \
           "SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o out -- $CMD > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i ($WL): $set" >> $R/gpurun_out/sq_$WL.txt
  if [ -n "$f" ]; then python3 $R/tools/pmc_sq.py "$f" attn_ gemm8p layernorm geglu >> $R/gpurun_out/sq_$WL.txt; else tail -3 $O/p$i.log >> $R/gpurun_out/sq_$WL.txt; fi
done
grep -c "per launch" $R/gpurun_out/sq_$WL.txt
