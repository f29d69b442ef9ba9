# SQ counters of the gemm8p k-loop (one shape per pass; counter-only passes, MI355X_MICROARCH.md "rocprofv3 PMC slots").
#   bash tools/ubench/gemm8p_pmc.sh      (GPU box, repository root)
R=$(pwd); O=$R/gpurun_out/pmc8p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name M N K epi a_kc b_kc splits
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES" \
             "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
    d=$O/$1_$(echo $set | cut -c1-12 | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $d -o out -- $R/tools/ubench/gemm_harness one $2 $3 $4 $5 $6 $7 $8 > $d.log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    echo "== $1 [$2 x $3 x $4] counters: $set"
    python3 $R/tools/pmc_sq.py "$f" gemm8p gemm256
  done
}
cd $R
run cube8192 8192 8192 8192 0 1 1 1
run fwdK768 131072 2304 768 0 1 1 1
run wgrad 2304 768 131072 1 0 0 9
