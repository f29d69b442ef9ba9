# Which LDS traffic of the fused global backward conflicts?  Ablation builds (-DCM3P_FABL=mask: 2 no tile DMA, 4 no dS image writes, 8 no dQ operand
# reads) under a counter-only pass (SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE), C4 shape.   bash tools/ubench/attn_bwd_lds.sh "0 2 4 8 12"
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/bwd_lds; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v audit | grep -v "/attention_bwd_fused.o")
for m in ${1:-0 2 4 8}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DCM3P_FABL=$m -c $C/attention_bwd_fused.hip -o $O/f_$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$m.so $OBJS $O/f_$m.o
  export CM3P_ALLOW_ABLATED_LIB=1 CM3P_HIP_LIB=$O/lib_$m.so
  (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/p_$m -o out -- python3 $R/tools/attn_probe.py bwd -1 c4 > $O/p_$m.log 2>&1)
  echo "== FABL=$m"
  python3 $R/tools/pmc_sq.py $(find $O/p_$m -name "*counter_collection.csv" | head -1) "attn_bwd_fused_kernel<true, false>"
done
