#!/bin/bash
# The fused global attention backward's ablation ladder with counters (the forward's is attn_fwd_power.sh): per timing-only build
# (-DCM3P_FABL=mask: 1 no barrier, 2 no tile DMA in the loop, 4 no dS image writes, 8 no dQ operand reads, 16 no slab stores, 32 no dQ MFMAs,
# 64 no exponentials, 128 no dS multiplies, 256 no bf16 packs, 512 no Q / dO row-fragment reloads, 1024 no transposed fragment reloads;
# results wrong by construction) one un-profiled timing leg and ONE counter pass at the C4 shape: matrix-busy share, cycles, and the clock the
# launch held (GRBM_GUI_ACTIVE / 8 / duration of the counter pass).
#   bash tools/ubench/attn_bwd_power.sh "0 4 8 12 28 60 62"
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/bwd_power; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "audit" | grep -v "/attention_bwd_fused.o")
export CM3P_ALLOW_ABLATED_LIB=1
cd /tmp && export TMPDIR=/tmp
for m in ${1:-0 4 8 12 28 60 62}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DCM3P_FABL=$m -c $C/attention_bwd_fused.hip -o $O/bwd_$m.o 2>/dev/null || { echo "build $m failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$m.so $OBJS $O/bwd_$m.o
  export CM3P_HIP_LIB=$O/lib_$m.so
  echo "== CM3P_FABL=$m"
  (cd $R && timeout -k 10 120 python3 tools/attn_bwd_ab.py --seq 8192 --batch 16 --rounds 3 --arms fused 2>&1 | grep -E "^fused ")
  d=$O/pmc_$m
  timeout -k 10 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $d -o out -- python3 $R/tools/attn_probe.py bwd -1 c4 > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_mfma.py "$f" $d.json | grep attn_bwd_fused; python3 $R/tools/pmc_sq.py "$f" attn_bwd_fused | grep -E "attn_bwd|INSTS|WAIT|WAVE_CYCLES"; else tail -5 $d.log; fi
done
