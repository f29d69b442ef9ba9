#!/bin/bash
# Timing-only ablation builds of the pipelined global attention forward (csrc/attention_fwd.hip, -DCM3P_GABL=mask: 1 no exponentials,
# 2 no max / reference decision, 4 no row sums, 8 no tile DMA in the loop, 16 no barrier, 32 no reference-move branch, 64 no K / V fragment reloads,
# 128 no bf16 packs, 256 score products start from the constant 0; results are wrong by construction) and
# variant builds (-DCM3P_FWD_U=2: 64 queries per wave), each timed by tools/attn_fwd_ab.py through CM3P_HIP_LIB.  Run from the repo
# root on the GPU box:   bash tools/ubench/attn_fwd_ablate.sh "0 1 2 4 8 16 7 31" ["-DCM3P_FWD_U=2" ...]
R=$(pwd)
C=$R/cm3p_amd/csrc
O=$R/gpurun_out/ablate_fwd
mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "audit" | grep -v "/attention_fwd.o")
build_time() {  # $1 tag, $2.. extra flags
  tag=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c $C/attention_fwd.hip -o $O/fwd_$tag.o 2>/dev/null || { echo "build $tag failed"; return; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$tag.so $OBJS $O/fwd_$tag.o
  echo "== $tag ($*)"
  CM3P_ALLOW_ABLATED_LIB=1 CM3P_HIP_LIB=$O/lib_$tag.so timeout -k 10 120 python3 tools/attn_fwd_ab.py time --iters 10 2>&1 | grep -E "^impl"
}
for m in ${1:-0 1 2 4 8 16 7 31}; do build_time abl$m -DCM3P_GABL=$m; done
shift
i=0
for f in "$@"; do i=$((i+1)); build_time var$i $f; done
