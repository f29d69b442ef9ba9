#!/bin/bash
# HERE (hipcc cross-compiles), repository root: a second build of the library that differs from the in-tree one in ONE object, for a
# one-call A/B on the GPU box (CM3P_HIP_LIB=..., tools/ubench/lib_ab.sh).  The alternative source is a file or a git revision of the
# in-tree file; every other object is the in-tree one (same ABI, same Python side).
#   bash tools/ubench/alt_lib.sh _ab/lib_old_reduce.so attention_bwd_fused.hip HEAD          (the committed version of that file)
#   bash tools/ubench/alt_lib.sh _ab/lib_variant.so attention_bwd_fused.hip tools/ubench/variants/attention_bwd_fused_even.hip [-DFOO=1 ...]
set -e
OUT=$1; NAME=$2; SRC=$3; shift 3
C=cm3p_amd/csrc
T=$(mktemp -d)
if [ -f "$SRC" ]; then cp "$SRC" $T/$NAME; else git show "$SRC:$C/$NAME" > $T/$NAME; fi
EXTRA=""
case $NAME in attention_fwd.hip|attention_bwd.hip|attention_bwd_fused.hip) EXTRA="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-inline-asm $EXTRA "$@" -I$C -Iinclude -c $T/$NAME -o $T/alt.o
OBJS=""
for f in norm elementwise gemm gemm256 gemm8p attention attention_fwd attention_bwd attention_bwd_fused attention_generic head conv muon; do
  if [ "$f.hip" = "$NAME" ]; then OBJS="$OBJS $T/alt.o"; else OBJS="$OBJS $C/$f.o"; fi
done
mkdir -p $(dirname $OUT)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
rm -rf $T
echo "built $OUT ($NAME from $SRC $*)"
