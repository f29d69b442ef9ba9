# Builds libcm3p_hip variants that differ only in CM3P_NT (common.h: which write-once / read-once global accesses are non-temporal)
# into _ab/ (git-ignored; the .so files travel to the GPU box with the snapshot).  Run HERE (hipcc cross-compiles), not on the GPU box:
#   bash tools/ubench/nt_variants.sh 0 1 3 7 15 31
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/_ab; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-inline-asm"
for m in "$@"; do
  mkdir -p $O/nt_$m
  for f in norm elementwise gemm gemm256 gemm8p attention attention_bwd attention_bwd_fused attention_generic head conv muon; do
    x=""; case $f in attention_bwd|attention_bwd_fused) x="-fno-slp-vectorize";; esac
    /opt/rocm/bin/hipcc $FLAGS $x -DCM3P_NT=$m -c $C/$f.hip -o $O/nt_$m/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libcm3p_nt_$m.so $O/nt_$m/*.o && rm -rf $O/nt_$m
  echo "built $O/libcm3p_nt_$m.so"
done
