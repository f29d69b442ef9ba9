# Builds _ab/libcm3p_g4w_<n>.so: the library with gemm4w.hip compiled with -DCM3P_G4W_SCHED=<n> (run here; the .so files travel)
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/_ab; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "audit.o" | grep -v "/gemm4w.o")
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-inline-asm -DCM3P_G4W_SCHED=$m -c $C/gemm4w.hip -o $O/g4w_$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libcm3p_g4w_$m.so $OBJS $O/g4w_$m.o && rm $O/g4w_$m.o && echo built $O/libcm3p_g4w_$m.so
done
