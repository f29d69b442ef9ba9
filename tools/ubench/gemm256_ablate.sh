set -e
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/ablate; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "/gemm256.o")
for m in 0 1 2 3; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCM3P_G256_ABL=$m -c $C/gemm256.hip -o $O/g256_$m.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libg_$m.so $OBJS $O/g256_$m.o
echo "== gemm256 ablation $m"; CM3P_HIP_LIB=$O/libg_$m.so timeout -k 10 200 python3 tools/bench_kernels.py gemm 2>&1 | grep -E "fwd   Wqkv|dgrad Wqkv|fwd   Wo |dgrad Wi"
done
