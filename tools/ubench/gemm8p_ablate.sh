# Timing-only ablations of gemm8p.hip (results invalid by construction): where a tile's time goes at the step's shapes.
#   bash tools/ubench/gemm8p_ablate.sh [masks...]       (on the GPU box, from the repository root)
set -e
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/ablate8p; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "/gemm8p.o")
for m in ${@:-0 1 2 4 6 8 14 32 64}; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -DCM3P_G8P_ABL=$m -c $C/gemm8p.hip -o $O/g8p_$m.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libg_$m.so $OBJS $O/g8p_$m.o
echo "== gemm8p ablation mask $m (1 no stores, 2 no epilogue, 4 no zeroing, 8 no vmcnt wait, 32 no B-lo reads in phase 1, 64 no LDS-DMA)"
CM3P_HIP_LIB=$O/libg_$m.so timeout -k 10 200 tools/ubench/gemm_harness time 2>&1
done
