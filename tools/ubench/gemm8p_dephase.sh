# De-phasing probe for gemm8p.hip: every second workgroup of an XCD starts late, so that one half's store bursts meet the other
# half's k-loops (results stay valid).   bash tools/ubench/gemm8p_dephase.sh   (GPU box, repository root)
set -e
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/dephase8p; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "/gemm8p.o")
for d in 0 4 8 12 16 24; do
if [ $d = 0 ]; then F="-DCM3P_G8P_ABL=0"; else F="-DCM3P_G8P_ABL=16 -DCM3P_G8P_DELAY=$d"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm $F -c $C/gemm8p.hip -o $O/g8p_$d.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libg_$d.so $OBJS $O/g8p_$d.o
echo "== start delay of every second workgroup: $d x ~0.5 us"
CM3P_HIP_LIB=$O/libg_$d.so timeout -k 10 200 tools/ubench/gemm_harness time 2>&1 | grep -E "fwd|shape"
done
