set -e
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/ablate; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "/attention.o")
for m in 0 1 2 4 3 5 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCM3P_BABL=$m -c $C/attention.hip -o $O/att_$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/liba_$m.so $OBJS $O/att_$m.o
  echo "== band dkv ablation $m"
  CM3P_HIP_LIB=$O/liba_$m.so timeout -k 10 120 python3 tools/attn_bwd_ab.py --rounds 5 --window 64 --arms band 2>&1 | grep -E "dkv_kernel"
done
