# GPU box, repository root: cache policy of the GEMM's output stores (CM3P_G8P_STORE = 0 plain, 1 sc1, 2 nt): do the output lines that
# plain stores leave in the XCD's L2 push the operand panels out of it?  Timing + FETCH_SIZE per policy.   bash tools/ubench/store_policy.sh
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/store_policy; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "/gemm8p.o")
for m in 0 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -DCM3P_G8P_STORE=$m -c $C/gemm8p.hip -o $O/g8p_$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libg_$m.so $OBJS $O/g8p_$m.o
done
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for m in 0 1 2; do
  export CM3P_HIP_LIB=$O/libg_$m.so
  echo "== store policy $m (round $round): timing"
  timeout -k 10 200 python3 $R/tools/rope_gemm_probe.py --more
done
done
for m in 0 1 2; do
  export CM3P_HIP_LIB=$O/libg_$m.so
  d=$O/fetch_$m
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -o out -- python3 $R/tools/rope_gemm_probe.py --iters 5 --more > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== store policy $m: bytes fetched (FETCH_SIZE x 2) per launch by kernel instance"
  python3 - "$f" <<'PY'
import collections, csv, re, sys
acc, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm8p" not in r["Kernel_Name"]:
        continue
    k = re.search(r"gemm8p_kernel<[^>]*>", r["Kernel_Name"]).group(0)
    acc[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(acc):
    print(f"   {k:60s} {acc[k] / cnt[k] * 2 * 1024 / 1e6:10.1f} MB fetched per launch ({cnt[k]} launches)")
PY
done
cd $R
CM3P_HIP_LIB=$O/libg_1.so timeout -k 10 300 python3 tools/gemm_ab.py check 2>&1 | tail -8
