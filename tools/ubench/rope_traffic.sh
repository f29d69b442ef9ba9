# GPU box, repository root: what the Wqkv + RoPE instance of gemm8p moves beyond the plain GEMM of the same shape (r03 verdict item 4).
# Builds a timing-only variant whose RoPE epilogue does not read its tables (CM3P_G8P_ABL=128), then per library: timing, FETCH_SIZE,
# WRITE_SIZE + L2 hit / miss, each in its own counter-only pass.      bash tools/ubench/rope_traffic.sh
R=$(pwd); C=$R/cm3p_amd/csrc; O=$R/gpurun_out/rope_traffic; mkdir -p $O
OBJS=$(ls $C/*.o | grep -v "/gemm8p.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -DCM3P_G8P_ABL=128 -c $C/gemm8p.hip -o $O/g8p_128.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libg_128.so $OBJS $O/g8p_128.o
cd /tmp && export TMPDIR=/tmp
for lib in shipped notables; do
  if [ $lib = notables ]; then export CM3P_HIP_LIB=$O/libg_128.so CM3P_ALLOW_ABLATED_LIB=1; fi
  echo "== $lib: timing"
  timeout -k 10 200 python3 $R/tools/rope_gemm_probe.py
  for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    d=$O/${lib}_$(echo $set | cut -c1-10 | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $d -o out -- python3 $R/tools/rope_gemm_probe.py --iters 5 > $d.log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    echo "== $lib counters: $set"
    python3 - "$f" <<'PY'
import collections, csv, sys
acc, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm8p" not in r["Kernel_Name"]:
        continue
    k = ("rope " if "true, true, 3" in r["Kernel_Name"] else "plain") + " " + r["Counter_Name"]
    acc[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(acc):
    print(f"   {k:36s} {acc[k] / cnt[k]:16.1f} per launch ({cnt[k]} launches)")
PY
  done
done
