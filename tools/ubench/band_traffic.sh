# GPU box, repository root: fabric bytes of the sliding-window backward, pair against merged launch (FETCH_SIZE x 2 and WRITE_SIZE, separate passes)
R=$(pwd); O=$R/gpurun_out/band_traffic; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE; do
  d=$O/$set
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $d -o out -- python3 $R/tools/band_bwd_ab.py --rounds 1 --iters 3 > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" $set <<'PY'
import collections, csv, re, sys
acc, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "attn_bwd" not in n and "attn_delta" not in n:
        continue
    k = re.search(r"(attn_\w+)(<[^>]*>)?", n).group(0)
    acc[k] += float(r["Counter_Value"]); cnt[k] += 1
mul = 2 * 1024 if sys.argv[2] == "FETCH_SIZE" else 1024
for k in sorted(acc):
    print(f"   {sys.argv[2]:10s} {k:44s} {acc[k] / cnt[k] * mul / 1e6:10.1f} MB per launch ({cnt[k]} launches)")
PY
done
