#!/bin/bash
# GPU box, repository root: rocprofv3 kernel stats of the C2 (or C4) bench step, printing the small kernels that sit between the big ones on
# the critical chain (split-K reduce, column sums, token order, slab reduce, prep).   bash tools/ubench/small_kernels.sh [c2|c4]
WL=${1:-c2}
R=$(pwd)
O=$R/gpurun_out/small_$WL
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o out -- python3 $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-optimizer --no-secondary > $O/log 2>&1
cd $R
python3 - <<PY
import csv, re
rows = list(csv.DictReader(open("$O/out_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 8e6
print("sum of kernel durations per step: %.2f ms" % tot)
for r in rows:
    n = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    if float(r["AverageNs"]) < 40e3 and float(r["TotalDurationNs"]) / 8e6 > 0.02:
        print("%-46s calls/step %6.1f  avg %6.1f us  per step %6.3f ms" % (n[:46], int(r["Calls"]) / 8, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 8e6))
PY
tail -2 $O/log | cut -c1-300
