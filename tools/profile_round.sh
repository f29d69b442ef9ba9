#!/bin/bash
# GPU box: the rocprofv3 passes whose summaries are committed under profiles/ (run from the repository root).
#   bash tools/profile_round.sh r04 [c2|c4]      (workload: the judged C2 by default, C4 = the north-star shape)
# 1. kernel trace + stats of the judged command; 2. FETCH_SIZE, 3. WRITE_SIZE and 4. matrix-pipe busy cycles in separate counter-only passes
# Every pass runs under its own timeout: in r01 one counter pass sat silent until the box watchdog ended the call; the logs of that
# call were not kept, so its cause could not be established afterwards (profiles/README.md).  Always `-- python3 <file>`: a script
# started through its shebang would re-exec under the profiler.
# (MI355X_MICROARCH.md: never combine --pmc with other trace domains), turned into HBM bytes per launch by tools/pmc_traffic.py.
set -e
TAG=${1:-r04}
WL=${2:-c2}
R=$(pwd)
OUT=$R/gpurun_out/prof_${TAG}_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-optimizer --no-secondary"
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o out -- $CMD > $OUT/stats.log 2>&1
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o out -- $CMD > $OUT/fetch.log 2>&1
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o out -- $CMD > $OUT/write.log 2>&1
timeout -k 10 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o out -- $CMD > $OUT/mfma.log 2>&1
cd $R
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
cp "$S" $R/gpurun_out/${TAG}_${WL}_kernel_stats.csv
python3 tools/pmc_traffic.py "$F" "$W" $R/gpurun_out/traffic_${WL}.json > $OUT/traffic.log
U=$(find $OUT/mfma -name "*counter_collection.csv" | head -1)
python3 tools/pmc_mfma.py "$U" $R/gpurun_out/mfma_util_${WL}.json > $OUT/mfma_util.log
cat $OUT/mfma_util.log | head -14
tail -3 $OUT/stats.log
head -12 $R/gpurun_out/${TAG}_${WL}_kernel_stats.csv | cut -c1-110
