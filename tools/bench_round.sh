#!/bin/bash
# GPU box: the bench lines committed under profiles/ for one round (run from the repository root):  bash tools/bench_round.sh r03
set -e
TAG=${1:-r03}
O=gpurun_out
python3 bench.py --steps 10 --warmup 3 > $O/${TAG}_c2_bench.json 2> $O/${TAG}_c2_bench.err
python3 bench.py --workload c4 --steps 5 --warmup 2 --no-optimizer > $O/${TAG}_c4_bench.json 2> $O/${TAG}_c4_bench.err
python3 bench.py --workload c5 --steps 5 --warmup 2 --no-optimizer --no-cpu-baseline > $O/${TAG}_c5_bench.json 2> $O/${TAG}_c5_bench.err
python3 bench.py --padded --steps 5 --warmup 2 --no-optimizer --no-cpu-baseline > $O/${TAG}_c2_padded.json 2> $O/${TAG}_c2_padded.err
python3 bench.py --padded --unpad --steps 5 --warmup 2 --no-optimizer --no-cpu-baseline > $O/${TAG}_c2_padded_unpad.json 2> $O/${TAG}_c2_padded_unpad.err
for f in c2_bench c4_bench c5_bench c2_padded c2_padded_unpad; do python3 -c "
import json,sys
d=json.loads(open('$O/${TAG}_$f.json').read().strip().splitlines()[-1])
print('$f', round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['unit'], 'step frac', round(d['step_mfma_frac'],4), 'dominant', d.get('roofline',{}).get('kernel'), round(d.get('roofline',{}).get('frac',0),3))
"; done
