# GPU box, repository root: the C2 and C4 steps under each _ab/libcm3p_nt_<mask>.so (built by tools/ubench/nt_variants.sh), two rounds,
# one call = one box.      bash tools/ubench/nt_ab.sh 0 1 3 7 ...
R=$(pwd)
for round in 1 2; do
for m in "$@"; do
  for wl in c2 c4; do
    CM3P_HIP_LIB=$R/_ab/libcm3p_nt_$m.so timeout -k 10 300 python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-optimizer --no-secondary --no-profile 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt mask $m round $round $wl: %.2f ms  loss %.6f' % (d['ms_per_step'], d['config']['loss']), flush=True)"
  done
done
done
