# GPU box, repository root: the C2 and C4 steps under two settings of one environment switch, two interleaved rounds, one call = one box.
#   bash tools/ubench/env_ab.sh CM3P_ATTN_BAND_MERGED 0 1
V=$1; shift
for round in 1 2; do
for val in "$@"; do
  for wl in c2 c4; do
    env $V=$val timeout -k 10 300 python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-optimizer --no-secondary --no-profile 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$val round $round $wl: %.2f ms  loss %.6f' % (d['ms_per_step'], d['config']['loss']), flush=True)"
  done
done
done
