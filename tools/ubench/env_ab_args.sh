# GPU box, repository root: like env_ab.sh, with extra bench.py arguments and one workload: the step under two settings of one environment switch.
#   bash tools/ubench/env_ab_args.sh CM3P_DEFER_REDUCE "0 1" c2 --batch 2
V=$1; VALS=$2; WL=$3; shift 3
for round in 1 2; do
for val in $VALS; do
    env $V=$val timeout -k 10 300 python3 bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer --no-secondary --no-profile "$@" 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$val round $round $WL $*: %.3f ms  loss %.6f' % (d['ms_per_step'], d['config']['loss']), flush=True)"
done
done
