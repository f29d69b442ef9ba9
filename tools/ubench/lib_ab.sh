# GPU box, repository root: the C2 and C4 steps with the in-tree library against another build of it (CM3P_HIP_LIB), two interleaved rounds.
#   bash tools/ubench/lib_ab.sh _ab/libcm3p_old_gelu.so
ALT=$(pwd)/$1
for round in 1 2; do
for lib in tree alt; do
  for wl in c2 c4; do
    if [ $lib = alt ]; then export CM3P_HIP_LIB=$ALT; else unset CM3P_HIP_LIB; fi
    timeout -k 10 300 python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-optimizer --no-secondary --no-profile 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib round $round $wl: %.2f ms  loss %.6f' % (d['ms_per_step'], d['config']['loss']), flush=True)"
  done
done
done
