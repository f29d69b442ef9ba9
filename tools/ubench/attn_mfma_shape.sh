#!/bin/bash
# GPU box, repository root (r06; the MFMA-shape question of the r05 verdict, answered in the real kernels instead of a microbenchmark):
# timing-only builds in which the ACCUMULATING products issue pairs of v_mfma_f32_16x16x32_bf16 on four-register accumulators where the
# product issues one v_mfma_f32_32x32x16_bf16 - the same FLOPs, the same operands, every other instruction of the stream unchanged, no
# layout fix-up (results wrong by construction: CM3P_GABL & 512 in attention_fwd.hip, CM3P_FABL & 2048 in attention_bwd_fused.hip).
# 1. kernel level, C4 shape, two interleaved rounds each: time, matrix-busy share, clock during the kernel (one counter pass per build);
# 2. step level: the C4 and C2 steps with each variant library against the product, two interleaved rounds.
R=$(pwd)
bash tools/ubench/attn_fwd_power.sh "0 512 0 512" 2>&1 | grep -v "^$"
bash tools/ubench/attn_bwd_power.sh "0 2048 0 2048" 2>&1 | grep -v "^$"
export CM3P_ALLOW_ABLATED_LIB=1
for round in 1 2; do
  for lib in product fwd16 bwd16; do
    case $lib in product) unset CM3P_HIP_LIB;; fwd16) export CM3P_HIP_LIB=$R/gpurun_out/fwd_power/lib_512.so;; bwd16) export CM3P_HIP_LIB=$R/gpurun_out/bwd_power/lib_2048.so;; esac
    for wl in c4 c2; do
      timeout -k 10 300 python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-optimizer --no-secondary --no-profile 2>/dev/null \
        | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step $lib round $round $wl: %.2f ms' % d['ms_per_step'], flush=True)"
    done
  done
done
