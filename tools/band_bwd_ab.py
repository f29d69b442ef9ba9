#!/usr/bin/env python3
"""Sliding-window backward, interleaved rounds in one process: one workgroup per block (the r03 pair), the merged launch
(CM3P_ATTN_BAND_MERGED=1) and resident workgroups that walk the blocks (CM3P_ATTN_BAND_PERSISTENT, default since r04).

    python tools/band_bwd_ab.py [--batch 32 --seq 4096] [--rounds 5]
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    B, S, nh = args.batch, args.seq, 12
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = (torch.randn(B, S, 3, nh, 64, device="cuda", generator=g) * 0.7).bfloat16()
    do = (torch.randn(B * S, nh * 64, device="cuda", generator=g) * 0.1).bfloat16()
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, device="cuda", dtype=torch.float32) / 64))
    tables = K.rope_table(torch.arange(S, device="cuda"), inv_freq)
    out, lse = K.attn_fwd(qkv, None, B, S, nh, 64, 0.125, prescaled=True)
    modes = {"pair": dict(CM3P_ATTN_BAND_PERSISTENT="0", CM3P_ATTN_BAND_MERGED="0"), "merged": dict(CM3P_ATTN_BAND_PERSISTENT="0", CM3P_ATTN_BAND_MERGED="1"),
             "persistent": dict(CM3P_ATTN_BAND_PERSISTENT="1", CM3P_ATTN_BAND_MERGED="0")}
    res = {m: [] for m in modes}
    outs = {}
    run = lambda: K.attn_bwd(qkv, out, do, lse, None, B, S, nh, 64, 0.125, tables, False, prescaled=True)
    for _ in range(args.rounds):
        for m, env in modes.items():
            os.environ.update(env)
            res[m].append(timeit(run, args.iters))
            outs[m] = run()
    torch.cuda.synchronize()
    print(f"B={B} S={S}: " + " | ".join(f"{m} {statistics.median(v):.3f} ms (min {min(v):.3f})" for m, v in res.items())
          + f" | identical: {all(torch.equal(outs['pair'], o) for o in outs.values())}", flush=True)


if __name__ == "__main__":
    main()
