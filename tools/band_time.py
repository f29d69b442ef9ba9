#!/usr/bin/env python3
"""Sliding-window attention forward and backward at the step's shapes (C2: B 32 x S 4096, C4: B 16 x S 8192; 12 heads, window +-64):
time per call, and the rate for the bytes one pass must move.  A/B two builds with CM3P_HIP_LIB.

    python tools/band_time.py [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    nh = 12
    g = torch.Generator(device="cuda").manual_seed(0)
    for B, S in ((32, 4096), (16, 8192)):
        T = B * S
        qkv = (torch.randn(T, 3 * nh * 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        do = (torch.randn(T, nh * 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, device="cuda").float() / 64))
        ang = torch.arange(S, device="cuda").float()[:, None] * inv[None]
        cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
        o, lse = K.attn_fwd(qkv, None, B, S, nh, 64, 0.125, prescaled=True)

        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters

        tf = timed(lambda: K.attn_fwd(qkv, None, B, S, nh, 64, 0.125, prescaled=True))
        tb = timed(lambda: K.attn_bwd(qkv, o, do, lse, None, B, S, nh, 64, 0.125, (cos, sin), False, prescaled=True))
        unit = T * nh * 64 * 2 / 1e9  # GB of one [T, 768] bf16 tensor
        print(f"B {B} S {S}: forward {tf * 1e3:6.1f} us ({4 * unit / tf:.2f} TB/s for q, k, v in + o out)   backward (dq + dk/dv sweeps) {tb * 1e3:6.1f} us "
              f"({8 * unit / tb:.2f} TB/s for q, k, v, o, dO in + dq, dk, dv out)", flush=True)


if __name__ == "__main__":
    main()
