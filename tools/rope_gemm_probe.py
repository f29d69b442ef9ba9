#!/usr/bin/env python3
"""20 launches each of the Wqkv + RoPE GEMM and of the plain bf16 GEMM of the same shape [T x 2304 x 768] (C2: T = 131072), for
rocprofv3 counter passes and for timing (the r03 verdict's question: what are the 0.65 GB per launch that the RoPE instance moves
beyond the plain one?).  CM3P_HIP_LIB selects the library (an ablated gemm8p build needs CM3P_ALLOW_ABLATED_LIB=1).

    python tools/rope_gemm_probe.py [--tokens 131072] [--seq 4096]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=131072)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--more", action="store_true")
    args = ap.parse_args()
    T, S, H = args.tokens, args.seq, 768
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(T, H, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(3 * H, H, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    cos = torch.randn(S, 32, device="cuda", generator=g)
    sin = torch.randn(S, 32, device="cuda", generator=g)
    cases = [("plain bf16", lambda: K.linear_fwd(x, w), 3 * H, H),
             ("Wqkv + RoPE", lambda: K.qkv_linear_rope(x, w, cos, sin, S, False, K.SOFTMAX_Q_SCALE), 3 * H, H)]
    if args.more:  # the other forward / input-gradient shapes of a layer
        wo = (torch.randn(H, H, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
        wo2 = (torch.randn(H, 1152, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
        g1 = torch.randn(T, 1152, device="cuda", generator=g).to(torch.bfloat16)
        r = torch.randn(T, H, device="cuda", generator=g)
        dy = torch.randn(T, 3 * H, device="cuda", generator=g).to(torch.bfloat16)
        wt = w.t().contiguous()  # [H, 3H]: dgrad through W^T, contraction 2304
        cases += [("Wo + resid", lambda: K.linear_fwd(x, wo, resid=r), H, H), ("Wo2 + resid", lambda: K.linear_fwd(g1, wo2, resid=r), H, 1152),
                  ("dgrad K=2304", lambda: K.linear_fwd(dy, wt), H, 3 * H)]
    for name, fn, N, Kd in cases:
        ms = [timeit(fn, args.iters) for _ in range(3)]
        print(f"{name:12s} [{T} x {N} x {Kd}]  {min(ms):.4f} ms (best of 3 x {args.iters})  {2.0 * T * N * Kd / min(ms) / 1e9:.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
