#!/usr/bin/env python3
"""Standalone timing of the hot kernels at the C2 / C4 shapes (development aid; the judged numbers come from bench.py).

    python tools/bench_kernels.py [gemm] [attn] [ln] [--iters 10] [--seq 4096]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402

DEV = "cuda"


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["gemm", "attn", "ln"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    B, S, nh, H, I = args.batch, args.seq, 12, 768, 1152
    T = B * S
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)

    if "gemm" in args.what:
        x, g1 = rnd(T, H), rnd(T, I)
        for name, N, Kd, a in (("Wqkv", 3 * H, H, x), ("Wi", 2 * I, H, x), ("Wo", H, H, x), ("Wo2", H, I, g1)):
            w = rnd(N, Kd) * 0.02
            dy = rnd(T, N)
            r = torch.randn(T, N, device=DEV, generator=g) if N == H else None
            fl = 2.0 * T * N * Kd
            ms = timeit(lambda: K.linear_fwd(a, w, resid=r), args.iters)
            print(f"gemm fwd   {name:5s} [{T}x{N}x{Kd}] {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
            ms = timeit(lambda: K.linear_dgrad(dy, w), args.iters)
            print(f"gemm dgrad {name:5s} [{T}x{Kd}x{N}] {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
            ms = timeit(lambda: K.linear_wgrad(dy, a), args.iters)
            print(f"gemm wgrad {name:5s} [{N}x{Kd}x{T}] {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s")
    if "cube" in args.what:
        # square shapes on uniform random [-1, 1) operands: comparable with the guide's 256^2 8-phase template numbers
        # (cdna_hip_programming.md "The 256^2 8-phase template": 1.32-1.34 PF at 4096^3, 1.47 PF at 8192^3)
        for n in (4096, 8192):
            a = (torch.rand(n, n, device=DEV, generator=g) * 2 - 1).to(torch.bfloat16)
            b = (torch.rand(n, n, device=DEV, generator=g) * 2 - 1).to(torch.bfloat16)
            fl = 2.0 * n ** 3
            for name, fn in (("fwd  (kc,kc)", lambda: K.linear_fwd(a, b)), ("dgrad(kc,ks)", lambda: K.linear_dgrad(a, b)),
                             ("wgrad(ks,ks)", lambda: K.linear_wgrad(a, b))):
                ms = min(timeit(fn, args.iters) for _ in range(3))
                print(f"cube {n}^3 {name} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s", flush=True)
    if "attn" in args.what:
        qkv = rnd(B, S, 3, nh, 64)
        do = rnd(T, nh * 64)
        for window in (-1, 64):
            keys = S if window < 0 else 129
            fl = 4.0 * B * nh * S * keys * 64
            out, lse = K.attn_fwd(qkv, None, B, S, nh, window, 0.125)
            ms = timeit(lambda: K.attn_fwd(qkv, None, B, S, nh, window, 0.125), args.iters)
            print(f"attn fwd  window={window:3d} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s (algorithmic)")
            ms = timeit(lambda: K.attn_bwd(qkv, out, do, lse, None, B, S, nh, window, 0.125), args.iters)
            print(f"attn bwd  window={window:3d} {ms:7.3f} ms  {2.5 * fl / ms / 1e9:7.1f} TF/s (5-product algorithmic)")
    if "ln" in args.what:
        x = torch.randn(T, H, device=DEV, generator=g)
        w = torch.ones(H, device=DEV)
        dy = rnd(T, H)
        dres = torch.randn(T, H, device=DEV, generator=g)
        _, y16, mean, rstd = K.layernorm_fwd(x, w, 1e-5, False, True)
        ms = timeit(lambda: K.layernorm_fwd(x, w, 1e-5, False, True), args.iters)
        print(f"ln fwd  {ms:7.3f} ms  {T * H * 6 / ms / 1e9:6.2f} TB/s")
        ms = timeit(lambda: K.layernorm_bwd(dy, x, w, mean, rstd, dres, True), args.iters)
        print(f"ln bwd  {ms:7.3f} ms  {T * H * 16 / ms / 1e9:6.2f} TB/s")
        h = rnd(T, 2 * I)
        ms = timeit(lambda: K.geglu_fwd(h), args.iters)
        print(f"geglu fwd {ms:7.3f} ms  {T * I * 6 / ms / 1e9:6.2f} TB/s")


if __name__ == "__main__":
    main()
