#!/usr/bin/env python3
"""The step's big GEMM shapes on the vendor library (torch.mm -> hipBLASLt / rocBLAS) beside this repo's kernels: a calibration of how
far the hand-written ring kernel is from what the library reaches on the same skinny shapes (N or K = 768 ... 2304, M = 131072).

    python tools/blas_compare.py [--iters 30]          (run under `rocprofv3 --kernel-trace --stats` to see the library's kernel names)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402

DEV = "cuda"


def timed(fn, iters):
    for _ in range(3):
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
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--T", type=int, default=32 * 4096)
    args = ap.parse_args()
    T = args.T
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    H, I = 768, 1152
    print(f"T = {T}; times in ms, rates in TFLOP/s (2mnk)", flush=True)
    # forward / input-gradient GEMMs: y[T, N] = x[T, Kd] w[N, Kd]^T, bf16 out
    for name, N, Kd in (("Wi forward", 2 * I, H), ("Wqkv forward (no RoPE)", 3 * H, H), ("Wo2 dgrad", I, H), ("Wi / Wqkv dgrad", H, 3 * H),
                        ("Wo dgrad", H, H), ("Wo2 forward (no residual)", H, I)):
        x, w = rnd(T, Kd), rnd(N, Kd)
        wt = w.t().contiguous()
        fl = 2.0 * T * N * Kd / 1e9
        t_ours = timed(lambda: K.gemm(x, w, T, N, Kd, True, True, K.EPI_BF16), args.iters)
        t_nt = timed(lambda: torch.mm(x, w.t()), args.iters)
        t_nn = timed(lambda: torch.mm(x, wt), args.iters)
        ref = torch.mm(x, w.t()).float()
        err = (K.gemm(x, w, T, N, Kd, True, True, K.EPI_BF16).float() - ref).abs().max().item()
        print(f"{name:28s} [{T} x {N} x {Kd}]: ours {t_ours:.3f} ({fl / t_ours:6.0f})  library x w^T {t_nt:.3f} ({fl / t_nt:6.0f})  "
              f"library x (w^T stored) {t_nn:.3f} ({fl / t_nn:6.0f})   max |ours - library| {err:.3g}", flush=True)
    # weight gradients: dW[N, Kd] = dy[T, N]^T x[T, Kd]; ours fp32 out with a deterministic split-K, the library's bf16 out
    for name, N, Kd in (("Wi / Wqkv wgrad", 3 * H, H), ("Wo wgrad", H, H), ("Wo2 wgrad", H, I)):
        dy, x = rnd(T, N), rnd(T, Kd)
        fl = 2.0 * T * N * Kd / 1e9
        t_ours = timed(lambda: K.linear_wgrad(dy, x), args.iters)
        t_lib = timed(lambda: torch.mm(dy.t(), x), args.iters)
        try:
            t_lib32 = timed(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32), args.iters)
        except Exception as e:  # noqa: BLE001
            t_lib32 = float("nan")
            print("   (fp32-out torch.mm not available:", type(e).__name__, ")")
        print(f"{name:28s} [{N} x {Kd} x {T}]: ours (fp32 out) {t_ours:.3f} ({fl / t_ours:6.0f})  library bf16 out {t_lib:.3f} ({fl / t_lib:6.0f})  "
              f"library fp32 out {t_lib32:.3f} ({fl / t_lib32:6.0f})", flush=True)


if __name__ == "__main__":
    main()
