#!/usr/bin/env python3
"""One-process A/B of the two big-shape GEMM kernels (CM3P_GEMM_IMPL=256: gemm256.hip, default: gemm8p.hip): results compared
element by element (same accumulation order: bit-identical expected), then interleaved timing rounds (median and min).

    python tools/gemm_ab.py [--rounds 5] [--iters 20] [check] [cube] [step]
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

DEV = "cuda"


def impl(name):
    os.environ["CM3P_GEMM_IMPL"] = name


IMPLS = ("256", "8p")  # --impls: any of 128 (gemm.hip's 128 x 128 kernel for every shape), 256 (gemm256.hip), 8p (gemm8p.hip, default)


def ab(label, fn, flops, rounds, iters):
    res = {n: [] for n in IMPLS}
    for _ in range(rounds):
        for name in IMPLS:
            impl(name)
            res[name].append(timeit(fn, iters))
    base = statistics.median(res[IMPLS[0]])
    print(f"{label:34s} " + " | ".join(f"{n}: {statistics.median(v):7.3f} ms (min {min(v):7.3f}) {flops / statistics.median(v) / 1e9:7.1f} TF/s x{base / statistics.median(v):.3f}"
                                       for n, v in res.items()), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["check", "cube", "step"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--impls", default="256,8p")
    args = ap.parse_args()
    global IMPLS
    IMPLS = tuple(args.impls.split(","))
    g = torch.Generator(device=DEV).manual_seed(0)
    uni = lambda *s: (torch.rand(*s, device=DEV, generator=g) * 2 - 1).to(torch.bfloat16)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)

    if "check" in args.what:
        bad = 0
        for (M, N, Kd) in ((4096, 4096, 4096), (131072, 2304, 768), (65536, 768, 1152), (65536 + 64, 1152, 768), (8192 + 8, 2304 + 64, 64),
                           (12800, 768, 192), (256 * 30, 256 * 7, 128), (8208, 2320, 96), (70000, 784, 768)):
            a, w = uni(M, Kd), uni(N, Kd)
            r = torch.randn(M, N, device=DEV, generator=g)
            S = 4096 if M % 4096 == 0 else M
            cos, sin = torch.randn(S, 32, device=DEV, generator=g), torch.randn(S, 32, device=DEV, generator=g)
            rope_ok = N % 768 == 0 and N >= 2304 and Kd % 64 == 0
            outs = {}
            for name in IMPLS:
                impl(name)
                outs[name] = [K.linear_fwd(a, w), K.linear_fwd(a, w, resid=r), K.gemm(a, w, M, N, Kd, True, True, K.EPI_F32)]
                if rope_ok:
                    outs[name].append(K.qkv_linear_rope(a, w, cos, sin, S, False, 0.18))
            torch.cuda.synchronize()
            for other in IMPLS[1:]:
                for i, kind in enumerate(("bf16", "f32+resid", "f32", "bf16+rope")[:len(outs[IMPLS[0]])]):
                    x, y = outs[IMPLS[0]][i].float(), outs[other][i].float()
                    d = (x - y).abs().max().item()
                    dr = 0.0
                    if i < 3:
                        ref = (a[:64].float() @ w.float().T) + (r[:64] if i == 1 else 0)
                        dr = (y[:64] - ref).abs().max().item()
                    ok = d == 0.0 and bool(torch.isfinite(y).all())
                    bad += (not ok)
                    print(f"check [{M}x{N}x{Kd}] {kind:9s} max|{IMPLS[0]}-{other}| = {d:.3e}  max|{other}-torch| (64 rows) = {dr:.3e}  {'OK' if ok else 'MISMATCH'}", flush=True)
        if bad:
            print(f"{bad} MISMATCHES")
            sys.exit(1)
    if "cube" in args.what:
        for n in (4096, 8192):
            a, b = uni(n, n), uni(n, n)
            ab(f"cube {n}^3 fwd bf16", lambda: K.linear_fwd(a, b), 2.0 * n ** 3, args.rounds, args.iters)
    if "step" in args.what:
        T, H, I = 131072, 768, 1152
        x, g1 = rnd(T, H), rnd(T, I)
        for name, N, Kd, a in (("Wqkv/Wi", 3 * H, H, x), ("Wo", H, H, x), ("Wo2", H, I, g1)):
            w = rnd(N, Kd) * 0.02
            r = torch.randn(T, N, device=DEV, generator=g) if N == H else None
            ab(f"fwd {name} [{T}x{N}x{Kd}]" + (" +resid" if r is not None else ""), lambda: K.linear_fwd(a, w, resid=r), 2.0 * T * N * Kd, args.rounds, args.iters)
            if r is not None:
                ab(f"fwd {name} [{T}x{N}x{Kd}] bf16", lambda: K.linear_fwd(a, w), 2.0 * T * N * Kd, args.rounds, args.iters)
        cos = torch.randn(4096, 32, device=DEV, generator=g)
        sin = torch.randn(4096, 32, device=DEV, generator=g)
        w = rnd(3 * H, H) * 0.02
        ab("fwd Wqkv+rope", lambda: K.qkv_linear_rope(x, w, cos, sin, 4096, False, K.SOFTMAX_Q_SCALE), 2.0 * T * 3 * H * H, args.rounds, args.iters)


if __name__ == "__main__":
    main()
