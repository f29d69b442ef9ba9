#!/usr/bin/env python3
"""Does an HBM-bound kernel of the backward chain (LayerNorm backward, GeGLU backward) hide under a weight-gradient GEMM that runs on a
second stream and leaves it some CUs?  (development probe; C2 shapes)

A gemm8p workgroup owns its CU (160 KiB of LDS, 2 x 248 VGPRs per SIMD), so the two kernels can only share the chip CU by CU: the GEMM
is launched first with CM3P_G8P_GRID workgroups, the streaming kernel with CM3P_LN_BWD_CAP workgroups lands on the CUs that are left.
Prints sequential and concurrent times per pair.

    python tools/overlap_ab.py [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402

DEV = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    T, H, N = 32 * 4096, 768, 2304
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)
    dy, a = rnd(T, N), rnd(T, H)                       # wgrad operands: dW[N, H] = dy^T a
    x = torch.randn(T, H, device=DEV, generator=g)
    w = torch.ones(H, device=DEV)
    dn = rnd(T, H)
    dres = torch.randn(T, H, device=DEV, generator=g)
    _, _, mean, rstd = K.layernorm_fwd(x, w, 1e-5, False, True)
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    split = [9]

    def gemm():
        return K.gemm(dy, a, N, H, T, False, False, K.EPI_F32, split_k=split[0])  # 27 tiles x split work items

    def ln():
        return K.layernorm_bwd(dn, x, w, mean, rstd, dres, True, inplace=False)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    def both_seq():
        gemm()
        ln()

    def both_conc():
        ev = torch.cuda.Event()
        ev.record(main_s)
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            gemm()                      # dispatched first: its workgroups take whole CUs
        ln()                            # lands on what is left
        main_s.wait_stream(side)

    K.gemm8p_set_grid(0)
    os.environ.pop("CM3P_LN_BWD_CAP", None)
    t_g, t_l, t_s = timed(gemm), timed(ln), timed(both_seq)
    print(f"alone: wgrad {t_g:.3f} ms, LN backward {t_l:.3f} ms, back to back {t_s:.3f} ms", flush=True)
    print(f"concurrent, untouched grids: {timed(both_conc):.3f} ms", flush=True)
    for grid in (243, 216, 189, 162):
        for per_cu in (4, 8):
            cap = (256 - grid) * per_cu
            split[0] = grid // 27  # one work item per workgroup, as the shipped split of 9 gives on 256 CUs
            K.gemm8p_set_grid(grid)
            os.environ["CM3P_LN_BWD_CAP"] = str(cap)
            tg, tl = timed(gemm), timed(ln)
            tc = timed(both_conc)
            print(f"GEMM on {grid} CUs ({tg:.3f} ms alone), LN backward with {cap} workgroups ({tl:.3f} ms alone): concurrent {tc:.3f} ms "
                  f"(back to back at full grids {t_s:.3f})", flush=True)


if __name__ == "__main__":
    main()
