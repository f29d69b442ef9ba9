#!/usr/bin/env python3
"""Can a weight-gradient GEMM with a SMALL per-workgroup footprint (the 128 x 128 kernel, two to three workgroups per CU) share CUs with
an HBM-bound kernel of the backward chain on two plain streams - the matrix cores doing the GEMM while the streaming kernel's waves wait
for memory?  (r03 measured the whole-CU ring kernel beside LayerNorm backward: no - it needs whole CUs.  The small kernel does not.)

For each pair: the streaming kernel alone, the weight gradient alone on the ring kernel (what the step runs) and on the 128 x 128 kernel,
then both at once (streaming kernel on the main stream, GEMM on a second one, joined at the end).  The pair pays if
`both at once` < `streaming alone + ring-kernel weight gradient alone`.

    python tools/coresident_probe.py [--iters 20]
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
    B, S, H, I, nh = 32, 4096, 768, 1152, 12
    T = B * S
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    x = torch.randn(T, H, device=DEV, generator=g)
    ones = torch.ones(H, device=DEV)
    _, xn, mean, rstd = K.layernorm_fwd(x, ones, 1e-5, False, True)
    dn = rnd(T, H)
    dres = torch.randn(T, H, device=DEV, generator=g)
    h, dg = rnd(T, 2 * I), rnd(T, I)
    qkv = rnd(T, 3 * H)
    o, lse = K.attn_fwd(qkv, None, B, S, nh, 64, 0.125, prescaled=True)
    do = rnd(T, H)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, device=DEV).float() / 64))
    ang = torch.arange(S, device=DEV).float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    dy3, dy1 = rnd(T, 3 * H), rnd(T, H)
    gact = rnd(T, I)
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    streaming = {
        "LayerNorm backward": lambda: K.layernorm_bwd(dn, x, ones, mean, rstd, dres, True, inplace=False),
        "GeGLU backward": lambda: K.geglu_bwd(dg, h),
        "sliding-window attention backward": lambda: K.attn_bwd(qkv, o, do, lse, None, B, S, nh, 64, 0.125, (cos, sin), False, prescaled=True),
    }
    wgrads = {
        "Wqkv / Wi (2304 x 768)": lambda: K.linear_wgrad(dy3, xn),
        "Wo (768 x 768)": lambda: K.linear_wgrad(dy1, xn),
        "Wo2 (768 x 1152)": lambda: K.linear_wgrad(dy1, gact),
    }

    def small(fn):
        def run():
            os.environ["CM3P_GEMM_IMPL"] = "128"
            try:
                return fn()
            finally:
                os.environ.pop("CM3P_GEMM_IMPL", None)
        return run

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    def both(sfn, gfn, gemm_first):
        def run():
            side.wait_stream(main_s)
            if gemm_first:
                with torch.cuda.stream(side):
                    gfn()
                sfn()
            else:
                sfn()
                with torch.cuda.stream(side):
                    gfn()
            main_s.wait_stream(side)
        return run

    t_g8 = {k: timed(f) for k, f in wgrads.items()}
    t_g1 = {k: timed(small(f)) for k, f in wgrads.items()}
    for sname, sfn in streaming.items():
        t_s = timed(sfn)
        print(f"{sname}: alone {t_s:.3f} ms", flush=True)
        for gname, gfn in wgrads.items():
            t_b1 = min(timed(both(sfn, small(gfn), True)), timed(both(sfn, small(gfn), False)))
            t_b8 = min(timed(both(sfn, gfn, True)), timed(both(sfn, gfn, False)))
            print(f"    + weight gradient {gname}: ring kernel alone {t_g8[gname]:.3f}, 128 x 128 kernel alone {t_g1[gname]:.3f}; back to back (ring) "
                  f"{t_s + t_g8[gname]:.3f}; at once with the 128 x 128 kernel {t_b1:.3f}, with the ring kernel {t_b8:.3f}", flush=True)


if __name__ == "__main__":
    main()
