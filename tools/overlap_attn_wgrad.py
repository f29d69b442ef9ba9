#!/usr/bin/env python3
"""Third overlap probe: a weight-gradient GEMM (off the backward's critical chain) on a second stream beside the global attention backward,
the sliding-window backward, or an input-gradient GEMM - all kernels whose workgroups own whole CUs, so only the tails can be shared."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402

DEV = "cuda"
B, S, nh, H, N = 32, 4096, 12, 768, 2304
T = B * S
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)
qkv = rnd(B, S, 3, nh, 64)
do = rnd(T, nh * 64) * 0.1
dy, a, dy768, wgt = rnd(T, N), rnd(T, H), rnd(T, H), rnd(N, H) * 0.02
side = torch.cuda.Stream()
main_s = torch.cuda.current_stream()
outs = {w: K.attn_fwd(qkv, None, B, S, nh, w, 0.125) for w in (-1, 64)}


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


chain = {"global attention backward": lambda: K.attn_bwd(qkv, outs[-1][0], do, outs[-1][1], None, B, S, nh, -1, 0.125),
         "sliding-window backward": lambda: K.attn_bwd(qkv, outs[64][0], do, outs[64][1], None, B, S, nh, 64, 0.125),
         "dgrad K=2304": lambda: K.linear_dgrad(dy, wgt)}
sides = {"Wo wgrad": lambda: K.linear_wgrad(dy768, a), "Wqkv wgrad": lambda: K.linear_wgrad(dy, a)}
for cn, cf in chain.items():
    for sn, sf in sides.items():
        def seq():
            cf()
            sf()

        def conc():
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                sf()
            cf()
            main_s.wait_stream(side)

        def conc2():  # chain kernel first, the GEMM behind it on the other stream
            cf()
            side.wait_stream(main_s) if False else None
            with torch.cuda.stream(side):
                sf()
            main_s.wait_stream(side)

        t_c, t_s = timed(cf), timed(sf)
        print(f"{cn} {t_c:.3f} ms + {sn} {t_s:.3f} ms: back to back {timed(seq):.3f}, GEMM issued first on the second stream {timed(conc):.3f}, "
              f"GEMM issued second {timed(conc2):.3f}", flush=True)
