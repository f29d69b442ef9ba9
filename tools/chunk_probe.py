#!/usr/bin/env python3
"""Temporal blocking over tokens: does a layer's chain of kernels run faster when it is walked in token chunks small enough for every
intermediate to stay in the 256 MiB Infinity Cache between its producer and its consumer?  Forward chain of one layer behind the
attention (Wo + residual -> LayerNorm -> Wi -> GeGLU -> Wo2 + residual) and the backward chain of the MLP half (dgrad Wo2 -> GeGLU
backward -> dgrad Wi + both weight gradients -> LayerNorm backward), whole (T = 131072 rows) against 2 / 4 chunks; interleaved rounds.

    python tools/chunk_probe.py
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

T, H, I = 131072, 768, 1152
g = torch.Generator(device="cuda").manual_seed(0)
rb = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
o, x = rb(T, H), torch.randn(T, H, device="cuda", generator=g)
wo, wi, wo2 = rb(H, H) * 0.05, rb(2 * I, H) * 0.05, rb(H, I) * 0.05
wi_t, wo2_t = wi.t().contiguous(), wo2.t().contiguous()
w_ln = torch.ones(H, device="cuda")
gx32, gx16 = torch.randn(T, H, device="cuda", generator=g), rb(T, H)


def fwd(nchunks):
    n = T // nchunks
    outs = []
    for c in range(nchunks):
        sl = slice(c * n, (c + 1) * n)
        x_mid = K.linear_fwd(o[sl], wo, resid=x[sl])
        _, xn2, mean, rstd = K.layernorm_fwd(x_mid, w_ln, 1e-5, False, True, True)
        h = K.linear_fwd(xn2, wi)
        gg = K.geglu_fwd(h)
        x_out = K.linear_fwd(gg, wo2, resid=x_mid)
        outs.append((x_mid, xn2, mean, rstd, h, gg, x_out))
    return outs


saved = fwd(1)[0]


def bwd(nchunks):
    n = T // nchunks
    x_mid, xn2, mean, rstd, h, gg, _ = saved
    for c in range(nchunks):
        sl = slice(c * n, (c + 1) * n)
        dg = K.linear_dgrad(gx16[sl], wo2, wo2_t)
        dwo2 = K.linear_wgrad(gx16[sl], gg[sl])
        dh = K.geglu_bwd(dg, h[sl])
        dxn2 = K.linear_dgrad(dh, wi, wi_t)
        dwi = K.linear_wgrad(dh, xn2[sl])
        K.layernorm_bwd(dxn2, x_mid[sl], w_ln, mean[sl], rstd[sl], gx32[sl].clone() if False else gx32[sl], True, inplace=False)


for name, fn in (("forward chain (Wo+resid, LN, Wi, GeGLU, Wo2+resid)", fwd), ("backward chain of the MLP half", bwd)):
    res = {1: [], 2: [], 4: []}
    for _ in range(5):
        for nc in res:
            res[nc].append(timeit(lambda: fn(nc), 5))
    print(name + ": " + " | ".join(f"{nc} chunk(s) {statistics.median(v):.3f} ms (min {min(v):.3f})" for nc, v in res.items()), flush=True)
