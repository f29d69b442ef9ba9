#!/usr/bin/env python3
"""GeGLU forward / backward at the step's shape (T = 131072, I = 1152): time per launch and the rate for the bytes they move.

    python tools/geglu_time.py [--iters 50]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    T, I = 32 * 4096, 1152
    g = torch.Generator(device="cuda").manual_seed(0)
    h = (torch.randn(T, 2 * I, device="cuda", generator=g)).to(torch.bfloat16)
    dg = (torch.randn(T, I, device="cuda", generator=g)).to(torch.bfloat16)

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

    tf = timed(lambda: K.geglu_fwd(h))
    tb = timed(lambda: K.geglu_bwd(dg, h))
    print(f"geglu forward  {tf * 1e3:7.1f} us  ({T * I * 6 / tf / 1e9:.2f} TB/s for 6 B per output)")
    print(f"geglu backward {tb * 1e3:7.1f} us  ({T * I * 10 / tb / 1e9:.2f} TB/s for 10 B per output pair)")
    # accuracy against fp64 on the bf16 inputs
    a, b = h[:4096, :I].double(), h[:4096, I:].double()
    ref = 0.5 * a * (1 + torch.erf(a / 2 ** 0.5)) * b
    out = K.geglu_fwd(h[:4096].contiguous()).double()
    print("forward  max |out - fp64| / (|fp64| + 1e-3):", ((out - ref).abs() / (ref.abs() + 1e-3)).max().item())
    d = dg[:4096].double()
    cdf = 0.5 * (1 + torch.erf(a / 2 ** 0.5))
    pdf = torch.exp(-0.5 * a * a) / (2 * torch.pi) ** 0.5
    dref = torch.cat([d * b * (cdf + a * pdf), d * a * cdf], dim=1)
    dout = K.geglu_bwd(dg[:4096].contiguous(), h[:4096].contiguous()).double()
    print("backward max |out - fp64| / (|fp64| + 1e-3):", ((dout - dref).abs() / (dref.abs() + 1e-3)).max().item())


if __name__ == "__main__":
    main()
