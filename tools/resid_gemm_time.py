#!/usr/bin/env python3
"""The fp32 + residual GEMMs of the step (Wo: K = 768, MLP Wo: K = 1152; M = 131072, N = 768) - time per launch, for one-call A/Bs of two
builds (CM3P_HIP_LIB).     python tools/resid_gemm_time.py [--iters 40]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    T, H = 32 * 4096, 768
    g = torch.Generator(device="cuda").manual_seed(0)
    r = torch.randn(T, H, device="cuda", generator=g)
    out = []
    for Kd in (768, 1152):
        a = (torch.randn(T, Kd, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn(H, Kd, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        for _ in range(3):
            x = K.linear_fwd(a, w, resid=r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            K.linear_fwd(a, w, resid=r)
        e1.record()
        torch.cuda.synchronize()
        out.append(f"K = {Kd}: {e0.elapsed_time(e1) / args.iters * 1e3:6.1f} us (checksum {x.double().sum().item():.6e})")
    print("   ".join(out), flush=True)


if __name__ == "__main__":
    main()
