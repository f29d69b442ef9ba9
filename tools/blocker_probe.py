#!/usr/bin/env python3
"""What happens to a ring-kernel GEMM (one workgroup per CU, fixed stride over the work items) when some CUs are held by another kernel -
an RCCL all-reduce beside the backward pass - and does launching MORE workgroups than CUs (the surplus is dispatched wherever a CU
comes free) fix it?  The stand-in for the communication kernel holds 64 KiB of LDS on `--held` CUs for `--us` microseconds.

    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libblocker.so tools/ubench/blocker.hip
    python tools/blocker_probe.py [--held 32] [--us 1500]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402

DEV = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--held", type=int, default=32)
    ap.add_argument("--us", type=int, default=1500)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libblocker.so"))
    lib.blocker_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    T, H, I = 32 * 4096, 768, 1152
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    sink = torch.zeros(1, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()
    x, w3, w1 = rnd(T, H), rnd(3 * H, H), rnd(H, 3 * H)
    dy3 = rnd(T, 3 * H)
    resid = torch.randn(T, H, device=DEV, generator=g)
    wo = rnd(H, H)
    cases = {
        "Wi forward (1536 items)": lambda: K.gemm(x, w3, T, 3 * H, H, True, True, K.EPI_BF16),
        "Wqkv dgrad (1536 items, K 2304)": lambda: K.gemm(dy3, w1, T, H, 3 * H, True, True, K.EPI_BF16),
        "Wo + residual (1536 items)": lambda: K.linear_fwd(x, wo, resid=resid),
        "Wqkv wgrad (243 items)": lambda: K.linear_wgrad(dy3, x),
    }

    def timed(fn, blocked):
        ts = []
        for _ in range(args.iters):
            torch.cuda.synchronize()
            if blocked:
                with torch.cuda.stream(side):
                    lib.blocker_launch(args.held, args.us, sink.data_ptr(), side.cuda_stream)
                # the blocker is resident before the GEMM is issued
                torch.cuda._sleep(200000)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4)
        ts.sort()
        return ts[len(ts) // 2]

    for name, fn in cases.items():
        row = []
        for grid in (None, 512, 768, 1024):
            if grid is None:
                K.gemm8p_set_grid(0)
            else:
                K.gemm8p_set_grid(grid)
            fn()
            row.append((grid or 256, timed(fn, False), timed(fn, True)))
        K.gemm8p_set_grid(0)
        print(name + ":  " + "   ".join(f"grid {g_}: free {a:.3f} ms, {args.held} CUs held {b:.3f} ms" for g_, a, b in row), flush=True)


if __name__ == "__main__":
    main()
