#!/usr/bin/env python3
"""The full-row GEMM + residual + LayerNorm measurement kernel (tools/ubench/fullrow_gemm_ln.hip) beside what the step runs
(the 256 x 256 ring kernel with the fp32 + residual epilogue, then LayerNorm forward), on the step's two shapes (Wo: K = 768, MLP Wo: K = 1152).

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/libfullrow.so tools/ubench/fullrow_gemm_ln.hip
    python tools/fullrow_probe.py [--iters 20]
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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--T", type=int, default=32 * 4096)
    args = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libfullrow.so"))
    lib.fullrow_gemm_ln.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    T, H = args.T, 768
    g = torch.Generator(device=DEV).manual_seed(0)
    gamma = (1.0 + 0.1 * torch.randn(H, device=DEV, generator=g)).contiguous()
    for Kd in (768, 1152):
        a = (torch.randn(T, Kd, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn(H, Kd, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
        r = torch.randn(T, H, device=DEV, generator=g)
        x = torch.empty(T, H, device=DEV)
        y = torch.empty(T, H, device=DEV, dtype=torch.bfloat16)
        mean, rstd = torch.empty(T, device=DEV), torch.empty(T, device=DEV)

        def fused(mode=0, grid=0):
            rc = lib.fullrow_gemm_ln(a.data_ptr(), w.data_ptr(), r.data_ptr(), gamma.data_ptr(), x.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                     rstd.data_ptr(), T, Kd, 1e-5, mode, grid, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc

        def pair():
            xm = K.linear_fwd(a, w, resid=r)
            return (xm,) + tuple(K.layernorm_fwd(xm, gamma, 1e-5, False, True))

        fused()
        torch.cuda.synchronize()
        xm, _, yn, mu, rs = pair()
        torch.cuda.synchronize()
        ex = (x - xm).abs().max().item()
        ey = (y.float() - yn.float()).abs().max().item()
        em = (mean - mu).abs().max().item()
        er = ((rstd - rs).abs() / rs).max().item()
        print(f"K = {Kd}: max |x - x_pair| {ex:.3g}, max |y - y_pair| {ey:.3g} (bf16), mean {em:.3g}, rstd rel {er:.3g}", flush=True)

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

        t_gemm = timed(lambda: K.linear_fwd(a, w, resid=r))
        t_ln = timed(lambda: K.layernorm_fwd(xm, gamma, 1e-5, False, True))
        t_pair = timed(pair)
        t_f = timed(fused)
        t_k = timed(lambda: fused(1))
        t_f512 = timed(lambda: fused(0, 512))
        fl = 2.0 * T * H * Kd / 1e9
        print(f"         ring GEMM + residual {t_gemm:.3f} ms, LayerNorm {t_ln:.3f} ms, the pair back to back {t_pair:.3f} ms;  full-row kernel {t_f:.3f} ms "
              f"(grid 512: {t_f512:.3f}), its k-loop alone {t_k:.3f} ms ({fl / t_k:.0f} TFLOP/s; the ring kernel's whole launch {fl / t_gemm:.0f})", flush=True)


if __name__ == "__main__":
    main()
