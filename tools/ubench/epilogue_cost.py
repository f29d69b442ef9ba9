#!/usr/bin/env python3
"""Cost of the three gemm256 epilogues on the Wo shape [131072 x 768 x 768]: bf16 / fp32 / fp32 + residual."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cm3p_amd import kernels as K
from cm3p_amd._lib import EPI_BF16, EPI_F32, EPI_F32_RESID

T, N, Kd = 131072, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 768
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(T, Kd, device="cuda", generator=g).bfloat16()
w = (torch.randn(N, Kd, device="cuda", generator=g) * 0.02).bfloat16()
r = torch.randn(T, N, device="cuda", generator=g)
def timeit(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for name, epi, res in (("bf16", EPI_BF16, None), ("f32", EPI_F32, None), ("f32+resid", EPI_F32_RESID, r)):
    ms = timeit(lambda: K.gemm(x, w, T, N, Kd, True, True, epi, res))
    byt = T * Kd * 2 + T * N * (2 if epi == EPI_BF16 else 4) * (2 if res is not None else 1)
    print(f"{name:10s} {ms:.3f} ms  {2.0*T*N*Kd/ms/1e9:7.1f} TF/s  {byt/ms/1e9:6.2f} TB/s algorithmic")
