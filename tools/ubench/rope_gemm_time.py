#!/usr/bin/env python3
"""Time the fused Wqkv + RoPE GEMM at the C2 shape (for one-session A/B of two library builds via CM3P_HIP_LIB)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cm3p_amd import kernels as K
T, H, S = 131072, 768, 4096
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(T, H, device="cuda", generator=g).bfloat16()
w = (torch.randn(3 * H, H, device="cuda", generator=g) * 0.02).bfloat16()
inv_freq = 1.0 / (160000.0 ** (torch.arange(0, 64, 2, device="cuda", dtype=torch.float32) / 64))
cos, sin = K.rope_table(torch.arange(S, device="cuda"), inv_freq)
def timeit(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
ms = timeit(lambda: K.qkv_linear_rope(x, w, cos, sin, S, False))
print(f"qkv+rope {ms:.3f} ms  {2.0*T*3*H*H/ms/1e9:.1f} TF/s")
ms = timeit(lambda: K.linear_fwd(x, w))
print(f"qkv plain {ms:.3f} ms  {2.0*T*3*H*H/ms/1e9:.1f} TF/s")
