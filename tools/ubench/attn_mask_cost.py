#!/usr/bin/env python3
"""What does an all-ones key mask cost the attention kernels compared with no mask (C2 global and local layers)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cm3p_amd import kernels as K
B, S, nh = 32, 4096, 12
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3, nh, 64, device="cuda", generator=g).bfloat16()
do = torch.randn(B * S, nh * 64, device="cuda", generator=g).bfloat16()
ones = torch.ones(B, S, dtype=torch.uint8, device="cuda")
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for window in (-1, 64):
    for name, m in (("no mask", None), ("all-ones mask", ones)):
        out, lse = K.attn_fwd(qkv, m, B, S, nh, window, 0.125)
        f = timeit(lambda: K.attn_fwd(qkv, m, B, S, nh, window, 0.125))
        b = timeit(lambda: K.attn_bwd(qkv, out, do, lse, m, B, S, nh, window, 0.125))
        print(f"window {window:3d} {name:14s} fwd {f:.3f} ms  bwd {b:.3f} ms")
