#!/usr/bin/env python3
"""The pipelined global forward at a constant number of workgroups (3072 = 12 per CU) and 8 ... 128 key tiles per workgroup: a line
time = fixed + tiles * per_tile through the points separates what a workgroup costs by existing (dispatch, prologue, epilogue) from its
sweep (development aid, r05)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K

def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

g = torch.Generator(device="cuda").manual_seed(0)
pts = []
for B, S in ((256, 512), (128, 1024), (64, 2048), (32, 4096), (16, 8192)):
    qkv = torch.randn(B, S, 3, 12, 64, device="cuda", generator=g).to(torch.bfloat16)
    qkv[:, :, 0] *= 0.18
    ms = timeit(lambda: K.attn_fwd(qkv, None, B, S, 12, -1, 0.125, True))
    pts.append((S // 64, ms))
    print(f"B={B} S={S}: {S // 64:4d} tiles per workgroup, {ms:.3f} ms  ({ms / 12 * 1e3:.1f} us per workgroup round)")
(x0, y0), (x1, y1) = pts[-2], pts[-1]
b = (y1 - y0) / (x1 - x0)
print(f"per tile {b / 12 * 1e3:.3f} us per workgroup; fixed {(y1 - b * x1) / 12 * 1e3:.1f} us per workgroup (from the two longest)")
