#!/usr/bin/env python3
"""Does a streaming kernel run faster when it reads its producer's output LAST-WRITTEN-FIRST?  The Infinity Cache (256 MiB, memory side)
still holds the tail of what the previous kernel wrote; a consumer that sweeps the rows in the producer's order reads the oldest
(evicted) rows first and evicts the newest with its own traffic before it gets to them.  Probe: Wo GEMM (fp32 + residual, writes x_mid
403 MB in row order) followed by LayerNorm forward over x_mid in ascending or descending row order (CM3P_LN_REVERSE), LayerNorm timed
alone with events, interleaved rounds.      python tools/mall_order_probe.py
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402

T, H = 131072, 768
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.randn(T, H, device="cuda", generator=g).to(torch.bfloat16)
wo = (torch.randn(H, H, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
x = torch.randn(T, H, device="cuda", generator=g)
w = torch.ones(H, device="cuda")
scratch = torch.empty(512 * 2 ** 20, dtype=torch.uint8, device="cuda")  # "cold": the cache holds something else


def one(reverse: bool, cold: bool):
    os.environ["CM3P_LN_REVERSE"] = "1" if reverse else "0"
    x_mid = K.linear_fwd(o, wo, resid=x)
    if cold:
        scratch.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.layernorm_fwd(x_mid, w, 1e-5, False, True, True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(3):
    one(False, False)
res = {(r, c): [] for r in (False, True) for c in (False, True)}
for _ in range(15):
    for key in res:
        res[key].append(one(*key))
for (r, c), v in res.items():
    print(f"LayerNorm forward behind the Wo GEMM, rows {'descending' if r else 'ascending '}, cache {'flushed by a 512 MiB fill' if c else 'as the GEMM left it   '}: "
          f"median {statistics.median(v) * 1e3:.1f} us  min {min(v) * 1e3:.1f} us", flush=True)
