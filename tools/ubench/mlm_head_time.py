#!/usr/bin/env python3
"""Per-kernel time of the MLM head (CM3PPredictionHead + decoder + masked-LM loss) at the C2 token count, forward + backward.

    python tools/ubench/mlm_head_time.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cm3p_amd import _lib  # noqa: E402
from cm3p_amd.modeling_cm3p import _MLMHeadLossFn  # noqa: E402

T, H, V = 32 * 4096, 768, 3167
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(T, H, device=dev, generator=g).requires_grad_(True)
Wd = (torch.randn(H, H, device=dev, generator=g) * 0.02).requires_grad_(True)
nw = torch.ones(H, device=dev).requires_grad_(True)
Wdec = (torch.randn(V, H, device=dev, generator=g) * 0.02).requires_grad_(True)
bdec = torch.zeros(V, device=dev).requires_grad_(True)
labels = torch.randint(3, V - 3, (T,), device=dev, generator=g)
labels = torch.where(torch.rand(T, device=dev, generator=g) < 0.15, labels, torch.full_like(labels, -100))


def step():
    _, loss = _MLMHeadLossFn.apply(h, Wd, None, nw, Wdec, bdec, 1e-5, labels, None)
    loss.backward()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    step()
e1.record()
torch.cuda.synchronize()
print(f"MLM head fwd+bwd at T={T}: {e0.elapsed_time(e1) / 3:.2f} ms")
_lib.profile_begin()
step()
prof = _lib.profile_end()
for tag, (n, ms, _) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"  {tag:48s} x{n:<3d} {ms:8.3f} ms")
print(f"  sum of kernels {sum(v[1] for v in prof.values()):.2f} ms")
