#!/usr/bin/env python3
"""Runs the attention forward or backward a few times (for rocprofv3 --pmc / --kernel-trace passes).

    python tools/attn_probe.py [fwd|bwd] [window: -1 global (default), 64 local] [c2 (default: B 32, S 4096) | c4 (B 16, S 8192)]
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K

B, S, nh = (16, 8192, 12) if (len(sys.argv) > 3 and sys.argv[3] == "c4") else (32, 4096, 12)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3, nh, 64, device="cuda", generator=g).to(torch.bfloat16)
qkv[:, :, 0] *= 0.18  # (q carries scale * log2 e)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
W = int(sys.argv[2]) if len(sys.argv) > 2 else -1
out, lse = K.attn_fwd(qkv, None, B, S, nh, W, 0.125, True)
do = torch.randn(B * S, nh * 64, device="cuda", generator=g).to(torch.bfloat16)
for _ in range(5):
    if which == "fwd":
        K.attn_fwd(qkv, None, B, S, nh, W, 0.125, True)
    else:
        K.attn_bwd(qkv, out, do, lse, None, B, S, nh, W, 0.125, None, False, True)
torch.cuda.synchronize()
