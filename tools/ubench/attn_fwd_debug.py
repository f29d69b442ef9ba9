#!/usr/bin/env python3
"""Where does the global attention forward differ from a torch fp32 reference?  (development aid)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cm3p_amd import kernels as K  # noqa: E402

torch.manual_seed(0)
for (B, S, nh, masked) in ((1, 256, 1, False), (1, 256, 1, True), (1, 512, 2, False), (2, 320, 2, True)):
    qkv = torch.randn(B, S, 3, nh, 64, device="cuda").to(torch.bfloat16)
    mask = torch.ones(B, S, dtype=torch.uint8, device="cuda") if masked else None
    out, lse = K.attn_fwd(qkv, mask, B, S, nh, -1, 0.125)
    q, k, v = (qkv[:, :, i].float().permute(0, 2, 1, 3) for i in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=0.125).permute(0, 2, 1, 3)
    got = out.view(B, S, nh, 64).float()
    err = (got - ref).abs()
    print(f"B={B} S={S} nh={nh} masked={masked}: max err {err.max().item():.4f}")
    e = err.amax(dim=(0, 2))  # [S, 64]
    badq = (e.amax(dim=1) > 0.02).nonzero().flatten().tolist()
    badd = (e.amax(dim=0) > 0.02).nonzero().flatten().tolist()
    print("   bad queries:", badq[:40], "... total", len(badq))
    print("   bad dims:", badd[:64], "total", len(badd))
    lref = torch.logsumexp(torch.einsum("bhqd,bhkd->bhqk", q, k) * 0.125, dim=-1)
    print("   lse max err", (lse.view(B, nh, S) - lref).abs().max().item())
