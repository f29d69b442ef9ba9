#!/usr/bin/env python3
"""Sliding-window backward of two builds of the library, bit for bit (four shapes: C2-like, key masks, no rotation, S = 256): run twice with --save
(the second time with CM3P_HIP_LIB pointing at the other build), then --compare.  Used for the resident / whole-row forms of the dQ sweep.
    python tools/band_resident_check.py --save /tmp/a.pt; CM3P_HIP_LIB=$PWD/_ab/other.so python tools/band_resident_check.py --save /tmp/b.pt
    python tools/band_resident_check.py --compare /tmp/a.pt /tmp/b.pt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if sys.argv[1] == "--compare":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        for k in a:
            same = torch.equal(a[k], b[k])
            print(k, "bit-identical" if same else f"DIFFERENT: max |d| {(a[k].float() - b[k].float()).abs().max().item():.3g}, {(a[k] != b[k]).float().mean().item():.2e} of elements")
        return
    from cm3p_amd import kernels as K

    out = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, B, S, nh, masked, rope in (("c2-like", 4, 4096, 12, False, True), ("masked", 3, 1024, 4, True, True), ("no-rope", 2, 512, 2, False, False), ("s256", 5, 256, 3, True, True)):
        qkv = (torch.randn(B * S, 3 * nh * 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        do = (torch.randn(B * S, nh * 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, device="cuda").float() / 64))
        ang = torch.arange(S, device="cuda").float()[:, None] * inv[None]
        cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
        km = None
        if masked:
            lens = torch.randint(S // 3, S + 1, (B,), generator=torch.Generator().manual_seed(1))
            km = (torch.arange(S)[None] < lens[:, None]).to(torch.uint8).cuda().contiguous()
        o, lse = K.attn_fwd(qkv, km, B, S, nh, 64, 0.125, prescaled=True)
        d = K.attn_bwd(qkv, o, do, lse, km, B, S, nh, 64, 0.125, (cos, sin) if rope else None, False, prescaled=True)
        out[name] = d.cpu()
    torch.save(out, sys.argv[2])
    print("saved", sys.argv[2])


if __name__ == "__main__":
    main()
