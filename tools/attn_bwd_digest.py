#!/usr/bin/env python3
"""SHA-256 of the fused global attention backward's dq / dk / dv on a few seeded shapes (several 256-key blocks, ragged key masks, the
four-key-block slab group, both q modes): a re-placement of the kernel's instruction stream must leave every digest unchanged.

    python tools/attn_bwd_digest.py                    # this build
    CM3P_HIP_LIB=$PWD/_ab/lib_prev_bwd.so python tools/attn_bwd_digest.py     # the build to compare with
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402


def main():
    dev = "cuda"
    for (B, S, nh, lens, pre, grp) in [(2, 1536, 2, None, True, None), (3, 700, 2, [700, 513, 64], True, None), (2, 2048, 3, [2048, 1999], True, "4"),
                                       (2, 330, 1, [330, 257], False, None), (16, 4096, 12, None, True, None), (4, 8192, 12, None, True, None)]:
        if grp:
            os.environ["CM3P_FUSED_SLAB_GROUP"] = grp
        else:
            os.environ.pop("CM3P_FUSED_SLAB_GROUP", None)
        g = torch.Generator(device=dev).manual_seed(B * 1000 + S)
        qkv = torch.randn(B, S, 3, nh, 64, device=dev, generator=g).to(torch.bfloat16)
        if pre:
            qkv[:, :, 0] *= 0.18
        km = None
        if lens is not None:
            km = (torch.arange(S, device=dev)[None] < torch.tensor(lens, device=dev)[:, None]).to(torch.uint8)
        out, lse = K.attn_fwd(qkv, km, B, S, nh, -1, 0.125, pre)
        do = torch.randn(B * S, nh * 64, device=dev, generator=g).to(torch.bfloat16)
        dqkv = K.attn_bwd(qkv, out, do, lse, km, B, S, nh, -1, 0.125, prescaled=pre)
        torch.cuda.synchronize()
        h = hashlib.sha256(dqkv.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"B={B} S={S} nh={nh} lens={lens} prescaled={pre} group={grp}: {h}  finite={bool(torch.isfinite(dqkv.float()).all())}")


if __name__ == "__main__":
    main()
