#!/usr/bin/env python3
"""Where do the waves of the fused global attention backward wait?  Needs a -DCM3P_FTRACE=1 build of csrc/attention_bwd_fused.hip linked into a
library given by CM3P_HIP_LIB (see tools/ubench/attn_bwd_wait.sh).  Per wave: cycles at the counted vmcnt wait (the DMA of tile t+1 not yet
landed), at the LDS drain + the tile's one barrier, and in the whole sweep; printed per tile (80 MFMAs per wave).

    CM3P_ALLOW_ABLATED_LIB=1 CM3P_HIP_LIB=... python tools/attn_bwd_trace.py [c2|c4]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402
from cm3p_amd import _lib  # noqa: E402


def main():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    B, S, nh = (16, 8192, 12) if "c4" in sys.argv else (32, 4096, 12)
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, S, 3, nh, 64, device="cuda", generator=g).to(torch.bfloat16)
    qkv[:, :, 0] *= 0.18
    out, lse = K.attn_fwd(qkv, None, B, S, nh, -1, 0.125, True)
    do = torch.randn(B * S, nh * 64, device="cuda", generator=g).to(torch.bfloat16)
    for _ in range(3):
        K.attn_bwd(qkv, out, do, lse, None, B, S, nh, -1, 0.125, prescaled=True)
    nwg = ((S + 255) // 256) * nh * B  # (upper bound over the launches of one call: each launch writes its first blocks)
    buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
    assert lib.cm3p_debug_set_bwdf_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    K.attn_bwd(qkv, out, do, lse, None, B, S, nh, -1, 0.125, prescaled=True)
    torch.cuda.synchronize()
    lib.cm3p_debug_set_bwdf_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
    t = t[t[:, 3] > 0]  # (the LAST launch of the call wrote these: an adding launch unless the sequence has one key block group)
    tiles = t[:, 3]
    print(f"{len(t)} waves, {int(tiles[0])} tiles each; per tile (80 MFMAs = 2560 matrix cycles): sweep {np.mean(t[:, 2] / tiles):7.0f} cycles,"
          f" at the counted vmcnt wait {np.mean(t[:, 0] / tiles):6.0f} (max wave {np.max(t[:, 0] / tiles):.0f}),"
          f" at the LDS drain + barrier {np.mean(t[:, 1] / tiles):6.0f} (max wave {np.max(t[:, 1] / tiles):.0f})")
    w = t.reshape(-1, 4, 4)  # [workgroup][wave][field]
    print("   per wave of a workgroup (mean over workgroups): vmcnt wait " + " ".join(f"{x:6.0f}" for x in (w[:, :, 0] / w[:, :, 3]).mean(axis=0)) +
          "   drain + barrier " + " ".join(f"{x:6.0f}" for x in (w[:, :, 1] / w[:, :, 3]).mean(axis=0)))


if __name__ == "__main__":
    main()
