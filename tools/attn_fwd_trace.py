#!/usr/bin/env python3
"""Where does a wave of the pipelined global forward spend its cycles?  Needs the trace build of the library:
    hipcc ... -fno-slp-vectorize -DCM3P_GTRACE=1 -c cm3p_amd/csrc/attention_fwd.hip -o _ab/fwd_trace.o   (linked like cm3p_amd/build.py into _ab/libcm3p_trace.so)
    CM3P_ALLOW_ABLATED_LIB=1 CM3P_HIP_LIB=$PWD/_ab/libcm3p_trace.so python tools/attn_fwd_trace.py [c2|c4] [mask] [fine (a -DCM3P_GTRACE=2 build)] [clock (any trace build; -DCM3P_GTRACE=3 has no stamp inside the sweep)]
Every wave sums the shader cycles of six regions of its sweep (csrc/attention_fwd.hip: GT_STAMP); printed per period (16 MFMAs)."""
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
    mask = torch.ones(B, S, dtype=torch.uint8, device="cuda") if "mask" in sys.argv else None
    warm = 800 if "clock" in sys.argv else 3  # (clock: ~2 s of back-to-back launches first, MI355X_MICROARCH.md "DVFS give-back" item 6)
    for _ in range(warm):
        K.attn_fwd(qkv, mask, B, S, nh, -1, 0.125, True)
    nwg = ((S + 511) // 512) * nh * B
    buf = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device="cuda")
    assert lib.cm3p_debug_set_fwd_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    K.attn_fwd(qkv, mask, B, S, nh, -1, 0.125, True)
    torch.cuda.synchronize()
    lib.cm3p_debug_set_fwd_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(nwg * 4, 8).astype(np.float64)
    ghz = t[:, :6].sum(axis=1) / np.maximum(t[:, 7], 1.0) * 0.1  # (s_memrealtime counts at 100 MHz)
    print(f"in-kernel clock (cycle total / s_memrealtime): median {np.median(ghz):.3f} GHz over {len(ghz)} waves (5 % .. 95 %: {np.percentile(ghz, 5):.3f} .. {np.percentile(ghz, 95):.3f})")
    if "clock" in sys.argv:
        return
    tiles = t[:, 6]
    periods = tiles * 4
    names = ["tile top: counted wait + barrier + validity (per tile)", "gaps 0-3: tail, maximum, decision", "gaps 4-7: PV k-steps 2-3, chunk 0 / 1 exponentials",
             "gaps 8-11: QK^T block 0", "gaps 12-15: QK^T block 1", "prologue + drain + epilogue (per workgroup)"]
    print(f"{nwg} workgroups x 4 waves, {int(tiles[0])} tiles each; cycles per wave: total {t[:, :6].sum(axis=1).mean():.0f}")
    for k in (1, 2, 3, 4):
        c = t[:, k] / periods
        print(f"  {names[k]:58s} {c.mean():7.1f} cycles per period  (min wave {c.min():.1f}, max {c.max():.1f})")
    c = t[:, 0] / tiles
    print(f"  {names[0]:58s} {c.mean():7.1f} cycles per tile = {c.mean() / 4:.1f} per period")
    print(f"  {names[5]:58s} {t[:, 5].mean():7.0f} cycles = {t[:, 5].mean() / 2.1e3:.1f} us at 2.1 GHz")
    if "fine" in sys.argv:  # (-DCM3P_GTRACE=2: gap 0, gap 1, gap 2, gap 3 up to the branch, the branch itself, the rest)
        for k, nm in enumerate(["gap 0", "gap 1", "gap 2", "gap 3 up to the branch", "the reference-move branch (not taken)"]):
            print(f"  fine: {nm:40s} {(t[:, k] / periods).mean():7.1f} cycles per period")
    per = (t[:, 1:5].sum(axis=1) / periods).mean() + c.mean() / 4
    print(f"  per period in all: {per:.1f} cycles (16 MFMAs = 512)")


if __name__ == "__main__":
    main()
