#!/usr/bin/env python3
"""Where does a workgroup of the sliding-window dQ sweep spend its time?  Needs the trace build of the library:
    hipcc ... -DCM3P_BAND_TRACE=1 -c cm3p_amd/csrc/attention.hip -o _ab/attention.trace.o   (and link like cm3p_amd/build.py into _ab/libcm3p_trace.so)
    CM3P_HIP_LIB=$PWD/_ab/libcm3p_trace.so python tools/band_trace.py
Thread 0 of every workgroup records the 100 MHz wall clock at entry, after the prologue's requests, when each tile is ready and swept,
after the last barrier and after the stores are issued (csrc/attention.hip: BAND_T)."""
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
    B, S, nh = 32, 4096, 12
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = (torch.randn(B * S, 3 * nh * 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    do = (torch.randn(B * S, nh * 64, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, device="cuda").float() / 64))
    ang = torch.arange(S, device="cuda").float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    o, lse = K.attn_fwd(qkv, None, B, S, nh, 64, 0.125, prescaled=True)
    nwg = (S // 128) * nh * B
    for _ in range(2):
        K.attn_bwd(qkv, o, do, lse, None, B, S, nh, 64, 0.125, (cos, sin), False, prescaled=True)
    buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
    assert lib.cm3p_debug_set_band_trace(ctypes.c_void_p(buf.data_ptr())) == 0
    torch.cuda.synchronize()
    K.attn_bwd(qkv, o, do, lse, None, B, S, nh, 64, 0.125, (cos, sin), False, prescaled=True)
    torch.cuda.synchronize()
    lib.cm3p_debug_set_band_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(nwg, 16).astype(np.float64) / 100.0  # microseconds
    t0 = t[:, 0].min()
    life = t[:, 15] - t[:, 0]
    print(f"{nwg} workgroups; kernel span {t[:, 15].max() - t0:.1f} us; workgroup life: mean {life.mean():.2f} us, median {np.median(life):.2f}, p90 {np.percentile(life, 90):.2f}")
    print(f"  resident workgroups on average: {life.sum() / (t[:, 15].max() - t0):.0f} (2 per CU = 512)")
    ph = [("entry -> prologue requests issued", 0, 1), ("prologue issued -> tile 0 ready", 1, 2)]
    ntile = int(((t[:, 2:12:2] > 0).sum(axis=1)).max())
    for i in range(ntile):
        ph.append((f"tile {i} ready -> swept", 2 + 2 * i, 3 + 2 * i))
        if i + 1 < ntile:
            ph.append((f"tile {i} swept -> tile {i + 1} ready", 3 + 2 * i, 4 + 2 * i))
    for name, a, b in ph:
        ok = (t[:, a] > 0) & (t[:, b] > 0)
        d = (t[ok, b] - t[ok, a])
        print(f"  {name:36s} n {ok.sum():6d}  mean {d.mean():6.2f} us  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}")
    last = np.where(t[:, 3:13:2] > 0, t[:, 3:13:2], 0).max(axis=1)
    d = t[:, 14] - last
    print(f"  {'last tile swept -> final barrier':36s} mean {d.mean():6.2f} us  median {np.median(d):6.2f}")
    d = t[:, 15] - t[:, 14]
    print(f"  {'rotary epilogue + stores issued':36s} mean {d.mean():6.2f} us  median {np.median(d):6.2f}")


if __name__ == "__main__":
    main()
