#!/usr/bin/env python3
"""Global attention forward: the pipelined kernel (attention_fwd.hip) against an fp32 restatement, and its time at the C2 / C4 shapes.

    python tools/attn_fwd_ab.py [check] [time] [--iters 20]
    CM3P_ATTN_FWD_IMPL=wave3 python tools/attn_fwd_ab.py time      # the three-waves-per-SIMD kernel (attention.hip), same shapes
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402

LN2 = math.log(2.0)


def reference(qkv, mask, B, S, nh):
    """q already carries scale * log2(e): p = 2^(q k)"""
    q, k, v = (qkv[:, :, i].float().permute(0, 2, 1, 3) for i in range(3))  # [B, nh, S, 64]
    s = (q @ k.transpose(-1, -2)) * LN2
    if mask is not None:
        s = s.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * S, nh * 64)
    return o, lse


def check():
    g = torch.Generator(device="cuda").manual_seed(1)
    worst = 0.0
    cases = [(2, 1024, 4, None), (1, 200, 2, None), (3, 64, 1, None), (2, 577, 3, None), (1, 2048, 2, "pad"), (2, 777, 2, "rand"), (1, 512, 1, "dead"),
             (1, 4096, 2, None), (1, 1, 1, None), (2, 130, 2, "pad")]
    for B, S, nh, mk in cases:
        qkv = (torch.randn(B, S, 3, nh, 64, device="cuda", generator=g) * 1.0).to(torch.bfloat16)
        qkv[:, :, 0] *= 0.35
        # rows whose maximum arrives late and large: exercises the reference move
        if S >= 512:
            qkv[:, S // 2:, 1] *= 3.0
        mask = None
        if mk == "pad":
            mask = torch.ones(B, S, dtype=torch.uint8, device="cuda")
            for b in range(B):
                mask[b, S - 37 - 100 * b:] = 0
        elif mk == "rand":
            mask = (torch.rand(B, S, device="cuda", generator=g) > 0.3).to(torch.uint8)
        elif mk == "dead":
            mask = torch.zeros(B, S, dtype=torch.uint8, device="cuda")
        out, lse = K.attn_fwd(qkv, mask, B, S, nh, -1, 0.125, True)
        torch.cuda.synchronize()
        ro, rl = reference(qkv, mask, B, S, nh)
        eo = (out.float() - ro).abs().max().item()
        fin = torch.isfinite(rl)
        el = (lse[fin] - rl[fin]).abs().max().item() if fin.any() else 0.0
        inf_ok = bool((lse[~fin] == float("inf")).all()) if (~fin).any() else True
        scale_o = ro.abs().max().item()
        print(f"B={B} S={S} nh={nh} mask={mk}: max|dO|={eo:.3e} (|O|max {scale_o:.2f}) max|dlse|={el:.3e} dead_rows_inf={inf_ok} nan={bool(torch.isnan(out.float()).any())}")
        worst = max(worst, eo / max(scale_o, 1e-6))
    print("worst relative output error", worst)
    return worst


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_shapes(iters):
    g = torch.Generator(device="cuda").manual_seed(0)
    for B, S, nh in ((32, 4096, 12), (16, 8192, 12)):
        qkv = torch.randn(B, S, 3, nh, 64, device="cuda", generator=g).to(torch.bfloat16)
        qkv[:, :, 0] *= 0.18
        ms = timeit(lambda: K.attn_fwd(qkv, None, B, S, nh, -1, 0.125, True), iters)
        fl = 4.0 * B * nh * S * S * 64
        print(f"impl={os.environ.get('CM3P_ATTN_FWD_IMPL', 'pipe')} B={B} S={S}: {ms:.3f} ms = {fl / ms / 1e9:.0f} TFLOP/s = {fl / ms / 1e9 / 2500:.3f} of 2.5 PF")
        mask = torch.ones(B, S, dtype=torch.uint8, device="cuda")
        ms = timeit(lambda: K.attn_fwd(qkv, mask, B, S, nh, -1, 0.125, True), iters)
        print(f"   with an all-ones key mask: {ms:.3f} ms")


if __name__ == "__main__":
    what = [a for a in sys.argv[1:] if not a.startswith("--")] or ["check", "time"]
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
    if "check" in what:
        check()
    if "time" in what:
        time_shapes(iters)
