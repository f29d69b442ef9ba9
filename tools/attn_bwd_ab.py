#!/usr/bin/env python3
"""One-process A/B of the attention backward kernels at the C2 / C4 global-layer shapes (development aid).

arm "fused": window = -1                         -> attention_bwd_fused.hip (prep + five-product kernel + slab reduce)
arm "pair":  window = -1, CM3P_ATTN_BWD_FUSED=0  -> attention_bwd.hip (attn_bwd_dq3_kernel + attn_bwd_dkv3_kernel, seven products)
arm "band":  window = S                          -> attention.hip's band kernels with every key inside the window (the same
                                                    mathematics), i.e. the kernels that served the global layers in round 1
Interleaved rounds, per-stage HIP-event timing, and a check that the arms agree (and against an fp32 torch reference with --ref).

    python tools/attn_bwd_ab.py [--seq 4096] [--batch 32] [--rounds 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import _lib  # noqa: E402
from cm3p_amd import kernels as K  # noqa: E402

DEV = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--padded", action="store_true", help="right-pad every second row to 3/4 of the length (key mask path)")
    ap.add_argument("--arms", default="fused,pair,band")
    ap.add_argument("--window", type=int, default=-1, help="sliding-window layers: |q - k| <= window (arms fused / band only)")
    ap.add_argument("--plain-q", action="store_true", help="q_prescaled = 0 kernels (the model runs the prescaled ones)")
    args = ap.parse_args()
    B, S, nh = args.batch, args.seq, 12
    g = torch.Generator(device=DEV).manual_seed(0)
    qkv = torch.randn(B, S, 3, nh, 64, device=DEV, generator=g)
    qkv[:, :, 0] *= 0.125 * 1.4426950408889634 if not args.plain_q else 1.0  # (prescaled q: keep the logits of the plain case)
    qkv = qkv.to(torch.bfloat16)
    do = (torch.randn(B * S, nh * 64, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    mask = None
    if args.padded:
        mask = torch.ones(B, S, dtype=torch.uint8, device=DEV)
        mask[1::2, (3 * S) // 4:] = 0
    pos = torch.arange(S, device=DEV).unsqueeze(0)
    inv = 1.0 / (160000.0 ** (torch.arange(0, 64, 2, dtype=torch.float) / 64)).to(DEV)
    rope = K.rope_table(pos.contiguous(), inv)
    pre = not args.plain_q
    out, lse = K.attn_fwd(qkv, mask, B, S, nh, args.window, 0.125, pre)
    all_arms = {"fused": (-1, "1"), "pair": (-1, "0"), "band": (S, "0")}
    if args.window >= 0:
        all_arms = {"fused": (args.window, "1"), "band": (args.window, "0")}
    arms = {k: all_arms[k] for k in args.arms.split(",") if k in all_arms}

    def run(name):
        w, fused = arms[name]
        os.environ["CM3P_ATTN_BWD_FUSED"] = fused
        return K.attn_bwd(qkv, out, do, lse, mask, B, S, nh, w, 0.125, rope, False, pre)

    res = {name: run(name) for name in arms}
    torch.cuda.synchronize()
    base = "band" if "band" in arms else list(arms)[-1]
    b = res[base].float()
    for name in arms:
        if name == base:
            continue
        a = res[name].float()
        for i, part in enumerate(("dq", "dk", "dv")):
            x, y = a[:, :, i], b[:, :, i]
            print(f"{name} vs {base} {part}: rel-L2 {((x - y).norm() / y.norm()).item():.3e}  max|diff| {(x - y).abs().max().item():.3e}  finite {bool(torch.isfinite(x).all())}")
    times = {k: {} for k in arms}
    for r in range(args.rounds):
        for name in arms:
            _lib.profile_begin()
            run(name)
            for tag, (n, ms, work) in _lib.profile_end().items():
                key = tag.split(" [")[0]
                key = key if key.startswith("attn_bwd_fused_kernel") else key.split("<")[0]  # (the fused kernel's two launches stay apart)
                times[name].setdefault(key, []).append(ms)
    fl = 2.0 * B * nh * S * (S if args.window < 0 else min(S, 2 * args.window + 1)) * 64
    for name in arms:
        tot = 0.0
        for tag, v in times[name].items():
            v = sorted(v)
            med = v[len(v) // 2]
            tot += med
            prods = 2.5 if tag.startswith("attn_bwd_fused_kernel") else \
                {"attn_bwd_prep_kernel": 0, "attn_bwd_dq_reduce_kernel": 0}.get(tag, 3 if "dq" in tag else 4)  # (each fused launch: half of five)
            print(f"{name:7s} {tag:36s} median {med:7.3f} ms  min {v[0]:7.3f}  executed {prods * fl / med / 1e9:7.1f} TF/s ({prods * fl / med / 1e9 / 2500:.1%} of peak)")
        print(f"{name:7s} total {tot:7.3f} ms   section-8d basis (4 products) {4 * fl / tot / 1e9:7.1f} TF/s = {4 * fl / tot / 1e9 / 2500:.1%} of peak")


if __name__ == "__main__":
    main()
