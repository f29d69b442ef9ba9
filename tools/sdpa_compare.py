#!/usr/bin/env python3
"""Global attention (head_dim 64, no mask) forward and backward: this repo's kernels beside the vendor flash attention that
torch.nn.functional.scaled_dot_product_attention dispatches to on ROCm.  A calibration only - the product never calls SDPA.

    python tools/sdpa_compare.py [--iters 10]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402

DEV = "cuda"


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    nh, hd = 12, 64
    g = torch.Generator(device=DEV).manual_seed(0)
    for B, S in ((32, 4096), (16, 8192)):
        qkv = (torch.randn(B * S, 3 * nh * hd, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
        do = (torch.randn(B * S, nh * hd, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
        fl_f = 4.0 * B * nh * S * S * hd / 1e9
        o, lse = K.attn_fwd(qkv, None, B, S, nh, -1, 0.125, prescaled=False)
        t_f = timed(lambda: K.attn_fwd(qkv, None, B, S, nh, -1, 0.125, prescaled=False), args.iters)
        t_b = timed(lambda: K.attn_bwd(qkv, o, do, lse, None, B, S, nh, -1, 0.125, None, False, prescaled=False), args.iters)
        print(f"B {B} S {S}: ours forward {t_f:.3f} ms ({fl_f / t_f:5.0f} TFLOP/s on 4 B nh S^2 d)  backward {t_b:.3f} ms "
              f"({2.5 * fl_f / t_b:5.0f} on 2.5 x forward, {2.0 * fl_f / t_b:5.0f} on the 2 x of SURVEY 8d)", flush=True)
        q4 = qkv.view(B, S, 3, nh, hd)
        q, k, v = (q4[:, :, j].transpose(1, 2).contiguous().requires_grad_(True) for j in range(3))
        do4 = do.view(B, S, nh, hd).transpose(1, 2).contiguous()
        for backend in ("FLASH_ATTENTION", "EFFICIENT_ATTENTION"):
            try:
                from torch.nn.attention import SDPBackend, sdpa_kernel
                with sdpa_kernel(getattr(SDPBackend, backend)):
                    out = F.scaled_dot_product_attention(q, k, v)
                    t_lf = timed(lambda: F.scaled_dot_product_attention(q, k, v), args.iters)

                    def fb():
                        out = F.scaled_dot_product_attention(q, k, v)
                        torch.autograd.grad(out, (q, k, v), do4)

                    t_lfb = timed(fb, args.iters)
                err = (out.transpose(1, 2).reshape(B * S, nh * hd).float() - o.float()).abs().max().item()
                print(f"          library {backend}: forward {t_lf:.3f} ms ({fl_f / t_lf:5.0f})  forward + backward {t_lfb:.3f} ms -> backward "
                      f"{t_lfb - t_lf:.3f} ms ({2.5 * fl_f / (t_lfb - t_lf):5.0f} on 2.5 x forward)   max |library - ours| {err:.3g}", flush=True)
            except Exception as e:  # noqa: BLE001
                print(f"          library {backend}: not available here ({type(e).__name__}: {str(e)[:120]})", flush=True)


if __name__ == "__main__":
    main()
