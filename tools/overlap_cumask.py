#!/usr/bin/env python3
"""Second form of tools/overlap_ab.py: the two kernels run on streams created with disjoint CU masks (hipExtStreamCreateWithCUMask),
so that the dispatcher cannot scatter the streaming kernel's workgroups over the CUs the GEMM's workgroups need whole.

    python tools/overlap_cumask.py [--iters 20]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K  # noqa: E402

DEV = "cuda"
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(pred):
    """torch stream whose kernels may only use the CUs i with pred(i) (mask bit i; 256 CUs = 8 words)."""
    words = (ctypes.c_uint32 * 8)()
    n = 0
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
            n += 1
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value), n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    T, H, N = 32 * 4096, 768, 2304
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)
    dy, wgt = rnd(T, N), rnd(N, H) * 0.02               # dgrad: dx[T, H] = dy W (1536 work items)
    a = rnd(T, H)
    x = torch.randn(T, H, device=DEV, generator=g)
    w = torch.ones(H, device=DEV)
    dn = rnd(T, H)
    dres = torch.randn(T, H, device=DEV, generator=g)
    _, _, mean, rstd = K.layernorm_fwd(x, w, 1e-5, False, True)
    main_s = torch.cuda.current_stream()
    split = [9]
    gemms = {"dgrad K=2304 (1536 items)": lambda: K.linear_dgrad(dy, wgt),
             "wgrad (27 tiles x split)": lambda: K.gemm(dy, a, N, H, T, False, False, K.EPI_F32, split_k=split[0])}

    def ln():
        return K.layernorm_bwd(dn, x, w, mean, rstd, dres, True, inplace=False)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    K.gemm8p_set_grid(0)
    os.environ.pop("CM3P_LN_BWD_CAP", None)
    t_l = timed(ln)
    print(f"LN backward alone, whole chip: {t_l:.3f} ms", flush=True)
    for small_per16 in (2, 3, 4):  # CUs of every 16 given to the streaming kernel
        s_small, n_small = masked_stream(lambda i: i % 16 >= 16 - small_per16)
        s_big, n_big = masked_stream(lambda i: i % 16 < 16 - small_per16)
        K.gemm8p_set_grid(n_big)
        os.environ["CM3P_LN_BWD_CAP"] = str(n_small * 4)
        split[0] = max(8, n_big // 27)

        def on(stream, fn):
            def run():
                stream.wait_stream(main_s)
                with torch.cuda.stream(stream):
                    fn()
                main_s.wait_stream(stream)
            return run

        t_ln_small = timed(on(s_small, ln))
        for name, gemm in gemms.items():
            K.gemm8p_set_grid(0)
            t_full = timed(gemm)
            K.gemm8p_set_grid(n_big)
            t_big = timed(on(s_big, gemm))

            def conc():
                s_big.wait_stream(main_s)
                s_small.wait_stream(main_s)
                with torch.cuda.stream(s_big):
                    gemm()
                with torch.cuda.stream(s_small):
                    ln()
                main_s.wait_stream(s_big)
                main_s.wait_stream(s_small)

            t_c = timed(conc)
            print(f"{n_big} + {n_small} CUs: {name}: whole chip {t_full:.3f} ms, on {n_big} CUs {t_big:.3f}; LN backward on {n_small} CUs "
                  f"{t_ln_small:.3f}; concurrent {t_c:.3f} ms vs back to back on the whole chip {t_full + t_l:.3f}", flush=True)


if __name__ == "__main__":
    main()
