#!/usr/bin/env python3
"""Can the weight-gradient GEMMs of the backward chain run on a FIXED slice of every XCD's CUs while the chain itself (input-gradient
GEMMs, attention backward, LayerNorm / GeGLU backward) runs on the rest?  The chain's streaming kernels are HBM-bound and leave the
matrix cores idle; a weight-gradient GEMM that owns k CUs per XCD would use that time.

r03's `overlap_cumask.py` masked CUs by `i % 16`, which (mask bit i = CU i // 8 of XCD i % 8) thinned two XCDs only; the dispatcher
deals workgroups to XCDs round-robin, so those XCDs became the whole chip's tail.  Here every XCD gives up the same k CUs.

One "layer" = the kernels of _EncoderLayerFn.backward at C2 shapes (T = 32 x 4096, H 768, I 1152), sliding-window or global attention.
Timed: `--layers` layers back to back, (a) everything on one stream on the whole chip, (b) chain on the big mask, weight gradients on the
small mask (each waits for the event of the kernel that produced its operand; the two streams join only at the end).

    python tools/partition_probe.py [--layers 6] [--iters 5] [--k 4 6 8]
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
    words = (ctypes.c_uint32 * 8)()
    n = 0
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
            n += 1
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    _MASKED.append(h)
    return torch.cuda.ExternalStream(h.value), n


_MASKED: list = []  # raw handles of every CU-masked stream: destroyed before the interpreter exits (see destroy_masked_streams)


def destroy_masked_streams():
    """r04: `rocprofv3 --kernel-trace -- python3 tools/partition_probe.py --only chain` died with SIGSEGV in __cxa_finalize AFTER the profiler
    had written its files (gpurun_out/r4_partition_prof.log).  The streams of hipExtStreamCreateWithCUMask were never destroyed: torch's
    ExternalStream does not own its handle, so the masked HSA queues were still alive when libamdhip64's static destructors ran, behind
    rocprofv3's "tool finalization" - the runtime then tears down queues the (already finalised) profiler had wrapped.  Destroying them while
    both are alive removes the crash; the un-profiled runs behind profiles/r04_partition_probe.log never hit it."""
    torch.cuda.synchronize()
    while _MASKED:
        hip.hipStreamDestroy(_MASKED.pop())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--k", type=int, nargs="+", default=[4, 6, 8])
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--S", type=int, default=4096)
    ap.add_argument("--only", choices=["serial", "chain", "side", "split"], default=None,
                    help="run just this arrangement (for a rocprofv3 --kernel-trace --stats pass per arrangement)")
    args = ap.parse_args()
    B, S, H, I, nh = args.B, args.S, 768, 1152, 12
    T = B * S
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    wq = lambda n, k: (torch.randn(n, k, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
    Wqkv, Wo, Wi, Wo2 = wq(3 * H, H), wq(H, H), wq(2 * I, H), wq(H, I)
    Wqkv_t, Wo_t, Wi_t, Wo2_t = (w.t().contiguous() for w in (Wqkv, Wo, Wi, Wo2))
    x = torch.randn(T, H, device=DEV, generator=g)
    ones = torch.ones(H, device=DEV)
    _, xn, mean, rstd = K.layernorm_fwd(x, ones, 1e-5, False, True)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, device=DEV).float() / 64))
    ang = torch.arange(S, device=DEV).float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    qkv = K.qkv_linear_rope(xn, Wqkv, cos, sin, S, False, q_scale=K.SOFTMAX_Q_SCALE)
    acts = {}
    for name, window in (("local", 64), ("global", -1)):
        o, lse = K.attn_fwd(qkv, None, B, S, nh, window, 0.125, prescaled=True)
        acts[name] = (window, o, lse)
    h = rnd(T, 2 * I)
    gact = K.geglu_fwd(h)
    gx32 = torch.randn(T, H, device=DEV, generator=g)
    gx16 = K.cast_bf16(gx32)
    torch.cuda.synchronize()
    main_s = torch.cuda.current_stream()

    marks = []  # (name, start event, end event) of every kernel call of the pass being timed (events on the stream it ran on)

    def tm(name, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        marks.append((name, e0, e1))
        return r

    def layer(kind, side, chain_done):
        """Issues one layer's backward on the current stream; weight gradients through side(fn) after the event of their operand."""
        window, o, lse = acts[kind]
        dg = tm("dgrad Wo2", lambda: K.linear_dgrad(gx16, Wo2, Wo2_t))
        side(lambda: tm("wgrad Wo2", lambda: K.linear_wgrad(gx16, gact)))
        dh = tm("geglu bwd", lambda: K.geglu_bwd(dg, h))
        dxn2 = tm("dgrad Wi", lambda: K.linear_dgrad(dh, Wi, Wi_t))
        side(lambda: tm("wgrad Wi", lambda: K.linear_wgrad(dh, xn)))
        g32, g16, _ = tm("LN bwd", lambda: K.layernorm_bwd(dxn2, x, ones, mean, rstd, gx32, True, inplace=False))
        do = tm("dgrad Wo", lambda: K.linear_dgrad(g16, Wo, Wo_t))
        side(lambda: tm("wgrad Wo", lambda: K.linear_wgrad(g16, o)))
        dqkv = tm("attn bwd " + kind, lambda: K.attn_bwd(qkv, o, do, lse, None, B, S, nh, window, 0.125, (cos, sin), False, prescaled=True))
        side(lambda: tm("wgrad Wqkv", lambda: K.linear_wgrad(dqkv, xn)))
        dxn = tm("dgrad Wqkv", lambda: K.linear_dgrad(dqkv, Wqkv, Wqkv_t))
        tm("LN bwd", lambda: K.layernorm_bwd(dxn, x, ones, mean, rstd, g32, True, inplace=False))
        chain_done()

    def per_kernel(fn):
        """One more pass of fn; -> {name: mean ms per call} from the events around every call."""
        marks.clear()
        fn()
        torch.cuda.synchronize()
        acc = {}
        for name, e0, e1 in marks:
            a = acc.setdefault(name, [0.0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += 1
        marks.clear()
        return {k: v[0] / v[1] for k, v in acc.items()}

    kinds = ["global" if i % 3 == 0 else "local" for i in range(args.layers)]

    def serial():
        for kd in kinds:
            layer(kd, lambda fn: fn(), lambda: None)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts), sum(ts) / len(ts)

    K.gemm8p_set_grid(0)
    os.environ.pop("CM3P_LN_BWD_CAP", None)
    t_ser = timed(serial) if args.only in (None, "serial") else (float("nan"), float("nan"))
    if args.only == "serial":
        return
    print(f"{args.layers} layers, one stream, whole chip: min {t_ser[0]:.3f} ms  mean {t_ser[1]:.3f} ms", flush=True)
    pk_ser = per_kernel(serial)

    for k in args.k:
        s_small, n_small = masked_stream(lambda i: (i // 8) >= 32 - k)
        s_big, n_big = masked_stream(lambda i: (i // 8) < 32 - k)

        def grid(n):
            K.gemm8p_set_grid(n)
            os.environ["CM3P_LN_BWD_CAP"] = str(n * 4)

        def split():
            s_big.wait_stream(main_s)
            s_small.wait_stream(main_s)
            with torch.cuda.stream(s_big):
                def side(fn):
                    ev = torch.cuda.Event()
                    ev.record(s_big)
                    grid(n_small)
                    with torch.cuda.stream(s_small):
                        s_small.wait_event(ev)
                        fn()
                    grid(n_big)
                grid(n_big)
                for kd in kinds:
                    layer(kd, side, lambda: None)
            main_s.wait_stream(s_big)
            main_s.wait_stream(s_small)

        def chain_only():
            s_big.wait_stream(main_s)
            with torch.cuda.stream(s_big):
                grid(n_big)
                for kd in kinds:
                    layer(kd, lambda fn: None, lambda: None)
            main_s.wait_stream(s_big)

        def side_only():
            s_small.wait_stream(main_s)
            with torch.cuda.stream(s_small):
                grid(n_small)
                for kd in kinds:
                    o = acts[kd][1]
                    K.linear_wgrad(gx16, gact), K.linear_wgrad(h, xn), K.linear_wgrad(gx16, o), K.linear_wgrad(qkv, xn)
            main_s.wait_stream(s_small)

        nan = (float("nan"), float("nan"))
        t_c = timed(chain_only) if args.only in (None, "chain") else nan
        t_s = timed(side_only) if args.only in (None, "side") else nan
        t_p = timed(split) if args.only in (None, "split") else nan
        print(f"k = {k}: chain on {n_big} CUs alone {t_c[0]:.3f} ms, weight gradients on {n_small} CUs alone {t_s[0]:.3f} ms, "
              f"both at once min {t_p[0]:.3f} mean {t_p[1]:.3f} ms  (one stream, whole chip: {t_ser[0]:.3f})", flush=True)
        pk_c, pk_p = per_kernel(chain_only), per_kernel(split)
        print(f"    {'ms per call':14s} {'whole chip':>10s} {'big mask':>10s} {'beside / as wgrad':>18s}")
        for name in pk_ser:
            print(f"    {name:14s} {pk_ser[name]:10.3f} {pk_c.get(name, float('nan')):10.3f} {pk_p.get(name, float('nan')):18.3f}", flush=True)
        K.gemm8p_set_grid(0)
        os.environ.pop("CM3P_LN_BWD_CAP", None)


if __name__ == "__main__":
    try:
        main()
    finally:
        destroy_masked_streams()
