#!/usr/bin/env python3
"""Bounds audit of every LDS-DMA staging stream (r03 verdict item 7; csrc/common.h CM3P_DMA_AUDIT).

Run with the audit twin of the library:  CM3P_HIP_LIB=cm3p_amd/csrc/libcm3p_hip_audit.so CM3P_ALLOW_ABLATED_LIB=1 python tools/dma_audit.py
For each case the recording buffer is reset, ONE C-ABI call runs, and every operand id the kernels reported must lie inside the tensor
the call was given for it: [data_ptr, data_ptr + nbytes).  Prints one JSON object per case and a final {"failed": n}; exit code 1 if
any address was outside.  Cases: the GEMM harness's edge shapes (extents that are not tile multiples, a single k-tile, the last work
item of a persistent workgroup, split-K tails, k-strided operands, the RoPE / GeGLU / batched instances) on both big-shape kernels,
the sliding-window backward and the fused global backward at odd lengths, with key masks, and on packed sequences with short tails.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import _lib  # noqa: E402
from cm3p_amd._lib import call, ptr, query, stream  # noqa: E402

DEV = "cuda"
NAMES = {0: "GEMM A", 1: "GEMM B", 2: "tile matrix 0", 3: "tile matrix 1", 4: "row statistics / mask", 5: "row statistics 2"}
g = torch.Generator(device=DEV).manual_seed(0)
bf = lambda *s: (torch.randn(*s, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
f32 = lambda *s: torch.randn(*s, device=DEV, generator=g)
buf = torch.zeros((8, 2), dtype=torch.int64, device=DEV)
failed = 0


def audited(name, fn, expect):
    """expect: {operand id: tensor}.  Runs fn() with a fresh recording buffer."""
    global failed
    buf[:, 0] = -1  # = 2^64 - 1 unsigned
    buf[:, 1] = 0
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    rec = buf.cpu().tolist()
    out = {"case": name, "operands": {}, "ok": True}
    for i, (lo, hi) in enumerate(rec):
        if hi == 0 and lo == -1:
            continue
        lo &= (1 << 64) - 1
        hi &= (1 << 64) - 1
        t = expect.get(i)
        if t is None:
            out["operands"][NAMES.get(i, str(i))] = "recorded but not expected"
            out["ok"] = False
            continue
        base, end = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
        inside = base <= lo and hi < end
        out["operands"][NAMES.get(i, str(i))] = {"first_byte_minus_base": lo - base, "bytes": end - base, "last_byte_minus_base": hi - base, "inside": inside}
        out["ok"] &= inside
    for i in expect:
        if NAMES.get(i, str(i)) not in out["operands"]:
            out["operands"][NAMES.get(i, str(i))] = "not recorded"  # (the case did not reach an LDS-DMA kernel: say so)
            out["ok"] = False
    failed += not out["ok"]
    print(json.dumps(out), flush=True)


def gemm_cases():
    for impl in ("8p", "256"):
        os.environ["CM3P_GEMM_IMPL"] = impl
        # forward orientation (both operands k-contiguous): edge tiles in both extents, one k-tile, more work items than workgroups
        for (M, N, Kd) in ((8200, 2312, 128), (256 * 30, 256 * 7, 64), (25600, 768, 192), (65536 + 64, 1152, 768), (70000, 776, 128)):
            a, w, c = bf(M, Kd), bf(N, Kd), torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
            audited(f"gemm {impl} fwd bf16 [{M}x{N}x{Kd}]",
                    lambda: call("cm3p_gemm_bf16", ptr(a), ptr(w), ptr(c), None, M, N, Kd, Kd, Kd, N, 1, 1, 0, 1, None, stream()), {0: a, 1: w})
        M, N, Kd = 8200, 2312, 128
        a, w = bf(M, Kd), bf(N, Kd)
        r, c32 = f32(M, N), torch.empty((M, N), dtype=torch.float32, device=DEV)
        audited(f"gemm {impl} fwd fp32+resid [{M}x{N}x{Kd}]",
                lambda: call("cm3p_gemm_bf16", ptr(a), ptr(w), ptr(c32), ptr(r), M, N, Kd, Kd, Kd, N, 1, 1, 2, 1, None, stream()), {0: a, 1: w})
        # input gradient through the k-strided weight: dx[M, N] = dy[M, K] W[K, N]  (b_kc = 0, ldb = N)
        for (M, N, Kd) in ((16400, 776, 2304), (40960, 1152, 768)):
            dy, w, c = bf(M, Kd), bf(Kd, N), torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
            audited(f"gemm {impl} dgrad (B k-strided) [{M}x{N}x{Kd}]",
                    lambda: call("cm3p_gemm_bf16", ptr(dy), ptr(w), ptr(c), None, M, N, Kd, Kd, N, N, 1, 0, 0, 1, None, stream()), {0: dy, 1: w})
        # weight gradient: dW[M, N] = dy^T x over T tokens, both operands k-strided, split-K with a short last split
        for (M, N, T) in ((2304, 768, 8192 + 576), (776, 1160, 16384)):
            dy, x = bf(T, M), bf(T, N)
            splits = max(1, query("cm3p_gemm_wgrad_splits", M, N, T))
            ws = torch.empty((splits, M, N), dtype=torch.float32, device=DEV)
            c32 = torch.empty((M, N), dtype=torch.float32, device=DEV)
            audited(f"gemm {impl} wgrad (both k-strided, split-K {splits}) [{M}x{N}x{T}]",
                    lambda: call("cm3p_gemm_bf16", ptr(dy), ptr(x), ptr(c32), None, M, N, T, M, N, N, 0, 0, 1, splits, ptr(ws), stream()), {0: dy, 1: x})
        # Wqkv + RoPE (per-token tables: M need not be a multiple of S) and Wi + GeGLU
        M, H = 8200, 768
        x, w, qkv = bf(M, H), bf(3 * H, H), torch.empty((M, 3 * H), dtype=torch.bfloat16, device=DEV)
        cos, sin = f32(M, 32), f32(M, 32)
        audited(f"gemm {impl} Wqkv + RoPE [{M}x{3 * H}x{H}]",
                lambda: call("cm3p_qkv_gemm_rope", ptr(x), ptr(w), ptr(qkv), M, 3 * H, H, ptr(cos), ptr(sin), M, 1, 2 * H, 0.18, stream()), {0: x, 1: w})
        if impl == "8p":
            I = 1152
            wi, aout = bf(2 * I, H), torch.empty((M, I), dtype=torch.bfloat16, device=DEV)
            audited(f"gemm 8p Wi + GeGLU [{M}x{2 * I}x{H}]", lambda: call("cm3p_gemm_geglu", ptr(x), ptr(wi), ptr(aout), M, I, H, stream()), {0: x, 1: wi})
            # strided batch (the Muon step's Newton-Schulz products): X^T X of 44 [2304 x 768] matrices, and a X + B X with a residual
            n, R, Cc = 44, 2304, 768
            X = bf(n, R, Cc)
            A = torch.empty((n, Cc, Cc), dtype=torch.bfloat16, device=DEV)
            audited(f"gemm 8p batched X^T X [{n} x {Cc}x{Cc}x{R}]",
                    lambda: call("cm3p_gemm_bf16_batched", ptr(X), ptr(X), ptr(A), None, n, Cc, Cc, R, Cc, Cc, Cc, R * Cc, R * Cc, Cc * Cc, 0, 0, 0, 1.0, 0.0, stream()),
                    {0: X, 1: X})
    os.environ.pop("CM3P_GEMM_IMPL", None)


def attention_cases():
    nh, scale = 2, 0.125
    for (B, S, masked) in ((1, 1000, False), (2, 1000, True), (1, 4097, True), (3, 65, False), (1, 64, True)):
        qkv, o, do = bf(B, S, 3, nh, 64), bf(B * S, nh * 64), bf(B * S, nh * 64)
        lse = f32(B, nh, S)
        delta, dqkv = torch.zeros_like(lse), torch.empty_like(qkv)
        mask = None
        if masked:
            mask = torch.ones((B, S), dtype=torch.uint8, device=DEV)
            mask[:, S - S // 3:] = 0
        # the pipelined global forward (attention_fwd.hip): K and V tiles, three tiles past the sequence's last one, and the mask bytes
        fo, flse = torch.empty_like(o), torch.empty_like(lse)
        audited(f"global forward (pipelined) B={B} S={S} mask={masked}",
                lambda: call("cm3p_attn_fwd", ptr(qkv), ptr(fo), ptr(flse), ptr(mask, torch.uint8), B, S, nh, -1, scale, 1, stream()),
                {2: qkv, 3: qkv, **({4: mask} if masked else {})})
        for stage, expect in ((1, {2: qkv, 3: qkv, **({4: mask} if masked else {})}), (2, {2: qkv, 3: do, 4: lse, 5: delta})):
            audited(f"sliding-window backward stage {stage} B={B} S={S} mask={masked}",
                    lambda: call("cm3p_attn_bwd", ptr(qkv), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dqkv), ptr(mask, torch.uint8), B, S, nh, 64, scale,
                                 None, None, 0, stage, 1, stream()), expect)
        ws = torch.empty(query("cm3p_attn_bwd_fused_workspace_bytes", B, S, nh), dtype=torch.uint8, device=DEV)
        call("cm3p_attn_bwd_fused", ptr(qkv), ptr(o), ptr(do), ptr(lse), ptr(dqkv), ptr(mask, torch.uint8), None, B, S, 0, nh, scale, None, None, 0, 1, 1,
             ptr(ws), ws.numel(), stream())  # prep: writes the score offsets the main kernel's statistics DMA reads
        for stage, nm in ((8, "even key blocks"), (16, "odd key blocks")):
            if stage == 16 and S <= 256:
                continue
            audited(f"fused global backward ({nm}) B={B} S={S} mask={masked}",
                    lambda: call("cm3p_attn_bwd_fused", ptr(qkv), ptr(o), ptr(do), ptr(lse), ptr(dqkv), ptr(mask, torch.uint8), None, B, S, 0, nh, scale,
                                 None, None, 0, stage, 1, ptr(ws), ws.numel(), stream()), {2: qkv, 3: do, 4: ws})
    # packed sequences: lengths that end inside a tile, a one-token sequence, a sequence shorter than the window
    lens = [700, 1, 63, 257, 1000, 40]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    total, Bv, max_s = sum(lens), len(lens), max(lens)
    qkv, o, do = bf(total, 3, nh, 64), bf(total, nh * 64), bf(total, nh * 64)
    lse = f32(nh, total)
    delta, dqkv = torch.zeros_like(lse), torch.empty_like(qkv)
    fo, flse = torch.empty_like(o), torch.empty_like(lse)
    audited(f"global forward (pipelined), packed {lens}",
            lambda: call("cm3p_attn_fwd_varlen", ptr(qkv), ptr(fo), ptr(flse), ptr(cu, torch.int32), Bv, max_s, total, nh, -1, scale, 1, stream()), {2: qkv, 3: qkv})
    for stage, expect in ((1, {2: qkv, 3: qkv}), (2, {2: qkv, 3: do, 4: lse, 5: delta})):
        audited(f"sliding-window backward stage {stage}, packed {lens}",
                lambda: call("cm3p_attn_bwd_varlen", ptr(qkv), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dqkv), ptr(cu, torch.int32), Bv, max_s, total, nh, 64,
                             scale, None, None, stage, 1, stream()), expect)
    ws = torch.empty(query("cm3p_attn_bwd_fused_workspace_bytes", Bv, max_s, nh), dtype=torch.uint8, device=DEV)
    call("cm3p_attn_bwd_fused", ptr(qkv), ptr(o), ptr(do), ptr(lse), ptr(dqkv), None, ptr(cu, torch.int32), Bv, max_s, total, nh, scale, None, None, 0, 1, 1,
         ptr(ws), ws.numel(), stream())
    for stage, nm in ((8, "even key blocks"), (16, "odd key blocks")):
        audited(f"fused global backward ({nm}), packed {lens}",
                lambda: call("cm3p_attn_bwd_fused", ptr(qkv), ptr(o), ptr(do), ptr(lse), ptr(dqkv), None, ptr(cu, torch.int32), Bv, max_s, total, nh, scale,
                             None, None, 0, stage, 1, ptr(ws), ws.numel(), stream()), {2: qkv, 3: do, 4: ws})


def main():
    lib = _lib.load()
    if not (lib.cm3p_build_ablation_flags() & 32):
        raise SystemExit("this library was built without the audit hooks: set CM3P_HIP_LIB to libcm3p_hip_audit.so (and CM3P_ALLOW_ABLATED_LIB=1)")
    rc = lib.cm3p_debug_set_dma_audit(buf.data_ptr())
    if rc != 0:
        raise SystemExit(f"cm3p_debug_set_dma_audit failed ({rc})")
    gemm_cases()
    attention_cases()
    lib.cm3p_debug_set_dma_audit(None)
    print(json.dumps({"failed": failed}), flush=True)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
