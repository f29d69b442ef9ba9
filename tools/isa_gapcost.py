#!/usr/bin/env python3
"""Issue-cost walk of a hand-placed MFMA stream (one wave per SIMD): splits a kernel's main loop at its MFMAs and prices what sits in
every gap with the measured per-instruction issue costs (MI355X_MICROARCH.md 'vector-instruction ISSUE cost'; v_cvt_pk_bf16_f32 = 8 as
tools/ubench/mfma_cvt_dep.hip measured it - NOT the 4-5 of a lone pack).  A gap runs max(MFMA cycles, its issue sum): the matrix pipe does
not buy back what a heavy gap overran.  Prints the gaps and the total, i.e. what the placement costs against a perfectly even one.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o k.s cm3p_amd/csrc/attention_bwd_fused.hip
    python tools/isa_gapcost.py k.s attn_bwd_fused_kernelILb1ELb0E 480 [-v]
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import isa_check  # noqa: E402

COST = [  # (regex, cycles)
    (r"v_mfma_f32_32x32x16", 8.0), (r"v_mfma_f32_16x16x32", 8.0),
    (r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", 8.0), (r"v_cvt_pk_bf16_f32", 8.0),
    (r"v_accvgpr", 4.0), (r"v_", 4.0),
    (r"ds_read_b64_tr_b16|ds_read_b128|ds_read_b64|ds_read_b32|ds_read2", 1.5), (r"ds_write_b64", 6.0), (r"ds_write2_b64|ds_write_b128", 13.0), (r"ds_write", 4.0),
    (r"global_load_lds", 30.0), (r"global_store|global_atomic|global_load|buffer_", 8.0),
    (r"s_barrier", 8.0), (r"s_waitcnt", 1.0), (r"s_", 1.0),
]


def cost(ins: str) -> float:
    op = ins.split()[0]
    if op == "s_nop":
        return 1.0 + float(ins.split()[1])
    for rx, c in COST:
        if re.match(rx, op):
            return c
    return 1.0


def gap_costs(isa: str, kernel: str, n_mfma: int, which: int = 0):
    """-> (gaps, cycles per MFMA): gaps = [(issue cycles, [opcodes])] of the `which`-th loop of `kernel` with n_mfma MFMAs, split at its MFMAs
    (hot path: blocks that forward branches jump over are skipped)."""
    body = next(iter(isa_check.kernel_bodies(isa, kernel).values()))
    seg = [seg for _, seg in isa_check.loops(body) if seg.count("v_mfma") == n_mfma][which]
    lines = body.split("\n")
    first = body[:body.index(seg)].count("\n")
    hot = [l.strip().split(";")[0].strip() for _, l in isa_check.hot_path(lines, first, first + seg.count("\n")) if l.startswith("\t") and l.strip() and not l.strip().startswith(";")]
    mfma_cyc = 16.0 if "16x16x32" in seg and "32x32x16" not in seg else 32.0
    gaps, cur, names = [], 0.0, []
    for ins in hot:
        if ins.startswith("v_mfma") and cur > 0:
            gaps.append((cur, names))
            cur, names = 0.0, []
        cur += cost(ins)
        names.append(ins.split()[0])
    gaps.append((cur, names))
    return gaps, mfma_cyc


def main():
    args = [a for a in sys.argv[1:] if a != "-v"]
    verbose = "-v" in sys.argv
    gaps, mfma_cyc = gap_costs(open(args[0]).read(), args[1], int(args[2]))
    tot_issue = sum(g for g, _ in gaps)
    tot_run = sum(max(mfma_cyc, g) for g, _ in gaps)
    print(f"{len(gaps)} MFMA gaps; issue sum {tot_issue:.0f} = {tot_issue / len(gaps):.1f} per MFMA; as placed sum(max({mfma_cyc:.0f}, gap)) = {tot_run:.0f} = {tot_run / len(gaps):.1f} per MFMA; "
          f"evenly spread {max(mfma_cyc, tot_issue / len(gaps)):.1f}")
    hist = {}
    for g, _ in gaps:
        b = int(g // 8) * 8
        hist[b] = hist.get(b, 0) + 1
    print("gap histogram (cycles: count): " + "  ".join(f"{b}-{b + 7}: {hist[b]}" for b in sorted(hist)))
    if verbose:
        import collections
        for i, (g, nm) in enumerate(gaps):
            c = collections.Counter(n for n in nm if not n.startswith("v_mfma"))
            print(f"gap {i:3d}  {g:6.1f}  " + " ".join(f"{k}x{v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()
