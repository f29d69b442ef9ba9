#!/usr/bin/env python3
"""Per-loop instruction census of one kernel in a hipcc -S listing (development aid for the hand-scheduled kernels).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o k.s cm3p_amd/csrc/attention_bwd_fused.hip
    python tools/isa_loops.py k.s attn_bwd_fused_kernelILb1
"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    name = sys.argv[2]
    m = re.search(r"^(_ZN\S*" + re.escape(name) + r"\S*):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M)
    body = m.group(2).split("\n")
    print(len(body), "lines")
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    pats = ["v_mfma", "scratch_", "v_accvgpr_write", "v_accvgpr_read", "v_exp", "ds_read", "ds_write", "global_load", "global_store",
            "s_waitcnt", "s_nop", "v_mov_b32", "s_barrier"]
    for i, l in enumerate(body):
        mm = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            seg = body[labels[mm.group(1)]:i]
            ins = [x for x in seg if x.startswith("\t") and not x.strip().startswith((".", ";"))]
            print("loop", mm.group(1), "lines", labels[mm.group(1)], i, "instructions", len(ins),
                  " ".join(f"{p}={sum(1 for x in ins if p in x)}" for p in pats))
    print("whole kernel:", " ".join(f"{p}={sum(1 for x in body if p in x)}" for p in pats))


if __name__ == "__main__":
    main()
