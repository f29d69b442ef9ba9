#!/usr/bin/env python3
"""Prints the HOT path of a kernel's main loop from a hipcc -S listing: basic blocks reached only through a forward conditional
branch that jumps over them (the cold blocks of a hand-placed stream: mask, reference move) are skipped, so what is left is the
stream a wave executes on an ordinary tile; per sched_barrier chunk it prints the instruction mix (development aid, r05).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o k.s cm3p_amd/csrc/attention_fwd.hip
    python tools/isa_hotpath.py k.s attn_fwd_g_kernelILi4ELb0 [first_line last_line] [-v]
"""
import re
import sys


sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cm3p_amd.isa_check import hot_path  # noqa: E402  (the same walker tests/test_kernel_isa.py pins the kernel with)


def hot_lines(body, lo, hi):
    return list(hot_path(body, lo, hi))


def main():
    args = [a for a in sys.argv[1:] if a != "-v"]
    verbose = "-v" in sys.argv
    s = open(args[0]).read()
    m = re.search(r"^(_ZN\S*" + re.escape(args[1]) + r"\S*):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M)
    body = m.group(2).split("\n")
    bars = [i for i, l in enumerate(body) if "s_barrier" in l]
    lo, hi = (int(args[2]), int(args[3])) if len(args) > 3 else (bars[1], bars[2])
    chunk, n = {}, 0
    tot = {}

    def flush():
        nonlocal chunk, n
        if chunk:
            print(f"chunk {n:3d}: " + " ".join(f"{k}={v}" for k, v in sorted(chunk.items())))
            n += 1
        chunk = {}

    for i, l in hot_lines(body, lo, hi):
        t = l.strip()
        if "sched_barrier" in t:
            flush()
            continue
        if not l.startswith("\t") or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        key = ("mfma" if op.startswith("v_mfma") else "exp" if op.startswith("v_exp") else "ds" if op.startswith("ds_") else
               "dma" if "load_lds" in op else "nop" if op == "s_nop" else "wait" if op == "s_waitcnt" else "salu" if op.startswith("s_") else
               "acc" if "accvgpr" in op else "valu")
        chunk[key] = chunk.get(key, 0) + 1
        tot[key] = tot.get(key, 0) + 1
        if verbose:
            print(f"    {i:6d} {t}")
    flush()
    print("total:", " ".join(f"{k}={v}" for k, v in sorted(tot.items())))


if __name__ == "__main__":
    main()
