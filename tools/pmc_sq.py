#!/usr/bin/env python3
"""Per-kernel averages of whatever counters one rocprofv3 --pmc pass collected (development aid for the hand-scheduled kernels).

    python tools/pmc_sq.py <counter_collection.csv> [kernel-name-substring ...]

Prints, per kernel, each counter's mean per launch and its ratio to SQ_WAVE_CYCLES when that counter is in the pass
(MI355X_MICROARCH.md, rocprofv3 PMC slots: SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES, quad-cycles).
"""
import collections
import csv
import re
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main():
    path, want = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if want and not any(w in k for w in want):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    for k in sorted(acc):
        print(k)
        wc = acc[k].get("SQ_WAVE_CYCLES", 0.0) / max(1, cnt[k].get("SQ_WAVE_CYCLES", 1))
        for c in sorted(acc[k]):
            mean = acc[k][c] / cnt[k][c]
            rel = f"  {mean / wc:7.3f} of SQ_WAVE_CYCLES" if wc > 0 else ""
            print(f"   {c:32s} {mean:16.1f} per launch ({cnt[k][c]} launches){rel}")


if __name__ == "__main__":
    main()
