#!/usr/bin/env python3
"""Per-kernel roofline table of one workload from the committed rocprofv3 passes under profiles/:
time per launch and per step (kernel stats), fabric bytes per launch (FETCH x 2 + WRITE), the rate they make, matrix-pipe busy and clock.
(Averages are over ALL launches of a row: the beatmap tower's and the 16 x smaller metadata tower's.)

    python tools/roofline_table.py [c2|c4] [steps in the trace = 8]      -> markdown on stdout
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    P = os.path.join(ROOT, "profiles")
    rows = list(csv.DictReader(open(os.path.join(P, f"r04_{wl}_kernel_stats.csv"))))
    traffic = json.load(open(os.path.join(P, f"traffic_{wl}.json")))
    busy = json.load(open(os.path.join(P, f"mfma_util_{wl}.json")))
    print("| kernel (rocprof row) | launches / step | µs / launch | ms / step | MB / launch (PMC) | TB/s | matrix-busy | clock GHz |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        ms = float(r["TotalDurationNs"]) / steps / 1e6
        if ms < 0.9:
            continue
        k = short(r["Name"])
        us = float(r["AverageNs"]) / 1e3
        tb = traffic.get(k)
        b = busy.get(k, {})
        print(f"| `{k}` | {int(r['Calls']) / steps:.0f} | {us:.1f} | {ms:.2f} | {tb / 1e6:.0f} | {tb / (us * 1e-6) / 1e12:.2f} | "
              + (f"{100 * b['mfma_util']:.1f} % | {b['clock_ghz']:.2f} |" if b.get("mfma_util") else "- | - |") if tb else
              f"| `{k}` | {int(r['Calls']) / steps:.0f} | {us:.1f} | {ms:.2f} | - | - | - | - |")


if __name__ == "__main__":
    main()
