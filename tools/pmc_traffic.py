#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; never combined with tracing of
other domains) into HBM bytes per launch per kernel, corrected as MI355X_MICROARCH.md 'HBM' prescribes for gfx950:
FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads (x2); both counters are in KiB.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import re
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def per_launch(path: str, counter: str):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    fetch, write, out = sys.argv[1:4]
    f = per_launch(fetch, "FETCH_SIZE")
    w = per_launch(write, "WRITE_SIZE")
    res = {k: (2.0 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024.0 for k in sorted(set(f) | set(w)) if "gemm" in k or "attn" in k or "layernorm" in k or "geglu" in k}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:60s} {v / 1e6:10.1f} MB/launch (fetch x2 {2 * f.get(k, 0) * 1024 / 1e6:.1f} + write {w.get(k, 0) * 1024 / 1e6:.1f})")


if __name__ == "__main__":
    main()
