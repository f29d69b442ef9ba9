#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from one rocprofv3 counter pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE):

    util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)

SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's SIMDs and counts cycles (32 per v_mfma_f32_32x32x16_bf16, 16 per
v_mfma_f32_16x16x32_bf16); GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, "DVFS give-back" and the constants
table), so GUI_ACTIVE / 8 is the launch's duration in shader cycles and also gives the clock it ran at.

    python tools/pmc_mfma.py <counter_collection.csv> <out.json>
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


def main():
    path, out = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
            dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) if "End_Timestamp" in r else 0.0
    res = {}
    for k, c in acc.items():
        gui, busy = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if gui <= 0 or busy <= 0:
            continue
        cycles = gui / 8.0
        res[k] = {"launches": cnt[k], "mfma_util": busy / (1024.0 * cycles),
                  "clock_ghz": (cycles / dur[k]) if dur[k] > 0 else None}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["mfma_util"]):
        ck = f"{v['clock_ghz']:.2f} GHz" if v["clock_ghz"] else ""
        print(f"{k:48s} launches {v['launches']:5d}  MFMA-busy {100 * v['mfma_util']:5.1f} %  {ck}")


if __name__ == "__main__":
    main()
