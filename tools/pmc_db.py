#!/usr/bin/env python3
"""Print per-kernel averages of the counters in rocprofv3 result databases: python tools/pmc_db.py <dir> [kernel substring]."""
import glob, sqlite3, sys
root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(root + "/**/*_results.db", recursive=True)):
    cur = sqlite3.connect(f).cursor()
    for name, counter, avg, n, dur in cur.execute(
            "select kernel_name, counter_name, avg(value), count(*), avg(end-start) from counters_collection group by kernel_name, counter_name"):
        if sub in name:
            print(f"{name.replace('(anonymous namespace)::','')[:44]:44s} {counter:34s} {avg:14.5g}  n={n} dur_us={dur/1e3:.1f}")
