#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSV rows per (kernel, counter): python tools/pmc_sum.py <counter_collection.csv> [name filter]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    k = r["Kernel_Name"][:60]
    if flt and flt not in k:
        continue
    a = acc[(k, r["Counter_Name"])]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:60s} {c:28s} {v / n:16.1f} per dispatch ({n} dispatches)")
