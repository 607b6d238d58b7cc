#!/usr/bin/env python3
"""Per-kernel means of every counter in a rocprofv3 counter_collection.csv (kernels filtered by substring)."""
import csv
import sys
from collections import defaultdict

path, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))      # kernel -> counter -> dispatch -> value
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, d in sorted(cs.items()):
        v = list(d.values())
        print(f"   {c:34s} launches {len(v):4d} mean {sum(v) / len(v):16.1f}")
