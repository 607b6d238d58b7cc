#!/usr/bin/env python3
"""Mean per-launch value of one rocprofv3 --pmc counter for the kernels whose name contains a pattern.
usage: pmc_traffic.py <counter_collection.csv> <counter> <kernel substring> [...]"""
import csv
import sys
from collections import defaultdict


def main(path, counter, *patterns):
    per = defaultdict(lambda: defaultdict(float))          # pattern -> dispatch id -> summed value (rows are per XCD / dimension)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        for p in patterns:
            if p in r["Kernel_Name"]:
                per[p][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for p in patterns:
        v = list(per[p].values())
        print(f"{counter} {p}: launches {len(v)} mean {sum(v) / max(len(v), 1):.4f}")


if __name__ == "__main__":
    main(*sys.argv[1:])
