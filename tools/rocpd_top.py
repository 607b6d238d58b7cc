#!/usr/bin/env python3
"""Top kernels of a rocprofv3 run that wrote the rocpd SQLite format: python tools/rocpd_top.py <results.db> [top] [name filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = [r for r in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels") if flt in r[0]]
print(f"{'kernel':100s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'pct':>6s}")
for name, calls, total, avg, pct in rows[:top]:
    print(f"{name[:100]:100s} {calls:6d} {total:11.1f} {avg:9.2f} {pct:6.2f}")
