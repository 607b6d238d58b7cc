#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV pair into the text summary committed under profiles/."""
import csv
import sys


def main(stats_csv, out_txt, title, top=40):
    rows = list(csv.DictReader(open(stats_csv)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out_txt, "w") as f:
        f.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats ; total kernel time {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} dispatches\n")
        f.write(f"{'kernel':100s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}\n")
        for r in rows[:top]:
            f.write(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e3:11.1f} {float(r['AverageNs']) / 1e3:9.2f} "
                    f"{float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f} {float(r['Percentage']):6.2f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
