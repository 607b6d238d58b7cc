#!/usr/bin/env python3
"""Text summary (the format committed under profiles/) of a rocprofv3 --kernel-trace --stats run that wrote the rocpd SQLite format:
python tools/rocpd_summary.py <results.db> <out.txt> "<title>" [top]"""
import sqlite3
import sys


def main(db_path, out_txt, title, top=45):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    mins = {}
    try:
        for name, mn, mx in db.execute("select name, min(duration), max(duration) from kernels group by name"):
            mins[name] = (mn / 1e3, mx / 1e3)
    except sqlite3.Error:
        pass
    tot = sum(r[2] for r in rows)
    with open(out_txt, "w") as f:
        f.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats (rocpd) ; total kernel time {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
        f.write(f"{'kernel':100s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}\n")
        for name, calls, total, avg, pct in rows[:top]:
            mn, mx = mins.get(name, (float('nan'), float('nan')))
            f.write(f"{name[:100]:100s} {calls:6d} {total:11.1f} {avg:9.2f} {mn:9.2f} {mx:9.2f} {pct:6.2f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 45)
