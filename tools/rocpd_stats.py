#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd SQLite result (`rocprofv3 --kernel-trace -d DIR -o NAME` writes
NAME_results.db on this ROCm): count, total, average, min, max duration per kernel name, sorted by total time.
Usage: python tools/rocpd_stats.py RESULTS.db [--csv OUT.csv] [--since-dispatch N] [--top K]"""
import argparse
import csv
import re
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--skip-frac", type=float, default=0.0, help="ignore this leading fraction of the dispatches (warm-up)")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    n = c.execute(f"select count(*) from {kd}").fetchone()[0]
    first = c.execute(f"select id from {kd} order by start limit 1 offset {int(n * a.skip_frac)}").fetchone()[0] if n else 0
    t0 = c.execute(f"select start from {kd} where id = {first}").fetchone()[0] if n else 0
    rows = c.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                     f"from {kd} d join {ks} s on d.kernel_id = s.id where d.start >= {t0} group by s.kernel_name order by 3 desc").fetchall()
    span = c.execute(f"select max(end) - min(start) from {kd} where start >= {t0}").fetchone()[0] or 1
    tot = sum(r[2] for r in rows) or 1
    short = lambda k: re.sub(r"\(.*", "", k)[:110]   # noqa: E731
    print(f"{n} dispatches; considered span {span / 1e6:.3f} ms, kernel time {tot / 1e6:.3f} ms ({100.0 * tot / span:.1f} % busy)")
    print(f"{'calls':>7} {'total_us':>11} {'avg_us':>9} {'min_us':>8} {'max_us':>8} {'%':>6}  kernel")
    for k, cnt, s, avg, mn, mx in rows[:a.top]:
        print(f"{cnt:7d} {s / 1e3:11.1f} {avg / 1e3:9.2f} {mn / 1e3:8.2f} {mx / 1e3:8.2f} {100.0 * s / tot:6.2f}  {short(k)}")
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
            for k, cnt, s, avg, mn, mx in rows:
                w.writerow([k, cnt, s, f"{avg:.1f}", mn, mx, f"{100.0 * s / tot:.3f}"])


if __name__ == "__main__":
    sys.exit(main())
