#!/usr/bin/env python3
"""Do the step kernels of a multi-stream run overlap?  From a rocprofv3 rocpd SQLite result: the last N dispatches whose
name matches, with start offset, duration, queue, and for each the time during which another matching dispatch was running.
Usage: python tools/rocpd_overlap.py RESULTS.db [--match step_kernel] [--last 64] [--json OUT.json]"""
import argparse
import json
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--match", default="step_kernel")
    ap.add_argument("--last", type=int, default=64)
    ap.add_argument("--json")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in cols else None
    sel = f"select d.start, d.end, {('d.' + qcol) if qcol else '0'}, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id " \
          f"where s.kernel_name like '%{a.match}%' order by d.start"
    rows = c.execute(sel).fetchall()[-a.last:]
    t0 = rows[0][0]
    out = []
    for i, (s, e, q, g) in enumerate(rows):
        ov = 0
        for j, (s2, e2, _, _) in enumerate(rows):
            if j != i:
                ov += max(0, min(e, e2) - max(s, s2))
        out.append({"start_us": (s - t0) / 1e3, "dur_us": (e - s) / 1e3, "queue": q, "grid": g, "overlapped_us": ov / 1e3})
    span = (rows[-1][1] - rows[0][0]) / 1e3
    busy_union = 0.0
    ev = sorted([(s, 1) for s, _, _, _ in rows] + [(e, -1) for _, e, _, _ in rows])
    depth, last = 0, None
    for t, d in ev:
        if depth > 0:
            busy_union += (t - last) / 1e3
        depth += d
        last = t
    summ = {"dispatches": len(rows), "span_us": span, "union_busy_us": busy_union, "idle_us": span - busy_union,
            "mean_dur_us": sum(o["dur_us"] for o in out) / len(out), "mean_overlapped_us": sum(o["overlapped_us"] for o in out) / len(out),
            "period_us_per_dispatch": span / len(rows), "columns": cols}
    print(json.dumps(summ))
    for o in out[:24]:
        print(o)
    if a.json:
        json.dump({"summary": summ, "dispatches": out}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
