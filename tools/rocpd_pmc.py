#!/usr/bin/env python3
"""Per-kernel PMC sums from a rocprofv3 rocpd SQLite result (`rocprofv3 --kernel-trace --pmc ... -d DIR -o NAME`):
for every kernel name the number of dispatches, the mean duration and the mean value of every collected counter per
dispatch.  Written because the raw .db of a whole PPO run exceeds what gpurun copies back: run it on the GPU box and
keep the JSON.  Usage: python tools/rocpd_pmc.py RESULTS.db OUT.json [--match pnr::]"""
import argparse
import json
import re
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db"); ap.add_argument("out"); ap.add_argument("--match", default="pnr")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda key: [t for t in tabs if key in t][0]   # noqa: E731
    kd, ks, pe, pi = T("kernel_dispatch"), T("kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    cols = lambda t: [r[1] for r in c.execute(f"pragma table_info({t})")]   # noqa: E731
    pecols, picols = cols(pe), cols(pi)
    name_col = "name" if "name" in picols else [x for x in picols if "name" in x][0]
    # a dispatch has one row per counter INSTANCE (XCD / shader engine): sum them, then average over dispatches
    q = (f"select s.kernel_name, p.{name_col}, count(distinct d.id), sum(e.value), avg(d.end - d.start) from {pe} e "
         f"join {pi} p on e.pmc_id = p.id join {kd} d on d.event_id = e.event_id join {ks} s on d.kernel_id = s.id "
         f"group by s.kernel_name, p.{name_col}")
    out = {}
    for kname, cname, n, total, avg_ns in c.execute(q):
        if a.match and a.match not in kname:
            continue
        k = re.sub(r"\(.*", "", kname)
        rec = out.setdefault(k, {"avg_duration_us": avg_ns / 1e3, "counters_per_dispatch": {}, "rows": 0})
        rec["counters_per_dispatch"][cname] = total / max(1, n)
        rec["rows"] = max(rec["rows"], n)
    meta = {"pmc_event_columns": pecols, "info_pmc_columns": picols}
    json.dump({"kernels": out, "meta": meta}, open(a.out, "w"), indent=1)
    for k, rec in out.items():
        print(k[:90], round(rec["avg_duration_us"], 2), {x: round(y, 1) for x, y in rec["counters_per_dispatch"].items()})


if __name__ == "__main__":
    sys.exit(main())
