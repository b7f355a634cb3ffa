#!/usr/bin/env python3
"""Timeline of one PPO iteration from a rocprofv3 rocpd SQLite result of tools/ppo_profile.py: the dispatches of the last
complete iteration in start order, split into the sampler (from the first pnr::mlp_forward_kernel<false> after the last
learner kernel to the first learner kernel) and the learner, with per-phase kernel time, idle time between consecutive
kernels and the largest gaps.  Answers "what would a resident closed-loop rollout kernel save" (VERDICT r01 #9): only the
sampler's gaps.  Usage: python tools/rocpd_timeline.py RESULTS.db [--json OUT.json]"""
import argparse
import json
import re
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--json")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    short = lambda k: re.sub(r"\(.*", "", k)  # noqa: E731
    is_learn = lambda k: "mlp_forward_kernelILb1" in k or "mlp_wgrad" in k or "mlp_adam" in k or "finish_split" in k or "permutation_kernel" in k  # noqa: E731
    is_act = lambda k: "mlp_forward_kernelILb0" in k  # noqa: E731
    # the last learner phase = the last maximal run of dispatches that starts at a permutation kernel ... ends with adam
    last_learn_end = max(i for i, r in enumerate(rows) if is_learn(r[0]))
    i = last_learn_end
    while i > 0 and not (is_act(rows[i][0]) or "step_kernel" in rows[i][0]):
        i -= 1
    learn = rows[i + 1:last_learn_end + 1]
    # the sampler phase before it: back to the previous learner kernel
    j = i
    while j > 0 and not is_learn(rows[j][0]):
        j -= 1
    sample = rows[j + 1:i + 1]

    def phase(seg):
        busy = sum(e - s for _, s, e in seg)
        span = seg[-1][2] - seg[0][1]
        gaps = sorted(((seg[k + 1][1] - seg[k][2], short(seg[k][0])[:60], short(seg[k + 1][0])[:60]) for k in range(len(seg) - 1)), reverse=True)
        by = {}
        for n, s, e in seg:
            d = by.setdefault(short(n)[:90], [0, 0]); d[0] += 1; d[1] += e - s
        top = sorted(by.items(), key=lambda kv: -kv[1][1])[:8]
        return {"dispatches": len(seg), "span_us": span / 1e3, "kernel_us": busy / 1e3, "idle_us": (span - busy) / 1e3,
                "median_gap_us": sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0,
                "largest_gaps_us": [(g[0] / 1e3, g[1], g[2]) for g in gaps[:5]],
                "top_kernels": [(k, v[0], v[1] / 1e3) for k, v in top]}

    out = {"sampler": phase(sample), "learner": phase(learn)}
    for name, p in out.items():
        print(f"{name}: {p['dispatches']} dispatches, span {p['span_us']:.1f} us, kernels {p['kernel_us']:.1f} us, idle {p['idle_us']:.1f} us, "
              f"median gap {p['median_gap_us']:.2f} us")
        for k, n, us in p["top_kernels"]:
            print(f"    {n:5d} x {us / n:8.2f} us  {k}")
        for g in p["largest_gaps_us"]:
            print(f"    gap {g[0]:8.2f} us  {g[1]} -> {g[2]}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    sys.exit(main())
