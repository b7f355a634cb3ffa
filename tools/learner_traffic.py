#!/usr/bin/env python3
"""HBM-side traffic of the learner kernels per 32 768-sample update from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
separate runs, as MI355X_MICROARCH.md's HBM section prescribes) over tools/mlp_step_bench.py, summarised per kernel by
tools/rocpd_pmc.py.  Counters are KiB per dispatch; FETCH_SIZE of wide coalesced reads counts 64 B per 128-B request on gfx950
and is doubled (the guide's correction; every learner kernel reads with 16-byte-per-lane loads).  Infinity-Cache hits are
counted by both counters (guide), so these are bytes that left the L2s, not DRAM bytes.
Writes profiles/<OUT>.json with the fingerprint of the learner sources; bench.py quotes it as ppo_loop.roofline while the
fingerprint matches.  OUT.json is keyed by operand precision ("planes1" = bf16, "planes3" = float32-accurate split operands): a pass over
`PLANES=3 python3 tools/mlp_step_bench.py` adds / replaces its key.
Usage: python tools/learner_traffic.py FETCH_PMC.json WRITE_PMC.json OUT.json LABEL [BATCH] [PLANES]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pioneer_amd import _lib  # noqa: E402

ROLES = {"mlp_forward_kernelILb1": "fused", "mlp_wgrad_kernel": "wgrad", "mlp_adam_kernel": "adam"}      # (mangled-name substrings)


def main():
    fetch, write, out, label = sys.argv[1:5]
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 32768
    planes = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    fk, wk = json.load(open(fetch))["kernels"], json.load(open(write))["kernels"]
    try:
        whole = json.load(open(out))
    except Exception:
        whole = {}
    sha = _lib.source_fingerprint(_lib.LEARNER_KERNEL_SOURCES)
    if whole.get("learner_sha16") != sha:
        whole = {}                                  # passes taken on other sources do not mix
    doc = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/mlp_step_bench.py; "
                   "KiB per dispatch; FETCH_SIZE doubled (gfx950 wide-read correction); Infinity-Cache hits are counted",
           "profile": label, "batch": batch, "learner_sha16": _lib.source_fingerprint(_lib.LEARNER_KERNEL_SOURCES), "kernels": {}}
    tot_r = tot_w = 0.0
    for name, rec in fk.items():
        role = next((r for k, r in ROLES.items() if k in name), None)
        if role is None or name not in wk:
            continue
        rd = 2.0 * rec["counters_per_dispatch"]["FETCH_SIZE"] * 1024.0
        wr = wk[name]["counters_per_dispatch"]["WRITE_SIZE"] * 1024.0
        doc["kernels"][role] = {"kernel": name, "dispatches": rec["rows"], "avg_duration_us_under_pmc": rec["avg_duration_us"],
                                "read_bytes_corrected": rd, "write_bytes": wr, "bytes": rd + wr}
        tot_r += rd
        tot_w += wr
    doc["bytes_per_update"] = tot_r + tot_w
    doc["read_bytes_per_update"] = tot_r
    doc["write_bytes_per_update"] = tot_w
    whole["learner_sha16"] = sha
    whole["_how"] = doc.pop("_how")
    doc.pop("learner_sha16")
    whole[f"planes{planes}"] = doc
    json.dump(whole, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    sys.exit(main())
