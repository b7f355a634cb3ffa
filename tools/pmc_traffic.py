#!/usr/bin/env python3
"""HBM traffic of `pnr::step_kernel` (or, with a fifth argument `dynamic`, of `pnr::dyn_step_kernel<…, RAND>` with per-env
randomised parameters: key "dynamic:env_major:65536:1", 842 algorithmic bytes per env-step) per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs,
as MI355X_MICROARCH.md's HBM section prescribes), summarised by tools/rocpd_pmc.py: writes the
"kinematic:env_major:65536:1" entry of profiles/pmc_traffic.json that bench.py quotes as roofline.traffic.
Counters are KiB per dispatch; the gfx950 FETCH_SIZE of wide coalesced reads counts 64 B per 128-B request and is doubled.
Usage: python tools/pmc_traffic.py FETCH_PMC.json WRITE_PMC.json OUT.json PROFILE_LABEL [kinematic|dynamic]"""
import json
import sys


def main():
    fetch, write, out, label = sys.argv[1:5]
    mode = sys.argv[5] if len(sys.argv) > 5 else "kinematic"
    def pick(path, counter):
        ks = json.load(open(path))["kernels"]
        if mode == "dynamic":
            k = [v for n, v in ks.items() if "dyn_step_kernelILb1ELb1ELb1E" in n][0]
        else:
            k = [v for n, v in ks.items() if "step_kernelILb1ELb1E" in n and "dyn" not in n][0]
        return k["counters_per_dispatch"][counter], k["rows"]
    f_kib, n = pick(fetch, "FETCH_SIZE")
    w_kib, _ = pick(write, "WRITE_SIZE")
    envs = 65536
    alg_r, alg_w = envs * (24 + 92 + (92 if mode == "dynamic" else 0)), envs * (80 + 548 + 6)
    rd, wr = 2.0 * f_kib * 1024.0, w_kib * 1024.0
    try:
        doc = json.load(open(out))
    except Exception:
        doc = {}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pioneer_amd import _lib
    doc["csrc_sha16"] = _lib.source_fingerprint()      # the env-kernel sources these passes ran on (bench.py checks it)
    doc.setdefault("_how", "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the bench's main leg; KiB per dispatch")
    key = f"{mode}:env_major:65536:1"
    doc[key] = {
        "profile": label, "fetch_size_kib_mean": f_kib, "write_size_kib_mean": w_kib, "dispatches": n,
        "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
        "algorithmic_read_bytes": alg_r, "algorithmic_write_bytes": alg_w, "algorithmic_bytes": alg_r + alg_w,
        "traffic_over_algorithmic": (rd + wr) / (alg_r + alg_w)}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc[key]))


if __name__ == "__main__":
    sys.exit(main())
