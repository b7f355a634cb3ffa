#!/usr/bin/env python3
"""A/B of north_star's "per-link spatial inertias staged in LDS" against the all-registers dynamics kernels (VERDICT r01
next #6c).  Builds the variant library with -DPNR_DYN_LDS_MODEL=1 if it is not there, then times both libraries in child
processes (PNR_LIB_PATH) on the dynamics legs of bench.py: pnr_step at 65 536 and 262 144 envs, pnr_rollout (32 steps
per launch) at 65 536.  Writes gpurun_out/dyn_lds_ab.json."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pioneer_amd import _lib  # noqa: E402

variant = os.path.join(_lib.CSRC, "libpioneer_amd_ldsmodel.so")
if not os.path.exists(variant):
    _lib.build_library(extra_flags=["-DPNR_DYN_LDS_MODEL=1"], out_path=variant)
out = {}
for name, lib in (("registers", os.path.join(_lib.CSRC, "libpioneer_amd.so")), ("lds_model", variant)):
    out[name] = {}
    for label, extra in (("step_65536", ["--envs", "65536"]), ("step_262144", ["--envs", "262144", "--ring", "8"]),
                         ("rollout32_65536", ["--envs", "65536", "--fused", "32"])):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "dynamic", "--randomize", "--gravity", "9.81", "--steps", "640",
               "--warmup", "64", "--no-cpu-baseline", "--ppo-iters", "0", "--large-envs", "0", "--dynamic-leg", "0", "--fused-leg", "0"] + extra
        res = subprocess.run(cmd, env=dict(os.environ, PNR_LIB_PATH=lib), capture_output=True, text=True, timeout=300)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")]
        r = json.loads(line[-1]) if line else {"error": res.stderr[-500:]}
        out[name][label] = {"us_per_step": r.get("ms_per_step", 0) * 1e3, "event_us_per_launch": r.get("roofline", {}).get("avg_launch_ms", 0) * 1e3}
        print(name, label, out[name][label], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "dyn_lds_ab.json"), "w"), indent=1)
