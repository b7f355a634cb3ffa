#!/usr/bin/env python3
"""Timing-only ablations of the fused learner kernel (mlp_forward_kernel<true>) at 32 768 samples: variant libraries built with
-DPNR_MLP_DIAG=bits (outputs are WRONG when a bit is set; see pnr_mlp.h for the bits), each timed by tools/mlp_step_bench.py in
a child process.  Answers "what is the launch waiting for": tile stores (64), the H1 reload (128), the loss phase (256), the
tanh epilogues (1 / 32), the products (4, 8).  Writes gpurun_out/r03/mlp_fused_ablation.json.  Run on the GPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pioneer_amd import _lib  # noqa: E402

sets = [int(x) for x in sys.argv[1:]] or [0, 64, 128, 256, 1, 32, 12, 64 + 128, 64 + 128 + 256, 64 + 128 + 256 + 32, 511 - 2]
out = {}
for bits in sets:
    lib = os.path.join(_lib.CSRC, f"libpioneer_amd_diag{bits}.so")
    _lib.build_library(extra_flags=[f"-DPNR_MLP_DIAG={bits}"], out_path=lib, units=("pnr_learn.hip",))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mlp_step_bench.py"), "32768", "200"], env=dict(os.environ, PNR_LIB_PATH=lib),
                         capture_output=True, text=True, timeout=300)
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    out[bits] = json.loads(line[-1]) if line else {"error": res.stderr[-300:]}
    print(bits, out[bits].get("train_step_us"), flush=True)
    os.remove(lib)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r03"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03", "mlp_fused_ablation.json"), "w"), indent=1)
