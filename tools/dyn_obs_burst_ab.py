#!/usr/bin/env python3
"""How much of a dynamics-mode step is the observation burst: dyn_step_kernel at 65 536 and 8 192 envs (randomised, gravity) as it
is, without the obs flush (PNR_DIAG=2) and without emit + flush (PNR_DIAG=6) — timing-only ablations of a -DPNR_DIAG_BUILD=1 variant of
the library (outputs are wrong when set).  Builds the variant if it is missing.  Prints one JSON line per batch size."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pioneer_amd import _lib  # noqa: E402

CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from pioneer_amd import PioneerVectorEnv, EngineConfig, SimulationConfig
n = int(sys.argv[1]); dev = torch.device("cuda:0")
env = PioneerVectorEnv(n, device=dev, seed=0, simulation_config=SimulationConfig(gravity=9.81), engine_config=EngineConfig(mode="dynamic", randomize=True))
env.reset()
acts = (torch.rand(8, n, 6, device=dev) * 2 - 1) * torch.from_numpy(env.a_max).to(dev)
ring = [{"obs": torch.empty(n, 137, device=dev), "reward": torch.empty(n, device=dev), "done": torch.empty(n, dtype=torch.uint8, device=dev),
         "truncated": torch.empty(n, dtype=torch.uint8, device=dev)} for _ in range(16)]
for i in range(50): env.vector_step(acts[i %% 8], out=ring[i %% 16])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
K = 600
for i in range(K): env.vector_step(acts[i %% 8], out=ring[i %% 16])
e1.record(); torch.cuda.synchronize()
print(json.dumps({"us": e0.elapsed_time(e1) / K * 1e3}))
''' % ROOT

lib = os.path.join(_lib.CSRC, "libpioneer_amd_diag.so")
if not os.path.exists(lib):
    _lib.build_library(extra_flags=["-DPNR_DIAG_BUILD=1"], out_path=lib, units=("pnr_api.hip",))
for n in (65536, 8192):
    row = {"envs": n}
    for name, diag in (("full", "0"), ("no_obs_flush", "2"), ("no_obs_emit_no_flush", "6")):
        r = subprocess.run([sys.executable, "-c", CHILD, str(n)], env=dict(os.environ, PNR_LIB_PATH=lib, PNR_DIAG=diag), capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        row[name + "_us"] = json.loads(line[-1])["us"] if line else r.stderr[-200:]
    print(json.dumps(row), flush=True)
