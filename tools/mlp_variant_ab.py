#!/usr/bin/env python3
"""A/B of build-time variants of the MLP kernels: by default the tile height PNR_MLP_BM = 64 (two to three workgroups
per CU) against 128 (one workgroup per CU, half the weight traffic from L2); any other set with
`--variants name:-DX=1,-DY=2 name2:...` (e.g. the prefetch ring depth PNR_MLP_RING).  Times the sampling forward and one
full training step (pnr_mlp_train_step) in child processes on every build.  Writes gpurun_out/mlp_variant_ab.json."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from pioneer_amd.ppo import ActorCritic, PPOConfig, gaussian_logp
from pioneer_amd.mlp import HipMLP
dev = torch.device("cuda", 0)
model = ActorCritic(PPOConfig()).to(dev)
res = {}
def timed(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
filt = (torch.zeros(137, device=dev), torch.ones(137, device=dev), torch.full((137,), -10.0, device=dev), torch.full((137,), 10.0, device=dev))
for B in (16384, 32768, 131072):
    mlp = HipMLP(model, B, dev); mlp.pack()
    obs = torch.randn(B, 137, device=dev); out = torch.empty(2, B, 16, device=dev)
    res["forward_%%d" %% B] = timed(lambda: mlp.forward_nograd(obs, None, filt, out=out))
    R = lambda *s: torch.randn(*s, device=dev)
    act, mean, ls = R(B, 6), 0.1 * R(B, 6), 0.1 * R(B, 6)
    rec = {"actions": act, "mean": mean, "log_std": ls, "logp": gaussian_logp(act, mean, ls), "values": R(B), "adv": R(B), "vtarg": R(B)}
    klc = torch.tensor(0.2, device=dev); entc = torch.tensor(0.01, device=dev); means = torch.zeros(8, device=dev)
    res["train_step_%%d" %% B] = timed(lambda: mlp.train_step(obs, None, filt, rec, klc, entc, 0.3, 10.0, 1.0, means, 2e-5))
print(json.dumps(res))
''' % ROOT

from pioneer_amd import _lib  # noqa: E402
variants = {"BM64": ["-DPNR_MLP_BM=64"], "BM128": ["-DPNR_MLP_BM=128"]}
if "--variants" in sys.argv:
    variants = {}
    for spec in sys.argv[sys.argv.index("--variants") + 1:]:
        name, _, flags = spec.partition(":")
        variants[name] = [f for f in flags.split(",") if f]
out = {}
for name, flags in variants.items():
    lib = os.path.join(_lib.CSRC, f"libpioneer_amd_{name}.so")
    if not os.path.exists(lib):
        _lib.build_library(extra_flags=flags, out_path=lib)
    res = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PNR_LIB_PATH=lib), capture_output=True, text=True, timeout=300)
    line = [l for l in res.stdout.splitlines() if l.startswith("{")]
    out[name] = json.loads(line[-1]) if line else {"error": res.stderr[-300:]}
    out[name]["flags"] = flags
    print(name, out[name], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "mlp_variant_ab.json"), "w"), indent=1)
