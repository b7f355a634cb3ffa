#!/usr/bin/env python3
"""Where the fused forward kernel's time goes: timing-only ablation builds (-DPNR_MLP_DIAG=bits; results are wrong when a
bit is set) timed on the sampling shape (16 384 raw observations, both nets, filter on load) and on a 32 768-sample
training forward.  bits: 1 no tanh, 2 no observation loads, 4 no layer-2 GEMM, 8 no layer-1 GEMM, 16 no layer 3,
32 no epilogues at all.  Writes gpurun_out/mlp_ablation.json."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from pioneer_amd.ppo import ActorCritic, PPOConfig
from pioneer_amd.mlp import HipMLP
dev = torch.device("cuda", 0)
model = ActorCritic(PPOConfig()).to(dev)
res = {}
for B in (16384, 32768, 131072):
    mlp = HipMLP(model, B, dev); mlp.pack()
    obs = torch.randn(B, 137, device=dev)
    filt = (torch.zeros(137, device=dev), torch.ones(137, device=dev), torch.full((137,), -10.0, device=dev), torch.full((137,), 10.0, device=dev))
    out = torch.empty(2, B, 16, device=dev)
    for _ in range(5):
        mlp.forward_nograd(obs, None, filt, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        mlp.forward_nograd(obs, None, filt, out=out)
    e1.record(); torch.cuda.synchronize()
    res[B] = e0.elapsed_time(e1) / 50 * 1e3
print(json.dumps(res))
''' % ROOT

from pioneer_amd import _lib  # noqa: E402
out = {}
for bits in (0, 1, 2, 32, 8, 4, 12, 16, 63):
    lib = os.path.join(_lib.CSRC, f"libpioneer_amd_diag{bits}.so")
    if not os.path.exists(lib):
        _lib.build_library(extra_flags=[f"-DPNR_MLP_DIAG={bits}"], out_path=lib)
    res = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PNR_LIB_PATH=lib), capture_output=True, text=True, timeout=300)
    line = [l for l in res.stdout.splitlines() if l.startswith("{")]
    out[bits] = json.loads(line[-1]) if line else {"error": res.stderr[-300:]}
    print(bits, out[bits], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "mlp_ablation.json"), "w"), indent=1)
