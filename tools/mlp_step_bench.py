#!/usr/bin/env python3
"""The learner's minibatch update as the PPO loop runs it (pnr_mlp_train_step on the epoch's pre-gathered rows, 32 768 samples,
both nets): microseconds per update from HIP events, and — under `rocprofv3 --kernel-trace -d DIR -o s -- python3
tools/mlp_step_bench.py` + tools/rocpd_stats.py — per kernel.  Also times the sampler's pnr_mlp_act at 16 384 samples.
Prints one JSON line.  PNR_LIB_PATH selects a variant build."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pioneer_amd.mlp import HipMLP  # noqa: E402
from pioneer_amd.ppo import ActorCritic, PPOConfig, gaussian_logp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = ActorCritic(PPOConfig()).to(dev)
PLANES = int(os.environ.get("PLANES", "1"))      # 1: bf16 operands; 2, 3: split float32 operands
mlp = HipMLP(model, B, dev, planes=PLANES)
mlp.w3_partials = os.environ.get("W3PART", "1") == "1"      # 0: H2 stored and read back (the A/B)
mlp.pack()
R = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
rows = 16 * B
obs = R(rows, 137)
filt = (torch.zeros(137, device=dev), torch.ones(137, device=dev), torch.full((137,), -10.0, device=dev), torch.full((137,), 10.0, device=dev))
act, mean, ls = R(rows, 6), 0.1 * R(rows, 6), 0.1 * R(rows, 6)
rec = {"actions": act, "mean": mean, "log_std": ls, "logp": gaussian_logp(act, mean, ls) + 0.1 * R(rows), "values": R(rows), "adv": R(rows), "vtarg": R(rows)}
perm = torch.randperm(rows, device=dev)
rec_rows = mlp.pack_record(rec)
g = mlp.gather_epoch(obs, perm, filt, None, rec_rows=rec_rows)
XS = (lambda s: g["xs"][s:s + B]) if PLANES == 1 else (lambda s: g["xs"][:, s:s + B])
klc = torch.tensor(0.2, device=dev)
entc = torch.tensor(0.01, device=dev)
means = torch.zeros(8, device=dev)


CHAINS = os.environ.get("CHAINS", "0") == "1"      # the two nets as two chains on two streams (single GPU, Adam fused per net)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
means2 = torch.zeros(2, 8, device=dev)


INKERNEL = os.environ.get("INKERNEL_GATHER", "0") == "1"      # the minibatch's rows gathered (and filtered) by the fused kernel's stage 0 instead of pnr_mlp_gather


def step(i):
    s = (i % 16) * B
    if INKERNEL:
        mlp.train_step(obs, perm[s:s + B], filt, rec, klc, entc, 0.3, 10.0, 1.0, means, 2e-5)
        return
    if CHAINS:
        for net, st in enumerate(streams):
            with torch.cuda.stream(st):
                mlp.train_step(None, None, None, {k: g[k][s:s + B] for k in mlp.REC_KEYS}, klc, entc, 0.3, 10.0, 1.0, means2[net], 2e-5,
                               xs_in=XS(s), nets=(net, 1))
        return
    mlp.train_step(None, None, None, {k: g[k][s:s + B] for k in mlp.REC_KEYS}, klc, entc, 0.3, 10.0, 1.0, means, 2e-5, xs_in=XS(s))


if CHAINS:
    mlp.sync_step_counters(True)
for i in range(20):
    step(i)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
if CHAINS:
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
for i in range(N):
    step(i)
if CHAINS:
    for st in streams:
        torch.cuda.current_stream(dev).wait_stream(st)
    means = means2.sum(0)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / N * 1e3

# the sampler's step at 16 384 samples
S = 16384
smp = HipMLP(model, S, dev, planes=PLANES)
smp.pack()
so = R(S, 137)
noise = R(S, 6)
a_max = torch.ones(6, device=dev)
out = {k: torch.empty(S, 6, device=dev) for k in ("mean", "log_std", "actions", "env")}
vals = torch.empty(S, device=dev)
xs = torch.empty(S, 144, dtype=torch.bfloat16, device=dev) if PLANES == 1 else None
act_call = lambda: smp.act(so, filt, noise, a_max, mean=out["mean"], log_std=out["log_std"], values=vals, actions=out["actions"],  # noqa: E731
                           env_actions=out["env"], xs_out=xs)
for _ in range(20):
    act_call()
torch.cuda.synchronize()
e0.record()
for _ in range(N):
    act_call()
e1.record()
torch.cuda.synchronize()
print(json.dumps({"lib": os.environ.get("PNR_LIB_PATH", "default"), "planes": PLANES, "in_kernel_gather": INKERNEL, "chains": CHAINS, "w3_partials": mlp.w3_partials, "batch": B, "train_step_us": us, "act_16384_us": e0.elapsed_time(e1) / N * 1e3,
                  "param_sums": [round(float(p_.double().sum()), 6) for p_ in mlp.params[:2]] + [round(float(p_.double().sum()), 6) for p_ in mlp.params[6:8]],
                  "means_finite": bool(torch.isfinite(means[:5]).all()), "means": [round(float(x), 5) for x in means[:5]]}))
