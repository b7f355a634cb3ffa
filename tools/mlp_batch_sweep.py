#!/usr/bin/env python3
"""pnr_mlp_train_step over a sweep of batch sizes (contiguous rows): run under `rocprofv3 --kernel-trace` and read the
fused kernel's duration per grid size from the rocpd database (tools/rocpd_stats.py groups by kernel name only; the grid
is in the kernel_dispatch table).  The numbers quoted in DESIGN.md section 6c (29.7 us for 384 workgroups ... 163.5 us
for 3 072) come from this script.  Usage: rocprofv3 --kernel-trace -d DIR -o s -- python3 tools/mlp_batch_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pioneer_amd.mlp import HipMLP  # noqa: E402
from pioneer_amd.ppo import ActorCritic, PPOConfig, gaussian_logp  # noqa: E402

dev = torch.device("cuda", 0)
model = ActorCritic(PPOConfig()).to(dev)
filt = (torch.zeros(137, device=dev), torch.ones(137, device=dev), torch.full((137,), -10.0, device=dev), torch.full((137,), 10.0, device=dev))
for B in (12288, 16384, 24576, 32768, 40960, 49152, 65536, 98304):
    mlp = HipMLP(model, B, dev)
    mlp.pack()
    obs = torch.randn(B, 137, device=dev)
    R = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    act, mean, ls = R(B, 6), 0.1 * R(B, 6), 0.1 * R(B, 6)
    rec = {"actions": act, "mean": mean, "log_std": ls, "logp": gaussian_logp(act, mean, ls), "values": R(B), "adv": R(B), "vtarg": R(B)}
    klc = torch.tensor(0.2, device=dev)
    entc = torch.tensor(0.01, device=dev)
    means = torch.zeros(8, device=dev)
    for _ in range(12):
        mlp.train_step(obs, None, filt, rec, klc, entc, 0.3, 10.0, 1.0, means, 2e-5)
    torch.cuda.synchronize()
