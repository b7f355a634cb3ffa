#!/usr/bin/env python3
"""What mlp_adam_kernel's time is made of: the same launch with ONE flat gradient instead of 32 slabs (pnr_mlp_adam: no slab traffic, no
loss-means block), timed by HIP events back to back and — under rocprofv3 --kernel-trace + tools/rocpd_stats.py — as kernel duration.
Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pioneer_amd.mlp import HipMLP  # noqa: E402
from pioneer_amd.ppo import ActorCritic, PPOConfig  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mlp = HipMLP(ActorCritic(PPOConfig()).to(dev), 32768, dev)
mlp.pack()
flat = 1e-3 * torch.randn(int(mlp.lib.pnr_mlp_grad_floats()), device=dev)
_, _, step = mlp.adam_state()
step.fill_(1.0)
for _ in range(20):
    mlp.adam(flat, 1.0, 2e-5)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(N):
    mlp.adam(flat, 1.0, 2e-5)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"adam_flat_gradient_us_back_to_back": e0.elapsed_time(e1) / N * 1e3}))
