#!/usr/bin/env python3
"""What the boundary costs when the caller lives on the HOST (the reference's callers do: RLlib hands numpy actions in and wants numpy
observations back).  Three forms, env-steps/s each, one JSON line:
  facade            PioneerKinematicEnv.step(action) — one env, numpy in / out (the reference's own call shape)
  rllib_vector_env  PioneerRLlibVectorEnv.vector_step(list of rows) -> lists (one device-to-host copy per step + the Python lists)
  pcie_inclusive    PioneerVectorEnv.vector_step on a HOST action array, observation / reward / flags copied back to pinned host
                    memory every step (no Python lists): the PCIe-inclusive rate of the device-resident hot path
The device-resident rate (bench.py's `value`) has none of this in its timed region."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pioneer_amd import PioneerVectorEnv  # noqa: E402
from pioneer_amd.env import make_env  # noqa: E402
from pioneer_amd.rllib_env import PioneerRLlibVectorEnv  # noqa: E402

out = {}
cfg = {"award_potential_slope": 10.0, "award_done": 5.0, "penalty_step": 0.01}

env = make_env(cfg)
env.reset()
a = np.zeros(6, np.float32)
for _ in range(50):
    env.step(a)
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    o, r, d, i = env.step(a)
    if d:
        env.reset()
    n += 1
out["facade_env_steps_per_s"] = n / (time.perf_counter() - t0)
env.close()

for N in (4096, 65536):
    v = PioneerRLlibVectorEnv(N, seed=0)
    v.vector_reset()
    acts = [np.zeros(6, np.float32)] * N
    for _ in range(3):
        v.vector_step(acts)
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        obs, rew, done, info = v.vector_step(acts)
        k += 1
    out[f"rllib_vector_env_{N}_env_steps_per_s"] = k * N / (time.perf_counter() - t0)
    v.close()

N = 65536
dev = torch.device("cuda", 0)
v = PioneerVectorEnv(N, device=dev, seed=0)
v.reset()
a_host = torch.zeros(N, 6).pin_memory()
o_host = torch.empty(N, v.obs_dim).pin_memory()
r_host = torch.empty(N).pin_memory()
f_host = torch.empty(N, dtype=torch.uint8).pin_memory()
a_dev = torch.empty(N, 6, device=dev)


def step():
    a_dev.copy_(a_host, non_blocking=True)
    res = v.vector_step(a_dev)
    o_host.copy_(res[0], non_blocking=True)
    r_host.copy_(res[1], non_blocking=True)
    f_host.copy_(res[2].view(torch.uint8) if res[2].dtype == torch.bool else res[2], non_blocking=True)
    torch.cuda.synchronize()


for _ in range(5):
    step()
k, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    step()
    k += 1
dt = time.perf_counter() - t0
out["pcie_inclusive_65536_env_steps_per_s"] = k * N / dt
out["pcie_inclusive_ms_per_step"] = dt / k * 1e3
out["pcie_bytes_per_step"] = N * (6 * 4 + v.obs_dim * 4 + 4 + 1)
out["pcie_GBps"] = out["pcie_bytes_per_step"] * k / dt / 1e9
print(json.dumps(out))
