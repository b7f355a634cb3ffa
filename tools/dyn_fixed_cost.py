"""Dynamics mode: time per env-step batch as a function of frame_skip (sub-steps per step), to split
the per-launch fixed cost (launches, loads, integrator, obs kernel) from the per-sub-step cost."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pioneer_amd import PioneerVectorEnv, EngineConfig, SimulationConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
res = {}
for fs in (1, 2, 5, 10, 20):
    env = PioneerVectorEnv(n, device=dev, seed=0, simulation_config=SimulationConfig(frame_skip=fs, gravity=9.81),
                           engine_config=EngineConfig(mode="dynamic", randomize=True))
    env.reset()
    acts = (torch.rand(8, n, 6, device=dev) * 2 - 1) * torch.from_numpy(env.a_max).to(dev)
    out = {"obs": torch.empty(n, 137, device=dev), "reward": torch.empty(n, device=dev),
           "done": torch.empty(n, dtype=torch.uint8, device=dev), "truncated": torch.empty(n, dtype=torch.uint8, device=dev)}
    for i in range(50):
        env.vector_step(acts[i % 8], out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    K = 400
    for i in range(K):
        env.vector_step(acts[i % 8], out=out)
    e1.record(); torch.cuda.synchronize()
    res[fs] = e0.elapsed_time(e1) / K * 1e3
    env.close()
fs = sorted(res)
slope = (res[fs[-1]] - res[fs[0]]) / (fs[-1] - fs[0])
print(json.dumps({"envs": n, "us_per_step_by_frame_skip": res, "us_per_substep": slope, "fixed_us": res[fs[0]] - slope * fs[0]}))
