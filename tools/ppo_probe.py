"""Probe: where the PPO loop's time goes (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pioneer_amd import PioneerVectorEnv, EngineConfig
from pioneer_amd.ppo import PPOConfig, PPOTrainer

for amp, mbs, graph in ((False, 131072, False), (False, 131072, True), (True, 131072, True)):
    if True:
        env = PioneerVectorEnv(16384, device="cuda:0", seed=0, engine_config=EngineConfig(max_episode_steps=500))
        tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=mbs, amp_bf16=amp), use_graph=graph)
        tr.train(); tr.train()
        rs = [tr.train() for _ in range(3)]
        s = sum(r["sample_time_s"] for r in rs) / 3; l = sum(r["learn_time_s"] for r in rs) / 3
        print(f"amp_bf16={amp} mbs={mbs} graph={graph}: sample {s*1e3:.1f} ms  learn {l*1e3:.1f} ms  -> {32*16384/(s+l)/1e6:.2f} M env-steps/s  kl {rs[-1]['kl']:.4f}", flush=True)
        env.close()
