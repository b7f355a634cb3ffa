"""Run a few PPO iterations (for rocprofv3 --kernel-trace --stats).  PPO_PROFILE_GRAPH=1 replays the
captured hipGraphs (kernel durations without host gaps), default eager."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pioneer_amd import PioneerVectorEnv, EngineConfig
from pioneer_amd.ppo import PPOConfig, PPOTrainer
g = os.environ.get("PPO_PROFILE_GRAPH") == "1"
env = PioneerVectorEnv(16384, device="cuda:0", seed=0, engine_config=EngineConfig(max_episode_steps=500))
prec = os.environ.get("PPO_PROFILE_PRECISION", "f32")      # f32 (the credited loop) | bf16 | bf16x3
tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=int(os.environ.get("PPO_PROFILE_MBS", "32768")),
                               hip_kernels=True if prec == "bf16" else prec), use_graph=g)
for _ in range(int(os.environ.get("PPO_PROFILE_ITERS", "4"))):
    r = tr.train()
print(r["sample_time_s"], r["learn_time_s"])
