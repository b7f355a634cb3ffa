"""Short PPO run printing the learning curve (evidence that the stack trains the reach task)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pioneer_amd import PioneerVectorEnv, EngineConfig
from pioneer_amd.ppo import PPOConfig, PPOTrainer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mode = sys.argv[2] if len(sys.argv) > 2 else "kinematic"       # "dynamic": ABA + PD tracking of the same commands
mbs = int(sys.argv[3]) if len(sys.argv) > 3 else 131072        # 32768: the contract's minibatch (four times as many updates)
kp = float(sys.argv[4]) if len(sys.argv) > 4 else 4000.0        # dynamics mode: PD gains of the motor
kd = float(sys.argv[5]) if len(sys.argv) > 5 else 400.0
scaled = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False  # gains per unit of each joint's articulated inertia (e.g. 400 40 1)
# the learner's arithmetic: 1 / bf16 = the HIP kernels with bf16 operands; f32 / bf16x3 = the same kernels with float32-accurate split
# operands (PPOConfig.hip_kernels); 0 = the float32 torch learner (autograd, BLAS)
_prec = sys.argv[7] if len(sys.argv) > 7 else "1"
bf16 = {"1": True, "0": False}.get(_prec, _prec)
lr = float(sys.argv[8]) if len(sys.argv) > 8 else 3e-4
env = PioneerVectorEnv(16384, device="cuda:0", seed=0, engine_config=EngineConfig(max_episode_steps=500, mode=mode, pd_kp=kp, pd_kd=kd, pd_inertia_scaled=scaled))
cfg = PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=mbs, lr=lr, hip_kernels=bf16,
                entropy_coeff_start=3e-3, entropy_decay_steps=100_000_000, seed=0)
tr = PPOTrainer(env, cfg, use_graph=True)
t0 = time.time(); rows = []
for it in range(1, iters + 1):
    r = tr.train()
    if it % 25 == 0 or it == 1:
        row = {"iter": it, "mlp": str(cfg.hip_kernels), "lr": lr, "timesteps_M": round(r["timesteps_total"] / 1e6, 1), "reward_mean": round(r["episode_reward_mean"], 2),
               "len_mean": round(r["episode_len_mean"], 1), "episodes": r["episodes_total"], "wall_s": round(time.time() - t0, 1)}
        rows.append(row); print(json.dumps(row), flush=True)
env.close()
