"""Train briefly (tools/train_curve.py's setting), save a checkpoint, record evaluation episodes as a GIF."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from pioneer_amd import PioneerVectorEnv, EngineConfig
from pioneer_amd.evaluate import evaluate
from pioneer_amd.ppo import PPOConfig, PPOTrainer

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dynamic = len(sys.argv) > 3 and sys.argv[3] == "dynamic"         # ABA + the inertia-scaled motor instead of the kinematic arm
os.makedirs(out, exist_ok=True)
eng = EngineConfig(max_episode_steps=500, mode="dynamic", pd_kp=400.0, pd_kd=40.0, pd_inertia_scaled=True) if dynamic \
    else EngineConfig(max_episode_steps=500)
env = PioneerVectorEnv(16384, device="cuda:0", seed=0, engine_config=eng)
tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=32768, lr=1e-4, 
                               entropy_coeff_start=3e-3, entropy_decay_steps=100_000_000, seed=0), use_graph=True)
for _ in range(iters):
    r = tr.train()
ck = tr.save(os.path.join(out, "demo_policy.pt"))
env.close()
res = evaluate(ck, episodes=4, max_episode_steps=500, gif_path=os.path.join(out, "demo_full.gif"), frame_stride=2, seed=11,
               engine_config=eng if dynamic else None)
im = Image.open(os.path.join(out, "demo_full.gif"))
small = []
for k in range(im.n_frames):
    im.seek(k)
    small.append(im.convert("RGB").resize((480, 300), Image.BILINEAR).quantize(32))
small[0].save(os.path.join(out, "demo.gif"), save_all=True, append_images=small[1:], duration=83, loop=0, optimize=True)
res.update({"train_iterations": iters, "train_reward_mean": r["episode_reward_mean"], "train_len_mean": r["episode_len_mean"]})
print(json.dumps(res))
