"""Roll a trained policy out in the single-env façade and record it — the role of the reference's
`agent.restore(...)` + `VideoRecorder` evaluation script (temp/pioneer_eval.py:53-78; the only artefact
the reference ships is such a recording, demo.gif).  Frames come from env.render("rgb_array")
(bullet_env.py:156-185; here the host-side stick-figure rasteriser of render.py)."""
from typing import Dict, List, Optional

import numpy as np
import torch

from .env import PioneerKinematicEnv, TimeLimit
from .ppo import ActorCritic, MeanStdFilter, NoFilter, PPOConfig


def load_policy(checkpoint: str, device):
    """(model, obs filter, PPOConfig) from a PPOTrainer.save() file."""
    ck = torch.load(checkpoint, map_location=device, weights_only=True)
    cfg = PPOConfig.from_dict(ck["cfg"])
    model = ActorCritic(cfg).to(device)
    model.load_state_dict(ck["model"])
    model.eval()
    if ck["filter"]:
        filt = MeanStdFilter(cfg.obs_dim, device, cfg.filter_clip)
        filt.load_state_dict(ck["filter"])
    else:
        filt = NoFilter()
    return model, filt, cfg


@torch.no_grad()
def evaluate(checkpoint: str, episodes: int = 3, max_episode_steps: int = 500, gif_path: Optional[str] = None,
             device="cuda:0", mode: str = "kinematic", seed: Optional[int] = 0, frame_stride: int = 1,
             deterministic: bool = True, engine_config=None) -> Dict:
    """Returns {"episode_rewards", "episode_lengths", "successes", "frames"}; writes an animated GIF of all
    episodes when gif_path is given (24 frames per second, the env's metadata rate)."""
    device = torch.device(device)
    model, filt, cfg = load_policy(checkpoint, device)
    env = TimeLimit(PioneerKinematicEnv(device=device, mode=mode, engine_config=engine_config), max_episode_steps=max_episode_steps)
    if seed is not None:
        env.seed(seed)
    a_max = torch.from_numpy(env.action_space.high).to(device)
    gen = torch.Generator(device=device).manual_seed(0 if seed is None else seed)
    frames: List[np.ndarray] = []
    rewards, lengths, successes = [], [], []
    for _ in range(episodes):
        obs = env.reset()
        total, steps, done, info = 0.0, 0, False, {}
        while not done:
            x = filt(torch.as_tensor(obs, dtype=torch.float32, device=device).unsqueeze(0))
            mean, log_std, _ = model(x)          # one env: the float32 master weights through torch
            act = mean if deterministic else mean + torch.exp(log_std) * torch.randn(mean.shape, generator=gen, device=device)
            if cfg.clip_actions:
                act = torch.maximum(torch.minimum(act, a_max), -a_max)
            obs, reward, done, info = env.step(act[0].cpu().numpy())
            total += reward; steps += 1
            if gif_path is not None and steps % frame_stride == 0:
                frames.append(env.render(mode="rgb_array"))
        rewards.append(total); lengths.append(steps)
        successes.append(float(info.get("r_done", "0")) > 0)      # the episode ended inside done_distance
    fps = env.metadata["video.frames_per_second"]
    env.close()
    if gif_path is not None and frames:
        from PIL import Image
        imgs = [Image.fromarray(f) for f in frames]
        imgs[0].save(gif_path, save_all=True, append_images=imgs[1:], duration=int(1000 * frame_stride / fps), loop=0)
    return {"episode_rewards": rewards, "episode_lengths": lengths, "successes": successes, "frames": len(frames)}
