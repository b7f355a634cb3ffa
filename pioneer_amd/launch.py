"""train(): the host-side counterpart of pioneer/launch/pioneer_knm_train.py:14-76.

Same arguments (results_dir, checkpoint_freq, num_samples, num_workers, monitor) and the same
result columns (episode_reward_{max,min,mean}, episode_len_mean, episodes_total, plus
experiment_id / trial_id as cli.py:32-38 selects them).  Ray Tune's trial fan-out becomes a
sequential loop of ``num_samples`` trials (each samples its entropy-coefficient start value
log-uniformly in [1e-3, 1e-1], pioneer_knm_train.py:32-41,63); RLlib's rollout workers become
one device-resident env batch per GPU (``num_workers`` scales the batch: workers x envs_per_worker).
``monitor`` is RLlib's episode recording switch there (:51): here it prints progress lines and writes one evaluation
episode as ``monitor_<iteration>.gif`` next to every checkpoint.  Launch with torch.distributed.run for several GPUs.
"""
import json
import os
import time
import uuid
from typing import Dict, List, Optional

import numpy as np
import torch

from . import dist as pdist
from .config import EngineConfig, PioneerKinematicConfig
from .ppo import PPOConfig, PPOTrainer, sample_entropy_start
from .tb import ScalarWriter
from .vector_env import PioneerVectorEnv

ENV_CONFIG = {"award_potential_slope": 10.0, "award_done": 5.0, "penalty_step": 1 / 100}  # pioneer_knm_train.py:53-57
RESULT_COLUMNS = ["experiment_id", "trial_id", "episode_reward_max", "episode_reward_min",
                  "episode_reward_mean", "episode_len_mean", "episodes_total"]                 # cli.py:32-38


def train(results_dir: str,
          checkpoint_freq: int,
          num_samples: int,
          num_workers: int,
          monitor: bool = False,
          training_iterations: int = 1000,          # stop={'training_iteration': 1000}, :68-70
          envs_per_worker: int = 4096,
          ppo_config: Optional[PPOConfig] = None,
          mode: str = "kinematic",
          log_every: int = 10,
          use_graph: bool = True,
          monitor_steps: int = 500,
          engine_config: Optional[EngineConfig] = None,
          trial_parallel: bool = False,
          restore: Optional[str] = None,
          hip_kernels: object = None):
    """Returns a pandas DataFrame (one row per trial, Tune-style) if pandas is importable, else the row list.

    Several GPUs (torch.distributed.run): by default every trial is data-parallel over the ranks (env shards, gradients
    all-reduced).  ``trial_parallel=True`` is the reference's own parallelism instead — Tune runs its ``num_samples`` trials
    independently (pioneer_knm_train.py:43-44; cli.py:15 defaults to 128 of them): rank r trains trials r, r + world, ... on
    its own GPU with ``num_workers x envs_per_worker`` envs each and NO traffic between the GPUs until the result rows are
    gathered at the end.  ``restore``: a PPOTrainer.save() checkpoint every trial starts from (Tune's restore=...).
    ``hip_kernels``: the learner's arithmetic when no ``ppo_config`` is given — "f32" (default: float32-accurate products on the
    hand-written kernels, the arithmetic of the reference's float32 torch learner, pioneer_knm_train.py:47), True / "bf16" (the
    reduced-precision fast variant), "bf16x3", or False (the float32 torch formulation itself); PPOConfig.hip_kernels."""
    # the engine's own options: as given, or — dynamics mode — the inertia-scaled motor (omega = 20 rad/s, zeta = 1 on every
    # joint: with plain torque gains PPO does not learn the task); TimeLimit(500) and auto-reset as
    # the reference's prepare_env wraps it (pioneer_knm_train.py:27)
    # A caller's engine_config is used AS GIVEN (its max_episode_steps / auto_reset included; auto_reset must be on: the
    # sampler never resets by hand); the defaults built here carry TimeLimit(500) and auto-reset.
    engine = engine_config or (EngineConfig(mode="dynamic", pd_kp=400.0, pd_kd=40.0, pd_inertia_scaled=True, max_episode_steps=500,
                                            auto_reset=True) if mode == "dynamic" else EngineConfig(max_episode_steps=500, auto_reset=True))
    if not engine.auto_reset:
        raise AssertionError("train(): engine_config.auto_reset must be True (the sampler relies on the in-kernel auto-reset)")
    mode = engine.mode
    _, local_rank, _ = pdist.world_info()
    if os.environ.get("PNR_DIST_BACKEND") == "gloo":        # rehearsal: ranks may share the visible GPUs
        local_rank %= max(1, torch.cuda.device_count())
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    pdist.init_distributed(device=device)
    results_dir = os.path.expanduser(results_dir)
    experiment_id = pdist.broadcast_object(uuid.uuid4().hex)

    def run_one(trial: int) -> Dict:
        # (inside run_trials' solo context with trial_parallel: this rank is then "rank 0 of 1")
        rank, _, world = pdist.world_info()
        total_envs = max(1, num_workers) * envs_per_worker * world
        start, count = pdist.shard_range(total_envs, world, rank)
        trial_id = f"{trial:05d}"
        tdir = os.path.join(results_dir, f"PPO_Pioneer-v1_{trial_id}")
        if rank == 0:
            os.makedirs(tdir, exist_ok=True)
        # default when no PPOConfig is given: the reference's learning rate, nets, filter and entropy schedule, but
        # GPU-scale batching: T = 32 steps of every env per iteration (131 072 samples per 4 096
        # envs instead of train_batch_size 8 000), 4 epochs of 32 768-sample minibatches instead of 20 x 128, float32-accurate products
        # The entropy schedule keeps the reference's LENGTH IN ITERATIONS: 1 M timesteps of 8 000-sample batches
        # = 125 iterations there (pioneer_knm_train.py:37-40, :62); left at 1 M timesteps it would be over after
        # 8 of these 131 072-sample iterations (2 at 16 384 envs).
        cfg = ppo_config or PPOConfig(num_sgd_iter=4, sgd_minibatch_size=max(1, 32768 // world),   # 32 768 samples per GLOBAL minibatch
                                      entropy_decay_steps=125 * 32 * total_envs,
                                      hip_kernels="f32" if hip_kernels is None else hip_kernels)
        ent_rng = np.random.RandomState(cfg.seed + 7919 * trial)
        cfg = PPOConfig(**{**cfg.__dict__, "entropy_coeff_start": sample_entropy_start(ent_rng),
                           "seed": cfg.seed + trial})
        pioneer_config = PioneerKinematicConfig(                      # prepare_env, :20-27
            award_potential_slope=float(ENV_CONFIG["award_potential_slope"]),
            award_done=float(ENV_CONFIG["award_done"]),
            penalty_step=float(ENV_CONFIG["penalty_step"]))
        env = PioneerVectorEnv(count, device=device, seed=cfg.seed, env_id_offset=start,
                               pioneer_config=pioneer_config,
                               engine_config=engine)
        trainer = PPOTrainer(env, cfg, use_graph=use_graph)   # hipGraph-captured sampling
        if restore:
            trainer.restore(os.path.expanduser(restore))
        last = {}
        # the files Tune leaves in a trial directory: params.json, result.json (one line per iteration), progress.csv,
        # a TensorBoard event file
        log = open(os.path.join(tdir, "result.json"), "a") if rank == 0 else None
        csv_log, csv_cols = None, None
        tb_log = ScalarWriter(tdir) if rank == 0 else None      # events.out.tfevents.*: `tensorboard --logdir results_dir`
        if rank == 0:
            with open(os.path.join(tdir, "params.json"), "w") as f:
                json.dump({"env": "Pioneer-v1", "env_config": ENV_CONFIG, "mode": mode, "num_envs": total_envs,
                           "max_episode_steps": engine.max_episode_steps,
                           # deliberate deviation when no PPOConfig is given: the reference's entropy schedule ends after 1 M
                           # timesteps = 125 of ITS iterations; the default here keeps the 125 iterations (see above)
                           "entropy_decay_steps_reference": 1_000_000,
                           **{k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.__dict__.items()}}, f, indent=2)
        t0 = time.time()
        for it in range(1, training_iterations + 1):
            last = trainer.train()
            last.update({"experiment_id": experiment_id, "trial_id": trial_id, "time_total_s": time.time() - t0})
            if log:
                log.write(json.dumps({k: v for k, v in last.items()}) + "\n"); log.flush()
                if csv_log is None:
                    csv_cols = list(last.keys())
                    csv_log = open(os.path.join(tdir, "progress.csv"), "a")
                    if csv_log.tell() == 0:
                        csv_log.write(",".join(csv_cols) + "\n")
                csv_log.write(",".join(str(last.get(c, "")) for c in csv_cols) + "\n"); csv_log.flush()
                tb_log.add_scalars(last, step=it)
                if monitor and it % log_every == 0:
                    print(f"[{trial_id}] iter {it} reward_mean {last['episode_reward_mean']:.2f} "
                          f"len {last['episode_len_mean']:.1f} steps/s {last['env_steps_per_s']:.3g}", flush=True)
            if checkpoint_freq and it % checkpoint_freq == 0:
                ck = trainer.save(os.path.join(tdir, f"checkpoint_{it}.pt"))
                _record(monitor and rank == 0, ck, os.path.join(tdir, f"monitor_{it}.gif"), device, mode, monitor_steps, engine)
        ck = trainer.save(os.path.join(tdir, "checkpoint_final.pt"))      # checkpoint_at_end=True, :73
        _record(monitor and rank == 0, ck, os.path.join(tdir, "monitor_final.gif"), device, mode, monitor_steps, engine)
        if log:
            log.close()
        if csv_log:
            csv_log.close()
        if tb_log:
            tb_log.close()
        env.close()
        return last

    rows: List[Dict] = pdist.run_trials(num_samples, run_one, trial_parallel)
    try:
        import pandas as pd
        return pd.DataFrame(rows)
    except Exception:
        return rows


def _record(enabled: bool, checkpoint: str, gif_path: str, device, mode: str, max_steps: int, engine_config=None) -> None:
    """RLlib's 'monitor': True (pioneer_knm_train.py:51) records episodes as videos into the results directory; here one
    evaluation episode of the checkpointed policy is written as a GIF next to each checkpoint."""
    if not enabled:
        return
    try:
        from .evaluate import evaluate
        evaluate(checkpoint, episodes=1, max_episode_steps=max_steps, gif_path=gif_path, device=device, mode=mode,
                 frame_stride=4, engine_config=engine_config)
    except Exception as exc:          # a recording problem must not end a training run
        print(f"monitor: recording {gif_path} failed: {type(exc).__name__}: {exc}", flush=True)


def dump(rows, cols=RESULT_COLUMNS) -> str:
    """pioneer/util.py:45-63: a github-style table of the chosen columns."""
    try:
        import pandas as pd
        from tabulate import tabulate
        df = rows if isinstance(rows, pd.DataFrame) else pd.DataFrame(rows)
        return tabulate(df[[c for c in cols if c in df.columns]], headers="keys", showindex=False, tablefmt="github")
    except Exception:
        return "\n".join(str({c: r.get(c) for c in cols}) for r in rows)
