"""GPU smoke of the full rollout + learn loop (BASELINE config[2] shape, small)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ppo_trainer_runs_and_checkpoints(tmp_path):
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    env = PioneerVectorEnv(2048, device="cuda:0", seed=0, engine_config=EngineConfig(max_episode_steps=40))
    cfg = PPOConfig(rollout_fragment_length=16, num_sgd_iter=2, sgd_minibatch_size=4096, lr=1e-4)
    tr = PPOTrainer(env, cfg)
    res = [tr.train() for _ in range(4)]
    last = res[-1]
    assert last["training_iteration"] == 4 and last["timesteps_total"] == 4 * 16 * 2048
    assert last["episodes_total"] >= 2048
    ended = [r for r in res if r["episodes_this_iter"] > 0]     # every env is cut by TimeLimit(40) in iteration 3
    assert ended and all(r["episode_len_mean"] <= 40 for r in ended)
    for k in ("episode_reward_mean", "episode_reward_max", "episode_reward_min"):
        assert all(math.isfinite(r[k]) for r in ended), k
    for k in ("kl", "entropy", "vf_loss", "total_loss"):
        assert math.isfinite(last[k]), k
    assert last["env_steps_per_s"] > 0
    # obs written in place by pnr_step: the rollout buffer really holds env observations
    assert torch.equal(tr.buf["raw_obs"][-1], tr.raw_obs)
    path = tr.save(str(tmp_path / "ck.pt"))
    w0 = torch.cat([p.detach().reshape(-1) for p in tr.learner.model.parameters()]).clone()
    tr2 = PPOTrainer(PioneerVectorEnv(2048, device="cuda:0", seed=0, engine_config=EngineConfig(max_episode_steps=40)), cfg)
    tr2.restore(path, restore_env=True)
    w1 = torch.cat([p.detach().reshape(-1) for p in tr2.learner.model.parameters()])
    assert torch.equal(w0, w1) and tr2.iteration == 4 and torch.equal(tr2.env.get_state(), env.get_state())
    assert float(tr2.filter.n) == float(tr.filter.n)
    env.close(); tr2.env.close()


def test_ppo_learns_to_approach_the_target():
    """A short run must raise the mean episode reward (potential-based shaping gives a dense signal)."""
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    env = PioneerVectorEnv(4096, device="cuda:0", seed=1, engine_config=EngineConfig(max_episode_steps=100))
    cfg = PPOConfig(rollout_fragment_length=100, num_sgd_iter=4, sgd_minibatch_size=16384, lr=3e-4,
                    entropy_coeff_start=1e-3, seed=1)
    tr = PPOTrainer(env, cfg)
    hist = [tr.train()["episode_reward_mean"] for _ in range(12)]
    assert hist[-1] > hist[0] + 1.0, hist
    env.close()


def test_launch_train_mirrors_reference_signature(tmp_path):
    """train(results_dir, checkpoint_freq, num_samples, num_workers, monitor) -> Tune-style rows
    (pioneer_knm_train.py:14-76; columns of cli.py:32-38) + checkpoints (checkpoint_freq, at_end)."""
    import os
    from pioneer_amd.launch import RESULT_COLUMNS, dump, train
    from pioneer_amd.ppo import PPOConfig
    df = train(results_dir=str(tmp_path), checkpoint_freq=2, num_samples=2, num_workers=1, monitor=False,
               training_iterations=3, envs_per_worker=512,
               ppo_config=PPOConfig(rollout_fragment_length=8, num_sgd_iter=1, sgd_minibatch_size=2048))
    assert len(df) == 2
    for c in RESULT_COLUMNS:
        assert c in df.columns
    assert list(df["trial_id"]) == ["00000", "00001"] and df["training_iteration"].tolist() == [3, 3]
    for t in ("00000", "00001"):
        d = tmp_path / f"PPO_Pioneer-v1_{t}"
        assert (d / "checkpoint_2.pt").exists() and (d / "checkpoint_final.pt").exists() and (d / "result.json").exists()
    assert "episode_reward_mean" in dump(df)


def test_plumbing_one_env_ppo_iteration():
    """BASELINE config[0] shape: ONE env behind the same driver (plumbing: runs, shapes/dtypes right)."""
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    env = PioneerVectorEnv(1, device="cuda:0", seed=0, engine_config=EngineConfig(max_episode_steps=50))
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=120, num_sgd_iter=2, sgd_minibatch_size=128))
    r1 = tr.train(); r2 = tr.train()
    assert r2["timesteps_total"] == 240 and r1["episodes_this_iter"] == 2 and r2["episodes_total"] == 4
    assert tr.buf["raw_obs"].shape == (120, 1, 137) and tr.buf["raw_obs"].dtype == torch.float32
    assert tr.buf["actions"].shape == (120, 1, 6) and math.isfinite(r2["total_loss"])
    env.close()


def test_graph_captured_learner_matches_eager():
    """hipGraph-captured minibatch updates (static buffers, capturable Adam) == eager updates."""
    from pioneer_amd.ppo import PPOConfig, PPOLearner, gaussian_logp
    cfg = PPOConfig(num_sgd_iter=3, sgd_minibatch_size=4096, lr=1e-3, seed=3)
    g = torch.Generator(device="cuda").manual_seed(0)
    B = 16384
    obs = torch.randn(B, 137, generator=g, device="cuda"); act = torch.randn(B, 6, generator=g, device="cuda")
    mean = torch.randn(B, 6, generator=g, device="cuda") * 0.1; log_std = torch.zeros(B, 6, device="cuda")
    batch = {"obs": obs, "actions": act, "mean": mean, "log_std": log_std, "logp": gaussian_logp(act, mean, log_std),
             "values": torch.randn(B, generator=g, device="cuda"), "adv": torch.randn(B, generator=g, device="cuda"),
             "vtarg": torch.randn(B, generator=g, device="cuda")}
    outs = []
    for use_graph in (False, True):
        L = PPOLearner(cfg, "cuda:0", use_graph=use_graph)
        gen = torch.Generator(device="cuda").manual_seed(11)
        infos = [L.update(dict(batch), gen) for _ in range(2)]           # 24 updates; the graph takes over after 3
        assert (L._graph is not None) == use_graph
        outs.append((torch.cat([p.detach().reshape(-1) for p in L.model.parameters()]), infos[-1]))
    (w0, i0), (w1, i1) = outs
    assert torch.allclose(w0, w1, atol=2e-5, rtol=1e-4)
    assert abs(i0["total_loss"] - i1["total_loss"]) < 1e-3 and abs(i0["kl"] - i1["kl"]) < 1e-4


@pytest.mark.parametrize("use_graph", [False, True])
def test_episode_statistics_exact_under_graph_replay(use_graph):
    """With TimeLimit(10) and T = 20 every env finishes exactly two 10-step episodes per iteration;
    the device-side statistics must say so in eager mode AND when the sampling loop is replayed from a
    captured hipGraph (state carried across replays has to be updated in place)."""
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    n = 1024
    env = PioneerVectorEnv(n, device="cuda:0", seed=3, engine_config=EngineConfig(max_episode_steps=10))
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=20, num_sgd_iter=1, sgd_minibatch_size=4096, lr=1e-5),
                    use_graph=use_graph)
    for it in range(1, 6):
        r = tr.train()
        early = int((tr.buf["done"] > 0).sum())                       # terminal successes end an episode early
        assert r["episodes_this_iter"] >= 2 * n and r["episodes_total"] >= 2 * n * it
        if early == 0:
            assert r["episodes_this_iter"] == 2 * n and r["episode_len_mean"] == 10.0
        assert math.isfinite(r["episode_reward_mean"]) and r["episode_reward_min"] <= r["episode_reward_mean"] <= r["episode_reward_max"]
        # the mean return of a 10-step episode is bounded by the potential scale
        assert -5.0 < r["episode_reward_mean"] < 100.0
    assert (tr._graph is not None) == use_graph
    env.close()
