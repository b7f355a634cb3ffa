"""GPU smoke of the full rollout + learn loop (BASELINE config[2] shape, small)."""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ppo_trainer_runs_and_checkpoints(tmp_path):
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    env = PioneerVectorEnv(2048, device="cuda:0", seed=0, engine_config=EngineConfig(max_episode_steps=40))
    cfg = PPOConfig(rollout_fragment_length=16, num_sgd_iter=2, sgd_minibatch_size=4096, lr=1e-4)
    tr = PPOTrainer(env, cfg)
    assert tr.learner.hip                      # PPOConfig() on a HIP device = the hand-written kernels
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
    assert tr.learner.hip
    hist = [tr.train()["episode_reward_mean"] for _ in range(12)]
    assert hist[-1] > hist[0] + 1.0, hist
    env.close()


@pytest.mark.parametrize("use_graph", [False, True])
def test_hip_trainer_learns_the_reach_task_at_the_contract_batching(use_graph):
    """The HIP loop on SURVEY 8(d) config 3's batching (T = 32, 4 epochs of 32 768-sample minibatches, 16 384 envs): within
    60 iterations = 31 M env-steps the mean episode return rises by more than 20 and episodes get shorter (targets reached)."""
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    env = PioneerVectorEnv(16384, device="cuda:0", seed=0, engine_config=EngineConfig(max_episode_steps=500))
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=32768, lr=3e-4,
                                   entropy_coeff_start=3e-3, entropy_decay_steps=100_000_000, seed=0), use_graph=use_graph)
    assert tr.learner.hip
    rows = [tr.train() for _ in range(60)]
    assert (tr._graph is not None) == use_graph
    ended = [r for r in rows if r["episodes_this_iter"] > 1000]
    first, last = ended[0], rows[-1]
    assert last["episode_reward_mean"] > first["episode_reward_mean"] + 20.0, (first["episode_reward_mean"], last["episode_reward_mean"])
    assert last["episode_len_mean"] < 0.8 * max(r["episode_len_mean"] for r in ended), [round(r["episode_len_mean"]) for r in ended[::5]]
    assert all(math.isfinite(r[k]) for r in rows for k in ("kl", "total_loss", "vf_loss", "entropy"))
    env.close()


@pytest.mark.parametrize("n_envs,max_steps,plain", [(4096, 25, False), (1000, 7, False), (777, 5, True), (65, 4, False), (1, 3, False)])
def test_resident_rollout_equals_the_two_launch_sampler(n_envs, max_steps, plain):
    """pnr_ppo_rollout (the sampler's T steps as ONE resident launch: a workgroup owns 64 envs, both nets' W2 and the env state
    stay on the CU) against T x (pnr_mlp_act, pnr_step): every buffer of the rollout — observations, actions, means, log-stds,
    values, rewards, flags, the nets' saved inputs, advantages — the env state and, after the update, the weights are identical
    over three iterations, with episodes ending (auto-reset) inside the rollouts and a last workgroup that is only partly full."""
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    cfg = PPOConfig(rollout_fragment_length=16, num_sgd_iter=2, sgd_minibatch_size=n_envs * 4, lr=3e-4, seed=4,
                    **(dict(observation_filter="NoFilter", clip_actions=False) if plain else {}))      # plain: no filter vectors, no a_max
    mk = lambda: PioneerVectorEnv(n_envs, device="cuda:0", seed=9, engine_config=EngineConfig(max_episode_steps=max_steps))   # noqa: E731
    a, b = PPOTrainer(mk(), cfg), PPOTrainer(mk(), cfg)
    assert a.resident_rollout
    b.resident_rollout = False
    for it in range(3):
        ra, rb = a.train(), b.train()
        assert torch.equal(a.raw_in, b.raw_in), it
        for k in a.buf:
            assert torch.equal(a.buf[k], b.buf[k]), (it, k)
        assert torch.equal(a.env.get_state(), b.env.get_state()), it
        assert ra["episodes_this_iter"] == rb["episodes_this_iter"]
    assert ra["episodes_total"] == rb["episodes_total"] and ra["episodes_total"] >= n_envs       # every env was cut and re-drawn at least once
    flat = lambda t: torch.cat([p.detach().reshape(-1) for p in t.learner.model.parameters()])   # noqa: E731
    assert torch.equal(flat(a), flat(b))
    a.env.close(); b.env.close()


def test_resident_rollout_refuses_what_it_does_not_implement():
    """pnr_ppo_rollout is the kinematic, env-major form: a dynamics-mode or feature-major handle, a handle that was never reset and
    T < 1 are reported as errors (PnrError) instead of being run; the trainer falls back to the per-step pair by itself."""
    from pioneer_amd import PioneerVectorEnv, EngineConfig, SimulationConfig, _lib
    from pioneer_amd.mlp import HipMLP
    from pioneer_amd.ppo import ActorCritic, PPOConfig, PPOTrainer
    dev = torch.device("cuda:0")
    N, T = 128, 2
    mlp = HipMLP(ActorCritic(PPOConfig()).to(dev), N, dev)
    mlp.pack()
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)   # noqa: E731
    def call(env, T=T):
        mlp.rollout(env, None, z(T, N, 6), None, obs=z(T + 1, N, 137), mean=z(T, N, 6), log_std=z(T, N, 6), values=z(T, N),
                    actions=z(T, N, 6), reward=z(T, N), done=z(T, N, dt=torch.uint8), truncated=z(T, N, dt=torch.uint8))
    dyn = PioneerVectorEnv(N, device=dev, seed=0, simulation_config=SimulationConfig(gravity=9.81), engine_config=EngineConfig(mode="dynamic"))
    dyn.reset()
    with pytest.raises(_lib.PnrError, match="kinematic"):
        call(dyn)
    fm = PioneerVectorEnv(N, device=dev, seed=0, engine_config=EngineConfig(obs_layout="feature_major"))
    fm.reset()
    with pytest.raises(_lib.PnrError, match="env-major"):
        call(fm)
    fresh = PioneerVectorEnv(N, device=dev, seed=0)
    with pytest.raises(_lib.PnrError, match="before the first pnr_reset"):
        call(fresh)
    fresh.reset()
    call(fresh)                                                  # and the plain case runs
    torch.cuda.synchronize()
    assert not PPOTrainer(dyn, PPOConfig(rollout_fragment_length=2, sgd_minibatch_size=N)).resident_rollout
    for e in (dyn, fm, fresh):
        e.close()


def test_hip_trainer_restore_continues_bit_identically(tmp_path):
    """save -> restore(restore_env=True) into a FRESH trainer -> the next iteration equals the uninterrupted run bit for
    bit: master weights, Adam's moments and update count (`hip_adam`), the shuffle's epoch counter, filter, KL coefficient,
    env shards, running episode accumulators and the noise generator all travel."""
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    cfg = PPOConfig(rollout_fragment_length=16, num_sgd_iter=2, sgd_minibatch_size=8192, lr=3e-4, seed=4)
    mk = lambda: PioneerVectorEnv(2048, device="cuda:0", seed=9, engine_config=EngineConfig(max_episode_steps=25))   # noqa: E731
    a = PPOTrainer(mk(), cfg)
    for _ in range(3):
        a.train()
    path = a.save(str(tmp_path / "ck.pt"))
    ra = a.train()
    b = PPOTrainer(mk(), cfg)
    b.restore(path, restore_env=True)
    assert b.learner.hip and b.iteration == 3
    rb = b.train()
    flat = lambda t: torch.cat([p.detach().reshape(-1) for p in t.learner.model.parameters()])   # noqa: E731
    assert torch.equal(flat(a), flat(b))
    for x, y in zip(a.learner.hip_mlp(1).adam_state(), b.learner.hip_mlp(1).adam_state()):
        assert torch.equal(x, y)
    assert float(a.learner.hip_mlp(1).adam_state()[2]) == 4 * 2 * 4              # 4 iterations x 2 epochs x 4 minibatches
    assert torch.equal(a.env.get_state(), b.env.get_state()) and torch.equal(a.raw_obs, b.raw_obs)
    for k in ("kl", "total_loss", "vf_loss", "policy_loss", "entropy", "episode_reward_mean", "episodes_total", "cur_kl_coeff"):
        assert ra[k] == rb[k] or (ra[k] != ra[k] and rb[k] != rb[k]), (k, ra[k], rb[k])
    # a checkpoint restores into the formulation that wrote it: the other one raises instead of dropping Adam's moments
    import dataclasses
    t = PPOTrainer(mk(), dataclasses.replace(cfg, hip_kernels=False))
    assert not t.learner.hip
    with pytest.raises(AssertionError, match="optimiser state"):
        t.restore(path)
    t.train()
    tpath = t.save(str(tmp_path / "ck_torch.pt"))
    with pytest.raises(AssertionError, match="optimiser state"):
        b.restore(tpath)
    for e in (a.env, b.env, t.env):
        e.close()


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
        # Tune's trial-directory files
        import csv as _csv, json as _json
        params = _json.load(open(d / "params.json"))
        assert params["env"] == "Pioneer-v1" and params["env_config"]["award_done"] == 5.0 and "lr" in params
        prog = list(_csv.DictReader(open(d / "progress.csv")))
        assert len(prog) == len(open(d / "result.json").read().strip().splitlines()) and "episode_reward_mean" in prog[0]
        from pioneer_amd.tb import read_events
        (evf,) = [f for f in os.listdir(d) if f.startswith("events.out.tfevents.")]
        ev = read_events(str(d / evf))
        assert [s_ for s_, _ in ev] == [0, 1, 2, 3] and "ray/tune/timesteps_total" in ev[-1][1]
    assert "episode_reward_mean" in dump(df)


def test_plumbing_one_env_ppo_iteration():
    """BASELINE config[0] shape: ONE env behind the same driver (plumbing: runs, shapes/dtypes right)."""
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    env = PioneerVectorEnv(1, device="cuda:0", seed=0, engine_config=EngineConfig(max_episode_steps=50))
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=120, num_sgd_iter=2, sgd_minibatch_size=128))
    assert tr.learner.hip
    r1 = tr.train(); r2 = tr.train()
    assert r2["timesteps_total"] == 240 and r1["episodes_this_iter"] == 2 and r2["episodes_total"] == 4
    assert tr.buf["raw_obs"].shape == (120, 1, 137) and tr.buf["raw_obs"].dtype == torch.float32
    assert tr.buf["actions"].shape == (120, 1, 6) and math.isfinite(r2["total_loss"])
    env.close()


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
    assert (tr._graph is not None) == use_graph and tr.learner.hip
    env.close()


def test_torch_formulation_trainer_still_runs_on_the_gpu():
    """hip_kernels=False (or nets of another shape): the float32 torch formulation, sampled eagerly; same result columns and
    the same exact episode statistics."""
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    n = 512
    for cfg in (PPOConfig(rollout_fragment_length=20, num_sgd_iter=1, sgd_minibatch_size=2048, hip_kernels=False),
                PPOConfig(rollout_fragment_length=20, num_sgd_iter=1, sgd_minibatch_size=2048, fcnet_hiddens=(64, 64))):
        env = PioneerVectorEnv(n, device="cuda:0", seed=3, engine_config=EngineConfig(max_episode_steps=10))
        tr = PPOTrainer(env, cfg, use_graph=True)
        assert not tr.learner.hip and not tr.use_graph
        for it in range(1, 3):
            r = tr.train()
            assert r["episodes_this_iter"] >= 2 * n and math.isfinite(r["total_loss"]) and math.isfinite(r["kl"])
            if int((tr.buf["done"] > 0).sum()) == 0:
                assert r["episodes_this_iter"] == 2 * n and r["episode_len_mean"] == 10.0
        env.close()


class FusedPPOLoss(torch.autograd.Function):
    """pnr_ppo_loss (the stand-alone loss kernel the autograd binding of pioneer_amd.mlp uses) behind autograd, for the test
    below: forward values and d loss / d head from ONE launch."""

    @staticmethod
    def forward(ctx, head_p, head_v, mb, kl_c, ent_c, clip, vf_clip, vf_coeff):
        import ctypes as C
        from pioneer_amd import _lib
        lib = _lib.load_library()
        B = head_p.shape[0]
        head_p, head_v = head_p.contiguous(), head_v.contiguous()
        t = {k: mb[k].contiguous() for k in ("actions", "logp", "mean", "log_std", "adv", "vtarg", "values")}
        g_p, g_v = torch.empty_like(head_p), torch.empty_like(head_v)
        rows = (B + 255) // 256
        partials = torch.empty((rows, 8), dtype=torch.float32, device=head_p.device)
        means = torch.empty(8, dtype=torch.float32, device=head_p.device)   # policy_loss, vf_loss, kl, entropy, total
        P = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
        _lib.check(lib.pnr_ppo_loss(B, None, P(head_p), P(head_v), P(t["actions"]), P(t["logp"]), P(t["mean"]), P(t["log_std"]),
                                    P(t["adv"]), P(t["vtarg"]), P(t["values"]), P(kl_c), P(ent_c),
                                    C.c_float(clip), C.c_float(vf_clip), C.c_float(vf_coeff), P(g_p), P(g_v), P(partials),
                                    rows, P(means), C.c_void_p(torch.cuda.current_stream(head_p.device).cuda_stream)))
        ctx.save_for_backward(g_p, g_v)
        ctx.mark_non_differentiable(means)
        return means[4].clone(), means

    @staticmethod
    def backward(ctx, g_total, _g_means):
        g_p, g_v = ctx.saved_tensors
        return g_p * g_total, g_v * g_total, None, None, None, None, None, None


def _torch_loss_from_heads(hp, hv, mb, kl_c, ent_c, cfg):
    """PPOLearner.loss()'s arithmetic (torch ops + autograd), starting from the raw head rows."""
    from pioneer_amd.ppo import gaussian_entropy, gaussian_kl, gaussian_logp
    mean, log_std, v = hp[:, :6], torch.clamp(hp[:, 6:12], -20.0, 2.0), hv[:, 0]
    logp = gaussian_logp(mb["actions"], mean, log_std)
    ratio = torch.exp(logp - mb["logp"])
    adv = mb["adv"]
    surr = torch.minimum(adv * ratio, adv * torch.clamp(ratio, 1 - cfg.clip_param, 1 + cfg.clip_param))
    kl = gaussian_kl(mb["mean"], mb["log_std"], mean, log_std)
    ent = gaussian_entropy(log_std)
    vf1 = (v - mb["vtarg"]) ** 2
    v_clipped = mb["values"] + torch.clamp(v - mb["values"], -cfg.vf_clip_param, cfg.vf_clip_param)
    vf = torch.maximum(vf1, (v_clipped - mb["vtarg"]) ** 2)
    total = (-surr + kl_c * kl + cfg.vf_loss_coeff * vf - ent_c * ent).mean()
    return total, torch.stack([(-surr).mean(), vf.mean(), kl.mean(), ent.mean(), total])


@pytest.mark.parametrize("B", [1, 255, 4099, 131072])
def test_fused_loss_matches_autograd(B):
    """pnr_ppo_loss (one HIP kernel, forward + backward) against the torch-ops loss under autograd: values and
    d loss / d head, with ratios on both sides of the clip range, active value clipping, log-stds beyond the
    clamp and exact ties (ratio == 1, v == v_old)."""
    from pioneer_amd.ppo import PPOConfig
    cfg = PPOConfig()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(B)
    R = lambda *s: torch.randn(*s, generator=g, device=dev)   # noqa: E731
    hp = torch.zeros(B, 16, device=dev); hv = torch.zeros(B, 16, device=dev)
    hp[:, :6] = R(B, 6); hp[:, 6:12] = 0.7 * R(B, 6) - 0.5; hv[:, 0] = 3 * R(B)
    hp[:, 12:] = R(B, 4); hv[:, 1:] = R(B, 15)                       # padding columns must not matter
    if B > 8:
        hp[0, 6:12] = torch.tensor([2.5, -21.0, 2.0, -20.0, 0.0, 1.0], device=dev)   # beyond / on the clamp
    mb = {"actions": hp[:, :6].detach() + 1.5 * R(B, 6), "mean": hp[:, :6].detach() + 0.3 * R(B, 6),
          "log_std": torch.clamp(hp[:, 6:12].detach() + 0.2 * R(B, 6), -20, 2), "adv": R(B), "vtarg": 3 * R(B),
          "values": hv[:, 0].detach() + 4 * R(B)}
    if B > 8:   # keep the row with the extreme log-stds well conditioned: z = 0 and the old policy equal to the new one
        mb["actions"][0] = hp[0, :6]; mb["mean"][0] = hp[0, :6]; mb["log_std"][0] = torch.clamp(hp[0, 6:12], -20, 2)
    with torch.no_grad():
        from pioneer_amd.ppo import gaussian_logp
        lp = gaussian_logp(mb["actions"], hp[:, :6], torch.clamp(hp[:, 6:12], -20, 2))
    mb["logp"] = lp + 0.4 * R(B)                                     # ratios spread over ~[0.3, 3]
    if B > 8:
        mb["logp"][1] = lp[1]                                        # ratio exactly 1: the minimum's tie
        mb["values"][2] = hv[2, 0]                                   # v == v_old: the maximum's tie
    kl_c = torch.tensor(0.37, device=dev); ent_c = torch.tensor(0.013, device=dev)

    a_p, a_v = hp.clone().requires_grad_(True), hv.clone().requires_grad_(True)
    ref_total, ref_means = _torch_loss_from_heads(a_p, a_v, mb, kl_c, ent_c, cfg)
    ref_total.backward()
    b_p, b_v = hp.clone().requires_grad_(True), hv.clone().requires_grad_(True)
    total, means = FusedPPOLoss.apply(b_p, b_v, mb, kl_c, ent_c, float(cfg.clip_param), float(cfg.vf_clip_param),
                                      float(cfg.vf_loss_coeff))
    (2.0 * total).backward()                                         # the upstream factor must reach the heads
    assert torch.allclose(means[:5], ref_means, rtol=2e-5, atol=1e-5), (means, ref_means)
    # fp32 against fp32 with different summation orders: compared per sample, relative to that sample's largest
    # gradient entry (a few samples have ratios far outside the clip range and gradients 100x the typical one)
    for fused, ref in ((b_p.grad / 2, a_p.grad), (b_v.grad / 2, a_v.grad)):
        row = ref.abs().max(1, keepdim=True).values
        err = (fused - ref).abs()
        bad = int((err > 2e-4 * row + 1e-9).any(1).sum())
        # a sample whose ratio (or value step) lands within rounding of a clip boundary may fall on different
        # sides in the two implementations: the gradient is discontinuous there.  Allow a handful in 131 072.
        assert bad <= B // 20000, (bad, float(err.max()), float(row.max()))
    assert float(b_p.grad[:, 12:].abs().max()) == 0.0 and float(b_v.grad[:, 1:].abs().max()) == 0.0
    if B > 8:
        assert float(b_p.grad[0, 6].abs()) == 0.0 and float(b_p.grad[0, 7].abs()) == 0.0     # clamped: no gradient
        assert float(b_p.grad[0, 8].abs()) > 0.0 and float(b_p.grad[0, 9].abs()) > 0.0       # on the bound: passes


def test_hip_and_torch_learners_report_the_same_losses():
    """Two learners from the same seed, one on the kernels (bf16 operands) and one on torch (float32), one full-batch update
    each: the reported loss means are those of the SAME weights on the same batch and agree to bf16 accuracy, and both updates
    move the weights in the same direction."""
    from pioneer_amd.ppo import PPOConfig, PPOLearner, gaussian_logp
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    B = 8192
    R = lambda *s_: torch.randn(*s_, generator=g, device=dev)   # noqa: E731
    act, mean, log_std = R(B, 6), 0.1 * R(B, 6), 0.1 * R(B, 6)
    batch = {"obs": R(B, 137), "actions": act, "mean": mean, "log_std": log_std, "logp": gaussian_logp(act, mean, log_std) + 0.2 * R(B),
             "values": R(B), "adv": R(B), "vtarg": R(B)}
    outs = []
    for hip in (False, True):
        L = PPOLearner(PPOConfig(num_sgd_iter=1, sgd_minibatch_size=B, lr=1e-3, seed=11, hip_kernels=hip), dev)
        assert L.hip == hip
        w0 = torch.cat([p.detach().reshape(-1) for p in L.model.parameters()]).clone()
        info = L.update(dict(batch))
        outs.append((torch.cat([p.detach().reshape(-1) for p in L.model.parameters()]) - w0, info))
    (d0, i0), (d1, i1) = outs
    for k in ("policy_loss", "vf_loss", "kl", "entropy", "total_loss"):
        assert abs(i0[k] - i1[k]) <= 2e-2 * max(1.0, abs(i0[k])), (k, i0[k], i1[k])
    # Adam's first step is lr * sign(g): the two gradients agree in sign wherever they are not tiny
    assert float((torch.sign(d0) == torch.sign(d1)).float().mean()) > 0.9


def test_evaluate_checkpoint_and_record_gif(tmp_path):
    """The reference's eval script role (temp/pioneer_eval.py:53-78): restore a checkpoint, roll the policy out in
    the single-env façade, record the frames."""
    from PIL import Image
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.evaluate import evaluate
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    env = PioneerVectorEnv(512, device="cuda:0", seed=1, engine_config=EngineConfig(max_episode_steps=20))
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=8, num_sgd_iter=1, sgd_minibatch_size=2048))
    assert tr.learner.hip
    tr.train(); tr.train()
    ck = tr.save(str(tmp_path / "ck.pt"))
    env.close()
    gif = tmp_path / "eval.gif"
    res = evaluate(ck, episodes=2, max_episode_steps=12, gif_path=str(gif), frame_stride=3)
    assert len(res["episode_rewards"]) == 2 and all(1 <= n <= 12 for n in res["episode_lengths"])
    assert all(np.isfinite(r) for r in res["episode_rewards"]) and res["frames"] >= 4
    im = Image.open(gif)
    assert im.is_animated and im.n_frames == res["frames"] and im.size == (1280, 800)     # RenderConfig defaults
    # the stochastic policy runs too, and in dynamics mode
    res2 = evaluate(ck, episodes=1, max_episode_steps=5, mode="dynamic", deterministic=False)
    assert res2["episode_lengths"] == [5] or res2["successes"][0]


def test_gae_logp_kernel_equals_the_host_formulas():
    """pnr_ppo_gae against compute_gae (bit for bit: same float32 operations in the same order) and gaussian_logp (to
    the rounding of exp), ragged N, with terminals of both kinds and lambda < 1."""
    from pioneer_amd.ppo import compute_gae, gaussian_logp, hip_gae_logp
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(4)
    T, N = 32, 4099
    r = torch.randn(T, N, generator=g, device=dev)
    v = torch.randn(T, N, generator=g, device=dev) * 3
    last = torch.randn(N, generator=g, device=dev)
    done = (torch.rand(T, N, generator=g, device=dev) < 0.05).to(torch.uint8)
    trunc = ((torch.rand(T, N, generator=g, device=dev) < 0.05) & (done == 0)).to(torch.uint8)
    mean = torch.randn(T, N, 6, generator=g, device=dev)
    log_std = torch.rand(T, N, 6, generator=g, device=dev) * 4 - 3
    act = mean + torch.exp(log_std) * torch.randn(T, N, 6, generator=g, device=dev)
    for gamma, lam in ((0.99, 1.0), (0.97, 0.9)):
        out = {k: torch.full((T, N), float("nan"), device=dev) for k in ("logp", "adv", "vtarg", "terminals")}
        hip_gae_logp(r, v, last, done, trunc, act, mean, log_std, gamma, lam, **out)
        term = (done | trunc).float()
        adv, vt = compute_gae(r, v, last, term, gamma, lam)
        assert torch.equal(out["terminals"], term)
        assert torch.equal(out["adv"], adv) and torch.equal(out["vtarg"], vt)
        lp = gaussian_logp(act, mean, log_std)
        assert float((out["logp"] - lp).abs().max()) <= 2e-5 * float(lp.abs().max())
        assert float(((out["logp"] - lp).abs() / (lp.abs() + 1)).max()) < 5e-6


def test_gae_kernel_bookkeeping_equals_episode_stats_and_advantage_moments():
    """The optional bookkeeping of pnr_ppo_gae over three consecutive rollouts (carry-over of running returns included,
    one rollout without any terminal, terminals on the first and last step) against EpisodeStats.step() on the same
    data, and the advantages' moments against float64 sums."""
    from pioneer_amd.ppo import EpisodeStats, compute_gae, hip_gae_logp
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(8)
    T, N = 32, 4099
    a, b = EpisodeStats(N, dev), EpisodeStats(N, dev)
    adv_stats = torch.zeros(3, dtype=torch.float64, device=dev)
    for it in range(3):
        r = torch.randn(T, N, generator=g, device=dev)
        v = torch.randn(T, N, generator=g, device=dev)
        last = torch.randn(N, generator=g, device=dev)
        p = 0.0 if it == 1 else 0.06
        done = (torch.rand(T, N, generator=g, device=dev) < p).to(torch.uint8)
        trunc = ((torch.rand(T, N, generator=g, device=dev) < p) & (done == 0)).to(torch.uint8)
        if it == 2:
            done[0] = 1; trunc[0] = 0; done[-1] = 0; trunc[-1] = 1
        act = torch.randn(T, N, 6, generator=g, device=dev); z = torch.zeros(T, N, 6, device=dev)
        out = {k: torch.empty(T, N, device=dev) for k in ("logp", "adv", "vtarg")}
        hip_gae_logp(r, v, last, done, trunc, act, z, z, 0.99, 0.95, stats=b, adv_stats=adv_stats, **out)
        term = (done | trunc).float()
        for t in range(T):
            a.step(r[t], term[t])
        assert torch.equal(a.ret, b.ret) and torch.equal(a.len, b.len)            # same float32 accumulation per env
        assert float(a.w_cnt) == float(b.w_cnt) and float(a.w_len) == float(b.w_len)
        # (step() sums the ended episodes' float32 returns in float32 per step, the kernel in float64)
        assert abs(float(a.w_sum) - float(b.w_sum)) <= 1e-6 * float(a.w_cnt) + 1e-9
        if float(a.w_cnt) > 0:
            assert float(a.w_max) == float(b.w_max) and float(a.w_min) == float(b.w_min)
        adv = compute_gae(r, v, last, term, 0.99, 0.95)[0]
        assert torch.equal(out["adv"], adv)
        want = torch.stack([adv.double().sum(), (adv.double() ** 2).sum(), torch.tensor(float(T * N), dtype=torch.float64, device=dev)])
        assert torch.allclose(adv_stats, want, rtol=1e-12, atol=1e-9)
        ra, rb = a.summarize(), b.summarize()
        assert all((ra[k] == rb[k]) or (ra[k] != ra[k] and rb[k] != rb[k]) or abs(ra[k] - rb[k]) <= 1e-6 for k in ra)


@pytest.mark.parametrize("n", [1, 2, 5, 4099, 32768, 524288, 1000003])
def test_permutation_kernel_is_a_permutation(n):
    from pioneer_amd.ppo import hip_permutation
    dev = torch.device("cuda", 0)
    out = torch.full((n + 3,), -7, dtype=torch.int64, device=dev)
    p0 = hip_permutation(n, 11, 0, out).clone()
    assert bool((out[n:] == -7).all())
    assert torch.equal(torch.sort(p0).values, torch.arange(n, device=dev))
    p1 = hip_permutation(n, 11, 1, out).clone()
    p0b = hip_permutation(n, 11, 0, out).clone()
    assert torch.equal(p0, p0b)                                  # a function of (seed, stream id)
    if n >= 4099:
        assert not torch.equal(p0, p1)
        # no visible structure: few fixed points, first-difference sign changes like a random sequence (2/3 of the positions)
        assert int((p0 == torch.arange(n, device=dev)).sum()) < 12
        d = torch.sign(p0[1:] - p0[:-1])
        turns = float((d[1:] != d[:-1]).float().mean())
        assert abs(turns - 2.0 / 3.0) < 0.03, turns
        # each quarter of the output draws evenly from the input range
        q = p0[: n // 4].double().mean() / n
        assert abs(float(q) - 0.5) < 0.02


@pytest.mark.parametrize("rows", [1, 511, 4099, 32 * 2048])
def test_filter_moment_kernel_equals_the_float64_sums(rows):
    """MeanStdFilter.observe on the GPU (pnr_filter_moments) against float64 column sums of the same deviations: relative
    error of a float32 partial sum over <= 512 rows; constant columns (36 of the 137 observation entries) exactly 0; two
    calls accumulate; sync() then gives the float64 mean and variance."""
    from pioneer_amd.ppo import MeanStdFilter
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(rows)
    x = torch.randn(rows, 137, generator=g, device=dev) * 3.0 + torch.linspace(-50, 50, 137, device=dev)
    x[:, 18:54] = torch.linspace(-3.1415927, 3.1415927, 36, device=dev)          # the constant block
    y = torch.randn(rows, 137, generator=g, device=dev)
    y[:, 18:54] = x[0, 18:54]
    f = MeanStdFilter(137, dev)
    f.observe(x); f.observe(y)
    piv = x[0].double()
    d = torch.cat([x, y]).double() - piv
    assert float(f._dn) == 2 * rows
    for got, want in ((f._dsum, d.sum(0)), (f._dsq, (d * d).sum(0))):
        assert bool((got[18:54] == 0).all())
        scale = d.abs().sum(0) if got is f._dsum else (d * d).sum(0)
        assert float(((got - want).abs() / (scale + 1e-30)).max()) < 2e-6
    f.sync()
    allx = torch.cat([x, y]).double()
    assert bool(((f.mean - allx.mean(0)).abs() <= 4e-6 * d.abs().mean(0) + 1e-12).all())    # the sums' bound, divided by n
    if rows > 1:
        assert torch.allclose(f.std, allx.std(0, unbiased=True), rtol=2e-5, atol=1e-9)
        assert bool((f.std[18:54] == 0).all())


def test_filter_merge_kernel_equals_the_host_formulation_bit_for_bit(monkeypatch):
    """MeanStdFilter.sync() through pnr_filter_merge against the element-wise float64 formulation on the same pending
    deltas, over three syncs (first merge into empty statistics, then two more), constant columns included."""
    from pioneer_amd import ppo
    from pioneer_amd.ppo import MeanStdFilter
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    a, b = MeanStdFilter(137, dev), MeanStdFilter(137, dev)
    for it in range(3):
        x = torch.randn(4099, 137, generator=g, device=dev) * (it + 1) + 5.0 * it
        x[:, 18:54] = 1.25
        a.observe(x); b.observe(x)
        a.sync()                                               # the kernel
        monkeypatch.setattr(ppo.pdist, "is_dist", lambda: True)         # forces the host formulation (its all-reduces are no-ops on one rank)
        monkeypatch.setattr(ppo.pdist, "allreduce_sum_", lambda t: t)
        b.sync()
        monkeypatch.undo()
        assert torch.equal(a.n, b.n) and torch.equal(a.mean, b.mean) and torch.equal(a.m2, b.m2)
        assert float(a._dn) == 0.0 and not bool(a._dsum.any()) and not bool(a._dsq.any())
    assert bool((a.std[18:54] == 0).all())


@pytest.mark.parametrize("clip", [10.0, 0.0])
def test_filter_prepare_kernel_equals_the_host_formulation_bit_for_bit(clip):
    """MeanStdFilter.prepare() through pnr_filter_prepare against the element-wise float64 formulation it replaces: the identity
    before two samples exist, then mean / 1 / (std + 1e-8) / +-clip (or +-inf) rounded to float32 once — constant columns (std 0) included."""
    from pioneer_amd.ppo import MeanStdFilter
    dev = torch.device("cuda", 0)
    f = MeanStdFilter(137, dev, clip=clip)

    def expected():
        ident = f.n < 2
        c = clip if clip else float("inf")
        loc = torch.where(ident, torch.zeros_like(f.mean), f.mean).float()
        inv = torch.where(ident, torch.ones_like(f.mean), 1.0 / (f.std + 1e-8)).float()
        hi = torch.where(ident, torch.full_like(f.mean, float("inf")), torch.full_like(f.mean, c)).float()
        return loc, inv, -hi, hi

    g = torch.Generator(device=dev).manual_seed(3)
    for it in range(3):
        f.prepare()
        for got, want in zip((f._loc, f._inv, f._lo, f._hi), expected()):
            assert torch.equal(got, want), it
        x = torch.randn(2048, 137, generator=g, device=dev) * (0.01 + it) + 3.0 * it
        x[:, 18:54] = -0.75
        f.observe(x); f.sync()
    f.prepare()
    assert bool((f._inv[18:54] == 1e8).all()) and float(f._loc[20]) == -0.75


RESTORE_CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, os.environ["PNR_ROOT"])
from pioneer_amd import PioneerVectorEnv, EngineConfig
from pioneer_amd.ppo import PPOConfig, PPOTrainer
cfg = PPOConfig(rollout_fragment_length=16, num_sgd_iter=2, sgd_minibatch_size=8192, lr=3e-4, seed=4)
env = PioneerVectorEnv(2048, device="cuda:0", seed=9, engine_config=EngineConfig(max_episode_steps=25))
tr = PPOTrainer(env, cfg, use_graph=True)
tr.restore(os.environ["PNR_CK"], restore_env=True)
it0 = tr.iteration
rows, graphs = [], []
for _ in range(3):
    rows.append(tr.train()); graphs.append(tr._graph is not None)
w = torch.cat([p.detach().reshape(-1).double().cpu() for p in tr.learner.model.parameters()])
json.dump({"iteration_at_restore": it0, "graph_after_each_train": graphs, "kl": [r["kl"] for r in rows], "total_loss": [r["total_loss"] for r in rows],
           "wsum": float(w.sum()), "wabs": float(w.abs().sum())}, open(os.environ["PNR_OUT"], "w"))
env.close()
'''


def test_restore_with_graph_capture_in_a_fresh_process_warms_up_eagerly_first(tmp_path):
    """restore() sets `iteration` from the checkpoint, and the graph capture used to be gated on it — a fresh
    process that restored would have captured the hipGraph on its very first collect, with the first launches and scratch
    allocations inside stream capture.  The gate is the trainer object's own count of eager collects: the first train() after a
    restore runs eagerly, the second captures, and the run equals the uninterrupted (eager) one bit for bit."""
    import json, os, subprocess, sys
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    cfg = PPOConfig(rollout_fragment_length=16, num_sgd_iter=2, sgd_minibatch_size=8192, lr=3e-4, seed=4)
    env = PioneerVectorEnv(2048, device="cuda:0", seed=9, engine_config=EngineConfig(max_episode_steps=25))
    a = PPOTrainer(env, cfg)
    for _ in range(2):
        a.train()
    path = a.save(str(tmp_path / "ck.pt"))
    ref = [a.train() for _ in range(3)]
    w = torch.cat([p.detach().reshape(-1).double().cpu() for p in a.learner.model.parameters()])
    env.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "child.py"
    script.write_text(RESTORE_CHILD)
    out = tmp_path / "child.json"
    res = subprocess.run([sys.executable, str(script)], env=dict(os.environ, PNR_ROOT=root, PNR_CK=path, PNR_OUT=str(out)),
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    r = json.load(open(out))
    assert r["iteration_at_restore"] == 2
    import numpy as np
    assert r["graph_after_each_train"] == [False, True, True]        # eager warm-up by THIS object, then capture + replay
    # the eager iteration continues the uninterrupted run bit for bit; from the capture on the action noise comes from the graph's
    # own generator state (re-seeded at capture, PPOTrainer.collect), so later iterations are another — equally valid — sample path
    assert r["kl"][0] == ref[0]["kl"] and r["total_loss"][0] == ref[0]["total_loss"]
    assert all(np.isfinite(x) for x in r["kl"] + r["total_loss"]) and np.isfinite(r["wsum"])
    assert abs(r["wabs"] - float(w.abs().sum())) <= 1e-2 * float(w.abs().sum())


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_hip_trainer_with_float32_accurate_kernels(precision):
    """PPOConfig(hip_kernels="f32"): the whole loop on the hand-written kernels with every MFMA operand as two scaled fp16 planes (the
    reference's learner is float32 torch, pioneer_knm_train.py:47).  The sampler runs per step (pnr_mlp_act + pnr_step: the resident
    rollout kernel is bf16-only), the learner gathers its input planes from the float32 observations.  (a) The first rollout — same
    initial weights, same noise stream — equals the float32 torch formulation's to float32 rounding; (b) the run learns, graph-captured."""
    import dataclasses
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    cfg = PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=16384, lr=3e-4, entropy_coeff_start=1e-3, seed=1,
                    hip_kernels=precision)
    mk = lambda: PioneerVectorEnv(4096, device="cuda:0", seed=1, engine_config=EngineConfig(max_episode_steps=100))   # noqa: E731
    tr = PPOTrainer(mk(), cfg, use_graph=True)
    assert tr.learner.hip and not tr.resident_rollout and tr.sample_mlp.planes == cfg.mlp_planes() == {"f32": 2, "bf16x3": 3}[precision]
    ref = PPOTrainer(mk(), dataclasses.replace(cfg, hip_kernels=False))
    assert not ref.learner.hip
    tr._collect_impl(); ref._collect_impl()
    torch.cuda.synchronize()
    tol = 2e-5
    for k in ("mean", "log_std", "values"):
        a, b = tr.buf[k][0].double(), ref.buf[k][0].double()          # step 0: the same observations through both forwards
        assert float((a - b).norm() / b.norm()) <= tol, (k, float((a - b).norm() / b.norm()))
    rows = [tr.train() for _ in range(12)]
    assert tr._graph is not None
    ret = [r["episode_reward_mean"] for r in rows if math.isfinite(r["episode_reward_mean"])]     # (episodes end every 100 steps = ~3 iterations)
    assert len(ret) >= 3 and ret[-1] > ret[0] + 1.0, ret
    assert all(math.isfinite(r[k]) for r in rows for k in ("kl", "total_loss", "vf_loss", "entropy"))
    assert float(tr.learner.hip_mlp(1).adam_state()[2]) == 12 * 4 * (32 * 4096 // 16384)
    tr.env.close(); ref.env.close()
