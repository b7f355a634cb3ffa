"""CPU tests of the host driver and its N>1 path (gloo, world_size 2; the GPU path uses the same
code over RCCL).  Envs need a GPU, so rollouts here are synthetic tensors; what is covered is
everything that crosses ranks: shard arithmetic, the flat gradient all-reduce (== one process
on the concatenated batch), the obs-filter moment merge, metric reductions."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pioneer_amd import dist as pdist                       # noqa: E402
from pioneer_amd.ppo import (ActorCritic, MeanStdFilter, PPOConfig, PPOLearner, compute_gae, EpisodeStats,  # noqa: E402
                             gaussian_entropy, gaussian_kl, gaussian_logp, sample_entropy_start)


def test_shard_range_partitions_the_env_axis():
    for total in (65536, 1000, 7):
        for world in (1, 2, 3, 8):
            spans = [pdist.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert pdist.shard_range(65536, 8, 3) == (24576, 8192)          # 8 192 envs per GPU
    with pytest.raises(AssertionError):
        pdist.shard_range(10, 2, 2)


def test_model_parameter_count_matches_survey():
    m = ActorCritic(PPOConfig())
    n_pol = sum(p.numel() for p in m.policy.parameters())
    n_val = sum(p.numel() for p in m.value.parameters())
    assert (n_pol, n_val, n_pol + n_val) == (104204, 101377, 205581)


def test_gaussian_helpers_against_torch_distributions():
    torch.manual_seed(0)
    mean, log_std = torch.randn(5, 6), torch.randn(5, 6) * 0.3
    mean1, log_std1 = torch.randn(5, 6), torch.randn(5, 6) * 0.3
    x = torch.randn(5, 6)
    d0 = torch.distributions.Normal(mean, log_std.exp()); d1 = torch.distributions.Normal(mean1, log_std1.exp())
    assert torch.allclose(gaussian_logp(x, mean, log_std), d0.log_prob(x).sum(-1), atol=1e-5)
    assert torch.allclose(gaussian_entropy(log_std), d0.entropy().sum(-1), atol=1e-5)
    assert torch.allclose(gaussian_kl(mean, log_std, mean1, log_std1), torch.distributions.kl_divergence(d0, d1).sum(-1), atol=1e-5)


def test_gae_terminal_masking():
    r = torch.tensor([[1.0], [1.0], [1.0]]); v = torch.tensor([[0.5], [0.5], [0.5]])
    term = torch.tensor([[0.0], [1.0], [0.0]])
    adv, vt = compute_gae(r, v, torch.tensor([2.0]), term, gamma=0.9, lam=1.0)
    # step 2 bootstraps from last_value; step 1 is terminal (no bootstrap, no carry); step 0 carries step 1
    a2 = 1 + 0.9 * 2.0 - 0.5; a1 = 1 - 0.5; a0 = (1 + 0.9 * 0.5 - 0.5) + 0.9 * a1
    assert torch.allclose(adv.squeeze(), torch.tensor([a0, a1, a2]))
    assert torch.allclose(vt, adv + v)


def test_entropy_schedule_and_sampler():
    rng = np.random.RandomState(0)
    xs = [sample_entropy_start(rng) for _ in range(500)]
    assert min(xs) >= 1e-3 and max(xs) <= 1e-1 and 5e-3 < np.exp(np.mean(np.log(xs))) < 2e-2   # log-uniform
    L = PPOLearner(PPOConfig(entropy_coeff_start=0.05, entropy_decay_steps=1000), "cpu")
    assert L.entropy_coeff() == 0.05
    L.timesteps_total = 500;  assert abs(L.entropy_coeff() - 0.025) < 1e-12
    L.timesteps_total = 5000; assert L.entropy_coeff() == 0.0


def test_episode_stats():
    st = EpisodeStats(3, "cpu")
    st.step(torch.tensor([1.0, 2.0, 3.0]), torch.tensor([0.0, 0.0, 1.0]))
    st.step(torch.tensor([1.0, 2.0, 3.0]), torch.tensor([1.0, 0.0, 0.0]))
    out = st.summarize()
    assert out["episodes_this_iter"] == 2 and out["episode_reward_max"] == 3.0 and out["episode_reward_min"] == 2.0
    assert out["episode_reward_mean"] == 2.5 and out["episode_len_mean"] == 1.5 and out["episodes_total"] == 2


def make_batch(B, seed):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(B, 137, generator=g)
    act = torch.randn(B, 6, generator=g)
    mean = torch.randn(B, 6, generator=g) * 0.1
    log_std = torch.zeros(B, 6)
    return {"obs": obs, "actions": act, "mean": mean, "log_std": log_std,
            "logp": gaussian_logp(act, mean, log_std), "values": torch.randn(B, generator=g),
            "adv": torch.randn(B, generator=g), "vtarg": torch.randn(B, generator=g)}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, w = pdist.init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and pdist.is_dist()
    # 1) gradient all-reduce == single process on the concatenated batch
    cfg = PPOConfig(num_sgd_iter=1, sgd_minibatch_size=64, lr=1e-3, seed=3)
    L = PPOLearner(cfg, "cpu")                     # broadcast makes weights identical
    full = make_batch(128, seed=11)
    mine = {k: v[rank * 64:(rank + 1) * 64] for k, v in full.items()}
    loss, _ = L.loss(mine)
    L.opt.zero_grad(); loss.backward()
    pdist.allreduce_mean_grads(L.model.parameters())
    g = torch.cat([p.grad.reshape(-1) for p in L.model.parameters()])
    # 2) filter moments
    f = MeanStdFilter(137, "cpu")
    f.observe(full["obs"][rank * 64:(rank + 1) * 64]); f.sync()
    # 3) metrics
    st = EpisodeStats(2, "cpu")
    st.step(torch.tensor([float(rank + 1), 10.0 * (rank + 1)]), torch.tensor([1.0, 1.0]))
    summ = st.summarize()
    # 4) one full update keeps ranks in lock-step
    info = L.update(dict(mine), torch.Generator().manual_seed(5))
    w_after = torch.cat([p.detach().reshape(-1) for p in L.model.parameters()])
    torch.save({"g": g, "mean": f.mean, "std": f.std, "n": f.n, "summ": summ, "w": w_after, "info": info},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True); r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    # ranks agree with each other
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["w"], r1["w"]) and torch.equal(r0["mean"], r1["mean"])
    # and with one process on the concatenated batch
    for k in ("RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    cfg = PPOConfig(num_sgd_iter=1, sgd_minibatch_size=64, lr=1e-3, seed=3)
    L = PPOLearner(cfg, "cpu")
    full = make_batch(128, seed=11)
    loss, _ = L.loss(full)            # mean over 128 == average of the two 64-sample means
    L.opt.zero_grad(); loss.backward()
    g = torch.cat([p.grad.reshape(-1) for p in L.model.parameters()])
    assert torch.allclose(r0["g"], g, atol=2e-6, rtol=1e-4)
    # float32 partial sums of the deviations from each rank's pivot row, merged in float64
    assert torch.allclose(r0["mean"], full["obs"].double().mean(0), rtol=0, atol=5e-7)
    assert torch.allclose(r0["std"], full["obs"].double().std(0), rtol=1e-6, atol=0) and float(r0["n"]) == 128
    s = r0["summ"]
    assert s["episodes_this_iter"] == 4 and s["episode_reward_max"] == 20.0 and s["episode_reward_min"] == 1.0
    assert abs(s["episode_reward_mean"] - (1 + 10 + 2 + 20) / 4) < 1e-9
    assert np.isfinite(r0["info"]["total_loss"]) and r0["info"]["kl"] == r1["info"]["kl"]


def test_learner_improves_surrogate_single_process():
    cfg = PPOConfig(num_sgd_iter=5, sgd_minibatch_size=256, lr=3e-3, seed=1, entropy_coeff_start=0.0)
    L = PPOLearner(cfg, "cpu")
    b = make_batch(1024, seed=2)
    with torch.no_grad():
        mean, log_std, v = L.model(b["obs"])
        b.update(mean=mean, log_std=log_std, logp=gaussian_logp(b["actions"], mean, log_std), values=v)
    before = float(L.loss(dict(b, adv=(b["adv"] - b["adv"].mean()) / b["adv"].std()))[0])
    info = L.update(b, torch.Generator().manual_seed(0))
    after = float(L.loss(dict(b, adv=(b["adv"] - b["adv"].mean()) / b["adv"].std()))[0])
    assert after < before and np.isfinite(info["kl"]) and info["kl"] >= 0


def _trial_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    pdist.init_distributed(backend="gloo")
    seen = []

    def run_one(t):
        # inside a rank's own trial nothing may reach for the process group
        assert not pdist.is_dist() and pdist.world_info() == (0, rank, 1)
        L = PPOLearner(PPOConfig(num_sgd_iter=1, sgd_minibatch_size=32, lr=1e-3, seed=t), "cpu")   # broadcast_module_ must be a no-op
        info = L.update(make_batch(64, seed=t), torch.Generator().manual_seed(t))
        seen.append(t)
        return {"trial_id": f"{t:05d}", "rank": rank, "kl": info["kl"]}

    rows = pdist.run_trials(5, run_one, trial_parallel=True)
    assert pdist.is_dist() and pdist.world_info() == (rank, rank, world)              # back to the two-rank job
    exp = pdist.broadcast_object(f"exp-from-{rank}")
    torch.save({"rows": rows, "seen": seen, "exp": exp}, os.path.join(out_dir, f"t{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_trial_parallel_runs_each_trial_on_one_rank_and_gathers_the_rows(tmp_path):
    """launch.train(trial_parallel=True)'s plumbing (pdist.run_trials / solo) under gloo, world size 2: rank r runs trials r,
    r + 2, ... as a one-rank job (no collective inside a trial), every rank ends up with all rows in trial order, and a trial's
    result equals the same trial run in a single process."""
    port = _free_port()
    mp.spawn(_trial_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "t0.pt", weights_only=False); r1 = torch.load(tmp_path / "t1.pt", weights_only=False)
    assert r0["seen"] == [0, 2, 4] and r1["seen"] == [1, 3]
    assert r0["rows"] == r1["rows"] and [r["trial_id"] for r in r0["rows"]] == [f"{t:05d}" for t in range(5)]
    assert [r["rank"] for r in r0["rows"]] == [0, 1, 0, 1, 0]
    assert r0["exp"] == r1["exp"] == "exp-from-0"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    L = PPOLearner(PPOConfig(num_sgd_iter=1, sgd_minibatch_size=32, lr=1e-3, seed=3), "cpu")
    assert L.update(make_batch(64, seed=3), torch.Generator().manual_seed(3))["kl"] == r0["rows"][3]["kl"]


def _slow_trial_worker(rank, world, port, out_dir):
    import datetime
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    # the job's own collective timeout is SHORT (here 4 s; 10 minutes on RCCL): rank 0 runs one more trial than rank 1 and that
    # trial outlasts it, so rank 1 waits in the final gather for longer than the default group would tolerate
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=4))

    def run_one(t):
        if t == 2:
            time.sleep(9.0)
        return {"trial_id": f"{t:05d}", "rank": pdist.world_info()[0], "real_rank": rank}

    rows = pdist.run_trials(3, run_one, trial_parallel=True)
    torch.save(rows, os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_trial_parallel_gather_outlasts_the_jobs_collective_timeout(tmp_path):
    """with trial_parallel the ranks exchange nothing until the final gather, so an early rank waits there
    for a whole trial — longer than the default group's collective timeout.  The rows travel through a gloo group with a timeout
    of days: one rank running one more (slow) trial than the other still ends with every row on every rank."""
    port = _free_port()
    mp.spawn(_slow_trial_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "s0.pt", weights_only=False); r1 = torch.load(tmp_path / "s1.pt", weights_only=False)
    assert r0 == r1 and [r["trial_id"] for r in r0] == ["00000", "00001", "00002"]
    assert [r["real_rank"] for r in r0] == [0, 1, 0] and all(r["rank"] == 0 for r in r0)
