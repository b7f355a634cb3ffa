"""CPU unit tests of the PPO driver's rollout bookkeeping (no GPU, no env): the vectorised
episode statistics, the prepared observation filter and the cached-weight inference path must
reproduce their step-by-step / autograd counterparts."""
import numpy as np
import pytest
import torch

from pioneer_amd.ppo import ActorCritic, EpisodeStats, MeanStdFilter, PPOConfig


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_rollout_statistics_equal_step_by_step(seed):
    g = torch.Generator().manual_seed(seed)
    T, N = 37, 53
    a, b = EpisodeStats(N, "cpu"), EpisodeStats(N, "cpu")
    for it in range(4):                                   # carry-over across rollouts included
        rew = torch.randn(T, N, generator=g)
        p = 0.0 if it == 2 else 0.08                      # one rollout without any terminal
        term = (torch.rand(T, N, generator=g) < p).float()
        if it == 3:
            term[0] = 1.0; term[-1] = 1.0                 # terminals on the first and the last step
        for t in range(T):
            a.step(rew[t], term[t])
        b.rollout(rew, term)
        assert torch.allclose(a.ret, b.ret, atol=1e-5) and torch.equal(a.len, b.len)
        assert float(a.w_cnt) == float(b.w_cnt) and float(a.w_len) == float(b.w_len)
        assert abs(float(a.w_sum) - float(b.w_sum)) < 1e-3 * max(1.0, abs(float(a.w_sum)))
        if float(a.w_cnt) > 0:
            assert abs(float(a.w_max) - float(b.w_max)) < 1e-5 and abs(float(a.w_min) - float(b.w_min)) < 1e-5
        ra, rb = a.summarize(), b.summarize()
        assert ra["episodes_this_iter"] == rb["episodes_this_iter"] and ra["episodes_total"] == rb["episodes_total"]


def test_prepared_filter_equals_call():
    f = MeanStdFilter(7, "cpu", clip=3.0)
    x = torch.randn(5, 7) * 4 + 1
    f.prepare()
    assert torch.equal(f.apply_(x, out=torch.empty_like(x)), f(x))          # identity before any statistics
    f.observe(torch.randn(200, 7) * 2 + 0.5); f.sync()
    f.prepare()
    y = f.apply_(x, out=torch.empty_like(x))
    assert torch.allclose(y, f(x), atol=2e-6) and float(y.abs().max()) <= 3.0
    f2 = MeanStdFilter(7, "cpu", clip=0.0)                                   # no clipping
    f2.observe(torch.randn(50, 7)); f2.sync(); f2.prepare()
    big = torch.full((1, 7), 1e4)
    assert torch.allclose(f2.apply_(big, out=torch.empty_like(big)), f2(big), rtol=1e-6)


def test_cached_inference_equals_forward():
    torch.manual_seed(0)
    m = ActorCritic(PPOConfig())
    obs = torch.randn(33, 137)
    mean, log_std, v = m(obs)
    m.refresh_inference_cache(False)
    xin = torch.zeros(33, 144); xin[:, :137] = obs
    head, vc = m.forward_cached(xin)
    assert torch.allclose(head[:, :6], mean, atol=1e-6) and torch.allclose(head[:, 6:].clamp(-20, 2), log_std, atol=1e-6)
    assert torch.allclose(vc.squeeze(-1), v, atol=1e-6)
    with torch.no_grad():                                                    # the cache follows the parameters
        for p in m.parameters():
            p.add_(0.01)
    m.refresh_inference_cache(False)
    mean2 = m(obs)[0]
    assert torch.allclose(m.forward_cached(xin)[0][:, :6], mean2, atol=1e-6) and not torch.allclose(mean2, mean)


def test_constant_obs_columns_filter_to_exactly_zero():
    """36 of the 137 obs entries are per-env constants (joint limits and their cos / sin, obs[18:54]).  RLlib's float64
    RunningStat gives them std 0 and a filtered value of exactly 0; float32 column sums gave +-clip."""
    g = torch.Generator().manual_seed(3)
    f = MeanStdFilter(5, "cpu", clip=10.0)
    consts = torch.tensor([-3.1415927, 0.2588190, 1.309, 5.0e-8, -1.0])
    for it in range(3):
        x = torch.randn(3, 16384, 5, generator=g) * 3 + 7
        x[..., 1:] = consts[1:]
        x[..., 0] = consts[0] if it < 2 else x[..., 0]            # column 0 stops being constant in the third batch
        f.observe(x[0]); f.observe(x[1:]); f.sync()
        f.prepare()
        row = x[0, :4]
        y = f.apply_(row, out=torch.empty_like(row))
        assert torch.equal(y[:, 1:], torch.zeros(4, 4)), y
        assert torch.equal(f(row)[:, 1:], torch.zeros(4, 4))
        assert torch.equal(f.mean[1:].float(), consts[1:]) and torch.equal(f.m2[1:], torch.zeros(4, dtype=torch.float64))
        if it < 2:
            assert torch.equal(y[:, 0], torch.zeros(4))
        else:
            assert float(f.std[0]) > 0.5 and float(y[:, 0].abs().max()) < 10.0


def test_filter_moments_match_float64_reference():
    g = torch.Generator().manual_seed(4)
    f = MeanStdFilter(9, "cpu")
    xs = []
    for _ in range(4):
        x = torch.randn(8192, 9, generator=g) * torch.arange(1, 10) + 100.0      # large mean, small spread
        f.observe(x); f.sync(); xs.append(x)
    allx = torch.cat(xs).double()
    assert float(f.n) == allx.shape[0]
    assert torch.allclose(f.mean, allx.mean(0), rtol=0, atol=1e-6)
    assert torch.allclose(f.std, allx.std(0), rtol=1e-6)
