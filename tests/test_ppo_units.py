"""CPU unit tests of the PPO driver's host-side pieces (no GPU, no env): the prepared observation filter against its
call form, the shifted-sum moments, the config's checkpoint compatibility."""
import numpy as np
import pytest
import torch

from pioneer_amd.ppo import ActorCritic, MeanStdFilter, PPOConfig


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


def test_config_from_dict_reads_old_checkpoints():
    """Checkpoints of r01 / r02 carry `amp_bf16` in their cfg dict: read as `hip_kernels`; unknown keys are dropped."""
    c = PPOConfig.from_dict({"fcnet_hiddens": [256, 256], "lr": 1e-4, "amp_bf16": False, "some_removed_option": 3})
    assert c.fcnet_hiddens == (256, 256) and c.lr == 1e-4 and c.hip_kernels is False
    assert PPOConfig.from_dict(PPOConfig().__dict__) == PPOConfig()
    # r04 checkpoints: "bf16x2" (two bf16 planes) no longer exists and reads as its successor, the two-fp16-plane "f32"; the names map to planes
    assert PPOConfig.from_dict({"hip_kernels": "bf16x2"}).hip_kernels == "f32"
    assert [PPOConfig(hip_kernels=k).mlp_planes() for k in (True, "bf16", "f32", "bf16x3")] == [1, 1, 2, 3]
    with pytest.raises(AssertionError):
        PPOConfig(hip_kernels="fp8").mlp_planes()
    assert not PPOConfig().wants_hip("cpu") and PPOConfig().wants_hip("cuda:0") and not PPOConfig(fcnet_hiddens=(64, 64)).wants_hip("cuda:0")
    m = ActorCritic(PPOConfig())
    mean, log_std, v = m(torch.randn(33, 137))
    assert mean.shape == (33, 6) and log_std.shape == (33, 6) and v.shape == (33,) and float(log_std.max()) <= 2.0


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
