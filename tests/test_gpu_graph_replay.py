"""The r01 finding "NaNs in unrelated tensors after hipGraph replay", pinned.

What was found (tools/graph_reduce_probe.py, profiles/r02_graph_reduce_probe.json, r02_nan_repro_*.json; the bisection script
tools/graph_nan_repro.py drove the torch learner machinery that r03 removed and lives in git history, commit 12bcd55): a torch
reduction over the middle axis of a large tensor inside the captured sampling loop (MeanStdFilter.observe on the [T-1, N, 137]
rollout buffer) returned wrong sums from the second replay on — only the second reduction of the variant without a temporary,
so the cause is layout-dependent (semaphore / memset-node aliasing in the graph's pool is the suspect) and NOT pinned below
torch.  The BLAS library was not involved: its 1- / 12-wide head GEMMs replay exactly between canaries.  The loop keeps its
reductions outside the captured region, and nothing captured reduces with torch any more."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_formerly_failing_configuration_stays_finite_and_blas_heads_respect_their_bounds():
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    dev = torch.device("cuda", 0)
    # (1) narrow head GEMMs under replay, between canaries allocated in the same capture pool
    for n in (1, 12):
        x = torch.randn(4096, 256, device=dev).bfloat16()
        w = (torch.randn(n, 256, device=dev) * 0.05).bfloat16()
        b = torch.randn(n, device=dev).bfloat16()
        F.linear(x, w, b); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            c1 = torch.full((1024,), 7.0, device=dev)
            out = F.linear(x, w, b)
            c2 = torch.full((1024,), 9.0, device=dev)
        for _ in range(3):
            x.copy_(torch.randn(4096, 256, device=dev).bfloat16())
            g.replay()
            ref = F.linear(x.float(), w.float(), b.float())
            assert float((out.float() - ref).abs().max()) < 0.06
            assert bool((c1 == 7.0).all()) and bool((c2 == 9.0).all())
    # (2) the configuration that produced garbage filter moments at the second replay (16 384 envs, T = 32,
    # 32 768-sample minibatches): the moments the loop accumulates must equal a float64 recomputation from the very
    # buffers it read, at every iteration, and everything stays finite
    env = PioneerVectorEnv(16384, device=dev, seed=0, engine_config=EngineConfig(max_episode_steps=500))
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=32, num_sgd_iter=2, sgd_minibatch_size=32768),
                    use_graph=True)
    T = tr.cfg.rollout_fragment_length
    prev = torch.empty_like(tr.raw_obs)
    for it in range(5):
        prev.copy_(tr.raw_obs)
        n0, mean0 = float(tr.filter.n), tr.filter.mean.clone()
        r = tr.train()
        assert np.isfinite(r["kl"]) and np.isfinite(r["total_loss"])
        x = torch.cat([prev, tr.buf["raw_obs"][:T - 1].reshape(-1, 137)])
        bmean = torch.zeros(137, dtype=torch.float64, device=dev)
        for c in x.split(65536):                                     # float64 in slices: no 0.5 GB temporaries
            bmean += c.double().sum(0)
        bmean /= x.shape[0]
        want = mean0 + (bmean - mean0) * (x.shape[0] / (n0 + x.shape[0]))
        assert float(tr.filter.n) == n0 + x.shape[0]
        assert torch.allclose(tr.filter.mean, want, rtol=0, atol=1e-4), float((tr.filter.mean - want).abs().max())
        assert bool(torch.isfinite(tr.filter.m2).all()) and float(tr.filter.mean.abs().max()) < 1e3
    assert tr._graph is not None and tr.learner.hip        # the sampling loop ran from its graph, the learner on the HIP kernels
    for p in tr.learner.model.parameters():
        assert bool(torch.isfinite(p).all())
    env.close()
