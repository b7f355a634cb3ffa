"""bench.py end to end on the GPU box: the N=1 line carries the contract's fields, and `--gpus 2` (two ranks sharing the
one GPU over gloo, PNR_BENCH_SHARE_GPU=1) goes through the self-launch path and reports the metric's configuration:
65 536 envs IN TOTAL, strong scaling, the PPO leg with its gradient all-reduce."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "64", "--warmup", "8", "--large-envs", "0", "--dynamic-leg", "0", "--fused-leg", "0",
         "--ppo-iters", "1", "--ppo-envs", "2048", "--ppo-minibatch", "8192", "--ppo-large-minibatch", "0"]


def _bench(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0]), res


def test_single_gpu_line_has_the_contract_fields():
    r, _ = _bench(SMALL + ["--cpu-seconds", "1"])
    assert r["metric"] == "env-steps/sec" and r["n_gpus"] == 1 and r["steps"] == 64 and r["warmup"] == 8
    assert r["scaling"] == "strong" and r["config"]["total_envs"] == 65536 and r["config"]["envs_per_gpu"] == 65536
    assert r["repeats"] >= 5                       # a 64-step block is far below 60 ms: many back-to-back blocks
    assert abs(r["value"] - 65536 * 64 / (r["ms_per_step"] * 64e-3)) < 1e-6 * r["value"]
    assert r["config"]["handles_per_gpu"] == 1 and not r["config"]["hip_graph"]
    assert r["split_streams"]["handles_per_gpu"] == 2 and 0.05 < r["split_streams"]["frac"] < 1.0
    assert r["config"]["block_event_spread_p10_p90"] < 0.2
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and 0.05 < rf["frac"] < 1.0 and rf["peak"] == 8000.0
    assert ("NOT measured in this run" in rf["traffic_source"]) if rf["traffic"] is not None else ("stale" in rf["traffic_source"] or "profiles/" in rf["traffic_source"])
    assert rf["launches_per_step"] == 1 and "single_launch" not in rf
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["value"] > 0
    p = r["ppo_loop"]
    assert "error" not in p and p["losses_finite"] and p["sgd_minibatch_size"] == 8192 and p["grad_allreduce"] is None


def test_two_ranks_self_launched_on_the_metric_configuration():
    r, res = _bench(["--gpus", "2"] + SMALL, env_extra={"PNR_BENCH_SHARE_GPU": "1"})
    assert "torch.distributed.run" in res.stderr
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert r["config"]["total_envs"] == 65536 and r["config"]["envs_per_gpu"] == 32768
    assert "cpu_baseline" not in r                 # contract: rank 0 at N=1 only
    assert r["weak_scaling"]["envs_per_gpu"] == 65536 and r["weak_scaling"]["total_envs"] == 131072
    p = r["ppo_loop"]
    assert "error" not in p and p["losses_finite"]
    assert p["total_envs"] == 2048 and p["envs_per_gpu"] == 1024 and p["sgd_minibatch_size_per_rank"] == 4096
    assert p["grad_allreduce"]["world_size"] == 2 and p["grad_allreduce"]["backend"] == "gloo"
    assert p["hip_graph"]["sampling"] and p["hip_graph"]["learner"].startswith("hip kernels")
