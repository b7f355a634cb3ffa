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
    # the PPO legs: the float32-accurate loop leads and is labelled as the credited one, bf16 is labelled the reduced-precision variant
    keys = list(r)
    assert keys.index("ppo_loop_f32") < keys.index("ppo_loop")
    for key, planes_word in (("ppo_loop_f32", "float32-accurate"), ("ppo_loop", "REDUCED-PRECISION")):
        p = r[key]
        assert "error" not in p and p["losses_finite"] and p["sgd_minibatch_size"] == 8192 and p["grad_allreduce"] is None
        assert planes_word in p["precision"]
        rf = p["roofline"]
        assert rf["us_per_update"] > 0 and rf["bound"] and ("bytes_per_update" not in rf or rf["bytes_per_update"] > 0)   # filled or dropped, never null
    assert r["ppo_loop_f32"]["mlp_dtype"].startswith("f32") and r["ppo_loop"]["mlp_dtype"] == "bf16"
    assert "ppo_loop_65536" not in r               # (--ppo-envs overrides the leg sizes: the anchor is skipped in this small run)
    b = r["build"]
    assert b["binary_is_this_tree"] and b["binary"] == b["tree"] and len(b["env_kernel_sources_sha16"]) == 16


def test_config3_anchor_and_dynamics_lane_roofline_are_in_the_line():
    """`ppo_loop_65536` (config[3] whole on one GPU, both precisions, each with its roofline), and the dynamics leg's
    second roofline figure in fp32 lane-operations."""
    r, _ = _bench(["--steps", "64", "--warmup", "8", "--large-envs", "0", "--fused-leg", "0", "--split-leg", "0", "--ppo-iters", "3",
                   "--ppo-f32", "0", "--ppo-large-minibatch", "0", "--no-cpu-baseline"])
    a = r["ppo_loop_65536"]
    for prec in ("f32", "bf16"):
        p = a[prec]
        assert "error" not in p and p["losses_finite"] and p["total_envs"] == 65536 and p["sgd_minibatch_size"] == 32768 and p["rollout_T"] == 32
        assert p["sgd_updates_per_iter"] == 4 * 64 and p["roofline"]["us_per_update"] > 0 and p["value"] > 1e6
    assert a["f32"]["value"] < a["bf16"]["value"]
    rf = r["dynamics_randomized"]["roofline"]
    if rf["frac"] is None:             # the instruction count is looked up from a committed counter pass: quoted only on the sources it ran on
        assert "stale" in rf["source"] or "profiles/" in rf["source"]
    else:
        lanes = rf["fp32_lanes"]
        assert lanes["unit"].startswith("T fp32 lane-operations") and 0.05 < lanes["frac"] < 1.0 and abs(lanes["peak"] - 78.65) < 0.1
        assert "second wave" in lanes["note"] and "valu_issue_probe" in lanes["note"]


def test_two_ranks_self_launched_on_the_metric_configuration():
    r, res = _bench(["--gpus", "2"] + SMALL, env_extra={"PNR_BENCH_SHARE_GPU": "1"})
    assert "torch.distributed.run" in res.stderr
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert r["config"]["total_envs"] == 65536 and r["config"]["envs_per_gpu"] == 32768
    assert "cpu_baseline" not in r                 # contract: rank 0 at N=1 only
    assert r["weak_scaling"]["envs_per_gpu"] == 65536 and r["weak_scaling"]["total_envs"] == 131072
    assert "error" not in r["ppo_loop_f32"] and r["ppo_loop_f32"]["grad_allreduce"]["world_size"] == 2     # the credited loop runs at N > 1 too
    p = r["ppo_loop"]
    assert "error" not in p and p["losses_finite"]
    assert p["total_envs"] == 2048 and p["envs_per_gpu"] == 1024 and p["sgd_minibatch_size_per_rank"] == 4096
    assert p["grad_allreduce"]["world_size"] == 2 and p["grad_allreduce"]["backend"] == "gloo"
    assert p["hip_graph"]["sampling"] and p["hip_graph"]["learner"].startswith("hip kernels")
