"""bench.py's rank plumbing on CPU: `python bench.py --gpus 2` with no torchrun environment must start its own two
ranks (torch.distributed.run child), rendezvous on 127.0.0.1, take the max over ranks and relay ONE JSON line from
rank 0.  PNR_BENCH_DRYRUN=1 replaces the kernels by a sleep (gloo, no GPU) and labels the line as a dry run; the real
two-rank run on the GPU box is tests/test_gpu_bench.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PNR_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_self_launch_two_ranks_prints_one_line():
    res = _run(["--gpus", "2", "--steps", "40", "--warmup", "4"])
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 40 and r["warmup"] == 4
    assert r["scaling"] == "strong" and r["data"] == "dryrun" and r["value"] is None
    assert r["config"]["total_envs"] == 65536 and r["config"]["envs_per_gpu"] == [32768, 32768]
    assert "torch.distributed.run" in res.stderr          # the self-launch path really ran


def test_uneven_shards_cover_the_env_axis():
    res = _run(["--gpus", "3", "--steps", "10", "--warmup", "1", "--envs", "1000"])
    assert res.returncode == 0, res.stderr[-3000:]
    r = json.loads(res.stdout.strip().splitlines()[-1])
    assert r["config"]["envs_per_gpu"] == [334, 333, 333] and r["config"]["total_envs"] == 1000


def test_single_rank_needs_no_launcher():
    res = _run(["--steps", "10", "--warmup", "1"])
    assert res.returncode == 0, res.stderr[-3000:]
    r = json.loads(res.stdout.strip().splitlines()[-1])
    assert r["n_gpus"] == 1 and "torch.distributed.run" not in res.stderr


def test_world_size_mismatch_is_an_error():
    res = _run(["--gpus", "2", "--steps", "10"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert res.returncode == 2


def test_a_failing_rank_fails_the_launcher():
    # rank 1 dies before the first barrier: the child's non-zero exit code must come back, with no result line
    res = _run(["--gpus", "2", "--steps", "10"], env_extra={"PNR_BENCH_DRYRUN_FAIL_RANK": "1"})
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if '"metric"' in l]


def _run_real(extra, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "PNR_BENCH_DRYRUN",
                                                              "PNR_BENCH_SHARE_GPU")}
    env.update(OMP_NUM_THREADS="1")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout)


def test_more_ranks_than_devices_is_one_clear_line_and_exit_2():
    """`bench.py --gpus N` on a node with fewer than N devices (the failure of gpurun_out/r03/bench_gpus2.err: rank 1 raised
    `HIP error: invalid device ordinal` from torch.cuda.set_device): the launcher says what is wrong and exits 2 before any rank
    starts; a rank started by an outside launcher says the same before its first GPU call."""
    import torch
    ndev = torch.cuda.device_count()
    want = ndev + 2
    res = _run_real(["--gpus", str(want), "--steps", "5", "--warmup", "1"])
    assert res.returncode == 2, (res.returncode, res.stderr[-2000:])
    assert f"--gpus {want} needs {want} visible devices (found {ndev})" in res.stderr
    assert "torch.distributed.run" not in res.stderr and not res.stdout.strip()
    # under an existing launcher environment: the rank itself reports it
    res = _run_real(["--gpus", str(want), "--steps", "5", "--warmup", "1"],
                    env_extra={"WORLD_SIZE": str(want), "RANK": str(want - 1), "LOCAL_RANK": str(want - 1), "MASTER_ADDR": "127.0.0.1",
                               "MASTER_PORT": "29999"})
    assert res.returncode == 2, (res.returncode, res.stderr[-2000:])
    assert f"needs {want} visible devices (found {ndev})" in res.stderr and "Traceback" not in res.stderr
