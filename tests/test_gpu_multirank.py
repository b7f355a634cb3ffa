"""N>1 path on the GPU box: two ranks (sharing the one GPU, gloo instead of RCCL) run the real
sharded PPO loop — env shards by global env id, flat gradient all-reduce, obs-filter merge."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, torch
sys.path.insert(0, os.environ["PNR_ROOT"])
from pioneer_amd import dist as pdist
from pioneer_amd.launch import train
from pioneer_amd.ppo import PPOConfig
out = os.environ["PNR_OUT"]
df = train(results_dir=out, checkpoint_freq=0, num_samples=1, num_workers=1, monitor=False,
           training_iterations=3, envs_per_worker=256,
           ppo_config=PPOConfig(rollout_fragment_length=8, num_sgd_iter=2, sgd_minibatch_size=1024, seed=5))
rank = int(os.environ["RANK"])
row = df.iloc[0].to_dict() if hasattr(df, "iloc") else df[0]
ck = torch.load(os.path.join(out, "PPO_Pioneer-v1_00000", "checkpoint_final.pt"), weights_only=True)
w = torch.cat([v.reshape(-1).double().cpu() for v in ck["model"].values()])
json.dump({"rank": rank, "timesteps_total": int(row["timesteps_total"]), "episodes_total": int(row["episodes_total"]),
           "kl": float(row["kl"]), "wsum": float(w.sum()), "wabs": float(w.abs().sum())},
          open(os.path.join(out, f"rank{rank}.json"), "w"))
'''


def test_two_ranks_train_in_lock_step(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, PNR_ROOT=ROOT, PNR_OUT=str(tmp_path), PNR_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    r0 = json.load(open(tmp_path / "rank0.json")); r1 = json.load(open(tmp_path / "rank1.json"))
    # world = 2 ranks x 256 envs x 8 steps x 3 iterations
    assert r0["timesteps_total"] == r1["timesteps_total"] == 2 * 256 * 8 * 3
    assert r0["kl"] == r1["kl"] and r0["episodes_total"] == r1["episodes_total"]   # all-reduced metrics
    assert r0["wsum"] == r1["wsum"] and r0["wabs"] == r1["wabs"]                   # one checkpoint, ranks in lock-step


TRIAL_WORKER = r'''
import json, os, sys, torch
sys.path.insert(0, os.environ["PNR_ROOT"])
from pioneer_amd.launch import train
from pioneer_amd.ppo import PPOConfig
out = os.environ["PNR_OUT"]
df = train(results_dir=out, checkpoint_freq=0, num_samples=3, num_workers=1, monitor=False, training_iterations=2, envs_per_worker=256,
           ppo_config=PPOConfig(rollout_fragment_length=8, num_sgd_iter=2, sgd_minibatch_size=1024, seed=5), trial_parallel=True)
rows = df.to_dict("records") if hasattr(df, "to_dict") else df
json.dump([{k: r[k] for k in ("trial_id", "experiment_id", "timesteps_total", "kl", "episodes_total")} for r in rows],
          open(os.path.join(out, f"trials_rank{os.environ['RANK']}.json"), "w"))
'''


def test_trial_parallel_training_on_two_ranks(tmp_path):
    """The reference's own scaling axis (Tune's num_samples trials, pioneer_knm_train.py:43-44): with trial_parallel each rank
    trains its own trials as a one-rank job — 256 envs each, not 512 shared — and every rank gets all rows; a trial's result
    equals (to the rounding of the CPU-side weight initialisation) the same trial trained by a single process."""
    script = tmp_path / "trials.py"
    script.write_text(TRIAL_WORKER)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, PNR_ROOT=ROOT, PNR_OUT=str(tmp_path), PNR_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    r0 = json.load(open(tmp_path / "trials_rank0.json")); r1 = json.load(open(tmp_path / "trials_rank1.json"))
    assert r0 == r1 and [r["trial_id"] for r in r0] == ["00000", "00001", "00002"]
    assert len({r["experiment_id"] for r in r0}) == 1
    assert all(r["timesteps_total"] == 2 * 8 * 256 for r in r0)       # one rank's envs per trial: nothing was sharded
    for t in ("00000", "00001", "00002"):
        assert (tmp_path / f"PPO_Pioneer-v1_{t}" / "checkpoint_final.pt").exists()
    # the same trial 1 in this process (one rank): identical result row
    from pioneer_amd.launch import train
    from pioneer_amd.ppo import PPOConfig
    solo = train(results_dir=str(tmp_path / "solo"), checkpoint_freq=0, num_samples=2, num_workers=1, monitor=False, training_iterations=2,
                 envs_per_worker=256, ppo_config=PPOConfig(rollout_fragment_length=8, num_sgd_iter=2, sgd_minibatch_size=1024, seed=5))
    row = (solo.to_dict("records") if hasattr(solo, "to_dict") else solo)[1]
    # (to rounding: the nets' orthogonal initialisation is a CPU QR factorisation whose last bits follow the BLAS thread count,
    # which differs between this process and the torchrun children)
    assert abs(row["kl"] - r0[1]["kl"]) <= 2e-2 * abs(row["kl"]) and row["episodes_total"] == r0[1]["episodes_total"]


LEARNER_WORKER = r'''
import json, os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["PNR_ROOT"])
from pioneer_amd.ppo import PPOConfig, PPOLearner, gaussian_logp
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
B = 4096
g = torch.Generator(device=dev).manual_seed(100 + rank)           # every rank its own share of the batch
R = lambda *s: torch.randn(*s, generator=g, device=dev)
act, mean, log_std = R(B, 6), 0.1 * R(B, 6), 0.1 * R(B, 6)
batch = {"obs": R(B, 137), "actions": act, "mean": mean, "log_std": log_std, "logp": gaussian_logp(act, mean, log_std) + 0.2 * R(B),
         "values": R(B), "adv": R(B), "vtarg": R(B)}
prec = os.environ.get("PNR_TEST_PRECISION", "bf16")
cfg = PPOConfig(num_sgd_iter=3, sgd_minibatch_size=B // 2, lr=1e-3, seed=11, hip_kernels=True if prec == "bf16" else prec)
out = {}
for chains in (True, False):
    L = PPOLearner(cfg, dev)
    L.net_chains = chains          # True (default): the two nets as two SGD chains on two streams, each with its own all-reduce
    w0 = torch.cat([p.detach().reshape(-1).double().cpu() for p in L.model.parameters()])
    infos = [L.update(dict(batch)) for _ in range(3)]                # 18 updates, each with its gradient all-reduce(s)
    torch.cuda.synchronize()
    w = torch.cat([p.detach().reshape(-1).double().cpu() for p in L.model.parameters()])
    m, v, step = L.hip_mlp(1).adam_state()
    out[str(chains)] = {"hip": bool(L.hip), "flat_bucket": L._flat_grad is not None, "streams": L._net_streams is not None,
                        "move": float((w - w0).abs().max()), "w": w, "m": m.double().cpu(), "v": v.double().cpu(), "step": float(step),
                        "kl": [i["kl"] for i in infos], "total_loss": [i["total_loss"] for i in infos]}
a, b = out["True"], out["False"]
same = bool(torch.equal(a["w"], b["w"]) and torch.equal(a["m"], b["m"]) and torch.equal(a["v"], b["v"]))
json.dump({"rank": rank, "hip": a["hip"] and b["hip"], "flat_bucket": a["flat_bucket"], "chains_ran": a["streams"] and not b["streams"],
           "chains_equal_one_bucket": same, "loss_gap": max(abs(x - y) for x, y in zip(a["total_loss"] + a["kl"], b["total_loss"] + b["kl"])),
           "move": a["move"], "wsum": float(a["w"].sum()), "wabs": float(a["w"].abs().sum()), "msum": float(a["m"].sum()),
           "vsum": float(a["v"].sum()), "step": a["step"], "kl": a["kl"], "total_loss": a["total_loss"]},
          open(os.path.join(os.environ["PNR_OUT"], f"learner{rank}.json"), "w"))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("precision", ["bf16", "f32"])
def test_hip_learner_keeps_two_ranks_in_lock_step(tmp_path, precision):
    """(bf16 operands, and the float32-accurate split-operand form: PPOConfig(hip_kernels="f32").)  Several ranks on the kernels: every update's gradient is all-reduced before Adam, so master weights, Adam's moments and
    the reported (rank-averaged) losses are identical on both ranks although each rank trains on its own share of the batch —
    and the two-chain form (each net on its own stream with its own all-reduce, what a multi-GPU run uses) leaves exactly the
    weights and moments of the serial one-bucket form."""
    script = tmp_path / "learner.py"
    script.write_text(LEARNER_WORKER)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, PNR_ROOT=ROOT, PNR_OUT=str(tmp_path), OMP_NUM_THREADS="2", PNR_TEST_PRECISION=precision)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    r0 = json.load(open(tmp_path / "learner0.json")); r1 = json.load(open(tmp_path / "learner1.json"))
    for r in (r0, r1):
        assert r["hip"] and r["flat_bucket"] and r["step"] == 18.0, r
        assert r["chains_ran"] and r["chains_equal_one_bucket"], r    # bit for bit: same sums per element, same Adam arithmetic
        assert r["loss_gap"] < 1e-5                                   # the means differ in the order of three additions only
        assert r["move"] > 1e-3                                    # the weights did move
        assert all(np.isfinite(x) for x in r["kl"] + r["total_loss"])
    for k in ("wsum", "wabs", "msum", "vsum", "kl", "total_loss"):
        assert r0[k] == r1[k], (k, r0[k], r1[k])                   # ranks in lock-step, bit for bit


RCCL_WORKER = r'''
import json, os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["PNR_ROOT"])
os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)     # "nccl" is RCCL on ROCm
from pioneer_amd import dist as pdist
pdist.force_collectives(True)       # take the multi-rank code paths with the one rank this box has
from pioneer_amd import PioneerVectorEnv, EngineConfig
from pioneer_amd.ppo import PPOConfig, PPOTrainer

def run(chains, precision):
    env = PioneerVectorEnv(2048, device=dev, seed=3, engine_config=EngineConfig(max_episode_steps=40))
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=16, num_sgd_iter=3, sgd_minibatch_size=8192, lr=3e-4, seed=3, hip_kernels=precision), use_graph=True)
    tr.learner.net_chains = chains
    rows = [tr.train() for _ in range(5)]
    torch.cuda.synchronize()
    mlp = tr.learner.hip_mlp(1)
    out = {"sampling_graph": tr._graph is not None, "hip": bool(tr.learner.hip), "flat_bucket": tr.learner._flat_grad is not None,
           "chains_ran": tr.learner._net_streams is not None,
           "kl": [r["kl"] for r in rows], "total_loss": [r["total_loss"] for r in rows],
           "timesteps_total": rows[-1]["timesteps_total"], "episodes_total": rows[-1]["episodes_total"],
           "weights": [p.detach().clone() for p in mlp.params], "adam": [t.clone() for t in mlp.adam_state()]}
    env.close()
    return out

res = {"backend": dist.get_backend()}
for precision in ("f32", True):
    a, b = run(True, precision), run(False, precision)       # two SGD chains on two streams, each with its own RCCL all-reduce | one bucket
    key = "f32" if precision == "f32" else "bf16"
    res[key] = {k: a[k] for k in ("sampling_graph", "hip", "flat_bucket", "chains_ran", "kl", "total_loss", "timesteps_total", "episodes_total")}
    res[key]["one_bucket_chains_ran"] = b["chains_ran"]
    res[key]["weights_equal"] = all(torch.equal(x, y) for x, y in zip(a["weights"], b["weights"]))
    res[key]["adam_equal"] = all(torch.equal(x, y) for x, y in zip(a["adam"][:2], b["adam"][:2]))
    res[key]["loss_gap"] = max(abs(x - y) for x, y in zip(a["total_loss"], b["total_loss"]))
json.dump(res, open(os.path.join(os.environ["PNR_OUT"], "rccl.json"), "w"))
dist.destroy_process_group()
'''


def test_graph_captured_loop_with_an_rccl_process_group(tmp_path):
    """The N>1 code paths (sampling hipGraph, the HIP learner's flat gradient bucket all-reduced between the gradient kernels and
    pnr_mlp_adam, filter/metric all-reduces) against a real RCCL process group — one rank, all this box has (pdist.force_collectives) —
    so that capture next to RCCL's watchdog thread and eager collectives are exercised.  VERDICT r04 #7a: the DEFAULT learner form
    at N > 1, the two nets as two SGD chains on two streams, each with its own all-reduce, meets RCCL's asynchronous stream
    semantics here (gloo's collectives are host-synchronous): its weights and Adam moments must equal the one-bucket form's bit for
    bit, at both precisions."""
    script = tmp_path / "rccl.py"
    script.write_text(RCCL_WORKER)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, PNR_ROOT=ROOT, PNR_OUT=str(tmp_path), MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    doc = json.load(open(tmp_path / "rccl.json"))
    assert doc["backend"] == "nccl"
    for key in ("f32", "bf16"):
        r = doc[key]
        assert r["sampling_graph"] and r["hip"] and r["flat_bucket"], r      # HIP learner: reduce -> RCCL all-reduce -> pnr_mlp_adam
        assert r["chains_ran"] and not r["one_bucket_chains_ran"], r
        assert r["weights_equal"] and r["adam_equal"], r                     # two chains == one bucket, bit for bit, under RCCL
        assert r["loss_gap"] < 1e-4                                         # (the reported means add three terms in another order)
        assert r["timesteps_total"] == 5 * 16 * 2048 and r["episodes_total"] > 0
        assert all(np.isfinite(x) for x in r["kl"] + r["total_loss"]), r


CONFIG3_WORKER = r'''
import json, os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["PNR_ROOT"])
from pioneer_amd import PioneerVectorEnv, EngineConfig
from pioneer_amd import dist as pdist
from pioneer_amd.ppo import PPOConfig, PPOTrainer
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)                       # every rank on the box's one GPU; gloo, because RCCL refuses two ranks per device
dev = torch.device("cuda", 0)
if world > 1:
    dist.init_process_group("gloo")
total, gmbs, iters = int(os.environ["PNR_TOTAL_ENVS"]), int(os.environ["PNR_GLOBAL_MBS"]), int(os.environ.get("PNR_ITERS", "1"))
start, cnt = pdist.shard_range(total, world, rank)
env = PioneerVectorEnv(cnt, device=dev, seed=0, env_id_offset=start, engine_config=EngineConfig(max_episode_steps=500, auto_reset=True))
mbs = gmbs // world
tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=mbs, seed=2), use_graph=False)
assert tr.learner.hip
rows = [tr.train() for _ in range(iters)]
torch.cuda.synchronize()
w = torch.cat([p.detach().reshape(-1).double().cpu() for p in tr.learner.model.parameters()])
m, v, step = tr.learner.hip_mlp(1).adam_state()
r = rows[-1]
json.dump({"rank": rank, "world": world, "envs": cnt, "mbs": mbs, "chains": bool(world > 1 and tr.learner.net_chains and tr.learner._net_streams is not None),
           "resident_rollout": bool(tr.resident_rollout), "updates": float(step), "timesteps_total": int(r["timesteps_total"]),
           "wsum": float(w.sum()), "wabs": float(w.abs().sum()), "msum": float(m.double().sum()), "vsum": float(v.double().sum()),
           "metrics": {k: float(r[k]) for k in ("kl", "total_loss", "vf_loss", "policy_loss", "entropy")},
           "episodes_total": int(r["episodes_total"])},
          open(os.path.join(os.environ["PNR_OUT"], f"c3_{world}_{rank}.json"), "w"))
env.close()
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def _run_config3(tmp_path, world, total, gmbs):
    script = tmp_path / "config3.py"
    script.write_text(CONFIG3_WORKER)
    env = dict(os.environ, PNR_ROOT=ROOT, PNR_OUT=str(tmp_path), OMP_NUM_THREADS="2", PNR_TOTAL_ENVS=str(total), PNR_GLOBAL_MBS=str(gmbs))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if world == 1:
        res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    else:
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                              "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                             env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    return [json.load(open(tmp_path / f"c3_{world}_{r}.json")) for r in range(world)]


def test_config3_whole_65536_envs_on_two_ranks(tmp_path):
    """BASELINE configs[3] as a whole, in the only form one GPU allows: 65 536 envs in total on two ranks (32 768 each, gloo, both
    on this GPU), T = 32, four epochs of 32 768-sample GLOBAL minibatches (16 384 per rank), the two-chain learner with one
    gradient all-reduce per net and update: 256 updates.  Ranks end bit-identical; against the one-rank run of the same 65 536 envs
    the update count is the same and the loss means agree statistically — NOT bit for bit: a rank shuffles its own shard and draws
    its own action noise, so the minibatches hold different samples (the per-update identity of the averaged gradient with one
    process on the concatenated batch is tests/test_ppo_dist_cpu.py's and test_hip_learner_keeps_two_ranks_in_lock_step's)."""
    two = _run_config3(tmp_path, 2, 65536, 32768)
    r0, r1 = two
    for r in two:
        assert r["envs"] == 32768 and r["mbs"] == 16384 and r["chains"] and r["resident_rollout"], r
        assert r["updates"] == 4 * (32 * 32768) // 16384 == 256 and r["timesteps_total"] == 65536 * 32
        assert all(np.isfinite(x) for x in r["metrics"].values()), r
    for k in ("wsum", "wabs", "msum", "vsum", "metrics", "episodes_total"):
        assert r0[k] == r1[k], (k, r0[k], r1[k])                   # lock-step, bit for bit
    (one,) = _run_config3(tmp_path, 1, 65536, 32768)
    assert one["updates"] == 256 and one["timesteps_total"] == 65536 * 32 and not one["chains"]
    for k, tol in (("vf_loss", 0.05), ("total_loss", 0.05), ("entropy", 0.01)):
        a, b = one["metrics"][k], r0["metrics"][k]
        assert abs(a - b) <= tol * max(abs(a), abs(b)), (k, a, b)
    assert abs(one["wabs"] - r0["wabs"]) <= 1e-3 * one["wabs"]       # 256 updates at lr 2e-5 from the same initial weights


def test_config3_per_rank_shape_of_eight_gpus_on_two_ranks(tmp_path):
    """The per-rank shape of configs[3] at N = 8 (8 192 envs and 4 096-sample minibatches per rank, 256 updates per iteration,
    each with its two 0.43 MB all-reduces), run on the two ranks this box can hold."""
    two = _run_config3(tmp_path, 2, 16384, 8192)
    for r in two:
        assert r["envs"] == 8192 and r["mbs"] == 4096 and r["chains"], r
        assert r["updates"] == 4 * (32 * 8192) // 4096 == 256 and r["timesteps_total"] == 16384 * 32
        assert all(np.isfinite(x) for x in r["metrics"].values()), r
    for k in ("wsum", "wabs", "msum", "vsum", "metrics"):
        assert two[0][k] == two[1][k], k
