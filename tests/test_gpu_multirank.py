"""N>1 path on the GPU box: two ranks (sharing the one GPU, gloo instead of RCCL) run the real
sharded PPO loop — env shards by global env id, flat gradient all-reduce, obs-filter merge."""
import json
import os
import socket
import subprocess
import sys

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
ck = torch.load(os.path.join(out, "PPO_Pioneer-v1_00000", "checkpoint_final.pt"), weights_only=False)
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
