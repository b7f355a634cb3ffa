#!/usr/bin/env python3
"""Freeze the CPU oracle: a small trajectory (inputs + expected outputs) written to oracle_traj_v1.npz.

Provenance: produced by oracle/pnr_oracle.c (ORC_REF precision) itself — NOT by the reference, which
cannot run here (pybullet/gym absent) and holds no vectors.  The fixture guards the checker against
accidental changes; the oracle's correctness rests on tests/test_oracle.py.
    python tests/golden/make_oracle_traj.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import COracle                      # noqa: E402
from oracle.binding import ORC_REF              # noqa: E402


def main():
    n, T = 8, 40
    o = COracle(n, seed=20261004, precision=ORC_REF, auto_reset=True, max_episode_steps=15)
    rng = np.random.RandomState(7)
    obs0 = o.reset()
    acts = (rng.uniform(-1.5, 1.5, (T, n, 6)) * o.a_max).astype(np.float32)
    acts[::4] = (np.sign(acts[::4]) * o.a_max).astype(np.float32)          # saturate on every 4th step
    obs, rew, done, trunc = [], [], [], []
    for t in range(T):
        ob, r, d, tr = o.step(acts[t])
        obs.append(ob); rew.append(r); done.append(d); trunc.append(tr)
    np.savez_compressed(os.path.join(HERE, "oracle_traj_v1.npz"), seed=20261004, max_episode_steps=15,
                        actions=acts, obs0=obs0, obs=np.array(obs), reward=np.array(rew),
                        done=np.array(done), truncated=np.array(trunc), state_words=o.state_words())
    print("wrote oracle_traj_v1.npz", np.array(obs).shape)


if __name__ == "__main__":
    main()
