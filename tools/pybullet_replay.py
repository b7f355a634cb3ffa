#!/usr/bin/env python3
"""Optional PyBullet replay of the reference's per-step call sequence (BASELINE.md plan B1).

Runs ONLY if `pybullet` is importable on this host; otherwise prints
"pybullet: unavailable on this host" and exits 0 — nothing is estimated or fabricated.
The URDF comes from the engine's own model table (pioneer_amd/model.py); per step the script
issues what pioneer_knm_env.py:111-211 issues: 6 x resetJointState, getLinkState(FK) + base pose
in act(), 10 x stepSimulation, the two pose queries again in observe(), and the NumPy obs pack.
With --check it also compares pointer positions with the CPU oracle's FK.
"""
import argparse
import importlib.util
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    if importlib.util.find_spec("pybullet") is None:
        print(json.dumps({"pybullet": "unavailable on this host"}))
        return 0
    import numpy as np
    import pybullet as pb
    from pioneer_amd.model import to_urdf

    cid = pb.connect(pb.DIRECT)
    pb.setGravity(0, 0, 0, physicsClientId=cid)            # bullet_env.py:41, bullet_scene.py:270
    pb.setTimeStep(1 / 240, physicsClientId=cid)
    with tempfile.NamedTemporaryFile("w", suffix=".urdf", delete=False) as f:
        f.write(to_urdf())
        path = f.name
    body = pb.loadURDF(path, flags=0, physicsClientId=cid)
    os.unlink(path)
    joints, pointer = [], None
    for i in range(pb.getNumJoints(body, physicsClientId=cid)):
        info = pb.getJointInfo(body, i, physicsClientId=cid)
        if info[2] == pb.JOINT_REVOLUTE:
            joints.append(i)
        if info[12].decode() == "robot:pointer":
            pointer = i
    lo = np.array([pb.getJointInfo(body, j, physicsClientId=cid)[8] for j in joints], dtype=np.float32)
    hi = np.array([pb.getJointInfo(body, j, physicsClientId=cid)[9] for j in joints], dtype=np.float32)
    vis = pb.createVisualShape(pb.GEOM_SPHERE, radius=0.2, physicsClientId=cid)
    target = pb.createMultiBody(baseMass=0.0, basePosition=(20, 0, 4), baseVisualShapeIndex=vis, physicsClientId=cid)
    rng = np.random.RandomState(0)
    r = rng.uniform(lo, hi).astype(np.float32)
    max_err = 0.0

    def step():
        nonlocal r, max_err
        r = np.clip(r + rng.uniform(-0.05, 0.05, 6).astype(np.float32), lo, hi)
        for q, j in zip(r, joints):
            pb.resetJointState(body, j, float(q), physicsClientId=cid)
        p = pb.getLinkState(body, pointer, computeLinkVelocity=1, computeForwardKinematics=1, physicsClientId=cid)[0]
        t = pb.getBasePositionAndOrientation(target, physicsClientId=cid)[0]
        np.linalg.norm(np.array(t) - np.array(p))
        for _ in range(10):
            pb.stepSimulation(physicsClientId=cid)
        p = pb.getLinkState(body, pointer, computeLinkVelocity=1, computeForwardKinematics=1, physicsClientId=cid)[0]
        t = pb.getBasePositionAndOrientation(target, physicsClientId=cid)[0]
        np.concatenate([r, np.cos(r), np.sin(r), lo, np.cos(lo), np.sin(lo), hi, np.cos(hi), np.sin(hi),
                        r - lo, np.cos(r - lo), np.sin(r - lo), hi - r, np.cos(hi - r), np.sin(hi - r),
                        np.zeros(18), np.zeros(18), np.array(p), np.array(t), np.array(t) - np.array(p), [0.0], [0.0]])
        return p

    if args.check:
        from oracle import COracle
        orc = COracle(1)
        for _ in range(200):
            p = step()
            max_err = max(max_err, float(np.abs(np.array(p) - orc.fk([r.astype(np.float64)])[0]).max()))
    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt = time.perf_counter() - t0
    out = {"pybullet": "ok", "env_steps_per_s": args.steps / dt, "steps": args.steps, "cores": 1}
    if args.check:
        out["fk_max_abs_err_vs_oracle"] = max_err
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
