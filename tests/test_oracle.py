"""CPU tests of the oracle: known answers, the reference's URDF data, the NumPy twin, properties.

PARITY UNPINNED: the reference ships no golden vectors and cannot run here
(pybullet/gym absent), so the oracle is pinned by (i) the URDF numbers extracted
from the reference's own data file (tests/golden/urdf_chain.json), (ii) the
analytic KATs of SURVEY.md Appendix C, (iii) Random123's Philox KATs, and
(iv) agreement of two independent restatements (C closed form vs NumPy generic chain).
"""
import json
import os

import numpy as np
import pytest

from oracle import COracle, numpy_twin as T
from oracle.binding import ORC_DEV, ORC_REF

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(GOLD, "kat_appendix_c.json")))
URDF = json.load(open(os.path.join(GOLD, "urdf_chain.json")))


@pytest.fixture(scope="module")
def orc(oracle_built):
    return COracle(1)


# ---- reference data: the URDF -----------------------------------------------------------
def urdf_chain_table():
    """(type, xyz, axis) in chain order from the extracted URDF numbers."""
    by_parent = {j["parent"]: j for j in URDF["joints"]}
    chain, link = [], "world"
    while link in by_parent:
        j = by_parent[link]
        assert j["rpy"] == [0, 0, 0]
        chain.append((j["type"], tuple(j["xyz"]), tuple(j["axis"]) if j["axis"] else None, j))
        link = j["child"]
    return chain, link


def test_urdf_fixture_matches_reference_file_when_present():
    """The committed fixture is exactly what the extraction script yields from the reference."""
    ref = "/root/reference/pioneer/envs/pioneer/assets/pioneer_knm_6dof.urdf"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    import subprocess, sys, tempfile, shutil
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(os.path.join(GOLD, "make_urdf_chain.py"), tmp)
        subprocess.run([sys.executable, os.path.join(tmp, "make_urdf_chain.py"), "/root/reference"], check=True,
                       stdout=subprocess.DEVNULL)
        assert json.load(open(os.path.join(tmp, "urdf_chain.json"))) == URDF
    finally:
        shutil.rmtree(tmp)


def test_urdf_structure_assumptions():
    chain, tip = urdf_chain_table()
    assert tip == "robot:pointer" and len(chain) == 11
    assert [c[0] for c in chain].count("revolute") == 6
    # no colliders, no joint dynamics, unit inertials: what makes stepSimulation a no-op
    assert all(l["n_collision"] == 0 for l in URDF["links"].values())
    assert not any(j["has_dynamics"] for j in URDF["joints"])
    for name, l in URDF["links"].items():
        if name == "world":
            assert not l["has_inertial"]
        else:
            assert l["mass"] == 1.0 and l["inertial_xyz"] == [0, 0, 0]
            assert l["inertia"] == {"ixx": 1.0, "ixy": 0.0, "ixz": 0.0, "iyy": 1.0, "iyz": 0.0, "izz": 1.0}


def test_twin_chain_table_equals_urdf():
    chain, _ = urdf_chain_table()
    assert len(chain) == len(T.CHAIN)
    for (jt, xyz, axis, j), (tt, txyz, taxis) in zip(chain, T.CHAIN):
        assert jt == tt and tuple(map(float, xyz)) == tuple(map(float, txyz))
        assert (axis is None and taxis is None) or tuple(map(float, axis)) == tuple(map(float, taxis))
    lims = [j["upper"] for (jt, _, _, j) in chain if jt == "revolute"]
    assert lims == list(T.LIMITS)
    assert all(j["lower"] == -j["upper"] for (jt, _, _, j) in chain if jt == "revolute")


def test_oracle_limits_equal_urdf(orc):
    chain, _ = urdf_chain_table()
    lims = np.array([j["upper"] for (jt, _, _, j) in chain if jt == "revolute"], dtype=np.float32)
    assert np.array_equal(orc.r_hi, lims) and np.array_equal(orc.r_lo, -lims)


def test_fk_closed_form_vs_generic_urdf_chain(orc):
    """C oracle's closed-form FK == generic 4x4 composition over the URDF numbers."""
    chain, _ = urdf_chain_table()
    table = [(jt, xyz, axis) for (jt, xyz, axis, _) in chain]
    rng = np.random.RandomState(0)
    q = rng.uniform(orc.r_lo, orc.r_hi, size=(500, 6))
    q = np.vstack([q, np.zeros(6), orc.r_lo, orc.r_hi])
    got = orc.fk(q)
    want = np.array([T.fk_chain(x, table) for x in q])
    assert np.abs(got - want).max() < 1e-12


# ---- known answers ----------------------------------------------------------------------
def test_fk_kats(orc):
    for case in KAT["fk"]:
        q = {"r_hi": orc.r_hi, "r_lo": orc.r_lo}.get(case["q"], case["q"]) if isinstance(case["q"], str) else case["q"]
        got = orc.fk([np.asarray(q, dtype=np.float64)])[0]
        assert np.abs(got - np.array(case["xyz"])).max() < 5e-9


def test_potential_kats(orc):
    for d, want in KAT["potential"]:
        assert abs(orc.potential(d) - want) < 5e-9


def test_constants(orc):
    c = KAT["constants"]
    assert np.array_equal(orc.r_hi, np.array(c["r_hi"], dtype=np.float32))
    assert np.allclose(orc.r_hi.astype(np.float64), c["r_hi_f32"], atol=5e-10)
    assert np.allclose(orc.v_max, c["v_max"], rtol=1e-6) and np.allclose(orc.a_max, c["a_max"], rtol=1e-6)
    assert orc.dt == c["dt"] == (1 / 240) * 10 and orc.p.eps == c["eps"]
    # float32 arithmetic of pioneer_knm_env.py:57-58
    span = orc.r_hi - orc.r_lo
    assert np.array_equal(orc.v_max, np.float32(2) * span)
    assert np.array_equal(orc.a_max, np.float32(10) * orc.v_max)


def test_integrator_trace(oracle_built):
    o = COracle(1, precision=ORC_REF, max_episode_steps=0)
    o.reset(joint_pos=np.zeros((1, 6)), target_pos=[[20, 0, 4]])
    rows = {r[0]: r[1:] for r in KAT["integrator_trace"]["rows"]}
    for t in range(1, 10):
        o.step(o.a_max[None])
        st = o.state[0]
        if t in rows:
            v0, r0, v1, r1 = rows[t]
            assert abs(st["v"][0] - v0) < 1e-6 and abs(st["r"][0] - r0) < 1e-6
            assert abs(st["v"][1] - v1) < 1e-6 and abs(st["r"][1] - r1) < 1e-6
    st = o.state[0]
    assert st["r"][1] == np.float32(1.309) and st["v"][1] == 0.0    # joint 1 parked at r_hi at t = 9


def test_philox_kats(orc):
    for v in KAT["philox4x32_10"]["vectors"]:
        out = orc.philox([int(x, 16) for x in v["ctr"]], [int(x, 16) for x in v["key"]])
        assert out == [int(x, 16) for x in v["out"]]


# ---- the two restatements against each other ----------------------------------------------
def test_c_oracle_vs_numpy_twin_trajectories(oracle_built):
    rng = np.random.RandomState(42)
    for ep in range(6):
        o = COracle(1, precision=ORC_REF, max_episode_steps=0)
        t = T.TwinEnv()
        jp = rng.uniform(t.r_lo, t.r_hi)
        tp = rng.uniform([15, -10, 2], [25, 10, 6])
        ob_c = o.reset(joint_pos=jp[None], target_pos=tp[None])[0]
        ob_t = t.reset_world(jp, tp)
        # right after reset r is float64 (quirk Q5): r, r - r_lo, r_hi - r groups are float64-exact;
        # limit / v / a groups are float32 arrays whose np.cos is within ~1.5 ulp of the rounded double
        f64_groups = np.r_[0:18, 54:90, 126:137]
        assert np.abs(ob_c - ob_t)[f64_groups].max() < 1e-12
        assert np.abs(ob_c - ob_t).max() < 2.5e-7
        scale = [1.0, 1.0, 3.0, 0.2, 1.0, 1.0][ep]
        for k in range(60):
            a = (rng.uniform(-1, 1, 6) * t.a_max * scale).astype(np.float32)
            if ep == 4:
                a = (np.sign(a) * t.a_max).astype(np.float32)   # bang-bang: saturation + limits
            ob_c, rw_c, dn_c, _ = o.step(a[None])
            ob_t, rw_t, dn_t, info = t.step(a)
            st = o.state[0]
            assert np.array_equal(st["v"], t.v) and np.array_equal(st["r"].astype(np.float32), t.r)
            assert np.array_equal(st["a"], t.a)
            # np.cos/np.sin of float32 arrays are within ~1.5 ulp of the rounded-double value
            assert np.abs(ob_c[0] - ob_t).max() < 2.5e-7
            lin = np.r_[0:6, 18:24, 36:42, 54:60, 72:78, 90:96, 108:114, 126:137]
            assert np.abs(ob_c[0][lin] - ob_t[lin]).max() < 1e-11
            assert abs(rw_c[0] - rw_t) < 1e-11 and bool(dn_c[0]) == dn_t


def test_first_step_integrates_zero_action_and_potential_quirk(oracle_built):
    """Q1: the action given at step t moves the arm at t+1.  Q3: potential starts at 0."""
    o = COracle(1, precision=ORC_REF)
    jp = np.array([[0.1, 0.2, -0.3, 0.4, 0.5, -0.6]])
    ob0 = o.reset(joint_pos=jp, target_pos=[[20, 1, 3]])[0]
    assert ob0[136] == 0.0
    ob1, rw, dn, tr = o.step(o.a_max[None])
    assert np.array_equal(ob1[0][0:6], jp[0].astype(np.float32))     # r unchanged (a was 0)
    assert np.array_equal(ob1[0][108:114], o.a_max)                  # obs already shows the new action
    dist = ob1[0][135]
    assert abs(rw[0] - (95 / (dist / 10 + 1) - 0.01)) < 1e-12        # full potential on the first step


def test_done_and_award(oracle_built):
    o = COracle(1, precision=ORC_REF)
    q = np.array([[0.5, -0.4, 0.9, 1.2, -0.7, 2.0]])
    tip = o.fk(q)[0]
    # reachable pose but target box is x in [15,25]: use the override, as the reference allows
    o.reset(joint_pos=q, target_pos=[tip + np.array([0.05, 0, 0])])
    ob, rw, dn, tr = o.step(np.zeros((1, 6), np.float32))
    assert dn[0] == 1 and abs(ob[0][135] - 0.05) < 1e-6
    assert abs(rw[0] - (95 / (0.05 / 10 + 1) - 0.01 + 5.0)) < 1e-4


# ---- storage model and properties ---------------------------------------------------------
def test_dev_vs_ref_precision_is_benign(oracle_built):
    """ORC_DEV (float32 storage at reset, what the GPU does) vs ORC_REF: v identical, r within 1 ulp-ish."""
    n = 512
    a = COracle(n, seed=3, precision=ORC_REF, max_episode_steps=0)
    b = COracle(n, seed=3, precision=ORC_DEV, max_episode_steps=0)
    oa, ob = a.reset(), b.reset()
    # rounding r to float32 moves a joint by <= 1.2e-7 rad; with a ~30-unit arm the pointer moves <= ~4e-6
    assert np.abs(oa - ob).max() < 1e-5
    rng = np.random.RandomState(1)
    for _ in range(50):
        act = (rng.uniform(-1, 1, (n, 6)) * a.a_max).astype(np.float32)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert np.abs(a.state["r"] - b.state["r"]).max() < 5e-7
        assert np.abs(a.state["v"] - b.state["v"]).max() < 1e-5
        assert np.abs(oa - ob).max() < 2e-5
        assert np.abs(ra - rb).max() < 5e-5


def test_properties_random_rollout(oracle_built):
    n = 2048
    o = COracle(n, seed=9, precision=ORC_DEV, auto_reset=True, max_episode_steps=37, nthreads=4)
    obs = o.reset()
    rng = np.random.RandomState(2)
    episodes_before = o.state["episode"].copy()
    for t in range(80):
        act = (rng.uniform(-1.5, 1.5, (n, 6)) * o.a_max).astype(np.float32)
        obs, rew, done, trunc = o.step(act)
        r, v = obs[:, 0:6], obs[:, 90:96]
        assert np.all(r >= o.r_lo) and np.all(r <= o.r_hi)
        assert np.all(np.abs(v) <= o.v_max)
        for base in (6, 24, 42, 60, 78, 96, 114):
            c, s = obs[:, base:base + 6], obs[:, base + 6:base + 12]
            assert np.abs(c * c + s * s - 1).max() < 3e-7
        assert np.allclose(obs[:, 129:132] - obs[:, 126:129], obs[:, 132:135], atol=1e-12)
        assert np.allclose(np.linalg.norm(obs[:, 132:135], axis=1), obs[:, 135], atol=1e-12)
        assert np.array_equal(obs[:, 54:60], (obs[:, 0:6].astype(np.float32) - o.r_lo).astype(np.float64))
        # TimeLimit: truncated only at the cap and never together with done
        assert not np.any(done & trunc)
        if (t + 1) % 37 == 0:
            assert np.all(trunc | done)
            assert np.all(obs[:, 136] == 0) and np.all(obs[:, 90:126:1][:, :6] == 0)   # fresh episodes: pot = 0, v = 0
    assert np.all(o.state["episode"] == episodes_before + 2)


def test_reset_draws_uniform_and_keyed_by_global_id(oracle_built):
    a = COracle(4096, seed=5, precision=ORC_DEV)
    a.reset(want_obs=False)
    r, tg = a.state["r"], a.state["target"]
    assert np.all(r >= a.r_lo) and np.all(r <= a.r_hi)
    assert np.all(tg >= [15, -10, 2]) and np.all(tg <= [25, 10, 6])
    assert np.abs(r.mean(0)).max() < 0.15 and np.abs(tg.mean(0) - [20, 0, 4]).max() < 0.3
    # a shard with an offset reproduces the same envs
    b = COracle(1024, seed=5, precision=ORC_DEV, env_id_offset=2048)
    b.reset(want_obs=False)
    assert np.array_equal(b.state["r"], r[2048:3072]) and np.array_equal(b.state["target"], tg[2048:3072])
    c = COracle(1024, seed=6, precision=ORC_DEV)
    c.reset(want_obs=False)
    assert not np.array_equal(c.state["r"], r[:1024])


def test_state_words_roundtrip(oracle_built):
    a = COracle(33, seed=1, precision=ORC_DEV)
    a.reset()
    a.step(np.ones((33, 6), np.float32))
    w = a.state_words()
    b = COracle(33, seed=1, precision=ORC_DEV)
    b.load_state_words(w)
    assert np.array_equal(b.state_words(), w)
    oa = a.step(np.ones((33, 6), np.float32))
    ob = b.step(np.ones((33, 6), np.float32))
    assert np.array_equal(oa[0], ob[0]) and np.array_equal(oa[1], ob[1])


def test_oracle_reproduces_its_frozen_trajectory(oracle_built):
    """tests/golden/oracle_traj_v1.npz (made by make_oracle_traj.py from the oracle itself) bit for bit."""
    z = np.load(os.path.join(GOLD, "oracle_traj_v1.npz"))
    n = z["actions"].shape[1]
    o = COracle(n, seed=int(z["seed"]), precision=ORC_REF, auto_reset=True, max_episode_steps=int(z["max_episode_steps"]))
    assert np.array_equal(o.reset(), z["obs0"])
    for t in range(z["actions"].shape[0]):
        ob, r, d, tr = o.step(z["actions"][t])
        assert np.array_equal(ob, z["obs"][t]) and np.array_equal(r, z["reward"][t])
        assert np.array_equal(d, z["done"][t]) and np.array_equal(tr, z["truncated"][t])
    assert np.array_equal(o.state_words(), z["state_words"])
