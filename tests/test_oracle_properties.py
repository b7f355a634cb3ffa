"""Property-based tests (hypothesis) of the oracle's step: invariants that must hold for ANY state/action."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import COracle, numpy_twin as T
from oracle.binding import ORC_REF

f32 = st.floats(min_value=-1.0, max_value=1.0, allow_nan=False, width=32)


@settings(max_examples=150, deadline=None)
@given(q=st.lists(f32, min_size=6, max_size=6), acts=st.lists(st.lists(f32, min_size=6, max_size=6), min_size=1, max_size=12),
       scale=st.floats(min_value=0.0, max_value=8.0), tgt=st.lists(f32, min_size=3, max_size=3))
def test_step_invariants_and_twin_agreement(q, acts, scale, tgt):
    o = COracle(1, precision=ORC_REF, max_episode_steps=0)
    t = T.TwinEnv()
    jp = np.array(q) * t.r_hi
    tp = np.array([20, 0, 4]) + np.array(tgt) * np.array([5, 10, 2])
    o.reset(joint_pos=jp[None], target_pos=tp[None]); t.reset_world(jp, tp)
    prev_pot = 0.0
    for a in acts:
        a = (np.array(a) * t.a_max * scale).astype(np.float32)
        obs, rew, done, trunc = o.step(a[None])
        tobs, trew, tdone, _ = t.step(a)
        s = o.state[0]
        # two independent restatements agree on the integrator bit for bit
        assert np.array_equal(s["v"], t.v) and np.array_equal(s["r"].astype(np.float32), t.r)
        # physical / structural invariants
        assert np.all(s["r"] >= o.r_lo) and np.all(s["r"] <= o.r_hi) and np.all(np.abs(s["v"]) <= o.v_max)
        parked = (s["r"].astype(np.float32) == o.r_hi) | (s["r"].astype(np.float32) == o.r_lo)
        assert np.all(s["v"][parked] == 0)                       # clamped joints have zero velocity
        d = obs[0][135]
        assert abs(d - np.linalg.norm(obs[0][129:132] - obs[0][126:129])) < 1e-12
        pot = 95.0 / (d / 10.0 + 1.0)
        assert abs(obs[0][136] - pot) < 1e-12 and 0 < pot <= 95.0
        assert abs(rew[0] - ((pot - prev_pot) - 0.01 + (5.0 if d < 0.1 else 0.0))) < 1e-12
        assert bool(done[0]) == (d < 0.1) and abs(rew[0] - trew) < 1e-11
        prev_pot = pot


@settings(max_examples=100, deadline=None)
@given(seed=st.integers(min_value=0, max_value=2**63 - 1), env=st.integers(min_value=0, max_value=2**40))
def test_reset_draws_in_range_for_any_seed_and_env_id(seed, env):
    o = COracle(3, seed=seed, env_id_offset=env)
    o.reset(want_obs=False)
    assert np.all(o.state["r"] >= o.r_lo) and np.all(o.state["r"] <= o.r_hi)
    assert np.all(o.state["target"] >= [15, -10, 2]) and np.all(o.state["target"] <= [25, 10, 6])
    assert len({tuple(r) for r in o.state["r"]}) == 3             # distinct envs draw distinct poses
