"""The reference's seed -> reset stream (pioneer_amd/seeding.py: gym.utils.seeding.np_random of gym <= 0.21, restated).

Known answers: gym's CartPole-v0/v1 draws its initial state as np_random.uniform(-0.05, 0.05, size=(4,)); the states
for env.seed(0) and env.seed(42) under gym 0.1x-0.21 are printed in countless public tutorials and issue threads.  They
pin the whole chain (create_seed, the sha512 hash with gym's padding quirk, RandomState.seed by array)."""
import numpy as np
import pytest

from pioneer_amd import seeding


def test_gym_seeding_known_answers():
    rng, s = seeding.np_random(0)
    assert s == 0
    assert np.allclose(rng.uniform(low=-0.05, high=0.05, size=(4,)), [-0.04456399, 0.04653909, 0.01326909, -0.02099827], atol=5e-9)
    rng, s = seeding.np_random(42)
    assert s == 42
    assert np.allclose(rng.uniform(low=-0.05, high=0.05, size=(4,)), [-0.01258566, -0.00156614, 0.04207708, -0.00180545], atol=5e-9)


def test_seed_rules():
    assert seeding.create_seed(2 ** 64 + 5) == 5                       # reduced modulo 2**64
    assert seeding._int_list_from_bigint(0) == [0] and seeding._int_list_from_bigint(2 ** 32 + 7) == [7, 1]
    assert seeding._bigint_from_bytes(b"\x01\x00\x00\x00\x02\x00\x00\x00") == 1 + 2 * 2 ** 32   # + a zero word of padding
    for bad in (-1, 1.5, "7"):
        with pytest.raises(ValueError):
            seeding.np_random(bad)
    a, sa = seeding.np_random(None); b, sb = seeding.np_random(None)
    assert sa != sb and 0 <= sa < 2 ** 64                               # fresh entropy when no seed is given
    r1, _ = seeding.np_random(123); r2, _ = seeding.np_random(123)
    assert np.array_equal(r1.uniform(size=9), r2.uniform(size=9))


@pytest.mark.gpu
def test_facade_reset_draws_follow_the_reference_stream():
    """env.seed(s); env.reset() starts from np_random.uniform(r_lo, r_hi) and then np_random.uniform(target_lo, target_hi)
    (pioneer_knm_env.py:80-90), rounded to the engine's float32 state; overrides consume no draws, as in the reference."""
    from pioneer_amd import PioneerKinematicEnv
    env = PioneerKinematicEnv()
    assert env.seed(7) == [7]
    rng, _ = seeding.np_random(7)
    lo, hi = env.joint_limits()
    cfg = env.config
    for episode in range(3):
        obs = env.reset()
        r = rng.uniform(lo, hi)
        t = rng.uniform(np.array(cfg.target_lo), np.array(cfg.target_hi))
        assert np.array_equal(env.r, r.astype(np.float32))
        assert np.array_equal(obs[129:132].astype(np.float32), t.astype(np.float32))       # the target entries of the observation
    env.reset_world(joint_positions=np.zeros(6))                        # only the target is drawn
    t = rng.uniform(np.array(cfg.target_lo), np.array(cfg.target_hi))
    assert np.array_equal(env._state()["target"][0], t.astype(np.float32)) and not env.r.any()
    env.reset_world(joint_positions=np.zeros(6), target_position=(20.0, 0.0, 4.0))      # nothing is drawn
    env.reset()
    assert np.array_equal(env.r, rng.uniform(lo, hi).astype(np.float32))
