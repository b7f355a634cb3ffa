"""The gym / RLlib branch of pioneer_amd/compat.py, exercised against stand-ins for the two packages' public base classes
(tests/stubs: neither package is installed here, so this is the only way that branch runs before a user's machine does).

The reference env is a ``gym.Env`` under ``gym.wrappers.TimeLimit`` (bullet_env.py:65, pioneer_knm_env.py:38,
pioneer_knm_train.py:27) and RLlib accepts ``gym.Env`` / ``VectorEnv`` objects from ``register_env`` creators."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "stubs")


def run_child(code: str, ray_style: str = "old", timeout: int = 600) -> str:
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUBS, ROOT, os.environ.get("PYTHONPATH", "")]), PNR_STUB_RAY_STYLE=ray_style)
    out = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout


def test_stubs_are_not_importable_from_the_test_process():
    import importlib.util
    assert importlib.util.find_spec("gym") is None or "stubs" not in (importlib.util.find_spec("gym").origin or "")


def test_facade_classes_derive_from_gym_and_rllib_when_they_exist():
    out = run_child("""
        import numpy as np
        import gym, gym.spaces
        from ray.rllib.env.vector_env import VectorEnv
        from pioneer_amd import compat
        assert compat.HAVE_GYM and compat.HAVE_RLLIB
        assert compat.GymEnv is gym.Env and compat.GymWrapper is gym.Wrapper and compat.RLlibVectorEnv is VectorEnv
        from pioneer_amd.env import PioneerKinematicEnv, TimeLimit
        from pioneer_amd.rllib_env import PioneerRLlibVectorEnv
        assert issubclass(PioneerKinematicEnv, gym.Env) and issubclass(TimeLimit, gym.Wrapper) and issubclass(PioneerRLlibVectorEnv, VectorEnv)
        from pioneer_amd.spaces import Box
        sp = compat.to_gym_space(Box(-np.inf, np.inf, shape=(137,), dtype=np.float64))
        assert isinstance(sp, gym.spaces.Box) and sp.shape == (137,) and sp.dtype == np.float64
        act = compat.to_gym_space(Box(-np.ones(6, np.float32), np.ones(6, np.float32), dtype=np.float32))
        assert isinstance(act, gym.spaces.Box) and act.dtype == np.float32 and act.contains(np.zeros(6, np.float32))

        # TimeLimit over a gym.Env: gym.wrappers.TimeLimit's protocol through gym.Wrapper's constructor
        class Dummy(gym.Env):
            action_space = act
            observation_space = sp
            marker = "dummy"
            def __init__(self): self.t = 0
            def reset(self): self.t = 0; return np.zeros(137)
            def step(self, a): self.t += 1; return np.full(137, float(self.t)), 1.0, self.t == 5, {}
        env = TimeLimit(Dummy(), max_episode_steps=3)
        assert isinstance(env, gym.Wrapper) and env.action_space is act and env.observation_space is sp and env.unwrapped.marker == "dummy"
        assert env.marker == "dummy"                       # public attributes are forwarded
        try:
            env._no_such_private
            raise SystemExit("private names must not be forwarded")
        except AttributeError:
            pass
        try:
            env.step(np.zeros(6))
            raise SystemExit("step before reset must assert")
        except AssertionError:
            pass
        env.reset()
        o, r, d, info = env.step(np.zeros(6)); assert not d and "TimeLimit.truncated" not in info
        o, r, d, info = env.step(np.zeros(6)); assert not d
        o, r, d, info = env.step(np.zeros(6)); assert d and info["TimeLimit.truncated"] is True      # the cut, env not done
        env2 = TimeLimit(Dummy(), max_episode_steps=5)
        env2.reset()
        for _ in range(4): env2.step(np.zeros(6))
        o, r, d, info = env2.step(np.zeros(6)); assert d and info["TimeLimit.truncated"] is False      # done at the cut: not truncated
        print("ok")
    """)
    assert out.strip().endswith("ok")


CHILD_GPU = """
    import pickle
    import numpy as np
    import gym, gym.spaces
    from ray.rllib.env.vector_env import VectorEnv
    from pioneer_amd.env import make_env, make_vector_env, PioneerKinematicEnv
    cfg = {"award_potential_slope": 10.0, "award_done": 5.0, "penalty_step": 0.01}
    env = make_env(cfg)                                     # pioneer_knm_train.py:20-27
    assert isinstance(env, gym.Wrapper) and isinstance(env.env, gym.Env) and isinstance(env.unwrapped, PioneerKinematicEnv)
    assert isinstance(env.action_space, gym.spaces.Box) and isinstance(env.observation_space, gym.spaces.Box)
    assert env.observation_space.dtype == np.float64 and env.observation_space.shape == (137,) and env.action_space.shape == (6,)
    env.seed(7)
    obs = env.reset()
    assert obs.shape == (137,) and obs.dtype == np.float64 and env.observation_space.contains(obs)
    total = 0.0
    for t in range(5):
        obs, rew, done, info = env.step(np.clip(np.full(6, 0.1, np.float32), env.action_space.low, env.action_space.high))
        assert isinstance(rew, float) and isinstance(done, bool) and env.observation_space.contains(obs)
        total += rew
    assert env.dof == 6 and env.metadata["video.frames_per_second"] > 0 and env.reward_range == (-float("inf"), float("inf"))
    clone = pickle.loads(pickle.dumps(env.unwrapped))       # EzPickle-style: by constructor arguments
    assert isinstance(clone, gym.Env)
    clone.close()
    env.close()

    venv = make_vector_env(dict(cfg, num_envs=64, seed=3))
    assert isinstance(venv, VectorEnv) and venv.num_envs == 64
    assert isinstance(venv.observation_space, gym.spaces.Box) and isinstance(venv.action_space, gym.spaces.Box)
    rows = venv.vector_reset()
    assert len(rows) == 64 and rows[0].shape == (137,) and rows[0].dtype == np.float64
    obs, rew, done, infos = venv.vector_step([np.zeros(6, np.float32)] * 64)
    assert len(obs) == len(rew) == len(done) == len(infos) == 64 and isinstance(rew[0], float) and isinstance(done[0], bool)
    assert venv.reset_at(5).shape == (137,)
    venv.close()
    print("ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("ray_style", ["old", "new"])
def test_creators_return_gym_and_rllib_objects_on_the_gpu(ray_style):
    """``make_env`` / ``make_vector_env`` (what ``register_env`` is handed) with gym and ray present: real subclasses, real
    ``gym.spaces.Box`` spaces, and both VectorEnv constructor generations (ray 0.8.x without arguments, ray >= 1.0 with three)."""
    assert run_child(CHILD_GPU, ray_style=ray_style).strip().endswith("ok")
