"""Single-env façade with the reference's ``gym.Env`` surface.

``PioneerKinematicEnv`` here is what RLlib's ``register_env`` creator would return
in place of pioneer.envs.pioneer.PioneerKinematicEnv (pioneer_knm_env.py:37-242):
``reset() -> obs[137]``, ``step(action[6]) -> (obs, reward, done, info)``,
``seed``, ``action_space``, ``observation_space``, ``reward_range``, ``metadata``,
``reset_world(joint_positions, target_position)``, ``dof``, ``joint_limits()``,
``joint_positions()``.  It is a one-env batch of the HIP engine (no CPU path);
``TimeLimit`` reproduces gym.wrappers.TimeLimit as used at pioneer_knm_train.py:27.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import compat, seeding
from .config import EngineConfig, PioneerKinematicConfig, RenderConfig, SimulationConfig
from .spaces import Box
from .vector_env import PioneerVectorEnv

Action = np.ndarray
Observation = np.ndarray


def arr2str(arr, fmt: str = ".3f") -> str:
    """pioneer/collections_util.py:13-14."""
    return "[" + ", ".join([f"{x:{fmt}}" for x in arr]) + "]"


class PioneerKinematicEnv(compat.GymEnv):
    """A ``gym.Env`` subclass where gym is importable (bullet_env.py:65, pioneer_knm_env.py:38: RLlib type-checks for it),
    a plain class with the same surface otherwise; the spaces are real ``gym.spaces.Box`` objects in the first case."""

    def __init__(self,
                 headless: bool = True,
                 pioneer_config: Optional[PioneerKinematicConfig] = None,
                 simulation_config: Optional[SimulationConfig] = None,
                 render_config: Optional[RenderConfig] = None,
                 device=None,
                 mode: str = "kinematic",
                 engine_config: Optional[EngineConfig] = None):
        """``engine_config`` (optional, no reference counterpart): the engine's own options, e.g. the dynamics-mode motor and
        scene; its ``mode`` wins over the ``mode`` argument, and the env itself never truncates or auto-resets."""
        self._ctor = dict(headless=headless, pioneer_config=pioneer_config, simulation_config=simulation_config,
                          render_config=render_config, device=device, mode=mode, engine_config=engine_config)
        self.headless = headless
        self.config = pioneer_config or PioneerKinematicConfig()
        self.simulation_config = simulation_config or SimulationConfig()
        self.render_config = render_config or RenderConfig()
        self.metadata = {                                                  # bullet_env.py:76-79
            "render.modes": ["human", "rgb_array"],
            "video.frames_per_second": self.simulation_config.frames_per_second,
        }
        self.world_index = -1                                              # bullet_env.py:85
        self.step_index = 0

        # the env itself never truncates or auto-resets: TimeLimit / the sampler do
        if engine_config is not None:
            import dataclasses
            engine = dataclasses.replace(engine_config, max_episode_steps=0, auto_reset=False)
        else:
            engine = EngineConfig(max_episode_steps=0, auto_reset=False, mode=mode)
        self.np_random = None                                              # :45
        self.seed()                                                        # :46
        self._base_engine = engine                                         # what reset() returns to (scene bodies of the constructor kept)
        self._vec = PioneerVectorEnv(1, device=device, seed=self._seed_value,
                                     pioneer_config=self.config,
                                     simulation_config=self.simulation_config, engine_config=engine)
        self.r_lo, self.r_hi = self.joint_limits()                         # :56
        self.v_max = self._vec.v_max                                       # :57
        self.a_max = self._vec.a_max                                       # :58
        self.dt = self._vec.dt                                             # :60
        self.eps = self._vec.eps                                           # :61

        from .scene import Scene, World
        self._motor_cmds = {}                                              # Joint.control_*: joint -> pnr_set_joint_motor arguments
        self.scene = Scene(self)                                           # bullet_env.py:86-88: the objects the demo drives
        self.world = World(self)
        self._obs = self.reset_world()                                     # :69 (+ reset_simulator in BulletEnv.__init__)
        self.action_space = compat.to_gym_space(Box(-self.a_max, self.a_max, dtype=np.float32))  # :72
        self.observation_space = compat.to_gym_space(self.observation_to_space(self.observe()))  # :73
        self.reward_range = (-float("inf"), float("inf"))                  # :74

    def _rebuild_engine(self, engine_config: EngineConfig) -> None:
        """A new engine handle with another EngineConfig (scene bodies are part of pnr_config at pnr_create), the env's state carried over."""
        old = self._vec
        state = old.get_state().clone()
        dyn = old.get_dyn_state().clone() if old.engine_config.mode == "dynamic" else None
        self._vec = PioneerVectorEnv(1, device=old.device, seed=self._seed_value, pioneer_config=self.config,
                                     simulation_config=self.simulation_config, engine_config=engine_config)
        self._vec.reset()
        self._vec.set_state(state)
        if dyn is not None:
            self._vec.set_dyn_state(dyn)
            for args in self._motor_cmds.values():
                self._vec.set_joint_motor(*args)
        old.close()

    # -- pickling: by constructor arguments, like gym.utils.EzPickle (pioneer_knm_env.py:38, :51) --
    def __reduce__(self):
        return (_rebuild_env, (self._ctor,))

    # -- randomness ------------------------------------------------------------------
    def seed(self, seed=None) -> List[int]:                                # :107-109
        """The reference's seeding: gym.utils.seeding.np_random (pioneer_amd/seeding.py restates it), a NumPy RandomState
        whose draws reset_world() takes exactly as the reference does — so ``env.seed(s); env.reset()`` starts from the
        reference's joint angles and target.  (The batched engine's own per-env Philox streams are for PioneerVectorEnv.)"""
        self.np_random, seed = seeding.np_random(seed)
        self._seed_value = int(seed)
        if getattr(self, "_vec", None) is not None:
            self._vec.seed(self._seed_value)
        return [seed]

    # -- state mirrors -------------------------------------------------------------------
    def _state(self):
        return self._vec.state_dict()

    @property
    def a(self) -> np.ndarray: return self._state()["a"][0]
    @property
    def v(self) -> np.ndarray: return self._state()["v"][0]
    @property
    def r(self) -> np.ndarray: return self._state()["r"][0]
    @property
    def potential(self) -> float: return float(self._state()["potential"][0])
    @property
    def dof(self) -> int: return self._vec.dof                            # :213-215

    def joint_limits(self) -> Tuple[np.ndarray, np.ndarray]:               # :217-220
        return self._vec.r_lo.copy(), self._vec.r_hi.copy()

    def joint_positions(self) -> np.ndarray:                               # :222-223
        return self.r.astype(np.float64)

    # -- reset / step ----------------------------------------------------------------------
    def reset_world(self, joint_positions: Optional[np.ndarray] = None,
                    target_position: Optional[Tuple[float, float, float]] = None) -> Observation:
        """pioneer_knm_env.py:76-105; returns the observation of the new state."""
        if joint_positions is None:
            joint_positions = self.np_random.uniform(self.r_lo, self.r_hi)  # :80-81 (float64 draws between the float32 limits)
        if target_position is None:
            assert len(self.config.target_lo) == 3                         # :84
            assert len(self.config.target_hi) == 3                         # :85
            target_position = tuple(self.np_random.uniform(np.array(self.config.target_lo), np.array(self.config.target_hi)))  # :87-90
        jp = tp = None
        if joint_positions is not None:
            positions_list = list(joint_positions)
            assert len(positions_list) == self.dof                         # :227
            jp = np.asarray(positions_list, dtype=np.float32)[None]
        if target_position is not None:
            assert len(target_position) == 3
            tp = np.asarray(target_position, dtype=np.float32)[None]
        obs = self._vec.reset(joint_positions=jp, target_positions=tp)
        self._obs = obs[0].double().cpu().numpy()
        # :96-102: the target marker is an item of the scene (visual only).  (The reference asserts on a second 'target' when
        # reset_world() is called twice without reset_simulator(), quirk Q9; here the item is replaced.)
        from .scene import Item
        tgt = Item("target", "sphere", self._obs[129:132], (0.0, 0.0, 0.0, 1.0), False, (self.config.target_radius, 0.0, 0.0))
        self.scene.items = [i for i in self.scene.items if i.name != "target"] + [tgt]
        self.scene.items_by_name["target"] = tgt
        self.scene.sync_from_env()
        return self._obs

    def reset(self) -> Observation:                                        # bullet_env.py:187-190
        # reset_simulator() (:90-101): a fresh World and Scene — bodies created through env.scene are gone, as in the reference
        from .scene import Scene, World
        if self._vec.engine_config != self._base_engine:
            self._rebuild_engine(self._base_engine)
        self._motor_cmds = {}
        self.scene, self.world = Scene(self), World(self)
        self.world_index += 1
        self.step_index = 0
        return self.reset_world()

    def step(self, action: Action) -> Tuple[Observation, float, bool, Dict]:  # bullet_env.py:192-197
        self.step_index += 1
        act = np.asarray(action, dtype=np.float32).reshape(1, self.dof)
        obs, rew, done, _trunc, info = self._vec.vector_step(torch.from_numpy(act), want_info=True)
        obs = obs[0].double().cpu().numpy()                                # float64[137], quirk Q6
        reward = float(rew[0].item())
        is_done = bool(done[0].item())
        r_pot, r_step, r_done, dist = (float(x) for x in info[0].cpu().numpy())
        st = self._state()
        info_dict = {                                                      # :167-179
            "r_pot": f"{r_pot:.3f}", "r_step": f"{r_step:.3f}", "r_done": f"{r_done:.3f}",
            "rw": f"{reward:.3f}",
            "dist": f"{dist:.3f}", "pot": f"{float(st['potential'][0]):.3f}",
            "a": arr2str(st["a"][0]), "v": arr2str(st["v"][0]), "r": arr2str(st["r"][0]),
        }
        self._obs = obs
        self.scene.sync_from_env()                                         # :148: act() teleported the joints into the simulator
        return obs, reward, is_done, info_dict

    def observe(self) -> Observation:                                      # :184-211
        return self._vec.observe()[0].double().cpu().numpy()

    def render(self, mode="human"):                                        # bullet_env.py:156-185
        if mode == "human":
            return None
        elif mode == "rgb_array":
            from .render import render_rgb                                 # host-side stick-figure rasteriser
            st = self._state()
            q = st["r"][0]
            if self._vec.engine_config.mode == "dynamic":                  # the simulated joints, not the command
                q = self._vec.get_dyn_state()[0:6, 0].cpu().numpy()
            return render_rgb(q, st["target"][0], self.render_config, self.config.target_radius)
        else:
            raise AssertionError(f'Render mode "{mode}" is not supported')

    @staticmethod
    def observation_to_space(observation: Observation) -> Box:             # :238-242
        low = np.full(observation.shape, -float("inf"), dtype=np.float32)
        high = np.full(observation.shape, float("inf"), dtype=np.float32)
        return Box(low, high, dtype=observation.dtype)

    def close(self):
        self._vec.close()


def _rebuild_env(ctor):
    return PioneerKinematicEnv(**ctor)


class TimeLimit(compat.GymWrapper):
    """gym.wrappers.TimeLimit semantics (pioneer_knm_train.py:27); a ``gym.Wrapper`` where gym is importable."""

    def __init__(self, env, max_episode_steps: int):
        if compat.HAVE_GYM:
            super().__init__(env)          # gym.Wrapper: env, spaces, reward_range, metadata
        else:
            self.env = env
            self.action_space = env.action_space
            self.observation_space = env.observation_space
            self.reward_range = env.reward_range
            self.metadata = env.metadata
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        assert self._elapsed_steps is not None, "Cannot call env.step() before calling reset()"
        observation, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info["TimeLimit.truncated"] = not done
            done = True
        return observation, reward, done, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)

    def __getattr__(self, name):
        if name.startswith("_"):           # as gym.Wrapper: private names are not forwarded (and `env` itself may not be set yet)
            raise AttributeError(name)
        return getattr(self.env, name)


def make_env(env_config: Dict) -> TimeLimit:
    """The ``prepare_env`` creator of pioneer_knm_train.py:20-27."""
    pioneer_config = PioneerKinematicConfig(
        award_potential_slope=float(env_config["award_potential_slope"]),
        award_done=float(env_config["award_done"]),
        penalty_step=float(env_config["penalty_step"]),
    )
    return TimeLimit(PioneerKinematicEnv(pioneer_config=pioneer_config), max_episode_steps=500)


def make_vector_env(env_config: Dict):
    """The same creator for the BATCHED engine: what ``register_env('Pioneer-v1', make_vector_env)`` hands RLlib as a
    ``ray.rllib.env.VectorEnv`` (pioneer_amd/rllib_env.py).  ``env_config`` carries the reference's three reward keys
    (pioneer_knm_train.py:20-26) plus ``num_envs`` (default 4 096), ``device``, ``seed``."""
    from .rllib_env import PioneerRLlibVectorEnv
    pioneer_config = PioneerKinematicConfig(
        award_potential_slope=float(env_config["award_potential_slope"]),
        award_done=float(env_config["award_done"]),
        penalty_step=float(env_config["penalty_step"]),
    )
    return PioneerRLlibVectorEnv(int(env_config.get("num_envs", 4096)), device=env_config.get("device"),
                                 seed=int(env_config.get("seed", 0)), pioneer_config=pioneer_config, max_episode_steps=500)
