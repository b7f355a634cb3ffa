"""The batched engine behind RLlib's ``VectorEnv`` contract (lists of per-env values, as ray 0.8.x's rollout worker consumes them).

The reference hands RLlib ONE ``gym.Env`` per rollout worker through ``register_env`` (pioneer/launch/pioneer_knm_train.py:20-29)
and RLlib wraps it into a ``VectorEnv`` itself (``num_envs_per_worker`` copies stepped in a Python loop).  A creator may
also return a ``VectorEnv`` directly — that is the drop-in for the batched engine: ``PioneerRLlibVectorEnv`` derives from
``ray.rllib.env.VectorEnv`` where ray is importable (``object`` otherwise; nothing is installed for it) and follows its contract:

    vector_reset() -> [obs]            reset_at(i) -> obs            get_unwrapped() -> [env]
    vector_step(actions) -> ([obs], [reward], [done], [info])

with the single-env semantics of the reference: float64 ``obs[137]`` rows (quirk Q6, pioneer_knm_env.py:194-211, :242), Python
``float`` rewards, ``done = done or TimeLimit cut`` with ``info['TimeLimit.truncated']`` exactly as ``gym.wrappers.TimeLimit`` sets it
(pioneer_knm_train.py:27), NO auto-reset (RLlib calls ``reset_at`` for the envs it saw finish).  The lists are made from ONE
device-to-host copy per step; the device-resident fast path (tensors, in-kernel auto-reset) stays ``PioneerVectorEnv``.
"""
import dataclasses
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import compat
from .config import EngineConfig, PioneerKinematicConfig, SimulationConfig
from .spaces import Box
from .vector_env import PioneerVectorEnv


class PioneerRLlibVectorEnv(compat.RLlibVectorEnv):
    def __init__(self, num_envs: int, device=None, seed: int = 0, env_id_offset: int = 0,
                 pioneer_config: Optional[PioneerKinematicConfig] = None,
                 simulation_config: Optional[SimulationConfig] = None,
                 engine_config: Optional[EngineConfig] = None,
                 max_episode_steps: int = 500, info: str = "none", batch_resets: bool = True):
        """``info``: "none" (only ``TimeLimit.truncated`` where gym sets it), "numeric" (+ r_pot, r_step, r_done, dist as floats) or
        "strings" (the reference's nine formatted entries, pioneer_knm_env.py:167-179; slow: Python formatting per env).
        ``batch_resets``: the first ``reset_at`` after a step resets ALL envs that finished in that step with one masked launch
        and serves the following ``reset_at`` calls from it (RLlib asks for every finished env anyway)."""
        assert info in ("none", "numeric", "strings")
        eng = engine_config or EngineConfig()
        # the adaptor owns the episode protocol: TimeLimit counted in-kernel, resets only when asked for, env-major rows
        eng = dataclasses.replace(eng, max_episode_steps=int(max_episode_steps), auto_reset=False,
                                  obs_layout="env_major", action_layout="env_major")
        self.vec = PioneerVectorEnv(num_envs, device=device, seed=seed, env_id_offset=env_id_offset, pioneer_config=pioneer_config,
                                    simulation_config=simulation_config, engine_config=eng)
        self._info_mode, self._batch_resets = info, batch_resets
        self._pending: Dict[int, np.ndarray] = {}          # env index -> obs row of a reset already done (batch_resets)
        self._finished = np.zeros(num_envs, dtype=bool)    # envs whose last step returned done
        self._elapsed = np.zeros(num_envs, dtype=np.int64) # steps since each env's reset (gym.wrappers.TimeLimit._elapsed_steps)
        observation_space = compat.to_gym_space(Box(-np.inf, np.inf, shape=(self.vec.obs_dim,), dtype=np.float64))   # Q6: float64
        action_space = compat.to_gym_space(Box(-self.vec.a_max, self.vec.a_max, dtype=np.float32))
        if compat.HAVE_RLLIB:
            try:
                super().__init__(observation_space, action_space, int(num_envs))      # ray >= 1.0
            except TypeError:
                super().__init__()                                                    # ray 0.8.x: no constructor arguments
        self.observation_space, self.action_space, self.num_envs = observation_space, action_space, int(num_envs)

    # -- VectorEnv contract ----------------------------------------------------------------------------------------------
    def vector_reset(self) -> List[np.ndarray]:
        obs = self.vec.reset().double().cpu().numpy()
        self._pending.clear()
        self._finished[:] = False
        self._elapsed[:] = 0
        return list(obs)

    def reset_at(self, index: Optional[int] = None) -> np.ndarray:
        index = 0 if index is None else int(index)
        if not 0 <= index < self.num_envs:
            raise AssertionError(f"env index {index} out of range [0, {self.num_envs})")
        row = self._pending.pop(index, None)
        if row is not None:
            return row
        which = np.zeros(self.num_envs, dtype=np.uint8)
        which[index] = 1
        if self._batch_resets and self._finished[index]:
            which |= self._finished.astype(np.uint8)       # every env that finished in the last step, one launch
        obs = self.vec.reset(mask=torch.from_numpy(which)).double().cpu().numpy()
        for j in np.nonzero(which)[0]:
            if j != index:
                self._pending[int(j)] = obs[j]
        self._finished[which.astype(bool)] = False
        self._elapsed[which.astype(bool)] = 0
        return obs[index]

    def vector_step(self, actions) -> Tuple[List[np.ndarray], List[float], List[bool], List[Dict]]:
        act = np.asarray(actions, dtype=np.float32)
        if act.shape != (self.num_envs, self.vec.dof):
            raise AssertionError(f"actions must be {self.num_envs} rows of {self.vec.dof}, got shape {act.shape}")
        want = self._info_mode != "none"
        res = self.vec.vector_step(torch.from_numpy(act), want_info=want)
        obs = res[0].double().cpu().numpy()
        rew = res[1].cpu().numpy().astype(np.float64)
        term = res[2].cpu().numpy().astype(bool)
        trunc = res[3].cpu().numpy().astype(bool)
        done = term | trunc                                   # gym.wrappers.TimeLimit: done = True at the cut
        self._elapsed += 1
        self._finished = done.copy()
        self._pending.clear()                                 # rows of a reset nobody asked for are stale now
        infos: List[Dict] = [{} for _ in range(self.num_envs)]
        if want:
            inf = res[4].cpu().numpy()
            st = self.vec.state_dict() if self._info_mode == "strings" else None
            for i in range(self.num_envs):
                r_pot, r_step, r_done, dist = (float(x) for x in inf[i])
                if st is None:
                    infos[i] = {"r_pot": r_pot, "r_step": r_step, "r_done": r_done, "rw": float(rew[i]), "dist": dist}
                else:
                    from .env import arr2str
                    infos[i] = {"r_pot": f"{r_pot:.3f}", "r_step": f"{r_step:.3f}", "r_done": f"{r_done:.3f}", "rw": f"{float(rew[i]):.3f}",
                                "dist": f"{dist:.3f}", "pot": f"{float(st['potential'][i]):.3f}",
                                "a": arr2str(st["a"][i]), "v": arr2str(st["v"][i]), "r": arr2str(st["r"][i])}
        # gym.wrappers.TimeLimit writes its key on every step at or past the limit: `not done` of the wrapped env
        max_steps = self.vec.engine_config.max_episode_steps
        if max_steps > 0:
            for i in np.nonzero(self._elapsed >= max_steps)[0]:
                infos[int(i)]["TimeLimit.truncated"] = bool(not term[i])
        return list(obs), rew.tolist(), done.tolist(), infos          # (tolist: Python floats / bools, as RLlib's sampler expects)

    def get_unwrapped(self) -> List:
        """RLlib asks for the underlying gym envs (for rendering / custom callbacks); a batched engine has none."""
        return []

    # (newer RLlib spellings of the same two methods)
    def get_sub_environments(self) -> List:
        return []

    def try_render_at(self, index: Optional[int] = None):
        return None

    def seed(self, seed=None):
        return self.vec.seed(seed)

    def close(self):
        self.vec.close()


def as_rllib_vector_env(num_envs: int, **kw) -> PioneerRLlibVectorEnv:
    return PioneerRLlibVectorEnv(num_envs, **kw)
