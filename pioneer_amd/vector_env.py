"""Batched, device-resident Pioneer-arm env (the RLlib ``VectorEnv`` shape with tensors).

Mirrors, for N envs at once, ``PioneerKinematicEnv`` of the reference
(pioneer/envs/pioneer/pioneer_knm_env.py:37-242): same constants, same
``reset_world`` override arguments, same step outputs; lists become torch
tensors that never leave the GPU.  All work happens in libpioneer_amd.so
(HIP); there is no Python or CPU compute path here.
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .config import EngineConfig, PioneerKinematicConfig, SimulationConfig, to_c_config
from .spaces import Box


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _rebuild_vector_env(ctor):
    return PioneerVectorEnv(**ctor)


class PioneerVectorEnv:
    """N independent Pioneer arms stepped by one HIP kernel launch.

    obs tensors are ``[N, 137]`` (``obs_layout="env_major"``) or ``[137, N]``
    (``"feature_major"``); actions ``[N, 6]`` / ``[6, N]``; rewards float32 ``[N]``;
    dones / truncated uint8 ``[N]``.
    """

    def __init__(self, num_envs: int, device=None, seed: int = 0, env_id_offset: int = 0,
                 pioneer_config: Optional[PioneerKinematicConfig] = None,
                 simulation_config: Optional[SimulationConfig] = None,
                 engine_config: Optional[EngineConfig] = None):
        self._h = None
        self._ctor = dict(num_envs=num_envs, device=device, seed=seed, env_id_offset=env_id_offset,
                          pioneer_config=pioneer_config, simulation_config=simulation_config,
                          engine_config=engine_config)
        self.lib = _lib.load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("pioneer_amd needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU backend")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise AssertionError(f"device must be a HIP device, got {self.device}")
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)

        self.config = pioneer_config or PioneerKinematicConfig()
        self.simulation_config = simulation_config or SimulationConfig()
        self.engine_config = engine_config or EngineConfig()
        self.num_envs = int(num_envs)
        self.env_id_offset = int(env_id_offset)
        self._c_cfg = to_c_config(self.config, self.simulation_config, self.engine_config)

        k = _lib.PnrConstants()
        _lib.check(self.lib.pnr_get_constants(self._c_cfg, k))
        self.r_lo = np.array(k.r_lo[:], dtype=np.float32)    # pioneer_knm_env.py:56
        self.r_hi = np.array(k.r_hi[:], dtype=np.float32)
        self.v_max = np.array(k.v_max[:], dtype=np.float32)  # :57
        self.a_max = np.array(k.a_max[:], dtype=np.float32)  # :58
        self.dt = k.dt                                       # :60
        self.eps = k.eps                                     # :61
        self.dof = _lib.DOF
        self.obs_dim = _lib.OBS_DIM

        h = C.c_void_p()
        _lib.check(self.lib.pnr_create(self._c_cfg, self.num_envs, self.env_id_offset, dev_index,
                                       C.c_uint64(seed & (2**64 - 1)), C.byref(h)))
        self._h = h
        self._seed = seed

        self.feature_major_obs = self.engine_config.obs_layout == "feature_major"
        self.feature_major_act = self.engine_config.action_layout == "feature_major"
        self.obs_shape = (self.obs_dim, self.num_envs) if self.feature_major_obs else (self.num_envs, self.obs_dim)
        self.action_shape = (self.dof, self.num_envs) if self.feature_major_act else (self.num_envs, self.dof)

        # spaces of ONE env, as the reference defines them (:72-74)
        self.action_space = Box(-self.a_max, self.a_max, dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, shape=(self.obs_dim,), dtype=np.float32)
        self.reward_range = (-float("inf"), float("inf"))

    # -- plumbing -------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_handle(self):
        if self._h is None:
            raise RuntimeError("env is closed")

    def _chk(self, code):
        _lib.check(code, self._h)

    def _in(self, t, shape, dtype, name):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t), dtype=dtype)
        if t.device != self.device or t.dtype != dtype:
            t = t.to(device=self.device, dtype=dtype)
        if tuple(t.shape) != tuple(shape):
            raise AssertionError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
        return t.contiguous()

    def _new(self, shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    # -- gym-ish surface ----------------------------------------------------------------
    def seed(self, seed=None):
        """pioneer_knm_env.py:107-109 (takes effect at the next reset)."""
        self._check_handle()
        if seed is None:
            seed = int(np.random.SeedSequence().entropy & (2**63 - 1))
        self._seed = int(seed)
        self._chk(self.lib.pnr_seed(self._h, C.c_uint64(self._seed & (2**64 - 1))))
        return [self._seed]

    def reset(self, mask=None, joint_positions=None, target_positions=None, out=None):
        """BulletEnv.reset + reset_world (bullet_env.py:187-190, pioneer_knm_env.py:76-105).

        ``mask`` selects envs (uint8/bool ``[N]``; None = all); ``joint_positions``
        ``[N,6]`` / ``target_positions`` ``[N,3]`` override the random draws as the
        reference's arguments do.  Returns the obs batch (rows of unselected envs
        are only meaningful if ``out`` already held them).
        """
        self._check_handle()
        n = self.num_envs
        m = None if mask is None else self._in(torch.as_tensor(mask).to(torch.uint8), (n,), torch.uint8, "mask")
        jp = None if joint_positions is None else self._in(joint_positions, (n, 6), torch.float32, "joint_positions")
        tp = None if target_positions is None else self._in(target_positions, (n, 3), torch.float32, "target_positions")
        if out is None:
            obs = self._new(self.obs_shape)
            if m is not None:
                self._chk(self.lib.pnr_observe(self._h, _ptr(obs), self._stream()))
        else:
            obs = self._in(out, self.obs_shape, torch.float32, "out")
        self._chk(self.lib.pnr_reset(self._h, _ptr(m), _ptr(jp), _ptr(tp), _ptr(obs), self._stream()))
        return obs

    def vector_reset(self):
        return self.reset()

    def reset_at(self, index: int):
        """RLlib VectorEnv.reset_at: reset one env, return its obs row (env-major order)."""
        mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
        mask[index] = 1
        obs = self.reset(mask=mask)
        return obs[:, index] if self.feature_major_obs else obs[index]

    def vector_step(self, actions, out=None, want_info=False):
        """BulletEnv.step for every env (bullet_env.py:192-197).

        Returns ``(obs, rewards, dones, truncated)`` (+ ``info [N,4]`` =
        r_pot, r_step, r_done, dist when ``want_info``).  ``out`` may carry
        preallocated ``obs/reward/done/truncated/info`` tensors.
        """
        self._check_handle()
        n = self.num_envs
        act = self._in(actions, self.action_shape, torch.float32, "actions")
        out = out or {}
        obs = out.get("obs") if out.get("obs") is not None else self._new(self.obs_shape)
        rew = out.get("reward") if out.get("reward") is not None else self._new((n,))
        done = out.get("done") if out.get("done") is not None else self._new((n,), torch.uint8)
        trunc = out.get("truncated") if out.get("truncated") is not None else self._new((n,), torch.uint8)
        info = None
        if want_info:
            info = out.get("info") if out.get("info") is not None else self._new((n, _lib.INFO_DIM))
        self._chk(self.lib.pnr_step(self._h, _ptr(act), _ptr(obs), _ptr(rew), _ptr(done), _ptr(trunc),
                                    _ptr(info), self._stream()))
        if want_info:
            return obs, rew, done, trunc, info
        return obs, rew, done, trunc

    def rollout(self, actions, out=None):
        """T open-loop steps in one launch; ``actions`` is ``[T, *action_shape]``."""
        self._check_handle()
        n = self.num_envs
        T = int(actions.shape[0])
        act = self._in(actions, (T,) + tuple(self.action_shape), torch.float32, "actions")
        out = out or {}
        obs = out.get("obs") if out.get("obs") is not None else self._new((T,) + tuple(self.obs_shape))
        rew = out.get("reward") if out.get("reward") is not None else self._new((T, n))
        done = out.get("done") if out.get("done") is not None else self._new((T, n), torch.uint8)
        trunc = out.get("truncated") if out.get("truncated") is not None else self._new((T, n), torch.uint8)
        self._chk(self.lib.pnr_rollout(self._h, T, _ptr(act), _ptr(obs), _ptr(rew), _ptr(done), _ptr(trunc),
                                       self._stream()))
        return obs, rew, done, trunc

    def world_step(self, joint_state=None):
        """World.step() alone (bullet_scene.py:273-275; pnr_world_step): the simulator's sub-steps and nothing else.  Dynamics mode:
        ``joint_state`` is None (the handle's q, qd move); kinematic mode: the caller's float32 [N, 12] device buffer (q | qd)."""
        self._check_handle()
        js = None
        if joint_state is not None:
            js = self._in(joint_state, (self.num_envs, 12), torch.float32, "joint_state")
            assert js.data_ptr() == joint_state.data_ptr(), "joint_state is updated in place: it must already be a contiguous float32 device tensor"
        self._chk(self.lib.pnr_world_step(self._h, _ptr(js) if js is not None else None, self._stream()))

    def set_joint_motor(self, joint, control_mode, target_position=float("nan"), target_velocity=float("nan"), position_gain=float("nan"),
                        velocity_gain=float("nan"), max_force=float("nan"), max_velocity=float("nan")):
        """Joint.control_position / control_velocity (bullet_scene.py:123-155; pnr_set_joint_motor) for ``joint`` of every env; NaN = the
        EngineConfig's value.  Honoured by ``world_step``."""
        self._check_handle()
        self._chk(self.lib.pnr_set_joint_motor(self._h, int(joint), int(control_mode), float(target_position), float(target_velocity),
                                               float(position_gain), float(velocity_gain), float(max_force), float(max_velocity)))

    def observe(self, out=None):
        """observe() without stepping (pioneer_knm_env.py:184-211)."""
        self._check_handle()
        obs = self._new(self.obs_shape) if out is None else self._in(out, self.obs_shape, torch.float32, "out")
        self._chk(self.lib.pnr_observe(self._h, _ptr(obs), self._stream()))
        return obs

    # -- raw state ------------------------------------------------------------------------
    def get_state(self):
        """Planar state words uint32-as-int32 ``[24, N]`` (see include/pioneer_amd.h)."""
        self._check_handle()
        w = torch.empty((_lib.STATE_WORDS, self.num_envs), dtype=torch.int32, device=self.device)
        self._chk(self.lib.pnr_get_state(self._h, _ptr(w), self._stream()))
        return w

    def set_state(self, words):
        self._check_handle()
        w = self._in(words, (_lib.STATE_WORDS, self.num_envs), torch.int32, "words")
        self._chk(self.lib.pnr_set_state(self._h, _ptr(w), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()  # `w` may be a temporary

    def get_dyn_state(self):
        self._check_handle()
        w = torch.empty((_lib.DYN_STATE_WORDS, self.num_envs), dtype=torch.float32, device=self.device)
        self._chk(self.lib.pnr_get_dyn_state(self._h, _ptr(w), self._stream()))
        return w

    def set_dyn_state(self, words):
        self._check_handle()
        w = self._in(words, (_lib.DYN_STATE_WORDS, self.num_envs), torch.float32, "words")
        self._chk(self.lib.pnr_set_dyn_state(self._h, _ptr(w), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    def state_dict(self):
        """Decoded state (numpy, host): a, v, r [N,6]; target [N,3]; potential, step_index, episode [N]."""
        w = self.get_state().cpu().numpy().view(np.uint32)
        f = w.view(np.float32)
        return dict(a=f[0:6].T.copy(), v=f[6:12].T.copy(), r=f[12:18].T.copy(), target=f[18:21].T.copy(),
                    potential=f[21].copy(), step_index=w[22].copy(), episode=w[23].copy())

    def __reduce__(self):
        """Pickled by constructor arguments (a fresh batch, like the reference's EzPickle envs);
        use get_state()/set_state() to carry the simulation state."""
        return (_rebuild_vector_env, (self._ctor,))

    def get_unwrapped(self):
        return []

    def close(self):
        if getattr(self, "_h", None) is not None:
            torch.cuda.synchronize(self.device)
            self.lib.pnr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
