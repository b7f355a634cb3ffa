"""``env.scene`` / ``env.world``: the object surface the reference's demo drives (pioneer_knm_env.py:245-296 ``__main__``:
``env.scene.create_body_box / create_body_plane``, ``env.scene.rpy2quat``, ``env.scene.joints_by_name[...]`` with ``position()``,
``lower_limit / upper_limit``, ``reset_state(position, velocity)``; ``env.world.step()``, ``env.world.step_time``) over the HIP engine.

The reference's ``Scene`` / ``Joint`` / ``World`` (bullet_scene.py:70-279) are views of a PyBullet client; here they are views of the
one-env engine batch of ``PioneerKinematicEnv``.  What maps and what does not:

* joints: the six revolute joints of the URDF, by the URDF's names (bullet_env.py:141-146 keeps exactly these); ``position() /
  velocity()`` read the engine's joint state (kinematic mode: r, v; dynamics mode: the simulated q, q̇), ``reset_state`` writes it —
  ``resetJointState``, bullet_scene.py:157-165.  ``control_position / control_velocity`` raise: the engine's motor law is one
  configuration for all joints (``EngineConfig.control_mode``), not a per-joint command.
* bodies: ``create_body_box / plane / sphere`` with ``mass == 0``.  The reference's arm has no ``<collision>`` shapes, so there a
  created body never touches it: in kinematic mode a body is a record (``items_by_name``), as inert as in the reference.  In dynamics
  mode a body with a collision shape becomes a static scene body of the engine (``EngineConfig.scene``; the handle is rebuilt with the
  env's state carried over).  ``mass > 0`` asserts: moving bodies are not modelled (the reference never creates one).
* ``world.step()``: ``frame_skip`` × ``stepSimulation``.  Under the env's defaults that is the identity on everything the env
  observes — except for a joint whose state was reset WITH a velocity, which Bullet then carries on at that velocity:
  kinematic mode advances such joints by ``velocity × step_time`` (clamped at the limits, velocity zeroed there) and nothing else.
  In dynamics mode the sub-steps belong to ``env.step(action)``; ``world.step()`` raises.
Build-defined behaviour where the reference delegates to Bullet: parity unpinned, like the rest of the Bullet boundary.
"""
import dataclasses
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import model
from .config import scene_box, scene_plane, scene_sphere


class Item:
    """bullet_scene.py:40-67's Item, for the bodies created through Scene (and the target marker)."""

    def __init__(self, name: Optional[str], shape: str, position, orientation, collision: bool, size):
        self.name, self.shape, self.collision, self.size = name, shape, bool(collision), tuple(size)
        self._position, self._orientation = tuple(map(float, position)), tuple(map(float, orientation))

    def pose(self):
        return self._position, self._orientation

    def __repr__(self) -> str:
        return f"Item(name={self.name}, shape={self.shape}, position={self._position}, collision={self.collision})"


class Joint:
    def __init__(self, env, index: int, jd: model.JointDef):
        self._env, self.index, self.name = env, index, jd.name
        self.joint_type = jd.type
        lo, hi = env.joint_limits()
        self.lower_limit, self.upper_limit = float(lo[index]), float(hi[index])      # the float32 limits the env uses (:56)
        self.max_force = model.EFFORT
        self.damping = self.friction = 0.0

    def __repr__(self) -> str:
        return f"Joint(name={self.name}, index={self.index}, lower_limit={self.lower_limit}, upper_limit={self.upper_limit})"

    def _dynamic(self) -> bool:
        return self._env._vec.engine_config.mode == "dynamic"

    def position(self) -> float:
        if self._dynamic():
            return float(self._env._vec.get_dyn_state()[self.index, 0])
        return float(self._env._vec.state_dict()["r"][0, self.index])

    def velocity(self) -> float:
        if self._dynamic():
            return float(self._env._vec.get_dyn_state()[6 + self.index, 0])
        return float(self._env._vec.state_dict()["v"][0, self.index])

    def reset_state(self, position: float, velocity: Optional[float] = None):
        """resetJointState (bullet_scene.py:157-165): the joint is put at ``position`` (and ``velocity``, else 0)."""
        vec = self._env._vec
        v = 0.0 if velocity is None else float(velocity)
        w = vec.get_state()
        f = w.view(torch.float32)
        f[12 + self.index, 0] = float(position)               # planar state words: a 0..5, v 6..11, r 12..17
        f[6 + self.index, 0] = v
        vec.set_state(w)
        if self._dynamic():
            d = vec.get_dyn_state()
            d[self.index, 0] = float(position)
            d[6 + self.index, 0] = v
            vec.set_dyn_state(d)

    def control_position(self, *args, **kwargs):
        raise NotImplementedError("per-joint motor commands are not part of the engine: its motor law is configured per env "
                                  "(EngineConfig.control_mode, pd_kp, pd_kd, max_velocity); commands go through env.step(action)")

    control_velocity = control_position


class Scene:
    def __init__(self, env):
        self._env = env
        self.items: List[Item] = []
        self.items_by_name: Dict[str, Item] = {}
        self.joints: List[Joint] = [Joint(env, i, jd) for i, jd in enumerate(model.revolute_joints())]
        self.joints_by_name: Dict[str, Joint] = {j.name: j for j in self.joints}

    # -- items -----------------------------------------------------------------------------------------------------------
    def add_item(self, item: Item):
        self.items.append(item)
        if item.name is not None:
            assert item.name not in self.items_by_name                        # bullet_scene.py:181-183
            self.items_by_name[item.name] = item

    def _create(self, item: Item, mass: float, body):
        assert float(mass) == 0.0, "only static bodies (mass 0) are modelled: the reference never creates a moving one"
        self.add_item(item)
        vec = self._env._vec
        if item.collision and vec.engine_config.mode == "dynamic":
            self._env._rebuild_engine(dataclasses.replace(vec.engine_config, scene=tuple(vec.engine_config.scene) + (body,)))

    def create_body_sphere(self, name, collision, mass, radius, position, orientation, rgba_color=None):   # bullet_scene.py:193-204
        self._create(Item(name, "sphere", position, orientation, collision, (radius, 0.0, 0.0)), mass, scene_sphere(radius, position, orientation))

    def create_body_box(self, name, collision, mass, half_extents, position, orientation, rgba_color=None):  # :206-217
        self._create(Item(name, "box", position, orientation, collision, half_extents), mass, scene_box(half_extents, position, orientation))

    def create_body_plane(self, name, mass, normal, position, orientation):                                   # :219-227
        self._create(Item(name, "plane", position, orientation, True, normal), mass, scene_plane(normal, position, orientation))

    # -- rotations (pybullet.getQuaternionFromEuler / getEulerFromQuaternion: x, y, z, w; roll about x, pitch about y, yaw about z) --
    @staticmethod
    def rpy2quat(rpy: Tuple[float, float, float]):
        r, p, y = (0.5 * float(a) for a in rpy)
        cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
        return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)

    @staticmethod
    def quat2rpy(quat: Tuple[float, float, float, float]):
        x, y, z, w = (float(a) for a in quat)
        roll = math.atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
        pitch = math.asin(max(-1.0, min(1.0, 2.0 * (w * y - z * x))))
        yaw = math.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
        return (roll, pitch, yaw)

    def __repr__(self) -> str:
        items_str = "\n".join(f"\t\t{x}" for x in self.items)
        joints_str = "\n".join(f"\t\t{x}" for x in self.joints)
        return f"Scene(\n\titems: \n{items_str} \n\tjoints: \n{joints_str} \n)"


class World:
    """bullet_scene.py:262-279."""

    def __init__(self, env):
        self._env = env
        sc = env.simulation_config
        self.timestep, self.frame_skip, self.gravity_force = sc.timestep, sc.frame_skip, sc.gravity

    @property
    def step_time(self) -> float:
        return self.timestep * self.frame_skip

    def step(self):
        vec = self._env._vec
        if vec.engine_config.mode == "dynamic":
            raise NotImplementedError("dynamics mode: the sub-steps run inside env.step(action) (one launch: command integration, "
                                      "frame_skip sub-steps, reward, observation)")
        w = vec.get_state()
        f = w.view(torch.float32)
        v = f[6:12, 0].cpu().numpy().astype(np.float64)
        if not v.any():
            return                                                            # the identity of SURVEY 8 a6
        r = f[12:18, 0].cpu().numpy().astype(np.float64) + v * self.step_time
        lo, hi = (x.astype(np.float64) for x in self._env.joint_limits())
        hit = (r >= hi) | (r <= lo)
        r = np.clip(r, lo, hi)
        v = np.where(hit, 0.0, v)
        f[12:18, 0] = torch.from_numpy(r.astype(np.float32)).to(f.device)
        f[6:12, 0] = torch.from_numpy(v.astype(np.float32)).to(f.device)
        vec.set_state(w)
