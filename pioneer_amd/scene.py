"""``env.scene`` / ``env.world``: the object surface the reference's demo drives (pioneer_knm_env.py:245-296 ``__main__``:
``env.scene.create_body_box / create_body_plane``, ``env.scene.rpy2quat``, ``env.scene.joints_by_name[...]`` with ``position()``,
``lower_limit / upper_limit``, ``reset_state(position, velocity)``, ``control_position / control_velocity``; ``env.world.step()``,
``env.world.step_time``) over the HIP engine.

The reference's ``Scene`` / ``Joint`` / ``World`` (bullet_scene.py:70-279) are views of a PyBullet client; here they are views of the
one-env engine batch of ``PioneerKinematicEnv``.  What maps and how:

* joints: the six revolute joints of the URDF, by the URDF's names (bullet_env.py:141-146 keeps exactly these).  The Bullet client's
  joint state is NOT the env's ``self.r / self.v`` (pioneer_knm_env.py:144-146): the env teleports its r into Bullet with velocity 0
  on every step (:148, bullet_scene.py:157-165) and nothing flows back.  So ``position() / velocity() / reset_state`` work on the
  SIMULATOR's joints and never on the env's command state:
  dynamics mode — the engine's simulated q, q̇ (``pnr_get_dyn_state``);
  kinematic mode — a [1, 12] device buffer of this Scene (q | q̇), set to (r, 0) by every ``env.step`` / ``reset_world`` — what act()
  leaves in Bullet — and advanced by ``world.step()``.  ``env.step(a)`` followed by ``env.world.step()`` therefore changes nothing.
* ``control_position / control_velocity`` (bullet_scene.py:123-155): dynamics mode: the joint's motor for ``world.step()``
  (``pnr_set_joint_motor``: the engine's one motor law with this joint's targets, gains, force and maxVelocity; an argument left
  ``None`` takes the EngineConfig's value).  Kinematic mode has no motors (the engine says so).
* bodies: ``create_body_box / plane / sphere`` with ``mass == 0``.  The reference's arm has no ``<collision>`` shapes, so there a
  created body never touches it: in kinematic mode a body is a record (``items_by_name``), as inert as in the reference.  In dynamics
  mode a body with a collision shape becomes a static scene body of the engine (``EngineConfig.scene``; the handle is rebuilt with the
  env's state and the joints' motors carried over).  ``mass > 0`` asserts: moving bodies are not modelled (the reference never creates one).
* ``world.step()``: ``frame_skip`` × ``stepSimulation`` = ONE engine launch (``pnr_world_step``), nothing on the host.  Dynamics
  mode: the articulated-body sub-steps (gravity, contacts with the scene's bodies, limits, every joint's motor) and nothing else —
  no command integration, reward or observation.  Kinematic mode: each joint carries on at the velocity it was reset with
  (q += q̇ × step_time, stopped at its limit): what Bullet does without gravity, motor torque and collision shapes.
Build-defined behaviour where the reference delegates to Bullet: parity unpinned, like the rest of the Bullet boundary.
"""
import dataclasses
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib, model
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
        return float(self._env.scene._bullet[0, self.index])

    def velocity(self) -> float:
        if self._dynamic():
            return float(self._env._vec.get_dyn_state()[6 + self.index, 0])
        return float(self._env.scene._bullet[0, 6 + self.index])

    def reset_state(self, position: float, velocity: Optional[float] = None):
        """resetJointState (bullet_scene.py:157-165): the SIMULATOR's joint is put at ``position`` (and ``velocity``, else 0); the
        env's own r / v are not touched (the reference's are not either)."""
        v = 0.0 if velocity is None else float(velocity)
        if self._dynamic():
            vec = self._env._vec
            d = vec.get_dyn_state()
            d[self.index, 0] = float(position)
            d[6 + self.index, 0] = v
            vec.set_dyn_state(d)
        else:
            b = self._env.scene._bullet
            b[0, self.index] = float(position)
            b[0, 6 + self.index] = v

    def _set_motor(self, mode, position, velocity, position_gain, velocity_gain, max_force, max_velocity):
        nan = float("nan")
        args = (self.index, mode, nan if position is None else float(position), nan if velocity is None else float(velocity),
                nan if position_gain is None else float(position_gain), nan if velocity_gain is None else float(velocity_gain),
                nan if max_force is None else float(max_force), nan if max_velocity is None else float(max_velocity))
        self._env._vec.set_joint_motor(*args)
        self._env._motor_cmds[self.index] = args                  # re-applied when the engine handle is rebuilt (a body created later)

    def control_position(self, position: float, velocity: Optional[float] = None, max_velocity: Optional[float] = None,
                         max_force: Optional[float] = None, position_gain: Optional[float] = None, velocity_gain: Optional[float] = None):
        """setJointMotorControl2(POSITION_CONTROL, ...) (bullet_scene.py:123-142) for ``world.step()``."""
        self._set_motor(_lib.CONTROL_POSITION, position, velocity, position_gain, velocity_gain, max_force, max_velocity)

    def control_velocity(self, velocity: float, max_force: Optional[float] = None):
        """setJointMotorControl2(VELOCITY_CONTROL, ...) (bullet_scene.py:144-155) for ``world.step()``."""
        self._set_motor(_lib.CONTROL_VELOCITY, None, velocity, None, None, max_force, None)


class Scene:
    def __init__(self, env):
        self._env = env
        self.items: List[Item] = []
        self.items_by_name: Dict[str, Item] = {}
        self.joints: List[Joint] = [Joint(env, i, jd) for i, jd in enumerate(model.revolute_joints())]
        self.joints_by_name: Dict[str, Joint] = {j.name: j for j in self.joints}
        # kinematic mode: the simulator's joints (q | qd) as act() leaves them in Bullet; see the module docstring
        self._bullet = torch.zeros((1, 12), dtype=torch.float32, device=env._vec.device)

    def sync_from_env(self):
        """What act() does to the Bullet client on every step (pioneer_knm_env.py:148): joints at the env's r, velocity 0."""
        if self._env._vec.engine_config.mode != "dynamic":
            self._bullet[0, 0:6] = self._env._vec.get_state().view(torch.float32)[12:18, 0]
            self._bullet[0, 6:12] = 0.0

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
        """frame_skip x stepSimulation: ONE launch of the engine (pnr_world_step), no host arithmetic."""
        vec = self._env._vec
        vec.world_step(None if vec.engine_config.mode == "dynamic" else self._env.scene._bullet)
