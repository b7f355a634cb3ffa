"""Configuration dataclasses.

``PioneerKinematicConfig`` / ``SimulationConfig`` / ``RenderConfig`` keep the
reference's field names and defaults (pioneer/envs/pioneer/pioneer_knm_env.py:19-34,
pioneer/envs/bullet/bullet_env.py:18-62); ``EngineConfig`` holds the options that
only exist in this engine.
"""
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from . import _lib


@dataclass
class PioneerKinematicConfig:
    max_v_to_r: float = 2       # seconds^-1
    max_a_to_v: float = 10      # seconds^-1

    done_distance: float = 0.1

    award_max: float = 100.0
    award_done: float = 5.0
    award_potential_slope: float = 10.0
    penalty_step: float = 1 / 100

    target_lo: Tuple[float, float, float] = (15, -10, 2)
    target_hi: Tuple[float, float, float] = (25, 10, 6)
    target_radius: float = 0.2
    target_rgba: Tuple[float, float, float, float] = (1.0, 0.0, 0.0, 0.5)


@dataclass
class SimulationConfig:
    timestep: float = 1 / 240
    frame_skip: int = 10

    gravity: float = 0

    self_collision: bool = False
    collision_parent: bool = True

    @property
    def frames_per_second(self) -> int:
        return int(np.round(1 / (self.timestep * self.frame_skip)))


@dataclass
class RenderConfig:
    camera_target: Tuple[float, float, float] = (0, 0, 0)
    camera_distance: float = 100.0
    camera_yaw: float = 120.0
    camera_pitch: float = -30.0
    camera_roll: float = 0.0
    render_width: int = 1280
    render_height: int = 800
    projection_fov: float = 30
    projection_near: float = 0.1
    projection_far: float = 200.0


@dataclass(frozen=True)
class SceneBody:
    """A static body of the scene (mass 0, with a collision shape): what Scene.create_body_plane / create_body_box /
    create_body_sphere (bullet_scene.py:193-228) add to the world.  Dynamics mode only; every contact sample of the arm
    collides with it.  Build with scene_plane / scene_box / scene_sphere, whose arguments are the reference's."""
    shape: str                                              # "plane" | "box" | "sphere"
    position: Tuple[float, float, float]
    orientation: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 1.0)   # Bullet quaternion (x, y, z, w)
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)     # plane normal | box half extents | (radius, 0, 0)


def scene_plane(normal, position=(0.0, 0.0, 0.0), orientation=(0.0, 0.0, 0.0, 1.0)) -> SceneBody:
    """create_body_plane(name, mass=0, normal, position, orientation), bullet_scene.py:219-227."""
    return SceneBody("plane", tuple(map(float, position)), tuple(map(float, orientation)), tuple(map(float, normal)))


def scene_box(half_extents, position, orientation=(0.0, 0.0, 0.0, 1.0)) -> SceneBody:
    """create_body_box(name, collision=True, mass=0, half_extents, position, orientation, rgba), bullet_scene.py:206-217."""
    return SceneBody("box", tuple(map(float, position)), tuple(map(float, orientation)), tuple(map(float, half_extents)))


def scene_sphere(radius, position, orientation=(0.0, 0.0, 0.0, 1.0)) -> SceneBody:
    """create_body_sphere(name, collision=True, mass=0, radius, position, orientation, rgba), bullet_scene.py:193-204."""
    return SceneBody("sphere", tuple(map(float, position)), tuple(map(float, orientation)), (float(radius), 0.0, 0.0))


@dataclass
class EngineConfig:
    """Options without a reference counterpart."""
    max_episode_steps: int = 500        # gym TimeLimit of pioneer_knm_train.py:27; 0 = off
    auto_reset: bool = True             # re-draw done|truncated envs inside the step kernel
    obs_layout: str = "env_major"       # "env_major" [N,137] | "feature_major" [137,N]
    action_layout: str = "env_major"    # "env_major" [N,6]   | "feature_major" [6,N]
    mode: str = "kinematic"             # "kinematic" (reference semantics) | "dynamic"
    # dynamics mode
    pd_kp: float = 4000.0
    pd_kd: float = 400.0
    torque_limit: float = 0.0
    joint_damping: float = 0.0
    joint_friction: float = 0.0
    teleport: bool = False
    randomize: bool = False
    rand_mass: Tuple[float, float] = (0.5, 1.5)
    rand_friction: Tuple[float, float] = (0.0, 0.1)
    rand_damping: Tuple[float, float] = (0.0, 0.1)
    ground_z: float = float("nan")
    contact_kp: float = 2000.0
    contact_kd: float = 50.0
    # static box obstacle for the pointer sphere (the reference demo's obstacle:1, pioneer_knm_env.py:249-255);
    # half extents <= 0 disable it
    obstacle_position: Tuple[float, float, float] = (10.0, 5.0, 0.0)
    obstacle_half_extents: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    pointer_radius: float = 0.2
    # the rest of the reference's motor surface (Joint.control_position / control_velocity, bullet_scene.py:123-155)
    control_mode: str = "position"      # "position" (POSITION_CONTROL) | "velocity" (VELOCITY_CONTROL: tracks the commanded velocity)
    max_velocity: float = 0.0           # control_position's maxVelocity: cap on the velocity the motor asks for; <= 0 = none
    # contacts of the LINKS with the plane / box: sample spheres along capsules fitted to the URDF's visual boxes (the
    # reference URDF carries no <collision> elements, so without this only the pointer sphere collides)
    link_contacts: bool = False
    # the motor asks for the acceleration pd_kp (r - q) + pd_kd (v* - qd), scaled by each joint's articulated-body inertia at
    # the current pose: pd_kp = omega^2, pd_kd = 2 zeta omega for every joint and pose alike (e.g. 400 / 40), stable whenever
    # pd_kd * timestep < 2 on the light wrist as on the heavy shoulder
    pd_inertia_scaled: bool = False
    # further static bodies (up to 8): planes of any normal, oriented boxes, spheres — see SceneBody
    scene: Tuple[SceneBody, ...] = ()


_LAYOUTS = {"env_major": _lib.ENV_MAJOR, "feature_major": _lib.FEATURE_MAJOR}
_MODES = {"kinematic": _lib.MODE_KINEMATIC, "dynamic": _lib.MODE_DYNAMIC}
_CONTROLS = {"position": _lib.CONTROL_POSITION, "velocity": _lib.CONTROL_VELOCITY}


def to_c_config(pioneer: PioneerKinematicConfig, sim: SimulationConfig, engine: EngineConfig) -> _lib.PnrConfig:
    """Dataclasses -> the C struct, starting from pnr_config_default()."""
    lib = _lib.load_library()
    c = _lib.PnrConfig()
    _lib.check(lib.pnr_config_default(c))
    assert len(pioneer.target_lo) == 3      # pioneer_knm_env.py:84
    assert len(pioneer.target_hi) == 3      # pioneer_knm_env.py:85
    for name in ("max_v_to_r", "max_a_to_v", "done_distance", "award_max", "award_done",
                 "award_potential_slope", "penalty_step", "target_radius"):
        setattr(c, name, float(getattr(pioneer, name)))
    for k in range(3):
        c.target_lo[k] = float(pioneer.target_lo[k])
        c.target_hi[k] = float(pioneer.target_hi[k])
    c.timestep = float(sim.timestep)
    c.frame_skip = int(sim.frame_skip)
    c.gravity = float(sim.gravity)
    if engine.obs_layout not in _LAYOUTS or engine.action_layout not in _LAYOUTS:
        raise AssertionError(f"layouts must be one of {sorted(_LAYOUTS)}")
    if engine.mode not in _MODES:
        raise AssertionError(f"mode must be one of {sorted(_MODES)}")
    c.max_episode_steps = int(engine.max_episode_steps)
    c.auto_reset = int(bool(engine.auto_reset))
    c.obs_layout = _LAYOUTS[engine.obs_layout]
    c.action_layout = _LAYOUTS[engine.action_layout]
    c.mode = _MODES[engine.mode]
    c.pd_kp, c.pd_kd, c.torque_limit = float(engine.pd_kp), float(engine.pd_kd), float(engine.torque_limit)
    c.joint_damping, c.joint_friction = float(engine.joint_damping), float(engine.joint_friction)
    c.teleport, c.randomize = int(bool(engine.teleport)), int(bool(engine.randomize))
    c.rand_mass_lo, c.rand_mass_hi = map(float, engine.rand_mass)
    c.rand_friction_lo, c.rand_friction_hi = map(float, engine.rand_friction)
    c.rand_damping_lo, c.rand_damping_hi = map(float, engine.rand_damping)
    c.ground_z = float(engine.ground_z)
    c.contact_kp, c.contact_kd = float(engine.contact_kp), float(engine.contact_kd)
    for k in range(3):
        c.obstacle_position[k] = float(engine.obstacle_position[k])
        c.obstacle_half_extents[k] = float(engine.obstacle_half_extents[k])
    c.pointer_radius = float(engine.pointer_radius)
    if engine.control_mode not in _CONTROLS:
        raise AssertionError(f"control_mode must be one of {sorted(_CONTROLS)}")
    c.control_mode = _CONTROLS[engine.control_mode]
    c.max_velocity = float(engine.max_velocity)
    c.link_contacts = int(bool(engine.link_contacts))
    c.pd_inertia_scaled = int(bool(engine.pd_inertia_scaled))
    shapes = {"plane": _lib.SHAPE_PLANE, "box": _lib.SHAPE_BOX, "sphere": _lib.SHAPE_SPHERE}
    if len(engine.scene) > _lib.MAX_SCENE:
        raise AssertionError(f"at most {_lib.MAX_SCENE} scene bodies")
    c.n_scene = len(engine.scene)
    for i, b in enumerate(engine.scene):
        if b.shape not in shapes:
            raise AssertionError(f"scene body {i}: shape must be one of {sorted(shapes)}")
        c.scene[i].shape = shapes[b.shape]
        for k in range(3):
            c.scene[i].position[k] = float(b.position[k]); c.scene[i].size[k] = float(b.size[k])
        for k in range(4):
            c.scene[i].orientation[k] = float(b.orientation[k])
    # SimulationConfig.self_collision / collision_parent (bullet_env.py:43-58) select pybullet's URDF_USE_SELF_COLLISION load
    # flags.  The reference URDF has no <collision> elements, so in the reference they change nothing — and here they
    # are accepted with that same meaning.  What is NOT modelled is self-collision between the build-defined link
    # capsules: asking for both is refused rather than silently ignored.
    if sim.self_collision and engine.link_contacts:
        raise AssertionError("self_collision between the link capsules (EngineConfig.link_contacts) is not modelled; the "
                             "reference's own URDF has no collision shapes, so self_collision alone is a no-op there and here")
    return c
