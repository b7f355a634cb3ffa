"""Python mirror of the engine's model table (csrc/pnr_model.h) + a URDF emitter.

The table is the engine's own compact description of the Pioneer 6-DoF arm (numbers from the
reference's assets/pioneer_knm_6dof.urdf:27-275).  ``to_urdf()`` writes a
kinematically and inertially equivalent URDF (visuals omitted) for the optional PyBullet replay
(tools/pybullet_replay.py) — the reference tree never travels to the GPU box.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple


@dataclass(frozen=True)
class JointDef:
    name: str
    type: str                                # "revolute" | "fixed"
    parent: str
    child: str
    xyz: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    axis: Optional[Tuple[float, float, float]] = None
    limit: Optional[float] = None            # symmetric +-limit, rad


LINKS: List[str] = ["world", "robot:base", "robot:rotator1", "robot:hinge1", "robot:arm1", "robot:arm2",
                    "robot:rotator2", "robot:hinge2", "robot:arm3", "robot:rotator3", "robot:effector",
                    "robot:pointer"]

JOINTS: List[JointDef] = [
    JointDef("world_to_base", "fixed", "world", "robot:base"),
    JointDef("robot:base_to_rotator1", "revolute", "robot:base", "robot:rotator1", (0, 0, 0), (0, 0, 1), 3.1416),
    JointDef("robot:rotator1_to_hinge1", "fixed", "robot:rotator1", "robot:hinge1"),
    JointDef("robot:hinge1_to_arm1", "revolute", "robot:hinge1", "robot:arm1", (0, 0, 3), (0, 1, 0), 1.309),
    JointDef("robot:arm1_to_arm2", "revolute", "robot:arm1", "robot:arm2", (0, 0, 11), (0, 1, 0), 1.309),
    JointDef("robot:arm2_to_rotator2", "revolute", "robot:arm2", "robot:rotator2", (0, 1, 0), (1, 0, 0), 3.1416),
    JointDef("robot:rotator2_to_hinge2", "fixed", "robot:rotator2", "robot:hinge2"),
    JointDef("robot:hinge2_to_arm3", "revolute", "robot:hinge2", "robot:arm3", (11, 0, 0), (0, 1, 0), 1.5708),
    JointDef("robot:arm3_to_rotator3", "revolute", "robot:arm3", "robot:rotator3", (0, 0, 0), (1, 0, 0), 3.1416),
    JointDef("robot:rotator3_to_effector", "fixed", "robot:rotator3", "robot:effector"),
    JointDef("robot:effector_to_pointer", "fixed", "robot:effector", "robot:pointer", (3.6, 0, 1.9)),
]

LINK_MASS = 1.0        # every non-world link
LINK_INERTIA = 1.0     # diag(1, 1, 1) about the link frame origin
EFFORT = 1.0


def revolute_joints() -> List[JointDef]:
    return [j for j in JOINTS if j.type == "revolute"]


def _fmt(v) -> str:
    return " ".join(f"{float(x):g}" for x in v)


def to_urdf(robot_name: str = "pioneer") -> str:
    out = ['<?xml version="1.0"?>', f'<robot name="{robot_name}">']
    for link in LINKS:
        if link == "world":
            out.append('  <link name="world"/>')
            continue
        out += [f'  <link name="{link}">', "    <inertial>", f'      <mass value="{LINK_MASS:g}"/>',
                f'      <inertia ixx="{LINK_INERTIA:g}" ixy="0" ixz="0" iyy="{LINK_INERTIA:g}" iyz="0" izz="{LINK_INERTIA:g}"/>',
                "    </inertial>", "  </link>"]
    for j in JOINTS:
        out.append(f'  <joint name="{j.name}" type="{j.type}">')
        out.append(f'    <parent link="{j.parent}"/>')
        out.append(f'    <child link="{j.child}"/>')
        if j.type == "revolute":
            out.append(f'    <limit lower="{-j.limit:g}" upper="{j.limit:g}" effort="{EFFORT:g}"/>')
        if any(j.xyz):
            out.append(f'    <origin xyz="{_fmt(j.xyz)}"/>')
        if j.axis is not None:
            out.append(f'    <axis xyz="{_fmt(j.axis)}"/>')
        out.append("  </joint>")
    out.append("</robot>")
    return "\n".join(out) + "\n"
