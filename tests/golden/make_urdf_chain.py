#!/usr/bin/env python3
"""Extract the kinematic chain DATA of the reference's URDF into urdf_chain.json.

Run in the build container (the reference tree is not on the GPU box):
    python tests/golden/make_urdf_chain.py [/root/reference]
Only numbers and names are copied (joint order, types, parents, origins, axes,
limits, link masses / inertias, collision count); no reference source text.
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
URDF = os.path.join(REF, "pioneer/envs/pioneer/assets/pioneer_knm_6dof.urdf")


def floats(text, default):
    return [float(x) for x in text.split()] if text else list(default)


def main():
    root = ET.parse(URDF).getroot()
    links = {}
    for link in root.findall("link"):
        inertial = link.find("inertial")
        rec = {"has_inertial": inertial is not None, "n_collision": len(link.findall("collision")),
               "n_visual": len(link.findall("visual"))}
        if inertial is not None:
            rec["mass"] = float(inertial.find("mass").get("value"))
            i = inertial.find("inertia")
            rec["inertia"] = {k: float(i.get(k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")}
            o = inertial.find("origin")
            rec["inertial_xyz"] = floats(o.get("xyz") if o is not None else None, (0, 0, 0))
        links[link.get("name")] = rec
    joints = []
    for j in root.findall("joint"):
        o = j.find("origin")
        ax = j.find("axis")
        lim = j.find("limit")
        dyn = j.find("dynamics")
        joints.append({
            "name": j.get("name"), "type": j.get("type"),
            "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
            "xyz": floats(o.get("xyz") if o is not None else None, (0, 0, 0)),
            "rpy": floats(o.get("rpy") if o is not None else None, (0, 0, 0)),
            "axis": floats(ax.get("xyz"), ()) if ax is not None else None,
            "lower": float(lim.get("lower")) if lim is not None and lim.get("lower") else None,
            "upper": float(lim.get("upper")) if lim is not None and lim.get("upper") else None,
            "effort": float(lim.get("effort")) if lim is not None and lim.get("effort") else None,
            "has_dynamics": dyn is not None,
        })
    out = {"source": "xdralex/pioneer pioneer/envs/pioneer/assets/pioneer_knm_6dof.urdf (numbers only)",
           "robot": root.get("name"), "links": links, "joints": joints}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "urdf_chain.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {dst}: {len(links)} links, {len(joints)} joints")


if __name__ == "__main__":
    main()
