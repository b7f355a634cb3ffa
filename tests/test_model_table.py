"""The engine's Python model table and its URDF emitter against the reference's URDF numbers."""
import json
import os
import subprocess
import sys
import xml.etree.ElementTree as ET

from pioneer_amd import model

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "urdf_chain.json")))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_table_matches_reference_urdf_numbers():
    assert [j["name"] for j in GOLD["joints"]] == [j.name for j in model.JOINTS]
    for g, j in zip(GOLD["joints"], model.JOINTS):
        assert (g["type"], g["parent"], g["child"]) == (j.type, j.parent, j.child)
        assert tuple(map(float, g["xyz"])) == tuple(map(float, j.xyz))
        assert (g["axis"] is None) == (j.axis is None)
        if j.axis is not None:
            assert tuple(map(float, g["axis"])) == tuple(map(float, j.axis))
            assert g["upper"] == j.limit and g["lower"] == -j.limit and g["effort"] == model.EFFORT
    assert sorted(GOLD["links"]) == sorted(model.LINKS)
    for name, l in GOLD["links"].items():
        if name != "world":
            assert l["mass"] == model.LINK_MASS and l["inertia"]["ixx"] == model.LINK_INERTIA


def test_emitted_urdf_round_trips_to_the_same_chain(tmp_path):
    root = ET.fromstring(model.to_urdf())
    joints = root.findall("joint")
    assert len(joints) == 11 and len(root.findall("link")) == 12
    for el, g in zip(joints, GOLD["joints"]):
        assert el.get("name") == g["name"] and el.get("type") == g["type"]
        o = el.find("origin")
        xyz = [float(x) for x in o.get("xyz").split()] if o is not None else [0, 0, 0]
        assert xyz == [float(x) for x in g["xyz"]]
        if g["axis"]:
            assert [float(x) for x in el.find("axis").get("xyz").split()] == g["axis"]
            assert float(el.find("limit").get("upper")) == g["upper"]
    assert not root.findall(".//collision")


def test_c_model_table_agrees_with_python_mirror():
    """pnr_model.h's joint rows (axis, origin, limit) are the revolute rows of model.JOINTS."""
    import re
    src = open(os.path.join(ROOT, "pioneer_amd", "csrc", "pnr_model.h")).read()
    rows = re.findall(r"\{A([XYZ]), ([\d.]+), ([\d.]+), ([\d.]+), ([\d.]+)\}", src)
    assert len(rows) == 6
    ax = {"X": (1, 0, 0), "Y": (0, 1, 0), "Z": (0, 0, 1)}
    # fixed joints between revolute ones are identity except the tip, so origins carry over directly
    for row, j in zip(rows, model.revolute_joints()):
        assert ax[row[0]] == tuple(int(a) for a in j.axis)
        assert tuple(float(x) for x in row[1:4]) == tuple(float(x) for x in j.xyz)
        assert float(row[4]) == j.limit
    tip = re.search(r"kTipX = ([\d.]+), kTipY = ([\d.]+), kTipZ = ([\d.]+)", src)
    assert tuple(float(x) for x in tip.groups()) == tuple(float(x) for x in model.JOINTS[-1].xyz)


def test_pybullet_replay_reports_unavailable_or_runs():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pybullet_replay.py"), "--steps", "50",
                          "--warmup", "5"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["pybullet"] in ("ok", "unavailable on this host")
