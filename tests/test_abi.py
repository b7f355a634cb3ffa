"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports
every symbol include/pioneer_amd.h declares; defaults mirror the reference's dataclasses;
the product path fails loudly when no HIP device is usable (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pioneer_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pnr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_surface():
    names = declared_functions()
    for must in ("pnr_create", "pnr_destroy", "pnr_reset", "pnr_step", "pnr_rollout", "pnr_observe",
                 "pnr_get_state", "pnr_set_state", "pnr_last_error", "pnr_config_default", "pnr_get_constants"):
        assert must in names


def test_library_exports_every_declared_symbol(hip_lib):
    from pioneer_amd import _lib
    names = declared_functions()
    for n in names:
        assert hasattr(hip_lib, n), f"libpioneer_amd.so lacks {n}"
    assert sorted(_lib.SIGNATURES) == names, "python binding table and header disagree"


def test_no_torch_types_in_signatures():
    src = open(HEADER).read()
    assert "torch" not in src.lower().replace("no torch types", "") and "at::" not in src and "c10" not in src
    assert 'extern "C"' in src


def test_config_defaults_mirror_reference_dataclasses(hip_lib):
    from pioneer_amd import _lib, PioneerKinematicConfig, SimulationConfig
    c = _lib.PnrConfig()
    assert hip_lib.pnr_config_default(c) == 0
    assert c.struct_size == C.sizeof(_lib.PnrConfig)
    pk, sim = PioneerKinematicConfig(), SimulationConfig()
    # pioneer_knm_env.py:19-34
    assert (c.max_v_to_r, c.max_a_to_v, c.done_distance) == (2, 10, 0.1) == (pk.max_v_to_r, pk.max_a_to_v, pk.done_distance)
    assert (c.award_max, c.award_done, c.award_potential_slope, c.penalty_step) == (100.0, 5.0, 10.0, 1 / 100)
    assert (pk.award_max, pk.award_done, pk.award_potential_slope, pk.penalty_step) == (100.0, 5.0, 10.0, 1 / 100)
    assert tuple(c.target_lo) == (15, -10, 2) == tuple(pk.target_lo)
    assert tuple(c.target_hi) == (25, 10, 6) == tuple(pk.target_hi)
    assert c.target_radius == 0.2 == pk.target_radius and pk.target_rgba == (1.0, 0.0, 0.0, 0.5)
    # bullet_env.py:36-44
    assert (c.timestep, c.frame_skip, c.gravity) == (1 / 240, 10, 0) == (sim.timestep, sim.frame_skip, sim.gravity)
    assert sim.frames_per_second == 24 and not sim.self_collision and sim.collision_parent
    assert c.max_episode_steps == 500                      # pioneer_knm_train.py:27


def test_constants_match_oracle(hip_lib, oracle_built):
    from pioneer_amd import _lib
    from oracle import COracle
    c = _lib.PnrConfig(); k = _lib.PnrConstants()
    hip_lib.pnr_config_default(c)
    assert hip_lib.pnr_get_constants(c, k) == 0
    o = COracle(1)
    assert np.array_equal(np.array(k.r_lo[:], np.float32), o.r_lo)
    assert np.array_equal(np.array(k.r_hi[:], np.float32), o.r_hi)
    assert np.array_equal(np.array(k.v_max[:], np.float32), o.v_max)
    assert np.array_equal(np.array(k.a_max[:], np.float32), o.a_max)
    assert k.dt == o.dt and k.eps == o.p.eps
    c.max_v_to_r = 3.0; c.max_a_to_v = 7.0; c.frame_skip = 4
    hip_lib.pnr_get_constants(c, k)
    o2 = COracle(1, max_v_to_r=3.0, max_a_to_v=7.0, frame_skip=4)
    assert np.array_equal(np.array(k.v_max[:], np.float32), o2.v_max)
    assert np.array_equal(np.array(k.a_max[:], np.float32), o2.a_max) and k.dt == o2.dt


def test_bad_config_is_rejected(hip_lib):
    from pioneer_amd import _lib
    c = _lib.PnrConfig(); hip_lib.pnr_config_default(c)
    k = _lib.PnrConstants()
    c.struct_size = 12
    assert hip_lib.pnr_get_constants(c, k) == -1
    assert b"size/version" in hip_lib.pnr_last_error(None)
    hip_lib.pnr_config_default(c); c.obs_layout = 7
    assert hip_lib.pnr_get_constants(c, k) == -1
    hip_lib.pnr_config_default(c); c.timestep = 0.0
    assert hip_lib.pnr_get_constants(c, k) == -1
    # static scene bodies (create_body_plane / _box / _sphere): dynamics mode only, sane shapes and sizes
    from pioneer_amd.config import EngineConfig, PioneerKinematicConfig, SimulationConfig, scene_box, scene_plane, scene_sphere, to_c_config
    good = (scene_plane((0, 0, 1), (0, 0, 2)), scene_box((1, 2, 3), (10, 0, 0), (0, 0, 0.5, 0.5)), scene_sphere(2.0, (15, 5, 5)))
    c = to_c_config(PioneerKinematicConfig(), SimulationConfig(), EngineConfig(mode="dynamic", scene=good))
    assert c.n_scene == 3 and hip_lib.pnr_get_constants(c, k) == 0
    c.mode = _lib.MODE_KINEMATIC
    assert hip_lib.pnr_get_constants(c, k) == -1 and b"dynamics mode only" in hip_lib.pnr_last_error(None)
    for bad, msg in ((scene_box((1, 0, 3), (0, 0, 0)), b"half extents"), (scene_sphere(0.0, (0, 0, 0)), b"radius"),
                     (scene_plane((0, 0, 0)), b"plane normal"), (scene_sphere(1.0, (0, 0, 0), (0, 0, 0, 0)), b"quaternion")):
        c = to_c_config(PioneerKinematicConfig(), SimulationConfig(), EngineConfig(mode="dynamic", scene=(bad,)))
        assert hip_lib.pnr_get_constants(c, k) == -1 and msg in hip_lib.pnr_last_error(None)
    c.n_scene = 9
    assert hip_lib.pnr_get_constants(c, k) == -1
    import pytest
    with pytest.raises(AssertionError):
        to_c_config(PioneerKinematicConfig(), SimulationConfig(), EngineConfig(mode="dynamic", scene=good * 3))


def test_create_fails_loudly_without_gpu(hip_lib):
    """No CPU backend: on a box without a HIP device pnr_create must fail with PNR_ERR_NODEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pioneer_amd import _lib
    c = _lib.PnrConfig(); hip_lib.pnr_config_default(c)
    h = C.c_void_p()
    rc = hip_lib.pnr_create(c, 16, 0, 0, 0, C.byref(h))
    assert rc == -4 and not h.value
    assert b"no HIP device" in hip_lib.pnr_last_error(None) or b"device" in hip_lib.pnr_last_error(None)
    from pioneer_amd import PioneerVectorEnv
    with pytest.raises(RuntimeError):
        PioneerVectorEnv(16)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under pioneer_amd/ may reference it."""
    pkg = os.path.join(ROOT, "pioneer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "pnr_oracle.h" not in text and "libpnr_oracle" not in text, f


def test_product_library_has_no_ablation_switches(hip_lib):
    """The timing-only ablations (PNR_DIAG, PNR_GRID_CAP*: wrong outputs when set) exist only in a -DPNR_DIAG_BUILD=1 variant:
    the product library must not read the environment, so it may not even hold such a string."""
    from pioneer_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    hits = sorted(set(re.findall(rb"PNR_[A-Z_]{3,}", blob)))
    assert hits == [], hits
    assert b"getenv" not in blob
    assert hip_lib.pnr_abi_version() == _lib.ABI_VERSION == 5


def test_missing_library_raises(monkeypatch, tmp_path):
    from pioneer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.load_library()


def test_library_built_from_other_sources_is_refused(tmp_path):
    """The build's identity is baked into the binary (content fingerprints per translation unit, -DPNR_UNIT_FINGERPRINT), not
    read off file times: a copy of the package whose source was touched after the build must refuse to load its library."""
    import shutil
    import subprocess
    import sys
    from pioneer_amd import _lib
    assert not _lib._stale(), "the in-tree library must carry the fingerprints of the in-tree sources"
    assert _lib.load_library().pnr_build_fingerprint().decode() == _lib.tree_fingerprint()
    assert _lib.embedded_fingerprints(_lib.LIB_PATH) == {_lib.UNIT_TAGS[u]: _lib.unit_fingerprint(u) for u in _lib.UNITS}
    root = tmp_path / "copy"
    shutil.copytree(os.path.join(ROOT, "pioneer_amd"), root / "pioneer_amd", ignore=shutil.ignore_patterns("*.o", "__pycache__", "libpioneer_amd_*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), root / "include")
    probe = "from pioneer_amd import _lib; _lib.load_library(); print('loaded', _lib._stale())"
    env = {k: v for k, v in os.environ.items() if k != "PNR_LIB_PATH"}
    r = subprocess.run([sys.executable, "-c", probe], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "loaded False" in r.stdout, r.stderr[-2000:]
    src = root / "pioneer_amd" / "csrc" / "pnr_device.h"
    src.write_text(src.read_text() + "\n// touched after the build\n")
    os.utime(src, (1, 1))                                     # .. and older than the library by its file time
    r = subprocess.run([sys.executable, "-c", probe], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "was built from other sources" in r.stderr, r.stdout + r.stderr[-2000:]


def test_spaces_and_helpers():
    from pioneer_amd.spaces import Box
    from pioneer_amd.env import arr2str
    b = Box(-np.ones(6, np.float32), np.ones(6, np.float32), dtype=np.float32)
    b.seed(0)
    x = b.sample()
    assert x.shape == (6,) and x.dtype == np.float32 and b.contains(x) and not b.contains(x + 3)
    o = Box(-np.inf, np.inf, shape=(137,), dtype=np.float64)
    assert o.shape == (137,) and o.dtype == np.float64
    assert arr2str(np.array([1.0, -2.5])) == "[1.000, -2.500]"     # collections_util.py:13-14


def test_optional_base_classes_follow_what_is_importable():
    """The façade derives from gym.Env / gym.Wrapper / ray.rllib.env.VectorEnv exactly where those packages can be imported
    (bullet_env.py:65, pioneer_knm_train.py:27) and from object otherwise — decided by importlib.util.find_spec, no stubs."""
    import collections
    import importlib.util
    from pioneer_amd import compat
    # the mechanism, on modules that do exist / do not exist here
    assert compat.optional_attr("collections", "OrderedDict") is collections.OrderedDict
    assert compat.optional_attr("os", "path.join") is os.path.join
    assert compat.optional_attr("no_such_module_xyz", "Env") is object
    assert compat.optional_attr("collections", "NoSuchAttr") is object
    # .. and its outcome in THIS interpreter, whichever it is
    have_gym = importlib.util.find_spec("gym") is not None
    have_ray = importlib.util.find_spec("ray") is not None
    assert compat.HAVE_GYM == have_gym or not have_gym
    from pioneer_amd.env import PioneerKinematicEnv, TimeLimit
    from pioneer_amd.rllib_env import PioneerRLlibVectorEnv
    if have_gym and compat.HAVE_GYM:
        import gym
        assert issubclass(PioneerKinematicEnv, gym.Env) and issubclass(TimeLimit, gym.Wrapper)
    else:
        assert PioneerKinematicEnv.__mro__[1] is object and TimeLimit.__mro__[1] is object
    if have_ray and compat.HAVE_RLLIB:
        from ray.rllib.env.vector_env import VectorEnv
        assert issubclass(PioneerRLlibVectorEnv, VectorEnv)
    else:
        assert PioneerRLlibVectorEnv.__mro__[1] is object
    # the VectorEnv contract's methods exist either way (RLlib 0.8.x names)
    for name in ("vector_reset", "reset_at", "vector_step", "get_unwrapped"):
        assert callable(getattr(PioneerRLlibVectorEnv, name))


def test_abi_version_is_the_same_in_every_document():
    """INTEGRATION.md's binding stub, DESIGN.md's boundary section and the header's changelog all name PNR_ABI_VERSION."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from pioneer_amd import _lib
    v = _lib.ABI_VERSION
    integ = open(os.path.join(root, "INTEGRATION.md")).read()
    assert re.findall(r"pnr_abi_version\(\) == (\d+)", integ) == [str(v)], "INTEGRATION.md's stub checks another ABI version"
    design = open(os.path.join(root, "DESIGN.md")).read()
    assert re.findall(r"ABI version (\d+)", design) and set(re.findall(r"ABI version (\d+)", design)) == {str(v)}
    header = open(os.path.join(root, "include", "pioneer_amd.h")).read()
    log = header[header.index("Bumped whenever"):header.index("#define PNR_ABI_VERSION")]
    assert [int(x) for x in re.findall(r"^ \*\s+(\d+)  ", log, re.M)] == list(range(1, v + 1)), "the header's ABI changelog must list every version"
