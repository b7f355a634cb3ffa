"""ctypes binding of libpioneer_amd.so (the C ABI of include/pioneer_amd.h).

There is no fallback: if the HIP library is missing or a call fails, this
module raises.  ``build_library()`` compiles it in-tree with hipcc for gfx950.
"""
import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# PNR_LIB_PATH: load another build of the library (A/B experiments such as tools/build_variant.py's); default in-tree
LIB_PATH = os.environ.get("PNR_LIB_PATH") or os.path.join(CSRC, "libpioneer_amd.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "pioneer_amd.h")



def _header_abi_version() -> int:
    """PNR_ABI_VERSION as include/pioneer_amd.h states it (the header is the single source)."""
    import re
    with open(HEADER) as f:
        m = re.search(r"^#define\s+PNR_ABI_VERSION\s+(\d+)", f.read(), re.M)
    if not m:
        raise ImportError(f"{HEADER}: no PNR_ABI_VERSION")
    return int(m.group(1))


ABI_VERSION = _header_abi_version()

DOF = 6
OBS_DIM = 137
STATE_WORDS = 24
INFO_DIM = 4
DYN_STATE_WORDS = 36

PNR_OK = 0
ENV_MAJOR, FEATURE_MAJOR = 0, 1
MODE_KINEMATIC, MODE_DYNAMIC = 0, 1
CONTROL_POSITION, CONTROL_VELOCITY = 0, 1


class PnrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pioneer_amd error {code}: {msg}")
        self.code = code


MAX_SCENE = 8
SHAPE_PLANE, SHAPE_BOX, SHAPE_SPHERE = 1, 2, 3


class PnrSceneBody(C.Structure):
    _fields_ = [("shape", C.c_int32), ("reserved", C.c_int32), ("position", C.c_double * 3), ("orientation", C.c_double * 4),
                ("size", C.c_double * 3)]


class PnrConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("max_v_to_r", C.c_double), ("max_a_to_v", C.c_double), ("done_distance", C.c_double),
        ("award_max", C.c_double), ("award_done", C.c_double),
        ("award_potential_slope", C.c_double), ("penalty_step", C.c_double),
        ("target_lo", C.c_double * 3), ("target_hi", C.c_double * 3), ("target_radius", C.c_double),
        ("timestep", C.c_double), ("frame_skip", C.c_int32), ("gravity", C.c_double),
        ("max_episode_steps", C.c_int32),
        ("auto_reset", C.c_int32), ("obs_layout", C.c_int32), ("action_layout", C.c_int32),
        ("mode", C.c_int32),
        ("pd_kp", C.c_double), ("pd_kd", C.c_double), ("torque_limit", C.c_double),
        ("joint_damping", C.c_double), ("joint_friction", C.c_double),
        ("teleport", C.c_int32), ("randomize", C.c_int32),
        ("rand_mass_lo", C.c_double), ("rand_mass_hi", C.c_double),
        ("rand_friction_lo", C.c_double), ("rand_friction_hi", C.c_double),
        ("rand_damping_lo", C.c_double), ("rand_damping_hi", C.c_double),
        ("ground_z", C.c_double), ("contact_kp", C.c_double), ("contact_kd", C.c_double),
        ("obstacle_position", C.c_double * 3), ("obstacle_half_extents", C.c_double * 3),
        ("pointer_radius", C.c_double),
        ("control_mode", C.c_int32), ("link_contacts", C.c_int32), ("max_velocity", C.c_double),
        ("n_scene", C.c_int32), ("pd_inertia_scaled", C.c_int32), ("scene", PnrSceneBody * MAX_SCENE),
    ]


class PnrConstants(C.Structure):
    _fields_ = [
        ("r_lo", C.c_float * DOF), ("r_hi", C.c_float * DOF),
        ("v_max", C.c_float * DOF), ("a_max", C.c_float * DOF),
        ("dt", C.c_double), ("eps", C.c_double),
    ]


class PnrRolloutStats(C.Structure):
    _fields_ = [("ep_ret", C.c_void_p), ("ep_len", C.c_void_p), ("scratch", C.c_void_p), ("scratch_doubles", C.c_int64),
                ("w_sum", C.c_void_p), ("w_len", C.c_void_p), ("w_cnt", C.c_void_p), ("w_max", C.c_void_p), ("w_min", C.c_void_p),
                ("adv_stats", C.c_void_p)]


class PnrMlpStep(C.Structure):
    """pnr_mlp_step of include/pioneer_amd.h (one PPO minibatch update, pnr_mlp_train_step)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("batch", C.c_int64),
        ("obs", C.c_void_p), ("idx", C.c_void_p),
        ("f_loc", C.c_void_p), ("f_inv", C.c_void_p), ("f_lo", C.c_void_p), ("f_hi", C.c_void_p),
        ("actions", C.c_void_p), ("logp_old", C.c_void_p), ("mean_old", C.c_void_p), ("log_std_old", C.c_void_p),
        ("adv", C.c_void_p), ("value_target", C.c_void_p), ("value_old", C.c_void_p),
        ("kl_coeff", C.c_void_p), ("entropy_coeff", C.c_void_p),
        ("clip_param", C.c_float), ("vf_clip_param", C.c_float), ("vf_loss_coeff", C.c_float),
        ("params", C.c_void_p * 12), ("n3_policy", C.c_int32), ("n3_value", C.c_int32),
        ("wpack", C.c_void_p), ("bias", C.c_void_p),
        ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("adam_step", C.c_void_p),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("head", C.c_void_p), ("g_head", C.c_void_p), ("xs", C.c_void_p), ("h1", C.c_void_p), ("h2", C.c_void_p),
        ("dz1", C.c_void_p), ("dz2", C.c_void_p),
        ("partials", C.c_void_p), ("partial_rows", C.c_int64), ("slabs", C.c_void_p), ("slab_floats", C.c_int64),
        ("means", C.c_void_p), ("flat_grad", C.c_void_p), ("xs_in", C.c_void_p),
        ("first_net", C.c_int32), ("n_nets", C.c_int32),
        ("w3_partials", C.c_void_p), ("w3_partial_floats", C.c_int64),
        ("planes", C.c_int32), ("xs_in_plane", C.c_int64),
    ]


# name -> (restype, argtypes); the list tests check against include/pioneer_amd.h
_VP = C.c_void_p
SIGNATURES = {
    "pnr_abi_version": (C.c_int, []),
    "pnr_build_fingerprint": (C.c_char_p, []),
    "pnr_config_default": (C.c_int, [C.POINTER(PnrConfig)]),
    "pnr_get_constants": (C.c_int, [C.POINTER(PnrConfig), C.POINTER(PnrConstants)]),
    "pnr_create": (C.c_int, [C.POINTER(PnrConfig), C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.POINTER(_VP)]),
    "pnr_destroy": (C.c_int, [_VP]),
    "pnr_seed": (C.c_int, [_VP, C.c_uint64]),
    "pnr_reset": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "pnr_step": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pnr_rollout": (C.c_int, [_VP, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pnr_observe": (C.c_int, [_VP, _VP, _VP]),
    "pnr_world_step": (C.c_int, [_VP, _VP, _VP]),
    "pnr_set_joint_motor": (C.c_int, [_VP, C.c_int32, C.c_int32] + [C.c_double] * 6),
    "pnr_get_state": (C.c_int, [_VP, _VP, _VP]),
    "pnr_set_state": (C.c_int, [_VP, _VP, _VP]),
    "pnr_get_dyn_state": (C.c_int, [_VP, _VP, _VP]),
    "pnr_set_dyn_state": (C.c_int, [_VP, _VP, _VP]),
    "pnr_diag_sincos": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int, _VP]),
    "pnr_ppo_loss": (C.c_int, [C.c_int64] + [_VP] * 12 + [C.c_float] * 3 + [_VP] * 3 + [C.c_int64, _VP, _VP]),
    "pnr_mlp_pack_elems": (C.c_int64, []),
    "pnr_mlp_bias_elems": (C.c_int64, []),
    "pnr_mlp_slab_floats": (C.c_int64, [C.c_int64]),
    "pnr_mlp_w3_partial_floats": (C.c_int64, [C.c_int64]),
    "pnr_mlp_pack": (C.c_int, [_VP, C.c_int32, C.c_int32, _VP, _VP, C.c_int32, _VP]),
    "pnr_mlp_forward": (C.c_int, [C.c_int64] + [_VP] * 12 + [C.c_int32, C.c_int32, C.c_int32, _VP]),
    "pnr_ppo_gae_scratch": (C.c_int64, [C.c_int64]),
    "pnr_ppo_gae": (C.c_int, [C.c_int32, C.c_int64] + [_VP] * 8 + [C.c_double, C.c_double] + [_VP] * 6),
    "pnr_filter_moments_scratch": (C.c_int64, [C.c_int64]),
    "pnr_filter_moments": (C.c_int, [C.c_int64, _VP, _VP, _VP, C.c_int64, _VP, _VP, _VP, _VP]),
    "pnr_filter_merge": (C.c_int, [_VP] * 8),
    "pnr_filter_prepare": (C.c_int, [_VP, _VP, _VP, C.c_double, _VP, _VP, _VP, _VP, _VP]),
    "pnr_permutation": (C.c_int, [C.c_int64, C.c_uint64, C.c_uint64, _VP, _VP]),
    "pnr_mlp_act": (C.c_int, [C.c_int64] + [_VP] * 16 + [C.c_int32, _VP]),
    "pnr_ppo_rollout": (C.c_int, [_VP, C.c_int32] + [_VP] * 18),
    "pnr_mlp_backward": (C.c_int, [C.c_int64] + [_VP] * 8 + [C.c_int64, _VP, C.c_int32, C.c_int32, C.c_int32, _VP, _VP]),
    "pnr_mlp_grad_floats": (C.c_int64, []),
    "pnr_ppo_pack_record": (C.c_int, [C.c_int64] + [_VP] * 11),
    "pnr_mlp_gather": (C.c_int, [C.c_int64] + [_VP] * 23 + [C.c_int32, _VP]),
    "pnr_mlp_train_step": (C.c_int, [C.POINTER(PnrMlpStep), _VP]),
    "pnr_mlp_adam": (C.c_int, [C.POINTER(PnrMlpStep), _VP, C.c_float, _VP]),
    "pnr_num_envs": (C.c_int64, [_VP]),
    "pnr_last_error": (C.c_char_p, [_VP]),
}


# the library's translation units and what each includes: a unit is recompiled when one of its files is newer than its object
UNITS = {
    "pnr_api.hip": ["pnr_api.hip", "pnr_host.h", "pnr_device.h", "pnr_model.h", "pnr_dyn.h", "pnr_env_kernels.h"],
    "pnr_learn.hip": ["pnr_learn.hip", "pnr_host.h", "pnr_device.h", "pnr_model.h", "pnr_ppo.h", "pnr_mlp.h", "pnr_sampler.h"],
}
SOURCES = sorted({f for deps in UNITS.values() for f in deps})      # every file a unit includes: what _stale() watches
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
               "-mllvm", "-amdgpu-kernarg-preload-count=16",      # leading scalar kernel args arrive preloaded in SGPRs
               "-Wall", "-Wno-unused-function"]
# the sources that define the env-side kernels (what the counter passes under profiles/ were taken on)
ENV_KERNEL_SOURCES = ["pnr_env_kernels.h", "pnr_device.h", "pnr_model.h", "pnr_dyn.h"]
# .. and the learner's (profiles/*learner_pmc_traffic.json: the bytes bench.py quotes in ppo_loop.roofline)
LEARNER_KERNEL_SOURCES = ["pnr_learn.hip", "pnr_mlp.h", "pnr_ppo.h"]


def source_fingerprint(files=ENV_KERNEL_SOURCES) -> str:
    """sha256 (first 16 hex digits) over the named csrc files and the compile flags: stored next to looked-up counter values
    (profiles/pmc_traffic.json, the dynamics SQ-counter summary) so that bench.py can tell when the kernels have changed since
    the counter pass."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


UNIT_TAGS = {"pnr_api.hip": "api", "pnr_learn.hip": "learn"}       # the names the units stamp themselves with (g_unit_fp)


def unit_fingerprint(unit: str, extra_flags=()) -> str:
    """What a unit's object must have been built from: the compile flags, its own files, the C-ABI header."""
    import hashlib
    h = hashlib.sha256(" ".join([*HIPCC_FLAGS, *extra_flags]).encode())
    for f in UNITS[unit]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    with open(HEADER, "rb") as fh:
        h.update(b"pioneer_amd.h\0" + fh.read())
    return h.hexdigest()[:16]


def tree_fingerprint(extra_flags=()) -> str:
    """The string pnr_build_fingerprint() of a library built from THIS tree returns."""
    return "".join(f"{UNIT_TAGS[u]}={unit_fingerprint(u, extra_flags)};" for u in UNITS)


def embedded_fingerprints(path: str) -> dict:
    """The pnr_build_fp tags inside an object or library file ({} if there are none): read from the bytes, nothing is loaded."""
    import re
    if not os.path.exists(path):
        return {}
    with open(path, "rb") as f:
        blob = f.read()
    return {m.group(1).decode(): m.group(2).decode() for m in re.finditer(rb"pnr_build_fp:([a-z]+)=([0-9a-f]{16});", blob)}


def _obj_path(unit: str, tag: str = "") -> str:
    return os.path.join(CSRC, os.path.splitext(unit)[0] + tag + ".o")


def _unit_stale(unit: str, obj: str, extra_flags=()) -> bool:
    return embedded_fingerprints(obj).get(UNIT_TAGS[unit]) != unit_fingerprint(unit, extra_flags)


def _stale() -> bool:
    """The in-tree library is stale unless every unit inside it carries the fingerprint of the tree's sources (content, not mtime)."""
    have = embedded_fingerprints(LIB_PATH)
    return any(have.get(UNIT_TAGS[u]) != unit_fingerprint(u) for u in UNITS)


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), out_path: str = None, units=None) -> str:
    """hipcc --offload-arch=gfx950 -> pioneer_amd/csrc/libpioneer_amd.so (in-tree): the two translation units are compiled in
    parallel (only those whose object does not carry the fingerprint of the current sources) and linked.  extra_flags / out_path
    build a variant next to it (e.g. -DPNR_DYN_LDS_MODEL=1 for the LDS-staging A/B, -DPNR_DIAG_BUILD=1 for the timing-only
    ablations), with objects of its own; `units` names the translation units the flags concern (e.g. ("pnr_learn.hip",) for an
    MLP kernel A/B) — the others are linked from the default build's objects.  A full forced build takes ~95 s on 8 cores
    (pnr_api.hip: ~95 s, pnr_learn.hip: ~20 s, in parallel)."""
    variant = out_path is not None or bool(extra_flags)
    if variant:
        force = True
        if out_path is None:
            raise ValueError("a variant build (extra_flags) needs its own out_path")
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libpioneer_amd.so")
    tag = ("." + os.path.splitext(os.path.basename(out_path))[0]) if variant else ""
    procs, objs = [], []
    own = []
    for unit, deps in UNITS.items():
        if variant and units is not None and unit not in units:
            objs.append(_obj_path(unit))                 # the default build's object (build_library() must have run)
            if _unit_stale(unit, objs[-1]):
                raise RuntimeError(f"{objs[-1]} is missing or not built from the current sources: build the default library first")
            continue
        obj = _obj_path(unit, tag)
        objs.append(obj)
        own.append(obj)
        flags = list(extra_flags) if (variant and (units is None or unit in units)) else []
        if not force and not _unit_stale(unit, obj, flags):
            continue
        cmd = [hipcc, *HIPCC_FLAGS, *flags, f'-DPNR_UNIT_FINGERPRINT="{unit_fingerprint(unit, flags)}"', "-c", "-o", obj,
               os.path.join(CSRC, unit)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_path or LIB_PATH, *objs]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=CSRC)
    if variant:
        for o in own:
            os.remove(o)
    return out_path or LIB_PATH


_lib = None


def load_library():
    """Load the HIP library; raise loudly if it is not there (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  pioneer_amd has no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as ROCm's).  If this
    # library were loaded first it would pull in /opt/rocm's copy, torch would then load its own, and
    # the two runtimes would not see each other's devices/streams.  Importing torch first makes our
    # DT_NEEDED libamdhip64.so.7 resolve to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pnr_abi_version() != ABI_VERSION:
        raise ImportError(f"libpioneer_amd.so reports ABI {lib.pnr_abi_version()}, include/pioneer_amd.h says {ABI_VERSION}; rebuild")
    # the binary must be the one these sources build (content fingerprints baked in at compile time, not file times); a library
    # named explicitly through PNR_LIB_PATH is somebody's A/B variant with flags of its own and is taken as it is
    built, tree = lib.pnr_build_fingerprint().decode(), tree_fingerprint()
    if built != tree and not os.environ.get("PNR_LIB_PATH"):
        raise ImportError(f"{LIB_PATH} was built from other sources (binary {built}, tree {tree}): rebuild it with "
                          "`python -c 'import __graft_entry__ as g; g.build()'`")
    _lib = lib
    return lib


def check(code, handle=None):
    if code != PNR_OK:
        msg = load_library().pnr_last_error(handle)
        raise PnrError(code, msg.decode() if msg else "unknown")
