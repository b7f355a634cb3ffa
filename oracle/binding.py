"""ctypes binding of the C oracle (test infrastructure; see pnr_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DOF = 6
OBS = 137


def oracle_lib_path() -> str:
    return os.path.join(_HERE, "libpnr_oracle.so")


def build_oracle(force: bool = False) -> str:
    """Compile libpnr_oracle.so with gcc (a few hundred ms)."""
    path = oracle_lib_path()
    srcs = [os.path.join(_HERE, f) for f in
            ("pnr_oracle.c", "pnr_oracle.h", "pnr_dyn_oracle.c", "pnr_dyn_oracle.h", "Makefile")]
    stale = (not os.path.exists(path)) or any(
        os.path.getmtime(s) > os.path.getmtime(path) for s in srcs if os.path.exists(s))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return path


class OrcParams(C.Structure):
    _fields_ = [
        ("max_v_to_r", C.c_double), ("max_a_to_v", C.c_double), ("done_distance", C.c_double),
        ("award_max", C.c_double), ("award_done", C.c_double),
        ("award_potential_slope", C.c_double), ("penalty_step", C.c_double),
        ("target_lo", C.c_double * 3), ("target_hi", C.c_double * 3),
        ("timestep", C.c_double),
        ("frame_skip", C.c_int32), ("max_episode_steps", C.c_int32),
        ("precision", C.c_int32), ("auto_reset", C.c_int32),
        ("seed", C.c_uint64),
        ("r_lo", C.c_float * DOF), ("r_hi", C.c_float * DOF),
        ("v_max", C.c_float * DOF), ("a_max", C.c_float * DOF),
        ("dt", C.c_double), ("eps", C.c_double),
    ]


class OrcState(C.Structure):
    _fields_ = [
        ("a", C.c_float * DOF), ("v", C.c_float * DOF), ("r", C.c_double * DOF),
        ("r_is_f64", C.c_int32),
        ("target", C.c_double * 3), ("potential", C.c_double),
        ("step_index", C.c_uint32), ("episode", C.c_uint32),
    ]


STATE_DTYPE = np.dtype([
    ("a", np.float32, DOF), ("v", np.float32, DOF), ("r", np.float64, DOF),
    ("r_is_f64", np.int32), ("_pad", np.int32),
    ("target", np.float64, 3), ("potential", np.float64),
    ("step_index", np.uint32), ("episode", np.uint32),
])

ORC_REF, ORC_DEV = 0, 1


def _ptr(arr, ctype):
    if arr is None:
        return None
    return arr.ctypes.data_as(C.POINTER(ctype))


class COracle:
    """Batched CPU oracle.  ``precision``: ORC_REF (reference-exact) or ORC_DEV."""

    def __init__(self, num_envs=1, seed=0, precision=ORC_REF, auto_reset=False,
                 max_episode_steps=500, env_id_offset=0, nthreads=1, **tunables):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        assert L.orc_sizeof_params() == C.sizeof(OrcParams), "orc_params layout drift"
        assert L.orc_sizeof_state() == C.sizeof(OrcState) == STATE_DTYPE.itemsize, \
            "orc_state layout drift"
        L.orc_potential.restype = C.c_double
        L.orc_potential.argtypes = [C.POINTER(OrcParams), C.c_double]
        self.p = OrcParams()
        L.orc_params_default(C.byref(self.p))
        for k, v in tunables.items():
            if k in ("target_lo", "target_hi"):
                for i in range(3):
                    getattr(self.p, k)[i] = float(v[i])
            else:
                assert hasattr(self.p, k), k
                setattr(self.p, k, v)
        self.p.seed = seed
        self.p.precision = precision
        self.p.auto_reset = int(auto_reset)
        self.p.max_episode_steps = max_episode_steps
        L.orc_params_derive(C.byref(self.p))
        self.n = int(num_envs)
        self.off = int(env_id_offset)
        self.nthreads = int(nthreads)
        self.state = np.zeros(self.n, dtype=STATE_DTYPE)

    # -- constants ---------------------------------------------------------------
    @property
    def r_lo(self): return np.array(self.p.r_lo[:], dtype=np.float32)
    @property
    def r_hi(self): return np.array(self.p.r_hi[:], dtype=np.float32)
    @property
    def v_max(self): return np.array(self.p.v_max[:], dtype=np.float32)
    @property
    def a_max(self): return np.array(self.p.a_max[:], dtype=np.float32)
    @property
    def dt(self): return self.p.dt

    def _sp(self):
        return self.state.ctypes.data_as(C.POINTER(OrcState))

    # -- calls ---------------------------------------------------------------------
    def fk(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, DOF)
        out = np.empty((q.shape[0], 3))
        for i in range(q.shape[0]):
            self.lib.orc_fk_pointer(_ptr(q[i], C.c_double), _ptr(out[i], C.c_double))
        return out

    def potential(self, d):
        return self.lib.orc_potential(C.byref(self.p), float(d))

    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        self.lib.orc_philox4x32_10(c, k, o)
        return list(o)

    def reset(self, mask=None, joint_pos=None, target_pos=None, want_obs=True):
        n = self.n
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        jp = None if joint_pos is None else np.ascontiguousarray(joint_pos, dtype=np.float64).reshape(n, DOF)
        tp = None if target_pos is None else np.ascontiguousarray(target_pos, dtype=np.float64).reshape(n, 3)
        obs = np.zeros((n, OBS)) if want_obs else None
        self.lib.orc_reset_batch(C.byref(self.p), self._sp(), C.c_int64(n), C.c_int64(self.off),
                                 _ptr(m, C.c_uint8), _ptr(jp, C.c_double), _ptr(tp, C.c_double),
                                 _ptr(obs, C.c_double), C.c_int(self.nthreads))
        return obs

    def step(self, actions, want_obs=True, want_info=False):
        n = self.n
        act = np.ascontiguousarray(actions, dtype=np.float32).reshape(n, DOF)
        obs = np.empty((n, OBS)) if want_obs else None
        rew = np.empty(n)
        done = np.empty(n, dtype=np.uint8)
        trunc = np.empty(n, dtype=np.uint8)
        info = np.empty((n, 4)) if want_info else None
        self.lib.orc_step_batch(C.byref(self.p), self._sp(), C.c_int64(n), C.c_int64(self.off),
                                _ptr(act, C.c_float), _ptr(obs, C.c_double), _ptr(rew, C.c_double),
                                _ptr(done, C.c_uint8), _ptr(trunc, C.c_uint8), _ptr(info, C.c_double),
                                C.c_int(self.nthreads))
        if want_info:
            return obs, rew, done, trunc, info
        return obs, rew, done, trunc

    def observe(self):
        obs = np.empty((self.n, OBS))
        for e in range(self.n):
            self.lib.orc_observe(C.byref(self.p), C.byref(OrcState.from_buffer(self.state, e * STATE_DTYPE.itemsize)),
                                 _ptr(obs[e], C.c_double))
        return obs

    # -- engine interchange -----------------------------------------------------------
    def state_words(self):
        w = np.empty((24, self.n), dtype=np.uint32)
        self.lib.orc_state_to_words(self._sp(), C.c_int64(self.n), _ptr(w, C.c_uint32))
        return w

    def load_state_words(self, w):
        w = np.ascontiguousarray(w, dtype=np.uint32).reshape(24, self.n)
        self.lib.orc_state_from_words(self._sp(), C.c_int64(self.n), _ptr(w, C.c_uint32))


# ---- dynamics-mode oracle (pnr_dyn_oracle.c) --------------------------------------------------
LINKS = 11


ORC_MAX_SCENE = 8
SHAPES = {"plane": 1, "box": 2, "sphere": 3}


class OrcSceneBody(C.Structure):
    _fields_ = [("shape", C.c_int32), ("pad", C.c_int32), ("position", C.c_double * 3), ("orientation", C.c_double * 4),
                ("size", C.c_double * 3)]


class OrcDynParams(C.Structure):
    _fields_ = [
        ("kp", C.c_double), ("kd", C.c_double), ("torque_limit", C.c_double), ("gravity", C.c_double),
        ("timestep", C.c_double), ("frame_skip", C.c_int32), ("teleport", C.c_int32),
        ("randomize", C.c_int32), ("pad", C.c_int32),
        ("joint_damping", C.c_double), ("joint_friction", C.c_double),
        ("rand_mass_lo", C.c_double), ("rand_mass_hi", C.c_double),
        ("rand_friction_lo", C.c_double), ("rand_friction_hi", C.c_double),
        ("rand_damping_lo", C.c_double), ("rand_damping_hi", C.c_double),
        ("ground_z", C.c_double), ("contact_kp", C.c_double), ("contact_kd", C.c_double),
        ("obstacle_position", C.c_double * 3), ("obstacle_half_extents", C.c_double * 3),
        ("pointer_radius", C.c_double),
        ("control_mode", C.c_int32), ("link_contacts", C.c_int32), ("max_velocity", C.c_double),
        ("n_scene", C.c_int32), ("pd_inertia_scaled", C.c_int32), ("scene", OrcSceneBody * ORC_MAX_SCENE),
    ]


class OrcContactSample(C.Structure):
    _fields_ = [("body", C.c_int32), ("c", C.c_double * 3), ("radius", C.c_double)]


DYN_STATE_DTYPE = np.dtype([
    ("q", np.float64, DOF), ("qd", np.float64, DOF), ("mass_scale", np.float64, LINKS),
    ("friction", np.float64, DOF), ("damping", np.float64, DOF),
])


class OrcJointMotor(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("control_mode", C.c_int32), ("target_position", C.c_double), ("target_velocity", C.c_double),
                ("position_gain", C.c_double), ("velocity_gain", C.c_double), ("max_force", C.c_double), ("max_velocity", C.c_double)]


class DynOracle(COracle):
    """Batched dynamics-mode oracle: kinematic command state + simulated (q, qd)."""

    def __init__(self, num_envs=1, dyn=None, **kw):
        kw.setdefault("precision", ORC_DEV)
        super().__init__(num_envs, **kw)
        L = self.lib
        assert L.orc_dyn_sizeof_params() == C.sizeof(OrcDynParams), "orc_dyn_params layout drift"
        assert L.orc_dyn_sizeof_state() == DYN_STATE_DTYPE.itemsize, "orc_dyn_state layout drift"
        self.d = OrcDynParams()
        L.orc_dyn_params_default(C.byref(self.d))
        self.d.timestep = self.p.timestep
        self.d.frame_skip = self.p.frame_skip
        for k, v in (dyn or {}).items():
            assert hasattr(self.d, k), k
            if k in ("obstacle_position", "obstacle_half_extents"):
                for i in range(3):
                    getattr(self.d, k)[i] = float(v[i])
            elif k == "scene":
                # [(shape, position, orientation xyzw, size3)], the static bodies of create_body_plane / _box / _sphere
                assert len(v) <= ORC_MAX_SCENE
                self.d.n_scene = len(v)
                for i, (shape, pos, quat, size) in enumerate(v):
                    b = self.d.scene[i]
                    b.shape = SHAPES[shape]
                    for j in range(3):
                        b.position[j] = float(pos[j]); b.size[j] = float(size[j])
                    for j in range(4):
                        b.orientation[j] = float(quat[j])
            else:
                setattr(self.d, k, v)
        self.dstate = np.zeros(self.n, dtype=DYN_STATE_DTYPE)
        self.dstate["mass_scale"] = 1.0

    def _dp(self):
        return self.dstate.ctypes.data_as(C.c_void_p)

    def reset(self, mask=None, joint_pos=None, target_pos=None, want_obs=True):
        n = self.n
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        jp = None if joint_pos is None else np.ascontiguousarray(joint_pos, dtype=np.float64).reshape(n, DOF)
        tp = None if target_pos is None else np.ascontiguousarray(target_pos, dtype=np.float64).reshape(n, 3)
        obs = np.zeros((n, OBS)) if want_obs else None
        self.lib.orc_dyn_reset_batch(C.byref(self.d), C.byref(self.p), self._sp(), self._dp(), C.c_int64(n),
                                     C.c_int64(self.off), _ptr(m, C.c_uint8), _ptr(jp, C.c_double),
                                     _ptr(tp, C.c_double), _ptr(obs, C.c_double), C.c_int(self.nthreads))
        return obs

    def step(self, actions, want_obs=True, want_info=False):
        n = self.n
        act = np.ascontiguousarray(actions, dtype=np.float32).reshape(n, DOF)
        obs = np.empty((n, OBS)) if want_obs else None
        rew = np.empty(n); done = np.empty(n, dtype=np.uint8); trunc = np.empty(n, dtype=np.uint8)
        info = np.empty((n, 4)) if want_info else None
        self.lib.orc_dyn_step_batch(C.byref(self.d), C.byref(self.p), self._sp(), self._dp(), C.c_int64(n),
                                    C.c_int64(self.off), _ptr(act, C.c_float), _ptr(obs, C.c_double),
                                    _ptr(rew, C.c_double), _ptr(done, C.c_uint8), _ptr(trunc, C.c_uint8),
                                    _ptr(info, C.c_double), C.c_int(self.nthreads))
        return (obs, rew, done, trunc, info) if want_info else (obs, rew, done, trunc)

    # single-env helpers on dstate[e]
    def _one(self, e=0):
        return C.c_void_p(self.dstate.ctypes.data + e * DYN_STATE_DTYPE.itemsize)

    def aba(self, tau, gravity=0.0, f_tip=None, e=0):
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        qdd = np.empty(DOF)
        f = None if f_tip is None else np.ascontiguousarray(f_tip, dtype=np.float64)
        self.lib.orc_dyn_aba(self._one(e), _ptr(tau, C.c_double), C.c_double(gravity), _ptr(f, C.c_double),
                             _ptr(qdd, C.c_double))
        return qdd

    def aba_ext(self, tau, gravity=0.0, fext=None, e=0):
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        qdd = np.empty(DOF)
        f = None if fext is None else np.ascontiguousarray(fext, dtype=np.float64).reshape(DOF, 6)
        self.lib.orc_dyn_aba_ext(self._one(e), _ptr(tau, C.c_double), C.c_double(gravity), _ptr(f, C.c_double), _ptr(qdd, C.c_double))
        return qdd

    def motor_torque(self, r_ref, v_ref, q, qd):
        self.lib.orc_dyn_motor_torque.restype = C.c_double
        return self.lib.orc_dyn_motor_torque(C.byref(self.d), C.c_double(r_ref), C.c_double(v_ref), C.c_double(q), C.c_double(qd))

    def nominal_inertia(self):
        J = np.zeros(DOF)
        self.lib.orc_dyn_nominal_inertia(_ptr(J, C.c_double))
        return J

    def contact_samples(self):
        arr = (OrcContactSample * 23)()
        n = self.lib.orc_dyn_contact_samples(C.byref(self.d), arr)
        return [(arr[i].body, np.array(arr[i].c[:]), arr[i].radius) for i in range(n)]

    def contact_wrenches(self, e=0):
        f = np.zeros((DOF, 6))
        any_ = self.lib.orc_dyn_contact_wrenches(C.byref(self.d), self._one(e), _ptr(f, C.c_double))
        return bool(any_), f

    def energy(self, gravity=0.0, e=0):
        ke, pe = C.c_double(), C.c_double()
        self.lib.orc_dyn_energy(self._one(e), C.c_double(gravity), C.byref(ke), C.byref(pe))
        return ke.value, pe.value

    def tip(self, e=0):
        pos, vel = np.empty(3), np.empty(3)
        self.lib.orc_dyn_tip(self._one(e), _ptr(pos, C.c_double), _ptr(vel, C.c_double))
        return pos, vel

    # World.step() alone with per-joint motors (orc_dyn_world_step): motors = {joint: dict(control_mode, target_position, target_velocity,
    # position_gain, velocity_gain, max_force, max_velocity)}; a joint that is not named keeps the env-wide law
    def set_joint_motor(self, joint, control_mode, target_position=0.0, target_velocity=0.0, position_gain=None, velocity_gain=None,
                        max_force=None, max_velocity=None):
        if not hasattr(self, "motors"):
            self.motors = (OrcJointMotor * DOF)()
        m = self.motors[joint]
        m.enabled, m.control_mode = 1, int(control_mode)
        m.target_position, m.target_velocity = float(target_position), float(target_velocity)
        m.position_gain = float(self.d.kp if position_gain is None else position_gain)
        m.velocity_gain = float(self.d.kd if velocity_gain is None else velocity_gain)
        m.max_force = float(self.d.torque_limit if max_force is None else max_force)
        m.max_velocity = float(self.d.max_velocity if max_velocity is None else max_velocity)

    def world_step(self):
        motors = getattr(self, "motors", None)
        for e in range(self.n):
            self.lib.orc_dyn_world_step(C.byref(self.d), C.byref(self.p), C.c_void_p(self.state.ctypes.data + e * self.state.dtype.itemsize),
                                        self._one(e), motors)

    def substep(self, r_ref, v_ref, e=0):
        r = np.ascontiguousarray(r_ref, dtype=np.float64); v = np.ascontiguousarray(v_ref, dtype=np.float64)
        self.lib.orc_dyn_substep(C.byref(self.d), C.byref(self.p), self._one(e), _ptr(r, C.c_double), _ptr(v, C.c_double))

    def dyn_words(self):
        """Engine interchange: planar float32 [36][n] (include/pioneer_amd.h pnr_get_dyn_state)."""
        w = np.zeros((36, self.n), dtype=np.float32)
        w[0:6] = self.dstate["q"].T; w[6:12] = self.dstate["qd"].T
        w[12:23] = self.dstate["mass_scale"].T; w[23:29] = self.dstate["friction"].T
        w[29:35] = self.dstate["damping"].T
        return w

    def load_dyn_words(self, w):
        w = np.asarray(w, dtype=np.float32).reshape(36, self.n)
        self.dstate["q"] = w[0:6].T; self.dstate["qd"] = w[6:12].T
        self.dstate["mass_scale"] = w[12:23].T; self.dstate["friction"] = w[23:29].T
        self.dstate["damping"] = w[29:35].T
