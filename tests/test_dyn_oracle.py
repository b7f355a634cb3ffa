"""CPU tests of the dynamics-mode oracle (PARITY UNPINNED: the reference pins no dynamics).

The ABA of oracle/pnr_dyn_oracle.c (body-coordinate spatial algebra over six merged bodies)
is checked against an independent Lagrangian formulation written here: per-LINK world-frame
Jacobians -> mass matrix M(q); Coriolis and gravity terms by numerical differentiation of
M(q) and of the potential energy.  Plus energy conservation, contact, limits, and the
reduction to the kinematic (reference) semantics under teleport.
"""
import math

import numpy as np
import pytest

from oracle import COracle, DynOracle
from oracle.binding import ORC_DEV, _ptr
from oracle.numpy_twin import CHAIN, _axis_angle

# URDF link order (tests/golden/urdf_chain.json): world, base, rotator1, hinge1, arm1, arm2,
# rotator2, hinge2, arm3, rotator3, effector, pointer.  Link l (1-based, skipping world) is the
# child of CHAIN[l-1].


def link_frames(q):
    """World rotation / origin of every non-world link + each revolute joint's world axis and position."""
    R, p = np.eye(3), np.zeros(3)
    frames, joints, qi = [], [], 0
    for jtype, origin, axis in CHAIN:
        p = p + R @ np.array(origin, dtype=float)
        if jtype == "revolute":
            joints.append((R @ np.array(axis, dtype=float), p.copy(), len(frames)))
            R = R @ _axis_angle(axis, q[qi]); qi += 1
        frames.append((R.copy(), p.copy()))
    return frames, joints


def mass_matrix(q, scale):
    frames, joints = link_frames(q)
    M = np.zeros((6, 6))
    for l, (R, p) in enumerate(frames):
        if l == 0:
            continue                      # robot:base is fixed to the world
        m, inertia = 1.0 * scale[l], 1.0 * scale[l]
        Jv, Jw = np.zeros((3, 6)), np.zeros((3, 6))
        for j, (z, o, first_link) in enumerate(joints):
            if first_link <= l:           # joint j is an ancestor of link l
                Jw[:, j] = z
                Jv[:, j] = np.cross(z, p - o)
        M += m * Jv.T @ Jv + inertia * Jw.T @ Jw      # isotropic link inertia: R I R^T = I
    return M


def potential_energy(q, scale, g):
    frames, _ = link_frames(q)
    return sum(1.0 * scale[l] * g * p[2] for l, (R, p) in enumerate(frames) if l > 0)


def lagrange_qdd(q, qd, tau, scale, g, f_tip=None, h=1e-6):
    M = mass_matrix(q, scale)
    dM = np.zeros((6, 6, 6))
    G = np.zeros(6)
    for k in range(6):
        e = np.zeros(6); e[k] = h
        dM[:, :, k] = (mass_matrix(q + e, scale) - mass_matrix(q - e, scale)) / (2 * h)
        G[k] = (potential_energy(q + e, scale, g) - potential_energy(q - e, scale, g)) / (2 * h)
    Cq = np.einsum("ijk,j,k->i", dM, qd, qd) - 0.5 * np.einsum("jki,j,k->i", dM, qd, qd)
    Q = np.array(tau, dtype=float)
    if f_tip is not None:
        frames, joints = link_frames(q)
        ptip = frames[-1][1]
        Jv = np.stack([np.cross(z, ptip - o) for (z, o, _) in joints], axis=1)
        Q = Q + Jv.T @ np.asarray(f_tip, dtype=float)
    return np.linalg.solve(M, Q - Cq - G)


@pytest.fixture(scope="module")
def dyn(oracle_built):
    return DynOracle(1)


def set_state(o, q, qd, scale=None):
    o.dstate["q"][0] = q; o.dstate["qd"][0] = qd
    o.dstate["mass_scale"][0] = np.ones(11) if scale is None else scale
    o.dstate["friction"][0] = 0; o.dstate["damping"][0] = 0


def test_aba_equals_lagrangian(dyn):
    rng = np.random.RandomState(0)
    for trial in range(12):
        q = rng.uniform(dyn.r_lo, dyn.r_hi).astype(np.float64)
        qd = rng.uniform(-2, 2, 6)
        tau = rng.uniform(-50, 50, 6)
        scale = np.ones(11) if trial < 4 else rng.uniform(0.5, 1.5, 11)
        g = [0.0, 9.81, 3.0][trial % 3]
        f = None if trial % 2 == 0 else rng.uniform(-20, 20, 3)
        set_state(dyn, q, qd, scale)
        got = dyn.aba(tau, gravity=g, f_tip=f)
        want = lagrange_qdd(q, qd, tau, scale, g, f)
        assert np.abs(got - want).max() < 2e-6 * max(1.0, np.abs(want).max())


def test_energy_matches_lagrangian_and_tip_matches_fk(dyn):
    rng = np.random.RandomState(1)
    q = rng.uniform(dyn.r_lo, dyn.r_hi).astype(np.float64); qd = rng.uniform(-1, 1, 6)
    scale = rng.uniform(0.5, 1.5, 11)
    set_state(dyn, q, qd, scale)
    ke, pe = dyn.energy(9.81)
    assert abs(ke - 0.5 * qd @ mass_matrix(q, scale) @ qd) < 1e-9
    assert abs(pe - potential_energy(q, scale, 9.81)) < 1e-9
    pos, vel = dyn.tip()
    assert np.abs(pos - dyn.fk([q])[0]).max() < 1e-12
    h = 1e-7
    num = (dyn.fk([q + h * qd])[0] - dyn.fk([q - h * qd])[0]) / (2 * h)
    assert np.abs(vel - num).max() < 1e-6


def test_energy_conservation_passive(oracle_built):
    """No torque, gravity on, damping off: E = KE + PE is conserved up to the integrator's O(dt) drift."""
    o = DynOracle(1, dyn=dict(teleport=1, gravity=9.81, timestep=2e-5))
    set_state(o, [0.2, 0.3, -0.4, 0.5, 0.3, -0.2], [0.3, -0.2, 0.1, 0.4, -0.3, 0.2])
    e0 = sum(o.energy(9.81))
    ke0 = o.energy(9.81)[0]
    z = np.zeros(6)
    for _ in range(5000):               # 0.1 s
        o.substep(z, z)
    ke1, pe1 = o.energy(9.81)
    assert abs(ke1 - ke0) > 1.0          # energy really moved between KE and PE
    assert abs(ke1 + pe1 - e0) < 2e-3 * abs(ke1 - ke0)


def test_teleport_zero_gravity_is_kinematic_mode(oracle_built):
    """with the reference's defaults (gravity 0, teleported joints, no colliders) the
    dynamics sub-steps are the identity, so dynamics mode reproduces kinematic mode bit for bit."""
    n = 64
    kin = COracle(n, seed=4, precision=ORC_DEV, auto_reset=True, max_episode_steps=9)
    dyn = DynOracle(n, seed=4, precision=ORC_DEV, auto_reset=True, max_episode_steps=9,
                    dyn=dict(teleport=1, gravity=0.0))
    assert np.array_equal(kin.reset(), dyn.reset())
    rng = np.random.RandomState(0)
    for _ in range(30):
        a = (rng.uniform(-1, 1, (n, 6)) * kin.a_max).astype(np.float32)
        ok, rk, dk, tk = kin.step(a)
        od, rd, dd, td = dyn.step(a)
        assert np.array_equal(ok, od) and np.array_equal(rk, rd)
        assert np.array_equal(dk, dd) and np.array_equal(tk, td)


def test_pd_tracks_the_kinematic_command(oracle_built):
    n = 16
    o = DynOracle(n, seed=2, precision=ORC_DEV, max_episode_steps=0, dyn=dict(kp=4000.0, kd=400.0))
    o.reset()
    rng = np.random.RandomState(3)
    for t in range(120):
        a = (rng.uniform(-0.05, 0.05, (n, 6)) * o.a_max).astype(np.float32)
        o.step(a)
    err = np.abs(o.dstate["q"] - o.state["r"])
    assert err[:, 3:].max() < 0.15       # light distal joints track tightly
    assert err.max() < 0.6               # the heavy base joints lag but stay bounded
    assert np.all(o.dstate["q"] <= o.r_hi + 1e-12) and np.all(o.dstate["q"] >= o.r_lo - 1e-12)


def test_ground_contact_pushes_the_pointer_up(oracle_built):
    """Arm released horizontally (tip at z = 15.9) above a plane at z = 10 under gravity."""
    z = np.zeros(6)
    traj = {}
    for name, gz in (("plane", 10.0), ("free", float("nan"))):
        o = DynOracle(1, dyn=dict(teleport=1, gravity=9.81, ground_z=gz, contact_kp=2000.0, contact_kd=50.0))
        set_state(o, np.zeros(6), np.zeros(6))
        zs = []
        for _ in range(2400):             # 10 s
            o.substep(z, z)
            zs.append(o.tip()[0][2])
        traj[name] = np.array(zs)
    assert traj["free"].min() < 10.0 - 5.0            # without the plane the pointer swings far below
    assert traj["plane"].min() > 10.0 - 1.0           # penetration bounded by the penalty spring
    assert abs(traj["plane"][-1] - 10.0) < 0.5        # and it comes to rest on the plane


def test_randomisation_draws(oracle_built):
    o = DynOracle(2048, seed=8, dyn=dict(randomize=1))
    o.reset(want_obs=False)
    ms, fr, dm = o.dstate["mass_scale"], o.dstate["friction"], o.dstate["damping"]
    assert ms.min() >= 0.5 and ms.max() <= 1.5 and abs(ms.mean() - 1.0) < 0.02
    assert fr.min() >= 0 and fr.max() <= 0.1 and dm.min() >= 0 and dm.max() <= 0.1
    assert np.array_equal(o.dstate["q"], o.state["r"]) and not o.dstate["qd"].any()
    o2 = DynOracle(1024, seed=8, env_id_offset=1024, dyn=dict(randomize=1))
    o2.reset(want_obs=False)
    assert np.array_equal(o2.dstate["mass_scale"], ms[1024:])


def test_box_contact_force_geometry(oracle_built):
    """Penalty force of the static box (the reference demo's obstacle:1 geometry) on the pointer sphere."""
    import ctypes as C
    o = DynOracle(1, dyn=dict(obstacle_position=(10.0, 5.0, 0.0), obstacle_half_extents=(0.5, 0.5, 5.0),
                              pointer_radius=0.2, contact_kp=2000.0, contact_kd=50.0))
    o.lib.orc_dyn_contact_force.restype = C.c_int

    def force(pos, vel=(0, 0, 0)):
        p = np.array(pos, dtype=np.float64); v = np.array(vel, dtype=np.float64); f = np.zeros(3)
        act = o.lib.orc_dyn_contact_force(C.byref(o.d), p.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
                                          f.ctypes.data_as(C.c_void_p))
        return act, f
    assert force((12.0, 5.0, 1.0))[0] == 0                                     # clear of the box
    act, f = force((10.6, 5.0, 1.0))                                           # 0.1 outside the +x face: depth 0.1
    assert act == 1 and np.allclose(f, [2000.0 * 0.1, 0, 0])
    act, f = force((10.6, 5.0, 1.0), vel=(-1.0, 0, 0))                         # approaching: damping adds
    assert np.allclose(f, [2000.0 * 0.1 + 50.0, 0, 0])
    act, f = force((10.6, 5.0, 1.0), vel=(10.0, 0, 0))                         # separating fast: no pull
    assert act == 0 and not f.any()
    act, f = force((10.0, 5.0, 5.1))                                           # above the top face
    assert np.allclose(f, [0, 0, 2000.0 * 0.1])
    act, f = force((10.4, 5.0, 1.0))                                           # inside: out through the nearest (+x) face
    assert f[0] > 0 and f[1] == 0 and f[2] == 0 and abs(f[0] - 2000.0 * (0.2 + 0.1)) < 1e-9
    k = 0.1 / np.sqrt(2)
    act, f = force((10.5 + k, 5.5 + k, 1.0))                                   # off an edge: normal along the diagonal
    assert act == 1 and abs(f[0] - f[1]) < 1e-9 and f[2] == 0 and abs(np.linalg.norm(f) - 2000.0 * 0.1) < 1e-6


# ---- the rest of the reference's motor surface (bullet_scene.py:123-155) and link contacts (r02) -----------------------
def body_point_world(q, body, c):
    """World position of point c (body frame) of moving body `body` (0-based, fixed joints merged): the body frame is
    the frame right behind revolute joint `body`."""
    frames, joints = link_frames(q)
    R, p = frames[joints[body][2]]
    return p + R @ np.asarray(c, dtype=float)


def test_motor_law_forms(oracle_built):
    pd = DynOracle(1, dyn=dict(kp=4000.0, kd=400.0))
    assert pd.motor_torque(0.3, 1.0, 0.1, -0.5) == 4000.0 * (0.3 - 0.1) + 400.0 * (1.0 - -0.5)     # plain PD, r01 arithmetic
    lim = DynOracle(1, dyn=dict(kp=4000.0, kd=400.0, torque_limit=500.0))
    assert lim.motor_torque(0.3, 1.0, 0.1, -0.5) == 500.0 and lim.motor_torque(-0.3, 0.0, 0.1, 0.0) == -500.0
    cap = DynOracle(1, dyn=dict(kp=4000.0, kd=400.0, max_velocity=1e9))
    assert abs(cap.motor_torque(0.3, 1.0, 0.1, -0.5) - pd.motor_torque(0.3, 1.0, 0.1, -0.5)) < 1e-9  # inactive cap == the PD
    cap = DynOracle(1, dyn=dict(kp=4000.0, kd=400.0, max_velocity=0.5))
    # asked-for velocity 1.0 + 10 * 0.2 = 3.0 -> capped at 0.5: tau = kd (0.5 - qd)
    assert abs(cap.motor_torque(0.3, 1.0, 0.1, -0.5) - 400.0 * (0.5 - -0.5)) < 1e-12
    vel = DynOracle(1, dyn=dict(kp=4000.0, kd=400.0, control_mode=1))
    assert vel.motor_torque(123.0, 1.0, 0.1, -0.5) == 400.0 * (1.0 - -0.5)                          # the position target is ignored


def test_max_velocity_caps_the_joint_speed(oracle_built):
    """A 2 rad position step on the wrist joint: the PD motor overshoots 4 rad/s on the way, the capped one never asks
    for more than 1 rad/s and still arrives."""
    peak = {}
    for name, vmax in (("pd", 0.0), ("capped", 1.0)):
        o = DynOracle(1, dyn=dict(kp=4000.0, kd=400.0, max_velocity=vmax))
        set_state(o, np.zeros(6), np.zeros(6))
        r = np.zeros(6); r[5] = 2.0
        speeds = []
        for _ in range(1200):
            o.substep(r, np.zeros(6))
            speeds.append(abs(o.dstate["qd"][0][5]))
        peak[name] = max(speeds)
        assert abs(o.dstate["q"][0][5] - 2.0) < 2e-2
    assert peak["pd"] > 4.0 and peak["capped"] <= 1.0 + 1e-3     # the motor never ASKS for more; coupling adds a hair


def test_velocity_control_tracks_the_commanded_velocity(oracle_built):
    o = DynOracle(1, dyn=dict(kp=4000.0, kd=400.0, control_mode=1))
    set_state(o, np.zeros(6), np.zeros(6))
    v = np.array([0.2, 0.0, 0.0, 0.5, 0.0, -0.7])
    for _ in range(480):                                     # 2 s (before the wrist joints reach their limits)
        o.substep(np.full(6, 99.0), v)                       # r is ignored in this mode
    err = np.abs(o.dstate["qd"][0] - v)
    assert err[1:].max() < 2e-2 and err[0] < 8e-2            # the heavy base joint's time constant I / kd is ~1 s


def test_contact_samples_cover_the_links(oracle_built):
    o = DynOracle(1, dyn=dict(link_contacts=1))
    smp = o.contact_samples()
    assert len(smp) == 23 and sorted({b for b, _, _ in smp}) == [1, 2, 3, 4, 5]
    assert np.allclose(smp[-1][1], [3.6, 0.0, 1.9]) and smp[-1][2] < 0          # the last sample is the pointer sphere
    # neighbours on one link are never further apart than their diameter + the demo obstacle's width (no tunnelling)
    for (b0, c0, r0), (b1, c1, r1) in zip(smp[:-1], smp[1:]):
        if b0 == b1:
            assert np.linalg.norm(c1 - c0) <= 2 * abs(r0 if r0 > 0 else 0.2) + 1.0 + 1e-9
    assert len(DynOracle(1).contact_samples()) == 1                              # default: the pointer alone


def test_link_contact_wrenches_equal_generalised_forces(oracle_built):
    """ABA with the per-body contact wrenches == ABA with tau + sum_k J_k^T F_k (Jacobians of the sample points by
    finite differences of an independent FK): the force transformation into body coordinates is right."""
    rng = np.random.RandomState(5)
    o = DynOracle(1, dyn=dict(link_contacts=1, ground_z=9.0, contact_kp=2000.0, contact_kd=50.0, gravity=9.81,
                              obstacle_position=(8.0, 1.0, 6.0), obstacle_half_extents=(2.0, 2.0, 2.0)))
    hits = 0
    for trial in range(40):
        q = rng.uniform(-1.0, 1.0, 6); qd = rng.uniform(-1.0, 1.0, 6); tau = rng.uniform(-50, 50, 6)
        set_state(o, q, qd)
        active, fext = o.contact_wrenches()
        if not active:
            continue
        hits += 1
        # generalised force of the same contacts: world forces recomputed here from the oracle's own force law per sample
        import ctypes as C
        Q = np.zeros(6)
        for b, c, rad in o.contact_samples():
            pos = body_point_world(q, b, c)
            J = np.zeros((3, 6))
            for k in range(6):
                e = np.zeros(6); e[k] = 1e-6
                J[:, k] = (body_point_world(q + e, b, c) - body_point_world(q - e, b, c)) / 2e-6
            vel = J @ qd
            f = np.zeros(3)
            radius = o.d.pointer_radius if rad < 0 else rad
            # plane
            depth = 9.0 - pos[2] + radius
            if depth > 0:
                fz = 2000.0 * depth - 50.0 * vel[2]
                if fz > 0:
                    f[2] += fz
            # box (outside or inside)
            dd = pos - np.array([8.0, 1.0, 6.0]); qq = np.abs(dd) - 2.0; oo = np.maximum(qq, 0)
            if (oo > 0).any():
                sdf = np.linalg.norm(oo); nrm = np.sign(dd) * oo / sdf
            else:
                km = int(np.argmax(qq)); sdf = qq[km]; nrm = np.zeros(3); nrm[km] = 1.0 if dd[km] >= 0 else -1.0
            dep = radius - sdf
            if dep > 0:
                fn = 2000.0 * dep - 50.0 * float(vel @ nrm)
                if fn > 0:
                    f += fn * nrm
            Q += J.T @ f
        a = o.aba_ext(tau, 9.81, fext)
        b_ = o.aba_ext(tau + Q, 9.81, None)
        assert np.allclose(a, b_, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(b_).max())), (trial, a, b_)
    assert hits >= 10


def test_an_arm_link_rests_on_the_ground_plane(oracle_built):
    """Released horizontally above a plane that only the LINK samples can reach first (the pointer ends higher): without
    link contacts arm2 sinks through, with them it is held."""
    z = np.zeros(6)
    low = {}
    for name, links in (("pointer_only", 0), ("links", 1)):
        o = DynOracle(1, dyn=dict(teleport=1, gravity=9.81, ground_z=12.5, contact_kp=4000.0, contact_kd=100.0, link_contacts=links))
        q0 = np.zeros(6); q0[4] = -1.2                       # wrist pitched up: the pointer sits well above arm2
        set_state(o, q0, np.zeros(6))
        zmin = 1e9
        for _ in range(1500):
            o.substep(z, z)
            zmin = min(zmin, body_point_world(o.dstate["q"][0], 2, [9.0, 1.0, 0.0])[2])      # far end of arm2
        low[name] = zmin
    assert low["links"] > 12.5 - 0.7 - 0.6                   # held near the plane (sphere radius 0.7 + bounded penetration)
    assert low["pointer_only"] < low["links"] - 1.0


def _wrench(dyn, q, qd=None):
    o = DynOracle(1, seed=0, dyn=dyn)
    o.reset(joint_pos=np.asarray(q, float)[None])
    o.dstate["q"][0] = q
    o.dstate["qd"][0] = 0.0 if qd is None else qd
    return o.contact_wrenches(0)


def test_scene_bodies_reduce_to_the_legacy_plane_and_box_and_rotate_consistently():
    """The static scene bodies (create_body_plane / _box / _sphere, bullet_scene.py:193-228) against what is already
    validated: a z-normal plane and an identity-orientation box ARE the legacy ground / obstacle (same wrenches);
    turning a box by 90 degrees about z equals swapping its x / y half extents; a plane given by a rotated body-frame
    normal equals the plane given by the rotated normal directly."""
    rng = np.random.RandomState(3)
    h = math.sqrt(0.5)
    for _ in range(20):
        q = rng.uniform(-1.2, 1.2, 6)
        qd = rng.uniform(-1, 1, 6)
        # plane (with link contacts the legacy ground also touches with the surface)
        a = _wrench(dict(ground_z=6.0, link_contacts=1), q, qd)
        b = _wrench(dict(link_contacts=1, scene=[("plane", (3.0, -2.0, 6.0), (0, 0, 0, 1), (0, 0, 5.0))]), q, qd)
        assert a[0] == b[0] and np.allclose(a[1], b[1], rtol=1e-12, atol=1e-12)
        # box
        box = dict(obstacle_position=(12.0, 0.0, 4.0), obstacle_half_extents=(4.0, 6.0, 3.0))
        a = _wrench(dict(link_contacts=1, **box), q, qd)
        b = _wrench(dict(link_contacts=1, scene=[("box", (12.0, 0.0, 4.0), (0, 0, 0, 1), (4.0, 6.0, 3.0))]), q, qd)
        c = _wrench(dict(link_contacts=1, scene=[("box", (12.0, 0.0, 4.0), (0, 0, h, h), (6.0, 4.0, 3.0))]), q, qd)
        assert a[0] == b[0] == c[0]
        assert np.allclose(a[1], b[1], rtol=1e-12, atol=1e-12) and np.allclose(a[1], c[1], rtol=1e-9, atol=1e-9)
        # a plane's normal lives in the body frame: normal x turned 90 degrees about y -> -z ... check with +z from -x
        a = _wrench(dict(link_contacts=1, scene=[("plane", (0, 0, 6.0), (0, 0, 0, 1), (0, 0, 1.0))]), q, qd)
        b = _wrench(dict(link_contacts=1, scene=[("plane", (0, 0, 6.0), (0, h, 0, h), (-1.0, 0, 0))]), q, qd)   # R_y(90) (-x) = +z
        assert a[0] == b[0] and np.allclose(a[1], b[1], rtol=1e-9, atol=1e-9)


def test_scene_sphere_pushes_the_pointer_radially():
    """A static sphere around the resting pointer: force on body 6 = kp * depth along (pointer - centre), zero torque about
    the pointer; outside reach nothing happens."""
    q = np.zeros(6)
    o = DynOracle(1, seed=0, dyn=dict())
    o.reset(joint_pos=q[None])
    tip = o.observe()[0, 126:129].copy() if hasattr(o, "observe") else None
    assert tip is not None
    centre = tip + np.array([0.3, -0.2, 0.1])
    R, r = 1.0, 0.2
    any_, f = _wrench(dict(scene=[("sphere", tuple(centre), (0, 0, 0, 1), (R, 0, 0))]), q)
    assert any_
    d = tip - centre
    depth = r + R - np.linalg.norm(d)
    want = 2000.0 * depth * d / np.linalg.norm(d)
    # the wrench is in body 6's frame; at q = 0 every body frame is parallel to the world's
    assert np.allclose(f[5, 3:6], want, rtol=1e-9, atol=1e-9)
    any_, f = _wrench(dict(scene=[("sphere", tuple(tip + 10.0), (0, 0, 0, 1), (R, 0, 0))]), q)
    assert not any_ and not f.any()


def test_nominal_joint_inertias_are_the_mass_matrix_diagonal_at_the_zero_pose(oracle_built):
    """A yardstick for motor gains: J_i = M_ii(q = 0) of the independent per-link Lagrangian formulation above; 6.6 on the
    wrist, 1 477 on the shoulder — why one pair of torque gains cannot serve all six joints."""
    o = DynOracle(1, seed=0)
    J = o.nominal_inertia()
    M = mass_matrix(np.zeros(6), np.ones(11))
    assert np.allclose(J, np.diag(M), rtol=1e-12)
    assert np.allclose(J, [592.16, 1476.57, 586.77, 9.61, 20.57, 6.61], atol=5e-3)


def test_inertia_scaled_motor_is_an_acceleration_request(oracle_built):
    """pd_inertia_scaled: joint i gets the torque D_i a_i with a_i = kp (r - q) + kd (v - qd) and D_i its articulated-body
    inertia.  With the other joints' requests at zero and no velocity, the ABA then returns qdd_i = a_i up to the coupling
    through the parent's acceleration — exactly a_i for the base joint, whose parent is the fixed world."""
    import ctypes as C
    rng = np.random.RandomState(2)
    o = DynOracle(1, seed=0, dyn=dict(kp=400.0, kd=40.0, pd_inertia_scaled=1))
    L = o.lib
    for _ in range(10):
        q = rng.uniform(-1.2, 1.2, 6)
        set_state(o, q, np.zeros(6))
        ades = np.zeros(6); ades[0] = rng.uniform(0.5, 5) * rng.choice([-1.0, 1.0])
        qdd = np.zeros(6)
        L.orc_dyn_aba_motor(o._one(0), _ptr(np.zeros(6), C.c_double), _ptr(ades, C.c_double), C.c_double(0.0), C.c_double(0.0), None,
                            _ptr(qdd, C.c_double))
        assert abs(qdd[0] - ades[0]) < 1e-9 * abs(ades[0])
        # and the cap acts on the torque D_i a_i
        qdd_c = np.zeros(6)
        L.orc_dyn_aba_motor(o._one(0), _ptr(np.zeros(6), C.c_double), _ptr(ades, C.c_double), C.c_double(1e-3), C.c_double(0.0), None,
                            _ptr(qdd_c, C.c_double))
        assert abs(qdd_c[0]) < abs(qdd[0]) * 1e-2


def test_inertia_scaled_gains_track_every_joint_alike(oracle_built):
    """omega = 20 rad/s, zeta = 1 on every joint: a step in the command is followed within half a second by the heavy base
    joints and the light wrist alike (cross-coupling leaves a few hundredths of a radian of ringing); with the plain gains
    4000 / 400 the base joints have covered less than two thirds of the way by then."""
    r = np.array([0.5, 0.3, -0.3, 0.5, 0.4, -0.5])
    err = {}
    for name, dyn in (("scaled", dict(kp=400.0, kd=40.0, pd_inertia_scaled=1)), ("plain", dict(kp=4000.0, kd=400.0))):
        o = DynOracle(1, seed=0, dyn=dyn)
        o.reset(joint_pos=np.zeros((1, 6)))
        o.state["r"][0] = r.astype(o.state["r"].dtype); o.state["v"][0] = 0.0
        for _ in range(12):                                    # twelve env steps = 0.5 s, zero action: the command stays put
            o.step(np.zeros((1, 6), np.float32))
        err[name] = np.abs(o.dstate["q"][0] - o.state["r"][0])
    assert err["scaled"].max() < 0.03, err
    assert err["plain"][:3].max() > 0.1, err


# ---- World.step() alone and the per-joint motors (r05: orc_dyn_world_step; include/pioneer_amd.h pnr_world_step, pnr_set_joint_motor) ------
def test_world_step_is_frame_skip_substeps_on_the_commanded_state(oracle_built):
    """Without per-joint motors World.step() (bullet_scene.py:273-275) is frame_skip sub-steps of the env-wide law tracking the env's own
    command state r, v: the same as calling orc_dyn_substep frame_skip times; it touches neither the command state nor the counters."""
    kw = dict(gravity=9.81, kp=4000.0, kd=400.0, torque_limit=900.0, joint_damping=0.3, joint_friction=0.05, ground_z=0.0, contact_kp=2000.0, contact_kd=50.0)
    a, b = DynOracle(1, seed=4, dyn=kw), DynOracle(1, seed=4, dyn=kw)
    a.reset(); b.reset()
    for o in (a, b):
        o.dstate["q"][0] += 0.05; o.dstate["qd"][0] = np.linspace(-0.4, 0.4, 6)
    before = a.state.copy()
    a.world_step()
    for _ in range(b.d.frame_skip):
        b.substep(b.state["r"][0], b.state["v"][0].astype(np.float64))
    assert np.array_equal(a.dstate["q"], b.dstate["q"]) and np.array_equal(a.dstate["qd"], b.dstate["qd"])
    assert a.state.tobytes() == before.tobytes()
    assert not np.array_equal(a.dstate["q"][0], a.state["r"][0])            # it did move: gravity and the motor pulled


def test_world_step_on_a_teleport_handle_coasts(oracle_built):
    """teleport = the reference's semantics (no motor torque): with no gravity, friction or contacts a joint reset with a velocity — the
    demo's reset_state(position(), velocity), pioneer_knm_env.py:282 — moves the whole coupled chain, and the kinetic energy is conserved."""
    o = DynOracle(1, seed=2, dyn=dict(teleport=1))
    o.reset()
    o.dstate["q"][0] = [0.2, 0.3, -0.4, 0.1, 0.2, 0.0]; o.dstate["qd"][0] = [0.0, 1.0, 0.0, 0.0, 0.0, 0.0]
    e0 = sum(o.energy())
    for _ in range(5):
        o.world_step()
    assert abs(sum(o.energy()) - e0) < 2e-3 * e0 and o.dstate["q"][0][1] > 0.3 + 0.15
    assert np.abs(o.dstate["qd"][0][[0, 2, 3, 4, 5]]).max() > 1e-3          # the other joints are dragged along (no motors hold them)


def test_per_joint_motors_of_world_step(oracle_built):
    """Joint.control_velocity / control_position (bullet_scene.py:123-155) as per-joint motors of World.step(): in zero gravity a velocity
    motor reaches its target velocity, a position motor with maxVelocity travels no faster than that and arrives, max_force bounds the
    joint's acceleration, and a joint nobody commanded keeps tracking the env's command state."""
    o = DynOracle(1, seed=1, dyn=dict(kp=4000.0, kd=400.0))
    o.reset()
    r0 = o.state["r"][0].copy()
    o.set_joint_motor(5, 1, target_velocity=0.7)                            # the wrist: light, converges fast
    o.set_joint_motor(3, 0, target_position=float(r0[3]) + 0.8, max_velocity=0.5)
    peak = 0.0
    for _ in range(60):
        o.world_step()
        peak = max(peak, abs(o.dstate["qd"][0][3]))
    assert abs(o.dstate["qd"][0][5] - 0.7) < 0.02
    assert peak < 0.5 * 1.1 and abs(o.dstate["q"][0][3] - (r0[3] + 0.8)) < 0.02
    assert np.abs(o.dstate["q"][0][[0, 1, 2, 4]] - r0[[0, 1, 2, 4]]).max() < 0.02          # the others hold the command state
    weak = DynOracle(1, seed=1, dyn=dict(kp=4000.0, kd=400.0)); weak.reset()
    strong = DynOracle(1, seed=1, dyn=dict(kp=4000.0, kd=400.0)); strong.reset()
    weak.set_joint_motor(0, 1, target_velocity=1.0, max_force=5.0); strong.set_joint_motor(0, 1, target_velocity=1.0, max_force=5000.0)
    weak.world_step(); strong.world_step()
    assert 0 < weak.dstate["qd"][0][0] < 0.2 * strong.dstate["qd"][0][0]
