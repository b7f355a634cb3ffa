"""Pure-NumPy twin of the C oracle — TEST INFRASTRUCTURE ONLY (see pnr_oracle.h).

An independent second restatement of the same reference lines, written the way
the reference is written (one env, NumPy scalars), used to cross-check
``pnr_oracle.c`` at small sizes.  The reference ran under NumPy 1.x promotion
(np.float32 scalar (op) Python float -> float64); NumPy 2 (NEP 50) would keep
float32, so every such operation spells its float64 cast out.

FK here is the *generic* URDF composition (4x4 homogeneous transforms built
from a joint table), not the closed form the C oracle uses, so the two FKs
check each other.
"""
import numpy as np

f32 = np.float32
f64 = np.float64

# (type, origin xyz, axis) for the 11 joints of assets/pioneer_knm_6dof.urdf:204-275
# in chain order; fixed joints without <origin> are identity.
CHAIN = [
    ("fixed", (0, 0, 0), None),            # world_to_base               :204-207
    ("revolute", (0, 0, 0), (0, 0, 1)),    # robot:base_to_rotator1      :209-214
    ("fixed", (0, 0, 0), None),            # robot:rotator1_to_hinge1    :216-219
    ("revolute", (0, 0, 3), (0, 1, 0)),    # robot:hinge1_to_arm1        :221-227
    ("revolute", (0, 0, 11), (0, 1, 0)),   # robot:arm1_to_arm2          :229-235
    ("revolute", (0, 1, 0), (1, 0, 0)),    # robot:arm2_to_rotator2      :237-243
    ("fixed", (0, 0, 0), None),            # robot:rotator2_to_hinge2    :245-248
    ("revolute", (11, 0, 0), (0, 1, 0)),   # robot:hinge2_to_arm3        :250-256
    ("revolute", (0, 0, 0), (1, 0, 0)),    # robot:arm3_to_rotator3      :258-264
    ("fixed", (0, 0, 0), None),            # robot:rotator3_to_effector  :266-269
    ("fixed", (3.6, 0, 1.9), None),        # robot:effector_to_pointer   :271-275
]
LIMITS = (3.1416, 1.309, 1.309, 3.1416, 1.5708, 3.1416)


def _axis_angle(axis, q):
    """Rodrigues rotation matrix about a unit axis."""
    x, y, z = axis
    K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=f64)
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)


def fk_chain(q, chain=CHAIN):
    """World position of the last link's frame origin (generic composition)."""
    T = np.eye(4)
    qi = 0
    for jtype, origin, axis in chain:
        J = np.eye(4)
        J[:3, 3] = origin
        if jtype == "revolute":
            J[:3, :3] = _axis_angle(axis, f64(q[qi]))
            qi += 1
        T = T @ J
    return T[:3, 3].copy()


class TwinEnv:
    """One env, reference-exact precision (ORC_REF).  Mirrors
    PioneerKinematicEnv's attributes: a, v, r, potential, r_lo, r_hi, v_max, a_max, dt, eps."""

    def __init__(self, max_v_to_r=2, max_a_to_v=10, done_distance=0.1, award_max=100.0,
                 award_done=5.0, award_potential_slope=10.0, penalty_step=1 / 100,
                 timestep=1 / 240, frame_skip=10):
        self.cfg = dict(done_distance=done_distance, award_max=award_max, award_done=award_done,
                        award_potential_slope=award_potential_slope, penalty_step=penalty_step)
        self.r_lo = np.array([-x for x in LIMITS], dtype=f32)     # pioneer_knm_env.py:217-220
        self.r_hi = np.array(LIMITS, dtype=f32)
        self.v_max = f32(max_v_to_r) * (self.r_hi - self.r_lo)    # :57
        self.a_max = f32(max_a_to_v) * self.v_max                 # :58
        self.dt = timestep * frame_skip                           # :60
        self.eps = 1e-5                                           # :61
        self.a = self.v = self.r = None
        self.target = None
        self.potential = None
        self.step_index = 0

    def reset_world(self, joint_positions, target_position):      # :76-105
        self.a = np.zeros(6, dtype=f32)
        self.v = np.zeros(6, dtype=f32)
        self.r = np.array(joint_positions, dtype=f64)             # float64 until the first step
        self.target = np.array(target_position, dtype=f64)
        self.potential = 0
        self.step_index = 0
        return self.observe()

    def compute_potential(self, distance):                        # :232-236
        m = self.cfg["award_max"] - self.cfg["award_done"]
        s = self.cfg["award_potential_slope"]
        return m / (distance / s + 1)

    def step(self, action):                                       # bullet_env.py:192-197
        self.step_index += 1
        action = np.asarray(action, dtype=f32)
        a0, v0, r0 = self.a, self.v, self.r                       # :113-115
        v1 = np.zeros(6, dtype=f32)
        r1 = np.zeros(6, dtype=f32)
        dt = f64(self.dt)
        for i in range(6):                                        # :120-141
            v1[i] = f64(v0[i]) + f64(a0[i]) * dt
            dt_p1 = dt
            dt_p2 = f64(0)
            if v1[i] > self.v_max[i]:
                num = f32(self.v_max[i] - v0[i])
                dt_p1 = np.clip(f64(num) / (f64(a0[i]) + f64(self.eps)), 0, dt)
                dt_p2 = dt - dt_p1
                v1[i] = self.v_max[i]
            elif v1[i] < -self.v_max[i]:
                num = f32(-self.v_max[i] - v0[i])
                dt_p1 = np.clip(f64(num) / (f64(a0[i]) + f64(self.eps)), 0, dt)
                dt_p2 = dt - dt_p1
                v1[i] = -self.v_max[i]
            vs = f32(f32(v0[i]) + f32(v1[i]))
            r1[i] = f64(r0[i]) + f64(0.5) * f64(vs) * dt_p1 + f64(v1[i]) * dt_p2
            if r1[i] >= self.r_hi[i]:
                r1[i] = self.r_hi[i]
                v1[i] = 0
            if r1[i] <= self.r_lo[i]:
                r1[i] = self.r_lo[i]
                v1[i] = 0
        self.a, self.v, self.r = action, v1, r1                   # :144-146

        pointer = fk_chain(self.r)                                # :151
        diff = self.target - pointer                              # :154
        distance = np.linalg.norm(diff)                           # :155
        old_potential = self.potential
        self.potential = self.compute_potential(distance)         # :158
        done = bool(distance < self.cfg["done_distance"])         # :160
        reward_potential = self.potential - old_potential
        reward_step = -self.cfg["penalty_step"]
        reward_done = self.cfg["award_done"] if done else 0
        reward = reward_potential + reward_step + reward_done     # :165
        info = dict(r_pot=reward_potential, r_step=reward_step, r_done=reward_done, dist=distance)
        return self.observe(), float(reward), done, info

    def observe(self):                                            # :184-211
        pointer = fk_chain(self.r)
        diff = self.target - pointer
        distance = np.linalg.norm(diff)
        r_lo_dist = self.r - self.r_lo
        r_hi_diff = self.r_hi - self.r
        return np.concatenate([
            self.r, np.cos(self.r), np.sin(self.r),
            self.r_lo, np.cos(self.r_lo), np.sin(self.r_lo),
            self.r_hi, np.cos(self.r_hi), np.sin(self.r_hi),
            r_lo_dist, np.cos(r_lo_dist), np.sin(r_lo_dist),
            r_hi_diff, np.cos(r_hi_diff), np.sin(r_hi_diff),
            self.v, np.cos(self.v), np.sin(self.v),
            self.a, np.cos(self.a), np.sin(self.a),
            pointer, self.target, diff,
            np.array([distance]), np.array([self.potential]),
        ]).astype(f64)
