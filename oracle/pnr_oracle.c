/*
 * pnr_oracle.c — CPU restatement of the reference's step path (see
 * pnr_oracle.h for the status of this file: TEST INFRASTRUCTURE, PARITY
 * UNPINNED).  Every function cites the reference lines it follows; paths are
 * relative to the reference tree (xdralex/pioneer).
 *
 * Arithmetic model: the reference mixes float32 NumPy arrays with Python
 * floats under NumPy 1.x promotion (scalar float32 (op) Python float ->
 * float64; float32 (op) float32 -> float32; stores into float32 arrays round).
 * Each expression below carries the cast sequence that reproduces it.  Build
 * with -ffp-contract=off so no a*b+c is fused.
 */
#include "pnr_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- model constants: assets/pioneer_knm_6dof.urdf ------------------------ */
/* revolute joints in chain order (bullet_env.py:141-142 keeps revolute only) */
static const double URDF_LIMIT[ORC_DOF] = {
    3.1416, /* robot:base_to_rotator1   urdf:209-214, axis z */
    1.309,  /* robot:hinge1_to_arm1     urdf:221-227, axis y, origin 0 0 3  */
    1.309,  /* robot:arm1_to_arm2       urdf:229-235, axis y, origin 0 0 11 */
    3.1416, /* robot:arm2_to_rotator2   urdf:237-243, axis x, origin 0 1 0  */
    1.5708, /* robot:hinge2_to_arm3     urdf:250-256, axis y, origin 11 0 0 */
    3.1416  /* robot:arm3_to_rotator3   urdf:258-264, axis x, origin 0 0 0  */
};

void orc_params_derive(orc_params* p)
{
    for (int i = 0; i < ORC_DOF; i++) {
        /* joint_limits(): np.array(..., dtype=np.float32), pioneer_knm_env.py:217-220 */
        p->r_lo[i] = (float)(-URDF_LIMIT[i]);
        p->r_hi[i] = (float)(URDF_LIMIT[i]);
        /* :57  python scalar * float32 array -> float32 array */
        float span = p->r_hi[i] - p->r_lo[i];
        p->v_max[i] = (float)p->max_v_to_r * span;
        /* :58 */
        p->a_max[i] = (float)p->max_a_to_v * p->v_max[i];
    }
    /* :60 world.step_time = timestep * frame_skip (bullet_scene.py:277-279) */
    p->dt = p->timestep * (double)p->frame_skip;
    p->eps = 1e-5; /* :61 */
}

void orc_params_default(orc_params* p)
{
    memset(p, 0, sizeof(*p));
    /* pioneer_knm_env.py:19-34 */
    p->max_v_to_r = 2;
    p->max_a_to_v = 10;
    p->done_distance = 0.1;
    p->award_max = 100.0;
    p->award_done = 5.0;
    p->award_potential_slope = 10.0;
    p->penalty_step = 1.0 / 100;
    p->target_lo[0] = 15; p->target_lo[1] = -10; p->target_lo[2] = 2;
    p->target_hi[0] = 25; p->target_hi[1] = 10;  p->target_hi[2] = 6;
    /* bullet_env.py:36-41 */
    p->timestep = 1.0 / 240;
    p->frame_skip = 10;
    p->max_episode_steps = 500; /* pioneer_knm_train.py:27 */
    p->precision = ORC_REF;
    p->auto_reset = 0;
    p->seed = 0;
    orc_params_derive(p);
}

/* ---- forward kinematics ---------------------------------------------------- */
/* Position of link robot:pointer (urdf:271-275 origin 3.6 0 1.9 on
 * robot:effector) in the world frame; fixed joints are identity transforms
 * except that last one.  Right-handed rotations about the joint axes. */
void orc_fk_pointer(const double q[ORC_DOF], double out[3])
{
    double c[ORC_DOF], s[ORC_DOF];
    for (int i = 0; i < ORC_DOF; i++) { c[i] = cos(q[i]); s[i] = sin(q[i]); }
    double x, y, z, t;
    /* Rx(q6) . (3.6, 0, 1.9) */
    x = 3.6; y = -s[5] * 1.9; z = c[5] * 1.9;
    /* Ry(q5) */
    t = c[4] * x + s[4] * z; z = -s[4] * x + c[4] * z; x = t;
    x += 11.0;                                  /* urdf:254 */
    /* Rx(q4) */
    t = c[3] * y - s[3] * z; z = s[3] * y + c[3] * z; y = t;
    y += 1.0;                                   /* urdf:241 */
    /* Ry(q3) */
    t = c[2] * x + s[2] * z; z = -s[2] * x + c[2] * z; x = t;
    z += 11.0;                                  /* urdf:233 */
    /* Ry(q2) */
    t = c[1] * x + s[1] * z; z = -s[1] * x + c[1] * z; x = t;
    z += 3.0;                                   /* urdf:225 */
    /* Rz(q1) */
    t = c[0] * x - s[0] * y; y = s[0] * x + c[0] * y; x = t;
    out[0] = x; out[1] = y; out[2] = z;
}

/* pioneer_knm_env.py:232-236 */
double orc_potential(const orc_params* p, double distance)
{
    double m = p->award_max - p->award_done;
    double s = p->award_potential_slope;
    return m / (distance / s + 1);
}

/* ---- Philox4x32-10 ----------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4])
{
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 24-bit uniform in [0,1): exact in float32 and float64 */
static double u01(uint32_t x) { return (double)(x >> 8) * (1.0 / 16777216.0); }

/* uniform draw w of Philox block b (4 words per block) for (seed, env, episode) */
void orc_draw_block(const orc_params* p, uint64_t genv, uint32_t episode, uint32_t block, double u[4])
{
    uint32_t key[2] = { (uint32_t)p->seed, (uint32_t)(p->seed >> 32) };
    uint32_t ctr[4] = { episode, (uint32_t)genv, (uint32_t)(genv >> 32), block };
    uint32_t w[4];
    orc_philox4x32_10(ctr, key, w);
    for (int i = 0; i < 4; i++) u[i] = u01(w[i]);
}

static void draw9(const orc_params* p, uint64_t genv, uint32_t episode, double u[9])
{
    uint32_t key[2] = { (uint32_t)p->seed, (uint32_t)(p->seed >> 32) };
    uint32_t w[12];
    for (uint32_t b = 0; b < 3; b++) {
        uint32_t ctr[4] = { episode, (uint32_t)genv, (uint32_t)(genv >> 32), b };
        orc_philox4x32_10(ctr, key, w + 4 * b);
    }
    for (int i = 0; i < 9; i++) u[i] = u01(w[i]);
}

/* ---- reset -------------------------------------------------------------------- */
/* reset_world, pioneer_knm_env.py:76-105 (+ BulletEnv.reset bullet_env.py:187-190:
 * reset_simulator sets step_index = 0, :99) */
void orc_reset(const orc_params* p, orc_state* s, uint64_t genv,
               const double* joint_pos, const double* target_pos)
{
    double u[9];
    draw9(p, genv, s->episode, u);
    for (int i = 0; i < ORC_DOF; i++) {
        /* :80-81 np_random.uniform(r_lo, r_hi): lo + (hi-lo)*u in float64 */
        double lo = (double)p->r_lo[i], hi = (double)p->r_hi[i];
        double r = joint_pos ? joint_pos[i] : lo + (hi - lo) * u[i];
        s->r[i] = (p->precision == ORC_DEV) ? (double)(float)r : r; /* :94 */
        s->a[i] = 0.0f; /* :92 */
        s->v[i] = 0.0f; /* :93 */
    }
    for (int k = 0; k < 3; k++) {
        /* :83-90 */
        double lo = p->target_lo[k], hi = p->target_hi[k];
        double t = target_pos ? target_pos[k] : lo + (hi - lo) * u[6 + k];
        s->target[k] = (p->precision == ORC_DEV) ? (double)(float)t : t;
    }
    s->r_is_f64 = (p->precision == ORC_REF);
    s->potential = 0.0; /* :105 (quirk Q3: not potential(distance)) */
    s->step_index = 0;  /* bullet_env.py:99 */
    s->episode += 1;
}

/* ---- observe ------------------------------------------------------------------ */
/* np.cos / np.sin of a float32 array return float32 (rounded); of a float64
 * array, float64. */
static void put_triple(double* dst, const double x[ORC_DOF], int as_f32)
{
    for (int i = 0; i < ORC_DOF; i++) {
        dst[i] = x[i];
        dst[6 + i] = as_f32 ? (double)(float)cos(x[i]) : cos(x[i]);
        dst[12 + i] = as_f32 ? (double)(float)sin(x[i]) : sin(x[i]);
    }
}

/* observe, pioneer_knm_env.py:184-211, on explicit joint positions / velocities
 * (kinematic mode: q = s->r, qd = s->v; dynamics mode: the simulated q, qd) */
void orc_observe_qv(const orc_params* p, const orc_state* s, const double q[ORC_DOF],
                    const double qd[ORC_DOF], int q_is_f64, double obs[ORC_OBS])
{
    double pointer[3], diff[3];
    orc_fk_pointer(q, pointer);                        /* :185, bullet_scene.py:58 */
    for (int k = 0; k < 3; k++) diff[k] = s->target[k] - pointer[k]; /* :188 */
    double distance = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]); /* :189 */

    int f32 = !q_is_f64;
    double x[ORC_DOF];
    /* :195 r, cos r, sin r */
    put_triple(obs + 0, q, f32);
    /* :196-197 limits (always float32 arrays) */
    for (int i = 0; i < ORC_DOF; i++) x[i] = (double)p->r_lo[i];
    put_triple(obs + 18, x, 1);
    for (int i = 0; i < ORC_DOF; i++) x[i] = (double)p->r_hi[i];
    put_triple(obs + 36, x, 1);
    /* :191, :199  r - r_lo : float32 - float32 -> float32; float64 - float32 -> float64 */
    for (int i = 0; i < ORC_DOF; i++)
        x[i] = f32 ? (double)((float)q[i] - p->r_lo[i]) : q[i] - (double)p->r_lo[i];
    put_triple(obs + 54, x, f32);
    /* :192, :200  r_hi - r */
    for (int i = 0; i < ORC_DOF; i++)
        x[i] = f32 ? (double)(p->r_hi[i] - (float)q[i]) : (double)p->r_hi[i] - q[i];
    put_triple(obs + 72, x, f32);
    /* :202-203 v, a */
    put_triple(obs + 90, qd, 1);
    for (int i = 0; i < ORC_DOF; i++) x[i] = (double)s->a[i];
    put_triple(obs + 108, x, 1);
    /* :205-210 */
    for (int k = 0; k < 3; k++) {
        obs[126 + k] = pointer[k];
        obs[129 + k] = s->target[k];
        obs[132 + k] = diff[k];
    }
    obs[135] = distance;
    obs[136] = s->potential;
}

void orc_observe(const orc_params* p, const orc_state* s, double obs[ORC_OBS])
{
    double qd[ORC_DOF];
    for (int i = 0; i < ORC_DOF; i++) qd[i] = (double)s->v[i];
    orc_observe_qv(p, s, s->r, qd, s->r_is_f64, obs);
}

/* ---- step --------------------------------------------------------------------- */
static double np_clip(double x, double lo, double hi)
{
    if (x != x) return x; /* np.clip propagates NaN */
    return x < lo ? lo : (x > hi ? hi : x);
}

/* the integrator loop of act(), pioneer_knm_env.py:113-146 (+ step_index, bullet_env.py:193) */
void orc_integrate(const orc_params* p, orc_state* s, const float action[ORC_DOF])
{
    s->step_index += 1; /* bullet_env.py:193 */

    const double dt = p->dt, eps = p->eps;
    for (int i = 0; i < ORC_DOF; i++) {      /* :120 */
        const float a0 = s->a[i], v0 = s->v[i];
        const double r0 = s->r[i];           /* float32-valued except right after reset */
        const float vmax = p->v_max[i];
        /* :121  v1[i] = v0[i] + a0[i] * self.dt   (f32*pyfloat -> f64; f32+f64 -> f64; store f32) */
        float v1 = (float)((double)v0 + (double)a0 * dt);
        double dt_p1 = dt;                   /* :122 */
        double dt_p2 = 0.0;                  /* :123 */
        if (v1 > vmax) {                     /* :125 */
            float num = vmax - v0;           /* f32 - f32 -> f32 */
            dt_p1 = np_clip((double)num / ((double)a0 + eps), 0.0, dt); /* :126 */
            dt_p2 = dt - dt_p1;              /* :127 */
            v1 = vmax;                       /* :128 */
        } else if (v1 < -vmax) {             /* :129 */
            float num = -vmax - v0;
            dt_p1 = np_clip((double)num / ((double)a0 + eps), 0.0, dt); /* :130 */
            dt_p2 = dt - dt_p1;              /* :131 */
            v1 = -vmax;                      /* :132 */
        }
        /* :134  r1[i] = r0[i] + 0.5 * (v0[i] + v1[i]) * dt_p1 + v1[i] * dt_p2 */
        float vs = v0 + v1;                  /* f32 + f32 -> f32 */
        double half = 0.5 * (double)vs;
        double r1d = (r0 + half * dt_p1) + (double)v1 * dt_p2;
        float r1 = (float)r1d;               /* store into float32 array */
        if (r1 >= p->r_hi[i]) { r1 = p->r_hi[i]; v1 = 0.0f; } /* :135-137 */
        if (r1 <= p->r_lo[i]) { r1 = p->r_lo[i]; v1 = 0.0f; } /* :139-141 */
        s->v[i] = v1;                        /* :145 */
        s->r[i] = (double)r1;                /* :146 */
    }
    for (int i = 0; i < ORC_DOF; i++) s->a[i] = action[i]; /* :144 (quirk Q1) */
    s->r_is_f64 = 0;
}

/* the reward block of act(), pioneer_knm_env.py:148-165, + TimeLimit, for joint positions q.
 * Returns done | truncated<<1. */
int orc_reward(const orc_params* p, orc_state* s, const double q[ORC_DOF], double* reward, double info[4])
{
    /* :148-155 teleport joints, FK, distance */
    double pointer[3], diff[3];
    orc_fk_pointer(q, pointer);
    for (int k = 0; k < 3; k++) diff[k] = s->target[k] - pointer[k];
    double distance = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);

    /* :157-165 */
    double old_potential = s->potential;
    double pot = orc_potential(p, distance);
    int done = distance < p->done_distance;
    double reward_potential = pot - old_potential;
    double reward_step = -p->penalty_step;
    double reward_done = done ? p->award_done : 0.0;
    double rw = reward_potential + reward_step + reward_done;
    s->potential = (p->precision == ORC_DEV) ? (double)(float)pot : pot;

    /* gym.wrappers.TimeLimit.step: truncated = elapsed >= max and not done */
    int trunc = (p->max_episode_steps > 0) &&
                (s->step_index >= (uint32_t)p->max_episode_steps) && !done;
    if (reward) *reward = rw;
    if (info) { info[0] = reward_potential; info[1] = reward_step; info[2] = reward_done; info[3] = distance; }
    return done | (trunc << 1);
}

/* act + observe: BulletEnv.step, bullet_env.py:192-197 */
void orc_step(const orc_params* p, orc_state* s, uint64_t genv,
              const float action[ORC_DOF], double obs[ORC_OBS], double* reward,
              uint8_t* done_out, uint8_t* trunc_out, double info[4])
{
    orc_integrate(p, s, action);
    int flags = orc_reward(p, s, s->r, reward, info);
    int done = flags & 1, trunc = (flags >> 1) & 1;

    /* :181 world.step(): 10 x stepSimulation with gravity 0, zero joint velocity,
     * no colliders -> identity on the observable state */

    if (done_out) *done_out = (uint8_t)done;
    if (trunc_out) *trunc_out = (uint8_t)trunc;

    if (p->auto_reset && (done || trunc)) orc_reset(p, s, genv, NULL, NULL);
    if (obs) orc_observe(p, s, obs);         /* bullet_env.py:196 */
}

/* ---- batches ------------------------------------------------------------------ */
void orc_reset_batch(const orc_params* p, orc_state* s, int64_t n, int64_t off,
                     const uint8_t* mask, const double* joint_pos, const double* target_pos,
                     double* obs, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < n; e++) {
        if (mask && !mask[e]) continue;
        orc_reset(p, &s[e], (uint64_t)(off + e),
                  joint_pos ? joint_pos + 6 * e : NULL, target_pos ? target_pos + 3 * e : NULL);
        if (obs) orc_observe(p, &s[e], obs + (int64_t)ORC_OBS * e);
    }
}

void orc_step_batch(const orc_params* p, orc_state* s, int64_t n, int64_t off,
                    const float* actions, double* obs, double* reward, uint8_t* done,
                    uint8_t* truncated, double* info, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < n; e++) {
        orc_step(p, &s[e], (uint64_t)(off + e), actions + 6 * e,
                 obs ? obs + (int64_t)ORC_OBS * e : NULL, reward ? reward + e : NULL,
                 done ? done + e : NULL, truncated ? truncated + e : NULL,
                 info ? info + 4 * e : NULL);
    }
}

/* ---- state words (engine interchange format) ----------------------------------- */
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

void orc_state_to_words(const orc_state* s, int64_t n, uint32_t* w)
{
    for (int64_t e = 0; e < n; e++) {
        for (int i = 0; i < 6; i++) {
            w[(0 + i) * n + e] = f2u(s[e].a[i]);
            w[(6 + i) * n + e] = f2u(s[e].v[i]);
            w[(12 + i) * n + e] = f2u((float)s[e].r[i]);
        }
        for (int k = 0; k < 3; k++) w[(18 + k) * n + e] = f2u((float)s[e].target[k]);
        w[21 * n + e] = f2u((float)s[e].potential);
        w[22 * n + e] = s[e].step_index;
        w[23 * n + e] = s[e].episode;
    }
}

void orc_state_from_words(orc_state* s, int64_t n, const uint32_t* w)
{
    for (int64_t e = 0; e < n; e++) {
        for (int i = 0; i < 6; i++) {
            s[e].a[i] = u2f(w[(0 + i) * n + e]);
            s[e].v[i] = u2f(w[(6 + i) * n + e]);
            s[e].r[i] = (double)u2f(w[(12 + i) * n + e]);
        }
        for (int k = 0; k < 3; k++) s[e].target[k] = (double)u2f(w[(18 + k) * n + e]);
        s[e].potential = (double)u2f(w[21 * n + e]);
        s[e].step_index = w[22 * n + e];
        s[e].episode = w[23 * n + e];
        s[e].r_is_f64 = 0;
    }
}

int orc_sizeof_state(void) { return (int)sizeof(orc_state); }
int orc_sizeof_params(void) { return (int)sizeof(orc_params); }
