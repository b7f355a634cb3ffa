/*
 * pnr_dyn_oracle.c — float64 dynamics-mode oracle (see pnr_dyn_oracle.h: TEST
 * INFRASTRUCTURE, PARITY UNPINNED).  Deliberately generic and slow: 6x6 spatial
 * matrices, explicit Pluecker transforms (Featherstone, "Rigid Body Dynamics
 * Algorithms", ch. 2 and Table 7.1), so that it shares no shortcuts with the
 * specialised HIP kernel it checks.
 *
 * Conventions: motion vectors [w; v], force vectors [n; f], all in body
 * coordinates at the body-frame origin; joint i connects body i-1 (body 0 = the
 * fixed base) to body i: translate by o_i in the parent frame, rotate by q_i
 * about a coordinate axis.
 */
#include "pnr_dyn_oracle.h"

#include <math.h>
#include <string.h>

/* ---- model table (assets/pioneer_knm_6dof.urdf, fixed joints merged) -------------------- */
static const int AXIS[ORC_DOF] = {2, 1, 1, 0, 1, 0};                 /* z y y x y x  (urdf:213,226,234,242,255,263) */
static const double ORIGIN[ORC_DOF][3] = {                            /* joint origins in the parent body frame */
    {0, 0, 0}, {0, 0, 3}, {0, 0, 11}, {0, 1, 0}, {11, 0, 0}, {0, 0, 0}};
static const double TIP[3] = {3.6, 0.0, 1.9};                         /* urdf:271-275 */
/* URDF link -> moving body (0-based), -1 = static base; link order:
 * base rotator1 hinge1 arm1 arm2 rotator2 hinge2 arm3 rotator3 effector pointer */
static const int LINK_BODY[ORC_LINKS] = {-1, 0, 0, 1, 2, 3, 3, 4, 5, 5, 5};
static const double FRICTION_EPS = 0.05;                              /* smooth sign(qd) = qd / sqrt(qd^2 + eps^2) */

typedef double vec6[6];
typedef double mat6[6][6];

static void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

/* rotation matrix R(axis, q): child coordinates -> parent coordinates */
static void rot_axis(int axis, double q, double R[3][3])
{
    double c = cos(q), s = sin(q);
    memset(R, 0, 9 * sizeof(double));
    int a = axis, b = (axis + 1) % 3, d = (axis + 2) % 3;
    R[a][a] = 1; R[b][b] = c; R[b][d] = -s; R[d][b] = s; R[d][d] = c;
}

/* 6x6 motion transform parent -> child: X = [E 0; -E rx  E], E = R^T */
static void xform_motion(int axis, double q, const double r[3], mat6 X)
{
    double R[3][3], E[3][3];
    rot_axis(axis, q, R);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) E[i][j] = R[j][i];
    double rx[3][3] = {{0, -r[2], r[1]}, {r[2], 0, -r[0]}, {-r[1], r[0], 0}};
    memset(X, 0, sizeof(mat6));
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        X[i][j] = E[i][j]; X[3 + i][3 + j] = E[i][j];
        double t = 0; for (int k = 0; k < 3; k++) t += E[i][k] * rx[k][j];
        X[3 + i][j] = -t;
    }
}

static void matvec6(const mat6 A, const vec6 x, vec6 y)
{
    for (int i = 0; i < 6; i++) { double t = 0; for (int j = 0; j < 6; j++) t += A[i][j] * x[j]; y[i] = t; }
}
static void matTvec6(const mat6 A, const vec6 x, vec6 y)
{
    for (int i = 0; i < 6; i++) { double t = 0; for (int j = 0; j < 6; j++) t += A[j][i] * x[j]; y[i] = t; }
}
/* v x m (motion) */
static void crm(const vec6 v, const vec6 m, vec6 o)
{
    double a[3], b[3], c[3];
    cross3(v, m, a); cross3(v, m + 3, b); cross3(v + 3, m, c);
    for (int i = 0; i < 3; i++) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
/* v x* f (force) */
static void crf(const vec6 v, const vec6 f, vec6 o)
{
    double a[3], b[3], c[3];
    cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
    for (int i = 0; i < 3; i++) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}

/* spatial inertia of each moving body about its frame origin from the per-link scales */
static void body_inertias(const double scale[ORC_LINKS], mat6 I[ORC_DOF], double mass[ORC_DOF], double h[ORC_DOF][3])
{
    double Ibar[ORC_DOF][3][3];
    memset(Ibar, 0, sizeof(Ibar)); memset(mass, 0, ORC_DOF * sizeof(double)); memset(h, 0, ORC_DOF * 3 * sizeof(double));
    for (int l = 0; l < ORC_LINKS; l++) {
        int b = LINK_BODY[l];
        if (b < 0) continue;
        double m = 1.0 * scale[l], in = 1.0 * scale[l];   /* mass 1, inertia diag(1,1,1) (e.g. urdf:44-47) */
        const double zero[3] = {0, 0, 0};
        const double* c = (l == ORC_LINKS - 1) ? TIP : zero; /* only the pointer link is offset in its body */
        mass[b] += m;
        for (int k = 0; k < 3; k++) h[b][k] += m * c[k];
        double c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            Ibar[b][i][j] += (i == j ? in + m * c2 : 0.0) - m * c[i] * c[j];   /* parallel axis */
    }
    for (int b = 0; b < ORC_DOF; b++) {
        memset(I[b], 0, sizeof(mat6));
        double hx[3][3] = {{0, -h[b][2], h[b][1]}, {h[b][2], 0, -h[b][0]}, {-h[b][1], h[b][0], 0}};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            I[b][i][j] = Ibar[b][i][j];
            I[b][i][3 + j] = hx[i][j];
            I[b][3 + i][j] = hx[j][i];
            I[b][3 + i][3 + j] = (i == j) ? mass[b] : 0.0;
        }
    }
}

/* world pose of every body: R0[i] (body -> world), p0[i] (origin) */
static void world_poses(const double q[ORC_DOF], double R0[ORC_DOF][3][3], double p0[ORC_DOF][3])
{
    double Rp[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, pp[3] = {0, 0, 0};
    for (int i = 0; i < ORC_DOF; i++) {
        double R[3][3];
        rot_axis(AXIS[i], q[i], R);
        for (int a = 0; a < 3; a++) {
            p0[i][a] = pp[a];
            for (int k = 0; k < 3; k++) p0[i][a] += Rp[a][k] * ORIGIN[i][k];
        }
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
            double t = 0; for (int k = 0; k < 3; k++) t += Rp[a][k] * R[k][b];
            R0[i][a][b] = t;
        }
        memcpy(Rp, R0[i], sizeof(Rp)); memcpy(pp, p0[i], sizeof(pp));
    }
}

static void body_velocities(const orc_dyn_state* s, mat6 X[ORC_DOF], vec6 v[ORC_DOF])
{
    vec6 vp = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ORC_DOF; i++) {
        xform_motion(AXIS[i], s->q[i], ORIGIN[i], X[i]);
        matvec6(X[i], vp, v[i]);
        v[i][AXIS[i]] += s->qd[i];
        memcpy(vp, v[i], sizeof(vec6));
    }
}

void orc_dyn_aba(const orc_dyn_state* s, const double tau[ORC_DOF], double gravity,
                 const double f_tip_world[3], double qdd[ORC_DOF])
{
    if (!f_tip_world) { orc_dyn_aba_ext(s, tau, gravity, NULL, qdd); return; }
    /* external force on the pointer, expressed in body-6 coordinates at its origin */
    double fext[ORC_DOF][6], R0[ORC_DOF][3][3], p0[ORC_DOF][3], fb[3], nb[3];
    memset(fext, 0, sizeof(fext));
    world_poses(s->q, R0, p0);
    for (int a = 0; a < 3; a++) { fb[a] = 0; for (int k = 0; k < 3; k++) fb[a] += R0[5][k][a] * f_tip_world[k]; }
    cross3(TIP, fb, nb);
    for (int a = 0; a < 3; a++) { fext[5][a] = nb[a]; fext[5][3 + a] = fb[a]; }
    orc_dyn_aba_ext(s, tau, gravity, (const double (*)[6])fext, qdd);
}

void orc_dyn_aba_ext(const orc_dyn_state* s, const double tau[ORC_DOF], double gravity,
                     const double fext[ORC_DOF][6], double qdd[ORC_DOF])
{
    orc_dyn_aba_motor(s, tau, NULL, 0.0, gravity, fext, qdd);
}

/* The same recursion with an acceleration-level motor term: joint i receives, on top of tau[i], the torque
 * clip(D_i * ades[i], +-tcap) (tcap <= 0: no cap), D_i being the joint's articulated-body inertia at this pose — the
 * number pass 2 forms anyway.  A PD law fed in as ades therefore acts with the same stiffness and damping PER UNIT OF
 * INERTIA on every joint and in every pose (the "pd_inertia_scaled" motor). */
static void aba_motor_caps(const orc_dyn_state* s, const double tau[ORC_DOF], const double ades[ORC_DOF], const double tcaps[ORC_DOF],
                           double gravity, const double fext[ORC_DOF][6], double qdd[ORC_DOF]);

void orc_dyn_aba_motor(const orc_dyn_state* s, const double tau[ORC_DOF], const double ades[ORC_DOF], double tcap, double gravity,
                       const double fext[ORC_DOF][6], double qdd[ORC_DOF])
{
    double caps[ORC_DOF];
    for (int i = 0; i < ORC_DOF; i++) caps[i] = tcap;
    aba_motor_caps(s, tau, ades, caps, gravity, fext, qdd);
}

/* .. with a torque cap per joint (pnr_world_step's per-joint motors) */
static void aba_motor_caps(const orc_dyn_state* s, const double tau[ORC_DOF], const double ades[ORC_DOF], const double tcaps[ORC_DOF],
                           double gravity, const double fext[ORC_DOF][6], double qdd[ORC_DOF])
{
    mat6 I[ORC_DOF], X[ORC_DOF], IA[ORC_DOF];
    vec6 v[ORC_DOF], c[ORC_DOF], pA[ORC_DOF], U[ORC_DOF];
    double mass[ORC_DOF], h[ORC_DOF][3], D[ORC_DOF], u[ORC_DOF];
    body_inertias(s->mass_scale, I, mass, h);
    body_velocities(s, X, v);

    /* pass 1: velocity-product accelerations and bias forces */
    for (int i = 0; i < ORC_DOF; i++) {
        vec6 vj = {0, 0, 0, 0, 0, 0}, Iv;
        vj[AXIS[i]] = s->qd[i];
        crm(v[i], vj, c[i]);
        memcpy(IA[i], I[i], sizeof(mat6));
        matvec6(I[i], v[i], Iv);
        crf(v[i], Iv, pA[i]);
    }
    if (fext)
        for (int i = 0; i < ORC_DOF; i++)
            for (int r = 0; r < 6; r++) pA[i][r] -= fext[i][r];
    /* pass 2: articulated inertias, tip to base */
    for (int i = ORC_DOF - 1; i >= 0; i--) {
        int k = AXIS[i];
        for (int r = 0; r < 6; r++) U[i][r] = IA[i][r][k];
        D[i] = U[i][k];
        double tq = tau[i];
        if (ades) {
            double m = D[i] * ades[i];
            const double tcap = tcaps[i];
            if (tcap > 0) m = m > tcap ? tcap : (m < -tcap ? -tcap : m);
            tq += m;
        }
        u[i] = tq - pA[i][k];
        if (i > 0) {
            mat6 Ia, T; vec6 pa, Iac, t6;
            for (int r = 0; r < 6; r++) for (int cc = 0; cc < 6; cc++) Ia[r][cc] = IA[i][r][cc] - U[i][r] * U[i][cc] / D[i];
            matvec6(Ia, c[i], Iac);
            for (int r = 0; r < 6; r++) pa[r] = pA[i][r] + Iac[r] + U[i][r] * u[i] / D[i];
            /* IA[parent] += X^T Ia X ; pA[parent] += X^T pa */
            for (int r = 0; r < 6; r++) for (int cc = 0; cc < 6; cc++) {
                double t = 0; for (int m = 0; m < 6; m++) t += Ia[r][m] * X[i][m][cc];
                T[r][cc] = t;
            }
            for (int r = 0; r < 6; r++) for (int cc = 0; cc < 6; cc++) {
                double t = 0; for (int m = 0; m < 6; m++) t += X[i][m][r] * T[m][cc];
                IA[i - 1][r][cc] += t;
            }
            matTvec6(X[i], pa, t6);
            for (int r = 0; r < 6; r++) pA[i - 1][r] += t6[r];
        }
    }
    /* pass 3: accelerations, base to tip; gravity as a base acceleration of +g along z */
    vec6 ap = {0, 0, 0, 0, 0, gravity};
    for (int i = 0; i < ORC_DOF; i++) {
        vec6 a;
        matvec6(X[i], ap, a);
        for (int r = 0; r < 6; r++) a[r] += c[i][r];
        double t = 0; for (int r = 0; r < 6; r++) t += U[i][r] * a[r];
        qdd[i] = (u[i] - t) / D[i];
        a[AXIS[i]] += qdd[i];
        memcpy(ap, a, sizeof(vec6));
    }
}

void orc_dyn_energy(const orc_dyn_state* s, double gravity, double* kinetic, double* potential)
{
    mat6 I[ORC_DOF], X[ORC_DOF]; vec6 v[ORC_DOF];
    double mass[ORC_DOF], h[ORC_DOF][3], R0[ORC_DOF][3][3], p0[ORC_DOF][3];
    body_inertias(s->mass_scale, I, mass, h);
    body_velocities(s, X, v);
    world_poses(s->q, R0, p0);
    double ke = 0, pe = 0;
    for (int i = 0; i < ORC_DOF; i++) {
        vec6 Iv; matvec6(I[i], v[i], Iv);
        for (int r = 0; r < 6; r++) ke += 0.5 * v[i][r] * Iv[r];
        double hz = 0; for (int k = 0; k < 3; k++) hz += R0[i][2][k] * h[i][k];
        pe += gravity * (mass[i] * p0[i][2] + hz);
    }
    *kinetic = ke; *potential = pe;
}

void orc_dyn_tip(const orc_dyn_state* s, double pos[3], double vel[3])
{
    mat6 X[ORC_DOF]; vec6 v[ORC_DOF];
    double R0[ORC_DOF][3][3], p0[ORC_DOF][3], wxt[3], vb[3];
    body_velocities(s, X, v);
    world_poses(s->q, R0, p0);
    cross3(v[5], TIP, wxt);
    for (int a = 0; a < 3; a++) vb[a] = v[5][3 + a] + wxt[a];
    for (int a = 0; a < 3; a++) {
        pos[a] = p0[5][a]; vel[a] = 0;
        for (int k = 0; k < 3; k++) { pos[a] += R0[5][a][k] * TIP[k]; vel[a] += R0[5][a][k] * vb[k]; }
    }
}

/* rotation matrix of Bullet's quaternion (x, y, z, w), normalised first */
static void quat_to_matrix(const double qq[4], double R[3][3])
{
    double n = sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
    double x = 0, y = 0, z = 0, w = 1;
    if (n > 0) { x = qq[0] / n; y = qq[1] / n; z = qq[2] / n; w = qq[3] / n; }
    R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - z * w);     R[0][2] = 2 * (x * z + y * w);
    R[1][0] = 2 * (x * y + z * w);     R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - x * w);
    R[2][0] = 2 * (x * z - y * w);     R[2][1] = 2 * (y * z + x * w);     R[2][2] = 1 - 2 * (x * x + y * y);
}

void orc_dyn_nominal_inertia(double J[ORC_DOF])
{
    /* zero pose: every body frame is parallel to the world's (no rpy in the URDF's joint origins); joint i sits at the sum
     * of the origins up to i.  A link of mass 1 and inertia diag(1,1,1) at position c adds 1 + |a x (c - p_i)|^2 about the
     * axis a through p_i (parallel-axis theorem; the isotropic inertia gives 1 about any axis). */
    double p[ORC_DOF][3];
    for (int i = 0; i < ORC_DOF; i++)
        for (int k = 0; k < 3; k++) p[i][k] = (i ? p[i - 1][k] : 0.0) + ORIGIN[i][k];
    for (int i = 0; i < ORC_DOF; i++) {
        J[i] = 0.0;
        for (int l = 0; l < ORC_LINKS; l++) {
            int b = LINK_BODY[l];
            if (b < i) continue;                               /* inboard of joint i (or the static base) */
            double c[3], d[3], x[3], a[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++) c[k] = p[b][k] + ((l == ORC_LINKS - 1) ? TIP[k] : 0.0);
            for (int k = 0; k < 3; k++) d[k] = c[k] - p[i][k];
            a[AXIS[i]] = 1.0;
            cross3(a, d, x);
            J[i] += 1.0 + x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
        }
    }
}

static int contact_force_sphere(const orc_dyn_params* d, const double pos[3], const double vel[3], double radius, double f[3]);

int orc_dyn_contact_force(const orc_dyn_params* d, const double pos[3], const double vel[3], double f[3])
{
    return contact_force_sphere(d, pos, vel, d->pointer_radius, f);
}

/* Contact sample spheres.  The pointer sphere is the reference's only sphere (urdf:190-196); the link samples are capsules
 * fitted to the URDF's VISUAL boxes (arm1 urdf:78-90, arm2 :92-104, rotator2 + hinge2 :106-132, arm3 :134-153,
 * effector :169-188): the reference URDF has no <collision> elements, so this geometry is build-defined. */
static const struct { int body; double a[3], b[3]; int n; double radius; } CAPSULES[5] = {
    {1, {0, 0, 0}, {0, 0, 11}, 8, 0.7},            /* arm1: 12 x 1 x 2 beam along its z */
    {2, {-1, 1, 0}, {9, 1, 0}, 7, 0.7},            /* arm2: 10 x 1 x 2 beam along its x at y = 1 */
    {3, {9, 0, 0}, {11, 0, 0}, 2, 0.7},            /* rotator2 + hinge2 on the roll axis */
    {4, {-0.5, 0, 0}, {2.5, 0, 0}, 3, 0.6},        /* arm3: the two 3 x 0.5 x 1 cheeks */
    {5, {3.6, 0, -0.75}, {3.6, 0, 1.9}, 3, -1.0},  /* the effector's needle; its last sample is the pointer sphere itself */
};

int orc_dyn_contact_samples(const orc_dyn_params* d, orc_contact_sample* out)
{
    if (!d->link_contacts) {
        out[0].body = 5; memcpy(out[0].c, TIP, sizeof(TIP)); out[0].radius = -1.0;
        return 1;
    }
    int k = 0;
    for (int c = 0; c < 5; c++)
        for (int i = 0; i < CAPSULES[c].n; i++, k++) {
            double t = CAPSULES[c].n > 1 ? (double)i / (CAPSULES[c].n - 1) : 0.0;
            out[k].body = CAPSULES[c].body;
            for (int a = 0; a < 3; a++) out[k].c[a] = CAPSULES[c].a[a] + t * (CAPSULES[c].b[a] - CAPSULES[c].a[a]);
            out[k].radius = CAPSULES[c].radius;
        }
    return k;
}

int orc_dyn_contact_wrenches(const orc_dyn_params* d, const orc_dyn_state* s, double fext[ORC_DOF][6])
{
    mat6 X[ORC_DOF]; vec6 v[ORC_DOF];
    double R0[ORC_DOF][3][3], p0[ORC_DOF][3];
    orc_contact_sample smp[ORC_CONTACT_SAMPLES];
    int n = orc_dyn_contact_samples(d, smp), any = 0;
    memset(fext, 0, ORC_DOF * 6 * sizeof(double));
    body_velocities(s, X, v);
    world_poses(s->q, R0, p0);
    for (int k = 0; k < n; k++) {
        const int b = smp[k].body;
        const double* c = smp[k].c;
        double wxc[3], vb[3], pos[3], vel[3], f[3], fb[3], nb[3];
        cross3(v[b], c, wxc);
        for (int a = 0; a < 3; a++) vb[a] = v[b][3 + a] + wxc[a];
        for (int a = 0; a < 3; a++) {
            pos[a] = p0[b][a]; vel[a] = 0;
            for (int m = 0; m < 3; m++) { pos[a] += R0[b][a][m] * c[m]; vel[a] += R0[b][a][m] * vb[m]; }
        }
        if (!contact_force_sphere(d, pos, vel, smp[k].radius < 0 ? d->pointer_radius : smp[k].radius, f)) continue;
        any = 1;
        for (int a = 0; a < 3; a++) { fb[a] = 0; for (int m = 0; m < 3; m++) fb[a] += R0[b][m][a] * f[m]; }
        cross3(c, fb, nb);
        for (int a = 0; a < 3; a++) { fext[b][a] += nb[a]; fext[b][3 + a] += fb[a]; }
    }
    return any;
}

double orc_dyn_motor_torque(const orc_dyn_params* d, double r_ref, double v_ref, double q, double qd)
{
    /* One law for the three motor forms of bullet_scene.py:123-155, written as a velocity servo:
     *   tau = clip(Kp (r - q) + kd (v* - qd), +-force),   v* = clamp(v + c (r - q), +-maxVelocity)
     * POSITION_CONTROL without maxVelocity: Kp = kp, c = 0 (plain PD, the r01 arithmetic, bit for bit);
     * with maxVelocity: Kp = 0, c = kp / kd (the same PD when the cap is inactive, the asked-for speed capped otherwise);
     * VELOCITY_CONTROL: Kp = 0, c = 0. */
    const int capped = d->max_velocity > 0;
    const double Kp = (d->control_mode == 1 || capped) ? 0.0 : d->kp;
    const double c = (d->control_mode == 0 && capped) ? d->kp / d->kd : 0.0;
    double vs = v_ref + c * (r_ref - q);
    if (capped) vs = vs > d->max_velocity ? d->max_velocity : (vs < -d->max_velocity ? -d->max_velocity : vs);
    double t = Kp * (r_ref - q) + d->kd * (vs - qd);
    if (d->torque_limit > 0) t = t > d->torque_limit ? d->torque_limit : (t < -d->torque_limit ? -d->torque_limit : t);
    return t;
}

static int contact_force_sphere(const orc_dyn_params* d, const double pos[3], const double vel[3], double radius, double f[3])
{
    int active = 0;
    f[0] = f[1] = f[2] = 0.0;
    if (d->ground_z == d->ground_z) {                      /* plane z = ground_z, normal +z; the POINTER touches with its
                                                            * centre (r01 form, kept), a link sample with its surface */
        double depth = d->ground_z - pos[2] + (d->link_contacts ? radius : 0.0);
        if (depth > 0) {
            double fz = d->contact_kp * depth - d->contact_kd * vel[2];
            if (fz > 0) { f[2] += fz; active = 1; }
        }
    }
    if (d->obstacle_half_extents[0] > 0 && d->obstacle_half_extents[1] > 0 && d->obstacle_half_extents[2] > 0) {
        /* signed distance of the pointer centre to the axis-aligned box, outward normal n */
        double q[3], o[3], n[3] = {0, 0, 0}, dd[3];
        double out2 = 0, qmax = -1e300; int kmax = 0;
        for (int k = 0; k < 3; k++) {
            dd[k] = pos[k] - d->obstacle_position[k];
            q[k] = fabs(dd[k]) - d->obstacle_half_extents[k];
            o[k] = q[k] > 0 ? q[k] : 0.0;
            out2 += o[k] * o[k];
            if (q[k] > qmax) { qmax = q[k]; kmax = k; }
        }
        double sdf;
        if (out2 > 0) {
            double len = sqrt(out2);
            sdf = len;
            for (int k = 0; k < 3; k++) n[k] = (dd[k] < 0 ? -o[k] : o[k]) / len;
        } else {
            sdf = qmax;                                    /* inside: push out through the nearest face */
            n[kmax] = dd[kmax] < 0 ? -1.0 : 1.0;
        }
        double depth = radius - sdf;
        if (depth > 0) {
            double vn = vel[0] * n[0] + vel[1] * n[1] + vel[2] * n[2];
            double fn = d->contact_kp * depth - d->contact_kd * vn;
            if (fn > 0) { for (int k = 0; k < 3; k++) f[k] += fn * n[k]; active = 1; }
        }
    }
    for (int b = 0; b < d->n_scene && b < ORC_MAX_SCENE; b++) {
        /* static scene bodies (create_body_plane / _box / _sphere, bullet_scene.py:193-228): signed distance of the sample's
         * centre to the shape and its outward normal n in the world frame; the sample touches with its surface */
        const orc_scene_body* B = &d->scene[b];
        double R[3][3], dd[3], n[3] = {0, 0, 0}, sdf;
        quat_to_matrix(B->orientation, R);
        for (int k = 0; k < 3; k++) dd[k] = pos[k] - B->position[k];
        if (B->shape == ORC_SHAPE_PLANE) {
            double nl = sqrt(B->size[0] * B->size[0] + B->size[1] * B->size[1] + B->size[2] * B->size[2]);
            for (int k = 0; k < 3; k++) n[k] = (R[k][0] * B->size[0] + R[k][1] * B->size[1] + R[k][2] * B->size[2]) / nl;
            sdf = n[0] * dd[0] + n[1] * dd[1] + n[2] * dd[2];
        } else if (B->shape == ORC_SHAPE_SPHERE) {
            double len = sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
            sdf = len - B->size[0];
            if (len > 0) { for (int k = 0; k < 3; k++) n[k] = dd[k] / len; } else n[2] = 1.0;
        } else if (B->shape == ORC_SHAPE_BOX) {
            double l[3], q[3], o[3], nl[3] = {0, 0, 0}, out2 = 0, qmax = -1e300; int kmax = 0;
            for (int k = 0; k < 3; k++) l[k] = R[0][k] * dd[0] + R[1][k] * dd[1] + R[2][k] * dd[2];     /* R^T dd */
            for (int k = 0; k < 3; k++) {
                q[k] = fabs(l[k]) - B->size[k];
                o[k] = q[k] > 0 ? q[k] : 0.0;
                out2 += o[k] * o[k];
                if (q[k] > qmax) { qmax = q[k]; kmax = k; }
            }
            if (out2 > 0) {
                double len = sqrt(out2);
                sdf = len;
                for (int k = 0; k < 3; k++) nl[k] = (l[k] < 0 ? -o[k] : o[k]) / len;
            } else {
                sdf = qmax;
                nl[kmax] = l[kmax] < 0 ? -1.0 : 1.0;
            }
            for (int k = 0; k < 3; k++) n[k] = R[k][0] * nl[0] + R[k][1] * nl[1] + R[k][2] * nl[2];
        } else continue;
        double depth = radius - sdf;
        if (depth > 0) {
            double vn = vel[0] * n[0] + vel[1] * n[1] + vel[2] * n[2];
            double fn = d->contact_kp * depth - d->contact_kd * vn;
            if (fn > 0) { for (int k = 0; k < 3; k++) f[k] += fn * n[k]; active = 1; }
        }
    }
    return active;
}

void orc_dyn_params_default(orc_dyn_params* d)
{
    memset(d, 0, sizeof(*d));
    d->kp = 4000.0; d->kd = 400.0; d->torque_limit = 0.0;
    d->gravity = 0.0; d->timestep = 1.0 / 240; d->frame_skip = 10;
    d->teleport = 0; d->randomize = 0;
    d->joint_damping = 0.0; d->joint_friction = 0.0;
    d->rand_mass_lo = 0.5; d->rand_mass_hi = 1.5;
    d->rand_friction_lo = 0.0; d->rand_friction_hi = 0.1;
    d->rand_damping_lo = 0.0; d->rand_damping_hi = 0.1;
    d->ground_z = NAN; d->contact_kp = 2000.0; d->contact_kd = 50.0;
    d->obstacle_position[0] = 10.0; d->obstacle_position[1] = 5.0; d->obstacle_position[2] = 0.0;   /* pioneer_knm_env.py:253 */
    d->obstacle_half_extents[0] = d->obstacle_half_extents[1] = d->obstacle_half_extents[2] = 0.0;  /* disabled */
    d->pointer_radius = 0.2;                                                                          /* urdf:193 */
    d->control_mode = 0; d->link_contacts = 0; d->max_velocity = 0.0;
    d->n_scene = 0; d->pd_inertia_scaled = 0;
}

void orc_dyn_substep(const orc_dyn_params* d, const orc_params* p, orc_dyn_state* s,
                     const double r_ref[ORC_DOF], const double v_ref[ORC_DOF])
{
    double tau[ORC_DOF], qdd[ORC_DOF], fext[ORC_DOF][6];
    int have_ext = 0;
    double ades[ORC_DOF];
    for (int i = 0; i < ORC_DOF; i++) {
        double t = 0.0;
        ades[i] = 0.0;
        if (!d->teleport) {
            if (d->pd_inertia_scaled) {          /* the motor law UNCAPPED, as an acceleration request; the cap acts on D_i * ades */
                orc_dyn_params u = *d; u.torque_limit = 0.0;
                ades[i] = orc_dyn_motor_torque(&u, r_ref[i], v_ref[i], s->q[i], s->qd[i]);
            } else t = orc_dyn_motor_torque(d, r_ref[i], v_ref[i], s->q[i], s->qd[i]);
        }
        t -= s->damping[i] * s->qd[i];
        t -= s->friction[i] * s->qd[i] / sqrt(s->qd[i] * s->qd[i] + FRICTION_EPS * FRICTION_EPS);
        tau[i] = t;
    }
    if (d->ground_z == d->ground_z || d->obstacle_half_extents[0] > 0 || d->n_scene > 0)
        have_ext = orc_dyn_contact_wrenches(d, s, fext);
    orc_dyn_aba_motor(s, tau, d->pd_inertia_scaled ? ades : NULL, d->torque_limit, d->gravity,
                      have_ext ? (const double (*)[6])fext : NULL, qdd);
    for (int i = 0; i < ORC_DOF; i++) {          /* semi-implicit Euler + inelastic joint limits */
        s->qd[i] += qdd[i] * d->timestep;
        s->q[i] += s->qd[i] * d->timestep;
        double hi = (double)p->r_hi[i], lo = (double)p->r_lo[i];
        if (s->q[i] > hi) { s->q[i] = hi; if (s->qd[i] > 0) s->qd[i] = 0; }
        if (s->q[i] < lo) { s->q[i] = lo; if (s->qd[i] < 0) s->qd[i] = 0; }
    }
}

/* World.step() alone (bullet_scene.py:273-275: frame_skip x stepSimulation; include/pioneer_amd.h pnr_world_step): the
 * sub-steps of orc_dyn_substep with each joint's own motor — m[i] where Joint.control_position / control_velocity set one
 * (bullet_scene.py:123-155; m[i].enabled), else the env-wide law on the env's command state ks->r, ks->v (teleport: none).
 * No command integration, reward or observation.  Build-defined like all of dynamics mode: parity unpinned. */
void orc_dyn_world_step(const orc_dyn_params* d, const orc_params* p, const orc_state* ks, orc_dyn_state* s, const orc_joint_motor* m)
{
    for (int k = 0; k < d->frame_skip; k++) {
        double tau[ORC_DOF], qdd[ORC_DOF], fext[ORC_DOF][6], ades[ORC_DOF], caps[ORC_DOF];
        int have_ext = 0;
        for (int i = 0; i < ORC_DOF; i++) {
            orc_dyn_params u = *d;                       /* this joint's motor in the terms of the one law */
            double r_ref = ks->r[i], v_ref = (double)ks->v[i];
            int on = !d->teleport;
            if (m && m[i].enabled) {
                u.control_mode = m[i].control_mode; u.kp = m[i].position_gain; u.kd = m[i].velocity_gain;
                u.torque_limit = m[i].max_force; u.max_velocity = m[i].control_mode == 1 ? 0.0 : m[i].max_velocity;
                r_ref = m[i].target_position; v_ref = m[i].target_velocity; on = 1;
            }
            caps[i] = u.torque_limit;
            double t = 0.0;
            ades[i] = 0.0;
            if (on) {
                if (d->pd_inertia_scaled) { orc_dyn_params w = u; w.torque_limit = 0.0; ades[i] = orc_dyn_motor_torque(&w, r_ref, v_ref, s->q[i], s->qd[i]); }
                else t = orc_dyn_motor_torque(&u, r_ref, v_ref, s->q[i], s->qd[i]);
            }
            t -= s->damping[i] * s->qd[i];
            t -= s->friction[i] * s->qd[i] / sqrt(s->qd[i] * s->qd[i] + FRICTION_EPS * FRICTION_EPS);
            tau[i] = t;
        }
        if (d->ground_z == d->ground_z || d->obstacle_half_extents[0] > 0 || d->n_scene > 0)
            have_ext = orc_dyn_contact_wrenches(d, s, fext);
        aba_motor_caps(s, tau, d->pd_inertia_scaled ? ades : NULL, caps, d->gravity, have_ext ? (const double (*)[6])fext : NULL, qdd);
        for (int i = 0; i < ORC_DOF; i++) {          /* semi-implicit Euler + inelastic joint limits (orc_dyn_substep's) */
            s->qd[i] += qdd[i] * d->timestep;
            s->q[i] += s->qd[i] * d->timestep;
            double hi = (double)p->r_hi[i], lo = (double)p->r_lo[i];
            if (s->q[i] > hi) { s->q[i] = hi; if (s->qd[i] > 0) s->qd[i] = 0; }
            if (s->q[i] < lo) { s->q[i] = lo; if (s->qd[i] < 0) s->qd[i] = 0; }
        }
    }
}

void orc_dyn_reset(const orc_dyn_params* d, const orc_params* p, const orc_state* ks, orc_dyn_state* s,
                   uint64_t genv, uint32_t episode_drawn)
{
    for (int i = 0; i < ORC_DOF; i++) { s->q[i] = (double)(float)ks->r[i]; s->qd[i] = 0.0; }
    double u[24];
    if (d->randomize)
        for (uint32_t b = 0; b < 6; b++) orc_draw_block(p, genv, episode_drawn, 3 + b, u + 4 * b);
    for (int l = 0; l < ORC_LINKS; l++)
        s->mass_scale[l] = d->randomize ? (double)(float)(d->rand_mass_lo + (d->rand_mass_hi - d->rand_mass_lo) * u[l]) : 1.0;
    for (int i = 0; i < ORC_DOF; i++) {
        s->friction[i] = d->randomize ? (double)(float)(d->rand_friction_lo + (d->rand_friction_hi - d->rand_friction_lo) * u[11 + i])
                                      : (double)(float)d->joint_friction;
        s->damping[i] = d->randomize ? (double)(float)(d->rand_damping_lo + (d->rand_damping_hi - d->rand_damping_lo) * u[17 + i])
                                     : (double)(float)d->joint_damping;
    }
}

void orc_dyn_step(const orc_dyn_params* d, const orc_params* p, orc_state* ks, orc_dyn_state* s,
                  uint64_t genv, const float action[ORC_DOF], double obs[ORC_OBS],
                  double* reward, uint8_t* done_out, uint8_t* trunc_out, double info[4])
{
    orc_integrate(p, ks, action);                 /* the kinematic command generator (parity-mode state) */
    double r_ref[ORC_DOF], v_ref[ORC_DOF];
    for (int i = 0; i < ORC_DOF; i++) { r_ref[i] = ks->r[i]; v_ref[i] = (double)ks->v[i]; }
    if (d->teleport)                              /* resetJointState: pioneer_knm_env.py:148, bullet_scene.py:157-165 */
        for (int i = 0; i < ORC_DOF; i++) { s->q[i] = r_ref[i]; s->qd[i] = 0.0; }
    for (int k = 0; k < d->frame_skip; k++) orc_dyn_substep(d, p, s, r_ref, v_ref);   /* World.step, bullet_scene.py:273-275 */
    /* device storage model: q, qd are float32 between steps */
    double qf[ORC_DOF], qdf[ORC_DOF];
    for (int i = 0; i < ORC_DOF; i++) {
        s->q[i] = qf[i] = (double)(float)s->q[i];
        s->qd[i] = qdf[i] = (double)(float)s->qd[i];
        /* teleport = the reference's semantics: its obs shows the env's own v (pioneer_knm_env.py:202),
         * not Bullet's joint velocity (which resetJointState zeroes) */
        if (d->teleport) qdf[i] = (double)ks->v[i];
    }
    double info_local[4];
    int flags = orc_reward(p, ks, qf, reward, info ? info : info_local);
    int done = flags & 1, trunc = (flags >> 1) & 1;
    {   /* a diverged (non-finite) simulation is cut like a time-out */
        double dist = (info ? info : info_local)[3];
        if (!(dist == dist && fabs(dist) <= 3.0e38)) trunc = !done;
    }
    if (done_out) *done_out = (uint8_t)done;
    if (trunc_out) *trunc_out = (uint8_t)trunc;
    if (p->auto_reset && (done || trunc)) {
        orc_reset(p, ks, genv, NULL, NULL);
        orc_dyn_reset(d, p, ks, s, genv, ks->episode - 1);
        for (int i = 0; i < ORC_DOF; i++) { qf[i] = s->q[i]; qdf[i] = 0.0; }
    }
    if (obs) orc_observe_qv(p, ks, qf, qdf, 0, obs);
}

void orc_dyn_reset_batch(const orc_dyn_params* d, const orc_params* p, orc_state* ks, orc_dyn_state* s,
                         int64_t n, int64_t off, const uint8_t* mask, const double* joint_pos,
                         const double* target_pos, double* obs, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < n; e++) {
        if (mask && !mask[e]) continue;
        orc_reset(p, &ks[e], (uint64_t)(off + e), joint_pos ? joint_pos + 6 * e : NULL,
                  target_pos ? target_pos + 3 * e : NULL);
        orc_dyn_reset(d, p, &ks[e], &s[e], (uint64_t)(off + e), ks[e].episode - 1);
        if (obs) {
            double qd0[ORC_DOF] = {0, 0, 0, 0, 0, 0};
            orc_observe_qv(p, &ks[e], s[e].q, qd0, 0, obs + (int64_t)ORC_OBS * e);
        }
    }
}

void orc_dyn_step_batch(const orc_dyn_params* d, const orc_params* p, orc_state* ks, orc_dyn_state* s,
                        int64_t n, int64_t off, const float* actions, double* obs,
                        double* reward, uint8_t* done, uint8_t* truncated, double* info, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < n; e++)
        orc_dyn_step(d, p, &ks[e], &s[e], (uint64_t)(off + e), actions + 6 * e,
                     obs ? obs + (int64_t)ORC_OBS * e : NULL, reward ? reward + e : NULL,
                     done ? done + e : NULL, truncated ? truncated + e : NULL, info ? info + 4 * e : NULL);
}

int orc_dyn_sizeof_params(void) { return (int)sizeof(orc_dyn_params); }
int orc_dyn_sizeof_state(void) { return (int)sizeof(orc_dyn_state); }
