// pnr_dyn.h — dynamics mode: articulated-body forward dynamics (Featherstone ABA)
// + PD joint torques + joint limits + pointer/ground penalty contact, frame_skip
// sub-steps per env-step (World.step of the reference, bullet_scene.py:273-275,
// is where Bullet would do this; under the reference's defaults it is a no-op,
// SURVEY.md a6).  PARITY UNPINNED: checked against oracle/pnr_dyn_oracle.c.
//
// One env per lane (the recursion over the chain is serial), float32, everything
// in registers.  The chain is the constexpr table of pnr_model.h: every joint
// axis is a coordinate axis, so S_i is a unit vector (U_i is a column of the
// articulated inertia, D_i a diagonal element) and every Pluecker transform is a
// Givens rotation plus a constexpr translation; the unrolled code drops the terms
// that vanish for this arm.  VALU-bound (~1.3 k FMA per ABA, 10 per env-step);
// no MFMA: per-lane 6x6 recursions with data-dependent pivots, not a contraction.
//
// State: planar float32 words [36][n] (the canonical pnr_get_dyn_state layout):
//   0-5 q, 6-11 qd, 12-22 per-link mass scale, 23-28 friction, 29-34 damping, 35 pad.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/pioneer_amd.h"
#include "pnr_device.h"

// The dynamics arithmetic has no bit-exactness contract (float32 vs a float64 oracle, tolerance-
// checked), so let a*b+c fuse here; integrate_joint (pnr_device.h) keeps its per-instruction
// no-contract flags when it is inlined into these kernels.  Restored at the end of this header.
#pragma clang fp contract(fast)

namespace pnr {

struct DynParams {
    float* dyn;                 // [36][n]
    float kp, kd, tau_max;      // PD gains, torque cap (<= 0: none)
    float gravity;
    float dt_sub;               // SimulationConfig.timestep
    int nsub;                   // SimulationConfig.frame_skip
    int teleport;
    int randomize;
    int has_ground;
    int has_box;
    float ground_z, ckp, ckd;
    float box_c[3], box_h[3], ptr_radius;   // static box obstacle for the pointer sphere
    float joint_damping, joint_friction;
    double mass_lo, mass_span, fric_lo, fric_span, damp_lo, damp_span;
};

constexpr float kFrictionEps = 0.05f;   // smooth sign(qd) = qd / sqrt(qd^2 + eps^2)

// 1/x: hardware reciprocal (1 ulp) + one Newton step, instead of the ~10-instruction IEEE division
__device__ __forceinline__ float fast_rcp(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return r * (2.0f - x * r);
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float comp(V3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
__device__ __forceinline__ void add_comp(V3& a, int k, float v) { if (k == 0) a.x += v; else if (k == 1) a.y += v; else a.z += v; }

// R(axis, angle) v : child coordinates -> parent coordinates.  R^T v = rot<AX>(v, c, -s).
template <int AXn>
__device__ __forceinline__ V3 rot(V3 v, float c, float s)
{
    if (AXn == (int)AX) return {v.x, c * v.y - s * v.z, s * v.y + c * v.z};
    if (AXn == (int)AY) return {c * v.x + s * v.z, v.y, c * v.z - s * v.x};
    return {c * v.x - s * v.y, s * v.x + c * v.y, v.z};
}

// r_J x v with the constexpr joint origin r_J; vanishing terms are dropped at compile time
template <int J>
__device__ __forceinline__ V3 cross_r(V3 v)
{
    constexpr float rx = (float)kJoints[J].ox, ry = (float)kJoints[J].oy, rz = (float)kJoints[J].oz;
    V3 o = {0.f, 0.f, 0.f};
    if (ry != 0.f) { o.x += ry * v.z; o.z -= ry * v.x; }
    if (rz != 0.f) { o.x -= rz * v.y; o.y += rz * v.x; }
    if (rx != 0.f) { o.y -= rx * v.z; o.z += rx * v.y; }
    return o;
}

// 3x3 matrix as rows
struct M3 { V3 r0, r1, r2; };
__device__ __forceinline__ V3 mul(const M3& m, V3 v) { return {dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)}; }
__device__ __forceinline__ V3 mulT(const M3& m, V3 v) { return v.x * m.r0 + v.y * m.r1 + v.z * m.r2; }
__device__ __forceinline__ V3 col(const M3& m, int k) { return {comp(m.r0, k), comp(m.r1, k), comp(m.r2, k)}; }
__device__ __forceinline__ V3 row(const M3& m, int k) { return k == 0 ? m.r0 : (k == 1 ? m.r1 : m.r2); }
__device__ __forceinline__ M3 operator+(const M3& a, const M3& b) { return {a.r0 + b.r0, a.r1 + b.r1, a.r2 + b.r2}; }
__device__ __forceinline__ M3 operator-(const M3& a, const M3& b) { return {a.r0 - b.r0, a.r1 - b.r1, a.r2 - b.r2}; }
__device__ __forceinline__ M3 transpose(const M3& m)
{
    return {{m.r0.x, m.r1.x, m.r2.x}, {m.r0.y, m.r1.y, m.r2.y}, {m.r0.z, m.r1.z, m.r2.z}};
}
__device__ __forceinline__ M3 outer(V3 a, V3 b) { return {a.x * b, a.y * b, a.z * b}; }
__device__ __forceinline__ M3 diag3(float d) { return {{d, 0.f, 0.f}, {0.f, d, 0.f}, {0.f, 0.f, d}}; }

// R M R^T
template <int AXn>
__device__ __forceinline__ M3 rot_block(const M3& m, float c, float s)
{
    // rows: R M  (each column transformed like a vector) == transform the row-vectors' mixing
    const M3 t = transpose(m);                                   // columns of m as rows
    const M3 rm = transpose(M3{rot<AXn>(t.r0, c, s), rot<AXn>(t.r1, c, s), rot<AXn>(t.r2, c, s)});  // R M
    return {rot<AXn>(rm.r0, c, s), rot<AXn>(rm.r1, c, s), rot<AXn>(rm.r2, c, s)};                   // (R M) R^T: rows times R^T
}

// R S R^T for a SYMMETRIC S: six unique entries.  With (a, b) the two coordinates the rotation mixes
// (v_a' = c v_a - s v_b, v_b' = s v_a + c v_b) and k the axis:
//   S'aa = c^2 Saa - 2cs Sab + s^2 Sbb     S'bb = s^2 Saa + 2cs Sab + c^2 Sbb
//   S'ab = cs (Saa - Sbb) + (c^2 - s^2) Sab  S'ak = c Sak - s Sbk   S'bk = s Sak + c Sbk   S'kk = Skk
template <int AXn>
__device__ __forceinline__ M3 rot_sym(const M3& m, float c, float s)
{
    constexpr int a = (AXn + 1) % 3, b = (AXn + 2) % 3, k = AXn;
    const float Saa = comp(row(m, a), a), Sbb = comp(row(m, b), b), Sab = comp(row(m, a), b);
    const float Sak = comp(row(m, a), k), Sbk = comp(row(m, b), k), Skk = comp(row(m, k), k);
    const float cc = c * c, ss = s * s, cs = c * s;
    const float naa = cc * Saa - 2.f * cs * Sab + ss * Sbb;
    const float nbb = ss * Saa + 2.f * cs * Sab + cc * Sbb;
    const float nab = cs * (Saa - Sbb) + (cc - ss) * Sab;
    const float nak = c * Sak - s * Sbk, nbk = s * Sak + c * Sbk;
    float e[3][3];
    e[a][a] = naa; e[b][b] = nbb; e[k][k] = Skk;
    e[a][b] = e[b][a] = nab; e[a][k] = e[k][a] = nak; e[b][k] = e[k][b] = nbk;
    return {{e[0][0], e[0][1], e[0][2]}, {e[1][0], e[1][1], e[1][2]}, {e[2][0], e[2][1], e[2][2]}};
}

// a a^T scaled: symmetric, six products
__device__ __forceinline__ M3 outer_sym(float k, V3 a)
{
    const float xx = k * a.x * a.x, yy = k * a.y * a.y, zz = k * a.z * a.z;
    const float xy = k * a.x * a.y, xz = k * a.x * a.z, yz = k * a.y * a.z;
    return {{xx, xy, xz}, {xy, yy, yz}, {xz, yz, zz}};
}

// spatial (articulated) inertia [[A, B], [B^T, C]] acting on [w; v]: n = A w + B v, f = B^T w + C v
struct SI { M3 A, B, C; };
struct SV { V3 a, l; };   // spatial vector: angular part, linear part

// skew(r_J) * M (each column crossed with r) and M * skew(r_J)
template <int J>
__device__ __forceinline__ M3 rx_mul(const M3& m)
{
    const M3 t = transpose(m);
    return transpose(M3{cross_r<J>(t.r0), cross_r<J>(t.r1), cross_r<J>(t.r2)});
}
template <int J>
__device__ __forceinline__ M3 mul_rx(const M3& m)
{
    // (M rx) = -(rx M^T)^T  since rx^T = -rx
    const M3 t = rx_mul<J>(transpose(m));
    const M3 tt = transpose(t);
    return {-1.f * tt.r0, -1.f * tt.r1, -1.f * tt.r2};
}

struct DynBody {   // what pass 3 needs from pass 2
    V3 Ua, Ul;
    float D, u;
};

struct DynModel {  // per-env rigid-body data from the 11 link scales
    float m[kDof];      // body masses (== isotropic inertia for bodies 0..4)
    V3 h6;              // first moment of body 6 (pointer offset)
    M3 I6;              // rotational inertia of body 6 about its origin
};

__device__ __forceinline__ void build_model(const float (&sc)[kNumLinks], DynModel& M)
{
    M.m[0] = sc[1] + sc[2]; M.m[1] = sc[3]; M.m[2] = sc[4]; M.m[3] = sc[5] + sc[6]; M.m[4] = sc[7];
    M.m[5] = sc[8] + sc[9] + sc[10];
    constexpr float tx = (float)kTipX, ty = (float)kTipY, tz = (float)kTipZ;
    const float mp = sc[10];
    M.h6 = {mp * tx, mp * ty, mp * tz};
    const float t2 = tx * tx + ty * ty + tz * tz;
    M.I6 = {{M.m[5] + mp * (t2 - tx * tx), -mp * tx * ty, -mp * tx * tz},
            {-mp * ty * tx, M.m[5] + mp * (t2 - ty * ty), -mp * ty * tz},
            {-mp * tz * tx, -mp * tz * ty, M.m[5] + mp * (t2 - tz * tz)}};
}

// one joint of pass 2 (tip -> base).  IA/pA: articulated inertia / bias force of body J in its own
// frame (children already folded in).  Emits U, D, u and folds body J into its parent (PA, pP).
template <int J>
__device__ __forceinline__ void aba_inward(const SI& IA, const SV& pA, const SV& vJ, float qdJ, float tauJ,
                                           float cJ, float sJ, DynBody& out, SI& IP, SV& pP)
{
    constexpr int k = (int)kJoints[J].axis;
    out.Ua = col(IA.A, k);
    out.Ul = row(IA.B, k);
    out.D = comp(out.Ua, k);
    out.u = tauJ - comp(pA.a, k);
    if (J == 0) return;
    const float invD = fast_rcp(out.D);
    // Ia = IA - U U^T / D
    SI Ia;
    Ia.A = IA.A - outer_sym(invD, out.Ua);
    Ia.B = IA.B - outer(invD * out.Ua, out.Ul);
    Ia.C = IA.C - outer_sym(invD, out.Ul);
    // c = v x (S qd)
    V3 ek = {k == 0 ? qdJ : 0.f, k == 1 ? qdJ : 0.f, k == 2 ? qdJ : 0.f};
    const V3 ca = cross(vJ.a, ek), cl = cross(vJ.l, ek);
    // pa = pA + Ia c + U u / D
    const float ud = out.u * invD;
    SV pa;
    pa.a = pA.a + mul(Ia.A, ca) + mul(Ia.B, cl) + ud * out.Ua;
    pa.l = pA.l + mulT(Ia.B, ca) + mul(Ia.C, cl) + ud * out.Ul;
    // rotate into the parent's orientation, then shift to the parent's origin:
    //   C_p = C', B_p = B' + rx C', A_p = A' - P - P^T - (rx C') rx  with P = B' rx
    constexpr Axis AXJ = kJoints[J].axis;
    const M3 A1 = rot_sym<(int)AXJ>(Ia.A, cJ, sJ), B1 = rot_block<(int)AXJ>(Ia.B, cJ, sJ), C1 = rot_sym<(int)AXJ>(Ia.C, cJ, sJ);
    const M3 T = rx_mul<J>(C1);
    const M3 P = mul_rx<J>(B1);
    const M3 Q = mul_rx<J>(T);
    IP.C = IP.C + C1;
    IP.B = IP.B + B1 + T;
    IP.A = IP.A + A1 - P - transpose(P) - Q;
    const V3 n1 = rot<(int)AXJ>(pa.a, cJ, sJ), f1 = rot<(int)AXJ>(pa.l, cJ, sJ);
    pP.a = pP.a + n1 + cross_r<J>(f1);
    pP.l = pP.l + f1;
}

// rigid-body inertia and velocity-product bias of body J
template <int J>
__device__ __forceinline__ void rigid_body(const DynModel& M, const SV& v, SI& I, SV& p)
{
    if (J < kDof - 1) {
        const float m = M.m[J];
        I.A = diag3(m); I.B = diag3(0.f); I.C = diag3(m);
        p.a = {0.f, 0.f, 0.f};
        p.l = m * cross(v.a, v.l);
    } else {
        const float m = M.m[J];
        const V3 h = M.h6;
        I.A = M.I6;
        I.B = {{0.f, -h.z, h.y}, {h.z, 0.f, -h.x}, {-h.y, h.x, 0.f}};
        I.C = diag3(m);
        const V3 n = mul(M.I6, v.a) + cross(h, v.l);
        const V3 f = m * v.l - cross(h, v.a);
        p.a = cross(v.a, n) + cross(v.l, f);
        p.l = cross(v.a, f);
    }
}

template <int J>
__device__ __forceinline__ void vel_outward(const SV& vp, float c, float s, float qd, SV& v)
{
    constexpr int AXJ = (int)kJoints[J].axis;
    v.a = rot<AXJ>(vp.a, c, -s);
    v.l = rot<AXJ>(vp.l - cross_r<J>(vp.a), c, -s);     // E (v_p - r x w_p)
    add_comp(v.a, AXJ, qd);
}

template <int J>
__device__ __forceinline__ void acc_outward(const SV& ap, const SV& vJ, float c, float s, float qd, const DynBody& b,
                                            float& qdd, SV& a)
{
    constexpr int AXJ = (int)kJoints[J].axis;
    a.a = rot<AXJ>(ap.a, c, -s);
    a.l = rot<AXJ>(ap.l - cross_r<J>(ap.a), c, -s);
    V3 ek = {AXJ == 0 ? qd : 0.f, AXJ == 1 ? qd : 0.f, AXJ == 2 ? qd : 0.f};
    a.a = a.a + cross(vJ.a, ek);
    a.l = a.l + cross(vJ.l, ek);
    qdd = (b.u - dot(b.Ua, a.a) - dot(b.Ul, a.l)) * fast_rcp(b.D);
    add_comp(a.a, AXJ, qdd);
}

// world pose of body J from its parent's (for the pointer/ground contact)
template <int J>
__device__ __forceinline__ void pose_outward(const M3& Rp, V3 pp, float c, float s, M3& R, V3& p)
{
    constexpr int AXJ = (int)kJoints[J].axis;
    constexpr float ox = (float)kJoints[J].ox, oy = (float)kJoints[J].oy, oz = (float)kJoints[J].oz;
    p = pp + mul(Rp, V3{ox, oy, oz});
    // R = Rp * R(axis, q): rows of Rp times R  -> row' = R^T-applied... (row * R) = rot(row, c, -s)
    R = {rot<AXJ>(Rp.r0, c, -s), rot<AXJ>(Rp.r1, c, -s), rot<AXJ>(Rp.r2, c, -s)};
}

// qdd = ABA(q, qd, tau); optional ground contact on the pointer
__device__ __forceinline__ void aba(const DynParams& D, const DynModel& M, const float (&q)[kDof], const float (&qd)[kDof],
                                    const float (&tau)[kDof], float (&qdd)[kDof])
{
    float c[kDof], s[kDof];
#pragma unroll
    for (int i = 0; i < kDof; ++i) sincos_bounded(q[i], s[i], c[i]);

    // pass 1: body velocities
    SV v[kDof];
    const SV v0 = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    vel_outward<0>(v0, c[0], s[0], qd[0], v[0]);
    vel_outward<1>(v[0], c[1], s[1], qd[1], v[1]);
    vel_outward<2>(v[1], c[2], s[2], qd[2], v[2]);
    vel_outward<3>(v[2], c[3], s[3], qd[3], v[3]);
    vel_outward<4>(v[3], c[4], s[4], qd[4], v[4]);
    vel_outward<5>(v[4], c[5], s[5], qd[5], v[5]);

    // external force on the pointer (body 6 coordinates): penalty contacts with the ground plane
    // and the static box (the reference demo's scene extras, pioneer_knm_env.py:249-261)
    SV fext = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (D.has_ground || D.has_box) {
        M3 R = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}}, Rn;
        V3 p = {0.f, 0.f, 0.f}, pn;
        pose_outward<0>(R, p, c[0], s[0], Rn, pn); R = Rn; p = pn;
        pose_outward<1>(R, p, c[1], s[1], Rn, pn); R = Rn; p = pn;
        pose_outward<2>(R, p, c[2], s[2], Rn, pn); R = Rn; p = pn;
        pose_outward<3>(R, p, c[3], s[3], Rn, pn); R = Rn; p = pn;
        pose_outward<4>(R, p, c[4], s[4], Rn, pn); R = Rn; p = pn;
        pose_outward<5>(R, p, c[5], s[5], Rn, pn); R = Rn; p = pn;
        const V3 t = {(float)kTipX, (float)kTipY, (float)kTipZ};
        const V3 tip = p + mul(R, t);                         // world position of the pointer
        const V3 vb = v[5].l + cross(v[5].a, t);
        const V3 vw = mul(R, vb);                             // world velocity of the pointer
        V3 F = {0.f, 0.f, 0.f};
        if (D.has_ground) {
            const float depth = D.ground_z - tip.z;
            const float fz = D.ckp * depth - D.ckd * vw.z;
            if (depth > 0.f && fz > 0.f) F.z += fz;
        }
        if (D.has_box) {
            // signed distance of the pointer centre to the axis-aligned box and its outward normal
            const V3 dd = {tip.x - D.box_c[0], tip.y - D.box_c[1], tip.z - D.box_c[2]};
            const V3 q = {fabsf(dd.x) - D.box_h[0], fabsf(dd.y) - D.box_h[1], fabsf(dd.z) - D.box_h[2]};
            const V3 o = {fmaxf(q.x, 0.f), fmaxf(q.y, 0.f), fmaxf(q.z, 0.f)};
            const float out2 = dot(o, o);
            V3 nrm; float sdf;
            if (out2 > 0.f) {
                const float len = sqrtf(out2);
                sdf = len;
                nrm = {(dd.x < 0.f ? -o.x : o.x) / len, (dd.y < 0.f ? -o.y : o.y) / len, (dd.z < 0.f ? -o.z : o.z) / len};
            } else {                                          // inside: out through the nearest face
                const int km = (q.x >= q.y && q.x >= q.z) ? 0 : (q.y >= q.z ? 1 : 2);
                sdf = comp(q, km);
                nrm = {km == 0 ? (dd.x < 0.f ? -1.f : 1.f) : 0.f, km == 1 ? (dd.y < 0.f ? -1.f : 1.f) : 0.f,
                       km == 2 ? (dd.z < 0.f ? -1.f : 1.f) : 0.f};
            }
            const float depth = D.ptr_radius - sdf;
            const float fn = D.ckp * depth - D.ckd * dot(vw, nrm);
            if (depth > 0.f && fn > 0.f) F = F + fn * nrm;
        }
        const V3 fb = mulT(R, F);                             // R^T F
        fext.a = cross(t, fb);
        fext.l = fb;
    }

    // pass 2: tip -> base
    DynBody B[kDof];
    SI IA, IP; SV pA, pP;
    rigid_body<5>(M, v[5], IA, pA);
    pA.a = pA.a - fext.a; pA.l = pA.l - fext.l;
    rigid_body<4>(M, v[4], IP, pP);
    aba_inward<5>(IA, pA, v[5], qd[5], tau[5], c[5], s[5], B[5], IP, pP);
    IA = IP; pA = pP; rigid_body<3>(M, v[3], IP, pP);
    aba_inward<4>(IA, pA, v[4], qd[4], tau[4], c[4], s[4], B[4], IP, pP);
    IA = IP; pA = pP; rigid_body<2>(M, v[2], IP, pP);
    aba_inward<3>(IA, pA, v[3], qd[3], tau[3], c[3], s[3], B[3], IP, pP);
    IA = IP; pA = pP; rigid_body<1>(M, v[1], IP, pP);
    aba_inward<2>(IA, pA, v[2], qd[2], tau[2], c[2], s[2], B[2], IP, pP);
    IA = IP; pA = pP; rigid_body<0>(M, v[0], IP, pP);
    aba_inward<1>(IA, pA, v[1], qd[1], tau[1], c[1], s[1], B[1], IP, pP);
    IA = IP; pA = pP;
    aba_inward<0>(IA, pA, v[0], qd[0], tau[0], c[0], s[0], B[0], IP, pP);

    // pass 3: base -> tip; gravity as a base acceleration +g along z
    SV a0 = {{0.f, 0.f, 0.f}, {0.f, 0.f, D.gravity}}, a1;
    acc_outward<0>(a0, v[0], c[0], s[0], qd[0], B[0], qdd[0], a1); a0 = a1;
    acc_outward<1>(a0, v[1], c[1], s[1], qd[1], B[1], qdd[1], a1); a0 = a1;
    acc_outward<2>(a0, v[2], c[2], s[2], qd[2], B[2], qdd[2], a1); a0 = a1;
    acc_outward<3>(a0, v[3], c[3], s[3], qd[3], B[3], qdd[3], a1); a0 = a1;
    acc_outward<4>(a0, v[4], c[4], s[4], qd[4], B[4], qdd[4], a1); a0 = a1;
    acc_outward<5>(a0, v[5], c[5], s[5], qd[5], B[5], qdd[5], a1);
}

// ---------------------------------------------------------------------------------
// kernel A: kinematic command integration (the parity-mode integrator) + nsub
// sub-steps of ABA + PD.  One env per lane.  Leaves step_index / reward / obs to
// step_kernel<..., DYN = true>.
// ---------------------------------------------------------------------------------
// RAND = per-env link scales (domain randomisation): loaded from the dyn words; otherwise every scale
// is 1 and the model folds into literals.
template <bool ACT_EM, bool RAND>
__global__ __launch_bounds__(256) void dyn_substeps_kernel(const KParams P, const DynParams D, int t)
{
    const long long n = P.n;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;

    // kinematic state: both half-records of this env
    const long long n2 = 2 * n;
    float4 k0[2], k1[2], k2[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        k0[p] = P.state[2 * e + p]; k1[p] = P.state[n2 + 2 * e + p]; k2[p] = P.state[2 * n2 + 2 * e + p];
    }
    float a[kDof], v[kDof], r[kDof];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        a[3 * p] = k0[p].x; a[3 * p + 1] = k0[p].y; a[3 * p + 2] = k0[p].z; v[3 * p] = k0[p].w;
        v[3 * p + 1] = k1[p].x; v[3 * p + 2] = k1[p].y; r[3 * p] = k1[p].z; r[3 * p + 1] = k1[p].w;
        r[3 * p + 2] = k2[p].x;
    }
    float act[kDof];
    const float* A = P.actions + (long long)t * n * kDof;
    if (ACT_EM) {
        const float2* a2 = reinterpret_cast<const float2*>(A + e * kDof);
        const float2 x0 = a2[0], x1 = a2[1], x2 = a2[2];
        act[0] = x0.x; act[1] = x0.y; act[2] = x1.x; act[3] = x1.y; act[4] = x2.x; act[5] = x2.y;
    } else {
#pragma unroll
        for (int i = 0; i < kDof; ++i) act[i] = A[(long long)i * n + e];
    }
#pragma unroll
    for (int i = 0; i < kDof; ++i) {
        integrate_joint(a[i], v[i], r[i], P.v_max[i], limit_lo(i), limit_hi(i), P.dt, P.eps, v[i], r[i]);
        a[i] = act[i];
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        P.state[2 * e + p] = make_float4(a[3 * p], a[3 * p + 1], a[3 * p + 2], v[3 * p]);
        P.state[n2 + 2 * e + p] = make_float4(v[3 * p + 1], v[3 * p + 2], r[3 * p], r[3 * p + 1]);
        P.state[2 * n2 + 2 * e + p] = make_float4(r[3 * p + 2], k2[p].y, k2[p].z, k2[p].w);
    }

    // dynamics state
    float q[kDof], qd[kDof], sc[kNumLinks], fric[kDof], damp[kDof];
#pragma unroll
    for (int i = 0; i < kDof; ++i) {
        q[i] = D.dyn[(long long)i * n + e]; qd[i] = D.dyn[(long long)(6 + i) * n + e];
        fric[i] = D.dyn[(long long)(23 + i) * n + e]; damp[i] = D.dyn[(long long)(29 + i) * n + e];
    }
#pragma unroll
    for (int l = 0; l < kNumLinks; ++l) sc[l] = RAND ? D.dyn[(long long)(12 + l) * n + e] : 1.0f;
    DynModel M;
    build_model(sc, M);

    if (D.teleport) {   // resetJointState semantics: pioneer_knm_env.py:148, bullet_scene.py:157-165
#pragma unroll
        for (int i = 0; i < kDof; ++i) { q[i] = r[i]; qd[i] = 0.f; }
    }
    for (int k = 0; k < D.nsub; ++k) {
        float tau[kDof], qdd[kDof];
#pragma unroll
        for (int i = 0; i < kDof; ++i) {
            float tq = 0.f;
            if (!D.teleport) {
                tq = D.kp * (r[i] - q[i]) + D.kd * (v[i] - qd[i]);
                if (D.tau_max > 0.f) tq = fminf(fmaxf(tq, -D.tau_max), D.tau_max);
            }
            tq -= damp[i] * qd[i];
            tq -= fric[i] * qd[i] * __builtin_amdgcn_rsqf(qd[i] * qd[i] + kFrictionEps * kFrictionEps);
            tau[i] = tq;
        }
        aba(D, M, q, qd, tau, qdd);
#pragma unroll
        for (int i = 0; i < kDof; ++i) {   // semi-implicit Euler + inelastic joint limits
            qd[i] += qdd[i] * D.dt_sub;
            q[i] += qd[i] * D.dt_sub;
            const float hi = limit_hi(i), lo = limit_lo(i);
            if (q[i] > hi) { q[i] = hi; if (qd[i] > 0.f) qd[i] = 0.f; }
            if (q[i] < lo) { q[i] = lo; if (qd[i] < 0.f) qd[i] = 0.f; }
        }
    }
#pragma unroll
    for (int i = 0; i < kDof; ++i) {
        D.dyn[(long long)i * n + e] = q[i];
        D.dyn[(long long)(6 + i) * n + e] = qd[i];
    }
}

// reset of the dynamics words for this lane's joints (called by the pair kernels after reset_env):
// q = r, qd = 0, and the per-env parameter draws (Philox blocks 3..8 of the same counter).
__device__ __forceinline__ void dyn_reset_lane(const KParams& P, const DynParams& D, const LaneState& s, int p,
                                               long long e, unsigned long long genv, uint32_t episode_drawn)
{
    const long long n = P.n;
#pragma unroll
    for (int i = 0; i < kJpl; ++i) {
        D.dyn[(long long)(kJpl * p + i) * n + e] = s.r[i];
        D.dyn[(long long)(6 + kJpl * p + i) * n + e] = 0.f;
    }
    float u[24];
    if (D.randomize) {
#pragma unroll
        for (uint32_t b = 0; b < 6; ++b) {
            uint32_t o[4];
            philox4x32_10(episode_drawn, (uint32_t)genv, (uint32_t)(genv >> 32), 3 + b, P.seed_lo, P.seed_hi, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) u[4 * b + k] = (float)u01(o[k]);
        }
    }
    // lane 0 writes the link scales, each lane its own joints' friction / damping
    if (p == 0) {
#pragma unroll
        for (int l = 0; l < kNumLinks; ++l)
            D.dyn[(long long)(12 + l) * n + e] = D.randomize ? (float)(D.mass_lo + D.mass_span * (double)u[l]) : 1.0f;
    }
#pragma unroll
    for (int i = 0; i < kJpl; ++i) {
        const int j = kJpl * p + i;
        const float uf = p ? u[11 + kJpl + i] : u[11 + i], ud = p ? u[17 + kJpl + i] : u[17 + i];
        D.dyn[(long long)(23 + j) * n + e] = D.randomize ? (float)(D.fric_lo + D.fric_span * (double)uf) : D.joint_friction;
        D.dyn[(long long)(29 + j) * n + e] = D.randomize ? (float)(D.damp_lo + D.damp_span * (double)ud) : D.joint_damping;
    }
}

}  // namespace pnr

#pragma clang fp contract(off)
