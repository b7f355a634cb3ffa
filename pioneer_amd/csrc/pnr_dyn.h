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

#include <type_traits>
#include <utility>

#include "../../include/pioneer_amd.h"
#include "pnr_device.h"

// The dynamics arithmetic has no bit-exactness contract (float32 vs a float64 oracle, tolerance-
// checked), so let a*b+c fuse here; integrate_joint (pnr_device.h) keeps its per-instruction
// no-contract flags when it is inlined into these kernels.  Restored at the end of this header.
#pragma clang fp contract(fast)

namespace pnr {

// a static scene body as the kernels want it: world position, rotation matrix (row-major; a plane keeps its unit world
// normal in rot[0..2]), size (box half extents | sphere radius in size[0])
constexpr int kMaxScene = 8;
struct SceneBody { int shape; float pos[3]; float rot[9]; float size[3]; };

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
    // motor law (bullet_scene.py:123-155), one form for POSITION_CONTROL, its maxVelocity cap and VELOCITY_CONTROL:
    //   tau = clip(kp_eff (r - q) + kd (clamp(v + c_pos (r - q), +-v_cap) - qd), +-tau_max)
    float kp_eff, c_pos, v_cap;
    int link_contacts;          // 1: the link capsules' sample spheres collide too (pnr_model.h kCapsules)
    int inertia_scaled;         // 1: the motor law is an acceleration request scaled by each joint's articulated inertia
    int n_scene;                // static bodies of create_body_plane / _box / _sphere (bullet_scene.py:193-228)
    const SceneBody* scene;     // [n_scene] in device memory (a by-value array indexed by the loop counter went to scratch)
};

// pnr_world_step (World.step, bullet_scene.py:273-275): the motor of every joint as Joint.control_position / control_velocity
// left it (bullet_scene.py:123-155), in the terms of the one motor law above — the same for every env of the handle.  A joint
// nobody commanded keeps the handle's own motor (DynParams), tracking the env's command state r, v (from_cmd).
struct JointMotorTable {
    float kp[kDof], kd[kDof], cpos[kDof], vcap[kDof], tcap[kDof];   // kp_eff, kd, c_pos, v_cap, torque cap (+inf: none) per joint
    float r_ref[kDof], v_ref[kDof];                                  // targetPosition / targetVelocity
    int from_cmd[kDof];                                              // 1: targets = the env's own r, v (no per-joint command)
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

// ---- packed spatial algebra ---------------------------------------------------------
// CDNA4's fp32 vector peak needs v_pk_{fma,mul,add}_f32 (two results per instruction), and the
// compiler's SLP pass cannot find the pairs in this code without burying them in shuffles.  So the
// pairing is done by hand, by layout: every spatial quantity keeps its ANGULAR and LINEAR halves in
// one 64-bit register pair, because the recursion treats the two halves alike almost everywhere
// (same rotation, same axis products).  hipcc turns `f2` arithmetic straight into v_pk_* ops: a
// scalar operand becomes an op_sel_hi splat and `.yx` an op_sel swap, neither costs an instruction.
typedef float f2 __attribute__((ext_vector_type(2)));

// compile-time loops: every index below is a constant in the AST already (no local array or struct is
// ever indexed by a loop variable, so nothing can be left behind in scratch)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int K> __device__ __forceinline__ float vc(const V3& a) { if constexpr (K == 0) return a.x; else if constexpr (K == 1) return a.y; else return a.z; }
template <int I, int J> __device__ __forceinline__ float mc(const M3& m)
{
    if constexpr (I == 0) return vc<J>(m.r0); else if constexpr (I == 1) return vc<J>(m.r1); else return vc<J>(m.r2);
}

// spatial vector: component i = (angular_i, linear_i)
struct P3 { f2 x, y, z; };
template <int K> __device__ __forceinline__ f2& pr(P3& a) { if constexpr (K == 0) return a.x; else if constexpr (K == 1) return a.y; else return a.z; }
template <int K> __device__ __forceinline__ const f2& pc(const P3& a) { if constexpr (K == 0) return a.x; else if constexpr (K == 1) return a.y; else return a.z; }
__device__ __forceinline__ P3 operator+(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ P3 operator-(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 ang(const P3& a) { return {a.x.x, a.y.x, a.z.x}; }
__device__ __forceinline__ V3 lin(const P3& a) { return {a.x.y, a.y.y, a.z.y}; }
__device__ __forceinline__ P3 pack(V3 a, V3 l) { return {{a.x, l.x}, {a.y, l.y}, {a.z, l.z}}; }

// both halves rotated by R(axis, angle): child -> parent; R^T = rotp<AX>(v, c, -s)
template <int AXn>
__device__ __forceinline__ P3 rotp(const P3& v, float c, float s)
{
    if (AXn == (int)AX) return {v.x, c * v.y - s * v.z, s * v.y + c * v.z};
    if (AXn == (int)AY) return {c * v.x + s * v.z, v.y, c * v.z - s * v.x};
    return {c * v.x - s * v.y, s * v.x + c * v.y, v.z};
}

// (v.a, v.l) x (e_k qd) for both halves: two packed products, component k is zero
template <int K>
__device__ __forceinline__ P3 cross_axis(const P3& v, float qd)
{
    constexpr int a = (K + 1) % 3, b = (K + 2) % 3;
    P3 c = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    pr<a>(c) = pc<b>(v) * qd;
    pr<b>(c) = pc<a>(v) * (-qd);
    return c;
}

// E (v - [0; r_J x v.a]): the motion transform parent -> child of joint J
template <int J>
__device__ __forceinline__ P3 to_child(const P3& vp, float c, float s)
{
    constexpr int AXJ = (int)kJoints[J].axis;
    const V3 rxw = cross_r<J>(ang(vp));
    P3 t = vp;
    t.x.y -= rxw.x; t.y.y -= rxw.y; t.z.y -= rxw.z;
    return rotp<AXJ>(t, c, -s);
}

// spatial (articulated) inertia [[A, B], [B^T, C]] acting on [w; v]: n = A w + B v, f = B^T w + C v.
// A and C are symmetric and always transform alike: stored as six pairs (A_ij, C_ij).  B is general;
// its transpose transforms like B itself, so the off-diagonal entries are the pairs (B_ij, B_ji).
struct SIp {
    f2 ac[6];     // (A_ij, C_ij) for ij = 00 01 02 11 12 22
    float bd[3];  // B_00 B_11 B_22
    f2 bo[3];     // (B_01, B_10) (B_02, B_20) (B_12, B_21)
};
// compile-time index maps (variable templates, so they are constants wherever they are used)
constexpr int sidx_fn(int i, int j) { const int a = i < j ? i : j, b = i < j ? j : i; return a == 0 ? b : (a == 1 ? 2 + b : 5); }
template <int I, int J> constexpr int SIDX = sidx_fn(I, J);
template <int I, int J> constexpr int OIDX = I + J - 1;
// (B_ij, B_ji)
template <int I_, int J_>
__device__ __forceinline__ f2 bpair(const SIp& I)
{
    if constexpr (I_ == J_) return (f2)(I.bd[I_]);
    else if constexpr (I_ < J_) return I.bo[OIDX<I_, J_>];
    else return I.bo[OIDX<I_, J_>].yx;
}
template <int I_, int J_>
__device__ __forceinline__ float bget(const SIp& I)
{
    if constexpr (I_ == J_) return I.bd[I_];
    else if constexpr (I_ < J_) return I.bo[OIDX<I_, J_>].x;
    else return I.bo[OIDX<I_, J_>].y;
}
template <int I_, int J_>
__device__ __forceinline__ void bset(SIp& I, f2 v)     // v = (B_ij, B_ji), I_ != J_
{
    if constexpr (I_ < J_) I.bo[OIDX<I_, J_>] = v; else I.bo[OIDX<I_, J_>] = v.yx;
}
__device__ __forceinline__ M3 unpackB(const SIp& I)
{
    return {{I.bd[0], I.bo[0].x, I.bo[1].x}, {I.bo[0].y, I.bd[1], I.bo[2].x}, {I.bo[1].y, I.bo[2].y, I.bd[2]}};
}
__device__ __forceinline__ M3 unpackC(const SIp& I)
{
    return {{I.ac[0].y, I.ac[1].y, I.ac[2].y}, {I.ac[1].y, I.ac[3].y, I.ac[4].y}, {I.ac[2].y, I.ac[4].y, I.ac[5].y}};
}

// R I R^T of all four blocks.  (a, b) are the coordinates the rotation mixes (v_a' = c v_a - s v_b,
// v_b' = s v_a + c v_b), k the axis.  Symmetric pairs as rot_sym; for B
//   B'aa = cc Baa - cs (Bab + Bba) + ss Bbb          B'bb = ss Baa + cs (Bab + Bba) + cc Bbb
//   (B'ab, B'ba) = cs (Baa - Bbb) + cc (Bab, Bba) - ss (Bba, Bab)
//   (B'ak, B'ka) = c (Bak, Bka) - s (Bbk, Bkb)       (B'bk, B'kb) = s (Bak, Bka) + c (Bbk, Bkb)
template <int AXn>
__device__ __forceinline__ SIp rot_inertia(const SIp& I, float c, float s)
{
    constexpr int a = (AXn + 1) % 3, b = (AXn + 2) % 3, k = AXn;
    const float cc = c * c, ss = s * s, cs = c * s;
    SIp o;
    const f2 Saa = I.ac[SIDX<a, a>], Sbb = I.ac[SIDX<b, b>], Sab = I.ac[SIDX<a, b>];
    const f2 Sak = I.ac[SIDX<a, k>], Sbk = I.ac[SIDX<b, k>];
    const f2 x2 = (2.f * cs) * Sab;
    o.ac[SIDX<a, a>] = cc * Saa - x2 + ss * Sbb;
    o.ac[SIDX<b, b>] = ss * Saa + x2 + cc * Sbb;
    o.ac[SIDX<a, b>] = cs * (Saa - Sbb) + (cc - ss) * Sab;
    o.ac[SIDX<a, k>] = c * Sak - s * Sbk;
    o.ac[SIDX<b, k>] = s * Sak + c * Sbk;
    o.ac[SIDX<k, k>] = I.ac[SIDX<k, k>];
    const float Baa = I.bd[a], Bbb = I.bd[b];
    const f2 pab = bpair<a, b>(I), pak = bpair<a, k>(I), pbk = bpair<b, k>(I);
    const float m = cs * (pab.x + pab.y);
    o.bd[a] = cc * Baa - m + ss * Bbb;
    o.bd[b] = ss * Baa + m + cc * Bbb;
    o.bd[k] = I.bd[k];
    const f2 nab = cs * (Baa - Bbb) + cc * pab - ss * pab.yx;
    const f2 nak = c * pak - s * pbk;
    const f2 nbk = s * pak + c * pbk;
    bset<a, b>(o, nab);
    bset<a, k>(o, nak);
    bset<b, k>(o, nbk);
    return o;
}

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
    P3 U;          // (Ua_i, Ul_i)
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
// frame (children already folded in).  Emits U, D, u and folds body J into its parent (IP, pP).
// SCALED: the joint's motor asks for the acceleration adesJ; its torque is clip(D adesJ, +-tcap), D being the articulated
// inertia about the joint axis that this step forms anyway (pnr_config.pd_inertia_scaled)
template <int J, bool SCALED>
__device__ __forceinline__ void aba_inward(const SIp& IA, const P3& pA, const P3& vJ, float qdJ, float tauJ, float adesJ, float tcap,
                                           float cJ, float sJ, DynBody& out, SIp& IP, P3& pP)
{
    constexpr int k = (int)kJoints[J].axis;
    // U = I S: (Ua_j, Ul_j) = (A_jk, B_kj)
    P3 u;
    static_for<3>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        pr<j>(u) = (f2){IA.ac[SIDX<j, k>].x, bget<k, j>(IA)};
    });
    out.U = u;
    out.D = pc<k>(u).x;
    float tq = tauJ;
    if constexpr (SCALED) tq += fminf(fmaxf(out.D * adesJ, -tcap), tcap);
    out.u = tq - pc<k>(pA).x;
    if (J == 0) return;
    const float invD = fast_rcp(out.D);
    // Ia = IA - U U^T / D
    const P3 w = {invD * u.x, invD * u.y, invD * u.z};
    SIp Ia;
    static_for<3>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        Ia.bd[i] = IA.bd[i] - pc<i>(w).x * pc<i>(u).y;                // B_ij -= Ua_i Ul_j / D
        static_for<3>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j >= i) Ia.ac[SIDX<i, j>] = IA.ac[SIDX<i, j>] - pc<i>(w) * pc<j>(u);
            if constexpr (j > i) {
                Ia.bo[OIDX<i, j>].x = IA.bo[OIDX<i, j>].x - pc<i>(w).x * pc<j>(u).y;
                Ia.bo[OIDX<i, j>].y = IA.bo[OIDX<i, j>].y - pc<j>(w).x * pc<i>(u).y;
            }
        });
    });
    // c = v x (S qd); pa = pA + Ia c + U u / D:
    //   (pa.a_i, pa.l_i) += (A_ij, C_ij) (ca_j, cl_j) + (B_ij, B_ji) (cl_j, ca_j)
    const P3 cv = cross_axis<k>(vJ, qdJ);
    const float ud = out.u * invD;
    P3 pa;
    static_for<3>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        f2 acc = pc<i>(pA) + ud * pc<i>(u);
        static_for<3>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j != k) {                                   // c_k = 0
                acc += Ia.ac[SIDX<i, j>] * pc<j>(cv);
                acc += bpair<i, j>(Ia) * pc<j>(cv).yx;
            }
        });
        pr<i>(pa) = acc;
    });
    // rotate into the parent's orientation, then shift to the parent's origin:
    //   C_p = C', B_p = B' + rx C', A_p = A' - P - P^T - (rx C') rx  with P = B' rx
    constexpr int AXJ = (int)kJoints[J].axis;
    const SIp I1 = rot_inertia<AXJ>(Ia, cJ, sJ);
    const M3 C1 = unpackC(I1), B1 = unpackB(I1);
    const M3 T = rx_mul<J>(C1);
    const M3 Pm = mul_rx<J>(B1);
    const M3 Q = mul_rx<J>(T);
    static_for<3>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        IP.bd[i] += I1.bd[i] + mc<i, i>(T);
        static_for<3>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j >= i) {
                IP.ac[SIDX<i, j>] += I1.ac[SIDX<i, j>];
                IP.ac[SIDX<i, j>].x -= mc<i, j>(Pm) + mc<j, i>(Pm) + mc<i, j>(Q);
            }
            if constexpr (j > i) {
                IP.bo[OIDX<i, j>] += I1.bo[OIDX<i, j>];
                IP.bo[OIDX<i, j>].x += mc<i, j>(T);
                IP.bo[OIDX<i, j>].y += mc<j, i>(T);
            }
        });
    });
    const P3 p1 = rotp<AXJ>(pa, cJ, sJ);
    const V3 rxf = cross_r<J>(lin(p1));
    pP = pP + p1;
    pP.x.x += rxf.x; pP.y.x += rxf.y; pP.z.x += rxf.z;
}

// rigid-body inertia and velocity-product bias of body J
template <int J>
__device__ __forceinline__ void rigid_body(const DynModel& M, const P3& vp, SIp& I, P3& p)
{
    const V3 va = ang(vp), vl = lin(vp);
    const float m = M.m[J];
    const f2 mm = {m, m}, zz = {0.f, 0.f};
    if (J < kDof - 1) {                    // A = C = m 1, B = 0
        I.ac[0] = mm; I.ac[1] = zz; I.ac[2] = zz; I.ac[3] = mm; I.ac[4] = zz; I.ac[5] = mm;
        I.bd[0] = I.bd[1] = I.bd[2] = 0.f;
        I.bo[0] = zz; I.bo[1] = zz; I.bo[2] = zz;
        p = pack(V3{0.f, 0.f, 0.f}, m * cross(va, vl));
    } else {                               // A = I6, B = skew(h), C = m 1
        const V3 h = M.h6;
        const float a00 = M.I6.r0.x, a01 = M.I6.r0.y, a02 = M.I6.r0.z, a11 = M.I6.r1.y, a12 = M.I6.r1.z, a22 = M.I6.r2.z;
        I.ac[0] = (f2){a00, m};   I.ac[1] = (f2){a01, 0.f}; I.ac[2] = (f2){a02, 0.f};
        I.ac[3] = (f2){a11, m};   I.ac[4] = (f2){a12, 0.f}; I.ac[5] = (f2){a22, m};
        I.bd[0] = I.bd[1] = I.bd[2] = 0.f;
        I.bo[0] = (f2){-h.z, h.z}; I.bo[1] = (f2){h.y, -h.y}; I.bo[2] = (f2){-h.x, h.x};
        const V3 n = V3{a00 * va.x + a01 * va.y + a02 * va.z, a01 * va.x + a11 * va.y + a12 * va.z,
                        a02 * va.x + a12 * va.y + a22 * va.z} + cross(h, vl);
        const V3 f = m * vl - cross(h, va);
        p = pack(cross(va, n) + cross(vl, f), cross(va, f));
    }
}

template <int J>
__device__ __forceinline__ void vel_outward(const P3& vp, float c, float s, float qd, P3& v)
{
    constexpr int AXJ = (int)kJoints[J].axis;
    v = to_child<J>(vp, c, s);
    pr<AXJ>(v).x += qd;
}

template <int J>
__device__ __forceinline__ void acc_outward(const P3& ap, const P3& vJ, float c, float s, float qd, const DynBody& b,
                                            float& qdd, P3& a)
{
    constexpr int AXJ = (int)kJoints[J].axis;
    a = to_child<J>(ap, c, s) + cross_axis<AXJ>(vJ, qd);
    const f2 d = b.U.x * a.x + b.U.y * a.y + b.U.z * a.z;             // (Ua . a.a, Ul . a.l)
    qdd = (b.u - (d.x + d.y)) * fast_rcp(b.D);
    pr<AXJ>(a).x += qdd;
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

// Penalty contact of ONE sample sphere (centre c in the frame of the body whose world pose is R, p and whose spatial
// velocity in body coordinates is vb) with the ground plane and the static axis-aligned box (the reference demo's scene
// extras, pioneer_knm_env.py:249-261; create_body_plane / create_body_box, bullet_scene.py:206-228): the contact force
// is added to the body's external spatial force in body coordinates.
__device__ __forceinline__ void sample_contact(const DynParams& D, const M3& R, V3 p, const P3& vb, V3 c, float radius, P3& fext)
{
    const V3 pos = p + mul(R, c);                           // world position of the sphere centre
    const V3 vw = mul(R, lin(vb) + cross(ang(vb), c));      // its world velocity
    V3 F = {0.f, 0.f, 0.f};
    if (D.has_ground) {
        // the pointer alone touches with its centre (the r01 form); with link contacts every sphere with its surface
        const float depth = D.ground_z - pos.z + (D.link_contacts ? radius : 0.f);
        const float fz = D.ckp * depth - D.ckd * vw.z;
        if (depth > 0.f && fz > 0.f) F.z += fz;
    }
    if (D.has_box) {
        // signed distance of the centre to the axis-aligned box and its outward normal
        const V3 dd = {pos.x - D.box_c[0], pos.y - D.box_c[1], pos.z - D.box_c[2]};
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
        const float depth = radius - sdf;
        const float fn = D.ckp * depth - D.ckd * dot(vw, nrm);
        if (depth > 0.f && fn > 0.f) F = F + fn * nrm;
    }
    // the scene's static bodies: wave-uniform loop over a small device array, the sample touches with its surface.  Same signed-distance forms as the oracle's contact_force_sphere.
#pragma unroll 1
    for (int b = 0; b < D.n_scene; ++b) {
        const SceneBody S = D.scene[b];
        const V3 dd = {pos.x - S.pos[0], pos.y - S.pos[1], pos.z - S.pos[2]};
        V3 nrm; float sdf;
        if (S.shape == 1) {                                // plane: unit world normal in rot[0..2]
            nrm = {S.rot[0], S.rot[1], S.rot[2]};
            sdf = dot(nrm, dd);
        } else if (S.shape == 3) {                         // sphere
            const float len = sqrtf(dot(dd, dd));
            sdf = len - S.size[0];
            nrm = len > 0.f ? (1.f / len) * dd : V3{0.f, 0.f, 1.f};
        } else {                                           // oriented box: into its frame, out again with the normal
            const V3 l = {S.rot[0] * dd.x + S.rot[3] * dd.y + S.rot[6] * dd.z, S.rot[1] * dd.x + S.rot[4] * dd.y + S.rot[7] * dd.z,
                          S.rot[2] * dd.x + S.rot[5] * dd.y + S.rot[8] * dd.z};
            const V3 q = {fabsf(l.x) - S.size[0], fabsf(l.y) - S.size[1], fabsf(l.z) - S.size[2]};
            const V3 o = {fmaxf(q.x, 0.f), fmaxf(q.y, 0.f), fmaxf(q.z, 0.f)};
            const float out2 = dot(o, o);
            V3 nl;
            if (out2 > 0.f) {
                const float len = sqrtf(out2);
                sdf = len;
                nl = {(l.x < 0.f ? -o.x : o.x) / len, (l.y < 0.f ? -o.y : o.y) / len, (l.z < 0.f ? -o.z : o.z) / len};
            } else {
                const int km = (q.x >= q.y && q.x >= q.z) ? 0 : (q.y >= q.z ? 1 : 2);
                sdf = comp(q, km);
                nl = {km == 0 ? (l.x < 0.f ? -1.f : 1.f) : 0.f, km == 1 ? (l.y < 0.f ? -1.f : 1.f) : 0.f,
                      km == 2 ? (l.z < 0.f ? -1.f : 1.f) : 0.f};
            }
            nrm = {S.rot[0] * nl.x + S.rot[1] * nl.y + S.rot[2] * nl.z, S.rot[3] * nl.x + S.rot[4] * nl.y + S.rot[5] * nl.z,
                   S.rot[6] * nl.x + S.rot[7] * nl.y + S.rot[8] * nl.z};
        }
        const float depth = radius - sdf;
        const float fn = D.ckp * depth - D.ckd * dot(vw, nrm);
        if (depth > 0.f && fn > 0.f) F = F + fn * nrm;
    }
    const V3 fb = mulT(R, F);                             // R^T F
    fext = fext + pack(cross(c, fb), fb);
}

// every sample sphere of moving body BODY (compile-time table kCapsules); without link contacts only the pointer
template <int BODY>
__device__ __forceinline__ void body_contacts(const DynParams& D, const M3& R, V3 p, const P3& vb, P3& fext)
{
    static_for<kNumCapsules>([&](auto ci_) {
        constexpr int ci = decltype(ci_)::value;
        if constexpr (kCapsules[ci].body == BODY) {
            static_for<kCapsules[ci].n>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                constexpr CapsuleDef K = kCapsules[ci];
                constexpr float t = K.n > 1 ? (float)((double)i / (double)(K.n - 1)) : 0.f;
                constexpr bool tip = (ci == kNumCapsules - 1) && (i == K.n - 1);
                const V3 c = tip ? V3{(float)kTipX, (float)kTipY, (float)kTipZ}
                                 : V3{K.ax + t * (K.bx - K.ax), K.ay + t * (K.by - K.ay), K.az + t * (K.bz - K.az)};
                if (tip || D.link_contacts) sample_contact(D, R, p, vb, c, K.radius < 0.f ? D.ptr_radius : K.radius, fext);
            });
        }
    });
}

// the external spatial force of all active contacts on every body, in body coordinates
__device__ __forceinline__ void contact_wrenches(const DynParams& D, const float (&c)[kDof], const float (&s)[kDof],
                                                 const P3 (&v)[kDof], P3 (&fext)[kDof])
{
    M3 R = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}}, Rn;
    V3 p = {0.f, 0.f, 0.f}, pn;
    pose_outward<0>(R, p, c[0], s[0], Rn, pn); R = Rn; p = pn;          // body 0 (rotator1 + hinge1) carries no samples
    pose_outward<1>(R, p, c[1], s[1], Rn, pn); R = Rn; p = pn; body_contacts<1>(D, R, p, v[1], fext[1]);
    pose_outward<2>(R, p, c[2], s[2], Rn, pn); R = Rn; p = pn; body_contacts<2>(D, R, p, v[2], fext[2]);
    pose_outward<3>(R, p, c[3], s[3], Rn, pn); R = Rn; p = pn; body_contacts<3>(D, R, p, v[3], fext[3]);
    pose_outward<4>(R, p, c[4], s[4], Rn, pn); R = Rn; p = pn; body_contacts<4>(D, R, p, v[4], fext[4]);
    pose_outward<5>(R, p, c[5], s[5], Rn, pn); R = Rn; p = pn; body_contacts<5>(D, R, p, v[5], fext[5]);
}

// qdd = ABA(q, qd, tau).  CONTACT: the penalty contacts' external forces are subtracted from the bodies' bias forces
// (a separate instantiation: the contact-free kernels carry none of that code or its registers)
// c, s: cos / sin of the joint angles (dyn_core carries them across the sub-steps)
// PHYS: bit 0 = contacts, bit 1 = the inertia-scaled motor (its acceleration requests in ades, torque cap tcap)
template <int PHYS>
__device__ __forceinline__ void aba(const DynParams& D, const DynModel& M, const float (&c)[kDof], const float (&s)[kDof],
                                    const float (&qd)[kDof], const float (&tau)[kDof], const float (&ades)[kDof], const float (&tcap)[kDof],
                                    float (&qdd)[kDof])
{
    constexpr bool CONTACT = (PHYS & 1) != 0, SCALED = (PHYS & 2) != 0;

    // pass 1: body velocities
    P3 v[kDof];
    const P3 v0 = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    vel_outward<0>(v0, c[0], s[0], qd[0], v[0]);
    vel_outward<1>(v[0], c[1], s[1], qd[1], v[1]);
    vel_outward<2>(v[1], c[2], s[2], qd[2], v[2]);
    vel_outward<3>(v[2], c[3], s[3], qd[3], v[3]);
    vel_outward<4>(v[3], c[4], s[4], qd[4], v[4]);
    vel_outward<5>(v[4], c[5], s[5], qd[5], v[5]);

    P3 fext[kDof];
    if (CONTACT) {
#pragma unroll
        for (int i = 0; i < kDof; ++i) fext[i] = v0;
        contact_wrenches(D, c, s, v, fext);
    }

    // pass 2: tip -> base
    DynBody B[kDof];
    SIp IA, IP; P3 pA, pP;
    rigid_body<5>(M, v[5], IA, pA);
    if (CONTACT) pA = pA - fext[5];
    rigid_body<4>(M, v[4], IP, pP);
    if (CONTACT) pP = pP - fext[4];
    aba_inward<5, SCALED>(IA, pA, v[5], qd[5], tau[5], ades[5], tcap[5], c[5], s[5], B[5], IP, pP);
    IA = IP; pA = pP; rigid_body<3>(M, v[3], IP, pP);
    if (CONTACT) pP = pP - fext[3];
    aba_inward<4, SCALED>(IA, pA, v[4], qd[4], tau[4], ades[4], tcap[4], c[4], s[4], B[4], IP, pP);
    IA = IP; pA = pP; rigid_body<2>(M, v[2], IP, pP);
    if (CONTACT) pP = pP - fext[2];
    aba_inward<3, SCALED>(IA, pA, v[3], qd[3], tau[3], ades[3], tcap[3], c[3], s[3], B[3], IP, pP);
    IA = IP; pA = pP; rigid_body<1>(M, v[1], IP, pP);
    if (CONTACT) pP = pP - fext[1];
    aba_inward<2, SCALED>(IA, pA, v[2], qd[2], tau[2], ades[2], tcap[2], c[2], s[2], B[2], IP, pP);
    IA = IP; pA = pP; rigid_body<0>(M, v[0], IP, pP);
    aba_inward<1, SCALED>(IA, pA, v[1], qd[1], tau[1], ades[1], tcap[1], c[1], s[1], B[1], IP, pP);
    IA = IP; pA = pP;
    aba_inward<0, SCALED>(IA, pA, v[0], qd[0], tau[0], ades[0], tcap[0], c[0], s[0], B[0], IP, pP);

    // pass 3: base -> tip; gravity as a base acceleration +g along z
    P3 a0 = {{0.f, 0.f}, {0.f, 0.f}, {0.f, D.gravity}}, a1;
    acc_outward<0>(a0, v[0], c[0], s[0], qd[0], B[0], qdd[0], a1); a0 = a1;
    acc_outward<1>(a0, v[1], c[1], s[1], qd[1], B[1], qdd[1], a1); a0 = a1;
    acc_outward<2>(a0, v[2], c[2], s[2], qd[2], B[2], qdd[2], a1); a0 = a1;
    acc_outward<3>(a0, v[3], c[3], s[3], qd[3], B[3], qdd[3], a1); a0 = a1;
    acc_outward<4>(a0, v[4], c[4], s[4], qd[4], B[4], qdd[4], a1); a0 = a1;
    acc_outward<5>(a0, v[5], c[5], s[5], qd[5], B[5], qdd[5], a1);
}

// ---------------------------------------------------------------------------------
// Phase A of a dynamics step for ONE env in this lane: kinematic command integration (the
// parity-mode integrator) + nsub sub-steps of ABA + PD.  In: the env's two state records as they
// lie in HBM (k0/k1/k2[p], see load_state_raw).  Out: the records with a, v, r updated, and the
// simulated q, qd.  Nothing is stored here: pnr::dyn_step_kernel hands the results to its pair
// lanes through LDS.
// ---------------------------------------------------------------------------------
// RAND = per-env link scales (domain randomisation): loaded from the dyn words; otherwise every scale
// is 1 and the model folds into literals.
// `in` carries the pointers and integrator constants the kernel received as preloaded leading arguments
// (so that the first loads and the integrator do not wait for the kernarg structs).
struct DynLead {
    const float4* state; const float* dyn; const float* actions;
    long long n; double dt, eps; float max_v_to_r;
};

// What a phase-A lane keeps in registers for its env across the steps of one launch.
struct DynLane {
    float a[kDof], v[kDof], r[kDof];            // the kinematic command state
    float q[kDof], qd[kDof];                    // the simulated joints
    float sc[kNumLinks], fric[kDof], damp[kDof];
    float cw[2][3];                             // the env's common words as loaded (target | potential, step, episode):
                                                // passed on to phase B with the first hand-off, not used after it
    float act[kDof];                            // the action of the NEXT advance, requested one step ahead
};

// All phase-A addresses are formed as (wave-uniform plane pointer incl. the workgroup's first env) + lane: the
// uniform part is scalar arithmetic and the loads take the SGPR-base + 32-bit-offset form.  Written as
// `ptr[i * n + e]` with a 64-bit per-lane `e`, the 40-odd plane addresses became v_mad_u64_u32 chains whose
// temporaries overlapped load destinations, so the compiler split the load burst with s_waitcnt vmcnt(0) —
// one more memory round trip per split before the first sub-step.
template <bool ACT_EM>
__device__ __forceinline__ void dyn_load_action(const float* __restrict__ actions, long long n, long long base, int lane,
                                                float (&act)[kDof])
{
    if (ACT_EM) {
        const float2* a2 = reinterpret_cast<const float2*>(actions + base * kDof);
        const float2 x0 = a2[3 * lane], x1 = a2[3 * lane + 1], x2 = a2[3 * lane + 2];
        act[0] = x0.x; act[1] = x0.y; act[2] = x1.x; act[3] = x1.y; act[4] = x2.x; act[5] = x2.y;
    } else {
#pragma unroll
        for (int i = 0; i < kDof; ++i) act[i] = (actions + (long long)i * n + base)[lane];
    }
}

// RAND = per-env link scales (domain randomisation): loaded from the dyn words; otherwise every scale
// is 1 and the model folds into literals.  Every load of the kernel is issued here, before the first wait.
template <bool ACT_EM, bool RAND>
__device__ __forceinline__ void dyn_lane_load(const DynLead& in, long long base, int lane, DynLane& L)
{
    const long long n = in.n;
    const long long n2 = 2 * n;
    float4 k0[2], k1[2], k2[2];
    const float4* s0 = in.state + 2 * base;                       // record 2 * env + p of plane 0 for this workgroup
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        k0[p] = s0[2 * lane + p]; k1[p] = (s0 + n2)[2 * lane + p]; k2[p] = (s0 + 2 * n2)[2 * lane + p];
    }
    // the 35 planar dyn words through ONE walking pointer (plane p at dyn + p * n): two address registers in all
    const float* w = in.dyn + base + lane;
#pragma unroll
    for (int i = 0; i < kDof; ++i) { L.q[i] = *w; w += n; }
#pragma unroll
    for (int i = 0; i < kDof; ++i) { L.qd[i] = *w; w += n; }
#pragma unroll
    for (int l = 0; l < kNumLinks; ++l) { L.sc[l] = RAND ? *w : 1.0f; w += n; }
#pragma unroll
    for (int i = 0; i < kDof; ++i) { L.fric[i] = *w; w += n; }
#pragma unroll
    for (int i = 0; i < kDof; ++i) { L.damp[i] = *w; w += n; }
    dyn_load_action<ACT_EM>(in.actions, n, base, lane, L.act);   // the first step's action, with everything else
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        L.a[3 * p] = k0[p].x; L.a[3 * p + 1] = k0[p].y; L.a[3 * p + 2] = k0[p].z; L.v[3 * p] = k0[p].w;
        L.v[3 * p + 1] = k1[p].x; L.v[3 * p + 2] = k1[p].y; L.r[3 * p] = k1[p].z; L.r[3 * p + 1] = k1[p].w;
        L.r[3 * p + 2] = k2[p].x;
        L.cw[p][0] = k2[p].y; L.cw[p][1] = k2[p].z; L.cw[p][2] = k2[p].w;
    }
}

// ---------------------------------------------------------------------------------
// Phase A of a dynamics step for ONE env in this lane: kinematic command integration (the
// parity-mode integrator) with the action of this step, then nsub sub-steps of ABA + PD.
// Nothing is stored here: pnr::dyn_step_kernel hands the results to its pair lanes through LDS.
// ---------------------------------------------------------------------------------
// The arithmetic of phase A, shared by the single-step and the rollout kernels (one text, so that both produce
// the same bits): command integration with the action latched for the next step, then nsub sub-steps of ABA + PD.
// PNR_DYN_LDS_MODEL=1 (a build-time A/B, north_star's "per-link spatial inertias staged in LDS"): the env's rigid-body model
// (6 masses, first moment and 6 inertia entries of body 6) and its 12 friction / damping coefficients are written to a
// per-lane LDS slice [kDynStageWords][64] once per step and re-read at the top of every sub-step instead of living in
// ~33 registers across the loop.  Default 0: everything in registers (the measured winner).
#ifndef PNR_DYN_LDS_MODEL
#define PNR_DYN_LDS_MODEL 0
#endif
constexpr int kDynStageWords = 33;

// WORLD (pnr_world_step): the sub-steps alone — no command integration, no teleport — with the per-joint motor table W.
template <int PHYS, bool WORLD = false>
__device__ __forceinline__ void dyn_core(const DynLead& in, const DynParams& D, float (&a)[kDof], float (&v)[kDof],
                                         float (&r)[kDof], float (&q)[kDof], float (&qd)[kDof], const float (&sc)[kNumLinks],
                                         const float (&fric_)[kDof], const float (&damp_)[kDof], const float (&act)[kDof],
                                         float* stage = nullptr, const JointMotorTable* W = nullptr)
{
    if constexpr (!WORLD) {
#pragma unroll
    for (int i = 0; i < kDof; ++i) {
        integrate_joint(a[i], v[i], r[i], in.max_v_to_r * (limit_hi(i) - limit_lo(i)), limit_lo(i), limit_hi(i),
                        in.dt, in.eps, v[i], r[i]);
        a[i] = act[i];
    }
    }

    DynModel M;
    build_model(sc, M);
    float fric[kDof], damp[kDof];
#pragma unroll
    for (int i = 0; i < kDof; ++i) { fric[i] = fric_[i]; damp[i] = damp_[i]; }
#if PNR_DYN_LDS_MODEL
    {
        float* w = stage + (threadIdx.x & 63);
#pragma unroll
        for (int i = 0; i < kDof; ++i) { w[i * 64] = M.m[i]; w[(6 + i) * 64] = fric[i]; w[(12 + i) * 64] = damp[i]; }
        w[18 * 64] = M.h6.x; w[19 * 64] = M.h6.y; w[20 * 64] = M.h6.z;
        w[21 * 64] = M.I6.r0.x; w[22 * 64] = M.I6.r0.y; w[23 * 64] = M.I6.r0.z;
        w[24 * 64] = M.I6.r1.y; w[25 * 64] = M.I6.r1.z; w[26 * 64] = M.I6.r2.z;
    }
#endif

    if (!WORLD && D.teleport) {   // resetJointState semantics: pioneer_knm_env.py:148, bullet_scene.py:157-165
#pragma unroll
        for (int i = 0; i < kDof; ++i) { q[i] = r[i]; qd[i] = 0.f; }
    }
    // wave-uniform options folded into the arithmetic once, so the sub-step loop carries no branches:
    // teleport = no motor torque (gains 0), no torque cap = cap at +inf
    const float kp = D.teleport ? 0.f : D.kp_eff, kd = D.teleport ? 0.f : D.kd;
    const float cpos = D.c_pos, vcap = D.v_cap;            // 0 and +inf for the plain PD motor: v_ask == v[i] exactly
    const float tcap = D.tau_max > 0.f ? D.tau_max : __builtin_inff();
    // cos / sin of the joint angles: evaluated once per step and then ROTATED by each sub-step's actual angle change
    // (|dq| = |qd| / 240 is normally a few hundredths of a radian; sin / cos of it by Taylor polynomials to dq^11 / dq^12:
    // exact to float32 up to |dq| = 1.3 — the parity suite's wildest transient spins at 265 rad/s = 1.1 rad per sub-step,
    // and polynomials two terms shorter left 7e-6 there, which that trajectory amplified to 1.4e-3 rad), as (c, s) pairs in
    // packed registers: ~10 instructions per joint and sub-step instead of the ~35 of a range-reduced sincos.
    // Ten rotations drift the pair's norm by < 1e-6, and the next step starts from the exact values again.
    f2 cs[kDof];
#pragma unroll
    for (int i = 0; i < kDof; ++i) { float sn, cn; sincos_bounded(q[i], sn, cn); cs[i] = (f2){cn, sn}; }
    for (int k = 0; k < D.nsub; ++k) {
        float tau[kDof], qdd[kDof], ades[kDof];
#if PNR_DYN_LDS_MODEL
        {   // re-read the staged model: the offset is opaque per iteration, so nothing is hoisted out of the loop
            int off = threadIdx.x & 63;
            asm volatile("" : "+v"(off));
            const float* w = stage + off;
#pragma unroll
            for (int i = 0; i < kDof; ++i) { M.m[i] = w[i * 64]; fric[i] = w[(6 + i) * 64]; damp[i] = w[(12 + i) * 64]; }
            M.h6 = {w[18 * 64], w[19 * 64], w[20 * 64]};
            const float i01 = w[22 * 64], i02 = w[23 * 64], i12 = w[25 * 64];
            M.I6 = {{w[21 * 64], i01, i02}, {i01, w[24 * 64], i12}, {i02, i12, w[26 * 64]}};
        }
#endif
#pragma unroll
        for (int i = 0; i < kDof; ++i) {
            float tq;
            if constexpr (WORLD) {      // the joint's own motor (wave-uniform table entries)
                const float rr = W->from_cmd[i] ? r[i] : W->r_ref[i], vr = W->from_cmd[i] ? v[i] : W->v_ref[i];
                const float dq = rr - q[i];
                const float v_ask = fminf(fmaxf(vr + W->cpos[i] * dq, -W->vcap[i]), W->vcap[i]);
                tq = W->kp[i] * dq + W->kd[i] * (v_ask - qd[i]);
                if constexpr ((PHYS & 2) != 0) { ades[i] = tq; tq = 0.f; }
                else { ades[i] = 0.f; tq = fminf(fmaxf(tq, -W->tcap[i]), W->tcap[i]); }
            } else {
            const float dq = r[i] - q[i];
            const float v_ask = fminf(fmaxf(v[i] + cpos * dq, -vcap), vcap);
            tq = kp * dq + kd * (v_ask - qd[i]);
            if constexpr ((PHYS & 2) != 0) { ades[i] = tq; tq = 0.f; }        // an acceleration request: scaled and capped inside the ABA
            else { ades[i] = 0.f; tq = fminf(fmaxf(tq, -tcap), tcap); }
            }
            tq -= damp[i] * qd[i];
            tq -= fric[i] * qd[i] * __builtin_amdgcn_rsqf(qd[i] * qd[i] + kFrictionEps * kFrictionEps);
            tau[i] = tq;
        }
        {
            float c[kDof], s[kDof];
#pragma unroll
            for (int i = 0; i < kDof; ++i) { c[i] = cs[i].x; s[i] = cs[i].y; }
            float tc[kDof];
#pragma unroll
            for (int i = 0; i < kDof; ++i) tc[i] = WORLD ? W->tcap[i] : tcap;
            aba<PHYS>(D, M, c, s, qd, tau, ades, tc, qdd);
        }
#pragma unroll
        for (int i = 0; i < kDof; ++i) {   // semi-implicit Euler + inelastic joint limits (selects, no branches)
            const float hi = limit_hi(i), lo = limit_lo(i);
            const float qdn = qd[i] + qdd[i] * D.dt_sub;
            const float qn = q[i] + qdn * D.dt_sub;
            const bool over = qn > hi, under = qn < lo;
            const float qnew = over ? hi : (under ? lo : qn);
            const float dq = qnew - q[i];                          // the angle actually turned (clamps included)
            q[i] = qnew;
            qd[i] = ((over && qdn > 0.f) || (under && qdn < 0.f)) ? 0.f : qdn;
            // (cd, sd) = (cos dq, sin dq) as one packed Horner chain in x = dq^2, then (c, s) <- cd (c, s) + sd (-s, c)
            const float x = dq * dq;
            f2 p = (f2){1.0f / 479001600.0f, -1.0f / 39916800.0f};
            p = p * x + (f2){-1.0f / 3628800.0f, 1.0f / 362880.0f};
            p = p * x + (f2){1.0f / 40320.0f, -1.0f / 5040.0f};
            p = p * x + (f2){-1.0f / 720.0f, 1.0f / 120.0f};
            p = p * x + (f2){1.0f / 24.0f, -1.0f / 6.0f};
            p = p * x + (f2){-0.5f, 1.0f};
            const float cd = p.x * x + 1.0f, sd = p.y * dq;
            const f2 rot = (f2){-cs[i].y, cs[i].x};
            cs[i] = cs[i] * cd + rot * sd;
        }
    }
}

template <bool ACT_EM, bool RAND, int PHYS>
__device__ __forceinline__ void dyn_lane_advance(const DynLead& in, const DynParams& D, long long base, int lane,
                                                 const float* __restrict__ next_actions, DynLane& L, float* stage = nullptr)
{
    float act[kDof];
#pragma unroll
    for (int i = 0; i < kDof; ++i) act[i] = L.act[i];
    // the action after this one is requested now: it arrives under the sub-steps (next_actions: null on the last step)
    if (next_actions) dyn_load_action<ACT_EM>(next_actions, in.n, base, lane, L.act);
    dyn_core<PHYS>(in, D, L.a, L.v, L.r, L.q, L.qd, L.sc, L.fric, L.damp, act, stage);
}

// The per-env parameter draws of a reset: Philox blocks 3..8 of the env's counter (the joints and the target
// use blocks 0..2, reset_env), 11 link-mass scales, 6 friction and 6 damping coefficients; the configured
// defaults without randomisation.
__device__ __forceinline__ void dyn_draw_params(const KParams& P, const DynParams& D, unsigned long long genv,
                                                uint32_t episode_drawn, float (&sc)[kNumLinks], float (&fric)[kDof],
                                                float (&damp)[kDof])
{
    float u[24] = {};
    if (D.randomize) {
#pragma unroll
        for (uint32_t b = 0; b < 6; ++b) {
            uint32_t o[4];
            philox4x32_10(episode_drawn, (uint32_t)genv, (uint32_t)(genv >> 32), 3 + b, P.seed_lo, P.seed_hi, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) u[4 * b + k] = (float)u01(o[k]);
        }
    }
#pragma unroll
    for (int l = 0; l < kNumLinks; ++l) sc[l] = D.randomize ? (float)(D.mass_lo + D.mass_span * (double)u[l]) : 1.0f;
#pragma unroll
    for (int j = 0; j < kDof; ++j) {
        fric[j] = D.randomize ? (float)(D.fric_lo + D.fric_span * (double)u[11 + j]) : D.joint_friction;
        damp[j] = D.randomize ? (float)(D.damp_lo + D.damp_span * (double)u[17 + j]) : D.joint_damping;
    }
}

// ---------------------------------------------------------------------------------
// Single-step form of phase A (pnr_step): load, dyn_core, re-pack, in one piece.  Kept next to the rollout
// kernel's dyn_lane_load + dyn_lane_advance: that split form has the same instruction counts but ran 2 % slower as
// the single-step kernel (A/B inside one library: 32.15 vs 32.85 us per 65 536-env step; SQ_WAIT_INST_ANY +40 %).
// In: nothing but the env index.  Out: the env's two state records as they lie in HBM (a, v, r updated), q, qd.
// ---------------------------------------------------------------------------------
template <bool ACT_EM, bool RAND, int PHYS>
__device__ __forceinline__ void dyn_substeps_lane(const DynLead& in, const DynParams& D, long long e,
                                                  float4 (&k0)[2], float4 (&k1)[2], float4 (&k2)[2],
                                                  float (&q)[kDof], float (&qd)[kDof], float* stage = nullptr)
{
    const long long n = in.n;
    const long long n2 = 2 * n;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        k0[p] = in.state[2 * e + p]; k1[p] = in.state[n2 + 2 * e + p]; k2[p] = in.state[2 * n2 + 2 * e + p];
    }
    // dynamics state: every load is issued here, before the first wait — with one wave per SIMD
    // (65 536 envs) nothing else hides a memory round trip
    float sc[kNumLinks], fric[kDof], damp[kDof];
#pragma unroll
    for (int i = 0; i < kDof; ++i) {
        q[i] = in.dyn[(long long)i * n + e]; qd[i] = in.dyn[(long long)(6 + i) * n + e];
        fric[i] = in.dyn[(long long)(23 + i) * n + e]; damp[i] = in.dyn[(long long)(29 + i) * n + e];
    }
#pragma unroll
    for (int l = 0; l < kNumLinks; ++l) sc[l] = RAND ? in.dyn[(long long)(12 + l) * n + e] : 1.0f;
    float a[kDof], v[kDof], r[kDof];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        a[3 * p] = k0[p].x; a[3 * p + 1] = k0[p].y; a[3 * p + 2] = k0[p].z; v[3 * p] = k0[p].w;
        v[3 * p + 1] = k1[p].x; v[3 * p + 2] = k1[p].y; r[3 * p] = k1[p].z; r[3 * p + 1] = k1[p].w;
        r[3 * p + 2] = k2[p].x;
    }
    float act[kDof];
    if (ACT_EM) {
        const float2* a2 = reinterpret_cast<const float2*>(in.actions + e * kDof);
        const float2 x0 = a2[0], x1 = a2[1], x2 = a2[2];
        act[0] = x0.x; act[1] = x0.y; act[2] = x1.x; act[3] = x1.y; act[4] = x2.x; act[5] = x2.y;
    } else {
#pragma unroll
        for (int i = 0; i < kDof; ++i) act[i] = in.actions[(long long)i * n + e];
    }
    dyn_core<PHYS>(in, D, a, v, r, q, qd, sc, fric, damp, act, stage);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        k0[p] = make_float4(a[3 * p], a[3 * p + 1], a[3 * p + 2], v[3 * p]);
        k1[p] = make_float4(v[3 * p + 1], v[3 * p + 2], r[3 * p], r[3 * p + 1]);
        k2[p].x = r[3 * p + 2];
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
    float sc[kNumLinks], fric[kDof], damp[kDof];
    dyn_draw_params(P, D, genv, episode_drawn, sc, fric, damp);
    // lane 0 writes the link scales, each lane its own joints' friction / damping
    if (p == 0) {
#pragma unroll
        for (int l = 0; l < kNumLinks; ++l) D.dyn[(long long)(12 + l) * n + e] = sc[l];
    }
#pragma unroll
    for (int i = 0; i < kJpl; ++i) {
        const int j = kJpl * p + i;
        // both candidates become plain register values first: left as `p ? x[a] : x[b]`, LLVM folds the
        // select into the address and the whole array is demoted to scratch (cf. lane_consts)
        float f0 = fric[i], f1 = fric[kJpl + i], d0 = damp[i], d1 = damp[kJpl + i];
        asm volatile("" : "+v"(f0), "+v"(f1), "+v"(d0), "+v"(d1));
        D.dyn[(long long)(23 + j) * n + e] = p ? f1 : f0;
        D.dyn[(long long)(29 + j) * n + e] = p ? d1 : d0;
    }
}

}  // namespace pnr

#pragma clang fp contract(off)
