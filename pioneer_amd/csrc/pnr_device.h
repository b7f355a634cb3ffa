// pnr_device.h — device-side building blocks of the kinematic ("parity mode")
// step: integrator, forward kinematics, reward, reset draws, obs packing.
//
// Work decomposition: TWO lanes per env (a lane pair 2e, 2e+1), three joints per
// lane.  Lane p of the pair owns joints 3p..3p+2: it integrates them, evaluates
// their 15 sin/cos pairs and writes their 63 observation entries; the 12 values
// the other half needs (cos/sin of the partner's joints for the forward
// kinematics) cross by DPP quad_perm, which never touches LDS.  A 64-lane wave
// therefore steps 32 envs and stages a 17.5 KB obs tile, so two waves fit per
// SIMD at 65 536 envs: one wave's stores overlap the other's arithmetic.
//
// Compiled with -ffp-contract=off: the integrator reproduces the reference's
// mixed float32/float64 arithmetic (NumPy 1.x promotion) operation by
// operation, so nothing may be fused behind its back.  Where a fused
// multiply-add is wanted (FK, polynomials) it is written as __builtin_fmaf.
//
// Reference lines are relative to xdralex/pioneer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pnr_model.h"

namespace pnr {

constexpr int kWave = 64;
constexpr int kLanesPerEnv = 2;
constexpr int kJpl = kDof / kLanesPerEnv;      // joints per lane
constexpr int kEnvsPerWave = kWave / kLanesPerEnv;
constexpr int kTileFloats = kEnvsPerWave * kObsDim;

// Kernel parameters (by value -> kernarg segment -> scalar loads).
struct KParams {
    float4* state;             // [3][2n] float4 planes (library-owned), see load_state
    const float* actions;      // step / rollout
    float* obs;
    float* reward;
    uint8_t* done;
    uint8_t* trunc;            // may be null
    float* info;               // may be null, [n][4]
    const uint8_t* mask;       // reset: may be null
    const float* joint_pos;    // reset: may be null, [n][6]
    const float* target_pos;   // reset: may be null, [n][3]
    long long n;
    unsigned long long env_off;
    unsigned seed_lo, seed_hi;
    int T;                     // steps per launch (rollout); 1 for step
    int max_steps;             // TimeLimit; 0 = off
    int auto_reset;
    int diag;                  // timing-only ablation bits (PNR_DIAG), 0 in production
    double dt, eps;            // pioneer_knm_env.py:60-61
    double tlo[3], tspan[3];   // target_lo, target_hi - target_lo
    float v_max[kDof];         // max_v_to_r * (r_hi - r_lo); the limits themselves are constexpr (pnr_model.h)
    float pot_m, pot_s;        // award_max - award_done, award_potential_slope
    float penalty, award_done, done_dist;
    float pad1;
    double done_dist_d;        // done_distance as the reference compares it (a Python float, pioneer_knm_env.py:160)
};

// Per-lane state: this lane's three joints plus the env's common words (held by
// both lanes of the pair after load).
struct LaneState {
    float a[kJpl], v[kJpl], r[kJpl];
    float tgt[3];
    float pot;
    uint32_t step, episode;
};

// LDS hand-off between the lanes of ONE wave (every workgroup here is a single wave).  A wave's DS
// instructions execute in issue order, so a ds_write followed by another lane's ds_read needs no
// s_barrier and, unlike __syncthreads(), no s_waitcnt vmcnt(0): outstanding global stores (the
// previous tile's obs flush) keep draining while the next tile is packed.  The fences only pin
// the compiler's ordering of LDS accesses.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// partner lane's value (lane ^ 1) by DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float xchg(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));
}

// State in HBM: three float4 planes [3][2n]; record (e, p) at index 2e + p:
//   plane 0: a[3p] a[3p+1] a[3p+2] v[3p]
//   plane 1: v[3p+1] v[3p+2] r[3p] r[3p+1]
//   plane 2: r[3p+2] C0 C1 C2     p = 0: C = target xyz; p = 1: C = potential, step_index, episode
// so a wave reads/writes each plane with one lane-contiguous 1-KiB dwordx4.
struct RawState { float4 p0, p1, p2; };

__device__ __forceinline__ RawState load_state_raw(const float4* __restrict__ st, long long n, long long rec)
{
    const long long n2 = 2 * n;
    return {st[rec], st[n2 + rec], st[2 * n2 + rec]};
}

__device__ __forceinline__ void unpack_state(const RawState& raw, int p, LaneState& s)
{
    const float4 p0 = raw.p0, p1 = raw.p1, p2 = raw.p2;
    s.a[0] = p0.x; s.a[1] = p0.y; s.a[2] = p0.z; s.v[0] = p0.w;
    s.v[1] = p1.x; s.v[2] = p1.y; s.r[0] = p1.z; s.r[1] = p1.w;
    s.r[2] = p2.x;
    const float o1 = xchg(p2.y), o2 = xchg(p2.z), o3 = xchg(p2.w);
    s.tgt[0] = p ? o1 : p2.y; s.tgt[1] = p ? o2 : p2.z; s.tgt[2] = p ? o3 : p2.w;
    s.pot = p ? p2.y : o1;
    s.step = __float_as_uint(p ? p2.z : o2);
    s.episode = __float_as_uint(p ? p2.w : o3);
}

__device__ __forceinline__ void load_state(const float4* __restrict__ st, long long n, long long rec, int p, LaneState& s)
{
    unpack_state(load_state_raw(st, n, rec), p, s);
}

__device__ __forceinline__ void store_state(float4* __restrict__ st, long long n, long long rec, int p, const LaneState& s)
{
    const long long n2 = 2 * n;
    st[rec] = make_float4(s.a[0], s.a[1], s.a[2], s.v[0]);
    st[n2 + rec] = make_float4(s.v[1], s.v[2], s.r[0], s.r[1]);
    st[2 * n2 + rec] = make_float4(s.r[2], p ? s.pot : s.tgt[0], p ? __uint_as_float(s.step) : s.tgt[1],
                                   p ? __uint_as_float(s.episode) : s.tgt[2]);
}

__device__ __forceinline__ void zero_state(LaneState& s)
{
#pragma unroll
    for (int i = 0; i < kJpl; ++i) { s.a[i] = 0.f; s.v[i] = 0.f; s.r[i] = 0.f; }
    s.tgt[0] = s.tgt[1] = s.tgt[2] = 0.f; s.pot = 0.f; s.step = 0; s.episode = 0;
}

// Per-joint constants of this lane's joints (joint 3p + i), built ONCE per kernel from constexpr
// locals.  Written this way on purpose: `p ? table[3 + i] : table[i]` makes LLVM fold the select
// into a dynamically indexed load from the constant array — a global load plus s_waitcnt vmcnt(0)
// (which also waits for every outstanding store) in the middle of the hot loop.
struct LaneConsts {
    float lim[kJpl];   // r_hi (= -r_lo)
    float lc[kJpl];    // cos(r_hi)
    float ls[kJpl];    // sin(r_hi)
};

__device__ __forceinline__ LaneConsts lane_consts(int p)
{
    constexpr float l0 = limit_hi(0), l1 = limit_hi(1), l2 = limit_hi(2), l3 = limit_hi(3), l4 = limit_hi(4), l5 = limit_hi(5);
    constexpr float c0 = kLimitCos[0], c1 = kLimitCos[1], c2 = kLimitCos[2], c3 = kLimitCos[3], c4 = kLimitCos[4], c5 = kLimitCos[5];
    constexpr float s0 = kLimitSin[0], s1 = kLimitSin[1], s2 = kLimitSin[2], s3 = kLimitSin[3], s4 = kLimitSin[4], s5 = kLimitSin[5];
    static_assert(kJpl == 3, "lane_consts is written for three joints per lane");
    LaneConsts k;
    k.lim[0] = p ? l3 : l0; k.lim[1] = p ? l4 : l1; k.lim[2] = p ? l5 : l2;
    k.lc[0] = p ? c3 : c0;  k.lc[1] = p ? c4 : c1;  k.lc[2] = p ? c5 : c2;
    k.ls[0] = p ? s3 : s0;  k.ls[1] = p ? s4 : s1;  k.ls[2] = p ? s5 : s2;
    return k;
}

// ---- sin/cos -------------------------------------------------------------------------
// np.sin / np.cos of the float32 observation pieces (pioneer_knm_env.py:195-203).
// Cody-Waite reduction by pi/2 with three float32 constants and FMA, then
// degree-7/8 minimax polynomials on [-pi/4, pi/4]; max abs error 9.3e-8 for
// |x| <= 2^16 (tests/test_gpu_sincos.py sweeps it against float64).  Every trig
// argument of the obs except the raw action is bounded by construction
// (|r| <= pi, |r - r_lo| <= 2 pi, |v| <= v_max); ~22 VALU ops per pair instead
// of ocml's sincosf with its inlined Payne-Hanek path.
__device__ __forceinline__ void sincos_bounded(float x, float& sn, float& cs)
{
    const float k = __builtin_rintf(x * 0x1.45f306p-1f);                 // x * 2/pi
    float r = __builtin_fmaf(-k, 0x1.921fb6p+0f, x);                     // pi/2 hi
    r = __builtin_fmaf(-k, -0x1.777a5cp-25f, r);                         // pi/2 mid
    r = __builtin_fmaf(-k, -0x1.ee59dap-50f, r);                         // pi/2 lo
    const float z = r * r;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
    const float s0 = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
    const float c0 = __builtin_fmaf(z * z, pc, __builtin_fmaf(z, -0.5f, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? c0 : s0;
    const float cc = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -ss : ss;
    cs = ((q + 1) & 2) ? -cc : cc;
}

// The raw action is not bounded by the env (RLlib clips it, the env does not):
// beyond 2^17 fall back to ocml's fully-reduced sincosf.
__device__ __forceinline__ void sincos_any(float x, float& sn, float& cs)
{
    if (__builtin_fabsf(x) <= 131072.0f) sincos_bounded(x, sn, cs);
    else sincosf(x, &sn, &cs);
}

// ---- integrator: pioneer_knm_env.py:113-146 -------------------------------------
// One joint.  Cast sequence = the reference's NumPy 1.x arithmetic (float32
// stores, float64 intermediates); bit-exact against oracle/pnr_oracle.c.
__device__ __forceinline__ void integrate_joint(float a0, float v0, float r0, float vmax, float rlo, float rhi,
                                                double dt, double eps, float& v_out, float& r_out)
{
    float v1 = (float)((double)v0 + (double)a0 * dt);            // :121
    double dt_p1 = dt, dt_p2 = 0.0;                              // :122-123
    const bool hi = v1 > vmax, lo = v1 < -vmax;                  // :125, :129
    if (hi || lo) {
        const float vsat = hi ? vmax : -vmax;
        const float num = vsat - v0;                             // float32 - float32
        double q = (double)num / ((double)a0 + eps);             // :126, :130 (eps added, not sign-matched)
        // np.clip(q, 0, dt), NaN-propagating
        q = (q != q) ? q : (q < 0.0 ? 0.0 : (q > dt ? dt : q));
        dt_p1 = q;
        dt_p2 = dt - dt_p1;                                      // :127, :131
        v1 = vsat;                                               // :128, :132
    }
    const float vs = v0 + v1;                                    // float32 + float32
    const double half = 0.5 * (double)vs;
    const double r1d = ((double)r0 + half * dt_p1) + (double)v1 * dt_p2;  // :134
    float r1 = (float)r1d;
    if (r1 >= rhi) { r1 = rhi; v1 = 0.0f; }                      // :135-137
    if (r1 <= rlo) { r1 = rlo; v1 = 0.0f; }                      // :139-141
    v_out = v1;
    r_out = r1;
}

// ---- forward kinematics of robot:pointer ---------------------------------------
// Replaces Item.pose() -> getLinkState(computeForwardKinematics=1)
// (bullet_scene.py:53-59) for the serial chain of pnr_model.h; unrolled over the
// constexpr joint table from the tip to the base.
__device__ __forceinline__ void fk_pointer(const float (&c)[kDof], const float (&s)[kDof], float (&p)[3])
{
    float x = (float)kTipX, y = (float)kTipY, z = (float)kTipZ;
#pragma unroll
    for (int j = kDof - 1; j >= 0; --j) {
        const float cj = c[j], sj = s[j];
        if (kJoints[j].axis == AX) {
            const float ny = __builtin_fmaf(cj, y, -(sj * z));
            const float nz = __builtin_fmaf(sj, y, cj * z);
            y = ny; z = nz;
        } else if (kJoints[j].axis == AY) {
            const float nx = __builtin_fmaf(cj, x, sj * z);
            const float nz = __builtin_fmaf(cj, z, -(sj * x));
            x = nx; z = nz;
        } else {
            const float nx = __builtin_fmaf(cj, x, -(sj * y));
            const float ny = __builtin_fmaf(sj, x, cj * y);
            x = nx; y = ny;
        }
        if (kJoints[j].ox != 0.0) x += (float)kJoints[j].ox;
        if (kJoints[j].oy != 0.0) y += (float)kJoints[j].oy;
        if (kJoints[j].oz != 0.0) z += (float)kJoints[j].oz;
    }
    p[0] = x; p[1] = y; p[2] = z;
}

// ---- Philox4x32-10 (Salmon et al. SC'11) ----------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u01(uint32_t x) { return (double)(x >> 8) * (1.0 / 16777216.0); }

// ---- reset: reset_world, pioneer_knm_env.py:76-105 --------------------------------
// Draws r ~ U(r_lo, r_hi), target ~ U(target_lo, target_hi) from
// Philox(key = seed, counter = (episode, global env id, block)): words 0..5 are
// the joints, 6..8 the target; a = v = 0; potential = 0 (quirk Q3);
// step_index = 0.  jp / tp (this ENV's rows) override the draws, as
// reset_world's joint_positions / target_position arguments do.  Both lanes of
// a pair run this and keep their own three joints.
__device__ __forceinline__ void reset_env(const KParams& P, const LaneConsts& K, LaneState& s, int p,
                                          unsigned long long genv, const float* jp, const float* tp)
{
    uint32_t w[12];
#pragma unroll
    for (uint32_t b = 0; b < 3; ++b) {
        uint32_t o[4];
        philox4x32_10(s.episode, (uint32_t)genv, (uint32_t)(genv >> 32), b, P.seed_lo, P.seed_hi, o);
        w[4 * b + 0] = o[0]; w[4 * b + 1] = o[1]; w[4 * b + 2] = o[2]; w[4 * b + 3] = o[3];
    }
#pragma unroll
    for (int i = 0; i < kJpl; ++i) {
        const double hi = (double)K.lim[i], lo = -hi;
        const uint32_t word = p ? w[kJpl + i] : w[i];
        const float drawn = (float)(lo + (hi - lo) * u01(word));             // :80-81
        s.r[i] = jp ? jp[kJpl * p + i] : drawn;                              // :94 (stored as float32)
        s.a[i] = 0.0f;                                                       // :92
        s.v[i] = 0.0f;                                                       // :93
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float drawn = (float)(P.tlo[k] + P.tspan[k] * u01(w[6 + k])); // :83-90
        s.tgt[k] = tp ? tp[k] : drawn;
    }
    s.pot = 0.0f;                                                            // :105
    s.step = 0;                                                              // bullet_env.py:99
    s.episode += 1;
}

// ---- pose-dependent quantities -----------------------------------------------------
struct Pose {
    float c[kJpl], s[kJpl];  // cos r, sin r of this lane's joints
    float ptr[3];            // pointer xyz
    float diff[3];
    float dist;
};

__device__ __forceinline__ void compute_pose(const LaneState& s, int p, Pose& q)
{
    float c6[kDof], s6[kDof];
#pragma unroll
    for (int i = 0; i < kJpl; ++i) {
        sincos_bounded(s.r[i], q.s[i], q.c[i]);
        const float oc = xchg(q.c[i]), os = xchg(q.s[i]);
        c6[i] = p ? oc : q.c[i];        s6[i] = p ? os : q.s[i];
        c6[kJpl + i] = p ? q.c[i] : oc; s6[kJpl + i] = p ? q.s[i] : os;
    }
    fk_pointer(c6, s6, q.ptr);
#pragma unroll
    for (int k = 0; k < 3; ++k) q.diff[k] = s.tgt[k] - q.ptr[k];          // :154
    q.dist = sqrtf(__builtin_fmaf(q.diff[2], q.diff[2],
                   __builtin_fmaf(q.diff[1], q.diff[1], q.diff[0] * q.diff[0])));  // :155
}

// ---- `done`, bit for bit: pioneer_knm_env.py:154-160 ------------------------------------
// `done = distance < done_distance` is a byte the caller branches on (episode end, +award_done, reset), so it is held to
// the integer bar: the reference forms it from Bullet's float64 link position and Python floats.  The float32 pose above is
// good to 3e-5; whenever its distance lies within kDoneBand of the threshold, the lanes concerned re-evaluate the forward
// kinematics and the distance in float64 from (double)r — the operation order of the float64 restatement
// (oracle/pnr_oracle.c orc_fk_pointer / orc_reward), no fused multiply-adds — and take the predicate from that.  The branch
// is rare (a pointer crosses the threshold at ~1 unit per step, the band is 2e-3 wide), pair-uniform (both lanes of an env
// hold the same float32 distance, so the DPP exchange below finds its partner live), and costs the common path one
// compare.
#ifndef PNR_DONE_BAND
#define PNR_DONE_BAND 1.0e-3f          // -DPNR_DONE_BAND=0.0f folds the branch away (the float32 predicate of rounds 1-4: A/B only)
#endif
constexpr float kDoneBand = PNR_DONE_BAND;

// The float64 constants of this rare path live in a table and are fetched into VECTOR registers by the lanes that need them
// (the index is made opaque, so the loads are vector loads inside the branch): as literals they would be 40 scalar
// registers materialised next to a kernel that already runs at the scalar-register limit, and the spills that makes room
// for them land on the common path (+112 instructions per step, +3.7 % on step_kernel at 65 536 envs, A/B r05).
constexpr double kPio2Inv = 6.36619772367581382433e-01, kPio2Hi = 1.57079632673412561417e+00, kPio2Lo = 6.07710050650619224932e-11;
__device__ const double kDoneTable[15 + 3 + 3 * kDof] = {
    kPio2Inv, kPio2Hi, kPio2Lo,
    // sin kernel, highest power first (Sun fdlibm __kernel_sin S6..S1), then cos kernel (C6..C1)
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06,
    -1.98412698298579493134e-04, 8.33333333332248946124e-03, -1.66666666666666324348e-01,
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07,
    2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02,
    kTipX, kTipY, kTipZ,
    kJoints[0].ox, kJoints[0].oy, kJoints[0].oz, kJoints[1].ox, kJoints[1].oy, kJoints[1].oz,
    kJoints[2].ox, kJoints[2].oy, kJoints[2].oz, kJoints[3].ox, kJoints[3].oy, kJoints[3].oz,
    kJoints[4].ox, kJoints[4].oy, kJoints[4].oz, kJoints[5].ox, kJoints[5].oy, kJoints[5].oz,
};

// sin / cos of a float64 argument bounded by the joint limits (|x| <= 2 pi here; accurate to < 1 ulp for |x| < 2^20):
// Cody-Waite reduction by pi/2 in two pieces (k * hi is exact: hi carries 33 bits), then the classic degree-13 / 14
// minimax kernels on [-pi/4, pi/4].  T = the table above.
__device__ __forceinline__ void sincos_f64(const double* __restrict__ T, double x, double& sn, double& cs)
{
    const double k = __builtin_rint(x * T[0]);
    double r = __builtin_fma(-k, T[1], x);
    r = __builtin_fma(-k, T[2], r);
    const double z = r * r;
    double ps = __builtin_fma(z, T[3], T[4]);
    ps = __builtin_fma(z, ps, T[5]);
    ps = __builtin_fma(z, ps, T[6]);
    ps = __builtin_fma(z, ps, T[7]);
    ps = __builtin_fma(z, ps, T[8]);
    const double s0 = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, T[9], T[10]);
    pc = __builtin_fma(z, pc, T[11]);
    pc = __builtin_fma(z, pc, T[12]);
    pc = __builtin_fma(z, pc, T[13]);
    pc = __builtin_fma(z, pc, T[14]);
    const double c0 = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)k;
    const double ss = (q & 1) ? c0 : s0;
    const double cc = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -ss : ss;
    cs = ((q + 1) & 2) ? -cc : cc;
}

// orc_fk_pointer's arithmetic: plain float64 multiplies and adds from the tip to the base, over the constexpr joint table
// (which offsets exist is decided at compile time; their values come from the table)
__device__ __forceinline__ void fk_pointer_f64(const double* __restrict__ T, const double (&c)[kDof], const double (&s)[kDof], double (&p)[3])
{
    double x = T[15], y = T[16], z = T[17];
#pragma unroll
    for (int j = kDof - 1; j >= 0; --j) {
        const double cj = c[j], sj = s[j];
        if (kJoints[j].axis == AX) {
            const double ny = cj * y - sj * z, nz = sj * y + cj * z;
            y = ny; z = nz;
        } else if (kJoints[j].axis == AY) {
            const double nx = cj * x + sj * z, nz = -sj * x + cj * z;
            x = nx; z = nz;
        } else {
            const double nx = cj * x - sj * y, ny = sj * x + cj * y;
            x = nx; y = ny;
        }
        if (kJoints[j].ox != 0.0) x += T[18 + 3 * j];
        if (kJoints[j].oy != 0.0) y += T[19 + 3 * j];
        if (kJoints[j].oz != 0.0) z += T[20 + 3 * j];
    }
    p[0] = x; p[1] = y; p[2] = z;
}

// `s` holds this lane's three joints (r) and the env's target; `dist` is the pair's float32 distance.
__device__ __forceinline__ bool done_predicate(const LaneState& s, int p, float dist, float done_dist, double done_dist_d)
{
    bool done = dist < done_dist;
    if (__builtin_fabsf(dist - done_dist) < kDoneBand) {
        int zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));      // opaque, per-lane: the table is read with vector loads, here
        const double* __restrict__ T = kDoneTable + zero;
        double c6[kDof], s6[kDof];
#pragma unroll
        for (int i = 0; i < kJpl; ++i) {
            const float mine = s.r[i], other = xchg(mine);
            sincos_f64(T, (double)(p ? other : mine), s6[i], c6[i]);
            sincos_f64(T, (double)(p ? mine : other), s6[kJpl + i], c6[kJpl + i]);
        }
        double ptr[3];
        fk_pointer_f64(T, c6, s6, ptr);
        const double d0 = (double)s.tgt[0] - ptr[0], d1 = (double)s.tgt[1] - ptr[1], d2 = (double)s.tgt[2] - ptr[2];
        done = __builtin_sqrt(d0 * d0 + d1 * d1 + d2 * d2) < done_dist_d;
    }
    return done;
}

// ---- observation: observe(), pioneer_knm_env.py:184-211 -----------------------------
// Sink::putj(f, v) writes per-joint feature f of this lane's joint column (the
// sink adds 3p); Sink::putt(f0, f1, v) writes tail feature f0 (lane 0) or f1
// (lane 1).  Tail pieces (126..136): lane 0 writes pointer + target, lane 1
// diff + distance + potential.
// [18:54]: r_lo, cos r_lo, sin r_lo, r_hi, cos r_hi, sin r_hi — the same 36 numbers for every env
// and every step.  With an LDS-tile sink they are written ONCE per kernel (the tile slots are never
// overwritten by anything else), at kernel start while the state loads are in flight.
template <class Sink>
__device__ __forceinline__ void emit_obs_const(const LaneConsts& K, Sink& out)
{
#pragma unroll
    for (int i = 0; i < kJpl; ++i) {
        const float lim = K.lim[i], lc = K.lc[i], ls = K.ls[i];
        out.putj(18 + i, -lim); out.putj(24 + i, lc); out.putj(30 + i, -ls);
        out.putj(36 + i, lim);  out.putj(42 + i, lc); out.putj(48 + i, ls);
    }
}

template <bool WITH_CONST = true, class Sink>
__device__ __forceinline__ void emit_obs(const LaneConsts& K, const LaneState& s, const Pose& q, int p, Sink& out)
{
    if (WITH_CONST) emit_obs_const(K, out);
#pragma unroll
    for (int i = 0; i < kJpl; ++i) {
        float sn, cs;
        const float lim = K.lim[i];
        // [0:18]  r, cos r, sin r
        out.putj(0 + i, s.r[i]); out.putj(6 + i, q.c[i]); out.putj(12 + i, q.s[i]);
        // [54:72] r - r_lo (float32 subtraction, :191)
        const float dlo = s.r[i] - (-lim);
        sincos_bounded(dlo, sn, cs);
        out.putj(54 + i, dlo); out.putj(60 + i, cs); out.putj(66 + i, sn);
        // [72:90] r_hi - r (:192)
        const float dhi = lim - s.r[i];
        sincos_bounded(dhi, sn, cs);
        out.putj(72 + i, dhi); out.putj(78 + i, cs); out.putj(84 + i, sn);
        // [90:108] v
        sincos_bounded(s.v[i], sn, cs);
        out.putj(90 + i, s.v[i]); out.putj(96 + i, cs); out.putj(102 + i, sn);
        // [108:126] a (the action just given, quirk Q1)
        sincos_any(s.a[i], sn, cs);
        out.putj(108 + i, s.a[i]); out.putj(114 + i, cs); out.putj(120 + i, sn);
    }
    // [126:137] p = 0: pointer xyz (126..128), target xyz (129..131)
    //           p = 1: diff (132..134), distance (135), potential (136)
#pragma unroll
    for (int k = 0; k < 3; ++k) out.putt(126 + k, 132 + k, p ? q.diff[k] : q.ptr[k]);
    out.putt(129, 135, p ? q.dist : s.tgt[0]);
    out.putt(130, 136, p ? s.pot : s.tgt[1]);
    out.putt0(131, s.tgt[2]);
}

// Sinks -------------------------------------------------------------------------
// env-major through an LDS tile [32][137]: the row stride of 137 dwords is odd, so
// a pair-interleaved ds_write_b32 is at worst 2-way conflicted (free for writes)
struct SinkLdsTile {
    float* row;    // tile + env_local * 137
    int jo;        // 3p
    int p;
    __device__ __forceinline__ void putj(int f, float v) { row[f + jo] = v; }
    __device__ __forceinline__ void putt(int f0, int f1, float v) { row[p ? f1 : f0] = v; }
    __device__ __forceinline__ void putt0(int f0, float v) { if (!p) row[f0] = v; }
};
// feature-major through an LDS tile [137][32]
struct SinkLdsFeatureTile {
    float* col;    // tile + env_local
    int jo;
    int p;
    __device__ __forceinline__ void putj(int f, float v) { col[(f + jo) * kEnvsPerWave] = v; }
    __device__ __forceinline__ void putt(int f0, int f1, float v) { col[(p ? f1 : f0) * kEnvsPerWave] = v; }
    __device__ __forceinline__ void putt0(int f0, float v) { if (!p) col[f0 * kEnvsPerWave] = v; }
};
// direct global stores with an element stride (masked reset only: rows / columns
// of other envs must stay untouched); stride 1 = env-major row, n = feature-major
struct SinkDirect {
    float* base; long long stride; int jo; int p; bool on;
    __device__ __forceinline__ void putj(int f, float v) { if (on) base[(long long)(f + jo) * stride] = v; }
    __device__ __forceinline__ void putt(int f0, int f1, float v) { if (on) base[(long long)(p ? f1 : f0) * stride] = v; }
    __device__ __forceinline__ void putt0(int f0, float v) { if (on && !p) base[(long long)f0 * stride] = v; }
};

// Streaming outputs (observations, rewards, flags) are written once and read by another kernel much
// later: non-temporal stores keep them from displacing the state planes in L2 and were worth 8 % at
// 65 536 envs and 31 % at 1 M envs per launch (178 -> 122 us).
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store(float4* dst, const float4& v)
{
    v4f_t x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<v4f_t*>(dst));
}
__device__ __forceinline__ void stream_store(float* dst, float v) { __builtin_nontemporal_store(v, dst); }
__device__ __forceinline__ void stream_store(uint8_t* dst, uint8_t v) { __builtin_nontemporal_store(v, dst); }

// Copy a wave's LDS tile (rows [0, nvalid) of 137 floats) to its contiguous
// place in an env-major obs batch with 16-byte lane-linear stores.
__device__ __forceinline__ void flush_tile(const float* __restrict__ lds, float* __restrict__ dst,
                                           int nvalid, int lane)
{
    const int total = nvalid * kObsDim;
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const int nvec = total >> 2;
        const float4* src4 = reinterpret_cast<const float4*>(lds);
        float4* dst4 = reinterpret_cast<float4*>(dst);
        for (int j = lane; j < nvec; j += kWave) stream_store(dst4 + j, src4[j]);
        for (int j = (nvec << 2) + lane; j < total; j += kWave) stream_store(dst + j, lds[j]);
    } else {
        for (int j = lane; j < total; j += kWave) stream_store(dst + j, lds[j]);
    }
}

// Flush a [137][32] feature tile to obs[f*n + tile0 + 0..31]: 8 lanes x 16 B cover
// one feature's 32 envs (128 B), a wave instruction covers eight features.
__device__ __forceinline__ void flush_feature_tile(const float* __restrict__ lds, float* __restrict__ dst,
                                                   long long n, int nvalid, int lane)
{
    if (nvalid == kEnvsPerWave && ((reinterpret_cast<uintptr_t>(dst) | (uintptr_t)(n * 4)) & 15u) == 0) {
        const int sub = lane >> 3, col4 = (lane & 7) * 4;
        for (int f0 = 0; f0 < kObsDim; f0 += 8) {
            const int f = f0 + sub;
            if (f < kObsDim) {
                const float4 v = *reinterpret_cast<const float4*>(lds + f * kEnvsPerWave + col4);
                stream_store(reinterpret_cast<float4*>(dst + (long long)f * n + col4), v);
            }
        }
    } else {
        const int col = lane & (kEnvsPerWave - 1), half = lane >> 5;
        for (int f = half; f < kObsDim; f += 2)
            if (col < nvalid) stream_store(dst + (long long)f * n + col, lds[f * kEnvsPerWave + col]);
    }
}

}  // namespace pnr
