// pnr_device.h — device-side building blocks of the kinematic ("parity mode")
// step: integrator, forward kinematics, reward, reset draws, obs packing.
//
// Compiled with -ffp-contract=off: the integrator reproduces the reference's
// mixed float32/float64 arithmetic (NumPy 1.x promotion) operation by
// operation, so nothing may be fused behind its back.  Where a fused
// multiply-add is wanted (FK) it is written as __builtin_fmaf.
//
// Reference lines are relative to xdralex/pioneer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pnr_model.h"

namespace pnr {

// Kernel parameters (by value -> kernarg segment -> scalar loads).
struct KParams {
    float4* state;             // [6][n] float4 planes (library-owned)
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
    int pad0;
    double dt, eps;            // pioneer_knm_env.py:60-61
    double tlo[3], tspan[3];   // target_lo, target_hi - target_lo
    float v_max[kDof];         // max_v_to_r * (r_hi - r_lo); the limits themselves are constexpr (pnr_model.h)
    float pot_m, pot_s;        // award_max - award_done, award_potential_slope
    float penalty, award_done, done_dist;
    float pad1;
};

// Per-env state held in registers.
struct EnvState {
    float a[kDof], v[kDof], r[kDof];
    float tgt[3];
    float pot;
    uint32_t step, episode;
};

__device__ __forceinline__ void load_state(const float4* __restrict__ st, long long n, long long e, EnvState& s)
{
    const float4 p0 = st[0 * n + e], p1 = st[1 * n + e], p2 = st[2 * n + e];
    const float4 p3 = st[3 * n + e], p4 = st[4 * n + e], p5 = st[5 * n + e];
    s.a[0] = p0.x; s.a[1] = p0.y; s.a[2] = p0.z; s.a[3] = p0.w;
    s.a[4] = p1.x; s.a[5] = p1.y; s.v[0] = p1.z; s.v[1] = p1.w;
    s.v[2] = p2.x; s.v[3] = p2.y; s.v[4] = p2.z; s.v[5] = p2.w;
    s.r[0] = p3.x; s.r[1] = p3.y; s.r[2] = p3.z; s.r[3] = p3.w;
    s.r[4] = p4.x; s.r[5] = p4.y; s.tgt[0] = p4.z; s.tgt[1] = p4.w;
    s.tgt[2] = p5.x; s.pot = p5.y;
    s.step = __float_as_uint(p5.z); s.episode = __float_as_uint(p5.w);
}

__device__ __forceinline__ void store_state(float4* __restrict__ st, long long n, long long e, const EnvState& s)
{
    st[0 * n + e] = make_float4(s.a[0], s.a[1], s.a[2], s.a[3]);
    st[1 * n + e] = make_float4(s.a[4], s.a[5], s.v[0], s.v[1]);
    st[2 * n + e] = make_float4(s.v[2], s.v[3], s.v[4], s.v[5]);
    st[3 * n + e] = make_float4(s.r[0], s.r[1], s.r[2], s.r[3]);
    st[4 * n + e] = make_float4(s.r[4], s.r[5], s.tgt[0], s.tgt[1]);
    st[5 * n + e] = make_float4(s.tgt[2], s.pot, __uint_as_float(s.step), __uint_as_float(s.episode));
}

// ---- sin/cos -------------------------------------------------------------------------
// np.sin / np.cos of the float32 observation pieces (pioneer_knm_env.py:195-203).
// Cody-Waite reduction by pi/2 with three float32 constants and FMA, then
// degree-7/8 minimax polynomials on [-pi/4, pi/4]; max abs error 9.3e-8 for
// |x| <= 2^16 (tests/test_sincos.py sweeps it against float64).  Every trig
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
// Philox(key = seed, counter = (episode, global env id, block)); a = v = 0;
// potential = 0 (quirk Q3); step_index = 0.  jp / tp override the draws
// (reset_world's joint_positions / target_position arguments).
__device__ __forceinline__ void reset_env(const KParams& P, EnvState& s, unsigned long long genv,
                                          const float* jp, const float* tp)
{
    uint32_t w[12];
#pragma unroll
    for (uint32_t b = 0; b < 3; ++b) {
        uint32_t o[4];
        philox4x32_10(s.episode, (uint32_t)genv, (uint32_t)(genv >> 32), b, P.seed_lo, P.seed_hi, o);
        w[4 * b + 0] = o[0]; w[4 * b + 1] = o[1]; w[4 * b + 2] = o[2]; w[4 * b + 3] = o[3];
    }
#pragma unroll
    for (int i = 0; i < kDof; ++i) {
        const double lo = (double)limit_lo(i), hi = (double)limit_hi(i);
        const float drawn = (float)(lo + (hi - lo) * u01(w[i]));          // :80-81
        s.r[i] = jp ? jp[i] : drawn;                                       // :94 (stored as float32)
        s.a[i] = 0.0f;                                                     // :92
        s.v[i] = 0.0f;                                                     // :93
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float drawn = (float)(P.tlo[k] + P.tspan[k] * u01(w[6 + k])); // :83-90
        s.tgt[k] = tp ? tp[k] : drawn;
    }
    s.pot = 0.0f;                                                          // :105
    s.step = 0;                                                            // bullet_env.py:99
    s.episode += 1;
}

// ---- pose-dependent quantities -----------------------------------------------------
struct Pose {
    float c[kDof], s[kDof];  // cos r, sin r
    float ptr[3];            // pointer xyz
    float diff[3];
    float dist;
};

__device__ __forceinline__ void compute_pose(const EnvState& s, Pose& q)
{
#pragma unroll
    for (int i = 0; i < kDof; ++i) sincos_bounded(s.r[i], q.s[i], q.c[i]);
    fk_pointer(q.c, q.s, q.ptr);
#pragma unroll
    for (int k = 0; k < 3; ++k) q.diff[k] = s.tgt[k] - q.ptr[k];          // :154
    q.dist = sqrtf(__builtin_fmaf(q.diff[2], q.diff[2],
                   __builtin_fmaf(q.diff[1], q.diff[1], q.diff[0] * q.diff[0])));  // :155
}

// ---- observation: observe(), pioneer_knm_env.py:184-211 -----------------------------
// Sink::put(f, value) receives feature f of this lane's env.
template <class Sink>
__device__ __forceinline__ void emit_obs(const KParams& P, const EnvState& s, const Pose& q, Sink& out)
{
#pragma unroll
    for (int i = 0; i < kDof; ++i) {
        float sn, cs;
        // [0:18]  r, cos r, sin r
        out.put(0 + i, s.r[i]); out.put(6 + i, q.c[i]); out.put(12 + i, q.s[i]);
        // [18:54] limits and their cos/sin (compile-time constants)
        out.put(18 + i, limit_lo(i)); out.put(24 + i, kLimitCos[i]); out.put(30 + i, -kLimitSin[i]);
        out.put(36 + i, limit_hi(i)); out.put(42 + i, kLimitCos[i]); out.put(48 + i, kLimitSin[i]);
        // [54:72] r - r_lo (float32 subtraction, :191)
        const float dlo = s.r[i] - limit_lo(i);
        sincos_bounded(dlo, sn, cs);
        out.put(54 + i, dlo); out.put(60 + i, cs); out.put(66 + i, sn);
        // [72:90] r_hi - r (:192)
        const float dhi = limit_hi(i) - s.r[i];
        sincos_bounded(dhi, sn, cs);
        out.put(72 + i, dhi); out.put(78 + i, cs); out.put(84 + i, sn);
        // [90:108] v
        sincos_bounded(s.v[i], sn, cs);
        out.put(90 + i, s.v[i]); out.put(96 + i, cs); out.put(102 + i, sn);
        // [108:126] a (the action just given, quirk Q1)
        sincos_any(s.a[i], sn, cs);
        out.put(108 + i, s.a[i]); out.put(114 + i, cs); out.put(120 + i, sn);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        out.put(126 + k, q.ptr[k]);   // pointer xyz
        out.put(129 + k, s.tgt[k]);   // target xyz
        out.put(132 + k, q.diff[k]);  // target - pointer
    }
    out.put(135, q.dist);
    out.put(136, s.pot);
}

// Sinks -------------------------------------------------------------------------
// feature-major [137][n]: lane-contiguous dword stores, coalesced as is
struct SinkFeatureMajor {
    float* base; long long n; bool valid;
    __device__ __forceinline__ void put(int f, float v) { if (valid) base[(long long)f * n] = v; }
};
// env-major through an LDS tile [64][137]: row stride 137 dwords (odd) -> the 64
// lanes of a ds_write_b32 hit distinct banks
struct SinkLdsTile {
    float* row;
    __device__ __forceinline__ void put(int f, float v) { row[f] = v; }
};
// env-major direct (masked reset only: rows of other envs must stay untouched)
struct SinkRowDirect {
    float* row; bool valid;
    __device__ __forceinline__ void put(int f, float v) { if (valid) row[f] = v; }
};

// feature-major through an LDS tile [137][64]: lane-contiguous ds_write_b32, then
// 16-byte stores where 16 lanes cover one feature's 64 envs (256 B) and a wave
// instruction covers four features
struct SinkLdsFeatureTile {
    float* col;   // tile + lane
    __device__ __forceinline__ void put(int f, float v) { col[f * 64] = v; }
};

// Flush a [137][64] feature tile to obs[f*n + tile0 + 0..63].
__device__ __forceinline__ void flush_feature_tile(const float* __restrict__ lds, float* __restrict__ dst,
                                                   long long n, int nvalid, int lane)
{
    // dst = obs_t + tile0; row f lives at dst + f*n
    if (nvalid == 64 && ((reinterpret_cast<uintptr_t>(dst) | (uintptr_t)(n * 4)) & 15u) == 0) {
        const int sub = lane >> 4, col4 = (lane & 15) * 4;
        for (int f0 = 0; f0 < kObsDim; f0 += 4) {
            const int f = f0 + sub;
            if (f < kObsDim) {
                const float4 v = *reinterpret_cast<const float4*>(lds + f * 64 + col4);
                *reinterpret_cast<float4*>(dst + (long long)f * n + col4) = v;
            }
        }
    } else {
        for (int f = 0; f < kObsDim; ++f)
            if (lane < nvalid) dst[(long long)f * n + lane] = lds[f * 64 + lane];
    }
}

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
        for (int j = lane; j < nvec; j += 64) dst4[j] = src4[j];
        for (int j = (nvec << 2) + lane; j < total; j += 64) dst[j] = lds[j];
    } else {
        for (int j = lane; j < total; j += 64) dst[j] = lds[j];
    }
}

}  // namespace pnr
