// pnr_api.hip — kernels + C ABI (include/pioneer_amd.h) of the MI355X-native
// Pioneer-arm engine.  gfx950 only; no CPU fallback.
//
// Execution shape: a lane PAIR per env (three joints per lane), one 64-lane wave
// = 32 envs per workgroup.  State lives in HBM as three float4 planes [3][2n]
// (24 words per env), so a wave moves each plane with one 1-KiB dwordx4
// instruction.  Observations are staged through a 17.5 KB LDS tile and leave as
// 16-byte stores: env-major rows of a wave are one contiguous 17.5 KB span;
// feature-major columns leave as 128-B segments, eight features per
// instruction, as non-temporal stores (write-once streaming data must not churn
// the L2s that hold the state planes).  The grid is persistent (<= 2 048
// one-wave workgroups striding over 32-env tiles, next tile prefetched ahead of
// the current tile's stores: a wave's VMEM operations retire in order), and the
// LDS hand-off inside the single-wave workgroup is a compiler-only barrier.
// Envs never interact, so there is no cross-workgroup traffic and block -> XCD
// placement only matters for L2 residency of the state planes (block b touches
// the same lines every launch).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/pioneer_amd.h"
#include "pnr_device.h"
#include "pnr_dyn.h"
#include "pnr_ppo.h"
#include "pnr_mlp.h"

namespace pnr {

// ---------------------------------------------------------------------------------
// step / rollout kernel: BulletEnv.step (bullet_env.py:192-197) for T steps.
// One wave = 32 envs (lane pair per env), one wave per workgroup.
// ---------------------------------------------------------------------------------
// The leading scalar parameters repeat P.state / P.actions / P.n / P.dt / P.eps and carry max_v_to_r (v_max
// is formed from it and the constexpr limits): plain leading arguments (up to 14 dwords) are preloaded into SGPRs by the command processor (-mllvm
// -amdgpu-kernarg-preload-count), so neither the first state and action loads nor the integrator wait for
// a kernarg fetch; the by-value struct, needed from the reward block on, is fetched behind them.
template <bool OBS_EM, bool ACT_EM>
__global__ __launch_bounds__(kWave) void step_kernel(float4* __restrict__ state_, const float* __restrict__ actions_,
                                                     const long long n_, const double dt_, const double eps_,
                                                     const float max_v_to_r_, const KParams P)
{
    __shared__ __attribute__((aligned(16))) float tile[kTileFloats];

    const int lane = threadIdx.x;
    const int p = lane & 1;                 // which half of the env's joints
    const int el = lane >> 1;               // env within the wave's tile
    const long long n = n_;
    const long long ntiles = (n + kEnvsPerWave - 1) / kEnvsPerWave;

    // PNR_DIAG timing-only ablations (outputs are wrong when set; see DESIGN.md "Where the time goes")
    const bool diag_noflush = P.diag & 2, diag_noemit = P.diag & 4, diag_nostate = P.diag & 8;

    const LaneConsts K = lane_consts(p);
    // v_max = max_v_to_r * (r_hi - r_lo) (pioneer_knm_env.py:57), the same float32 product pnr_get_constants forms
    const float vmax[kJpl] = {max_v_to_r_ * (K.lim[0] - (-K.lim[0])), max_v_to_r_ * (K.lim[1] - (-K.lim[1])),
                              max_v_to_r_ * (K.lim[2] - (-K.lim[2]))};

    // Persistent tile loop: the grid is capped (host: <= 8 waves per CU) and every wave strides over
    // tiles.  The NEXT tile's state and first action are requested before the current tile is
    // processed, so they never queue behind this CU's own obs stores.
    const auto load_act0 = [&](long long e_, float (&a_)[kJpl]) {
        if (ACT_EM) {
            const float* a3 = actions_ + e_ * kDof + kJpl * p;         // 12 B per lane, lanes contiguous
            a_[0] = a3[0]; a_[1] = a3[1]; a_[2] = a3[2];
        } else {
#pragma unroll
            for (int i = 0; i < kJpl; ++i) a_[i] = actions_[(long long)(kJpl * p + i) * n + e_];
        }
    };

    long long tix = blockIdx.x;
    RawState raw = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    float act0[kJpl] = {0.f, 0.f, 0.f};
    if (tix < ntiles && tix * kEnvsPerWave + el < n) {
        raw = load_state_raw(state_, n, 2 * tix * kEnvsPerWave + lane);
        load_act0(tix * kEnvsPerWave + el, act0);
    }
    bool first_tile = true;

    // the 36 constant obs entries of this lane's tile slots: once per kernel, under the load latency
    if (OBS_EM) { SinkLdsTile sink{tile + el * kObsDim, kJpl * p, p}; emit_obs_const(K, sink); }
    else { SinkLdsFeatureTile sink{tile + el, kJpl * p, p}; emit_obs_const(K, sink); }

    for (; tix < ntiles; tix += gridDim.x) {
    const long long tile0 = tix * kEnvsPerWave;
    const long long e = tile0 + el;
    const long long rec = 2 * tile0 + lane; // state record index (2e + p)
    const bool valid = e < n;               // the pair shares `valid`, so DPP partners are live
    const int nvalid = (int)((n - tile0) < kEnvsPerWave ? (n - tile0) : kEnvsPerWave);

    LaneState s;
    unpack_state(raw, p, s);                // all-zero records for lanes past the end
    float act_first[kJpl] = {act0[0], act0[1], act0[2]};

    // prefetch the next tile
    {
        const long long nt = tix + gridDim.x;
        if (nt < ntiles && nt * kEnvsPerWave + el < n) {
            raw = load_state_raw(state_, n, 2 * nt * kEnvsPerWave + lane);
            load_act0(nt * kEnvsPerWave + el, act0);
        } else {
            raw = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        }
    }

    if (P.diag & 16) { if (valid) P.reward[e] = s.pot; continue; }   // launch + state-load floor

    for (int t = 0; t < P.T; ++t) {
        {
            // -- action of this step (this lane's three joints) ----------------------
            // this step's action was requested one step (or one tile) ago; request the next one now,
            // ahead of this step's obs stores (VMEM ops of a wave retire in order)
            float act[kJpl] = {act_first[0], act_first[1], act_first[2]};
            if (t + 1 < P.T && valid) {
                const float* A = actions_ + (long long)(t + 1) * n * kDof;
                if (ACT_EM) {
                    const float* a3 = A + e * kDof + kJpl * p;
                    act_first[0] = a3[0]; act_first[1] = a3[1]; act_first[2] = a3[2];
                } else {
#pragma unroll
                    for (int i = 0; i < kJpl; ++i) act_first[i] = A[(long long)(kJpl * p + i) * n + e];
                }
            }
            // -- act(): integrate the PREVIOUS action, then latch the new one -------
#pragma unroll
            for (int i = 0; i < kJpl; ++i) {
                const float lim = K.lim[i];
                if (!(P.diag & 32)) integrate_joint(s.a[i], s.v[i], s.r[i], vmax[i], -lim, lim, dt_, eps_, s.v[i], s.r[i]);
            }
#pragma unroll
            for (int i = 0; i < kJpl; ++i) s.a[i] = act[i];           // :144 (quirk Q1)
        }
        s.step += 1;                                                  // bullet_env.py:193

        Pose q;
        compute_pose(s, p, q);

        // -- reward block, pioneer_knm_env.py:157-165 (both lanes, identical) ----------
        const float old_pot = s.pot;
        const float pot = P.pot_m / (q.dist / P.pot_s + 1.0f);        // :232-236
        const bool done = q.dist < P.done_dist;                       // :160
        const float r_pot = pot - old_pot;
        const float r_step = -P.penalty;
        const float r_done = done ? P.award_done : 0.0f;
        const float rw = (r_pot + r_step) + r_done;                   // :165
        s.pot = pot;
        // gym.wrappers.TimeLimit: truncated = elapsed >= max and not done
        const bool trunc = (P.max_steps > 0) && (s.step >= (uint32_t)P.max_steps) && !done;

        if (valid && p == 0) {
            const long long o = (long long)t * n + e;
            stream_store(P.reward + o, rw);
            stream_store(P.done + o, (uint8_t)done);
            if (P.trunc) stream_store(P.trunc + o, (uint8_t)trunc);
            if (P.info) stream_store(reinterpret_cast<float4*>(P.info) + o, make_float4(r_pot, r_step, r_done, q.dist));
        }

        // -- in-kernel auto-reset (BulletEnv.reset as the sampler would call it) -----
        // `done`/`trunc` are identical in both lanes of a pair, so pairs stay together
        if (P.auto_reset && (done || trunc)) {
            reset_env(P, K, s, p, P.env_off + (unsigned long long)e, nullptr, nullptr);
            compute_pose(s, p, q);
        }

        // state goes out before the obs is packed: its stores drain under the LDS emit
        if (t == P.T - 1 && valid && !diag_nostate) store_state(state_, n, rec, p, s);

        // -- observe() ----------------------------------------------------------------
        float* obs_t = P.obs + (long long)t * n * kObsDim;
        if (t > 0 || !first_tile) wave_lds_sync();   // previous flush done before the tile is rewritten
        if (OBS_EM) {
            SinkLdsTile sink{tile + el * kObsDim, kJpl * p, p};
            if (!diag_noemit) emit_obs<false>(K, s, q, p, sink);
            wave_lds_sync();
            if (!diag_noflush) flush_tile(tile, obs_t + tile0 * kObsDim, nvalid, lane);
        } else {
            SinkLdsFeatureTile sink{tile + el, kJpl * p, p};
            emit_obs<false>(K, s, q, p, sink);
            wave_lds_sync();
            flush_feature_tile(tile, obs_t + tile0, n, nvalid, lane);
        }
    }
    first_tile = false;
    }   // tile loop
}

// ---------------------------------------------------------------------------------
// dynamics-mode step: ONE launch per pnr_step / pnr_rollout.  A workgroup is one wave and owns 64 envs.
//   phase A  one env per lane: command integration + the ABA sub-steps (pnr_dyn.h); a, v, r, q, qd of the
//            env stay in that lane's registers for all T steps of the launch and are handed to phase B
//            through LDS every step;
//   phase B  the lanes regroup as pairs (as in step_kernel) and finish two 32-env tiles: reward,
//            TimeLimit, auto-reset, observation through the LDS tile; the pair lanes keep target,
//            potential, step and episode counters of their envs in registers.  A reset is reported back
//            to the env's phase-A lane through a small LDS note (episode counter + the new r), which
//            re-draws the per-env parameters itself.
// The hand-off area is the head of the obs tile: both tiles' values are read into registers before the
// first observation is packed.  With T > 1 the obs stores of step t drain under the sub-steps of t + 1.
// ---------------------------------------------------------------------------------
constexpr int kDynEnvsPerWg = kWave;                       // phase A: one env per lane
constexpr int kHandRecFloats = 3 * 2 * kDynEnvsPerWg * 4;  // three float4 planes of 2 records per env
constexpr int kHandFloats = kHandRecFloats + 2 * kDof * kDynEnvsPerWg;
static_assert(kHandFloats <= kTileFloats, "the hand-off area must fit into the obs tile it aliases");

template <bool OBS_EM>
__device__ __forceinline__ void dyn_finish_tile(const KParams& P, const DynParams& D, const LaneConsts& K, const RawState& raw,
                                                const float (&dq)[kJpl], const float (&dqd)[kJpl], float* tile,
                                                long long tile0, int lane, bool tile_in_use)
{
    const int p = lane & 1, el = lane >> 1;
    const long long n = P.n;
    const long long e = tile0 + el;
    const bool valid = e < n;               // the pair shares `valid`, so DPP partners are live
    const int nvalid = (int)((n - tile0) < kEnvsPerWave ? (n - tile0) : kEnvsPerWave);

    LaneState s;
    unpack_state(raw, p, s);                // all-zero records for lanes past the end
    LaneState o = s;                        // what reward / obs see: the simulated q, qd
#pragma unroll
    for (int i = 0; i < kJpl; ++i) {
        o.r[i] = dq[i];
        // teleport = reference semantics: obs shows the env's own v (pioneer_knm_env.py:202)
        o.v[i] = D.teleport ? s.v[i] : dqd[i];
    }
    s.step += 1;                                                  // bullet_env.py:193
    o.step = s.step;

    Pose q;
    compute_pose(o, p, q);

    // -- reward block, pioneer_knm_env.py:157-165 (both lanes, identical) ----------
    const float old_pot = s.pot;
    const float pot = P.pot_m / (q.dist / P.pot_s + 1.0f);        // :232-236
    const bool done = q.dist < P.done_dist;                       // :160
    const float r_pot = pot - old_pot;
    const float r_step = -P.penalty;
    const float r_done = done ? P.award_done : 0.0f;
    const float rw = (r_pot + r_step) + r_done;                   // :165
    s.pot = pot;
    // gym.wrappers.TimeLimit: truncated = elapsed >= max and not done
    bool trunc = (P.max_steps > 0) && (s.step >= (uint32_t)P.max_steps) && !done;
    // a lane whose simulation diverged (non-finite pose) is cut like a time-out, so auto-reset recovers
    // it instead of carrying NaNs forever (kinematic mode keeps the reference's NaN-propagating behaviour)
    if (!(q.dist == q.dist && __builtin_fabsf(q.dist) <= 3.0e38f)) trunc = !done;

    if (valid && p == 0) {
        stream_store(P.reward + e, rw);
        stream_store(P.done + e, (uint8_t)done);
        if (P.trunc) stream_store(P.trunc + e, (uint8_t)trunc);
        if (P.info) stream_store(reinterpret_cast<float4*>(P.info) + e, make_float4(r_pot, r_step, r_done, q.dist));
    }

    // -- in-kernel auto-reset (BulletEnv.reset as the sampler would call it) -----
    // `done`/`trunc` are identical in both lanes of a pair, so pairs stay together
    o.pot = pot;
    const bool redraw = P.auto_reset && (done || trunc);
    if (redraw) {
        reset_env(P, K, s, p, P.env_off + (unsigned long long)e, nullptr, nullptr);
        if (valid) dyn_reset_lane(P, D, s, p, e, P.env_off + (unsigned long long)e, s.episode - 1);   // q = r, qd = 0, new draws
        o = s;
        compute_pose(o, p, q);
    }
    if (valid) {
        store_state(P.state, n, 2 * tile0 + lane, p, s);
        if (!redraw) {
#pragma unroll
            for (int i = 0; i < kJpl; ++i) {
                D.dyn[(long long)(kJpl * p + i) * n + e] = dq[i];
                D.dyn[(long long)(6 + kJpl * p + i) * n + e] = dqd[i];
            }
        }
    }

    // -- observe() ----------------------------------------------------------------
    if (tile_in_use) wave_lds_sync();       // previous flush done before the tile is rewritten
    if (OBS_EM) {
        SinkLdsTile sink{tile + el * kObsDim, kJpl * p, p};
        emit_obs<false>(K, o, q, p, sink);
        wave_lds_sync();
        flush_tile(tile, P.obs + tile0 * kObsDim, nvalid, lane);
    } else {
        SinkLdsFeatureTile sink{tile + el, kJpl * p, p};
        emit_obs<false>(K, o, q, p, sink);
        wave_lds_sync();
        flush_feature_tile(tile, P.obs + tile0, n, nvalid, lane);
    }
}

// Leading scalar arguments as in step_kernel: preloaded into SGPRs, they repeat P.state / D.dyn / P.actions /
// P.n / P.dt / P.eps and carry max_v_to_r.
template <bool OBS_EM, bool ACT_EM, bool RAND, int PHYS>
__global__ __launch_bounds__(kWave) void dyn_step_kernel(const float4* __restrict__ state_, const float* __restrict__ dyn_,
                                                         const float* __restrict__ actions_, const long long n_,
                                                         const double dt_, const double eps_, const float max_v_to_r_,
                                                         const KParams P, const DynParams D)
{
    __shared__ __attribute__((aligned(16))) float tile[kTileFloats];
    const int lane = threadIdx.x;
    const long long n = n_;
    const long long base = (long long)blockIdx.x * kDynEnvsPerWg;
    float4* hrec = reinterpret_cast<float4*>(tile);               // [3][2 * 64] records, index 2 * env + p
    float* hq = tile + kHandRecFloats;                            // [12][64]: q then qd

    // ---- phase A: one env per lane
    {
        const long long e = base + lane;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 k0[2] = {z4, z4}, k1[2] = {z4, z4}, k2[2] = {z4, z4};
        float q[kDof] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, qd[kDof] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const DynLead lead = {state_, dyn_, actions_, n_, dt_, eps_, max_v_to_r_};
        if (e < n) dyn_substeps_lane<ACT_EM, RAND, PHYS>(lead, D, e, k0, k1, k2, q, qd, tile);   // tile: free during phase A
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            hrec[2 * lane + p] = k0[p];
            hrec[2 * kDynEnvsPerWg + 2 * lane + p] = k1[p];
            hrec[4 * kDynEnvsPerWg + 2 * lane + p] = k2[p];
        }
#pragma unroll
        for (int i = 0; i < kDof; ++i) {
            hq[i * kDynEnvsPerWg + lane] = q[i];
            hq[(kDof + i) * kDynEnvsPerWg + lane] = qd[i];
        }
    }
    wave_lds_sync();

    // ---- phase B: lane pairs; both tiles' hand-off records leave LDS before the tile is reused
    const int p = lane & 1, el = lane >> 1;
    RawState raw0, raw1;
    float dq0[kJpl], dqd0[kJpl], dq1[kJpl], dqd1[kJpl];
    {
        const int r0 = lane, r1 = 2 * kEnvsPerWave + lane;        // record 2 * env + p of env el / env 32 + el
        raw0 = {hrec[r0], hrec[2 * kDynEnvsPerWg + r0], hrec[4 * kDynEnvsPerWg + r0]};
        raw1 = {hrec[r1], hrec[2 * kDynEnvsPerWg + r1], hrec[4 * kDynEnvsPerWg + r1]};
#pragma unroll
        for (int i = 0; i < kJpl; ++i) {
            dq0[i] = hq[(kJpl * p + i) * kDynEnvsPerWg + el];
            dqd0[i] = hq[(kDof + kJpl * p + i) * kDynEnvsPerWg + el];
            dq1[i] = hq[(kJpl * p + i) * kDynEnvsPerWg + kEnvsPerWave + el];
            dqd1[i] = hq[(kDof + kJpl * p + i) * kDynEnvsPerWg + kEnvsPerWave + el];
        }
    }
    wave_lds_sync();
    const LaneConsts K = lane_consts(p);
    // the 36 constant obs entries of this lane's tile slots: once per kernel
    if (OBS_EM) { SinkLdsTile sink{tile + el * kObsDim, kJpl * p, p}; emit_obs_const(K, sink); }
    else { SinkLdsFeatureTile sink{tile + el, kJpl * p, p}; emit_obs_const(K, sink); }

    dyn_finish_tile<OBS_EM>(P, D, K, raw0, dq0, dqd0, tile, base, lane, false);
    if (base + kEnvsPerWave < n)
        dyn_finish_tile<OBS_EM>(P, D, K, raw1, dq1, dqd1, tile, base + kEnvsPerWave, lane, true);
}


constexpr int kComFloats = 2 * kWave * 4;                  // common words of each lane's two envs between steps: float4 [2][64]
constexpr int kRstFloats = (1 + kDof) * kDynEnvsPerWg;     // reset notes: episode flag + new r [7][64]

struct DynTileRegs {      // what phase A handed over for this lane's record of one env
    RawState raw;         // a, v, r of the lane's three joints (+ the common words on the first step)
    float q[kJpl], qd[kJpl];
};

__device__ __forceinline__ void dyn_write_handoff(float* hand, int lane, const DynLane& L)
{
    float4* hrec = reinterpret_cast<float4*>(hand);               // [3][2 * 64] records, index 2 * env + p
    float* hq = hand + kHandRecFloats;                            // [12][64]: q then qd
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        hrec[2 * lane + p] = make_float4(L.a[3 * p], L.a[3 * p + 1], L.a[3 * p + 2], L.v[3 * p]);
        hrec[2 * kDynEnvsPerWg + 2 * lane + p] = make_float4(L.v[3 * p + 1], L.v[3 * p + 2], L.r[3 * p], L.r[3 * p + 1]);
        hrec[4 * kDynEnvsPerWg + 2 * lane + p] = make_float4(L.r[3 * p + 2], L.cw[p][0], L.cw[p][1], L.cw[p][2]);
    }
#pragma unroll
    for (int i = 0; i < kDof; ++i) {
        hq[i * kDynEnvsPerWg + lane] = L.q[i];
        hq[(kDof + i) * kDynEnvsPerWg + lane] = L.qd[i];
    }
}

__device__ __forceinline__ void dyn_read_handoff(const float* hand, int env, int p, DynTileRegs& g)
{
    const float4* hrec = reinterpret_cast<const float4*>(hand);
    const float* hq = hand + kHandRecFloats;
    const int r = 2 * env + p;
    g.raw = {hrec[r], hrec[2 * kDynEnvsPerWg + r], hrec[4 * kDynEnvsPerWg + r]};
#pragma unroll
    for (int i = 0; i < kJpl; ++i) {
        g.q[i] = hq[(kJpl * p + i) * kDynEnvsPerWg + env];
        g.qd[i] = hq[(kDof + kJpl * p + i) * kDynEnvsPerWg + env];
    }
}

// One 32-env tile of phase B.  The env's common words (target | potential, step, episode) arrive with the first
// hand-off; between the steps of a looped launch they wait in `com` (LDS), so that nothing of phase B stays in
// registers during the sub-steps.
template <bool OBS_EM>
__device__ __forceinline__ void dyn_rollout_tile(const KParams& P, const DynParams& D, const LaneConsts& K, const DynTileRegs& in,
                                                float4* com, float* tile, float* rst, long long tile0, int t, int lane,
                                                bool tile_in_use)
{
    const int p = lane & 1, el = lane >> 1;
    const long long n = P.n;
    const long long e = tile0 + el;
    const bool valid = e < n;               // the pair shares `valid`, so DPP partners are live
    const int nvalid = (int)((n - tile0) < kEnvsPerWave ? (n - tile0) : kEnvsPerWave);
    const bool last = t == P.T - 1;

    LaneState s;
    {
        RawState raw = in.raw;
        if (t > 0) { const float4 c = *com; raw.p2.y = c.y; raw.p2.z = c.z; raw.p2.w = c.w; }
        unpack_state(raw, p, s);            // a, v, r of this lane's joints + the env's common words
    }
    LaneState o = s;                        // what reward / obs see: the simulated q, qd
#pragma unroll
    for (int i = 0; i < kJpl; ++i) {
        o.r[i] = in.q[i];
        // teleport = reference semantics: obs shows the env's own v (pioneer_knm_env.py:202)
        o.v[i] = D.teleport ? s.v[i] : in.qd[i];
    }
    s.step += 1;                                                  // bullet_env.py:193
    o.step = s.step;

    Pose q;
    compute_pose(o, p, q);

    // -- reward block, pioneer_knm_env.py:157-165 (both lanes, identical) ----------
    const float old_pot = s.pot;
    const float pot = P.pot_m / (q.dist / P.pot_s + 1.0f);        // :232-236
    const bool done = q.dist < P.done_dist;                       // :160
    const float r_pot = pot - old_pot;
    const float r_step = -P.penalty;
    const float r_done = done ? P.award_done : 0.0f;
    const float rw = (r_pot + r_step) + r_done;                   // :165
    s.pot = pot;
    // gym.wrappers.TimeLimit: truncated = elapsed >= max and not done
    bool trunc = (P.max_steps > 0) && (s.step >= (uint32_t)P.max_steps) && !done;
    // a lane whose simulation diverged (non-finite pose) is cut like a time-out, so auto-reset recovers
    // it instead of carrying NaNs forever (kinematic mode keeps the reference's NaN-propagating behaviour)
    if (!(q.dist == q.dist && __builtin_fabsf(q.dist) <= 3.0e38f)) trunc = !done;

    if (valid && p == 0) {
        const long long oi = (long long)t * n + e;
        stream_store(P.reward + oi, rw);
        stream_store(P.done + oi, (uint8_t)done);
        if (P.trunc) stream_store(P.trunc + oi, (uint8_t)trunc);
        if (P.info) stream_store(reinterpret_cast<float4*>(P.info) + oi, make_float4(r_pot, r_step, r_done, q.dist));
    }

    // -- in-kernel auto-reset (BulletEnv.reset as the sampler would call it) -----
    // `done`/`trunc` are identical in both lanes of a pair, so pairs stay together
    o.pot = pot;
    const bool redraw = P.auto_reset && (done || trunc);
    if (redraw) {
        reset_env(P, K, s, p, P.env_off + (unsigned long long)e, nullptr, nullptr);
        if (valid) {
            dyn_reset_lane(P, D, s, p, e, P.env_off + (unsigned long long)e, s.episode - 1);   // q = r, qd = 0, new draws
            // note for the env's phase-A lane: the counter after the reset (>= 1) and the new joints
            const int env = el + (int)(tile0 & (kDynEnvsPerWg - 1));
            if (p == 0) rst[env] = __uint_as_float(s.episode);
#pragma unroll
            for (int i = 0; i < kJpl; ++i) rst[(1 + kJpl * p + i) * kDynEnvsPerWg + env] = s.r[i];
        }
        o = s;
        compute_pose(o, p, q);
    }
    if (valid && last) {
        store_state(P.state, n, 2 * tile0 + lane, p, s);
        if (!redraw) {
#pragma unroll
            for (int i = 0; i < kJpl; ++i) {
                D.dyn[(long long)(kJpl * p + i) * n + e] = in.q[i];
                D.dyn[(long long)(6 + kJpl * p + i) * n + e] = in.qd[i];
            }
        }
    }
    *com = make_float4(0.f, p ? s.pot : s.tgt[0], p ? __uint_as_float(s.step) : s.tgt[1],
                        p ? __uint_as_float(s.episode) : s.tgt[2]);

    // -- observe() ----------------------------------------------------------------
    float* obs_t = P.obs + (long long)t * n * kObsDim;
    if (tile_in_use) wave_lds_sync();       // previous flush done before the tile is rewritten
    if (OBS_EM) {
        SinkLdsTile sink{tile + el * kObsDim, kJpl * p, p};
        emit_obs<false>(K, o, q, p, sink);
        wave_lds_sync();
        flush_tile(tile, obs_t + tile0 * kObsDim, nvalid, lane);
    } else {
        SinkLdsFeatureTile sink{tile + el, kJpl * p, p};
        emit_obs<false>(K, o, q, p, sink);
        wave_lds_sync();
        flush_feature_tile(tile, obs_t + tile0, n, nvalid, lane);
    }
}

// pnr_rollout in dynamics mode: P.T steps in ONE launch.  Same two phases as dyn_step_kernel, but the env's
// a, v, r, q, qd and parameters stay in its phase-A lane's registers from step to step, the pair lanes keep the
// common words (target | potential, step, episode) in LDS between steps, and a reset travels back to the phase-A
// lane as a small LDS note (episode counter + the new r) from which that lane re-draws the parameters itself.
// The obs stores of step t drain under the sub-steps of step t + 1.  The hand-off has its own 9 KB here (the
// kernel runs one wave per SIMD anyway), so a tile's values are read right before that tile is finished and
// nothing of phase B is live during the sub-steps.
template <bool OBS_EM, bool ACT_EM, bool RAND, int PHYS>
__global__ __launch_bounds__(kWave) void dyn_rollout_kernel(const float4* __restrict__ state_, const float* __restrict__ dyn_,
                                                         const float* __restrict__ actions_, const long long n_,
                                                         const double dt_, const double eps_, const float max_v_to_r_,
                                                         const KParams P, const DynParams D)
{
    __shared__ __attribute__((aligned(16))) float lds[kTileFloats + kComFloats + kRstFloats + kHandFloats + (PNR_DYN_LDS_MODEL ? kDynStageWords * 64 : 0)];
    float* tile = lds;
    float4* com = reinterpret_cast<float4*>(lds + kTileFloats);   // [2][64] common words between steps, index tile * 64 + lane
    float* rst = lds + kTileFloats + kComFloats;                  // [7][64] reset notes, phase B -> phase A
    float* hand = lds + kTileFloats + kComFloats + kRstFloats;    // records + q, qd planes (kHandFloats)
    const int lane = threadIdx.x;
    const int p = lane & 1, el = lane >> 1;
    const long long n = n_;
    const long long base = (long long)blockIdx.x * kDynEnvsPerWg;
    const long long eA = base + lane;                             // phase A: this lane's env
    const bool liveA = eA < n;
    const DynLead lead = {state_, dyn_, actions_, n_, dt_, eps_, max_v_to_r_};

    DynLane L;
#pragma unroll
    for (int i = 0; i < kDof; ++i) { L.a[i] = L.v[i] = L.r[i] = L.q[i] = L.qd[i] = 0.f; L.fric[i] = L.damp[i] = 0.f; }
#pragma unroll
    for (int l = 0; l < kNumLinks; ++l) L.sc[l] = 1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { L.cw[0][k] = 0.f; L.cw[1][k] = 0.f; }
#pragma unroll
    for (int i = 0; i < kDof; ++i) L.act[i] = 0.f;
    if (liveA) dyn_lane_load<ACT_EM, RAND>(lead, base, lane, L);
    {   // the 36 constant obs entries of this lane's tile slots: once per kernel (nothing else writes them)
        const LaneConsts K0 = lane_consts(p);
        if (OBS_EM) { SinkLdsTile sink{tile + el * kObsDim, kJpl * p, p}; emit_obs_const(K0, sink); }
        else { SinkLdsFeatureTile sink{tile + el, kJpl * p, p}; emit_obs_const(K0, sink); }
    }

    const int T = P.T;
    for (int t = 0; t < T; ++t) {
        // ---- phase A: one env per lane
        if (liveA) dyn_lane_advance<ACT_EM, RAND, PHYS>(lead, D, base, lane, t + 1 < T ? actions_ + (long long)(t + 1) * n * kDof : nullptr, L,
                                                           lds + kTileFloats + kComFloats + kRstFloats + kHandFloats);
        dyn_write_handoff(hand, lane, L);
        rst[lane] = 0.f;                                // no reset noted yet (episode counters are >= 1)
        wave_lds_sync();

        // ---- phase B: lane pairs
        {
            // everything phase B derives from the lane id (pair constants, tile and output addresses) is re-derived
            // from an opaque copy each step: hoisted out of the t loop it would sit in registers during the sub-steps
            int lane_b = lane;
            asm volatile("" : "+v"(lane_b));
            const int pb = lane_b & 1, elb = lane_b >> 1;
            const LaneConsts Kb = lane_consts(pb);
            {
                DynTileRegs g;
                dyn_read_handoff(hand, elb, pb, g);
                dyn_rollout_tile<OBS_EM>(P, D, Kb, g, com + lane_b, tile, rst, base, t, lane_b, t > 0);
            }
            if (base + kEnvsPerWave < n) {
                DynTileRegs g;
                dyn_read_handoff(hand, kEnvsPerWave + elb, pb, g);
                dyn_rollout_tile<OBS_EM>(P, D, Kb, g, com + kWave + lane_b, tile, rst, base + kEnvsPerWave, t, lane_b, true);
            }
            // ---- back to phase A: envs that were reset continue from the new draw
            if (t + 1 < T) {
                wave_lds_sync();
                const uint32_t ep = __float_as_uint(rst[lane]);
                if (liveA && ep != 0u) {
#pragma unroll
                    for (int j = 0; j < kDof; ++j) {
                        L.r[j] = rst[(1 + j) * kDynEnvsPerWg + lane];
                        L.a[j] = 0.f; L.v[j] = 0.f; L.q[j] = L.r[j]; L.qd[j] = 0.f;
                    }
                    if (RAND) dyn_draw_params(P, D, P.env_off + (unsigned long long)eA, ep - 1u, L.sc, L.fric, L.damp);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// reset / observe kernel.  MODE 0: reset (mask / overrides), 1: observe only.
// OBS: 0 none, 1 feature-major direct (masked reset), 2 env-major via LDS tile
//      (all rows written), 3 env-major direct rows (masked reset), 4 feature-major
//      via LDS tile (all columns written).
// ---------------------------------------------------------------------------------
template <int MODE, int OBS, bool DYN>
__global__ __launch_bounds__(kWave) void reset_kernel(const KParams P, const DynParams D)
{
    __shared__ __attribute__((aligned(16))) float tile[(OBS == 2 || OBS == 4) ? kTileFloats : 4];
    const int lane = threadIdx.x;
    const int p = lane & 1, el = lane >> 1;
    const long long n = P.n;
    const long long tile0 = (long long)blockIdx.x * kEnvsPerWave;
    const long long e = tile0 + el;
    const long long rec = 2 * tile0 + lane;
    const bool valid = e < n;
    const int nvalid = (int)((n - tile0) < kEnvsPerWave ? (n - tile0) : kEnvsPerWave);

    const LaneConsts K = lane_consts(p);
    LaneState s;
    if (valid) load_state(P.state, n, rec, p, s);
    else zero_state(s);
    bool active = valid;
    if (MODE == 0) {
        if (valid && P.mask) active = P.mask[e] != 0;
        if (active) {   // both lanes of a pair take the same branch
            reset_env(P, K, s, p, P.env_off + (unsigned long long)e,
                      P.joint_pos ? P.joint_pos + e * kDof : nullptr,
                      P.target_pos ? P.target_pos + e * 3 : nullptr);
            store_state(P.state, n, rec, p, s);
            if (DYN) dyn_reset_lane(P, D, s, p, e, P.env_off + (unsigned long long)e, s.episode - 1);
        }
    }
    if (DYN && !(MODE == 0 && active)) {
        // observe the simulated joints (a freshly reset env has q = r, qd = 0 already in s)
#pragma unroll
        for (int i = 0; i < kJpl; ++i) {
            const float qi = valid ? D.dyn[(long long)(kJpl * p + i) * n + e] : 0.f;
            const float qdi = valid ? D.dyn[(long long)(6 + kJpl * p + i) * n + e] : 0.f;
            s.r[i] = qi;
            if (!D.teleport) s.v[i] = qdi;
        }
    }
    if (OBS != 0) {
        Pose q;
        compute_pose(s, p, q);   // every lane takes part: the DPP exchange needs live partners
        if (OBS == 1) {
            SinkDirect sink{P.obs + e, n, kJpl * p, p, active};
            emit_obs(K, s, q, p, sink);
        } else if (OBS == 2) {
            SinkLdsTile sink{tile + el * kObsDim, kJpl * p, p};
            emit_obs(K, s, q, p, sink);
            wave_lds_sync();
            flush_tile(tile, P.obs + tile0 * kObsDim, nvalid, lane);
        } else if (OBS == 4) {
            SinkLdsFeatureTile sink{tile + el, kJpl * p, p};
            emit_obs(K, s, q, p, sink);
            wave_lds_sync();
            flush_feature_tile(tile, P.obs + tile0, n, nvalid, lane);
        } else {
            SinkDirect sink{P.obs + e * kObsDim, 1, kJpl * p, p, active};
            emit_obs(K, s, q, p, sink);
        }
    }
}

// canonical planar words [24][n] (include/pioneer_amd.h) <-> the engine's pair records
__device__ __forceinline__ int word_of(int p, int plane, int comp)
{
    // which canonical word sits in (half p, plane, component)
    const int k = plane * 4 + comp;            // 0..11 within the half record
    if (k < 3) return 0 + 3 * p + k;           // a
    if (k < 6) return 6 + 3 * p + (k - 3);     // v
    if (k < 9) return 12 + 3 * p + (k - 6);    // r
    return (p ? 21 : 18) + (k - 9);            // target xyz | potential, step, episode
}

__global__ void state_to_words_kernel(const float4* __restrict__ st, uint32_t* __restrict__ w, long long n)
{
    const long long rec = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (rec >= 2 * n) return;
    const long long e = rec >> 1; const int p = (int)(rec & 1);
#pragma unroll
    for (int pl = 0; pl < kStatePlanes; ++pl) {
        const float4 v = st[(long long)pl * 2 * n + rec];
        w[(long long)word_of(p, pl, 0) * n + e] = __float_as_uint(v.x);
        w[(long long)word_of(p, pl, 1) * n + e] = __float_as_uint(v.y);
        w[(long long)word_of(p, pl, 2) * n + e] = __float_as_uint(v.z);
        w[(long long)word_of(p, pl, 3) * n + e] = __float_as_uint(v.w);
    }
}

__global__ void words_to_state_kernel(float4* __restrict__ st, const uint32_t* __restrict__ w, long long n)
{
    const long long rec = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (rec >= 2 * n) return;
    const long long e = rec >> 1; const int p = (int)(rec & 1);
#pragma unroll
    for (int pl = 0; pl < kStatePlanes; ++pl) {
        st[(long long)pl * 2 * n + rec] =
            make_float4(__uint_as_float(w[(long long)word_of(p, pl, 0) * n + e]),
                        __uint_as_float(w[(long long)word_of(p, pl, 1) * n + e]),
                        __uint_as_float(w[(long long)word_of(p, pl, 2) * n + e]),
                        __uint_as_float(w[(long long)word_of(p, pl, 3) * n + e]));
    }
}

__global__ void diag_sincos_kernel(const float* __restrict__ x, float* __restrict__ sn, float* __restrict__ cs,
                                   long long n, int bounded)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    if (bounded) sincos_bounded(x[i], s, c); else sincos_any(x[i], s, c);
    sn[i] = s; cs[i] = c;
}

}  // namespace pnr

// =====================================================================================
// host side
// =====================================================================================
using namespace pnr;

struct pnr_env_s {
    pnr_config cfg;
    pnr_constants k;
    KParams base;        // constants pre-filled; pointers set per call
    DynParams dbase;     // dynamics-mode constants (zeroed in kinematic mode)
    long long n;
    unsigned long long env_off;
    int device;
    float4* state;
    float* dyn;          // dynamics-mode planar words [36][n] or null
    SceneBody* scene;    // dynamics-mode static scene bodies [kMaxScene] or null
    int diag;            // PNR_DIAG env var at create time (timing-only ablations; 0 in production)
    bool ready;          // a full reset has happened, or the state was set explicitly
    bool kin_set, dyn_set;  // pnr_set_state / pnr_set_dyn_state seen (both needed in dynamics mode)
    char err[512];
};

static thread_local char g_err[512] = "";

static int fail(pnr_handle h, int code, const char* fmt, ...)
{
    char* dst = h ? h->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    if (h) { strncpy(g_err, h->err, sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
    return code;
}

#define HIP_TRY(h, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(h, PNR_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));      \
    } while (0)

// RAII current-device switch: launches and allocations go to the handle's device.
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) { switched = hipSetDevice(dev) == hipSuccess; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

extern "C" {

int pnr_abi_version(void) { return PNR_ABI_VERSION; }

int pnr_config_default(pnr_config* c)
{
    if (!c) return fail(nullptr, PNR_ERR_INVALID, "pnr_config_default: null config");
    memset(c, 0, sizeof(*c));
    c->struct_size = (uint32_t)sizeof(pnr_config);
    c->abi_version = PNR_ABI_VERSION;
    // PioneerKinematicConfig, pioneer_knm_env.py:19-34
    c->max_v_to_r = 2; c->max_a_to_v = 10; c->done_distance = 0.1;
    c->award_max = 100.0; c->award_done = 5.0; c->award_potential_slope = 10.0;
    c->penalty_step = 1.0 / 100;
    c->target_lo[0] = 15; c->target_lo[1] = -10; c->target_lo[2] = 2;
    c->target_hi[0] = 25; c->target_hi[1] = 10; c->target_hi[2] = 6;
    c->target_radius = 0.2;
    // SimulationConfig, bullet_env.py:36-41
    c->timestep = 1.0 / 240; c->frame_skip = 10; c->gravity = 0;
    c->max_episode_steps = 500;  // pioneer_knm_train.py:27
    c->auto_reset = 1;
    c->obs_layout = PNR_ENV_MAJOR; c->action_layout = PNR_ENV_MAJOR;
    c->mode = PNR_MODE_KINEMATIC;
    // dynamics mode (no reference values)
    c->pd_kp = 4000.0; c->pd_kd = 400.0; c->torque_limit = 0.0;
    c->joint_damping = 0.0; c->joint_friction = 0.0;
    c->teleport = 0; c->randomize = 0;
    c->rand_mass_lo = 0.5; c->rand_mass_hi = 1.5;
    c->rand_friction_lo = 0.0; c->rand_friction_hi = 0.1;
    c->rand_damping_lo = 0.0; c->rand_damping_hi = 0.1;
    c->ground_z = NAN; c->contact_kp = 2000.0; c->contact_kd = 50.0;
    c->obstacle_position[0] = 10.0; c->obstacle_position[1] = 5.0; c->obstacle_position[2] = 0.0;  // pioneer_knm_env.py:253
    c->obstacle_half_extents[0] = c->obstacle_half_extents[1] = c->obstacle_half_extents[2] = 0.0; // disabled
    c->pointer_radius = 0.2;
    c->control_mode = PNR_CONTROL_POSITION; c->max_velocity = 0.0; c->link_contacts = 0;
    c->n_scene = 0;                                         // scene[] zeroed by the memset above
    c->pd_inertia_scaled = 0;
    return PNR_OK;
}

static int check_config(const pnr_config* c)
{
    if (!c) return fail(nullptr, PNR_ERR_INVALID, "null config");
    if (c->struct_size != sizeof(pnr_config) || c->abi_version != PNR_ABI_VERSION)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_config size/version mismatch (got %u/%u, want %zu/%d)",
                    c->struct_size, c->abi_version, sizeof(pnr_config), PNR_ABI_VERSION);
    if (!(c->timestep > 0) || c->frame_skip < 1)
        return fail(nullptr, PNR_ERR_INVALID, "timestep must be > 0 and frame_skip >= 1");
    if (c->obs_layout != PNR_ENV_MAJOR && c->obs_layout != PNR_FEATURE_MAJOR)
        return fail(nullptr, PNR_ERR_INVALID, "bad obs_layout %d", c->obs_layout);
    if (c->action_layout != PNR_ENV_MAJOR && c->action_layout != PNR_FEATURE_MAJOR)
        return fail(nullptr, PNR_ERR_INVALID, "bad action_layout %d", c->action_layout);
    if (c->mode != PNR_MODE_KINEMATIC && c->mode != PNR_MODE_DYNAMIC)
        return fail(nullptr, PNR_ERR_INVALID, "bad mode %d", c->mode);
    if (c->max_episode_steps < 0) return fail(nullptr, PNR_ERR_INVALID, "max_episode_steps < 0");
    if (c->control_mode != PNR_CONTROL_POSITION && c->control_mode != PNR_CONTROL_VELOCITY)
        return fail(nullptr, PNR_ERR_INVALID, "bad control_mode %d", c->control_mode);
    if (c->max_velocity > 0 && !(c->pd_kd > 0))
        return fail(nullptr, PNR_ERR_INVALID, "max_velocity needs pd_kd > 0 (the cap acts on the velocity the motor asks for)");
    for (int k = 0; k < 3; ++k)
        if (!(c->target_hi[k] >= c->target_lo[k]))
            return fail(nullptr, PNR_ERR_INVALID, "target_hi[%d] < target_lo[%d]", k, k);
    if (c->n_scene < 0 || c->n_scene > PNR_MAX_SCENE)
        return fail(nullptr, PNR_ERR_INVALID, "n_scene %d outside 0..%d", c->n_scene, PNR_MAX_SCENE);
    if (c->n_scene > 0 && c->mode != PNR_MODE_DYNAMIC)
        return fail(nullptr, PNR_ERR_INVALID, "scene bodies collide in dynamics mode only (the kinematic arm passes through everything, as the reference's does)");
    for (int b = 0; b < c->n_scene; ++b) {
        const pnr_scene_body& S = c->scene[b];
        const double qn = S.orientation[0] * S.orientation[0] + S.orientation[1] * S.orientation[1] +
                          S.orientation[2] * S.orientation[2] + S.orientation[3] * S.orientation[3];
        if (!(qn > 0)) return fail(nullptr, PNR_ERR_INVALID, "scene body %d: zero orientation quaternion", b);
        if (S.shape == PNR_SHAPE_PLANE) {
            if (!(S.size[0] * S.size[0] + S.size[1] * S.size[1] + S.size[2] * S.size[2] > 0))
                return fail(nullptr, PNR_ERR_INVALID, "scene body %d: zero plane normal", b);
        } else if (S.shape == PNR_SHAPE_BOX) {
            if (!(S.size[0] > 0 && S.size[1] > 0 && S.size[2] > 0))
                return fail(nullptr, PNR_ERR_INVALID, "scene body %d: box half extents must be > 0", b);
        } else if (S.shape == PNR_SHAPE_SPHERE) {
            if (!(S.size[0] > 0)) return fail(nullptr, PNR_ERR_INVALID, "scene body %d: sphere radius must be > 0", b);
        } else return fail(nullptr, PNR_ERR_INVALID, "scene body %d: bad shape %d", b, S.shape);
    }
    return PNR_OK;
}

int pnr_get_constants(const pnr_config* c, pnr_constants* out)
{
    int rc = check_config(c);
    if (rc) return rc;
    if (!out) return fail(nullptr, PNR_ERR_INVALID, "pnr_get_constants: null out");
    for (int i = 0; i < kDof; ++i) {
        // joint_limits(): float32 of the URDF limits, pioneer_knm_env.py:217-220
        out->r_lo[i] = (float)(-kJoints[i].limit);
        out->r_hi[i] = (float)(kJoints[i].limit);
        // :57-58  python scalar * float32 array stays float32
        const float span = out->r_hi[i] - out->r_lo[i];
        out->v_max[i] = (float)c->max_v_to_r * span;
        out->a_max[i] = (float)c->max_a_to_v * out->v_max[i];
    }
    out->dt = c->timestep * (double)c->frame_skip;  // bullet_scene.py:277-279
    out->eps = 1e-5;                                // pioneer_knm_env.py:61
    return PNR_OK;
}

static void fill_base(pnr_handle h)
{
    KParams& P = h->base;
    memset(&P, 0, sizeof(P));
    const pnr_config& c = h->cfg;
    P.state = h->state;
    P.n = h->n;
    P.env_off = h->env_off;
    P.T = 1;
    P.max_steps = c.max_episode_steps;
    P.auto_reset = c.auto_reset;
    P.dt = h->k.dt; P.eps = h->k.eps;
    for (int k = 0; k < 3; ++k) { P.tlo[k] = c.target_lo[k]; P.tspan[k] = c.target_hi[k] - c.target_lo[k]; }
    for (int i = 0; i < kDof; ++i) P.v_max[i] = h->k.v_max[i];
    DynParams& D = h->dbase;
    memset(&D, 0, sizeof(D));
    D.dyn = h->dyn;
    D.kp = (float)c.pd_kp; D.kd = (float)c.pd_kd; D.tau_max = (float)c.torque_limit;
    D.gravity = (float)c.gravity; D.dt_sub = (float)c.timestep; D.nsub = c.frame_skip;
    D.teleport = c.teleport; D.randomize = c.randomize;
    D.has_ground = (c.ground_z == c.ground_z) ? 1 : 0;
    D.ground_z = D.has_ground ? (float)c.ground_z : 0.f;
    D.ckp = (float)c.contact_kp; D.ckd = (float)c.contact_kd;
    D.has_box = (c.obstacle_half_extents[0] > 0 && c.obstacle_half_extents[1] > 0 && c.obstacle_half_extents[2] > 0) ? 1 : 0;
    for (int k = 0; k < 3; ++k) { D.box_c[k] = (float)c.obstacle_position[k]; D.box_h[k] = (float)c.obstacle_half_extents[k]; }
    D.ptr_radius = (float)c.pointer_radius;
    {   // the one motor law of pnr_dyn.h (oracle: orc_dyn_motor_torque)
        const bool capped = c.max_velocity > 0;
        const bool velocity = c.control_mode == PNR_CONTROL_VELOCITY;
        D.kp_eff = (velocity || capped) ? 0.f : (float)c.pd_kp;
        D.c_pos = (!velocity && capped) ? (float)(c.pd_kp / c.pd_kd) : 0.f;
        D.v_cap = capped ? (float)c.max_velocity : INFINITY;
    }
    D.link_contacts = c.link_contacts ? 1 : 0;
    D.inertia_scaled = c.pd_inertia_scaled ? 1 : 0;
    D.n_scene = h->scene ? c.n_scene : 0;
    D.scene = h->scene;
    SceneBody host_scene[kMaxScene];
    for (int b = 0; b < kMaxScene; ++b) {
        SceneBody& S = host_scene[b];
        S = SceneBody{};
        if (b >= c.n_scene) continue;
        const pnr_scene_body& B = c.scene[b];
        const double qn = std::sqrt(B.orientation[0] * B.orientation[0] + B.orientation[1] * B.orientation[1] +
                                    B.orientation[2] * B.orientation[2] + B.orientation[3] * B.orientation[3]);
        const double x = B.orientation[0] / qn, y = B.orientation[1] / qn, z = B.orientation[2] / qn, w = B.orientation[3] / qn;
        const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
        S.shape = B.shape;
        for (int k = 0; k < 3; ++k) { S.pos[k] = (float)B.position[k]; S.size[k] = (float)B.size[k]; }
        if (B.shape == PNR_SHAPE_PLANE) {               // unit world normal = R n / |n|
            const double nl = std::sqrt(B.size[0] * B.size[0] + B.size[1] * B.size[1] + B.size[2] * B.size[2]);
            for (int k = 0; k < 3; ++k) S.rot[k] = (float)((R[3 * k] * B.size[0] + R[3 * k + 1] * B.size[1] + R[3 * k + 2] * B.size[2]) / nl);
        } else {
            for (int k = 0; k < 9; ++k) S.rot[k] = (float)R[k];
        }
    }
    if (h->scene) (void)hipMemcpy(h->scene, host_scene, sizeof(host_scene), hipMemcpyHostToDevice);
    D.joint_damping = (float)c.joint_damping; D.joint_friction = (float)c.joint_friction;
    D.mass_lo = c.rand_mass_lo; D.mass_span = c.rand_mass_hi - c.rand_mass_lo;
    D.fric_lo = c.rand_friction_lo; D.fric_span = c.rand_friction_hi - c.rand_friction_lo;
    D.damp_lo = c.rand_damping_lo; D.damp_span = c.rand_damping_hi - c.rand_damping_lo;
    P.pot_m = (float)(c.award_max - c.award_done);
    P.pot_s = (float)c.award_potential_slope;
    P.penalty = (float)c.penalty_step;
    P.award_done = (float)c.award_done;
    P.done_dist = (float)c.done_distance;
}

int pnr_create(const pnr_config* cfg, int64_t num_envs, int64_t env_id_offset, int device_id,
               uint64_t seed, pnr_handle* out)
{
    if (!out) return fail(nullptr, PNR_ERR_INVALID, "pnr_create: null out");
    *out = nullptr;
    int rc = check_config(cfg);
    if (rc) return rc;
    if (num_envs < 1) return fail(nullptr, PNR_ERR_INVALID, "num_envs must be >= 1 (got %lld)", (long long)num_envs);
    if (env_id_offset < 0) return fail(nullptr, PNR_ERR_INVALID, "env_id_offset < 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, PNR_ERR_NODEVICE, "no HIP device visible (this engine has no CPU backend)");
    if (device_id < 0 || device_id >= ndev)
        return fail(nullptr, PNR_ERR_NODEVICE, "device_id %d out of range [0,%d)", device_id, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PNR_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only",
                    device_id, prop.gcnArchName);

    for (int i = 0; i < kDof; ++i) {
        // the obs constants of pnr_model.h must be np.cos/np.sin of the float32 limits
        const float c = (float)std::cos((double)limit_hi(i)), s = (float)std::sin((double)limit_hi(i));
        if (c != kLimitCos[i] || s != kLimitSin[i])
            return fail(nullptr, PNR_ERR_INVALID, "model table: cos/sin constants of joint %d are stale", i);
    }

    pnr_handle h = new (std::nothrow) pnr_env_s();
    if (!h) return fail(nullptr, PNR_ERR_NOMEM, "host allocation failed");
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->n = num_envs;
    h->env_off = (unsigned long long)env_id_offset;
    h->device = device_id;
    pnr_get_constants(cfg, &h->k);

    DeviceGuard g(device_id);
    hipError_t e = hipMalloc((void**)&h->state, sizeof(float4) * kStatePlanes * 2 * (size_t)num_envs);
    if (e != hipSuccess) { delete h; return fail(nullptr, PNR_ERR_NOMEM, "hipMalloc(state) failed: %s", hipGetErrorString(e)); }
    (void)hipMemset(h->state, 0, sizeof(float4) * kStatePlanes * 2 * (size_t)num_envs);
    if (cfg->mode == PNR_MODE_DYNAMIC) {
        const size_t bytes = sizeof(float) * PNR_DYN_STATE_WORDS * (size_t)num_envs;
        e = hipMalloc((void**)&h->dyn, bytes);
        if (e != hipSuccess) { (void)hipFree(h->state); delete h; return fail(nullptr, PNR_ERR_NOMEM, "hipMalloc(dyn) failed: %s", hipGetErrorString(e)); }
        (void)hipMemset(h->dyn, 0, bytes);
        if (cfg->n_scene > 0) {
            e = hipMalloc((void**)&h->scene, sizeof(SceneBody) * kMaxScene);
            if (e != hipSuccess) { (void)hipFree(h->dyn); (void)hipFree(h->state); delete h; return fail(nullptr, PNR_ERR_NOMEM, "hipMalloc(scene) failed: %s", hipGetErrorString(e)); }
        }
    }
    { const char* e_ = getenv("PNR_DIAG"); h->diag = e_ ? atoi(e_) : 0; }
    fill_base(h);
    h->base.seed_lo = (unsigned)seed; h->base.seed_hi = (unsigned)(seed >> 32);
    *out = h;
    return PNR_OK;
}

int pnr_destroy(pnr_handle h)
{
    if (!h) return PNR_OK;
    DeviceGuard g(h->device);
    if (h->state) (void)hipFree(h->state);
    if (h->dyn) (void)hipFree(h->dyn);
    if (h->scene) (void)hipFree(h->scene);
    delete h;
    return PNR_OK;
}

int pnr_seed(pnr_handle h, uint64_t seed)
{
    if (!h) return fail(nullptr, PNR_ERR_INVALID, "null handle");
    h->base.seed_lo = (unsigned)seed; h->base.seed_hi = (unsigned)(seed >> 32);
    return PNR_OK;
}

int64_t pnr_num_envs(pnr_handle h) { return h ? h->n : -1; }

const char* pnr_last_error(pnr_handle h) { return h ? h->err : g_err; }

static inline unsigned grid_for(long long n) { return (unsigned)((n + kEnvsPerWave - 1) / kEnvsPerWave); }

// step kernels are persistent over tiles: at most 8 one-wave workgroups per CU (256 CUs)
// Persistent grid of one-wave workgroups.  One launch per step: 2 048 (two waves per SIMD; with a single tile
// per wave the second wave is what overlaps one tile's stores with another's arithmetic).  Long rollouts
// (pnr_rollout, T >= 8): 1 024 = ONE wave per SIMD — each wave's stores of step t drain under its own step
// t + 1, and a second wave on the SIMD only contends for issue slots (6.5-7.2 vs 7.2-7.5 us per 65 536-env
// step at T = 32; 768 and 1 536 are worse than either; at T = 2, 4 the 2 048 grid still wins: 8.7 / 8.1 vs
// 10.0 / 8.6 us).  PNR_GRID_CAP / PNR_GRID_CAP_ROLLOUT override for experiments.
static inline unsigned step_grid_for(long long n, int T)
{
    static int cap = -1, cap_roll = -1;
    if (cap < 0) { const char* e_ = getenv("PNR_GRID_CAP"); cap = e_ ? atoi(e_) : 2048; if (cap < 1) cap = 2048; }
    if (cap_roll < 0) { const char* e_ = getenv("PNR_GRID_CAP_ROLLOUT"); cap_roll = e_ ? atoi(e_) : 1024; if (cap_roll < 1) cap_roll = 1024; }
    const unsigned tiles = grid_for(n);
    const unsigned c = (unsigned)(T >= 8 ? cap_roll : cap);
    return tiles < c ? tiles : c;
}

int pnr_reset(pnr_handle h, const uint8_t* mask, const float* joint_pos, const float* target_pos,
              float* obs_out, void* stream)
{
    if (!h) return fail(nullptr, PNR_ERR_INVALID, "null handle");
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    KParams P = h->base;
    P.mask = mask; P.joint_pos = joint_pos; P.target_pos = target_pos; P.obs = obs_out;
    const dim3 grid(grid_for(h->n)), block(kWave);
    const DynParams& D = h->dbase;
    const bool fm = h->cfg.obs_layout == PNR_FEATURE_MAJOR;
#define PNR_LAUNCH_RESET(OBSK)                                                                     \
    do {                                                                                           \
        if (h->dyn) hipLaunchKernelGGL((reset_kernel<0, OBSK, true>), grid, block, 0, st, P, D);   \
        else hipLaunchKernelGGL((reset_kernel<0, OBSK, false>), grid, block, 0, st, P, D);         \
    } while (0)
    if (mask && !h->ready)
        return fail(h, PNR_ERR_INVALID, "the first pnr_reset must be a full one (mask == NULL)");
    if (!mask) h->ready = true;
    if (!obs_out) PNR_LAUNCH_RESET(0);
    else if (fm && mask) PNR_LAUNCH_RESET(1);
    else if (fm) PNR_LAUNCH_RESET(4);
    else if (!mask) PNR_LAUNCH_RESET(2);
    else PNR_LAUNCH_RESET(3);
#undef PNR_LAUNCH_RESET
    HIP_TRY(h, hipGetLastError());
    return PNR_OK;
}

int pnr_observe(pnr_handle h, float* obs_out, void* stream)
{
    if (!h) return fail(nullptr, PNR_ERR_INVALID, "null handle");
    if (!obs_out) return fail(h, PNR_ERR_INVALID, "pnr_observe: null obs_out");
    if (!h->ready) return fail(h, PNR_ERR_INVALID, "pnr_observe before the first pnr_reset");
    DeviceGuard g(h->device);
    KParams P = h->base;
    P.obs = obs_out;
    const dim3 grid(grid_for(h->n)), block(kWave);
    const DynParams& D = h->dbase;
    hipStream_t st = (hipStream_t)stream;
    const bool fm = h->cfg.obs_layout == PNR_FEATURE_MAJOR;
    if (fm && h->dyn) hipLaunchKernelGGL((reset_kernel<1, 4, true>), grid, block, 0, st, P, D);
    else if (fm) hipLaunchKernelGGL((reset_kernel<1, 4, false>), grid, block, 0, st, P, D);
    else if (h->dyn) hipLaunchKernelGGL((reset_kernel<1, 2, true>), grid, block, 0, st, P, D);
    else hipLaunchKernelGGL((reset_kernel<1, 2, false>), grid, block, 0, st, P, D);
    HIP_TRY(h, hipGetLastError());
    return PNR_OK;
}

static int launch_step(pnr_handle h, int T, const float* actions, float* obs, float* reward,
                       uint8_t* done, uint8_t* truncated, float* info, void* stream)
{
    if (!h) return fail(nullptr, PNR_ERR_INVALID, "null handle");
    if (!actions || !obs || !reward || !done)
        return fail(h, PNR_ERR_INVALID, "actions, obs, reward and done must be non-null");
    if (T < 1) return fail(h, PNR_ERR_INVALID, "T must be >= 1 (got %d)", T);
    if (!h->ready) return fail(h, PNR_ERR_INVALID, "pnr_step before the first pnr_reset (or pnr_set_state)");
    if (info && (reinterpret_cast<uintptr_t>(info) & 15u))
        return fail(h, PNR_ERR_INVALID, "info must be 16-byte aligned");
    if (h->cfg.action_layout == PNR_ENV_MAJOR && (reinterpret_cast<uintptr_t>(actions) & 7u))
        return fail(h, PNR_ERR_INVALID, "env-major actions must be 8-byte aligned");
    DeviceGuard g(h->device);
    KParams P = h->base;
    P.diag = h->diag;
    P.T = T; P.actions = actions; P.obs = obs; P.reward = reward; P.done = done; P.trunc = truncated; P.info = info;
    const dim3 grid(step_grid_for(h->n, T)), block(kWave);
    hipStream_t st = (hipStream_t)stream;
    const DynParams& D = h->dbase;
    const bool oem = h->cfg.obs_layout == PNR_ENV_MAJOR, aem = h->cfg.action_layout == PNR_ENV_MAJOR;
    if (h->cfg.mode == PNR_MODE_DYNAMIC) {
        // one launch: dyn_step_kernel runs the sub-steps one env per lane, then finishes each step (reward /
        // TimeLimit / auto-reset / obs) as lane pairs, T times
        const dim3 gridD((unsigned)((h->n + kDynEnvsPerWg - 1) / kDynEnvsPerWg));
        // link scales differ from 1 only under randomisation (or after pnr_set_dyn_state, which marks it)
        const bool rnd = h->cfg.randomize || h->dyn_set;
        {
            const KParams& Pt = P;
            // T > 1: dyn_rollout_kernel loops over the steps inside the launch (state in registers, stores of step t
            // under the sub-steps of t + 1); T == 1: the lean single-step kernel
#define PNR_DYN_LAUNCH2(O, A, R, C) do { \
        if (T > 1) hipLaunchKernelGGL((dyn_rollout_kernel<O, A, R, C>), gridD, block, 0, st, Pt.state, D.dyn, Pt.actions, Pt.n, Pt.dt, \
                                      Pt.eps, (float)h->cfg.max_v_to_r, Pt, D); \
        else hipLaunchKernelGGL((dyn_step_kernel<O, A, R, C>), gridD, block, 0, st, Pt.state, D.dyn, Pt.actions, Pt.n, Pt.dt, \
                                Pt.eps, (float)h->cfg.max_v_to_r, Pt, D); } while (0)
            // contact-free handles run instantiations without any contact code
            // (PHYS: bit 0 contacts, bit 1 the inertia-scaled motor)
#define PNR_DYN_LAUNCH(O, A, R) do { const bool ct_ = D.has_ground || D.has_box || D.n_scene > 0; \
        if (D.inertia_scaled) { if (ct_) PNR_DYN_LAUNCH2(O, A, R, 3); else PNR_DYN_LAUNCH2(O, A, R, 2); } \
        else { if (ct_) PNR_DYN_LAUNCH2(O, A, R, 1); else PNR_DYN_LAUNCH2(O, A, R, 0); } } while (0)
            if (oem) {
                if (aem) { if (rnd) PNR_DYN_LAUNCH(true, true, true); else PNR_DYN_LAUNCH(true, true, false); }
                else { if (rnd) PNR_DYN_LAUNCH(true, false, true); else PNR_DYN_LAUNCH(true, false, false); }
            } else {
                if (aem) { if (rnd) PNR_DYN_LAUNCH(false, true, true); else PNR_DYN_LAUNCH(false, true, false); }
                else { if (rnd) PNR_DYN_LAUNCH(false, false, true); else PNR_DYN_LAUNCH(false, false, false); }
            }
#undef PNR_DYN_LAUNCH2
#undef PNR_DYN_LAUNCH
        }
        HIP_TRY(h, hipGetLastError());
        return PNR_OK;
    }
    if (oem && aem) hipLaunchKernelGGL((step_kernel<true, true>), grid, block, 0, st, P.state, P.actions, P.n, P.dt, P.eps, (float)h->cfg.max_v_to_r, P);
    else if (oem && !aem) hipLaunchKernelGGL((step_kernel<true, false>), grid, block, 0, st, P.state, P.actions, P.n, P.dt, P.eps, (float)h->cfg.max_v_to_r, P);
    else if (!oem && aem) hipLaunchKernelGGL((step_kernel<false, true>), grid, block, 0, st, P.state, P.actions, P.n, P.dt, P.eps, (float)h->cfg.max_v_to_r, P);
    else hipLaunchKernelGGL((step_kernel<false, false>), grid, block, 0, st, P.state, P.actions, P.n, P.dt, P.eps, (float)h->cfg.max_v_to_r, P);
    HIP_TRY(h, hipGetLastError());
    return PNR_OK;
}

int pnr_step(pnr_handle h, const float* actions, float* obs, float* reward, uint8_t* done,
             uint8_t* truncated, float* info, void* stream)
{
    return launch_step(h, 1, actions, obs, reward, done, truncated, info, stream);
}

int pnr_rollout(pnr_handle h, int32_t T, const float* actions, float* obs, float* reward,
                uint8_t* done, uint8_t* truncated, void* stream)
{
    return launch_step(h, T, actions, obs, reward, done, truncated, nullptr, stream);
}

int pnr_diag_sincos(const float* x, float* sin_out, float* cos_out, int64_t n, int bounded, void* stream)
{
    if (!x || !sin_out || !cos_out || n < 0) return fail(nullptr, PNR_ERR_INVALID, "pnr_diag_sincos: bad argument");
    if (n == 0) return PNR_OK;
    hipLaunchKernelGGL(diag_sincos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, sin_out, cos_out, (long long)n, bounded);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_ppo_loss(int64_t batch, const int64_t* idx, const float* head_policy, const float* head_value, const float* actions,
                 const float* logp_old, const float* mean_old, const float* log_std_old, const float* adv,
                 const float* value_target, const float* value_old, const float* kl_coeff, const float* entropy_coeff,
                 float clip_param, float vf_clip_param, float vf_loss_coeff, float* grad_head_policy,
                 float* grad_head_value, float* partial_sums, int64_t partial_rows, float* means, void* stream)
{
    if (batch <= 0 || !head_policy || !head_value || !actions || !logp_old || !mean_old || !log_std_old || !adv ||
        !value_target || !value_old || !kl_coeff || !entropy_coeff || !grad_head_policy || !grad_head_value || !partial_sums)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_ppo_loss: null argument or empty batch");
    const long long blocks = (batch + kPpoBlock - 1) / kPpoBlock;
    if (partial_rows < blocks)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_ppo_loss: partial_sums has %lld rows, the launch needs %lld",
                    (long long)partial_rows, blocks);
    PpoLossParams P;
    P.head_p = head_policy; P.head_v = head_value; P.actions = actions; P.logp_old = logp_old; P.mean_old = mean_old;
    P.idx = reinterpret_cast<const long long*>(idx);
    P.ls_old = log_std_old; P.adv = adv; P.vtarg = value_target; P.v_old = value_old; P.kl_coeff = kl_coeff;
    P.ent_coeff = entropy_coeff; P.g_head_p = grad_head_policy; P.g_head_v = grad_head_value; P.partials = partial_sums;
    P.B = batch; P.clip = clip_param; P.vf_clip = vf_clip_param; P.vf_coeff = vf_loss_coeff;
    hipLaunchKernelGGL(ppo_loss_kernel, dim3((unsigned)blocks), dim3(kPpoBlock), 0, (hipStream_t)stream, P);
    if (means)
        hipLaunchKernelGGL(ppo_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial_sums, (long long)blocks,
                           (long long)batch, means, (float*)nullptr);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

// ---- the host driver's MLPs (pnr_mlp.h) ------------------------------------------------------------------------
// 32 slices x 4 roles x 2 nets = 256 workgroups = one per CU, one round; 64 slices (two rounds) wrote and re-read twice the
// slab bytes for the same time in the multiply loop
constexpr long long kMaxSlices = 32;
static inline void mlp_slicing(long long B, long long* slices, long long* slice_rows)
{
    long long want = (B + kWgChunk - 1) / kWgChunk;          // at most one slice per 64-sample chunk ...
    if (want > kMaxSlices) want = kMaxSlices;                // ... and few enough that slices x 4 roles x 2 nets fill the CUs ONCE
    if (want < 1) want = 1;
    long long rows = (B + want - 1) / want;
    rows = (rows + kWgChunk - 1) / kWgChunk * kWgChunk;
    *slice_rows = rows;
    *slices = (B + rows - 1) / rows;
}

int64_t pnr_mlp_slab_floats(int64_t batch)
{
    if (batch < 1) return 0;
    long long slices, rows;
    mlp_slicing(batch, &slices, &rows);
    return (int64_t)(slices * kMlpNets * kGradElems);
}

int64_t pnr_mlp_pack_elems(void) { return (int64_t)kMlpNets * kPackElems; }
int64_t pnr_mlp_bias_elems(void) { return (int64_t)kMlpNets * kBiasElems; }

int pnr_mlp_pack(const float* const* params, int32_t n3_policy, int32_t n3_value, void* wpack, float* bias, void* stream)
{
    if (!params || !wpack || !bias) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_pack: null argument");
    if (n3_policy < 1 || n3_policy > kMlpHead || n3_value < 1 || n3_value > kMlpHead)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_pack: head widths must be in 1..16");
    MlpPackParams P;
    for (int n = 0; n < kMlpNets; ++n) {
        for (int k = 0; k < 6; ++k)
            if (!params[6 * n + k]) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_pack: null parameter %d of net %d", k, n);
        P.net[n] = {params[6 * n + 0], params[6 * n + 1], params[6 * n + 2], params[6 * n + 3], params[6 * n + 4], params[6 * n + 5],
                    n == 0 ? n3_policy : n3_value};
    }
    P.wpack = static_cast<__bf16*>(wpack); P.bias = bias;
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((kPackElems + kBiasElems + 255) / 256, kMlpNets), dim3(256), 0, (hipStream_t)stream, P);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_mlp_forward(int64_t batch, const float* obs, const int64_t* idx, const float* f_loc, const float* f_inv,
                    const float* f_lo, const float* f_hi, const void* wpack, const float* bias, float* head,
                    void* xs, void* h1, void* h2, int32_t first_net, int32_t n_nets, void* stream)
{
    if (batch < 1 || !obs || !wpack || !bias || !head) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_forward: null argument or empty batch");
    if (first_net < 0 || n_nets < 1 || first_net + n_nets > kMlpNets) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_forward: bad net range");
    if ((f_loc || f_inv || f_lo || f_hi) && !(f_loc && f_inv && f_lo && f_hi))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_forward: the four filter vectors come together or not at all");
    if ((h1 == nullptr) != (h2 == nullptr)) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_forward: h1 and h2 come together");
    MlpFwdParams P = {};
    P.obs = obs; P.idx = reinterpret_cast<const long long*>(idx); P.f_loc = f_loc; P.f_inv = f_inv; P.f_lo = f_lo; P.f_hi = f_hi;
    P.wpack = static_cast<const __bf16*>(wpack); P.bias = bias; P.head = head;
    P.xs = static_cast<__bf16*>(xs); P.h1 = static_cast<__bf16*>(h1); P.h2 = static_cast<__bf16*>(h2);
    P.B = batch; P.first_net = first_net; P.n_nets = n_nets;
    hipLaunchKernelGGL(mlp_forward_kernel<false>, dim3((unsigned)((batch + kMlpBM - 1) / kMlpBM), n_nets), dim3(kMlpThreads), 0,
                       (hipStream_t)stream, P);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int64_t pnr_ppo_gae_scratch(int64_t n) { return n < 1 ? 0 : ((n + 63) / 64) * 8; }

int pnr_ppo_gae(int32_t T, int64_t n, const float* reward, const float* values, const float* last_value, const uint8_t* done,
                const uint8_t* truncated, const float* actions, const float* mean, const float* log_std, double gamma,
                double lambda, float* logp, float* adv, float* value_target, float* terminals, const pnr_rollout_stats* stats,
                void* stream)
{
    if (T < 1 || n < 1 || !reward || !values || !last_value || !done || !adv || !value_target)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_ppo_gae: null argument or empty rollout");
    if (actions && (!mean || !log_std || !logp)) return fail(nullptr, PNR_ERR_INVALID, "pnr_ppo_gae: actions need mean, log_std and logp");
    const long long blocks = (n + 63) / 64;
    if (stats) {
        if (!stats->ep_ret || !stats->ep_len || !stats->scratch || !stats->w_sum || !stats->w_len || !stats->w_cnt || !stats->w_max ||
            !stats->w_min || !stats->adv_stats)
            return fail(nullptr, PNR_ERR_INVALID, "pnr_ppo_gae: null pointer in pnr_rollout_stats");
        if (stats->scratch_doubles < blocks * 8)
            return fail(nullptr, PNR_ERR_INVALID, "pnr_ppo_gae: scratch holds %lld doubles, the launch needs %lld",
                        (long long)stats->scratch_doubles, blocks * 8);
    }
    GaeParams P;
    P.reward = reward; P.values = values; P.last_value = last_value; P.done = done; P.trunc = truncated;
    P.actions = actions; P.mean = mean; P.log_std = log_std; P.logp = logp; P.adv = adv; P.vtarg = value_target;
    P.terminals = terminals; P.N = n; P.T = T;
    P.ep_ret = stats ? stats->ep_ret : nullptr; P.ep_len = stats ? stats->ep_len : nullptr; P.partials = stats ? stats->scratch : nullptr;
    P.gamma = (float)gamma; P.gamma_lam = (float)(gamma * lambda);      // the host formula's Python-float product, then float32
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gae_logp_kernel, dim3((unsigned)blocks, (unsigned)(1 + ((actions || terminals) ? T : 0))), dim3(64), 0, st, P);
    if (stats)
        hipLaunchKernelGGL(gae_finish_kernel, dim3(1), dim3(64), 0, st, stats->scratch, blocks, (long long)T * n, stats->w_sum, stats->w_len,
                           stats->w_cnt, stats->w_max, stats->w_min, stats->adv_stats);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int64_t pnr_filter_moments_scratch(int64_t rows) { return rows < 1 ? 0 : ((rows + kFmRows - 1) / kFmRows) * 2 * kFmCols; }

int pnr_filter_moments(int64_t rows, const float* obs, const float* pivot, float* scratch, int64_t scratch_floats, double* dsum,
                       double* dsq, double* dn, void* stream)
{
    if (rows < 1 || !obs || !pivot || !scratch || !dsum || !dsq || !dn)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_filter_moments: null argument or no rows");
    const long long blocks = (rows + kFmRows - 1) / kFmRows;
    if (scratch_floats < blocks * 2 * kFmCols)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_filter_moments: scratch holds %lld floats, the launch needs %lld",
                    (long long)scratch_floats, blocks * 2 * kFmCols);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(filter_moments_kernel, dim3((unsigned)blocks), dim3(kFmThreads), 0, st, obs, (long long)rows, pivot, scratch);
    hipLaunchKernelGGL(filter_moments_finish_kernel, dim3(kFmCols, 2), dim3(256), 0, st, scratch, blocks, (long long)rows, dsum, dsq, dn);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_filter_merge(double* dn, double* dsum, double* dsq, const float* pivot, double* n, double* mean, double* m2, void* stream)
{
    if (!dn || !dsum || !dsq || !pivot || !n || !mean || !m2) return fail(nullptr, PNR_ERR_INVALID, "pnr_filter_merge: null argument");
    hipLaunchKernelGGL(filter_merge_kernel, dim3(1), dim3(kFmThreads), 0, (hipStream_t)stream, dn, dsum, dsq, pivot, n, mean, m2);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_permutation(int64_t n, uint64_t seed, uint64_t stream_id, int64_t* out, void* stream)
{
    if (n < 1 || !out) return fail(nullptr, PNR_ERR_INVALID, "pnr_permutation: null argument or n < 1");
    if (n > (1ll << 40)) return fail(nullptr, PNR_ERR_INVALID, "pnr_permutation: n above 2^40");
    int bits = 1;
    while ((1ll << bits) < n) ++bits;                   // 2^bits >= n
    const int half = (bits + 1) / 2 < 1 ? 1 : (bits + 1) / 2;
    hipLaunchKernelGGL(permutation_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<long long*>(out), (long long)n, half, (unsigned long long)seed, (unsigned long long)stream_id);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_mlp_act(int64_t batch, const float* obs, const float* f_loc, const float* f_inv, const float* f_lo, const float* f_hi,
                const void* wpack, const float* bias, const float* noise, const float* a_max, float* head, float* mean,
                float* log_std, float* values, float* actions, float* env_actions, void* xs_out, void* stream)
{
    if (batch < 1 || !obs || !wpack || !bias || !noise || !mean || !log_std || !values || !actions)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_act: null argument or empty batch");
    if ((f_loc || f_inv || f_lo || f_hi) && !(f_loc && f_inv && f_lo && f_hi))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_act: the four filter vectors come together or not at all");
    if (a_max && (!env_actions || env_actions == actions))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_act: clipping (a_max) needs its own env_actions buffer");
    MlpFwdParams P = {};
    P.obs = obs; P.f_loc = f_loc; P.f_inv = f_inv; P.f_lo = f_lo; P.f_hi = f_hi;
    P.wpack = static_cast<const __bf16*>(wpack); P.bias = bias; P.head = head;
    P.B = batch; P.first_net = 0; P.n_nets = kMlpNets;
    P.noise = noise; P.a_max = a_max; P.mean = mean; P.log_std = log_std; P.values = values; P.actions = actions;
    P.env_actions = a_max ? env_actions : actions;
    P.xs = static_cast<__bf16*>(xs_out);                     // the nets' input as they saw it, for the learner (or null)
    hipLaunchKernelGGL(mlp_forward_kernel<false>, dim3((unsigned)((batch + kMlpBM - 1) / kMlpBM), kMlpNets), dim3(kMlpThreads), 0,
                       (hipStream_t)stream, P);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_mlp_backward(int64_t batch, const float* g_head, const void* wpack, const void* xs, const void* h1, const void* h2,
                     void* dz1, void* dz2, float* slabs, int64_t slab_floats, float* const* grads, int32_t n3_policy,
                     int32_t n3_value, int32_t accumulate, const float* scale, void* stream)
{
    if (batch < 1 || !g_head || !wpack || !xs || !h1 || !h2 || !dz1 || !dz2 || !slabs || !grads)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_backward: null argument or empty batch");
    long long slices, rows;
    mlp_slicing(batch, &slices, &rows);
    if (slab_floats < slices * kMlpNets * kGradElems)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_backward: slabs hold %lld floats, the launch needs %lld",
                    (long long)slab_floats, slices * kMlpNets * kGradElems);
    hipStream_t st = (hipStream_t)stream;
    MlpBwdParams Bp;
    Bp.g_head = g_head; Bp.wpack = static_cast<const __bf16*>(wpack); Bp.h1 = static_cast<const __bf16*>(h1);
    Bp.h2 = static_cast<const __bf16*>(h2); Bp.dz1 = static_cast<__bf16*>(dz1); Bp.dz2 = static_cast<__bf16*>(dz2); Bp.B = batch;
    hipLaunchKernelGGL(mlp_backward_data_kernel, dim3((unsigned)((batch + kMlpBM - 1) / kMlpBM), kMlpNets), dim3(kMlpThreads), 0, st, Bp);
    MlpWgradParams Wp;
    Wp.g_head = g_head; Wp.xs = static_cast<const __bf16*>(xs); Wp.h1 = Bp.h1; Wp.h2 = Bp.h2; Wp.dz1 = Bp.dz1; Wp.dz2 = Bp.dz2;
    Wp.slabs = slabs; Wp.B = batch; Wp.slice_rows = rows;
    hipLaunchKernelGGL(mlp_wgrad_kernel, dim3((unsigned)slices, kWgParts, kMlpNets), dim3(kMlpThreads), 0, st, Wp);
    MlpReduceParams Rp;
    Rp.slabs = slabs; Rp.slices = (int)slices; Rp.accumulate = accumulate; Rp.scale = scale;
    for (int n = 0; n < kMlpNets; ++n) {
        for (int k = 0; k < 6; ++k)
            if (!grads[6 * n + k]) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_backward: null gradient %d of net %d", k, n);
        Rp.gw1[n] = grads[6 * n + 0]; Rp.gb1[n] = grads[6 * n + 1]; Rp.gw2[n] = grads[6 * n + 2];
        Rp.gb2[n] = grads[6 * n + 3]; Rp.gw3[n] = grads[6 * n + 4]; Rp.gb3[n] = grads[6 * n + 5];
    }
    Rp.n3[0] = n3_policy; Rp.n3[1] = n3_value;
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3((kGradElems + 255) / 256, kMlpNets), dim3(256), 0, st, Rp);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int64_t pnr_mlp_grad_floats(void) { return (int64_t)kMlpNets * kGradElems; }

static int mlp_step_check(const pnr_mlp_step* s, const char* who)
{
    if (!s) return fail(nullptr, PNR_ERR_INVALID, "%s: null argument block", who);
    if (s->struct_size != sizeof(pnr_mlp_step))
        return fail(nullptr, PNR_ERR_INVALID, "%s: pnr_mlp_step size mismatch (got %u, want %zu)", who, s->struct_size, sizeof(pnr_mlp_step));
    for (int k = 0; k < 12; ++k)
        if (!s->params[k]) return fail(nullptr, PNR_ERR_INVALID, "%s: null parameter %d", who, k);
    if (!s->wpack || !s->bias || !s->adam_m || !s->adam_v || !s->adam_step)
        return fail(nullptr, PNR_ERR_INVALID, "%s: null weight / optimiser buffer", who);
    if (s->n3_policy < 1 || s->n3_policy > kMlpHead || s->n3_value < 1 || s->n3_value > kMlpHead)
        return fail(nullptr, PNR_ERR_INVALID, "%s: head widths must be in 1..16", who);
    return PNR_OK;
}

// loss_rows > 0: the launch also sums the update's loss means (rows of the fused kernel in s->partials)
static void mlp_launch_adam(const pnr_mlp_step* s, const float* grad, int slices, float scale, hipStream_t st, long long loss_rows = 0)
{
    MlpAdamParams A;
    A.partials = loss_rows > 0 ? s->partials : nullptr; A.loss_rows = loss_rows; A.batch = s->batch; A.means = s->means;
    A.kl_coeff = s->kl_coeff; A.ent_coeff = s->entropy_coeff; A.vf_coeff = s->vf_loss_coeff;
    A.grad = grad; A.slices = slices; A.grad_scale = scale;
    for (int n = 0; n < kMlpNets; ++n) {
        A.w1[n] = s->params[6 * n + 0]; A.b1[n] = s->params[6 * n + 1]; A.w2[n] = s->params[6 * n + 2];
        A.b2[n] = s->params[6 * n + 3]; A.w3[n] = s->params[6 * n + 4]; A.b3[n] = s->params[6 * n + 5];
    }
    A.n3[0] = s->n3_policy; A.n3[1] = s->n3_value;
    A.m = s->adam_m; A.v = s->adam_v; A.step = s->adam_step;
    A.lr = s->lr; A.beta1 = s->beta1; A.beta2 = s->beta2; A.eps = s->eps;
    A.wpack = static_cast<__bf16*>(s->wpack); A.bias = s->bias;
    hipLaunchKernelGGL(mlp_adam_kernel, dim3((kGradElems + 255) / 256 + 1, kMlpNets), dim3(256), 0, st, A);
}

int pnr_ppo_pack_record(int64_t rows, const float* actions, const float* logp_old, const float* mean_old, const float* log_std_old,
                        const float* adv, const float* value_target, const float* value_old, const float* adv_mu, const float* adv_den,
                        float* record_rows, void* stream)
{
    if (rows < 1 || !actions || !logp_old || !mean_old || !log_std_old || !adv || !value_target || !value_old || !record_rows)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_ppo_pack_record: null argument or no rows");
    if ((adv_mu == nullptr) != (adv_den == nullptr))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_ppo_pack_record: adv_mu and adv_den come together");
    RecordPackParams R;
    R.actions = actions; R.logp = logp_old; R.mean = mean_old; R.log_std = log_std_old; R.adv = adv; R.vtarg = value_target;
    R.values = value_old; R.adv_mu = adv_mu; R.adv_den = adv_den; R.aos = record_rows; R.rows = rows;
    hipLaunchKernelGGL(record_pack_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, R);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_mlp_gather(int64_t batch, const int64_t* idx, const float* obs, const float* f_loc, const float* f_inv, const float* f_lo,
                   const float* f_hi, const float* actions, const float* logp_old, const float* mean_old, const float* log_std_old,
                   const float* adv, const float* value_target, const float* value_old, void* xs_out, float* actions_out,
                   float* logp_out, float* mean_out, float* log_std_out, float* adv_out, float* value_target_out,
                   float* value_old_out, const float* record_rows, const void* xs_rows, void* stream)
{
    const bool soa = actions && logp_old && mean_old && log_std_old && adv && value_target && value_old;
    if (batch < 1 || !(obs || xs_rows) || !(soa || record_rows) || !xs_out ||
        !actions_out || !logp_out || !mean_out || !log_std_out || !adv_out || !value_target_out || !value_old_out)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_gather: null argument or empty batch");
    if ((f_loc || f_inv || f_lo || f_hi) && !(f_loc && f_inv && f_lo && f_hi))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_gather: the four filter vectors come together or not at all");
    MlpGatherParams G;
    G.obs = obs; G.idx = reinterpret_cast<const long long*>(idx); G.f_loc = f_loc; G.f_inv = f_inv; G.f_lo = f_lo; G.f_hi = f_hi;
    G.actions = actions; G.logp = logp_old; G.mean = mean_old; G.log_std = log_std_old; G.adv = adv; G.vtarg = value_target;
    G.values = value_old; G.rec_aos = record_rows; G.xs_src = static_cast<const __bf16*>(xs_rows); G.xs_out = static_cast<__bf16*>(xs_out); G.actions_out = actions_out; G.logp_out = logp_out;
    G.mean_out = mean_out; G.log_std_out = log_std_out; G.adv_out = adv_out; G.vtarg_out = value_target_out; G.values_out = value_old_out;
    G.B = batch;
    hipLaunchKernelGGL(mlp_gather_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(kMlpThreads), 0, (hipStream_t)stream, G);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_mlp_train_step(const pnr_mlp_step* s, void* stream)
{
    int rc = mlp_step_check(s, "pnr_mlp_train_step");
    if (rc) return rc;
    const long long B = s->batch;
    if (B < 1 || (!s->obs && !s->xs_in) || !s->actions || !s->logp_old || !s->mean_old || !s->log_std_old || !s->adv || !s->value_target ||
        !s->value_old || !s->kl_coeff || !s->entropy_coeff || !s->head || !s->g_head || !s->xs || !s->h1 || !s->h2 || !s->dz1 ||
        !s->dz2 || !s->partials || !s->slabs || !s->means)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: null argument or empty batch");
    if ((s->f_loc || s->f_inv || s->f_lo || s->f_hi) && !(s->f_loc && s->f_inv && s->f_lo && s->f_hi))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: the four filter vectors come together or not at all");
    long long slices, rows;
    mlp_slicing(B, &slices, &rows);
    if (s->slab_floats < slices * kMlpNets * kGradElems)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: slabs hold %lld floats, the launch needs %lld",
                    (long long)s->slab_floats, slices * kMlpNets * kGradElems);
    hipStream_t st = (hipStream_t)stream;
    const dim3 tiles((unsigned)((B + kMlpBM - 1) / kMlpBM), kMlpNets), thr(kMlpThreads);
    const long long prow = (long long)tiles.x * kMlpNets;       // one row of loss sums per workgroup of the fused kernel
    if (s->partial_rows < prow)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: partials hold %lld rows, the launch needs %lld",
                    (long long)s->partial_rows, prow);

    // forward + loss + backward-data of each 64-sample tile in ONE launch (mlp_forward_kernel<true>), then the means
    MlpFwdParams F = {};
    F.obs = s->obs; F.idx = reinterpret_cast<const long long*>(s->idx); F.f_loc = s->f_loc; F.f_inv = s->f_inv; F.f_lo = s->f_lo; F.f_hi = s->f_hi;
    F.wpack = static_cast<const __bf16*>(s->wpack); F.bias = s->bias; F.head = nullptr;
    F.xs_in = static_cast<const __bf16*>(s->xs_in);
    if (F.xs_in) { F.idx = nullptr; F.f_loc = F.f_inv = F.f_lo = F.f_hi = nullptr; }       // everything was applied by pnr_mlp_gather
    F.xs = F.xs_in ? nullptr : static_cast<__bf16*>(s->xs); F.h1 = static_cast<__bf16*>(s->h1); F.h2 = static_cast<__bf16*>(s->h2);
    F.B = B; F.first_net = 0; F.n_nets = kMlpNets;
    F.rec_actions = s->actions; F.rec_logp = s->logp_old; F.rec_mean = s->mean_old; F.rec_log_std = s->log_std_old;
    F.rec_adv = s->adv; F.rec_vtarg = s->value_target; F.rec_values = s->value_old;
    F.kl_coeff = s->kl_coeff; F.ent_coeff = s->entropy_coeff;
    F.clip = s->clip_param; F.vf_clip = s->vf_clip_param; F.vf_coeff = s->vf_loss_coeff;
    F.g_head = s->g_head; F.partials = s->partials; F.adam_step = s->adam_step;
    F.dz1 = static_cast<__bf16*>(s->dz1); F.dz2 = static_cast<__bf16*>(s->dz2);
    hipLaunchKernelGGL(mlp_forward_kernel<true>, tiles, thr, 0, st, F);
    if (s->flat_grad)       // no Adam launch here (the caller all-reduces first): the loss means get a small launch of their own
        hipLaunchKernelGGL(ppo_loss_finish_split_kernel, dim3(1), dim3(256), 0, st, s->partials, prow, B, s->means, (float*)nullptr,
                           s->kl_coeff, s->entropy_coeff, s->vf_loss_coeff);
    MlpWgradParams Wp;
    Wp.g_head = s->g_head; Wp.xs = F.xs_in ? F.xs_in : F.xs; Wp.h1 = F.h1; Wp.h2 = F.h2; Wp.dz1 = F.dz1; Wp.dz2 = F.dz2;
    Wp.slabs = s->slabs; Wp.B = B; Wp.slice_rows = rows;
    hipLaunchKernelGGL(mlp_wgrad_kernel, dim3((unsigned)slices, kWgParts, kMlpNets), thr, 0, st, Wp);
    if (s->flat_grad)
        hipLaunchKernelGGL(mlp_reduce_flat_kernel, dim3((kMlpNets * kGradElems + 255) / 256), dim3(256), 0, st, s->slabs, (int)slices, s->flat_grad);
    else
        mlp_launch_adam(s, s->slabs, (int)slices, 1.0f, st, prow);      // + the loss means, in the same launch
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_mlp_adam(const pnr_mlp_step* s, const float* flat_grad, float grad_scale, void* stream)
{
    int rc = mlp_step_check(s, "pnr_mlp_adam");
    if (rc) return rc;
    if (!flat_grad) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_adam: null gradient");
    mlp_launch_adam(s, flat_grad, 1, grad_scale, (hipStream_t)stream);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_get_state(pnr_handle h, uint32_t* words_out, void* stream)
{
    if (!h || !words_out) return fail(h, PNR_ERR_INVALID, "pnr_get_state: null argument");
    DeviceGuard g(h->device);
    hipLaunchKernelGGL(state_to_words_kernel, dim3((unsigned)((2 * h->n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, h->state, words_out, h->n);
    HIP_TRY(h, hipGetLastError());
    return PNR_OK;
}

int pnr_set_state(pnr_handle h, const uint32_t* words_in, void* stream)
{
    if (!h || !words_in) return fail(h, PNR_ERR_INVALID, "pnr_set_state: null argument");
    DeviceGuard g(h->device);
    h->kin_set = true;
    if (!h->dyn || h->dyn_set) h->ready = true;
    hipLaunchKernelGGL(words_to_state_kernel, dim3((unsigned)((2 * h->n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, h->state, words_in, h->n);
    HIP_TRY(h, hipGetLastError());
    return PNR_OK;
}

int pnr_get_dyn_state(pnr_handle h, float* words_out, void* stream)
{
    if (!h || !words_out) return fail(h, PNR_ERR_INVALID, "pnr_get_dyn_state: null argument");
    if (!h->dyn) return fail(h, PNR_ERR_UNSUPPORTED, "handle is not in dynamics mode");
    DeviceGuard g(h->device);
    HIP_TRY(h, hipMemcpyAsync(words_out, h->dyn, sizeof(float) * PNR_DYN_STATE_WORDS * (size_t)h->n,
                              hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PNR_OK;
}

int pnr_set_dyn_state(pnr_handle h, const float* words_in, void* stream)
{
    if (!h || !words_in) return fail(h, PNR_ERR_INVALID, "pnr_set_dyn_state: null argument");
    if (!h->dyn) return fail(h, PNR_ERR_UNSUPPORTED, "handle is not in dynamics mode");
    DeviceGuard g(h->device);
    h->dyn_set = true;
    if (h->kin_set) h->ready = true;
    HIP_TRY(h, hipMemcpyAsync(h->dyn, words_in, sizeof(float) * PNR_DYN_STATE_WORDS * (size_t)h->n,
                              hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PNR_OK;
}

}  // extern "C"
