// pnr_env_kernels.h — the env-side kernels of the engine: step / rollout (kinematic), the dynamics-mode step and rollout
// kernels, reset / observe, and the state-layout converters.  Included by pnr_api.hip only (which holds the C ABI and the
// launches); kept in a file of its own so that the counter passes under profiles/ can be tied to exactly the sources that
// define these kernels (pioneer_amd._lib.source_fingerprint, bench.py's `roofline.traffic`).
//
// Execution shape: see the header of pnr_api.hip and DESIGN.md section 3.
#pragma once

#include <hip/hip_runtime.h>

#include "pnr_device.h"
#include "pnr_dyn.h"

#ifndef PNR_DIAG_BUILD
#define PNR_DIAG_BUILD 0
#endif

namespace pnr {

// ---------------------------------------------------------------------------------
// step / rollout kernel: BulletEnv.step (bullet_env.py:192-197) for T steps.
// One wave = 32 envs (lane pair per env), one wave per workgroup.
// ---------------------------------------------------------------------------------
// The leading scalar parameters repeat P.state / P.actions / P.n / P.dt / P.eps and carry max_v_to_r (v_max
// is formed from it and the constexpr limits): plain leading arguments (up to 14 dwords) are preloaded into SGPRs by the command processor (-mllvm
// -amdgpu-kernarg-preload-count), so neither the first state and action loads nor the integrator wait for
// a kernarg fetch; the by-value struct, needed from the reward block on, is fetched behind them.
// PNR_STEP_WAVES waves per workgroup (default 4).  The waves of a workgroup share nothing — each has its own obs tile and walks its
// own tiles, no barrier — so this only changes what the dispatcher places: with four-wave workgroups a rollout's 1 024 waves are 256
// workgroups = exactly one per CU, one wave per SIMD, where 1 024 one-wave workgroups landed unevenly (A/B r03 in one run, twice:
// 7.36 -> 6.33 us per rollout step at 65 536 envs; single steps unchanged, 10.38 vs 10.31 us).
#ifndef PNR_STEP_WAVES
#define PNR_STEP_WAVES 4
#endif
constexpr int kStepWaves = PNR_STEP_WAVES;

template <bool OBS_EM, bool ACT_EM>
__global__ __launch_bounds__(kWave * kStepWaves) void step_kernel(float4* __restrict__ state_, const float* __restrict__ actions_,
                                                     const long long n_, const double dt_, const double eps_,
                                                     const float max_v_to_r_, const KParams P)
{
    __shared__ __attribute__((aligned(16))) float tiles_[kStepWaves * kTileFloats];
    float* tile = tiles_ + (threadIdx.x >> 6) * kTileFloats;

    const int lane = threadIdx.x & (kWave - 1);
    const int p = lane & 1;                 // which half of the env's joints
    const int el = lane >> 1;               // env within the wave's tile
    const long long n = n_;
    const long long ntiles = (n + kEnvsPerWave - 1) / kEnvsPerWave;

    // timing-only ablations (outputs are wrong when set; DESIGN.md "Where the time goes"): compiled in by -DPNR_DIAG_BUILD=1
    // only — in the product library `diag` is the literal 0 and every branch on it folds away
    const int diag = PNR_DIAG_BUILD ? P.diag : 0;
    const bool diag_noflush = diag & 2, diag_noemit = diag & 4, diag_nostate = diag & 8;

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

    const long long tstride = (long long)gridDim.x * kStepWaves;
    long long tix = (long long)blockIdx.x * kStepWaves + (threadIdx.x >> 6);
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

    for (; tix < ntiles; tix += tstride) {
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
        const long long nt = tix + tstride;
        if (nt < ntiles && nt * kEnvsPerWave + el < n) {
            raw = load_state_raw(state_, n, 2 * nt * kEnvsPerWave + lane);
            load_act0(nt * kEnvsPerWave + el, act0);
        } else {
            raw = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        }
    }

    if (diag & 16) { if (valid) P.reward[e] = s.pot; continue; }   // launch + state-load floor

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
                if (!(diag & 32)) integrate_joint(s.a[i], s.v[i], s.r[i], vmax[i], -lim, lim, dt_, eps_, s.v[i], s.r[i]);
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
        const bool done = done_predicate(s, p, q.dist, P.done_dist, P.done_dist_d);   // :160
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
    const bool done = done_predicate(o, p, q.dist, P.done_dist, P.done_dist_d);   // :160
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
    // (timing-only ablations of a -DPNR_DIAG_BUILD=1 variant, PNR_DIAG bits as in step_kernel: 2 no obs flush, 4 no obs emit; in the
    // product library `diag` is the literal 0)
    const int diag = PNR_DIAG_BUILD ? P.diag : 0;
    if (tile_in_use) wave_lds_sync();       // previous flush done before the tile is rewritten
    if (OBS_EM) {
        SinkLdsTile sink{tile + el * kObsDim, kJpl * p, p};
        if (!(diag & 4)) emit_obs<false>(K, o, q, p, sink);
        wave_lds_sync();
        if (!(diag & 2)) flush_tile(tile, P.obs + tile0 * kObsDim, nvalid, lane);
    } else {
        SinkLdsFeatureTile sink{tile + el, kJpl * p, p};
        emit_obs<false>(K, o, q, p, sink);
        wave_lds_sync();
        flush_feature_tile(tile, P.obs + tile0, n, nvalid, lane);
    }
}

// Leading scalar arguments as in step_kernel: preloaded into SGPRs, they repeat P.state / D.dyn / P.actions /
// P.n / P.dt / P.eps and carry max_v_to_r.
// PNR_DYN_STEP_WAVES = 2 (default): the workgroup is TWO waves.  Wave 0 runs phase A for the 64 envs (wave 1 waits at the barrier
// and takes no issue slot); in phase B each wave finishes ONE of the two 32-env tiles in an obs tile of its own, side by side on
// two SIMDs, instead of wave 0 finishing them one after the other (phase B is ~40 % of the step's fixed cost).  1 = the r02 form (A/B).
#ifndef PNR_DYN_STEP_WAVES
#define PNR_DYN_STEP_WAVES 2
#endif
constexpr int kDynStepWaves = PNR_DYN_STEP_WAVES;
static_assert(kDynStepWaves == 1 || kDynStepWaves == 2, "one or two waves per dynamics-step workgroup");

template <bool OBS_EM, bool ACT_EM, bool RAND, int PHYS>
__global__ __launch_bounds__(kWave * kDynStepWaves) void dyn_step_kernel(const float4* __restrict__ state_, const float* __restrict__ dyn_,
                                                         const float* __restrict__ actions_, const long long n_,
                                                         const double dt_, const double eps_, const float max_v_to_r_,
                                                         const KParams P, const DynParams D)
{
    __shared__ __attribute__((aligned(16))) float lds[kDynStepWaves * kTileFloats];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const long long n = n_;
    const long long base = (long long)blockIdx.x * kDynEnvsPerWg;
    float* tile = lds + wv * kTileFloats;                         // this wave's obs tile
    float4* hrec = reinterpret_cast<float4*>(lds);                // hand-off (head of tile 0): [3][2 * 64] records, index 2 * env + p
    float* hq = lds + kHandRecFloats;                             // [12][64]: q then qd
    const auto handoff_sync = [] { if (kDynStepWaves == 1) wave_lds_sync(); else __syncthreads(); };

    // ---- phase A: one env per lane (wave 0)
    if (wv == 0) {
        const long long e = base + lane;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 k0[2] = {z4, z4}, k1[2] = {z4, z4}, k2[2] = {z4, z4};
        float q[kDof] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, qd[kDof] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const DynLead lead = {state_, dyn_, actions_, n_, dt_, eps_, max_v_to_r_};
        if (e < n) dyn_substeps_lane<ACT_EM, RAND, PHYS>(lead, D, e, k0, k1, k2, q, qd, lds);   // the LDS is free during phase A
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
    handoff_sync();

    // ---- phase B: lane pairs; every tile's hand-off records leave LDS before tile 0 (which they alias) is reused
    const int p = lane & 1, el = lane >> 1;
    RawState raw0, raw1;
    float dq0[kJpl], dqd0[kJpl], dq1[kJpl], dqd1[kJpl];
    {
        // one wave: tiles 0 and 1; two waves: wave wv takes tile wv (raw0 / dq0 / dqd0)
        const int t0 = kDynStepWaves == 1 ? 0 : wv;
        const int r0 = 2 * kEnvsPerWave * t0 + lane, r1 = 2 * kEnvsPerWave + lane;        // record 2 * env + p
        raw0 = {hrec[r0], hrec[2 * kDynEnvsPerWg + r0], hrec[4 * kDynEnvsPerWg + r0]};
        raw1 = {hrec[r1], hrec[2 * kDynEnvsPerWg + r1], hrec[4 * kDynEnvsPerWg + r1]};
#pragma unroll
        for (int i = 0; i < kJpl; ++i) {
            dq0[i] = hq[(kJpl * p + i) * kDynEnvsPerWg + kEnvsPerWave * t0 + el];
            dqd0[i] = hq[(kDof + kJpl * p + i) * kDynEnvsPerWg + kEnvsPerWave * t0 + el];
            dq1[i] = hq[(kJpl * p + i) * kDynEnvsPerWg + kEnvsPerWave + el];
            dqd1[i] = hq[(kDof + kJpl * p + i) * kDynEnvsPerWg + kEnvsPerWave + el];
        }
    }
    handoff_sync();
    const LaneConsts K = lane_consts(p);
    // the 36 constant obs entries of this lane's tile slots: once per kernel
    if (OBS_EM) { SinkLdsTile sink{tile + el * kObsDim, kJpl * p, p}; emit_obs_const(K, sink); }
    else { SinkLdsFeatureTile sink{tile + el, kJpl * p, p}; emit_obs_const(K, sink); }

    if (kDynStepWaves == 1) {
        dyn_finish_tile<OBS_EM>(P, D, K, raw0, dq0, dqd0, tile, base, lane, false);
        if (base + kEnvsPerWave < n)
            dyn_finish_tile<OBS_EM>(P, D, K, raw1, dq1, dqd1, tile, base + kEnvsPerWave, lane, true);
    } else if (base + (long long)kEnvsPerWave * wv < n) {
        dyn_finish_tile<OBS_EM>(P, D, K, raw0, dq0, dqd0, tile, base + (long long)kEnvsPerWave * wv, lane, false);
    }
}


// independent waves per workgroup of dyn_rollout_kernel (own LDS slice, own 64 envs each).  A/B r03 in one run, twice: 25.60 us per
// rollout step with four (256 workgroups = one wave per SIMD by construction) against 25.65 with one: its waves already landed one per
// SIMD, so the r02 form stays.
#ifndef PNR_DYN_ROLLOUT_WAVES
#define PNR_DYN_ROLLOUT_WAVES 1
#endif
constexpr int kDynRolloutWaves = PNR_DYN_ROLLOUT_WAVES;
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
    const bool done = done_predicate(o, p, q.dist, P.done_dist, P.done_dist_d);   // :160
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
__global__ __launch_bounds__(kWave * kDynRolloutWaves) void dyn_rollout_kernel(const float4* __restrict__ state_, const float* __restrict__ dyn_,
                                                         const float* __restrict__ actions_, const long long n_,
                                                         const double dt_, const double eps_, const float max_v_to_r_,
                                                         const KParams P, const DynParams D)
{
    // kDynRolloutWaves independent waves per workgroup (each with its own LDS slice and its own 64 envs; no barrier between them):
    // at 65 536 envs the 1 024 waves are 256 workgroups = exactly one wave per SIMD, whatever the dispatcher does
    constexpr int kPerWave = kTileFloats + kComFloats + kRstFloats + kHandFloats + (PNR_DYN_LDS_MODEL ? kDynStageWords * 64 : 0);
    __shared__ __attribute__((aligned(16))) float lds_all[kDynRolloutWaves * kPerWave];
    float* lds = lds_all + (threadIdx.x >> 6) * kPerWave;
    float* tile = lds;
    float4* com = reinterpret_cast<float4*>(lds + kTileFloats);   // [2][64] common words between steps, index tile * 64 + lane
    float* rst = lds + kTileFloats + kComFloats;                  // [7][64] reset notes, phase B -> phase A
    float* hand = lds + kTileFloats + kComFloats + kRstFloats;    // records + q, qd planes (kHandFloats)
    const int lane = threadIdx.x & (kWave - 1);
    const int p = lane & 1, el = lane >> 1;
    const long long n = n_;
    const long long base = ((long long)blockIdx.x * kDynRolloutWaves + (threadIdx.x >> 6)) * kDynEnvsPerWg;
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
    if (base >= n) return;          // a wave past the end of the batch (whole wave: no barrier follows in this kernel)
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

// ---------------------------------------------------------------------------------
// World.step (bullet_scene.py:273-275), pnr_world_step: frame_skip sub-steps of the simulator and NOTHING else — no command
// integration, reward, TimeLimit, reset or observation.  One env per lane (the phase-A shape of dyn_step_kernel); the joints'
// motors are the per-joint table W (Joint.control_position / control_velocity, bullet_scene.py:123-155).
// ---------------------------------------------------------------------------------
template <bool RAND, int PHYS>
__global__ __launch_bounds__(kWave) void dyn_world_kernel(const float4* __restrict__ state, float* __restrict__ dyn, const long long n,
                                                          const DynParams D, const JointMotorTable W)
{
    const long long e = (long long)blockIdx.x * kWave + threadIdx.x;
    if (e >= n) return;
    const long long n2 = 2 * n;
    float a[kDof], v[kDof], r[kDof], q[kDof], qd[kDof], sc[kNumLinks], fric[kDof], damp[kDof], act[kDof];
#pragma unroll
    for (int p = 0; p < 2; ++p) {           // the env's command state r, v: what a joint without a command of its own tracks
        const float4 k0 = state[2 * e + p], k1 = state[n2 + 2 * e + p], k2 = state[2 * n2 + 2 * e + p];
        a[3 * p] = k0.x; a[3 * p + 1] = k0.y; a[3 * p + 2] = k0.z; v[3 * p] = k0.w;
        v[3 * p + 1] = k1.x; v[3 * p + 2] = k1.y; r[3 * p] = k1.z; r[3 * p + 1] = k1.w;
        r[3 * p + 2] = k2.x;
    }
#pragma unroll
    for (int i = 0; i < kDof; ++i) {
        q[i] = dyn[(long long)i * n + e]; qd[i] = dyn[(long long)(6 + i) * n + e];
        fric[i] = dyn[(long long)(23 + i) * n + e]; damp[i] = dyn[(long long)(29 + i) * n + e];
        act[i] = 0.f;
    }
#pragma unroll
    for (int l = 0; l < kNumLinks; ++l) sc[l] = RAND ? dyn[(long long)(12 + l) * n + e] : 1.0f;
    const DynLead lead = {state, dyn, nullptr, n, 0.0, 0.0, 0.f};
    dyn_core<PHYS, true>(lead, D, a, v, r, q, qd, sc, fric, damp, act, nullptr, &W);
#pragma unroll
    for (int i = 0; i < kDof; ++i) { dyn[(long long)i * n + e] = q[i]; dyn[(long long)(6 + i) * n + e] = qd[i]; }
}

// The same call on a kinematic-mode handle: there is no simulated state, the caller holds the joints as Bullet would after
// resetJointState(position, velocity) — js [n][12] = q[6] | qd[6] — and with no gravity, no motor and no collision shapes
// (the reference's URDF and defaults) frame_skip x stepSimulation carries each joint on at its velocity:
// q += qd * step_time, stopped at its limit with the velocity zeroed there.
__global__ void kin_world_kernel(float* __restrict__ js, long long n, float step_time)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * kDof) return;
    const long long e = i / kDof; const int j = (int)(i % kDof);
    constexpr float lim[kDof] = {limit_hi(0), limit_hi(1), limit_hi(2), limit_hi(3), limit_hi(4), limit_hi(5)};
    const float hi = lim[j];
    float q = js[e * 12 + j], qd = js[e * 12 + 6 + j];
    q = __builtin_fmaf(qd, step_time, q);
    if (q >= hi) { q = hi; qd = 0.f; }
    if (q <= -hi) { q = -hi; qd = 0.f; }
    js[e * 12 + j] = q; js[e * 12 + 6 + j] = qd;
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
