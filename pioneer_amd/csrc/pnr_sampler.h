// pnr_sampler.h — the PPO sampler's closed loop as ONE resident kernel (pnr_ppo_rollout).
//
// What it replaces: T x (pnr_mlp_act + pnr_step), two dependent launches per sampler step — 18.5 + 6.1 us of kernels and
// 5.4 us of launch boundaries per step at 16 384 envs (profiles/r03_j_ppo_loop_kernel_stats.csv), 13 % of a PPO iteration.
// Envs are independent and a 64-env tile needs nothing from any other tile, so ONE workgroup owns 64 envs for the whole
// rollout: per step it runs both nets on the tile's observation (the forward pass of mlp_forward_kernel<false>, same
// arithmetic in the same order), draws the action in the policy head's epilogue, steps its 64 envs (the body of
// step_kernel's loop: pnr_device.h), and makes the next input tile from the observation while it is still in LDS.  Nothing
// crosses a launch boundary inside the rollout, the env state stays on the CU, and each wave keeps its blocks of BOTH nets'
// W2 in registers for all T steps (weight-stationary: with one workgroup per CU there is no
// second workgroup to hide the L2 latency of a weight stream behind).  Outputs are those of the two-launch form, bit for bit
// (tests/test_gpu_ppo.py::test_resident_rollout_equals_the_two_launch_sampler).
// Kinematic mode, env-major layouts (pnr_env_rollout_params checks); grid = ceil(n / 64) workgroups of 512 threads.
#pragma once

#include "pnr_device.h"
#include "pnr_mlp.h"

#ifndef PNR_ROLL_HOOKS
#define PNR_ROLL_HOOKS 1          // 1: the step's global stores are issued from inside the layer-1 products (0: where they arise; the A/B)
#endif
#ifndef PNR_ROLL_DIAG
#define PNR_ROLL_DIAG 0           // timing-only ablations of ppo_rollout_kernel (variant builds; results are wrong when set): 1 no next-input
#endif                            // construction, 2 no xs store, 4 no env step, 8 no value net, 16 no policy head / draw

namespace pnr {

struct RolloutParams {
    KParams K;                 // the handle's constants; state; obs = slot 1 of the observation buffer ([T][n][137]), reward, done,
                               // trunc ([T][n]; trunc may be null)
    const float* obs0;         // [n][137] the observation the rollout starts from (slot 0)
    float max_v_to_r;
    int T;
    const float* f_loc; const float* f_inv; const float* f_lo; const float* f_hi;   // MeanStdFilter vectors, all four or none
    const __bf16* wpack;       // [2][kPackElems]
    const float* bias;         // [2][kBiasElems]
    const float* noise;        // [T][n][6] standard-normal draws
    const float* a_max;        // [6] or null: no clipping
    float* mean; float* log_std; float* actions;      // [T][n][6]
    float* values;             // [T][n]
    __bf16* xs;                // [T][n][144] the nets' inputs as they saw them, or null
};

constexpr int kRollLdsBytes = (kMlpBM * kXS + 2 * kMlpBM * kHS) * 2 + 2 * kTileFloats * 4 + kMlpNets * kMlpHead * kMlpHid * 2 +
                              4 * kMlpInPad * 4 + kMlpNets * kBiasElems * 4 + kMlpBM * 8 * 4 + 3 * 2 * kMlpBM * 16;
static_assert(kRollLdsBytes <= 160 * 1024, "the rollout tile's LDS fits one CU");
static_assert(kMlpBM == 2 * kEnvsPerWave, "a 64-sample tile is two 32-env lane-pair tiles");

__global__ __launch_bounds__(kFwdThreads, 2) void ppo_rollout_kernel(const RolloutParams S)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kRollLdsBytes];
    __bf16* xt = reinterpret_cast<__bf16*>(lds_raw);                 // [64][144] the nets' input tile
    __bf16* h1t = xt + kMlpBM * kXS;                                 // [64][256] H1
    __bf16* h2t = h1t + kMlpBM * kHS;                                // [64][256] H2
    float* ot = reinterpret_cast<float*>(h2t + kMlpBM * kHS);        // [2][32][137] the env waves' observation tiles (float32)
    __bf16* w3l = reinterpret_cast<__bf16*>(ot + 2 * kTileFloats);   // [2 nets] W3, fragment-native
    float* fv = reinterpret_cast<float*>(w3l + kMlpNets * kMlpHead * kMlpHid);   // loc | inv | lo | hi, 4 x 144
    float* bl = fv + 4 * kMlpInPad;                                  // [2 nets][b1 | b2 | b3]
    float* actl = bl + kMlpNets * kBiasElems;                        // [64][8] the step's env actions
    float4* stl = reinterpret_cast<float4*>(actl + kMlpBM * 8);      // [3][128] the 64 envs' state records (pnr_device.h's planes)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const long long n = S.K.n, e0 = (long long)blockIdx.x * kMlpBM;
    const bool filt = S.f_loc != nullptr;
    const int wv = w - 4;                                            // env waves: 4 and 5, each a lane-pair tile of 32 envs
    const bool env_wave = w == 4 || w == 5;

    // ---- once per launch: filter vectors, biases, W3 of both nets, the stationary W2 blocks, the envs' state, the constant
    // entries of the observation tiles
    for (int i = tid; i < 4 * kMlpInPad; i += kFwdThreads) {
        const int which = i / kMlpInPad, k = i % kMlpInPad;
        const float* src = which == 0 ? S.f_loc : (which == 1 ? S.f_inv : (which == 2 ? S.f_lo : S.f_hi));
        fv[i] = (filt && k < kMlpIn) ? src[k] : 0.f;
    }
    for (int i = tid; i < kMlpNets * kBiasElems; i += kFwdThreads) bl[i] = S.bias[i];
    static_assert(kMlpHead * kMlpHid / 8 == kFwdThreads, "W3: one 16-byte piece per thread and net");
#pragma unroll
    for (int net = 0; net < kMlpNets; ++net)
        *reinterpret_cast<uint4*>(w3l + net * kMlpHead * kMlpHid + 8 * tid) =
            *reinterpret_cast<const uint4*>(S.wpack + (size_t)net * kPackElems + kOffW3 + 8 * tid);
    bf16x8 w2p[kMlpHid / 16], w2v[kMlpHid / 16];                     // this wave's row block of W2: policy, value
#pragma unroll
    for (int ks = 0; ks < kMlpHid / 16; ++ks) {
        w2p[ks] = ld_global_bf16x8(S.wpack + kOffW2 + (w * (kMlpHid / 16) + ks) * 512 + lane * 8);
        w2v[ks] = ld_global_bf16x8(S.wpack + kPackElems + kOffW2 + (w * (kMlpHid / 16) + ks) * 512 + lane * 8);
    }
    MlpGemm1<kMlpInPad, kXS> g1;                                     // W1 streams through the ring, requested a phase ahead
    g1.prefetch(S.wpack + kOffW1 + w * (kMlpInPad / 16) * 512, lane);
    if (env_wave) {
        const int p = lane & 1, el = lane >> 1;
        const long long tile0 = e0 + kEnvsPerWave * wv;
        RawState raw = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        if (tile0 + el < n) raw = load_state_raw(S.K.state, n, 2 * tile0 + lane);
        stl[kWave * wv + lane] = raw.p0; stl[2 * kMlpBM + kWave * wv + lane] = raw.p1; stl[4 * kMlpBM + kWave * wv + lane] = raw.p2;
        const LaneConsts K = lane_consts(p);
        SinkLdsTile sink{ot + wv * kTileFloats + el * kObsDim, kJpl * p, p};
        emit_obs_const(K, sink);
    }
    mlp_barrier();

    // the nets' input: x = clamp((obs - loc) * inv, lo, hi) rounded to bf16, columns 137.. zero (mlp_forward_kernel's stage 0:
    // eight threads per row, 18 columns each); `row_ptr` = the row's 137 float32 observation entries, global or LDS
    constexpr int TPR = kFwdThreads / kMlpBM, CPT = kMlpInPad / TPR;
    static_assert(CPT == 18, "18 columns per thread");
    const auto make_x = [&](const float* row_ptr, bool live, int row, int part) {
        // (every load unconditional, from an address that exists — the padding columns re-read column 136 — and masked afterwards:
        // a load under a per-element condition becomes a branch with its own s_waitcnt, i.e. 18 round trips in a row)
        float x[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) x[j] = row_ptr[CPT * part + j < kMlpIn ? CPT * part + j : kMlpIn - 1];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int col = CPT * part + j;
            float y = x[j];
            if (filt) y = fminf(fmaxf((y - fv[col]) * fv[kMlpInPad + col], fv[2 * kMlpInPad + col]), fv[3 * kMlpInPad + col]);
            x[j] = (live && col < kMlpIn) ? y : 0.f;
        }
#pragma unroll
        for (int j = 0; j < CPT; j += 2) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const bf16x2 pk = {(__bf16)x[j], (__bf16)x[j + 1]};
            *reinterpret_cast<bf16x2*>(xt + row * kXS + CPT * part + j) = pk;
        }
    };
    {
        const int row = tid / TPR, part = tid % TPR;
        const bool live = e0 + row < n;
        make_x(S.obs0 + (live ? e0 + row : e0) * kMlpIn, live, row, part);
    }
    mlp_barrier();

    const auto bias16 = [&](const float* b) {                         // this wave's 16 bias values of a layer: an MFMA's C operand
        f32x16 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(b + 32 * w + 8 * k + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * k + j] = q[j];
        }
        return v;
    };
    const auto tanh_tile = [&](const f32x16 (&acc)[kMlpCB], __bf16* tile) {
#pragma unroll
        for (int cb = 0; cb < kMlpCB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<bf16x4*>(tile + (32 * cb + c) * kHS + 32 * w + 8 * q + 4 * h) = tanh_quad(acc[cb], q);
                __builtin_amdgcn_sched_barrier(0);      // one quad at a time: the stationary weights leave no room for 32 interleaved chains
            }
    };
    // layers 1 and 2 of one net on the tile in xt: H2 in h2t, behind a barrier
    // (`hook` runs inside layer 1's product, right behind its last weight-fragment request: where this step's global STORES go — a
    // wave's vector-memory operations retire in order, so a fragment requested behind a store waits for that store's acknowledgement
    // from HBM, ~2 us per step as first written: profiles/r03_g_rollout_ablation.json)
    const auto hidden_layers = [&](int net, const bf16x8 (&w2)[kMlpHid / 16], const __bf16* next_w1, auto&& hook) {
        f32x16 acc[kMlpCB];
        {
            const f32x16 b16 = bias16(bl + net * kBiasElems);
            g1.run(xt, acc, lane, hook, &b16);
        }
        g1.prefetch(next_w1, lane);                                   // the ring is free: the next forward's first fragments
        tanh_tile(acc, h1t);
        mlp_barrier();
        {
            const f32x16 b16 = bias16(bl + net * kBiasElems + kMlpHid);
            const int r = lane & 31;
            const __bf16* tb = h1t + r * kHS + 8 * h;
            bf16x8 b[2][kMlpCB];
#pragma unroll
            for (int cb = 0; cb < kMlpCB; ++cb) b[0][cb] = *reinterpret_cast<const bf16x8*>(tb + cb * 32 * kHS);
#pragma unroll
            for (int ks = 0; ks < kMlpHid / 16; ++ks) {
                if (ks + 1 < kMlpHid / 16) {
#pragma unroll
                    for (int cb = 0; cb < kMlpCB; ++cb) b[(ks + 1) & 1][cb] = *reinterpret_cast<const bf16x8*>(tb + cb * 32 * kHS + 16 * (ks + 1));
                }
#pragma unroll
                for (int cb = 0; cb < kMlpCB; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[ks], b[ks & 1][cb], ks == 0 ? b16 : acc[cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        tanh_tile(acc, h2t);
        mlp_barrier();
    };
    // head^T [16][16 samples of wave w < 4] = W3 . H2^T + b3 (16 x 16 x 32 MFMAs, W3's fragments from LDS)
    const auto head16 = [&](int net) {
        const int r16 = lane & 15, g = lane >> 4;
        f32x4 a3 = *reinterpret_cast<const f32x4*>(bl + net * kBiasElems + 2 * kMlpHid + 4 * g);
#pragma unroll
        for (int ks = 0; ks < kMlpHid / 32; ++ks) {
            const bf16x8 bq = *reinterpret_cast<const bf16x8*>(h2t + (16 * w + r16) * kHS + 32 * ks + 8 * g);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(w3l + net * kMlpHead * kMlpHid + 512 * ks + lane * 8), bq, a3, 0, 0, 0);
        }
        return a3;
    };

    for (int t = 0; t < S.T; ++t) {
        // the thread id as the loop body sees it: opaque, so that per-thread global addresses are formed where they are used
        // instead of living in registers around the loop
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const long long tn = (long long)t * n;
        const auto xs_store = [&] {
            if (S.xs && !(PNR_ROLL_DIAG & 2)) {                           // the nets' input as they see it, for the learner: 1 152 16-byte pieces
                constexpr int kCh = kMlpBM * (kMlpInPad / 8), kIt = (kCh + kFwdThreads - 1) / kFwdThreads;
                uint4 v[kIt];
    #pragma unroll
                for (int i = 0; i < kIt; ++i) {
                    const int ch = tv + kFwdThreads * i, row = ch / (kMlpInPad / 8), cc = ch % (kMlpInPad / 8);
                    if (ch < kCh) v[i] = *reinterpret_cast<const uint4*>(xt + row * kXS + cc * 8);
                }
                    if (e0 + kMlpBM <= n) {                               // a whole tile: 18 KB contiguous in xs
                    __bf16* base = S.xs + (tn + e0) * kMlpInPad + tv * 8;
#pragma unroll
                    for (int i = 0; i < kIt; ++i)
                        if (i < kCh / kFwdThreads || tv + kFwdThreads * i < kCh) *reinterpret_cast<uint4*>(base + (size_t)i * kFwdThreads * 8) = v[i];
                } else {
#pragma unroll
                    for (int i = 0; i < kIt; ++i) {
                        const int ch = tv + kFwdThreads * i, row = ch / (kMlpInPad / 8), cc = ch % (kMlpInPad / 8);
                        if (ch < kCh && e0 + row < n) *reinterpret_cast<uint4*>(S.xs + (tn + e0 + row) * kMlpInPad + cc * 8) = v[i];
                    }
                }
            }
        };
        // ---- policy net, the DiagGaussian draw in its head's epilogue (mlp_forward_kernel<false>'s, to the letter)
        if (!PNR_ROLL_HOOKS) xs_store();
        hidden_layers(0, w2p, S.wpack + kPackElems + kOffW1 + (tv >> 6) * (kMlpInPad / 16) * 512, [&] { if (PNR_ROLL_HOOKS) xs_store(); });
        // the head's outputs wait in registers for the value net's layer-1 hook (pm / pa: what this lane stores to mean | log_std / actions)
        f32x4 pm = {0.f, 0.f, 0.f, 0.f}, pa = pm;
        size_t po = 0;
        bool plive = false;
        if (w < 4 && !(PNR_ROLL_DIAG & 16)) {
            const f32x4 hq = head16(0);
            const int r16 = lane & 15, g = lane >> 4;
            const long long b = e0 + 16 * w + r16;
            // g = 0 holds means 0..3, g = 1 means 4, 5 and raw log-stds 0, 1, g = 2 raw log-stds 2..5; six cross-lane reads put
            // each mean next to its log-std (executed by all lanes: no divergence around them)
            const auto ls = [](float x) { return fminf(fmaxf(x, -20.f), 2.f); };
            const float l0 = ls(__shfl(hq[2], r16 + 16)), l1 = ls(__shfl(hq[3], r16 + 16));
            const float l2 = ls(__shfl(hq[0], r16 + 32)), l3 = ls(__shfl(hq[1], r16 + 32));
            const float l4 = ls(__shfl(hq[2], r16 + 32)), l5 = ls(__shfl(hq[3], r16 + 32));
            if (g <= 2) {
                const bool live = b < n;
                const size_t o = (size_t)(tn + (live ? b : 0)) * kMlpAct;
                const auto draw = [&](int j, float m, float l, float& a, float& e) {
                    a = fmaf(expf(l), live ? S.noise[o + j] : 0.f, m);
                    e = S.a_max ? fminf(fmaxf(a, -S.a_max[j]), S.a_max[j]) : a;
                };
                float* al = actl + (16 * w + r16) * 8;
                plive = live; po = o;
                if (g == 0) {
                    float a[4], e[4];
                    draw(0, hq[0], l0, a[0], e[0]); draw(1, hq[1], l1, a[1], e[1]);
                    draw(2, hq[2], l2, a[2], e[2]); draw(3, hq[3], l3, a[3], e[3]);
                    pm = hq; pa = (f32x4){a[0], a[1], a[2], a[3]};
                    *reinterpret_cast<f32x4*>(al) = (f32x4){e[0], e[1], e[2], e[3]};
                } else if (g == 1) {
                    float a[2], e[2];
                    draw(4, hq[0], l4, a[0], e[0]); draw(5, hq[1], l5, a[1], e[1]);
                    pm = (f32x4){hq[0], hq[1], ls(hq[2]), ls(hq[3])}; pa = (f32x4){a[0], a[1], 0.f, 0.f};
                    al[4] = e[0]; al[5] = e[1];
                } else {
                    pm = (f32x4){ls(hq[0]), ls(hq[1]), ls(hq[2]), ls(hq[3])};
                }
            }
        }
        const auto head_store = [&] {
            if (w < 4 && plive) {
                const int g = lane >> 4;
                typedef float f32x2s __attribute__((ext_vector_type(2)));
                const auto st2 = [&](float* dst, float x, float y) { *reinterpret_cast<f32x2s*>(dst) = (f32x2s){x, y}; };
                if (g == 0) {
                    st2(S.mean + po, pm[0], pm[1]); st2(S.mean + po + 2, pm[2], pm[3]);
                    st2(S.actions + po, pa[0], pa[1]); st2(S.actions + po + 2, pa[2], pa[3]);
                } else if (g == 1) {
                    st2(S.mean + po + 4, pm[0], pm[1]);
                    st2(S.actions + po + 4, pa[0], pa[1]);
                    st2(S.log_std + po, pm[2], pm[3]);
                } else if (g == 2) {
                    st2(S.log_std + po + 2, pm[0], pm[1]); st2(S.log_std + po + 4, pm[2], pm[3]);
                }
            }
        };
        // ---- value net
        if (!PNR_ROLL_HOOKS || (PNR_ROLL_DIAG & 8)) head_store();
        if (!(PNR_ROLL_DIAG & 8)) hidden_layers(1, w2v, S.wpack + kOffW1 + (tv >> 6) * (kMlpInPad / 16) * 512, [&] { if (PNR_ROLL_HOOKS) head_store(); });
        if (w < 4) {
            const f32x4 hq = head16(1);
            const long long b = e0 + 16 * w + (lane & 15);
            if (b < n && (lane >> 4) == 0) S.values[tn + b] = hq[0];
        } else if (env_wave && !(PNR_ROLL_DIAG & 4)) {
            // ---- the env step of this wave's 32 envs (step_kernel's loop body: BulletEnv.step, bullet_env.py:192-197), lane pair per env
            const int p = lane & 1, el = lane >> 1;
            const long long tile0 = e0 + kEnvsPerWave * wv, e = tile0 + el;
            const bool valid = e < n;
            const int nvalid = (int)((n - tile0) < kEnvsPerWave ? ((n - tile0) < 0 ? 0 : (n - tile0)) : kEnvsPerWave);
            const LaneConsts K = lane_consts(p);
            const float vmax[kJpl] = {S.max_v_to_r * (K.lim[0] - (-K.lim[0])), S.max_v_to_r * (K.lim[1] - (-K.lim[1])),
                                      S.max_v_to_r * (K.lim[2] - (-K.lim[2]))};
            LaneState s;
            {
                const RawState raw = {stl[kWave * wv + lane], stl[2 * kMlpBM + kWave * wv + lane], stl[4 * kMlpBM + kWave * wv + lane]};
                unpack_state(raw, p, s);
            }
            const float* al = actl + (kEnvsPerWave * wv + el) * 8 + kJpl * p;
            const float act[kJpl] = {al[0], al[1], al[2]};
            // act(): integrate the PREVIOUS action, then latch the new one (pioneer_knm_env.py:113-146, quirk Q1)
#pragma unroll
            for (int i = 0; i < kJpl; ++i) {
                const float lim = K.lim[i];
                integrate_joint(s.a[i], s.v[i], s.r[i], vmax[i], -lim, lim, S.K.dt, S.K.eps, s.v[i], s.r[i]);
            }
#pragma unroll
            for (int i = 0; i < kJpl; ++i) s.a[i] = act[i];
            s.step += 1;
            Pose q;
            compute_pose(s, p, q);
            // reward block, pioneer_knm_env.py:157-165 (both lanes, identical)
            const float old_pot = s.pot;
            const float pot = S.K.pot_m / (q.dist / S.K.pot_s + 1.0f);
            const bool done = done_predicate(s, p, q.dist, S.K.done_dist, S.K.done_dist_d);
            const float r_pot = pot - old_pot;
            const float r_step = -S.K.penalty;
            const float r_done = done ? S.K.award_done : 0.0f;
            const float rw = (r_pot + r_step) + r_done;
            s.pot = pot;
            const bool trunc = (S.K.max_steps > 0) && (s.step >= (uint32_t)S.K.max_steps) && !done;
            if (valid && p == 0) {
                stream_store(S.K.reward + tn + e, rw);
                stream_store(S.K.done + tn + e, (uint8_t)done);
                if (S.K.trunc) stream_store(S.K.trunc + tn + e, (uint8_t)trunc);
            }
            if (S.K.auto_reset && (done || trunc)) {
                reset_env(S.K, K, s, p, S.K.env_off + (unsigned long long)e, nullptr, nullptr);
                compute_pose(s, p, q);
            }
            store_state(stl, kMlpBM, kWave * wv + lane, p, s);
            // observe(): into this wave's tile (its constant entries are in place since the prologue), then to obs slot t + 1
            float* tile = ot + wv * kTileFloats;
            SinkLdsTile sink{tile + el * kObsDim, kJpl * p, p};
            emit_obs<false>(K, s, q, p, sink);
            wave_lds_sync();
            if (nvalid > 0) flush_tile(tile, S.K.obs + (tn + tile0) * kObsDim, nvalid, lane);
        }
        mlp_barrier();                                                // the observation tiles are complete
        // ---- the next step's input, from the tiles: thread = (column quad, group of eight rows) — the filter vectors of its four
        // columns are read once (16-byte LDS reads), the tile entries one by one (a row starts on a 4-byte boundary only)
        if (tv < (kMlpInPad / 4) * 8 && !(PNR_ROLL_DIAG & 1)) {
            const int cq = tv % (kMlpInPad / 4), rg = tv / (kMlpInPad / 4), col = 4 * cq;
            f32x4 loc = {0.f, 0.f, 0.f, 0.f}, inv = loc, lo = loc, hi = loc;
            if (filt) {
                loc = *reinterpret_cast<const f32x4*>(fv + col); inv = *reinterpret_cast<const f32x4*>(fv + kMlpInPad + col);
                lo = *reinterpret_cast<const f32x4*>(fv + 2 * kMlpInPad + col); hi = *reinterpret_cast<const f32x4*>(fv + 3 * kMlpInPad + col);
            }
            // (loads unconditional and all in flight before the first use; columns 137.. re-read column 136 and are masked)
            float raw[8][4];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* src = ot + ((8 * rg + k) >> 5) * kTileFloats + ((8 * rg + k) & 31) * kObsDim;
#pragma unroll
                for (int j = 0; j < 4; ++j) raw[k][j] = src[col + j < kMlpIn ? col + j : kMlpIn - 1];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = 8 * rg + k;
                const bool live = e0 + row < n;
                bf16x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float y = raw[k][j];
                    if (filt) y = fminf(fmaxf((y - loc[j]) * inv[j], lo[j]), hi[j]);
                    pk[j] = (__bf16)((live && col + j < kMlpIn) ? y : 0.f);
                }
                *reinterpret_cast<bf16x4*>(xt + row * kXS + col) = pk;
            }
        }
        mlp_barrier();
    }
    if (env_wave) {                                                   // the envs' state back to the handle
        const int el = lane >> 1;
        const long long tile0 = e0 + kEnvsPerWave * wv;
        if (tile0 + el < n) {
            const long long rec = 2 * tile0 + lane, n2 = 2 * n;
            S.K.state[rec] = stl[kWave * wv + lane];
            S.K.state[n2 + rec] = stl[2 * kMlpBM + kWave * wv + lane];
            S.K.state[2 * n2 + rec] = stl[4 * kMlpBM + kWave * wv + lane];
        }
    }
}

}  // namespace pnr
