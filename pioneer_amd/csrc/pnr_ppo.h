// pnr_ppo.h — the element-wise part of the PPO learner's loss, forward AND backward, as one kernel.
//
// The host driver (pioneer_amd/ppo.py) replaces RLlib's PPOTrainer loop with the hyper-parameters of
// pioneer/launch/pioneer_knm_train.py:45-67 (clip_param, vf_clip_param, kl_coeff, vf_loss_coeff,
// entropy_coeff).  Written with torch ops, the loss over a [B, 6] Gaussian head is ~40 small kernels
// forward and ~80 backward per minibatch — a quarter of the learn phase at B = 131 072.  Here one
// thread owns one sample: it reads the two heads' raw outputs and the rollout record, writes
// d loss / d head and accumulates the five reported means.  Semantics (ties included) are those of
// PPOLearner.loss() under torch autograd, which stays the CPU path and the checker
// (tests/test_gpu_ppo.py::test_fused_loss_matches_autograd).
#pragma once

#include <hip/hip_runtime.h>

namespace pnr {

constexpr int kPpoActDim = kDof;          // 6 means + 6 log-stds in a 16-float head row
constexpr int kPpoHeadStride = 16;
constexpr int kPpoBlock = 256;
constexpr int kPpoSums = 8;               // policy_loss, vf_loss, kl, entropy, total, 3 spare

struct PpoLossParams {
    const float* head_p;    // [B][16]: mean[0:6], raw log_std[6:12]
    const float* head_v;    // [B][16]: v at column 0
    const float* actions;   // [B][6]
    const float* logp_old;  // [B]
    const float* mean_old;  // [B][6]
    const float* ls_old;    // [B][6] (already clamped when it was recorded)
    const float* adv;       // [B]
    const float* vtarg;     // [B]
    const float* v_old;     // [B]
    const long long* idx;   // [B] row of the rollout record (actions .. v_old) for sample i, or null: row i
    const float* kl_coeff;  // device scalars: their values change between graph replays
    const float* ent_coeff;
    float* g_head_p;        // [B][16]  d loss / d head_p
    float* g_head_v;        // [B][16]  d loss / d head_v
    float* partials;        // [gridDim.x][8] per-block sums (summed by the caller: deterministic)
    long long B;
    float clip, vf_clip, vf_coeff;
};

// The policy half of one sample: head row (means m, raw log-stds) + rollout record -> d loss / d (mean, raw log-std)
// (already divided by the batch) and the sample's -surrogate, KL and entropy.
__device__ __forceinline__ void ppo_policy_sample(const float (&m)[6], const float (&raw)[6], const float (&a)[6], const float (&m0)[6],
                                                  const float (&l0)[6], float adv, float lp0, float clip, float klc, float entc,
                                                  float invB, float (&gm)[6], float (&gl)[6], float& neg_surr, float& kl_out,
                                                  float& ent_out)
{
    float ls[6], z[6], s[6];
    bool pass[6];
    float logp = -0.5f * 6.0f * 1.8378770664093453f;           // -3 log(2 pi)
    float kl = 0.f, ent = 6.0f * 1.4189385332046727f;          // 6 * 0.5 log(2 pi e)
    float dkl_m[6], dkl_l[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        pass[j] = raw[j] >= -20.0f && raw[j] <= 2.0f;          // torch.clamp passes the gradient on [min, max]
        ls[j] = fminf(fmaxf(raw[j], -20.0f), 2.0f);
        s[j] = expf(-ls[j]);
        z[j] = (a[j] - m[j]) * s[j];
        logp += -0.5f * z[j] * z[j] - ls[j];
        const float ivar = s[j] * s[j];
        const float d = m0[j] - m[j];
        const float q = (expf(2.0f * l0[j]) + d * d) * ivar;  // (var0 + (m0 - m)^2) / var
        kl += ls[j] - l0[j] + 0.5f * q - 0.5f;
        dkl_m[j] = -d * ivar;
        dkl_l[j] = 1.0f - q;
        ent += ls[j];
    }
    const float ratio = expf(logp - lp0);
    const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
    const float s1 = adv * ratio, s2 = adv * rc;
    const float surr = fminf(s1, s2);
    const bool inrange = ratio >= 1.0f - clip && ratio <= 1.0f + clip;
    // torch.minimum: the smaller argument takes the gradient, a tie splits it; the clipped branch is
    // constant outside the range
    float dsurr = 0.f;                                          // d surr / d logp
    if (s1 < s2) dsurr = s1;
    else if (s1 == s2) dsurr = 0.5f * s1 + (inrange ? 0.5f * s1 : 0.f);
    else dsurr = inrange ? s1 : 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        gm[j] = (-dsurr * z[j] * s[j] + klc * dkl_m[j]) * invB;
        gl[j] = pass[j] ? (-dsurr * (z[j] * z[j] - 1.0f) + klc * dkl_l[j] - entc) * invB : 0.f;
    }
    neg_surr = -surr; kl_out = kl; ent_out = ent;
}

// The value half: clipped value loss (RLlib: max of the plain and the clipped squared error) and d / d v
__device__ __forceinline__ void ppo_value_sample(float v, float vt, float v0, float vf_clip, float& vf, float& dvf)
{
    const float e1 = v - vt;
    const float dv = v - v0;
    const bool vin = dv >= -vf_clip && dv <= vf_clip;
    const float e2 = v0 + fminf(fmaxf(dv, -vf_clip), vf_clip) - vt;
    const float f1 = e1 * e1, f2 = e2 * e2;
    vf = fmaxf(f1, f2);
    const float g1 = 2.0f * e1, g2 = vin ? 2.0f * e2 : 0.f;
    dvf = f1 > f2 ? g1 : (f1 < f2 ? g2 : 0.5f * (g1 + g2));   // torch.maximum, ties split
}

__global__ __launch_bounds__(kPpoBlock) void ppo_loss_kernel(const PpoLossParams P)
{
    __shared__ float red[kPpoSums][kPpoBlock / 64];
    const long long i = (long long)blockIdx.x * kPpoBlock + threadIdx.x;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < P.B) {
        const float klc = *P.kl_coeff, entc = *P.ent_coeff;
        const float invB = 1.0f / (float)P.B;
        const float4* hp = reinterpret_cast<const float4*>(P.head_p + i * kPpoHeadStride);
        const float4 h0 = hp[0], h1 = hp[1], h2 = hp[2];
        const float m[6] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y};
        const float raw[6] = {h1.z, h1.w, h2.x, h2.y, h2.z, h2.w};
        const float v = P.head_v[i * kPpoHeadStride];
        const long long r = P.idx ? P.idx[i] : i;                   // minibatch gather of the rollout record
        float a[6], m0[6], l0[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { a[j] = P.actions[r * 6 + j]; m0[j] = P.mean_old[r * 6 + j]; l0[j] = P.ls_old[r * 6 + j]; }
        const float adv = P.adv[r], vt = P.vtarg[r], v0 = P.v_old[r], lp0 = P.logp_old[r];

        float gm[6], gl[6], neg_surr, kl, ent, vf, dvf;
        ppo_policy_sample(m, raw, a, m0, l0, adv, lp0, P.clip, klc, entc, invB, gm, gl, neg_surr, kl, ent);
        ppo_value_sample(v, vt, v0, P.vf_clip, vf, dvf);
        float4* gp = reinterpret_cast<float4*>(P.g_head_p + i * kPpoHeadStride);
        gp[0] = make_float4(gm[0], gm[1], gm[2], gm[3]);
        gp[1] = make_float4(gm[4], gm[5], gl[0], gl[1]);
        gp[2] = make_float4(gl[2], gl[3], gl[4], gl[5]);
        gp[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        float4* gv = reinterpret_cast<float4*>(P.g_head_v + i * kPpoHeadStride);
        gv[0] = make_float4(P.vf_coeff * dvf * invB, 0.f, 0.f, 0.f);
        gv[1] = gv[2] = gv[3] = make_float4(0.f, 0.f, 0.f, 0.f);

        acc[0] = neg_surr; acc[1] = vf; acc[2] = kl; acc[3] = ent;
        acc[4] = neg_surr + klc * kl + P.vf_coeff * vf - entc * ent;
    }
    // block sums: wave shuffle, then across the four waves through LDS
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float x = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x < kPpoSums) {
        float x = 0.f;
        if (threadIdx.x < 5) {
#pragma unroll
            for (int w = 0; w < kPpoBlock / 64; ++w) x += red[threadIdx.x][w];
        }
        P.partials[(long long)blockIdx.x * kPpoSums + threadIdx.x] = x;
    }
}

// Sum the per-block partial rows in row order (one block, deterministic) and divide by the batch: the five reported
// means.  A separate tiny launch instead of a last-block-done atomic: nothing here needs zeroed counters.
__global__ __launch_bounds__(64) void ppo_loss_finish_kernel(const float* __restrict__ partials, long long rows, long long B,
                                                             float* __restrict__ means, float* __restrict__ step_counter)
{
    if (step_counter && threadIdx.x == 0) *step_counter += 1.0f;    // the optimiser's update count (single block)
    const int k = threadIdx.x & 7, part = threadIdx.x >> 3;       // 8 lanes per sum slot walk the rows 8 apart
    float s = 0.f;
    for (long long r = part; r < rows; r += 8) s += partials[r * kPpoSums + k];
    s += __shfl_down(s, 32, 64);
    s += __shfl_down(s, 16, 64);
    s += __shfl_down(s, 8, 64);
    if (threadIdx.x < kPpoSums) means[k] = s / (float)B;
}

// The same for the fused forward + loss + backward kernel, whose partial rows come from the policy workgroups
// (-surrogate, KL, entropy) and the value workgroups (value loss) separately: total = the same combination of the
// four means that the per-sample form sums (linear, so equal up to float32 summation order).
// Tried and dropped (r02): no separate launch — the workgroup that draws the last ticket (a device-scope fence + atomicAdd
// after its partial row) sums the rows at the end of the fused kernel.  Correct, but every workgroup's fence writes its
// XCD's L2 back (the tile's 160 KB of activation stores are in flight): learn phase 7.7 -> 12.4 ms per iteration.
// One 256-thread block: thread t = (slot k = t & 7, walker t >> 3 of 32) adds rows walker, walker + 32, ... (eight loads
// in flight), the walkers of a wave are combined by a fixed shuffle tree and the four waves' sums in order: deterministic,
// ~1 us.  Used by the finishing launch below AND by the extra block of mlp_adam_kernel: same bits either way.
__device__ __forceinline__ void ppo_loss_means_block(const float* __restrict__ partials, long long rows, long long B, float* __restrict__ means,
                                                     const float* __restrict__ kl_coeff, const float* __restrict__ ent_coeff, float vf_coeff,
                                                     float (&red)[4][kPpoSums])
{
    const int t = threadIdx.x, k = t & 7, part = t >> 3;
    float s = 0.f;
    long long r = part;
    for (; r + 7 * 32 < rows; r += 8 * 32) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = partials[(r + 32 * j) * kPpoSums + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += x[j];
    }
    for (; r < rows; r += 32) s += partials[r * kPpoSums + k];
    s += __shfl_down(s, 32, 64);
    s += __shfl_down(s, 16, 64);
    s += __shfl_down(s, 8, 64);
    if ((t & 63) < kPpoSums) red[t >> 6][k] = s;
    __syncthreads();
    if (t < 64) {
        float v = 0.f;
        if (t < kPpoSums) v = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
        const float mean = v / (float)B;
        const float m0 = __shfl(mean, 0, 64), m1 = __shfl(mean, 1, 64), m2 = __shfl(mean, 2, 64), m3 = __shfl(mean, 3, 64);
        if (t < kPpoSums) means[t] = t == 4 ? m0 + *kl_coeff * m2 + vf_coeff * m1 - *ent_coeff * m3 : mean;
    }
}

__global__ __launch_bounds__(256) void ppo_loss_finish_split_kernel(const float* __restrict__ partials, long long rows, long long B,
                                                                    float* __restrict__ means, float* __restrict__ step_counter,
                                                                    const float* __restrict__ kl_coeff, const float* __restrict__ ent_coeff,
                                                                    float vf_coeff)
{
    __shared__ float red[4][kPpoSums];
    if (step_counter && threadIdx.x == 0) *step_counter += 1.0f;
    ppo_loss_means_block(partials, rows, B, means, kl_coeff, ent_coeff, vf_coeff, red);
}

// ---------------------------------------------------------------------------------------------------------------
// After the sampler's T steps: log-probabilities of the taken actions and GAE(lambda) advantages, one thread per env
// walking its T records backwards (the scan is serial in t, parallel in n; all accesses coalesced over n).  Replaces
// ~7 element-wise launches per step of the host loop with one launch per rollout.  Every product and sum is a
// separate rounded float32 operation in the order of the host formula (ppo.py compute_gae / gaussian_logp), so the
// advantages are bit-identical to it; RLlib of the reference's era (0.8.x postprocessing.compute_advantages) treats the
// TimeLimit cut as terminal, hence terminal = done | truncated.
// ---------------------------------------------------------------------------------------------------------------
struct GaeParams {
    const float* reward;       // [T][N]
    const float* values;       // [T][N]
    const float* last_value;   // [N] value of the state after the last step
    const unsigned char* done; // [T][N]
    const unsigned char* trunc;// [T][N] or null
    const float* actions;      // [T][N][6] or null (then no log-probs)
    const float* mean;         // [T][N][6]
    const float* log_std;      // [T][N][6]
    float* logp;               // [T][N]
    float* adv;                // [T][N]
    float* vtarg;              // [T][N]
    float* terminals;          // [T][N] 1.0 / 0.0, or null
    // the rollout's bookkeeping sums, optional (partials == null: none): the episode statistics of the reference's result
    // columns (cli.py:32-38) — per-env running return / length carried across rollouts in ep_ret / ep_len — and the
    // advantages' sum and sum of squares (PPO standardises them over the batch)
    float* ep_ret;             // [N] in / out
    float* ep_len;             // [N] in / out
    double* partials;          // [gridDim.x][8]: adv sum, adv sum of squares, episodes ended, their return sum, length sum,
                               //                 max return, min return, 0
    long long N;
    int T;
    float gamma, gamma_lam;
};

// grid (ceil(N / 64), 1 + T): row 0 of the grid is the per-env scan (one wave per block: 16 384 envs are 256 blocks =
// every CU; eight steps' loads in flight), rows 1 .. T do the log-probabilities and terminal flags of step t = y - 1, which
// need no scan (as part of the scan loop their 18 loads and 6 exp per step sat on its critical path: 49 us per rollout)
__global__ __launch_bounds__(64) void gae_logp_kernel(const GaeParams P)
{
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = n < P.N;
    if (blockIdx.y > 0) {
        if (!in) return;
        const long long i = (long long)(blockIdx.y - 1) * P.N + n;
        if (P.terminals) P.terminals[i] = (P.done[i] | (P.trunc ? P.trunc[i] : (unsigned char)0)) ? 1.f : 0.f;
        if (P.actions) {
            const float* a = P.actions + i * 6;
            const float* m = P.mean + i * 6;
            const float* l = P.log_std + i * 6;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float z = __fmul_rn(__fsub_rn(a[j], m[j]), expf(-l[j]));
                // (-0.5 z) z - log_std - 0.5 log(2 pi), summed over j in order
                const float term_j = __fsub_rn(__fsub_rn(__fmul_rn(__fmul_rn(-0.5f, z), z), l[j]), 0.91893853320467274178f);
                s = __fadd_rn(s, term_j);
            }
            P.logp[i] = s;
        }
        return;
    }
    double a_sum = 0.0, a_sq = 0.0, e_cnt = 0.0, e_ret = 0.0, e_len = 0.0;
    float e_max = -__builtin_inff(), e_min = __builtin_inff();
    if (in) {
        float nxt_v = P.last_value[n], nxt_a = 0.f;
        const float gl = P.gamma_lam;
        constexpr int U = 8;
        for (int t1 = P.T; t1 > 0; t1 -= U) {                           // steps t1 - 1 down to max(t1 - U, 0)
            float rw[U], vv[U]; bool tm[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t1 - 1 - u;
                const long long i = (long long)(t < 0 ? 0 : t) * P.N + n;
                rw[u] = P.reward[i]; vv[u] = P.values[i];
                tm[u] = P.done[i] | (P.trunc ? P.trunc[i] : (unsigned char)0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t1 - 1 - u;
                if (t < 0) break;
                const long long i = (long long)t * P.N + n;
                const float live = tm[u] ? 0.f : 1.f;
                const float v = vv[u];
                // delta = r + gamma * nxt_v * live - v ; adv = delta + gamma * lam * live * nxt_a
                const float delta = __fsub_rn(__fadd_rn(rw[u], __fmul_rn(__fmul_rn(P.gamma, nxt_v), live)), v);
                nxt_a = __fadd_rn(delta, __fmul_rn(__fmul_rn(gl, live), nxt_a));
                nxt_v = v;
                P.adv[i] = nxt_a;
                P.vtarg[i] = __fadd_rn(nxt_a, v);
                a_sum += (double)nxt_a; a_sq += (double)nxt_a * (double)nxt_a;
            }
        }
        if (P.partials) {
            // episode bookkeeping walks forward: EpisodeStats.step()'s arithmetic (return and length accumulate in float32)
            float ret = P.ep_ret[n], len = P.ep_len[n];
            for (int t0 = 0; t0 < P.T; t0 += U) {
                float rw[U]; bool tm[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + u;
                    const long long i = (long long)(t < P.T ? t : P.T - 1) * P.N + n;
                    rw[u] = P.reward[i];
                    tm[u] = P.done[i] | (P.trunc ? P.trunc[i] : (unsigned char)0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (t0 + u >= P.T) break;
                    ret = __fadd_rn(ret, rw[u]);
                    len += 1.0f;
                    if (tm[u]) {
                        e_cnt += 1.0; e_ret += (double)ret; e_len += (double)len;
                        e_max = fmaxf(e_max, ret); e_min = fminf(e_min, ret);
                        ret = 0.f; len = 0.f;
                    }
                }
            }
            P.ep_ret[n] = ret; P.ep_len[n] = len;
        }
    }
    if (!P.partials) return;
    double v[7] = {a_sum, a_sq, e_cnt, e_ret, e_len, (double)e_max, (double)e_min};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_down(v[k], off, 64);
            v[k] = k == 5 ? fmax(v[k], o) : (k == 6 ? fmin(v[k], o) : v[k] + o);
        }
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) P.partials[(size_t)blockIdx.x * 8 + k] = v[k];
        P.partials[(size_t)blockIdx.x * 8 + 7] = 0.0;
    }
}

// the partial rows in block order -> the window accumulators of EpisodeStats (in place) and the advantages' moments
__global__ __launch_bounds__(64) void gae_finish_kernel(const double* __restrict__ partials, long long blocks, long long count,
                                                       double* __restrict__ w_sum, double* __restrict__ w_len, double* __restrict__ w_cnt,
                                                       float* __restrict__ w_max, float* __restrict__ w_min, double* __restrict__ adv_stats)
{
    // eight walkers per value (rows 8 apart), combined in a fixed order
    const int k = threadIdx.x & 7, part = threadIdx.x >> 3;
    const bool mx = k == 5, mn = k == 6;
    double x = mx ? -__builtin_inf() : (mn ? __builtin_inf() : 0.0);
    for (long long b = part; b < blocks; b += 8) {
        const double p = partials[(size_t)b * 8 + k];
        x = mx ? fmax(x, p) : (mn ? fmin(x, p) : x + p);
    }
#pragma unroll
    for (int off = 32; off >= 8; off >>= 1) {
        const double o = __shfl_down(x, off, 64);
        x = mx ? fmax(x, o) : (mn ? fmin(x, o) : x + o);
    }
    if (threadIdx.x >= 7) return;
    if (k == 0) { adv_stats[0] = x; adv_stats[2] = (double)count; }
    else if (k == 1) adv_stats[1] = x;
    else if (k == 2) *w_cnt += x;
    else if (k == 3) *w_sum += x;
    else if (k == 4) *w_len += x;
    else if (k == 5) *w_max = fmaxf(*w_max, (float)x);
    else *w_min = fminf(*w_min, (float)x);
}

// ---------------------------------------------------------------------------------------------------------------
// A pseudo-random permutation of 0 .. n-1 without a sort: a 6-round Feistel network over the next even-bit-width
// power of two, keyed by (seed, stream), with cycle walking (re-encrypt until the value is below n; fewer than four
// rounds expected, since the domain is < 4n).  A bijection of [0, 2^k) restricted by cycle walking is a bijection of
// [0, n): every index appears exactly once, which is all minibatch shuffling needs (RLlib's sgd.py shuffles with
// np.random.permutation; the draws differ, the property "each sample once per epoch" is the same).  One launch of n
// threads instead of torch.randperm's radix sort + merge passes (~170 us at n = 524 288).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int perm_round(unsigned int x, unsigned int key)
{
    x = (x ^ key) * 0x9E3779B1u;
    x ^= x >> 15; x *= 0x85EBCA77u;
    x ^= x >> 13; x *= 0xC2B2AE3Du;
    x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void permutation_kernel(long long* __restrict__ out, long long n, int half_bits,
                                                         unsigned long long seed, unsigned long long stream)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int mask = (1u << half_bits) - 1u;
    unsigned int keys[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {                         // splitmix64 of (seed, stream, round)
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (stream * 6ull + (unsigned long long)r + 1ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        keys[r] = (unsigned int)(z ^ (z >> 31));
    }
    unsigned long long x = (unsigned long long)i;
    do {                                                  // terminates: the walk of a bijection returns to i < n at the latest
        unsigned int L = (unsigned int)(x >> half_bits) & mask, R = (unsigned int)x & mask;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const unsigned int f = perm_round(R, keys[r]) & mask;
            const unsigned int nl = R;
            R = L ^ f;
            L = nl;
        }
        x = ((unsigned long long)L << half_bits) | R;
    } while ((long long)x >= n);
    out[i] = (long long)x;
}

// ---------------------------------------------------------------------------------------------------------------
// MeanStdFilter's moment pass over a rollout's observations (the 'observation_filter' of pioneer_knm_train.py:66):
// column sums of d = x - pivot and of d^2 over rows x [rows][137], ONE read of the buffer (287 MB per 524 288 rows)
// instead of the four passes of the element-wise formulation (subtract, sum, square, sum).  Stage 1: a block walks
// kFmRows consecutive rows, thread = column (a row is 548 contiguous bytes), float32 partial sums (at most 512 terms
// each); stage 2: one block adds the partials in block order in float64 into the running accumulators.  A constant
// column has d == 0 exactly, so both of its sums stay exactly 0 (what the shifted-sum filter is built on).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFmCols = 137;
constexpr int kFmThreads = 160;
constexpr int kFmRows = 512;

__global__ __launch_bounds__(kFmThreads) void filter_moments_kernel(const float* __restrict__ x, long long rows,
                                                                   const float* __restrict__ pivot, float* __restrict__ partials)
{
    const int c = threadIdx.x;
    if (c >= kFmCols) return;
    const long long r0 = (long long)blockIdx.x * kFmRows;
    long long r1 = r0 + kFmRows;
    if (r1 > rows) r1 = rows;
    const float p = pivot[c];
    float s = 0.f, q = 0.f;
    const float* xp = x + r0 * kFmCols + c;
    long long r = r0;
    for (; r + 8 <= r1; r += 8) {                          // eight rows in flight
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = xp[(long long)j * kFmCols];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = __fsub_rn(v[j], p); s = __fadd_rn(s, d); q = __fadd_rn(q, __fmul_rn(d, d)); }
        xp += 8 * kFmCols;
    }
    for (; r < r1; ++r) { const float d = __fsub_rn(*xp, p); s = __fadd_rn(s, d); q = __fadd_rn(q, __fmul_rn(d, d)); xp += kFmCols; }
    partials[((size_t)blockIdx.x * 2 + 0) * kFmCols + c] = s;
    partials[((size_t)blockIdx.x * 2 + 1) * kFmCols + c] = q;
}

// grid (137 columns, 2 moments), 256 threads: thread t adds the partials of blocks t, t + 256, ... in float64, then the 256
// sums are added pairwise in a fixed tree: deterministic, and ~10 us where one thread per column walking all 1 024
// partial rows took 250
__global__ __launch_bounds__(256) void filter_moments_finish_kernel(const float* __restrict__ partials, long long blocks, long long rows,
                                                                   double* __restrict__ dsum, double* __restrict__ dsq, double* __restrict__ dn)
{
    __shared__ double red[256];
    const int c = blockIdx.x, m = blockIdx.y, t = threadIdx.x;
    double s = 0.0;
    for (long long b = t; b < blocks; b += 256) s += (double)partials[((size_t)b * 2 + m) * kFmCols + c];
    red[t] = s;
    __syncthreads();
#pragma unroll
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) {
        (m == 0 ? dsum : dsq)[c] += red[0];
        if (c == 0 && m == 0) *dn += (double)rows;
    }
}

// MeanStdFilter.sync() on ONE rank: the pending delta (dn, shifted sums about the pivot) merged into the running (n, mean,
// m2) by Chan's update, column by column in float64 — the operations of the host formulation (ppo.py) one by one, so the
// statistics are bit-identical to it; ~25 element-wise launches on 137-vectors become one.  (Several ranks: the host
// formulation stays, its two all-reduces sit between the same steps.)
__global__ __launch_bounds__(kFmThreads) void filter_merge_kernel(double* __restrict__ dn_p, double* __restrict__ dsum, double* __restrict__ dsq,
                                                                 const float* __restrict__ pivot, double* __restrict__ n_p,
                                                                 double* __restrict__ mean, double* __restrict__ m2)
{
    const int c = threadIdx.x;
    const double dn_r = *dn_p, n = *n_p;
    __syncthreads();                                   // every thread has read the scalars before thread 0 rewrites them
    if (c < kFmCols) {
        const double dn_r_safe = fmax(dn_r, 1.0);
        const double mean_r = (double)pivot[c] + dsum[c] / dn_r_safe;
        const double m2_r = fmax(dsq[c] - dsum[c] * dsum[c] / dn_r_safe, 0.0);
        const double dn = dn_r, dn_safe = fmax(dn, 1.0);
        const double bmean = (dn_r * mean_r) / dn_safe;
        const double dev = mean_r - bmean;
        const double bm2 = m2_r + dn_r * dev * dev;
        const double tot = n + dn, tot_safe = fmax(tot, 1.0);
        const double delta = bmean - mean[c];
        m2[c] = m2[c] + (bm2 + delta * delta * (n * dn / tot_safe));
        mean[c] = mean[c] + delta * (dn / tot_safe);
        dsum[c] = 0.0; dsq[c] = 0.0;
    }
    if (c == 0) { *n_p = n + dn_r; *dn_p = 0.0; }
}

// MeanStdFilter.prepare(): the vectors the kernels filter with — x' = clamp((x - loc) * inv, lo, hi) — from the running statistics:
// loc = mean, inv = 1 / (std + 1e-8) with std = sqrt(max(m2 / max(n - 1, 1), 0)), hi = clip (or +inf), lo = -hi; the identity
// (0, 1, -inf, +inf) until two samples exist.  float64 like the torch formulation it replaces (17 element-wise launches), rounded to
// float32 once: the same bits.
__global__ __launch_bounds__(kFmThreads) void filter_prepare_kernel(const double* __restrict__ n_p, const double* __restrict__ mean,
                                                                   const double* __restrict__ m2, double clip, float* __restrict__ loc,
                                                                   float* __restrict__ inv, float* __restrict__ lo, float* __restrict__ hi)
{
    const int c = threadIdx.x;
    if (c >= kFmCols) return;
    const double n = *n_p;
    const bool ident = n < 2.0;
    const double var = m2[c] / fmax(n - 1.0, 1.0);
    const double sd = sqrt(fmax(var, 0.0));
    const float h = ident ? __builtin_inff() : (float)clip;
    loc[c] = ident ? 0.f : (float)mean[c];
    inv[c] = ident ? 1.f : (float)(1.0 / (sd + 1e-8));
    hi[c] = h;
    lo[c] = -h;
}

}  // namespace pnr
