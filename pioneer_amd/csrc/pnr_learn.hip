// pnr_learn.hip — the PPO host driver's kernels (pnr_mlp.h: both MLPs as bf16 MFMA kernels, weight gradients, Adam;
// pnr_ppo.h: loss, GAE, filter moments, shuffle) and their C ABI (include/pioneer_amd.h, section "PPO driver").  A
// translation unit of its own: it shares nothing with the env engine but the error plumbing, and the two compile in parallel.
// gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/pioneer_amd.h"
#include "pnr_host.h"
#include "pnr_device.h"
#include "pnr_ppo.h"
#include "pnr_mlp.h"
#include "pnr_sampler.h"

using namespace pnr;

// handle-less calls: the message goes to the calling thread's buffer (pnr_last_error(NULL))
static int fail(pnr_handle, int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    const int rc = pnr_failv(nullptr, code, fmt, ap);
    va_end(ap);
    return rc;
}

static const char g_unit_fp[] = "pnr_build_fp:learn=" PNR_UNIT_FINGERPRINT ";";
extern "C" const char* pnr_unit_fingerprint_learn(void) { return g_unit_fp; }

#if PNR_MLP_STAMPS
// diagnostic variant only (tools/mlp_stamps.py): where the fused kernel's waves write their phase stamps
static unsigned long long* g_mlp_stamps = nullptr;
extern "C" int pnr_mlp_set_stamp_buffer(void* p) { g_mlp_stamps = static_cast<unsigned long long*>(p); return 0; }
static unsigned long long* g_wg_stamps = nullptr;      // the weight-gradient kernel's: [2 nets][4 roles][slices][8 waves][26]
extern "C" int pnr_mlp_set_wgrad_stamp_buffer(void* p) { g_wg_stamps = static_cast<unsigned long long*>(p); return 0; }
#endif

extern "C" {

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

int64_t pnr_mlp_w3_partial_floats(int64_t batch) { return batch < 1 ? 0 : ((batch + kMlpBM - 1) / kMlpBM) * kMlpNets * (int64_t)kW3PartFloats; }

int64_t pnr_mlp_slab_floats(int64_t batch)
{
    if (batch < 1) return 0;
    long long slices, rows;
    mlp_slicing(batch, &slices, &rows);
    return (int64_t)(slices * kMlpNets * kGradElems);
}

int64_t pnr_mlp_pack_elems(void) { return (int64_t)kMlpNets * kPackElems; }
int64_t pnr_mlp_bias_elems(void) { return (int64_t)kMlpNets * kBiasElems; }

static inline bool planes_ok(int32_t planes) { return planes >= 1 && planes <= kMlpMaxPlanes; }

int pnr_mlp_pack(const float* const* params, int32_t n3_policy, int32_t n3_value, void* wpack, float* bias, int32_t planes, void* stream)
{
    if (!params || !wpack || !bias) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_pack: null argument");
    if (!planes_ok(planes)) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_pack: planes must be 1, 2 or 3 (got %d)", planes);
    if (n3_policy < 1 || n3_policy > kMlpHead || n3_value < 1 || n3_value > kMlpHead)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_pack: head widths must be in 1..16");
    MlpPackParams P;
    for (int n = 0; n < kMlpNets; ++n) {
        for (int k = 0; k < 6; ++k)
            if (!params[6 * n + k]) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_pack: null parameter %d of net %d", k, n);
        P.net[n] = {params[6 * n + 0], params[6 * n + 1], params[6 * n + 2], params[6 * n + 3], params[6 * n + 4], params[6 * n + 5],
                    n == 0 ? n3_policy : n3_value};
    }
    P.wpack = static_cast<__bf16*>(wpack); P.bias = bias; P.planes = planes;
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((kPackElems + kBiasElems + 255) / 256, kMlpNets), dim3(256), 0, (hipStream_t)stream, P);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_mlp_forward(int64_t batch, const float* obs, const int64_t* idx, const float* f_loc, const float* f_inv,
                    const float* f_lo, const float* f_hi, const void* wpack, const float* bias, float* head,
                    void* xs, void* h1, void* h2, int32_t first_net, int32_t n_nets, int32_t planes, void* stream)
{
    if (batch < 1 || !obs || !wpack || !bias || !head) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_forward: null argument or empty batch");
    if (!planes_ok(planes)) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_forward: planes must be 1, 2 or 3 (got %d)", planes);
    if (planes > 1 && (xs || h1 || h2))
        return fail(nullptr, PNR_ERR_UNSUPPORTED, "pnr_mlp_forward: split operands (planes %d) save no activations: pnr_mlp_backward is bf16-only", planes);
    if (first_net < 0 || n_nets < 1 || first_net + n_nets > kMlpNets) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_forward: bad net range");
    if ((f_loc || f_inv || f_lo || f_hi) && !(f_loc && f_inv && f_lo && f_hi))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_forward: the four filter vectors come together or not at all");
    if ((h1 == nullptr) != (h2 == nullptr)) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_forward: h1 and h2 come together");
    MlpFwdParams P = {};
    P.gscale = 1.f;
    P.obs = obs; P.idx = reinterpret_cast<const long long*>(idx); P.f_loc = f_loc; P.f_inv = f_inv; P.f_lo = f_lo; P.f_hi = f_hi;
    P.wpack = static_cast<const __bf16*>(wpack); P.bias = bias; P.head = head;
    P.xs = static_cast<__bf16*>(xs); P.h1 = static_cast<__bf16*>(h1); P.h2 = static_cast<__bf16*>(h2);
    P.B = batch; P.first_net = first_net; P.n_nets = n_nets;
    const dim3 grid((unsigned)((batch + kMlpBM - 1) / kMlpBM), n_nets);
    if (planes == 1) hipLaunchKernelGGL((mlp_forward_kernel<false, 1>), grid, dim3(kFwdThreads), 0, (hipStream_t)stream, P);
    else if (planes == 2) hipLaunchKernelGGL((mlp_forward_kernel<false, 2>), grid, dim3(kFwdThreads), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL((mlp_forward_kernel<false, 3>), grid, dim3(kFwdThreads), 0, (hipStream_t)stream, P);
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

int pnr_filter_prepare(const double* n, const double* mean, const double* m2, double clip, float* loc, float* inv, float* lo, float* hi,
                       void* stream)
{
    if (!n || !mean || !m2 || !loc || !inv || !lo || !hi) return fail(nullptr, PNR_ERR_INVALID, "pnr_filter_prepare: null argument");
    hipLaunchKernelGGL(filter_prepare_kernel, dim3(1), dim3(kFmThreads), 0, (hipStream_t)stream, n, mean, m2, clip, loc, inv, lo, hi);
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
                float* log_std, float* values, float* actions, float* env_actions, void* xs_out, int32_t planes, void* stream)
{
    if (batch < 1 || !obs || !wpack || !bias || !noise || !mean || !log_std || !values || !actions)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_act: null argument or empty batch");
    if (!planes_ok(planes)) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_act: planes must be 1, 2 or 3 (got %d)", planes);
    if (planes > 1 && xs_out)
        return fail(nullptr, PNR_ERR_UNSUPPORTED, "pnr_mlp_act: split operands (planes %d): the learner gathers its inputs from the observations, xs_out must be NULL", planes);
    if ((f_loc || f_inv || f_lo || f_hi) && !(f_loc && f_inv && f_lo && f_hi))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_act: the four filter vectors come together or not at all");
    if (a_max && (!env_actions || env_actions == actions))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_act: clipping (a_max) needs its own env_actions buffer");
    MlpFwdParams P = {};
    P.gscale = 1.f;
    P.obs = obs; P.f_loc = f_loc; P.f_inv = f_inv; P.f_lo = f_lo; P.f_hi = f_hi;
    P.wpack = static_cast<const __bf16*>(wpack); P.bias = bias; P.head = head;
    P.B = batch; P.first_net = 0; P.n_nets = kMlpNets;
    P.noise = noise; P.a_max = a_max; P.mean = mean; P.log_std = log_std; P.values = values; P.actions = actions;
    P.env_actions = a_max ? env_actions : actions;
    P.xs = static_cast<__bf16*>(xs_out);                     // the nets' input as they saw it, for the learner (or null)
    const dim3 grid((unsigned)((batch + kMlpBM - 1) / kMlpBM), kMlpNets);
    if (planes == 1) hipLaunchKernelGGL((mlp_forward_kernel<false, 1>), grid, dim3(kFwdThreads), 0, (hipStream_t)stream, P);
    else if (planes == 2) hipLaunchKernelGGL((mlp_forward_kernel<false, 2>), grid, dim3(kFwdThreads), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL((mlp_forward_kernel<false, 3>), grid, dim3(kFwdThreads), 0, (hipStream_t)stream, P);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_ppo_rollout(pnr_handle h, int32_t T, const float* f_loc, const float* f_inv, const float* f_lo, const float* f_hi,
                    const void* wpack, const float* bias, const float* noise, const float* a_max, float* obs, float* mean,
                    float* log_std, float* values, float* actions, void* xs_out, float* reward, uint8_t* done, uint8_t* truncated,
                    void* stream)
{
    RolloutParams S = {};
    int device = 0;
    int rc = pnr_env_rollout_params(h, &S.K, &S.max_v_to_r, &device);
    if (rc) return rc;
    const auto hfail = [&](const char* msg) { return fail(nullptr, PNR_ERR_INVALID, "%s", msg); };
    if (T < 1) return hfail("pnr_ppo_rollout: T must be >= 1");
    if (!wpack || !bias || !noise || !obs || !mean || !log_std || !values || !actions || !reward || !done)
        return hfail("pnr_ppo_rollout: null argument");
    if ((f_loc || f_inv || f_lo || f_hi) && !(f_loc && f_inv && f_lo && f_hi))
        return hfail("pnr_ppo_rollout: the four filter vectors come together or not at all");
    if (reinterpret_cast<uintptr_t>(mean) & 7u || reinterpret_cast<uintptr_t>(log_std) & 7u || reinterpret_cast<uintptr_t>(actions) & 7u)
        return hfail("pnr_ppo_rollout: mean, log_std and actions must be 8-byte aligned");
    S.T = T;
    S.obs0 = obs;
    S.K.obs = obs + (size_t)S.K.n * kObsDim;                 // slot 1: where step 0 leaves its observation
    S.K.reward = reward; S.K.done = done; S.K.trunc = truncated; S.K.info = nullptr; S.K.actions = nullptr; S.K.T = T;
    S.f_loc = f_loc; S.f_inv = f_inv; S.f_lo = f_lo; S.f_hi = f_hi;
    S.wpack = static_cast<const __bf16*>(wpack); S.bias = bias; S.noise = noise; S.a_max = a_max;
    S.mean = mean; S.log_std = log_std; S.values = values; S.actions = actions; S.xs = static_cast<__bf16*>(xs_out);
    DeviceGuard g(device);
    hipLaunchKernelGGL(ppo_rollout_kernel, dim3((unsigned)((S.K.n + kMlpBM - 1) / kMlpBM)), dim3(kFwdThreads), 0, (hipStream_t)stream, S);
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
    MlpWgradParams Wp = {};
    Wp.gscale = 1.f;
    Wp.g_head = g_head; Wp.xs = static_cast<const __bf16*>(xs); Wp.h1 = Bp.h1; Wp.h2 = Bp.h2; Wp.dz1 = Bp.dz1; Wp.dz2 = Bp.dz2;
    Wp.slabs = slabs; Wp.B = batch; Wp.slice_rows = rows; Wp.first_net = 0; Wp.w3part = nullptr; Wp.n_nets = kMlpNets; Wp.stamps = nullptr;
    Wp.act_plane = 0; Wp.xs_plane = 0;
    hipLaunchKernelGGL(mlp_wgrad_kernel<1>, dim3((unsigned)slices, kWgParts, kMlpNets), dim3(kWgThreads), 0, st, Wp);
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
    if (s->first_net < 0 || s->n_nets < 0 || s->first_net + (s->n_nets ? s->n_nets : kMlpNets) > kMlpNets)
        return fail(nullptr, PNR_ERR_INVALID, "%s: bad net range (first_net %d, n_nets %d)", who, s->first_net, s->n_nets);
    if (s->planes < 0 || s->planes > kMlpMaxPlanes) return fail(nullptr, PNR_ERR_INVALID, "%s: planes must be 0 .. 3 (got %d)", who, s->planes);
    if ((reinterpret_cast<uintptr_t>(s->wpack) | reinterpret_cast<uintptr_t>(s->bias) | reinterpret_cast<uintptr_t>(s->adam_m) | reinterpret_cast<uintptr_t>(s->adam_v)) & 15)
        return fail(nullptr, PNR_ERR_INVALID, "%s: wpack, bias, adam_m and adam_v must be 16-byte aligned (mlp_adam_kernel moves them four floats at a time)", who);
    return PNR_OK;
}

// loss_rows > 0: the launch also sums the update's loss means (rows of the fused kernel in s->partials)
static inline int step_nets(const pnr_mlp_step* s) { return s->n_nets ? s->n_nets : kMlpNets; }

static void mlp_launch_adam(const pnr_mlp_step* s, const float* grad, int slices, float scale, hipStream_t st, long long loss_rows = 0)
{
    MlpAdamParams A;
    A.first_net = s->first_net;
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
    A.wpack = static_cast<__bf16*>(s->wpack); A.bias = s->bias; A.planes = s->planes > 1 ? s->planes : 1;
    hipLaunchKernelGGL(mlp_adam_kernel, dim3(kAdamBlocks + 1, step_nets(s)), dim3(256), 0, st, A);
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
                   float* value_old_out, const float* record_rows, const void* xs_rows, int32_t planes, void* stream)
{
    if (!planes_ok(planes)) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_gather: planes must be 1, 2 or 3 (got %d)", planes);
    if (planes > 1 && xs_rows)
        return fail(nullptr, PNR_ERR_UNSUPPORTED, "pnr_mlp_gather: split operands (planes %d) are made from the float32 observations: xs_rows must be NULL", planes);
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
    G.B = batch; G.planes = planes;
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
        !s->value_old || !s->kl_coeff || !s->entropy_coeff || !s->head || (!s->g_head && !s->w3_partials) || !s->xs || !s->h1 || !s->h2 ||
        !s->dz1 || !s->dz2 || !s->partials || !s->slabs || !s->means)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: null argument or empty batch");
    if ((s->f_loc || s->f_inv || s->f_lo || s->f_hi) && !(s->f_loc && s->f_inv && s->f_lo && s->f_hi))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: the four filter vectors come together or not at all");
    if (reinterpret_cast<uintptr_t>(s->slabs) & 15) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: slabs must be 16-byte aligned");
    long long slices, rows;
    mlp_slicing(B, &slices, &rows);
    if (s->slab_floats < slices * kMlpNets * kGradElems)
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: slabs hold %lld floats, the launch needs %lld",
                    (long long)s->slab_floats, slices * kMlpNets * kGradElems);
    hipStream_t st = (hipStream_t)stream;
    const int nets = step_nets(s);
    const int planes = s->planes > 1 ? s->planes : 1;
    if (planes > 1 && (!s->w3_partials || !s->xs_in))
        return fail(nullptr, PNR_ERR_UNSUPPORTED, "pnr_mlp_train_step: split operands (planes %d) need w3_partials and the pre-gathered input planes (xs_in, pnr_mlp_gather)", planes);
    if (planes > kMlpMaxPlanes) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: planes %d (1, 2 or 3)", planes);
    // the kernels read the pre-gathered rows as 16-byte vectors at xs_in + plane * xs_in_plane: checked before anything is launched
    if (s->xs_in && (reinterpret_cast<uintptr_t>(s->xs_in) & 15))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: xs_in must be 16-byte aligned");
    if (planes > 1 && s->xs_in_plane != 0 && (s->xs_in_plane % 8 != 0 || s->xs_in_plane < B * kMlpInPad))
        return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: xs_in_plane %lld: a multiple of 8 elements, at least batch * 144 = %lld (or 0: exactly that)",
                    (long long)s->xs_in_plane, B * kMlpInPad);
    const dim3 tiles((unsigned)((B + kMlpBM - 1) / kMlpBM), nets);
    const long long prow = (long long)tiles.x * nets;           // one row of loss sums per workgroup of the fused kernel
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
    F.B = B; F.first_net = s->first_net; F.n_nets = nets;
    F.rec_actions = s->actions; F.rec_logp = s->logp_old; F.rec_mean = s->mean_old; F.rec_log_std = s->log_std_old;
    F.rec_adv = s->adv; F.rec_vtarg = s->value_target; F.rec_values = s->value_old;
    F.kl_coeff = s->kl_coeff; F.ent_coeff = s->entropy_coeff;
    F.clip = s->clip_param; F.vf_clip = s->vf_clip_param; F.vf_coeff = s->vf_loss_coeff;
    F.g_head = s->g_head; F.partials = s->partials; F.adam_step = s->adam_step;
    F.dz1 = static_cast<__bf16*>(s->dz1); F.dz2 = static_cast<__bf16*>(s->dz2);
    F.act_plane = (size_t)kMlpNets * (size_t)B * kMlpHid;            // planes of h1 / dz1 / dz2: [planes][2][B][256]
    F.xs_plane = s->xs_in_plane > 0 ? (size_t)s->xs_in_plane : (size_t)B * kMlpInPad;
    F.gscale = mlp_grad_scale(planes, B);
    // layer 3's weight gradients per tile from the fused kernel (then H2 never leaves the CU) when the caller gave the scratch for it
    if (s->w3_partials) {
        if (s->w3_partial_floats < (long long)tiles.x * nets * kW3PartFloats)
            return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_train_step: w3_partials hold %lld floats, the launch needs %lld",
                        (long long)s->w3_partial_floats, (long long)tiles.x * nets * kW3PartFloats);
        F.w3part = s->w3_partials;
        F.h2 = nullptr;
        F.g_head = nullptr;          // its only reader outside the tile was the weight-gradient kernel's layer-3 role: 4 MB less per update
    }
#if PNR_MLP_STAMPS
    F.stamps = g_mlp_stamps;
#endif
    if (planes == 1) hipLaunchKernelGGL((mlp_forward_kernel<true, 1>), tiles, dim3(kFwdThreads), 0, st, F);
    else if (planes == 2) hipLaunchKernelGGL((mlp_forward_kernel<true, 2>), tiles, dim3(kFwdThreads), 0, st, F);
    else hipLaunchKernelGGL((mlp_forward_kernel<true, 3>), tiles, dim3(kFwdThreads), 0, st, F);
    if (s->flat_grad)       // no Adam launch here (the caller all-reduces first): the loss means get a small launch of their own
        hipLaunchKernelGGL(ppo_loss_finish_split_kernel, dim3(1), dim3(256), 0, st, s->partials, prow, B, s->means, (float*)nullptr,
                           s->kl_coeff, s->entropy_coeff, s->vf_loss_coeff);
    MlpWgradParams Wp = {};
    Wp.gscale = 1.f;
    Wp.g_head = s->g_head; Wp.xs = F.xs_in ? F.xs_in : F.xs; Wp.h1 = F.h1; Wp.h2 = F.h2; Wp.dz1 = F.dz1; Wp.dz2 = F.dz2;
    Wp.slabs = s->slabs; Wp.B = B; Wp.slice_rows = rows; Wp.first_net = s->first_net; Wp.w3part = F.w3part; Wp.n_nets = nets;
    Wp.stamps = nullptr;
#if PNR_MLP_STAMPS
    Wp.stamps = g_wg_stamps;
#endif
    Wp.act_plane = F.act_plane; Wp.xs_plane = F.xs_plane; Wp.gscale = F.gscale;
    const dim3 wgrid((unsigned)slices, kWgParts, nets);
    if (planes == 1) hipLaunchKernelGGL(mlp_wgrad_kernel<1>, wgrid, dim3(kWgThreads), 0, st, Wp);
    else if (planes == 2) hipLaunchKernelGGL(mlp_wgrad_kernel<2>, wgrid, dim3(kWgThreads), 0, st, Wp);
    else hipLaunchKernelGGL(mlp_wgrad_kernel<3>, wgrid, dim3(kWgThreads), 0, st, Wp);
    if (s->flat_grad)
        hipLaunchKernelGGL(mlp_reduce_flat_kernel, dim3((nets * kGradElems + 255) / 256), dim3(256), 0, st, s->slabs, (int)slices, s->flat_grad,
                           s->first_net * kGradElems, nets * kGradElems);
    else
        mlp_launch_adam(s, s->slabs, (int)slices, 1.0f, st, prow);      // + the loss means, in the same launch
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

int pnr_mlp_adam(const pnr_mlp_step* s, const float* flat_grad, float grad_scale, void* stream)
{
    int rc = mlp_step_check(s, "pnr_mlp_adam");
    if (rc) return rc;
    if (!flat_grad || (reinterpret_cast<uintptr_t>(flat_grad) & 15)) return fail(nullptr, PNR_ERR_INVALID, "pnr_mlp_adam: null or not 16-byte aligned gradient");
    mlp_launch_adam(s, flat_grad, 1, grad_scale, (hipStream_t)stream);
    HIP_TRY(nullptr, hipGetLastError());
    return PNR_OK;
}

}  // extern "C"
