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

// Ablation switches (timing-only builds whose OUTPUTS ARE WRONG: stores, the integrator or the obs flush gated off; grid-cap
// overrides) exist only in a variant library built with -DPNR_DIAG_BUILD=1 (`_lib.build_library(extra_flags=["-DPNR_DIAG_BUILD=1"],
// out_path=...)`), which reads them from the environment at pnr_create.  The product library ignores the environment
// (tests/test_abi.py checks that it holds no such string).
#ifndef PNR_DIAG_BUILD
#define PNR_DIAG_BUILD 0
#endif

#include "pnr_host.h"
#include "pnr_device.h"
#include "pnr_dyn.h"
#include "pnr_env_kernels.h"

// =====================================================================================
// host side
// =====================================================================================
using namespace pnr;

struct pnr_env_s {
    pnr_config cfg;
    pnr_constants k;
    KParams base;        // constants pre-filled; pointers set per call
    DynParams dbase;     // dynamics-mode constants (zeroed in kinematic mode)
    JointMotorTable motors;   // pnr_world_step's per-joint motors (pnr_set_joint_motor); fill_base: every joint on the handle's own law
    long long n;
    unsigned long long env_off;
    int device;
    float4* state;
    float* dyn;          // dynamics-mode planar words [36][n] or null
    SceneBody* scene;    // dynamics-mode static scene bodies [kMaxScene] or null
    int diag;            // -DPNR_DIAG_BUILD=1 variant only: the PNR_DIAG environment variable at create time; else 0
    bool ready;          // a full reset has happened, or the state was set explicitly
    bool kin_set, dyn_set;  // pnr_set_state / pnr_set_dyn_state seen (both needed in dynamics mode)
    char err[512];
};

static thread_local char g_err[512] = "";

// the one place error messages are written (pnr_host.h: shared with pnr_learn.hip, not exported)
int pnr_failv(char* handle_err, int code, const char* fmt, va_list ap)
{
    char* dst = handle_err ? handle_err : g_err;
    vsnprintf(dst, 512, fmt, ap);
    if (handle_err) { strncpy(g_err, handle_err, sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
    return code;
}

static int fail(pnr_handle h, int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    const int rc = pnr_failv(h ? h->err : nullptr, code, fmt, ap);
    va_end(ap);
    return rc;
}


static const char g_unit_fp[] = "pnr_build_fp:api=" PNR_UNIT_FINGERPRINT ";";

extern "C" {

int pnr_abi_version(void) { return PNR_ABI_VERSION; }

const char* pnr_build_fingerprint(void)
{
    static char buf[96] = "";
    if (!buf[0]) snprintf(buf, sizeof(buf), "%s%s", g_unit_fp + 13, pnr_unit_fingerprint_learn() + 13);   // past "pnr_build_fp:"
    return buf;
}

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
        // pnr_world_step: until a joint is commanded it runs the handle's motor on the env's own r, v (teleport: no motor at all)
        for (int i = 0; i < kDof; ++i) {
            JointMotorTable& W = h->motors;
            W.kp[i] = c.teleport ? 0.f : D.kp_eff; W.kd[i] = c.teleport ? 0.f : (float)c.pd_kd; W.cpos[i] = D.c_pos; W.vcap[i] = D.v_cap;
            W.tcap[i] = c.torque_limit > 0 ? (float)c.torque_limit : INFINITY;
            W.r_ref[i] = 0.f; W.v_ref[i] = 0.f; W.from_cmd[i] = 1;
        }
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
    P.done_dist_d = c.done_distance;
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
    // The zero fills above run on the NULL stream; the caller's first pnr_reset may be issued on a non-blocking stream (every
    // torch.cuda.Stream is one), which is not ordered after NULL-stream work, and reset_env reads the episode counters from
    // these planes.  pnr_create is the one call that synchronises (include/pioneer_amd.h, Conventions): when it returns, the
    // planes are zero for every stream.
    e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        if (h->scene) (void)hipFree(h->scene);
        if (h->dyn) (void)hipFree(h->dyn);
        (void)hipFree(h->state); delete h;
        return fail(nullptr, PNR_ERR_HIP, "pnr_create: zero fill of the state planes failed: %s", hipGetErrorString(e));
    }
#if PNR_DIAG_BUILD
    { const char* e_ = getenv("PNR_DIAG"); h->diag = e_ ? atoi(e_) : 0; }
#endif
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
// 10.0 / 8.6 us).  The -DPNR_DIAG_BUILD=1 variant reads overrides from the environment (grid-cap experiments).
static inline unsigned step_grid_for(long long n, int T)
{
#if PNR_DIAG_BUILD
    static int cap = -1, cap_roll = -1;
    if (cap < 0) { const char* e_ = getenv("PNR_GRID_CAP"); cap = e_ ? atoi(e_) : 2048; if (cap < 1) cap = 2048; }
    if (cap_roll < 0) { const char* e_ = getenv("PNR_GRID_CAP_ROLLOUT"); cap_roll = e_ ? atoi(e_) : 1024; if (cap_roll < 1) cap_roll = 1024; }
#else
    constexpr int cap = 2048, cap_roll = 1024;
#endif
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
    const dim3 grid((step_grid_for(h->n, T) + kStepWaves - 1) / kStepWaves), block(kWave * kStepWaves);   // the cap counts WAVES
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
        if (T > 1) hipLaunchKernelGGL((dyn_rollout_kernel<O, A, R, C>), dim3((gridD.x + kDynRolloutWaves - 1) / kDynRolloutWaves), dim3(kWave * kDynRolloutWaves), 0, st, Pt.state, D.dyn, Pt.actions, Pt.n, Pt.dt, \
                                      Pt.eps, (float)h->cfg.max_v_to_r, Pt, D); \
        else hipLaunchKernelGGL((dyn_step_kernel<O, A, R, C>), gridD, dim3(kWave * kDynStepWaves), 0, st, Pt.state, D.dyn, Pt.actions, Pt.n, Pt.dt, \
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

int pnr_set_joint_motor(pnr_handle h, int joint, int control_mode, double target_position, double target_velocity,
                        double position_gain, double velocity_gain, double max_force, double max_velocity)
{
    if (!h) return fail(nullptr, PNR_ERR_INVALID, "null handle");
    if (h->cfg.mode != PNR_MODE_DYNAMIC) return fail(h, PNR_ERR_INVALID, "pnr_set_joint_motor: motors exist in dynamics mode only");
    if (joint < 0 || joint >= kDof) return fail(h, PNR_ERR_INVALID, "pnr_set_joint_motor: joint %d out of [0, 6)", joint);
    if (control_mode != PNR_CONTROL_POSITION && control_mode != PNR_CONTROL_VELOCITY)
        return fail(h, PNR_ERR_INVALID, "pnr_set_joint_motor: control_mode %d", control_mode);
    const pnr_config& c = h->cfg;
    // an argument left out (NaN) takes the handle's own value, as setJointMotorControl2's optional arguments take Bullet's defaults
    const auto pick = [](double given, double dflt) { return given == given ? given : dflt; };
    const double kp = pick(position_gain, c.pd_kp), kd = pick(velocity_gain, c.pd_kd);
    const double fmax = pick(max_force, c.torque_limit), vmax = pick(max_velocity, c.max_velocity);
    if (kd <= 0 && control_mode == PNR_CONTROL_POSITION && vmax > 0)
        return fail(h, PNR_ERR_INVALID, "pnr_set_joint_motor: maxVelocity needs velocity_gain > 0");
    const bool velocity = control_mode == PNR_CONTROL_VELOCITY, capped = !velocity && vmax > 0;
    JointMotorTable& W = h->motors;       // the folding of fill_base (oracle: orc_dyn_motor_torque), per joint
    W.kp[joint] = (velocity || capped) ? 0.f : (float)kp;
    W.kd[joint] = (float)kd;
    W.cpos[joint] = capped ? (float)(kp / kd) : 0.f;
    W.vcap[joint] = capped ? (float)vmax : INFINITY;
    W.tcap[joint] = fmax > 0 ? (float)fmax : INFINITY;
    W.r_ref[joint] = velocity ? 0.f : (float)target_position;
    W.v_ref[joint] = (float)pick(target_velocity, 0.0);
    W.from_cmd[joint] = 0;
    return PNR_OK;
}

int pnr_world_step(pnr_handle h, float* joint_state, void* stream)
{
    if (!h) return fail(nullptr, PNR_ERR_INVALID, "null handle");
    if (!h->ready) return fail(h, PNR_ERR_INVALID, "pnr_world_step before the first pnr_reset (or pnr_set_state)");
    DeviceGuard g(h->device);
    hipStream_t st = (hipStream_t)stream;
    if (h->cfg.mode != PNR_MODE_DYNAMIC) {
        if (!joint_state) return fail(h, PNR_ERR_INVALID, "pnr_world_step: a kinematic-mode handle steps the caller's joint_state [n][12]");
        const long long items = h->n * kDof;
        hipLaunchKernelGGL(kin_world_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, joint_state, (long long)h->n, (float)h->k.dt);
        HIP_TRY(h, hipGetLastError());
        return PNR_OK;
    }
    if (joint_state) return fail(h, PNR_ERR_INVALID, "pnr_world_step: joint_state must be NULL in dynamics mode (the simulated joints are the handle's)");
    const DynParams& D = h->dbase;
    const dim3 grid((unsigned)((h->n + kWave - 1) / kWave));
    const bool ct = D.has_ground || D.has_box || D.n_scene > 0;
    // (always the instantiation that reads the per-env link scales: without randomisation they are stored as 1)
#define PNR_WORLD_LAUNCH(C) hipLaunchKernelGGL((dyn_world_kernel<true, C>), grid, dim3(kWave), 0, st, h->state, h->dyn, (long long)h->n, D, h->motors)
    if (D.inertia_scaled) { if (ct) PNR_WORLD_LAUNCH(3); else PNR_WORLD_LAUNCH(2); }
    else { if (ct) PNR_WORLD_LAUNCH(1); else PNR_WORLD_LAUNCH(0); }
#undef PNR_WORLD_LAUNCH
    HIP_TRY(h, hipGetLastError());
    return PNR_OK;
}

}  // extern "C"

int pnr_env_rollout_params(pnr_handle h, pnr::KParams* out, float* max_v_to_r, int* device)
{
    if (!h || !out || !max_v_to_r || !device) return fail(nullptr, PNR_ERR_INVALID, "pnr_ppo_rollout: null handle");
    if (h->cfg.mode != PNR_MODE_KINEMATIC)
        return fail(h, PNR_ERR_INVALID, "pnr_ppo_rollout runs kinematic-mode handles (dynamics mode: pnr_mlp_act + pnr_step per step)");
    if (h->cfg.obs_layout != PNR_ENV_MAJOR || h->cfg.action_layout != PNR_ENV_MAJOR)
        return fail(h, PNR_ERR_INVALID, "pnr_ppo_rollout needs env-major observations and actions");
    if (!h->ready) return fail(h, PNR_ERR_INVALID, "pnr_ppo_rollout before the first pnr_reset (or pnr_set_state)");
    *out = h->base;
    out->diag = 0;
    *max_v_to_r = (float)h->cfg.max_v_to_r;
    *device = h->device;
    return PNR_OK;
}

extern "C" {

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
    const size_t bytes = sizeof(float) * PNR_DYN_STATE_WORDS * (size_t)h->n;
    HIP_TRY(h, hipMemcpyAsync(words_out, h->dyn, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PNR_OK;
}

int pnr_set_dyn_state(pnr_handle h, const float* words_in, void* stream)
{
    if (!h || !words_in) return fail(h, PNR_ERR_INVALID, "pnr_set_dyn_state: null argument");
    if (!h->dyn) return fail(h, PNR_ERR_UNSUPPORTED, "handle is not in dynamics mode");
    DeviceGuard g(h->device);
    h->dyn_set = true;
    if (h->kin_set) h->ready = true;
    const size_t bytes = sizeof(float) * PNR_DYN_STATE_WORDS * (size_t)h->n;
    HIP_TRY(h, hipMemcpyAsync(h->dyn, words_in, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PNR_OK;
}

}  // extern "C"
