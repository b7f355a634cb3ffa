/*
 * abi_smoke.c — drives libpioneer_amd.so through its C ABI only (no Python, no torch): the calls a
 * maintainer's cgo/JNI/ctypes stub would make.  Checks known answers of SURVEY.md Appendix C:
 *   - FK at q = 0: pointer = (14.6, 1.0, 15.9), obs layout indices, potential starts at 0 (quirk Q3);
 *   - the 8-step integrator trace with action = a_max (one-step latency, mid-step saturation).
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ tests/c/abi_smoke.c -I/opt/rocm/include -Iinclude -Lpioneer_amd/csrc -lpioneer_amd -L/opt/rocm/lib -lamdhip64 -lm
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pioneer_amd.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != PNR_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, pnr_last_error(h)); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int close_to(float got, double want, double tol, const char* what)
{
    if (fabs((double)got - want) > tol) { fprintf(stderr, "MISMATCH %s: got %.9g want %.9g\n", what, got, want); return 0; }
    return 1;
}

int main(void)
{
    pnr_handle h = NULL;
    enum { N = 3 };
    pnr_config cfg;
    pnr_constants k;
    if (pnr_config_default(&cfg) != PNR_OK || pnr_get_constants(&cfg, &k) != PNR_OK) return 1;
    cfg.auto_reset = 0;
    cfg.max_episode_steps = 0;
    CHECK(pnr_create(&cfg, N, 0, 0, 123u, &h));
    if (pnr_num_envs(h) != N) return 1;

    float jp[N * 6] = {0}, tp[N * 3], act[N * 6], obs[N * PNR_OBS_DIM], rew[N];
    uint8_t done[N], trunc[N];
    for (int e = 0; e < N; e++) { tp[3 * e] = 20.f; tp[3 * e + 1] = 0.f; tp[3 * e + 2] = 4.f; }
    for (int e = 0; e < N; e++) for (int j = 0; j < 6; j++) act[6 * e + j] = k.a_max[j];

    float *d_jp, *d_tp, *d_act, *d_obs, *d_rew; uint8_t *d_done, *d_trunc;
    HIP(hipMalloc((void**)&d_jp, sizeof jp)); HIP(hipMalloc((void**)&d_tp, sizeof tp)); HIP(hipMalloc((void**)&d_act, sizeof act));
    HIP(hipMalloc((void**)&d_obs, sizeof obs)); HIP(hipMalloc((void**)&d_rew, sizeof rew));
    HIP(hipMalloc((void**)&d_done, sizeof done)); HIP(hipMalloc((void**)&d_trunc, sizeof trunc));
    HIP(hipMemcpy(d_jp, jp, sizeof jp, hipMemcpyHostToDevice)); HIP(hipMemcpy(d_tp, tp, sizeof tp, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_act, act, sizeof act, hipMemcpyHostToDevice));

    /* a step before the first reset must fail loudly */
    if (pnr_step(h, d_act, d_obs, d_rew, d_done, d_trunc, NULL, NULL) != PNR_ERR_INVALID) { fprintf(stderr, "step before reset did not fail\n"); return 1; }

    CHECK(pnr_reset(h, NULL, d_jp, d_tp, d_obs, NULL));
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(obs, d_obs, sizeof obs, hipMemcpyDeviceToHost));
    int ok = 1;
    for (int e = 0; e < N; e++) {
        const float* o = obs + e * PNR_OBS_DIM;
        ok &= close_to(o[126], 14.6, 2e-5, "pointer.x") & close_to(o[127], 1.0, 2e-5, "pointer.y") & close_to(o[128], 15.9, 2e-5, "pointer.z");
        ok &= close_to(o[129], 20.0, 0, "target.x") & close_to(o[136], 0.0, 0, "potential after reset");
        ok &= close_to(o[36], 3.141599894, 1e-7, "r_hi[0]") & close_to(o[6], 1.0, 1e-7, "cos r[0]");
    }
    static const double trace[8][4] = {   /* v0, r0, v1, r1 after step t = 1..8 */
        {0, 0, 0, 0}, {5.236000, 0.109083, 2.181667, 0.045451}, {10.471999, 0.436333, 4.363333, 0.181806},
        {12.566400, 0.942480, 5.236000, 0.392700}, {12.566400, 1.466080, 5.236000, 0.610867},
        {12.566400, 1.989680, 5.236000, 0.829033}, {12.566400, 2.513280, 5.236000, 1.047200},
        {12.566400, 3.036880, 5.236000, 1.265367}};
    for (int t = 0; t < 8; t++) {
        CHECK(pnr_step(h, d_act, d_obs, d_rew, d_done, d_trunc, NULL, NULL));
        HIP(hipDeviceSynchronize());
        HIP(hipMemcpy(obs, d_obs, sizeof obs, hipMemcpyDeviceToHost));
        const float* o = obs + 1 * PNR_OBS_DIM;
        ok &= close_to(o[90], trace[t][0], 2e-6, "v[0]") & close_to(o[0], trace[t][1], 2e-6, "r[0]");
        ok &= close_to(o[91], trace[t][2], 2e-6, "v[1]") & close_to(o[1], trace[t][3], 2e-6, "r[1]");
        ok &= close_to(o[108], k.a_max[0], 0, "obs shows the action just given");
    }
    /* World.step() alone (bullet_scene.py:273-275; ABI 5): on a kinematic-mode handle the caller's joint buffer [N][12] = q | qd moves by
     * qd * step_time and stops at the joint limit — the reference's demo loop (pioneer_knm_env.py:282-295) through the plain C ABI */
    {
        float js[N * 12] = {0}, *d_js;
        js[12 * 1 + 1] = 0.25f; js[12 * 1 + 6 + 1] = 1.0f;             /* env 1, joint robot:hinge1_to_arm1: position 0.25, velocity 1 */
        js[12 * 2 + 0] = 3.1f;  js[12 * 2 + 6 + 0] = 2.0f;             /* env 2, joint 0: one step from its limit */
        HIP(hipMalloc((void**)&d_js, sizeof js)); HIP(hipMemcpy(d_js, js, sizeof js, hipMemcpyHostToDevice));
        if (pnr_world_step(h, NULL, NULL) != PNR_ERR_INVALID) { fprintf(stderr, "kinematic world step without a joint buffer did not fail\n"); return 1; }
        if (pnr_set_joint_motor(h, 0, PNR_CONTROL_VELOCITY, 0, 1.0, NAN, NAN, NAN, NAN) != PNR_ERR_INVALID) { fprintf(stderr, "a motor on a kinematic handle did not fail\n"); return 1; }
        for (int t = 0; t < 3; t++) CHECK(pnr_world_step(h, d_js, NULL));
        HIP(hipDeviceSynchronize());
        HIP(hipMemcpy(js, d_js, sizeof js, hipMemcpyDeviceToHost));
        ok &= close_to(js[12 * 1 + 1], 0.25 + 3 * k.dt, 1e-6, "world.step: q += qd * step_time") & close_to(js[12 * 1 + 7], 1.0, 0, "velocity kept");
        ok &= close_to(js[12 * 2 + 0], k.r_hi[0], 0, "world.step: stopped at the limit") & close_to(js[12 * 2 + 6], 0.0, 0, "velocity zeroed there");
        ok &= close_to(js[12 * 0 + 3], 0.0, 0, "a joint at rest stays");
        hipFree(d_js);
    }
    {   /* the binary says what it was built from */
        const char* fp = pnr_build_fingerprint();
        if (!fp || strncmp(fp, "api=", 4) != 0 || !strstr(fp, ";learn=")) { fprintf(stderr, "pnr_build_fingerprint: %s\n", fp ? fp : "(null)"); return 1; }
    }
    CHECK(pnr_destroy(h));
    hipFree(d_jp); hipFree(d_tp); hipFree(d_act); hipFree(d_obs); hipFree(d_rew); hipFree(d_done); hipFree(d_trunc);
    if (!ok) return 2;
    printf("abi_smoke ok\n");
    return 0;
}
