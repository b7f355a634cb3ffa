// pnr_host.h — host-side plumbing shared by the library's two translation units (pnr_api.hip: the env engine and its
// C ABI; pnr_learn.hip: the PPO driver's kernels and their C ABI).  Nothing here is exported.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>

#include "../../include/pioneer_amd.h"

// Build identity (pioneer_amd/_lib.py): every translation unit is compiled with -DPNR_UNIT_FINGERPRINT="<sha16>" — sha256 over the
// compile flags, the unit's own files and include/pioneer_amd.h — and keeps it as a tagged string, so that the loader can refuse a
// binary that was not built from the sources lying next to it (the tag is also what _lib.py scans objects and the .so for).
#ifndef PNR_UNIT_FINGERPRINT
#define PNR_UNIT_FINGERPRINT "unstamped"
#endif
extern "C" __attribute__((visibility("hidden"))) const char* pnr_unit_fingerprint_learn(void);

// writes the message into the handle's buffer (or the thread's, for handle-less calls) and returns `code`
__attribute__((visibility("hidden"))) int pnr_failv(char* handle_err, int code, const char* fmt, va_list ap);

#define HIP_TRY(h, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(h, PNR_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));      \
    } while (0)

// pnr_ppo_rollout (pnr_learn.hip) steps a handle's envs from inside its own kernel: the env engine (pnr_api.hip) checks the handle
// (kinematic mode, env-major layouts, reset done) and hands out its kernel constants (state pointer, counts, reward constants, seed)
// as a pnr::KParams, plus max_v_to_r and the device.  Returns PNR_OK or the error it recorded in the handle.
namespace pnr { struct KParams; }
__attribute__((visibility("hidden"))) int pnr_env_rollout_params(pnr_handle h, pnr::KParams* out, float* max_v_to_r, int* device);

// RAII current-device switch: launches and allocations go to the handle's device.
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) { switched = hipSetDevice(dev) == hipSuccess; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
