// pnr_host.h — host-side plumbing shared by the library's two translation units (pnr_api.hip: the env engine and its
// C ABI; pnr_learn.hip: the PPO driver's kernels and their C ABI).  Nothing here is exported.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>

#include "../../include/pioneer_amd.h"

// writes the message into the handle's buffer (or the thread's, for handle-less calls) and returns `code`
__attribute__((visibility("hidden"))) int pnr_failv(char* handle_err, int code, const char* fmt, va_list ap);

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
