// pnr_dyn.h — dynamics mode (ABA + PD).  Placeholder until the dynamics kernel lands.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pioneer_amd.h"
#include "pnr_device.h"
namespace pnr {
inline int dyn_reset_launch(float4*, const KParams&, const pnr_config&, hipStream_t) { return 1; }
inline int dyn_step_launch(float4*, const KParams&, const pnr_config&, hipStream_t) { return 1; }
inline int dyn_words_launch(float4*, float*, const float*, long long, hipStream_t) { return 1; }
}
