// hk_swish.h — the actor's Swish on the device: the same value as hk_swishf (include/hk_detmath.h) bit for bit.  The f32 MFMAs
// of the actor do not co-execute with vector instructions on gfx950, so instructions saved in the epilogue come off the
// kernel's time: the clamp of the exp argument is one v_med3_f32 instead of two compare / select pairs (a NaN input still
// ends as NaN through the final product).  tests/swish_device_check.hip compares both forms over all 2^32 inputs on the GPU.
// (Also tried: the reciprocal as v_rcp_f32 + the six fmas of the IEEE division sequence without its range scaling —
// bit-identical once -s is held below 64, but the freer schedule spills at 128 VGPRs and the kernel gains nothing.)
#pragma once
#include "../../include/hk_detmath.h"

namespace hk {

__device__ __forceinline__ float swish(float s)
{
    const float x = __builtin_amdgcn_fmed3f(-s, -87.0f, 88.0f);
    return s * (1.0f / (1.0f + hk_expf_fast_core(x)));
}

}  // namespace hk
