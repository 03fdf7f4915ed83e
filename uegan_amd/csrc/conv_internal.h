// Internal (non-ABI) entry points shared between the convolution translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/uegan_hip.h"

namespace uegan {
// narrow-head VALU kernels (heads.hip): stride-1 KxK reflect-padded convs with <= 4 real output channels
bool heads_applicable(const uegan_conv_desc* d);
int heads_fwd(const uegan_conv_desc* d, const void* x, const void* w_ohwi, const float* bias, const float* scale, void* y, hipStream_t s);
bool heads_dgrad_applicable(const uegan_conv_desc* d);
int heads_dgrad(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const float* scale, void* dx, hipStream_t s);
int heads_wgrad_blocks(const uegan_conv_desc* d);
int heads_wgrad(const uegan_conv_desc* d, const void* x, const void* dz, float* ws, hipStream_t s);
}  // namespace uegan
