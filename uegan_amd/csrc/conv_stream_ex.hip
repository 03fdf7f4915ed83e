// The streaming convolution kernel's instantiations with hi + lo PAIRS of 16-bit planes and / or epilogue extras (uegan_conv2d_fwd_ex, round 6):
// the generator's full-resolution layers in the `precise` mode (enc1, ga1, dec4, dec5.0: operands and results carry ~2 x the significant bits of the
// storage format) and dec4's forward with `y4.mul(x1)` (models.py:69) formed in its epilogue.  A translation unit of its own: the
// plain instantiations in conv.hip keep their compile time and register budgets.
#include "conv_core.h"

#include <type_traits>

namespace uegan {

#define UEGAN_CONV_STREAM_KERNEL_ONLY
#include "conv_stream.h"

namespace {
// (TN, PF, LDS class, waves, STATS, PR, EPX) of the instantiations below
struct ExKey { int tn, pf, lc, nw, stats, pr, epx; };
constexpr ExKey kEx[] = {
    {2, 2, 2, 8, 0, 0, 2},      // dec4 (3x3, 32 + 32 -> 32), plain operands, product epilogue
    {2, 4, 1, 4, 0, 1, 1},      // enc1 (7x7, 8 -> 32): weight pair (the image's own pair rides in the spare channels of its 8-channel pixels)
    {2, 4, 1, 4, 1, 2, 1},      // ga1 (1x1, 32 -> 32) with the InstanceNorm moments
    {2, 4, 1, 4, 0, 2, 1},      // ... without
    {2, 2, 2, 8, 0, 2, 1},      // dec5.0 (3x3, 32 -> 32): one block of 8 waves per CU
    {2, 1, 2, 8, 0, 3, 2},      // dec4 with the attention branch as a pair, product epilogue: one block of 8 waves (8 x 16 tile, one row each)
    {2, 4, 1, 4, 0, 1, 0},      // upsample4's 1x1 (64 -> 32, in front of the bilinear x2): weight pair, plain source and result
};
int find(const ConvStreamPlan& p) {
  for (int i = 0; i < (int)(sizeof(kEx) / sizeof(kEx[0])); ++i) {
    const ExKey& k = kEx[i];
    if (k.tn == p.tn && k.pf == p.pf && k.lc == p.lc && k.nw == p.nw && k.stats == (p.stats ? 1 : 0) && k.pr == p.pr && k.epx == p.epx) return i;
  }
  return -1;
}
}  // namespace

bool conv_stream_ex_available(const ConvStreamPlan& p) {
  if (p.a.cls || p.a.xmir) return false;
  ConvStreamPlan q = p;
  q.stats = false;             // (the caller decides about the moments after planning: both forms of a STATS-capable plan exist)
  if (find(q) >= 0) return true;
  q.stats = true;
  return find(q) >= 0;
}

bool conv_stream_launch_ex(const ConvStreamPlan& p, hipStream_t s) {
  const int blocks = p.blocks;
  switch (find(p)) {
    case 0: hipLaunchKernelGGL((conv_stream_kernel<2, 2, 2, false, false, 8, false, 0, 2>), dim3(blocks), dim3(512), 0, s, p.a); return true;
    case 1: hipLaunchKernelGGL((conv_stream_kernel<2, 4, 1, false, false, 4, false, 1, 1>), dim3(blocks), dim3(256), 0, s, p.a); return true;
    case 2: hipLaunchKernelGGL((conv_stream_kernel<2, 4, 1, false, false, 4, true, 2, 1>), dim3(blocks), dim3(256), 0, s, p.a); return true;
    case 3: hipLaunchKernelGGL((conv_stream_kernel<2, 4, 1, false, false, 4, false, 2, 1>), dim3(blocks), dim3(256), 0, s, p.a); return true;
    case 4: hipLaunchKernelGGL((conv_stream_kernel<2, 2, 2, false, false, 8, false, 2, 1>), dim3(blocks), dim3(512), 0, s, p.a); return true;
    case 5: hipLaunchKernelGGL((conv_stream_kernel<2, 1, 2, false, false, 8, false, 3, 2>), dim3(blocks), dim3(512), 0, s, p.a); return true;
    case 6: hipLaunchKernelGGL((conv_stream_kernel<2, 4, 1, false, false, 4, false, 1, 0>), dim3(blocks), dim3(256), 0, s, p.a); return true;
    default: return false;
  }
}

}  // namespace uegan
