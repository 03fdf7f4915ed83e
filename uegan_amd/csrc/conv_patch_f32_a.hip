// conv_patch_kernel instantiations: f32, kernel-size set a (see conv_patch.h)
#include "conv_patch.h"

namespace uegan {

int conv_patch_f32_a(ConvArgs& a, hipStream_t s, int ks) {
  switch (ks) {
#define X(K) case K: return launch_conv_patch<float, K>(a, s);
    X(1) X(2) X(3)
#undef X
    default: return 1;
  }
}

}  // namespace uegan
