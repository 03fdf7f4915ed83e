// Input pipeline on the device (data_loader.py:74-82 train, :95-100 test): what torchvision's
//     RandomCrop -> Resize -> RandomHorizontalFlip -> RandomVerticalFlip -> ToTensor -> Normalize(0.5, 0.5)
// does to a decoded 8-bit RGB image, as two kernels over a batch of crop windows that were copied to the device as raw bytes.
//
// Resize on a PIL image is Pillow's two-pass resampler (horizontal, then vertical, an 8-bit intermediate image between the
// passes) with 22-bit fixed-point coefficients of the triangle filter stretched by the scale factor.  The coefficient tables
// are built on the host in double precision exactly as Pillow builds them (uegan_amd/data.py: resample_table) and the device
// does the integer arithmetic -- so the result is BIT-identical to the reference's loader, not "a bilinear resize".
//   pass 1   tmp[b][y][xo][c]  = clip8((2^21 + sum_k pix[b][y][xmin(xo) + k][c] * hc[xo][k]) >> 22)
//   pass 2   v                 = clip8((2^21 + sum_k tmp[b][ymin(yo) + k][xo][c] * vc[yo][k]) >> 22)
//            out[b][c][yo'][xo'] = (v / 255 - 0.5) / 0.5      (fp32, ToTensor's division then Normalize's two operations)
//            with (yo', xo') = (yo, xo) mirrored per image by the flip bits (a flip after the resize, as in the reference).
#include "common.h"

namespace uegan {

constexpr int INPUT_MAX_IMAGES = 64;
constexpr int RESAMPLE_PRECISION_BITS = 32 - 8 - 2;      // Pillow's PRECISION_BITS for 8-bit channels
struct FlipTable { int32_t bits[INPUT_MAX_IMAGES]; };     // bit 0: horizontal flip, bit 1: vertical flip (by value: no copy)

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// table row of output index o: [first input index, number of taps n <= K, K coefficients]
__global__ void resample_h_kernel(const uint8_t* pix, uint8_t* tmp, const int32_t* tab, int K, int B, int H, int W, int OW) {
  const size_t total = (size_t)B * H * OW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xo = (int)(i % OW);
    const size_t row = i / OW;                                   // (b, y)
    const int32_t* t = tab + (size_t)xo * (K + 2);
    const int x0 = t[0], n = t[1];
    const uint8_t* p = pix + (row * W + x0) * 3;
    int s0 = 1 << (RESAMPLE_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int k = 0; k < n; ++k) {
      const int c = t[2 + k];
      s0 += p[3 * k] * c; s1 += p[3 * k + 1] * c; s2 += p[3 * k + 2] * c;
    }
    uint8_t* q = tmp + i * 3;
    q[0] = (uint8_t)clip8(s0 >> RESAMPLE_PRECISION_BITS);
    q[1] = (uint8_t)clip8(s1 >> RESAMPLE_PRECISION_BITS);
    q[2] = (uint8_t)clip8(s2 >> RESAMPLE_PRECISION_BITS);
  }
}

__global__ void resample_v_norm_kernel(const uint8_t* tmp, float* out, const int32_t* tab, int K, FlipTable flips, int B, int H, int OH, int OW) {
  const size_t total = (size_t)B * OH * OW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xo = (int)(i % OW);
    const size_t r = i / OW;
    const int yo = (int)(r % OH), b = (int)(r / OH);
    const int32_t* t = tab + (size_t)yo * (K + 2);
    const int y0 = t[0], n = t[1];
    const uint8_t* p = tmp + (((size_t)b * H + y0) * OW + xo) * 3;
    int s0 = 1 << (RESAMPLE_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int k = 0; k < n; ++k) {
      const int c = t[2 + k];
      const uint8_t* pk = p + (size_t)k * OW * 3;
      s0 += pk[0] * c; s1 += pk[1] * c; s2 += pk[2] * c;
    }
    const int fb = flips.bits[b];
    const int xd = (fb & 1) ? OW - 1 - xo : xo, yd = (fb & 2) ? OH - 1 - yo : yo;
    float* o = out + (((size_t)b * 3) * OH + yd) * OW + xd;
    const size_t plane = (size_t)OH * OW;
    const int v[3] = {clip8(s0 >> RESAMPLE_PRECISION_BITS), clip8(s1 >> RESAMPLE_PRECISION_BITS), clip8(s2 >> RESAMPLE_PRECISION_BITS)};
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v[c], 255.f), 0.5f), 0.5f);
  }
}

}  // namespace uegan

using namespace uegan;

extern "C" int uegan_input_transform(const uint8_t* pixels, int B, int in_h, int in_w, int out_h, int out_w, const int32_t* htab, int hk,
                                     const int32_t* vtab, int vk, const int32_t* flips, uint8_t* tmp, float* out_nchw, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(pixels && htab && vtab && tmp && out_nchw, "input_transform: null pointer");
  UEGAN_CHECK_ARG(B >= 1 && B <= INPUT_MAX_IMAGES, "input_transform takes 1..%d images per call (got %d)", INPUT_MAX_IMAGES, B);
  UEGAN_CHECK_ARG(in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0 && hk >= 1 && vk >= 1, "input_transform: bad geometry");
  FlipTable ft;
  for (int i = 0; i < INPUT_MAX_IMAGES; ++i) ft.bits[i] = (flips && i < B) ? flips[i] : 0;
  hipStream_t s = (hipStream_t)stream;
  const size_t n1 = (size_t)B * in_h * out_w, n2 = (size_t)B * out_h * out_w;
  const int b1 = (int)((n1 + 255) / 256 < 16384 ? (n1 + 255) / 256 : 16384), b2 = (int)((n2 + 255) / 256 < 16384 ? (n2 + 255) / 256 : 16384);
  hipLaunchKernelGGL(resample_h_kernel, dim3(b1), dim3(256), 0, s, pixels, tmp, htab, hk, B, in_h, in_w, out_w);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3(b2), dim3(256), 0, s, (const uint8_t*)tmp, out_nchw, vtab, vk, ft, B, in_h, out_h, out_w);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}
