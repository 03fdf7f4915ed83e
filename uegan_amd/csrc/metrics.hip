// Image-level helpers around the generator: the history pool's gather/scatter (utils.py:23-50) and the evaluation metrics of
// the inference configuration -- 8-bit quantisation (torchvision save_image as called at tester.py:70-71), PSNR
// (metrics/CalcPSNR.py:85-92 with the 4-pixel border crop of :24,56) and SSIM with skimage's defaults as called at
// metrics/CalcSSIM.py:63 (7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance, data_range 255, mean over the three
// channels) -- all on the device, so validation never copies images to the host.
#include "common.h"

namespace uegan {

// ---- copy whole images between two stacks by index table (the tables travel by value: no host-to-device copy) ----
constexpr int COPY_MAX_IMAGES = 64;
struct CopyTable {
  int dst[COPY_MAX_IMAGES];
  int src[COPY_MAX_IMAGES];      // >= 0: image of stack A, < 0: image ~src of stack B
};

template <int V>
__global__ void copy_images_kernel(float* dst, const float* src_a, const float* src_b, CopyTable t, size_t elems) {
  const int img = blockIdx.y;
  const int si = t.src[img];
  const float* s = si >= 0 ? src_a + (size_t)si * elems : src_b + (size_t)(~si) * elems;
  float* d = dst + (size_t)t.dst[img] * elems;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < elems; i += (size_t)gridDim.x * blockDim.x * V) {
    if (V == 4) *reinterpret_cast<f32x4*>(d + i) = *reinterpret_cast<const f32x4*>(s + i);
    else d[i] = s[i];
  }
}

// ---- [-1,1] NCHW fp32 -> uint8 NHWC: denorm (utils.py:128-130) then save_image's mul(255).add(0.5).clamp(0,255).to(uint8) ----
__global__ void quantize_u8_kernel(const float* x, uint8_t* y, int B, int C, int HW) {
  const size_t total = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t p = i / C;
    const int b = (int)(p / HW);
    const size_t hw = p - (size_t)b * HW;
    float v = (x[((size_t)b * C + c) * HW + hw] + 1.f) / 2.f;
    v = fminf(fmaxf(v, 0.f), 1.f);
    v = v * 255.f + 0.5f;
    v = fminf(fmaxf(v, 0.f), 255.f);
    y[i] = (uint8_t)v;                       // truncation, like Tensor.to(torch.uint8)
  }
}

// ---- sum of squared differences over the cropped region, per image (double accumulation; the terms are integers and the sums stay
// below 2^53, so the double atomicAdd per block is exact and the result does not depend on the order of the blocks) ----
__global__ void sqdiff_u8_kernel(const uint8_t* a, const uint8_t* b, double* out, int H, int W, int C, int crop) {
  const int img = blockIdx.y;
  const int h = H - 2 * crop, w = W - 2 * crop;
  const size_t n = (size_t)h * w * C;
  const uint8_t* pa = a + (size_t)img * H * W * C;
  const uint8_t* pb = b + (size_t)img * H * W * C;
  double sd = 0.0;      // (terms are integers <= 65025: fp32 sums would lose bits after ~258 of them)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t p = i / C;
    const int x = (int)(p % w), y = (int)(p / w);
    const size_t o = ((size_t)(y + crop) * W + (x + crop)) * C + c;
    const float d = (float)pa[o] - (float)pb[o];
    sd += (double)(d * d);
  }
  for (int o = 32; o > 0; o >>= 1) sd += __shfl_xor(sd, o, 64);
  __shared__ double dred[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) dred[wv] = sd;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) t += dred[i];
    atomicAdd(out + img, t);
  }
}

// ---- SSIM (skimage.metrics.structural_similarity defaults, channel_axis = -1, data_range = 255) ----
// One thread per (valid window position, channel): the 7x7 window sums in double-free fp32 are exact for 8-bit data
// (49 * 255^2 < 2^24).  S is accumulated per image in double.
__global__ void ssim_u8_kernel(const uint8_t* a, const uint8_t* b, double* out, int H, int W, int C, int crop) {
  constexpr int WIN = 7;
  const int img = blockIdx.y;
  const int h = H - 2 * crop, w = W - 2 * crop;          // the cropped image skimage sees
  const int vh = h - (WIN - 1), vw = w - (WIN - 1);      // window centres left after skimage's own (WIN-1)/2 border crop
  const size_t n = (size_t)vh * vw * C;
  const uint8_t* pa = a + (size_t)img * H * W * C;
  const uint8_t* pb = b + (size_t)img * H * W * C;
  const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
  const double NP = WIN * WIN, cov_norm = NP / (NP - 1.0);
  double sd = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t p = i / C;
    const int x = (int)(p % vw), y = (int)(p / vw);
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
    for (int dy = 0; dy < WIN; ++dy)
      for (int dx = 0; dx < WIN; ++dx) {
        const size_t o = ((size_t)(y + dy + crop) * W + (x + dx + crop)) * C + c;
        const float u = (float)pa[o], v = (float)pb[o];
        sx += u; sy += v; sxx += u * u; syy += v * v; sxy += u * v;
      }
    const double ux = sx / NP, uy = sy / NP, uxx = sxx / NP, uyy = syy / NP, uxy = sxy / NP;
    const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
    const double A1 = 2.0 * ux * uy + C1, A2 = 2.0 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
    sd += (A1 * A2) / (B1 * B2);
  }
  for (int o = 32; o > 0; o >>= 1) sd += __shfl_xor(sd, o, 64);
  __shared__ double dred[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) dred[wv] = sd;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) t += dred[i];
    // order-independent accumulation: the block sum as 2^-32 fixed point in a 64-bit integer (|S| <= 1 per window: no overflow below
    // 2^31 windows per image); ssim_fix_kernel turns the slot into the double the caller reads
    atomicAdd(reinterpret_cast<unsigned long long*>(out) + img, (unsigned long long)(long long)llrint(t * 4294967296.0));
  }
}
__global__ void ssim_fix_kernel(double* out, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) out[i] = (double)(long long)reinterpret_cast<unsigned long long*>(out)[i] * (1.0 / 4294967296.0);
}

// total = sum_i w_i * t_i (accumulated left to right, like `a*x + b*y + c*z` in the reference's trainer.py:104-115), scaled[i] = w_i * t_i
struct ScalarSumArgs {
  const float* t[8];
  float w[8];
  int n;
};
__global__ void scalar_wsum_kernel(ScalarSumArgs a, float* total, float* scaled) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float acc = 0.f;
    for (int i = 0; i < a.n; ++i) {
      const float v = a.w[i] * a.t[i][0];
      if (scaled) scaled[i] = v;
      acc = i == 0 ? v : acc + v;
    }
    *total = acc;
  }
}
// gout[i] = w_i * g
__global__ void scalar_wsum_bwd_kernel(ScalarSumArgs a, const float* g, float* gout) {
  if (blockIdx.x == 0 && (int)threadIdx.x < a.n) gout[threadIdx.x] = a.w[threadIdx.x] * g[0];
}

// out[i] = t_i[0]
__global__ void gather_scalars_kernel(ScalarSumArgs a, float* out) {
  if (blockIdx.x == 0 && (int)threadIdx.x < a.n) out[threadIdx.x] = a.t[threadIdx.x][0];
}

}  // namespace uegan

using namespace uegan;

extern "C" int uegan_gather_scalars(int n, const float* const* src, float* out, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(n >= 1 && n <= 8 && src && out, "gather_scalars: 1..8 scalars");
  ScalarSumArgs a;
  a.n = n;
  for (int i = 0; i < 8; ++i) { a.t[i] = i < n ? src[i] : nullptr; a.w[i] = 0.f; }
  for (int i = 0; i < n; ++i) UEGAN_CHECK_ARG(src[i], "null scalar %d", i);
  hipLaunchKernelGGL(gather_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_fill_zero(void* p, size_t bytes, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(p || bytes == 0, "null pointer");
  if (bytes == 0) return UEGAN_OK;
  hipError_t e = hipMemsetAsync(p, 0, bytes, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); return UEGAN_E_HIP; }
  return UEGAN_OK;
}

extern "C" int uegan_scalar_wsum(int n, const float* const* terms, const float* weights, float* total, float* scaled, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(n >= 1 && n <= 8 && terms && weights && total, "scalar_wsum: 1..8 terms");
  ScalarSumArgs a;
  a.n = n;
  for (int i = 0; i < 8; ++i) { a.t[i] = i < n ? terms[i] : nullptr; a.w[i] = i < n ? weights[i] : 0.f; }
  for (int i = 0; i < n; ++i) UEGAN_CHECK_ARG(terms[i], "null term %d", i);
  hipLaunchKernelGGL(scalar_wsum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, total, scaled);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_scalar_wsum_bwd(int n, const float* weights, const float* g, float* gout, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(n >= 1 && n <= 8 && weights && g && gout, "scalar_wsum_bwd: 1..8 terms");
  ScalarSumArgs a;
  a.n = n;
  for (int i = 0; i < 8; ++i) { a.t[i] = nullptr; a.w[i] = i < n ? weights[i] : 0.f; }
  hipLaunchKernelGGL(scalar_wsum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, g, gout);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_copy_images(float* dst, const float* src_a, const float* src_b, const int32_t* dst_idx, const int32_t* src_idx,
                                 int n_images, int64_t image_elems, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(dst && src_a && src_b && dst_idx && src_idx && image_elems > 0, "bad copy_images args");
  UEGAN_CHECK_ARG(n_images >= 0 && n_images <= COPY_MAX_IMAGES, "copy_images takes at most %d images per call (got %d)", COPY_MAX_IMAGES, n_images);
  if (n_images == 0) return UEGAN_OK;
  CopyTable t;
  for (int i = 0; i < n_images; ++i) {
    UEGAN_CHECK_ARG(dst_idx[i] >= 0, "negative destination index");
    t.dst[i] = dst_idx[i];
    t.src[i] = src_idx[i];
  }
  const bool vec = image_elems % 4 == 0 && ((uintptr_t)dst % 16 == 0) && ((uintptr_t)src_a % 16 == 0) && ((uintptr_t)src_b % 16 == 0);
  const size_t work = vec ? (size_t)image_elems / 4 : (size_t)image_elems;
  const int bx = (int)((work + 255) / 256 < 256 ? (work + 255) / 256 : 256);
  if (vec) hipLaunchKernelGGL((copy_images_kernel<4>), dim3(bx, n_images), dim3(256), 0, (hipStream_t)stream, dst, src_a, src_b, t, (size_t)image_elems);
  else hipLaunchKernelGGL((copy_images_kernel<1>), dim3(bx, n_images), dim3(256), 0, (hipStream_t)stream, dst, src_a, src_b, t, (size_t)image_elems);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_quantize_u8(const float* x_nchw, uint8_t* y_nhwc, int B, int C, int H, int W, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x_nchw && y_nhwc && B > 0 && C > 0 && H > 0 && W > 0, "bad quantize_u8 args");
  const size_t n = (size_t)B * C * H * W;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(quantize_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x_nchw, y_nhwc, B, C, H * W);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_image_metrics_u8(const uint8_t* a_nhwc, const uint8_t* b_nhwc, double* sqdiff_sum, double* ssim_sum, int B, int H, int W,
                                      int C, int crop_border, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(a_nhwc && b_nhwc && (sqdiff_sum || ssim_sum) && B > 0 && C > 0 && crop_border >= 0, "bad image_metrics args");
  UEGAN_CHECK_ARG(H - 2 * crop_border >= 7 && W - 2 * crop_border >= 7, "image too small for a 7x7 SSIM window after the border crop");
  hipStream_t s = (hipStream_t)stream;
  const int h = H - 2 * crop_border, w = W - 2 * crop_border;
  if (sqdiff_sum) {
    UEGAN_CHECK_ARG(hipMemsetAsync(sqdiff_sum, 0, sizeof(double) * B, s) == hipSuccess, "memset failed");
    const size_t n = (size_t)h * w * C;
    const int bx = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
    hipLaunchKernelGGL(sqdiff_u8_kernel, dim3(bx, B), dim3(256), 0, s, a_nhwc, b_nhwc, sqdiff_sum, H, W, C, crop_border);
    UEGAN_CHECK_LAUNCH();
  }
  if (ssim_sum) {
    UEGAN_CHECK_ARG(hipMemsetAsync(ssim_sum, 0, sizeof(double) * B, s) == hipSuccess, "memset failed");
    const size_t n = (size_t)(h - 6) * (w - 6) * C;
    const int bx = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(ssim_u8_kernel, dim3(bx, B), dim3(256), 0, s, a_nhwc, b_nhwc, ssim_sum, H, W, C, crop_border);
    UEGAN_CHECK_LAUNCH();
    hipLaunchKernelGGL(ssim_fix_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ssim_sum, B);
    UEGAN_CHECK_LAUNCH();
  }
  return UEGAN_OK;
}
