// Spectral-norm power iteration / gradient, fused multi-tensor Adam, and the library's error plumbing.
// Reference arithmetic: torch.nn.utils.spectral_norm (models.py:185-188): 1 power iteration per training
// forward, eps 1e-12, sigma = u^T W v, weight = weight_orig / sigma, u/v constants in backward;
// torch.optim.Adam with weight_decay (L2 added to the gradient), trainer.py:337-338.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>

namespace uegan {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// t[j] += sum_{i in this block's row slab} W[i][j] * u[i]   (thread per column, coalesced along j; grid.y = row slabs of 16;
// t is zeroed by the caller)
__global__ void sn_wt_u_kernel(const float* w, const float* u, float* t, int rows, int cols) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  const int i0 = blockIdx.y * 16;
  int i1 = i0 + 16;
  if (i1 > rows) i1 = rows;
  float acc = 0.f;
  for (int i = i0; i < i1; ++i) acc += w[(size_t)i * cols + j] * u[i];
  atomicAdd(t + j, acc);
}

// one block per row i: (optionally) v = t / max(||t||, eps) [block 0 stores it], s[i] = sum_j W[i][j] * v[j]
__global__ void sn_w_v_kernel(const float* w, const float* t, float* v, float* s, int rows, int cols, int normalize, float eps) {
  __shared__ float red[16];
  const int i = blockIdx.x;
  float inv = 1.f;
  if (normalize) {
    float q = 0.f;
    for (int j = threadIdx.x; j < cols; j += blockDim.x) q += t[j] * t[j];
    q = block_sum(q, red);
    inv = 1.f / fmaxf(sqrtf(q), eps);
  }
  const float* src = normalize ? t : v;
  float acc = 0.f;
  for (int j = threadIdx.x; j < cols; j += blockDim.x) {
    const float vj = src[j] * inv;
    acc += w[(size_t)i * cols + j] * vj;
    if (normalize && i == 0) v[j] = vj;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) s[i] = acc;
}

// single block: (optionally) u = s / max(||s||, eps); sigma = dot(u, s)
__global__ void sn_finish_kernel(float* u, const float* s, float* sigma_out, int rows, int update_u, float eps) {
  __shared__ float red[16];
  float inv = 1.f;
  if (update_u) {
    float q = 0.f;
    for (int i = threadIdx.x; i < rows; i += blockDim.x) q += s[i] * s[i];
    q = block_sum(q, red);
    inv = 1.f / fmaxf(sqrtf(q), eps);
  }
  float d = 0.f;
  for (int i = threadIdx.x; i < rows; i += blockDim.x) {
    const float ui = update_u ? s[i] * inv : u[i];
    if (update_u) u[i] = ui;
    d += ui * s[i];
  }
  d = block_sum(d, red);
  if (threadIdx.x == 0) {
    sigma_out[0] = d;
    sigma_out[1] = 1.f / d;
  }
}

__global__ void dot_kernel(const float* a, const float* b, float* out, size_t n) {
  __shared__ float red[16];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i] * b[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, acc);
}

// dw = g - (dot * inv_sigma) * u v^T
__global__ void sn_grad_kernel(const float* g, const float* u, const float* v, const float* sigma, const float* dot, float* dw, int rows,
                               int cols) {
  const size_t n = (size_t)rows * cols;
  const float k = dot[0] * sigma[1];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
    dw[i] = g[i] - k * u[r] * v[c];
  }
}

__global__ void adam_kernel(const uegan_adam_tensor* desc, float step_size, float beta1, float beta2, float inv_sqrt_bc2, float eps,
                            float weight_decay, float grad_scale) {
  const uegan_adam_tensor d = desc[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
    const float p = d.p[i];
    float g = d.g[i] * grad_scale + weight_decay * p;
    const float m = beta1 * d.m[i] + (1.f - beta1) * g;
    const float v = beta2 * d.v[i] + (1.f - beta2) * g * g;
    d.m[i] = m;
    d.v[i] = v;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    d.p[i] = p - step_size * (m / denom);
  }
}

}  // namespace uegan

using namespace uegan;

extern "C" int uegan_version(void) { return UEGAN_VERSION; }
extern "C" const char* uegan_last_error(void) { return g_err; }

extern "C" int uegan_specnorm_sigma(const float* w, float* u, float* v, int rows, int cols, int do_iter, float eps, float* sigma_out,
                                    float* tmp, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(w && u && v && sigma_out && tmp && rows > 0 && cols > 0, "bad specnorm args");
  hipStream_t s = (hipStream_t)stream;
  float* t = tmp;          // [cols]
  float* sv = tmp + cols;  // [rows]
  if (do_iter) {
    hipError_t e = hipMemsetAsync(t, 0, sizeof(float) * cols, s);
    if (e != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); return UEGAN_E_HIP; }
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3((cols + 255) / 256, (rows + 15) / 16), dim3(256), 0, s, w, u, t, rows, cols);
    UEGAN_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(sn_w_v_kernel, dim3(rows), dim3(256), 0, s, w, t, v, sv, rows, cols, do_iter ? 1 : 0, eps);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sn_finish_kernel, dim3(1), dim3(256), 0, s, u, sv, sigma_out, rows, do_iter ? 1 : 0, eps);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_specnorm_grad(const float* g, const float* w, const float* u, const float* v, const float* sigma, float* dw, int rows,
                                   int cols, float* tmp, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(g && w && u && v && sigma && dw && tmp && rows > 0 && cols > 0, "bad specnorm_grad args");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(tmp, 0, sizeof(float), s);
  if (e != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); return UEGAN_E_HIP; }
  const size_t n = (size_t)rows * cols;
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(dot_kernel, dim3(blocks), dim3(256), 0, s, g, w, tmp, n);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sn_grad_kernel, dim3(blocks), dim3(256), 0, s, g, u, v, sigma, tmp, dw, rows, cols);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_adam_l2_step(const uegan_adam_tensor* desc_dev, int n_tensors, int64_t max_n, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, float grad_scale, int step, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(desc_dev && n_tensors > 0 && max_n > 0 && step >= 1, "bad adam args");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  int bx = (int)((max_n + 1023) / 1024);
  if (bx > 128) bx = 128;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(adam_kernel, dim3(bx, n_tensors), dim3(256), 0, (hipStream_t)stream, desc_dev, (float)(lr / bc1), beta1, beta2,
                     (float)(1.0 / sqrt(bc2)), eps, weight_decay, grad_scale);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}
