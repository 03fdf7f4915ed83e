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

// <a, b>: one partial per block (<= SN_DOTB blocks: 64 made the largest layers' dot 2.5x slower than the atomic version); the consumers
// add the partials in a fixed order (dot_fold) -- deterministic, where one float atomicAdd per block depended on the order the blocks
// finished in
constexpr int SN_DOTB = 256;
__global__ void dot_kernel(const float* a, const float* b, float* part, size_t n) {
  __shared__ float red[16];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i] * b[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__device__ __forceinline__ float dot_fold(const float* part, int nb) {      // every thread of the block calls; result in every thread
  __shared__ float sh;
  if (threadIdx.x < 64) {
    float v = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) v += part[i];
    v = wave_sum(v);
    if (threadIdx.x == 0) sh = v;
  }
  __syncthreads();
  return sh;
}

// dw = g - (dot * inv_sigma) * u v^T
__global__ void sn_grad_kernel(const float* g, const float* u, const float* v, const float* sigma, const float* dot, int ndot, float* dw,
                               int rows, int cols) {
  const size_t n = (size_t)rows * cols;
  const float k = dot_fold(dot, ndot) * sigma[1];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
    dw[i] = g[i] - k * u[r] * v[c];
  }
}

// ----------------------------------------------------------------------------------------------------
// All spectral-normalised layers of a network, several consecutive power-iteration rounds, in four launches per round and
// with a FIXED summation order (no atomics): replicas of a data-parallel run then advance bit-identical u / v, and a batched
// discriminator pass gets the sigma of each of its image groups (round r = the r-th application of the layer, models.py:185-188)
// from one call.  Stages per round (blockIdx.z / .y = layer):
//   1. part[slab][j] = sum_{i in slab} W[i][j] u[i]           (128-row slabs; thread per column, coalesced along j)
//   2. t[j] = sum_slab part[slab][j]; v = t / max(||t||, eps)  (one block per layer)
//   3. s[i] = sum_j W[i][j] v[j]                               (one block per row)
//   4. u = s / max(||s||, eps); sigma = u . s                  (one block per layer); round outputs: sigma[r], 1/sigma[r], u, v snapshots
// ----------------------------------------------------------------------------------------------------
constexpr int SN_MAXL = 8, SN_SLAB = 128;
struct SnLayer {
  const float* w;
  float* u;
  float* v;
  float* sigma;       // [rounds]
  float* inv_sigma;   // [rounds]
  float* u_hist;      // [rounds][rows] or null
  float* v_hist;      // [rounds][cols] or null
  float* tmp;         // [nslab * cols + rows]
  int rows, cols;
};
struct SnArgs {
  SnLayer l[SN_MAXL];
  int nlayers, round, do_iter;
  float eps;
};

__global__ void snm_wt_u_kernel(SnArgs a) {
  const SnLayer& L = a.l[blockIdx.z];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i0 = blockIdx.y * SN_SLAB;
  if (j >= L.cols || i0 >= L.rows) return;
  int i1 = i0 + SN_SLAB;
  if (i1 > L.rows) i1 = L.rows;
  float acc = 0.f;
  for (int i = i0; i < i1; ++i) acc += L.w[(size_t)i * L.cols + j] * L.u[i];
  L.tmp[(size_t)blockIdx.y * L.cols + j] = acc;
}

__global__ void snm_norm_v_kernel(SnArgs a) {
  __shared__ float red[16];
  const SnLayer& L = a.l[blockIdx.x];
  const int nslab = (L.rows + SN_SLAB - 1) / SN_SLAB;
  float q = 0.f;
  for (int j = threadIdx.x; j < L.cols; j += blockDim.x) {
    float t = 0.f;
    for (int sl = 0; sl < nslab; ++sl) t += L.tmp[(size_t)sl * L.cols + j];
    L.tmp[j] = t;                       // slab 0's row becomes the sum (only this thread touches column j)
    q += t * t;
  }
  q = block_sum(q, red);
  const float inv = 1.f / fmaxf(sqrtf(q), a.eps);
  for (int j = threadIdx.x; j < L.cols; j += blockDim.x) {
    const float vj = L.tmp[j] * inv;
    L.v[j] = vj;
    if (L.v_hist) L.v_hist[(size_t)a.round * L.cols + j] = vj;
  }
}

__global__ void snm_w_v_kernel(SnArgs a) {
  __shared__ float red[16];
  const SnLayer& L = a.l[blockIdx.y];
  const int i = blockIdx.x;
  if (i >= L.rows) return;
  const int nslab = (L.rows + SN_SLAB - 1) / SN_SLAB;
  float acc = 0.f;
  for (int j = threadIdx.x; j < L.cols; j += blockDim.x) acc += L.w[(size_t)i * L.cols + j] * L.v[j];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) L.tmp[(size_t)nslab * L.cols + i] = acc;
}

__global__ void snm_finish_kernel(SnArgs a) {
  __shared__ float red[16];
  const SnLayer& L = a.l[blockIdx.x];
  const int nslab = (L.rows + SN_SLAB - 1) / SN_SLAB;
  const float* s = L.tmp + (size_t)nslab * L.cols;
  float inv = 1.f;
  if (a.do_iter) {
    float q = 0.f;
    for (int i = threadIdx.x; i < L.rows; i += blockDim.x) q += s[i] * s[i];
    q = block_sum(q, red);
    inv = 1.f / fmaxf(sqrtf(q), a.eps);
  }
  float d = 0.f;
  for (int i = threadIdx.x; i < L.rows; i += blockDim.x) {
    const float ui = a.do_iter ? s[i] * inv : L.u[i];
    if (a.do_iter) L.u[i] = ui;
    if (L.u_hist) L.u_hist[(size_t)a.round * L.rows + i] = ui;
    d += ui * s[i];
  }
  d = block_sum(d, red);
  if (threadIdx.x == 0) {
    L.sigma[a.round] = d;
    L.inv_sigma[a.round] = 1.f / d;
  }
  if (!a.do_iter && L.v_hist)
    for (int j = threadIdx.x; j < L.cols; j += blockDim.x) L.v_hist[(size_t)a.round * L.cols + j] = L.v[j];
}

// dw (+)= g - (<g,w> * inv_sigma) * u v^T   (dot: ndot block partials)
__global__ void sn_grad_acc_kernel(const float* g, const float* u, const float* v, const float* inv_sigma, const float* dot, int ndot, float* dw,
                                   int rows, int cols, int acc) {
  const size_t n = (size_t)rows * cols;
  const float k = dot_fold(dot, ndot) * inv_sigma[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
    const float val = g[i] - k * u[r] * v[c];
    dw[i] = acc ? dw[i] + val : val;
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

// torch.optim.RMSprop(lr, alpha, eps) with its defaults weight_decay 0, momentum 0, centered False (trainer.py:339-342):
//   square_avg = alpha * square_avg + (1 - alpha) * g^2;   p -= lr * g / (sqrt(square_avg) + eps)      (square_avg lives in desc.v)
__global__ void rmsprop_kernel(const uegan_adam_tensor* desc, float lr, float alpha, float eps, float grad_scale) {
  const uegan_adam_tensor d = desc[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = d.g[i] * grad_scale;
    const float v = alpha * d.v[i] + (1.f - alpha) * g * g;
    d.v[i] = v;
    d.p[i] = d.p[i] - lr * (g / (sqrtf(v) + eps));
  }
}

}  // namespace uegan

using namespace uegan;

extern "C" int uegan_version(void) { return UEGAN_VERSION; }
extern "C" const char* uegan_last_error(void) { return g_err; }

extern "C" int uegan_specnorm_grad(const float* g, const float* w, const float* u, const float* v, const float* sigma, float* dw, int rows,
                                   int cols, float* tmp, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(g && w && u && v && sigma && dw && tmp && rows > 0 && cols > 0, "bad specnorm_grad args");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)rows * cols;
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  const int nd = blocks < SN_DOTB ? blocks : SN_DOTB;
  hipLaunchKernelGGL(dot_kernel, dim3(nd), dim3(256), 0, s, g, w, tmp, n);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sn_grad_kernel, dim3(blocks), dim3(256), 0, s, g, u, v, sigma, tmp, nd, dw, rows, cols);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" size_t uegan_specnorm_grad_workspace_floats(void) { return SN_DOTB; }

extern "C" size_t uegan_specnorm_multi_workspace_floats(int rows, int cols) {
  return (size_t)((rows + SN_SLAB - 1) / SN_SLAB) * cols + rows;
}

extern "C" int uegan_specnorm_multi(const uegan_sn_layer* layers, int n_layers, int n_rounds, int do_iter, float eps, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(layers && n_layers >= 1 && n_layers <= SN_MAXL && n_rounds >= 1, "specnorm_multi: 1..%d layers, >= 1 round", SN_MAXL);
  SnArgs a;
  int maxrows = 0, maxcols = 0;
  for (int i = 0; i < n_layers; ++i) {
    const uegan_sn_layer& s = layers[i];
    UEGAN_CHECK_ARG(s.w && s.u && s.v && s.sigma && s.inv_sigma && s.tmp && s.rows > 0 && s.cols > 0, "bad specnorm layer %d", i);
    a.l[i].w = s.w; a.l[i].u = s.u; a.l[i].v = s.v; a.l[i].sigma = s.sigma; a.l[i].inv_sigma = s.inv_sigma; a.l[i].u_hist = s.u_hist;
    a.l[i].v_hist = s.v_hist; a.l[i].tmp = s.tmp; a.l[i].rows = s.rows; a.l[i].cols = s.cols;
    if (s.rows > maxrows) maxrows = s.rows;
    if (s.cols > maxcols) maxcols = s.cols;
  }
  a.nlayers = n_layers; a.do_iter = do_iter ? 1 : 0; a.eps = eps;
  hipStream_t st = (hipStream_t)stream;
  for (int r = 0; r < n_rounds; ++r) {
    a.round = r;
    if (do_iter) {
      hipLaunchKernelGGL(snm_wt_u_kernel, dim3((maxcols + 255) / 256, (maxrows + SN_SLAB - 1) / SN_SLAB, n_layers), dim3(256), 0, st, a);
      UEGAN_CHECK_LAUNCH();
      hipLaunchKernelGGL(snm_norm_v_kernel, dim3(n_layers), dim3(1024), 0, st, a);
      UEGAN_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(snm_w_v_kernel, dim3(maxrows, n_layers), dim3(256), 0, st, a);
    UEGAN_CHECK_LAUNCH();
    hipLaunchKernelGGL(snm_finish_kernel, dim3(n_layers), dim3(256), 0, st, a);
    UEGAN_CHECK_LAUNCH();
  }
  return UEGAN_OK;
}

extern "C" int uegan_specnorm_sigma(const float* w, float* u, float* v, int rows, int cols, int do_iter, float eps, float* sigma_out,
                                    float* tmp, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(w && u && v && sigma_out && tmp && rows > 0 && cols > 0, "bad specnorm args");
  uegan_sn_layer l;
  l.w = w; l.u = u; l.v = v; l.sigma = sigma_out; l.inv_sigma = sigma_out + 1; l.u_hist = nullptr; l.v_hist = nullptr; l.tmp = tmp;
  l.rows = rows; l.cols = cols;
  return uegan_specnorm_multi(&l, 1, 1, do_iter, eps, stream);
}

extern "C" int uegan_specnorm_grad_acc(const float* g, const float* w, const float* u, const float* v, const float* inv_sigma, float* dw,
                                       int rows, int cols, float* tmp, int accumulate, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(g && w && u && v && inv_sigma && dw && tmp && rows > 0 && cols > 0, "bad specnorm_grad args");
  UEGAN_CHECK_ARG(!(accumulate && g == dw), "specnorm_grad_acc: accumulate needs g and dw in different buffers");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)rows * cols;
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  const int nd = blocks < SN_DOTB ? blocks : SN_DOTB;
  hipLaunchKernelGGL(dot_kernel, dim3(nd), dim3(256), 0, s, g, w, tmp, n);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sn_grad_acc_kernel, dim3(blocks), dim3(256), 0, s, g, u, v, inv_sigma, tmp, nd, dw, rows, cols, accumulate ? 1 : 0);
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

extern "C" int uegan_rmsprop_step(const uegan_adam_tensor* desc_dev, int n_tensors, int64_t max_n, float lr, float alpha, float eps,
                                  float grad_scale, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(desc_dev && n_tensors > 0 && max_n > 0, "bad rmsprop args");
  int bx = (int)((max_n + 1023) / 1024);
  if (bx > 128) bx = 128;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(rmsprop_kernel, dim3(bx, n_tensors), dim3(256), 0, (hipStream_t)stream, desc_dev, lr, alpha, eps, grad_scale);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}
