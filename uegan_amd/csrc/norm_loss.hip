// InstanceNorm (non-affine) forward/backward and the three loss kernels.  All HBM-bound reductions:
// per-thread fp32 accumulation, wave shuffles + LDS for the block stage, fp32 partials combined with
// Chan's formula (means/M2) so the variance never suffers E[x^2]-E[x]^2 cancellation.
//
// Reference arithmetic: nn.InstanceNorm2d(affine=False) (models.py:227,236; losses.py:18,30-34),
// GANLoss 'rahinge' (losses.py:348-362, 393-409), MultiscaleRecLoss (losses.py:219-231),
// PerceptualLoss tap term (losses.py:30-34).
#include "common.h"

namespace uegan {

// ----------------------------------------------------------------------------------------------------
// work decomposition for per-(b,c) reductions over HW pixels of an NHWC tensor.
// A thread owns V consecutive channels (V = one 16-byte chunk when C allows it, else 1) of a strided set of pixels;
// a block = CG channel lanes x PL pixel lanes over one pixel split; partials are combined by the consumer kernels.
// ----------------------------------------------------------------------------------------------------
struct RedPlan {
  int B, HW, C;
  int V;       // channels per thread
  int CG;      // channel lanes per block (power of two <= 64)
  int PL;      // pixel lanes per block = 256 / CG
  int ncg;     // channel groups
  int S;       // pixel splits
  int chunk;   // pixels per split
};

static RedPlan make_plan(int B, int HW, int C, int epc) {
  RedPlan p;
  p.B = B; p.HW = HW; p.C = C;
  p.V = (C % epc == 0) ? epc : 1;
  const int lanes = C / p.V;
  int cg = 1;
  while (cg < lanes && cg < 64) cg <<= 1;
  p.CG = cg;
  p.PL = 256 / cg;
  p.ncg = (lanes + cg - 1) / cg;
  int s = (HW + 1023) / 1024;
  int cap = 64;
  while ((long)B * p.ncg * cap < 512 && cap < 512) cap *= 2;      // few images (inference: B = 1): more splits, so that the grid still covers the chip
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  // ... and SHORTER splits (down to 128 pixels) while the grid is under ~4 blocks per CU: at B = 1 a 512 x 512 x 32 map was 256 blocks of
  // 4 waves -- 53 us for a 33 MB pass (the streaming kernels need many more waves in flight than that to reach HBM speed)
#if defined(UEGAN_EMU)
  constexpr long kGridTarget = 64;      // (CPU emulator: every block is 256 fibers -- same code path, CI-sized grids)
#else
  constexpr long kGridTarget = 1024;
#endif
  while ((long)B * p.ncg * s < kGridTarget && s < 2048 && HW / (2 * s) >= 128) s *= 2;
  // ... and, on the small maps of the attention modules (<= 64 x 64: ga4 / ga5 of a 512^2 input), down to 16 pixels while ONE image's blocks are a fraction
  // of the chip: at batch 1 these were 8 / 32 blocks whose threads walked 32 / 16 dependent loads (moments 9.7 / 8.5 us, apply 8.7 / 5.4 us for 1 MB).
  // Per image, not per batch: the split count of a map must not depend on how many images share the launch (Generator.forward_pair is bit-identical to
  // two passes, tests/test_fused.py)
  while (HW <= 4096 && (long)p.ncg * s < 128 && s < 2048 && HW / (2 * s) >= 16) s *= 2;
  p.chunk = (HW + s - 1) / s;
  p.S = (HW + p.chunk - 1) / p.chunk;
  return p;
}

#define RED_THREAD_SETUP()                                           \
  const int s = blockIdx.x, cg = blockIdx.y, b = blockIdx.z;         \
  const int cl = threadIdx.x % p.CG, pl = threadIdx.x / p.CG;        \
  const int c0 = (cg * p.CG + cl) * V;                               \
  const bool cvalid = c0 < p.C;                                      \
  const int p0 = s * p.chunk;                                        \
  int p1 = p0 + p.chunk;                                             \
  if (p1 > p.HW) p1 = p.HW;                                          \
  const size_t base = (size_t)b * p.HW * p.C + c0;

// partial moments of one tensor: part[((b*S + s)*C + c)*3 + {0,1,2}] = {count, mean, M2}
template <typename T, int V>
__global__ void moments_partial_kernel(const T* x, float* part, RedPlan p) {
  __shared__ float sh[3][V][256];
  RED_THREAD_SETUP();
  float n = 0.f, s1[V], s2[V], K[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { s1[e] = 0.f; s2[e] = 0.f; K[e] = 0.f; }
  if (cvalid) {
    if (p0 + pl < p1) Vec<T, V>::ld(x + base + (size_t)(p0 + pl) * p.C, K);     // shift: first sample of this thread
    _Pragma("unroll 4") for (int q = p0 + pl; q < p1; q += p.PL) {
      float v[V];
      Vec<T, V>::ld(x + base + (size_t)q * p.C, v);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float d = v[e] - K[e];
        s1[e] += d;
        s2[e] += d * d;
      }
      n += 1.f;
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) {
    sh[0][e][threadIdx.x] = n;
    sh[1][e][threadIdx.x] = n > 0.f ? K[e] + s1[e] / n : 0.f;
    sh[2][e][threadIdx.x] = n > 0.f ? s2[e] - s1[e] * s1[e] / n : 0.f;
  }
  __syncthreads();
  // pairwise (Chan) merge over the pixel lanes, all threads working: log2(PL) steps instead of a PL-long serial chain on
  // the CG threads of pixel lane 0 (that tail used to cost as much as the streaming loop)
  for (int half = p.PL >> 1; half > 0; half >>= 1) {
    if (pl < half) {
      const int o = threadIdx.x + half * p.CG;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float na = sh[0][e][threadIdx.x], nb = sh[0][e][o];
        if (nb > 0.f) {
          const float nt = na + nb, d = sh[1][e][o] - sh[1][e][threadIdx.x], r = nb / nt;
          sh[1][e][threadIdx.x] += d * r;
          sh[2][e][threadIdx.x] += sh[2][e][o] + d * d * na * r;
          sh[0][e][threadIdx.x] = nt;
        }
      }
    }
    __syncthreads();
  }
  if (pl == 0 && cvalid) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float* o = part + (((size_t)b * p.S + s) * p.C + c0 + e) * 3;
      o[0] = sh[0][e][threadIdx.x]; o[1] = sh[1][e][threadIdx.x]; o[2] = sh[2][e][threadIdx.x];
    }
  }
}

// one WAVE per (b,c): combine the split partials once (consumers then read 2 floats per channel).  Lane l merges partials
// l, l+64, ..., then a butterfly of pairwise Chan merges (fixed order: deterministic); a serial loop over the splits by one thread
// per channel used to take 13-26 us -- more than the streaming pass it follows on small tensors.
__global__ void moments_finalize_kernel(const float* part, float* mean_out, float* rstd_out, RedPlan p, float eps) {
  const int w = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (w >= p.B * p.C) return;
  const int b = w / p.C, c = w - b * p.C;
  float N = 0.f, M = 0.f, Q = 0.f;
  for (int sp = lane; sp < p.S; sp += 64) {
    const float* o = part + (((size_t)b * p.S + sp) * p.C + c) * 3;
    const float nb = o[0], mb = o[1], qb = o[2];
    if (nb > 0.f) {
      const float nt = N + nb, d = mb - M;
      M += d * nb / nt;
      Q += qb + d * d * N * nb / nt;
      N = nt;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float nb = __shfl_xor(N, o, 64), mb = __shfl_xor(M, o, 64), qb = __shfl_xor(Q, o, 64);
    const float nt = N + nb;
    if (nt > 0.f) {
      // symmetric form: both partners compute the same merged triple
      const float wa = N / nt, wb = nb / nt, d = mb - M;
      Q = Q + qb + d * d * N * wb;
      M = M * wa + mb * wb;
      N = nt;
    }
  }
  if (lane == 0) {
    const float var = N > 0.f ? Q / N : 0.f;
    mean_out[w] = M;
    rstd_out[w] = eps < 0.f ? var : 1.f / sqrtf(var + eps);      // (eps < 0: the caller wants the biased variance itself)
  }
}
// out[(b*C + c)*K + k] = sum_s part[((b*S + s)*C + c)*K + k]: one wave per output element
__global__ void sums_finalize_kernel(const float* part, float* out, RedPlan p, int K) {
  const int w = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (w >= p.B * p.C * K) return;
  const int k = w % K, bc = w / K, b = bc / p.C, c = bc % p.C;
  float t = 0.f;
  for (int sp = lane; sp < p.S; sp += 64) t += part[(((size_t)b * p.S + sp) * p.C + c) * K + k];
  t = wave_sum(t);
  if (lane == 0) out[w] = t;
}

template <typename T, int V>
__global__ void instnorm_apply_kernel(const T* x, T* y, const float* mean_in, const float* rstd_in, RedPlan p) {
  RED_THREAD_SETUP();
  if (!cvalid) return;
  float mean[V], rstd[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    mean[e] = mean_in[(size_t)b * p.C + c0 + e];
    rstd[e] = rstd_in[(size_t)b * p.C + c0 + e];
  }
  _Pragma("unroll 4") for (int q = p0 + pl; q < p1; q += p.PL) {
    float v[V];
    Vec<T, V>::ld(x + base + (size_t)q * p.C, v);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = (v[e] - mean[e]) * rstd[e];
    Vec<T, V>::st(y + base + (size_t)q * p.C, v);
  }
}

// the same on a hi + lo pair of 16-bit planes (value = hi + lo), result as a pair: the attention module's InstanceNorm in the `precise` mode
template <typename T, int V>
__global__ void instnorm_apply_pair_kernel(const T* x, const T* xl, T* y, T* yl, const float* mean_in, const float* rstd_in, RedPlan p) {
  RED_THREAD_SETUP();
  if (!cvalid) return;
  float mean[V], rstd[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    mean[e] = mean_in[(size_t)b * p.C + c0 + e];
    rstd[e] = rstd_in[(size_t)b * p.C + c0 + e];
  }
  _Pragma("unroll 4") for (int q = p0 + pl; q < p1; q += p.PL) {
    float v[V], l[V], h[V];
    Vec<T, V>::ld(x + base + (size_t)q * p.C, v);
    Vec<T, V>::ld(xl + base + (size_t)q * p.C, l);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = ((v[e] - mean[e]) + l[e]) * rstd[e];
    Vec<T, V>::st(y + base + (size_t)q * p.C, v);
#pragma unroll
    for (int e = 0; e < V; ++e) {                      // (what the store rounded to: the lo plane takes the rest)
      T r;
      DT<T>::st(&r, v[e]);
      h[e] = DT<T>::ld(&r);
      l[e] = v[e] - h[e];
    }
    Vec<T, V>::st(yl + base + (size_t)q * p.C, l);
  }
}

// backward partial sums: part[((b*S+s)*C + c)*2 + {0,1}] = {sum dy, sum dy*y}
template <typename T, int V>
__global__ void instnorm_bwd_partial_kernel(const T* dy, const T* y, float* part, RedPlan p) {
  __shared__ float sh[2][V][256];
  RED_THREAD_SETUP();
  float a0[V], a1[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
  if (cvalid) {
    _Pragma("unroll 4") for (int q = p0 + pl; q < p1; q += p.PL) {
      float gv[V], yv[V];
      Vec<T, V>::ld(dy + base + (size_t)q * p.C, gv);
      Vec<T, V>::ld(y + base + (size_t)q * p.C, yv);
#pragma unroll
      for (int e = 0; e < V; ++e) { a0[e] += gv[e]; a1[e] += gv[e] * yv[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) { sh[0][e][threadIdx.x] = a0[e]; sh[1][e][threadIdx.x] = a1[e]; }
  __syncthreads();
  for (int half = p.PL >> 1; half > 0; half >>= 1) {        // tree over the pixel lanes (PL is a power of two)
    if (pl < half) {
      const int o = threadIdx.x + half * p.CG;
#pragma unroll
      for (int e = 0; e < V; ++e) { sh[0][e][threadIdx.x] += sh[0][e][o]; sh[1][e][threadIdx.x] += sh[1][e][o]; }
    }
    __syncthreads();
  }
  if (pl == 0 && cvalid) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float* o = part + (((size_t)b * p.S + s) * p.C + c0 + e) * 2;
      o[0] = sh[0][e][threadIdx.x]; o[1] = sh[1][e][threadIdx.x];
    }
  }
}

template <typename T, int V>
__global__ void instnorm_bwd_apply_kernel(const T* dy, const T* y, const float* rstd, const float* tot, T* dx, RedPlan p) {
  RED_THREAD_SETUP();
  if (!cvalid) return;
  float m0[V], m1[V], r[V];
  const float inv_n = 1.f / (float)p.HW;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float* o = tot + ((size_t)b * p.C + c0 + e) * 2;
    m0[e] = o[0] * inv_n; m1[e] = o[1] * inv_n; r[e] = rstd[(size_t)b * p.C + c0 + e];
  }
  _Pragma("unroll 4") for (int q = p0 + pl; q < p1; q += p.PL) {
    float gv[V], yv[V];
    Vec<T, V>::ld(dy + base + (size_t)q * p.C, gv);
    Vec<T, V>::ld(y + base + (size_t)q * p.C, yv);
#pragma unroll
    for (int e = 0; e < V; ++e) gv[e] = r[e] * (gv[e] - m0[e] - yv[e] * m1[e]);
    Vec<T, V>::st(dx + base + (size_t)q * p.C, gv);
  }
}

// ----------------------------------------------------------------------------------------------------
// Affine normalisation + activation on explicit per-(b,c) coefficients: the norm_fun / act_fun variants of ConvBlock
// (models.py:88-101, 249-281: BatchNorm2d / InstanceNorm2d(affine, running statistics) followed by LeakyReLU | ReLU | Swish | SELU).
//   forward   y = act(x * scale[b,c] + shift[b,c])           (scale = gamma * rstd, shift = beta - mean * gamma * rstd; NULL = 1 / 0)
//   backward  g = gy * act'(x * scale + shift);   sums[b,c] = {sum g, sum g * x}    (-> d beta, d gamma, the mean terms of dx)
//             gx = g * ca[b,c] + x * cb[b,c] + cc[b,c]
// Which statistics feed the coefficients (per sample / per batch / running) is the caller's arithmetic on [B,C] arrays
// (uegan_amd/ops.py: NormAct); the pre-activation is recomputed from x, never stored.
// ----------------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void affine_act_fwd_kernel(const T* x, T* y, const float* scale, const float* shift, int act, RedPlan p) {
  RED_THREAD_SETUP();
  if (!cvalid) return;
  float sc[V], sf[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    sc[e] = scale ? scale[(size_t)b * p.C + c0 + e] : 1.f;
    sf[e] = shift ? shift[(size_t)b * p.C + c0 + e] : 0.f;
  }
  _Pragma("unroll 4") for (int q = p0 + pl; q < p1; q += p.PL) {
    float v[V];
    Vec<T, V>::ld(x + base + (size_t)q * p.C, v);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = act_of_pre(v[e] * sc[e] + sf[e], act);
    Vec<T, V>::st(y + base + (size_t)q * p.C, v);
  }
}

template <typename T, int V>
__global__ void affine_act_bwd_partial_kernel(const T* gy, const T* x, const float* scale, const float* shift, int act, float* part, RedPlan p) {
  __shared__ float sh[2][V][256];
  RED_THREAD_SETUP();
  float a0[V], a1[V], sc[V], sf[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    a0[e] = 0.f; a1[e] = 0.f;
    sc[e] = (scale && cvalid) ? scale[(size_t)b * p.C + c0 + e] : 1.f;
    sf[e] = (shift && cvalid) ? shift[(size_t)b * p.C + c0 + e] : 0.f;
  }
  if (cvalid) {
    _Pragma("unroll 4") for (int q = p0 + pl; q < p1; q += p.PL) {
      float gv[V], xv[V];
      Vec<T, V>::ld(gy + base + (size_t)q * p.C, gv);
      Vec<T, V>::ld(x + base + (size_t)q * p.C, xv);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float g = gv[e] * act_grad_of_pre(xv[e] * sc[e] + sf[e], act);
        a0[e] += g; a1[e] += g * xv[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) { sh[0][e][threadIdx.x] = a0[e]; sh[1][e][threadIdx.x] = a1[e]; }
  __syncthreads();
  for (int half = p.PL >> 1; half > 0; half >>= 1) {
    if (pl < half) {
      const int o = threadIdx.x + half * p.CG;
#pragma unroll
      for (int e = 0; e < V; ++e) { sh[0][e][threadIdx.x] += sh[0][e][o]; sh[1][e][threadIdx.x] += sh[1][e][o]; }
    }
    __syncthreads();
  }
  if (pl == 0 && cvalid) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float* o = part + (((size_t)b * p.S + s) * p.C + c0 + e) * 2;
      o[0] = sh[0][e][threadIdx.x]; o[1] = sh[1][e][threadIdx.x];
    }
  }
}

template <typename T, int V>
__global__ void affine_act_bwd_apply_kernel(const T* gy, const T* x, const float* scale, const float* shift, int act, const float* ca,
                                            const float* cb, const float* cc, T* gx, RedPlan p) {
  RED_THREAD_SETUP();
  if (!cvalid) return;
  float sc[V], sf[V], ka[V], kb[V], kc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const size_t i = (size_t)b * p.C + c0 + e;
    sc[e] = scale ? scale[i] : 1.f; sf[e] = shift ? shift[i] : 0.f;
    ka[e] = ca ? ca[i] : 1.f; kb[e] = cb ? cb[i] : 0.f; kc[e] = cc ? cc[i] : 0.f;
  }
  _Pragma("unroll 4") for (int q = p0 + pl; q < p1; q += p.PL) {
    float gv[V], xv[V];
    Vec<T, V>::ld(gy + base + (size_t)q * p.C, gv);
    Vec<T, V>::ld(x + base + (size_t)q * p.C, xv);
#pragma unroll
    for (int e = 0; e < V; ++e) gv[e] = gv[e] * act_grad_of_pre(xv[e] * sc[e] + sf[e], act) * ka[e] + xv[e] * kb[e] + kc[e];
    Vec<T, V>::st(gx + base + (size_t)q * p.C, gv);
  }
}

// ----------------------------------------------------------------------------------------------------
// perceptual tap: weight * MSE(IN(x), IN(y)) and its gradient w.r.t. x
// ----------------------------------------------------------------------------------------------------
// sums over the block's pixel range, per (b,c): {sum (xh-yh)^2, sum (xh-yh), sum (xh-yh)*xh}
template <typename T, int V>
__global__ void percep_sums_kernel(const T* x, const T* y, const float* st, float* sums, RedPlan p) {
  __shared__ float sh[3][V][256];
  RED_THREAD_SETUP();
  float a0[V], a1[V], a2[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { a0[e] = 0.f; a1[e] = 0.f; a2[e] = 0.f; }
  if (cvalid) {
    // st = [mean_x | rstd_x | mean_y | rstd_y], each B*C
    const size_t bc = (size_t)p.B * p.C, o0 = (size_t)b * p.C + c0;
    float mx[V], rx[V], my[V], ry[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      mx[e] = st[o0 + e]; rx[e] = st[bc + o0 + e]; my[e] = st[2 * bc + o0 + e]; ry[e] = st[3 * bc + o0 + e];
    }
    _Pragma("unroll 4") for (int q = p0 + pl; q < p1; q += p.PL) {
      float xv[V], yv[V];
      Vec<T, V>::ld(x + base + (size_t)q * p.C, xv);
      Vec<T, V>::ld(y + base + (size_t)q * p.C, yv);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float xh = (xv[e] - mx[e]) * rx[e], yh = (yv[e] - my[e]) * ry[e];
        const float d = xh - yh;
        a0[e] += d * d;
        a1[e] += d;
        a2[e] += d * xh;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) { sh[0][e][threadIdx.x] = a0[e]; sh[1][e][threadIdx.x] = a1[e]; sh[2][e][threadIdx.x] = a2[e]; }
  __syncthreads();
  for (int half = p.PL >> 1; half > 0; half >>= 1) {        // tree over the pixel lanes
    if (pl < half) {
      const int o = threadIdx.x + half * p.CG;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        sh[0][e][threadIdx.x] += sh[0][e][o]; sh[1][e][threadIdx.x] += sh[1][e][o]; sh[2][e][threadIdx.x] += sh[2][e][o];
      }
    }
    __syncthreads();
  }
  if (pl == 0 && cvalid) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float* o = sums + (((size_t)b * p.S + s) * p.C + c0 + e) * 3;
      o[0] = sh[0][e][threadIdx.x]; o[1] = sh[1][e][threadIdx.x]; o[2] = sh[2][e][threadIdx.x];
    }
  }
}

// loss += weight * sum_{b,c} sum (xh-yh)^2 / nel   (ONE block: a fixed summation order; the taps add up in launch order)
__global__ void percep_loss_kernel(const float* tot, float weight, float* loss, RedPlan p) {
  __shared__ float red[16];
  const int total = p.B * p.C;
  float acc = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) acc += tot[(size_t)i * 3];
  acc = block_sum(acc, red);
  const float nel = (float)p.B * (float)p.HW * (float)p.C;
  if (threadIdx.x == 0) *loss += weight * acc / nel;
}

// gx = gscale * d(weight * MSE(IN(x), IN(y)))/dx
template <typename T, int V, bool RELU, bool ACC = false>      // RELU: the act argument is UEGAN_ACT_RELU (the only one the model uses), resolved at compile time
__global__ void percep_grad_kernel(const T* x, const T* y, const float* st, const float* tot, float weight, const float* gscale, T* gx,
                                   RedPlan p, int act) {       // ACC: gx += ... (the tap has a second consumer whose gradient is already in gx)
  RED_THREAD_SETUP();
  if (!cvalid) return;
  const float nel = (float)p.B * (float)p.HW * (float)p.C;
  // g = dL/dxh = k*(xh-yh), k = 2*weight*gscale/nel ; dx = rx*(g - mean(g) - xh*mean(g*xh))
  const float k = 2.f * weight * (gscale ? *gscale : 1.f) / nel, inv_n = 1.f / (float)p.HW;
  const size_t bc = (size_t)p.B * p.C, o0 = (size_t)b * p.C + c0;
  float mx[V], rx[V], my[V], ry[V], mg[V], mgx[V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    mx[e] = st[o0 + e]; rx[e] = st[bc + o0 + e]; my[e] = st[2 * bc + o0 + e]; ry[e] = st[3 * bc + o0 + e];
    mg[e] = k * tot[(o0 + e) * 3 + 1] * inv_n;
    mgx[e] = k * tot[(o0 + e) * 3 + 2] * inv_n;
  }
  _Pragma("unroll 4") for (int q = p0 + pl; q < p1; q += p.PL) {
    float xv[V], yv[V];
    Vec<T, V>::ld(x + base + (size_t)q * p.C, xv);
    Vec<T, V>::ld(y + base + (size_t)q * p.C, yv);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float xh = (xv[e] - mx[e]) * rx[e], yh = (yv[e] - my[e]) * ry[e];
      const float gv = rx[e] * (k * (xh - yh) - mg[e] - xh * mgx[e]);
      xv[e] = RELU ? (xv[e] > 0.f ? gv : 0.f) : gv * act_grad_from_out(xv[e], act);      // (act: x's producer's deferred act')
    }
    if (ACC) {
      float pv[V];
      Vec<T, V>::ld(gx + base + (size_t)q * p.C, pv);
#pragma unroll
      for (int e = 0; e < V; ++e) xv[e] += pv[e];
    }
    Vec<T, V>::st(gx + base + (size_t)q * p.C, xv);
  }
}

// ----------------------------------------------------------------------------------------------------
// relativistic average hinge (losses.py:348-362), all scales in one launch per stage
// ----------------------------------------------------------------------------------------------------
// Deterministic reductions: a multi-block reduction stage stores ONE partial sum per block and quantity (grid.x <= RB) and its consumer --
// the next kernel of the chain -- adds the partials in a fixed order (a 64-lane butterfly), so the result does not depend on the order
// in which the blocks ran (a float atomicAdd per block did, at rounding level, and the relativistic means feed the gradient).
constexpr int RB = 64;
// sum of the nb (<= 64) block partials of one quantity: call from ONE whole wave, result in every lane
__device__ __forceinline__ float fold_partials(const float* part, int nb) {
  const int lane = threadIdx.x & 63;
  return wave_sum(lane < nb ? part[lane] : 0.f);
}

struct RaArgs {
  const float* real[8];
  const float* fake[8];
  float* greal[8];
  float* gfake[8];
  long long n[8];
  float* tmp;     // [nscales][8]: {sum r, sum f, sum A, sum B, cnt A, cnt B, -, -}, then the block partials [nscales][6][RB]
  float* loss;
  int nscales, nbx;      // nbx: blocks (partials) per scale of the reduction stages
  float sgn;      // +1 discriminator, -1 generator
};
__device__ __forceinline__ float* ra_part(const RaArgs& a, int sc, int q) { return a.tmp + 8 * a.nscales + (sc * 6 + q) * RB; }
// the two means of a scale from the partials of rahinge_means_kernel (waves 0 / 1 fold one each); block x == 0 also files them in the slots
__device__ __forceinline__ void ra_fold_means(const RaArgs& a, int sc, float* sh, float& rsum, float& fsum) {
  const int wv = threadIdx.x >> 6;
  if (wv < 2) {
    const float v = fold_partials(ra_part(a, sc, wv), a.nbx);
    if ((threadIdx.x & 63) == 0) {
      sh[wv] = v;
      if (blockIdx.x == 0) a.tmp[sc * 8 + wv] = v;
    }
  }
  __syncthreads();
  rsum = sh[0]; fsum = sh[1];
  __syncthreads();
}

__global__ void rahinge_means_kernel(RaArgs a) {
  __shared__ float red[16];
  const int sc = blockIdx.y;
  const long long n = a.n[sc];
  float sr = 0.f, sf = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    sr += a.real[sc][i];
    sf += a.fake[sc][i];
  }
  sr = block_sum(sr, red);
  sf = block_sum(sf, red);
  if (threadIdx.x == 0) {
    ra_part(a, sc, 0)[blockIdx.x] = sr;
    ra_part(a, sc, 1)[blockIdx.x] = sf;
  }
}

__global__ void rahinge_terms_kernel(RaArgs a) {
  __shared__ float red[16];
  const int sc = blockIdx.y;
  const long long n = a.n[sc];
  float rsum, fsum;
  ra_fold_means(a, sc, red, rsum, fsum);
  const float rbar = rsum / (float)n, fbar = fsum / (float)n;
  float sa = 0.f, sb = 0.f, ca = 0.f, cb = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float A = 1.f - a.sgn * (a.real[sc][i] - fbar);
    const float Bv = 1.f + a.sgn * (a.fake[sc][i] - rbar);
    if (A > 0.f) { sa += A; ca += 1.f; }
    if (Bv > 0.f) { sb += Bv; cb += 1.f; }
  }
  sa = block_sum(sa, red);
  sb = block_sum(sb, red);
  ca = block_sum(ca, red);
  cb = block_sum(cb, red);
  if (threadIdx.x == 0) {
    ra_part(a, sc, 2)[blockIdx.x] = sa;
    ra_part(a, sc, 3)[blockIdx.x] = sb;
    ra_part(a, sc, 4)[blockIdx.x] = ca;
    ra_part(a, sc, 5)[blockIdx.x] = cb;
  }
}

// one wave: folds the term partials of every scale into the slots (the gradient kernels read them there) and adds up the loss
__global__ void rahinge_loss_kernel(RaArgs a) {
  float L = 0.f;
  for (int k = 0; k < a.nscales; ++k) {
    const float nk = (float)a.n[k];
    float t[4];
    for (int q = 0; q < 4; ++q) {
      t[q] = fold_partials(ra_part(a, k, 2 + q), a.nbx);
      if (threadIdx.x == 0) a.tmp[k * 8 + 2 + q] = t[q];
    }
    L += 0.5f * (t[0] / nk + t[1] / nk);
  }
  if (threadIdx.x == 0) *a.loss = L;
}

// d loss / d real_i = -(sgn/2n) (1[A_i>0] + cntB/n) ; d loss / d fake_j = (sgn/2n) (1[B_j>0] + cntA/n), times gscale
__global__ void rahinge_grad_kernel(RaArgs a, const float* gscale) {
  const int sc = blockIdx.y;
  const long long n = a.n[sc];
  const float fn = (float)n;
  const float rbar = a.tmp[sc * 8 + 0] / fn, fbar = a.tmp[sc * 8 + 1] / fn;
  const float ca = a.tmp[sc * 8 + 4] / fn, cb = a.tmp[sc * 8 + 5] / fn;
  const float k = a.sgn * 0.5f / fn * (gscale ? *gscale : 1.f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (a.greal[sc]) {
      const float A = 1.f - a.sgn * (a.real[sc][i] - fbar);
      a.greal[sc][i] = -k * ((A > 0.f ? 1.f : 0.f) + cb);
    }
    if (a.gfake[sc]) {
      const float Bv = 1.f + a.sgn * (a.fake[sc][i] - rbar);
      a.gfake[sc][i] = k * ((Bv > 0.f ? 1.f : 0.f) + ca);
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// The other adversarial losses of GANLoss.loss (losses.py:312-392).  'rals' (relativistic average least squares, :363-376) shares
// the two-stage shape of 'rahinge' (means, then terms); the non-relativistic modes ('original' :313-323, 'ls' :324-332, 'hinge'
// :333-347, the wgan fallback :378-392) are a mean of an elementwise function of ONE prediction list.
// ----------------------------------------------------------------------------------------------------
// A_i = (r_i - fbar) - sgn, B_j = (f_j - rbar) + sgn; loss = sum_scales (mean A^2 + mean B^2) / 2
__global__ void rals_terms_kernel(RaArgs a) {
  __shared__ float red[16];
  const int sc = blockIdx.y;
  const long long n = a.n[sc];
  float rsum, fsum;
  ra_fold_means(a, sc, red, rsum, fsum);
  const float rbar = rsum / (float)n, fbar = fsum / (float)n;
  float sa = 0.f, sb = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float A = a.real[sc][i] - fbar - a.sgn;
    const float Bv = a.fake[sc][i] - rbar + a.sgn;
    sa += A * A;
    sb += Bv * Bv;
  }
  sa = block_sum(sa, red);
  sb = block_sum(sb, red);
  if (threadIdx.x == 0) {
    ra_part(a, sc, 2)[blockIdx.x] = sa;
    ra_part(a, sc, 3)[blockIdx.x] = sb;
    ra_part(a, sc, 4)[blockIdx.x] = 0.f;      // (the shared loss kernel folds four quantities)
    ra_part(a, sc, 5)[blockIdx.x] = 0.f;
  }
}
// d loss / d r_i = (A_i - mean B) / n,  d loss / d f_j = (B_j - mean A) / n   (mean A = rbar - fbar - sgn, mean B = fbar - rbar + sgn)
__global__ void rals_grad_kernel(RaArgs a, const float* gscale) {
  const int sc = blockIdx.y;
  const long long n = a.n[sc];
  const float fn = (float)n;
  const float rbar = a.tmp[sc * 8 + 0] / fn, fbar = a.tmp[sc * 8 + 1] / fn;
  const float ma = rbar - fbar - a.sgn, mb = fbar - rbar + a.sgn;
  const float k = (gscale ? *gscale : 1.f) / fn;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (a.greal[sc]) a.greal[sc][i] = k * ((a.real[sc][i] - fbar - a.sgn) - mb);
    if (a.gfake[sc]) a.gfake[sc][i] = k * ((a.fake[sc][i] - rbar + a.sgn) - ma);
  }
}

struct PredArgs {
  const float* p[8];
  float* g[8];
  long long n[8];
  float* tmp;      // [nscales][RB] block partials
  float* loss;
  int nscales, fid, nbx;
  float target;
};
__device__ __forceinline__ float pred_term(float p, int fid, float t) {
  switch (fid) {
    case UEGAN_PRED_BCE: return fmaxf(p, 0.f) - p * t + log1pf(expf(-fabsf(p)));      // binary_cross_entropy_with_logits
    case UEGAN_PRED_LS: return (p - t) * (p - t);
    case UEGAN_PRED_HINGE_REAL: return -fminf(p - 1.f, 0.f);
    case UEGAN_PRED_HINGE_FAKE: return -fminf(-p - 1.f, 0.f);
    case UEGAN_PRED_NEG_MEAN: return -p;
    default: return p;
  }
}
__device__ __forceinline__ float pred_term_grad(float p, int fid, float t) {
  switch (fid) {
    case UEGAN_PRED_BCE: return 1.f / (1.f + expf(-p)) - t;
    case UEGAN_PRED_LS: return 2.f * (p - t);
    case UEGAN_PRED_HINGE_REAL: return p - 1.f < 0.f ? -1.f : 0.f;
    case UEGAN_PRED_HINGE_FAKE: return -p - 1.f < 0.f ? 1.f : 0.f;
    case UEGAN_PRED_NEG_MEAN: return -1.f;
    default: return 1.f;
  }
}
__global__ void pred_terms_kernel(PredArgs a) {
  __shared__ float red[16];
  const int sc = blockIdx.y;
  const long long n = a.n[sc];
  float sa = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) sa += pred_term(a.p[sc][i], a.fid, a.target);
  sa = block_sum(sa, red);
  if (threadIdx.x == 0) a.tmp[sc * RB + blockIdx.x] = sa;
}
__global__ void pred_loss_kernel(PredArgs a) {      // one wave
  float L = 0.f;
  for (int k = 0; k < a.nscales; ++k) L += fold_partials(a.tmp + k * RB, a.nbx) / (float)a.n[k];
  if (threadIdx.x == 0) *a.loss = L;
}
__global__ void pred_grad_kernel(PredArgs a, const float* gscale) {
  const int sc = blockIdx.y;
  const long long n = a.n[sc];
  const float k = (gscale ? *gscale : 1.f) / (float)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    a.g[sc][i] = k * pred_term_grad(a.p[sc][i], a.fid, a.target);
}

// ----------------------------------------------------------------------------------------------------
// The same loss read straight off the prediction-head maps of a BATCHED discriminator pass (uegan_amd/fused.py): the maps are
// NHWC with channel 0 = tanh output (the other channels of the 16-byte chunk are padding), image groups of nb images lie one
// after the other in the batch, and the loss is a sum over (real group, fake group) pairs -- trainer.py:92+95 is
// {(exp, fake_store), (exp, raw)}, :104 is {(exp, fake)}.  The gradient comes back in the same layout already multiplied by
// tanh'(P) = 1 - P^2, i.e. it IS the head convolution's pre-activation gradient.
// ----------------------------------------------------------------------------------------------------
constexpr int RH_MAXG = 4, RH_MAXP = 4;
struct RaHeadArgs {
  const void* maps[8];
  void* gmaps[8];
  long long npg[8];       // prediction pixels per group (nb * h * w) of each scale
  int pr[RH_MAXP], pf[RH_MAXP];
  float* tmp;             // [nscales][RH_MAXG] group sums, then [nscales][RH_MAXP][4] {sum A, sum B, cnt A, cnt B}, then the block
                          // partials of both: [nscales][RH_MAXG][RB], [nscales][RH_MAXP][4][RB]
  float* loss;
  int nscales, ngroups, npairs, cp, nbx;
  unsigned gmask;         // groups whose gradient is wanted
  float sgn;
};

__device__ __forceinline__ float* rh_gpart(const RaHeadArgs& a, int sc, int g) {
  return a.tmp + a.nscales * (RH_MAXG + RH_MAXP * 4) + (sc * RH_MAXG + g) * RB;
}
__device__ __forceinline__ float* rh_ppart(const RaHeadArgs& a, int sc, int pi, int q) {
  return a.tmp + a.nscales * (RH_MAXG + RH_MAXP * 4) + a.nscales * RH_MAXG * RB + ((sc * RH_MAXP + pi) * 4 + q) * RB;
}

template <typename T>
__global__ void rahead_means_kernel(RaHeadArgs a) {
  __shared__ float red[16];
  const int sc = blockIdx.y, g = blockIdx.z;
  const long long n = a.npg[sc];
  const T* p = static_cast<const T*>(a.maps[sc]) + (size_t)g * n * a.cp;
  float sm = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) sm += DT<T>::ld(p + i * a.cp);
  sm = block_sum(sm, red);
  if (threadIdx.x == 0) rh_gpart(a, sc, g)[blockIdx.x] = sm;
}

template <typename T>
__global__ void rahead_terms_kernel(RaHeadArgs a) {
  __shared__ float red[16];
  const int sc = blockIdx.y, pi = blockIdx.z;
  const long long n = a.npg[sc];
  const int gr = a.pr[pi], gf = a.pf[pi];
  const T* pr = static_cast<const T*>(a.maps[sc]) + (size_t)gr * n * a.cp;
  const T* pf = static_cast<const T*>(a.maps[sc]) + (size_t)gf * n * a.cp;
  {      // the two group sums from the partials of rahead_means_kernel (waves 0 / 1); block x == 0 files them in the slots
    const int wv = threadIdx.x >> 6;
    if (wv < 2) {
      const int g = wv == 0 ? gr : gf;
      const float v = fold_partials(rh_gpart(a, sc, g), a.nbx);
      if ((threadIdx.x & 63) == 0) {
        red[wv] = v;
        if (blockIdx.x == 0) a.tmp[sc * RH_MAXG + g] = v;      // (pairs sharing a group store the same value)
      }
    }
    __syncthreads();
  }
  const float rbar = red[0] / (float)n, fbar = red[1] / (float)n;
  __syncthreads();
  float sa = 0.f, sb = 0.f, ca = 0.f, cb = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float A = 1.f - a.sgn * (DT<T>::ld(pr + i * a.cp) - fbar);
    const float Bv = 1.f + a.sgn * (DT<T>::ld(pf + i * a.cp) - rbar);
    if (A > 0.f) { sa += A; ca += 1.f; }
    if (Bv > 0.f) { sb += Bv; cb += 1.f; }
  }
  sa = block_sum(sa, red);
  sb = block_sum(sb, red);
  ca = block_sum(ca, red);
  cb = block_sum(cb, red);
  if (threadIdx.x == 0) {
    rh_ppart(a, sc, pi, 0)[blockIdx.x] = sa; rh_ppart(a, sc, pi, 1)[blockIdx.x] = sb;
    rh_ppart(a, sc, pi, 2)[blockIdx.x] = ca; rh_ppart(a, sc, pi, 3)[blockIdx.x] = cb;
  }
}

// one wave: folds the pair-term partials into the slots (the gradient kernel reads them there) and adds up the loss
__global__ void rahead_loss_kernel(RaHeadArgs a) {
  float Ltot = 0.f;
  for (int pi = 0; pi < a.npairs; ++pi)        // pair-major, scale-minor: the order trainer.py:92,95 adds the two GANLoss calls
    for (int k = 0; k < a.nscales; ++k) {
      float* o = a.tmp + a.nscales * RH_MAXG + (k * RH_MAXP + pi) * 4;
      float t[4];
      for (int q = 0; q < 4; ++q) {
        t[q] = fold_partials(rh_ppart(a, k, pi, q), a.nbx);
        if (threadIdx.x == 0) o[q] = t[q];
      }
      const float nk = (float)a.npg[k];
      Ltot += 0.5f * (t[0] / nk + t[1] / nk);
    }
  if (threadIdx.x == 0) *a.loss = Ltot;
}

// one thread per prediction pixel: dP summed over the pairs the pixel's group takes part in, times tanh'(P); one 16-byte (bf16) /
// two-chunk (fp32, cp = 4: one chunk) store with zeros in the padding channels
template <typename T>
__global__ void rahead_grad_kernel(RaHeadArgs a, const float* gscale) {
  const int sc = blockIdx.y, g = blockIdx.z;
  if (!((a.gmask >> g) & 1u)) return;
  const long long n = a.npg[sc];
  const float fn = (float)n;
  const T* p = static_cast<const T*>(a.maps[sc]) + (size_t)g * n * a.cp;
  T* o = static_cast<T*>(a.gmaps[sc]) + (size_t)g * n * a.cp;
  const float gs = a.sgn * 0.5f / fn * (gscale ? *gscale : 1.f);
  // per pair this group is in: threshold mean and the count term
  float bar[RH_MAXP], cnt[RH_MAXP];
  int role[RH_MAXP];       // 0: not in the pair, 1: real, 2: fake
#pragma unroll
  for (int pi = 0; pi < RH_MAXP; ++pi) {
    role[pi] = 0; bar[pi] = 0.f; cnt[pi] = 0.f;
    if (pi < a.npairs) {
      const float* t = a.tmp + a.nscales * RH_MAXG + (sc * RH_MAXP + pi) * 4;
      if (a.pr[pi] == g) { role[pi] = 1; bar[pi] = a.tmp[sc * RH_MAXG + a.pf[pi]] / fn; cnt[pi] = t[3] / fn; }
      else if (a.pf[pi] == g) { role[pi] = 2; bar[pi] = a.tmp[sc * RH_MAXG + a.pr[pi]] / fn; cnt[pi] = t[2] / fn; }
    }
  }
  constexpr int EPC = DT<T>::EPC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float P = DT<T>::ld(p + i * a.cp);
    float d = 0.f;
#pragma unroll
    for (int pi = 0; pi < RH_MAXP; ++pi) {
      if (role[pi] == 1) {
        const float A = 1.f - a.sgn * (P - bar[pi]);
        d -= gs * ((A > 0.f ? 1.f : 0.f) + cnt[pi]);
      } else if (role[pi] == 2) {
        const float Bv = 1.f + a.sgn * (P - bar[pi]);
        d += gs * ((Bv > 0.f ? 1.f : 0.f) + cnt[pi]);
      }
    }
    float v[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) v[e] = 0.f;
    v[0] = d * (1.f - P * P);
    for (int c0 = 0; c0 < a.cp; c0 += EPC) {
      Vec<T, EPC>::st(o + i * a.cp + c0, v);
      v[0] = 0.f;
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// MultiscaleRecLoss (losses.py:202-231): criterion at `nscales` scales with AvgPool2d(2,2) between, weights 1, 1/2, 1/4.
// KIND 0 L1Loss, 1 SmoothL1Loss (beta = 1), 2 MSELoss.  One thread per 4x4 block of one channel plane; per-block partial sums, added up
// in a fixed order by msrec_final_kernel (deterministic).
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }
template <int KIND> __device__ __forceinline__ float rec_term(float d) {
  if (KIND == 0) return fabsf(d);
  if (KIND == 1) { const float ad = fabsf(d); return ad < 1.f ? 0.5f * d * d : ad - 0.5f; }
  return d * d;
}
template <int KIND> __device__ __forceinline__ float rec_grad(float d) {
  if (KIND == 0) return sgnf(d);
  if (KIND == 1) return fabsf(d) < 1.f ? d : sgnf(d);
  return 2.f * d;
}

template <int KIND>
__global__ void msrec_kernel(const float* pred, const float* gt, float* part, float* gpred, const float* gscale, int planes, int H, int W, int nscales) {
  __shared__ float red[16];
  const int bw = W / 4, bh = H / 4;
  const size_t total = (size_t)planes * bh * bw;
  const float n0 = (float)planes * (float)H * (float)W;
  const float c0 = 1.f / n0, c1 = nscales > 1 ? 0.5f / (n0 / 4.f) : 0.f, c2 = nscales > 2 ? 0.25f / (n0 / 16.f) : 0.f;
  const float gs = (gpred && gscale) ? *gscale : 1.f;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int bx = (int)(i % bw);
    size_t t = i / bw;
    const int by = (int)(t % bh);
    const size_t pl = t / bh;
    const size_t base = (pl * H + (size_t)by * 4) * W + (size_t)bx * 4;
    float d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(pred + base + (size_t)r * W);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gt + base + (size_t)r * W);
      d[r][0] = pv.x - gv.x; d[r][1] = pv.y - gv.y; d[r][2] = pv.z - gv.z; d[r][3] = pv.w - gv.w;
    }
    float d1[2][2], l0 = 0.f, l1 = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        d1[r][c] = 0.25f * (d[2 * r][2 * c] + d[2 * r][2 * c + 1] + d[2 * r + 1][2 * c] + d[2 * r + 1][2 * c + 1]);
        l1 += rec_term<KIND>(d1[r][c]);
      }
    const float d2 = 0.25f * (d1[0][0] + d1[0][1] + d1[1][0] + d1[1][1]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) l0 += rec_term<KIND>(d[r][c]);
    acc += c0 * l0 + c1 * l1 + c2 * rec_term<KIND>(d2);
    if (gpred) {
      const float g2 = gs * c2 * rec_grad<KIND>(d2) * (1.f / 16.f);
      const float g0 = gs * c0, g1 = gs * c1 * 0.25f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 o;
        o.x = g0 * rec_grad<KIND>(d[r][0]) + g1 * rec_grad<KIND>(d1[r / 2][0]) + g2;
        o.y = g0 * rec_grad<KIND>(d[r][1]) + g1 * rec_grad<KIND>(d1[r / 2][0]) + g2;
        o.z = g0 * rec_grad<KIND>(d[r][2]) + g1 * rec_grad<KIND>(d1[r / 2][1]) + g2;
        o.w = g0 * rec_grad<KIND>(d[r][3]) + g1 * rec_grad<KIND>(d1[r / 2][1]) + g2;
        *reinterpret_cast<f32x4*>(gpred + base + (size_t)r * W) = o;
      }
    }
  }
  acc = block_sum(acc, red);
  if (part && threadIdx.x == 0) part[blockIdx.x] = acc;
}

// any H x W (AvgPool2d(2, 2) floors: a last odd row / column does not reach the next scale, losses.py:225-227): one thread per 4 x 4
// block of the ceil grid, scalar accesses, per-element validity.  The denominators are the element counts of the floored maps.
template <int KIND>
__global__ void msrec_ragged_kernel(const float* pred, const float* gt, float* part, float* gpred, const float* gscale, int planes, int H, int W,
                                    int nscales) {
  __shared__ float red[16];
  const int bw = (W + 3) / 4, bh = (H + 3) / 4;
  const int H1 = H / 2, W1 = W / 2, H2 = H1 / 2, W2 = W1 / 2;
  const size_t total = (size_t)planes * bh * bw;
  const float c0 = 1.f / ((float)planes * (float)H * (float)W);
  const float c1 = nscales > 1 ? 0.5f / ((float)planes * (float)H1 * (float)W1) : 0.f;
  const float c2 = nscales > 2 ? 0.25f / ((float)planes * (float)H2 * (float)W2) : 0.f;
  const float gs = (gpred && gscale) ? *gscale : 1.f;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int bx = (int)(i % bw);
    size_t t = i / bw;
    const int by = (int)(t % bh);
    const size_t pl = t / bh;
    const size_t base = (pl * H + (size_t)by * 4) * W + (size_t)bx * 4;
    float d[4][4];
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = by * 4 + r < H && bx * 4 + c < W;
        d[r][c] = ok ? pred[base + (size_t)r * W + c] - gt[base + (size_t)r * W + c] : 0.f;
        l0 += ok ? rec_term<KIND>(d[r][c]) : 0.f;
      }
    float d1[2][2];
    bool ok1[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        ok1[r][c] = nscales > 1 && by * 2 + r < H1 && bx * 2 + c < W1;
        d1[r][c] = 0.25f * (d[2 * r][2 * c] + d[2 * r][2 * c + 1] + d[2 * r + 1][2 * c] + d[2 * r + 1][2 * c + 1]);
        l1 += ok1[r][c] ? rec_term<KIND>(d1[r][c]) : 0.f;
      }
    const bool ok2 = nscales > 2 && by < H2 && bx < W2;
    const float d2 = 0.25f * (d1[0][0] + d1[0][1] + d1[1][0] + d1[1][1]);
    l2 = ok2 ? rec_term<KIND>(d2) : 0.f;
    acc += c0 * l0 + c1 * l1 + c2 * l2;
    if (gpred) {
      const float g2 = ok2 ? gs * c2 * rec_grad<KIND>(d2) * (1.f / 16.f) : 0.f;
      const float g0 = gs * c0, g1 = gs * c1 * 0.25f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (by * 4 + r < H && bx * 4 + c < W)
            gpred[base + (size_t)r * W + c] = g0 * rec_grad<KIND>(d[r][c]) + (ok1[r / 2][c / 2] ? g1 * rec_grad<KIND>(d1[r / 2][c / 2]) : 0.f) + g2;
    }
  }
  acc = block_sum(acc, red);
  if (part && threadIdx.x == 0) part[blockIdx.x] = acc;
}

// single scale (multiscale=False, or scale=1), any H x W: mean criterion(pred - gt)
template <int KIND>
__global__ void rec_flat_kernel(const float* pred, const float* gt, float* part, float* gpred, const float* gscale, size_t n) {
  __shared__ float red[16];
  const float c0 = 1.f / (float)n, gs = (gpred && gscale) ? *gscale : 1.f;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = pred[i] - gt[i];
    acc += c0 * rec_term<KIND>(d);
    if (gpred) gpred[i] = gs * c0 * rec_grad<KIND>(d);
  }
  acc = block_sum(acc, red);
  if (part && threadIdx.x == 0) part[blockIdx.x] = acc;
}

constexpr int MSREC_MAXB = 2048;
__global__ void msrec_final_kernel(const float* part, int nb, float* loss) {      // one block: fixed summation order
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += part[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) *loss = acc;
}

}  // namespace uegan

using namespace uegan;

// binds T (storage type) and V (channels per thread: one 16-byte chunk when the plan allows it, else 1)
#define DISPATCH_TV(dtype, vec, ...)                                                                  \
  do {                                                                                                \
    if ((dtype) == UEGAN_F32) { using T = float; if ((vec) == 1) { constexpr int V = 1; __VA_ARGS__; } else { constexpr int V = 4; __VA_ARGS__; } } \
    else if ((dtype) == UEGAN_BF16) { using T = bf16_t; if ((vec) == 1) { constexpr int V = 1; __VA_ARGS__; } else { constexpr int V = 8; __VA_ARGS__; } } \
    else { set_error("bad dtype %d", (int)(dtype)); return UEGAN_E_INVALID; }                        \
  } while (0)
static inline int epc_of(int dtype) { return dtype == UEGAN_BF16 ? 8 : 4; }
#define DISPATCH_T(dtype, ...)                                   \
  do {                                                           \
    if ((dtype) == UEGAN_F32) { using T = float; __VA_ARGS__; }  \
    else if ((dtype) == UEGAN_BF16) { using T = bf16_t; __VA_ARGS__; } \
    else { set_error("bad dtype %d", (int)(dtype)); return UEGAN_E_INVALID; } \
  } while (0)

// scratch per reduction pass: split partials (3 per (b,s,c)) + 8 floats per (b,c) for finalized statistics / totals
extern "C" size_t uegan_reduce_workspace_floats(int B, int HW, int C) {
  const RedPlan p4 = make_plan(B, HW, C, 4), p8 = make_plan(B, HW, C, 8);      // (the split count depends on the storage type's chunk width)
  const int S = p4.S > p8.S ? p4.S : p8.S;
  return (size_t)B * S * C * 3 + (size_t)B * C * 8;
}

static inline int bc_blocks(const RedPlan& p, int K) { return (p.B * p.C * K + 3) / 4; }      // one wave per (b, c, k): 4 per 256-thread block

extern "C" int uegan_instnorm_fwd(int dtype, const void* x, void* y, float* mean, float* rstd, float* tmp, int B, int HW, int C, float eps,
                                  uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && mean && rstd && tmp && B > 0 && HW > 0 && C > 0, "bad instnorm args");
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  dim3 grid(p.S, p.ncg, B);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((moments_partial_kernel<T, V>), grid, dim3(256), 0, s, (const T*)x, tmp, p));
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(moments_finalize_kernel, dim3(bc_blocks(p, 1)), dim3(256), 0, s, tmp, mean, rstd, p, eps);
  UEGAN_CHECK_LAUNCH();
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((instnorm_apply_kernel<T, V>), grid, dim3(256), 0, s, (const T*)x, (T*)y, mean, rstd, p));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_instnorm_apply(int dtype, const void* x, void* y, const float* mean, const float* rstd, int B, int HW, int C, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && mean && rstd && B > 0 && HW > 0 && C > 0, "bad instnorm args");
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  dim3 grid(p.S, p.ncg, B);
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((instnorm_apply_kernel<T, V>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, mean, rstd, p));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_instnorm_apply_pair(int dtype, const void* x, const void* x_lo, void* y, void* y_lo, const float* mean, const float* rstd, int B, int HW,
                                         int C, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && x_lo && y && y_lo && mean && rstd && B > 0 && HW > 0 && C > 0, "bad instnorm args");
  UEGAN_CHECK_ARG(dtype == UEGAN_BF16, "hi + lo pairs exist for the 16-bit storage format");
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  dim3 grid(p.S, p.ncg, B);
  if (p.V == 8) hipLaunchKernelGGL((instnorm_apply_pair_kernel<bf16_t, 8>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)x_lo, (bf16_t*)y,
                                   (bf16_t*)y_lo, mean, rstd, p);
  else hipLaunchKernelGGL((instnorm_apply_pair_kernel<bf16_t, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)x_lo, (bf16_t*)y,
                          (bf16_t*)y_lo, mean, rstd, p);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_instnorm_bwd(int dtype, const void* dy, const void* y, const float* rstd, void* dx, float* tmp, int B, int HW, int C,
                                  uegan_stream_t stream) {
  UEGAN_CHECK_ARG(dy && y && rstd && dx && tmp && B > 0 && HW > 0 && C > 0, "bad instnorm args");
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  dim3 grid(p.S, p.ncg, B);
  hipStream_t s = (hipStream_t)stream;
  float* tot = tmp + (size_t)B * p.S * C * 3;
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((instnorm_bwd_partial_kernel<T, V>), grid, dim3(256), 0, s, (const T*)dy, (const T*)y, tmp, p));
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sums_finalize_kernel, dim3(bc_blocks(p, 2)), dim3(256), 0, s, tmp, tot, p, 2);
  UEGAN_CHECK_LAUNCH();
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((instnorm_bwd_apply_kernel<T, V>), grid, dim3(256), 0, s, (const T*)dy, (const T*)y, rstd, tot, (T*)dx, p));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_moments(int dtype, const void* x, float* mean, float* var, float* tmp, int B, int HW, int C, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && mean && var && tmp && B > 0 && HW > 0 && C > 0, "bad moments args");
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((moments_partial_kernel<T, V>), dim3(p.S, p.ncg, B), dim3(256), 0, s, (const T*)x, tmp, p));
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(moments_finalize_kernel, dim3(bc_blocks(p, 1)), dim3(256), 0, s, tmp, mean, var, p, -1.f);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_affine_act_fwd(int dtype, int act, const void* x, const float* scale, const float* shift, void* y, int B, int HW, int C,
                                    uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && B > 0 && HW > 0 && C > 0 && act >= UEGAN_ACT_NONE && act <= UEGAN_ACT_SELU, "bad affine_act args");
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((affine_act_fwd_kernel<T, V>), dim3(p.S, p.ncg, B), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                             (T*)y, scale, shift, act, p));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_affine_act_bwd_sums(int dtype, int act, const void* gy, const void* x, const float* scale, const float* shift, float* sums,
                                         float* tmp, int B, int HW, int C, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(gy && x && sums && tmp && B > 0 && HW > 0 && C > 0 && act >= UEGAN_ACT_NONE && act <= UEGAN_ACT_SELU, "bad affine_act args");
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((affine_act_bwd_partial_kernel<T, V>), dim3(p.S, p.ncg, B), dim3(256), 0, s, (const T*)gy, (const T*)x,
                                             scale, shift, act, tmp, p));
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sums_finalize_kernel, dim3(bc_blocks(p, 2)), dim3(256), 0, s, tmp, sums, p, 2);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_affine_act_bwd_apply(int dtype, int act, const void* gy, const void* x, const float* scale, const float* shift,
                                          const float* ca, const float* cb, const float* cc, void* gx, int B, int HW, int C, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(gy && x && gx && B > 0 && HW > 0 && C > 0 && act >= UEGAN_ACT_NONE && act <= UEGAN_ACT_SELU, "bad affine_act args");
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((affine_act_bwd_apply_kernel<T, V>), dim3(p.S, p.ncg, B), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)gy, (const T*)x, scale, shift, act, ca, cb, cc, (T*)gx, p));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

// percep scratch (3 reduction workspaces): region 0 = x partials + [mean_x|rstd_x|mean_y|rstd_y] (4*B*C in its 8*B*C tail),
// region 1 = y partials, region 2 = sum partials + totals (3*B*C in its tail)
static void percep_layout(const RedPlan& p, float* tmp, float*& px, float*& py, float*& sums, float*& st, float*& tot) {
  const size_t part = (size_t)p.B * p.S * p.C * 3, region = part + (size_t)p.B * p.C * 8;
  px = tmp; py = tmp + region; sums = tmp + 2 * region;
  st = tmp + part;
  tot = tmp + 2 * region + part;
}

extern "C" int uegan_percep_tap_fwd(int dtype, const void* x, const void* y, float weight, float* loss, float* tmp, int B, int HW, int C,
                                    float eps, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && loss && tmp && B > 0 && HW > 0 && C > 0, "bad percep args");
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  dim3 grid(p.S, p.ncg, B);
  hipStream_t s = (hipStream_t)stream;
  float *px, *py, *sums, *st, *tot;
  percep_layout(p, tmp, px, py, sums, st, tot);
  const size_t bc = (size_t)B * C;
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((moments_partial_kernel<T, V>), grid, dim3(256), 0, s, (const T*)x, px, p));
  UEGAN_CHECK_LAUNCH();
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((moments_partial_kernel<T, V>), grid, dim3(256), 0, s, (const T*)y, py, p));
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(moments_finalize_kernel, dim3(bc_blocks(p, 1)), dim3(256), 0, s, px, st, st + bc, p, eps);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(moments_finalize_kernel, dim3(bc_blocks(p, 1)), dim3(256), 0, s, py, st + 2 * bc, st + 3 * bc, p, eps);
  UEGAN_CHECK_LAUNCH();
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((percep_sums_kernel<T, V>), grid, dim3(256), 0, s, (const T*)x, (const T*)y, st, sums, p));
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sums_finalize_kernel, dim3(bc_blocks(p, 3)), dim3(256), 0, s, sums, tot, p, 3);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(percep_loss_kernel, dim3(1), dim3(1024), 0, s, tot, weight, loss, p);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

// ... with the InstanceNorm moments of both taps GIVEN (uegan_conv2d_fwd_stats emitted them from the tap conv's epilogue: VGG conv1_1, the 1-GB
// tap): the four moment launches become one copy of [mean_x | rstd_x | mean_y | rstd_y] into the scratch the backward reads them from
__global__ void percep_stats_copy_kernel(const float* mx, const float* rx, const float* my, const float* ry, float* st, int bc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * bc) return;
  const int k = i / bc, j = i - k * bc;
  st[i] = k == 0 ? mx[j] : (k == 1 ? rx[j] : (k == 2 ? my[j] : ry[j]));
}
extern "C" int uegan_percep_tap_fwd_given(int dtype, const void* x, const void* y, float weight, float* loss, float* tmp, int B, int HW, int C,
                                          const float* mean_x, const float* rstd_x, const float* mean_y, const float* rstd_y, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && loss && tmp && mean_x && rstd_x && mean_y && rstd_y && B > 0 && HW > 0 && C > 0, "bad percep args");
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  dim3 grid(p.S, p.ncg, B);
  hipStream_t s = (hipStream_t)stream;
  float *px, *py, *sums, *st, *tot;
  percep_layout(p, tmp, px, py, sums, st, tot);
  const int bc = B * C;
  hipLaunchKernelGGL(percep_stats_copy_kernel, dim3((4 * bc + 255) / 256), dim3(256), 0, s, mean_x, rstd_x, mean_y, rstd_y, st, bc);
  UEGAN_CHECK_LAUNCH();
  DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((percep_sums_kernel<T, V>), grid, dim3(256), 0, s, (const T*)x, (const T*)y, st, sums, p));
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(sums_finalize_kernel, dim3(bc_blocks(p, 3)), dim3(256), 0, s, sums, tot, p, 3);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(percep_loss_kernel, dim3(1), dim3(1024), 0, s, tot, weight, loss, p);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_percep_tap_bwd(int dtype, const void* x, const void* y, float weight, const float* gscale, void* gx, const float* tmp,
                                    int B, int HW, int C, float eps, uegan_stream_t stream) {
  return uegan_percep_tap_bwd_act(dtype, UEGAN_ACT_NONE, x, y, weight, gscale, gx, tmp, B, HW, C, eps, stream);
}
extern "C" int uegan_percep_tap_bwd_act(int dtype, int act, const void* x, const void* y, float weight, const float* gscale, void* gx,
                                        const float* tmp, int B, int HW, int C, float eps, uegan_stream_t stream) {
  return uegan_percep_tap_bwd_acc(dtype, act, x, y, weight, gscale, gx, tmp, B, HW, C, eps, 0, stream);
}
extern "C" int uegan_percep_tap_bwd_acc(int dtype, int act, const void* x, const void* y, float weight, const float* gscale, void* gx,
                                        const float* tmp, int B, int HW, int C, float eps, int accumulate, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && gx && tmp && B > 0 && HW > 0 && C > 0, "bad percep args");
  (void)eps;
  RedPlan p = make_plan(B, HW, C, epc_of(dtype));
  dim3 grid(p.S, p.ncg, B);
  float *px, *py, *sums, *st, *tot;
  percep_layout(p, const_cast<float*>(tmp), px, py, sums, st, tot);
  if (accumulate) {
    if (act == UEGAN_ACT_RELU) {
      DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((percep_grad_kernel<T, V, true, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)y, st, tot, weight, gscale, (T*)gx, p, act));
    } else {
      DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((percep_grad_kernel<T, V, false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)y, st, tot, weight, gscale, (T*)gx, p, act));
    }
  } else if (act == UEGAN_ACT_RELU) {
    DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((percep_grad_kernel<T, V, true>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)y, st, tot, weight, gscale, (T*)gx, p, act));
  } else {
    DISPATCH_TV(dtype, p.V, hipLaunchKernelGGL((percep_grad_kernel<T, V, false>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)y, st, tot, weight, gscale, (T*)gx, p, act));
  }
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

static int ra_fill(RaArgs& a, int nscales, const float* const* real, const float* const* fake, const int64_t* n, int for_discriminator,
                   float* const* greal, float* const* gfake, float* tmp, long long& maxn) {
  UEGAN_CHECK_ARG(nscales >= 1 && nscales <= 8 && real && fake && n && tmp, "bad rahinge args");
  maxn = 0;
  for (int i = 0; i < 8; ++i) {
    a.real[i] = i < nscales ? real[i] : nullptr;
    a.fake[i] = i < nscales ? fake[i] : nullptr;
    a.greal[i] = (i < nscales && greal) ? greal[i] : nullptr;
    a.gfake[i] = (i < nscales && gfake) ? gfake[i] : nullptr;
    a.n[i] = i < nscales ? (long long)n[i] : 0;
    if (i < nscales) {
      UEGAN_CHECK_ARG(real[i] && fake[i] && n[i] > 0, "bad rahinge scale %d", i);
      if (a.n[i] > maxn) maxn = a.n[i];
    }
  }
  a.tmp = tmp; a.loss = nullptr; a.nscales = nscales; a.sgn = for_discriminator ? 1.f : -1.f;
  a.nbx = (int)((maxn + 1023) / 1024);
  if (a.nbx > RB) a.nbx = RB;
  if (a.nbx < 1) a.nbx = 1;
  return UEGAN_OK;
}

extern "C" size_t uegan_rahinge_workspace_floats(int nscales) { return (size_t)nscales * (8 + 6 * RB); }
extern "C" size_t uegan_pred_loss_workspace_floats(int nscales) { return (size_t)nscales * RB; }

extern "C" int uegan_rahinge_fwd(int nscales, const float* const* real, const float* const* fake, const int64_t* n, int for_discriminator,
                                 float* loss, float* tmp, uegan_stream_t stream) {
  RaArgs a;
  long long maxn;
  int rc = ra_fill(a, nscales, real, fake, n, for_discriminator, nullptr, nullptr, tmp, maxn);
  if (rc) return rc;
  UEGAN_CHECK_ARG(loss, "null loss");
  a.loss = loss;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(a.nbx, nscales);
  hipLaunchKernelGGL(rahinge_means_kernel, grid, dim3(256), 0, s, a);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(rahinge_terms_kernel, grid, dim3(256), 0, s, a);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(rahinge_loss_kernel, dim3(1), dim3(64), 0, s, a);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_rahinge_bwd(int nscales, const float* const* real, const float* const* fake, const int64_t* n, int for_discriminator,
                                 const float* tmp, const float* gscale, float* const* greal, float* const* gfake, uegan_stream_t stream) {
  RaArgs a;
  long long maxn;
  int rc = ra_fill(a, nscales, real, fake, n, for_discriminator, greal, gfake, const_cast<float*>(tmp), maxn);
  if (rc) return rc;
  int bx = (int)((maxn + 1023) / 1024);
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(rahinge_grad_kernel, dim3(bx, nscales), dim3(256), 0, (hipStream_t)stream, a, gscale);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_rals_fwd(int nscales, const float* const* real, const float* const* fake, const int64_t* n, int for_discriminator,
                              float* loss, float* tmp, uegan_stream_t stream) {
  RaArgs a;
  long long maxn;
  int rc = ra_fill(a, nscales, real, fake, n, for_discriminator, nullptr, nullptr, tmp, maxn);
  if (rc) return rc;
  UEGAN_CHECK_ARG(loss, "null loss");
  a.loss = loss;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(a.nbx, nscales);
  hipLaunchKernelGGL(rahinge_means_kernel, grid, dim3(256), 0, s, a);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(rals_terms_kernel, grid, dim3(256), 0, s, a);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(rahinge_loss_kernel, dim3(1), dim3(64), 0, s, a);      // (same combination: sum_k (tmp2 / n + tmp3 / n) / 2)
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_rals_bwd(int nscales, const float* const* real, const float* const* fake, const int64_t* n, int for_discriminator,
                              const float* tmp, const float* gscale, float* const* greal, float* const* gfake, uegan_stream_t stream) {
  RaArgs a;
  long long maxn;
  int rc = ra_fill(a, nscales, real, fake, n, for_discriminator, greal, gfake, const_cast<float*>(tmp), maxn);
  if (rc) return rc;
  int bx = (int)((maxn + 1023) / 1024);
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(rals_grad_kernel, dim3(bx, nscales), dim3(256), 0, (hipStream_t)stream, a, gscale);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

static int pred_fill(PredArgs& a, int fid, float target, int nscales, const float* const* preds, const int64_t* n, float* const* gpreds, float* tmp,
                     long long& maxn) {
  UEGAN_CHECK_ARG(nscales >= 1 && nscales <= 8 && preds && n && tmp, "bad pred_loss args");
  UEGAN_CHECK_ARG(fid >= UEGAN_PRED_BCE && fid <= UEGAN_PRED_POS_MEAN, "bad pred_loss term %d", fid);
  maxn = 0;
  for (int i = 0; i < 8; ++i) {
    a.p[i] = i < nscales ? preds[i] : nullptr;
    a.g[i] = (i < nscales && gpreds) ? gpreds[i] : nullptr;
    a.n[i] = i < nscales ? (long long)n[i] : 0;
    if (i < nscales) {
      UEGAN_CHECK_ARG(preds[i] && n[i] > 0 && (!gpreds || gpreds[i]), "bad pred_loss scale %d", i);
      if (a.n[i] > maxn) maxn = a.n[i];
    }
  }
  a.tmp = tmp; a.loss = nullptr; a.nscales = nscales; a.fid = fid; a.target = target;
  return UEGAN_OK;
}

extern "C" int uegan_pred_loss_fwd(int term, float target, int nscales, const float* const* preds, const int64_t* n, float* loss, float* tmp,
                                   uegan_stream_t stream) {
  PredArgs a;
  long long maxn;
  int rc = pred_fill(a, term, target, nscales, preds, n, nullptr, tmp, maxn);
  if (rc) return rc;
  UEGAN_CHECK_ARG(loss, "null loss");
  a.loss = loss;
  hipStream_t s = (hipStream_t)stream;
  int bx = (int)((maxn + 1023) / 1024);
  if (bx > RB) bx = RB;
  if (bx < 1) bx = 1;
  a.nbx = bx;
  hipLaunchKernelGGL(pred_terms_kernel, dim3(bx, nscales), dim3(256), 0, s, a);
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(pred_loss_kernel, dim3(1), dim3(64), 0, s, a);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_pred_loss_bwd(int term, float target, int nscales, const float* const* preds, const int64_t* n, const float* gscale,
                                   float* const* gpreds, uegan_stream_t stream) {
  PredArgs a;
  long long maxn;
  float dummy;
  UEGAN_CHECK_ARG(gpreds, "null gradient table");
  int rc = pred_fill(a, term, target, nscales, preds, n, gpreds, &dummy, maxn);
  if (rc) return rc;
  int bx = (int)((maxn + 1023) / 1024);
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(pred_grad_kernel, dim3(bx, nscales), dim3(256), 0, (hipStream_t)stream, a, gscale);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

static int rahead_fill(RaHeadArgs& a, int nscales, const void* const* maps, const int64_t* pix_per_image, int nb, int cp, int ngroups,
                       int npairs, const int32_t* pairs, int for_discriminator, float* tmp, long long& maxn) {
  UEGAN_CHECK_ARG(nscales >= 1 && nscales <= 8 && maps && pix_per_image && tmp && nb > 0 && cp > 0, "bad rahinge_heads args");
  UEGAN_CHECK_ARG(ngroups >= 2 && ngroups <= RH_MAXG && npairs >= 1 && npairs <= RH_MAXP && pairs, "rahinge_heads: 2..%d groups, 1..%d pairs", RH_MAXG, RH_MAXP);
  maxn = 0;
  for (int i = 0; i < 8; ++i) {
    a.maps[i] = i < nscales ? maps[i] : nullptr;
    a.gmaps[i] = nullptr;
    a.npg[i] = i < nscales ? (long long)nb * pix_per_image[i] : 0;
    if (i < nscales) {
      UEGAN_CHECK_ARG(maps[i] && pix_per_image[i] > 0, "bad rahinge_heads scale %d", i);
      if (a.npg[i] > maxn) maxn = a.npg[i];
    }
  }
  for (int i = 0; i < RH_MAXP; ++i) {
    a.pr[i] = i < npairs ? pairs[2 * i] : -1;
    a.pf[i] = i < npairs ? pairs[2 * i + 1] : -1;
    if (i < npairs) UEGAN_CHECK_ARG(a.pr[i] >= 0 && a.pr[i] < ngroups && a.pf[i] >= 0 && a.pf[i] < ngroups && a.pr[i] != a.pf[i], "bad pair %d", i);
  }
  a.tmp = tmp; a.loss = nullptr; a.nscales = nscales; a.ngroups = ngroups; a.npairs = npairs; a.cp = cp; a.gmask = 0;
  a.sgn = for_discriminator ? 1.f : -1.f;
  return UEGAN_OK;
}

extern "C" size_t uegan_rahinge_heads_workspace_floats(int nscales) { return (size_t)nscales * (RH_MAXG + RH_MAXP * 4) * (1 + RB); }

extern "C" int uegan_rahinge_heads_fwd(int dtype, int nscales, const void* const* maps, const int64_t* pix_per_image, int nb, int cp,
                                       int ngroups, int npairs, const int32_t* pairs, int for_discriminator, float* loss, float* tmp,
                                       uegan_stream_t stream) {
  RaHeadArgs a;
  long long maxn;
  int rc = rahead_fill(a, nscales, maps, pix_per_image, nb, cp, ngroups, npairs, pairs, for_discriminator, tmp, maxn);
  if (rc) return rc;
  UEGAN_CHECK_ARG(loss, "null loss");
  UEGAN_CHECK_ARG(cp % epc_of(dtype) == 0, "head maps must carry whole 16-byte chunks per pixel");
  a.loss = loss;
  hipStream_t s = (hipStream_t)stream;
  int bx = (int)((maxn + 1023) / 1024);
  if (bx > RB) bx = RB;
  if (bx < 1) bx = 1;
  a.nbx = bx;
  DISPATCH_T(dtype, hipLaunchKernelGGL((rahead_means_kernel<T>), dim3(bx, nscales, ngroups), dim3(256), 0, s, a));
  UEGAN_CHECK_LAUNCH();
  DISPATCH_T(dtype, hipLaunchKernelGGL((rahead_terms_kernel<T>), dim3(bx, nscales, npairs), dim3(256), 0, s, a));
  UEGAN_CHECK_LAUNCH();
  hipLaunchKernelGGL(rahead_loss_kernel, dim3(1), dim3(64), 0, s, a);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_rahinge_heads_bwd(int dtype, int nscales, const void* const* maps, const int64_t* pix_per_image, int nb, int cp,
                                       int ngroups, int npairs, const int32_t* pairs, int for_discriminator, const float* tmp,
                                       const float* gscale, void* const* gmaps, uint32_t group_mask, uegan_stream_t stream) {
  RaHeadArgs a;
  long long maxn;
  int rc = rahead_fill(a, nscales, maps, pix_per_image, nb, cp, ngroups, npairs, pairs, for_discriminator, const_cast<float*>(tmp), maxn);
  if (rc) return rc;
  UEGAN_CHECK_ARG(gmaps && group_mask, "rahinge_heads_bwd: no gradient requested");
  for (int i = 0; i < nscales; ++i) {
    UEGAN_CHECK_ARG(gmaps[i], "null gradient map %d", i);
    a.gmaps[i] = gmaps[i];
  }
  a.gmask = group_mask;
  int bx = (int)((maxn + 255) / 256);
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  DISPATCH_T(dtype, hipLaunchKernelGGL((rahead_grad_kernel<T>), dim3(bx, nscales, ngroups), dim3(256), 0, (hipStream_t)stream, a, gscale));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

static int msrec_launch(const float* pred, const float* gt, float* loss, float* scratch, float* gpred, const float* gscale, int B, int C,
                        int H, int W, int kind, int nscales, hipStream_t s) {
  UEGAN_CHECK_ARG(pred && gt && B > 0 && C > 0 && H > 0 && W > 0, "bad multiscale-rec args");
  UEGAN_CHECK_ARG(kind >= 0 && kind <= 2 && nscales >= 1 && nscales <= 3, "multiscale rec: kind 0..2 (l1 / smoothl1 / l2), 1..3 scales");
  // (like AvgPool2d, which raises "Output size is too small" when a pooled map would be empty)
  UEGAN_CHECK_ARG(nscales == 1 || ((H >> (nscales - 1)) > 0 && (W >> (nscales - 1)) > 0), "multiscale rec loss: %dx%d is too small for %d scales", H, W,
                  nscales);
  const bool ragged = nscales > 1 && (H % 4 != 0 || W % 4 != 0);
  const size_t total = nscales == 1 ? (size_t)B * C * H * W : (size_t)B * C * ((H + 3) / 4) * ((W + 3) / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > MSREC_MAXB) blocks = MSREC_MAXB;
  if (blocks < 1) blocks = 1;
#define UEGAN_MSREC(K)                                                                                                              \
  do {                                                                                                                              \
    if (nscales == 1) hipLaunchKernelGGL((rec_flat_kernel<K>), dim3(blocks), dim3(256), 0, s, pred, gt, scratch, gpred, gscale, total); \
    else if (ragged) hipLaunchKernelGGL((msrec_ragged_kernel<K>), dim3(blocks), dim3(256), 0, s, pred, gt, scratch, gpred, gscale, B * C, H, W, nscales); \
    else hipLaunchKernelGGL((msrec_kernel<K>), dim3(blocks), dim3(256), 0, s, pred, gt, scratch, gpred, gscale, B * C, H, W, nscales); \
  } while (0)
  if (kind == 0) UEGAN_MSREC(0); else if (kind == 1) UEGAN_MSREC(1); else UEGAN_MSREC(2);
#undef UEGAN_MSREC
  UEGAN_CHECK_LAUNCH();
  if (loss) {
    hipLaunchKernelGGL(msrec_final_kernel, dim3(1), dim3(1024), 0, s, scratch, blocks, loss);
    UEGAN_CHECK_LAUNCH();
  }
  return UEGAN_OK;
}

extern "C" size_t uegan_msrec_scratch_floats(void) { return MSREC_MAXB; }

extern "C" int uegan_msrec_fwd(const float* pred, const float* gt, float* loss, float* scratch, int B, int C, int H, int W, int kind, int nscales,
                               uegan_stream_t stream) {
  UEGAN_CHECK_ARG(loss && scratch, "null loss / scratch");
  return msrec_launch(pred, gt, loss, scratch, nullptr, nullptr, B, C, H, W, kind, nscales, (hipStream_t)stream);
}

extern "C" int uegan_msrec_bwd(const float* pred, const float* gt, const float* gscale, float* gpred, int B, int C, int H, int W, int kind,
                               int nscales, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(gpred, "null gpred");
  return msrec_launch(pred, gt, nullptr, nullptr, gpred, gscale, B, C, H, W, kind, nscales, (hipStream_t)stream);
}

// the round-1/2 entry points of the default identity loss (three-scale L1): kept as aliases of uegan_msrec_* (kind 0, 3 scales)
extern "C" size_t uegan_msl1_scratch_floats(void) { return MSREC_MAXB; }
extern "C" int uegan_msl1_fwd(const float* pred, const float* gt, float* loss, float* scratch, int B, int C, int H, int W, uegan_stream_t stream) {
  return uegan_msrec_fwd(pred, gt, loss, scratch, B, C, H, W, 0, 3, stream);
}
extern "C" int uegan_msl1_bwd(const float* pred, const float* gt, const float* gscale, float* gpred, int B, int C, int H, int W,
                              uegan_stream_t stream) {
  return uegan_msrec_bwd(pred, gt, gscale, gpred, B, C, H, W, 0, 3, stream);
}
