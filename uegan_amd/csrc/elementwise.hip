// HBM-bound elementwise / resampling kernels (NHWC, channel-contiguous, fp32 math).
// Reference arithmetic: data_loader.py:79-81 tensor contract (NCHW fp32), models.py:70-72 (mul, residual, clamp),
// models.py:191-201 (bilinear x2 align_corners=True), torchvision VGG MaxPool2d(2,2), trainer.py:108 + losses.py:26-27
// (input rescale + ImageNet normalisation, folded into the NCHW->NHWC conversion).
#include "common.h"

namespace uegan {

struct Affine4 {
  float a[4], b[4];
  int on;
};

// NCHW fp32 -> NHWC T.  One thread per (pixel, 16-byte chunk of channels): the plane reads are coalesced along W, the write is one
// 16-byte store (the earlier thread-per-element form issued 2-byte stores and ran at 1.7 TB/s).  Cp is a multiple of one chunk.
// pair (16-bit storage, 2 C <= Cp): channels [C, 2C) receive what the rounding of channels [0, C) left, lo = rn16(v - rn16(v)) -- the image as a hi + lo
// pair in the spare channels of its own pixels (the generator's first convolution in the `precise` mode, uegan_conv2d_fwd_ex)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, T* y, int B, int C, int Cp, int HW, Affine4 af, int pair) {
  constexpr int EPC = DT<T>::EPC;
  const int nch = Cp / EPC;
  const size_t total = (size_t)B * HW * nch;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const size_t p = i / nch;
    const int b = (int)(p / HW);
    const size_t hw = p - (size_t)b * HW;
    float v[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = ch * EPC + e;
      v[e] = 0.f;                                    // channels C..Cp-1 are zero padding
      if (c < C) {
        v[e] = x[((size_t)b * C + c) * HW + hw];
        if (af.on) v[e] = v[e] * af.a[c] + af.b[c];
      } else if (pair && c < 2 * C) {
        float f = x[((size_t)b * C + (c - C)) * HW + hw];
        if (af.on) f = f * af.a[c - C] + af.b[c - C];
        T h;
        DT<T>::st(&h, f);
        v[e] = f - DT<T>::ld(&h);
      }
    }
    Vec<T, EPC>::st(y + p * Cp + ch * EPC, v);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* x, float* y, int B, int C, int Cp, int HW, Affine4 af) {
  const size_t total = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i indexes the NCHW output so that the fp32 writes are coalesced
    const size_t hw = i % HW;
    const size_t t = i / HW;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    float v = DT<T>::ld(x + ((size_t)b * HW + hw) * Cp + c);
    if (af.on) v *= af.a[c];
    y[i] = v;
  }
}

template <typename T>
__global__ void residual_clamp_fwd_kernel(const T* res, const float* x, float* out, int B, int C, int Cp, int HW) {
  const size_t total = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = i % HW;
    const size_t t = i / HW;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const float s = DT<T>::ld(res + ((size_t)b * HW + hw) * Cp + c) + x[i];
    out[i] = fminf(fmaxf(s, -1.f), 1.f);
  }
}

template <typename T>
__global__ void residual_clamp_bwd_kernel(const float* g, const T* res, const float* x, T* dres, float* dx, int B, int C, int Cp, int HW, int act) {
  // indexed over the padded NHWC gradient so that the padding channels are written (zero) too
  const size_t total = (size_t)B * HW * Cp;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(j % Cp);
    const size_t p = j / Cp;
    const int b = (int)(p / HW);
    const size_t hw = p - (size_t)b * HW;
    float m = 0.f;
    if (c < C) {
      const size_t i = ((size_t)b * C + c) * HW + hw;
      const float s = DT<T>::ld(res + j) + x[i];
      m = (s >= -1.f && s <= 1.f) ? g[i] : 0.f;   // torch.clamp backward: inclusive bounds
      if (dx) dx[i] = m;
      m *= act_grad_from_out(DT<T>::ld(res + j), act);     // act: the activation that produced res (tanh), its gradient deferred to here
    }
    DT<T>::st(dres + j, m);
  }
}

// flat elementwise kernels: V = one 16-byte chunk per thread when n allows it
template <typename T, int V>
__global__ void mul_fwd_kernel(const T* a, const T* b, T* y, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * blockDim.x * V) {
    float av[V], bv[V];
    Vec<T, V>::ld(a + i, av);
    Vec<T, V>::ld(b + i, bv);
#pragma unroll
    for (int e = 0; e < V; ++e) av[e] *= bv[e];
    Vec<T, V>::st(y + i, av);
  }
}
template <typename T, int V>
__global__ void mul_bwd_kernel(const T* g, const T* a, const T* b, T* da, T* db, size_t n, int act_a, int act_b) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * blockDim.x * V) {
    float gv[V], av[V], bv[V];
    Vec<T, V>::ld(g + i, gv);
    Vec<T, V>::ld(a + i, av);
    Vec<T, V>::ld(b + i, bv);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float t = gv[e];
      gv[e] = t * bv[e] * act_grad_from_out(av[e], act_a);      // act_a / act_b: deferred activation gradients of the operands' producers
      bv[e] = t * av[e] * act_grad_from_out(bv[e], act_b);
    }
    Vec<T, V>::st(da + i, gv);
    Vec<T, V>::st(db + i, bv);
  }
}
// bilinear x2, align_corners=True: src = dst * (in-1)/(out-1)   (torch upsample_bilinear2d area_pixel_compute_scale)
__device__ __forceinline__ void bilinear_src(int o, int in_n, int out_n, int& i0, int& i1, float& l1) {
  const float scale = out_n > 1 ? (float)(in_n - 1) / (float)(out_n - 1) : 0.f;
  const float s = scale * (float)o;
  i0 = (int)s;
  i1 = i0 + (i0 < in_n - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

// one thread = V channels of one output pixel; blockIdx.y = (image, output row): the row's source rows and weight are block-uniform
// (scalar), and no thread divides a 64-bit linear index (the linear form ran at 3.0 TB/s, VALU-bound, next to 5 TB/s neighbours)
template <typename T, int V>
__global__ void upsample2x_fwd_kernel(const T* x, T* y, int B, int H, int W, int C) {
  const int OH = 2 * H, OW = 2 * W, CV = C / V;
  const int row = blockIdx.y, b = row / OH, oy = row - b * OH;
  int y0, y1;
  float ly;
  bilinear_src(oy, H, OH, y0, y1, ly);
  const float hy = 1.f - ly;
  const T* r0 = x + ((size_t)b * H + y0) * W * C;
  const T* r1 = x + ((size_t)b * H + y1) * W * C;
  T* yo = y + (size_t)row * OW * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < OW * CV; i += gridDim.x * blockDim.x) {
    const int ox = i / CV, c = (i - ox * CV) * V;
    int x0, x1;
    float lx;
    bilinear_src(ox, W, OW, x0, x1, lx);
    float v00[V], v01[V], v10[V], v11[V];
    Vec<T, V>::ld(r0 + (size_t)x0 * C + c, v00);
    Vec<T, V>::ld(r0 + (size_t)x1 * C + c, v01);
    Vec<T, V>::ld(r1 + (size_t)x0 * C + c, v10);
    Vec<T, V>::ld(r1 + (size_t)x1 * C + c, v11);
    const float hx = 1.f - lx;
#pragma unroll
    for (int e = 0; e < V; ++e) v00[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    Vec<T, V>::st(yo + (size_t)ox * C + c, v00);
  }
}

// adjoint of the above in gather form: every input pixel scans the <=6 output rows/cols that can touch it; blockIdx.y = (image, input
// row): the live output rows and their weights are block-uniform
template <typename T, int V>
__global__ void upsample2x_bwd_kernel(const T* gy, T* gx, int B, int H, int W, int C) {
  const int OH = 2 * H, OW = 2 * W, CV = C / V;
  // XCD-banded block order (round 5): an output row is read by the 2 - 3 input rows it touches; in launch order those blocks sit on different XCDs
  // (round-robin by linear id) and every L2 fetches the row again -- FETCH_SIZE 2.07 GB/step for 1.0 GB of gradients.  Here XCD x takes the x-th
  // eighth of the (row, column-block) list, so neighbouring input rows share one L2.  (Time unchanged: the kernel is write / latency bound.)
  int bxi = blockIdx.x, row = blockIdx.y;
  {
    const unsigned total = gridDim.x * gridDim.y;
    if (total % 8 == 0) {
      const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, pidx = (L & 7) * (total >> 3) + (L >> 3);
      row = (int)(pidx / gridDim.x);
      bxi = (int)(pidx - (unsigned)row * gridDim.x);
    }
  }
  const int b = row / H, iy = row - b * H;
  float wyv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int oy = 2 * iy - 2 + k;
    float wy = 0.f;
    if (oy >= 0 && oy < OH) {
      int y0, y1;
      float ly;
      bilinear_src(oy, H, OH, y0, y1, ly);
      if (y0 == iy) wy += 1.f - ly;
      if (y1 == iy) wy += ly;
    }
    wyv[k] = wy;
  }
  const T* gb = gy + (size_t)b * OH * OW * C;
  T* go = gx + (size_t)row * W * C;
  for (int i = bxi * blockDim.x + threadIdx.x; i < W * CV; i += gridDim.x * blockDim.x) {
    const int ix = i / CV, c = (i - ix * CV) * V;
    float wxv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int ox = 2 * ix - 2 + k;
      float wx = 0.f;
      if (ox >= 0 && ox < OW) {
        int x0, x1;
        float lx;
        bilinear_src(ox, W, OW, x0, x1, lx);
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
      }
      wxv[k] = wx;
    }
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
      const float wy = wyv[ky];
      if (wy == 0.f) continue;                         // (block-uniform)
      const T* grow = gb + (size_t)(2 * iy - 2 + ky) * OW * C + c;
#pragma unroll
      for (int kx = 0; kx < 6; ++kx) {
        const float wx = wxv[kx];
        if (wx == 0.f) continue;
        float gv[V];
        Vec<T, V>::ld(grow + (size_t)(2 * ix - 2 + kx) * C, gv);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += wy * wx * gv[e];
      }
    }
    Vec<T, V>::st(go + (size_t)ix * C + c, acc);
  }
}

template <typename T, int V>
__global__ void maxpool2x2_fwd_kernel(const T* x, T* y, int B, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2, CV = C / V;
  const size_t total = (size_t)B * OH * OW * CV;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V;
    size_t p = i / CV;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const T* xb = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
    float m[V], t[V];
    Vec<T, V>::ld(xb, m);
    Vec<T, V>::ld(xb + C, t);
#pragma unroll
    for (int e = 0; e < V; ++e) m[e] = fmaxf(m[e], t[e]);
    Vec<T, V>::ld(xb + (size_t)W * C, t);
#pragma unroll
    for (int e = 0; e < V; ++e) m[e] = fmaxf(m[e], t[e]);
    Vec<T, V>::ld(xb + (size_t)W * C + C, t);
#pragma unroll
    for (int e = 0; e < V; ++e) m[e] = fmaxf(m[e], t[e]);
    Vec<T, V>::st(y + (i / CV) * C + c, m);
  }
}

// the same with the window position (dy * 2 + dx) of the FIRST maximum per element, one byte each: what the backward needs instead of x
template <typename T, int V>
__global__ void maxpool2x2_fwd_idx_kernel(const T* x, T* y, unsigned char* idx, int B, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2, CV = C / V;
  const size_t total = (size_t)B * OH * OW * CV;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V;
    size_t p = i / CV;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const T* xb = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
    const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    float m[V], t[V];
    unsigned char arg[V];
    Vec<T, V>::ld(xb, m);
#pragma unroll
    for (int e = 0; e < V; ++e) arg[e] = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      Vec<T, V>::ld(xb + off[k], t);
#pragma unroll
      for (int e = 0; e < V; ++e)
        if (t[e] > m[e]) { m[e] = t[e]; arg[e] = (unsigned char)k; }
    }
    Vec<T, V>::st(y + (i / CV) * C + c, m);
#pragma unroll
    for (int e = 0; e < V; ++e) idx[(i / CV) * C + c + e] = arg[e];
  }
}
// gx from the pooled tensor and the positions: gx[window position] = position == idx ? gy * act'(pooled) : 0 -- the values
// maxpool2x2_bwd_kernel computes from x (the maximum IS the pooled value), without reading x
template <typename T, int V>
__global__ void maxpool2x2_bwd_idx_kernel(const T* yp, const unsigned char* idx, const T* gy, T* gx, int B, int H, int W, int C, int act) {
  const int OH = H / 2, OW = W / 2, CV = C / V;
  const size_t total = (size_t)B * OH * OW * CV;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V;
    size_t p = i / CV;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const size_t base = (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
    const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    const size_t po = (i / CV) * C + c;
    float m[V], g[V];
    Vec<T, V>::ld(yp + po, m);
    Vec<T, V>::ld(gy + po, g);
    unsigned char arg[V];
    if constexpr (V == 8) {
      const u32x2 a8 = *reinterpret_cast<const u32x2*>(idx + po);
#pragma unroll
      for (int e = 0; e < 8; ++e) arg[e] = (unsigned char)((a8[e >> 2] >> (8 * (e & 3))) & 0xffu);
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) arg[e] = idx[po + e];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] *= act_grad_from_out(m[e], act);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o[V];
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = arg[e] == k ? g[e] : 0.f;
      Vec<T, V>::st(gx + base + off[k], o);
    }
  }
}

// gradient goes to the first maximum in (row, col) scan order, like ATen's max_pool2d_with_indices
template <typename T, int V>
__global__ void maxpool2x2_bwd_kernel(const T* x, const T* gy, T* gx, int B, int H, int W, int C, int act) {
  const int OH = H / 2, OW = W / 2, CV = C / V;
  const size_t total = (size_t)B * OH * OW * CV;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V;
    size_t p = i / CV;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const size_t base = (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
    const size_t off[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    float v[4][V], g[V];
#pragma unroll
    for (int k = 0; k < 4; ++k) Vec<T, V>::ld(x + base + off[k], v[k]);
    Vec<T, V>::ld(gy + (i / CV) * C + c, g);
    int arg[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float m = v[0][e];
      arg[e] = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][e] > m) { m = v[k][e]; arg[e] = k; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o[V];
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = arg[e] == k ? g[e] * act_grad_from_out(v[k][e], act) : 0.f;      // (act: x's producer's deferred act')
      Vec<T, V>::st(gx + base + off[k], o);
    }
  }
}

// threads of a block that owns one row of n work items: whole waves, at most 256
static inline int row_threads(size_t n) { return n >= 256 ? 256 : (int)((n + 63) / 64) * 64; }
static inline int grid_for(size_t n, int cap = 8192) {
  size_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b < (size_t)cap ? b : (size_t)cap);
}

}  // namespace uegan

using namespace uegan;

#define DISPATCH_T(dtype, ...)                                   \
  do {                                                           \
    if ((dtype) == UEGAN_F32) { using T = float; __VA_ARGS__; }  \
    else if ((dtype) == UEGAN_BF16) { using T = bf16_t; __VA_ARGS__; } \
    else { set_error("bad dtype %d", (int)(dtype)); return UEGAN_E_INVALID; } \
  } while (0)
// binds T and V (V = one 16-byte chunk per thread when `vec_ok`, else 1)
#define DISPATCH_TV(dtype, vec_ok, ...)                                                               \
  do {                                                                                                \
    if ((dtype) == UEGAN_F32) { using T = float; if (vec_ok) { constexpr int V = 4; __VA_ARGS__; } else { constexpr int V = 1; __VA_ARGS__; } } \
    else if ((dtype) == UEGAN_BF16) { using T = bf16_t; if (vec_ok) { constexpr int V = 8; __VA_ARGS__; } else { constexpr int V = 1; __VA_ARGS__; } } \
    else { set_error("bad dtype %d", (int)(dtype)); return UEGAN_E_INVALID; }                        \
  } while (0)
static inline int epc_of(int dtype) { return dtype == UEGAN_BF16 ? 8 : 4; }

static int make_affine(Affine4& af, int C, const float* a, const float* b) {
  af.on = (a != nullptr);
  for (int i = 0; i < 4; ++i) { af.a[i] = 1.f; af.b[i] = 0.f; }
  if (af.on) {
    UEGAN_CHECK_ARG(C <= 4, "per-channel affine supports C <= 4 (got %d)", C);
    for (int i = 0; i < C; ++i) { af.a[i] = a[i]; af.b[i] = b ? b[i] : 0.f; }
  }
  return UEGAN_OK;
}

extern "C" int uegan_nchw_to_nhwc(int dtype, const float* x, void* y, int B, int C, int Cp, int H, int W, const float* a, const float* b,
                                  uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && B > 0 && C > 0 && Cp >= C && H > 0 && W > 0, "bad args");
  UEGAN_CHECK_ARG(Cp % (dtype == UEGAN_BF16 ? 8 : 4) == 0, "Cp must be a multiple of one 16-byte chunk");
  Affine4 af;
  int rc = make_affine(af, C, a, b);
  if (rc) return rc;
  const size_t n = (size_t)B * Cp * H * W / (dtype == UEGAN_BF16 ? 8 : 4);
  DISPATCH_T(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (T*)y, B, C, Cp, H * W, af, 0));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_nchw_to_nhwc_pair(int dtype, const float* x, void* y, int B, int C, int Cp, int H, int W, const float* a, const float* b,
                                       uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && B > 0 && C > 0 && Cp >= 2 * C && H > 0 && W > 0, "bad args (the pair needs 2 C <= Cp)");
  UEGAN_CHECK_ARG(dtype == UEGAN_BF16 && Cp % 8 == 0, "hi + lo pairs exist for the 16-bit storage format; Cp must be a multiple of one 16-byte chunk");
  Affine4 af;
  int rc = make_affine(af, C, a, b);
  if (rc) return rc;
  const size_t n = (size_t)B * Cp * H * W / 8;
  hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, B, C, Cp, H * W, af, 1);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_nhwc_to_nchw(int dtype, const void* x, float* y, int B, int C, int Cp, int H, int W, const float* a,
                                  uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && B > 0 && C > 0 && Cp >= C && H > 0 && W > 0, "bad args");
  Affine4 af;
  int rc = make_affine(af, C, a, nullptr);
  if (rc) return rc;
  const size_t n = (size_t)B * C * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const T*)x, y, B, C, Cp, H * W, af));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_residual_clamp_fwd(int dtype, const void* res, const float* x, float* out, int B, int C, int Cp, int H, int W,
                                        uegan_stream_t stream) {
  UEGAN_CHECK_ARG(res && x && out && Cp >= C, "bad args");
  const size_t n = (size_t)B * C * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL((residual_clamp_fwd_kernel<T>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const T*)res, x, out, B, C, Cp, H * W));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_residual_clamp_bwd(int dtype, const float* g, const void* res, const float* x, void* dres, float* dx, int B, int C,
                                        int Cp, int H, int W, uegan_stream_t stream) {
  return uegan_residual_clamp_bwd_act(dtype, UEGAN_ACT_NONE, g, res, x, dres, dx, B, C, Cp, H, W, stream);
}
extern "C" int uegan_residual_clamp_bwd_act(int dtype, int act, const float* g, const void* res, const float* x, void* dres, float* dx, int B,
                                            int C, int Cp, int H, int W, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(g && res && x && dres && Cp >= C, "bad args");
  const size_t n = (size_t)B * Cp * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL((residual_clamp_bwd_kernel<T>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, (const T*)res, x, (T*)dres, dx, B, C, Cp, H * W, act));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_mul_fwd(int dtype, const void* a, const void* b, void* y, int64_t n, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(a && b && y && n > 0, "bad args");
  DISPATCH_TV(dtype, n % epc_of(dtype) == 0, hipLaunchKernelGGL((mul_fwd_kernel<T, V>), dim3(grid_for((size_t)n / V)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, (T*)y, (size_t)n));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}
extern "C" int uegan_mul_bwd(int dtype, const void* g, const void* a, const void* b, void* da, void* db, int64_t n, uegan_stream_t stream) {
  return uegan_mul_bwd_act(dtype, UEGAN_ACT_NONE, UEGAN_ACT_NONE, g, a, b, da, db, n, stream);
}
extern "C" int uegan_mul_bwd_act(int dtype, int act_a, int act_b, const void* g, const void* a, const void* b, void* da, void* db, int64_t n,
                                 uegan_stream_t stream) {
  UEGAN_CHECK_ARG(g && a && b && da && db && n > 0, "bad args");
  DISPATCH_TV(dtype, n % epc_of(dtype) == 0, hipLaunchKernelGGL((mul_bwd_kernel<T, V>), dim3(grid_for((size_t)n / V)), dim3(256), 0, (hipStream_t)stream, (const T*)g, (const T*)a, (const T*)b, (T*)da, (T*)db, (size_t)n, act_a, act_b));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}
extern "C" int uegan_upsample2x_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0, "bad args");
  UEGAN_CHECK_ARG((long long)B * 2 * H <= 65535, "upsample2x: B * 2H rows exceed the grid's y extent");
  DISPATCH_TV(dtype, C % epc_of(dtype) == 0, hipLaunchKernelGGL((upsample2x_fwd_kernel<T, V>), dim3(grid_for((size_t)2 * W * C / V, 64), B * 2 * H), dim3(row_threads((size_t)2 * W * C / V)), 0, (hipStream_t)stream, (const T*)x, (T*)y, B, H, W, C));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}
extern "C" int uegan_upsample2x_bwd(int dtype, const void* gy, void* gx, int B, int H, int W, int C, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(gy && gx && B > 0 && H > 0 && W > 0 && C > 0, "bad args");
  UEGAN_CHECK_ARG((long long)B * H <= 65535, "upsample2x: B * H rows exceed the grid's y extent");
  DISPATCH_TV(dtype, C % epc_of(dtype) == 0, hipLaunchKernelGGL((upsample2x_bwd_kernel<T, V>), dim3(grid_for((size_t)W * C / V, 64), B * H), dim3(row_threads((size_t)W * C / V)), 0, (hipStream_t)stream, (const T*)gy, (T*)gx, B, H, W, C));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_maxpool2x2_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && B > 0 && H > 1 && W > 1 && C > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2x2 needs even H,W");
  const size_t n = (size_t)B * (H / 2) * (W / 2) * C;
  DISPATCH_TV(dtype, C % epc_of(dtype) == 0, hipLaunchKernelGGL((maxpool2x2_fwd_kernel<T, V>), dim3(grid_for(n / V)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, B, H, W, C));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}
extern "C" int uegan_maxpool2x2_fwd_idx(int dtype, const void* x, void* y, void* idx, int B, int H, int W, int C, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && y && idx && B > 0 && H > 1 && W > 1 && C > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2x2 needs even H,W");
  const size_t n = (size_t)B * (H / 2) * (W / 2) * C;
  DISPATCH_TV(dtype, C % epc_of(dtype) == 0, hipLaunchKernelGGL((maxpool2x2_fwd_idx_kernel<T, V>), dim3(grid_for(n / V)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, (unsigned char*)idx, B, H, W, C));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}
extern "C" int uegan_maxpool2x2_bwd_idx(int dtype, int act, const void* y_pool, const void* idx, const void* gy, void* gx, int B, int H, int W, int C,
                                        uegan_stream_t stream) {
  UEGAN_CHECK_ARG(y_pool && idx && gy && gx && B > 0 && H > 1 && W > 1 && C > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2x2 needs even H,W");
  const size_t n = (size_t)B * (H / 2) * (W / 2) * C;
  DISPATCH_TV(dtype, C % epc_of(dtype) == 0, hipLaunchKernelGGL((maxpool2x2_bwd_idx_kernel<T, V>), dim3(grid_for(n / V)), dim3(256), 0, (hipStream_t)stream, (const T*)y_pool, (const unsigned char*)idx, (const T*)gy, (T*)gx, B, H, W, C, act));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}
extern "C" int uegan_maxpool2x2_bwd(int dtype, const void* x, const void* gy, void* gx, int B, int H, int W, int C, uegan_stream_t stream) {
  return uegan_maxpool2x2_bwd_act(dtype, UEGAN_ACT_NONE, x, gy, gx, B, H, W, C, stream);
}
extern "C" int uegan_maxpool2x2_bwd_act(int dtype, int act, const void* x, const void* gy, void* gx, int B, int H, int W, int C,
                                        uegan_stream_t stream) {
  UEGAN_CHECK_ARG(x && gy && gx && B > 0 && H > 1 && W > 1 && C > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2x2 needs even H,W");
  const size_t n = (size_t)B * (H / 2) * (W / 2) * C;
  DISPATCH_TV(dtype, C % epc_of(dtype) == 0, hipLaunchKernelGGL((maxpool2x2_bwd_kernel<T, V>), dim3(grid_for(n / V)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)gy, (T*)gx, B, H, W, C, act));
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}
