// Narrow-head convolutions: stride-1 KxK convs with <= 4 real output channels -- the five discriminator prediction
// heads (C -> 1, 7x7 / 5x5, tanh; models.py:170-182) and the generator's last layer (32 -> 3, 7x7, tanh; models.py:34-35).
//
// With 1 or 3 output channels a matrix-core tile is >= 80 % padding and the work per staged byte is tiny, so these
// layers are HBM/LDS-bound, not MFMA-bound.  They run on the vector ALU instead: an 8 x 32 pixel tile keeps its
// (8+K-1) x (32+K-1) input patch in LDS for a 32-channel chunk and every thread walks the taps over it
// (bf16: v_dot2c_f32_bf16, two MACs per lane-instruction, no unpacking; fp32: v_fma).
//   forward : thread = one output pixel, <= 4 accumulators
//   dgrad   : one real output channel on >= 64 input channels (the discriminator's d2..d5 heads): head_dgrad_kernel, thread = (pixel, 8
//             channels), mirrored images of the reflection padding inline; other shapes stay on the MFMA / streaming kernels
//   wgrad   : thread = a set of (tap, channel) weights, register accumulators over a persistent sweep of pixel tiles;
//             per-block partials -> the same reduce kernel as the MFMA wgrad
#include <cstdlib>

#include "common.h"
#include "conv_internal.h"

namespace uegan {

constexpr int HT_H = 8, HT_W = 32;          // pixel tile
constexpr int HCH = 32;                      // channels per chunk
constexpr int HNCO = 4;                      // max real output channels

#ifdef UEGAN_HALF_FP16
typedef _Float16 bf16x2_t __attribute__((ext_vector_type(2)));
#else
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
#endif

// acc += dot(x[0..EPC), w[0..EPC)) for one 16-byte chunk of each
__device__ __forceinline__ float dot_chunk(float acc, u32x4 x, u32x4 w, bf16_t*) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t xd = x[d], wd = w[d];       // scalars first: bit_cast of a vector-element expression is miscompiled by host clang
#ifdef UEGAN_HALF_FP16
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(bf16x2_t, xd), __builtin_bit_cast(bf16x2_t, wd), acc, false);
#else
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xd), __builtin_bit_cast(bf16x2_t, wd), acc, false);
#endif
  }
  return acc;
}
__device__ __forceinline__ float dot_chunk(float acc, u32x4 x, u32x4 w, float*) {
#pragma unroll
  for (int d = 0; d < 4; ++d) acc = fmaf(bits_to_f32(x[d]), bits_to_f32(w[d]), acc);
  return acc;
}

struct HeadArgs {
  const void* x;        // conv input  [B][H][W][C]   (fwd, wgrad)
  const void* dz;       // grad of pre-activation output [B][H][W][Zc]   (wgrad)
  const void* w;        // OHWI [Zc][Kp]
  const float* bias;
  const float* scale;
  int scale_group;      // head_dgrad_kernel: images per scale group (0: one scalar)
  void* out;            // y [B][H][W][Zc]
  float* ws;            // wgrad partials [nblocks][nco][KS*KS*C]
  int B, H, W, C, Zc, nco, Kp, pad, act, nbias;
  int nty, ntx, ntiles;
};

// source pixel index of patch position (piy, pix) for a forward-style gather with reflection padding, or -1
__device__ __forceinline__ int head_src_pixel(const HeadArgs& a, int b, int y0, int x0, int piy, int pix) {
  int sy = reflect_idx(y0 + piy - a.pad, a.H), sx = reflect_idx(x0 + pix - a.pad, a.W);
  if (sy < 0 || sy >= a.H || sx < 0 || sx >= a.W) return -1;
  return (b * a.H + sy) * a.W + sx;
}

// ---------------------------------------------------------------------------------------------------------------------
// NCO: accumulators per thread (1 for the discriminator heads, else HNCO).  CG: threads per output pixel -- on maps too
// small to fill the chip with 8 x 32 tiles (D.d4/d5 heads: 32^2 and 16^2 maps with 256 / 512 channels) the tile shrinks to
// 8 x 8 pixels and 4 threads share a pixel, each taking one 16-byte channel group of every chunk (shuffle-reduced at the end).
template <typename T, int KS, int NCO, int CG>
__global__ void __launch_bounds__(256) head_fwd_kernel(HeadArgs a) {
  constexpr int EPC = DT<T>::EPC, NT = KS * KS;
  constexpr int TW = HT_W / CG;
  constexpr int PH = HT_H + KS - 1, PW = TW + KS - 1;
  constexpr int PITCH = HCH * (int)sizeof(T) + 16;        // bytes per patch pixel (+16: conflict-free 16-byte reads across pixels)
  constexpr int NCH16 = HCH / EPC;                        // 16-byte chunks per pixel per channel chunk
  static_assert(NCH16 % CG == 0, "channel groups must divide the chunk");
  __shared__ __attribute__((aligned(16))) unsigned char patch[PH * PW * PITCH];
  __shared__ __attribute__((aligned(16))) T wl[NCO * NT * HCH];

  const T* x = static_cast<const T*>(a.x);
  const T* w = static_cast<const T*>(a.w);
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tile_x = t % a.ntx; t /= a.ntx;
  const int tile_y = t % a.nty;
  const int b = t / a.nty;
  const int y0 = tile_y * HT_H, x0 = tile_x * TW;
  const int q = tid % CG, pxl = tid / CG;
  const int r = pxl / TW, c = pxl % TW;

  float acc[NCO];
#pragma unroll
  for (int co = 0; co < NCO; ++co) acc[co] = 0.f;
  for (int c0 = 0; c0 < a.C; c0 += HCH) {
    for (int i = tid; i < PH * PW * NCH16; i += 256) {
      const int pp = i / NCH16, g = i - pp * NCH16;
      const int piy = pp / PW, pix = pp - piy * PW;
      const int sp = head_src_pixel(a, b, y0, x0, piy, pix);
      const int cc = c0 + g * EPC;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (sp >= 0 && cc < a.C) v = *reinterpret_cast<const u32x4*>(x + (size_t)sp * a.C + cc);
      *reinterpret_cast<u32x4*>(patch + pp * PITCH + g * 16) = v;
    }
    for (int i = tid; i < NCO * NT * NCH16; i += 256) {
      const int g = i % NCH16, ct = i / NCH16, tp = ct % NT, co = ct / NT;
      const int cc = c0 + g * EPC;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (co < a.nco && cc < a.C) v = *reinterpret_cast<const u32x4*>(w + (size_t)co * a.Kp + (size_t)tp * a.C + cc);
      *reinterpret_cast<u32x4*>(&wl[(co * NT + tp) * HCH + g * EPC]) = v;
    }
    __syncthreads();
    for (int tp = 0; tp < NT; ++tp) {
      const int ky = tp / KS, kx = tp - ky * KS;
      const unsigned char* prow = patch + ((r + ky) * PW + c + kx) * PITCH;
#pragma unroll
      for (int gi = 0; gi < NCH16 / CG; ++gi) {
        const int g = gi * CG + q;
        const u32x4 xv = *reinterpret_cast<const u32x4*>(prow + g * 16);
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
          const u32x4 wv = *reinterpret_cast<const u32x4*>(&wl[(co * NT + tp) * HCH + g * EPC]);
          acc[co] = dot_chunk(acc[co], xv, wv, (T*)nullptr);
        }
      }
    }
    __syncthreads();
  }
  if (CG > 1) {
#pragma unroll
    for (int co = 0; co < NCO; ++co)
#pragma unroll
      for (int m = 1; m < CG; m <<= 1) acc[co] += __shfl_xor(acc[co], m);
  }
  const int oy = y0 + r, ox = x0 + c;
  if (q == 0 && oy < a.H && ox < a.W) {
    const float scale = a.scale ? *a.scale : 1.f;
    T* o = static_cast<T*>(a.out) + (((size_t)b * a.H + oy) * a.W + ox) * a.Zc;
    for (int n = 0; n < a.Zc; n += 4) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int co = n + k;
        v[k] = (co < a.nco) ? apply_act_ext(acc[co < NCO ? co : 0] * scale + ((a.bias && co < a.nbias) ? a.bias[co] : 0.f), a.act)
                            : apply_act_ext(0.f, a.act);
      }
      store4(o + n, v[0], v[1], v[2], v[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// partial[block][co][tap*C + ci] = sum over this block's pixel tiles of dz[p][co] * x[reflect(p + tap - pad)][ci]
template <typename T, int KS>
__global__ void __launch_bounds__(256) head_wgrad_kernel(HeadArgs a) {
  constexpr int EPC = DT<T>::EPC, NT = KS * KS;
  constexpr int PH = HT_H + KS - 1, PW = HT_W + KS - 1;
  constexpr int NCH16 = HCH / EPC;
  constexpr int NITEM = (NT * HCH + 255) / 256;          // (tap, channel) weights per thread
  __shared__ __attribute__((aligned(16))) T patch[PH * PW * HCH];
  __shared__ __attribute__((aligned(16))) float dzl[HT_H * HT_W * HNCO];

  const T* x = static_cast<const T*>(a.x);
  const T* dz = static_cast<const T*>(a.dz);
  const int tid = threadIdx.x;
  const int ktot = NT * a.C;
  // my items: item = tid + 256*j -> (tap, ci); patch offset of the item relative to the pixel = (ky*PW + kx)*HCH + ci
  int ioff[NITEM], itap[NITEM], ici[NITEM];
#pragma unroll
  for (int j = 0; j < NITEM; ++j) {
    const int it = tid + 256 * j;
    itap[j] = it / HCH;
    ici[j] = it - itap[j] * HCH;
    if (itap[j] >= NT) { itap[j] = -1; ioff[j] = 0; }
    else { const int ky = itap[j] / KS, kx = itap[j] - ky * KS; ioff[j] = (ky * PW + kx) * HCH + ici[j]; }
  }
  float* wsb = a.ws + (size_t)blockIdx.x * a.nco * ktot;
  for (int c0 = 0; c0 < a.C; c0 += HCH) {
    float acc[NITEM][HNCO];
#pragma unroll
    for (int j = 0; j < NITEM; ++j)
#pragma unroll
      for (int co = 0; co < HNCO; ++co) acc[j][co] = 0.f;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
      int t = tile;
      const int tile_x = t % a.ntx; t /= a.ntx;
      const int tile_y = t % a.nty;
      const int b = t / a.nty;
      const int y0 = tile_y * HT_H, x0 = tile_x * HT_W;
      __syncthreads();
      for (int i = tid; i < PH * PW * NCH16; i += 256) {
        const int pp = i / NCH16, g = i - pp * NCH16;
        const int piy = pp / PW, pix = pp - piy * PW;
        const int sp = head_src_pixel(a, b, y0, x0, piy, pix);
        const int cc = c0 + g * EPC;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (sp >= 0 && cc < a.C) v = *reinterpret_cast<const u32x4*>(x + (size_t)sp * a.C + cc);
        *reinterpret_cast<u32x4*>(&patch[pp * HCH + g * EPC]) = v;
      }
      {
        const int r = tid >> 5, c = tid & 31, oy = y0 + r, ox = x0 + c;
        float v[HNCO] = {0.f, 0.f, 0.f, 0.f};
        if (oy < a.H && ox < a.W) {
          const T* p = dz + (((size_t)b * a.H + oy) * a.W + ox) * a.Zc;
          for (int co = 0; co < a.nco; ++co) v[co] = DT<T>::ld(p + co);
        }
        *reinterpret_cast<f32x4*>(&dzl[tid * HNCO]) = f32x4{v[0], v[1], v[2], v[3]};
      }
      __syncthreads();
      for (int p = 0; p < HT_H * HT_W; ++p) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(&dzl[p * HNCO]);        // broadcast
        const int pbase = ((p >> 5) * PW + (p & 31)) * HCH;
#pragma unroll
        for (int j = 0; j < NITEM; ++j) {
          const float xv = DT<T>::ld(&patch[pbase + ioff[j]]);
          acc[j][0] = fmaf(xv, d.x, acc[j][0]);
          acc[j][1] = fmaf(xv, d.y, acc[j][1]);
          acc[j][2] = fmaf(xv, d.z, acc[j][2]);
          acc[j][3] = fmaf(xv, d.w, acc[j][3]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NITEM; ++j)
      if (itap[j] >= 0 && c0 + ici[j] < a.C)
        for (int co = 0; co < a.nco; ++co) wsb[(size_t)co * ktot + (size_t)itap[j] * a.C + c0 + ici[j]] = acc[j][co];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Data gradient of a ONE-output-channel head (the discriminator's prediction heads, models.py:170-182), reflection padding included:
//     dx[q][c] = sum over the virtual images u of q in padded space (q itself; -q for 1 <= q <= pad; 2(n-1) - q for n-1-pad <= q <= n-2, per
//                axis) and the taps t of  dz[u - t + pad] * w[t][c]
// The gradient map is ONE scalar per pixel, so on the matrix cores 7/8 of every K chunk is channel padding and the pad-grid + fold route needs
// two passes over a padded workspace; here a thread owns (pixel, 8 channels): 8 accumulators, K*K taps of {one dz scalar from an LDS patch,
// eight weights broadcast from LDS}.  Interior tiles run the unrolled tap loop, tiles on a border the general one over the images.
// a.dz: [B][H][W][Zc] (channel 0), a.w: IHWO pack [C][Kp] with k = tap * Zc + co, a.out: dx [B][H][W][C]
template <typename T, int KS>
__global__ void __launch_bounds__(256) head_dgrad_kernel(HeadArgs a) {
  constexpr int NT = KS * KS, PADK = (KS - 1) / 2;
  constexpr int PH = HT_H + KS - 1, PW = HT_W + KS - 1;
  constexpr int CCH = 64;                                  // channels per block: 8 groups of 8
  __shared__ float zp[PH * PW];
  __shared__ __attribute__((aligned(16))) float wl[NT * CCH];
  const T* dz = static_cast<const T*>(a.dz);
  const T* w = static_cast<const T*>(a.w);
  T* out = static_cast<T*>(a.out);
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tile_x = t % a.ntx; t /= a.ntx;
  const int tile_y = t % a.nty;
  const int b = t / a.nty;
  const int y0 = tile_y * HT_H, x0 = tile_x * HT_W;
  const int c0 = blockIdx.y * CCH;
  for (int i = tid; i < PH * PW; i += 256) {
    const int py = i / PW, px = i - py * PW;
    const int sy = y0 - PADK + py, sx = x0 - PADK + px;
    float v = 0.f;
    if (sy >= 0 && sy < a.H && sx >= 0 && sx < a.W) v = DT<T>::ld(dz + (((size_t)b * a.H + sy) * a.W + sx) * a.Zc);
    zp[i] = v;
  }
  for (int i = tid; i < NT * CCH; i += 256) {
    const int tp = i / CCH, c = i - tp * CCH;
    wl[i] = (c0 + c < a.C) ? DT<T>::ld(w + (size_t)(c0 + c) * a.Kp + (size_t)tp * a.Zc) : 0.f;
  }
  __syncthreads();
  const int g = tid & 7, pc = tid >> 3;                    // channel group, tile column
  const int cg = c0 + g * 8;
  if (cg >= a.C) return;
  const int qx = x0 + pc;
  const bool interior = y0 >= PADK + 1 && y0 + HT_H - 1 <= a.H - 2 - PADK && x0 >= PADK + 1 && x0 + HT_W - 1 <= a.W - 2 - PADK;      // (block-uniform)
  const float scale = a.scale ? a.scale[a.scale_group ? b / a.scale_group : 0] : 1.f;
  const float* wg = wl + g * 8;
  // (taps outermost with the tile's 8 rows inner -- each tap's weights read once for 64 FMAs -- measured 25 % SLOWER: 64 accumulators; the
  // interior rows as a loop of their own in front of the border code 4x slower: the compiler then unrolls 8 x 49 taps)
  for (int r = 0; r < HT_H; ++r) {
    const int qy = y0 + r;
    if (qy >= a.H || qx >= a.W) continue;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (interior) {
#pragma unroll
      for (int ty = 0; ty < KS; ++ty)
#pragma unroll
        for (int tx = 0; tx < KS; ++tx) {
          const float z = zp[(r + KS - 1 - ty) * PW + pc + KS - 1 - tx];
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wg + (ty * KS + tx) * CCH), w1 = *reinterpret_cast<const f32x4*>(wg + (ty * KS + tx) * CCH + 4);
          acc[0] = fmaf(z, w0.x, acc[0]); acc[1] = fmaf(z, w0.y, acc[1]); acc[2] = fmaf(z, w0.z, acc[2]); acc[3] = fmaf(z, w0.w, acc[3]);
          acc[4] = fmaf(z, w1.x, acc[4]); acc[5] = fmaf(z, w1.y, acc[5]); acc[6] = fmaf(z, w1.z, acc[6]); acc[7] = fmaf(z, w1.w, acc[7]);
        }
    } else {
      int uy[3], ux[3], ny = 1, nx = 1;
      uy[0] = qy; ux[0] = qx;
      if (qy >= 1 && qy <= PADK) uy[ny++] = -qy;
      if (qy >= a.H - 1 - PADK && qy <= a.H - 2) uy[ny++] = 2 * (a.H - 1) - qy;
      if (qx >= 1 && qx <= PADK) ux[nx++] = -qx;
      if (qx >= a.W - 1 - PADK && qx <= a.W - 2) ux[nx++] = 2 * (a.W - 1) - qx;
      for (int iy = 0; iy < ny; ++iy)
        for (int ty = 0; ty < KS; ++ty) {
          const int sy = uy[iy] - ty + PADK;
          if (sy < 0 || sy >= a.H) continue;
          const int py = sy - (y0 - PADK);
          if (py < 0 || py >= PH) continue;
          for (int ix = 0; ix < nx; ++ix)
            for (int tx = 0; tx < KS; ++tx) {
              const int sx = ux[ix] - tx + PADK;
              if (sx < 0 || sx >= a.W) continue;
              const int px = sx - (x0 - PADK);
              if (px < 0 || px >= PW) continue;
              const float z = zp[py * PW + px];
              const float* wt = wg + (ty * KS + tx) * CCH;
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[e] = fmaf(z, wt[e], acc[e]);
            }
        }
    }
    T* o = out + (((size_t)b * a.H + qy) * a.W + qx) * a.C + cg;
    store4(o, acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale);
    store4(o + 4, acc[4] * scale, acc[5] * scale, acc[6] * scale, acc[7] * scale);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
static int heads_launch(int which, int KS, HeadArgs& a, int nblocks, hipStream_t s) {
  dim3 block(256), grid(nblocks);
#define HEADS_CASE(K)                                                                                  \
  case K:                                                                                              \
    if (which == 0) hipLaunchKernelGGL((head_fwd_kernel<T, K, HNCO, 1>), grid, block, 0, s, a);        \
    else if (which == 1) hipLaunchKernelGGL((head_fwd_kernel<T, K, 1, 1>), grid, block, 0, s, a);      \
    else if (which == 3) hipLaunchKernelGGL((head_fwd_kernel<T, K, 1, 4>), grid, block, 0, s, a);      \
    else hipLaunchKernelGGL((head_wgrad_kernel<T, K>), grid, block, 0, s, a);                          \
    break;
  switch (KS) {
    HEADS_CASE(3)
    HEADS_CASE(5)
    HEADS_CASE(7)
    default: set_error("head kernels: unsupported kernel size %d", KS); return UEGAN_E_UNSUPPORTED;
  }
#undef HEADS_CASE
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

bool heads_applicable(const uegan_conv_desc* d) {
  const int cw = d->Cout_w ? d->Cout_w : d->Cout;
  return cw <= HNCO && d->stride == 1 && d->KH == d->KW && (d->KH == 3 || d->KH == 5 || d->KH == 7) && d->C2 == 0 &&
         d->pad_mode == UEGAN_PAD_REFLECT && d->pad == (d->KH - 1) / 2 && d->Ho == d->H && d->Wo == d->W;
}

static void heads_fill(const uegan_conv_desc* d, HeadArgs& a) {
  a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C1; a.Zc = d->Cout; a.nco = d->Cout_w ? d->Cout_w : d->Cout;
  a.pad = d->pad; a.act = d->act; a.nbias = a.nco;
  a.nty = (d->H + HT_H - 1) / HT_H; a.ntx = (d->W + HT_W - 1) / HT_W; a.ntiles = d->B * a.nty * a.ntx;
  a.x = a.dz = a.w = nullptr; a.bias = a.scale = nullptr; a.out = nullptr; a.ws = nullptr; a.Kp = 0; a.scale_group = 0;
}

int heads_fwd(const uegan_conv_desc* d, const void* x, const void* w_ohwi, const float* bias, const float* scale, void* y, hipStream_t s) {
  UEGAN_CHECK_ARG(!(scale && d->scale_group), "the narrow-head kernels take one scalar scale (no model layer combines a head with spectral norm)");
  HeadArgs a;
  heads_fill(d, a);
  a.x = x; a.w = w_ohwi; a.bias = bias; a.scale = scale; a.out = y;
  a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * d->C1);
  int which = a.nco == 1 ? 1 : 0, nblocks = a.ntiles;
  const bool no_cg = g_tuning[UEGAN_TUNE_HEADS_NO_CG] != 0;      // (uegan_set_tuning: the tests flip it)
  if (which == 1 && a.ntiles < 512 && !no_cg) {      // small map: 8 x 8 tiles, 4 threads per pixel
    which = 3;
    a.ntx = (d->W + HT_W / 4 - 1) / (HT_W / 4);
    nblocks = a.ntiles = d->B * a.nty * a.ntx;
  }
  return d->dtype == UEGAN_F32 ? heads_launch<float>(which, d->KH, a, nblocks, s) : heads_launch<bf16_t>(which, d->KH, a, nblocks, s);
}

// the one-output-channel heads' data gradient on the vector ALU (head_dgrad_kernel): dx complete, mirrored images included, no workspace
bool heads_dgrad_applicable(const uegan_conv_desc* d) {
  const int cw = d->Cout_w ? d->Cout_w : d->Cout;
  return heads_applicable(d) && cw == 1 && (d->KH == 5 || d->KH == 7) && d->C1 % 8 == 0 && d->C1 >= 64 && d->H > 2 * d->pad + 1 && d->W > 2 * d->pad + 1 &&
         d->Cout == (d->dtype == UEGAN_F32 ? 4 : 8);
}

int heads_dgrad(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const float* scale, void* dx, hipStream_t s) {
  HeadArgs a;
  heads_fill(d, a);
  a.dz = dz; a.w = w_ihwo; a.scale = scale; a.scale_group = d->scale_group; a.out = dx;
  a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * d->Cout);
  const dim3 grid(a.ntiles, (d->C1 + 63) / 64), block(256);
  if (d->dtype == UEGAN_F32) {
    if (d->KH == 5) hipLaunchKernelGGL((head_dgrad_kernel<float, 5>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((head_dgrad_kernel<float, 7>), grid, block, 0, s, a);
  } else {
    if (d->KH == 5) hipLaunchKernelGGL((head_dgrad_kernel<bf16_t, 5>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((head_dgrad_kernel<bf16_t, 7>), grid, block, 0, s, a);
  }
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

int heads_wgrad_blocks(const uegan_conv_desc* d) {
  const int ntiles = d->B * ((d->H + HT_H - 1) / HT_H) * ((d->W + HT_W - 1) / HT_W);
  return ntiles < 512 ? ntiles : 512;
}

int heads_wgrad(const uegan_conv_desc* d, const void* x, const void* dz, float* ws, hipStream_t s) {
  HeadArgs a;
  heads_fill(d, a);
  a.x = x; a.dz = dz; a.ws = ws;
  const int nb = heads_wgrad_blocks(d);
  return d->dtype == UEGAN_F32 ? heads_launch<float>(2, d->KH, a, nb, s) : heads_launch<bf16_t>(2, d->KH, a, nb, s);
}

}  // namespace uegan
