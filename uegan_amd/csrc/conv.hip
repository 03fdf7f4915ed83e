// Convolution family for gfx950: MFMA implicit-GEMM forward / dgrad (one gather-GEMM kernel), split-K MFMA
// wgrad, weight packing, and scalar "direct" kernels used only for on-GPU cross-checks.
//
// Reference arithmetic: nn.ReflectionPad2d + nn.Conv2d (+bias) + LeakyReLU/ReLU/tanh and autograd's
// convolution_backward (models.py:80-84, 92-98, 161-166, 173-178; torchvision VGG conv3x3 + ReLU).
//
// Data layout: activations NHWC (channel contiguous), packed weights [rows][Kp] with the GEMM reduction
// index k = (kh, kw, c) contiguous -- both MFMA operands are "row-major with K contiguous", so a lane's
// fragment is one 16-byte LDS read.  The weight matrix is the MFMA A operand and the pixel tile the B
// operand: D[channel][pixel], so each lane ends up with 4 consecutive channels of one pixel and the
// NHWC store is a single 8/16-byte vector store per fragment.
#include "conv_core.h"

#include <type_traits>

namespace uegan {

static int g_conv_impl = UEGAN_IMPL_AUTO;
// launch-variant thresholds (uegan_set_tuning): process-wide, set explicitly through the C ABI -- the library never reads the environment
int g_tuning[UEGAN_TUNE_COUNT] = {256, -1, 0, 192, 192, 0, 1, 1, 1, 1, 1, 1};
int g_abl_stream = 0, g_abl_wide = 0;
#ifdef UEGAN_TOOLS_BUILD
extern "C" int uegan_tools_set_ablation(int stream_wgrad_bits, int wide_variant) {
  g_abl_stream = stream_wgrad_bits;
  g_abl_wide = wide_variant;
  return UEGAN_OK;
}
#endif

bool g_prof_on = false;
std::vector<ProfRecord> g_prof_records;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
size_t g_prof_used = 0;

// patch-resident kernel instantiations live in conv_patch_{bf16,f32}_{a,b}.hip (a: KS 1..3, b: KS 4, 5, 7); 1 = no such KS
int conv_patch_bf16_a(ConvArgs& a, hipStream_t s, int ks);
int conv_patch_bf16_b(ConvArgs& a, hipStream_t s, int ks);
int conv_patch_f32_a(ConvArgs& a, hipStream_t s, int ks);
int conv_patch_f32_b(ConvArgs& a, hipStream_t s, int ks);
int conv_toep_run(ConvArgs& a, int dtype, hipStream_t s);         // conv_toep.hip: <= 4 output channels as a Toeplitz product; 1 = not taken
bool conv_toep_takes(const ConvArgs& a, int dtype);
bool heads_dgrad_mfma_applicable(const uegan_conv_desc* d);      // heads_mfma.hip: one-channel heads' data gradient over the padded grid on the MFMA
int heads_dgrad_mfma(const uegan_conv_desc* d, const void* dz, const void* w_ohwi, void* out, hipStream_t s);
int conv_wide_run(ConvArgs& a, int dtype, hipStream_t s, bool interior = false);      // conv_wide.hip: 256-channel tiles, one wave per SIMD; 1 = not taken
int conv_tall_run(ConvArgs& a, int dtype, hipStream_t s, bool interior = false);      // conv_wide.hip: 64- / 128-channel blocks on 16 x 32-pixel tiles, one wave per SIMD; 1 = not taken
int conv_interior_run(ConvArgs& a, int dtype, hipStream_t s);    // conv_wide.hip: the image-free interior of a reflection-padded data gradient on those two; 1 = not taken
int conv_s2fwd_run(ConvArgs& a, int dtype, hipStream_t s);       // conv_s2.hip: stride-2 forwards by input parity classes; 1 = not taken
bool conv_flat_applicable(const uegan_conv_desc* d);             // conv_flat.hip: stride-2 data gradients over the padded grid, all parity classes in one launch
int conv_flat_run(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const float* scale, void* out, hipStream_t s);      // 1 = not taken
template <typename T> static int patch_run(ConvArgs& a, hipStream_t s, int ks);
template <> int patch_run<bf16_t>(ConvArgs& a, hipStream_t s, int ks) { return ks <= 3 ? conv_patch_bf16_a(a, s, ks) : conv_patch_bf16_b(a, s, ks); }
template <> int patch_run<float>(ConvArgs& a, hipStream_t s, int ks) { return ks <= 3 ? conv_patch_f32_a(a, s, ks) : conv_patch_f32_b(a, s, ks); }


template <typename T, int BN, int WARPS_M, int WARPS_N, bool GLDS>
__global__ void __launch_bounds__(256) conv_gemm_kernel(ConvArgs a) {
  constexpr int BM = CONV_BM, ROWB = CONV_ROWB;
  constexpr int EPC = DT<T>::EPC;
  constexpr int BK = ROWB / (int)sizeof(T);          // reduction elements per K step
  constexpr int NI_X = BM / 32;                      // staging instructions per thread for the pixel tile (8 rows each)
  constexpr int WROWG = BN / 8;                      // 8-row groups of the weight tile
  constexpr int NI_W = (WROWG + 3) / 4;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int NCHUNK = Mma<T>::NCHUNK;
  constexpr int NSUB = BK / 32;                      // 32-wide MFMA K sub-steps per K step (bf16: 2, fp32: 1)
  constexpr int BUFB = (BM + BN) * ROWB;
  static_assert(WARPS_M * WARPS_N == 4 && TM >= 1 && TN >= 1, "tile");

  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFB];

  const ConvGeom& g = a.g;
  const T* in1 = static_cast<const T*>(a.in1);
  const T* in2 = static_cast<const T*>(a.in2);
  const T* w = static_cast<const T*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int n0 = blockIdx.y * BN;

  // ---- tile decode: (image b, parity class, tile_y, tile_x)
  const int sub = (g.mode == 1) ? g.stride : 1;      // pixel stride inside the tile (dgrad parity classes)
  int t = blockIdx.x;
  int tile_x, tile_y, pcls = 0, b;
  tile_x = t % a.ntx; t /= a.ntx;
  tile_y = t % a.nty; t /= a.nty;
  pcls = t % (sub * sub);
  b = t / (sub * sub);
  const int py = pcls / sub, px = pcls - py * sub;
  // taps this tile iterates: dgrad keeps ty with (py + pad - ty) % stride == 0
  const int ty0 = (g.mode == 1) ? (py + g.pad) % sub : 0;
  const int tx0 = (g.mode == 1) ? (px + g.pad) % sub : 0;
  const int nty_t = ty0 < g.KH ? (g.KH - ty0 + sub - 1) / sub : 0;
  const int ntx_t = tx0 < g.KW ? (g.KW - tx0 + sub - 1) / sub : 0;
  const int kvalid = nty_t * ntx_t * g.C;            // flattened (tap, channel) reduction length of this tile
  const int nk = (kvalid + BK - 1) / BK;

  // ---- staging role of this thread: LDS (row, pos) per instruction i -> row = (i*4 + wave)*8 + (lane>>3), pos = lane&7
  const int srow = lane >> 3;
  const int spos = lane & 7;
  const int sdc = spos ^ (((lane >> 4) + 4 * (wave & 1)) & 7);     // data chunk held at that position (same for every i)
  // initial (tap, channel) of my chunk: flattened offset sdc*EPC
  int tyi0, txi0, c0;
  {
    const int q = sdc * EPC;
    const int ti = q / g.C;
    c0 = q - ti * g.C;
    tyi0 = ntx_t > 0 ? ti / ntx_t : 0;
    txi0 = ntx_t > 0 ? ti - tyi0 * ntx_t : 0;
  }
  // my pixel rows
  int roy[NI_X], rox[NI_X];
  bool rv[NI_X];
#pragma unroll
  for (int i = 0; i < NI_X; ++i) {
    const int r = (i * 4 + wave) * 8 + srow;
    roy[i] = py + sub * (tile_y * CONV_TH + (r >> 4));
    rox[i] = px + sub * (tile_x * CONV_TW + (r & 15));
    rv[i] = roy[i] < g.OH && rox[i] < g.OW;
  }
  // block-uniform list of padded-space images (4 bits per entry), from the tile's coordinate range: an image is
  // listed when some row of the tile MAY have it (rows that do not simply gather nothing for it)
  unsigned long long imgs = 0;
  int nimg = 0;
  if (g.mode == 1 && g.pad_mode == UEGAN_PAD_REFLECT) {
    const int y_lo = py + sub * tile_y * CONV_TH, y_hi = py + sub * (tile_y * CONV_TH + CONV_TH - 1);
    const int x_lo = px + sub * tile_x * CONV_TW, x_hi = px + sub * (tile_x * CONV_TW + CONV_TW - 1);
    bool hy[3], hx[3];
    hy[0] = hx[0] = true;
    hy[1] = y_lo <= g.pad && y_hi >= 1;
    hy[2] = y_lo <= g.OH - 2 && y_hi >= g.OH - 1 - g.pad;
    hx[1] = x_lo <= g.pad && x_hi >= 1;
    hx[2] = x_lo <= g.OW - 2 && x_hi >= g.OW - 1 - g.pad;
    for (int q = 0; q < 9; ++q)
      if (hy[q / 3] && hx[q % 3]) {
        imgs |= (unsigned long long)q << (4 * nimg);
        ++nimg;
      }
  } else {
    nimg = 1;
  }
  const int nsteps = nimg * nk;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // running decode state of my chunk
  int tyi = tyi0, txi = txi0, cc = c0, ks_in_img = 0, img_i = 0;
  u32x4 xreg[NI_X], wreg[NI_W];

  // gathered source pixel of my NI_X rows for the current (image, tap): recomputed only when the tap changes -- with >= 128
  // channels several consecutive K steps (1x1 convs: all of them) read the same pixels at different channel offsets
  int pixoff[NI_X];
  const T* pbase[NI_X];          // single-source tensors: in1 + pixel * C1 of the cached pixel (a K step only adds the channel offset)
  const T* wrow[NI_W];           // start of my weight rows (null: row beyond N)
#pragma unroll
  for (int i = 0; i < NI_X; ++i) pbase[i] = nullptr;
#pragma unroll
  for (int i = 0; i < NI_W; ++i) {
    const int rg = i * 4 + wave;
    const int n = n0 + rg * 8 + srow;
    wrow[i] = (rg < WROWG && n < a.N) ? w + (size_t)n * a.Kp : nullptr;
  }
  const bool one_src = g.C2 == 0;
  int pix_key = -1;
  auto stage = [&](unsigned char* buf) {
    // addresses for the current step, then advance the state by one K step
    const bool kv = tyi < nty_t;
    const int key = (img_i * 16 + tyi) * 16 + txi;
    if (key != pix_key) {
      pix_key = key;
      const int q = (int)((imgs >> (4 * img_i)) & 15ull);
      const int iy = q / 3, ix = q - iy * 3;
      const int ty = ty0 + sub * tyi, tx = tx0 + sub * txi;
#pragma unroll
      for (int i = 0; i < NI_X; ++i) {
        int off = -1;
        if (kv && rv[i]) {
          const int sy = src_coord(g, roy[i], ty, iy, g.IH, g.OH);
          const int sx = src_coord(g, rox[i], tx, ix, g.IW, g.OW);
          if (sy >= 0 && sx >= 0) off = (b * g.IH + sy) * g.IW + sx;
        }
        pixoff[i] = off;
        pbase[i] = off >= 0 ? in1 + (size_t)off * g.C1 : nullptr;
      }
    }
    const int ty = ty0 + sub * tyi, tx = tx0 + sub * txi;
#pragma unroll
    for (int i = 0; i < NI_X; ++i) {
      const void* src = g_zero16;
      if (pixoff[i] >= 0) {
        if (one_src) {
          src = pbase[i] + cc;
        } else {
          const size_t pix = (size_t)pixoff[i];
          src = (cc < g.C1) ? (const void*)(in1 + pix * g.C1 + cc) : (const void*)(in2 + pix * g.C2 + (cc - g.C1));
        }
      }
      if (GLDS) glds16(src, buf + ((i * 4 + wave) * 8) * ROWB);
      else xreg[i] = *reinterpret_cast<const u32x4*>(src);
    }
#pragma unroll
    for (int i = 0; i < NI_W; ++i) {
      const int rg = i * 4 + wave;
      if (rg < WROWG) {
        const void* src = g_zero16;
        if (kv && wrow[i]) src = wrow[i] + ((ty * g.KW + tx) * g.C + cc);
        if (GLDS) glds16(src, buf + (BM + rg * 8) * ROWB);
        else wreg[i] = *reinterpret_cast<const u32x4*>(src);
      }
    }
    // advance
    ++ks_in_img;
    if (ks_in_img == nk) {
      ks_in_img = 0; ++img_i; tyi = tyi0; txi = txi0; cc = c0;
    } else {
      cc += BK;
      while (cc >= g.C) {
        cc -= g.C;
        if (++txi == ntx_t) { txi = 0; ++tyi; }
      }
    }
  };
  auto commit = [&](unsigned char* buf) {   // register-staged mode: VGPRs -> LDS
#pragma unroll
    for (int i = 0; i < NI_X; ++i)
      *reinterpret_cast<u32x4*>(buf + ((i * 4 + wave) * 8 + srow) * ROWB + spos * 16) = xreg[i];
#pragma unroll
    for (int i = 0; i < NI_W; ++i) {
      const int rg = i * 4 + wave;
      if (rg < WROWG) *reinterpret_cast<u32x4*>(buf + (BM + rg * 8 + srow) * ROWB + spos * 16) = wreg[i];
    }
  };

  if (nsteps > 0) {
    stage(lds);
    if (!GLDS) commit(lds);
  }
  const int fr = lane & 15, fg = lane >> 4;
  for (int s = 0; s < nsteps; ++s) {
    unsigned char* cur = lds + (s & 1) * BUFB;
    unsigned char* nxt = lds + ((s + 1) & 1) * BUFB;
    __syncthreads();                       // step s staged (the compiler drains vmcnt here); buffer nxt is free again
    if (s + 1 < nsteps) stage(nxt);
#pragma unroll
    for (int ksub = 0; ksub < NSUB; ++ksub) {
      u32x4 xf[TM][NCHUNK], wf[TN][NCHUNK];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = wm * WTM + j * 16 + fr;
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
          const int q = ksub * 4 + c * 4 * (NCHUNK - 1) + fg;      // data chunk index within the 128-byte row
          xf[j][c] = *reinterpret_cast<const u32x4*>(cur + row * ROWB + ((q ^ ((row >> 1) & 7)) << 4));
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn * WTN + i * 16 + fr;
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
          const int q = ksub * 4 + c * 4 * (NCHUNK - 1) + fg;
          wf[i][c] = *reinterpret_cast<const u32x4*>(cur + (BM + row) * ROWB + ((q ^ ((row >> 1) & 7)) << 4));
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mma<T>::step(wf[i], xf[j], acc[i][j]);
    }
    if (!GLDS && s + 1 < nsteps) commit(nxt);
  }

  // ---- epilogue: lane holds channels n..n+3 of pixel (tile row m)
  const float scale = a.scale ? a.scale[a.scale_group ? b / a.scale_group : 0] : 1.f;
  T* out = static_cast<T*>(a.out);
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * WTN + i * 16 + (lane >> 4) * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.nbias) bv[r] = a.bias[n + r];
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = wm * WTM + j * 16 + (lane & 15);
      const int oy = py + sub * (tile_y * CONV_TH + (m >> 4));
      const int ox = px + sub * (tile_x * CONV_TW + (m & 15));
      if (oy >= g.OH || ox >= g.OW || n >= a.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(acc[i][j][r] * scale + bv[r], a.act);
      const size_t pixo = ((size_t)b * g.OH + oy) * g.OW + ox;
      T* p = (a.out2 && n >= a.n_out1) ? static_cast<T*>(a.out2) + pixo * (a.N - a.n_out1) + (n - a.n_out1)
                                       : out + pixo * (a.out2 ? a.n_out1 : a.N) + n;
      store4(p, v[0], v[1], v[2], v[3]);      // channel counts are multiples of 4 (padded tensors)
    }
  }
}



static bool g_use_patch = true;
static bool g_use_heads = true;

static bool g_use_glds = true;

template <typename T, bool GLDS>
static int launch_conv_gemm(ConvArgs& a, hipStream_t s) {
  const ConvGeom& g = a.g;
  const int sub = g.mode == 1 ? g.stride : 1;
  const int sh = (g.OH + sub - 1) / sub, sw = (g.OW + sub - 1) / sub;
  a.nty = (sh + CONV_TH - 1) / CONV_TH;
  a.ntx = (sw + CONV_TW - 1) / CONV_TW;
  const int gm = g.B * sub * sub * a.nty * a.ntx;
  if (gm == 0) return UEGAN_OK;
  dim3 block(256);
  const int bn_idx = a.N > 64 ? 3 : (a.N > 32 ? 2 : (a.N > 16 ? 1 : 0));
  const double rows = g.mode == 0 ? (double)g.B * g.OH * g.OW : (double)g.B * g.IH * g.IW;   // algorithmic MACs: conv-output pixels
  static const int kBn[4] = {16, 32, 64, 128};
  ProfScope prof(prof_key(0, DT<T>::kDtype == UEGAN_BF16, kBn[bn_idx], 0, 0, 8, GLDS), 2.0 * rows * a.N * (double)(g.KH * g.KW * g.C), s,
                 sizeof(T) * (rows * a.N + (double)g.B * g.IH * g.IW * g.C));
  const int small_grid = g_tuning[UEGAN_TUNE_SMALL_GRID];
  if (a.N > 64 && gm * ((a.N + 127) / 128) < small_grid) {         // small maps: 64-channel blocks so the grid covers the chip
    dim3 grid(gm, (a.N + 63) / 64);
    hipLaunchKernelGGL((conv_gemm_kernel<T, 64, 2, 2, GLDS>), grid, block, 0, s, a);
  } else if (a.N > 64) {
    dim3 grid(gm, (a.N + 127) / 128);
    hipLaunchKernelGGL((conv_gemm_kernel<T, 128, 2, 2, GLDS>), grid, block, 0, s, a);
  } else if (a.N > 32) {
    dim3 grid(gm, 1);
    hipLaunchKernelGGL((conv_gemm_kernel<T, 64, 2, 2, GLDS>), grid, block, 0, s, a);
  } else if (a.N > 16) {
    dim3 grid(gm, 1);
    hipLaunchKernelGGL((conv_gemm_kernel<T, 32, 4, 1, GLDS>), grid, block, 0, s, a);
  } else {
    dim3 grid(gm, 1);
    hipLaunchKernelGGL((conv_gemm_kernel<T, 16, 4, 1, GLDS>), grid, block, 0, s, a);
  }
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

template <typename T>
static int dispatch_conv_gemm(ConvArgs& a, hipStream_t s) {
  const ConvGeom& g = a.g;
  // patch-resident kernel: stride 1 and every 64-wide (bf16) K step fully populated; thin-channel layers (3-channel
  // images, 1/3-channel heads, 32-channel full-resolution layers) pack several taps per K step in the generic kernel
  constexpr int BKE = CONV_ROWB / (int)sizeof(T);
  if (g_use_patch && g_use_glds && g.KH == g.KW && g.stride == 2 && g.mode == 0 && sizeof(T) == 2 && g.C == 32) {
    const int rc = conv_s2fwd_run(a, DT<T>::kDtype, s);      // 32-channel stride-2 forwards (D.d2, G.enc2): pixel-pair rows
    if (rc != 1) return rc;
  }
  if (g_use_patch && g_use_glds && g.KH == g.KW && g.C % BKE == 0) {
    int ks = 0;
    if (g.stride == 1) {
      // 1x1 convs (the attention modules' fuse conv, the decoder's upsample convs; pad 0): plain GEMMs -- the patch is the tile itself
      if (g.KH == 1 && g.pad == 0) ks = 1;
      else if (g.KH == 3 || g.KH == 5 || g.KH == 7) ks = g.KH;
    } else if (g.stride == 2 && g.mode == 1) {     // stride-2 dgrad: per parity class a stride-1 problem with (K+1)/2 taps
      if (g.KH == 3 || g.KH == 5 || g.KH == 7) ks = (g.KH + 1) / 2;
    }
    if (ks == 3 && g.stride == 1) {                 // wide layers on maps that fill 256 x 256 tiles
      int rc = conv_tall_run(a, DT<T>::kDtype, s);
      if (rc != 1) return rc;
      rc = conv_wide_run(a, DT<T>::kDtype, s);
      if (rc != 1) return rc;
      rc = conv_interior_run(a, DT<T>::kDtype, s);      // (sets a.border_only: the patch launch below takes the frame with the mirrored images)
      if (rc != 1 && rc != UEGAN_OK) return rc;
    }
    if (ks) return patch_run<T>(a, s, ks);
    const int rc = conv_s2fwd_run(a, DT<T>::kDtype, s);
    if (rc != 1) return rc;
  }
  return g_use_glds ? launch_conv_gemm<T, true>(a, s) : launch_conv_gemm<T, false>(a, s);
}

// ----------------------------------------------------------------------------------------------------
// wgrad: dW[co][kk] = sum_pixels dz[pix][co] * gather(pix, kk), split over pixel ranges (split-K), partials
// to workspace, then a reduce kernel that sums the splits, scales and permutes to OIHW fp32.
// ----------------------------------------------------------------------------------------------------
struct WgradArgs {
  ConvGeom g;          // forward gather geometry (mode 0): rows = conv outputs, source = conv input
  const void* in1;
  const void* in2;
  const void* dz;      // [B][OH][OW][zC]
  float* ws;           // [nsplit][N][ktot]
  int N, zC, ktot;     // N = rows computed (true Cout), zC = channel stride of dz (padded Cout), ktot = KH*KW*C (padded C)
  int WS, WSlog, R;    // pixel strip: WS columns (power of two) x R rows = 32 slots
  int nxb, nyb;        // strips per row / per image
  int steps_total, steps_per_split;
};

constexpr int WG_BK = 128;   // kk columns per block

// In-register transpose of an E x E block of 16-bit (E=8) or 32-bit (E=4) elements held as E 16-byte rows.
__device__ __forceinline__ void transpose_chunks(const u32x4 (&in)[4], u32x4 (&out)[4]) {   // fp32: 4x4
  out[0] = u32x4{in[0].x, in[1].x, in[2].x, in[3].x};
  out[1] = u32x4{in[0].y, in[1].y, in[2].y, in[3].y};
  out[2] = u32x4{in[0].z, in[1].z, in[2].z, in[3].z};
  out[3] = u32x4{in[0].w, in[1].w, in[2].w, in[3].w};
}
__device__ __forceinline__ void transpose_chunks(const u32x4 (&in)[8], u32x4 (&out)[8]) {   // bf16: 8x8
  // in[p] = 8 channels of pixel p (dword d holds channels 2d, 2d+1); out[c] = 8 pixels of channel c
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const uint32_t a = in[2 * pp][d], b = in[2 * pp + 1][d];
      lo[pp] = (a & 0xffffu) | (b << 16);            // channel 2d   of pixels 2pp, 2pp+1
      hi[pp] = (a >> 16) | (b & 0xffff0000u);        // channel 2d+1
    }
    out[2 * d] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    out[2 * d + 1] = u32x4{hi[0], hi[1], hi[2], hi[3]};
  }
}

// wgrad block: BN output-channel rows x 128 kk columns, reduction over a range of pixel strips (split-K).
// One K step = 128 bytes of pixels per row (64 bf16 / 32 fp32 pixel slots).  Both operands arrive pixel-major from HBM
// (NHWC) but MFMA wants the reduction index contiguous per lane, so each thread loads an E x E block (E pixels x one
// 16-byte channel chunk), transposes it in registers and writes E 16-byte rows [channel][E pixels] into the swizzled
// LDS tile.  Lane mapping: the 8 lanes of a ds_write_b128 lane group hold 8 different pixel groups of one channel chunk,
// which makes the transposed writes bank-conflict free.  Two LDS buffers, one barrier per step.
template <typename T, int BN>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgradArgs a) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int ROWB = CONV_ROWB;
  constexpr int NPIX = ROWB / (int)sizeof(T);                 // pixel slots per step (64 / 32)
  constexpr int NCHUNK = Mma<T>::NCHUNK;
  constexpr int NSUB = NPIX / 32;
  constexpr int ZCH = BN / EPC, XCH = WG_BK / EPC;            // channel chunks of the two tiles
  constexpr int NUNIT = 8 * (ZCH + XCH);                      // 8 pixel groups x chunks
  constexpr int NU = (NUNIT + 255) / 256;
  constexpr int WZ = BN >= 64 ? 2 : 1, WX = 4 / WZ;           // wave grid (co x kk)
  constexpr int TN = BN / WZ / 16, TM = WG_BK / WX / 16;
  constexpr int BUFB = (BN + WG_BK) * ROWB;
  static_assert(TN >= 1 && TM >= 1 && NPIX / EPC == 8, "tile");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFB];

  const ConvGeom& g = a.g;
  const T* in1 = static_cast<const T*>(a.in1);
  const T* in2 = static_cast<const T*>(a.in2);
  const T* dz = static_cast<const T*>(a.dz);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wz = wave / WX, wx = wave % WX;
  const int kk_base = blockIdx.x * WG_BK, n_base = blockIdx.y * BN, split = blockIdx.z;
  int s_begin = split * a.steps_per_split;
  int s_end = s_begin + a.steps_per_split;
  if (s_end > a.steps_total) s_end = a.steps_total;

  // my units: unit id = u*256 + tid -> pixel group pq = id & 7, chunk index ch = id >> 3 (dz chunks first, then x chunks)
  int u_kind[NU], u_pq[NU], u_row0[NU], u_ty[NU], u_tx[NU], u_c[NU];    // kind: 0 dz, 1 x, 2 idle
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int id = u * 256 + tid;
    u_pq[u] = id & 7;
    const int ch = id >> 3;
    u_kind[u] = ch < ZCH ? 0 : (ch < ZCH + XCH ? 1 : 2);
    const int lc = ch < ZCH ? ch : ch - ZCH;
    u_row0[u] = (ch < ZCH ? 0 : BN) + lc * EPC;               // first LDS row (channel) of the unit
    u_c[u] = -1; u_ty[u] = 0; u_tx[u] = 0;
    if (u_kind[u] == 0) {
      u_c[u] = n_base + lc * EPC;                              // dz channel
      if (u_c[u] >= a.zC) u_c[u] = -1;
    } else if (u_kind[u] == 1) {
      const int kk = kk_base + lc * EPC;
      if (kk < a.ktot) {
        const int tap = kk / g.C;
        u_c[u] = kk - tap * g.C;
        u_ty[u] = tap / g.KW;
        u_tx[u] = tap - u_ty[u] * g.KW;
      }
    }
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 regs[NU][EPC];

  auto load_units = [&](int s) {
    const int xb = s % a.nxb;
    const int t = s / a.nxb;
    const int yb = t % a.nyb;
    const int b = t / a.nyb;
    const int oy0 = yb * a.R, ox0 = xb * a.WS;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int p = 0; p < EPC; ++p) {
        const int slot = u_pq[u] * EPC + p;
        const int oy = oy0 + (slot >> a.WSlog), ox = ox0 + (slot & (a.WS - 1));
        const bool pv = oy < g.OH && ox < g.OW && u_c[u] >= 0;
        const void* src = g_zero16;
        if (pv) {
          if (u_kind[u] == 0) {
            src = dz + (((size_t)b * g.OH + oy) * g.OW + ox) * a.zC + u_c[u];
          } else if (u_kind[u] == 1) {
            const int sy = src_coord(g, oy, u_ty[u], 0, g.IH, g.OH);
            const int sx = src_coord(g, ox, u_tx[u], 0, g.IW, g.OW);
            if (sy >= 0 && sx >= 0) {
              const size_t pix = ((size_t)b * g.IH + sy) * g.IW + sx;
              const int c = u_c[u];
              src = (c < g.C1) ? (const void*)(in1 + pix * g.C1 + c) : (const void*)(in2 + pix * g.C2 + (c - g.C1));
            }
          }
        }
        regs[u][p] = *reinterpret_cast<const u32x4*>(src);
      }
    }
  };
  auto commit = [&](unsigned char* buf) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (u_kind[u] == 2) continue;
      u32x4 tr[EPC];
      transpose_chunks(regs[u], tr);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int row = u_row0[u] + e;
        *reinterpret_cast<u32x4*>(buf + row * ROWB + ((u_pq[u] ^ ((row >> 1) & 7)) << 4)) = tr[e];
      }
    }
  };

  if (s_begin < s_end) {
    load_units(s_begin);
    commit(lds);
  }
  const int fr = lane & 15, fg = lane >> 4;
  for (int s = s_begin; s < s_end; ++s) {
    unsigned char* cur = lds + ((s - s_begin) & 1) * BUFB;
    unsigned char* nxt = lds + ((s - s_begin + 1) & 1) * BUFB;
    __syncthreads();
    if (s + 1 < s_end) load_units(s + 1);
#pragma unroll
    for (int ksub = 0; ksub < NSUB; ++ksub) {
      u32x4 zf[TN][NCHUNK], xf[TM][NCHUNK];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wz * (BN / WZ) + i * 16 + fr;
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
          const int q = ksub * 4 + c * 4 * (NCHUNK - 1) + fg;
          zf[i][c] = *reinterpret_cast<const u32x4*>(cur + row * ROWB + ((q ^ ((row >> 1) & 7)) << 4));
        }
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = BN + wx * (WG_BK / WX) + j * 16 + fr;
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
          const int q = ksub * 4 + c * 4 * (NCHUNK - 1) + fg;
          xf[j][c] = *reinterpret_cast<const u32x4*>(cur + row * ROWB + ((q ^ ((row >> 1) & 7)) << 4));
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mma<T>::step(zf[i], xf[j], acc[i][j]);
    }
    if (s + 1 < s_end) commit(nxt);
  }

  // partial tile -> workspace [split][N][ktot]; D rows = co, cols = kk
  float* ws = a.ws + (size_t)split * a.N * a.ktot;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int kk = kk_base + wx * (WG_BK / WX) + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_base + wz * (BN / WZ) + i * 16 + (lane >> 4) * 4 + r;
        if (n < a.N && kk < a.ktot) ws[(size_t)n * a.ktot + kk] = acc[i][j][r];
      }
    }
}

// sum splits, scale, permute [co][(ty,tx,c_padded)] -> OIHW [co][ci][ty][tx] (padding channels dropped).  Partials are
// [nsplit][pstride] with the N*ktot weight sums first; when dbias is given, N bias sums follow (unscaled).
// Block = 32 consecutive elements x 8 split lanes (fixed summation order: deterministic).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* ws, float* dw, float* dbias, const float* scale, int nsplit, int N,
                                                            int C, int Cin_w, int KH, int KW, size_t pstride, int acc, int accb, int Cin_row) {
  // Cin_row: input channels per row of the OIHW destination (>= Cin_w: the convolution may use a column slice of a wider master weight)
  __shared__ float red[8][32];
  const int ktot = KH * KW * C;
  const size_t nw = (size_t)N * ktot, total = nw + (dbias ? (size_t)N : 0);
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + e;
  float p[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < total) {
    int k = sl;
    for (; k + 24 < nsplit; k += 32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] += ws[(size_t)(k + 8 * u) * pstride + i];
    }
    for (; k < nsplit; k += 8) p[0] += ws[(size_t)k * pstride + i];
  }
  red[sl][e] = (p[0] + p[1]) + (p[2] + p[3]);
  __syncthreads();
  if (sl == 0 && i < total) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += red[u][e];
    if (i >= nw) {
      dbias[i - nw] = s + (accb ? dbias[i - nw] : 0.f);
    } else {
      const int n = (int)(i / ktot), kk = (int)(i - (size_t)n * ktot);
      const int tap = kk / C, c = kk - tap * C;
      if (c < Cin_w) {
        float* o = dw + ((size_t)n * Cin_row + c) * (KH * KW) + tap;
        *o = s * (scale ? *scale : 1.f) + (acc ? *o : 0.f);      // acc: gradient accumulation into a live bucket (beta = 1)
      }
    }
  }
}

#include "wgrad_tr.h"
#include "conv_stream.h"

// dbias[c] = sum over pixels of dz[pix][c] for c < C (dz channel stride zC, a multiple of one 16-byte chunk).
// Two stages (same-address fp32 atomics from ~1000 blocks serialise in L2): per-block partial sums -> part[block][zC],
// then one small kernel sums the <= BIAS_BLOCKS partials per channel.
constexpr int BIAS_BLOCKS = 512;
template <typename T>
__global__ void bias_grad_partial_kernel(const T* dz, float* part, size_t npix, int zC) {
  constexpr int V = DT<T>::EPC;
  __shared__ float red[V][256];
  const int nch = zC / V;                    // channel chunks per pixel
  int cp = 1;
  while (cp < nch && cp < 64) cp <<= 1;      // chunk lanes per block
  const int rows = 256 / cp;
  const int c_lane = threadIdx.x % cp, r_lane = threadIdx.x / cp;
  for (int ch0 = 0; ch0 < nch; ch0 += cp) {
    const int ch = ch0 + c_lane;
    float s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.f;
    if (ch < nch)
      for (size_t p = (size_t)blockIdx.x * rows + r_lane; p < npix; p += (size_t)gridDim.x * rows) {
        float v[V];
        Vec<T, V>::ld(dz + p * zC + ch * V, v);
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] += v[e];
      }
#pragma unroll
    for (int e = 0; e < V; ++e) red[e][threadIdx.x] = s[e];
    __syncthreads();
    for (int half = rows >> 1; half > 0; half >>= 1) {      // tree over the pixel lanes (rows is a power of two)
      if (r_lane < half) {
#pragma unroll
        for (int e = 0; e < V; ++e) red[e][threadIdx.x] += red[e][threadIdx.x + half * cp];
      }
      __syncthreads();
    }
    if (r_lane == 0 && ch < nch) {
#pragma unroll
      for (int e = 0; e < V; ++e) part[(size_t)blockIdx.x * zC + ch * V + e] = red[e][c_lane];
    }
    __syncthreads();
  }
}
// one block per channel: 256 threads split the partials
__global__ void bias_grad_final_kernel(const float* part, float* dbias, int nblocks, int C, int zC, int acc) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < nblocks; k += blockDim.x) s += part[(size_t)k * zC + c];
  s = block_sum(s, red);
  if (threadIdx.x == 0) dbias[c] = s + (acc ? dbias[c] : 0.f);
}

// ----------------------------------------------------------------------------------------------------
// weight packing: OIHW fp32 [Cout][Cin][KH][KW] -> ohwi [Cout_p][Kp] (k = (kh,kw,ci_padded)) and
//                                                  ihwo [Cin_p ][Kp2] (k = (kh,kw,co_padded)), zero padded
// ----------------------------------------------------------------------------------------------------
// ohwi_lo (optional): what the rounding of each OHWI element left, rn(w - rn(w)) -- the weights as a hi + lo pair (uegan_conv2d_fwd_ex);
// dup 1: input channels [Cin, 2 Cin) of the OHWI copies repeat [0, Cin) (a source that carries ITS lo plane in those channels, uegan_nchw_to_nhwc_pair);
// dup 2: they hold the LO part of [0, Cin) instead (the pair inside ONE matrix: a kernel that reads the source's channels twice multiplies by both)
template <typename T>
__device__ __forceinline__ void st_pair(T* hi, T* lo, size_t i, float v, bool as_lo = false) {
  if (as_lo) {
    T h;
    DT<T>::st(&h, v);
    v -= DT<T>::ld(&h);
  }
  DT<T>::st(hi + i, v);
  if (lo) DT<T>::st(lo + i, v - DT<T>::ld(hi + i));
}
template <typename T>
__global__ void pack_weights_kernel(const float* w, T* ohwi, T* ihwo, int Cout, int Cin, int KH, int KW, int Cout_p, int Cin_p, int Kp,
                                    int Kp2, int Cin_row, T* ohwi_lo = nullptr, int dup = 0) {
  const int taps = KH * KW;
  const size_t n1 = (size_t)Cout_p * Kp, n2 = ihwo ? (size_t)Cin_p * Kp2 : 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (size_t)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int co = (int)(i / Kp), kk = (int)(i - (size_t)co * Kp);
      float v = 0.f;
      bool as_lo = false;
      if (co < Cout && kk < taps * Cin_p) {
        const int tap = kk / Cin_p;
        int ci = kk - tap * Cin_p;
        if (dup && ci >= Cin && ci < 2 * Cin) { ci -= Cin; as_lo = dup == 2; }
        if (ci < Cin) v = w[((size_t)co * Cin_row + ci) * taps + tap];
      }
      st_pair<T>(ohwi, ohwi_lo, i, v, as_lo);
    } else {
      const size_t j = i - n1;
      const int ci = (int)(j / Kp2), kk = (int)(j - (size_t)ci * Kp2);
      float v = 0.f;
      if (ci < Cin && kk < taps * Cout_p) {
        const int tap = kk / Cout_p, co = kk - tap * Cout_p;
        if (co < Cout) v = w[((size_t)co * Cin_row + ci) * taps + tap];
      }
      DT<T>::st(ihwo + j, v);
    }
  }
}

// All conv weights of one optimizer in ONE launch (uegan_pack_weights_multi): entry e owns the element range [start, start + n1 + n2) of
// the concatenated (OHWI, IHWO) destinations; a thread finds its entry by bisection over the (<= a few hundred) range starts.
template <typename T>
__global__ void pack_weights_multi_kernel(const uegan_pack_entry* __restrict__ tab, int n_entries, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tab[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const uegan_pack_entry& e = tab[lo];
    const long long r = i - e.start;
    const int taps = e.KH * e.KW;
    const long long n1 = (long long)e.Cout_pad * e.Kp;
    const float* w = e.w_oihw;
    if (r < n1) {
      const int co = (int)(r / e.Kp), kk = (int)(r - (long long)co * e.Kp);
      float v = 0.f;
      bool as_lo = false;
      if (co < e.Cout && kk < taps * e.Cin_pad) {
        const int tap = kk / e.Cin_pad;
        int ci = kk - tap * e.Cin_pad;
        if ((e.flags & 3) && ci >= e.Cin && ci < 2 * e.Cin) { ci -= e.Cin; as_lo = (e.flags & 3) == 2; }
        if (ci < e.Cin) v = w[((size_t)co * e.Cin_total + ci) * taps + tap];
      }
      st_pair<T>(static_cast<T*>(e.w_ohwi), static_cast<T*>(e.w_ohwi_lo), (size_t)r, v, as_lo);
    } else {
      const long long j = r - n1;
      const int ci = (int)(j / e.Kp2), kk = (int)(j - (long long)ci * e.Kp2);
      float v = 0.f;
      if (ci < e.Cin && kk < taps * e.Cout_pad) {
        const int tap = kk / e.Cout_pad, co = kk - tap * e.Cout_pad;
        if (co < e.Cout) v = w[((size_t)co * e.Cin_total + ci) * taps + tap];
      }
      DT<T>::st(static_cast<T*>(e.w_ihwo) + j, v);
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// Direct (scalar) kernels: ground truth on the GPU for the MFMA path; never the default.
// ----------------------------------------------------------------------------------------------------
template <typename T>
__global__ void conv_direct_kernel(ConvArgs a) {
  const ConvGeom& g = a.g;
  const T* in1 = static_cast<const T*>(a.in1);
  const T* in2 = static_cast<const T*>(a.in2);
  const T* w = static_cast<const T*>(a.w);
  T* out = static_cast<T*>(a.out);
  const size_t total = (size_t)g.B * g.OH * g.OW * a.N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / a.N), n = (int)(idx - (size_t)m * a.N);
    const int ohw = g.OH * g.OW;
    const int b = m / ohw, r = m - b * ohw, oy = r / g.OW, ox = r - oy * g.OW;
    const float scale = a.scale ? a.scale[a.scale_group ? b / a.scale_group : 0] : 1.f;
    float acc = 0.f;
    const int nimg = (g.mode == 1 && g.pad_mode == UEGAN_PAD_REFLECT) ? 3 : 1;
    for (int iy = 0; iy < nimg; ++iy)
      for (int ix = 0; ix < nimg; ++ix)
        for (int ty = 0; ty < g.KH; ++ty) {
          const int sy = src_coord(g, oy, ty, iy, g.IH, g.OH);
          if (sy < 0) continue;
          for (int tx = 0; tx < g.KW; ++tx) {
            const int sx = src_coord(g, ox, tx, ix, g.IW, g.OW);
            if (sx < 0) continue;
            const size_t pix = ((size_t)b * g.IH + sy) * g.IW + sx;
            const T* wp = w + (size_t)n * a.Kp + (size_t)(ty * g.KW + tx) * g.C;
            for (int c = 0; c < g.C; ++c) {
              const float xv = (c < g.C1) ? DT<T>::ld(in1 + pix * g.C1 + c) : DT<T>::ld(in2 + pix * g.C2 + (c - g.C1));
              acc += xv * DT<T>::ld(wp + c);
            }
          }
        }
    float v = acc * scale + ((a.bias && n < a.nbias) ? a.bias[n] : 0.f);
    T* p = (a.out2 && n >= a.n_out1) ? static_cast<T*>(a.out2) + (size_t)m * (a.N - a.n_out1) + (n - a.n_out1)
                                     : out + (size_t)m * (a.out2 ? a.n_out1 : a.N) + n;
    DT<T>::st(p, apply_act_ext(v, a.act));
  }
}

// one thread per (co, kk): loops over all pixels (slow; tests only)
template <typename T>
__global__ void wgrad_direct_kernel(WgradArgs a, float* dw, const float* scale_p, int Cin_w, int accum, int Cin_row) {
  const ConvGeom& g = a.g;
  const T* in1 = static_cast<const T*>(a.in1);
  const T* in2 = static_cast<const T*>(a.in2);
  const T* dz = static_cast<const T*>(a.dz);
  const size_t total = (size_t)a.N * a.ktot;
  const float scale = scale_p ? *scale_p : 1.f;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx / a.ktot), kk = (int)(idx - (size_t)n * a.ktot);
    const int tap = kk / g.C, c = kk - tap * g.C, ty = tap / g.KW, tx = tap - ty * g.KW;
    if (c >= Cin_w) continue;
    float acc = 0.f;
    for (int b = 0; b < g.B; ++b)
      for (int oy = 0; oy < g.OH; ++oy) {
        const int sy = src_coord(g, oy, ty, 0, g.IH, g.OH);
        if (sy < 0) continue;
        for (int ox = 0; ox < g.OW; ++ox) {
          const int sx = src_coord(g, ox, tx, 0, g.IW, g.OW);
          if (sx < 0) continue;
          const size_t pix = ((size_t)b * g.IH + sy) * g.IW + sx;
          const float xv = (c < g.C1) ? DT<T>::ld(in1 + pix * g.C1 + c) : DT<T>::ld(in2 + pix * g.C2 + (c - g.C1));
          acc += xv * DT<T>::ld(dz + (((size_t)b * g.OH + oy) * g.OW + ox) * a.zC + n);
        }
      }
    float* o = dw + ((size_t)n * Cin_row + c) * (g.KH * g.KW) + tap;
    *o = acc * scale + (accum ? *o : 0.f);
  }
}

template <typename T, int V>
__global__ void act_bwd_kernel(const T* g, const T* g2, const T* g3, const T* a, T* dz, size_t n, int act) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * blockDim.x * V) {
    float gv[V], av[V];
    Vec<T, V>::ld(g + i, gv);
    if (g2) {                 // further consumers of the activation: the sum of their gradients never exists in memory
      Vec<T, V>::ld(g2 + i, av);
#pragma unroll
      for (int e = 0; e < V; ++e) gv[e] += av[e];
    }
    if (g3) {
      Vec<T, V>::ld(g3 + i, av);
#pragma unroll
      for (int e = 0; e < V; ++e) gv[e] += av[e];
    }
    Vec<T, V>::ld(a + i, av);
#pragma unroll
    for (int e = 0; e < V; ++e) gv[e] *= act_grad_from_out_ext(av[e], act);
    Vec<T, V>::st(dz + i, gv);
  }
}

// One 16-byte chunk of a gradient that arrives on the PADDED grid of a reflection-padded consumer (conv_flat_kernel, head_dgrad_mfma_kernel:
// [B][H + 2 pad][W + 2 pad][C]): the adjoint of nn.ReflectionPad2d (models.py:80) adds the mirror images -- up to 2 x 2 sources on the border
// ring, one elsewhere -- while the activation backward reads the gradient, so no fold pass and no folded copy exist.  pad = 0: a plain tensor.
// (two steps, so that a caller can issue the direct loads of several pixels back to back before any of the rare mirror terms)
__device__ __forceinline__ size_t padded_offset(int pad, int b, int y, int x, int H, int W, int C, int c) {
  return (((size_t)b * (H + 2 * pad) + (y + pad)) * (W + 2 * pad) + (x + pad)) * C + c;
}
__device__ __forceinline__ bool on_mirror_ring(int pad, int y, int x, int H, int W) {
  return pad > 0 && ((y >= 1 && y <= pad) || (y <= H - 2 && y >= H - 1 - pad) || (x >= 1 && x <= pad) || (x <= W - 2 && x >= W - 1 - pad));
}
// v += the mirror images of pixel (y, x) (everything but the direct source).  All candidate loads are issued before the first sum (predicated:
// a lane without that image issues nothing), so a ring pixel costs one memory round trip, not one per image.
template <typename T, int V>
__device__ __forceinline__ void add_mirrors(const T* __restrict__ src, int pad, int b, int y, int x, int H, int W, int C, int c, float (&v)[V]) {
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  int ys[3], xs[3], ny = 1, nx = 1;
  ys[0] = y + pad; xs[0] = x + pad; ys[1] = ys[2] = ys[0]; xs[1] = xs[2] = xs[0];
  if (y >= 1 && y <= pad) ys[ny++] = pad - y;                                  // mirrored across row 0
  if (y <= H - 2 && y >= H - 1 - pad) ys[ny++] = pad + 2 * (H - 1) - y;        // mirrored across row H-1
  if (x >= 1 && x <= pad) xs[nx++] = pad - x;
  if (x <= W - 2 && x >= W - 1 - pad) xs[nx++] = pad + 2 * (W - 1) - x;
  float t[8][V];
#pragma unroll
  for (int k = 1; k < 9; ++k) {
    const int iy = k / 3, ix = k - 3 * iy;
#pragma unroll
    for (int e = 0; e < V; ++e) t[k - 1][e] = 0.f;
    if (iy < ny && ix < nx) Vec<T, V>::ld(src + (((size_t)b * Hp + ys[iy]) * Wp + xs[ix]) * C + c, t[k - 1]);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] += t[k][e];
}
template <typename T, int V>
__device__ __forceinline__ void ld_folded(const T* __restrict__ src, int pad, int b, int y, int x, int H, int W, int C, int c, float (&v)[V]) {
  Vec<T, V>::ld(src + padded_offset(pad, b, y, x, H, W, C, c), v);
  if (on_mirror_ring(pad, y, x, H, W)) add_mirrors<T, V>(src, pad, b, y, x, H, W, C, c, v);
}

// dz = (g + g2) * act'(a) with g / g2 optionally on padded grids (above); one thread per 16-byte chunk
template <typename T>
__global__ void __launch_bounds__(256) act_bwd_p_kernel(const T* g, int pad_g, const T* g2, int pad_g2, const T* a, T* dz, int B, int H, int W, int C, int act) {
  constexpr int V = DT<T>::EPC;
  const int cpp = C / V;
  const unsigned total = (unsigned)B * H * W * cpp;        // (< 2^32: checked by the launcher; 32-bit divisions)
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int q = (int)(i % (unsigned)cpp);
    unsigned r = i / (unsigned)cpp;
    const int x = (int)(r % (unsigned)W); r /= (unsigned)W;
    const int y = (int)(r % (unsigned)H);
    const int b = (int)(r / (unsigned)H);
    float gv[V], t[V];
    ld_folded<T, V>(g, pad_g, b, y, x, H, W, C, q * V, gv);
    if (g2) {
      ld_folded<T, V>(g2, pad_g2, b, y, x, H, W, C, q * V, t);
#pragma unroll
      for (int e = 0; e < V; ++e) gv[e] += t[e];
    }
    const size_t o = (((size_t)b * H + y) * W + x) * C + q * V;
    Vec<T, V>::ld(a + o, t);
#pragma unroll
    for (int e = 0; e < V; ++e) gv[e] *= act_grad_from_out_ext(t[e], act);
    Vec<T, V>::st(dz + o, gv);
  }
}

// ----------------------------------------------------------------------------------------------------
// Activation backward of a spectral-normalised trunk conv inside the batched discriminator pass (fused.py).  Image group r of the batch was
// convolved with W / sigma_r; with dz_raw = (g + g2) * act'(y) the weight side needs, per group,
//     G_r = wgrad(x_r, dz_raw_r) / sigma_r            and   dW += G_r - (<G_r, W> / sigma_r) u_r v_r^T        (torch spectral_norm, u, v constant)
// Writing dz = dz_raw / sigma_r into the stored gradient makes ONE weight-gradient launch over all groups give sum_r G_r and lets the data
// gradient run without a per-group scale; and because the forward computed W (*) x = sigma_r (z - bias), the projection coefficient is a
// reduction over the activation instead of a dot product over the weights:   <G_r, W> / sigma_r = sum_{pixels of r, c} dz (z - bias_c).
// This kernel stores dz and emits per-block partials of c_r and of the bias gradient sum dz_raw (folded in a fixed order by
// sn_grad_finish_kernel).  15 weight-gradient + 15 dot + 15 rank-1 launches of a D update become 5 + 0 + 5.
// ----------------------------------------------------------------------------------------------------
constexpr int SNB = 256;     // partial blocks per group (round 5: 128 + up to 128 ring blocks); each thread keeps UNR pixels in flight (the 100-MB maps of d1 need ~10 MB of loads in the air)
// Padded-grid gradients (pad_g / pad_g2 > 0): a pixel on the border RING (within max pad of a border) also receives mirror images, which cost
// dependent loads with per-lane trip counts -- spread over the map they would sit in half of all wave iterations (measured 1.7 - 3 x the plain
// kernel).  The work is therefore split inside one launch: blocks [0, bx_main) take every pixel that is NOT on the ring (one load per source,
// the plain kernel's speed), blocks [bx_main, gridDim.x) walk a dense enumeration of the ring pixels only.  `ring_all`: maps too small for a
// ring-free interior -- every pixel goes the ring blocks' way.
struct SnRing {
  int pmax, n_row, n_ring, ring_all;      // max pad; 2 pmax W (the row bands); ring pixels per image
};
__device__ __forceinline__ void sn_ring_pixel(const SnRing& r, int k, int H, int W, int& y, int& x) {
  if (r.ring_all) { y = k / W; x = k - y * W; return; }
  const int p = r.pmax;
  if (k < r.n_row) {                       // rows 1 .. p and H-1-p .. H-2, all columns
    const int j = k / W;
    x = k - j * W;
    y = j < p ? 1 + j : H - 1 - p + (j - p);
  } else {                                 // columns 1 .. p and W-1-p .. W-2 of the other rows (0, p+1 .. H-2-p, H-1)
    const int k2 = k - r.n_row, t = k2 / (2 * p), j = k2 - t * (2 * p);
    x = j < p ? 1 + j : W - 1 - p + (j - p);
    y = t == 0 ? 0 : (t == H - 2 * p - 1 ? H - 1 : p + t);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) sn_act_bwd_kernel(const T* g, const T* g2, const T* y, const float* bias, int nbias, const float* inv_sigma, T* dz,
                                                         float* cpart, float* dbpart, long long pix_per_group, int C, int act, int pad_g, int pad_g2,
                                                         int H, int W, int bx_main, SnRing ring) {
  constexpr int V = DT<T>::EPC;
  __shared__ float sh[256][V + 1];
  __shared__ float red[16];
  const int grp = blockIdx.y;
  const int cpp = C / V, pl = 256 / cpp;                 // 16-byte chunks per pixel, pixels per block iteration
  const int q = threadIdx.x % cpp, pr = threadIdx.x / cpp;
  const float inv = inv_sigma[grp];
  const float slope = act == UEGAN_ACT_LRELU ? 0.2f : (act == UEGAN_ACT_RELU ? 0.f : 1.f);
  const float islope = act == UEGAN_ACT_LRELU ? 5.f : 1.f;      // z from y = act(z) (ReLU: z - b only matters where act' != 0)
  float bv[V], dbs[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { bv[e] = q * V + e < nbias ? bias[q * V + e] : 0.f; dbs[e] = 0.f; }
  float csum = 0.f;
  const size_t base = (size_t)grp * pix_per_group * C;
  const bool padded = (pad_g | pad_g2) != 0;
  const int ipg = padded ? (int)(pix_per_group / ((long long)H * W)) : 0;      // images per group
  auto finish = [&](float (&gv)[V], const float (&g2v)[V], const float (&av)[V], size_t o) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float gsum = g2 ? gv[e] + g2v[e] : gv[e];
      const float raw = gsum * (av[e] > 0.f ? 1.f : slope);
      const float z = av[e] > 0.f ? av[e] : av[e] * islope;
      dbs[e] += raw;
      gv[e] = raw * inv;
      csum += gv[e] * (z - bv[e]);
    }
    Vec<T, V>::st(dz + o, gv);
  };
  if ((int)blockIdx.x < bx_main) {
    constexpr int UNR = 4;
    const long long stride = (long long)bx_main * pl;
    for (long long p0 = (long long)blockIdx.x * pl + pr; p0 < pix_per_group; p0 += UNR * stride) {
      float gv[UNR][V], g2v[UNR][V], av[UNR][V];
      bool skip[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long long p = p0 + u * stride;
        skip[u] = false;
        if (p < pix_per_group) {
          const size_t i = base + (size_t)p * C + q * V;
          size_t og = i, og2 = i;
          if (padded) {      // a gradient on the padded grid of its reflection-padded consumer (32-bit arithmetic: a 64-bit division per pixel tripled the kernel's time)
            const unsigned hw = (unsigned)(H * W), pu = (unsigned)p, bl = pu / hw, rem = pu - bl * hw;
            const int yy = (int)(rem / (unsigned)W), xx = (int)(rem - (unsigned)yy * (unsigned)W), bg = grp * ipg + (int)bl;
            skip[u] = on_mirror_ring(ring.pmax, yy, xx, H, W);      // (the ring blocks' pixel: loaded like the others -- no branch around the loads -- and dropped)
            og = padded_offset(pad_g, bg, yy, xx, H, W, C, q * V);
            og2 = padded_offset(pad_g2, bg, yy, xx, H, W, C, q * V);
          }
          Vec<T, V>::ld(g + og, gv[u]);
          if (g2) Vec<T, V>::ld(g2 + og2, g2v[u]);
          Vec<T, V>::ld(y + i, av[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long long p = p0 + u * stride;
        if (p >= pix_per_group) break;
        if (skip[u]) continue;
        finish(gv[u], g2v[u], av[u], base + (size_t)p * C + q * V);
      }
    }
  } else {
    // ring blocks: dense over (image of the group, ring pixel)
    const int nrb = gridDim.x - bx_main, rb = blockIdx.x - bx_main;
    const int total = ipg * ring.n_ring;
    for (int r0 = rb * pl + pr; r0 < total; r0 += nrb * pl) {
      const int bl = r0 / ring.n_ring, k = r0 - bl * ring.n_ring, bg = grp * ipg + bl;
      int yy, xx;
      sn_ring_pixel(ring, k, H, W, yy, xx);
      float gv[V], g2v[V], av[V];
      ld_folded<T, V>(g, pad_g, bg, yy, xx, H, W, C, q * V, gv);
      if (g2) ld_folded<T, V>(g2, pad_g2, bg, yy, xx, H, W, C, q * V, g2v);
      const size_t o = (((size_t)bg * H + yy) * W + xx) * C + q * V;
      Vec<T, V>::ld(y + o, av);
      finish(gv, g2v, av, o);
    }
  }
  csum = block_sum(csum, red);
  const int slot = grp * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0) cpart[slot] = csum;
#pragma unroll
  for (int e = 0; e < V; ++e) sh[threadIdx.x][e] = dbs[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int cq = c / V, ce = c - cq * V;
    float t = 0.f;
    for (int r = 0; r < pl; ++r) t += sh[r * cpp + cq][ce];
    dbpart[(size_t)slot * C + c] = t;
  }
}

// dw -= sum_r c_r u_r v_r^T (c_r folded from its block partials) and db (+)= sum of the bias partials; one launch per layer.
// Blocks [0, nbb) also finish 16 bias channels each (16 threads per channel over interleaved partials, combined in a fixed order).
__global__ void sn_grad_finish_kernel(float* dw, float* db, const float* cpart, const float* dbpart, int nbx, int ngroups, const float* uh,
                                      const float* vh, int rows, int cols, int C, int accb, int nbb) {
  __shared__ float cr[8];
  __shared__ float bsum[16][17];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = wv; r < ngroups; r += 4) {
    float v = 0.f;
    for (int i = lane; i < nbx; i += 64) v += cpart[r * nbx + i];
    v = wave_sum(v);
    if (lane == 0) cr[r] = v;
  }
  const bool bias_block = db && (int)blockIdx.x < nbb;
  if (bias_block) {
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl, np = ngroups * nbx;
    float t = 0.f;
    if (c < rows)
      for (int pp = sl; pp < np; pp += 16) t += dbpart[(size_t)pp * C + c];
    bsum[sl][cl] = t;
  }
  __syncthreads();
  if (bias_block && threadIdx.x < 16) {
    const int c = blockIdx.x * 16 + threadIdx.x;
    if (c < rows) {
      float t = 0.f;
      for (int k = 0; k < 16; ++k) t += bsum[k][threadIdx.x];
      db[c] = accb ? db[c] + t : t;
    }
  }
  const size_t n = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int r0 = (int)(i / cols), c0 = (int)(i - (size_t)r0 * cols);
    float t = 0.f;
    for (int r = 0; r < ngroups; ++r) t += cr[r] * uh[(size_t)r * rows + r0] * vh[(size_t)r * cols + c0];
    dw[i] -= t;
  }
}

// MFMA layout self-test: D = A*B with A = I (16x16 padded in K) and an asymmetric B.
__global__ void selftest_mfma_kernel(float* out) {
  const int lane = threadIdx.x & 63;
  // f32: A[i][k] (k<4): identity block k==i for i<4; B[k][j] = 100*k + j  -> D[i][j] = 100*i + j for i<4
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int ai = lane & 15, ak = lane >> 4;
  acc = mfma_f32(ai == ak ? 1.f : 0.f, 100.f * (lane >> 4) + (lane & 15), acc);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
  // bf16: A[i][k] = (k == i) (K = 32), B[k][j] = (8k + j)/2: exactly representable for the rows that matter
  __attribute__((aligned(16))) unsigned short av[8];
  __attribute__((aligned(16))) unsigned short bv[8];
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * (lane >> 4) + e;
    av[e] = f32_to_bf16((lane & 15) == k ? 1.f : 0.f);
    bv[e] = f32_to_bf16((float)(k * 8 + (lane & 15)) * 0.5f);
  }
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  acc2 = mfma_bf16(*reinterpret_cast<u32x4*>(av), *reinterpret_cast<u32x4*>(bv), acc2);
  for (int r = 0; r < 4; ++r) out[256 + lane * 4 + r] = acc2[r];
  // bf16 32x32x16 (conv_wide.hip): A[i][k] = (k == i) (K = 16); B[k][j] = k + 1, then j + 1  ->  D[i][j] = i + 1 / j + 1 for i < 16
  typedef float f32x16_t __attribute__((ext_vector_type(16)));
  __attribute__((aligned(16))) unsigned short bk[8];
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * (lane >> 5) + e;
    av[e] = f32_to_bf16((lane & 31) == k ? 1.f : 0.f);
    bk[e] = f32_to_bf16((float)(k + 1));
    bv[e] = f32_to_bf16((float)((lane & 31) + 1));
  }
  f32x16_t z16;
  for (int r = 0; r < 16; ++r) z16[r] = 0.f;
  const bf16x8_t a8 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<u32x4*>(av));
#ifdef UEGAN_HALF_FP16
  const f32x16_t d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, __builtin_bit_cast(bf16x8_t, *reinterpret_cast<u32x4*>(bk)), z16, 0, 0, 0);
  const f32x16_t d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, __builtin_bit_cast(bf16x8_t, *reinterpret_cast<u32x4*>(bv)), z16, 0, 0, 0);
#else
  const f32x16_t d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8_t, *reinterpret_cast<u32x4*>(bk)), z16, 0, 0, 0);
  const f32x16_t d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8_t, *reinterpret_cast<u32x4*>(bv)), z16, 0, 0, 0);
#endif
  for (int r = 0; r < 16; ++r) {
    out[512 + lane * 16 + r] = d1[r];
    out[1536 + lane * 16 + r] = d2[r];
  }
}

}  // namespace uegan

using namespace uegan;

// ----------------------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------------------
static int check_desc(const uegan_conv_desc* d) {
  UEGAN_CHECK_ARG(d != nullptr, "conv desc is null");
  UEGAN_CHECK_ARG(d->dtype == UEGAN_F32 || d->dtype == UEGAN_BF16, "bad dtype %d", d->dtype);
  UEGAN_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->C1 > 0 && d->C2 >= 0 && d->Cout > 0, "bad conv dims");
  UEGAN_CHECK_ARG(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "bad conv kernel/stride/pad");
  UEGAN_CHECK_ARG(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
                  "Ho/Wo inconsistent with H,W,pad,K,stride");
  if (d->pad_mode == UEGAN_PAD_REFLECT)
    UEGAN_CHECK_ARG(d->pad < d->H && d->pad < d->W, "reflection pad %d must be smaller than the input (%d x %d)", d->pad, d->H, d->W);
  else
    UEGAN_CHECK_ARG(d->pad_mode == UEGAN_PAD_ZERO, "bad pad mode");
  const int epc = d->dtype == UEGAN_F32 ? 4 : 8;
  UEGAN_CHECK_ARG(d->C1 % epc == 0 && d->C2 % epc == 0 && d->Cout % epc == 0,
                  "tensor channel counts must be multiples of %d (one 16-byte chunk): pad them (C1=%d C2=%d Cout=%d)", epc, d->C1, d->C2, d->Cout);
  UEGAN_CHECK_ARG(d->Cin_w >= 0 && d->Cin_w <= d->C1 + d->C2 && d->Cout_w >= 0 && d->Cout_w <= d->Cout, "bad true weight dims");
  UEGAN_CHECK_ARG(d->stride <= 2, "stride > 2 is not built");
  UEGAN_CHECK_ARG(d->scale_group >= 0, "bad scale_group");
  UEGAN_CHECK_ARG(d->Cin_total == 0 || d->Cin_total >= (d->Cin_w ? d->Cin_w : d->C1 + d->C2), "Cin_total must cover the input channels used");
  return UEGAN_OK;
}
static inline int cin_w(const uegan_conv_desc* d) { return d->Cin_w ? d->Cin_w : d->C1 + d->C2; }
static inline int cin_row(const uegan_conv_desc* d) { return d->Cin_total ? d->Cin_total : cin_w(d); }
static inline int cout_w(const uegan_conv_desc* d) { return d->Cout_w ? d->Cout_w : d->Cout; }

static ConvGeom fwd_geom(const uegan_conv_desc* d) {
  ConvGeom g;
  g.B = d->B; g.IH = d->H; g.IW = d->W; g.C1 = d->C1; g.C2 = d->C2; g.C = d->C1 + d->C2;
  g.OH = d->Ho; g.OW = d->Wo; g.KH = d->KH; g.KW = d->KW; g.stride = d->stride; g.pad = d->pad; g.pad_mode = d->pad_mode;
  g.mode = 0;
  return g;
}

extern "C" int uegan_set_tuning(int knob, int value, int* previous) {
  UEGAN_CHECK_ARG(knob >= 0 && knob < UEGAN_TUNE_COUNT, "unknown tuning knob %d", knob);
  if (previous) *previous = g_tuning[knob];
  g_tuning[knob] = value;
  return UEGAN_OK;
}

extern "C" int uegan_set_conv_impl(int impl) {
  int old = g_conv_impl;
  g_use_glds = true; g_use_patch = true; g_use_heads = true; g_use_wgtr = true; g_use_stream = true;
  if (impl == UEGAN_IMPL_MFMA_REGSTAGE) { g_use_glds = false; g_use_heads = false; g_use_wgtr = false; g_conv_impl = UEGAN_IMPL_MFMA; }
  else if (impl == UEGAN_IMPL_MFMA_GENERIC) { g_use_patch = false; g_use_heads = false; g_use_wgtr = false; g_use_stream = false; g_conv_impl = UEGAN_IMPL_MFMA; }
  else g_conv_impl = impl;
  return old;
}

extern "C" int64_t uegan_packed_k(int64_t k) { return (k + 7) / 8 * 8; }

extern "C" int uegan_pack_weights(int dtype, const float* w_oihw, int Cout, int Cin, int KH, int KW, int Cout_pad, int Cin_pad, void* w_ohwi,
                                  void* w_ihwo, uegan_stream_t stream) {
  return uegan_pack_weights_slice(dtype, w_oihw, Cout, Cin, Cin, KH, KW, Cout_pad, Cin_pad, w_ohwi, w_ihwo, stream);
}

extern "C" int uegan_pack_weights_slice(int dtype, const float* w_oihw, int Cout, int Cin, int Cin_total, int KH, int KW, int Cout_pad,
                                        int Cin_pad, void* w_ohwi, void* w_ihwo, uegan_stream_t stream) {
  return uegan_pack_weights_pair(dtype, w_oihw, Cout, Cin, Cin_total, KH, KW, Cout_pad, Cin_pad, w_ohwi, w_ihwo, nullptr, 0, stream);
}

extern "C" int uegan_pack_weights_pair(int dtype, const float* w_oihw, int Cout, int Cin, int Cin_total, int KH, int KW, int Cout_pad, int Cin_pad,
                                       void* w_ohwi, void* w_ihwo, void* w_ohwi_lo, int dup_cin, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(w_oihw && w_ohwi && Cout_pad >= Cout && Cin_pad >= Cin && Cin_total >= Cin, "bad pack_weights args");
  UEGAN_CHECK_ARG(!w_ohwi_lo || dtype == UEGAN_BF16, "hi + lo pairs exist for the 16-bit storage format");
  UEGAN_CHECK_ARG(dup_cin >= 0 && dup_cin <= 2 && (!dup_cin || 2 * Cin <= Cin_pad), "dup_cin: 0, 1 or 2; the repeated channels must fit the padding (2 Cin <= Cin_pad)");
  UEGAN_CHECK_ARG(dup_cin != 2 || dtype == UEGAN_BF16, "hi + lo pairs exist for the 16-bit storage format");
  const int Kp = (int)uegan_packed_k((int64_t)KH * KW * Cin_pad), Kp2 = (int)uegan_packed_k((int64_t)KH * KW * Cout_pad);
  const size_t total = (size_t)Cout_pad * Kp + (w_ihwo ? (size_t)Cin_pad * Kp2 : 0);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UEGAN_F32)
    hipLaunchKernelGGL((pack_weights_kernel<float>), dim3(blocks), dim3(256), 0, s, w_oihw, (float*)w_ohwi, (float*)w_ihwo, Cout, Cin, KH, KW,
                       Cout_pad, Cin_pad, Kp, Kp2, Cin_total, (float*)nullptr, dup_cin);
  else if (dtype == UEGAN_BF16)
    hipLaunchKernelGGL((pack_weights_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, w_oihw, (bf16_t*)w_ohwi, (bf16_t*)w_ihwo, Cout, Cin, KH,
                       KW, Cout_pad, Cin_pad, Kp, Kp2, Cin_total, (bf16_t*)w_ohwi_lo, dup_cin);
  else
    UEGAN_CHECK_ARG(false, "bad dtype");
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_pack_weights_multi(int dtype, const uegan_pack_entry* table_dev, int n_entries, int64_t total, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(table_dev && n_entries > 0 && total > 0, "bad pack_weights_multi args");
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UEGAN_F32) hipLaunchKernelGGL((pack_weights_multi_kernel<float>), dim3(blocks), dim3(256), 0, s, table_dev, n_entries, (long long)total);
  else if (dtype == UEGAN_BF16) hipLaunchKernelGGL((pack_weights_multi_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, table_dev, n_entries, (long long)total);
  else UEGAN_CHECK_ARG(false, "bad dtype");
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

// ----------------------------------------------------------------------------------------------------
// Mirrored images of a reflection-padded data gradient, for the pixels that have any (the adjoint of nn.ReflectionPad2d,
// models.py:80): dx[o] += sum over the image pairs (iy, ix) != (0, 0) of sum_{taps, c} dz[src] * w.  Only pixels in rows
// 1..pad / OH-1-pad..OH-2 or the same columns have images -- 4 lines of a map for pad 1.  The streaming kernel has already
// written the direct image of EVERY pixel; this kernel reads, adds and writes back the affected ones (VALU: a few thousand MACs
// per pixel, <= 1 % of the pixels).  One thread = one affected pixel x one 16-byte chunk of output channels.
// ----------------------------------------------------------------------------------------------------
// per axis: the taps of output coordinate o that reach its direct image (img 0) and its (at most one) mirrored image -- see
// src_coord: t = t0, t0 + stride, ... (n of them), source (q - t) / stride.  A mirrored image only sees the <= pad taps that
// cross the border.
struct AxisTaps {
  int n0, t00, q0;     // direct image
  int n1, t01, q1;     // mirrored image (n1 = 0: none)
};
__device__ __forceinline__ void axis_range(const ConvGeom& g, int pp, int in_n, int K, int& n, int& t0, int& q) {
  q = pp + g.pad;                                   // t2 = q - t >= 0, (q - t) % stride == 0, (q - t) / stride <= in_n - 1
  int t1 = q < K - 1 ? q : K - 1;
  t0 = q - g.stride * (in_n - 1);
  if (t0 < 0) t0 = 0;
  if (g.stride == 2 && ((q - t0) & 1)) ++t0;
  n = t1 >= t0 ? (t1 - t0) / g.stride + 1 : 0;
}
__device__ __forceinline__ AxisTaps axis_taps(const ConvGeom& g, int o, int in_n, int out_n, int K) {
  AxisTaps r;
  axis_range(g, o, in_n, K, r.n0, r.t00, r.q0);
  r.n1 = 0; r.t01 = 0; r.q1 = 0;
  if (o >= 1 && o <= g.pad) axis_range(g, -o, in_n, K, r.n1, r.t01, r.q1);
  else if (o >= out_n - 1 - g.pad && o <= out_n - 2) axis_range(g, 2 * (out_n - 1) - o, in_n, K, r.n1, r.t01, r.q1);
  return r;
}
// j-th (tap, source) of an axis: direct taps first, then the mirrored image's
__device__ __forceinline__ void axis_pick(const ConvGeom& g, const AxisTaps& r, int j, int& t, int& src) {
  if (j < r.n0) { t = r.t00 + j * g.stride; src = (r.q0 - t) / g.stride; }
  else { t = r.t01 + (j - r.n0) * g.stride; src = (r.q1 - t) / g.stride; }
}

// ONE WAVE per affected pixel.  Work units = (tap pair with at least one mirrored axis) x (16-byte chunk of dz channels); lane =
// (chunk of output channels) + NCH * part: the 64 / NCH parts share the units round-robin, partial sums meet through shuffles, the
// part-0 lanes add them into dx.  (A thread-per-pixel loop was a chain of dependent HBM round trips: 170-280 us per launch.)
// rows_only: the x-mirrored images of the direct rows were added inside the streaming kernel (conv_stream.h XMIR); what is left are
// the y-mirrored images (with any x image) of rows 1..pad / OH-1-pad..OH-2 -- whole, contiguous rows.
template <typename T>
__global__ void __launch_bounds__(256) dgrad_images_kernel(ConvArgs a, int n_aff, int nch_log, int rows_only, int multi) {
  constexpr int E = DT<T>::EPC;
  const ConvGeom& g = a.g;
  const int lane = threadIdx.x & 63;
  // multi: a wave takes 64 / nch pixels, every lane the whole unit list of its pixel (no partial sums to shuffle) -- the thin layers' units are
  // so few (3-21 taps x 1-4 dz chunks) that one wave per pixel was bound by wave launches (10^5 waves of a few loads each)
  const size_t wv = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t wid = multi ? wv * (size_t)(64 >> nch_log) + (size_t)(lane >> nch_log) : wv;
  if (wid >= (size_t)g.B * n_aff) return;              // (wave-uniform unless multi; nothing below needs the whole wave then)
  const int q = (int)(wid % n_aff), b = (int)(wid / n_aff);
  const int nyr = 2 * g.pad, nxc = 2 * g.pad;
  int y, x;
  if (q < nyr * g.OW) {                                // whole rows 1..pad and OH-1-pad..OH-2
    const int ri = q / g.OW;
    x = q - ri * g.OW;
    y = ri < g.pad ? 1 + ri : g.OH - 1 - g.pad + (ri - g.pad);
  } else {                                             // the remaining rows: columns 1..pad and OW-1-pad..OW-2
    const int q2 = q - nyr * g.OW;
    const int rr = q2 / nxc, ci = q2 - rr * nxc;
    const int nrest = g.OH - nyr;
    y = rr == 0 ? 0 : (rr == nrest - 1 ? g.OH - 1 : g.pad + rr);
    x = ci < g.pad ? 1 + ci : g.OW - 1 - g.pad + (ci - g.pad);
  }
  const AxisTaps ay = axis_taps(g, y, g.IH, g.OH, g.KH), ax = axis_taps(g, x, g.IW, g.OW, g.KW);
  const int nx = ax.n0 + ax.n1;
  const int items = ay.n1 * nx + (rows_only ? 0 : ay.n0 * ax.n1);        // (mirrored y) x (all x)  +  (direct y) x (mirrored x)
  const int kc = g.C / E;
  const int nch = 1 << nch_log, nparts = multi ? 1 : 64 >> nch_log;
  const int mych = lane & (nch - 1), part = multi ? 0 : lane >> nch_log;
  const int n0 = mych * E;
  const T* dz = static_cast<const T*>(a.in1);
  const T* w = static_cast<const T*>(a.w);
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  const bool nvalid = n0 < a.N;
  // the value to add to (and the deferred-activation mask): loaded up front by the lanes that will write, so that this round trip
  // overlaps the gathers below (a wave lives for a handful of dependent memory round trips: their number is its run time)
  const size_t pixo = ((size_t)b * g.OH + y) * g.OW + x;
  T* p = (a.out2 && n0 >= a.n_out1) ? static_cast<T*>(a.out2) + pixo * (a.N - a.n_out1) + (n0 - a.n_out1)
                                    : static_cast<T*>(a.out) + pixo * (a.out2 ? a.n_out1 : a.N) + n0;
  const bool writer = part == 0 && nvalid;
  typedef typename std::conditional<sizeof(T) == 2, u32x4, f32x4>::type chunk_t;      // one 16-byte chunk, still packed
  chunk_t curp = {}, mkp = {};
  if (writer) {
    curp = *reinterpret_cast<const chunk_t*>(p);
    if (a.mask) mkp = *reinterpret_cast<const chunk_t*>(static_cast<const T*>(a.mask) + pixo * a.N + n0);
  }
  for (int u = part; u < items * kc; u += nparts) {
    const int it = u / kc, c = (u - it * kc) * E;
    int jy, jx;
    if (it < ay.n1 * nx) { jy = ay.n0 + it / nx; jx = it % nx; }
    else { const int i2 = it - ay.n1 * nx; jy = i2 / ax.n1; jx = ax.n0 + i2 % ax.n1; }
    int ty, sy, tx, sx;
    axis_pick(g, ay, jy, ty, sy);
    axis_pick(g, ax, jx, tx, sx);
    const T* zp = dz + (((size_t)b * g.IH + sy) * g.IW + sx) * g.C + c;
    const T* wp = w + (size_t)(nvalid ? n0 : 0) * a.Kp + (size_t)(ty * g.KW + tx) * g.C + c;
    // E + 1 unconditional 16-byte loads in flight together, kept packed until used (registers = resident waves = throughput here)
    chunk_t zq = *reinterpret_cast<const chunk_t*>(zp), wq[E];
#pragma unroll
    for (int e = 0; e < E; ++e) wq[e] = *reinterpret_cast<const chunk_t*>(wp + (size_t)e * a.Kp);
    float zv[E];
    Vec<T, E>::ld(reinterpret_cast<const T*>(&zq), zv);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float wv[E];
      Vec<T, E>::ld(reinterpret_cast<const T*>(&wq[e]), wv);
#pragma unroll
      for (int k = 0; k < E; ++k) acc[e] = fmaf(zv[k], wv[k], acc[e]);
    }
  }
  if (!multi)
    for (int o = 32; o >= nch; o >>= 1) {
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    }
  if (!writer) return;
  const float scale = a.scale ? a.scale[a.scale_group ? b / a.scale_group : 0] : 1.f;
  float cur[E];
  Vec<T, E>::ld(reinterpret_cast<const T*>(&curp), cur);
  if (a.mask) {
    float mv[E];
    Vec<T, E>::ld(reinterpret_cast<const T*>(&mkp), mv);
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] *= act_grad_from_out(mv[e], a.mask_act);
  }
#pragma unroll
  for (int e = 0; e < E; ++e) cur[e] += acc[e] * scale;
  Vec<T, E>::st(p, cur);
}

template <typename T>
static int launch_dgrad_images(ConvArgs& a, hipStream_t s, bool rows_only) {
  const ConvGeom& g = a.g;
  const int n_aff = 2 * g.pad * g.OW + (rows_only ? 0 : (g.OH - 2 * g.pad) * 2 * g.pad);
  const int chunks = a.N / DT<T>::EPC;               // <= 8 for the layers the streaming kernel takes (N <= 64)
  int nch_log = 0;
  while ((1 << nch_log) < chunks) ++nch_log;
  UEGAN_CHECK_ARG(nch_log <= 6 && g.C % DT<T>::EPC == 0, "dgrad_images: unsupported channel counts");
  const int multi = nch_log <= 3 ? 1 : 0;      // (<= 8 output chunks: >= 8 pixels per wave)
  const size_t waves = multi ? ((size_t)g.B * n_aff + (64 >> nch_log) - 1) / (64 >> nch_log) : (size_t)g.B * n_aff;
  hipLaunchKernelGGL((dgrad_images_kernel<T>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a, n_aff, nch_log, rows_only ? 1 : 0, multi);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

// *mask_applied (when asked for): whether the route taken multiplied by act'(a.mask) in its epilogue -- only the streaming kernel
// and the patch kernel's zero-padded 3x3 dgrads do, otherwise the caller runs act_bwd in place
template <typename T>
static int run_gather_gemm(ConvArgs& a, hipStream_t s, bool* mask_applied = nullptr) {
  if (mask_applied) *mask_applied = false;
  if (g_conv_impl == UEGAN_IMPL_DIRECT) {
    const size_t total = (size_t)a.g.B * a.g.OH * a.g.OW * a.N;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL((conv_direct_kernel<T>), dim3(blocks), dim3(256), 0, s, a);
    UEGAN_CHECK_LAUNCH();
    return UEGAN_OK;
  }
  if (g_use_glds && g_use_stream) {                                // <= 4 output channels on 32 input channels: Toeplitz kernel
    const int rc = conv_toep_run(a, DT<T>::kDtype, s);
    if (rc != 1) {
      if (mask_applied) *mask_applied = false;
      return rc;
    }
  }
  ConvStreamPlan sp;
  if (g_use_glds && conv_stream_plan(a, DT<T>::kDtype, sp)) {      // thin full-resolution layers: persistent streaming kernel
    if (mask_applied) *mask_applied = a.mask != nullptr;
    {
      ProfScope prof(prof_key(4, true, sp.tn, sp.pf, a.g.mode, 8, sp.lc == 2),
                     2.0 * (double)sp.a.tiles_total * sp.a.TH * 16 * a.N * (double)(a.g.KH * a.g.KW * a.g.C), s,
                     2.0 * ((double)a.g.B * a.g.OH * a.g.OW * a.N + (double)a.g.B * a.g.IH * a.g.IW * a.g.C));
      conv_stream_launch(sp, s);
      UEGAN_CHECK_LAUNCH();
    }
    if (sp.fixup) return launch_dgrad_images<T>(a, s, sp.a.xmir != 0);
    return UEGAN_OK;
  }
  // the masked epilogue exists for the patch kernel's zero-padded stride-1 3x3 data gradients (the VGG chain) and its 1x1 ones
  // (the generator's upsample / attention convs)
  const bool mask3 = a.g.pad_mode != UEGAN_PAD_REFLECT && a.g.KH == 3 && a.g.KW == 3;
  const bool mask1 = a.g.KH == 1 && a.g.KW == 1 && a.g.pad == 0;
  if (!(g_use_patch && g_use_glds && a.g.mode == 1 && (mask3 || mask1) && a.g.stride == 1 && a.g.C % (CONV_ROWB / (int)sizeof(T)) == 0))
    a.mask = nullptr;
  if (mask_applied) *mask_applied = a.mask != nullptr;
  return dispatch_conv_gemm<T>(a, s);
}

extern "C" int uegan_conv2d_fwd(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias,
                                const float* scale, void* y, uegan_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  UEGAN_CHECK_ARG(x1 && w_ohwi && y && (d->C2 == 0 || x2), "null pointer");
  ConvArgs a;
  a.g = fwd_geom(d);
  a.in1 = x1; a.in2 = d->C2 ? x2 : x1; a.w = w_ohwi; a.bias = bias; a.scale = scale; a.scale_group = d->scale_group; a.out = y; a.out2 = nullptr; a.n_out1 = 0;
  a.N = d->Cout; a.nbias = cout_w(d); a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * a.g.C); a.act = d->act;
  a.frame = 0; a.fy0 = a.fy1 = a.fx0 = a.fx1 = 0; a.mask = nullptr; a.mask_act = UEGAN_ACT_NONE;
  hipStream_t s = (hipStream_t)stream;
  if (g_conv_impl != UEGAN_IMPL_DIRECT && g_use_heads && heads_applicable(d)) {
    ConvStreamPlan sp;
    // (<= 4 output channels: the Toeplitz MFMA kernel where it applies -- 32 k input channels, bf16 -- else the vector-ALU head kernel)
    const bool toep = g_use_glds && g_use_stream && d->act != UEGAN_ACT_SIGMOID && conv_toep_takes(a, d->dtype);
    if (!toep && (d->act == UEGAN_ACT_SIGMOID || !(g_use_glds && conv_stream_plan(a, d->dtype, sp)))) return heads_fwd(d, x1, w_ohwi, bias, scale, y, s);
  }
  // (the MFMA kernels' epilogues evaluate NONE / LRELU / RELU / TANH; the sigmoid exists for the prediction heads, above, and in the direct kernel)
  UEGAN_CHECK_ARG(d->act <= UEGAN_ACT_TANH || (d->act == UEGAN_ACT_SIGMOID && g_conv_impl == UEGAN_IMPL_DIRECT),
                  "activation %d is not available in this convolution's epilogue (sigmoid: prediction heads only; Swish / SELU: uegan_affine_act_fwd)", d->act);
  return d->dtype == UEGAN_F32 ? run_gather_gemm<float>(a, s) : run_gather_gemm<bf16_t>(a, s);
}

// ----------------------------------------------------------------------------------------------------
// Split-K forward: see ConvArgs::kws.  The reduce pass adds the parts in order (deterministic) and applies scale, bias and activation.
// ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* ws, int parts, size_t part_stride, const float* bias, int nbias, const float* scale,
                                                            int scale_group, int act, bf16_t* out, size_t quads, int N, int hw) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  const size_t e = q * 4;
  const int n = (int)(e % N);
  f32x4 v = *reinterpret_cast<const f32x4*>(ws + e);
  for (int p = 1; p < parts; ++p) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(ws + (size_t)p * part_stride + e);
    v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
  }
  const int b = (int)((e / N) / hw);
  const float sc = scale ? scale[scale_group ? b / scale_group : 0] : 1.f;
  float r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = apply_act(v[k] * sc + ((bias && n + k < nbias) ? bias[n + k] : 0.f), act);
  store4(out + e, r[0], r[1], r[2], r[3]);
}
int uegan::splitk_reduce_launch(const ConvArgs& a, hipStream_t s) {
  const size_t elems = (size_t)a.g.B * a.g.OH * a.g.OW * a.N, quads = elems / 4;      // (channel counts are multiples of 4)
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a.kws, a.kparts, elems, a.bias, a.nbias, a.scale,
                     a.scale_group, a.act, static_cast<bf16_t*>(a.out), quads, a.N, a.g.OH * a.g.OW);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

// upper bound of what a split-K forward of this layer would use (0: no kernel would split it -- call uegan_conv2d_fwd)
extern "C" size_t uegan_conv2d_fwd_splitk_workspace_bytes(const uegan_conv_desc* d) {
  if (check_desc(d) || d->dtype == UEGAN_F32 || !g_use_patch || !g_use_glds) return 0;
  const ConvGeom g = fwd_geom(d);
  if (g.KH != g.KW || g.C % 64 || g.C < 128 || d->Cout < 64) return 0;
  int blocks;
  if (g.stride == 1 && g.KH == 3 && g.pad == 1 && d->Cout > 64 && g.OH >= 16) blocks = g.B * ((g.OH + 15) / 16) * ((g.OW + CONV_TW - 1) / CONV_TW) * ((d->Cout + 63) / 64);
  else if (g.stride == 2 && (g.KH == 3 || g.KH == 5 || g.KH == 7) && g.C2 == 0 && g.OH >= 8 && g.OW >= 16) blocks = g.B * ((g.OH + 7) / 8) * ((g.OW + CONV_TW - 1) / CONV_TW) * ((d->Cout + 63) / 64);
  else return 0;
  int parts = 1;
  while (parts * 2 <= g.C / 64 && blocks * parts * 2 <= 256) parts *= 2;
  return parts < 2 ? 0 : (size_t)parts * g.B * g.OH * g.OW * d->Cout * sizeof(float);
}

// uegan_conv2d_fwd with a workspace the kernels may use for a split-K launch (see include/uegan_hip.h)
extern "C" int uegan_conv2d_fwd_splitk(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias, const float* scale,
                                       void* y, void* workspace, size_t workspace_bytes, uegan_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!workspace || !workspace_bytes || d->dtype == UEGAN_F32 || g_conv_impl == UEGAN_IMPL_DIRECT || (g_use_heads && heads_applicable(d)))
    return uegan_conv2d_fwd(d, x1, x2, w_ohwi, bias, scale, y, stream);
  UEGAN_CHECK_ARG(x1 && w_ohwi && y && (d->C2 == 0 || x2), "null pointer");
  UEGAN_CHECK_ARG(d->act <= UEGAN_ACT_TANH, "activation %d is not available in this convolution's epilogue", d->act);
  ConvArgs a;
  a.g = fwd_geom(d);
  a.in1 = x1; a.in2 = d->C2 ? x2 : x1; a.w = w_ohwi; a.bias = bias; a.scale = scale; a.scale_group = d->scale_group; a.out = y; a.out2 = nullptr; a.n_out1 = 0;
  a.N = d->Cout; a.nbias = cout_w(d); a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * a.g.C); a.act = d->act;
  a.frame = 0; a.fy0 = a.fy1 = a.fx0 = a.fx1 = 0; a.mask = nullptr; a.mask_act = UEGAN_ACT_NONE;
  a.kws = static_cast<float*>(workspace); a.kws_bytes = workspace_bytes;
  return run_gather_gemm<bf16_t>(a, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------------------
// Forward + the per-(image, channel) moments of its result (InstanceNorm behind a conv: the generator's attention modules, models.py:227,
// 230-237): where the streaming kernel takes the layer it accumulates sum / sum of squares of its fp32 results on the way out
// (conv_stream_kernel<..., STATS>) and stream_stats_finalize_kernel folds the per-(block, image, wave) partials in a fixed order into
// mean[b][c] and rstd[b][c] = 1 / sqrt(biased variance + eps) (eps < 0: the variance itself) -- the moments pass over the tensor is gone.
// *produced = 0: no such kernel for this layer, y is computed as by uegan_conv2d_fwd and mean / rstd are untouched (the caller runs uegan_moments).
// ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) stream_stats_finalize_kernel(const float* part, float* mean_out, float* rstd_out, int B, int C, int Cs, int HW,
                                                                    int tpi, int tpb, int NW, float eps) {
  // one BLOCK per (b, c): at batch 1 an image is spread over all 512 blocks of the forward (2048 partials per channel); fixed summation order
  __shared__ float red[16];
  const int w = blockIdx.x;
  const int b = w / C, c = w - b * C;
  const int k0 = (b * tpi) / tpb, k1 = ((b + 1) * tpi - 1) / tpb;      // blocks whose tile range touches image b
  const int n = (k1 - k0 + 1) * NW;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
  for (int i = threadIdx.x; i < n; i += 256) {
    const int k = k0 + i / NW, wv = i - (i / NW) * NW;
    const int j = b - (k * tpb) / tpi;                                 // image index inside block k's range (0 or 1)
    const float* o = part + ((size_t)((k * 2 + j) * NW + wv) * Cs + c) * 2;
    s1 += o[0]; s2 += o[1];
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    const float m = s1 / (float)HW;
    float var = s2 / (float)HW - m * m;
    var = var > 0.f ? var : 0.f;
    mean_out[w] = m;
    rstd_out[w] = eps < 0.f ? var : 1.f / sqrtf(var + eps);
  }
}

static bool fwd_stats_plan(const uegan_conv_desc* d, ConvArgs& a, ConvStreamPlan& sp) {
  if (d->dtype != UEGAN_BF16 || g_conv_impl == UEGAN_IMPL_DIRECT || !g_use_glds || g_tuning[UEGAN_TUNE_FWD_STATS] == 0) return false;
  if (d->act > UEGAN_ACT_TANH) return false;
  // (64 output channels on 16-row tiles sit at the 256-register limit without the 32 sum registers: 8-row tiles there -- VGG conv1_1 609 -> 462 us; 420 without the sums)
  if (!conv_stream_plan(a, d->dtype, sp, a.N > 32 ? 2 : 4) || !conv_stream_stats_ok(sp)) return false;
  return true;
}
static void fwd_args(const uegan_conv_desc* d, ConvArgs& a, const void* x1, const void* x2, const void* w_ohwi, const float* bias, const float* scale, void* y) {
  a.g = fwd_geom(d);
  a.in1 = x1; a.in2 = d->C2 ? x2 : x1; a.w = w_ohwi; a.bias = bias; a.scale = scale; a.scale_group = d->scale_group; a.out = y; a.out2 = nullptr; a.n_out1 = 0;
  a.N = d->Cout; a.nbias = cout_w(d); a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * a.g.C); a.act = d->act;
  a.frame = 0; a.fy0 = a.fy1 = a.fx0 = a.fx1 = 0; a.mask = nullptr; a.mask_act = UEGAN_ACT_NONE;
}
extern "C" size_t uegan_conv2d_fwd_stats_workspace_bytes(const uegan_conv_desc* d) {
  if (check_desc(d)) return 0;
  ConvArgs a;
  ConvStreamPlan sp;
  fwd_args(d, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (!fwd_stats_plan(d, a, sp)) return 0;
  return (size_t)sp.blocks * 2 * sp.nw * (sp.tn * 16) * 2 * sizeof(float);
}
extern "C" int uegan_conv2d_fwd_stats(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias, const float* scale,
                                      void* y, float* mean, float* rstd, float eps, void* workspace, size_t workspace_bytes, int* produced,
                                      uegan_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  UEGAN_CHECK_ARG(x1 && w_ohwi && y && produced && (d->C2 == 0 || x2), "null pointer");
  *produced = 0;
  ConvArgs a;
  ConvStreamPlan sp;
  fwd_args(d, a, x1, x2, w_ohwi, bias, scale, y);
  if (!fwd_stats_plan(d, a, sp)) return uegan_conv2d_fwd(d, x1, x2, w_ohwi, bias, scale, y, stream);
  UEGAN_CHECK_ARG(mean && rstd && workspace && workspace_bytes >= uegan_conv2d_fwd_stats_workspace_bytes(d), "conv2d_fwd_stats: mean / rstd / workspace");
  hipStream_t s = (hipStream_t)stream;
  const int tpi = (sp.a.ty1 - sp.a.ty0) * (sp.a.tx1 - sp.a.tx0);
  sp.a.c.stats_part = static_cast<float*>(workspace);
  sp.a.c.stats_tpi = tpi;
  sp.stats = true;
  {
    ProfScope prof(prof_key(4, true, sp.tn, sp.pf, a.g.mode, 8, sp.lc == 2), 2.0 * (double)sp.a.tiles_total * sp.a.TH * 16 * a.N * (double)(a.g.KH * a.g.KW * a.g.C), s,
                   2.0 * ((double)a.g.B * a.g.OH * a.g.OW * a.N + (double)a.g.B * a.g.IH * a.g.IW * a.g.C));
    conv_stream_launch(sp, s);
    UEGAN_CHECK_LAUNCH();
  }
  const int C = d->Cout;      // (padding channels: exact zeros in y, mean 0 and rstd 1 / sqrt(eps): what uegan_moments reports for them)
  hipLaunchKernelGGL(stream_stats_finalize_kernel, dim3(d->B * C), dim3(256), 0, s, (const float*)workspace, mean, rstd, d->B, C, sp.tn * 16,
                     d->Ho * d->Wo, tpi, sp.a.tiles_per_block, sp.nw, eps);
  UEGAN_CHECK_LAUNCH();
  *produced = 1;
  return UEGAN_OK;
}

// ----------------------------------------------------------------------------------------------------
// Forward with extras (round 6): hi + lo pairs, the product epilogue, the generator's final residual + clamp, the moments (include/uegan_hip.h).
// ----------------------------------------------------------------------------------------------------
static bool ex_plan(const uegan_conv_desc* d, const uegan_conv_ex* ex, ConvArgs& a, ConvStreamPlan& sp, bool* toep) {
  *toep = false;
  if (d->dtype != UEGAN_BF16 || g_conv_impl == UEGAN_IMPL_DIRECT || !g_use_glds || !g_use_stream || d->act > UEGAN_ACT_TANH) return false;
  a.in1_lo = ex->x1_lo; a.in2_lo = d->C2 ? ex->x2_lo : nullptr; a.w_lo = ex->w_lo; a.out_lo = ex->y_lo;
  a.mul = ex->mul; a.mul_lo = ex->mul_lo; a.out_mul = ex->y_mul; a.out_mul_lo = ex->y_mul_lo;
  a.res_x = ex->res_x; a.res_x2 = ex->res_x2; a.res_out = ex->res_out; a.res_out2 = ex->res_out2; a.res_split = ex->res_split > 0 ? ex->res_split : (1 << 30);
  if (ex->w_interleaved) {               // a stride-2 forward against a [hi | lo] weight matrix: the source's channels read twice (ConvArgs::src_wrap)
    if (d->stride != 2 || d->C2 || (d->C1 != 32 && d->C1 != 64) || a.in1_lo || a.w_lo || a.out_lo || a.mul || a.res_out) return false;
    a.g.C = 2 * d->C1;
    a.src_wrap = d->C1 - 1;
    a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * a.g.C);
    *toep = false;
    return true;
  }
  if (a.res_out || cout_w(d) <= 4) {      // dec5.1: the Toeplitz kernel
    *toep = conv_toep_takes(a, d->dtype) && !(a.res_out && a.res_split < d->B && !(a.res_x2 && a.res_out2));
    return *toep;
  }
  return conv_stream_plan(a, d->dtype, sp);
}
extern "C" size_t uegan_conv2d_fwd_ex_workspace_bytes(const uegan_conv_desc* d, const uegan_conv_ex* ex) {
  if (check_desc(d) || !ex) return 0;
  ConvArgs a;
  ConvStreamPlan sp;
  bool toep;
  fwd_args(d, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (!ex_plan(d, ex, a, sp, &toep) || toep || a.src_wrap || g_tuning[UEGAN_TUNE_FWD_STATS] == 0 || !conv_stream_stats_ok(sp)) return 0;
  sp.stats = true;
  if (!conv_stream_ex_available(sp)) return 0;
  return (size_t)sp.blocks * 2 * sp.nw * (sp.tn * 16) * 2 * sizeof(float);
}
extern "C" int uegan_conv2d_fwd_ex(const uegan_conv_desc* d, const uegan_conv_ex* ex, const void* x1, const void* x2, const void* w_ohwi, const float* bias,
                                   const float* scale, void* y, int* taken, uegan_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  UEGAN_CHECK_ARG(ex && taken && x1 && w_ohwi && y && (d->C2 == 0 || x2), "null pointer");
  *taken = 0;
  ConvArgs a;
  ConvStreamPlan sp;
  bool toep;
  fwd_args(d, a, x1, x2, w_ohwi, bias, scale, y);
  if (!ex_plan(d, ex, a, sp, &toep)) return UEGAN_OK;
  hipStream_t s = (hipStream_t)stream;
  if (toep) {
    rc = conv_toep_run(a, d->dtype, s);
    if (rc == 1) return UEGAN_OK;
    if (rc == UEGAN_OK) *taken = 1;
    return rc;
  }
  if (a.src_wrap) {
    rc = conv_s2fwd_run(a, d->dtype, s);
    if (rc == 1) return UEGAN_OK;
    if (rc == UEGAN_OK) *taken = 1;
    return rc;
  }
  const size_t wsb = ex->mean ? uegan_conv2d_fwd_ex_workspace_bytes(d, ex) : 0;
  const bool stats = wsb != 0 && ex->rstd && ex->stats_workspace && ex->stats_workspace_bytes >= wsb;
  const int tpi = (sp.a.ty1 - sp.a.ty0) * (sp.a.tx1 - sp.a.tx0);
  if (stats) {
    sp.a.c.stats_part = static_cast<float*>(ex->stats_workspace);
    sp.a.c.stats_tpi = tpi;
    sp.stats = true;
  }
  {
    const int npl = 1 + (a.in1_lo || a.in2_lo ? 1 : 0) + (a.w_lo ? 1 : 0);      // MFMA passes per operand pair
    ProfScope prof(prof_key(4, true, sp.tn, sp.pf, 8 + sp.pr, 8, sp.lc >= 2), 2.0 * npl * (double)sp.a.tiles_total * sp.a.TH * 16 * a.N * (double)(a.g.KH * a.g.KW * a.g.C), s,
                   2.0 * ((double)a.g.B * a.g.OH * a.g.OW * a.N * (1 + (a.out_lo ? 1 : 0) + (a.mul ? 2 : 0) + (a.out_mul_lo ? 1 : 0) + (a.mul_lo ? 1 : 0)) +
                          (double)a.g.B * a.g.IH * a.g.IW * (a.g.C + sp.a.c_lo)));
    if (!conv_stream_launch_ex(sp, s)) return UEGAN_OK;
    UEGAN_CHECK_LAUNCH();
  }
  *taken = 1;
  if (stats) {
    const int C = d->Cout;
    hipLaunchKernelGGL(stream_stats_finalize_kernel, dim3(d->B * C), dim3(256), 0, s, (const float*)ex->stats_workspace, ex->mean, ex->rstd, d->B, C, sp.tn * 16,
                       d->Ho * d->Wo, tpi, sp.a.tiles_per_block, sp.nw, ex->eps);
    UEGAN_CHECK_LAUNCH();
    *taken |= 2;
  }
  return UEGAN_OK;
}

// forward + 2x2 max-pool of the result (the VGG chain, losses.py:74-104: conv, ReLU, MaxPool2d(2)): y as uegan_conv2d_fwd, y_pool =
// maxpool2x2(y) written by the convolution's epilogue where the kernel that takes the layer can, else by the pooling kernel
extern "C" int uegan_conv2d_fwd_pool(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias,
                                     const float* scale, void* y, void* y_pool, uegan_stream_t stream) {
  return uegan_conv2d_fwd_pool_part(d, x1, x2, w_ohwi, bias, scale, y, y_pool, d ? d->B : 0, stream);
}

// ... where only the first n_full images need y itself: y[n_full:] is UNDEFINED afterwards (a kernel with a pooling epilogue does not store
// those rows -- the store-bound half of its epilogue; the fallback writes them).  The fidelity loss's reference images (losses.py:29-30: no
// gradient reaches them) and every no-grad VGG pass: the outputs of conv1_2 / conv2_2 / conv3_4 / conv4_4 feed nothing but their pool.
extern "C" int uegan_conv2d_fwd_pool_part(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias,
                                          const float* scale, void* y, void* y_pool, int n_full, uegan_stream_t stream) {
  return uegan_conv2d_fwd_pool_idx(d, x1, x2, w_ohwi, bias, scale, y, y_pool, nullptr, n_full, 0, stream);
}

// ... and with the window position of each maximum (one byte per pooled element, first maximum in (row, column) order as ATen's
// max_pool2d_with_indices picks it) for the first n_idx images: uegan_maxpool2x2_bwd_idx routes the gradient with them and the pooled
// tensor alone, so an image that needs a gradient does not need its full-resolution y either (n_full = 0, n_idx = B: the fidelity loss's
// enhanced batch -- y is then neither written by the forward nor read by the backward).  idx[n_idx:] is undefined afterwards.
extern "C" int uegan_conv2d_fwd_pool_idx(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias,
                                         const float* scale, void* y, void* y_pool, void* idx, int n_full, int n_idx, uegan_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  UEGAN_CHECK_ARG(n_full >= 0 && n_full <= d->B && n_idx >= 0 && n_idx <= d->B && (idx || n_idx == 0), "conv2d_fwd_pool: n_full / n_idx outside [0, B], or positions wanted without a buffer");
  UEGAN_CHECK_ARG(x1 && w_ohwi && y && y_pool && (d->C2 == 0 || x2), "null pointer");
  UEGAN_CHECK_ARG(d->Ho % 2 == 0 && d->Wo % 2 == 0, "conv2d_fwd_pool needs an even output map");
  UEGAN_CHECK_ARG(d->act <= UEGAN_ACT_TANH, "activation %d is not available in this convolution's epilogue", d->act);
  ConvArgs a;
  a.g = fwd_geom(d);
  a.in1 = x1; a.in2 = d->C2 ? x2 : x1; a.w = w_ohwi; a.bias = bias; a.scale = scale; a.scale_group = d->scale_group; a.out = y; a.out2 = nullptr; a.n_out1 = 0;
  a.N = d->Cout; a.nbias = cout_w(d); a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * a.g.C); a.act = d->act;
  a.frame = 0; a.fy0 = a.fy1 = a.fx0 = a.fx1 = 0; a.mask = nullptr; a.mask_act = UEGAN_ACT_NONE;
  a.pool_out = y_pool;
  a.n_full = n_full;
  a.pool_idx = n_idx > 0 ? idx : nullptr;
  a.n_idx = n_idx;
  hipStream_t s = (hipStream_t)stream;
  rc = d->dtype == UEGAN_F32 ? run_gather_gemm<float>(a, s) : run_gather_gemm<bf16_t>(a, s);
  if (rc || a.pool_done) return rc;
  // (no kernel with a pooling epilogue took the layer: y is complete -- the plain kernels ignore n_full -- and is pooled here)
  if (a.pool_idx) return uegan_maxpool2x2_fwd_idx(d->dtype, y, y_pool, idx, d->B, d->Ho, d->Wo, d->Cout, stream);
  return uegan_maxpool2x2_fwd(d->dtype, y, y_pool, d->B, d->Ho, d->Wo, d->Cout, stream);
}

extern "C" int uegan_conv2d_dgrad(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const float* scale, void* dx1,
                                  void* dx2, uegan_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  UEGAN_CHECK_ARG(dz && w_ihwo && dx1 && (d->C2 == 0 || dx2), "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (g_conv_impl != UEGAN_IMPL_DIRECT && g_use_heads && heads_dgrad_applicable(d)) return heads_dgrad(d, dz, w_ihwo, scale, dx1, s);
  ConvArgs a;
  ConvGeom& g = a.g;
  g.B = d->B; g.IH = d->Ho; g.IW = d->Wo; g.C1 = d->Cout; g.C2 = 0; g.C = d->Cout;
  g.OH = d->H; g.OW = d->W; g.KH = d->KH; g.KW = d->KW; g.stride = d->stride; g.pad = d->pad; g.pad_mode = d->pad_mode;
  g.mode = 1;
  a.in1 = dz; a.in2 = dz; a.bias = nullptr; a.nbias = 0; a.scale = scale; a.scale_group = d->scale_group; a.act = UEGAN_ACT_NONE;
  a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * d->Cout);
  a.w = w_ihwo; a.N = d->C1 + d->C2;
  a.out = dx1; a.out2 = d->C2 ? dx2 : nullptr; a.n_out1 = d->C1;      // virtual concat: one launch, two destinations
  a.frame = 0; a.fy0 = a.fy1 = a.fx0 = a.fx1 = 0; a.mask = nullptr; a.mask_act = UEGAN_ACT_NONE;
  return d->dtype == UEGAN_F32 ? run_gather_gemm<float>(a, s) : run_gather_gemm<bf16_t>(a, s);
}

// ----------------------------------------------------------------------------------------------------
// Reflection-padded dgrad on small maps: "pad-grid dgrad + fold".  On a map of a few tiles every tile touches the border,
// so the mirrored-image passes of the MODE 2 kernels (a full MFMA pass per left / right image) cost 2-3x the direct
// work.  Instead the dgrad runs image-free over the PADDED grid (a zero-pad, pad = 0 problem of (H+2p) x (W+2p) pixels)
// into a workspace and fold_reflect_kernel adds each pixel's up to 3 x 3 mirror sources (the adjoint of
// nn.ReflectionPad2d, models.py:80) while copying the interior out.  The workspace is ~(1 + 2p/H)^2 x the size of dx,
// which is why large maps keep the interior / frame split.
// ----------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) fold_reflect_kernel(const T* __restrict__ ws, T* __restrict__ dx1, T* __restrict__ dx2,
                                                           int B, int H, int W, int p, int Ct, int C1) {
  constexpr int E = 16 / (int)sizeof(T);
  const int cch = Ct / E;
  const size_t total = (size_t)B * H * W * cch;
  const int Hp = H + 2 * p, Wp = W + 2 * p;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % cch);
    size_t r = i / cch;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    int ys[3], xs[3], ny = 1, nx = 1;
    ys[0] = y + p; xs[0] = x + p;
    if (y >= 1 && y <= p) ys[ny++] = p - y;                                  // mirrored across row 0
    if (y <= H - 2 && y >= H - 1 - p) ys[ny++] = p + 2 * (H - 1) - y;        // mirrored across row H-1
    if (x >= 1 && x <= p) xs[nx++] = p - x;
    if (x <= W - 2 && x >= W - 1 - p) xs[nx++] = p + 2 * (W - 1) - x;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    for (int iy = 0; iy < ny; ++iy)
      for (int ix = 0; ix < nx; ++ix) {
        float v[E];
        Vec<T, E>::ld(ws + (((size_t)b * Hp + ys[iy]) * Wp + xs[ix]) * Ct + cc * E, v);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += v[e];
      }
    const int c = cc * E;
    const size_t pix = ((size_t)b * H + y) * W + x;
    T* o = (dx2 && c >= C1) ? dx2 + pix * (Ct - C1) + (c - C1) : dx1 + pix * (dx2 ? C1 : Ct) + c;
    Vec<T, E>::st(o, acc);
  }
}

// Which reflection-padded dgrads take the pad-grid + fold route.  Measured on the model's layers (bf16, batch 16; direct -> fold):
// 5x5s2 256->512 @32^2 0.313 -> 0.141 ms, 7x7s2 64->128 @128^2 0.270 -> 0.168, the 5x5 / 7x7 prediction heads @<=128^2
// 1.3-2.1x faster; every 3x3 (pad 1: one mirrored row, cheap images) equal or slower, maps >= 256^2 slower (workspace
// traffic), and 5x5s2 128->256 @64^2 slower (0.105 -> 0.133: its 32^2 parity-class grids are exactly 2 x 2 tiles, the
// padded 34^2 ones 3 x 3).  uegan_set_tuning(UEGAN_TUNE_FOLD_MAX, n) (full-resolution pixels; the tests flip it) overrides the map
// limit, n = 0 disables the route, -1 (default) is this rule.
static bool dgrad_folds(const uegan_conv_desc* d) {
  if (d->pad_mode != UEGAN_PAD_REFLECT || d->pad == 0 || g_conv_impl == UEGAN_IMPL_DIRECT) return false;
  if (g_use_heads && heads_dgrad_applicable(d)) return false;      // (the one-channel heads: uegan_conv2d_dgrad's VALU kernel, no workspace)
  if (g_use_glds && conv_flat_applicable(d)) return true;          // stride-2 layers: conv_flat_kernel computes the padded grid in one launch
  if (g_tuning[UEGAN_TUNE_FOLD_MAX] >= 0) return (long)d->H * d->W <= (long)g_tuning[UEGAN_TUNE_FOLD_MAX];
  if (d->pad < 2 || (long)d->H * d->W > 128L * 128L) return false;
  if (d->Cout <= 8) return true;      // prediction heads (gather-GEMM dgrad, no tile quantisation): always faster folded
  auto tiles = [&](int h, int w) { return (((h + d->stride - 1) / d->stride + 15) / 16) * (((w + d->stride - 1) / d->stride + 15) / 16); };
  const int direct = tiles(d->H, d->W), padded = tiles(d->H + 2 * d->pad, d->W + 2 * d->pad);
  return direct == 1 || padded <= 2 * direct;
}

extern "C" size_t uegan_conv2d_dgrad_workspace_bytes(const uegan_conv_desc* d) {
  if (check_desc(d) || !dgrad_folds(d)) return 0;
  const size_t es = d->dtype == UEGAN_F32 ? 4 : 2;
  return (size_t)d->B * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * (d->C1 + d->C2) * es;
}

extern "C" int uegan_conv2d_dgrad_ws(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const float* scale, void* dx1,
                                     void* dx2, void* workspace, size_t workspace_bytes, uegan_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!dgrad_folds(d)) return uegan_conv2d_dgrad(d, dz, w_ihwo, scale, dx1, dx2, stream);
  UEGAN_CHECK_ARG(dz && w_ihwo && dx1 && (d->C2 == 0 || dx2), "null pointer");
  UEGAN_CHECK_ARG(workspace && workspace_bytes >= uegan_conv2d_dgrad_workspace_bytes(d), "dgrad workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ConvArgs a;
  ConvGeom& g = a.g;
  g.B = d->B; g.IH = d->Ho; g.IW = d->Wo; g.C1 = d->Cout; g.C2 = 0; g.C = d->Cout;
  g.OH = d->H + 2 * d->pad; g.OW = d->W + 2 * d->pad;       // the padded grid: dz -> d(pad(x)) is a pad-0 transposed conv
  g.KH = d->KH; g.KW = d->KW; g.stride = d->stride; g.pad = 0; g.pad_mode = UEGAN_PAD_ZERO;
  g.mode = 1;
  a.in1 = dz; a.in2 = dz; a.bias = nullptr; a.nbias = 0; a.scale = scale; a.scale_group = d->scale_group; a.act = UEGAN_ACT_NONE;
  a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * d->Cout);
  a.w = w_ihwo; a.N = d->C1 + d->C2;
  a.out = workspace; a.out2 = nullptr; a.n_out1 = 0;
  a.frame = 0; a.fy0 = a.fy1 = a.fx0 = a.fx1 = 0; a.mask = nullptr; a.mask_act = UEGAN_ACT_NONE;
  rc = g_use_glds ? conv_flat_run(d, dz, w_ihwo, scale, workspace, s) : 1;
  if (rc == 1) rc = d->dtype == UEGAN_F32 ? run_gather_gemm<float>(a, s) : run_gather_gemm<bf16_t>(a, s);
  if (rc) return rc;
  const int Ct = d->C1 + d->C2;
  const size_t es = d->dtype == UEGAN_F32 ? 4 : 2;
  const size_t total = (size_t)d->B * d->H * d->W * (Ct * es / 16);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (d->dtype == UEGAN_F32)
    hipLaunchKernelGGL((fold_reflect_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)workspace, (float*)dx1,
                       d->C2 ? (float*)dx2 : nullptr, d->B, d->H, d->W, d->pad, Ct, d->C1);
  else
    hipLaunchKernelGGL((fold_reflect_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)workspace, (bf16_t*)dx1,
                       d->C2 ? (bf16_t*)dx2 : nullptr, d->B, d->H, d->W, d->pad, Ct, d->C1);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

// The data gradient with respect to the PADDED input, for a caller whose next kernel can add the mirror images itself (uegan_sn_act_bwd_p,
// uegan_act_bwd_p): workspace = [B][H + 2 pad][W + 2 pad][C1], *pad_out = pad.  Only where a kernel computes the padded grid in one launch --
// stride-2 layers on conv_flat_kernel (w_ihwo), the one-channel prediction heads on head_dgrad_mfma_kernel (w_ohwi: the FORWARD pack) --
// else *pad_out = -1, nothing is launched and the caller takes uegan_conv2d_dgrad_ws.
extern "C" int uegan_conv2d_dgrad_padded(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const void* w_ohwi, const float* scale,
                                         void* workspace, size_t workspace_bytes, int* pad_out, uegan_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  UEGAN_CHECK_ARG(dz && pad_out, "null pointer");
  *pad_out = -1;
  if (d->C2 || d->pad_mode != UEGAN_PAD_REFLECT || d->pad == 0 || g_conv_impl == UEGAN_IMPL_DIRECT || !g_use_glds) return UEGAN_OK;
  const size_t need = (size_t)d->B * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * d->C1 * (d->dtype == UEGAN_F32 ? 4 : 2);
  hipStream_t s = (hipStream_t)stream;
  if (g_use_heads && w_ohwi && !scale && heads_dgrad_mfma_applicable(d)) {
    UEGAN_CHECK_ARG(workspace && workspace_bytes >= need, "dgrad_padded: workspace too small");
    rc = heads_dgrad_mfma(d, dz, w_ohwi, workspace, s);
    if (rc == UEGAN_OK) *pad_out = d->pad;
    return rc;
  }
  if (w_ihwo && conv_flat_applicable(d)) {
    UEGAN_CHECK_ARG(workspace && workspace_bytes >= need, "dgrad_padded: workspace too small");
    rc = conv_flat_run(d, dz, w_ihwo, scale, workspace, s);
    if (rc == 1) return UEGAN_OK;
    if (rc == UEGAN_OK) *pad_out = d->pad;
    return rc;
  }
  return UEGAN_OK;
}
extern "C" size_t uegan_conv2d_dgrad_padded_bytes(const uegan_conv_desc* d) {
  if (check_desc(d)) return 0;
  // (0 where uegan_conv2d_dgrad_padded would decline the layer whatever packs it is handed: the caller then does not allocate the padded grid at all)
  if (d->C2 || d->pad_mode != UEGAN_PAD_REFLECT || d->pad == 0 || g_conv_impl == UEGAN_IMPL_DIRECT || !g_use_glds) return 0;
  if (!(g_use_heads && heads_dgrad_mfma_applicable(d)) && !conv_flat_applicable(d)) return 0;
  return (size_t)d->B * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * (d->C1 + d->C2) * (d->dtype == UEGAN_F32 ? 4 : 2);
}

// dx = dgrad(dz) * act'(x_act): the data gradient with the activation gradient of the layer that PRODUCED the conv input folded
// into the epilogue (act' is a function of the activated output, which is this conv's saved input).  The producer then skips its
// own act_bwd pass -- valid when every consumer of that tensor applies the factor (uegan_amd/losses.py VGG19_relu).
extern "C" int uegan_conv2d_dgrad_act(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const float* scale, void* dx1,
                                      void* workspace, size_t workspace_bytes, int in_act, const void* x_act, uegan_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  UEGAN_CHECK_ARG(d->C2 == 0, "dgrad_act takes one destination");
  if (in_act == UEGAN_ACT_NONE) return uegan_conv2d_dgrad_ws(d, dz, w_ihwo, scale, dx1, nullptr, workspace, workspace_bytes, stream);
  UEGAN_CHECK_ARG(dz && w_ihwo && dx1 && x_act, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  bool applied = false;
  if (dgrad_folds(d)) {
    rc = uegan_conv2d_dgrad_ws(d, dz, w_ihwo, scale, dx1, nullptr, workspace, workspace_bytes, stream);
  } else {
    ConvArgs a;
    ConvGeom& g = a.g;
    g.B = d->B; g.IH = d->Ho; g.IW = d->Wo; g.C1 = d->Cout; g.C2 = 0; g.C = d->Cout;
    g.OH = d->H; g.OW = d->W; g.KH = d->KH; g.KW = d->KW; g.stride = d->stride; g.pad = d->pad; g.pad_mode = d->pad_mode;
    g.mode = 1;
    a.in1 = dz; a.in2 = dz; a.bias = nullptr; a.nbias = 0; a.scale = scale; a.scale_group = d->scale_group; a.act = UEGAN_ACT_NONE;
    a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * d->Cout);
    a.w = w_ihwo; a.N = d->C1;
    a.out = dx1; a.out2 = nullptr; a.n_out1 = d->C1;
    a.frame = 0; a.fy0 = a.fy1 = a.fx0 = a.fx1 = 0; a.mask = nullptr; a.mask_act = UEGAN_ACT_NONE;
    a.mask = x_act; a.mask_act = in_act;
    rc = d->dtype == UEGAN_F32 ? run_gather_gemm<float>(a, s, &applied) : run_gather_gemm<bf16_t>(a, s, &applied);
  }
  if (rc || applied) return rc;
  return uegan_act_bwd(d->dtype, in_act, dx1, x_act, dx1, (int64_t)d->B * d->H * d->W * d->C1, stream);
}

static void wgrad_plan(const uegan_conv_desc* d, WgradArgs& a, int& nsplit, dim3& grid, int& bn, WgradTrPlan& tr) {
  a.g = fwd_geom(d);
  a.N = cout_w(d);
  a.zC = d->Cout;
  a.ktot = d->KH * d->KW * (d->C1 + d->C2);
  if (g_conv_impl != UEGAN_IMPL_DIRECT && wgtr_plan(d, a.g, tr)) {      // bf16 transpose-read kernel
    bn = -1;
    nsplit = tr.nsplit_eff;
    grid = tr.grid;
    return;
  }
  const int npix = d->dtype == UEGAN_BF16 ? 64 : 32;     // pixel slots per K step (128-byte LDS rows)
  int ws = 1, wl = 0;
  while (ws < d->Wo && ws < npix) { ws <<= 1; ++wl; }
  a.WS = ws; a.WSlog = wl; a.R = npix / ws;
  a.nxb = (d->Wo + ws - 1) / ws;
  a.nyb = (d->Ho + a.R - 1) / a.R;
  a.steps_total = d->B * a.nyb * a.nxb;
  bn = a.N <= 16 ? 16 : (a.N <= 32 ? 32 : (a.N <= 64 ? 64 : 128));
  if (g_conv_impl != UEGAN_IMPL_DIRECT && g_use_heads && heads_applicable(d)) {     // VALU head kernel: one partial per block
    bn = 0;
    nsplit = heads_wgrad_blocks(d);
    grid = dim3(nsplit);
    return;
  }
  const int tiles = ((a.ktot + WG_BK - 1) / WG_BK) * ((a.N + bn - 1) / bn);
  int want = (1536 + tiles - 1) / tiles;
  if (want < 1) want = 1;
  if (want > a.steps_total) want = a.steps_total;
  a.steps_per_split = (a.steps_total + want - 1) / want;
  nsplit = (a.steps_total + a.steps_per_split - 1) / a.steps_per_split;
  grid = dim3((a.ktot + WG_BK - 1) / WG_BK, (a.N + bn - 1) / bn, nsplit);
}

extern "C" size_t uegan_conv2d_wgrad_workspace_bytes(const uegan_conv_desc* d) {
  if (check_desc(d)) return 0;
  WgradArgs a;
  WgradTrPlan tr;
  int nsplit, bn;
  dim3 grid;
  wgrad_plan(d, a, nsplit, grid, bn, tr);
  return ((size_t)nsplit * ((size_t)a.N * a.ktot + a.N) + (size_t)BIAS_BLOCKS * d->Cout) * sizeof(float);
}

template <typename T>
static int run_wgrad(const uegan_conv_desc* d, WgradArgs& a, WgradTrPlan& tr, int nsplit, dim3 grid, int bn, const float* scale, float* dw,
                     float* dbias, int accmask, hipStream_t s) {
  const int acc = accmask & 1, accb = (accmask >> 1) & 1;      // accumulate into dw / into dbias
  if (g_conv_impl == UEGAN_IMPL_DIRECT) {
    const size_t total = (size_t)a.N * a.ktot;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((wgrad_direct_kernel<T>), dim3(blocks), dim3(256), 0, s, a, dw, scale, cin_w(d), acc, cin_row(d));
  } else if (bn == -1) {
    tr.a.in1 = a.in1; tr.a.in2 = a.in2; tr.a.dz = a.dz; tr.a.ws = a.ws;
    tr.a.want_bias = dbias ? 1 : 0;
    {
      ProfScope prof(prof_key(3, true, tr.tn, tr.tm, 0, 8, tr.big), 2.0 * (double)d->B * d->Ho * d->Wo * a.N * (double)a.ktot, s,
                     2.0 * ((double)d->B * d->H * d->W * (d->C1 + d->C2) + (double)d->B * d->Ho * d->Wo * d->Cout));
      wgtr_launch(tr, s);
      UEGAN_CHECK_LAUNCH();
    }
    const size_t total = (size_t)a.N * a.ktot + (dbias ? a.N : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, a.ws, dw, dbias, scale, nsplit, a.N, a.g.C,
                       cin_w(d), a.g.KH, a.g.KW, (size_t)tr.a.pstride, acc, accb, cin_row(d));
    UEGAN_CHECK_LAUNCH();
    return UEGAN_OK;
  } else if (bn == 0) {
    int rc = heads_wgrad(d, a.in1, a.dz, a.ws, s);
    if (rc) return rc;
    const size_t total = (size_t)a.N * a.ktot;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, a.ws, dw, (float*)nullptr, scale, nsplit, a.N,
                       a.g.C, cin_w(d), a.g.KH, a.g.KW, total, acc, accb, cin_row(d));
  } else {
    {
      ProfScope prof(prof_key(2, DT<T>::kDtype == UEGAN_BF16, bn, 0, 0, 8, false), 2.0 * (double)d->B * d->Ho * d->Wo * a.N * (double)a.ktot, s,
                     sizeof(T) * ((double)d->B * d->H * d->W * (d->C1 + d->C2) + (double)d->B * d->Ho * d->Wo * d->Cout));
      if (bn == 128) hipLaunchKernelGGL((conv_wgrad_kernel<T, 128>), grid, dim3(256), 0, s, a);
      else if (bn == 64) hipLaunchKernelGGL((conv_wgrad_kernel<T, 64>), grid, dim3(256), 0, s, a);
      else if (bn == 32) hipLaunchKernelGGL((conv_wgrad_kernel<T, 32>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((conv_wgrad_kernel<T, 16>), grid, dim3(256), 0, s, a);
      UEGAN_CHECK_LAUNCH();
    }
    const size_t total = (size_t)a.N * a.ktot;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, a.ws, dw, (float*)nullptr, scale, nsplit, a.N,
                       a.g.C, cin_w(d), a.g.KH, a.g.KW, total, acc, accb, cin_row(d));
  }
  UEGAN_CHECK_LAUNCH();
  if (dbias) {
    const size_t npix = (size_t)d->B * d->Ho * d->Wo;
    const int nch = a.zC / DT<T>::EPC;
    int cp = 1;
    while (cp < nch && cp < 64) cp <<= 1;
    const size_t rows = 256 / cp;
    size_t blocks = (npix + rows * 8 - 1) / (rows * 8);
    if (blocks > BIAS_BLOCKS) blocks = BIAS_BLOCKS;
    if (blocks < 1) blocks = 1;
    float* part = a.ws + (size_t)nsplit * a.N * a.ktot;      // tail of the wgrad workspace
    hipLaunchKernelGGL((bias_grad_partial_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const T*>(a.dz), part, npix, a.zC);
    UEGAN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3(a.N), dim3(256), 0, s, part, dbias, (int)blocks, a.N, a.zC, accb);
    UEGAN_CHECK_LAUNCH();
  }
  return UEGAN_OK;
}

extern "C" int uegan_conv2d_wgrad(const uegan_conv_desc* d, const void* x1, const void* x2, const void* dz, const float* scale,
                                  float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes, uegan_stream_t stream) {
  return uegan_conv2d_wgrad_acc(d, x1, x2, dz, scale, dw_oihw, dbias, workspace, workspace_bytes, 0, stream);
}

extern "C" int uegan_conv2d_wgrad_acc(const uegan_conv_desc* d, const void* x1, const void* x2, const void* dz, const float* scale,
                                      float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes, int accumulate,
                                      uegan_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  UEGAN_CHECK_ARG(x1 && dz && dw_oihw && (d->C2 == 0 || x2), "null pointer");
  WgradArgs a;
  WgradTrPlan tr;
  int nsplit, bn;
  dim3 grid;
  wgrad_plan(d, a, nsplit, grid, bn, tr);
  const size_t need = ((size_t)nsplit * ((size_t)a.N * a.ktot + a.N) + (size_t)BIAS_BLOCKS * d->Cout) * sizeof(float);
  UEGAN_CHECK_ARG(workspace && workspace_bytes >= need, "wgrad workspace too small: %zu < %zu", workspace_bytes, need);
  a.in1 = x1; a.in2 = d->C2 ? x2 : x1; a.dz = dz; a.ws = static_cast<float*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  UEGAN_CHECK_ARG(accumulate >= 0 && accumulate <= 3, "accumulate is a bit mask: 1 = dw, 2 = dbias");
  return d->dtype == UEGAN_F32 ? run_wgrad<float>(d, a, tr, nsplit, grid, bn, scale, dw_oihw, dbias, accumulate, s)
                               : run_wgrad<bf16_t>(d, a, tr, nsplit, grid, bn, scale, dw_oihw, dbias, accumulate, s);
}

extern "C" int uegan_act_bwd(int dtype, int act, const void* g, const void* a, void* dz, int64_t n, uegan_stream_t stream) {
  return uegan_act_bwd3(dtype, act, g, nullptr, nullptr, a, dz, n, stream);
}
extern "C" int uegan_act_bwd2(int dtype, int act, const void* g, const void* g2, const void* a, void* dz, int64_t n, uegan_stream_t stream) {
  return uegan_act_bwd3(dtype, act, g, g2, nullptr, a, dz, n, stream);
}

extern "C" int uegan_act_bwd3(int dtype, int act, const void* g, const void* g2, const void* g3, const void* a, void* dz, int64_t n,
                              uegan_stream_t stream) {
  UEGAN_CHECK_ARG(g && a && dz && n >= 0, "bad act_bwd args");
  if (n == 0) return UEGAN_OK;
  hipStream_t s = (hipStream_t)stream;
  const int epc = dtype == UEGAN_F32 ? 4 : 8;
  const bool vec = n % epc == 0;
  const size_t work = vec ? (size_t)n / epc : (size_t)n;
  const int blocks = (int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192);
  if (dtype == UEGAN_F32) {
    if (vec) hipLaunchKernelGGL((act_bwd_kernel<float, 4>), dim3(blocks), dim3(256), 0, s, (const float*)g, (const float*)g2, (const float*)g3, (const float*)a, (float*)dz, (size_t)n, act);
    else hipLaunchKernelGGL((act_bwd_kernel<float, 1>), dim3(blocks), dim3(256), 0, s, (const float*)g, (const float*)g2, (const float*)g3, (const float*)a, (float*)dz, (size_t)n, act);
  } else {
    if (vec) hipLaunchKernelGGL((act_bwd_kernel<bf16_t, 8>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)g2, (const bf16_t*)g3, (const bf16_t*)a, (bf16_t*)dz, (size_t)n, act);
    else hipLaunchKernelGGL((act_bwd_kernel<bf16_t, 1>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)g2, (const bf16_t*)g3, (const bf16_t*)a, (bf16_t*)dz, (size_t)n, act);
  }
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" size_t uegan_sn_act_bwd_workspace_floats(int ngroups, int C) { return (size_t)ngroups * SNB * (1 + (size_t)C); }

extern "C" int uegan_sn_act_bwd(int dtype, int act, const void* g, const void* g2, const void* y, const float* bias, int nbias,
                                const float* inv_sigma, void* dz, float* workspace, int64_t pix_per_group, int C, int ngroups,
                                uegan_stream_t stream) {
  return uegan_sn_act_bwd_p(dtype, act, g, 0, g2, 0, y, bias, nbias, inv_sigma, dz, workspace, pix_per_group, 0, 0, C, ngroups, stream);
}

// ... with g / g2 optionally on the PADDED grid of their reflection-padded consumer ([images][H + 2 pad][W + 2 pad][C], pad_g / pad_g2 > 0: what
// uegan_conv2d_dgrad_padded returns): the mirror images of the padding are added while the gradient is read
extern "C" int uegan_sn_act_bwd_p(int dtype, int act, const void* g, int pad_g, const void* g2, int pad_g2, const void* y, const float* bias, int nbias,
                                  const float* inv_sigma, void* dz, float* workspace, int64_t pix_per_group, int H, int W, int C, int ngroups,
                                  uegan_stream_t stream) {
  UEGAN_CHECK_ARG(g && y && inv_sigma && dz && workspace && pix_per_group > 0 && ngroups >= 1 && ngroups <= 8, "bad sn_act_bwd args");
  UEGAN_CHECK_ARG(act == UEGAN_ACT_NONE || act == UEGAN_ACT_LRELU || act == UEGAN_ACT_RELU, "sn_act_bwd: none / LeakyReLU / ReLU");
  UEGAN_CHECK_ARG(pad_g >= 0 && pad_g2 >= 0 && (g2 || pad_g2 == 0), "sn_act_bwd: bad padding");
  if (pad_g || pad_g2)
    UEGAN_CHECK_ARG(H > 0 && W > 0 && pix_per_group % ((int64_t)H * W) == 0 && pad_g < H && pad_g < W && pad_g2 < H && pad_g2 < W,
                    "sn_act_bwd: a padded-grid gradient needs the map size (H, W) and whole images per group");
  UEGAN_CHECK_ARG(!(pad_g || pad_g2) || pix_per_group < (1ll << 31), "sn_act_bwd: more than 2^31 pixels per group");
  const int epc = dtype == UEGAN_F32 ? 4 : 8;
  UEGAN_CHECK_ARG(C % epc == 0 && 256 % (C / epc) == 0, "sn_act_bwd: channel chunks per pixel must divide 256 (C = %d)", C);
  const int pl = 256 / (C / epc);
  long long bx = (pix_per_group + pl * 4 - 1) / (pl * 4);
  if (bx > SNB / 2) bx = SNB / 2;
  if (bx < 1) bx = 1;
  // padded-grid gradients: the pixels on the mirror ring go to blocks of their own (see the kernel); both kinds share the SNB partial slots
  SnRing ring = {0, 0, 0, 0};
  int bx_main = (int)bx;
  if (pad_g || pad_g2) {
    ring.pmax = pad_g > pad_g2 ? pad_g : pad_g2;
    ring.ring_all = (H < 2 * ring.pmax + 3 || W < 2 * ring.pmax + 3) ? 1 : 0;
    ring.n_row = 2 * ring.pmax * W;
    ring.n_ring = ring.ring_all ? H * W : ring.n_row + 2 * ring.pmax * (H - 2 * ring.pmax);
    const long long ring_pix = (pix_per_group / ((long long)H * W)) * ring.n_ring;
    long long brg = (ring_pix + pl - 1) / pl;
    if (brg > SNB / 2) brg = SNB / 2;
    if (brg < 1) brg = 1;
    if (ring.ring_all) bx_main = 0;
    bx = bx_main + brg;
  }
  // (the partial arrays are laid out for SNB blocks per group whatever the launch uses: the finish kernel is told the actual count)
  float* cpart = workspace;
  float* dbpart = workspace + (size_t)ngroups * SNB;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UEGAN_F32)
    hipLaunchKernelGGL((sn_act_bwd_kernel<float>), dim3((unsigned)bx, ngroups), dim3(256), 0, s, (const float*)g, (const float*)g2, (const float*)y, bias, nbias,
                       inv_sigma, (float*)dz, cpart, dbpart, (long long)pix_per_group, C, act, pad_g, pad_g2, H, W, bx_main, ring);
  else if (dtype == UEGAN_BF16)
    hipLaunchKernelGGL((sn_act_bwd_kernel<bf16_t>), dim3((unsigned)bx, ngroups), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)g2, (const bf16_t*)y, bias,
                       nbias, inv_sigma, (bf16_t*)dz, cpart, dbpart, (long long)pix_per_group, C, act, pad_g, pad_g2, H, W, bx_main, ring);
  else UEGAN_CHECK_ARG(false, "bad dtype %d", dtype);
  UEGAN_CHECK_LAUNCH();
  return (int)bx;                                    // > 0: the number of partial blocks per group (for uegan_sn_grad_finish)
}

// dz = (g + g2) * act'(a) with g / g2 optionally on padded grids (as above); a, dz: [B][H][W][C]
extern "C" int uegan_act_bwd_p(int dtype, int act, const void* g, int pad_g, const void* g2, int pad_g2, const void* a, void* dz, int B, int H, int W,
                               int C, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(g && a && dz && B > 0 && H > 0 && W > 0 && C > 0, "bad act_bwd_p args");
  UEGAN_CHECK_ARG(pad_g >= 0 && pad_g2 >= 0 && (g2 || pad_g2 == 0) && pad_g < H && pad_g < W && pad_g2 < H && pad_g2 < W, "act_bwd_p: bad padding");
  const int epc = dtype == UEGAN_F32 ? 4 : 8;
  UEGAN_CHECK_ARG(C % epc == 0, "act_bwd_p: whole 16-byte chunks per pixel (C = %d)", C);
  const size_t work = (size_t)B * H * W * (C / epc);
  UEGAN_CHECK_ARG(work < (1ull << 32) - 8192ull * 256, "act_bwd_p: more than 2^32 chunks");
  const int blocks = (int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UEGAN_F32)
    hipLaunchKernelGGL((act_bwd_p_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)g, pad_g, (const float*)g2, pad_g2, (const float*)a, (float*)dz, B, H, W, C, act);
  else if (dtype == UEGAN_BF16)
    hipLaunchKernelGGL((act_bwd_p_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)g, pad_g, (const bf16_t*)g2, pad_g2, (const bf16_t*)a, (bf16_t*)dz, B, H, W, C, act);
  else UEGAN_CHECK_ARG(false, "bad dtype %d", dtype);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_sn_grad_finish(float* dw, float* db, const float* workspace, int nbx, int ngroups, const float* u_hist, const float* v_hist,
                                    int rows, int cols, int C, int acc_bias, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(dw && workspace && u_hist && v_hist && nbx >= 1 && nbx <= SNB && ngroups >= 1 && ngroups <= 8 && rows > 0 && cols > 0 && rows <= C,
                  "bad sn_grad_finish args");
  const size_t n = (size_t)rows * cols;
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 512) blocks = 512;
  const int nbb = (rows + 15) / 16;                  // blocks that also finish 16 bias channels each
  if (blocks < nbb) blocks = nbb;
  hipLaunchKernelGGL(sn_grad_finish_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dw, db, workspace, workspace + (size_t)ngroups * SNB, nbx,
                     ngroups, u_hist, v_hist, rows, cols, C, acc_bias ? 1 : 0, nbb);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_selftest_mfma(void* scratch, uegan_stream_t stream) {
  UEGAN_CHECK_ARG(scratch, "null scratch");
  hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (float*)scratch);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

extern "C" int uegan_profile_begin(int max_records) {
  UEGAN_CHECK_ARG(max_records > 0, "max_records must be positive");
  while ((int)g_prof_pool.size() < max_records) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      set_error("hipEventCreate failed");
      return UEGAN_E_HIP;
    }
    g_prof_pool.push_back(std::make_pair(a, b));
  }
  g_prof_records.clear();
  g_prof_used = 0;
  g_prof_on = true;
  return UEGAN_OK;
}

extern "C" int uegan_profile_end(uegan_profile_entry* out, int max_entries, int* n_entries) {
  UEGAN_CHECK_ARG(out && n_entries && max_entries > 0, "bad profile_end args");
  g_prof_on = false;
  std::vector<int> keys;
  std::vector<double> ms, fl, by;
  std::vector<long long> cnt;
  for (const ProfRecord& r : g_prof_records) {
    if (hipEventSynchronize(r.stop) != hipSuccess) { set_error("hipEventSynchronize failed"); return UEGAN_E_HIP; }
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) { set_error("hipEventElapsedTime failed"); return UEGAN_E_HIP; }
    size_t i = 0;
    while (i < keys.size() && keys[i] != r.kernel_id) ++i;
    if (i == keys.size()) { keys.push_back(r.kernel_id); ms.push_back(0); fl.push_back(0); by.push_back(0); cnt.push_back(0); }
    ms[i] += t; fl[i] += r.flops; by[i] += r.bytes; cnt[i] += 1;
  }
  int n = 0;
  for (size_t i = 0; i < keys.size() && n < max_entries; ++i, ++n) {
    prof_kernel_name(keys[i], out[n].name, sizeof(out[n].name));
    out[n].launches = cnt[i];
    out[n].total_ms = ms[i];
    out[n].total_flops = fl[i];
    out[n].total_bytes = by[i];
  }
  *n_entries = n;
  g_prof_records.clear();
  g_prof_used = 0;
  return UEGAN_OK;
}
