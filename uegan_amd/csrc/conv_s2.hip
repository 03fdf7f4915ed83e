// Stride-2 forward convolutions with >= 64 input channels (G.enc3-5 3x3, D.d3 7x7, D.d4/d5 5x5; models.py:15-19, 113-133) on the
// patch-resident MFMA structure of conv_patch.h.
//
// A stride-2 K x K convolution reads every input pixel of its window, but an output pixel's taps hop two input pixels at a time:
// staging the input window as ONE patch would make neighbouring taps' fragments overlap awkwardly (and conv_gemm_kernel, which
// these layers ran on, re-gathers the pixel tile from L2 for every tap: 320-480 TFLOP/s).  Instead the input is split into its four
// PARITY CLASSES (cy, cx) = (padded row parity, padded column parity): tap (ty, tx) only touches class (ty & 1, tx & 1), and
// restricted to one class the convolution is a plain STRIDE-1 convolution of the class sub-grid with ceil((K - c) / 2) taps per
// axis:
//     out(i, j) = sum_{cy, cx} sum_{tq, tp} W[cy + 2 tq][cx + 2 tp] . in_pad(2 (i + tq) + cy, 2 (j + tp) + cx)
// So the K loop runs over phases (class, 64-channel chunk): per phase a (TH + KSH - 1) x (16 + KSH - 1) patch of the class
// sub-grid is staged once (direct-to-LDS loads, reflection / zero padding resolved in the per-lane source address, the
// 16-byte-chunk XOR swizzle of conv_gemm_kernel applied on that address) and the class's taps walk over it as LDS row offsets;
// per tap only a BN x 128 B weight slice moves, through a 3-deep ring staged two steps ahead.  Accumulators live across all
// phases.  bf16 and fp32 storage; 8 waves, tile = 16 x 16 output pixels x 128 channels (each wave 64 pixels x 64 channels).
#include <stdlib.h>

#include "conv_core.h"

namespace uegan {

// KSH = ceil(K / 2): taps per axis the class patch is sized for
// HALF (bf16, 32 input channels: D.d2 7x7, G.enc2 3x3): a 128-byte LDS row holds the 32 channels of TWO horizontally adjacent input
// pixels, i.e. of the column classes cx = 0 and 1 at once -- phases are the two ROW classes, a K step covers the tap pair
// (ty, 2 tp) | (ty, 2 tp + 1) (the second half is zero when 2 tp + 1 is past the kernel), whose weights are 64 contiguous elements
// of the [Cout][K*K*C] pack.
// ONEP: one patch buffer (a phase's patch is loaded in place at the phase switch, its latency covered by the CU's other block): 70 instead of
// 116 KB of LDS for the 32-input-channel variant, i.e. two blocks per CU
// SPLITK: blockIdx.z = part of the 64-channel chunks (ConvArgs::kws), fp32 partial sums instead of the epilogue
template <typename T, int BN, int WARPS_M, int WARPS_N, int KSH, int TH, bool HALF = false, bool ONEP = false, int NWBUF = 3, bool SPLITK = false>
__global__ void __launch_bounds__(64 * WARPS_M * WARPS_N) conv_s2fwd_kernel(ConvArgs a) {
  constexpr int ROWB = CONV_ROWB, TW = CONV_TW, BM = TH * TW, NWAVES = WARPS_M * WARPS_N;
  constexpr int EPC = DT<T>::EPC;
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int PH = TH + KSH - 1, PW = TW + KSH - 1;
  constexpr int NPG = (PH * PW + 7) / 8;
  constexpr int NI_P = (NPG + NWAVES - 1) / NWAVES;
  constexpr int WROWG = BN / 8;
  constexpr int NI_W = (WROWG + NWAVES - 1) / NWAVES;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int NCHUNK = Mma<T>::NCHUNK;
  constexpr int NSUB = BK / 32;
  constexpr int PBUFB = NPG * 8 * ROWB, WSLICE = BN * ROWB;
  static_assert(TM >= 1 && TN >= 1 && WROWG % NWAVES == 0, "tile");

  __shared__ __attribute__((aligned(16))) unsigned char lds[(ONEP ? 1 : 2) * PBUFB + NWBUF * WSLICE];
  unsigned char* const lds_w = lds + (ONEP ? 1 : 2) * PBUFB;

  const ConvGeom& g = a.g;
  const T* in1 = static_cast<const T*>(a.in1);
  const T* w = static_cast<const T*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int n0 = blockIdx.y * BN;
  const bool refl = g.pad_mode == UEGAN_PAD_REFLECT;
  int t = blockIdx.x;
  const int tile_x = t % a.ntx; t /= a.ntx;
  const int tile_y = t % a.nty;
  const int b = t / a.nty;
  const int y0 = tile_y * TH, x0 = tile_x * TW;
  const int chunk0 = SPLITK ? (int)blockIdx.z * a.kchunks : 0;
  const int nchunk = HALF ? 1 : (SPLITK ? ((g.C / BK - chunk0) < a.kchunks ? (g.C / BK - chunk0) : a.kchunks) : g.C / BK);      // (launched only when C is a whole number of chunks)
  const int nph = HALF ? 2 : 4 * nchunk;              // phases: class-major, chunk-minor

  // staging role (identical LDS row / position scheme to conv_gemm_kernel and conv_patch_kernel)
  const int srow = lane >> 3, spos = lane & 7;
  const int sdc = spos ^ swz128(lane >> 3);
  const int c_in_chunk = sdc * EPC;

  const int half = sdc >> 2;                          // HALF: which pixel of the pair (column class) my 16-byte chunk belongs to
  auto cls_taps = [&](int cls, int& nty, int& ntx) {
    if (HALF) { nty = (g.KH - cls + 1) >> 1; ntx = (g.KW + 1) >> 1; return; }      // (cls = row class; tap PAIRS along x)
    nty = (g.KH - (cls >> 1) + 1) >> 1;
    ntx = (g.KW - (cls & 1) + 1) >> 1;
  };
  auto stage_patch = [&](unsigned char* buf, int ph) {
    const int cls = ph / nchunk, chunk = ph - cls * nchunk;
    const int cy = HALF ? cls : cls >> 1, cx = HALF ? half : cls & 1;
    const int cc0 = HALF ? (sdc & 3) * EPC : (chunk0 + chunk) * BK + c_in_chunk;
    const int cc = a.src_wrap ? (cc0 & a.src_wrap) : cc0;      // (ConvArgs::src_wrap: the source's channels against a [hi | lo] weight pair)
#pragma unroll
    for (int ii = 0; ii < NI_P; ++ii) {
      const int rg = ii * NWAVES + wave;
      if (rg < NPG) {
        const int pr = rg * 8 + srow;
        const void* src = g_zero16;
        if (pr < PH * PW) {
          const int piy = pr / PW, pix = pr - piy * PW;
          int sy = 2 * (y0 + piy) + cy - g.pad, sx = 2 * (x0 + pix) + cx - g.pad;      // class sub-grid (r, c) = padded pixel (2 r + cy, 2 c + cx)
          if (refl) { sy = reflect_idx(sy, g.IH); sx = reflect_idx(sx, g.IW); }
          // (tiles may overhang the map and the last sub-grid row / column may lie beyond the padded image: those gather zero)
          if (sy >= 0 && sy < g.IH && sx >= 0 && sx < g.IW) src = in1 + ((size_t)(b * g.IH + sy) * g.IW + sx) * g.C1 + cc;
        }
        glds16(src, buf + rg * 8 * ROWB);
      }
    }
  };
  const T* wbase[NI_W];
#pragma unroll
  for (int i = 0; i < NI_W; ++i) {
    const int n = n0 + (i * NWAVES + wave) * 8 + srow;
    wbase[i] = n < a.N ? w + (size_t)n * a.Kp + c_in_chunk : nullptr;
  }
  auto stage_w = [&](unsigned char* buf, int ph, int tq, int tp) {
    const int cls = ph / nchunk, chunk = ph - cls * nchunk;
    const int off = HALF ? (cls + 2 * tq) * g.KW * g.C + 2 * tp * g.C      // (wbase carries the pair half: c_in_chunk = half * 32 + channel)
                         : ((cls >> 1) + 2 * tq) * g.KW * g.C + ((cls & 1) + 2 * tp) * g.C + (chunk0 + chunk) * BK;
    const bool live = !HALF || 2 * tp + half < g.KW;
#pragma unroll
    for (int i = 0; i < NI_W; ++i) {
      const void* src = (wbase[i] && live) ? (const void*)(wbase[i] + off) : (const void*)g_zero16;
      glds16(src, buf + (i * NWAVES + wave) * 8 * ROWB);
    }
  };
  auto advance = [&](int& ph, int& tq, int& tp) {
    int nty, ntx;
    cls_taps(ph / nchunk, nty, ntx);
    if (++tp == ntx) { tp = 0; if (++tq == nty) { tq = 0; ++ph; } }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  int wad[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int row = wn * WTN + i * 16 + fr;
    wad[i] = row * ROWB + ((fg ^ swz128(row)) << 4);
  }

  int c_ph = 0, c_tq = 0, c_tp = 0;                  // compute cursor
  int w_ph = 0, w_tq = 0, w_tp = 0;                  // next weight slice to stage (two steps ahead)
  // prologue: patch of phase 0, weight slices of steps 0 and 1
  stage_patch(lds, 0);
  stage_w(lds_w, w_ph, w_tq, w_tp);
  advance(w_ph, w_tq, w_tp);
  if (NWBUF == 3 && w_ph < nph) {
    stage_w(lds_w + WSLICE, w_ph, w_tq, w_tp);
    advance(w_ph, w_tq, w_tp);
  }
  int slot = 0, pbuf = 0;
  bool phase_start = true;
  while (c_ph < nph) {
    // slice of this step (and anything older, incl. this phase's patch) must have landed; the most recent slice may stay in flight
    int nph_next = c_ph, ntq = c_tq, ntp = c_tp;
    advance(nph_next, ntq, ntp);
    if (NWBUF == 2 || nph_next >= nph) wait_vmcnt<0>(); else wait_vmcnt<NI_W>();      // (ring of 2: the next slice is requested after the barrier)
    raw_barrier();
    // issue order matters for the vmcnt accounting: first the NEXT phase's patch (on the first step of the current phase: its buffer was
    // last read one phase ago), then the weight slice two steps ahead (its ring slot was read at the previous step)
    if (!ONEP && phase_start && c_ph + 1 < nph) stage_patch(lds + (pbuf ^ 1) * PBUFB, c_ph + 1);
    if (w_ph < nph) {
      const int wslot = slot == 0 ? NWBUF - 1 : slot - 1;
      stage_w(lds_w + wslot * WSLICE, w_ph, w_tq, w_tp);
      advance(w_ph, w_tq, w_tp);
    }
    if (ONEP && phase_start && c_ph > 0) {      // every wave is past the previous phase's last tap (the barrier above)
      stage_patch(lds, c_ph);
      wait_vmcnt<0>();
      raw_barrier();
    }
    const unsigned char* pcur = lds + (ONEP ? 0 : pbuf) * PBUFB;
    const unsigned char* wcur = lds_w + slot * WSLICE;
    int xad[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int pr = (wm * (TH / WARPS_M) + j + c_tq) * PW + fr + c_tp;      // stride-1 walk over the class patch
      xad[j] = pr * ROWB + ((fg ^ swz128(pr)) << 4);
    }
#pragma unroll
    for (int ksub = 0; ksub < NSUB; ++ksub) {
      u32x4 xf[TM][NCHUNK], wf[TN][NCHUNK];
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) xf[j][c] = *reinterpret_cast<const u32x4*>(pcur + (xad[j] ^ ((ksub + c * (NCHUNK - 1)) << 6)));
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) wf[i][c] = *reinterpret_cast<const u32x4*>(wcur + (wad[i] ^ ((ksub + c * (NCHUNK - 1)) << 6)));
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mma<T>::step(wf[i], xf[j], acc[i][j]);
    }
    {
      const int before = c_ph;
      advance(c_ph, c_tq, c_tp);
      phase_start = c_ph != before;
      if (phase_start) pbuf ^= 1;
    }
    slot = slot + 1 == NWBUF ? 0 : slot + 1;
  }

  if constexpr (SPLITK) {      // fp32 partial sums of this part (splitk_reduce_kernel adds the parts and applies the epilogue)
    float* ws = a.kws + (size_t)blockIdx.z * ((size_t)g.B * g.OH * g.OW * a.N);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + wn * WTN + i * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int oy = y0 + wm * (TH / WARPS_M) + j, ox = x0 + fr;
        if (oy < g.OH && ox < g.OW && n < a.N) *reinterpret_cast<f32x4*>(ws + (((size_t)b * g.OH + oy) * g.OW + ox) * a.N + n) = acc[i][j];
      }
    }
    return;
  }
  // ---- epilogue: lane holds channels n..n+3 of pixel (tile row, column fr)
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
      const int oy = y0 + wm * (TH / WARPS_M) + j, ox = x0 + fr;
      if (oy >= g.OH || ox >= g.OW || n >= a.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(acc[i][j][r] * scale + bv[r], a.act);
      store4(out + (((size_t)b * g.OH + oy) * g.OW + ox) * a.N + n, v[0], v[1], v[2], v[3]);
    }
  }
}

template <typename T, int KSH>
static int launch_s2(ConvArgs& a, hipStream_t s) {
  const ConvGeom& g = a.g;
  constexpr int TH = 16;
  a.nty = (g.OH + TH - 1) / TH;
  a.ntx = (g.OW + CONV_TW - 1) / CONV_TW;
  const int gm = g.B * a.nty * a.ntx;
  if (gm == 0) return UEGAN_OK;
  // (uegan_set_tuning: the tests lower it to reach both variants on emulator-sized maps)
  const int small_grid = g_tuning[UEGAN_TUNE_SMALL_GRID];
  if (gm * ((a.N + 127) / 128) < small_grid / 2) {
    // small maps (the deep layers of a single-image inference): 8 x 16 tiles x 64 channels, 4 waves -- 4x the blocks.  (Half the patch
    // kernel's threshold: D.d5 at 192 blocks is still faster on the large tiles.)
    a.nty = (g.OH + 7) / 8;
    const int gs = g.B * a.nty * a.ntx;
    ProfScope prof(prof_key(5, DT<T>::kDtype == UEGAN_BF16, 64, 2 * KSH - 1, 0, 8, true),
                   2.0 * (double)g.B * g.OH * g.OW * a.N * (double)(g.KH * g.KW * g.C), s,
                   sizeof(T) * ((double)g.B * g.OH * g.OW * a.N + (double)g.B * g.IH * g.IW * g.C));
    if constexpr (sizeof(T) == 2) {
      // ... and, where even that grid leaves CUs empty (enc4 / enc5 of one 512^2 image: 128 / 64 blocks of 18 / 36 K steps), the chunks of the K loop over
      // several blocks per tile (ConvArgs::kws: the caller's workspace)
      int kchunks;
      const int parts = splitk_parts(a, gs * ((a.N + 63) / 64), g.C / (CONV_ROWB / (int)sizeof(T)), &kchunks);
      if (parts > 1) {
        a.kparts = parts; a.kchunks = kchunks;
        hipLaunchKernelGGL((conv_s2fwd_kernel<T, 64, 2, 2, KSH, 8, false, false, 3, true>), dim3(gs, (a.N + 63) / 64, parts), dim3(256), 0, s, a);
        UEGAN_CHECK_LAUNCH();
        return splitk_reduce_launch(a, s);
      }
    }
    hipLaunchKernelGGL((conv_s2fwd_kernel<T, 64, 2, 2, KSH, 8>), dim3(gs, (a.N + 63) / 64), dim3(256), 0, s, a);
    UEGAN_CHECK_LAUNCH();
    return UEGAN_OK;
  }
  ProfScope prof(prof_key(5, DT<T>::kDtype == UEGAN_BF16, 128, 2 * KSH - 1, 0, TH, true),
                 2.0 * (double)g.B * g.OH * g.OW * a.N * (double)(g.KH * g.KW * g.C), s,
                 sizeof(T) * ((double)g.B * g.OH * g.OW * a.N + (double)g.B * g.IH * g.IW * g.C));
  // one patch buffer + a 2-deep weight ring (69-78 instead of 122-140 KB: two blocks per CU): enc3 / enc4 / enc5 / d3 forwards 0.20 / 0.15 / 0.13 / 0.14 ->
  // 0.15 / 0.12 / 0.10 / 0.11 ms at batch 32, the 5x5 layers unchanged
  if constexpr (KSH == 2 && DT<T>::kDtype == UEGAN_BF16) {
    // <= 64 output channels on 64 (virtual) input channels: G.enc2 against its [hi | lo] weight pair (ConvArgs::src_wrap) -- a 64-channel block instead of a
    // 128-channel one whose upper half multiplies zeros
    if (a.N <= 64) {
      hipLaunchKernelGGL((conv_s2fwd_kernel<T, 64, 4, 2, KSH, TH, false, true, 2>), dim3(gm, 1), dim3(512), 0, s, a);
      UEGAN_CHECK_LAUNCH();
      return UEGAN_OK;
    }
  }
  hipLaunchKernelGGL((conv_s2fwd_kernel<T, 128, 4, 2, KSH, TH, false, true, 2>), dim3(gm, (a.N + 127) / 128), dim3(512), 0, s, a);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

// 32 input channels (bf16): pixel-pair rows, 64-channel blocks (D.d2, G.enc2: 64 output channels)
template <int KSH>
static int launch_s2_half(ConvArgs& a, hipStream_t s) {
  const ConvGeom& g = a.g;
  constexpr int TH = 16;
  a.nty = (g.OH + TH - 1) / TH;
  a.ntx = (g.OW + CONV_TW - 1) / CONV_TW;
  const int gm = g.B * a.nty * a.ntx;
  if (gm == 0) return UEGAN_OK;
  ProfScope prof(prof_key(5, true, 64, 2 * KSH - 1, 0, TH, true), 2.0 * (double)g.B * g.OH * g.OW * a.N * (double)(g.KH * g.KW * g.C), s,
                 2.0 * ((double)g.B * g.OH * g.OW * a.N + (double)g.B * g.IH * g.IW * g.C));
  hipLaunchKernelGGL((conv_s2fwd_kernel<bf16_t, 64, 4, 2, KSH, TH, true, true>), dim3(gm, (a.N + 63) / 64), dim3(512), 0, s, a);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

// 1: not a layer this kernel takes (the caller falls back to conv_gemm_kernel)
int conv_s2fwd_run(ConvArgs& a, int dtype, hipStream_t s) {
  const ConvGeom& g = a.g;
  const int bk = dtype == UEGAN_BF16 ? 64 : 32;
  if (g.mode != 0 || g.stride != 2 || g.KH != g.KW || g.C2 != 0 || a.N < 64 || a.out2 || a.mask) return 1;
  if (g.KH != 3 && g.KH != 5 && g.KH != 7) return 1;
  if (g.OH < 8 || g.OW < 16) return 1;               // (tiny maps: the 16 x 16 tile would be mostly padding)
  if (dtype == UEGAN_BF16 && g.C == 32) {
    if (g.KH == 3) return launch_s2_half<2>(a, s);
    if (g.KH == 5) return launch_s2_half<3>(a, s);
    return launch_s2_half<4>(a, s);
  }
  if (g.C % bk) return 1;
  if (dtype == UEGAN_BF16) {
    if (g.KH == 3) return launch_s2<bf16_t, 2>(a, s);
    if (g.KH == 5) return launch_s2<bf16_t, 3>(a, s);
    return launch_s2<bf16_t, 4>(a, s);
  }
  if (g.KH == 3) return launch_s2<float, 2>(a, s);
  if (g.KH == 5) return launch_s2<float, 3>(a, s);
  return launch_s2<float, 4>(a, s);
}

}  // namespace uegan
