// Data gradient of the discriminator's one-channel prediction heads (models.py:170-182: ReflectionPad2d(p) -> Conv2d(C, 1, K)) on the MFMA,
// over the PADDED grid (round 5).
//
// With ONE output channel the gradient with respect to the padded input is a correlation of a SCALAR map with the K x K x C weights:
//     F[b][uy][ux][c] = sum_{ty, tx} w[c][ty][tx] * dzp[b][uy - ty][ux - tx]          (dzp zero outside the map; uy, ux on the (H + 2p) x (W + 2p) grid)
// i.e. a GEMM  D[c][pixel] = sum_k A[c][k] B[k][pixel]  with k = (ty, tx), A = the weights and B = a Toeplitz gather of the scalar map.  Eight
// consecutive k of one lane are the eight tx of one tap row: for the pixel at column ux that is dzp[uy - ty][ux .. ux - 7] -- eight CONSECUTIVE
// scalars of the row stored right-to-left.  The tile's scalar patch is kept in the LDS reversed and in eight copies shifted by 0 .. 7 elements,
// so every lane's eight values are ONE aligned 16-byte read (7 KB for an 8 x 32 tile with K = 7), and one 32x32x16 MFMA covers two tap rows
// (K = 7: 4 MFMAs per 32 channels x 32 pixels, tx = 7 and ty = 7 multiplied by zero weights).  The weights' fragments come from row 0 of the OHWI
// pack ([tap][c]: 32 lanes = 32 consecutive channels per load) once per block and stay in registers over the block's tiles.
// The vector-ALU kernel this replaces (head_dgrad_kernel, heads.hip: thread = pixel x 8 channels, 49 FMAs per output element with the weights
// re-read from the LDS) ran 111 / 45 us per launch (7 x 7 / 5 x 5) -- 6 % of the pipe's rate; this one is bound by its stores.
// The mirror images of the reflection padding are NOT added here: the result is the padded grid [B][H + 2p][W + 2p][C], and the consumer -- the
// trunk activation's backward (uegan_sn_act_bwd_p / uegan_act_bwd_p), which reads this gradient exactly once anyway -- folds them in while it
// reads (up to 2 x 2 sources on the border ring, one elsewhere): no fold pass, no second copy of the gradient.
#include "conv_core.h"

namespace uegan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32_head(u32x4 a, u32x4 b, f32x16 c) {
#ifdef UEGAN_HALF_FP16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}

struct HeadDgArgs {
  const void* dz;       // [B][H][W][Zc]: channel 0 = the gradient of the head's pre-activation output
  const void* w;        // OHWI [Zc][Kp], row 0 = [tap][c]
  void* out;            // [B][H + 2p][W + 2p][C]
  int B, H, W, C, Zc, Kp, p;
  int Hp, Wp, nty, ntx, ntiles;
};

template <int KS, int NI>
__global__ void __launch_bounds__(256, 2) head_dgrad_mfma_kernel(HeadDgArgs a) {
  constexpr int TH = 8, TW = 32, PH = TH + KS - 1, PWC = TW + 7;      // patch: rows, columns (source columns x0 - 7 .. x0 + 31)
  constexpr int NM = (KS + 1) / 2;                                     // MFMAs (pairs of tap rows) per fragment pair
  constexpr int BN = NI * 32, EROW = BN * 2 + 8, WPIX = 64;
  constexpr int ZROWB = 8 * 64;                                        // bytes per patch row: 8 shifted copies x 32 elements
  constexpr int ZB = (PH + 1) * ZROWB;                                 // (+ one row of zeros: the tap row KS of an odd kernel's last pair)
  __shared__ __attribute__((aligned(16))) unsigned char zl[ZB];
  __shared__ __attribute__((aligned(16))) unsigned char est_all[4 * WPIX * EROW];
  const bf16_t* dz = static_cast<const bf16_t*>(a.dz);
  const bf16_t* w = static_cast<const bf16_t*>(a.w);
  bf16_t* out = static_cast<bf16_t*>(a.out);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * BN;

  // ---- weight fragments: channel n0 + 32 f + l31, tap row 2 m + lh, tx = 0 .. 7 (zero for tx >= KS / rows >= KS)
  u32x4 wf[NI][NM];
#pragma unroll
  for (int f = 0; f < NI; ++f)
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int ty = 2 * m + lh, c = n0 + f * 32 + l31;
      uint32_t v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (ty < KS && e < KS && c < a.C) ? (uint32_t)w[(size_t)(ty * KS + e) * a.C + c] : 0u;
      wf[f][m] = u32x4{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
    }
  // the extra zero row of the patch (read by the lanes whose tap row is KS: an odd kernel's last pair)
  for (int i = tid; i < ZROWB / 4; i += 256) reinterpret_cast<uint32_t*>(zl + PH * ZROWB)[i] = 0u;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx_ = t % a.ntx; t /= a.ntx;
    const int ty_ = t % a.nty;
    const int b = t / a.nty;
    const int y0 = ty_ * TH, x0 = tx_ * TW;
    __syncthreads();                                   // every wave is done with the previous tile's patch
    // ---- the scalar patch, reversed, in 8 shifted copies: copy s, element m = rev[m + s], rev[j] = dzp[row][x0 + 31 - j]
    for (int i = tid; i < PH * PWC; i += 256) {
      const int pr = i / PWC, j = i - pr * PWC;
      const int sy = y0 - (KS - 1) + pr, sx = x0 + 31 - j;
      bf16_t v = 0;
      if (sy >= 0 && sy < a.H && sx >= 0 && sx < a.W) v = dz[((size_t)(b * a.H + sy) * a.W + sx) * a.Zc];
      bf16_t* row = reinterpret_cast<bf16_t*>(zl + pr * ZROWB);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int m = j - s;
        if (m >= 0 && m < 32) row[s * 32 + m] = v;
      }
    }
    __syncthreads();
    f32x16 acc[NI][2];
#pragma unroll
    for (int f = 0; f < NI; ++f)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][rr][r] = 0.f;
    // pixel (row 2 wave + rr, column l31), tap row ty = 2 m + lh: patch row (2 wave + rr) + (KS - 1) - ty; elements rev[31 - l31 .. + 7]
    const int j0 = 31 - l31, cs = j0 & 7, ca = j0 >> 3;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int ty = 2 * m + lh;
      u32x4 zf[2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int pr = ty < KS ? 2 * wave + rr + (KS - 1) - ty : PH;      // (tap row KS: the row of zeros)
        zf[rr] = *reinterpret_cast<const u32x4*>(zl + pr * ZROWB + cs * 64 + ca * 16);
      }
#pragma unroll
      for (int f = 0; f < NI; ++f)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) acc[f][rr] = mfma32_head(wf[f][m], zf[rr], acc[f][rr]);
    }
    // ---- epilogue: 16-bit, through wave-private LDS rows (BN channels + 8 B per pixel), 16 bytes per lane, BN / 8 lanes per pixel
    unsigned char* const est = est_all + wave * (WPIX * EROW);
#pragma unroll
    for (int f = 0; f < NI; ++f)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        unsigned char* row = est + (rr * 32 + l31) * EROW + f * 64 + 8 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x2 pk;
          pk.x = pack_bf16x2(acc[f][rr][4 * q], acc[f][rr][4 * q + 1]);
          pk.y = pack_bf16x2(acc[f][rr][4 * q + 2], acc[f][rr][4 * q + 3]);
          *reinterpret_cast<u32x2*>(row + q * 16) = pk;
        }
      }
    __builtin_amdgcn_wave_barrier();                   // (each wave reads back only what it wrote)
    constexpr int LPP = BN / 8, PPI = 64 / LPP;
    const int lc = lane % LPP, nl = n0 + lc * 8;
#pragma unroll 4
    for (int it = 0; it < WPIX / PPI; ++it) {
      const int px = it * PPI + lane / LPP;            // pixel inside the wave's 2 rows x 32 columns
      const int uy = y0 + 2 * wave + (px >> 5), ux = x0 + (px & 31);
      const u32x2 v01 = *reinterpret_cast<const u32x2*>(est + px * EROW + lc * 16);
      const u32x2 v23 = *reinterpret_cast<const u32x2*>(est + px * EROW + lc * 16 + 8);
      if (uy >= a.Hp || ux >= a.Wp || nl >= a.C) continue;
      *reinterpret_cast<u32x4*>(out + (((size_t)b * a.Hp + uy) * a.Wp + ux) * a.C + nl) = u32x4{v01.x, v01.y, v23.x, v23.y};
    }
    __builtin_amdgcn_wave_barrier();
  }
}

bool heads_dgrad_mfma_applicable(const uegan_conv_desc* d) {
  const int cw = d->Cout_w ? d->Cout_w : d->Cout;
  return g_tuning[UEGAN_TUNE_HEADS_MFMA] != 0 && d->dtype == UEGAN_BF16 && cw == 1 && d->Cout == 8 && d->stride == 1 && d->KH == d->KW &&
         (d->KH == 5 || d->KH == 7) && d->C2 == 0 && d->pad_mode == UEGAN_PAD_REFLECT && d->pad == (d->KH - 1) / 2 && d->Ho == d->H && d->Wo == d->W &&
         d->C1 % 32 == 0 && (d->C1 <= 64 || d->C1 % 128 == 0) && d->H > 2 * d->pad + 1 && d->W > 2 * d->pad + 1;
}

// out: [B][H + 2 pad][W + 2 pad][C1] (the caller's consumer folds the mirror images); w_ohwi: the head's FORWARD pack
int heads_dgrad_mfma(const uegan_conv_desc* d, const void* dz, const void* w_ohwi, void* out, hipStream_t s) {
  HeadDgArgs a;
  a.dz = dz; a.w = w_ohwi; a.out = out;
  a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C1; a.Zc = d->Cout; a.p = d->pad;
  a.Kp = (int)uegan_packed_k((int64_t)d->KH * d->KW * d->C1);
  a.Hp = d->H + 2 * d->pad; a.Wp = d->W + 2 * d->pad;
  a.nty = (a.Hp + 7) / 8; a.ntx = (a.Wp + 31) / 32; a.ntiles = d->B * a.nty * a.ntx;
  const int ni = d->C1 >= 128 ? 4 : d->C1 / 32;
  const int nby = d->C1 / (32 * ni);
  int gx = a.ntiles < 2048 / nby ? a.ntiles : 2048 / nby;      // persistent: the weight fragments are fetched once per block
  if (gx < 1) gx = 1;
  const dim3 grid(gx, nby), block(256);
#define UEGAN_HDG(KS)                                                                              \
  do {                                                                                             \
    if (ni == 4) hipLaunchKernelGGL((head_dgrad_mfma_kernel<KS, 4>), grid, block, 0, s, a);        \
    else if (ni == 2) hipLaunchKernelGGL((head_dgrad_mfma_kernel<KS, 2>), grid, block, 0, s, a);   \
    else hipLaunchKernelGGL((head_dgrad_mfma_kernel<KS, 1>), grid, block, 0, s, a);                \
  } while (0)
  if (d->KH == 7) UEGAN_HDG(7); else UEGAN_HDG(5);
#undef UEGAN_HDG
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

}  // namespace uegan
