// Stride-1 convolutions with <= 4 output channels on 32 k input channels (G.dec5.1: 7x7, 32 -> 3 + tanh, models.py:34; round 5: the
// discriminator's prediction heads d2 - d5, 64 ... 512 -> 1, models.py:175-178) as a TOEPLITZ product on the MFMA.
//
// With 3 output channels the implicit GEMM of the other kernels fills 3 of the 16 rows of v_mfma_f32_16x16x32_bf16 and still
// fetches one pixel fragment per tap: 85 TFLOP/s and 0.7 TB/s on a layer whose input is read once.  Here
// the 16 rows are (dx, co) = 4 neighbouring output columns x 4 channel slots, and the reduction runs over (ty, u, c) with u the
// input column relative to the pixel QUAD:
//     out[y][4q + dx][co] = sum_{ty} sum_{u = 0}^{K + 2} sum_c  Wt[(dx, co)][ty][u][c] * in[y + ty - p][4q + u - p][c],
//     Wt[(dx, co)][ty][u][c] = W[co][ty][u - dx][c]  if 0 <= u - dx < K and co < Cout, else 0.
// One MFMA = 16 image rows (the fragment columns) x one quad, K + 3 products per tap row instead of 4 K, and the pixel fragment of
// input column x' serves every quad it overlaps (2.5 on average): 0.45 LDS fragment reads per MFMA instead of 1.  A lane's four
// accumulator rows are the four channel slots of ONE output pixel, so the epilogue is a plain 16-byte store.
//
//   * block = 4 waves, tile = 16 rows x 32 columns (8 quads); persistent over the tile list; two blocks per CU
//   * the Toeplitz weights never exist in memory: wave w owns tap rows {w, w + 4} (split-K) and keeps their K + 3 fragments in
//     REGISTERS, read from the packed [Cout][K*K*C] weights with per-lane (dx, co) addressing -- once per kernel on 32 input channels,
//     once per (tile, 32-channel chunk) on more (round 5: the accumulators then live across the chunks of a tile, the patch is re-staged per
//     chunk; the VALU head kernel these layers used ran at 6 % of the vector pipe's dot-product rate, 0.8 TB/s)
//   * LDS holds only the input patch, (16 + K - 1) rows of ((32 + K - 1) pixels x 64 B + 16 B): the 16-byte row pad makes the
//     16 fragment lanes (consecutive image rows, same column) hit 16 different bank quads; staged with direct-to-LDS loads whose
//     per-lane source address undoes the linear LDS offset (reflection / zero padding resolved there)
//   * the four waves' partial sums meet in LDS (the patch's space, after a barrier), each wave finishes two quads: bias, activation,
//     bf16, store
#include "conv_core.h"

namespace uegan {

// EX (round 6, 32 input channels; uegan_conv2d_fwd_ex): the operands as hi + lo PAIRS -- the "chunks" of a tile are then (hi plane, Whi), (hi plane,
// Wlo: the patch stays, only the weight fragments change), (lo plane, Whi) -- and / or the end of the generator in the epilogue: clamp(tanh(conv) + x,
// -1, 1) (models.py:70-72) from the fp32 accumulator straight into the NCHW fp32 result, beside the 16-bit `out` the backward reads
template <int KS, bool MULTI, int EX = 0>      // MULTI: more than one 32-channel chunk; EX: 1 the residual epilogue, 2 pairs (+ the epilogue)
__global__ void __launch_bounds__(256, 2) conv_toep_kernel(ConvArgs a, int tiles_x, int tiles_total) {
  constexpr int TH = 16, TWX = 32, Q = TWX / 4, NU = KS + 3, PH = TH + KS - 1, PW = TWX + KS - 1, PAD = (KS - 1) / 2;
  constexpr int PXB = 64, RP = PW * PXB + 16, PATCHB = PH * RP, NINST = (PATCHB + 1023) / 1024;
  constexpr int NTY = (KS + 3) / 4;                     // tap rows per wave (wave w: w, w + 4, ...)
  constexpr int REDB = 4 * Q * 1024;                    // partial sums: [wave][quad][lane] x 16 B
  constexpr int LDSB = NINST * 1024 > REDB ? NINST * 1024 : REDB;
  static_assert(KS == 3 || KS == 5 || KS == 7, "odd kernels up to 7");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];

  const ConvGeom& g = a.g;
  const bf16_t* in = static_cast<const bf16_t*>(a.in1);
  const bf16_t* w = static_cast<const bf16_t*>(a.w);
  bf16_t* out = static_cast<bf16_t*>(a.out);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const bool refl = g.pad_mode == UEGAN_PAD_REFLECT;

  // ---- my Toeplitz weight fragments: row (dx, co) = (fr >> 2, fr & 3), channels c0 + 8 fg .. + 7 of tap (ty, u - dx) ----
  u32x4 wf[NTY][NU];
  auto load_w = [&](const bf16_t* w, int c0) {
    const int dx = fr >> 2, co = fr & 3;
#pragma unroll
    for (int i = 0; i < NTY; ++i) {
      const int ty = wave + 4 * i;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int tx = u - dx;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (ty < KS && tx >= 0 && tx < KS && co < a.nbias) {
          v = *reinterpret_cast<const u32x4*>(w + (size_t)co * a.Kp + (size_t)(ty * KS + tx) * g.C + c0 + fg * 8);
        }
        wf[i][u] = v;
      }
    }
  };
  constexpr bool pairs = EX == 2;
  const int nchunk = pairs ? (a.in1_lo ? 3 : 2) : (MULTI ? g.C / 32 : 1);
  if (!MULTI && !pairs) load_w(w, 0);
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = (a.bias && r < a.nbias) ? a.bias[r] : 0.f;

  for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
    int t = tile;
    const int tx_ = t % tiles_x; t /= tiles_x;
    const int ty_ = t % a.nty;
    const int b = t / a.nty;
    const int y0 = ty_ * TH, x0 = tx_ * TWX;

    f32x4 acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // EX: the image values the residual epilogue adds (this wave finishes quads wave * Q / 4 + k: pixel (y0 + fr, x0 + 4 q + fg)), fetched now and
    // consumed behind the tile's MFMAs -- loaded in the epilogue their latency was exposed once per tile
    float rxv[EX != 0 ? Q / 4 : 1][4];
    if constexpr (EX != 0) {
      if (a.res_out) {
        const bool second = b >= a.res_split;
        const float* rx = second ? a.res_x2 : a.res_x;
        const size_t plane = (size_t)g.OH * g.OW, ib = ((size_t)(second ? b - a.res_split : b) * a.nbias) * plane;
#pragma unroll
        for (int k = 0; k < Q / 4; ++k) {
          const int oy = y0 + fr, ox = x0 + 4 * (wave * (Q / 4) + k) + fg;
#pragma unroll
          for (int r = 0; r < 4; ++r) rxv[k][r] = (r < a.nbias && oy < g.OH && ox < g.OW) ? rx[ib + r * plane + (size_t)oy * g.OW + ox] : 0.f;
        }
      }
    }
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      const bool restage = !(pairs && chunk == 1);        // (pairs, second chunk: the same patch against the lo part of the weights)
      const bf16_t* src_t = (pairs && chunk == 2) ? static_cast<const bf16_t*>(a.in1_lo) : in;
      const int c_off = pairs ? 0 : chunk * 32;
      if (chunk && restage) __syncthreads();           // every wave is done with the previous chunk's patch
      // ---- stage the patch: linear LDS offset -> (row, pixel, chunk of 8 channels) per lane ----
      if (restage) {
#pragma unroll
      for (int ii = 0; ii < (NINST + 3) / 4; ++ii) {
        const int inst = ii * 4 + wave;
        if (inst < NINST) {
          const int off = inst * 1024 + lane * 16;
          const int row = off / RP, within = off - row * RP;
          const void* src = g_zero16;
          if (row < PH && within < PW * PXB) {
            const int px = within >> 6, ch = (within >> 4) & 3;
            int sy = y0 + row - PAD, sx = x0 + px - PAD;
            if (refl) { sy = reflect_idx(sy, g.IH); sx = reflect_idx(sx, g.IW); }
            // (tiles may overhang the map; rows / columns more than one reflection away belong to outputs that are never stored)
            if (sy >= 0 && sy < g.IH && sx >= 0 && sx < g.IW) src = src_t + (((size_t)b * g.IH + sy) * g.IW + sx) * g.C + c_off + ch * 8;
          }
          glds16(src, lds + inst * 1024);
        }
      }
      }
      if (MULTI) load_w(w, chunk * 32);                // (this chunk's weight fragments travel beside the patch)
      if (pairs) load_w(chunk == 1 ? static_cast<const bf16_t*>(a.w_lo) : w, 0);
      if (restage) {
        wait_vmcnt<0>();
        __syncthreads();
      }

      // ---- my tap rows over the whole tile ----
#pragma unroll
      for (int i = 0; i < NTY; ++i) {
        const int ty = wave + 4 * i;
        if (ty < KS) {
          const unsigned char* rowp = lds + (fr + ty) * RP + fg * 16;
#pragma unroll
          for (int xc = 0; xc < PW; ++xc) {
            const u32x4 xf = *reinterpret_cast<const u32x4*>(rowp + xc * PXB);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
              const int u = xc - 4 * q;
              if (u >= 0 && u < NU) acc[q] = mfma_bf16(wf[i][u], xf, acc[q]);
            }
          }
        }
      }
    }
    __syncthreads();                                   // every wave is done with the patch: its space takes the partial sums

    // ---- split-K reduction through LDS; wave v finishes quads 2v, 2v + 1 ----
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int q = 0; q < Q; ++q) *reinterpret_cast<f32x4*>(red + ((wave * Q + q) * 64 + lane) * 4) = acc[q];
    __syncthreads();
    const float scale = a.scale ? a.scale[a.scale_group ? b / a.scale_group : 0] : 1.f;
#pragma unroll
    for (int k = 0; k < Q / 4; ++k) {
      const int q = wave * (Q / 4) + k;
      f32x4 s = *reinterpret_cast<const f32x4*>(red + ((0 * Q + q) * 64 + lane) * 4);
#pragma unroll
      for (int v = 1; v < 4; ++v) {
        if (v < KS) {                                  // (waves beyond the tap rows hold zeros; fixed order: deterministic)
          const f32x4 p = *reinterpret_cast<const f32x4*>(red + ((v * Q + q) * 64 + lane) * 4);
          s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
        }
      }
      const int oy = y0 + fr, ox = x0 + 4 * q + fg;
      if (oy < g.OH && ox < g.OW) {
        float v4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] = r < a.nbias ? apply_act(s[r] * scale + bv[r], a.act) : 0.f;
        bf16_t* o = out + (((size_t)b * g.OH + oy) * g.OW + ox) * a.N;
        u32x4 pk;
        pk[0] = pack_bf16x2(v4[0], v4[1]); pk[1] = pack_bf16x2(v4[2], v4[3]); pk[2] = 0u; pk[3] = 0u;
        *reinterpret_cast<u32x4*>(o) = pk;             // (a.N == 8: one 16-byte chunk per pixel, channels 4..7 zero)
        if constexpr (EX != 0) {
          if (a.res_out) {      // models.py:72: clamp(res + x, -1, 1), NCHW fp32, from the fp32 result
            const bool second = b >= a.res_split;
            float* ro = second ? a.res_out2 : a.res_out;
            const size_t i0 = (((size_t)(second ? b - a.res_split : b) * a.nbias) * g.OH + oy) * g.OW + ox, plane = (size_t)g.OH * g.OW;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (r < a.nbias) ro[i0 + r * plane] = fminf(fmaxf(v4[r] + rxv[k][r], -1.f), 1.f);
          }
        }
      }
    }
    __syncthreads();                                   // the partial sums are consumed: the next tile's patch may land
  }
}

// is this a layer the kernel takes?  (more than 32 input channels -- the prediction heads d2 - d5 -- only with UEGAN_TUNE_TOEP_HEADS, the default)
bool conv_toep_takes(const ConvArgs& a, int dtype) {
  const ConvGeom& g = a.g;
  if (dtype != UEGAN_BF16 || g.stride != 1 || g.KH != g.KW || g.C2 != 0 || g.C % 32 != 0 || g.C > 1024 || a.N != 8 || a.nbias > 4 || a.out2 || a.mask) return false;
  const bool ex = a.w_lo || a.in1_lo || a.res_out;
  if (ex && (g.C != 32 || g.KH != 7 || (a.in1_lo && !a.w_lo) || a.in2_lo || a.out_lo || a.mul || (a.res_out && !a.res_x))) return false;      // (one instantiation: dec5.1)
  if (!ex && (a.out_lo || a.mul || a.in2_lo)) return false;
  // (a tile walks its chunks one after the other -- stage, wait, multiply: on the 256 / 512-channel heads' 32^2 / 16^2 maps that chain is longer than the
  // vector-ALU kernel's whole launch, 0.032 / 0.058 vs 0.016 / 0.028 ms at batch 16; the 64 / 128-channel heads gain 0.050 -> 0.026 and 0.039 -> 0.022.
  // Knob value 2 lifts the limit: tests)
  if (g.C != 32 && (g_tuning[UEGAN_TUNE_TOEP_HEADS] == 0 || (g.C > 128 && g_tuning[UEGAN_TUNE_TOEP_HEADS] != 2))) return false;
  if (g.KH != 7 && g.KH != 5 && g.KH != 3) return false;
  if (g.pad != (g.KH - 1) / 2 || g.OH != g.IH || g.OW != g.IW) return false;
  if (g.mode != 0) return false;                    // (forward only: no data gradient in the networks has this shape)
  if (g.C == 32 ? (g.OH < 16 || g.OW < 32) : (g.OH < 8 || g.OW < 8)) return false;      // (tiles may overhang: masked stores, out-of-range sources are zeros)
  return true;
}

// 1: not a layer this kernel takes
int conv_toep_run(ConvArgs& a, int dtype, hipStream_t s) {
  const ConvGeom& g = a.g;
  if (!conv_toep_takes(a, dtype)) return 1;
  a.nty = (g.OH + 15) / 16;
  const int tiles_x = (g.OW + 31) / 32;
  const int total = g.B * a.nty * tiles_x;
  const int grid = total < 512 ? total : 512;
  ProfScope prof(prof_key(6, true, 4, g.KH, g.mode, 16, true), 2.0 * (double)g.B * g.OH * g.OW * a.nbias * (double)(g.KH * g.KW * g.C), s,
                 2.0 * ((double)g.B * g.OH * g.OW * a.N + (double)g.B * g.IH * g.IW * g.C));
  if (a.w_lo) {
    hipLaunchKernelGGL((conv_toep_kernel<7, false, 2>), dim3(grid), dim3(256), 0, s, a, tiles_x, total);
  } else if (a.res_out) {
    hipLaunchKernelGGL((conv_toep_kernel<7, false, 1>), dim3(grid), dim3(256), 0, s, a, tiles_x, total);
  } else if (g.C == 32) {
    if (g.KH == 7) hipLaunchKernelGGL((conv_toep_kernel<7, false>), dim3(grid), dim3(256), 0, s, a, tiles_x, total);
    else if (g.KH == 5) hipLaunchKernelGGL((conv_toep_kernel<5, false>), dim3(grid), dim3(256), 0, s, a, tiles_x, total);
    else hipLaunchKernelGGL((conv_toep_kernel<3, false>), dim3(grid), dim3(256), 0, s, a, tiles_x, total);
  } else {
    if (g.KH == 7) hipLaunchKernelGGL((conv_toep_kernel<7, true>), dim3(grid), dim3(256), 0, s, a, tiles_x, total);
    else if (g.KH == 5) hipLaunchKernelGGL((conv_toep_kernel<5, true>), dim3(grid), dim3(256), 0, s, a, tiles_x, total);
    else hipLaunchKernelGGL((conv_toep_kernel<3, true>), dim3(grid), dim3(256), 0, s, a, tiles_x, total);
  }
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

}  // namespace uegan
