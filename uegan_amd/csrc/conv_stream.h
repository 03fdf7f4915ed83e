// Streaming convolution kernel for the thin full-resolution layers (bf16, stride 1 or forward stride 2, <= 64 input and <= 64 output
// channels: enc1, dec4, dec5, GAM-1, the 7x7 heads and their data gradients).  Included by conv.hip only.
//
// These layers move hundreds of MB through a few GFLOP: they are bound by HBM and by launch-to-launch latency, not by
// MFMA.  The tile-per-block kernels above re-stage the weights for every tile and expose the load latency of every tile.
// Here a block is persistent over a range of tiles:
//   * the whole packed weight matrix [N][taps*C] lives in LDS for the life of the block (<= 40 KB)
//   * the (TH+K-1) x (16+K-1) pixel patch of tile t+1 streams into the second LDS buffer (direct-to-LDS loads issued
//     from inline asm, see wgtr_glds16 in wgrad_tr.h) while the taps of tile t walk over the first one
//   * a K step is 32 (tap, channel) reduction elements: with 8/16-channel tensors one MFMA covers 4/2 taps, each lane
//     group reading its own tap's pixel row; the per-lane patch offset of every K step is tile independent and kept in a
//     small LDS table
//   * output: D rows = output channels (4 consecutive per lane), columns = the 16 pixels of one tile row
// Forward (either padding) handles image borders itself (reflected / zero-filled patch loads).  The data gradient of a
// reflection-padded convolution is computed as the plain flipped-tap correlation with zero fill (every pixel's direct image); the
// mirrored images of the pixels within `pad` of a border are added by dgrad_images_kernel (conv.hip) afterwards.
#pragma once

struct ConvStreamArgs {
  ConvArgs c;                 // geometry, tensors, epilogue
  int flip;                   // 1: taps read the patch at (K-1-ty, K-1-tx) (data gradient)
  int org;                    // patch origin relative to the tile's first pixel, both axes: -pad (fwd), pad-(K-1) (dgrad)
  int zero_fill;              // 1: patch pixels outside the source image are zero, 0: reflected
  int cls;                    // 1: data gradient of a stride-2 conv: the tile is TH x 16 positions of the half-resolution grid, its
                              // four parity classes (output pixel 2i+py, 2j+px) are computed one after the other from ONE dz patch;
                              // the K steps are grouped by class (kstart), each with its own taps
  int kstart[5];
  int ymin;                   // cls: patch origin relative to the tile's first half-resolution position (both axes)
  int sx;                     // convolution stride (forward only: 1 or 2); output pixel (i, j) reads patch pixel (sx*i + ty, sx*j + tx)
  int TH, PW, PH, PWmagic, KWmagic;
  int rb, rblog, Clog;        // patch LDS bytes per pixel (= C*2), log2, log2(C)
  int ksteps, taps;
  int wrow, wrows;            // weight LDS row bytes, rows held
  int wbytes, tbytes, xbytes; // LDS regions: weights, per-K-step lane offsets, one patch buffer
  int mtab_off;               // byte offset (from the table region) of the mirror tables [2][ksteps][64] + flags [2][ksteps]
  int xmir;                   // 1: reflection-padded data gradient (stride 1, OW a multiple of 16): the tiles of the first / last tile column
                              // add the x-mirrored images of their pixels 1..pad / OW-1-pad..OW-2 as extra K steps (tables mtab below)
  int ty0, ty1, tx0, tx1;     // tile rectangle to process (units of TH x 16 tiles)
  int tiles_total, tiles_per_block;
  int abl;                    // timing ablations (tools build only, UEGAN_ABL_BITS; results are garbage): 1 no staging after the first tile, 2 no K loop, 4 no stores
  // hi + lo pairs (template parameter PR, see the kernel): the lo plane of the patch follows the hi plane inside each patch buffer at byte lo_xoff,
  // c_lo channels = rb_lo bytes per pixel; the lo part of the weights follows the hi matrix at wlo_off;
  // PR 3: the lo plane's own per-tap lane offsets at tab + tlo_off
  int lo_xoff, c_lo, rb_lo, rblog_lo, wlo_off, tlo_off;
};

// XOR on the 16-byte chunk index of a patch pixel in column pcol: swz128 / swz64 of conv_core.h on the COLUMN (a fragment reads 16
// consecutive columns of one patch row, and the row's first pixel only shifts the pattern), so a lane's offset is tile independent.
// Conflict free for stride 1; the stride-2 forwards (every second column) keep a 2-way conflict on their pixel fragments.
__device__ __forceinline__ int cs_swz(int rb, int pcol) {
  return rb == 128 ? swz128(pcol) : (rb == 64 ? swz64(pcol) : 0);
}

// LDS classes (static size = occupancy): 0: 53 KB, three blocks per CU; 1: 80 KB, two; 2: 152 KB, one (weights + patches too
// large otherwise)
constexpr int CS_LDS_KB[4] = {53, 80, 152, 160};      // (3: the whole LDS; no instantiation since dec4's pair planes are packed exactly and fit class 2)

// NW: waves per block (4; 8 for the one-block-per-CU class, so that a SIMD still holds two waves to hide each other's LDS / load latency:
// dec4's forward, whose 39 KB of weights + two 45-KB patches leave room for one block, ran 4 waves per CU at 1.7 TB/s)
// STATS: the forward also accumulates sum / sum of squares of its (fp32, activated) results per channel in registers and writes them per image of
// the block's tile range (ConvArgs::stats_part): a block's tiles are consecutive, so it flushes once or twice per launch
// PR (round 6, forward only): operands as hi + lo PAIRS of 16-bit planes (value = hi + lo, ~2 x the significant bits of the storage format) --
//   1: the weights (Whi from c.w, Wlo from c.w_lo; both matrices in LDS): every K step runs Whi b + Wlo b
//   2: ... and the (single) source: the patch holds the hi plane and, behind it, the lo plane in the same layout; a step runs Whi bhi + Wlo bhi + Whi blo
//      (Wlo blo, 2^-22 relative, is dropped)
//   3: ... two 32-channel sources of which the SECOND has a lo plane (dec4: upsampled branch plain, attention branch hi + lo): the odd K steps (the second
//      source's channels of a tap) add Whi blo with the fragment of the 32-channel lo plane
// EPX: epilogue extras of uegan_conv2d_fwd_ex; no mask, no second destination --
//   1: the lo plane of the result (out_lo)
//   2: the product with a second tensor formed from the fp32 result (mul / mul_lo -> out_mul / out_mul_lo).  The multiplier's values of a tile are
//      fetched BEFORE the tile's K loop and consumed behind it: loaded in the epilogue their latency was exposed once per tile (one block per CU in
//      lockstep: dec4 + product 0.96 ms per 32 images against 0.58 + 0.32 for the two separate kernels)
template <int TN, int PF, int LC, bool CLS, bool XMIR = false, int NW = 4, bool STATS = false, int PR = 0, int EPX = 0>
// (second launch bound = waves per SIMD: blocks per CU x NW / 4)
__global__ void __launch_bounds__(64 * NW, (LC >= 2 ? 1 : (LC == 1 ? 2 : 3)) * NW / 4) conv_stream_kernel(ConvStreamArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int MAXIX = (LC >= 2 ? 16 : 10) * 4 / NW;
  constexpr bool WLO = PR >= 1;
  static_assert(PR == 0 || (!CLS && !XMIR), "hi + lo pairs: plain forward only");
  __shared__ __attribute__((aligned(16))) unsigned char lds[CS_LDS_KB[LC] * 1024];
  const ConvArgs& ca = a.c;
  const ConvGeom& g = ca.g;
  const unsigned char* in1 = static_cast<const unsigned char*>(ca.in1);
  const unsigned char* in2 = static_cast<const unsigned char*>(ca.in2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fj = lane & 15, fg = lane >> 4;
  unsigned char* wl = lds;
  unsigned char* tab = lds + a.wbytes;
  unsigned char* xb0 = tab + a.tbytes;

  if (XMIR) {
    int* mflag = reinterpret_cast<int*>(tab + a.mtab_off) + 2 * a.ksteps * 64;
    for (int i = tid; i < 2 * a.ksteps; i += NT) mflag[i] = 0;
    __syncthreads();
  }
  // ---- weights -> LDS once: rows n < wrows (N rounded up to 8), ksteps*64 bytes each, zero beyond Kp; chunk position g ^ swz64(n)
  {
    const int cpr = a.ksteps * 4;
    const int total = a.wrows * cpr;
    const bf16_t* w = static_cast<const bf16_t*>(ca.w);
    for (int idx = tid; idx < total; idx += NT) {
      const int n = idx / cpr, q = idx - n * cpr;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (n < ca.N && q * 8 < ca.Kp) v = *reinterpret_cast<const u32x4*>(w + (size_t)n * ca.Kp + q * 8);
      *reinterpret_cast<u32x4*>(wl + n * a.wrow + (q >> 2) * 64 + (((q & 3) ^ swz64(n)) << 4)) = v;
      if constexpr (WLO) {
        u32x4 vl = u32x4{0u, 0u, 0u, 0u};
        if (n < ca.N && q * 8 < ca.Kp) vl = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(ca.w_lo) + (size_t)n * ca.Kp + q * 8);
        *reinterpret_cast<u32x4*>(wl + a.wlo_off + n * a.wrow + (q >> 2) * 64 + (((q & 3) ^ swz64(n)) << 4)) = vl;
      }
    }
    // per-K-step patch offset of every lane (B fragment: lane (j = pixel column, g) holds k = 32 s + 8 g .. + 7) and the
    // byte offset of the step's 64-byte slice inside a weight row (tab2: K steps follow the packed order unless cls)
    int* tab2 = reinterpret_cast<int*>(tab + a.ksteps * 256);
    if (!CLS) {
      for (int idx = tid; idx < a.ksteps * 64; idx += NT) {
        const int s = idx >> 6, l = idx & 63;
        const int k = 32 * s + 8 * (l >> 4);
        int tap = k >> a.Clog;
        const int chunk = (k & ((1 << a.Clog) - 1)) >> 3;
        if (tap >= a.taps) tap = 0;                    // padding K steps: the weights there are zero, any finite pixel will do
        int ty = (tap * a.KWmagic) >> 16, tx = tap - ty * g.KW;
        if (a.flip) { ty = g.KH - 1 - ty; tx = g.KW - 1 - tx; }
        const int pcol = a.sx * (l & 15) + tx;
        *reinterpret_cast<int*>(tab + idx * 4) = (ty * a.PW + pcol) * a.rb + ((chunk ^ cs_swz(a.rb, pcol)) << 4);
        if (l == 0) tab2[s] = s * 64;
        if (XMIR) {
          // x-mirrored images (data gradient, flipped taps: the direct image of tile column j reads patch column j + txf).
          //   left tile (x0 = 0):   pixel j in 1..pad, taps with txf >= pad + j   -> patch column txf - j
          //   right tile (D = 15):  pixel j in D-pad..D-1, taps with txf <= pad - (D - j) -> patch column 2 D - j + txf
          // entries without an image carry bit 31: the fragment is zeroed after the load
          int* mt = reinterpret_cast<int*>(tab + a.mtab_off);
          int* mflag = mt + 2 * a.ksteps * 64;         // [2][ksteps]: the step has an image for some lane (zeroed below, before the barrier)
          const int j = l & 15, txf = tx, D = 15;
          const bool real_tap = (k >> a.Clog) < a.taps;
          int oL = (int)0x80000000, oR = (int)0x80000000;
          if (real_tap && j >= 1 && j <= g.pad && txf >= g.pad + j) {
            const int pc = txf - j;
            oL = (ty * a.PW + pc) * a.rb + ((chunk ^ cs_swz(a.rb, pc)) << 4);
          }
          if (real_tap && j >= D - g.pad && j <= D - 1 && txf <= g.pad - (D - j)) {
            const int pc = 2 * D - j + txf;
            oR = (ty * a.PW + pc) * a.rb + ((chunk ^ cs_swz(a.rb, pc)) << 4);
          }
          mt[idx] = oL;
          mt[a.ksteps * 64 + idx] = oR;
          if (oL >= 0) mflag[s] = 1;                   // (benign race: every writer stores 1)
          if (oR >= 0) mflag[a.ksteps + s] = 1;
        }
      }
      if constexpr (PR == 3) {
        // the second source's lo plane (32 channels, 64-byte pixels): lane (j, g) of tap t reads channels 8 g .. 8 g + 7 of patch pixel (ty, j + tx)
        for (int idx = tid; idx < a.taps * 64; idx += NT) {
          const int tap = idx >> 6, l = idx & 63;
          const int ty = (tap * a.KWmagic) >> 16, tx = tap - ty * g.KW;
          const int pcol = (l & 15) + tx;
          *reinterpret_cast<int*>(tab + a.tlo_off + idx * 4) = (ty * a.PW + pcol) * a.rb_lo + (((l >> 4) ^ cs_swz(a.rb_lo, pcol)) << 4);
        }
      }
    } else {
      // class c = 2 py + px owns the taps with (py + pad - ty) and (px + pad - tx) even; source = i + (py + pad - ty) / 2
      const int spt = g.C >> 5;                        // K steps per tap (C = 32 or 64)
      for (int idx = tid; idx < a.ksteps * 64; idx += NT) {
        const int s = idx >> 6, l = idx & 63;
        int c = 0;
        while (c < 3 && s >= a.kstart[c + 1]) ++c;
        const int py = c >> 1, px = c & 1;
        int rem = (s - a.kstart[c]) / spt;             // index of the tap inside the class (ty-major)
        const int half = (s - a.kstart[c]) - rem * spt;
        const int ty0 = (py + g.pad) & 1, tx0 = (px + g.pad) & 1;
        const int ntx = (g.KW - tx0 + 1) >> 1;
        const int tyq = rem / ntx, txq = rem - tyq * ntx;
        const int ty = ty0 + 2 * tyq, tx = tx0 + 2 * txq;
        const int prow = (py + g.pad - ty) / 2 - a.ymin;
        const int pcol = (l & 15) + (px + g.pad - tx) / 2 - a.ymin;
        const int chunk = half * 4 + (l >> 4);
        *reinterpret_cast<int*>(tab + idx * 4) = (prow * a.PW + pcol) * a.rb + ((chunk ^ cs_swz(a.rb, pcol)) << 4);
        if (l == 0) tab2[s] = ((ty * g.KW + tx) * g.C) * 2 + half * 64;
      }
    }
  }

  // ---- per-thread staging table: byte offset of my chunk of round `it` from the patch origin (interior tiles)
  const int cprlog = a.rblog - 4;
  const int nxc = (a.PH * a.PW) << cprlog;
  // (PR >= 2: the lo plane's chunks follow the hi plane's directly, from chunk nlo0; bit 30 of a table entry = "from the lo source")
  const int cprlog_lo = PR >= 2 ? a.rblog_lo - 4 : 0;
  const int nlo0 = PR >= 2 ? a.lo_xoff >> 4 : 0, nxcl = PR >= 2 ? (a.PH * a.PW) << cprlog_lo : 0;
  const int nix = PR >= 2 ? (nlo0 + nxcl + NT - 1) / NT : (nxc + NT - 1) / NT;
  const unsigned char* in_lo = PR == 2 ? static_cast<const unsigned char*>(ca.in1_lo) : (PR == 3 ? static_cast<const unsigned char*>(ca.in2_lo) : nullptr);
  const bool two_src = g.C2 != 0;
  // chunk L of a patch buffer -> patch pixel, first channel inside its source, source (0: in1, 1: in2, 2: the lo plane); false: padding chunk
  auto decode = [&](int L, int& prow, int& pcol, int& c, int& sel) -> bool {
    if (PR >= 2 && L >= nlo0) {
      const int L2 = L - nlo0;
      if (L2 >= nxcl) return false;
      const int r = L2 >> cprlog_lo, pos = L2 & ((1 << cprlog_lo) - 1);
      prow = (r * a.PWmagic) >> 16; pcol = r - prow * a.PW;
      c = (pos ^ cs_swz(a.rb_lo, pcol)) << 3;
      sel = 2;
      return true;
    }
    if (PR >= 2 && L >= nxc) return false;
    const int r = L >> cprlog, pos = L & ((1 << cprlog) - 1);
    prow = (r * a.PWmagic) >> 16; pcol = r - prow * a.PW;
    c = (pos ^ cs_swz(a.rb, pcol)) << 3;
    sel = c < g.C1 ? 0 : 1;
    if (sel) c -= g.C1;
    return true;
  };
  uint32_t xoff[MAXIX];
#pragma unroll
  for (int it = 0; it < MAXIX; ++it) {
    const int L = it * NT + tid;
    xoff[it] = PR >= 2 ? 0xffffffffu : 0u;      // (PR >= 2: the planes are packed exactly -- a lane without a chunk must not write)
    int prow, pcol, c, sel;
    if (L < (PR >= 2 ? nlo0 + nxcl : nxc) && decode(L, prow, pcol, c, sel)) {
      if (sel == 0) xoff[it] = (uint32_t)(((prow * g.IW + pcol) * g.C1 + c) * 2);
      else if (sel == 1) xoff[it] = (uint32_t)(((prow * g.IW + pcol) * g.C2 + c) * 2) | 0x80000000u;
      else xoff[it] = (uint32_t)(((prow * g.IW + pcol) * a.c_lo + c) * 2) | 0x40000000u;
    }
  }
  // A fragment rows of this lane
  int abase[TN];
#pragma unroll
  for (int nf = 0; nf < TN; ++nf) {
    int n = nf * 16 + fj;
    if (n >= a.wrows) n -= a.wrows;                    // fragment rows beyond N: any stored row (those outputs are dropped)
    abase[nf] = n * a.wrow + ((fg ^ swz64(n)) << 4);
  }
  float bv[TN][4];
#pragma unroll
  for (int nf = 0; nf < TN; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nf * 16 + fg * 4 + r;
      bv[nf][r] = (ca.bias && n < ca.nbias) ? ca.bias[n] : 0.f;
    }
  __syncthreads();

  const int nrx = a.tx1 - a.tx0, nry = a.ty1 - a.ty0;
  int t_begin = blockIdx.x * a.tiles_per_block, t_end = t_begin + a.tiles_per_block;
  if (t_end > a.tiles_total) t_end = a.tiles_total;

  auto tile_origin = [&](int t, int& b, int& oy0, int& ox0) {
    const int txi = t % nrx;
    const int tq = t / nrx;
    const int tyi = tq % nry;
    b = tq / nry;
    oy0 = (a.ty0 + tyi) * a.TH;
    ox0 = (a.tx0 + txi) * 16;
  };

  auto stage = [&](int t, int bufi) {
    unsigned char* xb = xb0 + bufi * a.xbytes;
    int b, oy0, ox0;
    tile_origin(t, b, oy0, ox0);
    const int iy0 = CLS ? oy0 + a.ymin : a.sx * oy0 + a.org, ix0 = CLS ? ox0 + a.ymin : a.sx * ox0 + a.org;
    const bool inside = iy0 >= 0 && iy0 + a.PH <= g.IH && ix0 >= 0 && ix0 + a.PW <= g.IW;
    if (inside) {
      const long long pix0 = ((long long)b * g.IH + iy0) * g.IW + ix0;
      const unsigned char* o1 = in1 + pix0 * g.C1 * 2;
      const unsigned char* o2 = in2 + pix0 * g.C2 * 2;
      const unsigned char* o3 = PR >= 2 ? in_lo + pix0 * a.c_lo * 2 : nullptr;
#pragma unroll
      for (int it = 0; it < MAXIX; ++it)
        if (it < nix) {
          const uint32_t o = xoff[it];
          unsigned char* dst = xb + (it * NT + wave * 64) * 16;
          if (PR >= 2 && o == 0xffffffffu) continue;
          if (PR >= 2 && (o & 0x40000000u)) wgtr_glds16(o3, o & 0x3fffffffu, dst);
          else if (two_src && (o >> 31)) wgtr_glds16(o2, o & 0x7fffffffu, dst);
          else wgtr_glds16(o1, o, dst);
        }
    } else if (!a.zero_fill) {
      // reflected border: every patch pixel is a real pixel (rows/columns beyond a single reflection are never read by a
      // valid output pixel: clamp them)
      const long long img = (long long)b * g.IH * g.IW;
      const unsigned char* o1 = in1 + img * g.C1 * 2;
      const unsigned char* o2 = in2 + img * g.C2 * 2;
      const unsigned char* o3 = PR >= 2 ? in_lo + img * a.c_lo * 2 : nullptr;
#pragma unroll 1
      for (int it = 0; it < nix; ++it) {
        const int L = it * NT + tid;
        int prow, pcol, c, sel;
        unsigned char* dst = xb + (it * NT + wave * 64) * 16;
        if (!decode(L, prow, pcol, c, sel)) continue;      // (PR >= 2 only: lanes beyond the last chunk)
        int iy = iy0 + prow, ix = ix0 + pcol;
        iy = iy < 0 ? -iy : iy;
        iy = iy >= g.IH ? 2 * (g.IH - 1) - iy : iy;
        iy = iy < 0 ? 0 : (iy >= g.IH ? g.IH - 1 : iy);
        ix = ix < 0 ? -ix : ix;
        ix = ix >= g.IW ? 2 * (g.IW - 1) - ix : ix;
        ix = ix < 0 ? 0 : (ix >= g.IW ? g.IW - 1 : ix);
        const uint32_t pix = (uint32_t)(iy * g.IW + ix);
        if (sel == 0) wgtr_glds16(o1, (pix * (uint32_t)g.C1 + (uint32_t)c) * 2u, dst);
        else if (sel == 1) wgtr_glds16(o2, (pix * (uint32_t)g.C2 + (uint32_t)c) * 2u, dst);
        else wgtr_glds16(o3, (pix * (uint32_t)a.c_lo + (uint32_t)c) * 2u, dst);
      }
    } else {
#pragma unroll 1
      for (int it = 0; it < nix; ++it) {
        const int L = it * NT + tid;
        const void* src = g_zero16;
        if (L < nxc) {
          const int r = L >> cprlog, pos = L & ((1 << cprlog) - 1);
          const int prow = (r * a.PWmagic) >> 16, pcol = r - prow * a.PW;
          const int c = (pos ^ cs_swz(a.rb, pcol)) << 3;
          const int iy = iy0 + prow, ix = ix0 + pcol;
          if (iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW) {
            const long long pix = ((long long)b * g.IH + iy) * g.IW + ix;
            src = c < g.C1 ? in1 + (pix * g.C1 + c) * 2 : in2 + (pix * g.C2 + (c - g.C1)) * 2;
          }
        }
        wgtr_glds16(src, xb + (it * NT + wave * 64) * 16);
      }
    }
  };

  bf16_t* out = static_cast<bf16_t*>(ca.out);
  float st1[STATS ? TN : 1][4], st2[STATS ? TN : 1][4];
  int st_b = -1;
  const int st_b0 = STATS ? t_begin / (ca.stats_tpi > 0 ? ca.stats_tpi : 1) : 0;      // first image of this block's tile range
  if (STATS) {
#pragma unroll
    for (int nf = 0; nf < TN; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) { st1[nf][r] = 0.f; st2[nf][r] = 0.f; }
  }
  // sums over the 16 pixel-column lanes of a channel group, then one (sum, sum of squares) pair per channel of this wave for image `bimg`
  auto stats_flush = [&](int bimg) {
    if constexpr (STATS) {
      float* dst = ca.stats_part + ((size_t)((blockIdx.x * 2 + (bimg - st_b0)) * NW + wave) * (TN * 16)) * 2;
#pragma unroll
      for (int nf = 0; nf < TN; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a1 = st1[nf][r], a2 = st2[nf][r];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
          if (fj == 0) { dst[(nf * 16 + fg * 4 + r) * 2] = a1; dst[(nf * 16 + fg * 4 + r) * 2 + 1] = a2; }
          st1[nf][r] = 0.f; st2[nf][r] = 0.f;
        }
    }
  };
  const int row0 = wave * PF;
  int sc_grp = -1;
  float scale = 1.f;
  const int rowpitch = a.sx * a.PW * a.rb;            // LDS distance between the patch rows of consecutive output rows
  for (int t = t_begin - 1; t < t_end; ++t) {
    const bool have = t >= t_begin;
    const int bufi = (t - t_begin) & 1;
    if (have) {
      wgtr_wait_loads();           // (a counted wait that leaves the previous epilogue's stores in flight measured no gain)
      raw_barrier();               // tile t landed for every wave; everyone is done reading the other buffer
    }
    if (t + 1 < t_end && !((UEGAN_ABL_BITS(a.abl) & 1) && have)) stage(t + 1, bufi ^ 1);
    if (!have) continue;
    const unsigned char* xw = xb0 + bufi * a.xbytes + row0 * rowpitch;
    const int* tab2 = reinterpret_cast<const int*>(tab + a.ksteps * 256);
    constexpr int ncls = CLS ? 4 : 1;
   for (int cl = 0; cl < ncls; ++cl) {
    const int ks0 = CLS ? a.kstart[cl] : 0, ks1 = CLS ? a.kstart[cl + 1] : a.ksteps;
    f32x4 acc[TN][PF];
#pragma unroll
    for (int nf = 0; nf < TN; ++nf)
#pragma unroll
      for (int i = 0; i < PF; ++i) acc[nf][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x2 mulh[EPX == 2 ? TN : 1][EPX == 2 ? PF : 1], mull[EPX == 2 ? TN : 1][EPX == 2 ? PF : 1];
    if constexpr (EPX == 2) {
      // the multiplier's 4 channels (8 bytes; + its lo plane) of every result this lane will hold: in flight under the K loop
      int bq, oyq, oxq;
      tile_origin(t, bq, oyq, oxq);
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int oy = oyq + row0 + i, oxp = oxq + fj;
        const bool pvq = oy < g.OH && oxp < g.OW;
        const size_t pq = ((size_t)bq * g.OH + oy) * g.OW + oxp;
#pragma unroll
        for (int nf = 0; nf < TN; ++nf) {
          mulh[nf][i] = u32x2{0u, 0u}; mull[nf][i] = u32x2{0u, 0u};
          if (pvq && nf * 16 + fg * 4 < ca.N) {
            const size_t eo = pq * ca.N + nf * 16 + fg * 4;
            mulh[nf][i] = *reinterpret_cast<const u32x2*>(static_cast<const bf16_t*>(ca.mul) + eo);
            if (ca.mul_lo) mull[nf][i] = *reinterpret_cast<const u32x2*>(static_cast<const bf16_t*>(ca.mul_lo) + eo);
          }
        }
      }
    }
    // K loop, software pipelined by ping-pong: two register sets, the fragments of step s+1 are in flight behind the MFMAs of
    // step s (unrolled by two, so no register copies -- with 2-8 MFMAs per step the copies of a rotating pipeline cost
    // more than the MFMAs)
    u32x4 a0[TN], b0[PF], a1[TN], b1[PF];
   if constexpr (PR != 0) {
    // hi + lo pairs (see the template parameter): per K step the weight fragments of both matrices, the pixel fragment of the hi plane and -- PR 2: every
    // step, PR 3: the odd steps = the second source's channels -- of the lo plane; 2 or 3 MFMAs per (fragment, row)
    const int rowpitch_lo = a.sx * a.PW * a.rb_lo;
    const unsigned char* xwl = xb0 + bufi * a.xbytes + a.lo_xoff + row0 * rowpitch_lo;
    u32x4 al0[TN], al1[TN], bl0[PF], bl1[PF];
    auto load_p = [&](int s, auto odd_c, u32x4 (&af)[TN], u32x4 (&afl)[TN], u32x4 (&bf)[PF], u32x4 (&bfl)[PF]) {
      constexpr bool ODD = decltype(odd_c)::value;
      const int boff = *reinterpret_cast<const int*>(tab + (s * 64 + lane) * 4);
#pragma unroll
      for (int nf = 0; nf < TN; ++nf) {
        af[nf] = *reinterpret_cast<const u32x4*>(wl + abase[nf] + s * 64);
        afl[nf] = *reinterpret_cast<const u32x4*>(wl + a.wlo_off + abase[nf] + s * 64);
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) bf[i] = *reinterpret_cast<const u32x4*>(xw + boff + i * rowpitch);
      if constexpr (PR == 2) {
#pragma unroll
        for (int i = 0; i < PF; ++i) bfl[i] = *reinterpret_cast<const u32x4*>(xwl + boff + i * rowpitch_lo);      // (same layout as the hi plane)
      }
      if constexpr (PR == 3 && ODD) {
        const int bl = *reinterpret_cast<const int*>(tab + a.tlo_off + ((s >> 1) * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < PF; ++i) bfl[i] = *reinterpret_cast<const u32x4*>(xwl + bl + i * rowpitch_lo);
      }
    };
    auto mma_p = [&](auto odd_c, const u32x4 (&af)[TN], const u32x4 (&afl)[TN], const u32x4 (&bf)[PF], const u32x4 (&bfl)[PF]) {
      constexpr bool ODD = decltype(odd_c)::value;
#pragma unroll
      for (int nf = 0; nf < TN; ++nf)
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          if constexpr (PR == 2 || (PR == 3 && ODD)) acc[nf][i] = mfma_bf16(af[nf], bfl[i], acc[nf][i]);
          acc[nf][i] = mfma_bf16(afl[nf], bf[i], acc[nf][i]);
          acc[nf][i] = mfma_bf16(af[nf], bf[i], acc[nf][i]);
        }
    };
    const std::false_type even_c{};
    const std::true_type odd_c{};
    load_p(0, even_c, a0, al0, b0, bl0);
    for (int s = 0; s < ks1; s += 2) {
      if (s + 1 < ks1) load_p(s + 1, odd_c, a1, al1, b1, bl1);
      mma_p(even_c, a0, al0, b0, bl0);
      if (s + 1 >= ks1) break;
      if (s + 2 < ks1) load_p(s + 2, even_c, a0, al0, b0, bl0);
      mma_p(odd_c, a1, al1, b1, bl1);
    }
   } else {
    // (the lane offset of step s + 1 is fetched with the fragments of step s: the table read is not on the fragment reads' critical path)
    int boff_nx = *reinterpret_cast<const int*>(tab + (ks0 * 64 + lane) * 4);
    auto load_frags = [&](int s, u32x4 (&af)[TN], u32x4 (&bf)[PF]) {
      const int boff = boff_nx;
      if (s + 1 < ks1) boff_nx = *reinterpret_cast<const int*>(tab + ((s + 1) * 64 + lane) * 4);
      const int aoff = CLS ? tab2[s] : s * 64;
#pragma unroll
      for (int nf = 0; nf < TN; ++nf) af[nf] = *reinterpret_cast<const u32x4*>(wl + abase[nf] + aoff);
#pragma unroll
      for (int i = 0; i < PF; ++i) bf[i] = *reinterpret_cast<const u32x4*>(xw + boff + i * rowpitch);
    };
    auto mma = [&](const u32x4 (&af)[TN], const u32x4 (&bf)[PF]) {
#pragma unroll
      for (int nf = 0; nf < TN; ++nf)
#pragma unroll
        for (int i = 0; i < PF; ++i) acc[nf][i] = mfma_bf16(af[nf], bf[i], acc[nf][i]);
    };
    load_frags(ks0, a0, b0);
    for (int s = ks0; s < ((UEGAN_ABL_BITS(a.abl) & 2) ? ks0 + 1 : ks1); s += 2) {
      if (s + 1 < ks1) load_frags(s + 1, a1, b1);
      mma(a0, b0);
      if (s + 1 >= ks1) break;
      if (s + 2 < ks1) load_frags(s + 2, a0, b0);
      mma(a1, b1);
    }
   }
    if (XMIR && !CLS) {
      int bq, oyq, oxq;
      tile_origin(t, bq, oyq, oxq);
      const int side = oxq == 0 ? 0 : (oxq + 16 == g.OW ? 1 : -1);      // (block-uniform)
      if (side >= 0) {
        const int* mt = reinterpret_cast<const int*>(tab + a.mtab_off) + side * a.ksteps * 64;
        const int* mflag = reinterpret_cast<const int*>(tab + a.mtab_off) + 2 * a.ksteps * 64 + side * a.ksteps;
        for (int s = 0; s < a.ksteps; ++s) {
          if (!__builtin_amdgcn_readfirstlane(mflag[s])) continue;      // no lane of this step has an image
          const int boff = mt[s * 64 + lane];
          const uint32_t m = boff >= 0 ? 0xffffffffu : 0u;
          const int bo = boff & 0x7fffffff;
#pragma unroll
          for (int nf = 0; nf < TN; ++nf) a0[nf] = *reinterpret_cast<const u32x4*>(wl + abase[nf] + s * 64);
#pragma unroll
          for (int i = 0; i < PF; ++i) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xw + (boff >= 0 ? bo : 0) + i * rowpitch);
            b0[i] = v & u32x4{m, m, m, m};
          }
#pragma unroll
          for (int nf = 0; nf < TN; ++nf)
#pragma unroll
            for (int i = 0; i < PF; ++i) acc[nf][i] = mfma_bf16(a0[nf], b0[i], acc[nf][i]);
        }
      }
    }
    // epilogue.  The MFMA result gives a lane 4 consecutive channels (4g .. 4g+3 of each 16-channel block) of pixel
    // (tile row row0+i, column fj): lane pairs (g even, g odd) swap halves so that every lane owns one whole 16-byte chunk
    // (8 channels) -- a wave then writes the 16 pixels of a tile row as one contiguous run instead of 8-byte pieces.
    // Specialised per activation: with a run-time switch per element the epilogue VALU work exceeded the MFMA time.
    int b, oy0, ox0;
    tile_origin(t, b, oy0, ox0);
    if (STATS && b != st_b) {
      if (st_b >= 0) stats_flush(st_b);
      st_b = b;
    }
    // (1 / sigma of the image's group.  A plain load here put an `s_waitcnt vmcnt(0)` into every epilogue: the NEXT tile's staging loads and the previous
    // parity class's stores had to land before the epilogue's arithmetic could start.  Fetched when the group changes, the wait stays in that branch)
    if (ca.scale) {
      const int grp = ca.scale_group ? b / ca.scale_group : 0;
      if (grp != sc_grp) {
        sc_grp = grp;
        scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ca.scale[grp])));
      }
    }
    const int osx = CLS ? 2 : 1, opy = CLS ? (cl >> 1) : 0, opx = CLS ? (cl & 1) : 0;      // output pixel = osx * position + parity
    const int ox = osx * (ox0 + fj) + opx;
    const bool odd = fg & 1;
    auto epilogue = [&](auto act_c) {
      constexpr int ACT = decltype(act_c)::value;
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int oy = osx * (oy0 + row0 + i) + opy;
        const bool pv = oy < g.OH && ox < g.OW;
        const size_t pixo = ((size_t)b * g.OH + oy) * g.OW + ox;
        uint32_t pk[TN][2];
        uint32_t pkl[EPX == 1 ? TN : 1][2], pkm[EPX == 2 ? TN : 1][2], pkml[EPX == 2 ? TN : 1][2];      // EPX: lo plane of the result, product, its lo plane
#pragma unroll
        for (int nf = 0; nf < TN; ++nf) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act_c<ACT>(acc[nf][i][r] * scale + bv[nf][r]);
          if (STATS && pv) {      // (channels beyond N: never read)
#pragma unroll
            for (int r = 0; r < 4; ++r) { st1[nf][r] += v[r]; st2[nf][r] += v[r] * v[r]; }
          }
          pk[nf][0] = pack_bf16x2(v[0], v[1]);
          pk[nf][1] = pack_bf16x2(v[2], v[3]);
          if constexpr (EPX != 0) {
            // (what the 16-bit pair just packed leaves of each fp32 value)
            auto rest = [](float f, uint32_t packed, int hi) { return f - (hi ? half_hi_to_f32(packed) : half_lo_to_f32(packed)); };
            if constexpr (EPX == 1) {
              pkl[nf][0] = pack_bf16x2(rest(v[0], pk[nf][0], 0), rest(v[1], pk[nf][0], 1));
              pkl[nf][1] = pack_bf16x2(rest(v[2], pk[nf][1], 0), rest(v[3], pk[nf][1], 1));
            }
            if constexpr (EPX == 2) {
              // the product with the multiplier fetched before the K loop, formed from the fp32 result
              const u32x2 mh = mulh[nf][i], ml = mull[nf][i];
              float q[4];
              q[0] = v[0] * (half_lo_to_f32(mh.x) + half_lo_to_f32(ml.x)); q[1] = v[1] * (half_hi_to_f32(mh.x) + half_hi_to_f32(ml.x));
              q[2] = v[2] * (half_lo_to_f32(mh.y) + half_lo_to_f32(ml.y)); q[3] = v[3] * (half_hi_to_f32(mh.y) + half_hi_to_f32(ml.y));
              pkm[nf][0] = pack_bf16x2(q[0], q[1]);
              pkm[nf][1] = pack_bf16x2(q[2], q[3]);
              if (ca.out_mul_lo) {
                pkml[nf][0] = pack_bf16x2(rest(q[0], pkm[nf][0], 0), rest(q[1], pkm[nf][0], 1));
                pkml[nf][1] = pack_bf16x2(rest(q[2], pkm[nf][1], 0), rest(q[3], pkm[nf][1], 1));
              }
            }
          }
        }
        if constexpr (EPX != 0) {
          // lane pairs swap halves (see above) and store one 16-byte chunk per destination tensor
          auto emit = [&](const uint32_t (&k)[TN][2], void* dstp) {
#pragma unroll
            for (int pr = 0; pr < (TN + 1) / 2; ++pr) {
              const int nfa = 2 * pr, nfb = 2 * pr + 1 < TN ? 2 * pr + 1 : 2 * pr;
              const uint32_t s0 = odd ? k[nfa][0] : k[nfb][0], s1 = odd ? k[nfa][1] : k[nfb][1];
              const uint32_t r0 = __shfl_xor(s0, 16, 64), r1 = __shfl_xor(s1, 16, 64);
              const int nsel = odd ? nfb : nfa;
              const u32x4 chunk = odd ? u32x4{r0, r1, k[nfb][0], k[nfb][1]} : u32x4{k[nfa][0], k[nfa][1], r0, r1};
              const int n = nsel * 16 + (fg >> 1) * 8;
              if (!pv || n >= ca.N || (TN == 1 && odd)) continue;
              *reinterpret_cast<u32x4*>(static_cast<bf16_t*>(dstp) + pixo * ca.N + n) = chunk;
            }
          };
          emit(pk, ca.out);
          if constexpr (EPX == 1) emit(pkl, ca.out_lo);
          if constexpr (EPX == 2) {
            emit(pkm, ca.out_mul);
            if (ca.out_mul_lo) emit(pkml, ca.out_mul_lo);
          }
          continue;
        }
#pragma unroll
        for (int pr = 0; pr < (TN + 1) / 2; ++pr) {
          const int nfa = 2 * pr, nfb = 2 * pr + 1 < TN ? 2 * pr + 1 : 2 * pr;     // TN = 1: the odd lanes just feed the even ones
          const uint32_t s0 = odd ? pk[nfa][0] : pk[nfb][0], s1 = odd ? pk[nfa][1] : pk[nfb][1];
          const uint32_t r0 = __shfl_xor(s0, 16, 64), r1 = __shfl_xor(s1, 16, 64);
          const int nsel = odd ? nfb : nfa;
          u32x4 chunk = odd ? u32x4{r0, r1, pk[nfb][0], pk[nfb][1]} : u32x4{pk[nfa][0], pk[nfa][1], r0, r1};
          const int n = nsel * 16 + (fg >> 1) * 8;
          if (!pv || n >= ca.N || (TN == 1 && odd) || (UEGAN_ABL_BITS(a.abl) & 4)) continue;
          if (ca.mask) {      // deferred activation gradient of the layer that produced this conv's input (one destination)
            const u32x4 mk = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(ca.mask) + pixo * ca.N + n);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const uint32_t cd = chunk[d], md = mk[d];
              chunk[d] = pack_bf16x2(half_lo_to_f32(cd) * act_grad_from_out(half_lo_to_f32(md), ca.mask_act),
                                     half_hi_to_f32(cd) * act_grad_from_out(half_hi_to_f32(md), ca.mask_act));
            }
          }
          bf16_t* p = (ca.out2 && n >= ca.n_out1) ? static_cast<bf16_t*>(ca.out2) + pixo * (ca.N - ca.n_out1) + (n - ca.n_out1)
                                                  : out + pixo * (ca.out2 ? ca.n_out1 : ca.N) + n;
          *reinterpret_cast<u32x4*>(p) = chunk;
        }
      }
    };
    switch (ca.act) {
      case UEGAN_ACT_LRELU: epilogue(std::integral_constant<int, UEGAN_ACT_LRELU>{}); break;
      case UEGAN_ACT_RELU: epilogue(std::integral_constant<int, UEGAN_ACT_RELU>{}); break;
      case UEGAN_ACT_TANH: epilogue(std::integral_constant<int, UEGAN_ACT_TANH>{}); break;
      default: epilogue(std::integral_constant<int, UEGAN_ACT_NONE>{}); break;
    }
   }      // parity classes
  }
  if (STATS && st_b >= 0) stats_flush(st_b);
}

// ---- host side ----
static bool g_use_stream = true;

struct ConvStreamPlan {
  ConvStreamArgs a;
  int tn, pf, blocks;
  int nw;                // waves per block (4 or 8)
  int lc;                // LDS class (CS_LDS_KB)
  bool fixup;            // reflection-padded data gradient: the mirrored images of the border pixels are added by dgrad_images_kernel
  bool stats = false;    // launch the STATS instantiation (conv_stream_stats_ok)
  int pr = 0;            // hi + lo pairs (the kernel's PR): from ConvArgs::w_lo / in1_lo / in2_lo
  int epx = 0;           // epilogue extras (the kernel's EPX): 1 ConvArgs::out_lo, 2 ConvArgs::mul
};
// conv_stream_ex.hip: the instantiations with hi + lo pairs / epilogue extras (false: none for this plan)
bool conv_stream_launch_ex(const ConvStreamPlan& p, hipStream_t s);
bool conv_stream_ex_available(const ConvStreamPlan& p);

#ifndef UEGAN_CONV_STREAM_KERNEL_ONLY      // (conv_stream_ex.hip wants the kernel template and the plan struct only)
static bool conv_stream_plan(const ConvArgs& c, int dtype, ConvStreamPlan& p, int max_pf = 4) {
  const ConvGeom& g = c.g;
  if (!g_use_stream || dtype != UEGAN_BF16 || g.KH != g.KW || !(g.KH & 1) || g.pad != (g.KH - 1) / 2) return false;
  // hi + lo pairs / epilogue extras (uegan_conv2d_fwd_ex): plain stride-1 forwards, weights always as a pair when anything is
  int pr = 0;
  if (c.w_lo) pr = (c.in1_lo && g.C2 == 0) ? 2 : ((c.in2_lo && !c.in1_lo && g.C1 == 32 && g.C2 == 32) ? 3 : ((c.in1_lo || c.in2_lo) ? -1 : 1));
  else if (c.in1_lo || c.in2_lo) pr = -1;
  const int epx = c.mul ? 2 : (c.out_lo ? 1 : 0);
  if (c.mul && c.out_lo) return false;                                      // (no instantiation writes both)
  if (pr >= 2 && !epx) return false;                                        // (the source-pair instantiations all carry an epilogue extra)
  if (pr < 0 || ((pr || epx) && (g.mode != 0 || g.stride != 1 || c.mask || c.out2))) return false;
  if (pr >= 2 && g.pad_mode != UEGAN_PAD_REFLECT && g.pad != 0) return false;      // (the zero-filling staging path knows no lo plane)
  if (c.mul && !c.out_mul) return false;
  p.pr = pr; p.epx = epx;
  const int sx = g.stride;
  const bool cls = sx == 2 && g.mode == 1;      // data gradient of a stride-2 conv: four parity classes per tile
  if (cls) {
    if (!(g.C == 32 || g.C == 64) || g.OH != 2 * g.IH || g.OW != 2 * g.IW || g.pad_mode != UEGAN_PAD_REFLECT || g.C2 != 0) return false;
  } else if (sx == 2) {
    if (g.mode != 0 || g.OH != (g.IH + 2 * g.pad - g.KH) / 2 + 1 || g.OW != (g.IW + 2 * g.pad - g.KW) / 2 + 1) return false;
  } else if (sx != 1 || g.IH != g.OH || g.IW != g.OW) {
    return false;
  }
  if (!(g.C == 8 || g.C == 16 || g.C == 32 || g.C == 64) || c.N > 64 || c.N % 8 || (c.out2 && c.n_out1 % 8)) return false;
  if (g.C1 % 8 || g.C2 % 8) return false;
  if ((cls ? g.IH : g.OH) < 16 || (cls ? g.IW : g.OW) < 32) return false;
  ConvStreamArgs& a = p.a;
  a.c = c;
  a.abl = UEGAN_ABL_BITS(g_abl_stream);
  a.sx = cls ? 1 : sx;
  a.cls = cls ? 1 : 0;
  a.flip = g.mode == 1;
  a.org = g.mode == 1 ? g.pad - (g.KH - 1) : -g.pad;
  a.zero_fill = (g.mode == 1 || g.pad_mode != UEGAN_PAD_REFLECT) ? 1 : 0;
  a.taps = g.KH * g.KW;
  a.Clog = g.C == 8 ? 3 : (g.C == 16 ? 4 : (g.C == 32 ? 5 : 6));
  a.rb = g.C * 2;
  a.rblog = a.Clog + 1;
  a.ksteps = (a.taps * g.C + 31) / 32;
  a.KWmagic = 65536 / g.KW + 1;
  for (int tap = 0; tap < a.taps; ++tap)
    if (((tap * a.KWmagic) >> 16) != tap / g.KW) return false;
  a.wrow = a.ksteps * 64;
  // rows 64 B apart modulo the 256-byte bank row: conflict-free A reads (192 = -64 serves as well -- slot 12 n mod 16 is the same permutation of n & 3 --
  // and is what lets dec4's weight PAIR fit the LDS beside its patches; the plain launches keep the layout they were measured with)
  while (a.wrow % 256 != 64 && !(pr && a.wrow % 256 == 192)) a.wrow += 64;
  p.tn = c.N <= 16 ? 1 : (c.N <= 32 ? 2 : 4);
  a.wrows = (c.N + 7) / 8 * 8;
  if (a.wrows > p.tn * 16) a.wrows = p.tn * 16;
  a.wbytes = (a.wrows * a.wrow + 15) / 16 * 16;
  a.wlo_off = a.wbytes;
  if (pr) a.wbytes *= 2;                                             // (the lo part of the weights behind the hi matrix)
  a.tbytes = a.ksteps * 256 + (a.ksteps * 4 + 255) / 256 * 256;      // lane offsets + weight-slice offsets per K step
  // reflection-padded stride-1 data gradient on a map whose width is whole tiles: x-mirrored images inside the kernel (two more tables)
  a.xmir = (!cls && g.mode == 1 && g.pad_mode == UEGAN_PAD_REFLECT && g.pad > 0 && g.pad < 8 && g.OW % 16 == 0 && g.KW == 2 * g.pad + 1) ? 1 : 0;
  a.mtab_off = a.tbytes;
  if (a.xmir) a.tbytes += 2 * a.ksteps * 256 + (2 * a.ksteps * 4 + 255) / 256 * 256;
  if (pr && (a.xmir || cls)) return false;
  a.tlo_off = a.tbytes;
  if (pr == 3) a.tbytes += a.taps * 256;                             // the lo plane's per-tap lane offsets
  a.c_lo = pr == 2 ? g.C : (pr == 3 ? g.C2 : 0);
  a.rb_lo = a.c_lo * 2;
  a.rblog_lo = a.c_lo == 8 ? 4 : (a.c_lo == 16 ? 5 : (a.c_lo == 32 ? 6 : 7));
  a.lo_xoff = 0;
  a.PW = sx * 15 + g.KW;
  a.ymin = 0;
  int cspan = 0;
  for (int c = 0; c < 5; ++c) a.kstart[c] = 0;
  if (cls) {
    // source = i + (py + pad - t) / 2 over the taps t of class py: from (py - pad)/2 (t = K-1) up to (py + pad - t0)/2
    const int ymin = (g.pad & 1) ? (1 - g.pad) / 2 : -(g.pad / 2);
    int ymax = 0;
    for (int py = 0; py < 2; ++py) {
      const int t0 = (py + g.pad) & 1, v = (py + g.pad - t0) / 2;
      if (v > ymax) ymax = v;
    }
    a.ymin = ymin;
    cspan = ymax - ymin;
    a.PW = 16 + cspan;
    const int spt = g.C / 32;
    for (int c = 0; c < 4; ++c) {
      const int py = c >> 1, px = c & 1;
      const int nty = (g.KH - ((py + g.pad) & 1) + 1) / 2, ntx = (g.KW - ((px + g.pad) & 1) + 1) / 2;
      a.kstart[c + 1] = a.kstart[c] + nty * ntx * spt;
    }
    if (a.kstart[4] != a.ksteps) return false;
  }
  a.PWmagic = 65536 / a.PW + 1;
  p.pf = 0;
  p.lc = 1;
  for (int pass = 1; pass < (pr == 3 ? 4 : 3) && !p.pf; ++pass) {      // 80 KB (two blocks per CU) if it fits, else 152 KB (PR 3: else all 160)
    const int kb = CS_LDS_KB[pass], maxix = pass >= 2 ? 16 : 10;
    for (int pf : {4, 2}) {
      if (pf > max_pf) continue;
      const int th = 4 * pf, ph = cls ? th + cspan : sx * (th - 1) + g.KH;
      // (PR >= 2: both planes packed exactly, lanes without a chunk masked; else whole staging rounds: every lane of a round writes)
      const int xbh = pr >= 2 ? ph * a.PW * a.rb : (ph * a.PW * a.rb + 4095) / 4096 * 4096;
      const int xb = xbh + (pr >= 2 ? ph * a.PW * a.rb_lo : 0);
      if (a.wbytes + a.tbytes + 2 * xb > kb * 1024 || (xb + 4095) / 4096 > maxix) continue;
      bool ok = true;
      for (int r = 0; r < ph * a.PW && ok; ++r) ok = ((r * a.PWmagic) >> 16) == r / a.PW;
      if (!ok) continue;
      p.pf = pf; a.TH = th; a.PH = ph; a.xbytes = xb; p.lc = pass;
      if (pr >= 2) a.lo_xoff = xbh;
      break;
    }
  }
  if (!p.pf) return false;
  p.nw = 4;
  if (p.pf == 4 && (!cls || p.lc == 2) && p.lc == 2 && p.tn <= 2 && !a.xmir && !pr) {
    // the same 16-row tile on 8 waves of 2 rows each (staging rounds of 512 lanes)
    const int xb8 = (a.PH * a.PW * a.rb + 8191) / 8192 * 8192;
    if (a.wbytes + a.tbytes + 2 * xb8 <= CS_LDS_KB[p.lc] * 1024 && xb8 / 8192 <= (p.lc == 2 ? 8 : 5)) { p.nw = 8; p.pf = 2; a.xbytes = xb8; }
  }
  // pairs on one block per CU: 8 waves as well (the same tile, half the rows per wave; the planes are packed exactly, so only the round count changes):
  // dec5.0 0.96 -> 0.74 ms, dec4 2.09 -> 1.51 ms per 32 images
  if (pr >= 2 && p.lc >= 2 && p.tn <= 2 && (a.xbytes + 8191) / 8192 <= 8) { p.nw = 8; p.pf /= 2; }
  // one block per CU only pays for the thin layers: with 64 output channels (VGG conv1_2) or four parity classes per tile the
  // patch kernel measured faster
  if (p.lc == 2 && ((p.tn == 4 && sx == 1) || (cls && p.nw != 8))) return false;
  // Every tile of the map.  The data gradient of a reflection-padded conv is computed as if the padding were zeros (the direct
  // image of every pixel); the few pixels within `pad` of a border that also receive MIRRORED images get those added afterwards
  // by dgrad_images_kernel (conv.hip) -- 0.8 % of a 512^2 map for pad 1, instead of a second MFMA launch over every border tile.
  p.fixup = g.mode == 1 && g.pad_mode == UEGAN_PAD_REFLECT && g.pad > 0;
  if (p.fixup && (g.OH <= 2 * g.pad + 2 || g.OW <= 2 * g.pad + 2)) return false;
  a.ty0 = 0; a.tx0 = 0;
  a.ty1 = ((cls ? g.IH : g.OH) + a.TH - 1) / a.TH;
  a.tx1 = ((cls ? g.IW : g.OW) + 15) / 16;
  a.tiles_total = g.B * (a.ty1 - a.ty0) * (a.tx1 - a.tx0);
  const int maxb = p.lc == 2 ? 256 : 512;
  int blocks = a.tiles_total < maxb ? a.tiles_total : maxb;
  a.tiles_per_block = (a.tiles_total + blocks - 1) / blocks;
  p.blocks = (a.tiles_total + a.tiles_per_block - 1) / a.tiles_per_block;
  if ((pr || epx) && !conv_stream_ex_available(p)) return false;      // (a handful of instantiations: the generator's full-resolution layers)
  return true;
}

template <int TN, int PF>
static void conv_stream_launch2(const ConvStreamPlan& p, hipStream_t s) {
  const int blocks = p.blocks;
  if (p.a.cls) {        // parity-class data gradient: own instantiation, so the plain kernel keeps its straight-line K loop
    if constexpr (PF == 2 && TN <= 2) {
      if (p.nw == 8) {
        hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 2, true, false, 8>), dim3(blocks), dim3(512), 0, s, p.a);
        return;
      }
    }
    if (p.lc == 2) hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 2, true>), dim3(blocks), dim3(256), 0, s, p.a);
    else hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 1, true>), dim3(blocks), dim3(256), 0, s, p.a);
    return;
  }
  if constexpr (PF == 2 && TN <= 2) {
    if (p.nw == 8) {
      if (p.a.xmir) hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 1, false, true, 8>), dim3(blocks), dim3(512), 0, s, p.a);
      else if (p.lc == 2) hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 2, false, false, 8>), dim3(blocks), dim3(512), 0, s, p.a);
      else hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 1, false, false, 8>), dim3(blocks), dim3(512), 0, s, p.a);
      return;
    }
  }
  if (p.stats) {        // (forward with per-channel sums; conv_stream_stats_ok: 4 waves, 80-KB class, no mirrors, >= 32 output channels)
    if constexpr (TN >= 2) {
      hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 1, false, false, 4, true>), dim3(blocks), dim3(256), 0, s, p.a);
      return;
    }
  }
  if (p.a.xmir) {       // (own instantiation: the forward kernels keep their register budget)
    if (p.lc == 2) hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 2, false, true>), dim3(blocks), dim3(256), 0, s, p.a);
    else hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 1, false, true>), dim3(blocks), dim3(256), 0, s, p.a);
    return;
  }
  if (p.lc == 2) hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 2, false>), dim3(blocks), dim3(256), 0, s, p.a);
  else hipLaunchKernelGGL((conv_stream_kernel<TN, PF, 1, false>), dim3(blocks), dim3(256), 0, s, p.a);
}
// can this planned launch carry the per-channel sums?  (a block's tile range must span at most two images)
static bool conv_stream_stats_ok(const ConvStreamPlan& p) {
  const ConvGeom& g = p.a.c.g;
  const int tpi = (p.a.ty1 - p.a.ty0) * (p.a.tx1 - p.a.tx0);
  return g.mode == 0 && !p.a.cls && !p.a.xmir && p.nw == 4 && p.lc == 1 && !p.a.c.out2 && !p.a.c.mask && p.a.tiles_per_block <= tpi && p.tn >= 2;
}
static void conv_stream_launch(const ConvStreamPlan& p, hipStream_t s) {
  if (p.pr || p.epx) {
    (void)conv_stream_launch_ex(p, s);      // (the planner's caller checked conv_stream_ex_available)
    return;
  }
  if (p.tn == 1 && p.pf == 4) conv_stream_launch2<1, 4>(p, s);
  else if (p.tn == 1) conv_stream_launch2<1, 2>(p, s);
  else if (p.tn == 2 && p.pf == 4) conv_stream_launch2<2, 4>(p, s);
  else if (p.tn == 2) conv_stream_launch2<2, 2>(p, s);
  else if (p.pf == 4) conv_stream_launch2<4, 4>(p, s);
  else conv_stream_launch2<4, 2>(p, s);
}
#endif
