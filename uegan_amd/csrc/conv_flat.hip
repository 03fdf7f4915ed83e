// conv_flat_kernel: data gradient of a STRIDE-2 convolution (G.enc3 - enc5, D.d3 - d5) over the PADDED input grid, all four parity classes in
// one launch (round 5).
//
// The gradient with respect to the padded input of a stride-2 K x K convolution splits into four parity classes (ry, rx) of padded pixels
// i = 2c + r: with k = 2a + r,   dxp[2cy + ry][2cx + rx] = sum_{ay < A_ry, ax < A_rx, co} w[co][2ay + ry][2ax + rx] * dz[cy - ay][cx - ax],
// A_0 = (K + 1) / 2 =: A, A_1 = K / 2 -- a stride-1 correlation of dz on the class grid CH x CW = (Ho + A - 1) x (Wo + A - 1) with 16 / 12 / 12 / 9
// (7 x 7), 9 / 6 / 6 / 4 (5 x 5) or 4 / 2 / 2 / 1 (3 x 3) taps.  Round 4 ran one conv_patch_kernel launch per class on 16-pixel-wide tiles: the class
// grids of the deep layers are 18 / 34 / 67 positions wide -- "just over" the tile quantum, 1.7 - 2.5 x wasted tiles -- and each launch staged
// the same dz patch again (D.d4 / d5 at 200 - 320 TFLOP/s).
//
// Here the positions of ALL images are FLATTENED: m = (b * CH + cy) * PP + cx with the row pitch PP = CW + A - 1, so that a tap (ay, ax) is the
// uniform shift (A-1-ay) * PP + (A-1-ax) into a virtual zero-padded dz with the same pitch (out-of-range sources are zero pages; the rows
// an image's bottom taps reach are the zero rows above the next image).  A tile is 256 consecutive m -- no 2-D quantisation, the only waste is the
// A - 1 overhang columns per row (2 of 20 on D.d5) -- and its patch is the 256 + (A-1)(PP+1) virtual positions behind it, staged ONCE per
// 32-channel chunk of dz.  Block = (tile, parity class, 128- or 64-channel block of the input channels): conv_tall_kernel's one-slice-per-
// block structure (four waves share the weight slice [BN][32 channels] of a step, each wave owns 2 x 32 positions; ring of four slices,
// double-buffered patch, 32x32x16 MFMAs, two blocks per CU), with the tap list of the block's class driving the steps: per class a table
// of (pixel shift, weight column) per tap.  The blocks of one tile (4 classes x N / BN) run back to back on one XCD and share the patch in its
// L2.  A class with fewer taps than the patch needs steps to arrive (3 x 3: 1 - 2 taps) pads its chunk with steps on a slice of zeros (a
// branch around the MFMAs of such a step cost 40 - 90 registers: spills in the 128-channel variant).
// The result goes to the padded grid [B][H + 2p][W + 2p][N] (pixel-shuffled store); fold_reflect_kernel (conv.hip) adds the mirror images of
// the reflection padding.  Weights: the ordinary IHWO pack [ci][(ky * K + kx) * Cout + co], read per tap -- no second pack.
#include "conv_core.h"

#include <type_traits>

namespace uegan {

static __device__ __attribute__((aligned(128))) const unsigned int g_zero_page_flat[32] = {0u};

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32_flat(u32x4 a, u32x4 b, f32x16 c) {
#ifdef UEGAN_HALF_FP16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}

#define UEGAN_SB() __builtin_amdgcn_sched_barrier(0)

constexpr int FLAT_NT_MAX = 16, FLAT_TM = 256;

struct FlatArgs {
  const void* dz;          // [B][Ho][Wo][C]
  const void* w;           // IHWO [N][Kp]
  void* out;               // [B][OHp][OWp][N]: the padded grid
  const float* scale;      // device scalar(s) or null: image b is multiplied by scale[scale_group ? b / scale_group : 0]
  int scale_group;
  int B, Ho, Wo, C;
  int N, Kp;
  int A;                   // (K + 1) / 2
  int PP, CH, CW;          // flattened row pitch, class grid rows / columns
  int OHp, OWp;
  unsigned mPP, mCH;       // ceil(2^32 / PP), ceil(2^32 / CH): x / d = umulhi(x, m) for the x < 2^21 that occur (launcher checks)
  int npg;                 // 1-KB pieces (16 positions x 64 B) of one patch: ceil((256 + (A-1)(PP+1)) / 16)
  int nbc;                 // channel blocks per class = N / BN
  int ntiles;              // tiles of 256 positions (the grid is padded to a multiple of 8 for the XCD mapping)
  int xcd_map;
  int mode2d;              // 1: tiles are 8 rows x 32 columns of ONE image's class grid (maps too wide for the flattened patch: D.d2, and the 1024^2 configuration)
  int PWt, nty2, ntx2;     // 2-D tiles: patch pitch 32 + A - 1; tiles per image (rows, columns)
  unsigned mPWt, mntx2, mnty2;
  int pair;                // 1: 32 input channels -- a block takes BOTH column classes of its row class as its two channel fragments (they are neighbouring
                           // output pixels: 64 contiguous channels in memory); classes are then ry only, wcol[ry] / wcol[2 + ry] the two fragments' columns
  int nt[4], nlive[4];     // per class ry * 2 + rx: steps per 32-channel chunk (>= what the patch needs to arrive), of which the first nlive carry a tap
  int shift[4][FLAT_NT_MAX];      // per class and step: patch row offset of the tap
  int wcol[4][FLAT_NT_MAX];       // ... and its column (elements) in the IHWO row; dead step: -1
};

__device__ __forceinline__ unsigned fdiv(unsigned x, unsigned m) { return (unsigned)(((unsigned long long)x * m) >> 32); }

template <int NI>
__global__ void __launch_bounds__(256, 2) conv_flat_kernel(FlatArgs a) {
  static_assert(NI == 2 || NI == 4, "channel fragments per block");
  constexpr int RPW = 2, BN = NI * 32, NWAVES = 4;
  constexpr int NPG = NI == 4 ? 22 : 30;               // 1-KB pieces of one patch buffer (the launcher checks a.npg <= NPG)
  constexpr int NI_P = (NPG + NWAVES - 1) / NWAVES;    // pieces per wave
  constexpr int PPS = 3;                               // patch pieces requested per wave per step
  constexpr int PBUFB = NPG * 1024, WSL = BN * 64, NWP = WSL / 1024 / NWAVES;
  constexpr int DUMPB = 1024;
  constexpr int NLOAD = NWP + PPS;                     // direct-to-LDS loads per wave per step (nothing to fetch: the dump area)
  constexpr int EROW = BN * 2 + 8;                     // epilogue staging row: one position's BN channels + 8 B (bank spread)
  constexpr int WPIX = RPW * 32;                       // positions per wave
  constexpr int MAINB = 2 * PBUFB + 4 * WSL, EPIB = NWAVES * WPIX * EROW;
  constexpr int BODYB = MAINB > EPIB ? MAINB : EPIB;
  constexpr int TABB = 2 * FLAT_NT_MAX * 4;
  static_assert(2 * (BODYB + DUMPB + TABB) <= 160 * 1024, "LDS budget (two blocks per CU)");
  static_assert(NWP >= 1, "at least one weight piece per wave per slice");

  __shared__ __attribute__((aligned(16))) unsigned char lds[BODYB + DUMPB + TABB];
  unsigned char* const lds_w = lds + 2 * PBUFB;
  unsigned char* const lds_dump = lds + BODYB;
  int* const lds_wcol = reinterpret_cast<int*>(lds_dump + DUMPB);      // this class's weight column per step (read by the staging cursor, which runs 3 steps ahead)

  const bf16_t* dz = static_cast<const bf16_t*>(a.dz);
  const bf16_t* w = static_cast<const bf16_t*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  int by = blockIdx.y, t = blockIdx.x;
  if (a.xcd_map) {      // the (class, channel block) blocks of ONE tile back to back on one XCD: all but the first find the patch in that L2
    const int L = blockIdx.x + gridDim.x * blockIdx.y, nby = gridDim.y;
    const int xcd = L & 7, grp = L >> 3;
    by = grp % nby;
    t = (grp / nby) * 8 + xcd;
  }
  if (t >= a.ntiles) return;      // (block-uniform: the grid's padding)
  const int cls = by / a.nbc, n0 = a.pair ? 0 : (by - cls * a.nbc) * BN;
  const int ry = a.pair ? cls : cls >> 1, rx = a.pair ? 0 : cls & 1;
  const int m0 = t * FLAT_TM;
  // 2-D tiles: tile t = (image, tile row, tile column) of the class grid
  int b2 = 0, cy0 = 0, cx0 = 0;
  if (a.mode2d) {
    const unsigned q1 = fdiv((unsigned)t, a.mntx2), txi = (unsigned)t - q1 * a.ntx2;
    const unsigned q2 = fdiv(q1, a.mnty2), tyi = q1 - q2 * a.nty2;
    b2 = (int)q2; cy0 = (int)tyi * 8; cx0 = (int)txi * 32;
  }
  const int fpitch = a.mode2d ? a.PWt : 32;            // patch rows between the wave's position fragments
  const int nchunk = a.C / 32;
  const int nt = a.nt[cls];
  const int nsteps = nchunk * nt;
  // this class's tap table: pixel shifts in scalar registers (compile-time step index below), weight columns in the LDS
  int tshift[FLAT_NT_MAX];
#pragma unroll
  for (int i = 0; i < FLAT_NT_MAX; ++i) tshift[i] = a.shift[cls][i];
  if (tid < FLAT_NT_MAX) lds_wcol[tid] = a.wcol[cls][tid];
  if (a.pair && tid >= FLAT_NT_MAX && tid < 2 * FLAT_NT_MAX) lds_wcol[tid] = a.wcol[2 + cls][tid - FLAT_NT_MAX];

  // ---- patch staging role: piece rg = ii*4 + wave covers patch rows 16*rg .. 16*rg+15; lane -> (row lane>>2, LDS position lane&3)
  const int q_src = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;      // source channel offset inside the chunk = (position ^ ((row>>2)&3)) * 8
  int ppix[NI_P];                                              // my source pixel of piece ii (linear over [B][Ho][Wo]); -1: zeros
#pragma unroll
  for (int ii = 0; ii < NI_P; ++ii) {
    const int rg = ii * NWAVES + wave;
    int pix = -1;
    if (!a.mode2d) {
      const unsigned v = (unsigned)(m0 + rg * 16 + (lane >> 2));      // virtual position (row R = b * CH + vy of pitch PP, column vx)
      const unsigned R = fdiv(v, a.mPP), vx = v - R * a.PP;
      const unsigned b = fdiv(R, a.mCH), vy = R - b * a.CH;
      const int sy = (int)vy - (a.A - 1), sx = (int)vx - (a.A - 1);
      if (rg < a.npg && (int)b < a.B && sy >= 0 && sx >= 0 && sx < a.Wo) pix = ((int)b * a.Ho + sy) * a.Wo + sx;      // (sy < Ho: vy < CH = Ho + A - 1)
    } else {
      const unsigned pr = (unsigned)(rg * 16 + (lane >> 2));          // patch row = (patch row piy, patch column pix) of pitch PWt
      const unsigned piy = fdiv(pr, a.mPWt), pxx = pr - piy * a.PWt;
      const int sy = cy0 + (int)piy - (a.A - 1), sx = cx0 + (int)pxx - (a.A - 1);
      if (rg < a.npg && sy >= 0 && sy < a.Ho && sx >= 0 && sx < a.Wo) pix = (b2 * a.Ho + sy) * a.Wo + sx;
    }
    ppix[ii] = pix;
  }
  const unsigned char* const zero16 = reinterpret_cast<const unsigned char*>(g_zero_page_flat) + (lane & 3) * 16;
  const unsigned char* p_src = zero16;
  unsigned char* p_dst = lds_dump;
  // piece `pidx` (compile-time) of `chunk`'s patch; no such piece: a load into the dump area
  auto patch_piece_prepare = [&](int pidx, int chunk, bool live) {
    const int rg = pidx * NWAVES + wave;
    live = live && pidx < NI_P && rg < a.npg;
    const int pix = ppix[pidx < NI_P ? pidx : 0];
    p_src = (live && pix >= 0) ? reinterpret_cast<const unsigned char*>(dz + ((size_t)pix * a.C + chunk * 32 + q_src)) : zero16;
    p_dst = live ? lds + (chunk & 1) * PBUFB + rg * 1024 : lds_dump;
  };
  auto patch_piece_issue = [&]() { glds16(p_src, p_dst); };
  // ---- weight staging role: a 1-KB piece is 16 rows x 64 B, lane -> (row lane>>2, position lane&3); NWP pieces per wave per slice
  // (pair: rows 0 - 31 of the slice are the 32 channels of column class 0, rows 32 - 63 those of class 1 -- waves 0, 1 / 2, 3)
  const bf16_t* const wlane = w + (size_t)(a.pair ? (wave & 1) * 16 + (lane >> 2) : n0 + wave * 16 + (lane >> 2)) * a.Kp + q_src;
  const int wtab = a.pair ? (wave >> 1) * FLAT_NT_MAX : 0;
  const int wrow64b = 64 * a.Kp * (int)sizeof(bf16_t);
  int s_chunk = 0, s_tap = 0, s_idx = 0;               // staging cursor: the next slice to request
  const unsigned char* wsrc_cur = zero16;
  unsigned char* wdst_cur = lds_dump;
  int wdst_stride = 0;
  int wsrc_stride = 0;                                 // bytes between the 64-row pieces of a slice (< 2^31: checked by the launcher)
  int wc_pre = 0;                                      // this class's column of the staging cursor's tap, read one step ahead (the wait for it is then never a wait for the fragment reads around it)
  auto stage_w_prepare = [&]() {
    const int wc = __builtin_amdgcn_readfirstlane(wc_pre);
    const bool slot = s_idx < nsteps;                  // past the end: the dump area
    const bool real = slot && wc >= 0;                 // a step without a tap (classes padded to the steps the patch needs to arrive): a slice of zeros
    wsrc_cur = real ? reinterpret_cast<const unsigned char*>(wlane + (wc + s_chunk * 32)) : zero16;
    wsrc_stride = real ? wrow64b : 0;
    wdst_cur = slot ? lds_w + (s_idx & 3) * WSL + wave * 1024 : lds_dump;
    wdst_stride = slot ? 4096 : 0;
    ++s_idx;
    ++s_tap;
    const int wrap = s_tap == nt ? 1 : 0;
    s_tap = wrap ? 0 : s_tap;
    s_chunk = (s_chunk + wrap < nchunk) ? s_chunk + wrap : nchunk - 1;
    wc_pre = lds_wcol[wtab + s_tap];
  };
  auto stage_w_piece = [&](int i) { glds16(wsrc_cur + i * wsrc_stride, wdst_cur + i * wdst_stride); };

  // ---- fragment addresses
  int wad[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) wad[i] = (i * 32 + l31) * 64 + ((lh ^ ((l31 >> 2) & 3)) << 4);
  const int xbase = wave * RPW * fpitch + l31;          // patch row of my position in my first fragment, shift 0
  int xad[RPW];
  auto set_xad = [&](int shift, int j) {
    const int pr = xbase + j * fpitch + shift;
    xad[j] = pr * 64 + ((lh ^ ((pr >> 2) & 3)) << 4);
  };
  u32x4 wf0[NI], xf0[RPW], wf1[NI], xf1[RPW];
  f32x16 acc[NI][RPW];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  __syncthreads();                                     // the weight-column table is visible to the staging cursor
  wc_pre = lds_wcol[wtab];
  // ---- prologue: patch of chunk 0, slices 0, 1, 2
#pragma unroll
  for (int ii = 0; ii < NI_P; ++ii) { patch_piece_prepare(ii, 0, true); patch_piece_issue(); }
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    stage_w_prepare();
#pragma unroll
    for (int i = 0; i < NWP; ++i) stage_w_piece(i);
  }
  wait_vmcnt<NWP>();                                   // all but slice 2
  raw_barrier();                                       // barrier 0: patch 0 and slices 0, 1 are visible
#pragma unroll
  for (int j = 0; j < RPW; ++j) set_xad(tshift[0], j);
#pragma unroll
  for (int i = 0; i < NI; ++i) wf0[i] = *reinterpret_cast<const u32x4*>(lds_w + wad[i]);
#pragma unroll
  for (int j = 0; j < RPW; ++j) xf0[j] = *reinterpret_cast<const u32x4*>(lds + xad[j]);

#define MF(W, X, i, j) acc[i][j] = mfma32_flat(W[i], X[j], acc[i][j]); UEGAN_SB();
#define LDW(F, slot, ksub, i) F[i] = *reinterpret_cast<const u32x4*>((slot) + (wad[i] ^ ((ksub) << 5)));
#define LDX(F, pb, ksub, j) F[j] = *reinterpret_cast<const u32x4*>((pb) + (xad[j] ^ ((ksub) << 5)));
#define PIECE(k) patch_piece_prepare(PPS * tap + (k), chunk + 1, more); patch_piece_issue();
  int step = 0, chunk = 0;
  const unsigned char* pcur = lds;
  bool more = false;
  // one step; `tap` is a compile-time constant (the tap loop is expanded below: a `#pragma unroll` loop with a run-time exit is not unrolled by
  // the compiler, and the per-tap tables and patch-piece indices must be register indices)
  auto step_body = [&](auto tapc) __attribute__((always_inline)) {
    constexpr int tap = decltype(tapc)::value;
    const bool last = tap + 1 == nt;
    const unsigned char* ws = lds_w + (step & 3) * WSL;
    const unsigned char* wnext = lds_w + ((step + 1) & 3) * WSL;
    const unsigned char* pnext = last ? lds + ((chunk + 1) & 1) * PBUFB : pcur;
    const int nshift = last ? tshift[0] : tshift[tap + 1 < FLAT_NT_MAX ? tap + 1 : 0];
    UEGAN_SB();
    if constexpr (NI == 4) {
      // 8 + 8 MFMAs (four channel fragments x two position fragments); the partner wave on this SIMD (the CU's other block) fills what this one leaves
      MF(wf0, xf0, 0, 0) LDW(wf1, ws, 1, 0) LDW(wf1, ws, 1, 1) UEGAN_SB();
      MF(wf0, xf0, 0, 1) LDW(wf1, ws, 1, 2) LDW(wf1, ws, 1, 3) UEGAN_SB();
      MF(wf0, xf0, 1, 0) LDX(xf1, pcur, 1, 0) LDX(xf1, pcur, 1, 1) UEGAN_SB();
      MF(wf0, xf0, 1, 1) stage_w_prepare(); UEGAN_SB();
      MF(wf0, xf0, 2, 0) stage_w_piece(0); UEGAN_SB();
      MF(wf0, xf0, 2, 1) stage_w_piece(1); UEGAN_SB();
      MF(wf0, xf0, 3, 0) PIECE(0) UEGAN_SB();
      MF(wf0, xf0, 3, 1) PIECE(1) UEGAN_SB();
      MF(wf1, xf1, 0, 0) PIECE(2) UEGAN_SB();
      MF(wf1, xf1, 0, 1) set_xad(nshift, 0); set_xad(nshift, 1); UEGAN_SB();
      MF(wf1, xf1, 1, 0) LDW(wf0, wnext, 0, 0) LDW(wf0, wnext, 0, 1) UEGAN_SB();
      MF(wf1, xf1, 1, 1) LDW(wf0, wnext, 0, 2) LDW(wf0, wnext, 0, 3) UEGAN_SB();
      MF(wf1, xf1, 2, 0) LDX(xf0, pnext, 0, 0) LDX(xf0, pnext, 0, 1) UEGAN_SB();
      MF(wf1, xf1, 2, 1) MF(wf1, xf1, 3, 0) MF(wf1, xf1, 3, 1)
    } else {
      // 4 + 4 MFMAs (two channel fragments x two position fragments)
      MF(wf0, xf0, 0, 0) LDW(wf1, ws, 1, 0) LDW(wf1, ws, 1, 1) UEGAN_SB();
      MF(wf0, xf0, 0, 1) LDX(xf1, pcur, 1, 0) LDX(xf1, pcur, 1, 1) UEGAN_SB();
      MF(wf0, xf0, 1, 0) stage_w_prepare(); stage_w_piece(0); UEGAN_SB();
      MF(wf0, xf0, 1, 1) PIECE(0) PIECE(1) PIECE(2) UEGAN_SB();
      MF(wf1, xf1, 0, 0) set_xad(nshift, 0); set_xad(nshift, 1); UEGAN_SB();
      MF(wf1, xf1, 0, 1) LDW(wf0, wnext, 0, 0) LDW(wf0, wnext, 0, 1) UEGAN_SB();
      MF(wf1, xf1, 1, 0) LDX(xf0, pnext, 0, 0) LDX(xf0, pnext, 0, 1) UEGAN_SB();
      MF(wf1, xf1, 1, 1)
    }
    // everything older than this step's batch has landed: slice step+2, older patch pieces
    wait_vmcnt<NLOAD>();
    raw_barrier();
    ++step;
  };
#define STEP(k) if (k < nt) step_body(std::integral_constant<int, k>{});
  for (chunk = 0; chunk < nchunk; ++chunk) {
    pcur = lds + (chunk & 1) * PBUFB;
    more = chunk + 1 < nchunk;
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
    STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
  }
#undef STEP
#undef MF
#undef LDW
#undef LDX
#undef PIECE
  wait_vmcnt<0>();                                     // (the dump-area loads of the last steps)

  // ---- epilogue: (scale) fp32 -> 16-bit -> through the LDS (wave-private rows of BN channels + 8 B) -> the padded grid, 16 bytes per lane,
  // BN / 8 lanes per position
  unsigned char* const est = lds + wave * (WPIX * EROW);
  {
    float sc[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) sc[j] = 1.f;
    if (a.scale) {
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        const unsigned m = (unsigned)(m0 + wave * WPIX + j * 32 + l31);
        const unsigned b = a.mode2d ? (unsigned)b2 : fdiv(fdiv(m, a.mPP), a.mCH);
        sc[j] = (int)b < a.B ? a.scale[a.scale_group ? (int)b / a.scale_group : 0] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        unsigned char* row = est + (j * 32 + l31) * EROW + i * 64 + 8 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x2 pk;
          pk.x = pack_bf16x2(acc[i][j][4 * q] * sc[j], acc[i][j][4 * q + 1] * sc[j]);
          pk.y = pack_bf16x2(acc[i][j][4 * q + 2] * sc[j], acc[i][j][4 * q + 3] * sc[j]);
          *reinterpret_cast<u32x2*>(row + q * 16) = pk;
        }
      }
  }
  __builtin_amdgcn_wave_barrier();                     // (each wave reads back only what it wrote)
  {
    constexpr int LPP = BN / 8;                        // lanes per position (16 bytes each)
    constexpr int PPI = 64 / LPP;                      // positions per store instruction
    bf16_t* out = static_cast<bf16_t*>(a.out);
    const int lc = lane % LPP, nl = n0 + lc * 8;
#pragma unroll 4
    for (int it = 0; it < WPIX / PPI; ++it) {
      const int rr = it * PPI + lane / LPP;            // position inside the wave's 64
      const u32x2 v01 = *reinterpret_cast<const u32x2*>(est + rr * EROW + lc * 16);        // (rows are 8-byte aligned only)
      const u32x2 v23 = *reinterpret_cast<const u32x2*>(est + rr * EROW + lc * 16 + 8);
      const u32x4 v = {v01.x, v01.y, v23.x, v23.y};
      unsigned b, cy, cx;
      if (!a.mode2d) {
        const unsigned m = (unsigned)(m0 + wave * WPIX + rr);
        const unsigned R = fdiv(m, a.mPP);
        cx = m - R * a.PP;
        b = fdiv(R, a.mCH); cy = R - b * a.CH;
      } else {
        b = (unsigned)b2; cy = (unsigned)(cy0 + wave * RPW + (rr >> 5)); cx = (unsigned)(cx0 + (rr & 31));
        if ((int)cy >= a.CH) continue;
      }
      if ((int)cx >= a.CW || (int)b >= a.B) continue;  // the A - 1 overhang columns of a row / tile overhang; past the last image
      const size_t pix = ((size_t)b * a.OHp + (2 * cy + ry)) * a.OWp + (2 * cx + rx);
      *reinterpret_cast<u32x4*>(out + pix * a.N + nl) = v;
    }
  }
}

// 0 / error code when the launch was taken, 1 when the problem is not one of this kernel's.  `out` = the padded-grid workspace of
// uegan_conv2d_dgrad_ws ([B][H + 2 pad][W + 2 pad][C1]); the caller folds the mirror images.
struct FlatPlan {
  bool ok, mode2d, pair;
  int bn, rows;
};
static FlatPlan flat_plan(const uegan_conv_desc* d) {
  FlatPlan p = {false, false, false, 0, 0};
  if (g_tuning[UEGAN_TUNE_FLAT_S2] == 0) return p;
  if (d->dtype != UEGAN_BF16 || d->stride != 2 || d->KH != d->KW || !(d->KH == 3 || d->KH == 5 || d->KH == 7) || d->C2) return p;
  if (d->pad_mode != UEGAN_PAD_REFLECT || d->pad != d->KH / 2 || d->H % 2 || d->W % 2) return p;
  if (d->Cout % 32 || (d->C1 != 32 && d->C1 != 64 && d->C1 % 128)) return p;
  if (d->KH == 3 && d->C1 <= 64) return p;      // (G.enc3: 9 taps in 20 steps on 64-channel blocks + the fold of a 128 x 128 map: 0.193 vs 0.158 ms on the class
                                                // launches; G.enc2 is HBM-bound on the streaming kernel)
  const int A = (d->KH + 1) / 2, PP = d->Wo + 2 * (A - 1), CH = d->Ho + A - 1;
  if (d->Ho != (d->H + 2 * d->pad - d->KH) / 2 + 1 || d->Wo != (d->W + 2 * d->pad - d->KW) / 2 + 1) return p;
  p.pair = d->C1 == 32;
  p.bn = d->C1 <= 64 ? 64 : 128;
  const int maxpg = p.bn == 64 ? 30 : 22;
  const int rows_flat = FLAT_TM + (A - 1) * (PP + 1), rows_2d = (8 + A - 1) * (32 + A - 1);
  const bool fits_flat = (rows_flat + 15) / 16 <= maxpg && (long long)d->B * CH * PP + rows_flat < (1LL << 21) && PP < 2048 && CH < 2048;
  const bool fits_2d = (rows_2d + 15) / 16 <= maxpg && (long long)d->B * ((CH + 7) / 8) * ((d->Wo + A - 1 + 31) / 32) < (1LL << 21);
  if (!fits_flat && !fits_2d) return p;
  p.mode2d = !fits_flat;
  p.rows = p.mode2d ? rows_2d : rows_flat;
  p.ok = true;
  return p;
}
bool conv_flat_applicable(const uegan_conv_desc* d) { return flat_plan(d).ok; }

int conv_flat_run(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const float* scale, void* out, hipStream_t s) {
  const FlatPlan fp = flat_plan(d);
  if (!fp.ok) return 1;
  FlatArgs a;
  const int K = d->KH, A = (K + 1) / 2;
  a.dz = dz; a.w = w_ihwo; a.out = out; a.scale = scale; a.scale_group = d->scale_group;
  a.B = d->B; a.Ho = d->Ho; a.Wo = d->Wo; a.C = d->Cout;
  a.N = d->C1; a.Kp = (int)uegan_packed_k((int64_t)K * K * d->Cout);
  a.A = A;
  a.CW = d->Wo + A - 1; a.CH = d->Ho + A - 1; a.PP = a.CW + A - 1;
  a.OHp = d->H + 2 * d->pad; a.OWp = d->W + 2 * d->pad;
  a.mPP = (unsigned)(((1ULL << 32) + a.PP - 1) / a.PP);
  a.mCH = (unsigned)(((1ULL << 32) + a.CH - 1) / a.CH);
  a.mode2d = fp.mode2d ? 1 : 0;
  a.pair = fp.pair ? 1 : 0;
  a.PWt = 32 + A - 1; a.nty2 = (a.CH + 7) / 8; a.ntx2 = (a.CW + 31) / 32;
  a.mPWt = (unsigned)(((1ULL << 32) + a.PWt - 1) / a.PWt);
  a.mntx2 = (unsigned)(((1ULL << 32) + a.ntx2 - 1) / a.ntx2);
  a.mnty2 = (unsigned)(((1ULL << 32) + a.nty2 - 1) / a.nty2);
  const int rows = fp.rows;
  a.npg = (rows + 15) / 16;
  const int bn = fp.bn;
  a.nbc = fp.pair ? 1 : d->C1 / bn;
  const int pitch = fp.mode2d ? a.PWt : a.PP;          // patch rows per class-grid row
  // steps per chunk: the next chunk's patch pieces go out 3 per wave per step and must have been requested two steps before the chunk ends
  const int pieces_per_wave = (a.npg + 3) / 4;
  const int min_nt = (pieces_per_wave + 2) / 3 + 2;
  for (int c = 0; c < 4; ++c) { a.nt[c] = a.nlive[c] = 0; for (int n = 0; n < FLAT_NT_MAX; ++n) { a.shift[c][n] = 0; a.wcol[c][n] = -1; } }
  if (!fp.pair) {
    for (int ry = 0; ry < 2; ++ry)
      for (int rx = 0; rx < 2; ++rx) {
        const int c = ry * 2 + rx, ay_n = ry ? K / 2 : A, ax_n = rx ? K / 2 : A;
        int n = 0;
        for (int ay = 0; ay < ay_n; ++ay)
          for (int ax = 0; ax < ax_n; ++ax, ++n) {
            a.shift[c][n] = (A - 1 - ay) * pitch + (A - 1 - ax);
            a.wcol[c][n] = ((2 * ay + ry) * K + (2 * ax + rx)) * d->Cout;
          }
        a.nlive[c] = n;
        a.nt[c] = n > min_nt ? n : min_nt;
        if (a.nt[c] > FLAT_NT_MAX) return 1;
      }
  } else {
    // 32 input channels: class = ry, steps = (ay, ax) over the WIDER column class; the narrower one (rx = 1) has no tap at ax = A - 1: a slice of zeros
    for (int ry = 0; ry < 2; ++ry) {
      const int ay_n = ry ? K / 2 : A;
      int n = 0;
      for (int ay = 0; ay < ay_n; ++ay)
        for (int ax = 0; ax < A; ++ax, ++n) {
          a.shift[ry][n] = (A - 1 - ay) * pitch + (A - 1 - ax);
          a.wcol[ry][n] = ((2 * ay + ry) * K + 2 * ax) * d->Cout;
          a.wcol[2 + ry][n] = 2 * ax + 1 < K ? ((2 * ay + ry) * K + 2 * ax + 1) * d->Cout : -1;
        }
      a.nlive[ry] = n;
      a.nt[ry] = n > min_nt ? n : min_nt;
      if (a.nt[ry] > FLAT_NT_MAX) return 1;
    }
  }
  const long long total = (long long)d->B * a.CH * a.PP;
  const int ntiles = fp.mode2d ? d->B * a.nty2 * a.ntx2 : (int)((total + FLAT_TM - 1) / FLAT_TM);
  a.ntiles = ntiles;
  const dim3 grid((ntiles + 7) / 8 * 8, fp.pair ? 2 : 4 * a.nbc), block(256);
  a.xcd_map = 1;
  const double rowsd = (double)d->B * a.OHp * a.OWp;
  ProfScope prof(prof_key(7, true, bn, K, 1, 4, true), 2.0 * (double)d->B * d->Ho * d->Wo * d->C1 * (double)(K * K * d->Cout), s,
                 2.0 * (rowsd * d->C1 + (double)d->B * d->Ho * d->Wo * d->Cout));
  if (bn == 64) hipLaunchKernelGGL((conv_flat_kernel<2>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((conv_flat_kernel<4>), grid, block, 0, s, a);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

}  // namespace uegan
