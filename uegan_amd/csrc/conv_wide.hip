// Wide-tile patch-resident convolution kernel (bf16): one wave per SIMD, 32x32x16 MFMA fragments, the whole 512-entry register file.
//
// conv_patch_kernel (conv_patch.h) runs 8 waves of 256 registers per CU: two waves share each SIMD's matrix pipe, all eight meet at one
// barrier per K step and then all re-read their fragments from LDS at once -- its 256-channel instantiation measured 43 % MFMA-busy with
// 37 % of the wave time parked at the barrier (profiles/r02_pmc_patch256.txt).  This kernel gives each SIMD ONE wave that owns a
// 128-channel x 128-pixel sub-tile: 16 accumulators of 32x32 (256 registers, the accumulator half of the file), fragments of the next
// 16-deep K sub-step read from LDS while the 16 MFMAs of the current one run (8 ds_read_b128 per 16 x 32-cycle MFMAs), and the first
// fragments of the NEXT half step prefetched before its barrier, so the matrix pipe does not drain across a barrier.
//
//   block  = 256 channels x (8 rows x 32 columns) pixels, 4 waves as 2 (channel halves) x 2 (row halves)
//   K step = one tap x 64 input channels, split in two half steps of 32 channels: the weights stream through a ring of FOUR 16-KB half
//            slices [256 rows][64 B] -- half slice h+3 is requested after barrier h and must have landed before barrier h+2 (counted
//            vmcnt), so at barrier h the half slices <= h+1 are visible and the prefetch across the barrier is legal
//   patch  = (8+KS-1) x (32+KS-1) pixels x 64 channels (128-byte rows), double buffered per 64-channel chunk; the next chunk's patch is
//            requested one 1-KB piece per wave per half step, so it never forms a burst and every piece has >= one half step to land
//   LDS rows are XOR-swizzled on the 16-byte chunk index (patch: chunk ^ ((row>>1)&7), weights: chunk ^ ((row>>2)&3)): a 32x32x16
//   fragment read (32 consecutive rows, one chunk per half wave) is bank-conflict free for ANY first row, i.e. for every tap offset.
//
// Fragment layouts (gfx950, checked on hardware by uegan_selftest_mfma): A lane l = A[i = l&31][k = 8*(l>>5)+e], B lane l =
// B[k = 8*(l>>5)+e][j = l&31], D lane l register r = D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].  A = weights (i = channel),
// B = pixels (j = column inside one tile row): a lane ends up with 4 x 4 consecutive channels of one pixel per fragment.
#include "conv_core.h"

namespace uegan {

// zeros in global memory behind every masked lane of a patch load: one 128-byte line per 64-channel chunk, up to 1024 input channels
static __device__ __attribute__((aligned(128))) const unsigned int g_zero_page[512] = {0u};

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

#define UEGAN_SB() __builtin_amdgcn_sched_barrier(0)

// MODE 0: forward (zero or reflection padding); MODE 1: data gradient of a zero-padded stride-1 convolution (flipped taps, no mirrored
// images).  MASK: the epilogue multiplies by act'(a.mask) (deferred activation gradient of the producer).
//
// Instruction placement.  A 32x32x16 MFMA occupies the SIMD's matrix pipe for 32 cycles = 8 issue slots, and this wave is alone on its
// SIMD: whatever is issued between two MFMAs runs in the shadow of the first, whatever is issued in a block of its own leaves the pipe
// idle.  The half step is therefore ONE basic block (no branches: loads that have nothing to fetch go to a dump area of the LDS, so
// every half step issues exactly 5 direct-to-LDS loads per wave and the vmcnt wait is a constant) written as 32 slots of {one MFMA, a
// few other instructions}, pinned with sched_barrier.  Fragment reads are placed >= 5 MFMAs before their first use and never directly
// in front of a wait for older reads (the compiler's lgkmcnt wait is then always for reads issued long ago).
template <int KS, int MODE, bool MASK, int ABL = 0>      // ABL: timing ablations (tools only; results are garbage): 1 no loads, 2 no LDS reads, 3 neither, 4 + no barriers
__global__ void __launch_bounds__(256, 1) conv_wide_kernel(ConvArgs a) {
  constexpr int TH = 8, TW = 32, BN = 256, NWAVES = 4;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int NPG = (PH * PW + 7) / 8;               // 1-KB pieces (8 patch rows) of one patch buffer
  constexpr int NI_P = (NPG + NWAVES - 1) / NWAVES;    // pieces per wave
  constexpr int PBUFB = NPG * 1024, WHALF = BN * 64, DUMPB = 4096;
  constexpr int NT = KS * KS, NHC = 2 * NT;            // taps, half steps per 64-channel chunk
  constexpr bool DGRAD = MODE != 0;
  static_assert(NI_P <= NHC - 2, "the next chunk's patch pieces must be requested two half steps before the chunk ends");
  static_assert(2 * PBUFB + 4 * WHALF + DUMPB <= 160 * 1024, "LDS budget");

  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PBUFB + 4 * WHALF + DUMPB];
  unsigned char* const lds_w = lds + 2 * PBUFB;
  unsigned char* const lds_dump = lds_w + 4 * WHALF;

  const ConvGeom& g = a.g;
  const bf16_t* in1 = static_cast<const bf16_t*>(a.in1);
  const bf16_t* w = static_cast<const bf16_t*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int l31 = lane & 31, lh = lane >> 5;
  // XCD-aware block -> (tile, channel block) mapping.  Workgroups go to the 8 XCDs round-robin by their linear id, so the ids L and L + 8 run
  // back to back on ONE XCD: with a.xcd_map those two (.. four) are the 256-channel blocks of the SAME tile, and the second finds the tile's
  // patch in that XCD's L2.  (The launch's natural order runs all tiles of channel block 0 before any of block 1: every patch came from
  // HBM / MALL once per channel block -- FETCH_SIZE of a 512 -> 512 layer at 64^2 x 32: 452 -> 340 MB per launch, 273 MB algorithmic.  The
  // kernel is MFMA / power bound, so its time did not move; the step's other stream gets the bandwidth.)
  int n0 = blockIdx.y * BN;
  int t = blockIdx.x;
  if (a.xcd_map) {
    const int L = blockIdx.x + gridDim.x * blockIdx.y, nby = gridDim.y;      // (gridDim.x is a multiple of 8: checked by the launcher)
    const int xcd = L & 7, grp = L >> 3;
    n0 = (grp % nby) * BN;
    t = (grp / nby) * 8 + xcd;
  }
  const int tile_x = t % a.ntx; t /= a.ntx;
  const int tile_y = t % a.nty;
  const int b = t / a.nty;
  const int y0 = tile_y * TH, x0 = tile_x * TW;
  const int nchunk = g.C / 64;
  const int nsteps = nchunk * NT, nhs = 2 * nsteps;

  // ---- patch staging role: piece rg = ii*4 + wave covers patch rows 8*rg .. 8*rg+7, lane -> (row srow, LDS position spos)
  const int srow = lane >> 3, spos = lane & 7;
  const int c_in_chunk = (spos ^ (((lane >> 4) + 4 * (wave & 1)) & 7)) * 8;      // source chunk = position ^ ((row>>1)&7)
  // my 16 source bytes of piece ii in chunk 0 (outside the image: the zero page, which is as long as the chunk offsets reach)
  const unsigned char* pptr[NI_P];
  {
    const int vy0 = DGRAD ? y0 + g.pad - (KS - 1) : y0 - g.pad;
    const int vx0 = DGRAD ? x0 + g.pad - (KS - 1) : x0 - g.pad;
    const bool refl = !DGRAD && g.pad_mode == UEGAN_PAD_REFLECT;
#pragma unroll
    for (int ii = 0; ii < NI_P; ++ii) {
      const int pr = (ii * NWAVES + wave) * 8 + srow;
      const unsigned char* ptr = reinterpret_cast<const unsigned char*>(g_zero_page);
      if (pr < PH * PW) {
        const int piy = pr / PW, pix = pr - piy * PW;
        int sy = vy0 + piy, sx = vx0 + pix;
        if (refl) { sy = reflect_idx(sy, g.IH); sx = reflect_idx(sx, g.IW); }      // (tiles may overhang: out-of-range mirrors gather zero)
        if (sy >= 0 && sy < g.IH && sx >= 0 && sx < g.IW)
          ptr = reinterpret_cast<const unsigned char*>(in1 + ((size_t)(b * g.IH + sy) * g.IW + sx) * g.C1 + c_in_chunk);
      }
      pptr[ii] = ptr;
    }
  }
  // piece ii of `chunk`'s patch (no piece: a load into the dump area -- the load count per half step stays uniform)
  const unsigned char* p_src = pptr[0];
  unsigned char* p_dst = lds_dump;
  auto patch_piece_prepare = [&](int ii, int chunk, bool live) {
    const int rg = ii * NWAVES + wave;
    live = live && rg < NPG;
    p_src = pptr[ii < NI_P ? ii : 0] + (live ? chunk * 128 : 0);
    p_dst = live ? lds + (chunk & 1) * PBUFB + rg * 1024 : lds_dump + wave * 1024;
  };
  auto patch_piece_issue = [&]() { if (!(ABL & 1)) glds16(p_src, p_dst); };
  // ---- weight staging role: a 1-KB piece is 16 rows x 64 B, lane -> (row lane>>2, position lane&3); 4 pieces per wave per half slice
  const bf16_t* const wlane = w + (size_t)(n0 + wave * 16 + (lane >> 2)) * a.Kp + (((lane & 3) ^ ((lane >> 4) & 3)) << 3);
  const size_t wrow64 = (size_t)64 * a.Kp;
  int s_chunk = 0, s_tap = 0, s_half = 0, s_hs = 0;      // staging cursor: the next half slice to request
  const bf16_t* wsrc_cur = wlane;
  unsigned char* wdst_cur = lds_w;
  int wdst_stride = 4096;
  auto stage_w_prepare = [&]() {                     // source / destination of half slice s_hs, then advance the cursor (scalar work)
    const bool live = s_hs < nhs;                    // (past the end: the clamped cursor re-reads the last chunk into the dump area)
    wsrc_cur = wlane + (s_tap * g.C + s_chunk * 64 + s_half * 32);
    wdst_cur = live ? lds_w + (s_hs & 3) * WHALF + wave * 1024 : lds_dump + wave * 1024;
    wdst_stride = live ? 4096 : 0;
    ++s_hs;
    s_half ^= 1;
    s_tap += s_half == 0 ? 1 : 0;
    const int wrap = s_tap == NT ? 1 : 0;
    s_tap = wrap ? 0 : s_tap;
    s_chunk = (s_chunk + wrap < nchunk) ? s_chunk + wrap : nchunk - 1;
  };
  auto stage_w_piece = [&](int i) { if (!(ABL & 1) || s_hs <= 3) glds16(wsrc_cur + (size_t)i * wrow64, wdst_cur + i * wdst_stride); };

  // ---- fragment addresses
  int wad[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wad[i] = (wn * 128 + i * 32 + l31) * 64 + ((lh ^ ((l31 >> 2) & 3)) << 4);
  int xad[4];
  auto set_xad1 = [&](int tap, int j) {
    const int ty = tap / KS, tx = tap - ty * KS;
    const int pty = DGRAD ? KS - 1 - ty : ty, ptx = DGRAD ? KS - 1 - tx : tx;
    const int pr = (wm * 4 + j + pty) * PW + l31 + ptx;
    xad[j] = pr * 128 + ((lh ^ ((pr >> 1) & 7)) << 4);
  };
  u32x4 wf0[4], xf0[4], wf1[4], xf1[4];
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: patch of chunk 0, half slices 0, 1, 2
#pragma unroll
  for (int ii = 0; ii < NI_P; ++ii) { patch_piece_prepare(ii, 0, true); patch_piece_issue(); }
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    stage_w_prepare();
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_w_piece(i);
  }
  wait_vmcnt<4>();
  raw_barrier();                                     // barrier 0: patch 0 and half slices 0, 1 are visible
#pragma unroll
  for (int j = 0; j < 4; ++j) set_xad1(0, j);
#pragma unroll
  for (int i = 0; i < 4; ++i) wf0[i] = *reinterpret_cast<const u32x4*>(lds_w + wad[i]);
#pragma unroll
  for (int j = 0; j < 4; ++j) xf0[j] = *reinterpret_cast<const u32x4*>(lds + xad[j]);

#define MF(W, X, i, j) acc[i][j] = mfma32_bf16(W[i], X[j], acc[i][j]); UEGAN_SB();
#define LDW(F, slot, ksub, i) if (!(ABL & 2)) F[i] = *reinterpret_cast<const u32x4*>((slot) + (wad[i] ^ ((ksub) << 5)));
#define LDX(F, pb, kq, j) if (!(ABL & 2)) F[j] = *reinterpret_cast<const u32x4*>((pb) + (xad[j] ^ ((kq) << 5)));
  int chunk = 0, tap = 0;
  for (int step = 0; step < nsteps; ++step) {
    const unsigned char* pcur = lds + (chunk & 1) * PBUFB;
    const unsigned char* ws0 = lds_w + (step & 1) * (2 * WHALF);
    const unsigned char* ws1 = ws0 + WHALF;
    // =============== half step 0 of this tap.  Batch: half slice 2*step+3 (its ring slot held 2*step-1, which every wave has left) and,
    // during the first NI_P half steps of a chunk, one piece of the next chunk's patch.  The compiler waits lgkmcnt(0) in front of the
    // first MFMA of a fragment group: the group's LDS reads sit >= 5 MFMAs earlier, in pairs (one read per slot measured 4 % slower).
    UEGAN_SB();
    MF(wf0, xf0, 0, 0) stage_w_prepare(); UEGAN_SB();
    MF(wf0, xf0, 0, 1) stage_w_piece(0); UEGAN_SB();
    MF(wf0, xf0, 0, 2) stage_w_piece(1); UEGAN_SB();
    MF(wf0, xf0, 0, 3) stage_w_piece(2); UEGAN_SB();
    MF(wf0, xf0, 1, 0) stage_w_piece(3); UEGAN_SB();
    MF(wf0, xf0, 1, 1) patch_piece_prepare(2 * tap, chunk + 1, 2 * tap < NI_P && chunk + 1 < nchunk); UEGAN_SB();
    MF(wf0, xf0, 1, 2) patch_piece_issue(); UEGAN_SB();
    MF(wf0, xf0, 1, 3) LDW(wf1, ws0, 1, 0) LDW(wf1, ws0, 1, 1) UEGAN_SB();
    MF(wf0, xf0, 2, 0) LDW(wf1, ws0, 1, 2) LDW(wf1, ws0, 1, 3) UEGAN_SB();
    MF(wf0, xf0, 2, 1) LDX(xf1, pcur, 1, 0) LDX(xf1, pcur, 1, 1) UEGAN_SB();
    MF(wf0, xf0, 2, 2) LDX(xf1, pcur, 1, 2) LDX(xf1, pcur, 1, 3) UEGAN_SB();
    MF(wf0, xf0, 2, 3) MF(wf0, xf0, 3, 0) MF(wf0, xf0, 3, 1) MF(wf0, xf0, 3, 2) MF(wf0, xf0, 3, 3)
    MF(wf1, xf1, 0, 0) MF(wf1, xf1, 0, 1) MF(wf1, xf1, 0, 2) MF(wf1, xf1, 0, 3)
    MF(wf1, xf1, 1, 0) LDW(wf0, ws1, 0, 0) LDW(wf0, ws1, 0, 1) UEGAN_SB();        // (half slice 2*step+1 is visible since barrier 2*step)
    MF(wf1, xf1, 1, 1) LDW(wf0, ws1, 0, 2) LDW(wf0, ws1, 0, 3) UEGAN_SB();
    MF(wf1, xf1, 1, 2) LDX(xf0, pcur, 2, 0) LDX(xf0, pcur, 2, 1) UEGAN_SB();
    MF(wf1, xf1, 1, 3) LDX(xf0, pcur, 2, 2) LDX(xf0, pcur, 2, 3) UEGAN_SB();
    MF(wf1, xf1, 2, 0) MF(wf1, xf1, 2, 1) MF(wf1, xf1, 2, 2) MF(wf1, xf1, 2, 3)
    MF(wf1, xf1, 3, 0) MF(wf1, xf1, 3, 1) MF(wf1, xf1, 3, 2) MF(wf1, xf1, 3, 3)
    wait_vmcnt<5>();                                 // everything older than this half step's batch has landed: half slice 2*step+2
    if (ABL < 4) raw_barrier();
    // =============== half step 1
    UEGAN_SB();
    MF(wf0, xf0, 0, 0) stage_w_prepare(); UEGAN_SB();
    MF(wf0, xf0, 0, 1) stage_w_piece(0); UEGAN_SB();
    MF(wf0, xf0, 0, 2) stage_w_piece(1); UEGAN_SB();
    MF(wf0, xf0, 0, 3) stage_w_piece(2); UEGAN_SB();
    MF(wf0, xf0, 1, 0) stage_w_piece(3); UEGAN_SB();
    MF(wf0, xf0, 1, 1) patch_piece_prepare(2 * tap + 1, chunk + 1, 2 * tap + 1 < NI_P && chunk + 1 < nchunk); UEGAN_SB();
    MF(wf0, xf0, 1, 2) patch_piece_issue(); UEGAN_SB();
    MF(wf0, xf0, 1, 3) LDW(wf1, ws1, 1, 0) LDW(wf1, ws1, 1, 1) UEGAN_SB();
    MF(wf0, xf0, 2, 0) LDW(wf1, ws1, 1, 2) LDW(wf1, ws1, 1, 3) UEGAN_SB();
    MF(wf0, xf0, 2, 1) LDX(xf1, pcur, 3, 0) LDX(xf1, pcur, 3, 1) UEGAN_SB();
    MF(wf0, xf0, 2, 2) LDX(xf1, pcur, 3, 2) LDX(xf1, pcur, 3, 3) UEGAN_SB();
    MF(wf0, xf0, 2, 3) MF(wf0, xf0, 3, 0) MF(wf0, xf0, 3, 1) MF(wf0, xf0, 3, 2) MF(wf0, xf0, 3, 3)
    {                                                // (past the last step: the reads below fetch LDS bytes nobody uses)
      const int wrap = tap + 1 == NT ? 1 : 0;
      tap = wrap ? 0 : tap + 1;
      chunk += wrap;
    }
    const unsigned char* pnext = lds + (chunk & 1) * PBUFB;
    const unsigned char* wnext = lds_w + ((step + 1) & 1) * (2 * WHALF);
    MF(wf1, xf1, 0, 0) set_xad1(tap, 0); UEGAN_SB();
    MF(wf1, xf1, 0, 1) set_xad1(tap, 1); UEGAN_SB();
    MF(wf1, xf1, 0, 2) set_xad1(tap, 2); UEGAN_SB();
    MF(wf1, xf1, 0, 3) set_xad1(tap, 3); UEGAN_SB();
    MF(wf1, xf1, 1, 0) LDW(wf0, wnext, 0, 0) LDW(wf0, wnext, 0, 1) UEGAN_SB();
    MF(wf1, xf1, 1, 1) LDW(wf0, wnext, 0, 2) LDW(wf0, wnext, 0, 3) UEGAN_SB();
    MF(wf1, xf1, 1, 2) LDX(xf0, pnext, 0, 0) LDX(xf0, pnext, 0, 1) UEGAN_SB();
    MF(wf1, xf1, 1, 3) LDX(xf0, pnext, 0, 2) LDX(xf0, pnext, 0, 3) UEGAN_SB();
    MF(wf1, xf1, 2, 0) MF(wf1, xf1, 2, 1) MF(wf1, xf1, 2, 2) MF(wf1, xf1, 2, 3)
    MF(wf1, xf1, 3, 0) MF(wf1, xf1, 3, 1) MF(wf1, xf1, 3, 2) MF(wf1, xf1, 3, 3)
    wait_vmcnt<5>();
    if (ABL < 4) raw_barrier();
  }
#undef MF
#undef LDW
#undef LDX
  wait_vmcnt<0>();                                   // (the dump-area loads of the last half steps)

  // ---- epilogue: scale, bias, activation in fp32 -> bf16 -> through the LDS -> NHWC rows.
  // A lane holds 4 consecutive channels of a pixel per register quad, so stores straight from the accumulators are 8 bytes per lane, 32
  // different 128-byte lines per instruction and 64 instructions per wave.  The tile is therefore transposed through the (now idle) LDS:
  // each wave writes its 128 pixels x 128 channels as rows of 256 + 8 bytes (the 8-byte pad spreads a ds_write_b64 lane group over all
  // banks) and reads them back 16 bytes per lane (two 8-byte reads: the padded rows are 8-byte aligned), 16 lanes per pixel: a store instruction covers 4 pixels x 256 contiguous bytes (8 whole
  // lines), 32 instructions per wave; the deferred activation gradient reads its mask the same way.
  // (activations of this kernel's layers: none / LeakyReLU / ReLU = max(v, slope*v); their derivative from the output: a > 0 ? 1 : slope)
  constexpr int EROW = 264;
  static_assert(4 * 128 * EROW <= 2 * PBUFB + 4 * WHALF, "epilogue staging fits the main loop's LDS");
  unsigned char* const est = lds + wave * (128 * EROW);
  {
    const float scale = a.scale ? a.scale[a.scale_group ? b / a.scale_group : 0] : 1.f;
    const float slope = a.act == UEGAN_ACT_LRELU ? 0.2f : (a.act == UEGAN_ACT_RELU ? 0.f : 1.f);
    const bool plain = !a.scale && !a.bias && a.act == UEGAN_ACT_NONE;      // (data gradients: nothing but the rounding)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nb0 = n0 + wn * 128 + i * 32 + 4 * lh;
      float bv[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[q][r] = (!plain && a.bias && nb0 + 8 * q + r < a.nbias) ? a.bias[nb0 + 8 * q + r] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned char* row = est + (j * 32 + l31) * EROW + i * 64 + 8 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = plain ? acc[i][j][4 * q + r] : acc[i][j][4 * q + r] * scale + bv[q][r];
            v[r] = plain ? z : fmaxf(z, slope * z);
          }
          u32x2 pk;
          pk.x = pack_bf16x2(v[0], v[1]);
          pk.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<u32x2*>(row + q * 16) = pk;
        }
      }
    }
  }
  // (each wave reads back only what it wrote: no workgroup barrier, the compiler's lgkmcnt wait orders the LDS accesses of one wave)
  __builtin_amdgcn_wave_barrier();
  {
    const float mslope = a.mask_act == UEGAN_ACT_LRELU ? 0.2f : (a.mask_act == UEGAN_ACT_RELU ? 0.f : 1.f);
    bf16_t* out = static_cast<bf16_t*>(a.out);
    const int nl = n0 + wn * 128 + (lane & 15) * 8;
#pragma unroll 4
    for (int it = 0; it < 32; ++it) {
      const int rr = it * 4 + (lane >> 4);             // pixel of my 16 bytes inside the wave's 4 rows x 32 columns
      const int oy = y0 + wm * 4 + (rr >> 5), ox = x0 + (rr & 31);
      const u32x2 v01 = *reinterpret_cast<const u32x2*>(est + rr * EROW + (lane & 15) * 16);        // (rows are 8-byte aligned only)
      const u32x2 v23 = *reinterpret_cast<const u32x2*>(est + rr * EROW + (lane & 15) * 16 + 8);
      u32x4 v = {v01.x, v01.y, v23.x, v23.y};
      if (oy >= g.OH || ox >= g.OW) continue;
      const size_t o = (((size_t)b * g.OH + oy) * g.OW + ox) * a.N + nl;
      if (MASK) {
        const u32x4 m = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(a.mask) + o);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float lo = bits_to_f32(v[d] << 16) * (bits_to_f32(m[d] << 16) > 0.f ? 1.f : mslope);
          const float hi = bits_to_f32(v[d] & 0xffff0000u) * (bits_to_f32(m[d] & 0xffff0000u) > 0.f ? 1.f : mslope);
          v[d] = pack_bf16x2(lo, hi);
        }
      }
      *reinterpret_cast<u32x4*>(out + o) = v;
    }
  }
}

// 0 / error code when the launch was taken, 1 when the problem is not one of this kernel's (the caller falls through to conv_patch)
int conv_wide_run(ConvArgs& a, int dtype, hipStream_t s) {
  const ConvGeom& g = a.g;
  const int min_grid = g_tuning[UEGAN_TUNE_WIDE_MIN_GRID];      // fewer blocks than CUs: the smaller tiles of conv_patch cover the chip better
  if (min_grid < 0) return 1;                        // (uegan_set_tuning: < 0 switches this kernel off, the tests set 1 to reach it on small maps)
  if (dtype != UEGAN_BF16 || g.stride != 1 || g.KH != 3 || g.KW != 3 || a.out2 || a.frame != 0) return 1;
  if (g.C % 64 || g.C > 1024 || g.C2 || a.N % 256 || g.OW < 32 || g.OH < 8) return 1;
  if (g.mode == 1 && g.pad_mode == UEGAN_PAD_REFLECT && g.pad != 0) return 1;       // mirrored images: conv_patch MODE 2
  auto simple = [](int act) { return act == UEGAN_ACT_NONE || act == UEGAN_ACT_LRELU || act == UEGAN_ACT_RELU; };
  if (!simple(a.act) || (a.mask && !simple(a.mask_act))) return 1;
  a.nty = (g.OH + 7) / 8;
  a.ntx = (g.OW + 31) / 32;
  const int gm = g.B * a.nty * a.ntx;
  if (gm * (a.N / 256) < min_grid) return 1;
  const double rows = g.mode == 0 ? (double)g.B * g.OH * g.OW : (double)g.B * g.IH * g.IW;
  ProfScope prof(prof_key(7, true, 256, 3, g.mode, 8, true), 2.0 * rows * a.N * (double)(9 * g.C), s,
                 2.0 * (rows * a.N + (double)g.B * g.IH * g.IW * g.C));
  const dim3 grid(gm, a.N / 256), block(256);
  const int nby = a.N / 256;
  a.xcd_map = (nby > 1 && gm % 8 == 0) ? 1 : 0;
#ifdef UEGAN_TOOLS_BUILD
  const int abl = g_abl_wide;
  if (g.mode == 0 && abl == 1) hipLaunchKernelGGL((conv_wide_kernel<3, 0, false, 1>), grid, block, 0, s, a);
  else if (g.mode == 0 && abl == 2) hipLaunchKernelGGL((conv_wide_kernel<3, 0, false, 2>), grid, block, 0, s, a);
  else if (g.mode == 0 && abl == 3) hipLaunchKernelGGL((conv_wide_kernel<3, 0, false, 3>), grid, block, 0, s, a);
  else if (g.mode == 0 && abl == 4) hipLaunchKernelGGL((conv_wide_kernel<3, 0, false, 7>), grid, block, 0, s, a);
  else
#endif
  if (g.mode == 0) hipLaunchKernelGGL((conv_wide_kernel<3, 0, false>), grid, block, 0, s, a);
  else if (a.mask) hipLaunchKernelGGL((conv_wide_kernel<3, 1, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((conv_wide_kernel<3, 1, false>), grid, block, 0, s, a);
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

}  // namespace uegan
