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
#ifdef UEGAN_HALF_FP16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
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
  static_assert(2 * PBUFB + 4 * WHALF + DUMPB + BN * 4 <= 160 * 1024, "LDS budget");

  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PBUFB + 4 * WHALF + DUMPB + BN * 4];
  unsigned char* const lds_w = lds + 2 * PBUFB;
  unsigned char* const lds_dump = lds_w + 4 * WHALF;
  unsigned char* const lds_bias = lds_dump + DUMPB;    // fp32 [BN]: the epilogue's bias, fetched by the prologue (below)

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
  const int y0 = a.rect_y0 + tile_y * TH, x0 = a.rect_x0 + tile_x * TW;      // (rect: the image-free interior of a reflection-padded data gradient)
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

  // the epilogue's per-channel bias goes to the LDS now, by the direct-to-LDS path (4 bytes per lane, 64 channels per wave): a global load
  // in the epilogue is a full memory round trip with nothing to hide it behind (one wave per SIMD) -- four of them, one per channel fragment
  const bool plain = !a.scale && !a.bias && a.act == UEGAN_ACT_NONE;      // (data gradients: nothing but the rounding)
  if (!(ABL & 1)) {
    const int n = n0 + wave * 64 + lane;
    const bool live = !plain && a.bias && n < a.nbias;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(live ? a.bias + n : reinterpret_cast<const float*>(g_zero_page) + lane),
                                     (__attribute__((address_space(3))) void*)(lds_bias + wave * 256), 4, 0, 0);
  }
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const f32x4*>(lds_bias + (wn * 128 + i * 32 + 4 * lh + 8 * q) * 4);
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
      const size_t pix = ((size_t)b * g.OH + oy) * g.OW + ox;
      const size_t o = pix * a.N + nl;
      if (MASK) {
        const u32x4 m = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(a.mask) + o);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float lo = half_lo_to_f32(v[d]) * (half_lo_to_f32(m[d]) > 0.f ? 1.f : mslope);
          const float hi = half_hi_to_f32(v[d]) * (half_hi_to_f32(m[d]) > 0.f ? 1.f : mslope);
          v[d] = pack_bf16x2(lo, hi);
        }
      }
      // (virtual concat: channels [0, n_out1) and [n_out1, N) of a data gradient go to two tensors; never together with MASK)
      bf16_t* dst = !a.out2 ? out + o : (nl < a.n_out1 ? out + pix * a.n_out1 + nl : static_cast<bf16_t*>(a.out2) + pix * (a.N - a.n_out1) + (nl - a.n_out1));
      *reinterpret_cast<u32x4*>(dst) = v;
    }
  }
}

// =====================================================================================================================================
// conv_tall_kernel: the same one-wave-per-SIMD structure for 3x3 stride-1 layers with 64 / 128 OUTPUT channels (VGG conv1_2 .. conv2_2,
// G.dec2 / dec3 forward): a 256-channel block does not exist there, so the four waves of a block share ONE weight slice and split the
// pixels instead -- block = BN channels x (16 rows x 32 columns), wave w owns rows 4w .. 4w+3 (NI x 4 accumulators of 32x32, NI = BN/32).
//   patch  = 18 x 34 pixels x 32 channels (64-byte rows, 39 KB), double buffered per 32-CHANNEL chunk -- a 64-channel chunk of this tile
//            would be 2 x 78 KB.  The chunk may come from the second source tensor of a virtual concat (dec2 / dec3: chunk-uniform select).
//   K step = one tap x 32 channels = two 16-deep sub-steps (NI x 4 MFMAs each); the weight slices [BN][64 B] stream through a ring of
//            four (slice s+3 requested in step s, counted vmcnt before barrier s+1: visible from barrier s+2 on, read in step s+3 -- and
//            its first fragments already at the end of step s+2); the next chunk's patch arrives <= 2 one-KB pieces per wave per step.
//   The tap loop is unrolled (9 steps per chunk): tap offsets and piece indices are compile-time, a step is one basic block of
//   {MFMA + a few other instructions} slots like conv_wide_kernel's half step.  Every step issues exactly NWP + 2 direct-to-LDS loads per
//   wave (nothing to fetch: the dump area), so the vmcnt wait is a constant.
//   Swizzle of both LDS tiles (64-byte rows): 16-byte chunk q of row r at position q ^ ((r >> 2) & 3) -- conflict-free ds_read_b128 of 32
//   consecutive rows from any first row.
// MODE / MASK as conv_wide_kernel; POOL: the epilogue also writes the 2x2 max-pool of the tile (VGG conv1_2 / conv2_2; bit-identical to
// pooling the stored tensor because rounding to bf16 is monotonic).
// RPW: tile rows per wave.  4: the 16-row tile above, 256 accumulator registers, one block per CU.  2 (128 channels only): an 8-row tile, 128
// accumulator registers and <= 78 KB of LDS -- TWO blocks per CU, i.e. two waves per SIMD from different blocks: one block's prologue and
// store-issue-bound epilogue (20 k of a 76-k-cycle tile, cycle stamps in DESIGN.md 3.1) run under the other block's K loop.
// MODE 2 (round 5): data gradient of a REFLECTION-padded 3x3 convolution (pad 1; G.dec1 - dec3, two destinations), every tile in one launch.
// The adjoint of the padding gives pixel 1 of an axis the term w[t = 0] * dz[0] and pixel n-2 the term w[t = 2] * dz[n-1] on top of the
// zero-padded gradient -- the SAME weight slice as the direct term of that tap, so the mirrored image is folded into the pixel operand instead
// of costing MFMAs of its own:  B(pixel, tap) = dz[direct source] + dz[mirrored source]  (8 halves added in fp32, rounded to the storage format:
// one more rounding of the same size as the one dz already carries, on two rows and two columns of the map).  Only the waves / tiles that
// hold such a row or column take the extra LDS reads (a scalar branch elsewhere).  This replaces the split into an image-free rectangle on
// this kernel + a frame launch on conv_patch_kernel's MODE 2, where the mirrored images were extra MFMAs on masked fragments (+110 % on the
// border tiles, 1.7 ms/step in round 4).
template <int NI, int MODE, bool MASK, bool POOL, int RPW = 4>
__global__ void __launch_bounds__(256, RPW == 2 ? 2 : 1) conv_tall_kernel(ConvArgs a) {
  static_assert(RPW == 4 || RPW == 2, "rows per wave");
  constexpr bool REFL = MODE == 2;
  static_assert(!REFL || (NI == 4 && !MASK && !POOL), "mirrored images: 128-channel blocks, plain epilogue");
  constexpr int KS = 3, TH = 4 * RPW, TW = 32, BN = NI * 32, NWAVES = 4;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1, NPIX = PH * PW;
  constexpr int NPG = (NPIX + 15) / 16;                // 1-KB pieces (16 patch pixels x 64 B) of one patch buffer
  constexpr int NI_P = (NPG + NWAVES - 1) / NWAVES;    // pieces per wave
  constexpr int PBUFB = NPG * 1024, WSL = BN * 64, NWP = WSL / 1024 / NWAVES;
  constexpr int DUMPB = RPW == 2 ? 1024 : 4096, DUMPW = RPW == 2 ? 0 : 1024;      // dump area: one KB per wave, or (two blocks per CU) one shared KB of garbage
  constexpr int NT = KS * KS, NLOAD = NWP + 2;         // taps; direct-to-LDS loads per wave per step
  constexpr bool DGRAD = MODE != 0;
  constexpr int EROW = BN * 2 + 8;                     // epilogue staging row: one pixel's BN channels + 8 B (bank spread)
  constexpr int WPIX = RPW * 32;                       // pixels per wave
  constexpr int MAINB = 2 * PBUFB + 4 * WSL, EPIB = NWAVES * WPIX * EROW;
  constexpr int BODYB = MAINB > EPIB ? MAINB : EPIB;
  static_assert(NI_P <= 2 * (NT - 2), "the next chunk's patch pieces must be requested two steps before the chunk ends");
  static_assert((RPW == 2 ? 2 : 1) * (BODYB + DUMPB + BN * 4) <= 160 * 1024, "LDS budget");
  static_assert(NWP >= 1, "at least one weight piece per wave per slice");

  __shared__ __attribute__((aligned(16))) unsigned char lds[BODYB + DUMPB + BN * 4];
  unsigned char* const lds_w = lds + 2 * PBUFB;
  unsigned char* const lds_dump = lds + BODYB;         // (behind the epilogue staging as well: other waves' dump loads may still land there)
  unsigned char* const lds_bias = lds_dump + DUMPB;    // fp32 [BN]

  const ConvGeom& g = a.g;
  const bf16_t* in1 = static_cast<const bf16_t*>(a.in1);
  const bf16_t* in2 = static_cast<const bf16_t*>(a.in2);
  const bf16_t* w = static_cast<const bf16_t*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  int n0 = blockIdx.y * BN;
  int t = blockIdx.x;
  if (a.xcd_map) {      // the channel blocks of ONE tile back to back on one XCD (see conv_wide_kernel): the second .. fourth find the patch in that L2
    const int L = blockIdx.x + gridDim.x * blockIdx.y, nby = gridDim.y;
    const int xcd = L & 7, grp = L >> 3;
    n0 = (grp % nby) * BN;
    t = (grp / nby) * 8 + xcd;
  }
  const int tile_x = t % a.ntx; t /= a.ntx;
  const int tile_y = t % a.nty;
  const int b = t / a.nty;
  const int y0 = a.rect_y0 + tile_y * TH, x0 = a.rect_x0 + tile_x * TW;      // (rect: the image-free interior of a reflection-padded data gradient)
  const int nchunk = g.C / 32, nchunk1 = g.C1 / 32;
  const int nsteps = nchunk * NT;

  // ---- patch staging role: piece rg = ii*4 + wave covers patch pixels 16*rg .. 16*rg+15; lane -> (pixel lane>>2, LDS position lane&3)
  const int q_src = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;      // source channel offset inside the chunk = (position ^ ((row>>2)&3)) * 8
  int ppix[NI_P];                                              // my source pixel of piece ii (linear over [B][IH][IW]); -1: zeros
  {
    const int vy0 = DGRAD ? y0 + g.pad - (KS - 1) : y0 - g.pad;
    const int vx0 = DGRAD ? x0 + g.pad - (KS - 1) : x0 - g.pad;
    const bool refl = !DGRAD && g.pad_mode == UEGAN_PAD_REFLECT;
#pragma unroll
    for (int ii = 0; ii < NI_P; ++ii) {
      const int pr = (ii * NWAVES + wave) * 16 + (lane >> 2);
      int pix = -1;
      if (pr < NPIX) {
        const int piy = pr / PW, pix_x = pr - piy * PW;
        int sy = vy0 + piy, sx = vx0 + pix_x;
        if (refl) { sy = reflect_idx(sy, g.IH); sx = reflect_idx(sx, g.IW); }      // (tiles may overhang: out-of-range mirrors gather zero)
        if (sy >= 0 && sy < g.IH && sx >= 0 && sx < g.IW) pix = (b * g.IH + sy) * g.IW + sx;
      }
      ppix[ii] = pix;
    }
  }
  const unsigned char* const zero16 = reinterpret_cast<const unsigned char*>(g_zero_page) + (lane & 3) * 16;
  const unsigned char* p_src = zero16;
  unsigned char* p_dst = lds_dump;
  // piece `pidx` (compile-time) of `chunk`'s patch; no such piece: a load into the dump area
  auto patch_piece_prepare = [&](int pidx, int chunk, bool live) {
    const int rg = pidx * NWAVES + wave;
    live = live && pidx < NI_P && rg < NPG;
    const int pix = ppix[pidx < NI_P ? pidx : 0];
    const bool first = chunk < nchunk1;
    const bf16_t* base = first ? in1 : in2;
    const int cs = first ? g.C1 : g.C2, c0 = (first ? chunk : chunk - nchunk1) * 32 + q_src;
    p_src = (live && pix >= 0) ? reinterpret_cast<const unsigned char*>(base + ((size_t)pix * cs + c0)) : zero16;
    p_dst = live ? lds + (chunk & 1) * PBUFB + rg * 1024 : lds_dump + wave * DUMPW;
  };
  auto patch_piece_issue = [&]() { glds16(p_src, p_dst); };
  // ---- weight staging role: a 1-KB piece is 16 rows x 64 B, lane -> (row lane>>2, position lane&3); NWP pieces per wave per slice
  const bf16_t* const wlane = w + (size_t)(n0 + wave * 16 + (lane >> 2)) * a.Kp + q_src;
  const size_t wrow64 = (size_t)64 * a.Kp;
  int s_chunk = 0, s_tap = 0, s_idx = 0;               // staging cursor: the next slice to request
  const bf16_t* wsrc_cur = wlane;
  unsigned char* wdst_cur = lds_w;
  int wdst_stride = 4096;
  auto stage_w_prepare = [&]() {
    const bool live = s_idx < nsteps;                  // (past the end: the clamped cursor re-reads the last slice into the dump area)
    wsrc_cur = wlane + (s_tap * g.C + s_chunk * 32);
    wdst_cur = live ? lds_w + (s_idx & 3) * WSL + wave * 1024 : lds_dump + wave * DUMPW;
    wdst_stride = live ? 4096 : 0;
    ++s_idx;
    ++s_tap;
    const int wrap = s_tap == NT ? 1 : 0;
    s_tap = wrap ? 0 : s_tap;
    s_chunk = (s_chunk + wrap < nchunk) ? s_chunk + wrap : nchunk - 1;
  };
  auto stage_w_piece = [&](int i) { glds16(wsrc_cur + (size_t)i * wrow64, wdst_cur + i * wdst_stride); };

  // ---- fragment addresses
  int wad[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) wad[i] = (i * 32 + l31) * 64 + ((lh ^ ((l31 >> 2) & 3)) << 4);
  const int xbase = wave * RPW * PW + l31;               // patch pixel of my column in my first row, tap (0, 0)
  int xad[RPW];
  int xpr[REFL ? RPW : 1];                             // REFL: the patch pixel behind xad[j] (the mirrored sources are 2 rows / 2 columns away)
  auto set_xad = [&](int tap, int j) {                 // (tap is a compile-time constant at every call)
    const int ty = tap / KS, tx = tap - ty * KS;
    const int pty = DGRAD ? KS - 1 - ty : ty, ptx = DGRAD ? KS - 1 - tx : tx;
    const int pr = xbase + (j + pty) * PW + ptx;
    xad[j] = pr * 64 + ((lh ^ ((pr >> 2) & 3)) << 4);
    if constexpr (REFL) xpr[j] = pr;
  };
  // REFL: rows of this wave / columns of this tile that receive a mirrored image (pad 1: pixel 1 through tap 0, pixel n-2 through tap 2)
  bool r_top[RPW], r_bot[RPW];
  bool e_left = false, e_right = false, c_left = false, c_right = false, edge = false;
  if constexpr (REFL) {
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int y = y0 + wave * RPW + j;
      r_top[j] = y == 1;
      r_bot[j] = y == g.OH - 2;
      edge = edge || r_top[j] || r_bot[j];
    }
    e_left = x0 == 0;
    e_right = x0 <= g.OW - 2 && g.OW - 2 < x0 + TW;
    c_left = x0 + l31 == 1;
    c_right = x0 + l31 == g.OW - 2;
    edge = edge || e_left || e_right;
  } else {
#pragma unroll
    for (int j = 0; j < RPW; ++j) r_top[j] = r_bot[j] = false;
  }
  // fragment F = pixels of row j, tap `tap`, sub-step ksub read from patch buffer pb: add the mirrored sources of that (row, tap)
  auto fold1 = [&](u32x4& F, const unsigned char* pb, int ksub, int j, int tap) {
    const int ty = tap / KS, tx = tap - ty * KS;
    const bool ym = ty == 0 ? r_top[j] : (ty == 2 ? r_bot[j] : false);          // wave-uniform
    const bool xany = tx == 0 ? e_left : (tx == 2 ? e_right : false);           // tile-uniform: some lane of the row has an x mirror
    if (!ym && !xany) return;
    const int dpy = ty == 0 ? -2 * PW : 2 * PW, dpx = tx == 0 ? -2 : 2;
    auto rd = [&](int pr) { return *reinterpret_cast<const u32x4*>(pb + ((pr * 64 + ((lh ^ ((pr >> 2) & 3)) << 4)) ^ (ksub << 5))); };
    if (ym) F = add_frag<bf16_t>(F, rd(xpr[j] + dpy));
    if (xany) {
      const uint32_t m = (tx == 0 ? c_left : c_right) ? 0xffffffffu : 0u;
      u32x4 v = rd(xpr[j] + dpx);
      if (ym) v = add_frag<bf16_t>(v, rd(xpr[j] + dpy + dpx));                            // the corner: mirrored in both axes
      F = add_frag<bf16_t>(F, v & u32x4{m, m, m, m});
    }
  };
  u32x4 wf0[NI], xf0[RPW], wf1[NI], xf1[RPW];
  f32x16 acc[NI][RPW];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // the epilogue's per-channel bias goes to the LDS NOW, by the same direct-to-LDS path (4 bytes per lane): a load in the epilogue is a
  // full memory round trip with nothing to hide it behind (one wave per SIMD)
  const bool plain = !a.scale && !a.bias && a.act == UEGAN_ACT_NONE;      // (data gradients: nothing but the rounding)
  {
    const int n = n0 + wave * 64 + lane;
    const bool live = wave * 64 < BN && !plain && a.bias && n < a.nbias;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(live ? a.bias + n : reinterpret_cast<const float*>(g_zero_page) + lane),
                                     (__attribute__((address_space(3))) void*)(wave * 64 < BN ? lds_bias + wave * 256 : lds_dump + wave * DUMPW), 4, 0, 0);
  }
  const float scale = a.scale ? a.scale[a.scale_group ? b / a.scale_group : 0] : 1.f;

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
  for (int j = 0; j < RPW; ++j) set_xad(0, j);
#pragma unroll
  for (int i = 0; i < NI; ++i) wf0[i] = *reinterpret_cast<const u32x4*>(lds_w + wad[i]);
#pragma unroll
  for (int j = 0; j < RPW; ++j) xf0[j] = *reinterpret_cast<const u32x4*>(lds + xad[j]);

#define MF(W, X, i, j) acc[i][j] = mfma32_bf16(W[i], X[j], acc[i][j]); UEGAN_SB();
#define LDW(F, slot, ksub, i) F[i] = *reinterpret_cast<const u32x4*>((slot) + (wad[i] ^ ((ksub) << 5)));
#define LDX(F, pb, ksub, j) F[j] = *reinterpret_cast<const u32x4*>((pb) + (xad[j] ^ ((ksub) << 5)));
// REFL: the mirrored sources of fragment set F (all rows of the wave; taps whose xad / xpr are the current ones)
#define FOLDX(F, pb, ksub, tapv)                                                           \
  if constexpr (REFL) {                                                                    \
    if (edge) {                                                                            \
      _Pragma("unroll") for (int jf = 0; jf < RPW; ++jf) fold1(F[jf], pb, ksub, jf, tapv); \
    }                                                                                      \
  }
  FOLDX(xf0, lds, 0, 0)
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const unsigned char* pcur = lds + (chunk & 1) * PBUFB;
    const bool more = chunk + 1 < nchunk;
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int step = chunk * NT + tap;
      const unsigned char* ws = lds_w + (step & 3) * WSL;
      const unsigned char* wnext = lds_w + ((step + 1) & 3) * WSL;
      const unsigned char* pnext = tap + 1 == NT ? lds + ((chunk + 1) & 1) * PBUFB : pcur;
      const int ntap = tap + 1 == NT ? 0 : tap + 1;
      UEGAN_SB();
      if constexpr (RPW == 2 && NI == 2) {
        // 4 + 4 MFMAs (two channel fragments x two rows): one LDS fragment read per MFMA -- the CU's other block runs in what that leaves
        MF(wf0, xf0, 0, 0) LDW(wf1, ws, 1, 0) LDW(wf1, ws, 1, 1) UEGAN_SB();
        MF(wf0, xf0, 0, 1) LDX(xf1, pcur, 1, 0) LDX(xf1, pcur, 1, 1) UEGAN_SB();
        MF(wf0, xf0, 1, 0) stage_w_prepare(); stage_w_piece(0); UEGAN_SB();
        // (this variant issues 8.6 vector instructions per MFMA -- PMC, conv1_2 -- most of them address arithmetic of the direct-to-LDS loads: the
        // six patch pieces per wave go out in the first three taps and the other taps issue no patch load at all, not even into the dump area)
        if (2 * tap < NI_P) {      // (compile-time after unrolling)
          MF(wf0, xf0, 1, 1) patch_piece_prepare(2 * tap, chunk + 1, more); patch_piece_issue(); patch_piece_prepare(2 * tap + 1, chunk + 1, more); patch_piece_issue(); UEGAN_SB();
        } else {
          MF(wf0, xf0, 1, 1)
        }
        MF(wf1, xf1, 0, 0) set_xad(ntap, 0); set_xad(ntap, 1); UEGAN_SB();
        MF(wf1, xf1, 0, 1) LDW(wf0, wnext, 0, 0) LDW(wf0, wnext, 0, 1) UEGAN_SB();
        MF(wf1, xf1, 1, 0) LDX(xf0, pnext, 0, 0) LDX(xf0, pnext, 0, 1) UEGAN_SB();
        MF(wf1, xf1, 1, 1)
      } else if constexpr (RPW == 2) {
        // 8 + 8 MFMAs (four channel fragments x two rows); the partner wave on this SIMD (the CU's other block) fills what this one leaves
        MF(wf0, xf0, 0, 0) LDW(wf1, ws, 1, 0) LDW(wf1, ws, 1, 1) UEGAN_SB();
        MF(wf0, xf0, 0, 1) LDW(wf1, ws, 1, 2) LDW(wf1, ws, 1, 3) UEGAN_SB();
        MF(wf0, xf0, 1, 0) LDX(xf1, pcur, 1, 0) LDX(xf1, pcur, 1, 1) UEGAN_SB();
        MF(wf0, xf0, 1, 1) stage_w_prepare(); UEGAN_SB();
        MF(wf0, xf0, 2, 0) stage_w_piece(0); UEGAN_SB();
        MF(wf0, xf0, 2, 1) stage_w_piece(1); UEGAN_SB();
        MF(wf0, xf0, 3, 0) patch_piece_prepare(2 * tap, chunk + 1, more); UEGAN_SB();
        MF(wf0, xf0, 3, 1) patch_piece_issue(); patch_piece_prepare(2 * tap + 1, chunk + 1, more); patch_piece_issue(); UEGAN_SB();
        FOLDX(xf1, pcur, 1, tap)
        MF(wf1, xf1, 0, 0) set_xad(ntap, 0); set_xad(ntap, 1); UEGAN_SB();
        MF(wf1, xf1, 0, 1) LDW(wf0, wnext, 0, 0) LDW(wf0, wnext, 0, 1) UEGAN_SB();
        MF(wf1, xf1, 1, 0) LDW(wf0, wnext, 0, 2) LDW(wf0, wnext, 0, 3) UEGAN_SB();
        MF(wf1, xf1, 1, 1) LDX(xf0, pnext, 0, 0) LDX(xf0, pnext, 0, 1) UEGAN_SB();
        MF(wf1, xf1, 2, 0) MF(wf1, xf1, 2, 1) MF(wf1, xf1, 3, 0) MF(wf1, xf1, 3, 1)
        FOLDX(xf0, pnext, 0, ntap)
      } else if constexpr (NI == 4) {
        // 16 MFMAs on sub-step 0 (fragments read during the previous step), the step's loads and the fragments of sub-step 1 in their
        // shadow; then 16 MFMAs on sub-step 1 with the next step's tap addresses and first fragments in theirs
        MF(wf0, xf0, 0, 0) stage_w_prepare(); UEGAN_SB();
        MF(wf0, xf0, 0, 1) stage_w_piece(0); UEGAN_SB();
        MF(wf0, xf0, 0, 2) stage_w_piece(1); UEGAN_SB();
        MF(wf0, xf0, 0, 3) patch_piece_prepare(2 * tap, chunk + 1, more); UEGAN_SB();
        MF(wf0, xf0, 1, 0) patch_piece_issue(); UEGAN_SB();
        MF(wf0, xf0, 1, 1) patch_piece_prepare(2 * tap + 1, chunk + 1, more); UEGAN_SB();
        MF(wf0, xf0, 1, 2) patch_piece_issue(); UEGAN_SB();
        MF(wf0, xf0, 1, 3) LDW(wf1, ws, 1, 0) LDW(wf1, ws, 1, 1) UEGAN_SB();
        MF(wf0, xf0, 2, 0) LDW(wf1, ws, 1, 2) LDW(wf1, ws, 1, 3) UEGAN_SB();
        MF(wf0, xf0, 2, 1) LDX(xf1, pcur, 1, 0) LDX(xf1, pcur, 1, 1) UEGAN_SB();
        MF(wf0, xf0, 2, 2) LDX(xf1, pcur, 1, 2) LDX(xf1, pcur, 1, 3) UEGAN_SB();
        MF(wf0, xf0, 2, 3) MF(wf0, xf0, 3, 0) MF(wf0, xf0, 3, 1) MF(wf0, xf0, 3, 2) MF(wf0, xf0, 3, 3)
        FOLDX(xf1, pcur, 1, tap)
        MF(wf1, xf1, 0, 0) set_xad(ntap, 0); UEGAN_SB();
        MF(wf1, xf1, 0, 1) set_xad(ntap, 1); UEGAN_SB();
        MF(wf1, xf1, 0, 2) set_xad(ntap, 2); UEGAN_SB();
        MF(wf1, xf1, 0, 3) set_xad(ntap, 3); UEGAN_SB();
        MF(wf1, xf1, 1, 0) LDW(wf0, wnext, 0, 0) LDW(wf0, wnext, 0, 1) UEGAN_SB();      // (slice step+1 is visible since the last barrier)
        MF(wf1, xf1, 1, 1) LDW(wf0, wnext, 0, 2) LDW(wf0, wnext, 0, 3) UEGAN_SB();
        MF(wf1, xf1, 1, 2) LDX(xf0, pnext, 0, 0) LDX(xf0, pnext, 0, 1) UEGAN_SB();
        MF(wf1, xf1, 1, 3) LDX(xf0, pnext, 0, 2) LDX(xf0, pnext, 0, 3) UEGAN_SB();
        MF(wf1, xf1, 2, 0) MF(wf1, xf1, 2, 1) MF(wf1, xf1, 2, 2) MF(wf1, xf1, 2, 3)
        MF(wf1, xf1, 3, 0) MF(wf1, xf1, 3, 1) MF(wf1, xf1, 3, 2) MF(wf1, xf1, 3, 3)
        FOLDX(xf0, pnext, 0, ntap)
      } else {
        static_assert(NI == 2 || NI == 4, "channel fragments per wave");
        // 8 + 8 MFMAs: the reads of sub-step 1 first (they are needed after 8 MFMAs), the loads behind them
        MF(wf0, xf0, 0, 0) LDW(wf1, ws, 1, 0) LDW(wf1, ws, 1, 1) UEGAN_SB();
        MF(wf0, xf0, 0, 1) LDX(xf1, pcur, 1, 0) LDX(xf1, pcur, 1, 1) UEGAN_SB();
        MF(wf0, xf0, 0, 2) LDX(xf1, pcur, 1, 2) LDX(xf1, pcur, 1, 3) UEGAN_SB();
        MF(wf0, xf0, 0, 3) stage_w_prepare(); UEGAN_SB();
        MF(wf0, xf0, 1, 0) stage_w_piece(0); UEGAN_SB();
        MF(wf0, xf0, 1, 1) patch_piece_prepare(2 * tap, chunk + 1, more); UEGAN_SB();
        MF(wf0, xf0, 1, 2) patch_piece_issue(); patch_piece_prepare(2 * tap + 1, chunk + 1, more); UEGAN_SB();
        MF(wf0, xf0, 1, 3) patch_piece_issue(); UEGAN_SB();
        MF(wf1, xf1, 0, 0) set_xad(ntap, 0); set_xad(ntap, 1); UEGAN_SB();
        MF(wf1, xf1, 0, 1) set_xad(ntap, 2); set_xad(ntap, 3); LDW(wf0, wnext, 0, 0) LDW(wf0, wnext, 0, 1) UEGAN_SB();
        MF(wf1, xf1, 0, 2) LDX(xf0, pnext, 0, 0) LDX(xf0, pnext, 0, 1) UEGAN_SB();
        MF(wf1, xf1, 0, 3) LDX(xf0, pnext, 0, 2) LDX(xf0, pnext, 0, 3) UEGAN_SB();
        MF(wf1, xf1, 1, 0) MF(wf1, xf1, 1, 1) MF(wf1, xf1, 1, 2) MF(wf1, xf1, 1, 3)
      }
      // everything older than this step's batch has landed: slice step+2, older patch pieces
      if (RPW == 2 && NI == 2 && 2 * tap >= NI_P) wait_vmcnt<NWP>();
      else wait_vmcnt<NLOAD>();
      raw_barrier();
    }
  }
#undef MF
#undef LDW
#undef LDX
#undef FOLDX
  wait_vmcnt<0>();                                     // (the dump-area loads of the last steps)

  // ---- epilogue: scale, bias, activation in fp32 -> bf16 -> through the LDS (wave-private rows of BN channels + 8 B) -> NHWC rows,
  // 16 bytes per lane, BN/8 lanes per pixel (see conv_wide_kernel); the deferred activation gradient reads its mask the same way
  unsigned char* const est = lds + wave * (WPIX * EROW);
  {
    const float slope = a.act == UEGAN_ACT_LRELU ? 0.2f : (a.act == UEGAN_ACT_RELU ? 0.f : 1.f);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f32x4 bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const f32x4*>(lds_bias + (i * 32 + 4 * lh + 8 * q) * 4);
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
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
  __builtin_amdgcn_wave_barrier();                     // (each wave reads back only what it wrote)
  {
    constexpr int LPP = BN / 8;                        // lanes per pixel (16 bytes each)
    constexpr int PPI = 64 / LPP;                      // pixels per store instruction
    const float mslope = a.mask_act == UEGAN_ACT_LRELU ? 0.2f : (a.mask_act == UEGAN_ACT_RELU ? 0.f : 1.f);
    bf16_t* out = static_cast<bf16_t*>(a.out);
    const int lc = lane % LPP, nl = n0 + lc * 8;
    const int n_store = (POOL && b >= a.n_full) ? 0 : WPIX / PPI;      // (pooled result only: the store-bound half of this epilogue is skipped)
#pragma unroll 4
    for (int it = 0; it < n_store; ++it) {
      const int rr = it * PPI + lane / LPP;            // pixel inside the wave's 4 rows x 32 columns
      const int oy = y0 + wave * RPW + (rr >> 5), ox = x0 + (rr & 31);
      const u32x2 v01 = *reinterpret_cast<const u32x2*>(est + rr * EROW + lc * 16);        // (rows are 8-byte aligned only)
      const u32x2 v23 = *reinterpret_cast<const u32x2*>(est + rr * EROW + lc * 16 + 8);
      u32x4 v = {v01.x, v01.y, v23.x, v23.y};
      if (oy >= g.OH || ox >= g.OW) continue;
      const size_t pix = ((size_t)b * g.OH + oy) * g.OW + ox;
      const size_t o = pix * a.N + nl;
      if (MASK) {
        const u32x4 m = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(a.mask) + o);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float lo = half_lo_to_f32(v[d]) * (half_lo_to_f32(m[d]) > 0.f ? 1.f : mslope);
          const float hi = half_hi_to_f32(v[d]) * (half_hi_to_f32(m[d]) > 0.f ? 1.f : mslope);
          v[d] = pack_bf16x2(lo, hi);
        }
      }
      bf16_t* dst = !a.out2 ? out + o : (nl < a.n_out1 ? out + pix * a.n_out1 + nl : static_cast<bf16_t*>(a.out2) + pix * (a.N - a.n_out1) + (nl - a.n_out1));
      *reinterpret_cast<u32x4*>(dst) = v;
    }
    if constexpr (POOL) {
      // 2x2 max-pool of the wave's 4 x 32 pixels: 2 x 16 pooled pixels (OH, OW even and y0, x0 even: a window never straddles tiles)
      bf16_t* pout = static_cast<bf16_t*>(a.pool_out);
      const int PH2 = g.OH >> 1, PW2 = g.OW >> 1;
#pragma unroll 2
      for (int it = 0; it < WPIX / 4 / PPI; ++it) {
        const int pp = it * PPI + lane / LPP;
        const int pr = pp >> 4, pc = pp & 15;
        const int py = ((y0 + wave * RPW) >> 1) + pr, px = (x0 >> 1) + pc;
        float m[8];
        uint32_t arg[8];                               // window position (dy * 2 + dx) of the first maximum (maxpool2x2_bwd_kernel's rule)
#pragma unroll
        for (int e = 0; e < 8; ++e) { m[e] = -3.0e38f; arg[e] = 0; }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int rr = (2 * pr + dy) * 32 + 2 * pc + dx;
            const u32x2 v01 = *reinterpret_cast<const u32x2*>(est + rr * EROW + lc * 16);
            const u32x2 v23 = *reinterpret_cast<const u32x2*>(est + rr * EROW + lc * 16 + 8);
            const uint32_t vv[4] = {v01.x, v01.y, v23.x, v23.y};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const float lo = half_lo_to_f32(vv[d]), hi = half_hi_to_f32(vv[d]);
              if (lo > m[2 * d]) { m[2 * d] = lo; arg[2 * d] = dy * 2 + dx; }
              if (hi > m[2 * d + 1]) { m[2 * d + 1] = hi; arg[2 * d + 1] = dy * 2 + dx; }
            }
          }
        if (py >= PH2 || px >= PW2) continue;
        const size_t po = (((size_t)b * PH2 + py) * PW2 + px) * a.N + nl;
        u32x4 o4 = {pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7])};
        *reinterpret_cast<u32x4*>(pout + po) = o4;
        if (a.pool_idx && b < a.n_idx) {
          const u32x2 a8 = {arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24), arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24)};
          *reinterpret_cast<u32x2*>(static_cast<unsigned char*>(a.pool_idx) + po) = a8;
        }
      }
    }
  }
}

// 0 / error code when the launch was taken, 1 when the problem is not one of this kernel's
// interior: called by conv_interior_run with a.rect_* set -- the image-free rectangle of a reflection-padded data gradient (two destinations allowed)
int conv_tall_run(ConvArgs& a, int dtype, hipStream_t s, bool interior) {
  const ConvGeom& g = a.g;
  const int min_grid = g_tuning[UEGAN_TUNE_TALL_MIN_GRID];
  if (min_grid < 0) return 1;
  if (dtype != UEGAN_BF16 || g.stride != 1 || g.KH != 3 || g.KW != 3 || a.frame != 0) return 1;
  if (g.C1 % 32 || g.C2 % 32 || g.C > 1024 || (a.N != 64 && a.N % 128) || g.OW < 32 || g.OH < 16) return 1;
  // reflection-padded data gradient (pad 1), every tile in this launch: the mirrored images folded into the pixel operand (MODE 2, 128-channel blocks)
  const bool refl = !interior && g.mode == 1 && g.pad_mode == UEGAN_PAD_REFLECT && g.pad == 1 && !g.C2 && !a.mask && a.N % 128 == 0 &&
                    (!a.out2 || a.n_out1 % 8 == 0) && g.OH >= 4 && g.OW >= 4 && g_tuning[UEGAN_TUNE_TALL_REFLECT] != 0;
  if (!interior && !refl && (a.out2 || (g.mode == 1 && (g.C2 || (g.pad_mode == UEGAN_PAD_REFLECT && g.pad != 0))))) return 1;       // mirrored images: conv_interior_run / conv_patch MODE 2
  if (interior && (g.mode != 1 || g.C2 || a.mask || (a.out2 && a.n_out1 % 8))) return 1;
  auto simple = [](int act) { return act == UEGAN_ACT_NONE || act == UEGAN_ACT_LRELU || act == UEGAN_ACT_RELU; };
  if (!simple(a.act) || (a.mask && !simple(a.mask_act))) return 1;
  // 128-channel blocks: 8-row tiles, two blocks per CU (RPW = 2) while the K loop is short (< 512 input channels: the prologue and epilogue
  // are then 25 % and more of a tile, and the second block hides them: -5 ... -31 % per layer); at 512 input channels the 16-row tile's
  // lower LDS traffic per MFMA wins by 2 ... 5 %.  UEGAN_TUNE_TALL_RPW = 2 / 4 forces one of them (A/B, tests).
  const int rpw_knob = g_tuning[UEGAN_TUNE_TALL_RPW];
  bool rpw2 = rpw_knob == 2 || (rpw_knob != 4 && g.C < 512);      // (64-channel blocks: 61 KB of LDS and 4 accumulators per wave on 8-row tiles)
  const int nb_n = a.N == 64 ? 1 : a.N / 128;
  {      // the minimum grid counts 16-row tiles (one block per CU); a map too small for that still fills the chip with 8-row tiles (VGG conv5_1 at batch 16)
    const int ntx = interior ? (a.rect_x1 - a.rect_x0) / 32 : (g.OW + 31) / 32;
    const int nty16 = interior ? (a.rect_y1 - a.rect_y0) / 16 : (g.OH + 15) / 16, nty8 = interior ? (a.rect_y1 - a.rect_y0) / 8 : (g.OH + 7) / 8;
    if (g.B * nty16 * ntx * nb_n < min_grid) {
      if (rpw_knob == 4 || g.B * nty8 * ntx * nb_n < min_grid) return 1;
      rpw2 = true;
    }
  }
  const int th = rpw2 ? 8 : 16;
  a.nty = interior ? (a.rect_y1 - a.rect_y0) / th : (g.OH + th - 1) / th;
  a.ntx = interior ? (a.rect_x1 - a.rect_x0) / 32 : (g.OW + 31) / 32;
  const int gm = g.B * a.nty * a.ntx;
  const bool pool = a.pool_out && g.mode == 0 && !a.mask && g.OH % 2 == 0 && g.OW % 2 == 0;
  const double rows = interior ? (double)g.B * (a.rect_y1 - a.rect_y0) * (a.rect_x1 - a.rect_x0) : (g.mode == 0 ? (double)g.B * g.OH * g.OW : (double)g.B * g.IH * g.IW);
  ProfScope prof(prof_key(7, true, a.N == 64 ? 64 : 128, 3, refl ? 2 : g.mode, 16, !pool), 2.0 * rows * a.N * (double)(9 * g.C), s,
                 2.0 * (rows * a.N + (interior ? rows : (double)g.B * g.IH * g.IW) * g.C));
  const dim3 grid(gm, a.N == 64 ? 1 : a.N / 128), block(256);
  a.xcd_map = (grid.y > 1 && gm % 8 == 0) ? 1 : 0;
#define UEGAN_TALL(NI, RPW)                                                                                          \
  do {                                                                                                               \
    if (g.mode == 0 && pool) hipLaunchKernelGGL((conv_tall_kernel<NI, 0, false, true, RPW>), grid, block, 0, s, a);  \
    else if (g.mode == 0) hipLaunchKernelGGL((conv_tall_kernel<NI, 0, false, false, RPW>), grid, block, 0, s, a);    \
    else if (a.mask) hipLaunchKernelGGL((conv_tall_kernel<NI, 1, true, false, RPW>), grid, block, 0, s, a);          \
    else hipLaunchKernelGGL((conv_tall_kernel<NI, 1, false, false, RPW>), grid, block, 0, s, a);                     \
  } while (0)
  if (refl) {
    if (rpw2) hipLaunchKernelGGL((conv_tall_kernel<4, 2, false, false, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_tall_kernel<4, 2, false, false, 4>), grid, block, 0, s, a);
  }
  else if (a.N == 64) { if (rpw2) UEGAN_TALL(2, 2); else UEGAN_TALL(2, 4); }
  else if (rpw2) UEGAN_TALL(4, 2);
  else UEGAN_TALL(4, 4);
#undef UEGAN_TALL
  if (g.mode == 0 && pool) a.pool_done = 1;
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

// 0 / error code when the launch was taken, 1 when the problem is not one of this kernel's (the caller falls through to conv_patch)
int conv_wide_run(ConvArgs& a, int dtype, hipStream_t s, bool interior) {
  const ConvGeom& g = a.g;
  const int min_grid = g_tuning[UEGAN_TUNE_WIDE_MIN_GRID];      // fewer blocks than CUs: the smaller tiles of conv_patch cover the chip better
  if (min_grid < 0) return 1;                        // (uegan_set_tuning: < 0 switches this kernel off, the tests set 1 to reach it on small maps)
  if (dtype != UEGAN_BF16 || g.stride != 1 || g.KH != 3 || g.KW != 3 || a.frame != 0) return 1;
  if (g.C % 64 || g.C > 1024 || g.C2 || a.N % 256 || g.OW < 32 || g.OH < 8) return 1;
  if (!interior && (a.out2 || (g.mode == 1 && g.pad_mode == UEGAN_PAD_REFLECT && g.pad != 0))) return 1;       // mirrored images: conv_interior_run / conv_patch MODE 2
  if (interior && (g.mode != 1 || a.mask || (a.out2 && a.n_out1 % 8))) return 1;
  auto simple = [](int act) { return act == UEGAN_ACT_NONE || act == UEGAN_ACT_LRELU || act == UEGAN_ACT_RELU; };
  if (!simple(a.act) || (a.mask && !simple(a.mask_act))) return 1;
  // (conv_tall_kernel goes first: with N / 128 channel blocks per tile it moves a third less weight + patch data per MFMA -- four waves share a
  // slice -- and with its 8-row tiles it measured equal or faster on every layer this kernel used to take: conv5_1 at batch 32 0.115 vs 0.120 ms,
  // conv4_1's data gradient 0.120 vs 0.126, dec2's interior 0.218 vs 0.249.  What is left here are launches too small for its minimum grid.)
  a.nty = interior ? (a.rect_y1 - a.rect_y0) / 8 : (g.OH + 7) / 8;
  a.ntx = interior ? (a.rect_x1 - a.rect_x0) / 32 : (g.OW + 31) / 32;
  const int gm = g.B * a.nty * a.ntx;
  if (gm * (a.N / 256) < min_grid) return 1;
  const double rows = interior ? (double)g.B * (a.rect_y1 - a.rect_y0) * (a.rect_x1 - a.rect_x0) : (g.mode == 0 ? (double)g.B * g.OH * g.OW : (double)g.B * g.IH * g.IW);
  ProfScope prof(prof_key(7, true, 256, 3, g.mode, 8, true), 2.0 * rows * a.N * (double)(9 * g.C), s,
                 2.0 * (rows * a.N + (interior ? rows : (double)g.B * g.IH * g.IW) * g.C));
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

// Reflection-padded stride-1 3x3 data gradient (G.dec1 - dec3: two destinations): only pixels within `pad` of a border receive mirrored images.
// The rectangle of 16-row x 32-column tiles without such a pixel runs image-free on conv_wide_kernel / conv_tall_kernel; the caller then
// launches the patch kernel's mirrored-image variant over the frame around it (a.border_only).  0: interior launched, 1: not taken.
int conv_interior_run(ConvArgs& a, int dtype, hipStream_t s) {
  const ConvGeom& g = a.g;
  if (dtype != UEGAN_BF16 || g.mode != 1 || g.stride != 1 || g.KH != 3 || g.KW != 3 || g.pad_mode != UEGAN_PAD_REFLECT || g.pad != 1 || a.frame != 0 || a.mask)
    return 1;
  // pixels 1 .. pad and n-1-pad .. n-2 of an axis have a mirrored image: the clean tiles are 1 .. floor((n - 1 - pad) / T) - 1
  // (the rectangle starts one patch-kernel tile -- 16 pixels -- inside the border on both axes and is whole 16 x 32 tiles of these kernels from there)
  const int y0 = 16, y1 = (g.OH - 1 - g.pad) / 16 * 16, x0 = 16, x1 = 16 + (g.OW - 1 - g.pad - 16) / 32 * 32;
  if (y1 <= y0 || x1 <= x0 || 2LL * (y1 - y0) * (x1 - x0) < (long long)g.OH * g.OW) return 1;      // (small maps: the frame is most of it)
  a.rect_y0 = y0; a.rect_y1 = y1; a.rect_x0 = x0; a.rect_x1 = x1;
  int rc = conv_tall_run(a, dtype, s, true);
  if (rc == 1) rc = conv_wide_run(a, dtype, s, true);
  if (rc != UEGAN_OK) { a.rect_y0 = a.rect_y1 = a.rect_x0 = a.rect_x1 = 0; return rc; }
  a.border_only = 1;
  return UEGAN_OK;
}

}  // namespace uegan
