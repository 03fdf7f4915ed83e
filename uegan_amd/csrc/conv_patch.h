// Patch-resident gather-GEMM kernel and its launchers (instantiated per dtype / kernel-size set by conv_patch_*.hip so that the
// instantiations compile in parallel).  Internal (non-ABI).
#pragma once
#include "conv_core.h"

#ifndef UEGAN_PATCH_PIPE
#define UEGAN_PATCH_PIPE 1
#endif

namespace uegan {

// ----------------------------------------------------------------------------------------------------
// Patch-resident gather-GEMM for stride-1 KSxKS convolutions (forward, and dgrad incl. its reflected images).
//
// The generic kernel above re-gathers the 128-pixel operand tile from L2 for every tap (KS*KS times).  Here the
// (8+KS-1) x (16+KS-1) pixel patch a tile needs is staged ONCE per 64-channel chunk and the taps walk over it in LDS:
// a tap is just a different LDS row offset for the pixel fragments, so per tap only the BN x 128 B weight slice moves.
// L2->LDS traffic per MFMA drops ~1.7x for 128-wide channel tiles and >6x for the narrow heads (Cout 1/3).
//
// Per axis and image the gather is src = v0 + patch_index, patch_index = (ri ? T-1-i : i) + (rt ? KS-1-t : t):
//   forward            ri=0 rt=0  v0 = o0 - pad                     source row = pad_map(src)
//   dgrad, image 0     ri=0 rt=1  v0 = o0 + pad - (KS-1)            source row = src if 0 <= src < n
//   dgrad, mirror 0    ri=1 rt=1  v0 = -(o0+T-1) + pad - (KS-1)     (pixels 1..pad only)
//   dgrad, mirror n-1  ri=1 rt=1  v0 = 2(n-1) - (o0+T-1) + pad - (KS-1)   (pixels n-1-pad..n-2 only)
// ----------------------------------------------------------------------------------------------------
// MODE: 0 forward (either padding), 1 dgrad with zero padding (no images), 2 dgrad with reflection padding (images)
// TH: tile height in pixels (8 -> 128-pixel tile, 4 waves; 16 -> 256-pixel tile, 8 waves: every weight slice then feeds
//     twice the MFMA work, which is what a latency-bound L2->LDS stream needs); NWBUF: weight ring depth (2 or 3)
// POOL (forward): the epilogue also writes the 2x2 max-pool of its tile (rows pair up inside a wave's row fragments, columns across lane pairs)
template <typename T, int BN, int WARPS_M, int WARPS_N, int KS, int MODE, int TH, int NWBUF, int TPS = 1, bool ONEP = false, bool MASK = false, bool POOL = false,
          bool SPLITK = false>      // SPLITK: blockIdx.z = part of the K loop (ConvArgs::kws), fp32 partial sums instead of the epilogue
__global__ void __launch_bounds__(64 * WARPS_M * WARPS_N) conv_patch_kernel(ConvArgs a) {
  // ONEP: a single patch buffer, for layers with one 64-channel chunk (no next phase to prefetch): the block then fits twice per CU
  // TPS = taps per step (per barrier): 2 for the 64-channel blocks, whose steps are otherwise too short for their fixed cost
  // KS = taps per axis the patch is sized for: the kernel size for stride 1; for a stride-2 dgrad each parity class
  // of input pixels sees a stride-1 sub-convolution with ceil(K/2) or floor(K/2) taps per axis (KS = (K+1)/2)
  constexpr int ROWB = CONV_ROWB, TW = CONV_TW, BM = TH * TW, NWAVES = WARPS_M * WARPS_N;
  constexpr int EPC = DT<T>::EPC;
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int NPG = (PH * PW + 7) / 8;             // 8-row groups of the patch
  constexpr int NI_P = (NPG + NWAVES - 1) / NWAVES;  // patch staging instructions per thread
  constexpr int WROWG = BN / 8;
  constexpr int NI_W = (WROWG + NWAVES - 1) / NWAVES;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int NCHUNK = Mma<T>::NCHUNK;
  constexpr int NSUB = BK / 32;
  constexpr int PBUFB = NPG * 8 * ROWB, WSLICE = BN * ROWB, WBUFB = TPS * WSLICE;
  constexpr bool DGRAD = MODE != 0, IMAGES = MODE == 2;
  static_assert((NWAVES == 4 || NWAVES == 8) && TM >= 1 && TN >= 1 && (NWBUF == 2 || NWBUF == 3), "tile");

  __shared__ __attribute__((aligned(16))) unsigned char lds[(ONEP ? 1 : 2) * PBUFB + NWBUF * WBUFB];
  unsigned char* const lds_w = lds + (ONEP ? 1 : 2) * PBUFB;
  __shared__ int img_par[9][8];     // MODE 2: per mirrored image of this tile {tyl, tyh, txl, txh, dvy, dvx, riy, rix} (block-uniform)

  const ConvGeom& g = a.g;
  const T* in1 = static_cast<const T*>(a.in1);
  const T* in2 = static_cast<const T*>(a.in2);
  const T* w = static_cast<const T*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int n0 = blockIdx.y * BN;
  const bool refl = g.pad_mode == UEGAN_PAD_REFLECT;
  const int sub = DGRAD ? g.stride : 1;             // pixel stride inside the tile (parity classes of a stride-2 dgrad)
  int t = blockIdx.x;
  int tile_x, tile_y;
  if (a.frame == 0) {
    tile_x = t % a.ntx; t /= a.ntx;
    tile_y = t % a.nty; t /= a.nty;
  } else if (a.frame == 2) {                        // only the tile rectangle
    const int rw = a.fx1 - a.fx0, rh = a.fy1 - a.fy0;
    tile_x = a.fx0 + t % rw; t /= rw;
    tile_y = a.fy0 + t % rh; t /= rh;
  } else {                                          // only the border tiles: top band, bottom band, side columns
    const int top = a.fy0 * a.ntx, bot = (a.nty - a.fy1) * a.ntx, side = a.fx0 + (a.ntx - a.fx1);
    const int per = top + bot + (a.fy1 - a.fy0) * side;
    int i = t % per;
    t /= per;
    if (i < top) { tile_y = i / a.ntx; tile_x = i - tile_y * a.ntx; }
    else if (i < top + bot) { i -= top; const int r = i / a.ntx; tile_y = a.fy1 + r; tile_x = i - r * a.ntx; }
    else { i -= top + bot; const int r = i / side, k = i - r * side; tile_y = a.fy0 + r; tile_x = k < a.fx0 ? k : a.fx1 + (k - a.fx0); }
  }
  const int pcls = t % (sub * sub);
  const int b = t / (sub * sub);
  const int py = pcls / sub, px = pcls - py * sub;
  const int y0s = tile_y * TH, x0s = tile_x * TW;   // tile origin on the (sub-)grid
  // taps of this parity class (all taps when sub == 1)
  const int ty0 = DGRAD ? (py + g.pad) % sub : 0, tx0 = DGRAD ? (px + g.pad) % sub : 0;
  const int nty_t = (g.KH - ty0 + sub - 1) / sub, ntx_t = (g.KW - tx0 + sub - 1) / sub;
  // actual coordinate range of the tile
  const int y_lo = py + sub * y0s, y_hi = py + sub * (y0s + TH - 1);
  const int x_lo = px + sub * x0s, x_hi = px + sub * (x0s + TW - 1);

  // image list (block-uniform, analytic), 4 bits per entry
  unsigned long long imgs = 0;
  int nimg = 0;
  if (IMAGES) {
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
  const int chunk0 = SPLITK ? (int)blockIdx.z * a.kchunks : 0;      // first 64-channel chunk of this block's part of the K loop
  const int nchunk = SPLITK ? (((g.C + BK - 1) / BK - chunk0) < a.kchunks ? ((g.C + BK - 1) / BK - chunk0) : a.kchunks) : (g.C + BK - 1) / BK;
  // Phases are the 64-channel chunks of the DIRECT image.  The mirrored images of a reflection-padded dgrad read the same
  // source pixels the direct image already staged (they only reach a few rows/columns across the border), so they ride
  // along as extra MFMAs on the current patch and weight slice (below) instead of extra phases with their own patch loads.
  (void)nimg;

  // live tap range of an image along one axis (class-local tap index t', true tap t = t0 + sub*t'): mirrored images only
  // see the taps that reach across the border.  With o the true coordinate, in_n the gathered tensor's extent:
  //   mirror 0   : sub*src = -o + pad - t >= 0            for some o >= max(1, lo)   <=>  t <= pad - max(1, lo)
  //   mirror n-1 : sub*src = 2(n-1) - o + pad - t <= sub*(in_n-1)  for some o <= min(n-2, hi)
  //                                                                               <=>  t >= 2(n-1) + pad - sub*(in_n-1) - min(n-2, hi)
  // (supersets are safe: rows without the image are masked and out-of-range sources gather zero)
  auto tap_range = [&](int img, int lo, int hi, int n, int in_n, int t0, int nt, int& t_lo, int& t_hi) {
    t_lo = 0; t_hi = nt - 1;
    if (IMAGES) {
      if (img == 1) {
        const int tmax = g.pad - (lo > 1 ? lo : 1) - t0;                  // t' <= floor(tmax / sub)
        const int m = tmax >= 0 ? tmax / sub : -1;
        if (m < t_hi) t_hi = m;
      }
      if (img == 2) {
        const int tmin = 2 * (n - 1) + g.pad - sub * (in_n - 1) - ((n - 2) < hi ? (n - 2) : hi) - t0;   // t' >= ceil(tmin / sub)
        const int m = tmin > 0 ? (tmin + sub - 1) / sub : 0;
        if (m > t_lo) t_lo = m;
      }
    }
  };

  // staging role (identical LDS row/position scheme to conv_gemm_kernel)
  const int srow = lane >> 3, spos = lane & 7;
  const int sdc = spos ^ swz128(lane >> 3);      // (row = 8 * group + (lane >> 3): swz128 only looks at bits 1-2 of the row)
  const int c_in_chunk = sdc * EPC;

  // per-axis gather parameters: src = v0 + patch_index, patch_index = (ri ? T-1-i : i) + (dgrad ? nt-1-t' : t')
  auto axis = [&](int img, int o0s, int Tn, int n, int pcl, int t0, int nt, int& v0, bool& ri) {
    if (!DGRAD) { v0 = o0s - g.pad; ri = false; return; }
    const int c_dir = (pcl + g.pad - t0) / sub;      // (o + pad - t)/sub      = i' - t' + c_dir
    const int c_mir = (g.pad - pcl - t0) / sub;      // (-o + pad - t)/sub     = -i' - t' + c_mir   (exact: numerator is even)
    if (img == 0) { v0 = o0s + c_dir - (nt - 1); ri = false; }
    else if (img == 1) { v0 = -(o0s + Tn - 1) + c_mir - (nt - 1); ri = true; }
    else { v0 = -(o0s + Tn - 1) + c_mir + 2 * (n - 1) / sub - (nt - 1); ri = true; }
  };
  // pixel offsets of my patch rows for one image (-1: contributes zero)
  int poff[NI_P];
  auto setup_patch_rows = [&](int q) {
    const int iy = q / 3, ix = q - iy * 3;
    int vy0, vx0; bool r0, r1;
    axis(iy, y0s, TH, g.OH, py, ty0, nty_t, vy0, r0);
    axis(ix, x0s, TW, g.OW, px, tx0, ntx_t, vx0, r1);
#pragma unroll
    for (int ii = 0; ii < NI_P; ++ii) {
      const int pr = (ii * NWAVES + wave) * 8 + srow;
      int off = -1;
      if (pr < PH * PW) {
        const int piy = pr / PW, pix = pr - piy * PW;
        int sy = vy0 + piy, sx = vx0 + pix;
        if (!DGRAD && refl) {     // forward + reflection (tiles may overhang the image: out-of-range mirrors gather zero)
          sy = reflect_idx(sy, g.IH);
          sx = reflect_idx(sx, g.IW);
        }
        if (sy < 0 || sy >= g.IH) sy = -1;
        if (sx < 0 || sx >= g.IW) sx = -1;
        if (sy >= 0 && sx >= 0) off = (b * g.IH + sy) * g.IW + sx;
      }
      poff[ii] = off;
    }
  };
  auto stage_patch = [&](unsigned char* buf, int chunk) {
    const int cc = (chunk0 + chunk) * BK + c_in_chunk;
#pragma unroll
    for (int ii = 0; ii < NI_P; ++ii) {
      const int rg = ii * NWAVES + wave;
      if (rg < NPG) {
        const void* src = g_zero16;
        if (poff[ii] >= 0 && cc < g.C)
          src = (cc < g.C1) ? (const void*)(in1 + (size_t)poff[ii] * g.C1 + cc) : (const void*)(in2 + (size_t)poff[ii] * g.C2 + (cc - g.C1));
        glds16(src, buf + rg * 8 * ROWB);
      }
    }
  };
  // weight slice staging: the per-lane part of the source address (row n, channel offset inside the chunk) never changes, so it is
  // computed once; a step only adds the block-uniform (tap, chunk) offset
  const T* wbase[NI_W];
#pragma unroll
  for (int i = 0; i < NI_W; ++i) {
    const int rg = i * NWAVES + wave;
    const int n = n0 + rg * 8 + srow;
    wbase[i] = (rg < WROWG && n < a.N && c_in_chunk < g.C) ? w + (size_t)n * a.Kp + c_in_chunk : nullptr;
  }
  auto stage_w = [&](unsigned char* buf, int chunk, int tyq, int txq) {      // (tyq, txq): class-local first tap of the step
#pragma unroll
    for (int u = 0; u < TPS; ++u) {
      // u-th tap of the step; beyond the last tap of the chunk the slice is loaded from the zero page (same load count)
      const bool tv = tyq < nty_t;
      const int wtap = (ty0 + sub * tyq) * g.KW + (tx0 + sub * txq);
      const int off = wtap * g.C + (chunk0 + chunk) * BK;                                 // (the patch kernel runs only when C % BK == 0)
#pragma unroll
      for (int i = 0; i < NI_W; ++i) {
        const int rg = i * NWAVES + wave;
        if (rg < WROWG) {
          const void* src = (tv && wbase[i]) ? (const void*)(wbase[i] + off) : (const void*)g_zero16;
          glds16(src, buf + u * WSLICE + rg * 8 * ROWB);
        }
      }
      if (++txq == ntx_t) { txq = 0; ++tyq; }
    }
  };
  // my wave's vmcnt budget: the number of direct-to-LDS loads of ONE weight slice (what may stay in flight at a barrier)
  auto wait_all_but_one_slice = [&]() {
    if (NWBUF == 2) wait_vmcnt<0>();                      // ring of 2: the slice of the next step is issued after the barrier
    else if (WROWG % NWAVES == 0) wait_vmcnt<NI_W * TPS>();     // every wave issues exactly NI_W loads per slice
    else if (wave < WROWG) wait_vmcnt<TPS>();
    else wait_vmcnt<0>();
  };

  // schedule: step s = (chunk, ty, tx) in chunk-major order over the class's nty_t x ntx_t taps; plain running counters (the
  // earlier per-step cursor objects with tap rectangles cost ~250 scalar instructions per step, for 32 MFMAs)
  const int nsteps = nchunk * ((nty_t * ntx_t + TPS - 1) / TPS);
  auto advance = [&](int& chunk, int& ty, int& tx) {           // to the first tap of the next step
#pragma unroll
    for (int u = 0; u < TPS; ++u) {
      if (ty < nty_t && ++tx == ntx_t) { tx = 0; ++ty; }
    }
    if (ty >= nty_t) { ty = 0; tx = 0; ++chunk; }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int oxl = px + sub * (x0s + fr);         // my pixel column (all fragments)

  int c_chunk = 0, c_ty = 0, c_tx = 0;           // compute position
  int w_chunk = 0, w_ty = 0, w_tx = 0, w_step = 0;      // next weight slice to stage (runs NWBUF-1 steps ahead)
  if (IMAGES && nimg > 1) {        // parameters of the mirrored images, once per tile (thread e fills entry e)
    if (tid >= 1 && tid < nimg) {
      const int qi = (int)((imgs >> (4 * tid)) & 15ull);
      const int iy = qi / 3, ix = qi - iy * 3;
      int tyl, tyh, txl, txh, vy, vx, vyd, vxd;
      bool r0, r1, rd;
      tap_range(iy, y_lo, y_hi, g.OH, g.IH, ty0, nty_t, tyl, tyh);
      tap_range(ix, x_lo, x_hi, g.OW, g.IW, tx0, ntx_t, txl, txh);
      axis(iy, y0s, TH, g.OH, py, ty0, nty_t, vy, r0);
      axis(ix, x0s, TW, g.OW, px, tx0, ntx_t, vx, r1);
      axis(0, y0s, TH, g.OH, py, ty0, nty_t, vyd, rd);
      axis(0, x0s, TW, g.OW, px, tx0, ntx_t, vxd, rd);
      int* o = img_par[tid];
      o[0] = tyl; o[1] = tyh; o[2] = txl; o[3] = txh; o[4] = vy - vyd; o[5] = vx - vxd; o[6] = r0 ? 1 : 0; o[7] = r1 ? 1 : 0;
    }
    __syncthreads();
  }
  // the first NHI mirrored images' parameters in (scalar) registers for the whole tile: the K loop below visits every image twice per tap
  // step, and eight LDS reads + readfirstlanes per visit sat in front of its fragment addresses
  constexpr int NHI = 3;
  int hp[NHI][8];
#pragma unroll
  for (int e = 0; e < NHI; ++e)
#pragma unroll
    for (int k = 0; k < 8; ++k) hp[e][k] = (IMAGES && e + 1 < nimg) ? __builtin_amdgcn_readfirstlane(img_par[e + 1][k]) : 0;
  int pbuf = 0;                    // patch buffer of the phase being computed (toggles per LIVE phase)
  bool phase_start = true;         // the compute cursor is on the first step of its phase
  // prologue: patch of the first phase, weight slices of steps 0 and 1
  if (nsteps > 0) {
    setup_patch_rows(0);
    stage_patch(lds, 0);
    stage_w(lds_w, w_chunk, w_ty, w_tx);
    advance(w_chunk, w_ty, w_tx); ++w_step;
    if (NWBUF == 3 && w_step < nsteps) {
      stage_w(lds_w + WBUFB, w_chunk, w_ty, w_tx);
      advance(w_chunk, w_ty, w_tx); ++w_step;
    }
  }
  int wad[TN];             // weight fragment byte offsets inside a slice (step independent)
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int row = wn * WTN + i * 16 + fr;
    wad[i] = row * ROWB + ((fg ^ swz128(row)) << 4);
  }
  int slot = 0;            // weight ring slot of the step being computed
  for (int sidx = 0; sidx < nsteps; ++sidx) {
    // step s: slice s (and anything older) must have landed; slice s+1, the most recent loads, may stay in flight
    if (sidx + 1 == nsteps) wait_vmcnt<0>(); else wait_all_but_one_slice();
    raw_barrier();
    // issue order matters for the vmcnt accounting: first the NEXT phase's patch (once, on the first step of the
    // current phase; its buffer was last read one phase ago), then weight slice s+NWBUF-1 (its ring slot was read at step s-1)
    if (!ONEP && phase_start && c_chunk + 1 < nchunk) stage_patch(lds + (pbuf ^ 1) * PBUFB, c_chunk + 1);      // (same patch rows for every chunk)
    if (w_step < nsteps) {
      const int wslot = slot == 0 ? NWBUF - 1 : slot - 1;
      stage_w(lds_w + wslot * WBUFB, w_chunk, w_ty, w_tx);
      advance(w_chunk, w_ty, w_tx); ++w_step;
    }
    if (ONEP && phase_start && c_chunk > 0) {
      // single patch buffer, several chunks: the next chunk's patch can only be loaded once every wave is past the previous
      // chunk's last tap (the barrier above); its latency is exposed once per chunk and covered by the CU's other block
      stage_patch(lds, c_chunk);
      wait_vmcnt<0>();
      raw_barrier();
    }
    // compute step s: its TPS taps one after the other (all of them staged behind the same barrier)
    const unsigned char* pcur = lds + (ONEP ? 0 : pbuf) * PBUFB;
    int u_ty = c_ty, u_tx = c_tx;
#pragma unroll
   for (int u = 0; u < TPS; ++u) {
    if (u_ty >= nty_t) break;                      // odd tap count: the last step of a chunk has one tap less
    const unsigned char* wcur = lds_w + slot * WBUFB + u * WSLICE;
    const int pty = DGRAD ? nty_t - 1 - u_ty : u_ty, ptx = DGRAD ? ntx_t - 1 - u_tx : u_tx;
    const int pix = fr + ptx;
    // fragment addresses once per step: chunk q = 4*(ksub or c) + fg only flips bit 2 of the swizzled chunk index, i.e. XORs 64
    // into the byte address, so the second half of the K step costs one XOR per fragment instead of the whole swizzle again
    int xad[TM];
    const int tapoff = pty * PW + pix;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int pr = (wm * (TH / WARPS_M) + j) * PW + tapoff;      // patch pixel of this fragment's lane
      xad[j] = pr * ROWB + ((fg ^ swz128(pr)) << 4);
    }
    // Image-free kernels with >= 4 channel fragments per wave: the K step as a software pipeline over pairs of weight fragments --
    // the LDS reads of the NEXT pair (at the end of a half step: the next half's pixel fragments too) are issued before the MFMAs
    // of the current pair and pinned there, so the compiler's waits become counted (lgkmcnt(n)) instead of four full LDS round
    // trips per 32 MFMAs with the matrix pipe idle behind each (the ISA of the un-pipelined loop: read, lgkmcnt(0), 8 MFMAs, ...)
    // (not under MASK: its epilogue's extra live registers push the pipelined loop over the 256-VGPR budget -- 95-140 spilled registers)
    constexpr bool PIPE = UEGAN_PATCH_PIPE && !IMAGES && !MASK && TN >= 4 && TN % 4 == 0;
    if constexpr (PIPE) {
      constexpr int WG = 2, NG = TN / WG;
      u32x4 xf[2][TM][NCHUNK], wf[2][WG][NCHUNK];
      auto ld_x = [&](int ksub, int buf) {
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int c = 0; c < NCHUNK; ++c)
            xf[buf][j][c] = *reinterpret_cast<const u32x4*>(pcur + (xad[j] ^ ((ksub + c * (NCHUNK - 1)) << 6)));
      };
      auto ld_w = [&](int ksub, int grp, int buf) {
#pragma unroll
        for (int i = 0; i < WG; ++i)
#pragma unroll
          for (int c = 0; c < NCHUNK; ++c)
            wf[buf][i][c] = *reinterpret_cast<const u32x4*>(wcur + (wad[grp * WG + i] ^ ((ksub + c * (NCHUNK - 1)) << 6)));
      };
      ld_x(0, 0);
      ld_w(0, 0, 0);
#pragma unroll
      for (int ksub = 0; ksub < NSUB; ++ksub) {
#pragma unroll
        for (int gp = 0; gp < NG; ++gp) {
          const int cur = (ksub * NG + gp) & 1;
          if (gp + 1 < NG) ld_w(ksub, gp + 1, cur ^ 1);
          else if (ksub + 1 < NSUB) { ld_x(ksub + 1, (ksub + 1) & 1); ld_w(ksub + 1, 0, cur ^ 1); }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < WG; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) Mma<T>::step(wf[cur][i], xf[ksub & 1][j], acc[gp * WG + i][j]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
    for (int ksub = 0; ksub < NSUB; ++ksub) {
      u32x4 xf[TM][NCHUNK], wf[TN][NCHUNK];
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
          xf[j][c] = *reinterpret_cast<const u32x4*>(pcur + (xad[j] ^ ((ksub + c * (NCHUNK - 1)) << 6)));
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
          wf[i][c] = *reinterpret_cast<const u32x4*>(wcur + (wad[i] ^ ((ksub + c * (NCHUNK - 1)) << 6)));
      if (IMAGES && nimg > 1) {
        // mirrored images of this tile.  The adjoint of the reflection padding gives a border pixel, for tap t, the term w[t] * dz[mirrored
        // source] on top of w[t] * dz[direct source] -- the SAME weight fragment -- so the images are folded into the PIXEL operand:
        // xf[j] += dz[mirrored source] (fp32 add, rounded once to the storage format; the masked lanes add zero, which is exact), re-read
        // from the direct patch at the mirrored coordinates, and the MFMAs below run once.  (Rounds 1-4 issued TN extra MFMAs per image and
        // row fragment on a masked operand: +110 % on border tiles, and on the deep layers' small class grids every tile is one.)
        // A y-mirror only exists for <= pad rows of the tile (row fragments without it are skipped, block-uniformly), an x-mirror for
        // <= pad columns (other lanes masked).
        auto one_image = [&](int qi, int tyl, int tyh, int txl, int txh, int dvy, int dvx, bool r0, bool r1) {
          const int iy = qi / 3, ix = qi - iy * 3;
          if (u_ty < tyl || u_ty > tyh || u_tx < txl || u_tx > txh) return;
          const int pixm = dvx + (r1 ? TW - 1 - fr : fr) + ptx;              // column in the direct patch (per lane)
          const bool xok = has_image(g, oxl, ix, g.OW) && pixm >= 0 && pixm < PW;
          const uint32_t m = xok ? 0xffffffffu : 0u;
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            const int i = wm * (TH / WARPS_M) + j;
            if (!has_image(g, py + sub * (y0s + i), iy, g.OH)) continue;
            const int piym = dvy + (r0 ? TH - 1 - i : i) + pty;
            if (piym < 0 || piym >= PH) continue;
            const int pr = piym * PW + (xok ? pixm : 0);
            u32x4 xm[NCHUNK];
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) {
              const int q = ksub * 4 + c * 4 * (NCHUNK - 1) + fg;
              const u32x4 v = *reinterpret_cast<const u32x4*>(pcur + pr * ROWB + ((q ^ swz128(pr)) << 4));
              xm[c] = v & u32x4{m, m, m, m};
            }
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) xf[j][c] = add_frag<T>(xf[j][c], xm[c]);
          }
        };
#pragma unroll
        for (int e = 0; e < NHI; ++e)
          if (e + 1 < nimg)
            one_image((int)((imgs >> (4 * (e + 1))) & 15ull), hp[e][0], hp[e][1], hp[e][2], hp[e][3], hp[e][4], hp[e][5], hp[e][6] != 0, hp[e][7] != 0);
        for (int e = NHI + 1; e < nimg; ++e)       // (corner tiles of tiny maps: up to 8 images)
          one_image((int)((imgs >> (4 * e)) & 15ull), __builtin_amdgcn_readfirstlane(img_par[e][0]), __builtin_amdgcn_readfirstlane(img_par[e][1]),
                    __builtin_amdgcn_readfirstlane(img_par[e][2]), __builtin_amdgcn_readfirstlane(img_par[e][3]),
                    __builtin_amdgcn_readfirstlane(img_par[e][4]), __builtin_amdgcn_readfirstlane(img_par[e][5]),
                    __builtin_amdgcn_readfirstlane(img_par[e][6]) != 0, __builtin_amdgcn_readfirstlane(img_par[e][7]) != 0);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mma<T>::step(wf[i], xf[j], acc[i][j]);
    }
    }      // (!PIPE)
    if (++u_tx == ntx_t) { u_tx = 0; ++u_ty; }
   }
    {
      const int chunk_before = c_chunk;
      advance(c_chunk, c_ty, c_tx);
      phase_start = c_chunk != chunk_before;
      if (phase_start) pbuf ^= 1;
    }
    slot = slot + 1 == NWBUF ? 0 : slot + 1;
  }

  if constexpr (SPLITK) {      // fp32 partial sums of this part: a lane's 4 channels of a pixel as one 16-byte store; splitk_reduce_kernel finishes
    static_assert(MODE == 0 && !MASK && !POOL, "split-K: plain forwards");
    float* ws = a.kws + (size_t)blockIdx.z * ((size_t)g.B * g.OH * g.OW * a.N);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + wn * WTN + i * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int oy = y0s + wm * (TH / WARPS_M) + j, ox = x0s + fr;
        if (oy < g.OH && ox < g.OW && n < a.N) *reinterpret_cast<f32x4*>(ws + (((size_t)b * g.OH + oy) * g.OW + ox) * a.N + n) = acc[i][j];
      }
    }
    return;
  }
  // ---- epilogue (same as conv_gemm_kernel)
  const float scale = a.scale ? a.scale[a.scale_group ? b / a.scale_group : 0] : 1.f;
  T* out = static_cast<T*>(a.out);
  // deferred activation gradient factors.  bf16: ALL mask chunks of the tile (8 bytes per fragment, TN x TM x 2 registers -- the main
  // loop's fragment registers are dead by now) are requested before the first store, one memory round trip instead of one per channel
  // group (the compiler cannot move loads across the stores itself: out and mask may alias for all it knows).  fp32: per channel group.
  constexpr bool MASK_ALL = MASK && sizeof(T) == 2;
  u32x2 mraw[MASK_ALL ? TN : 1][MASK_ALL ? TM : 1];
  if constexpr (MASK_ALL) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + wn * WTN + i * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int oy = py + sub * (y0s + wm * (TH / WARPS_M) + j), ox = px + sub * (x0s + fr);
        u32x2 m = u32x2{0u, 0u};
        if (oy < g.OH && ox < g.OW && n < a.N)
          m = *reinterpret_cast<const u32x2*>(static_cast<const T*>(a.mask) + (((size_t)b * g.OH + oy) * g.OW + ox) * a.N + n);
        mraw[i][j] = m;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * WTN + i * 16 + (lane >> 4) * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < a.nbias) bv[r] = a.bias[n + r];
    }
    float mg[TM][4];
    if constexpr (MASK_ALL) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const u32x2 m = mraw[i][j];
        mg[j][0] = act_grad_from_out(half_lo_to_f32(m[0]), a.mask_act);
        mg[j][1] = act_grad_from_out(half_hi_to_f32(m[0]), a.mask_act);
        mg[j][2] = act_grad_from_out(half_lo_to_f32(m[1]), a.mask_act);
        mg[j][3] = act_grad_from_out(half_hi_to_f32(m[1]), a.mask_act);
      }
    } else if (MASK) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int oy = py + sub * (y0s + wm * (TH / WARPS_M) + j), ox = px + sub * (x0s + fr);
        float mv[4] = {0.f, 0.f, 0.f, 0.f};
        if (oy < g.OH && ox < g.OW && n < a.N) load4(static_cast<const T*>(a.mask) + (((size_t)b * g.OH + oy) * g.OW + ox) * a.N + n, mv);
#pragma unroll
        for (int r = 0; r < 4; ++r) mg[j][r] = act_grad_from_out(mv[r], a.mask_act);
      }
    }
    float vprev[4] = {0.f, 0.f, 0.f, 0.f};      // POOL: the previous (even) row fragment's values
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int oy = py + sub * (y0s + wm * (TH / WARPS_M) + j), ox = px + sub * (x0s + fr);
      const bool live = oy < g.OH && ox < g.OW && n < a.N;
      if (!POOL && !live) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(acc[i][j][r] * scale + bv[r], a.act);
      const size_t pixo = ((size_t)b * g.OH + oy) * g.OW + ox;
      if (MASK) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= mg[j][r];
      }
      if (live) {
        T* p = (a.out2 && n >= a.n_out1) ? static_cast<T*>(a.out2) + pixo * (a.N - a.n_out1) + (n - a.n_out1)
                                         : out + pixo * (a.out2 ? a.n_out1 : a.N) + n;
        store4(p, v[0], v[1], v[2], v[3]);      // channel counts are multiples of 4 (padded tensors)
      }
      if constexpr (POOL) {
        // rows (j - 1, j) of this wave and columns (fr, fr ^ 1) of neighbouring lanes form one 2x2 window (tile origins and the wave's first
        // row are even, OH and OW are even: a window lies wholly inside the map or wholly outside).  Rounding to T is monotonic, so the max of
        // the fp32 values rounded once equals the max of the stored values: bit-identical to a pooling pass over the written tensor.
        if (j & 1) {
          float m[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            m[r] = fmaxf(vprev[r], v[r]);
            m[r] = fmaxf(m[r], __shfl_xor(m[r], 1, 64));
          }
          if (live && !(fr & 1))
            store4(static_cast<T*>(a.pool_out) + (((size_t)b * (g.OH >> 1) + (oy >> 1)) * (g.OW >> 1) + (ox >> 1)) * a.N + n, m[0], m[1], m[2], m[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) vprev[r] = v[r];
        }
      }
    }
  }
}

// tile height of the patch kernel for a problem: 32 (8 waves, each 64 px x 128 channels, one patch buffer) for 65..128 output
// channels on maps >= 32 rows, 16 (8 waves) for other wide layers on maps >= 16 rows, else 8 (4 waves)
template <typename T, int KS>
static int patch_tile_h(const ConvArgs& a, int sh) {
  const bool big = KS <= 4 && a.N > 32 && sh >= 16;
  // (stride-2 class data gradients on the 4-wave 8 x 16 tiles -- two blocks per CU instead of one -- measured the same within 4 %, r3)
  // (not for reflection-padded dgrads: their interior/frame split loses more to the taller border tiles than the tile gains)
  if (big && a.N > 64 && a.N <= 128 && sh >= 32 && !(a.g.mode == 1 && a.g.pad_mode == UEGAN_PAD_REFLECT)) return 32;
  return big ? 16 : CONV_TH;
}

// MASK: the instantiations whose epilogue multiplies by act'(a.mask) (a separate set: the same code behind a run-time test
// cost every patch launch 1.6 % of the step in register allocation)
template <typename T, int KS, int MODE, bool MASK = false>
static int launch_conv_patch_m(ConvArgs& a, hipStream_t s) {
  const ConvGeom& g = a.g;
  const int sub = g.mode == 1 ? g.stride : 1;
  const int sh = (g.OH + sub - 1) / sub, sw = (g.OW + sub - 1) / sub;
  // 256-pixel tiles (8 waves, 3-deep weight ring) for wide layers on maps that fill them; the LDS budget allows them up to KS = 4
  int th = patch_tile_h<T, KS>(a, sh);
  // (uegan_set_tuning: the tests lower it to reach the large-grid variants on emulator-sized maps)
  const int small_grid = g_tuning[UEGAN_TUNE_SMALL_GRID];
  if (th == 32 && g.B * sub * sub * ((sh + 31) / 32) * ((sw + CONV_TW - 1) / CONV_TW) < small_grid) th = 16;   // small maps: see below
  // 1x1 convs with 65..128 output channels run 64-channel blocks on 16-row tiles (below): the tile grid must be counted with THAT height.
  // (Round 3 counted it with the 32-row tiles of the 128-channel block this replaced: on grids >= small_grid -- ga3 / up2 at batch 32 --
  // every second band of 16 rows was never written.  Found by tests/test_parity_full.py::test_step_does_not_depend_on_uninitialised_memory.)
  if (KS == 1 && th == 32) th = 16;
  const bool big = th >= 16;
  a.nty = (sh + th - 1) / th;
  a.ntx = (sw + CONV_TW - 1) / CONV_TW;
  int per = a.nty * a.ntx;
  if (a.frame == 2) per = (a.fy1 - a.fy0) * (a.fx1 - a.fx0);
  else if (a.frame == 1) per = a.fy0 * a.ntx + (a.nty - a.fy1) * a.ntx + (a.fy1 - a.fy0) * (a.fx0 + a.ntx - a.fx1);
  const int gm = g.B * sub * sub * per;
  if (gm == 0) return UEGAN_OK;
  const int bn_idx = a.N > 64 ? 3 : (a.N > 32 ? 2 : (a.N > 16 ? 1 : 0));
  double rows = g.mode == 0 ? (double)g.B * g.OH * g.OW : (double)g.B * g.IH * g.IW;
  if (a.frame) rows *= (double)per / (a.nty * a.ntx);
  static const int kBn[4] = {16, 32, 64, 128};
  const bool use256 = big && a.N >= 256;
  ProfScope prof(prof_key(1, DT<T>::kDtype == UEGAN_BF16, use256 ? 256 : kBn[bn_idx], KS, MODE, (big && a.N > 32) ? th : 8, true),
                 2.0 * rows * a.N * (double)(g.KH * g.KW * g.C), s,
                 sizeof(T) * (rows * a.N + (double)g.B * g.IH * g.IW * g.C * (a.frame ? (double)per / (a.nty * a.ntx) : 1.0)));
  constexpr int KB = KS <= 4 ? KS : 2;      // instantiate the 256-pixel variants only where they fit
  // fused 2x2 max-pool (a.pool_out): the 3x3 forwards on the 64-channel / 128-channel-32-row tiles (VGG conv1_2, conv2_2); anything else
  // leaves pool_done = 0 and the caller runs the pooling kernel
  constexpr bool pool_ok = KS == 3 && MODE == 0 && !MASK;
  const bool pool = pool_ok && a.pool_out && !a.pool_idx && a.frame == 0 && g.OH % 2 == 0 && g.OW % 2 == 0 && !a.out2;      // (positions: conv_tall_kernel's epilogue or the pooling kernel)
  // 64-channel blocks on 256-pixel tiles always run with ONE patch buffer (61 instead of 98 KB of LDS): two blocks share a CU and cover
  // each other's prologue (a block waits ~2 us for its first patch and then runs 1-18 K steps) and chunk switches.  Measured at batch 32:
  // the parity-class data gradients of enc3 / d3 0.40 -> 0.27 / 0.33 -> 0.23 ms, G.dec3 forward 0.64 -> 0.47, VGG conv2_1 data gradient
  // 0.62 -> 0.46.  (128- and 256-channel blocks: a single-buffer variant with a 2-deep weight ring measured equal or slower.)
  if constexpr (KS == 1) {
    // 1x1 convs with 65..128 output channels (K = a few 64-channel chunks): 64-channel single-buffer blocks, two per CU, instead of one
    // 128-channel block per CU (ga3 / up2 forward 0.094 / 0.034 -> 0.051 / 0.020 ms at batch 32; 256+ channels: the 256-channel blocks win)
    if (a.N > 64 && a.N <= 128 && big) {
      hipLaunchKernelGGL((conv_patch_kernel<T, 64, 4, 2, KB, MODE, 16, 3, 1, true, MASK>), dim3(gm, (a.N + 63) / 64), dim3(512), 0, s, a);
      UEGAN_CHECK_LAUNCH();
      return UEGAN_OK;
    }
  }
  if constexpr (MODE == 0 && !MASK && KS == 3 && sizeof(T) == 2) {
    // a grid that leaves most CUs empty (single-image inference: G.dec1 / dec2 = 64 / 128 blocks of 72 / 36 K steps): the chunks of the K loop over several
    // blocks per tile, fp32 partials to the caller's workspace, the epilogue in splitk_reduce_kernel
    if (a.N > 64 && th == 16 && a.kws) {      // (th: a.nty above counts 16-row tiles -- not the 32-row tiles a lowered UEGAN_TUNE_SMALL_GRID keeps on small maps)
      int kchunks;
      const int parts = splitk_parts(a, gm * ((a.N + 63) / 64), g.C / (CONV_ROWB / (int)sizeof(T)), &kchunks);
      if (parts > 1) {
        a.kparts = parts; a.kchunks = kchunks;
        hipLaunchKernelGGL((conv_patch_kernel<T, 64, 4, 2, KB, MODE, 16, 3, 1, false, false, false, true>), dim3(gm, (a.N + 63) / 64, parts), dim3(512), 0, s, a);
        UEGAN_CHECK_LAUNCH();
        return splitk_reduce_launch(a, s);
      }
    }
  }
  if (a.N > 64 && gm * ((a.N + 127) / 128) < small_grid) {         // small maps: 64-channel blocks so the grid covers the chip
    if (big) hipLaunchKernelGGL((conv_patch_kernel<T, 64, 4, 2, KB, MODE, 16, 3, 1, false, MASK>), dim3(gm, (a.N + 63) / 64), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv_patch_kernel<T, 64, 2, 2, KS, MODE, 8, 2, 1, false, MASK>), dim3(gm, (a.N + 63) / 64), dim3(256), 0, s, a);
  } else if (th == 32) {
    if constexpr (pool_ok) {
      if (pool) {
        hipLaunchKernelGGL((conv_patch_kernel<T, 128, 8, 1, KB, MODE, 32, 3, 1, true, false, true>), dim3(gm, 1), dim3(512), 0, s, a);
        a.pool_done = 1;
        UEGAN_CHECK_LAUNCH();
        return UEGAN_OK;
      }
    }
    hipLaunchKernelGGL((conv_patch_kernel<T, 128, 8, 1, KB, MODE, 32, 3, 1, true, MASK>), dim3(gm, 1), dim3(512), 0, s, a);
  } else if (a.N > 64) {
    // >= 256 output channels: 256-channel blocks (each wave 64 px x 128 ch: 12 LDS fragment reads per 32 MFMAs instead of 8 per
    // 16, and twice the MFMAs behind every barrier), 2-deep weight ring to stay inside 160 KB.  VGG 512->512: 950 -> 1170 TFLOP/s
    if (big && a.N >= 256) {
      hipLaunchKernelGGL((conv_patch_kernel<T, 256, 4, 2, KB, MODE, 16, 2, 1, false, MASK>), dim3(gm, (a.N + 255) / 256), dim3(512), 0, s, a);
      UEGAN_CHECK_LAUNCH();
      return UEGAN_OK;
    }
    if (big) hipLaunchKernelGGL((conv_patch_kernel<T, 128, 4, 2, KB, MODE, 16, 3, 1, false, MASK>), dim3(gm, (a.N + 127) / 128), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv_patch_kernel<T, 128, 2, 2, KS, MODE, 8, 2, 1, false, MASK>), dim3(gm, (a.N + 127) / 128), dim3(256), 0, s, a);
  } else if (a.N > 32) {
    if constexpr (pool_ok) {
      if (big && pool) {
        hipLaunchKernelGGL((conv_patch_kernel<T, 64, 4, 2, KB, MODE, 16, 3, 1, true, false, true>), dim3(gm, 1), dim3(512), 0, s, a);
        a.pool_done = 1;
        UEGAN_CHECK_LAUNCH();
        return UEGAN_OK;
      }
    }
    if (big) hipLaunchKernelGGL((conv_patch_kernel<T, 64, 4, 2, KB, MODE, 16, 3, 1, true, MASK>), dim3(gm, 1), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv_patch_kernel<T, 64, 2, 2, KS, MODE, 8, 2, 1, false, MASK>), dim3(gm, 1), dim3(256), 0, s, a);
  } else if (a.N > 16) {
    // one 64-channel chunk: nothing to prefetch into a second patch buffer -- without it four blocks share a CU instead of two
    // (D.d2's parity-class data gradient, 64 dz channels -> 32: 0.34 + 0.35 -> 0.22 + 0.23 ms at batch 48)
    if (g.C <= CONV_ROWB / (int)sizeof(T)) hipLaunchKernelGGL((conv_patch_kernel<T, 32, 4, 1, KS, MODE, 8, 2, 1, true, MASK>), dim3(gm, 1), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_patch_kernel<T, 32, 4, 1, KS, MODE, 8, 2, 1, false, MASK>), dim3(gm, 1), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((conv_patch_kernel<T, 16, 4, 1, KS, MODE, 8, 2, 1, false, MASK>), dim3(gm, 1), dim3(256), 0, s, a);
  }
  UEGAN_CHECK_LAUNCH();
  return UEGAN_OK;
}

template <typename T, int KS>
static int launch_conv_patch(ConvArgs& a, hipStream_t s) {
  if (a.g.mode == 0) return launch_conv_patch_m<T, KS, 0>(a, s);
  if (a.g.pad_mode != UEGAN_PAD_REFLECT || a.g.pad == 0) {      // (pad 0: a reflection pad of nothing has no mirrored images)
    if constexpr (KS == 3 || KS == 1) {       // (masked epilogues exist for the VGG chain's 3x3 and the generator's 1x1 data gradients)
      if (a.mask) return launch_conv_patch_m<T, KS, 1, true>(a, s);
    }
    return launch_conv_patch_m<T, KS, 1>(a, s);
  }
  // reflection-padded dgrad: only the border tiles can carry mirrored images.  The tile rectangle that cannot runs the
  // image-free instantiation (no per-fragment masks, one phase per chunk), the frame around it the full one.
  if constexpr (KS == 1) return UEGAN_E_UNSUPPORTED;      // (1x1 convs have pad 0: handled above; no MODE 2 instantiation for them)
  else {
  const ConvGeom& g = a.g;
  const int sub = g.stride;
  const int sh = (g.OH + sub - 1) / sub, sw = (g.OW + sub - 1) / sub;
  const int th = patch_tile_h<T, KS>(a, sh);        // (same tile choice as launch_conv_patch_m)
  const int nty = (sh + th - 1) / th, ntx = (sw + CONV_TW - 1) / CONV_TW;
  if (a.border_only) {
    // the image-free rectangle was computed by conv_wide_kernel / conv_tall_kernel (conv_interior_run): multiples of 16 rows x 32 columns,
    // i.e. whole tiles of this kernel (th = 8 or 16, 16 columns); only the frame with the mirrored images is left
    // (whole tiles of THIS launch, or rows / columns between the rectangle and the next tile boundary would be written by neither kernel)
    UEGAN_CHECK_ARG(a.rect_y0 % th == 0 && a.rect_y1 % th == 0 && a.rect_x0 % CONV_TW == 0 && a.rect_x1 % CONV_TW == 0,
                    "reflect dgrad: the image-free rectangle [%d,%d) x [%d,%d) is not whole %d x %d tiles of the frame launch", a.rect_y0, a.rect_y1,
                    a.rect_x0, a.rect_x1, th, CONV_TW);
    a.fy0 = a.rect_y0 / th; a.fy1 = a.rect_y1 / th; a.fx0 = a.rect_x0 / CONV_TW; a.fx1 = a.rect_x1 / CONV_TW;
    a.frame = 1;
    const int rc = launch_conv_patch_m<T, KS, 2>(a, s);
    a.frame = 0;
    return rc;
  }
  // the far mirror (bottom / right) exists only if the forward reads the far padding at all: a stride-2 3x3 conv on an even extent (G.enc2-5)
  // stops one row short of it, and the tiles along those two edges are image-free
  auto clean = [&](int tile, int tn, int n, int in_n, int K) {      // no pixel of this tile (any parity class) has a mirrored image
    const int lo = sub * tile * tn, hi = (sub - 1) + sub * (tile * tn + tn - 1);
    const bool far = (in_n - 1) * g.stride + K - 1 >= n + g.pad;
    const bool m0 = lo <= g.pad && hi >= 1, m1 = far && lo <= n - 2 && hi >= n - 1 - g.pad;
    return !m0 && !m1;
  };
  int y0 = 0, x0 = 0;
  while (y0 < nty && !clean(y0, th, g.OH, g.IH, g.KH)) ++y0;
  int y1 = y0;
  while (y1 < nty && clean(y1, th, g.OH, g.IH, g.KH)) ++y1;
  while (x0 < ntx && !clean(x0, CONV_TW, g.OW, g.IW, g.KW)) ++x0;
  int x1 = x0;
  while (x1 < ntx && clean(x1, CONV_TW, g.OW, g.IW, g.KW)) ++x1;
  // (two launches only when the image-free one fills the chip by itself: G.enc5's 2 x 2 class tiles with one clean tile measured +24 % split)
  if (y1 <= y0 || x1 <= x0 || (y1 - y0) * (x1 - x0) * 4 < nty * ntx || g.B * sub * sub * (y1 - y0) * (x1 - x0) * ((a.N + 255) / 256) < g_tuning[UEGAN_TUNE_SMALL_GRID])
    return launch_conv_patch_m<T, KS, 2>(a, s);
  a.fy0 = y0; a.fy1 = y1; a.fx0 = x0; a.fx1 = x1;
  a.frame = 2;
  int rc = launch_conv_patch_m<T, KS, 1>(a, s);
  if (rc) return rc;
  a.frame = 1;
  rc = launch_conv_patch_m<T, KS, 2>(a, s);
  a.frame = 0;
  return rc;
  }
}

}  // namespace uegan
