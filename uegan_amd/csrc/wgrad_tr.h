// Weight-gradient kernel for bf16 built on the gfx950 LDS transpose read (ds_read_b64_tr_b16).  Included by conv.hip only.
//
//   dW[n][tap][c] = sum_pixels dz[pixel][n] * x[pixel*stride + tap - pad][c]          (reference: autograd of F.conv2d)
//
// The contraction index is the PIXEL, but both tensors are stored pixel-major (NHWC): an MFMA lane needs 8 pixels of
// ONE channel.  The older conv_wgrad_kernel transposes 8x8 blocks in registers and re-reads x once per tap (KH*KW times).
// Here the x patch and the dz tile are staged ONCE per tile, untransposed, straight into LDS (global_load_lds), and
// the fragments are fetched with ds_read_b64_tr_b16: within a 16-lane group, lane s supplies the address of 4 consecutive
// channels [4*(s&3), +4) of pixel (s>>2) and receives the 4 pixels of channel s -- so every lane picks its own pixel row,
// which makes a tap shift (and a stride of 2) just a different address.  (Semantics probed on hardware: tools/probe_tr16.hip.)
//
//   * block = (64-channel chunk of x  x  a range of (tap, 16-channel) fragment slots) x (<= 64 dz channels) x a range of
//     pixel tiles (split-K); 4 waves = WK (fragment slots) x WS (k-steps); accumulators stay in registers over the whole
//     tile range; partial sums go to the workspace [split][N][ktot] and wgrad_reduce_kernel finishes (sum, 1/sigma
//     scale, permute to OIHW)
//   * tile = TH x TW output pixels; a k-step is 32 pixels: TW = 32 -> one tile row, TW = 16 -> two rows.  Pixel of MFMA
//     k index (g = lane group, e = 4h + j):  dx = j + 4(g&1) + 8h (+ 16(g>>1) if TW = 32), dy = g>>1 if TW = 16
//   * LDS rows are one pixel (xrb / zrb bytes = channels of the chunk); the 16-byte chunk q of the pixel in patch column
//     pc sits at q ^ swz(pc), with swz chosen so the 8 pixels x 32 bytes a half-wave reads hit 64 distinct banks.  The
//     swizzle phase of a lane depends on its column only (h adds 8 columns, k-steps and kernel rows add whole rows), so
//     all per-lane addresses are computed once per kernel and the patch needs no padding columns
//   * two LDS buffers: the loads of tile t+1 are in flight while tile t is multiplied; one barrier per tile.  The loads
//     are issued from inline asm (see wgtr_glds16) -- with the builtin, hipcc drains them before the first LDS read
//   * heads (1-3 real dz channels) would use 1-3 of the 16 MFMA rows.  Head mode re-indexes the reduction by the patch column
//     q = p + tx:  dW[n][ty][tx][c] = sum_q dzx[q][(tx,n)] * x[q + ty*row][c]  with  dzx[q][(tx,n)] = dz[q - tx][n] (zero outside the
//     tile): the rows of the MFMA become the KW*N (tx, n) pairs, the x fragments depend on ty only -- KW times fewer MFMAs and
//     LDS reads for 64 instead of 32 reduction columns per tile row.  dzx is built in LDS from the staged dz tile (one extra barrier)
//   * bias gradient (sum of dz over pixels) rides along: the blocks of the first x slot range multiply the dz fragments
//     with an all-ones B fragment (TN extra MFMAs per k-step) instead of a second pass over dz
#pragma once

struct WgradTrArgs {
  ConvGeom g;          // forward gather geometry (mode 0)
  const void* in1;
  const void* in2;
  const void* dz;      // [B][OH][OW][zC]
  float* ws;           // [nsplit * WS][pstride]: N*ktot weight partials, then N bias partials
  int N, zC, ktot, pstride, want_bias;
  int TW, TWlog, TH, PW, PWmagic, PWused, nks, dyk;
  int xrb, xrblog, zrb, zrblog;     // LDS bytes per pixel row (x patch / dz tile) and log2
  int xcb;                          // x channels per chunk (<= 64)
  int cfpc, fslots, fpb, nfr;       // 16-channel fragments per chunk (0: 8-channel tensors, a fragment = 2 taps), slots per chunk,
                                    // slots per block (WK*TM), slot ranges per chunk
  int WK, WS;
  int tiles_x, tiles_y, tiles_total, tiles_per_split;
  int xbytes, zbytes;               // LDS bytes per buffer: x patch (the rows the widest slot range needs), dz tile
  int rs;                           // source rows per staged patch row: 2 when every slot range is ONE kernel row of a stride-2 conv (only the
                                    // input rows of that row's parity are read: they are staged densely), else 1
  int nbuf;                         // LDS buffers (2): tile t+1 is in flight while tile t is multiplied
  int xcd_map;                      // 0: launch order; G = 8 / 4 / 2 / 1: XCD-aware block order in runs of G splits (see the kernel)
  int abl;                          // timing ablations (tools build only, UEGAN_ABL_BITS): 1 no staging after the first tile, 2 no MFMA loop
  int head, hE, hEB, hEBlog;        // head mode (<= 4 real dz channels, KW >= 3): MFMA rows = (tx, n) pairs from an im2col of dz over tx
                                    // built in LDS per tile (hE = KW*N rows, hEB = bytes per dzx pixel); slots = (ty, 16 channels)
};

__device__ __forceinline__ int wgtr_swz(int rb, int r) {   // XOR term for the 16-byte chunk index of LDS row r
  return rb == 128 ? (r & 6) : (rb == 64 ? ((r >> 1) & 2) : 0);
}

template <int KB>
__device__ __forceinline__ unsigned char* wgtr_lds() {
  __shared__ __attribute__((aligned(16))) unsigned char buf[KB * 1024];
  return buf;
}

// 4 consecutive bf16 per supplier lane -> 4 pixels of one channel per receiver lane (see header)
__device__ __forceinline__ u32x2 lds_read_tr16(const unsigned char* p) {
  typedef short v4s_t __attribute__((ext_vector_type(4)));
  const v4s_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)p);
  return __builtin_bit_cast(u32x2, r);
}

constexpr int WGTR_SMALL_KB = 80, WGTR_BIG_KB = 152;

// (wgtr_glds16 / wgtr_wait_loads, the asm-issued direct-to-LDS loads: conv_core.h)

template <int TN, int TM, bool BIG, bool HEAD>
__global__ void __launch_bounds__(256, BIG ? 1 : 2) wgrad_tr_kernel(WgradTrArgs a) {
  constexpr int MAXIX = BIG ? 19 : 10, MAXIZ = 8;
  unsigned char* lds = wgtr_lds<BIG ? WGTR_BIG_KB : WGTR_SMALL_KB>();
  const ConvGeom& g = a.g;
  const unsigned char* in1 = static_cast<const unsigned char*>(a.in1);
  const unsigned char* in2 = static_cast<const unsigned char*>(a.in2);
  const unsigned char* dz = static_cast<const unsigned char*>(a.dz);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: lets LDS bases and branches live in SGPRs
  const int wk = wave % a.WK, wsid = wave / a.WK;
  // XCD-aware block order (round 5).  Workgroups go to the 8 XCDs round-robin by their linear id; every (x chunk, slot range, dz block) block of one
  // pixel SPLIT streams the same pixels, so with a.xcd_map the linear ids are regrouped in runs of 8 x (blocks per split): inside a run, id % 8 picks
  // the split and id / 8 the block -- all blocks of a split sit behind ONE L2, which then serves the re-reads of its x and dz tiles (8 x-chunk blocks
  // share a dz tile, 4 dz blocks an x patch) instead of the MALL / HBM.  Time: within noise (the kernel is latency-bound, DESIGN history); traffic: see
  // profiles/r05_final_step_traffic.txt.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd_map) {
    // G = splits per run: 8 (one XCD each), or all 1 / 2 / 4 splits of a launch with fewer (8 / G XCDs each: the deep discriminator layers)
    const int P = gridDim.x * gridDim.y, G = a.xcd_map, m = 8 / G;
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int run = L / (G * P), r = L - run * G * P;      // (gridDim.z is a multiple of G and G * P of 8: checked by the launcher)
    const int xcd = r & 7, k = r >> 3;
    bz = run * G + xcd % G;
    const int q = k * m + xcd / G;
    by = q / gridDim.x;
    bx = q - by * gridDim.x;
  }
  const int cc = bx / a.nfr, fr = bx - cc * a.nfr;
  const int nb = by, split = bz;
  const int taps = g.KH * g.KW;
  const int f0 = fr * a.fpb;
  // kernel rows this block's fragment slots touch -> patch rows to stage
  int tapA, tapB;
  if (a.cfpc) { tapA = f0 / a.cfpc; tapB = (f0 + a.fpb - 1) / a.cfpc; }
  else { tapA = 2 * f0; tapB = 2 * (f0 + a.fpb) - 1; }
  if (HEAD) { tapA *= g.KW; tapB = tapB * g.KW + g.KW - 1; }       // head mode: a slot is a whole kernel row
  if (tapB > taps - 1) tapB = taps - 1;
  if (tapA > taps - 1) tapA = taps - 1;
  const int ty_lo = tapA / g.KW, ty_hi = tapB / g.KW;
  const int rstep = g.stride / a.rs;                                   // LDS patch rows per output row
  const int PH = (a.TH - 1) * rstep + (ty_hi - ty_lo) + 1;
  const int xcpr_log = a.xrblog - 4, zcpr_log = a.zrblog - 4;          // log2(16-byte chunks per row)
  const int nxc = (PH * a.PW) << xcpr_log, nzc = (a.TH * a.TW) << zcpr_log;
  const int c_chunk0 = cc * 64, n_chunk0 = nb * 64;
  const int nix = (nxc + 255) >> 8, niz = (nzc + 255) >> 8;      // 256-lane staging rounds per buffer

  // ---- per-thread staging tables (tile independent) ----
  // x: chunk L = it*256 + tid -> LDS row r = L >> xcpr_log (patch pixel prow*PW + pcol), position L & (cpr-1)
  uint32_t xoff[MAXIX];   // interior tiles: byte offset from the tile's patch origin in its source (bit 31: second source).  Chunks
                          // nothing reads (padding columns, staging-round tail) load offset 0: valid memory, harmless LDS slot
  const bool two_src = g.C2 != 0 && c_chunk0 < g.C1 && c_chunk0 + a.xcb > g.C1;     // the chunk straddles both sources
  const bool chunk_in2 = g.C2 != 0 && c_chunk0 >= g.C1;
#pragma unroll
  for (int it = 0; it < MAXIX; ++it) {
    const int L = it * 256 + tid;
    xoff[it] = 0;
    if (L < nxc) {
      const int r = L >> xcpr_log, pos = L & ((1 << xcpr_log) - 1);
      const int prow = (r * a.PWmagic) >> 16, pcol = r - prow * a.PW;
      const int c = c_chunk0 + ((pos ^ wgtr_swz(a.xrb, pcol)) << 3);
      if (pcol < a.PWused && c < g.C) {
        if (c < g.C1) xoff[it] = (uint32_t)(((prow * a.rs * g.IW + pcol) * g.C1 + c) * 2);
        else xoff[it] = (uint32_t)(((prow * a.rs * g.IW + pcol) * g.C2 + (c - g.C1)) * 2) | (two_src ? 0x80000000u : 0u);
      }
    }
  }
  uint32_t zoff[MAXIZ];
#pragma unroll
  for (int it = 0; it < MAXIZ; ++it) {
    const int L = it * 256 + tid;
    zoff[it] = 0;
    if (L < nzc) {
      const int r = L >> zcpr_log, pos = L & ((1 << zcpr_log) - 1);
      const int oy = r >> a.TWlog, ox = r & (a.TW - 1);
      const int c = n_chunk0 + ((pos ^ wgtr_swz(a.zrb, r)) << 3);
      if (c < a.zC) zoff[it] = (uint32_t)(((oy * g.OW + ox) * a.zC + c) * 2);
    }
  }

  // ---- per-lane fragment addresses (supplier role of ds_read_b64_tr_b16) ----
  const int g4 = lane >> 4, sj = (lane & 15) >> 2, seg = lane & 3;
  const int dx = sj + 4 * (g4 & 1) + (a.TW == 32 ? 16 * (g4 >> 1) : 0);
  const int dy = a.TW == 32 ? 0 : (g4 >> 1);
  int zaddr[TN], xaddr[TM], xaddr1[TM];      // xaddr1: head mode, second half (columns 32..63) of a tile row
  if (HEAD) {
    // A: dzx pixel (tile row, column q = dx), row block nf: channels nf*16 + 4*seg .. +3 of hEB/2 stored ones
    const int zch = a.hEB >> 1;
#pragma unroll
    for (int nf = 0; nf < TN; ++nf) {
      int ch = nf * 16 + 4 * seg;
      if (ch >= zch) ch = 4 * (seg & 1);
      zaddr[nf] = dx * a.hEB + ((((ch >> 3) ^ wgtr_swz(a.hEB, dx))) << 4) + ((ch >> 2) & 1) * 8;
    }
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      const int f = f0 + wk * TM + m;
      int ty = f / a.cfpc;
      const int ch = (f - ty * a.cfpc) * 16 + 4 * seg;
      if (ty > ty_hi) ty = ty_lo;
      const int q1 = dx + 32 < a.PWused ? dx + 32 : a.PWused - 1;       // columns beyond the patch: dzx is zero there, any finite x
      xaddr[m] = ((ty - ty_lo) * a.PW + dx) * a.xrb + ((((ch >> 3) ^ wgtr_swz(a.xrb, dx))) << 4) + ((ch >> 2) & 1) * 8;
      xaddr1[m] = ((ty - ty_lo) * a.PW + q1) * a.xrb + ((((ch >> 3) ^ wgtr_swz(a.xrb, q1))) << 4) + ((ch >> 2) & 1) * 8;
    }
  } else {
#pragma unroll
    for (int m = 0; m < TM; ++m) xaddr1[m] = 0;
    const int rz = dy * a.TW + dx;
    const int zcb = a.zC - n_chunk0 < 64 ? a.zC - n_chunk0 : 64;
#pragma unroll
    for (int nf = 0; nf < TN; ++nf) {
      int ch = nf * 16 + 4 * seg;
      if (ch >= zcb) ch = 4 * (seg & 1);              // rows of the fragment that do not exist: any valid address (results dropped)
      zaddr[nf] = rz * a.zrb + ((((ch >> 3) ^ wgtr_swz(a.zrb, rz))) << 4) + ((ch >> 2) & 1) * 8;
    }
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      const int f = f0 + wk * TM + m;
      int tap, ch;
      if (a.cfpc) { tap = f / a.cfpc; ch = (f - tap * a.cfpc) * 16 + 4 * seg; }
      else { tap = 2 * f + (seg >> 1); ch = 4 * (seg & 1); }
      if (tap > tapB) tap = tapA;                     // slot beyond the kernel: any valid address (results dropped)
      const int ty = tap / g.KW, tx = tap - ty * g.KW;
      const int pcx = dx * g.stride + tx;
      const int rx = (dy * rstep + ty - ty_lo) * a.PW + pcx;
      xaddr[m] = rx * a.xrb + ((((ch >> 3) ^ wgtr_swz(a.xrb, pcx))) << 4) + ((ch >> 2) & 1) * 8;
    }
  }
  const int z_ks = HEAD ? 32 * a.hEB : a.dyk * a.TW * a.zrb, z_h = 8 * (HEAD ? a.hEB : a.zrb);     // head: a k-step = half a dzx row
  const int x_ks = a.dyk * rstep * a.PW * a.xrb, x_h = 8 * g.stride * a.xrb;
  unsigned char* const dzx = lds + 2 * (a.xbytes + a.zbytes);

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int m = 0; m < TM; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 accb[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = a.want_bias && bx == 0 && wk == 0;
  const uint32_t one2 = (uint32_t)f32_to_bf16(1.f) * 0x10001u;                       // a pair of 1.0 in the 16-bit storage format
  const u32x4 ones = u32x4{one2, one2, one2, one2};

  int t_begin = split * a.tiles_per_split, t_end = t_begin + a.tiles_per_split;
  if (t_end > a.tiles_total) t_end = a.tiles_total;

  auto stage = [&](int t, int bufi) {
    unsigned char* xb = lds + bufi * (a.xbytes + a.zbytes);
    unsigned char* zb = xb + a.xbytes;
    const int txi = t % a.tiles_x;
    const int tq = t / a.tiles_x;
    const int tyi = tq % a.tiles_y, b = tq / a.tiles_y;      // (row-major sweeps; column-major strips -- halo rows as L2 hits -- measured 8-20 % slower)
    const int oy0 = tyi * a.TH, ox0 = txi * a.TW;
    const int iy0 = oy0 * g.stride + ty_lo - g.pad, ix0 = ox0 * g.stride - g.pad;
    const bool z_inside = oy0 + a.TH <= g.OH && ox0 + a.TW <= g.OW;
    const bool x_inside = iy0 >= 0 && iy0 + (PH - 1) * a.rs + 1 <= g.IH && ix0 >= 0 && ix0 + a.PWused <= g.IW;
    // ---- x patch ----
    if (x_inside && z_inside) {
      // uniform 64-bit base + per-lane 32-bit offset: the loads use the SGPR-base addressing mode, no per-lane address math.
      // A chunk that straddles both sources issues two exec-masked loads per round instead.
      const long long pix0 = ((long long)b * g.IH + iy0) * g.IW + ix0;
      const unsigned char* o2 = in2 + pix0 * g.C2 * 2;
      const unsigned char* o1 = chunk_in2 ? o2 : in1 + pix0 * g.C1 * 2;
#pragma unroll
      for (int it = 0; it < MAXIX; ++it)
        if (it < nix) {
          const uint32_t o = xoff[it];
          unsigned char* dst = xb + (it * 256 + wave * 64) * 16;
          if (two_src && (o >> 31)) wgtr_glds16(o2, o & 0x7fffffffu, dst);
          else wgtr_glds16(o1, o, dst);
        }
    } else if (g.pad_mode == UEGAN_PAD_REFLECT) {
      // border tile, reflection padding: every patch pixel maps to a real pixel (coordinates of rows/columns nothing
      // reads are clamped), so the loads keep the uniform-base form with 32-bit offsets from the image origin
      const long long img = (long long)b * g.IH * g.IW;
      const unsigned char* o2 = in2 + img * g.C2 * 2;
      const unsigned char* o1 = chunk_in2 ? o2 : in1 + img * g.C1 * 2;
#pragma unroll 1
      for (int it = 0; it < nix; ++it) {
        const int L = it * 256 + tid;
        const int r = L >> xcpr_log, pos = L & ((1 << xcpr_log) - 1);
        const int prow = (r * a.PWmagic) >> 16, pcol = r - prow * a.PW;
        int c = c_chunk0 + ((pos ^ wgtr_swz(a.xrb, pcol)) << 3);
        if (c >= g.C) c = c_chunk0;
        int iy = iy0 + prow * a.rs, ix = ix0 + pcol;
        iy = iy < 0 ? -iy : iy;
        iy = iy >= g.IH ? 2 * (g.IH - 1) - iy : iy;
        iy = iy < 0 ? 0 : (iy >= g.IH ? g.IH - 1 : iy);
        ix = ix < 0 ? -ix : ix;
        ix = ix >= g.IW ? 2 * (g.IW - 1) - ix : ix;
        ix = ix < 0 ? 0 : (ix >= g.IW ? g.IW - 1 : ix);
        const uint32_t pix = (uint32_t)(iy * g.IW + ix);
        unsigned char* dst = xb + (it * 256 + wave * 64) * 16;
        if (c < g.C1 || chunk_in2) wgtr_glds16(o1, (pix * (uint32_t)(chunk_in2 ? g.C2 : g.C1) + (uint32_t)(chunk_in2 ? c - g.C1 : c)) * 2u, dst);
        else wgtr_glds16(o2, (pix * (uint32_t)g.C2 + (uint32_t)(c - g.C1)) * 2u, dst);
      }
    } else {
#pragma unroll 1
      for (int it = 0; it < nix; ++it) {
        const int L = it * 256 + tid;
        const void* src = g_zero16;
        if (L < nxc) {
          const int r = L >> xcpr_log, pos = L & ((1 << xcpr_log) - 1);
          const int prow = (r * a.PWmagic) >> 16, pcol = r - prow * a.PW;
          const int c = c_chunk0 + ((pos ^ wgtr_swz(a.xrb, pcol)) << 3);
          const int iy = iy0 + prow * a.rs, ix = ix0 + pcol;
          if (pcol < a.PWused && c < g.C && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW) {
            const long long pix = ((long long)b * g.IH + iy) * g.IW + ix;
            src = c < g.C1 ? in1 + (pix * g.C1 + c) * 2 : in2 + (pix * g.C2 + (c - g.C1)) * 2;
          }
        }
        wgtr_glds16(src, xb + (it * 256 + wave * 64) * 16);
      }
    }
    // ---- dz tile ----
    if (z_inside) {
      const unsigned char* oz = dz + (((long long)b * g.OH + oy0) * g.OW + ox0) * a.zC * 2;
#pragma unroll
      for (int it = 0; it < MAXIZ; ++it)
        if (it < niz) wgtr_glds16(oz, zoff[it], zb + (it * 256 + wave * 64) * 16);
    } else {
#pragma unroll 1
      for (int it = 0; it < niz; ++it) {
        const int L = it * 256 + tid;
        const void* src = g_zero16;
        if (L < nzc) {
          const int r = L >> zcpr_log, pos = L & ((1 << zcpr_log) - 1);
          const int oy = oy0 + (r >> a.TWlog), ox = ox0 + (r & (a.TW - 1));
          const int c = n_chunk0 + ((pos ^ wgtr_swz(a.zrb, r)) << 3);
          if (oy < g.OH && ox < g.OW && c < a.zC) src = dz + ((((long long)b * g.OH + oy) * g.OW + ox) * a.zC + c) * 2;
        }
        wgtr_glds16(src, zb + (it * 256 + wave * 64) * 16);
      }
    }
  };

  // iteration t: wait for tile t (later tiles may stay in flight), start the loads of tile t+nbuf-1 into the buffer that
  // was multiplied in iteration t-1, multiply tile t.  The first nbuf-1 iterations (t < t_begin) only start loads.
  const int nb1 = a.nbuf - 1;
  for (int t = t_begin - nb1; t < t_end; ++t) {
    const bool have = t >= t_begin;
    int bufi = 0;
    if (have) {
      bufi = (t - t_begin) % a.nbuf;
      wgtr_wait_loads();           // (with more than 2 buffers a counted wait would do; 3 buffers measured slower)
      raw_barrier();               // tile t landed for every wave; everyone is done reading the buffer of tile t-1
    }
    const int tn = t + nb1;
    if (tn < t_end && !((UEGAN_ABL_BITS(a.abl) & 1) && have)) stage(tn, (tn - t_begin) % a.nbuf);
    if (!have) continue;
    const unsigned char* xb = lds + bufi * (a.xbytes + a.zbytes);
    const unsigned char* zb = xb + a.xbytes;
    if (HEAD) {
      // im2col of the dz tile over tx: dzx[row][q][(tx, n)] = dz[row][q - tx][n] for 0 <= q - tx < TW, else 0 (64 columns per row).
      // One thread per dzx pixel: 7 eight-byte reads (the <= 4 channels of dz pixels q .. q-6), the (tx, n) entries packed in registers
      // (all indices compile-time: N is a template argument of the builder, KW a predicate on the unrolled tx loop), <= 4 sixteen-byte
      // writes.  (Round 2 built it chunk by chunk from 2-byte LDS reads with two divisions per chunk: 0.4 of dec5.1's 0.76 ms, UEGAN_ABL.)
      auto build = [&](auto n_c) {
        constexpr int NN = decltype(n_c)::value;
        const int cpp = a.hEB >> 4;
        for (int pq = tid; pq < a.TH * 64; pq += 256) {
          const int q = pq & 63, row = pq >> 6;
          uint32_t hw[32];                     // entry (tx, n) at index tx * NN + n, 16 bits each
#pragma unroll
          for (int e = 0; e < 32; ++e) hw[e] = 0u;
#pragma unroll
          for (int tx = 0; tx < 7; ++tx) {
            const int src = q - tx;
            u32x2 v = u32x2{0u, 0u};
            if (tx < g.KW && src >= 0 && src < a.TW) v = *reinterpret_cast<const u32x2*>(zb + ((row << a.TWlog) + src) * 16);
#pragma unroll
            for (int n = 0; n < NN; ++n) {
              const uint32_t word = n < 2 ? v.x : v.y;
              hw[tx * NN + n] = (n & 1) ? (word >> 16) : (word & 0xffffu);
            }
          }
          unsigned char* dst = dzx + pq * a.hEB;
          const int sw = wgtr_swz(a.hEB, q);
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c < cpp)
              *reinterpret_cast<u32x4*>(dst + ((c ^ sw) << 4)) = u32x4{hw[8 * c] | (hw[8 * c + 1] << 16), hw[8 * c + 2] | (hw[8 * c + 3] << 16),
                                                                        hw[8 * c + 4] | (hw[8 * c + 5] << 16), hw[8 * c + 6] | (hw[8 * c + 7] << 16)};
        }
      };
      if (!(UEGAN_ABL_BITS(a.abl) & 8))
      switch (a.N) {
        case 1: build(std::integral_constant<int, 1>{}); break;
        case 2: build(std::integral_constant<int, 2>{}); break;
        case 3: build(std::integral_constant<int, 3>{}); break;
        default: build(std::integral_constant<int, 4>{}); break;
      }
      __syncthreads();
    }
    for (int ks = wsid; ks < ((UEGAN_ABL_BITS(a.abl) & 2) ? 0 : a.nks); ks += a.WS) {
      const unsigned char* zk = HEAD ? dzx + ks * z_ks : zb + ks * z_ks;
      const unsigned char* xk = HEAD ? xb + (ks >> 1) * a.PW * a.xrb : xb + ks * x_ks;
      const bool half1 = HEAD && (ks & 1);
      // dz fragments first, then a rolling window of x fragments PD ahead of the MFMAs that consume them
      constexpr int PD0 = TN >= 4 ? 2 : (TN == 2 ? 4 : 8), PD = PD0 < TM ? PD0 : TM;
      u32x4 af[TN], bf[PD];
      auto read_x = [&](int m) {
        // head mode, second half of a row: columns 40..63 lie beyond the patch (KW <= 7) and dzx is zero there, so the
        // h = 1 read just repeats the (clamped) h = 0 address -- any finite value will do
        const int xa = half1 ? xaddr1[m] : xaddr[m];
        const u32x2 lo = lds_read_tr16(xk + xa), hi = lds_read_tr16(xk + (half1 ? xa : xa + x_h));
        return u32x4{lo.x, lo.y, hi.x, hi.y};
      };
#pragma unroll
      for (int nf = 0; nf < TN; ++nf) {
        const u32x2 lo = lds_read_tr16(zk + zaddr[nf]), hi = lds_read_tr16(zk + zaddr[nf] + z_h);
        af[nf] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
#pragma unroll
      for (int m = 0; m < PD; ++m) bf[m] = read_x(m);
      if (do_bias) {
#pragma unroll
        for (int nf = 0; nf < TN; ++nf) accb[nf] = mfma_bf16(af[nf], ones, accb[nf]);
      }
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const u32x4 b = bf[m % PD];
        if (m + PD < TM) bf[m % PD] = read_x(m + PD);
#pragma unroll
        for (int nf = 0; nf < TN; ++nf) acc[nf][m] = mfma_bf16(af[nf], b, acc[nf][m]);
      }
    }
  }

  // partial sums -> workspace [split*WS + wsid][N][ktot]; D rows = dz channel, cols = kk
  float* ws = a.ws + (size_t)(split * a.WS + wsid) * a.pstride;
  const int col = lane & 15;
  if (do_bias && col == 0) {
#pragma unroll
    for (int nf = 0; nf < TN; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_chunk0 + nf * 16 + (lane >> 4) * 4 + r;      // head mode: rows (tx = 0, n) are rows 0 .. N-1
        if (n < a.N) ws[(size_t)a.N * a.ktot + n] = accb[nf][r];
      }
  }
  if (HEAD) {
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      const int f = f0 + wk * TM + m;
      const int ty = f / a.cfpc;
      if (ty >= g.KH) continue;
      const int kc = c_chunk0 + (f - ty * a.cfpc) * 16 + col;
#pragma unroll
      for (int nf = 0; nf < TN; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = nf * 16 + (lane >> 4) * 4 + r;                // MFMA row = (tx, n)
          if (i >= a.hE) continue;
          const int tx = i / a.N, n = i - tx * a.N;
          ws[(size_t)n * a.ktot + (ty * g.KW + tx) * g.C + kc] = acc[nf][m][r];
        }
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < TM; ++m) {
    const int f = f0 + wk * TM + m;
    int tap, kk;
    if (a.cfpc) { tap = f / a.cfpc; kk = tap * g.C + c_chunk0 + (f - tap * a.cfpc) * 16 + col; }
    else { tap = 2 * f + (col >> 3); kk = tap * g.C + (col & 7); }
    if (tap >= taps) continue;
#pragma unroll
    for (int nf = 0; nf < TN; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n_chunk0 + nf * 16 + (lane >> 4) * 4 + r;
        if (n < a.N) ws[(size_t)n * a.ktot + kk] = acc[nf][m][r];
      }
  }
}

// ---- host side: plan + launch ----
static bool g_use_wgtr = true;

struct WgradTrPlan {
  WgradTrArgs a;
  dim3 grid;
  int nsplit_eff;    // workspace splits (pixel splits x WS)
  int tn, tm;
  bool big;
};

static bool wgtr_ch_ok(int c) { return c == 8 || c == 16 || c == 32 || (c >= 64 && c % 64 == 0); }

static bool wgtr_plan(const uegan_conv_desc* d, const ConvGeom& g, WgradTrPlan& p) {
  if (!g_use_wgtr || d->dtype != UEGAN_BF16) return false;
  const int C = d->C1 + d->C2, zC = d->Cout;
  if (!wgtr_ch_ok(C) || !wgtr_ch_ok(zC)) return false;
  if (C < 64 && d->C2 != 0) { /* fine: sources are selected per 8-channel chunk */ }
  WgradTrArgs& a = p.a;
  a.g = g;
  a.abl = UEGAN_ABL_BITS(g_abl_stream);
  a.N = d->Cout_w ? d->Cout_w : d->Cout;
  a.zC = zC;
  a.ktot = d->KH * d->KW * C;
  a.pstride = a.N * a.ktot + a.N;
  a.want_bias = 1;
  const int s = d->stride;
  a.head = 0; a.hE = 0; a.hEB = 16; a.hEBlog = 4;
  if (a.N <= 4 && zC == 8 && s == 1 && d->KW >= 3 && d->KW <= 7 && d->Wo >= 32 && C >= 16) {
    a.head = 1;
    a.hE = d->KW * a.N;
    a.hEB = a.hE <= 8 ? 16 : (a.hE <= 16 ? 32 : 64);
    a.hEBlog = a.hEB == 16 ? 4 : (a.hEB == 32 ? 5 : 6);
  }
  // 32-wide one-row k-steps for the 1x1 layers and the heads; K x K layers take 16-wide tiles, which the LDS budget lets be 8 rows tall
  // instead of 2 (3x3 on 64-channel chunks: halo 1.4x instead of 2.1x the tile, four k-steps per barrier instead of two: dec1 / dec2 / dec3
  // 0.46 -> 0.37 / 0.40 -> 0.34 / 0.39 -> 0.35 ms at batch 32)
  a.TW = (s == 1 && d->Wo >= 32 && (a.head || (d->KH == 1 && d->KW == 1))) ? 32 : 16;
  a.TWlog = a.TW == 32 ? 5 : 4;
  a.dyk = a.TW == 32 ? 1 : 2;
  a.PWused = (a.TW - 1) * s + d->KW;
  a.PW = a.PWused;
  a.PWmagic = 65536 / a.PW + 1;
  a.xcb = C < 64 ? C : 64;
  a.xrb = a.xcb * 2;
  const int zcb = zC < 64 ? zC : 64;
  a.zrb = zcb * 2;
  a.xrblog = a.xrb == 128 ? 7 : (a.xrb == 64 ? 6 : (a.xrb == 32 ? 5 : 4));
  a.zrblog = a.zrb == 128 ? 7 : (a.zrb == 64 ? 6 : (a.zrb == 32 ? 5 : 4));
  // fragment slots
  const int taps = d->KH * d->KW;
  a.cfpc = a.xcb >= 16 ? a.xcb / 16 : 0;
  a.fslots = a.head ? d->KH * a.cfpc : (a.cfpc ? taps * a.cfpc : (taps + 1) / 2);
  const int F = a.fslots;
  static const int tms[7] = {1, 2, 3, 4, 5, 7, 9};
  if (F >= 20) {
    a.WK = 4;
    int best = 0, best_slots = 1 << 30;
    for (int tm : {9, 7, 5}) {
      const int slots = (F + 4 * tm - 1) / (4 * tm) * 4 * tm;
      if (slots < best_slots) { best_slots = slots; best = tm; }
    }
    p.tm = best;
  } else {
    a.WK = F > 9 ? 2 : 1;
    const int need = (F + a.WK - 1) / a.WK;
    p.tm = 9;
    for (int tm : tms)
      if (tm >= need) { p.tm = tm; break; }
  }
  a.WS = 4 / a.WK;
  a.fpb = a.WK * p.tm;
  a.nfr = (F + a.fpb - 1) / a.fpb;
  // kernel rows a slot range spans (the patch rows a block stages): ranges that are exactly one kernel row of a stride-2 conv read
  // only input rows of one parity and stage them densely (rs = 2)
  int krows = d->KH;
  if (!a.head && a.cfpc) {
    krows = 1;
    for (int fr = 0; fr < a.nfr; ++fr) {
      const int tA = fr * a.fpb / a.cfpc;
      int tB = (fr * a.fpb + a.fpb - 1) / a.cfpc;
      if (tB > taps - 1) tB = taps - 1;
      const int span = tB / d->KW - tA / d->KW + 1;
      if (span > krows) krows = span;
    }
  }
  a.rs = (s == 2 && krows == 1) ? 2 : 1;
  const int rstep = s / a.rs;
  // tile height: largest that double-buffers inside the LDS budget
  static const int th32[4] = {8, 4, 2, 1}, th16[4] = {16, 8, 4, 2};
  const int* ths = a.TW == 32 ? th32 : th16;
  int cap = ths[3];
  while (cap < d->Ho && cap < ths[0]) cap *= 2;
  auto dzx_bytes = [&](int th) { return a.head ? (th * 64 * a.hEB + 4095) / 4096 * 4096 : 0; };
  auto bytes = [&](int th, int& xb, int& zb) {
    // rounded to whole 256-lane staging rounds (4 KB): the last round of a buffer must not spill into its neighbour
    xb = (((th - 1) * rstep + krows) * a.PW * a.xrb + 4095) / 4096 * 4096;
    zb = (th * a.TW * a.zrb + 4095) / 4096 * 4096;
    return 2 * (xb + zb);
  };
  // tallest tile whose two buffers fit the 80 KB budget (2 blocks per CU), else the 152 KB variant (1 block per CU).
  // (Three buffers of shorter tiles measured slower: the extra halo rows cost more than the deeper pipeline buys.)
  auto fits = [&](int th, int kb) {
    int xb, zb;
    return th <= cap && bytes(th, xb, zb) + dzx_bytes(th) <= kb * 1024;
  };
  a.TH = 0;
  a.nbuf = 2;
  p.big = false;
  for (int i = 0; i < 4 && !a.TH; ++i)
    if (fits(ths[i], WGTR_SMALL_KB)) a.TH = ths[i];
  if (!a.TH) {
    p.big = true;
    for (int i = 0; i < 4 && !a.TH; ++i)
      if (fits(ths[i], WGTR_BIG_KB)) a.TH = ths[i];
  }
  if (!a.TH) return false;
  a.nks = a.head ? 2 * a.TH : a.TH / a.dyk;
  bytes(a.TH, a.xbytes, a.zbytes);
  {   // check the multiply-shift division used for patch decoding
    const int rows = ((a.TH - 1) * rstep + krows) * a.PW;
    for (int r = 0; r < rows; ++r)
      if (((r * a.PWmagic) >> 16) != r / a.PW) return false;
    const int nix = ((rows * a.xrb / 16) + 255) / 256, niz = ((a.TH * a.TW * a.zrb / 16) + 255) / 256;
    if (nix > (p.big ? 19 : 10) || niz > 8) return false;
  }
  p.tn = zcb >= 64 ? 4 : (zcb >= 32 ? 2 : 1);
  if (a.head) p.tn = a.hE <= 16 ? 1 : 2;
  const int cchunks = (C + 63) / 64, nblk = (zC + 63) / 64;
  a.tiles_x = (d->Wo + a.TW - 1) / a.TW;
  a.tiles_y = (d->Ho + a.TH - 1) / a.TH;
  a.tiles_total = d->B * a.tiles_x * a.tiles_y;
  const int per_split = cchunks * a.nfr * nblk;
  constexpr int target = 512;        // blocks per launch the split-K aims for (two per CU)
  int want = (target + per_split - 1) / per_split;
  if (want < 1) want = 1;
  if (want > 8 && g_tuning[UEGAN_TUNE_WGRAD_XCD] != 0) want = (want + 7) / 8 * 8;      // (a split count the XCD-aware order can use)
  if (want > a.tiles_total) want = a.tiles_total;
  a.tiles_per_split = (a.tiles_total + want - 1) / want;
  const int nsplit = (a.tiles_total + a.tiles_per_split - 1) / a.tiles_per_split;
  p.nsplit_eff = nsplit * a.WS;
  p.grid = dim3(cchunks * a.nfr, nblk, nsplit);
  a.xcd_map = 0;
  if (per_split > 1 && g_tuning[UEGAN_TUNE_WGRAD_XCD] != 0) {
    if (nsplit % 8 == 0) a.xcd_map = 8;
    else if ((nsplit == 1 || nsplit == 2 || nsplit == 4) && (nsplit * per_split) % 8 == 0 && per_split % (8 / nsplit) == 0) a.xcd_map = nsplit;
  }
  return true;
}

template <int TN, int TM>
static void wgtr_launch2(const WgradTrPlan& p, hipStream_t s) {
  if (p.a.head) {       // head mode is its own instantiation (TN <= 2): the plain kernel keeps its select-free k loop
    if constexpr (TN <= 2) {
      if (p.big) hipLaunchKernelGGL((wgrad_tr_kernel<TN, TM, true, true>), p.grid, dim3(256), 0, s, p.a);
      else hipLaunchKernelGGL((wgrad_tr_kernel<TN, TM, false, true>), p.grid, dim3(256), 0, s, p.a);
    }
    return;
  }
  if (p.big) hipLaunchKernelGGL((wgrad_tr_kernel<TN, TM, true, false>), p.grid, dim3(256), 0, s, p.a);
  else hipLaunchKernelGGL((wgrad_tr_kernel<TN, TM, false, false>), p.grid, dim3(256), 0, s, p.a);
}
template <int TN>
static void wgtr_launch1(const WgradTrPlan& p, hipStream_t s) {
  switch (p.tm) {
    case 1: wgtr_launch2<TN, 1>(p, s); break;
    case 2: wgtr_launch2<TN, 2>(p, s); break;
    case 3: wgtr_launch2<TN, 3>(p, s); break;
    case 4: wgtr_launch2<TN, 4>(p, s); break;
    case 5: wgtr_launch2<TN, 5>(p, s); break;
    case 7: wgtr_launch2<TN, 7>(p, s); break;
    default: wgtr_launch2<TN, 9>(p, s); break;
  }
}
static void wgtr_launch(const WgradTrPlan& p, hipStream_t s) {
  if (p.tn == 1) wgtr_launch1<1>(p, s);
  else if (p.tn == 2) wgtr_launch1<2>(p, s);
  else wgtr_launch1<4>(p, s);
}
