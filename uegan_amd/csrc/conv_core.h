// Shared pieces of the convolution translation units (conv.hip, conv_patch_*.hip): launch-timing scope, MFMA wrappers,
// gather geometry, the argument block of the gather-GEMM kernels.  Internal (non-ABI).
#pragma once
#include "common.h"
#include "conv_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <utility>
#include <vector>

namespace uegan {

// ----------------------------------------------------------------------------------------------------
// Optional per-launch timing of the MFMA kernels with HIP events on the launch stream (bench.py's
// `roofline` object).  Off by default; zero cost when off.
// ----------------------------------------------------------------------------------------------------
struct ProfRecord {
  hipEvent_t start, stop;
  int kernel_id;
  double flops, bytes;      // algorithmic: 2 x MACs; source tensor read once + result written once
};
extern bool g_prof_on;
extern std::vector<ProfRecord> g_prof_records;
extern std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
extern size_t g_prof_used;

// key = kind<<28 | bf16<<24 | BN<<12 | KS<<8 | MODE<<4 | log2(TH/8)<<1 | glds   (kind: 0 gather-GEMM, 1 patch, 2 wgrad, 3 transpose-read wgrad: BN=TN, KS=TM)
static inline int prof_key(int kind, bool bf16, int bn, int ks, int mode, int th, bool glds) {
  return (kind << 28) | ((bf16 ? 1 : 0) << 24) | (bn << 12) | (ks << 8) | (mode << 4) | ((th == 32 ? 2 : (th == 16 ? 1 : (th == 4 ? 3 : 0))) << 1) | (glds ? 1 : 0);
}
static void prof_kernel_name(int key, char* buf, size_t n) {
  const int kind = (key >> 28) & 7, bn = (key >> 12) & 0xfff, ks = (key >> 8) & 15, mode = (key >> 4) & 15;
  const char* dt = ((key >> 24) & 1) ? "bf16" : "f32";
  if (kind == 0) snprintf(buf, n, "conv_gemm_kernel<%s,BN=%d,%s>", dt, bn, (key & 1) ? "glds" : "regstage");
  else if (kind == 1) snprintf(buf, n, "conv_patch_kernel<%s,BN=%d,KS=%d,MODE=%d,TH=%d>", dt, bn, ks, mode, 8 << ((key >> 1) & 3));
  else if (kind == 4) snprintf(buf, n, "conv_stream_kernel<%s,TN=%d,PF=%d,MODE=%d>", dt, bn, ks, mode);
  else if (kind == 5) snprintf(buf, n, "conv_s2fwd_kernel<%s,BN=%d,K=%d,TH=%d>", dt, bn, ks, 8 << ((key >> 1) & 3));
  else if (kind == 6) snprintf(buf, n, "conv_toep_kernel<%s,K=%d,MODE=%d>", dt, ks, mode);
  else if (kind == 7 && ((key >> 1) & 3) == 1) snprintf(buf, n, "conv_tall_kernel<%s,BN=%d,KS=%d,MODE=%d%s>", dt, bn, ks, mode, (key & 1) ? "" : ",POOL");
  else if (kind == 7 && ((key >> 1) & 3) == 3) snprintf(buf, n, "conv_flat_kernel<%s,BN=%d,KS=%d>", dt, bn, ks);
  else if (kind == 7) snprintf(buf, n, "conv_wide_kernel<%s,BN=%d,KS=%d,MODE=%d>", dt, bn, ks, mode);
  else if (kind == 3) snprintf(buf, n, "wgrad_tr_kernel<%s,TN=%d,TM=%d%s>", dt, bn, ks, (key & 1) ? ",big" : "");
  else snprintf(buf, n, "conv_wgrad_kernel<%s,BN=%d>", dt, bn);
}

struct ProfScope {
  bool on;
  hipStream_t s;
  ProfRecord rec;
  ProfScope(int kernel_id, double flops, hipStream_t stream, double bytes = 0.0) : on(g_prof_on && g_prof_used < g_prof_pool.size()), s(stream) {
    if (!on) return;
    rec.bytes = bytes;
    rec.start = g_prof_pool[g_prof_used].first;
    rec.stop = g_prof_pool[g_prof_used].second;
    ++g_prof_used;
    rec.kernel_id = kernel_id;
    rec.flops = flops;
    (void)hipEventRecord(rec.start, s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(rec.stop, s);
    g_prof_records.push_back(rec);
  }
};

// ----------------------------------------------------------------------------------------------------
// MFMA wrappers.  Fragment layouts (gfx950):
//   16x16x32 bf16: A lane l = A[i=l&15][k=8*(l>>4)+e], B lane l = B[k=8*(l>>4)+e][j=l&15], e=0..7
//   16x16x4  f32 : A lane l = A[i=l&15][k=l>>4],       B lane l = B[k=l>>4][j=l&15]
//   C/D          : lane l, reg r -> row i = 4*(l>>4)+r, col j = l&15
// ----------------------------------------------------------------------------------------------------
// (mfma_bf16 = "the MFMA of the 16-bit storage format": bf16, or fp16 in the -DUEGAN_HALF_FP16 build -- same rate, same fragment layout)
#ifdef UEGAN_HALF_FP16
typedef _Float16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
#else
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
#endif
__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// element-wise sum of two 16-byte operand chunks in the storage format (8 halves: fp32 add, rounded back once; 4 floats): the mirrored
// images of a reflection-padded data gradient folded into the pixel operand (conv_patch.h MODE 2, conv_wide.hip MODE 2)
template <typename T> __device__ __forceinline__ u32x4 add_frag(u32x4 a, u32x4 b);
template <> __device__ __forceinline__ u32x4 add_frag<bf16_t>(u32x4 a, u32x4 b) {
  u32x4 r;
#pragma unroll
  for (int d = 0; d < 4; ++d) r[d] = pack_bf16x2(half_lo_to_f32(a[d]) + half_lo_to_f32(b[d]), half_hi_to_f32(a[d]) + half_hi_to_f32(b[d]));
  return r;
}
template <> __device__ __forceinline__ u32x4 add_frag<float>(u32x4 a, u32x4 b) {
  u32x4 r;
#pragma unroll
  for (int d = 0; d < 4; ++d) r[d] = f32_to_bits(bits_to_f32(a[d]) + bits_to_f32(b[d]));
  return r;
}

// one K-step (32 reduction elements) of fragment products. a/b are the 16-byte LDS chunks of this lane.
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int NCHUNK = 1;  // 16B chunks per lane per 32-wide K step
  // lane (r=l&15, g=l>>4) reads elements k = 8g..8g+7
  static __device__ __forceinline__ int chunk_byte(int g, int /*c*/) { return g * 16; }
  static __device__ __forceinline__ void step(const u32x4* a, const u32x4* b, f32x4& acc) { acc = mfma_bf16(a[0], b[0], acc); }
};
template <> struct Mma<float> {
  static constexpr int NCHUNK = 2;
  // chunk c covers k = 16c + 4g .. 16c + 4g + 3; element j of the chunk feeds MFMA #j of that chunk.  Both
  // operands use the same k permutation, so the sum over k is unchanged.
  static __device__ __forceinline__ int chunk_byte(int g, int c) { return (c * 16 + g * 4) * 4; }
  static __device__ __forceinline__ void step(const u32x4* a, const u32x4* b, f32x4& acc) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      acc = mfma_f32(bits_to_f32(a[c].x), bits_to_f32(b[c].x), acc);
      acc = mfma_f32(bits_to_f32(a[c].y), bits_to_f32(b[c].y), acc);
      acc = mfma_f32(bits_to_f32(a[c].z), bits_to_f32(b[c].z), acc);
      acc = mfma_f32(bits_to_f32(a[c].w), bits_to_f32(b[c].w), acc);
    }
  }
};

// ----------------------------------------------------------------------------------------------------
// Gather geometry shared by forward, dgrad and wgrad.  Every tensor the MFMA kernels touch has a channel count
// that is a multiple of one 16-byte chunk (8 bf16 / 4 fp32): 3-channel images and 1/3-channel heads are carried
// zero-padded (uegan_amd/ops.py), so every gather is one aligned 16-byte load.
// ----------------------------------------------------------------------------------------------------
struct ConvGeom {
  int B, IH, IW;   // spatial dims of the tensor being gathered from
  int C1, C2, C;   // its (padded) channels: two sources, C = C1 + C2
  int OH, OW;      // grid of GEMM pixel rows
  int KH, KW, stride, pad, pad_mode;
  int mode;        // 0: forward gather (rows = conv outputs, source = conv input)
                   // 1: dgrad gather   (rows = conv inputs,  source = dz on the conv-output grid)
};

// Source coordinate along one axis. Returns -1 when the tap contributes nothing.
//   forward: s = pad_map(o*stride + t - pad)
//   dgrad  : image `img` of input coordinate o in padded space (0: itself, 1: mirrored across 0,
//            2: mirrored across n-1; adjoint of reflection padding), then s = (pp + pad - t)/stride.
__device__ __forceinline__ int src_coord(const ConvGeom& g, int o, int t, int img, int in_n, int out_n) {
  if (g.mode == 0) {
    int s = o * g.stride + t - g.pad;
    if (g.pad_mode == UEGAN_PAD_REFLECT) return reflect_idx(s, in_n);
    return (s >= 0 && s < in_n) ? s : -1;
  }
  int pp;
  if (img == 0) {
    pp = o;
  } else if (img == 1) {
    if (o < 1 || o > g.pad) return -1;
    pp = -o;
  } else {
    if (o < out_n - 1 - g.pad || o > out_n - 2) return -1;
    pp = 2 * (out_n - 1) - o;
  }
  const int t2 = pp + g.pad - t;
  if (t2 < 0) return -1;
  int s = t2;
  if (g.stride == 2) {             // strides are 1 or 2 (checked at the API): no integer division in the inner loop
    if (t2 & 1) return -1;
    s = t2 >> 1;
  }
  return s < in_n ? s : -1;
}

__device__ __forceinline__ bool has_image(const ConvGeom& g, int o, int img, int out_n) {
  if (img == 0) return true;
  if (g.mode == 0 || g.pad_mode != UEGAN_PAD_REFLECT) return false;
  if (img == 1) return o >= 1 && o <= g.pad;
  return o >= out_n - 1 - g.pad && o <= out_n - 2;
}

// XOR term on the 16-byte chunk position of row r of a 128-byte-row LDS tile whose 16x16 MFMA fragments are read with ds_read_b128 from
// 16 consecutive rows starting ANYWHERE (a tap offset into a pixel patch).  The LDS serves a wave's ds_read_b128 in four groups of 16
// lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- i.e. rows j in {0-3, 12-15} of K
// chunk g together with rows {4-11} of chunk g ^ 1, and a group is conflict free when its 16 chunks fall on the 16 different 16-byte
// slots of the 256-byte bank line: slot = 8 (r & 1) + position.  With position = q ^ 2 ((r >> 1) & 3) the two rows of equal parity that
// share (r >> 1) & 3 are 8 rows apart, hence in different halves of the group, hence read chunks that differ in bit 0: all 16 slots are
// distinct from any first row.  (The round-1 term (r >> 1) & 7 is conflict free only from first rows that are multiples of 16: measured
// SQ_LDS_BANK_CONFLICT = 18-31 % of the LDS cycles of the patch kernels, 40 % of the streaming kernel's, profiles/r03_lds_bank.txt.)
__device__ __forceinline__ int swz128(int r) { return ((r >> 1) & 3) << 1; }
// the same for 64-byte rows (4 chunks): slot = 4 (r & 3) + position; rows of equal r & 3 that share (r >> 2) & 1 are 8 apart
__device__ __forceinline__ int swz64(int r) { return ((r >> 2) & 1) << 1; }

// 16 zero bytes in global memory: the source of every masked lane of a direct-to-LDS load
static __device__ __attribute__((aligned(16))) const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// one lane's 16 bytes global -> LDS without a VGPR round trip; the destination is (wave-uniform base) + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Direct-to-LDS load issued from inline asm, so that hipcc does not know an LDS DMA is pending: with the builtin form it
// puts `s_waitcnt vmcnt(0)` in front of the first transpose read of every tile (the intrinsic carries no alias scope),
// which serialises the load of tile t+1 with the MFMAs of tile t (measured: total = load time + compute time).
// The waits for these loads are therefore also asm (wgtr_wait_loads); hipcc would drop a builtin s_waitcnt it believes
// redundant.  M0 (LDS destination base) is saved and restored inside the statement.
// (uniform 64-bit base in SGPRs + per-lane 32-bit byte offset; lds_dst: wave-uniform LDS byte address)
__device__ __forceinline__ void wgtr_glds16(const unsigned char* base, uint32_t off, unsigned char* lds_wave_base) {
#ifdef UEGAN_EMU
  glds16(base + off, lds_wave_base);
#else
  const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
#endif
}
// per-lane 64-bit source address (slow staging paths)
__device__ __forceinline__ void wgtr_glds16(const void* src, unsigned char* lds_wave_base) {
#ifdef UEGAN_EMU
  glds16(src, lds_wave_base);
#else
  const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
#endif
}
__device__ __forceinline__ void wgtr_wait_loads() {
#ifndef UEGAN_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ----------------------------------------------------------------------------------------------------
// Gather-GEMM kernel: out[pixel][n] = epi( sum_{image, tap, c} gather(pixel, tap, c) * w[n][tap][c] )
//
//   * block tile: 8 x 16 output pixels (2-D, so a KxK window re-reads a 10x18 patch from L2 instead of 3 rows, and
//     only border tiles pay for reflected images) x BN channels; 4 waves, each a (128/WARPS_M) x (BN/WARPS_N) sub-tile
//   * dgrad with stride 2: a tile holds pixels of ONE parity class (oy%2, ox%2), so exactly the taps that hit
//     integer output coordinates are iterated (no MFMA work on structural zeros)
//   * K step = 128 bytes per row (64 bf16 / 32 fp32).  LDS rows are 128 B, the 16-byte chunk q of row r lives at
//     position q ^ ((r>>1)&7) in this (generic) kernel, whose fragment rows start at multiples of 16: conflict free.  The
//     patch-resident kernels read 16 consecutive rows from ANY first row (tap offsets) and use swz128() below
//   * staging: GLDS=true  -> global_load_lds_dwordx4 (direct to LDS, swizzle applied on the per-lane SOURCE address),
//              GLDS=false -> 16-byte global loads to VGPRs, ds_write_b128 after the MFMAs of the previous step;
//     two LDS buffers, one __syncthreads() per K step
// ----------------------------------------------------------------------------------------------------
struct ConvArgs {
  ConvGeom g;
  const void* in1;
  const void* in2;
  const void* w;       // [N][Kp]
  const float* bias;   // [nbias] or null
  const float* scale;  // device scalar(s) or null: image b is multiplied by scale[scale_group ? b / scale_group : 0]
  int scale_group;     // images per scale group (one spectral-norm sigma per group of a batched multi-pass forward); 0: one scalar
  void* out;           // NHWC [B][OH][OW][N]   (channels [0, n_out1) when out2 is set)
  void* out2;          // optional second destination (virtual-concat dgrad): channels [n_out1, N), NHWC stride N - n_out1
  int n_out1;
  int N, Kp, act, nbias;
  int nty, ntx;        // tiles per (parity class of an) image
  int frame;           // tile subset: 0 all tiles, 1 only the border tiles around the tile rectangle [fy0,fy1) x [fx0,fx1)
  int fy0, fy1, fx0, fx1;      // (the ones that can carry mirrored images of a reflection-padded dgrad), 2 only the rectangle
  const void* mask;    // optional (dgrad, one destination): the activated tensor this gradient is for, same shape as out;
  int mask_act;        // the epilogue multiplies by act'(mask) -- the producer's deferred activation gradient
  void* pool_out = nullptr;    // forward, optional: NHWC [B][OH/2][OW/2][N], the 2x2 max-pool of `out` (losses.py:74-104: every VGG pool follows a
                               // conv + ReLU), written by the epilogue of the kernels that can (they set pool_done), else by the caller
  int pool_done = 0;
  void* pool_idx = nullptr;    // with pool_out, optional: bytes [B][OH/2][OW/2][N], the window position (dy * 2 + dx) of the first maximum -- stored for images
                               // b < n_idx by the kernels that can (only those may set pool_done when this is wanted)
  int n_idx = 0;
  int n_full = 1 << 30;         // with pool_out: images b >= n_full need only the POOLED result (no gradient will flow through them: the reference batch of the
                               // fidelity loss) -- a kernel with a pooling epilogue skips their full-resolution store; the others ignore this and write everything
  int xcd_map = 0;             // conv_wide_kernel: XCD-aware (tile, channel block) mapping (see there)
  // Reflection-padded stride-1 data gradients, split by conv_interior_run (conv_wide.hip): the pixel rectangle [rect_y0, rect_y1) x [rect_x0,
  // rect_x1) holds no pixel with a mirrored image; conv_wide_kernel / conv_tall_kernel compute it image-free (their tiles start at the
  // rectangle's origin), the patch kernel's MODE 2 launch then takes the frame around it (border_only)
  int rect_y0 = 0, rect_y1 = 0, rect_x0 = 0, rect_x1 = 0;
  int border_only = 0;
  // conv_stream_kernel<..., STATS>: per-(image, channel) sums of the activated result (sum, sum of squares) ride along with the forward -- per
  // (block, image of the block's tile range, wave) partials [((block * 2 + j) * NW + wave) * Cs + n][2], folded by stream_stats_finalize_kernel
  // (uegan_conv2d_fwd_stats: the InstanceNorm behind the attention convs, models.py:227, needs no pass of its own over the tensor)
  float* stats_part = nullptr;
  int stats_tpi = 0;              // tiles per image
  // Round 6, forward only (uegan_conv2d_fwd_ex; conv_stream_kernel<..., PR, EPX>, conv_toep_kernel<..., EX>).  A tensor may travel as a hi + lo
  // PAIR of 16-bit planes, value = hi + lo with hi = rn16(v), lo = rn16(v - hi) (~2 x the significant bits of the storage format): the lo plane of
  // either source (same shape and layout as the source), of the packed weights (same layout as w) and of the result.
  const void* in1_lo = nullptr;
  const void* in2_lo = nullptr;
  const void* w_lo = nullptr;
  void* out_lo = nullptr;
  // ... and the epilogue may multiply the activated result by a second tensor of the output's shape (models.py:69 `y4.mul(x1)`, formed from the
  // fp32 accumulator): out_mul (+ out_mul_lo) = act(...) * (mul + mul_lo); `out` still receives act(...) itself
  const void* mul = nullptr;
  const void* mul_lo = nullptr;
  void* out_mul = nullptr;
  void* out_mul_lo = nullptr;
  // ... or finish the generator (models.py:70-72, conv_toep_kernel only): res_out[NCHW fp32, nbias channels] = clamp(act(...) + res_x, -1, 1) from the
  // fp32 accumulator (`out` still receives act(...): the backward's tanh').  Images b >= res_split belong to a second image set (res_x2 / res_out2,
  // indexed b - res_split): the generator pass over two batch-concatenated sets, trainer.py:85 + :112
  // conv_s2fwd_kernel: channel mask of the source gather (0: none).  With C = 2 C1 and src_wrap = C1 - 1 the source's C1 channels are read twice per
  // tap -- against a weight matrix that holds, per tap, the hi part of the C1 weights and then their lo part (uegan_pack_weights_pair, dup_cin = 2):
  // the weight PAIR of a stride-2 forward (G.enc2 in the precise mode) on the unchanged 64-channel kernel
  int src_wrap = 0;
  const float* res_x = nullptr;
  const float* res_x2 = nullptr;
  float* res_out = nullptr;
  float* res_out2 = nullptr;
  int res_split = 1 << 30;
  // Split-K forward (uegan_conv2d_fwd_splitk; conv_patch_kernel / conv_s2fwd_kernel <..., SPLITK>): on a grid that leaves most CUs without a block (the
  // deep layers of a single-image inference) the launchers split the 64-channel chunks of the K loop over gridDim.z = kparts blocks per tile, each writing
  // its fp32 partial sums to kws[part][pixel][N]; splitk_reduce_kernel adds them in part order and applies the epilogue.  kws / kws_bytes: the caller's
  // workspace (host side: the launchers decide whether to use it); kparts / kchunks (chunks per part): what the kernel sees
  float* kws = nullptr;
  size_t kws_bytes = 0;
  int kparts = 0, kchunks = 0;
};

// parts of a split-K forward: the largest power of two that keeps blocks * parts within the chip's 256 CUs with at least one 64-channel chunk per part, and
// fits the workspace; 1: no split.  *kchunks = chunks per part (every part has at least one)
static inline int splitk_parts(const ConvArgs& a, int blocks, int nchunk, int* kchunks) {
  *kchunks = nchunk;
  if (!a.kws || a.g.mode != 0 || a.frame || a.out2 || a.mask || a.pool_out || a.stats_part || a.out_lo || a.mul || a.res_out || blocks <= 0) return 1;
  int parts = 1;
  while (parts * 2 <= nchunk && blocks * parts * 2 <= 256) parts *= 2;
  const size_t out_bytes = (size_t)a.g.B * a.g.OH * a.g.OW * a.N * sizeof(float);
  while (parts > 1 && (size_t)parts * out_bytes > a.kws_bytes) parts >>= 1;
  if (parts < 2) return 1;
  *kchunks = (nchunk + parts - 1) / parts;
  return (nchunk + *kchunks - 1) / *kchunks;
}

// out[pixel][n] = act(scale * sum_part kws[part][pixel][n] + bias[n]) behind a split-K launch (conv.hip: splitk_reduce_kernel; 16-bit storage types)
int splitk_reduce_launch(const ConvArgs& a, hipStream_t s);


constexpr int CONV_TH = 8, CONV_TW = 16, CONV_BM = CONV_TH * CONV_TW;
constexpr int CONV_ROWB = 128;   // bytes per LDS row = one K step

}  // namespace uegan
