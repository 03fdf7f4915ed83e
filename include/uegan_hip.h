/* libuegan_hip.so -- C ABI of the MI355X-native UEGAN hot path.
 *
 * The reference (eezkni/UEGAN) has no FFI / operator-plugin interface: its hot path is the Python class
 * API of models.py / losses.py dispatching to ATen/cuDNN kernels.  Every entry point
 * below replaces one group of those implicit kernels; the comment on each cites the reference lines whose
 * arithmetic it implements.  The Python mirror of the reference classes (uegan_amd/models.py, losses.py)
 * binds these through ctypes.
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures: device pointers are `void*` / `float*`, the stream is the
 *    raw hipStream_t passed as `void*` (NULL = default stream).
 *  - every function enqueues asynchronously on `stream`, allocates nothing, never synchronises, and returns
 *    0 on success or a negative UEGAN_E_* code; `uegan_last_error()` returns a thread-local message.
 *  - activations are NHWC ("[B][H][W][C]", C contiguous) in storage dtype T = float or bfloat16
 *    (`dtype` argument); all arithmetic and all reductions are fp32.  Master weights, gradients of
 *    weights, statistics and loss scalars are fp32.  Image tensors at the module boundary are the
 *    reference's NCHW fp32 (data_loader.py:79-81).
 *  - the caller owns every buffer including workspaces (`*_workspace_bytes` queries).
 */
#ifndef UEGAN_HIP_H_
#define UEGAN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UEGAN_VERSION 105

enum { UEGAN_OK = 0, UEGAN_E_INVALID = -1, UEGAN_E_HIP = -2, UEGAN_E_UNSUPPORTED = -3 };
/* UEGAN_BF16 = "the 16-bit storage format of this build": bfloat16 in libuegan_hip.so; IEEE fp16 in libuegan_hip_f16.so, the same sources
 * compiled with -DUEGAN_HALF_FP16 (same ABI, same bytes and MFMA rate, 11 instead of 8 significant bits; gradients then need a loss scale:
 * uegan_amd.trainer.Trainer(loss_scale=...)).  A process may load both libraries; a tensor belongs to the build that wrote it. */
enum { UEGAN_F32 = 0, UEGAN_BF16 = 1 };
enum { UEGAN_PAD_ZERO = 0, UEGAN_PAD_REFLECT = 1 };
enum { UEGAN_ACT_NONE = 0, UEGAN_ACT_LRELU = 1, UEGAN_ACT_RELU = 2, UEGAN_ACT_TANH = 3,
       UEGAN_ACT_SIGMOID = 4,                       /* conv epilogues too: the prediction heads of 'ls' / 'rals' (models.py:175-176) */
       UEGAN_ACT_SWISH = 5, UEGAN_ACT_SELU = 6 };   /* uegan_affine_act_* only (models.py:255-258) */
enum { UEGAN_IMPL_AUTO = 0, UEGAN_IMPL_MFMA = 1, UEGAN_IMPL_DIRECT = 2, UEGAN_IMPL_MFMA_REGSTAGE = 3, UEGAN_IMPL_MFMA_GENERIC = 4 };

typedef void* uegan_stream_t;

int uegan_version(void);
const char* uegan_last_error(void);
/* select the convolution implementation (AUTO/MFMA = MFMA implicit GEMM staged with direct-to-LDS loads;
 * MFMA_REGSTAGE = generic kernel staged through VGPRs (A/B); MFMA_GENERIC = never use the patch-resident kernel; DIRECT = scalar reference kernels that exist for
 * cross-checking on the GPU). Returns the previous setting. */
int uegan_set_conv_impl(int impl);
/* Launch-variant thresholds (process-wide; the library itself never reads the environment, so nothing outside an explicit call can
 * change which kernel a layer runs on).  The defaults are the measured ones; the tests lower them to reach every variant on small maps.
 * `previous` (optional) receives the old value. */
enum {
  UEGAN_TUNE_SMALL_GRID = 0,     /* default 256: a launch with fewer workgroups takes the smaller-tile variant; a reflection-padded data gradient is split into image-free tiles + frame only when the former are at least this many */
  UEGAN_TUNE_FOLD_MAX = 1,       /* default -1 (rule in conv.hip: dgrad_folds); n >= 0: reflection-padded data gradients of maps with
                                    <= n pixels take the pad-grid + fold route, 0 = never */
  UEGAN_TUNE_HEADS_NO_CG = 2,    /* default 0; 1: one-output-channel heads always one thread per pixel */
  UEGAN_TUNE_WIDE_MIN_GRID = 3,  /* default 192: minimum workgroups for the one-wave-per-SIMD kernels (conv_wide.hip); < 0: off */
  UEGAN_TUNE_TALL_MIN_GRID = 4,  /* default 192: the same for the 64- / 128-channel form (conv_tall_kernel); < 0: off */
  UEGAN_TUNE_TALL_RPW = 5,       /* default 0: 128-channel blocks of conv_tall_kernel on 8-row tiles (two blocks per CU) below 512 input channels, 16-row tiles from there; 2 / 4: always 8- / 16-row tiles */
  UEGAN_TUNE_TALL_REFLECT = 6,   /* default 1: reflection-padded stride-1 3x3 data gradients with 128 k output channels run WHOLE on conv_tall_kernel, the mirrored images folded into the pixel operand of the border tiles; 0: image-free rectangle + frame launch on the patch kernel (round 4) */
  UEGAN_TUNE_FLAT_S2 = 7,       /* default 1: stride-2 data gradients with 64 / 128 k input channels run as ONE conv_flat_kernel launch over the padded grid (all four parity classes, flattened positions) + fold; 0: one parity-class launch each (round 4) */
  UEGAN_TUNE_TOEP_HEADS = 8,    /* default 1: forwards with <= 4 output channels on 64 / 128 input channels (the discriminator's prediction heads d2, d3) run on the Toeplitz MFMA kernel, one 32-channel chunk at a time; 2: up to 1024 input channels; 0: the vector-ALU head kernel (round 4) */
  UEGAN_TUNE_HEADS_MFMA = 9,    /* default 1: uegan_conv2d_dgrad_padded takes the one-channel prediction heads (head_dgrad_mfma_kernel); 0: it declines them (the caller's uegan_conv2d_dgrad_ws then runs the vector-ALU kernel of round 3) */
  UEGAN_TUNE_FWD_STATS = 10,    /* default 1: uegan_conv2d_fwd_stats lets the streaming kernel emit the per-channel moments; 0: it declines (plain forward, the caller's moments pass) */
  UEGAN_TUNE_WGRAD_XCD = 11,    /* default 1: wgrad_tr_kernel orders its blocks so that all blocks of one pixel split run behind one XCD's L2 (needs a split count that is a multiple of 8); 0: launch order */
  UEGAN_TUNE_COUNT = 12
};
int uegan_set_tuning(int knob, int value, int* previous);
/* on-device check of the MFMA fragment layouts this library assumes (A=I, asymmetric B). 0 = ok. */
int uegan_selftest_mfma(void* scratch_4096_floats, uegan_stream_t stream);

/* Per-launch timing of the MFMA convolution kernels with HIP events recorded on the launch stream (used by
 * bench.py's `roofline` object). begin: allocate/enable up to max_records launches; end: synchronise the events and
 * return one aggregated entry per kernel instantiation (algorithmic FLOPs = 2 * conv MACs of each launch, algorithmic bytes =
 * tensors in + out once). */
typedef struct {
  char name[96];
  int64_t launches;
  double total_ms;
  double total_flops;
  double total_bytes;      /* algorithmic HBM bytes: the source tensor(s) read once + the result written once */
} uegan_profile_entry;
int uegan_profile_begin(int max_records);
int uegan_profile_end(uegan_profile_entry* out, int max_entries, int* n_entries);

/* ---------------------------------------------------------------------------------------------------
 * Convolution family.  Replaces nn.ReflectionPad2d + nn.Conv2d (+bias) + LeakyReLU/ReLU/tanh and their
 * autograd backward (models.py:80-84, 92-98, 161-166, 173-178; torchvision VGG conv3x3 pad 1 + ReLU,
 * losses.py:68-114).  dilation = 1, groups = 1 everywhere in the reference.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;            /* UEGAN_F32 / UEGAN_BF16: activations and packed weights */
  int32_t B, H, W;          /* conv INPUT batch / height / width */
  int32_t C1, C2;           /* channels of the input TENSORS: two sources (virtual torch.cat, models.py:55,59,63,67), C2 = 0 for
                               one.  Every tensor channel count is a multiple of one 16-byte chunk (8 bf16 / 4 fp32): 3-channel
                               images and 1/3-channel heads are carried zero-padded. */
  int32_t Ho, Wo, Cout;     /* conv OUTPUT dims (Cout = channels of the output TENSOR, padded): Ho = (H + 2*pad - KH)/stride + 1 */
  int32_t KH, KW, stride, pad;
  int32_t pad_mode;         /* UEGAN_PAD_REFLECT (G, D) or UEGAN_PAD_ZERO (VGG) */
  int32_t act;              /* epilogue activation of the forward */
  int32_t Cin_w, Cout_w;    /* TRUE weight dims (OIHW master) when smaller than the padded tensor dims; 0 = same */
  int32_t Cin_total;        /* weight gradient: input channels per row of the OIHW destination when the convolution uses only the first
                               Cin_w input channels of a wider master weight (a column slice: the attention module's fuse conv,
                               models.py:230-237, see uegan_pack_weights_slice); the other columns are not written.  0 = Cin_w */
  int32_t scale_group;      /* forward / data gradient: images per scale group -- image b is multiplied by scale[b / scale_group]
                               (several applications of one spectral-normalised layer batched into one launch, each with the
                               sigma of ITS power-iteration state, models.py:185-188); 0 = `scale` is one scalar */
} uegan_conv_desc;

/* padded K (row length, in elements) of a packed weight matrix with k = KH*KW*C true columns */
int64_t uegan_packed_k(int64_t k);
/* OIHW fp32 master weight [Cout][Cin][KH][KW] -> two packed, zero-padded copies in dtype:
 *   w_ohwi [Cout_pad][packed_k(KH*KW*Cin_pad)]  (forward / wgrad ordering, k = (kh,kw,ci))
 *   w_ihwo [Cin_pad ][packed_k(KH*KW*Cout_pad)] (dgrad ordering,          k = (kh,kw,co))   (may be NULL) */
int uegan_pack_weights(int dtype, const float* w_oihw, int Cout, int Cin, int KH, int KW, int Cout_pad, int Cin_pad, void* w_ohwi,
                       void* w_ihwo, uegan_stream_t stream);
/* the same from the first Cin input channels of a master weight with Cin_total >= Cin input channels (rows of Cin_total*KH*KW) */
int uegan_pack_weights_slice(int dtype, const float* w_oihw, int Cout, int Cin, int Cin_total, int KH, int KW, int Cout_pad, int Cin_pad,
                             void* w_ohwi, void* w_ihwo, uegan_stream_t stream);
/* ... and the weights as a hi + lo PAIR of 16-bit matrices (round 6, 16-bit storage only): w_ohwi_lo (optional, same shape as w_ohwi) receives what the
 * rounding of each element left, rn16(w - rn16(w)), so that w_ohwi + w_ohwi_lo carries ~2 x the significant bits of the storage format
 * (uegan_conv2d_fwd_ex multiplies by both).  dup_cin = 1: input channels [Cin, 2 Cin) of the OHWI copies repeat [0, Cin) -- for a source that carries
 * ITS OWN lo plane in those channels (uegan_nchw_to_nhwc_pair: the 3-channel image in the 8-channel pixels of the generator's first convolution);
 * dup_cin = 2: they hold the LO part of [0, Cin) instead -- the pair inside one matrix (uegan_conv_ex::w_interleaved). */
int uegan_pack_weights_pair(int dtype, const float* w_oihw, int Cout, int Cin, int Cin_total, int KH, int KW, int Cout_pad, int Cin_pad,
                            void* w_ohwi, void* w_ihwo, void* w_ohwi_lo, int dup_cin, uegan_stream_t stream);
/* Every conv weight an optimizer step touched re-packed by ONE launch (trainer.py:337-338 updates all of a network's weights at once):
 * a device table of entries, each the arguments of uegan_pack_weights_slice plus `start`, the running sum of the entries'
 * Cout_pad*Kp + Cin_pad*Kp2 destination elements (w_ihwo must not be NULL here); `total` = that sum over all entries. */
typedef struct uegan_pack_entry {
  const float* w_oihw;
  void* w_ohwi;
  void* w_ihwo;
  int64_t start;
  int32_t Cout, Cin, Cin_total, KH, KW, Cout_pad, Cin_pad, Kp, Kp2;
  int32_t flags;            /* bits 0-1: dup_cin of uegan_pack_weights_pair */
  void* w_ohwi_lo;          /* optional: the lo part of the OHWI copy (uegan_pack_weights_pair); NULL = none */
} uegan_pack_entry;
int uegan_pack_weights_multi(int dtype, const uegan_pack_entry* table_dev, int n_entries, int64_t total, uegan_stream_t stream);
/* y = act(scale * conv(pad(x), w) + bias);  bias (fp32[Cout]) and scale (device fp32 scalar, the 1/sigma of
 * spectral norm, torch spectral_norm compute_weight) may be NULL. */
int uegan_conv2d_fwd(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias,
                     const float* scale, void* y, uegan_stream_t stream);
/* uegan_conv2d_fwd with a workspace the kernels may use to split the K loop (round 6).  The deep layers of a SINGLE-image forward (models.py:54-66 at batch
 * 1: G.enc4 / enc5 / dec1 / dec2 of one 512 x 512 image are 64-128 workgroups of 18-72 K steps on a 256-CU device) are launched with gridDim.z = parts blocks
 * per tile, each over its share of the 64-channel chunks, fp32 partial sums into `workspace` ([part][pixel][Cout]), and a second kernel adds the parts in
 * order (deterministic) and applies scale, bias and activation.  uegan_conv2d_fwd_splitk_workspace_bytes: an upper bound of what such a launch of this layer
 * uses, 0 if no kernel would split it (call uegan_conv2d_fwd).  With a NULL / short workspace, or on a layer no kernel splits, the call IS uegan_conv2d_fwd.
 * 16-bit storage types only (fp32: plain forward).  `workspace`: device memory, 16-byte aligned, not read before it is written (no initialisation needed). */
size_t uegan_conv2d_fwd_splitk_workspace_bytes(const uegan_conv_desc* d);
int uegan_conv2d_fwd_splitk(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias, const float* scale, void* y,
                            void* workspace, size_t workspace_bytes, uegan_stream_t stream);
/* Forward convolution + the per-(image, channel) moments of its result, for an InstanceNorm that follows the conv directly (the generator's
 * attention modules: models.py:227, 230-237; non-affine, biased variance, eps 1e-5).  Where the streaming kernel takes the layer (16-bit storage,
 * <= 64 input and 32 / 64 output channels on a full-resolution map) the sums ride along in its epilogue and a small kernel folds the per-block
 * partials in a fixed order: mean[B][Cout], rstd[B][Cout] = 1 / sqrt(var + eps) (eps < 0: the biased variance itself), *produced = 1.
 * Otherwise the call is uegan_conv2d_fwd, *produced = 0, and the caller computes the moments from y (uegan_moments / uegan_instnorm_fwd).
 * workspace: uegan_conv2d_fwd_stats_workspace_bytes(desc) (0 = no such kernel for this layer). */
size_t uegan_conv2d_fwd_stats_workspace_bytes(const uegan_conv_desc* d);
int uegan_conv2d_fwd_stats(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias, const float* scale, void* y,
                           float* mean, float* rstd, float eps, void* workspace, size_t workspace_bytes, int* produced, uegan_stream_t stream);
/* y = (x - mean[b][c]) * rstd[b][c] with the moments given (uegan_conv2d_fwd_stats): the apply half of uegan_instnorm_fwd */
int uegan_instnorm_apply(int dtype, const void* x, void* y, const float* mean, const float* rstd, int B, int HW, int C, uegan_stream_t stream);

/* Round 6 -- forward convolution with extras, for the generator's full-resolution layers (models.py:15, 32-36, 66-72: enc1, ga1, dec4, dec5).
 *
 * (1) hi + lo PAIRS.  A tensor may travel as two 16-bit planes of one shape, value = hi + lo, hi = rn16(v), lo = rn16(v - hi): ~2 x the significant
 *     bits of the storage format at 2 x its bytes.  x1_lo / x2_lo: the lo planes of the sources; w_lo: the lo part of the packed weights
 *     (uegan_pack_weights_pair); y_lo: the lo plane of the result.  The products Whi xhi + Wlo xhi + Whi xlo are accumulated in fp32 (Wlo xlo, 2^-22
 *     relative, is dropped).  This is what puts the fp16-storage generator inside north_star's 1e-3 on the enhanced pixels;
 *     the backward pass reads the hi planes only.
 * (2) mul / mul_lo -> y_mul / y_mul_lo: the epilogue also forms act(...) * (mul + mul_lo) from the fp32 result (models.py:69 `y4.mul(x1)`); y still
 *     receives act(...) itself.
 * (3) res_x -> res_out (<= 4 output channels, 7x7 on 32 input channels: dec5.1): res_out[NCHW fp32] = clamp(act(...) + res_x, -1, 1) (models.py:70-72)
 *     from the fp32 result, y still receives act(...) (the backward's tanh').  Images b >= res_split read res_x2 / write res_out2 at b - res_split
 *     (one generator pass over two image sets).
 * (4) mean / rstd / stats_workspace: the moments of uegan_conv2d_fwd_stats.
 * *taken: 0 = no kernel of this library honours the request for this layer, NOTHING was launched (the caller runs the plain sequence or refuses);
 * bit 0 = launched; bit 1 = mean / rstd were produced.  16-bit storage only. */
typedef struct {
  const void* x1_lo;
  const void* x2_lo;
  const void* w_lo;
  void* y_lo;
  const void* mul;
  const void* mul_lo;
  void* y_mul;
  void* y_mul_lo;
  const float* res_x;
  const float* res_x2;
  float* res_out;
  float* res_out2;
  float* mean;
  float* rstd;
  void* stats_workspace;
  size_t stats_workspace_bytes;
  float eps;
  int32_t res_split;
  int32_t w_interleaved;    /* stride-2 forwards (G.enc2): w_ohwi is [Cout][packed_k(KH*KW*2*C1)] with, per tap, the hi part of the C1 weights and then their lo
                               part (uegan_pack_weights_pair with Cin_pad = 2 C1, dup_cin = 2); the source's channels are read twice: the weight PAIR on the
                               unchanged 64-/128-channel stride-2 kernel.  No other extra may be combined with it. */
  int32_t reserved;
} uegan_conv_ex;
size_t uegan_conv2d_fwd_ex_workspace_bytes(const uegan_conv_desc* d, const uegan_conv_ex* ex);      /* of the moments; 0 = none would be produced */
int uegan_conv2d_fwd_ex(const uegan_conv_desc* d, const uegan_conv_ex* ex, const void* x1, const void* x2, const void* w_ohwi, const float* bias,
                        const float* scale, void* y, int* taken, uegan_stream_t stream);
/* NCHW fp32 -> NHWC with the image as a hi + lo pair inside its own pixels: channels [0, C) = rn16(a x + b), [C, 2C) = what that rounding left,
 * the rest zero (2 C <= Cp; 16-bit storage).  Pairs with uegan_pack_weights_pair(dup_cin = 1). */
int uegan_nchw_to_nhwc_pair(int dtype, const float* x, void* y, int B, int C, int Cp, int H, int W, const float* a, const float* b,
                            uegan_stream_t stream);
/* uegan_instnorm_apply on a pair, result as a pair */
int uegan_instnorm_apply_pair(int dtype, const void* x, const void* x_lo, void* y, void* y_lo, const float* mean, const float* rstd, int B, int HW,
                              int C, uegan_stream_t stream);

/* the same plus y_pool = 2x2 max-pool of y (NHWC [B][Ho/2][Wo/2][Cout]; Ho, Wo even): VGG19's conv -> ReLU -> MaxPool2d(2) stages
 * (losses.py:74-104).  The pooled tensor is written by the convolution's epilogue where the kernel taking the layer can (the 64- and
 * 128-channel 3x3 layers: conv1_2, conv2_2), otherwise by uegan_maxpool2x2_fwd behind it -- the results are bit-identical. */
int uegan_conv2d_fwd_pool(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias,
                          const float* scale, void* y, void* y_pool, uegan_stream_t stream);
/* ... when only the first n_full images need y itself (no gradient flows through the others: the reference batch of the fidelity loss,
 * losses.py:29-30, or a no-grad pass): y[n_full:] is UNDEFINED afterwards -- a kernel with a pooling epilogue skips those stores, the
 * fallback writes them; y must still hold B images.  y_pool is complete either way. */
int uegan_conv2d_fwd_pool_part(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias,
                               const float* scale, void* y, void* y_pool, int n_full, uegan_stream_t stream);
/* ... and idx (bytes, [B][Ho/2][Wo/2][Cout]): the window position dy * 2 + dx of each maximum -- the first one in (row, column) order, as
 * ATen's max_pool2d_with_indices picks it -- for the first n_idx images (idx[n_idx:] undefined).  uegan_maxpool2x2_bwd_idx needs only
 * these and y_pool, so an image whose gradient is wanted does not need y either: n_full = 0, n_idx = B. */
int uegan_conv2d_fwd_pool_idx(const uegan_conv_desc* d, const void* x1, const void* x2, const void* w_ohwi, const float* bias,
                              const float* scale, void* y, void* y_pool, void* idx, int n_full, int n_idx, uegan_stream_t stream);
/* dx = scale * conv_transpose(dz, w) folded through the padding (adjoint of reflect / zero pad).
 * dz is the gradient w.r.t. the PRE-activation output. dx2 receives channels [C1, C1+C2) when C2 > 0. */
int uegan_conv2d_dgrad(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const float* scale, void* dx1,
                       void* dx2, uegan_stream_t stream);
/* Same result through an optional workspace: on small reflection-padded maps (where every tile touches the border) the
 * library computes d(pad(x)) image-free over the padded grid into `workspace` and folds the border back (the adjoint of
 * nn.ReflectionPad2d, models.py:80).  uegan_conv2d_dgrad_workspace_bytes() == 0 means the plain route is taken and
 * workspace may be NULL. */
size_t uegan_conv2d_dgrad_workspace_bytes(const uegan_conv_desc* d);
int uegan_conv2d_dgrad_ws(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const float* scale, void* dx1,
                          void* dx2, void* workspace, size_t workspace_bytes, uegan_stream_t stream);
/* The data gradient with respect to the PADDED input of a reflection-padded convolution (models.py:80-82, 160-162), for a caller whose next
 * kernel adds the mirror images itself (uegan_sn_act_bwd_p / uegan_act_bwd_p): workspace = [B][H + 2 pad][W + 2 pad][C1]
 * (uegan_conv2d_dgrad_padded_bytes), *pad_out = pad.  Taken only where one launch computes the whole padded grid: stride-2 layers with 64 / 128 k
 * input channels (all four parity classes on conv_flat_kernel; needs w_ihwo) and the discriminator's one-channel prediction heads
 * (head_dgrad_mfma_kernel; needs w_ohwi, the FORWARD pack, and scale == NULL).  Otherwise *pad_out = -1, nothing is launched and the caller uses
 * uegan_conv2d_dgrad_ws.  uegan_conv2d_dgrad_padded_bytes is 0 for a layer (dtype, implementation setting, tuning knobs) that would be declined. */
int uegan_conv2d_dgrad_padded(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const void* w_ohwi, const float* scale,
                              void* workspace, size_t workspace_bytes, int* pad_out, uegan_stream_t stream);
size_t uegan_conv2d_dgrad_padded_bytes(const uegan_conv_desc* d);

/* Deferred activation gradients (an exact restructuring: act' of LeakyReLU / ReLU / tanh is a function of the activated OUTPUT).
 * A conv whose input x is the output of an activated layer can fold that layer's act'(x) into its own data-gradient epilogue:
 *   dx = dgrad(dz) * act'(x_act),  x_act = this conv's input, in_act = the activation that produced it
 * and the producer skips its uegan_act_bwd pass (3 tensor passes) -- valid only when EVERY consumer of x applies the factor; the
 * max-pool and fidelity-loss gradients have the same `_act` form below.  C2 must be 0; workspace as for _dgrad_ws. */
int uegan_conv2d_dgrad_act(const uegan_conv_desc* d, const void* dz, const void* w_ihwo, const float* scale, void* dx1,
                           void* workspace, size_t workspace_bytes, int in_act, const void* x_act, uegan_stream_t stream);
size_t uegan_conv2d_wgrad_workspace_bytes(const uegan_conv_desc* d);
/* dw_oihw (fp32, OIHW) = scale * sum_pixels pad(x) (x) dz ; dbias (fp32[Cout], may be NULL) = sum_pixels dz.
 * Both are OVERWRITTEN. */
int uegan_conv2d_wgrad(const uegan_conv_desc* d, const void* x1, const void* x2, const void* dz, const float* scale,
                       float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes, uegan_stream_t stream);
/* The same with an accumulate bit mask: bit 0: dw_oihw += ..., bit 1: dbias += ... (beta = 1).  Lets the gradient of a layer that
 * is applied several times per step land directly in the flat gradient bucket the optimizer / RCCL all-reduce reads, with no
 * separate add. */
int uegan_conv2d_wgrad_acc(const uegan_conv_desc* d, const void* x1, const void* x2, const void* dz, const float* scale,
                           float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes, int accumulate,
                           uegan_stream_t stream);
/* dz = g * act'(a), a = saved activation OUTPUT (LeakyReLU/ReLU/tanh backward: models.py:252,35,178) */
int uegan_act_bwd(int dtype, int act, const void* g, const void* a, void* dz, int64_t n, uegan_stream_t stream);
/* dz = (g + g2) * act'(a): an activation with two consumers (g2 may be NULL) -- the sum is formed in registers */
int uegan_act_bwd2(int dtype, int act, const void* g, const void* g2, const void* a, void* dz, int64_t n, uegan_stream_t stream);
/* three consumers: dz = (g + g2 + g3) * act'(a)  (g2, g3 may be NULL; act may be UEGAN_ACT_NONE: a plain fused sum) */
int uegan_act_bwd3(int dtype, int act, const void* g, const void* g2, const void* g3, const void* a, void* dz, int64_t n,
                   uegan_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Layout / elementwise boundary ops
 * ------------------------------------------------------------------------------------------------- */
/* NCHW fp32 [B][C][H][W] -> NHWC dtype [B][H][W][Cp] (Cp >= C, channels C..Cp-1 written as zero padding) with
 * per-channel affine y = x*a[c] + b[c] (a,b HOST arrays of C <= 4 floats, NULL = identity).  Module boundary, and
 * trainer.py:108 `(x+1)/2` + losses.py:26-27 ImageNet normalisation folded in. */
int uegan_nchw_to_nhwc(int dtype, const float* x_nchw, void* y_nhwc, int B, int C, int Cp, int H, int W, const float* a,
                       const float* b, uegan_stream_t stream);
/* NHWC dtype [B][H][W][Cp] -> NCHW fp32 [B][C][H][W], y = x * a[c] (backward of the above; D prediction maps) */
int uegan_nhwc_to_nchw(int dtype, const void* x_nhwc, float* y_nchw, int B, int C, int Cp, int H, int W, const float* a,
                       uegan_stream_t stream);
/* out_nchw[B][C][H][W] = clamp(res_nhwc[B][H][W][Cp] + x_nchw, -1, 1)   (models.py:72) */
int uegan_residual_clamp_fwd(int dtype, const void* res_nhwc, const float* x_nchw, float* out_nchw, int B, int C, int Cp,
                             int H, int W, uegan_stream_t stream);
/* dres_nhwc = g * 1[-1 <= res+x <= 1] (padding channels zero); dx_nchw (may be NULL) likewise (torch.clamp backward) */
int uegan_residual_clamp_bwd(int dtype, const float* g_nchw, const void* res_nhwc, const float* x_nchw, void* dres_nhwc,
                             float* dx_nchw, int B, int C, int Cp, int H, int W, uegan_stream_t stream);
/* the same with the deferred activation gradient of res's producer folded in: dres additionally * act'(res) (models.py:35 nn.Tanh) */
int uegan_residual_clamp_bwd_act(int dtype, int act, const float* g_nchw, const void* res_nhwc, const float* x_nchw, void* dres_nhwc,
                                 float* dx_nchw, int B, int C, int Cp, int H, int W, uegan_stream_t stream);
/* y = a * b (models.py:70 `y4.mul(x1)`) and its backward da = g*b, db = g*a */
int uegan_mul_fwd(int dtype, const void* a, const void* b, void* y, int64_t n, uegan_stream_t stream);
int uegan_mul_bwd(int dtype, const void* g, const void* a, const void* b, void* da, void* db, int64_t n,
                  uegan_stream_t stream);
/* backward with the operands' producers' deferred activation gradients: da = g*b*act_a'(a), db = g*a*act_b'(b) */
int uegan_mul_bwd_act(int dtype, int act_a, int act_b, const void* g, const void* a, const void* b, void* da, void* db, int64_t n,
                      uegan_stream_t stream);
/* bilinear x2, align_corners=True (models.py:191-201) and its adjoint */
int uegan_upsample2x_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, uegan_stream_t stream);
int uegan_upsample2x_bwd(int dtype, const void* gy, void* gx, int B, int H, int W, int C, uegan_stream_t stream);
/* MaxPool2d(2,2) (torchvision VGG features idx 4,9,18,27) and backward (first max in scan order) */
int uegan_maxpool2x2_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, uegan_stream_t stream);
int uegan_maxpool2x2_bwd(int dtype, const void* x, const void* gy, void* gx, int B, int H, int W, int C,
                         uegan_stream_t stream);
/* gx additionally multiplied by act'(x): x is the output of an activated conv whose act_bwd is deferred to its consumers */
int uegan_maxpool2x2_bwd_act(int dtype, int act, const void* x, const void* gy, void* gx, int B, int H, int W, int C,
                             uegan_stream_t stream);
/* the same pair through window positions (one byte per pooled element, see uegan_conv2d_fwd_pool_idx): the forward also stores them, the
 * backward computes gx from y_pool (act'(maximum) = act'(y_pool)), idx and gy without reading x -- bit-identical to uegan_maxpool2x2_bwd_act */
int uegan_maxpool2x2_fwd_idx(int dtype, const void* x, void* y, void* idx, int B, int H, int W, int C, uegan_stream_t stream);
int uegan_maxpool2x2_bwd_idx(int dtype, int act, const void* y_pool, const void* idx, const void* gy, void* gx, int B, int H, int W, int C,
                             uegan_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Image stacks: history pool (utils.py:23-50) and the evaluation metrics of the inference configuration
 * ------------------------------------------------------------------------------------------------- */
/* dst[dst_idx[i]] = src_idx[i] >= 0 ? src_a[src_idx[i]] : src_b[~src_idx[i]]  for i < n_images (<= 64): whole fp32 images of
 * image_elems elements; the index tables are HOST arrays (passed to the kernel by value).  ImagePool.query's "return an old
 * image, keep the new one" (utils.py:41-46) is one gather (pool/batch -> output) + one scatter (batch -> pool). */
int uegan_copy_images(float* dst, const float* src_a, const float* src_b, const int32_t* dst_idx, const int32_t* src_idx,
                      int n_images, int64_t image_elems, uegan_stream_t stream);
/* [-1,1] NCHW fp32 -> uint8 NHWC exactly as tester.py:70-71 writes a PNG: denorm (utils.py:128-130: (x+1)/2 clamped to [0,1]),
 * then torchvision save_image's mul(255).add(0.5).clamp(0,255).to(uint8). */
int uegan_quantize_u8(const float* x_nchw, uint8_t* y_nhwc, int B, int C, int H, int W, uegan_stream_t stream);
/* Per image b < B of two uint8 NHWC stacks, after cropping crop_border pixels on every side (CalcPSNR.py:24,56 / CalcSSIM.py:24,56):
 *   sqdiff_sum[b] (may be NULL) = sum (a - b)^2               -> PSNR = 10 log10(255^2 / (sqdiff_sum / n)), CalcPSNR.py:85-92
 *   ssim_sum[b]   (may be NULL) = sum over channels and valid 7x7 windows of the SSIM index with skimage's defaults as called at
 *                 CalcSSIM.py:63 (uniform window, K1 .01, K2 .03, sample covariance, data_range 255); mean SSIM =
 *                 ssim_sum / (C * (H-2c-6) * (W-2c-6)).   Both outputs are DEVICE fp64 [B], overwritten. */
int uegan_image_metrics_u8(const uint8_t* a_nhwc, const uint8_t* b_nhwc, double* sqdiff_sum, double* ssim_sum, int B, int H, int W,
                           int C, int crop_border, uegan_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Input pipeline on the device (data_loader.py:74-82 train transform, :95-100 test transform, :126 H2D)
 * ------------------------------------------------------------------------------------------------- */
/* pixels: B (<= 64) crop windows of decoded 8-bit RGB images, DEVICE uint8 [B][in_h][in_w][3] (the RandomCrop of
 * data_loader.py:75 is the choice of the bytes that were copied).  Resize([out_h, out_w]) as Pillow's two-pass fixed-point
 * resampler (what torchvision's Resize runs on a PIL image, :76 / :97): htab / vtab are DEVICE int32 tables with one row of
 * (2 + hk) / (2 + vk) entries per output column / row: [first input index, tap count n <= k, k 22-bit fixed-point coefficients]
 * (built on the host as Pillow builds them: uegan_amd/data.py:resample_table).  Then the flips of :77-78 (flips: HOST int32 [B]
 * or NULL, bit 0 horizontal, bit 1 vertical), ToTensor's /255 and Normalize(0.5, 0.5) (:79-81) -> out_nchw fp32 [B][3][out_h][out_w],
 * the layout InputFetcher hands to the trainer (:124-127).  tmp: DEVICE uint8 [B][in_h][out_w][3] scratch.  Bit-exact. */
int uegan_input_transform(const uint8_t* pixels, int B, int in_h, int in_w, int out_h, int out_w, const int32_t* htab, int hk,
                          const int32_t* vtab, int vk, const int32_t* flips, uint8_t* tmp, float* out_nchw, uegan_stream_t stream);

/* Device-scalar plumbing of the step driver: zero a buffer (gradient buckets, loss accumulators); total[0] = sum_i weights[i] * terms[i][0]
 * accumulated left to right (trainer.py:104-115: g_loss = lambda_adv*adv + lambda_percep*percep + lambda_idt*idt), scaled[i] (may be NULL) =
 * weights[i] * terms[i][0] (the logged per-term values); its backward gout[i] = weights[i] * g[0].  terms: HOST table of device pointers. */
int uegan_fill_zero(void* p, size_t bytes, uegan_stream_t stream);
int uegan_scalar_wsum(int n, const float* const* terms, const float* weights, float* total, float* scaled, uegan_stream_t stream);
int uegan_scalar_wsum_bwd(int n, const float* weights, const float* g, float* gout, uegan_stream_t stream);
/* out[i] = src[i][0], i < n <= 8: the step's logged loss scalars (trainer.py:98-119 reads five of them with .item(): five host syncs)
 * collected into ONE device vector, so that the end-of-step readback is a single copy.  src: HOST table of device pointers. */
int uegan_gather_scalars(int n, const float* const* src, float* out, uegan_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * InstanceNorm2d (non-affine, eps 1e-5, biased variance): GAM (models.py:227,236), losses.py:18
 * ------------------------------------------------------------------------------------------------- */
/* scratch size (floats) ONE reduction pass needs for a [B][HW][C] tensor (split partials) */
size_t uegan_reduce_workspace_floats(int B, int HW, int C);
/* y = (x - mean) * rstd per (b,c); writes mean/rstd (fp32 [B*C]); tmp = fp32 [uegan_reduce_workspace_floats] */
int uegan_instnorm_fwd(int dtype, const void* x, void* y, float* mean, float* rstd, float* tmp, int B, int HW, int C, float eps,
                       uegan_stream_t stream);
/* dx = rstd * (dy - mean(dy) - y * mean(dy*y)); tmp = fp32 [uegan_reduce_workspace_floats] scratch */
int uegan_instnorm_bwd(int dtype, const void* dy, const void* y, const float* rstd, void* dx, float* tmp, int B, int HW,
                       int C, uegan_stream_t stream);

/* The norm_fun / act_fun variants of ConvBlock (models.py:88-101, 249-281): BatchNorm2d / InstanceNorm2d(affine=True,
 * track_running_stats=True) followed by LeakyReLU(0.2) | ReLU | Swish | SELU, as an affine map with explicit per-(b,c) coefficients
 * (fp32 [B*C]; which statistics feed them -- per sample, per batch, running -- is the caller's arithmetic on those arrays).
 *   uegan_moments             mean[b,c], var[b,c] (biased) of an NHWC tensor; tmp = fp32 [uegan_reduce_workspace_floats]
 *   uegan_affine_act_fwd      y = act(x * scale[b,c] + shift[b,c])                       (scale / shift NULL: 1 / 0)
 *   uegan_affine_act_bwd_sums g = gy * act'(x * scale + shift); sums[b,c] = {sum g, sum g * x}   (fp32 [B*C][2]; tmp as above)
 *   uegan_affine_act_bwd_apply gx = g * ca[b,c] + x * cb[b,c] + cc[b,c]                  (ca / cb / cc NULL: 1 / 0 / 0)
 * act: any UEGAN_ACT_* (evaluated on the pre-activation, which is recomputed from x and never stored). */
int uegan_moments(int dtype, const void* x, float* mean, float* var, float* tmp, int B, int HW, int C, uegan_stream_t stream);
int uegan_affine_act_fwd(int dtype, int act, const void* x, const float* scale, const float* shift, void* y, int B, int HW, int C,
                         uegan_stream_t stream);
int uegan_affine_act_bwd_sums(int dtype, int act, const void* gy, const void* x, const float* scale, const float* shift, float* sums,
                              float* tmp, int B, int HW, int C, uegan_stream_t stream);
int uegan_affine_act_bwd_apply(int dtype, int act, const void* gy, const void* x, const float* scale, const float* shift, const float* ca,
                               const float* cb, const float* cc, void* gx, int B, int HW, int C, uegan_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Losses
 * ------------------------------------------------------------------------------------------------- */
/* Relativistic average hinge over up to 8 scales (losses.py:348-362 per scale, summed 393-409).
 * real[i]/fake[i]: fp32 maps of n[i] elements (HOST pointer tables of device pointers).
 * fwd: loss[0] = sum_i 0.5*(mean relu(1 -/+ (r - mean f)) + mean relu(1 +/- (f - mean r)));
 *      tmp = fp32 [uegan_rahinge_workspace_floats(nscales)] keeps the per-scale sums for bwd (and the per-block partial sums they are
 *      folded from in a fixed order: every reduction of this library is bit-reproducible from run to run).
 * bwd: greal[i] / gfake[i] (entries or tables may be NULL) = gscale[0] * d loss / d real[i] | fake[i];
 *      gscale is a DEVICE scalar (autograd's grad_output), NULL = 1. */
size_t uegan_rahinge_workspace_floats(int nscales);      /* also uegan_rals_fwd's */
int uegan_rahinge_fwd(int nscales, const float* const* real, const float* const* fake, const int64_t* n,
                      int for_discriminator, float* loss, float* tmp, uegan_stream_t stream);
int uegan_rahinge_bwd(int nscales, const float* const* real, const float* const* fake, const int64_t* n,
                      int for_discriminator, const float* tmp, const float* gscale, float* const* greal,
                      float* const* gfake, uegan_stream_t stream);
/* The same loss on the raw prediction-head maps of a BATCHED discriminator pass: maps[s] = NHWC dtype [ngroups*nb][h_s][w_s][cp]
 * with channel 0 = the tanh prediction (models.py:170-182), pix_per_image[s] = h_s*w_s; image group g = images [g*nb, (g+1)*nb).
 * loss = sum over the npairs (real group, fake group) pairs (HOST int32 [2*npairs]) of the rahinge loss above -- trainer.py:92+95 is
 * {(exp, fake_store), (exp, raw)} on one pass, :104 is {(exp, fake)}.  tmp = fp32 [uegan_rahinge_heads_workspace_floats(nscales)].
 * bwd: gmaps[s] (same layout as maps[s]) = gscale[0] * d loss / d(pre-tanh head output) = d loss/dP * (1 - P^2) in channel 0, zeros in
 * the padding channels, written for the groups in group_mask only (bit g). */
size_t uegan_rahinge_heads_workspace_floats(int nscales);
int uegan_rahinge_heads_fwd(int dtype, int nscales, const void* const* maps, const int64_t* pix_per_image, int nb, int cp, int ngroups,
                            int npairs, const int32_t* pairs, int for_discriminator, float* loss, float* tmp, uegan_stream_t stream);
int uegan_rahinge_heads_bwd(int dtype, int nscales, const void* const* maps, const int64_t* pix_per_image, int nb, int cp, int ngroups,
                            int npairs, const int32_t* pairs, int for_discriminator, const float* tmp, const float* gscale,
                            void* const* gmaps, uint32_t group_mask, uegan_stream_t stream);
/* MultiscaleRecLoss(scale, rec_loss_type, multiscale) (losses.py:202-231) on NCHW fp32:
 * loss = sum_{i < nscales} 2^-i * criterion(avgpool^i(pred), avgpool^i(gt)), criterion (`kind`) 0 = L1Loss, 1 = SmoothL1Loss (beta 1),
 * 2 = MSELoss, all with mean reduction; nscales = min(scale, 3) (the reference's weight list has three entries), 1 for multiscale=False.
 * Any H x W: AvgPool2d(2, 2) floors, so a last odd row / column does not reach the next scale (a size that would pool to an empty map is an
 * error, as in torch).  gpred = gscale[0] * d loss / d pred.  scratch = fp32 [uegan_msrec_scratch_floats()]: the
 * per-block partial sums, added up in a fixed order (the loss is bit-reproducible). */
size_t uegan_msrec_scratch_floats(void);
int uegan_msrec_fwd(const float* pred, const float* gt, float* loss, float* scratch, int B, int C, int H, int W, int kind, int nscales,
                    uegan_stream_t stream);
int uegan_msrec_bwd(const float* pred, const float* gt, const float* gscale, float* gpred, int B, int C, int H, int W, int kind,
                    int nscales, uegan_stream_t stream);
/* Aliases of the default identity loss MultiscaleRecLoss(scale=3, 'l1', multiscale=True) (losses.py:219-231; trainer.py:43,113) =
 * uegan_msrec_*(kind 0, 3 scales): the entry points rounds 1-2 exported. */
size_t uegan_msl1_scratch_floats(void);
int uegan_msl1_fwd(const float* pred, const float* gt, float* loss, float* scratch, int B, int C, int H, int W, uegan_stream_t stream);
int uegan_msl1_bwd(const float* pred, const float* gt, const float* gscale, float* gpred, int B, int C, int H, int W, uegan_stream_t stream);
/* One VGG tap of PerceptualLoss (losses.py:30-34): fwd: loss += weight * MSE(IN(x), IN(y)) (ACCUMULATED with
 * atomicAdd: zero loss before the first tap); tmp = fp32 [3 * uegan_reduce_workspace_floats(B,HW,C)] keeps the
 * statistics for bwd.  bwd: gx = gscale[0] * d(weight*MSE)/dx. */
int uegan_percep_tap_fwd(int dtype, const void* x, const void* y, float weight, float* loss, float* tmp, int B, int HW,
                         int C, float eps, uegan_stream_t stream);
/* uegan_percep_tap_fwd with the InstanceNorm moments of both taps given ([B][C] each, rstd = 1 / sqrt(var + eps): what uegan_conv2d_fwd_stats
 * returns for the conv that produced the tap, losses.py:30-34): the moment passes over x and y are skipped; tmp as for uegan_percep_tap_fwd
 * (uegan_percep_tap_bwd* read the moments back from it). */
int uegan_percep_tap_fwd_given(int dtype, const void* x, const void* y, float weight, float* loss, float* tmp, int B, int HW, int C,
                               const float* mean_x, const float* rstd_x, const float* mean_y, const float* rstd_y, uegan_stream_t stream);
int uegan_percep_tap_bwd(int dtype, const void* x, const void* y, float weight, const float* gscale, void* gx,
                         const float* tmp, int B, int HW, int C, float eps, uegan_stream_t stream);
/* gx additionally multiplied by act'(x) (deferred activation gradient of the VGG conv that produced the tap) */
int uegan_percep_tap_bwd_act(int dtype, int act, const void* x, const void* y, float weight, const float* gscale, void* gx,
                             const float* tmp, int B, int HW, int C, float eps, uegan_stream_t stream);

/* the same with accumulate != 0: gx += ... -- a tap has two consumers (the next VGG layer and the loss); the loss gradient is
 * added into the buffer the next layer's data gradient was written to, instead of a separate elementwise add */
int uegan_percep_tap_bwd_acc(int dtype, int act, const void* x, const void* y, float weight, const float* gscale, void* gx,
                             const float* tmp, int B, int HW, int C, float eps, int accumulate, uegan_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Spectral norm (models.py:185-188 -> torch.nn.utils.spectral_norm, 1 power iteration, eps 1e-12)
 * ------------------------------------------------------------------------------------------------- */
/* w: fp32 [rows][cols] (= weight_orig.view(Cout,-1)); u[rows], v[cols] updated IN PLACE when do_iter != 0;
 * sigma_out[0] = sigma = u^T W v, sigma_out[1] = 1/sigma.  tmp = fp32 [uegan_specnorm_multi_workspace_floats(rows, cols)] scratch.
 * (One layer, one round of uegan_specnorm_multi below: fixed summation order.) */
int uegan_specnorm_sigma(const float* w, float* u, float* v, int rows, int cols, int do_iter, float eps,
                         float* sigma_out, float* tmp, uegan_stream_t stream);
/* gradient through W/sigma with u,v constant: dw = g - (<g,w> * inv_sigma) * u v^T, where g = dL/d(W/sigma) * inv_sigma
 * (already scaled).  In place on g allowed (dw == g). tmp = fp32 [uegan_specnorm_grad_workspace_floats()] (block partials of <g,w>,
 * folded in a fixed order). */
size_t uegan_specnorm_grad_workspace_floats(void);
int uegan_specnorm_grad(const float* g, const float* w, const float* u, const float* v, const float* sigma, float* dw,
                        int rows, int cols, float* tmp, uegan_stream_t stream);

/* All spectral-normalised layers of a network (<= 8) in one call, `n_rounds` consecutive power-iteration rounds (round r = the
 * r-th application of the layers within an optimizer step: a batched discriminator pass over several image groups uses the sigma of
 * round g for group g, uegan_conv_desc.scale_group).  Fixed summation order, no atomics: data-parallel replicas advance bit-identical
 * u / v.  Per layer: u, v updated IN PLACE when do_iter (eval mode: 0, sigma only); sigma[r], inv_sigma[r] written per round; u_hist
 * [n_rounds][rows] / v_hist [n_rounds][cols] (may be NULL) receive the u, v of each round (the constants of that round's gradient).
 * tmp = fp32 [uegan_specnorm_multi_workspace_floats(rows, cols)].  `layers` is a HOST array (passed to the kernels by value). */
typedef struct {
  const float* w;
  float* u;
  float* v;
  float* sigma;
  float* inv_sigma;
  float* u_hist;
  float* v_hist;
  float* tmp;
  int32_t rows, cols;
} uegan_sn_layer;
size_t uegan_specnorm_multi_workspace_floats(int rows, int cols);
int uegan_specnorm_multi(const uegan_sn_layer* layers, int n_layers, int n_rounds, int do_iter, float eps, uegan_stream_t stream);
/* Activation backward of a spectral-normalised conv whose batch holds `ngroups` image groups of pix_per_group pixels, group r convolved with
 * W / sigma_r (a batched discriminator pass, models.py:139-155 applied to several image sets): dz = (g + g2) * act'(y) * inv_sigma[r]
 * (g2 may be NULL; act none / LeakyReLU / ReLU).  With dz pre-scaled, ONE uegan_conv2d_wgrad over all groups (scale NULL) gives
 * sum_r G_r of torch's spectral-norm gradient, the data gradient needs no scale either, and the projection coefficients
 * c_r = <G_r, W> / sigma_r = sum_{group r} dz * (z - bias) (z = act^-1(y); W (*) x = sigma_r (z - bias)) are reduced here from the
 * activation instead of by a dot product over the weights.  Returns (> 0) the number of partial blocks per group; negative = error.
 * workspace = fp32 [uegan_sn_act_bwd_workspace_floats(ngroups, C)]: per-block partials of c_r and of the bias gradient sum dz_raw.
 * uegan_sn_grad_finish: dw -= sum_r c_r u_r v_r^T (u_hist [ngroups][rows], v_hist [ngroups][cols]), db (+)= bias gradient (db may be NULL);
 * every sum in a fixed order. */
size_t uegan_sn_act_bwd_workspace_floats(int ngroups, int C);
int uegan_sn_act_bwd(int dtype, int act, const void* g, const void* g2, const void* y, const float* bias, int nbias, const float* inv_sigma,
                     void* dz, float* workspace, int64_t pix_per_group, int C, int ngroups, uegan_stream_t stream);
int uegan_sn_grad_finish(float* dw, float* db, const float* workspace, int nbx, int ngroups, const float* u_hist, const float* v_hist, int rows,
                         int cols, int C, int acc_bias, uegan_stream_t stream);
/* Round 5: the same two activation backwards with g / g2 optionally on the PADDED grid of their reflection-padded consumer
 * ([images][H + 2 pad][W + 2 pad][C], pad_g / pad_g2 > 0; 0 = a plain [images][H][W][C] tensor): the adjoint of nn.ReflectionPad2d
 * (models.py:80, 160: up to 2 x 2 mirror sources on the border ring) is applied while the gradient is read -- the gradients of the trunk
 * activation coming back from the next stride-2 trunk conv and from the prediction head (models.py:139-155) are produced on the padded grid
 * by uegan_conv2d_dgrad_padded and are neither folded by a pass of their own nor stored a second time.  pix_per_group = images per group x H x W. */
int uegan_sn_act_bwd_p(int dtype, int act, const void* g, int pad_g, const void* g2, int pad_g2, const void* y, const float* bias, int nbias,
                       const float* inv_sigma, void* dz, float* workspace, int64_t pix_per_group, int H, int W, int C, int ngroups,
                       uegan_stream_t stream);
int uegan_act_bwd_p(int dtype, int act, const void* g, int pad_g, const void* g2, int pad_g2, const void* a, void* dz, int B, int H, int W, int C,
                    uegan_stream_t stream);
/* uegan_specnorm_grad with 1/sigma given directly and an accumulate mode: dw (+)= g - (<g,w> * inv_sigma[0]) * u v^T.
 * accumulate != 0 needs g != dw (several applications of one layer add their gradients into one bucket). */
int uegan_specnorm_grad_acc(const float* g, const float* w, const float* u, const float* v, const float* inv_sigma, float* dw, int rows,
                            int cols, float* tmp, int accumulate, uegan_stream_t stream);

/* The other adversarial losses of GANLoss.loss (losses.py:312-392); argument conventions of uegan_rahinge_fwd / _bwd.
 * 'rals' (losses.py:363-376): loss = sum_scales (mean (r - mean f -+ 1)^2 + mean (f - mean r +- 1)^2) / 2. */
int uegan_rals_fwd(int nscales, const float* const* real, const float* const* fake, const int64_t* n, int for_discriminator, float* loss,
                   float* tmp, uegan_stream_t stream);
int uegan_rals_bwd(int nscales, const float* const* real, const float* const* fake, const int64_t* n, int for_discriminator,
                   const float* tmp, const float* gscale, float* const* greal, float* const* gfake, uegan_stream_t stream);
/* Non-relativistic modes: loss = sum_scales mean term(pred), one prediction list (for_real / for_fake selects it, losses.py:313-347,
 * 378-392).  term: BCE = binary_cross_entropy_with_logits against the constant `target` ('original'), LS = (p - target)^2 ('ls'),
 * HINGE_REAL = -min(p - 1, 0), HINGE_FAKE = -min(-p - 1, 0) (discriminator 'hinge'), NEG_MEAN = -p (generator 'hinge', wgan real),
 * POS_MEAN = p (wgan fake).  tmp: fp32 [uegan_pred_loss_workspace_floats(nscales)].  bwd: gpreds[k][i] = gscale[0] * term'(p) / n[k]. */
enum { UEGAN_PRED_BCE = 0, UEGAN_PRED_LS = 1, UEGAN_PRED_HINGE_REAL = 2, UEGAN_PRED_HINGE_FAKE = 3, UEGAN_PRED_NEG_MEAN = 4,
       UEGAN_PRED_POS_MEAN = 5 };
size_t uegan_pred_loss_workspace_floats(int nscales);
int uegan_pred_loss_fwd(int term, float target, int nscales, const float* const* preds, const int64_t* n, float* loss, float* tmp,
                        uegan_stream_t stream);
int uegan_pred_loss_bwd(int term, float target, int nscales, const float* const* preds, const int64_t* n, const float* gscale,
                        float* const* gpreds, uegan_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Adam with L2-in-gradient weight decay (torch.optim.Adam; trainer.py:337-338), one launch over many tensors.
 * desc: DEVICE array of n_tensors uegan_adam_tensor; step is 1-based. g is scaled by grad_scale first
 * (1/world_size after the RCCL sum).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} uegan_adam_tensor;
int uegan_adam_l2_step(const uegan_adam_tensor* desc_dev, int n_tensors, int64_t max_n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, float grad_scale, int step, uegan_stream_t stream);

/* torch.optim.RMSprop(lr, alpha, eps = 1e-8) with its defaults weight_decay 0, momentum 0, centered False (trainer.py:339-342):
 * square_avg (desc.v) = alpha * square_avg + (1 - alpha) g^2; p -= lr * g / (sqrt(square_avg) + eps).  desc.m is not touched. */
int uegan_rmsprop_step(const uegan_adam_tensor* desc_dev, int n_tensors, int64_t max_n, float lr, float alpha, float eps,
                       float grad_scale, uegan_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UEGAN_HIP_H_ */
