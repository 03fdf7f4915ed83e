// Shared device helpers for the UEGAN gfx950 kernels.
// Storage type T is either float or bf16_t (raw 16-bit pattern); all arithmetic is fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/uegan_hip.h"

// Timing ablations (kernels that skip their loads / multiplies / stores to show where a layer's time goes; results are garbage) exist
// only in the tools build (tools/build_tools.sh: -DUEGAN_TOOLS_BUILD, a separate .so that uegan_tools_set_ablation() switches).
// In libuegan_hip.so UEGAN_ABL_BITS() is the constant 0: the branches fold away and nothing at run time can reach them.
#ifdef UEGAN_TOOLS_BUILD
#define UEGAN_ABL_BITS(x) (x)
#else
#define UEGAN_ABL_BITS(x) 0
#endif

namespace uegan {

extern int g_tuning[UEGAN_TUNE_COUNT];      // uegan_set_tuning (conv.hip): launch-variant thresholds; the library never reads the environment
extern int g_abl_stream, g_abl_wide;        // tools build only (always 0 in the product)


typedef uint16_t bf16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ float bits_to_f32(uint32_t u) {
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}
__host__ __device__ __forceinline__ uint32_t f32_to_bits(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  return c.u;
}
// THE 16-BIT STORAGE FORMAT.  `bf16_t` is the library's 16-bit container.  In libuegan_hip.so it holds bfloat16; the SAME sources built
// with -DUEGAN_HALF_FP16 (libuegan_hip_f16.so, uegan_amd.set_compute_dtype(torch.float16)) hold IEEE fp16 in it: same bytes, same
// MFMA rate (v_mfma_*_f16), 11 instead of 8 significant bits -- the format this generator's O(1) activations want (DESIGN.md section 4:
// 66 dB instead of 57 dB on the enhanced pixels); gradients need a loss scale in that mode (trainer.Trainer(loss_scale=...)).
// Every conversion goes through the helpers below; nothing else in the library knows the format.
#ifdef UEGAN_HALF_FP16
typedef _Float16 h16x2_hw __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }      // round-to-nearest-even
// the two halves of a packed pair
__host__ __device__ __forceinline__ float half_lo_to_f32(uint32_t u) { return (float)__builtin_bit_cast(h16x2_hw, u).x; }
__host__ __device__ __forceinline__ float half_hi_to_f32(uint32_t u) { return (float)__builtin_bit_cast(h16x2_hw, u).y; }
#else
__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t v) { return bits_to_f32(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN kept quiet
__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = f32_to_bits(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__host__ __device__ __forceinline__ float half_lo_to_f32(uint32_t u) { return bits_to_f32(u << 16); }
__host__ __device__ __forceinline__ float half_hi_to_f32(uint32_t u) { return bits_to_f32(u & 0xffff0000u); }
#endif

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int kDtype = UEGAN_F32;
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  static __host__ __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __host__ __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
  static constexpr int kDtype = UEGAN_BF16;
  static constexpr int EPC = 8;
  static __host__ __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __host__ __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }      // (scalar tails only)
};

// two fp32 -> packed bf16 pair (lo in bits 15:0), round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950 (the software
// rounding above costs ~8 VALU per value, which made the epilogues of the memory-bound kernels VALU-bound)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#if defined(UEGAN_EMU)
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
#elif defined(UEGAN_HALF_FP16)
  typedef float f32x2_hw __attribute__((ext_vector_type(2)));
  const f32x2_hw v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h16x2_hw));      // v_cvt_pk_f16_f32 on gfx950 (round-to-nearest-even)
#else
  typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
  typedef float f32x2_hw __attribute__((ext_vector_type(2)));
  const f32x2_hw v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
#endif
}

// pack 4 consecutive fp32 results into T and store (p must be 4-element aligned)
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  f32x4 v = {a, b, c, d};
  *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
  u32x2 v;
  v.x = pack_bf16x2(a, b);
  v.y = pack_bf16x2(c, d);
  *reinterpret_cast<u32x2*>(p) = v;
}

// 4 consecutive elements -> fp32 (p must be 4-element aligned)
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
  const u32x2 t = *reinterpret_cast<const u32x2*>(p);
  v[0] = half_lo_to_f32(t.x); v[1] = half_hi_to_f32(t.x);
  v[2] = half_lo_to_f32(t.y); v[3] = half_hi_to_f32(t.y);
}

// V consecutive elements (V = 1, or one 16-byte chunk = DT<T>::EPC) <-> fp32 registers
template <typename T, int V> struct Vec;
template <typename T> struct Vec<T, 1> {
  static __device__ __forceinline__ void ld(const T* p, float (&v)[1]) { v[0] = DT<T>::ld(p); }
  static __device__ __forceinline__ void st(T* p, const float (&v)[1]) { DT<T>::st(p, v[0]); }
};
template <> struct Vec<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  }
};
template <> struct Vec<bf16_t, 8> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      v[2 * d] = half_lo_to_f32(t[d]);
      v[2 * d + 1] = half_hi_to_f32(t[d]);
    }
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
    u32x4 t;
#pragma unroll
    for (int d = 0; d < 4; ++d) t[d] = pack_bf16x2(v[2 * d], v[2 * d + 1]);
    *reinterpret_cast<u32x4*>(p) = t;
  }
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case UEGAN_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case UEGAN_ACT_RELU: return v > 0.f ? v : 0.f;
    case UEGAN_ACT_TANH: return tanhf(v);
    default: return v;
  }
}
template <int ACT>
__device__ __forceinline__ float apply_act_c(float v) {      // activation known at compile time (specialised epilogues)
  if (ACT == UEGAN_ACT_LRELU) return fmaxf(v, 0.2f * v);
  if (ACT == UEGAN_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == UEGAN_ACT_TANH) return tanhf(v);
  return v;
}
// derivative of the activation expressed through its OUTPUT a = act(z)
__device__ __forceinline__ float act_grad_from_out(float a, int act) {
  switch (act) {
    case UEGAN_ACT_LRELU: return a > 0.f ? 1.f : 0.2f;
    case UEGAN_ACT_RELU: return a > 0.f ? 1.f : 0.f;
    case UEGAN_ACT_TANH: return 1.f - a * a;
    default: return 1.f;
  }
}
// ... plus the sigmoid of the 'ls' / 'rals' prediction heads (models.py:175-176).  Separate functions, used by the head kernels and
// the activation-backward kernel only: one more case in the MFMA kernels' epilogues (above) pushed the masked 256-channel patch
// kernel over its VGPR budget (95-140 spilled registers, 359 -> 729 us per launch).
__device__ __forceinline__ float apply_act_ext(float v, int act) { return act == UEGAN_ACT_SIGMOID ? 1.f / (1.f + expf(-v)) : apply_act(v, act); }
__device__ __forceinline__ float act_grad_from_out_ext(float a, int act) { return act == UEGAN_ACT_SIGMOID ? a * (1.f - a) : act_grad_from_out(a, act); }
// the full activation set of get_act_fun (models.py:249-263) evaluated on the PRE-activation (standalone norm/activation kernels
// only: Swish is not monotonic, so its derivative cannot be read off its output like the epilogue activations' above)
#define UEGAN_SELU_ALPHA 1.6732632423543772848170429916717f
#define UEGAN_SELU_SCALE 1.0507009873554804934193349852946f
__device__ __forceinline__ float act_of_pre(float v, int act) {
  switch (act) {
    case UEGAN_ACT_SWISH: return v / (1.f + expf(-v));
    case UEGAN_ACT_SELU: return v > 0.f ? UEGAN_SELU_SCALE * v : (UEGAN_SELU_SCALE * UEGAN_SELU_ALPHA) * expm1f(v);
    default: return apply_act_ext(v, act);
  }
}
__device__ __forceinline__ float act_grad_of_pre(float v, int act) {
  switch (act) {
    case UEGAN_ACT_LRELU: return v > 0.f ? 1.f : 0.2f;
    case UEGAN_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case UEGAN_ACT_TANH: { const float t = tanhf(v); return 1.f - t * t; }
    case UEGAN_ACT_SIGMOID: { const float sg = 1.f / (1.f + expf(-v)); return sg * (1.f - sg); }
    case UEGAN_ACT_SWISH: { const float sg = 1.f / (1.f + expf(-v)); return sg * (1.f + v * (1.f - sg)); }
    case UEGAN_ACT_SELU: return v > 0.f ? UEGAN_SELU_SCALE : (UEGAN_SELU_SCALE * UEGAN_SELU_ALPHA) * expf(v);
    default: return 1.f;
  }
}

// reflection index: valid for -n < i < 2n-1 (single reflection, pad < n as nn.ReflectionPad2d requires)
__host__ __device__ __forceinline__ int reflect_idx(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// wave-level sum over 64 lanes (result valid in every lane)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-level sum for blocks of up to 1024 threads; red must hold 16 floats; result in all threads
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// wait until at most N of this wave's vector-memory operations (incl. direct-to-LDS loads) are outstanding; LDS and
// scalar counters untouched (gfx9+ s_waitcnt immediate: vmcnt[3:0]=bits 3:0, expcnt=6:4, lgkmcnt=11:8, vmcnt[5:4]=15:14)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
// workgroup barrier WITHOUT the memory fence of __syncthreads(): the compiler does not drain vmcnt for it, so
// direct-to-LDS loads can stay in flight across it (the data they carry is ordered by wait_vmcnt + this barrier)
__device__ __forceinline__ void raw_barrier() { __builtin_amdgcn_s_barrier(); }

void set_error(const char* fmt, ...);

}  // namespace uegan

#define UEGAN_CHECK_ARG(cond, ...)                \
  do {                                            \
    if (!(cond)) {                                \
      uegan::set_error(__VA_ARGS__);              \
      return UEGAN_E_INVALID;                     \
    }                                             \
  } while (0)

#define UEGAN_CHECK_LAUNCH()                                             \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      uegan::set_error("HIP launch failed: %s", hipGetErrorString(e__)); \
      return UEGAN_E_HIP;                                                \
    }                                                                    \
  } while (0)
