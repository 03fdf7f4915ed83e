// TEST INFRASTRUCTURE: a tiny CPU emulation of the slice of HIP that uegan_amd/csrc uses.
//
// The build container has no GPU and GPU time is rationed, so the kernel sources are ALSO compiled, unmodified,
// for the host against this header (tests/emu/build_emu.sh puts tests/emu first on the include path, so
// `#include <hip/hip_runtime.h>` resolves here).  Every GPU thread is a ucontext fiber; __syncthreads(), wave
// shuffles and MFMA are rendezvous points handled by a per-block scheduler; blocks are spread over OS threads.
// The MFMA emulation encodes the gfx950 fragment layouts documented in /opt/skills/guides
// (16x16x32 bf16 and 16x16x4 f32) -- the real hardware check of those layouts is uegan_selftest_mfma().
//
// This is NOT a product backend: uegan_amd never loads the emulated library by itself; only tests inject it.
#pragma once
#include <ucontext.h>
#include <sys/mman.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include <time.h>

// On the host, clang's __bf16 is a conversion-happy arithmetic type: bit_casts through __bf16 vectors are not
// bit-preserving at -O2.  The kernels only use it as a 16-bit storage lane for the MFMA / dot2 builtins, so make it one.
#define UEGAN_EMU 1            /* the two inline-asm statements of the kernels have a builtin twin behind this */
#define __bf16 unsigned short
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return hipSuccess; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)malloc(n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum hipMemcpyKind { hipMemcpyDeviceToHost = 2 };
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return hipSuccess;
}

// events: wall-clock stubs (launches are synchronous in the emulator)
struct emuEvent { double t; };
typedef emuEvent* hipEvent_t;
inline double emu_now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent{0}; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu_now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

namespace emu {

enum State { READY = 0, AT_BLOCK = 1, AT_WAVE = 2, DONE = 3 };

struct Fiber {
  ucontext_t ctx;
  void* stack = nullptr;
  State state = DONE;
  dim3 tid;
  unsigned linear = 0;
  unsigned wave_seq = 0;
};

struct WaveBuf {
  // double-buffered exchange area: [parity][lane][16 words]
  uint32_t w[2][64][16];
};

struct Worker {
  ucontext_t main_ctx;
  std::vector<Fiber> fibers;
  std::vector<WaveBuf> waves;
  Fiber* cur = nullptr;
  dim3 block_idx, block_dim, grid_dim;
  int or_accum = 0, or_result = 0;
  const std::function<void()>* body = nullptr;
};

inline Worker*& tl_worker() {
  static thread_local Worker* w = nullptr;
  return w;
}

static const size_t kStackBytes = 256 * 1024;

inline void fiber_entry() {
  Worker* w = tl_worker();
  Fiber* f = w->cur;
  (*w->body)();
  f->state = DONE;
  swapcontext(&f->ctx, &w->main_ctx);
}

inline void yield_to_scheduler(State st) {
  Worker* w = tl_worker();
  Fiber* f = w->cur;
  f->state = st;
  swapcontext(&f->ctx, &w->main_ctx);
}

inline void run_block(Worker* w, dim3 bidx, dim3 bdim, dim3 gdim) {
  const unsigned nthreads = bdim.x * bdim.y * bdim.z;
  if (w->fibers.size() < nthreads) {
    size_t old = w->fibers.size();
    w->fibers.resize(nthreads);
    // contexts hold self-pointers: (re)create all stacks lazily below
    for (size_t i = 0; i < old; ++i) { /* stacks stay valid, ctx re-made per block */ }
  }
  const unsigned nwaves = (nthreads + 63) / 64;
  if (w->waves.size() < nwaves) w->waves.resize(nwaves);
  w->block_idx = bidx;
  w->block_dim = bdim;
  w->grid_dim = gdim;
  w->or_accum = 0;
  for (unsigned t = 0; t < nthreads; ++t) {
    Fiber& f = w->fibers[t];
    if (!f.stack) {
      f.stack = mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (f.stack == MAP_FAILED) { perror("mmap fiber stack"); abort(); }
    }
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStackBytes;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    f.state = READY;
    f.linear = t;
    f.tid = dim3(t % bdim.x, (t / bdim.x) % bdim.y, t / (bdim.x * bdim.y));
    f.wave_seq = 0;
  }
  unsigned done = 0;
  while (done < nthreads) {
    bool progressed = false;
    for (unsigned t = 0; t < nthreads; ++t) {
      Fiber& f = w->fibers[t];
      if (f.state != READY) continue;
      w->cur = &f;
      swapcontext(&w->main_ctx, &f.ctx);
      progressed = true;
      if (f.state == DONE) ++done;
    }
    // block barrier: every live fiber waits at it
    unsigned at_block = 0, live = 0;
    for (unsigned t = 0; t < nthreads; ++t) {
      const State s = w->fibers[t].state;
      if (s != DONE) ++live;
      if (s == AT_BLOCK) ++at_block;
    }
    if (live && at_block == live) {
      w->or_result = w->or_accum;
      w->or_accum = 0;
      for (unsigned t = 0; t < nthreads; ++t)
        if (w->fibers[t].state == AT_BLOCK) w->fibers[t].state = READY;
      progressed = true;
    }
    // wave rendezvous
    for (unsigned wv = 0; wv < nwaves; ++wv) {
      unsigned lo = wv * 64, hi = lo + 64 < nthreads ? lo + 64 : nthreads, at = 0, lv = 0;
      for (unsigned t = lo; t < hi; ++t) {
        const State s = w->fibers[t].state;
        if (s != DONE) ++lv;
        if (s == AT_WAVE) ++at;
      }
      if (lv && at == lv) {
        for (unsigned t = lo; t < hi; ++t)
          if (w->fibers[t].state == AT_WAVE) w->fibers[t].state = READY;
        progressed = true;
      }
    }
    if (!progressed && done < nthreads) {
      fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier\n", bidx.x, bidx.y, bidx.z);
      abort();
    }
  }
}

inline int num_workers() {
  const char* e = getenv("UEGAN_EMU_THREADS");
  int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  std::atomic<size_t> next(0);
  static std::vector<Worker*> pool;   // launches are issued from one host thread at a time (tests)
  int nw = num_workers();
  if ((size_t)nw > nblocks) nw = (int)nblocks;
  if (nw < 1) nw = 1;
  while ((int)pool.size() < nw) pool.push_back(new Worker());
  auto work = [&](int wi) {
    Worker* my = pool[wi];
    tl_worker() = my;
    my->body = &body;
    for (;;) {
      const size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)));
      run_block(my, bidx, block, grid);
    }
  };
  if (nw <= 1) {
    work(0);
    return;
  }
  std::vector<std::thread> th;
  for (int i = 0; i < nw; ++i) th.emplace_back(work, i);
  for (auto& t : th) t.join();
}

// ---- wave exchange -------------------------------------------------------------------------------------
inline uint32_t* wave_slot(unsigned lane_linear, unsigned parity) {
  Worker* w = tl_worker();
  return w->waves[lane_linear / 64].w[parity][lane_linear % 64];
}

template <typename V>
inline void wave_publish_and_sync(const V* vals, int nvals) {
  Worker* w = tl_worker();
  Fiber* f = w->cur;
  uint32_t* slot = wave_slot(f->linear, f->wave_seq & 1);
  memcpy(slot, vals, sizeof(V) * nvals);
  yield_to_scheduler(AT_WAVE);
}
inline const uint32_t* wave_peer(unsigned lane_in_wave) {
  Worker* w = tl_worker();
  Fiber* f = w->cur;
  return w->waves[f->linear / 64].w[f->wave_seq & 1][lane_in_wave];
}
inline void wave_op_done() { ++tl_worker()->cur->wave_seq; }
inline unsigned my_lane() { return tl_worker()->cur->linear % 64; }

typedef float f32x4_e __attribute__((ext_vector_type(4)));

// the 16-bit operand format of the MFMA / dot emulation follows the build: bfloat16, or IEEE fp16 under -DUEGAN_HALF_FP16 (csrc/common.h)
inline float bf16_bits_to_f32(uint16_t v) {
#ifdef UEGAN_HALF_FP16
  _Float16 h;
  memcpy(&h, &v, 2);
  return (float)h;
#else
  uint32_t u = ((uint32_t)v) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

// D = A*B + C, 16x16x32 bf16: A lane l = A[i=l&15][k=8*(l>>4)+e], B lane l = B[k=8*(l>>4)+e][j=l&15]
template <typename AV>
inline f32x4_e mfma_16x16x32_bf16(AV a, AV b, f32x4_e c) {
  uint32_t pub[8];
  memcpy(pub, &a, 16);
  memcpy(pub + 4, &b, 16);
  wave_publish_and_sync(pub, 8);
  const unsigned l = my_lane();
  const unsigned j = l & 15;
  f32x4_e d = c;
  for (int r = 0; r < 4; ++r) {
    const unsigned i = 4 * (l >> 4) + r;
    float acc = d[r];
    for (int k = 0; k < 32; ++k) {
      const uint16_t* pa = reinterpret_cast<const uint16_t*>(wave_peer(i + 16 * (k / 8)));
      const uint16_t* pb = reinterpret_cast<const uint16_t*>(wave_peer(j + 16 * (k / 8))) + 8;
      acc += bf16_bits_to_f32(pa[k % 8]) * bf16_bits_to_f32(pb[k % 8]);
    }
    d[r] = acc;
  }
  wave_op_done();
  return d;
}

// 32x32x16 bf16: A lane l = A[i=l&31][k=8*(l>>5)+e], B lane l = B[k=8*(l>>5)+e][j=l&31]; D lane l reg r = D[(r&3)+8*(r>>2)+4*(l>>5)][l&31]
typedef float f32x16_e __attribute__((ext_vector_type(16)));
template <typename AV>
inline f32x16_e mfma_32x32x16_bf16(AV a, AV b, f32x16_e c) {
  uint32_t pub[8];
  memcpy(pub, &a, 16);
  memcpy(pub + 4, &b, 16);
  wave_publish_and_sync(pub, 8);
  const unsigned l = my_lane();
  const unsigned j = l & 31;
  f32x16_e d = c;
  for (int r = 0; r < 16; ++r) {
    const unsigned i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = d[r];
    for (int k = 0; k < 16; ++k) {
      const uint16_t* pa = reinterpret_cast<const uint16_t*>(wave_peer(i + 32 * (k / 8)));
      const uint16_t* pb = reinterpret_cast<const uint16_t*>(wave_peer(j + 32 * (k / 8))) + 8;
      acc += bf16_bits_to_f32(pa[k % 8]) * bf16_bits_to_f32(pb[k % 8]);
    }
    d[r] = acc;
  }
  wave_op_done();
  return d;
}

// 16x16x4 f32: A lane l = A[i=l&15][k=l>>4], B lane l = B[k=l>>4][j=l&15]; exact fmaf chain in k order
inline f32x4_e mfma_16x16x4_f32(float a, float b, f32x4_e c) {
  float pub[2] = {a, b};
  wave_publish_and_sync(pub, 2);
  const unsigned l = my_lane();
  const unsigned j = l & 15;
  f32x4_e d = c;
  for (int r = 0; r < 4; ++r) {
    const unsigned i = 4 * (l >> 4) + r;
    float acc = d[r];
    for (int k = 0; k < 4; ++k) {
      const float* pa = reinterpret_cast<const float*>(wave_peer(i + 16 * k));
      const float* pb = reinterpret_cast<const float*>(wave_peer(j + 16 * k));
      acc = fmaf(pa[0], pb[1], acc);
    }
    d[r] = acc;
  }
  wave_op_done();
  return d;
}

template <typename V>
inline V shfl_xor(V v, int mask) {
  static_assert(sizeof(V) == 4 || sizeof(V) == 8, "32- and 64-bit shuffles only");
  wave_publish_and_sync(&v, 1);
  V out;
  memcpy(&out, wave_peer((my_lane() ^ (unsigned)mask) & 63), sizeof(V));
  wave_op_done();
  return out;
}

}  // namespace emu

#define threadIdx (::emu::tl_worker()->cur->tid)
#define blockIdx (::emu::tl_worker()->block_idx)
#define blockDim (::emu::tl_worker()->block_dim)
#define gridDim (::emu::tl_worker()->grid_dim)

inline void __syncthreads() { ::emu::yield_to_scheduler(::emu::AT_BLOCK); }
inline int __syncthreads_or(int pred) {
  ::emu::Worker* w = ::emu::tl_worker();
  w->or_accum |= (pred != 0);
  ::emu::yield_to_scheduler(::emu::AT_BLOCK);
  return w->or_result;
}
template <typename V>
inline V __shfl_xor(V v, int mask, int = 64) { return ::emu::shfl_xor(v, mask); }

inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicAdd(float* p, float v) {
  uint32_t* u = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
  for (;;) {
    float f;
    memcpy(&f, &old, 4);
    f += v;
    uint32_t nu;
    memcpy(&nu, &f, 4);
    if (__atomic_compare_exchange_n(u, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      float r;
      memcpy(&r, &old, 4);
      return r;
    }
  }
}

inline double atomicAdd(double* p, double v) {
  uint64_t* u = reinterpret_cast<uint64_t*>(p);
  uint64_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
  for (;;) {
    double f;
    memcpy(&f, &old, 8);
    f += v;
    uint64_t nu;
    memcpy(&nu, &f, 8);
    if (__atomic_compare_exchange_n(u, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      double r;
      memcpy(&r, &old, 8);
      return r;
    }
  }
}

#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) ::emu::mfma_16x16x32_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) ::emu::mfma_32x32x16_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) ::emu::mfma_16x16x4_f32((a), (b), (c))
// fp16 build: same fragment layouts, operands converted by bf16_bits_to_f32 above (which is then the fp16 conversion)
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) ::emu::mfma_16x16x32_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) ::emu::mfma_32x32x16_bf16((a), (b), (c))

// direct-to-LDS load: destination = (wave-uniform LDS base) + lane * size; synchronous in the emulator
inline void emu_global_load_lds(const void* g, void* lds_base, int size) {
  memcpy(static_cast<char*>(lds_base) + (size_t)::emu::my_lane() * size, g, size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu_global_load_lds((const void*)(g), (void*)(l), (size))

// v_dot2c_f32_bf16: acc + a.x*b.x + a.y*b.y on packed bf16 pairs (operands taken as raw 32-bit patterns: passing
// __bf16 ext-vectors by value through the host ABI is not bit-preserving)
inline float emu_fdot2_bf16(uint32_t ua, uint32_t ub, float c) {
  return c + ::emu::bf16_bits_to_f32((uint16_t)(ua & 0xffff)) * ::emu::bf16_bits_to_f32((uint16_t)(ub & 0xffff)) +
         ::emu::bf16_bits_to_f32((uint16_t)(ua >> 16)) * ::emu::bf16_bits_to_f32((uint16_t)(ub >> 16));
}
#define __builtin_amdgcn_fdot2_f32_bf16(a, b, c, clamp) \
  emu_fdot2_bf16(__builtin_bit_cast(uint32_t, (a)), __builtin_bit_cast(uint32_t, (b)), (c))
#define __builtin_amdgcn_fdot2(a, b, c, clamp) \
  emu_fdot2_bf16(__builtin_bit_cast(uint32_t, (a)), __builtin_bit_cast(uint32_t, (b)), (c))
// ds_read_b64_tr_b16 (gfx950), semantics as probed on hardware (tools/probe_tr16.hip): within a 16-lane group, lane i
// element j = element (i & 3) of the 8 bytes addressed by lane 4*j + (i >> 2)
typedef short emu_v4s __attribute__((ext_vector_type(4)));
inline emu_v4s emu_ds_read_tr16_b64(const void* p) {
  uint32_t pub[2];
  memcpy(pub, p, 8);
  ::emu::wave_publish_and_sync(pub, 2);
  const unsigned l = ::emu::my_lane(), base = l & ~15u, i = l & 15u;
  emu_v4s out;
  for (unsigned j = 0; j < 4; ++j) {
    const uint16_t* src = reinterpret_cast<const uint16_t*>(::emu::wave_peer(base + 4 * j + (i >> 2)));
    out[j] = (short)src[i & 3];
  }
  ::emu::wave_op_done();
  return out;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu_ds_read_tr16_b64((const void*)(p))
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)            /* loads are synchronous in the emulator */
#define __builtin_amdgcn_s_barrier() __syncthreads()
/* the lanes of a wave run in lockstep on the hardware; the emulator's fibers meet here (LDS hand-offs inside one wave) */
inline void emu_wave_barrier() { uint32_t z = 0; ::emu::wave_publish_and_sync(&z, 1); ::emu::wave_op_done(); }
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
#define __builtin_amdgcn_sched_barrier(m) ((void)0)      /* instruction-scheduling fence: no meaning on the host */
/* correctly rounded fp32 arithmetic: what the host compiler does anyway (no -ffast-math, no contraction across these calls) */
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
#define __builtin_amdgcn_readfirstlane(x) (x)                 /* callers only pass wave-uniform values */

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  ::emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
