// HBM / L2->LDS bandwidth probe for MI355X: what do plain streaming kernels reach on this box?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_read(const u32x4* p, size_t n, unsigned* out, int unroll) {
  unsigned acc = 0;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i + 3 * st < n; i += 4 * st) {
    u32x4 a = p[i], b = p[i + st], c = p[i + 2 * st], d = p[i + 3 * st];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  for (; i < n; i += st) acc += p[i].x;
  if (acc == 0x12345678u) out[0] = acc;
}
// block-contiguous variant: each block streams its own contiguous slab
__global__ void __launch_bounds__(256) k_read_slab(const u32x4* p, size_t n, unsigned* out) {
  unsigned acc = 0;
  const size_t per = n / gridDim.x;
  const u32x4* q = p + per * blockIdx.x;
  for (size_t i = threadIdx.x; i + 768 < per; i += 1024) {
    u32x4 a = q[i], b = q[i + 256], c = q[i + 512], d = q[i + 768];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_copy(const u32x4* p, u32x4* q, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i + 3 * st < n; i += 4 * st) {
    u32x4 a = p[i], b = p[i + st], c = p[i + 2 * st], d = p[i + 3 * st];
    q[i] = a; q[i + st] = b; q[i + 2 * st] = c; q[i + 3 * st] = d;
  }
  for (; i < n; i += st) q[i] = p[i];
}
__global__ void __launch_bounds__(256) k_write(u32x4* q, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  const u32x4 v = {1u, 2u, 3u, (unsigned)i};
  for (; i < n; i += st) q[i] = v;
}
// direct-to-LDS streaming: each block stages CH KB chunks of its slab into LDS (NBUF buffers), no consumer
template <int KB, int NBUF>
__global__ void __launch_bounds__(256) k_glds(const u32x4* p, size_t n, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * KB * 1024];
  const size_t per = n / gridDim.x;                 // 16-byte chunks per block
  const u32x4* q = p + per * blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int ROUNDS = KB * 1024 / 16 / 256;      // 256-lane rounds per tile
  const size_t tiles = per / (ROUNDS * 256);
  for (size_t t = 0; t < tiles; ++t) {
    unsigned char* buf = lds + (t % NBUF) * KB * 1024;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q + (t * ROUNDS + r) * 256 + threadIdx.x),
                                       (__attribute__((address_space(3))) void*)(buf + (r * 256 + wave * 64) * 16), 16, 0, 0);
    if (NBUF == 1) { __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8)); }                      // vmcnt(0)
    else if (NBUF == 2) { __builtin_amdgcn_s_waitcnt((ROUNDS & 15) | (7 << 4) | (15 << 8) | ((ROUNDS >> 4) << 14)); }   // one tile in flight
    else { __builtin_amdgcn_s_waitcnt(((2 * ROUNDS) & 15) | (7 << 4) | (15 << 8) | (((2 * ROUNDS) >> 4) << 14)); }
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));
  __syncthreads();
  if (lds[threadIdx.x] == 77 && lds[threadIdx.x + 300] == 78) out[0] = 1;
}

// as k_glds<KB,2> but with access patterns of the convolution kernels: PERM: the 16-byte chunks of each 128-byte line
// are fetched in XOR-permuted lane order (LDS swizzle applied on the source address); STRIDE: each 8-lane group reads
// 128 contiguous bytes, groups are STRIDE bytes apart (pixel rows of a wider tensor)
template <int KB, int PERM, int STRIDE>
__global__ void __launch_bounds__(256) k_glds_pat(const unsigned char* p, size_t nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * KB * 1024];
  const size_t per = nbytes / gridDim.x;
  const unsigned char* q = p + per * blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int ROUNDS = KB * 1024 / 16 / 256;
  const size_t span = (size_t)ROUNDS * 32 * STRIDE;            // bytes of address space one tile covers (32 groups per round)
  const size_t tiles = per / span;
  const int grp = threadIdx.x >> 3, pos = threadIdx.x & 7;
  const unsigned lane_off = grp * STRIDE + ((pos ^ (PERM ? (grp & 6) : 0)) << 4);
  for (size_t t = 0; t < tiles; ++t) {
    unsigned char* buf = lds + (t & 1) * KB * 1024;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q + t * span + (size_t)r * 32 * STRIDE + lane_off),
                                       (__attribute__((address_space(3))) void*)(buf + (r * 256 + wave * 64) * 16), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt((ROUNDS & 15) | (7 << 4) | (15 << 8) | ((ROUNDS >> 4) << 14));
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));
  __syncthreads();
  if (lds[threadIdx.x] == 77 && lds[threadIdx.x + 300] == 78) out[0] = 1;
}

// patch-like access: a tile = ROWS rows of RB contiguous bytes at a row pitch of PITCH bytes; consecutive tiles of a block
// move along the row by RB_ADV bytes (halo overlap when RB_ADV < RB); a block owns a band of rows.  2 LDS buffers.
template <int ROWS, int RB, int RB_ADV>
__global__ void __launch_bounds__(256) k_glds_rows(const unsigned char* p, size_t pitch, int tiles_per_row, int bands, int reps, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 32 * 1024];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int CHUNKS = ROWS * RB / 16, ROUNDS = (CHUNKS + 255) / 256;
  const int band = blockIdx.x % bands;
  const unsigned char* q = p + (size_t)band * ROWS * pitch;
  int n = 0;
  for (int rep = 0; rep < reps; ++rep)
    for (int t = blockIdx.x / bands; t < tiles_per_row; t += gridDim.x / bands, ++n) {
      unsigned char* buf = lds + (n & 1) * 32 * 1024;
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        int L = r * 256 + threadIdx.x;
        if (L >= CHUNKS) L = 0;
        const int row = L / (RB / 16), cb = L % (RB / 16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q + (size_t)row * pitch + (size_t)t * RB_ADV + cb * 16),
                                         (__attribute__((address_space(3))) void*)(buf + (r * 256 + wave * 64) * 16), 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt((ROUNDS & 15) | (7 << 4) | (15 << 8) | ((ROUNDS >> 4) << 14));
      __builtin_amdgcn_s_barrier();
    }
  __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));
  __syncthreads();
  if (lds[threadIdx.x] == 77 && lds[threadIdx.x + 300] == 78) out[0] = 1;
}

template <typename F>
static float timeit(F f, int iters = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}
int main() {
  const size_t bytes = (size_t)1 << 30;     // 1 GiB
  const size_t n = bytes / 16;
  u32x4 *p, *q; unsigned* out;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(p, 1, bytes)); CK(hipMemset(q, 2, bytes));
  for (int blocks : {512, 1024, 2048, 4096, 8192}) {
    float t = timeit([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, p, n, out, 4); });
    printf("read  grid-stride x4  blocks %5d : %.3f ms  %.2f TB/s\n", blocks, t, bytes / t / 1e9);
  }
  for (int blocks : {512, 1024, 2048, 4096}) {
    float t = timeit([&] { hipLaunchKernelGGL(k_read_slab, dim3(blocks), dim3(256), 0, 0, p, n, out); });
    printf("read  slab per block  blocks %5d : %.3f ms  %.2f TB/s\n", blocks, t, bytes / t / 1e9);
  }
  for (int blocks : {1024, 4096}) {
    float t = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, p, q, n); });
    printf("copy                  blocks %5d : %.3f ms  %.2f TB/s (r+w)\n", blocks, t, 2.0 * bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, q, n); });
    printf("write                 blocks %5d : %.3f ms  %.2f TB/s\n", blocks, t, bytes / t / 1e9);
  }
  for (int blocks : {512, 1024, 2048}) {
    float t;
    t = timeit([&] { hipLaunchKernelGGL((k_glds<32, 1>), dim3(blocks), dim3(256), 0, 0, p, n, out); });
    printf("glds 32KB x1 buffers  blocks %5d : %.3f ms  %.2f TB/s\n", blocks, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((k_glds<32, 2>), dim3(blocks), dim3(256), 0, 0, p, n, out); });
    printf("glds 32KB x2 buffers  blocks %5d : %.3f ms  %.2f TB/s\n", blocks, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((k_glds<16, 3>), dim3(blocks), dim3(256), 0, 0, p, n, out); });
    printf("glds 16KB x3 buffers  blocks %5d : %.3f ms  %.2f TB/s\n", blocks, t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((k_glds<8, 3>), dim3(blocks), dim3(256), 0, 0, p, n, out); });
    printf("glds  8KB x3 buffers  blocks %5d : %.3f ms  %.2f TB/s\n", blocks, t, bytes / t / 1e9);
  }
  {
    const unsigned char* pb = (const unsigned char*)p;
    float t;
    t = timeit([&] { hipLaunchKernelGGL((k_glds_pat<32, 0, 128>), dim3(512), dim3(256), 0, 0, pb, bytes, out); });
    printf("glds 32KB x2 contiguous lines            : %.3f ms  %.2f TB/s\n", t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((k_glds_pat<32, 1, 128>), dim3(512), dim3(256), 0, 0, pb, bytes, out); });
    printf("glds 32KB x2 XOR-permuted chunks in line : %.3f ms  %.2f TB/s\n", t, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((k_glds_pat<32, 0, 256>), dim3(512), dim3(256), 0, 0, pb, bytes, out); });
    printf("glds 32KB x2 128 B of every 256 B        : %.3f ms  %.2f TB/s (bytes requested: half)\n", t, bytes / 2 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((k_glds_pat<32, 1, 256>), dim3(512), dim3(256), 0, 0, pb, bytes, out); });
    printf("glds 32KB x2 128 of 256 B, permuted      : %.3f ms  %.2f TB/s (bytes requested: half)\n", t, bytes / 2 / t / 1e9);
  }
  {
    // "image" of 16384 rows; 10-row x 2304-byte patches (18 px x 64 ch bf16) advancing 2048 bytes, as conv_stream stages them
    const unsigned char* pb = (const unsigned char*)p;
    for (size_t pitch : {(size_t)65536, (size_t)65536 + 256, (size_t)65536 + 2048, (size_t)32768}) {
      const int tiles_per_row = 28, rows = (int)(bytes / pitch), bands = rows / 10 < 512 ? rows / 10 : 512;
      const int reps = 8;
      float t = timeit([&] { hipLaunchKernelGGL((k_glds_rows<10, 2304, 2048>), dim3(bands), dim3(256), 0, 0, pb, pitch, tiles_per_row, bands, reps, out); });
      const double moved = (double)bands * tiles_per_row * reps * 10 * 2304;
      printf("glds patch rows 10 x 2304 B, row pitch %6zu B, %d blocks : %.3f ms  %.2f TB/s staged\n", pitch, bands, t, moved / t / 1e9);
    }
  }
  for (size_t mb : {8, 32, 128, 512}) {        // working set re-read 'reps' times: L2 (4 MB / XCD) and MALL (256 MB) residency
    const size_t wbytes = mb << 20;
    const int reps = (int)(bytes / wbytes) * 2;
    float t = timeit([&] {
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_glds<32, 2>), dim3(512), dim3(256), 0, 0, p, wbytes / 16, out);
    });
    printf("glds 32KB x2, %4zu MB working set re-read %3d x : %.3f ms  %.2f TB/s\n", mb, reps, t, (double)wbytes * reps / t / 1e9);
  }
  return 0;
}
