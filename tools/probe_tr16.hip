// Probe of gfx950 ds_read_b64_tr_b16 lane/element semantics (prints what each lane receives for known addresses).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, const int* addr) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
  int h_addr[64]; unsigned short h_out[256];
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h_addr[l] = 4 * l;
      else if (pat == 1) h_addr[l] = 64 * l + 4 * (l % 7);           // distinct rows
      else h_addr[l] = 4 * ((l * 37 + 11) % 1024);
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_addr);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int src_lane = 16 * (l >> 4) + 4 * j + ((l & 15) >> 2);
        const int expect = h_addr[src_lane] + (l & 3);
        if (h_out[l * 4 + j] != expect) ++bad;
      }
    printf("pattern %d: hypothesis mismatches %d\n", pat, bad);
    if (bad || pat == 0)
      for (int l = 0; l < 64; ++l) printf("  lane %2d addr %5d -> %5d %5d %5d %5d\n", l, h_addr[l], h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
  }
  return 0;
}
