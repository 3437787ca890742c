// Diagnostic: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds s[i] = i (16-bit); lane j of every
// 16-lane group supplies the address of row (4*group + (j >> 2)), 4-element chunk (j & 3) of a [rows][64]
// image; prints which (row, col) elements each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (short)i;
  __syncthreads();
  const int j = threadIdx.x & 15, grp = threadIdx.x >> 4;
  __attribute__((address_space(3))) s4* p = (__attribute__((address_space(3))) s4*)(s + (grp * 4 + (j >> 2)) * 64 + (j & 3) * 4);
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  short* d; (void)hipMalloc(&d, 64 * 4 * 2);
  k<<<1, 64>>>(d);
  short h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf("  (r%d,c%2d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
    printf("\n");
    if (l == 19) { printf("...\n"); l = 47; }
  }
  return 0;
}
