// Diagnostic: internal precision of v_mfma_f32_32x32x16_f16 (products + accumulation) vs exact.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A [32][16], B [16][32] (as Bt [32][16]), C [32][32]
__global__ void k(const _Float16* A, const _Float16* Bt, const float* C, float* D) {
  int l = threadIdx.x;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e]; b[e] = Bt[(l & 31) * 16 + 8 * (l >> 5) + e]; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
  std::vector<_Float16> A(512), Bt(512); std::vector<float> C(1024), D(1024);
  _Float16 *dA, *dB; float *dC, *dD;
  (void)hipMalloc(&dA, 1024); (void)hipMalloc(&dB, 1024); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dD, 4096);
  for (int trial = 0; trial < 4; ++trial) {
    srand(trial);
    double scaleA = trial == 3 ? 100.0 : 1.0;
    for (int i = 0; i < 512; ++i) { A[i] = (_Float16)(scaleA * (rand() / (double)RAND_MAX * 2 - 1)); Bt[i] = (_Float16)((rand() / (double)RAND_MAX * 2 - 1)); }
    for (int i = 0; i < 1024; ++i) C[i] = trial == 0 ? 0.f : (float)((rand() / (double)RAND_MAX * 2 - 1) * (trial == 2 ? 1000.0 : 1.0));
    (void)hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, dA, dB, dC, dD);
    (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double maxrel = 0, maxabs = 0, maxulp = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ex = C[i * 32 + j], mag = fabs(ex);
      for (int kk = 0; kk < 16; ++kk) { double p = (double)A[i * 16 + kk] * (double)Bt[j * 16 + kk]; ex += p; mag = fmax(mag, fabs(p)); }
      double err = fabs(D[i * 32 + j] - ex);
      maxabs = fmax(maxabs, err); maxrel = fmax(maxrel, err / mag);
      double ulp = ldexp(1.0, (int)floor(log2(fmax(fabs(ex), 1e-30))) - 23);
      maxulp = fmax(maxulp, err / ulp);
    }
    printf("trial %d: max abs err %.3e, max err/largest-term %.3e (2^%.1f), max err in result-ulps %.2f\n", trial, maxabs, maxrel, log2(maxrel), maxulp);
  }
  return 0;
}
