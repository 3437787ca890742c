// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float aval, float bval) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  a[0] = (_Float16)aval; b[0] = (_Float16)bval;
  f32x16 c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  float vals[][2] = {{1e-5f, 1024.f}, {3e-6f, 1024.f}, {6e-8f, 16384.f}, {1.0f, 1e-5f}, {1e-5f, 1e-5f}, {2e-4f, 1.f}};
  for (auto& v : vals) {
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d, v[0], v[1]);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    float ea = (float)(_Float16)v[0], eb = (float)(_Float16)v[1];
    printf("a=%g (fp16 %g) b=%g (fp16 %g): mfma=%g expected=%g\n", v[0], ea, v[1], eb, h, 2.0f * ea * eb);
  }
  return 0;
}
