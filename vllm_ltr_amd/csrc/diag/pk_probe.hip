// Diagnostic (round 6): does a packed-f32 VALU chain with op_sel operands (the form hipcc emits for
//   x = (r - mean) * rstd * gamma + beta     with (mean, rstd) one float2 shared by the four components)
// compute the same values when another kernel's waves (MFMA, memory streaming, plain VALU) share the SIMD?
// Written after profiles/r06_rln_fault.txt traced the round-5 wrong-score fault to the lo halves of exactly this chain in lanes
// 48-63 of splitk_epilogue_kernel<LNP, true>.  Victim: the chain as inline asm (so the instruction sequence is fixed) over a large
// array; run alone and beside each aggressor, outputs compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 diag/pk_probe.hip -o pk_probe && ./pk_probe [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef WITH_ROCBLAS
#include <rocblas/rocblas.h>
#endif

typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: asm chain with the compiler's s_nop placement (pk_add, s_nop 0, pk_mul, s_nop 0, pk_fma), MODE 1: no s_nops between,
// MODE 2: plain C (whatever hipcc makes of it)
template <int MODE>
__global__ void __launch_bounds__(256) victim(const float4* __restrict__ r, const float2* __restrict__ st, const float4* __restrict__ g,
                                              const float4* __restrict__ b, float4* __restrict__ o, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 rr = r[i];
    const float2 s2 = st[i >> 3];
    const float4 gg = g[i & 255], bb = b[i & 255];
    if (MODE == 2) {
      float4 x;
      x.x = (rr.x - s2.x) * s2.y * gg.x + bb.x; x.y = (rr.y - s2.x) * s2.y * gg.y + bb.y;
      x.z = (rr.z - s2.x) * s2.y * gg.z + bb.z; x.w = (rr.w - s2.x) * s2.y * gg.w + bb.w;
      o[i] = x;
    } else {
      v2f a0 = {rr.x, rr.y}, a1 = {rr.z, rr.w}, s = {s2.x, s2.y}, g0 = {gg.x, gg.y}, g1 = {gg.z, gg.w}, b0 = {bb.x, bb.y}, b1 = {bb.z, bb.w};
      if (MODE == 0)
        asm volatile("v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_add_f32 %1, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_mul_f32 %0, %0, %2 op_sel:[0,1]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_fma_f32 %0, %0, %3, %5\n\t"
                     "v_pk_mul_f32 %1, %1, %2 op_sel:[0,1]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_fma_f32 %1, %1, %4, %6"
                     : "+v"(a0), "+v"(a1) : "v"(s), "v"(g0), "v"(g1), "v"(b0), "v"(b1));
      else if (MODE == 3) {
        v2f sy = {s2.y, s2.y};
        asm volatile("v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_add_f32 %1, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_mul_f32 %0, %0, %7 op_sel_hi:[1,0]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_fma_f32 %0, %0, %3, %5\n\t"
                     "v_pk_mul_f32 %1, %1, %7 op_sel_hi:[1,0]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_fma_f32 %1, %1, %4, %6"
                     : "+v"(a0), "+v"(a1) : "v"(s), "v"(g0), "v"(g1), "v"(b0), "v"(b1), "v"(sy));
      } else
        asm volatile("v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_mul_f32 %0, %0, %2 op_sel:[0,1]\n\t"
                     "v_pk_fma_f32 %0, %0, %3, %5\n\t"
                     "v_pk_add_f32 %1, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_mul_f32 %1, %1, %2 op_sel:[0,1]\n\t"
                     "v_pk_fma_f32 %1, %1, %4, %6"
                     : "+v"(a0), "+v"(a1) : "v"(s), "v"(g0), "v"(g1), "v"(b0), "v"(b1));
      o[i] = make_float4(a0.x, a0.y, a1.x, a1.y);
    }
  }
}

__global__ void __launch_bounds__(256) aggr_mfma(float* sink, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  for (int it = 0; it < iters; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc3, 0, 0, 0);
  }
  const float v = acc0[0] + acc1[1] + acc2[2] + acc3[3];
  if (v == 1.2345e-30f) sink[0] = v;
}
__global__ void __launch_bounds__(256) aggr_valu(float* sink, int iters) {
  float x = threadIdx.x * 0.5f, y = 1.0001f;
  for (int it = 0; it < iters; ++it) { x = fmaf(x, y, 0.25f); y = fmaf(y, 0.99999f, 1e-6f); }
  if (x + y == 1.2345e-30f) sink[0] = x;
}
__global__ void __launch_bounds__(256) aggr_mem(const float4* __restrict__ src, float* sink, size_t n4, int rounds) {
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = src[i]; acc += v.x + v.y * v.z + v.w; }
  if (acc == 1.2345e-30f) sink[0] = acc;
}
__global__ void __launch_bounds__(256) aggr_lds(float* sink, int iters) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = i;
  __syncthreads();
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) acc += sm[(threadIdx.x * 17 + it * 33) & 4095];
  if (acc == 1.2345e-30f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20;
  const size_t n = (size_t)16 << 20;            // 16 M float4 in, 16 M out (256 MB each)
  std::vector<float4> hr(n), hg(256), hb(256);
  std::vector<float2> hs(n / 8);
  srand(7);
  auto fr = [] { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; };
  for (auto& v : hr) v = make_float4(fr() * 2, fr() * 2, fr() * 2, fr() * 2);
  for (auto& v : hs) v = make_float2(fr() * 0.03f, 0.9f + 0.05f * fr());
  for (auto& v : hg) v = make_float4(1 + .3f * fr(), 1 + .3f * fr(), 1 + .3f * fr(), 1 + .3f * fr());
  for (auto& v : hb) v = make_float4(.3f * fr(), .3f * fr(), .3f * fr(), .3f * fr());
  float4 *dr, *dg, *db, *dout, *dsrc; float2* ds; float* sink;
  (void)hipMalloc(&dr, n * 16); (void)hipMalloc(&dout, n * 16); (void)hipMalloc(&dg, 4096); (void)hipMalloc(&db, 4096); (void)hipMalloc(&ds, n);
  (void)hipMalloc(&dsrc, (size_t)1 << 30); (void)hipMalloc(&sink, 64);
  (void)hipMemcpy(dr, hr.data(), n * 16, hipMemcpyHostToDevice); (void)hipMemcpy(dg, hg.data(), 4096, hipMemcpyHostToDevice);
  (void)hipMemcpy(db, hb.data(), 4096, hipMemcpyHostToDevice); (void)hipMemcpy(ds, hs.data(), n, hipMemcpyHostToDevice);
  (void)hipMemset(dsrc, 0, (size_t)1 << 30);
  hipStream_t sa, sb;
  (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  std::vector<float4> ref(n), got(n);
  auto run_victim = [&](int mode) {
    (void)hipMemsetAsync(dout, 0xff, n * 16, sa);
    if (mode == 0) victim<0><<<2048, 256, 0, sa>>>(dr, ds, dg, db, dout, n);
    else if (mode == 1) victim<1><<<2048, 256, 0, sa>>>(dr, ds, dg, db, dout, n);
    else if (mode == 2) victim<2><<<2048, 256, 0, sa>>>(dr, ds, dg, db, dout, n);
    else victim<3><<<2048, 256, 0, sa>>>(dr, ds, dg, db, dout, n);
  };
#ifdef WITH_ROCBLAS
  rocblas_handle rh; rocblas_create_handle(&rh); rocblas_set_stream(rh, sb);
  const int G = 2048;
  void *gz, *gc_;
  (void)hipMalloc(&gz, (size_t)G * G * 2); (void)hipMalloc(&gc_, (size_t)G * G * 2);
  (void)hipMemset(gz, 0, (size_t)G * G * 2);
#endif
  const char* aggr_name[] = {"none", "mfma", "valu", "mem", "lds", "mfma (few waves)", "library fp16 GEMM"};
  for (int mode = 0; mode < 4; ++mode) {
    run_victim(mode);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(ref.data(), dout, n * 16, hipMemcpyDeviceToHost);
    // host check of the idle run (same roundings: sub, mul, fused multiply-add)
    long host_bad = 0;
    for (size_t i = 0; i < n; i += 97) {
      const float2 s2 = hs[i >> 3]; const float4 gg = hg[i & 255], bb = hb[i & 255], rr = hr[i];
      const float w[4] = {__builtin_fmaf((rr.x - s2.x) * s2.y, gg.x, bb.x), __builtin_fmaf((rr.y - s2.x) * s2.y, gg.y, bb.y),
                          __builtin_fmaf((rr.z - s2.x) * s2.y, gg.z, bb.z), __builtin_fmaf((rr.w - s2.x) * s2.y, gg.w, bb.w)};
      const float gq[4] = {ref[i].x, ref[i].y, ref[i].z, ref[i].w};
      for (int c = 0; c < 4; ++c) if (!(__builtin_fabsf(w[c] - gq[c]) <= 1e-6f * (1 + __builtin_fabsf(w[c])))) ++host_bad;
    }
    printf("mode %d: idle run vs host (sampled): %ld bad\n", mode, host_bad);
#ifdef WITH_ROCBLAS
    const int n_ag = 7;
#else
    const int n_ag = 6;
#endif
    for (int ag = 0; ag < n_ag; ++ag) {
      long bad_runs = 0, bad_elems = 0, by_comp[4] = {0, 0, 0, 0}, by_quarter[4] = {0, 0, 0, 0};
      for (int it = 0; it < rounds; ++it) {
        (void)hipDeviceSynchronize();
        if (ag == 1) aggr_mfma<<<2048, 256, 0, sb>>>(sink, 40000);
        if (ag == 2) aggr_valu<<<2048, 256, 0, sb>>>(sink, 200000);
        if (ag == 3) aggr_mem<<<1024, 256, 0, sb>>>(dsrc, sink, (size_t)64 << 20, 3);
        if (ag == 4) aggr_lds<<<2048, 256, 0, sb>>>(sink, 200000);
        if (ag == 5) aggr_mfma<<<256, 256, 0, sb>>>(sink, 160000);
#ifdef WITH_ROCBLAS
        if (ag == 6) {
          const float one = 1.f, nul = 0.f;
          for (int k = 0; k < 30; ++k)
            rocblas_gemm_ex(rh, rocblas_operation_none, rocblas_operation_none, G, G, G, &one, gz, rocblas_datatype_f16_r, G, gz, rocblas_datatype_f16_r, G, &nul,
                            gc_, rocblas_datatype_f16_r, G, gc_, rocblas_datatype_f16_r, G, rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
        }
#endif
        run_victim(mode);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(got.data(), dout, n * 16, hipMemcpyDeviceToHost);
        long e = 0;
        if (memcmp(got.data(), ref.data(), n * 16)) {
          for (size_t i = 0; i < n; ++i) {
            const float a[4] = {got[i].x, got[i].y, got[i].z, got[i].w}, w[4] = {ref[i].x, ref[i].y, ref[i].z, ref[i].w};
            for (int c = 0; c < 4; ++c) if (memcmp(&a[c], &w[c], 4)) { ++e; ++by_comp[c]; ++by_quarter[(i & 63) >> 4]; }
          }
        }
        bad_runs += e != 0; bad_elems += e;
      }
      printf("mode %d beside %-16s: %ld of %d runs differ, %ld elements (x %ld y %ld z %ld w %ld | lanes 0-15 %ld 16-31 %ld 32-47 %ld 48-63 %ld)\n", mode,
             aggr_name[ag], bad_runs, rounds, bad_elems, by_comp[0], by_comp[1], by_comp[2], by_comp[3], by_quarter[0], by_quarter[1], by_quarter[2], by_quarter[3]);
      fflush(stdout);
    }
  }
  return 0;
}
