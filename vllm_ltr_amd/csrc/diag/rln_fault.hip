// Diagnostic (round 6): kernel-level reproducer of round 5's "concurrency-dependent wrong scores" (profiles/r05_rln_probe.txt;
// root cause in profiles/r06_rln_fault.txt: a packed-f32 operand-select form that MI355X mis-executes in lanes 48-63 while a library
// fp16 GEMM shares the CU).  Includes ltr_gemm.hip itself (the kernels live in an anonymous namespace), so that ONE kernel
// instance can be launched on FIXED inputs - written once by the host, never touched again - next to unrelated work on another
// stream; every run is compared bit for bit with the same launch on an idle device:
//   E1  splitk_epilogue_kernel<LNP, true> beside a streaming kernel (b) / another lane's fc2 launch (c);
//   E2  the fc2 launch of an OPT-350m layer through launch_gemm (128 x 256 kernel in 4 K parts + the reduce kernel) beside (b);
//   E3  E2's launch pair on two streams at once (the lanes of run_forward), each with its own buffers;
//   E4  the reduce kernel beside a long MFMA kernel of our own (an 8,192-row GEMM);
//   E5  the reduce kernel beside LIBRARY fp16 GEMMs (rocBLAS; -DWITH_ROCBLAS) - the co-runner a serving engine's backbone is, and
//       the only one that exposes the fault: built with -DLTR_RLN_FAULT_SHAPE (round 5's expression: `v_pk_mul_f32 ... op_sel:[0,1]`
//       in the kernel) 7-11 of 20 runs differ, 160-2,400 values each; the shipped expression: 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DLTR_RLN_FAULT_SHAPE] -DWITH_ROCBLAS -I<csrc> -I<repo>/include diag/rln_fault.hip -lrocblas
//   rln_fault [iters] [M] [quick]         quick: E5 only.  Exit code 1 when any run differs.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ltr_gemm.hip"
#ifdef WITH_ROCBLAS
#include <rocblas/rocblas.h>
#endif

namespace ltr { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
using namespace ltr;

static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }
template <class T> T* dev(const std::vector<T>& v) { T* p; (void)hipMalloc(&p, v.size() * sizeof(T)); (void)hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return p; }
template <class T> T* dalloc(size_t n) { T* p; (void)hipMalloc(&p, n * sizeof(T)); (void)hipMemset(p, 0xff, n * sizeof(T)); return p; }
template <class T> std::vector<T> host(const T* p, size_t n) { std::vector<T> v(n); (void)hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost); return v; }

// unrelated work for the other stream: streams a buffer and keeps the VALU busy, many small workgroups
__global__ void __launch_bounds__(256) noise_kernel(const float4* __restrict__ src, float* __restrict__ sink, size_t n4, int rounds) {
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
      const float4 v = src[i];
      acc = fmaf(v.x, v.y, acc) + v.z * v.w;
    }
  if (acc == 1.2345e-30f) sink[0] = acc;
}

struct Bufs {          // everything one fc2 launch of a post-LN layer touches
  int M, N, K;
  __half *a_hi, *a_lo, *w;
  float *bias, *resid0, *x, *ln_gamma, *r_gamma, *r_beta;
  float2 *r_stats, *stats_out;
  __half *ln_hi, *ln_lo;
  float* splitk;
  size_t splitk_bytes;
  int32_t* err;
};

static Bufs make_bufs(int M, int N, int K, unsigned seed) {
  srand(seed);
  Bufs b{};
  b.M = M; b.N = N; b.K = K;
  std::vector<__half> ahi((size_t)M * K), alo((size_t)M * K), w((size_t)N * K);
  for (size_t i = 0; i < ahi.size(); ++i) { const float a = frand(); ahi[i] = __float2half(a); alo[i] = __float2half(a - __half2float(ahi[i])); }
  for (auto& x : w) x = __float2half(frand() * 0.05f);
  std::vector<float> bias(N), resid((size_t)M * N), g(N), rg(N), rb(N);
  for (auto& x : bias) x = frand();
  for (auto& x : resid) x = frand() * 2.f;
  for (auto& x : g) x = 1.f + 0.5f * frand();
  for (auto& x : rg) x = 1.f + 0.5f * frand();
  for (auto& x : rb) x = 0.3f * frand();
  std::vector<float2> st((size_t)(N / 64) * M);
  for (int m = 0; m < M; ++m)
    for (int p = 0; p < N / 64; ++p) {
      double mu = 0, q = 0;
      for (int c = 0; c < 64; ++c) mu += resid[(size_t)m * N + p * 64 + c];
      mu /= 64;
      for (int c = 0; c < 64; ++c) { const double d = resid[(size_t)m * N + p * 64 + c] - mu; q += d * d; }
      st[(size_t)p * M + m] = make_float2((float)mu, (float)q);
    }
  // slab-major A image [K/32][M][32]
  std::vector<__half> shi(ahi.size()), slo(alo.size());
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) {
      const size_t o = ((size_t)(k >> 5) * M + m) * 32 + (k & 31);
      shi[o] = ahi[(size_t)m * K + k]; slo[o] = alo[(size_t)m * K + k];
    }
  b.a_hi = dev(shi); b.a_lo = dev(slo);
  __half* wr = dev(w);
  b.w = dalloc<__half>(w.size());
  launch_pack_weight(wr, b.w, N, K, 0);
  b.bias = dev(bias); b.resid0 = dev(resid); b.x = dalloc<float>((size_t)M * N);
  b.ln_gamma = dev(g); b.r_gamma = dev(rg); b.r_beta = dev(rb);
  b.r_stats = dev(st); b.stats_out = dalloc<float2>((size_t)(N / 64) * M);
  b.ln_hi = dalloc<__half>((size_t)M * N); b.ln_lo = dalloc<__half>((size_t)M * N);
  b.splitk_bytes = (size_t)4 * M * N * 4;
  b.splitk = dalloc<float>(b.splitk_bytes / 4);
  b.err = dalloc<int32_t>(1);
  (void)hipMemset(b.err, 0, 4);
  (void)hipDeviceSynchronize();
  return b;
}

static GemmArgs fc2_args(const Bufs& b) {      // ChunkRun::layer's fc2 of a post-LN layer with the fold (in place: resid == out)
  GemmArgs g{};
  g.a = AOp{b.a_hi, b.a_lo}; g.w = b.w; g.bias = b.bias; g.resid = b.x; g.out_f32 = b.x; g.M = b.M; g.N = b.N; g.K = b.K; g.a_slab = 1;
  g.splitk_ws = b.splitk; g.splitk_ws_bytes = b.splitk_bytes;
  g.ln_gamma = b.ln_gamma; g.ln_out = AOp{b.ln_hi, b.ln_lo}; g.ln_stats_out = b.stats_out; g.err_flag = b.err;
  g.rln_stats = b.r_stats; g.rln_gamma = b.r_gamma; g.rln_beta = b.r_beta; g.rln_parts = b.N / 64;
  return g;
}

static Epilogue reduce_ep(const Bufs& b) {      // what launch_gemm hands to splitk_epilogue_kernel for that launch
  Epilogue ep{};
  ep.bias = b.bias; ep.resid = b.x; ep.out_f32 = b.x; ep.M = b.M; ep.N = b.N; ep.a_slab = 1;
  ep.ln_gamma = b.ln_gamma; ep.ln_hi = b.ln_hi; ep.ln_lo = b.ln_lo; ep.stats_out = b.stats_out; ep.err_flag = b.err;
  ep.r_stats = b.r_stats; ep.r_gamma = b.r_gamma; ep.r_beta = b.r_beta; ep.r_parts = b.N / 64;
  ep.row0 = 0; ep.ldm = b.M;
  return ep;
}

struct Snap { std::vector<float> x; std::vector<__half> hi, lo; std::vector<float2> st; };
static Snap snap(const Bufs& b) {
  return Snap{host(b.x, (size_t)b.M * b.N), host(b.ln_hi, (size_t)b.M * b.N), host(b.ln_lo, (size_t)b.M * b.N), host(b.stats_out, (size_t)(b.N / 64) * b.M)};
}
static void reset_outputs(const Bufs& b, hipStream_t s) {       // x = the residual again (in-place launch), outputs poisoned
  (void)hipMemcpyAsync(b.x, b.resid0, (size_t)b.M * b.N * 4, hipMemcpyDeviceToDevice, s);
  (void)hipMemsetAsync(b.ln_hi, 0xff, (size_t)b.M * b.N * 2, s);
  (void)hipMemsetAsync(b.ln_lo, 0xff, (size_t)b.M * b.N * 2, s);
  (void)hipMemsetAsync(b.stats_out, 0xff, (size_t)(b.N / 64) * b.M * 8, s);
}
static long compare(const char* tag, int it, const Bufs& b, const Snap& ref, const Snap& got, int verbose) {
  long bx = 0, bh = 0, bl = 0, bs = 0;
  int first_row = -1, last_row = -1;
  std::vector<int> rows_bad(b.M, 0);
  for (size_t i = 0; i < ref.x.size(); ++i)
    if (memcmp(&ref.x[i], &got.x[i], 4)) {
      const int row = (int)(i / b.N);
      if (bx < verbose) printf("  %s it %d: x[%d,%d] = %.7g want %.7g\n", tag, it, row, (int)(i % b.N), got.x[i], ref.x[i]);
      ++bx; ++rows_bad[row];
      if (first_row < 0) first_row = row;
      last_row = row;
    }
  for (size_t i = 0; i < ref.hi.size(); ++i) { if (memcmp(&ref.hi[i], &got.hi[i], 2)) ++bh; if (memcmp(&ref.lo[i], &got.lo[i], 2)) ++bl; }
  for (size_t i = 0; i < ref.st.size(); ++i) if (memcmp(&ref.st[i], &got.st[i], 8)) ++bs;
  if (bx || bh || bl || bs) {
    int nrows = 0, full = 0;
    for (int r = 0; r < b.M; ++r) { nrows += rows_bad[r] > 0; full += rows_bad[r] == b.N; }
    printf("%s it %d: MISMATCH x %ld (rows %d, whole rows %d, first %d last %d)  a'.hi %ld  a'.lo %ld  stats %ld\n", tag, it, bx, nrows, full,
           first_row, last_row, bh, bl, bs);
  }
  return bx + bh + bl + bs;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int M = argc > 2 ? atoi(argv[2]) : 1383;
  const bool quick = argc > 3 && !strcmp(argv[3], "quick");
  const int N = 1024, K = 4096;
  hipStream_t sa, sb, sn;
  (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&sn, hipStreamNonBlocking);
  Bufs A = make_bufs(M, N, K, 1), B = make_bufs(M - 100, N, K, 2);
  const size_t noise_n4 = (size_t)64 << 20;     // 1 GiB
  float4* noise_src = dalloc<float4>(noise_n4);
  float* sink = dalloc<float>(4);
  auto noise = [&](int rounds) { noise_kernel<<<1024, 256, 0, sn>>>(noise_src, sink, noise_n4, rounds); };

  // ---- reference of the whole fc2 launch on an idle device
  GemmArgs ga = fc2_args(A), gb = fc2_args(B);
  reset_outputs(A, sa); reset_outputs(B, sb);
  (void)hipDeviceSynchronize();
  int rc = launch_gemm(LTR_W_F16, ga, sa);
  (void)hipDeviceSynchronize();
  rc |= launch_gemm(LTR_W_F16, gb, sb);
  (void)hipDeviceSynchronize();
  printf("reference launches rc=%d (%s)\n", rc, hipGetErrorString(hipGetLastError()));
  const Snap refA = snap(A), refB = snap(B);
  // the K parts of A's launch stay in A.splitk: E1 reduces exactly those
  const Epilogue epA = reduce_ep(A);
  const unsigned nthreads = (unsigned)M * (unsigned)(N / 8);
  const dim3 rgrid((nthreads + 255) / 256);

  long bad = 0;
  // ---- E1a: the reduce kernel alone on fixed inputs, idle device (sanity: equals the reference)
  for (int it = 0; it < 3; ++it) {
    reset_outputs(A, sa);
    splitk_epilogue_kernel<LNP, true><<<rgrid, 256, 0, sa>>>(A.splitk, 4, M, N, epA);
    (void)hipDeviceSynchronize();
    bad += compare("E1a reduce, idle", it, A, refA, snap(A), 4);
  }
  printf("E1a done, bad so far %ld\n", bad);
  // ---- E1b: the same beside the noise kernel on another stream
  long e1b = 0, e1c = 0, e2 = 0, e3 = 0, e4 = 0, e5_total = 0;
  if (!quick) {
  for (int it = 0; it < iters; ++it) {
    reset_outputs(A, sa);
    (void)hipStreamSynchronize(sa);
    noise(1);
    splitk_epilogue_kernel<LNP, true><<<rgrid, 256, 0, sa>>>(A.splitk, 4, M, N, epA);
    (void)hipDeviceSynchronize();
    e1b += compare("E1b reduce + noise", it, A, refA, snap(A), e1b ? 0 : 6) != 0;
  }
  printf("E1b: %ld of %d runs differ\n", e1b, iters);
  // ---- E1c: the same beside B's whole fc2 launch on stream sb
  for (int it = 0; it < iters; ++it) {
    reset_outputs(A, sa); reset_outputs(B, sb);
    (void)hipDeviceSynchronize();
    (void)launch_gemm(LTR_W_F16, gb, sb);
    splitk_epilogue_kernel<LNP, true><<<rgrid, 256, 0, sa>>>(A.splitk, 4, M, N, epA);
    (void)hipDeviceSynchronize();
    e1c += compare("E1c reduce + other lane", it, A, refA, snap(A), e1c ? 0 : 6) != 0;
  }
  printf("E1c: %ld of %d runs differ\n", e1c, iters);
  // ---- E2: the whole fc2 launch beside the noise kernel
  for (int it = 0; it < iters; ++it) {
    reset_outputs(A, sa);
    (void)hipMemsetAsync(A.splitk, 0xff, A.splitk_bytes, sa);
    (void)hipStreamSynchronize(sa);
    noise(1);
    (void)launch_gemm(LTR_W_F16, ga, sa);
    (void)hipDeviceSynchronize();
    e2 += compare("E2 fc2 + noise", it, A, refA, snap(A), e2 ? 0 : 6) != 0;
  }
  printf("E2: %ld of %d runs differ\n", e2, iters);
  // ---- E3: two lanes, alternately issued (a: GEMM parts, b: GEMM parts + reduce interleave as launch_gemm issues them)
  for (int it = 0; it < iters; ++it) {
    reset_outputs(A, sa); reset_outputs(B, sb);
    (void)hipMemsetAsync(A.splitk, 0xff, A.splitk_bytes, sa);
    (void)hipMemsetAsync(B.splitk, 0xff, B.splitk_bytes, sb);
    (void)hipDeviceSynchronize();
    (void)launch_gemm(LTR_W_F16, ga, sa);
    (void)launch_gemm(LTR_W_F16, gb, sb);
    (void)hipDeviceSynchronize();
    const long da = compare("E3 lane a", it, A, refA, snap(A), e3 ? 0 : 6), db = compare("E3 lane b", it, B, refB, snap(B), e3 ? 0 : 6);
    e3 += (da || db);
  }
  printf("E3: %ld of %d runs differ\n", e3, iters);
  // ---- E4: the reduce kernel on fixed inputs beside a LONG MFMA kernel of another stream (a 8,192-row fc2 on the 128 x 256 kernel,
  // three launches ~ 1 ms): the reduce's waves share SIMDs with MFMA waves for their whole life
  {
    Bufs Cb = make_bufs(8192, N, K, 3);
    GemmArgs gc = fc2_args(Cb);
    gc.splitk_ws = nullptr; gc.splitk_ws_bytes = 0;
    (void)launch_gemm(LTR_W_F16, ga, sa);          // A.splitk holds A's K parts again (E2 / E3 poisoned and rewrote it)
    (void)hipDeviceSynchronize();
    for (int it = 0; it < iters; ++it) {
      reset_outputs(A, sa);
      (void)hipMemcpyAsync(Cb.x, Cb.resid0, (size_t)Cb.M * N * 4, hipMemcpyDeviceToDevice, sb);
      (void)hipDeviceSynchronize();
      for (int k = 0; k < 3; ++k) (void)launch_gemm(LTR_W_F16, gc, sb);
      splitk_epilogue_kernel<LNP, true><<<rgrid, 256, 0, sa>>>(A.splitk, 4, M, N, epA);
      (void)hipDeviceSynchronize();
      e4 += compare("E4 reduce + long MFMA kernel", it, A, refA, snap(A), e4 ? 0 : 6) != 0;
    }
    printf("E4: %ld of %d runs differ\n", e4, iters);
  }
  }   // !quick
#ifdef WITH_ROCBLAS
  // ---- E5: the reduce kernel on fixed inputs beside library fp16 GEMMs (what a serving engine's backbone runs; torch's matmul
  // was the co-runner that exposed the round-5 fault): random operands and all-zero operands (same kernel, far less power)
  {
    rocblas_handle rh; rocblas_create_handle(&rh); rocblas_set_stream(rh, sb);
    const int G = 2048;
    std::vector<__half> hm((size_t)G * G);
    for (auto& v : hm) v = __float2half(frand());
    __half *ga_ = dev(hm), *gb_ = dev(hm), *gz = dalloc<__half>((size_t)G * G), *gc_ = dalloc<__half>((size_t)G * G);
    (void)hipMemset(gz, 0, (size_t)G * G * 2);
    (void)launch_gemm(LTR_W_F16, ga, sa);
    (void)hipDeviceSynchronize();
    for (int zero = 0; zero < 2; ++zero) {
      long e5 = 0, waves = 0;
      for (int it = 0; it < iters; ++it) {
        reset_outputs(A, sa);
        (void)hipDeviceSynchronize();
        const float one = 1.f, nul = 0.f;
        for (int k = 0; k < 20; ++k)
          rocblas_gemm_ex(rh, rocblas_operation_none, rocblas_operation_none, G, G, G, &one, zero ? gz : ga_, rocblas_datatype_f16_r, G, zero ? gz : gb_,
                          rocblas_datatype_f16_r, G, &nul, gc_, rocblas_datatype_f16_r, G, gc_, rocblas_datatype_f16_r, G, rocblas_datatype_f32_r,
                          rocblas_gemm_algo_standard, 0, 0);
        splitk_epilogue_kernel<LNP, true><<<rgrid, 256, 0, sa>>>(A.splitk, 4, M, N, epA);
        (void)hipDeviceSynchronize();
        const long d = compare(zero ? "E5 reduce + library fp16 GEMM on zeros" : "E5 reduce + library fp16 GEMM", it, A, refA, snap(A), e5 ? 0 : 6);
        e5 += d != 0; waves += d;
      }
      printf("E5 (%s operands): %ld of %d runs differ (%ld values)\n", zero ? "zero" : "random", e5, iters, waves);
      e5_total += e5;
    }
  }
#endif
  printf("rln_fault: E1b %ld E1c %ld E2 %ld E3 %ld E4 %ld E5 %ld of %d runs differ\n", e1b, e1c, e2, e3, e4, e5_total, iters);
  return (bad || e1b || e1c || e2 || e3 || e4 || e5_total) ? 1 : 0;
}
