// Diagnostic: the F16 GEMM's LayerNorm-fold producer (LNP) and consumer (LNC) epilogues against a host
// computation, element by element (links libltr_hip.so, calls ltr::launch_gemm directly).
// Built by vllm_ltr_amd/csrc/build.py into csrc/build/gemm_check and run by tests/test_gpu_gemm_epilogue.py
// (NaN-safe comparisons: a stale-register store can leave NaN bit patterns behind).
//   gemm_check M N K [row0 rows]     row0 / rows: run both GEMMs on the ROW WINDOW [row0, row0 + rows) of the M-row tensors
//   (GemmArgs::row0 / ldm) - rows outside the window must come back untouched.  The producer gets the split-K scratch the
//   scoring path provides, so launch_gemm's own choice (tile configuration, K parts) is what is checked; LTR_GEMM_FORCE_CFG /
//   LTR_GEMM_FORCE_SPLIT / LTR_GEMM_TAIL in the environment select the other paths.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include <hip/hip_fp16.h>
#include "ltr_internal.h"
using namespace ltr;

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
template <class T> T* dev(const std::vector<T>& v) { T* p; (void)hipMalloc(&p, v.size() * sizeof(T)); (void)hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return p; }
template <class T> std::vector<T> host(const T* p, size_t n) { std::vector<T> v(n); (void)hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost); return v; }

// The host reference is a scalar triple loop; its rows are independent: row ranges on up to 32 host threads
// (GEMM_CHECK_THREADS overrides).  body(m_begin, m_end, counters of this thread); the counters are summed afterwards, the
// "first few" printouts are per thread.
struct Bad { int out = 0, ln = 0, st = 0, c = 0; };
static Bad over_rows(int M, const std::function<void(int, int, Bad&)>& body) {
  const char* e = getenv("GEMM_CHECK_THREADS");
  int nt = e ? atoi(e) : (int)std::thread::hardware_concurrency();
  nt = nt < 1 ? 1 : (nt > 32 ? 32 : nt);
  if (nt > M) nt = M;
  std::vector<Bad> bad(nt);
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) {
    const int m0 = (int)((long long)M * t / nt), m1 = (int)((long long)M * (t + 1) / nt);
    th.emplace_back([&, t, m0, m1] { body(m0, m1, bad[t]); });
  }
  for (auto& x : th) x.join();
  Bad s;
  for (auto& b : bad) { s.out += b.out; s.ln += b.ln; s.st += b.st; s.c += b.c; }
  return s;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 200, N = argc > 2 ? atoi(argv[2]) : 128, K = argc > 3 ? atoi(argv[3]) : 128;
  const int row0 = argc > 5 ? atoi(argv[4]) : 0, rows = argc > 5 ? atoi(argv[5]) : M;
  if (row0 < 0 || rows <= 0 || row0 + rows > M) { printf("bad window\n"); return 2; }
  auto inside = [&](int m) { return m >= row0 && m < row0 + rows; };
  srand(1);
  std::vector<float> a((size_t)M * K), resid((size_t)M * N), bias(N), gamma(N);
  std::vector<__half> ahi(a.size()), alo(a.size()), w((size_t)N * K);
  for (size_t i = 0; i < a.size(); ++i) { a[i] = frand(); ahi[i] = __float2half(a[i]); alo[i] = __float2half(a[i] - __half2float(ahi[i])); }
  for (auto& x : w) x = __float2half(frand() * 0.1f);
  for (auto& x : resid) x = frand();
  for (auto& x : bias) x = frand();
  for (auto& x : gamma) x = 1.f + 0.5f * frand();
  __half *d_ahi = dev(ahi), *d_alo = dev(alo), *d_w = dev(w), *d_wp;
  (void)hipMalloc(&d_wp, w.size() * 2);
  launch_pack_weight(d_w, d_wp, N, K, 0);
  float *d_res = dev(resid), *d_bias = dev(bias), *d_gamma = dev(gamma), *d_out;
  (void)hipMalloc(&d_out, (size_t)M * N * 4);
  (void)hipMemcpy(d_out, d_res, (size_t)M * N * 4, hipMemcpyDeviceToDevice);
  __half* d_ln; (void)hipMalloc(&d_ln, (size_t)M * N * 4); (void)hipMemset(d_ln, 0xff, (size_t)M * N * 4);
  float2* d_stats; (void)hipMalloc(&d_stats, (size_t)(N / 64) * M * 8); (void)hipMemset(d_stats, 0xff, (size_t)(N / 64) * M * 8);
  GemmArgs g{};
  g.a = AOp{d_ahi, d_alo}; g.w = d_wp; g.bias = d_bias; g.resid = d_out; g.out_f32 = d_out; g.M = rows; g.N = N; g.K = K;
  g.row0 = row0; g.ldm = M;
  g.ln_gamma = d_gamma; g.ln_out = AOp{d_ln, d_ln + (size_t)M * N}; g.ln_stats_out = d_stats;
  const size_t skb = (size_t)4 * (rows < 4800 ? rows : 4800) * N * 4;      // what forward_chunk hands to the residual producers
  void* d_sk; (void)hipMalloc(&d_sk, skb); (void)hipMemset(d_sk, 0xff, skb);
  g.splitk_ws = d_sk; g.splitk_ws_bytes = skb;
  int rc = launch_gemm(LTR_W_F16, g, 0);
  (void)hipDeviceSynchronize();
  printf("LNP rc=%d\n", rc);
  auto out = host(d_out, (size_t)M * N);
  auto ln = host(d_ln, (size_t)M * N * 2);
  auto st = host(d_stats, (size_t)(N / 64) * M);
  std::vector<double> ref((size_t)M * N);
  const Bad bp = over_rows(M, [&](int mb, int me, Bad& bd) {
  int &bad_out = bd.out, &bad_ln = bd.ln, &bad_st = bd.st;
  for (int m = mb; m < me; ++m) {
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += ((double)__half2float(ahi[(size_t)m * K + k]) + (double)__half2float(alo[(size_t)m * K + k])) * (double)__half2float(w[(size_t)n * K + k]);
      s += bias[n] + resid[(size_t)m * N + n];
      ref[(size_t)m * N + n] = s;
      const size_t so = ((size_t)(n >> 5) * M + m) * 32 + (n & 31);
      if (!inside(m)) {     // outside the window: the residual row and the 0xff fill must still be there
        unsigned short raw; memcpy(&raw, &ln[so], 2);
        if (out[(size_t)m * N + n] != resid[(size_t)m * N + n] || raw != 0xffff) { if (bad_out++ < 5) printf("row %d outside the window was written\n", m); }
        continue;
      }
      if (!(fabs(out[(size_t)m * N + n] - s) <= 1e-4)) { if (bad_out++ < 5) printf("out[%d,%d] = %g want %g\n", m, n, out[(size_t)m * N + n], s); }
      const double got = (double)__half2float(ln[so]) + (double)__half2float(ln[(size_t)M * N + so]);
      const double want = s * gamma[n] * 16.0;
      if (!(fabs(got - want) <= 1e-3 * (1 + fabs(want)))) { if (bad_ln++ < 12 || (bad_ln % 97) == 0) printf("a'[%d,%d] = %g (hi %g lo %g) want %g\n", m, n, got, __half2float(ln[so]), __half2float(ln[(size_t)M * N + so]), want); }
    }
    for (int p = 0; p < N / 64 && inside(m); ++p) {
      double mu = 0, q = 0;
      for (int c = 0; c < 64; ++c) mu += ref[(size_t)m * N + p * 64 + c];
      mu /= 64;
      for (int c = 0; c < 64; ++c) { double d = ref[(size_t)m * N + p * 64 + c] - mu; q += d * d; }
      const float2 s2 = st[(size_t)p * M + m];
      if (!(fabs(s2.x - mu) <= 1e-4) || !(fabs(s2.y - q) <= 1e-3 * (1 + q))) { if (bad_st++ < 12) printf("stats[piece %d, row %d] = (%g, %g) want (%g, %g)\n", p, m, s2.x, s2.y, mu, q); }
    }
  }
  });
  const int bad_out = bp.out, bad_ln = bp.ln, bad_st = bp.st;
  printf("LNP: bad out %d, bad a' %d, bad stats %d\n", bad_out, bad_ln, bad_st);

  // consumer: A = a' planes (slab-major), stats; W2 [N2, N]; expected = LN(ref) W2^T + b2 with LN affine (gamma, beta)
  const int N2 = 256;
  std::vector<__half> w2((size_t)N2 * N);
  std::vector<float> beta(N), b2(N2);
  for (auto& x : w2) x = __float2half(frand() * 0.1f);
  for (auto& x : beta) x = 0.3f * frand();
  for (auto& x : b2) x = frand();
  __half *d_w2 = dev(w2), *d_w2p; (void)hipMalloc(&d_w2p, w2.size() * 2);
  launch_pack_weight(d_w2, d_w2p, N2, N, 0);
  float *d_beta = dev(beta), *d_b2 = dev(b2), *d_c, *d_d;
  (void)hipMalloc(&d_c, N2 * 4); (void)hipMalloc(&d_d, N2 * 4);
  launch_ln_fold_coeff(d_w2, d_gamma, d_beta, d_b2, N2, N, d_c, d_d, 0);
  __half* d_o2; (void)hipMalloc(&d_o2, (size_t)M * N2 * 4);
  GemmArgs c{};
  (void)hipMemset(d_o2, 0xff, (size_t)M * N2 * 4);
  c.a = g.ln_out; c.w = d_w2p; c.bias = d_d; c.out_split = AOp{d_o2, d_o2 + (size_t)M * N2}; c.M = rows; c.N = N2; c.K = N;
  c.row0 = row0; c.ldm = M;
  c.a_slab = 1; c.ln_stats_in = d_stats; c.ln_c = d_c; c.ln_parts = N / 64;
  rc = launch_gemm(LTR_W_F16, c, 0);
  (void)hipDeviceSynchronize();
  printf("LNC rc=%d\n", rc);
  auto o2 = host(d_o2, (size_t)M * N2 * 2);
  const Bad bc = over_rows(M, [&](int mb, int me, Bad& bd) {
  int& bad_c = bd.c;
  for (int m = mb; m < me; ++m) {
    if (!inside(m)) {
      for (int n2 = 0; n2 < N2; ++n2) {
        unsigned short raw; memcpy(&raw, &o2[(size_t)m * N2 + n2], 2);
        if (raw != 0xffff) { if (bad_c++ < 5) printf("lnc row %d outside the window was written\n", m); break; }
      }
      continue;
    }
    double mu = 0, var = 0;
    for (int n = 0; n < N; ++n) mu += ref[(size_t)m * N + n];
    mu /= N;
    for (int n = 0; n < N; ++n) { double d = ref[(size_t)m * N + n] - mu; var += d * d; }
    const double rstd = 1.0 / sqrt(var / N + 1e-5);
    for (int n2 = 0; n2 < N2; ++n2) {
      double s = b2[n2];
      for (int n = 0; n < N; ++n) s += ((ref[(size_t)m * N + n] - mu) * rstd * gamma[n] + beta[n]) * (double)__half2float(w2[(size_t)n2 * N + n]);
      const double got = (double)__half2float(o2[(size_t)m * N2 + n2]) + (double)__half2float(o2[(size_t)M * N2 + (size_t)m * N2 + n2]);
      if (!(fabs(got - s) <= 1e-3 * (1 + fabs(s)))) { if (bad_c++ < 12) printf("lnc[%d,%d] = %g want %g\n", m, n2, got, s); }
    }
  }
  });
  const int bad_c = bc.c;
  printf("LNC: bad %d of %d\n", bad_c, M * N2);
  return (bad_out || bad_ln || bad_st || bad_c) ? 1 : 0;
}
