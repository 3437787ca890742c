// Diagnostic: attention kernels alone vs an fp64 host reference (links libltr_hip.so).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../ltr_internal.h"
using namespace ltr;
static float h2f(__half h) { return __half2float(h); }
int main() {
  const int H = 64, nh = 1;
  std::vector<int> lens = {1, 2, 33, 64, 65, 200, 129};
  std::vector<int32_t> cu(1, 0);
  for (int L : lens) cu.push_back(cu.back() + L);
  const int T = cu.back(), n = lens.size();
  std::vector<float> qkv((size_t)T * 3 * H);
  srand(1);
  auto rnd = []() { double u = 0; for (int i = 0; i < 12; ++i) u += rand() / (double)RAND_MAX; return u - 6.0; };
  for (int t = 0; t < T; ++t) for (int c = 0; c < 3 * H; ++c) qkv[(size_t)t * 3 * H + c] = (float)(rnd() * (c < 2 * H ? 1.66 : 0.55));
  std::vector<__half> hi(qkv.size()), lo(qkv.size());
  for (size_t i = 0; i < qkv.size(); ++i) { hi[i] = __float2half_rn(qkv[i]); lo[i] = __float2half_rn(qkv[i] - h2f(hi[i])); }
  // reference from the SPLIT values (what the kernel is given)
  std::vector<double> ref((size_t)T * H);
  for (int r = 0; r < n; ++r) for (int i = 0; i < lens[r]; ++i) {
    int ti = cu[r] + i; std::vector<double> s(i + 1); double mx = -1e300;
    for (int j = 0; j <= i; ++j) { int tj = cu[r] + j; double a = 0; for (int d = 0; d < H; ++d) { double q = (double)h2f(hi[(size_t)ti*3*H+d]) + h2f(lo[(size_t)ti*3*H+d]); double k = (double)h2f(hi[(size_t)tj*3*H+H+d]) + h2f(lo[(size_t)tj*3*H+H+d]); a += q * k; } s[j] = a * 0.125; mx = fmax(mx, s[j]); }
    double den = 0; for (int j = 0; j <= i; ++j) { s[j] = exp(s[j] - mx); den += s[j]; }
    for (int d = 0; d < H; ++d) { double o = 0; for (int j = 0; j <= i; ++j) { int tj = cu[r] + j; o += s[j] * ((double)h2f(hi[(size_t)tj*3*H+2*H+d]) + h2f(lo[(size_t)tj*3*H+2*H+d])); } ref[(size_t)ti * H + d] = o / den; }
  }
  float* d_qkv; __half *d_hi, *d_lo, *d_ohi, *d_olo; float* d_of; int32_t *d_cu, *d_blk;
  (void)hipMalloc(&d_qkv, qkv.size() * 4); (void)hipMalloc(&d_hi, qkv.size() * 2); (void)hipMalloc(&d_lo, qkv.size() * 2);
  (void)hipMalloc(&d_ohi, (size_t)T * H * 2); (void)hipMalloc(&d_olo, (size_t)T * H * 2); (void)hipMalloc(&d_of, (size_t)T * H * 4);
  (void)hipMalloc(&d_cu, (n + 1) * 4); (void)hipMalloc(&d_blk, (n + 2) * 4 + (T / 32 + n + 1) * 8);
  (void)hipMemcpy(d_qkv, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_hi, hi.data(), qkv.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(d_lo, lo.data(), qkv.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_cu, cu.data(), (n + 1) * 4, hipMemcpyHostToDevice);
  int rc = launch_attention(LTR_W_F16, AOp{d_hi, d_lo}, d_cu, n, T, H, nh, d_blk, AOp{d_ohi, d_olo}, 0);
  rc |= launch_attention(LTR_W_F32, AOp{d_qkv, nullptr}, d_cu, n, T, H, nh, d_blk, AOp{d_of, nullptr}, 0);
  (void)hipDeviceSynchronize();
  printf("rc=%d\n", rc);
  std::vector<__half> ohi((size_t)T * H), olo((size_t)T * H); std::vector<float> of((size_t)T * H);
  (void)hipMemcpy(ohi.data(), d_ohi, ohi.size() * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(olo.data(), d_olo, olo.size() * 2, hipMemcpyDeviceToHost);
  (void)hipMemcpy(of.data(), d_of, of.size() * 4, hipMemcpyDeviceToHost);
  for (int r = 0; r < n; ++r) {
    double e16 = 0, e32 = 0; int w16 = -1, wd = -1;
    for (int i = 0; i < lens[r]; ++i) for (int d = 0; d < H; ++d) {
      size_t o = (size_t)(cu[r] + i) * H + d;
      double a = fabs((double)h2f(ohi[o]) + h2f(olo[o]) - ref[o]);
      if (a > e16) { e16 = a; w16 = i; wd = d; }
      e32 = fmax(e32, fabs(of[o] - ref[o]));
    }
    printf("req %d L=%3d: mfma-split max err %.3e (pos %d d %d)   f32-valu max err %.3e\n", r, lens[r], e16, w16, wd, e32);
  }
  for (int r : {4, 5}) {
    printf("req %d per-position max err (x1e-7):", r);
    for (int i = 0; i < lens[r]; ++i) { double e = 0; for (int d = 0; d < H; ++d) { size_t o = (size_t)(cu[r] + i) * H + d; e = fmax(e, fabs((double)h2f(ohi[o]) + h2f(olo[o]) - ref[o])); } if (i % 32 == 0) printf("\n  %3d:", i); printf(" %4.0f", e * 1e7); }
    printf("\n");
  }
  { int ti = cu[4] + 48; for (int d : {41, 42, 43}) { size_t o = (size_t)ti * H + d; printf("VAL d=%d ref=%.10g hi=%.10g lo=%.10g sum=%.10g f32kernel=%.10g\n", d, ref[o], h2f(ohi[o]), h2f(olo[o]), (double)h2f(ohi[o]) + h2f(olo[o]), of[o]); }
    ti = cu[5] + 191; { int d = 60; size_t o = (size_t)ti * H + d; printf("VAL d=%d ref=%.10g hi=%.10g lo=%.10g sum=%.10g f32kernel=%.10g\n", d, ref[o], h2f(ohi[o]), h2f(olo[o]), (double)h2f(ohi[o]) + h2f(olo[o]), of[o]); } }
  // dump one row
  int ti = cu[2] + 1;
  printf("row req2 pos1: "); for (int d = 0; d < 8; ++d) printf("[%g %g] ", (double)h2f(ohi[(size_t)ti*H+d]) + h2f(olo[(size_t)ti*H+d]), ref[(size_t)ti*H+d]); printf("\n");
  return 0;
}
