// Diagnostic: where a small-batch GEMM launch spends its time (needs the library built with -DLTR_GEMM_TIMELINE:
// diag/run_small_timeline.sh).  Per workgroup four cycle stamps: entry, first K stage landed, K loop done, stores issued.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../ltr_internal.h"
namespace ltr { int gemm_timeline_read(unsigned long long* host); }
using namespace ltr;
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 262, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
  __half *ahi, *w, *ohi, *wp; float *bias, *of;
  (void)hipMalloc(&ahi, (size_t)M * K * 4); (void)hipMalloc(&w, (size_t)N * K * 2); (void)hipMalloc(&wp, (size_t)N * K * 2);
  (void)hipMalloc(&ohi, (size_t)M * N * 4); (void)hipMalloc(&of, (size_t)M * N * 4); (void)hipMalloc(&bias, N * 4);
  (void)hipMemset(ahi, 0x11, (size_t)M * K * 4); (void)hipMemset(w, 0x22, (size_t)N * K * 2); (void)hipMemset(bias, 0, N * 4);
  launch_pack_weight(w, wp, N, K, 0);
  GemmArgs g{}; g.a = AOp{ahi, ahi + (size_t)M * K}; g.w = wp; g.bias = bias; g.M = M; g.N = N; g.K = K; g.a_slab = 1;
  if (N > 1024) g.out_split = AOp{ohi, ohi + (size_t)M * N}; else g.out_f32 = of;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) launch_gemm(LTR_W_F16, g, 0);
  (void)hipEventRecord(e0, 0);
  for (int it = 0; it < 20; ++it) launch_gemm(LTR_W_F16, g, 0);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> tl(8192 * 4);
  gemm_timeline_read(tl.data());
  std::vector<double> pro, loop, epi;
  unsigned long long tmin = ~0ull, tmax = 0;
  int nb = 0;
  for (int b = 0; b < 8192; ++b) {
    if (!tl[b * 4 + 3] || !tl[b * 4 + 1]) continue;
    ++nb;
    tmin = std::min(tmin, tl[b * 4]); tmax = std::max(tmax, tl[b * 4 + 3]);
    pro.push_back((double)(tl[b * 4 + 1] - tl[b * 4])); loop.push_back((double)(tl[b * 4 + 2] - tl[b * 4 + 1]));
    epi.push_back((double)(tl[b * 4 + 3] - tl[b * 4 + 2]));
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  auto mx = [](std::vector<double> v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
  std::vector<double> start;
  for (int b = 0; b < 8192; ++b) if (tl[b * 4 + 3] && tl[b * 4 + 1]) start.push_back((double)(tl[b * 4] - tmin));
  printf("M=%d N=%d K=%d: %.2f us per launch in the stream; %d workgroups stamped; launch span %llu ticks (first entry -> last exit)\n",
         M, N, K, ms / 20 * 1e3, nb, tmax - tmin);
  printf("  ticks (s_memtime; 100 MHz = 10 ns per tick if constant-rate):  entry spread median %.0f max %.0f | entry -> first stage median %.0f max %.0f | "
         "K loop median %.0f max %.0f | epilogue median %.0f max %.0f\n", med(start), mx(start), med(pro), mx(pro), med(loop), mx(loop), med(epi), mx(epi));
  return 0;
}
