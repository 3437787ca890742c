// Diagnostic: per-workgroup timeline of one QKV-shaped GEMM launch (needs the library built with
// -DLTR_GEMM_TIMELINE).  Prints, per CU, how the main loops and epilogues of its workgroups interleave.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include "../ltr_internal.h"
namespace ltr { int gemm_timeline_read(unsigned long long* host); int gemm_waits_read(unsigned long long* host); }
using namespace ltr;
int main() {
  const int M = 65536, N = 2304, K = 768;
  __half *ahi, *alo, *w, *ohi; float* bias;
  (void)hipMalloc(&ahi, (size_t)M * K * 2); (void)hipMalloc(&alo, (size_t)M * K * 2); (void)hipMalloc(&w, (size_t)N * K * 2);
  (void)hipMalloc(&ohi, (size_t)M * N * 4); (void)hipMalloc(&bias, N * 4);
  (void)hipMemset(ahi, 0x11, (size_t)M * K * 2); (void)hipMemset(alo, 0x01, (size_t)M * K * 2); (void)hipMemset(w, 0x22, (size_t)N * K * 2);
  (void)hipMemset(bias, 0, N * 4);
  __half* wp; (void)hipMalloc(&wp, (size_t)N * K * 2); launch_pack_weight(w, wp, N, K, 0);
  GemmArgs g{}; g.a = AOp{ahi, alo}; g.w = wp; g.bias = bias; g.out_split = AOp{ohi, (char*)ohi + (size_t)M * N * 2}; g.M = M; g.N = N; g.K = K;
  for (int it = 0; it < 3; ++it) launch_gemm(LTR_W_F16, g, 0);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> tl(8192 * 4);
  gemm_timeline_read(tl.data());
  const int nb = (M / 128) * (N / 256);
  unsigned long long t0 = ~0ull;
  for (int b = 0; b < nb && b < 8192; ++b) t0 = std::min(t0, tl[b * 4]);
  std::map<unsigned long long, std::vector<int>> by_cu;
  double main_sum = 0, epi_sum = 0;
  for (int b = 0; b < nb && b < 8192; ++b) {
    unsigned long long id = tl[b * 4 + 3];
    unsigned hw = (unsigned)id; unsigned xcc = (unsigned)(id >> 32);
    unsigned cu = (hw >> 8) & 0xf, se = (hw >> 13) & 0x7, sh = (hw >> 12) & 1;   // gfx9 HW_ID: CU_ID[11:8], SH_ID[12], SE_ID[15:13]
    by_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
    main_sum += tl[b * 4 + 1] - tl[b * 4]; epi_sum += tl[b * 4 + 2] - tl[b * 4 + 1];
  }
  {
    std::vector<unsigned long long> wt(8192 * 2);
    gemm_waits_read(wt.data());
    double d = 0, b = 0;
    for (int i = 0; i < std::min(nb, 8192); ++i) { d += wt[i * 2]; b += wt[i * 2 + 1]; }
    printf("wave0 per block: dma wait %.0f cyc, barrier wait %.0f cyc\n", d / std::min(nb, 8192), b / std::min(nb, 8192));
  }
  printf("blocks %d  CUs seen %zu  avg main %.0f cyc  avg epilogue %.0f cyc\n", nb, by_cu.size(), main_sum / std::min(nb, 8192), epi_sum / std::min(nb, 8192));
  int shown = 0;
  for (auto& kv : by_cu) {
    if (shown++ >= 3) break;
    auto v = kv.second;
    std::sort(v.begin(), v.end(), [&](int a, int b) { return tl[a * 4] < tl[b * 4]; });
    printf("CU %llx: %zu blocks\n", kv.first, v.size());
    for (size_t i = 0; i < v.size() && i < 14; ++i) {
      int b = v[i];
      printf("   blk %5d start %8llu  main_end %8llu  end %8llu\n", b, tl[b * 4] - t0, tl[b * 4 + 1] - t0, tl[b * 4 + 2] - t0);
    }
  }
  // chip-wide: fraction of time with >X% of resident blocks in epilogue
  unsigned long long tend = 0;
  for (int b = 0; b < nb && b < 8192; ++b) tend = std::max(tend, tl[b * 4 + 2]);
  const int NB = 60;
  std::vector<int> in_epi(NB, 0), in_main(NB, 0);
  for (int b = 0; b < nb && b < 8192; ++b)
    for (int k = 0; k < NB; ++k) {
      unsigned long long t = t0 + (tend - t0) * (2 * k + 1) / (2 * NB);
      if (t >= tl[b * 4] && t < tl[b * 4 + 1]) in_main[k]++;
      else if (t >= tl[b * 4 + 1] && t < tl[b * 4 + 2]) in_epi[k]++;
    }
  printf("time-slice: blocks in main loop / in epilogue (launch = %llu cycles)\n", tend - t0);
  for (int k = 0; k < NB; ++k) printf("%d/%d ", in_main[k], in_epi[k]);
  printf("\n");
  return 0;
}
