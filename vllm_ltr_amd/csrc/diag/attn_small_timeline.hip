// Diagnostic: phase stamps of the attention workgroups of a small pass (library built with -DLTR_ATTN_TIMELINE -DLTR_GEMM_TIMELINE:
// diag/run_small_timeline.sh).  One request of L tokens, OPT-125m geometry (12 heads of 64).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../ltr_internal.h"
namespace ltr { int attn_timeline_read(unsigned long long* host); }
using namespace ltr;
int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 262, H = 768, heads = 12;
  __half *qkv, *out; int32_t *cu, *blk;
  (void)hipMalloc(&qkv, (size_t)L * 3 * H * 4); (void)hipMalloc(&out, (size_t)L * H * 4);
  (void)hipMemset(qkv, 0x11, (size_t)L * 3 * H * 4);
  int32_t cu_h[2] = {0, L};
  (void)hipMalloc(&cu, 8); (void)hipMemcpy(cu, cu_h, 8, hipMemcpyHostToDevice);
  (void)hipMalloc(&blk, (1 + 4) * 4 + (L / 64 + 2) * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  AOp in{qkv, qkv + (size_t)L * 3 * H}, o{out, out + (size_t)L * H};
  for (int it = 0; it < 5; ++it) launch_attention(LTR_W_F16, in, cu, 1, L, H, heads, blk, o, it == 0, 0);
  (void)hipEventRecord(e0, 0);
  for (int it = 0; it < 20; ++it) launch_attention(LTR_W_F16, in, cu, 1, L, H, heads, blk, o, 0, 0);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> tl(4096 * 5);
  attn_timeline_read(tl.data());
  const int nblk = (L + 127) / 128, gx = L / 128 + 1;
  printf("attention, one %d-token request: %.2f us per launch in the stream; per query block (head 0 / head 11): cycles entry->descriptor, ->first tile, tile loop, epilogue\n", L, ms / 20 * 1e3);
  for (int b = 0; b < nblk; ++b)
    for (int h : {0, 11}) {
      const unsigned long long* t = &tl[(h * gx + b) * 5];
      printf("  block %d head %2d: %6llu %6llu %7llu %6llu   (life %llu)\n", b, h, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[4] - t[0]);
    }
  return 0;
}
