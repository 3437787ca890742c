// Diagnostic: do the GEMMs of two independent half batches overlap when they are issued onto two streams?  The four dense
// layers of a decoder layer (roles as in diag/gemm_bench.hip) at M rows on ONE stream, against two sets of M / 2 rows on TWO
// streams (launches issued alternately), same weights.  Usage: lanes_probe [M=5928] [H=768] [F=3072] [reps=40]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_fp16.h>
#include "ltr_internal.h"
using namespace ltr;

__global__ void fill_half(__half* p, size_t n, float scale, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = __float2half(((float)(x & 0xffff) / 32768.f - 1.f) * scale);
  }
}
__global__ void fill_f32(float* p, size_t n, float scale, float off, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((float)(x & 0xffff) / 32768.f - 1.f) * scale + off;
  }
}
template <class T> T* alloc(size_t n) { T* p; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { printf("alloc failed\n"); exit(1); } return p; }

struct Layer { GemmArgs g[4]; };
static __half *w_qkv, *w_out, *w_fc1, *w_fc2;
static float* vec;

Layer make(int M, int H, int F, unsigned seed) {
  __half* a = alloc<__half>((size_t)M * H * 2);
  __half* a2 = alloc<__half>((size_t)M * H * 2);
  __half* qkv = alloc<__half>((size_t)M * 3 * H * 2);
  __half* f = alloc<__half>((size_t)M * F * 2);
  float* h = alloc<float>((size_t)M * H);
  float2* st1 = alloc<float2>((size_t)(H / 64) * M);
  float2* st2 = alloc<float2>((size_t)(H / 64) * M);
  fill_half<<<1024, 256>>>(a, (size_t)M * H, 1.f, seed + 1); fill_half<<<1024, 256>>>(a + (size_t)M * H, (size_t)M * H, 2e-4f, seed + 2);
  fill_half<<<1024, 256>>>(a2, (size_t)M * H, 1.f, seed + 3); fill_half<<<1024, 256>>>(a2 + (size_t)M * H, (size_t)M * H, 2e-4f, seed + 4);
  fill_half<<<1024, 256>>>(f, (size_t)M * F, 1.f, seed + 5); fill_half<<<1024, 256>>>(f + (size_t)M * F, (size_t)M * F, 2e-4f, seed + 6);
  fill_f32<<<1024, 256>>>(h, (size_t)M * H, 1.f, 0.f, seed + 7);
  fill_f32<<<1024, 256>>>((float*)st1, (size_t)(H / 64) * M * 2, 0.1f, 1.f, seed + 8);
  fill_f32<<<1024, 256>>>((float*)st2, (size_t)(H / 64) * M * 2, 0.1f, 1.f, seed + 9);
  const size_t skb = (size_t)4 * (M < 4800 ? M : 4800) * H * 4;
  void* sk = alloc<char>(skb);
  Layer L{};
  GemmArgs &q = L.g[0], &o = L.g[1], &f1 = L.g[2], &f2 = L.g[3];
  q.a = AOp{a, a + (size_t)M * H}; q.w = w_qkv; q.bias = vec; q.out_split = AOp{qkv, qkv + (size_t)M * 3 * H};
  q.M = M; q.N = 3 * H; q.K = H; q.a_slab = 1; q.ln_stats_in = st1; q.ln_c = vec + F; q.ln_parts = H / 64;
  o.a = AOp{a, a + (size_t)M * H}; o.w = w_out; o.bias = vec; o.resid = h; o.out_f32 = h; o.M = M; o.N = H; o.K = H;
  o.ln_gamma = vec + 2 * F; o.ln_out = AOp{a2, a2 + (size_t)M * H}; o.ln_stats_out = st2; o.splitk_ws = sk; o.splitk_ws_bytes = skb;
  f1.a = AOp{a2, a2 + (size_t)M * H}; f1.w = w_fc1; f1.bias = vec; f1.out_split = AOp{f, f + (size_t)M * F}; f1.relu = 1;
  f1.M = M; f1.N = F; f1.K = H; f1.a_slab = f1.out_slab = 1; f1.ln_stats_in = st2; f1.ln_c = vec + F; f1.ln_parts = H / 64;
  f2.a = AOp{f, f + (size_t)M * F}; f2.w = w_fc2; f2.bias = vec; f2.resid = h; f2.out_f32 = h; f2.M = M; f2.N = H; f2.K = F; f2.a_slab = 1;
  f2.ln_gamma = vec + 2 * F; f2.ln_out = AOp{a, a + (size_t)M * H}; f2.ln_stats_out = st1; f2.splitk_ws = sk; f2.splitk_ws_bytes = skb;
  return L;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 5928, H = argc > 2 ? atoi(argv[2]) : 768, F = argc > 3 ? atoi(argv[3]) : 3072;
  const int reps = argc > 4 ? atoi(argv[4]) : 40;
  auto weight = [&](int N, int K, unsigned seed) {
    __half* w = alloc<__half>((size_t)N * K); __half* wp = alloc<__half>((size_t)N * K);
    fill_half<<<1024, 256>>>(w, (size_t)N * K, 0.05f, seed);
    launch_pack_weight(w, wp, N, K, 0);
    return wp;
  };
  w_qkv = weight(3 * H, H, 11); w_out = weight(H, H, 12); w_fc1 = weight(F, H, 13); w_fc2 = weight(H, F, 14);
  vec = alloc<float>(4 * (size_t)F);
  fill_f32<<<64, 256>>>(vec, 4 * (size_t)F, 0.1f, 1.f, 8);
  Layer whole = make(M, H, F, 100), ha = make(M / 2, H, F, 200), hb = make(M - M / 2, H, F, 300);
  (void)hipDeviceSynchronize();
  hipStream_t s1, s2;
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e0, e1, ej; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&ej);
  auto timed = [&](const char* what, auto&& body) {
    for (int w = 0; w < 3; ++w) body();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, s1);
    (void)hipStreamWaitEvent(s2, e0, 0);
    for (int r = 0; r < reps; ++r) body();
    (void)hipEventRecord(ej, s2); (void)hipStreamWaitEvent(s1, ej, 0);
    (void)hipEventRecord(e1, s1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %8.1f us per layer\n", what, ms / reps * 1e3);
  };
  timed("whole batch, one stream", [&] { for (auto& g : whole.g) launch_gemm(LTR_W_F16, g, s1); });
  timed("half A alone, one stream", [&] { for (auto& g : ha.g) launch_gemm(LTR_W_F16, g, s1); });
  timed("halves A then B, one stream", [&] { for (auto& g : ha.g) launch_gemm(LTR_W_F16, g, s1); for (auto& g : hb.g) launch_gemm(LTR_W_F16, g, s1); });
  timed("halves A | B, two streams (alternate issue)", [&] { for (int i = 0; i < 4; ++i) { launch_gemm(LTR_W_F16, ha.g[i], s1); launch_gemm(LTR_W_F16, hb.g[i], s2); } });
  timed("halves A | B, two streams, B one GEMM behind", [&] {
    launch_gemm(LTR_W_F16, ha.g[0], s1);
    for (int i = 1; i < 4; ++i) { launch_gemm(LTR_W_F16, ha.g[i], s1); launch_gemm(LTR_W_F16, hb.g[i - 1], s2); }
    launch_gemm(LTR_W_F16, hb.g[3], s2); });
  {   // three and four lanes
    hipStream_t sx[4] = {s1, s2, nullptr, nullptr};
    (void)hipStreamCreateWithFlags(&sx[2], hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sx[3], hipStreamNonBlocking);
    hipEvent_t ef, ejx[4]; (void)hipEventCreate(&ef); for (auto& e : ejx) (void)hipEventCreate(&e);
    for (int nl = 3; nl <= 4; ++nl) {
      std::vector<Layer> parts;
      for (int i = 0; i < nl; ++i) parts.push_back(make(M / nl, H, F, 400 + 50 * i + nl));
      (void)hipDeviceSynchronize();
      auto body = [&] { for (int i = 0; i < 4; ++i) for (int l = 0; l < nl; ++l) launch_gemm(LTR_W_F16, parts[l].g[i], sx[l]); };
      for (int w = 0; w < 3; ++w) body();
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0, s1);
      for (int l = 1; l < nl; ++l) (void)hipStreamWaitEvent(sx[l], e0, 0);
      for (int r = 0; r < reps; ++r) body();
      for (int l = 1; l < nl; ++l) { (void)hipEventRecord(ejx[l], sx[l]); (void)hipStreamWaitEvent(s1, ejx[l], 0); }
      (void)hipEventRecord(e1, s1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%d parts on %d streams                           %8.1f us per layer\n", nl, nl, ms / reps * 1e3);
    }
  }
  for (int i = 0; i < 4; ++i) {
    const char* names[4] = {"qkv", "out_proj", "fc1", "fc2"};
    char buf[96];
    snprintf(buf, sizeof buf, "  %s: whole, one stream", names[i]);
    timed(buf, [&] { launch_gemm(LTR_W_F16, whole.g[i], s1); });
    snprintf(buf, sizeof buf, "  %s: halves on two streams", names[i]);
    timed(buf, [&] { launch_gemm(LTR_W_F16, ha.g[i], s1); launch_gemm(LTR_W_F16, hb.g[i], s2); });
  }
  return 0;
}
