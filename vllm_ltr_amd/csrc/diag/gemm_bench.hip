// Diagnostic: times the four dense layers of an OPT decoder layer at a pass of M tokens through ltr::launch_gemm
// (links libltr_hip.so), random operands (the kernel is power-bound: zeros would flatter it), with the LayerNorm fold
// roles they have in the model: QKV = consumer, out_proj = producer, fc1 = consumer + ReLU + slab output, fc2 = producer.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I. -I../../include diag/gemm_bench.hip -L. -lltr_hip -o /tmp/gemm_bench
//   LD_LIBRARY_PATH=. /tmp/gemm_bench [M=196608] [H=768] [F=3072] [reps=20]
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
// ReLU-shaped operand (probe builds whose fc1 does not store: fc2 must still see what fc1 writes in the model - half the
// entries zero in BOTH planes - because MFMA power, and with it the clock, depends on the data)
__global__ void relu_planes(__half* hi, __half* lo, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (__half2float(hi[i]) < 0.f) { hi[i] = __float2half(0.f); lo[i] = __float2half(0.f); }
}
template <class T> T* alloc(size_t n) { T* p; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { printf("alloc failed\n"); exit(1); } return p; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 196608, H = argc > 2 ? atoi(argv[2]) : 768, F = argc > 3 ? atoi(argv[3]) : 3072;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  __half* a = alloc<__half>((size_t)M * H * 2);        // hi | lo planes, slab-major or row-major: layout irrelevant for timing
  __half* a2 = alloc<__half>((size_t)M * H * 2);
  __half* qkv = alloc<__half>((size_t)M * 3 * H * 2);
  __half* f = alloc<__half>((size_t)M * F * 2);
  float* h = alloc<float>((size_t)M * H);
  float2* st1 = alloc<float2>((size_t)(H / 64) * M);
  float2* st2 = alloc<float2>((size_t)(H / 64) * M);
  fill_half<<<2048, 256>>>(a, (size_t)M * H, 1.f, 1); fill_half<<<2048, 256>>>(a + (size_t)M * H, (size_t)M * H, 2e-4f, 2);
  fill_half<<<2048, 256>>>(f, (size_t)M * F, 1.f, 3); fill_half<<<2048, 256>>>(f + (size_t)M * F, (size_t)M * F, 2e-4f, 4);
  relu_planes<<<2048, 256>>>(f, f + (size_t)M * F, (size_t)M * F);
  fill_f32<<<2048, 256>>>(h, (size_t)M * H, 1.f, 0.f, 5);
  fill_f32<<<2048, 256>>>((float*)st1, (size_t)(H / 64) * M * 2, 0.1f, 1.f, 6);
  fill_f32<<<2048, 256>>>((float*)st2, (size_t)(H / 64) * M * 2, 0.1f, 1.f, 7);
  auto weight = [&](int N, int K, unsigned seed) {
    __half* w = alloc<__half>((size_t)N * K); __half* wp = alloc<__half>((size_t)N * K);
    fill_half<<<2048, 256>>>(w, (size_t)N * K, 0.05f, seed);
    launch_pack_weight(w, wp, N, K, 0);
    return wp;
  };
  __half *w_qkv = weight(3 * H, H, 11), *w_out = weight(H, H, 12), *w_fc1 = weight(F, H, 13), *w_fc2 = weight(H, F, 14);
  // BENCH_ROTATE=R: R copies of every weight matrix, launch r uses copy r % R - as the R layers of a forward do (the same
  // weights every launch stay in the XCDs' L2s: 1.2-4.7 MB each; a forward streams 85 MB of them)
  const int rot = getenv("BENCH_ROTATE") ? atoi(getenv("BENCH_ROTATE")) : 1;
  std::vector<__half*> r_qkv{w_qkv}, r_out{w_out}, r_fc1{w_fc1}, r_fc2{w_fc2};
  for (int i = 1; i < rot; ++i) {
    r_qkv.push_back(weight(3 * H, H, 11 + 10 * i)); r_out.push_back(weight(H, H, 12 + 10 * i));
    r_fc1.push_back(weight(F, H, 13 + 10 * i)); r_fc2.push_back(weight(H, F, 14 + 10 * i));
  }
  const size_t skb = (size_t)4 * (M < 4800 ? M : 4800) * H * 4;       // what forward_chunk provides
  void* sk = alloc<char>(skb);
  float* vec = alloc<float>(4 * (size_t)F);
  fill_f32<<<64, 256>>>(vec, 4 * (size_t)F, 0.1f, 1.f, 8);
  (void)hipDeviceSynchronize();
  const bool fold = !(getenv("BENCH_NO_FOLD") && getenv("BENCH_NO_FOLD")[0] == '1');
  GemmArgs g_qkv{}, g_out{}, g_fc1{}, g_fc2{};
  g_qkv.a = AOp{a, a + (size_t)M * H}; g_qkv.w = w_qkv; g_qkv.bias = vec; g_qkv.out_split = AOp{qkv, qkv + (size_t)M * 3 * H};
  g_qkv.M = M; g_qkv.N = 3 * H; g_qkv.K = H; g_qkv.a_slab = 1;
  if (fold) { g_qkv.ln_stats_in = st1; g_qkv.ln_c = vec + F; g_qkv.ln_parts = H / 64; }
  g_out.a = AOp{a, a + (size_t)M * H}; g_out.w = w_out; g_out.bias = vec; g_out.resid = h; g_out.out_f32 = h; g_out.M = M; g_out.N = H; g_out.K = H;
  if (fold) { g_out.ln_gamma = vec + 2 * F; g_out.ln_out = AOp{a2, a2 + (size_t)M * H}; g_out.ln_stats_out = st2; }
  { g_out.splitk_ws = sk; g_out.splitk_ws_bytes = skb; g_fc2.splitk_ws = sk; g_fc2.splitk_ws_bytes = skb; }
  g_fc1.a = AOp{a2, a2 + (size_t)M * H}; g_fc1.w = w_fc1; g_fc1.bias = vec; g_fc1.out_split = AOp{f, f + (size_t)M * F}; g_fc1.relu = 1;
  g_fc1.M = M; g_fc1.N = F; g_fc1.K = H; g_fc1.a_slab = g_fc1.out_slab = 1;
  if (fold) { g_fc1.ln_stats_in = st2; g_fc1.ln_c = vec + F; g_fc1.ln_parts = H / 64; }
  g_fc2.a = AOp{f, f + (size_t)M * F}; g_fc2.w = w_fc2; g_fc2.bias = vec; g_fc2.resid = h; g_fc2.out_f32 = h; g_fc2.M = M; g_fc2.N = H; g_fc2.K = F; g_fc2.a_slab = 1;
  if (fold) { g_fc2.ln_gamma = vec + 2 * F; g_fc2.ln_out = AOp{a, a + (size_t)M * H}; g_fc2.ln_stats_out = st1; }
  // fc1 overwrites f with relu(...) of random data; refill f's planes each rep would distort timing: fc2 then reads
  // whatever fc1 wrote (realistic values, half zeros - as in the model)
  struct Item { const char* name; GemmArgs* g; double flop; } items[4] = {
      {"qkv", &g_qkv, 2.0 * M * 3 * H * H}, {"out_proj", &g_out, 2.0 * M * H * H}, {"fc1", &g_fc1, 2.0 * M * F * H}, {"fc2", &g_fc2, 2.0 * M * H * F}};
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  if (const char* only = getenv("BENCH_ONLY")) {   // PMC passes: one shape only, `reps` launches (counters are summed per kernel name)
    for (auto& it : items)
      if (!strcmp(only, it.name)) {
        for (int r = 0; r < reps; ++r) launch_gemm(LTR_W_F16, *it.g, 0);
        (void)hipDeviceSynchronize();
        printf("%s x %d at M=%d\n", it.name, reps, M);
        return 0;
      }
    printf("BENCH_ONLY=%s: no such shape\n", only);
    return 1;
  }
  // the layer sequence, as in the model (keeps the chip in the model's power state), then each shape alone
  for (int w = 0; w < 3; ++w) for (auto& it : items) if (launch_gemm(LTR_W_F16, *it.g, 0)) { printf("launch failed: %s\n", it.name); return 1; }
  (void)hipDeviceSynchronize();
  // per shape: `reps` launches back to back in the stream (what the launch costs inside a forward: kernel + boundary;
  // an event pair around a single launch adds ~5 us of its own at this scale)
  std::vector<float> tot(4, 0.f);
  for (int i = 0; i < 4; ++i) {
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch_gemm(LTR_W_F16, *items[i].g, 0);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); tot[i] = ms;
  }
  double sum = 0, fl = 0;
  for (int i = 0; i < 4; ++i) {
    const double ms = tot[i] / reps;
    printf("%-9s %8.1f us  %7.1f TFLOP/s\n", items[i].name, ms * 1e3, items[i].flop / ms / 1e9);
    sum += ms; fl += items[i].flop;
  }
  printf("layer     %8.1f us  %7.1f TFLOP/s   (M=%d H=%d F=%d fold=%d)\n", sum * 1e3, fl / sum / 1e9, M, H, F, (int)fold);
  {  // the four launches back to back, as the model issues them (launch gaps included)
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) {
      g_qkv.w = r_qkv[r % rot]; g_out.w = r_out[r % rot]; g_fc1.w = r_fc1[r % rot]; g_fc2.w = r_fc2[r % rot];
      for (auto& it : items) launch_gemm(LTR_W_F16, *it.g, 0);
    }
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("sequence  %8.1f us per layer (4 GEMMs back to back)\n", ms / reps * 1e3);
  }
  return 0;
}
