// Diagnostic (round 6): what does the LAUNCH CHAIN of a one-request scoring call cost with nothing in the kernels?
// The k = 1 call of OPT-125m is 84 dependent launches in ~650 us (profiles/r04_k1_timeline.txt): one-workgroup kernels 4.1-6.0 us,
// small GEMMs 7.4-10.5 us, while the guide prices a dependent kernel boundary at 1.2-1.9 us.  This replays the call's launch
// list - same grids, block sizes, dynamic-LDS sizes, kernarg bytes, same order - with EMPTY kernels, eager and as a replayed
// hipGraph, and then adds one attribute at a time:
//   trivial    84 x <<<1, 64>>>, no LDS, 8-byte kernarg
//   grids      the real grids / blocks, no LDS, 8-byte kernarg
//   +lds       ... with the real dynamic-LDS sizes (64 KiB rings: hipFuncSetAttribute'd)
//   +kernarg   ... with the real kernarg sizes (the GEMMs pass a 200-byte Epilogue by value: 264 bytes)
//   +touch     ... every workgroup also s_loads its kernargs and reads one cache line (a kernel that starts like the real ones)
//   +dirty     ... every workgroup leaves 4 KiB of nt stores behind (the write-back a real epilogue leaves for the kernel end)
// per-launch = total / 84.  hipcc --offload-arch=gfx950 -O3 diag/empty_chain.hip -o empty_chain && ./empty_chain [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BYTES> struct Blob { unsigned w[BYTES / 4]; };

template <int KARG, int MODE>    // MODE 0 empty, 1 touch, 2 touch + dirty
__global__ void chain_k(Blob<KARG> b, const float4* __restrict__ src, float4* __restrict__ dst) {
  extern __shared__ float sm[];
  if (MODE >= 1) {
    const float4 v = src[(blockIdx.x * 8 + (b.w[KARG / 4 - 1] & 7)) & 4095];
    if (v.x == 1.2345e-30f) dst[0] = v;
  }
  if (MODE >= 2) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* d = reinterpret_cast<f4*>(dst) + (size_t)blockIdx.x * 256 + threadIdx.x % 256;
    __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, (float)b.w[0]}, d);
  }
}

struct L { int grid, block, lds, karg; };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  // the k = 1 call (262 tokens, OPT-125m), profiles/r04_k1_timeline.txt: kernel, workgroups; blocks / LDS / kernarg from the launchers
  const int RING = 65536 + 256;          // SmallCfg<32, 64, 2, 2, 2, 4>::LDS_BYTES
  std::vector<L> call;
  call.push_back({17, 256, 0, 96});      // embed_gather
  call.push_back({33, 256, 0, 64});      // layernorm
  for (int l = 0; l < 12; ++l) {
    const bool last = l == 11;
    if (!last) {
      call.push_back({360, 256, RING, 264});                       // QKV
      if (l == 0) call.push_back({1, 256, 0, 32});                 // attn_blocks
      call.push_back({108, 256, 40960, 96});                       // attention
      call.push_back({120, 256, RING, 264});                       // out_proj
      call.push_back({432, 256, RING, 264});                       // fc1
      call.push_back({480, 256, RING, 264});                       // fc2 (4 K parts)
      call.push_back({99, 256, 0, 232});                           // K-part reduce
    } else {
      call.push_back({240, 256, RING, 264});                       // K | V
      call.push_back({1, 256, 0, 64});                             // gather_last_rows
      call.push_back({1, 256, 0, 64});                             // layernorm (1 row)
      call.push_back({16, 256, RING, 264});                        // Q
      call.push_back({12, 256, 0, 64});                            // attn_lastq
      call.push_back({16, 256, RING, 264});                        // out_proj
      call.push_back({48, 256, RING, 264});                        // fc1
      call.push_back({64, 256, RING, 264});                        // fc2
      call.push_back({1, 256, 0, 232});                            // reduce
      call.push_back({1, 256, 0, 96});                             // pool_head
    }
  }
  printf("launch list: %zu launches\n", call.size());
  float4 *src, *dst;
  (void)hipMalloc(&src, 4096 * 16); (void)hipMalloc(&dst, (size_t)512 * 256 * 16);
  (void)hipMemset(src, 0, 4096 * 16);
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
#define ATTR(K, M) (void)hipFuncSetAttribute((const void*)chain_k<K, M>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)
  ATTR(8, 0); ATTR(264, 0); ATTR(264, 1); ATTR(264, 2); ATTR(8, 1); ATTR(8, 2);
  // variant: 0 trivial, 1 grids, 2 +lds, 3 +kernarg, 4 +touch, 5 +dirty
  auto launch = [&](const L& l, int variant) {
    const int grid = variant == 0 ? 1 : l.grid, block = variant == 0 ? 64 : l.block, lds = variant >= 2 ? l.lds : 0;
    const bool big = variant >= 3 && l.karg > 128;
    if (variant <= 3) { if (big) chain_k<264, 0><<<grid, block, lds, s>>>(Blob<264>{}, src, dst); else chain_k<8, 0><<<grid, block, lds, s>>>(Blob<8>{}, src, dst); }
    else if (variant == 4) { if (big) chain_k<264, 1><<<grid, block, lds, s>>>(Blob<264>{}, src, dst); else chain_k<8, 1><<<grid, block, lds, s>>>(Blob<8>{}, src, dst); }
    else { if (big) chain_k<264, 2><<<grid, block, lds, s>>>(Blob<264>{}, src, dst); else chain_k<8, 2><<<grid, block, lds, s>>>(Blob<8>{}, src, dst); }
  };
  const char* names[] = {"trivial <<<1,64>>>", "real grids", "+ dynamic LDS", "+ 264-byte kernargs", "+ kernarg / line read", "+ 4 KiB nt stores per workgroup"};
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int variant = 0; variant < 6; ++variant) {
    // eager: GPU time of the chain between events, host far ahead (queue never empty) - median of reps
    std::vector<float> t;
    for (int w = 0; w < 3; ++w) for (auto& l : call) launch(l, variant);
    (void)hipStreamSynchronize(s);
    for (int r = 0; r < reps; ++r) {
      (void)hipEventRecord(e0, s);
      for (auto& l : call) launch(l, variant);
      (void)hipEventRecord(e1, s);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    const float eager = t[t.size() / 2];
    // back-to-back eager chains without a sync in between (the host stays ahead): total / reps
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) for (auto& l : call) launch(l, variant);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms_all; (void)hipEventElapsedTime(&ms_all, e0, e1);
    // graph replay
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (auto& l : call) launch(l, variant);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 3; ++w) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms_g; (void)hipEventElapsedTime(&ms_g, e0, e1);
    const int n = (int)call.size();
    printf("%-34s eager (one call between events) %7.1f us = %5.2f us per launch | eager, host ahead %7.1f us = %5.2f | graph replay %7.1f us = %5.2f\n",
           names[variant], eager, eager / n, ms_all * 1e3f / reps, ms_all * 1e3f / reps / n, ms_g * 1e3f / reps, ms_g * 1e3f / reps / n);
    fflush(stdout);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  }
  return 0;
}
