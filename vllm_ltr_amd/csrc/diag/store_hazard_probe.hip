// Diagnostic: does MI355X need wait states between a > 64-bit vector store and a VALU write of the store's DATA registers?
// (LLVM keeps two behind the stores it emits for gfx940-class targets; a store inside an asm statement gets none - the epilogue
// stores of ltr_gemm.hip, rounds 2-6.)  Each lane stores the pattern A = {0xA0, 0xA1, 0xA2, 0xA3} | lane id with ONE
// global_store_dwordx4 and overwrites data register `R` with B = 0xBAD00000 after `WS` wait states; the host counts the B words
// that reached memory.  Variants: R = first / last data register, WS = 0 / 1 / 2, nt / default policy, with and without other
// memory traffic queued in front of the store (a burst of loads: the store then waits in the memory pipeline while the VALU runs on).
//   hipcc --offload-arch=gfx950 -O3 diag/store_hazard_probe.hip -o build/store_hazard_probe && build/store_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R, int WS, bool NT, bool BUSY>
__global__ void __launch_bounds__(256) probe(unsigned* out, const unsigned* junk, unsigned* sink) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  unsigned* p = out + (size_t)gid * 4;
  unsigned acc = 0;
  if (BUSY) {
#pragma unroll
    for (int i = 0; i < 12; ++i) acc ^= junk[(size_t)((gid * 9973u + i * 524287u) & ((1u << 24) - 1))];   // loads in flight in front of the store
  }
  const unsigned a0 = 0xA0000000u | gid, a1 = 0xA1000000u | gid, a2 = 0xA2000000u | gid, a3 = 0xA3000000u | gid;
#define BODY(STORE, NOPS, REG)                                                                                   \
  asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\ts_nop 4\n\t"  \
               STORE NOPS "v_mov_b32 " REG ", 0xBAD00000\n\ts_nop 4" ::"v"(p), "v"(a0), "v"(a1), "v"(a2), "v"(a3) \
               : "v10", "v11", "v12", "v13", "memory")
#define ST_NT "global_store_dwordx4 %0, v[10:13], off nt\n\t"
#define ST_PL "global_store_dwordx4 %0, v[10:13], off\n\t"
#define ST_AD "v_lshl_add_u64 v[14:15], %0, 0, 0\n\tglobal_store_dwordx4 v[14:15], v[10:13], off nt\n\t"   /* address from the VALU one instruction earlier */
#define N0 ""
#define N1 "s_nop 0\n\t"
#define N2 "s_nop 1\n\t"
  if (R == 2) {          // the shape the shipped epilogues had: address computed by the VALU right in front of the store
    if (WS == 0) { asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\ts_nop 4\n\t" ST_AD
                                "v_mov_b32 v10, 0xBAD00000\n\tv_mov_b32 v11, 0xBAD00000\n\tv_mov_b32 v12, 0xBAD00000\n\tv_mov_b32 v13, 0xBAD00000\n\ts_nop 4"
                                ::"v"(p), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "v10", "v11", "v12", "v13", "v14", "v15", "memory"); }
  } else if (NT) {
    if (R == 0) { if (WS == 0) BODY(ST_NT, N0, "v10"); else if (WS == 1) BODY(ST_NT, N1, "v10"); else BODY(ST_NT, N2, "v10"); }
    else        { if (WS == 0) BODY(ST_NT, N0, "v13"); else if (WS == 1) BODY(ST_NT, N1, "v13"); else BODY(ST_NT, N2, "v13"); }
  } else {
    if (R == 0) { if (WS == 0) BODY(ST_PL, N0, "v10"); else if (WS == 1) BODY(ST_PL, N1, "v10"); else BODY(ST_PL, N2, "v10"); }
    else        { if (WS == 0) BODY(ST_PL, N0, "v13"); else if (WS == 1) BODY(ST_PL, N1, "v13"); else BODY(ST_PL, N2, "v13"); }
  }
  if (BUSY && acc == 0x12345u) sink[0] = acc;
}

template <int R, int WS, bool NT, bool BUSY>
static long run(unsigned* d_out, const unsigned* junk, unsigned* sink, std::vector<unsigned>& h, int blocks) {
  long bad = 0;
  for (int rep = 0; rep < 20; ++rep) {
    (void)hipMemset(d_out, 0, h.size() * 4);
    probe<R, WS, NT, BUSY><<<blocks, 256>>>(d_out, junk, sink);
    (void)hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size(); ++i) {
      const unsigned want = (0xA0000000u + ((unsigned)(i & 3) << 24)) | (unsigned)(i >> 2);
      if (h[i] != want) ++bad;
    }
  }
  return bad;
}

int main() {
  const int blocks = 256 * 16;
  std::vector<unsigned> h((size_t)blocks * 256 * 4);
  unsigned *d_out, *junk, *sink;
  (void)hipMalloc(&d_out, h.size() * 4); (void)hipMalloc(&junk, (size_t)(1u << 24) * 4); (void)hipMalloc(&sink, 64);
  (void)hipMemset(junk, 1, (size_t)(1u << 24) * 4);
  long total = 0;
#define RUN(R, WS, NT, BUSY) { long b = run<R, WS, NT, BUSY>(d_out, junk, sink, h, blocks); total += (WS < 2) ? 0 : b; \
    printf("overwrite data reg %s after %d wait state(s), %s store, %s: %ld wrong words of %zu x 20\n", R ? "3 (last)" : "0 (first)", WS, \
           NT ? "nt" : "default", BUSY ? "loads queued in front" : "idle memory pipe", b, h.size()); }
  RUN(0, 0, true, false) RUN(1, 0, true, false) RUN(0, 1, true, false) RUN(1, 1, true, false) RUN(0, 2, true, false) RUN(1, 2, true, false)
  RUN(0, 0, true, true)  RUN(1, 0, true, true)  RUN(0, 1, true, true)  RUN(1, 1, true, true)  RUN(0, 2, true, true)  RUN(1, 2, true, true)
  RUN(1, 0, false, false) RUN(1, 0, false, true) RUN(1, 1, false, true) RUN(1, 2, false, true)
  { long b = run<2, 0, true, false>(d_out, junk, sink, h, blocks); long c = run<2, 0, true, true>(d_out, junk, sink, h, blocks);
    printf("all four data regs overwritten right behind an nt store whose ADDRESS the VALU computed one instruction earlier: %ld (idle) / %ld (loads queued) wrong words\n", b, c); }
  printf("wrong words with two wait states: %ld\n", total);
  return total ? 1 : 0;
}
