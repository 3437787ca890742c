// Diagnostic: achievable L2 -> CU streaming rate on gfx950 by access form (perf only).
//   mode 0: global_load_lds_dwordx4 (LDS DMA, 1 KiB per wave instruction)
//   mode 1: global_load_lds_dword   (LDS DMA, 256 B per wave instruction)
//   mode 2: global_load_dwordx4 into registers (1 KiB per wave instruction)
//   mode 3: mode 2 followed by ds_write_b128 (register staging)
//   mode 4: batches of DEPTH LDS-DMA loads + DEPTH register loads in flight together (does the register path add to what a CU
//           pulls, or do both queue behind the same limit?)      mode 5: batches of 2 x DEPTH LDS-DMA loads (its baseline)
// Every wave walks 1-KiB chunks of a buffer small enough to stay in L2 (per XCD), keeping DEPTH
// loads in flight.  Output: GB/s chip-wide and per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int MODE, int DEPTH>
__global__ void __launch_bounds__(512) rate_kernel(const char* __restrict__ buf, size_t buf_bytes, int iters, int stride_rows,
                                                   unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t gw = (size_t)blockIdx.x * nw + wave;
  char* my = lds + wave * DEPTH * 1024 * (MODE >= 4 ? 2 : 1);
  uint4 acc = make_uint4(0, 0, 0, 0);
  const size_t nchunk = buf_bytes / 1024;
  // stride_rows == 0: 1 KiB contiguous per instruction; otherwise 16 rows x 64 B with a row pitch
  // of stride_rows bytes (the row-major K-slab access of the GEMM)
  auto src = [&](int it) -> const char* {
    const size_t c = (gw * 977 + (size_t)it * 131) & (nchunk - 1);   // buf_bytes is a power of two
    if (stride_rows == 0) return buf + c * 1024 + lane * 16;
    const size_t base = (c * 1024) & (buf_bytes / 2 - 1);
    return buf + (base & ~(size_t)63) + (size_t)(lane >> 2) * stride_rows + (lane & 3) * 16;
  };
  if (MODE == 4 || MODE == 5) {
    for (int it = 0; it < iters; it += 2 * DEPTH) {
      uint4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) __builtin_amdgcn_global_load_lds((gbl_void*)src(it + d), (lds_void*)(my + d * 1024), 16, 0, 0);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (MODE == 4) v[d] = *reinterpret_cast<const uint4*>(src(it + DEPTH + d));
        else __builtin_amdgcn_global_load_lds((gbl_void*)src(it + DEPTH + d), (lds_void*)(my + (DEPTH + d) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (MODE == 4) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
      }
    }
  } else if (MODE >= 2) {
    for (int it = 0; it < iters; it += DEPTH) {
      uint4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const uint4*>(src(it + d));
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (MODE == 3) *reinterpret_cast<uint4*>(my + d * 1024 + lane * 16) = v[d];
        else { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
      }
    }
  } else
  for (int it = 0; it < iters; ++it) {
    const char* p = src(it);
    char* dst = my + (it % DEPTH) * 1024;
    if (MODE == 0) {
      __builtin_amdgcn_global_load_lds((gbl_void*)p, (lds_void*)dst, 16, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((gbl_void*)(p - lane * 16 + q * 256 + lane * 4), (lds_void*)(dst + q * 256), 4, 0, 0);
    } else {
      uint4 v = *reinterpret_cast<const uint4*>(p);
      if (MODE == 3) *reinterpret_cast<uint4*>(dst + lane * 16) = v;
      else { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    if (MODE == 0) { if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); }
    if (MODE == 1) { if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE >= 2 || sink == nullptr) {
    unsigned t = acc.x ^ acc.y ^ acc.z ^ acc.w ^ (unsigned)lds[threadIdx.x];
    if (t == 0x12345u && sink) sink[0] = t;
  }
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* buf, size_t buf_bytes, int blocks_per_cu, int waves, int stride_rows, unsigned* sink) {
  const int iters = 2000;
  const int grid = 256 * blocks_per_cu;
  const size_t lds = (size_t)waves * DEPTH * 1024 * (MODE >= 4 ? 2 : 1);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  rate_kernel<MODE, DEPTH><<<grid, waves * 64, lds>>>(buf, buf_bytes, 200, stride_rows, sink);
  (void)hipEventRecord(e0);
  rate_kernel<MODE, DEPTH><<<grid, waves * 64, lds>>>(buf, buf_bytes, iters, stride_rows, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * waves * iters * 1024.0;
  printf("%-10s depth %2d  wg/CU %d  waves/wg %2d  pitch %5d  buf %4zu MiB : %7.0f GB/s chip  %6.1f GB/s per CU\n", name, DEPTH,
         blocks_per_cu, waves, stride_rows, buf_bytes >> 20, bytes / ms * 1e-6, bytes / ms * 1e-6 / 256);
}

int main(int argc, char** argv) {
  const size_t max_bytes = (size_t)1 << 30;
  char* buf; unsigned* sink;
  (void)hipMalloc(&buf, max_bytes); (void)hipMemset(buf, 1, max_bytes); (void)hipMalloc(&sink, 64);
  if (argc > 1) {      // dma_rate mix: the mixed-path question only, L2-resident / Infinity-Cache-resident / HBM buffers
    for (size_t mb : {2, 64, 1024}) {
      const size_t bb = mb << 20;
      for (int bpc : {1, 2}) {
        run<5, 4>("dma+dma", buf, bb, bpc, 4, 0, sink);
        run<4, 4>("dma+reg", buf, bb, bpc, 4, 0, sink);
        run<5, 6>("dma+dma", buf, bb, bpc, 4, 0, sink);
        run<4, 6>("dma+reg", buf, bb, bpc, 4, 0, sink);
        run<2, 8>("reg128", buf, bb, bpc, 4, 0, sink);
        run<2, 12>("reg128", buf, bb, bpc, 4, 0, sink);
      }
    }
    return 0;
  }
  for (size_t mb : {1, 2, 16}) {
    const size_t bb = mb << 20;
    for (int pitch : {0, 1536}) {
      for (int waves : {4, 8}) {
        for (int bpc : {1, 2}) {
          run<0, 1>("dma128", buf, bb, bpc, waves, pitch, sink);
          run<0, 4>("dma128", buf, bb, bpc, waves, pitch, sink);
          run<2, 4>("reg128", buf, bb, bpc, waves, pitch, sink);
        }
      }
      run<0, 8>("dma128", buf, bb, 2, 8, pitch, sink);
      run<2, 8>("reg128", buf, bb, 2, 8, pitch, sink);
      run<3, 8>("reg+dsw", buf, bb, 2, 8, pitch, sink);
      run<1, 8>("dma32", buf, bb, 2, 8, pitch, sink);
      run<0, 16>("dma128", buf, bb, 2, 8, pitch, sink);
    }
  }
  return 0;
}
