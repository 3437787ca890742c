// Diagnostic: known-byte kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// forms the library uses (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own
// access pattern before trusting an absolute").  Every kernel touches each byte of a buffer far larger than
// the 256 MiB Infinity Cache exactly once, so algorithmic bytes == compulsory HBM bytes.
//   calib_dma_contig   global_load_lds_dwordx4, 1 KiB contiguous per wave instruction   (GEMM slab-major operands)
//   calib_dma_rows64   global_load_lds_dwordx4, 16 rows x 64 B per instruction          (row-major K-slab pieces)
//   calib_dma_rows128  global_load_lds_dwordx4, 8 rows x 128 B per instruction          (attention K/V tiles)
//   calib_reg_x4       global_load_dwordx4 into registers                               (LayerNorm / epilogue residual reads)
//   calib_reg_x2       global_load_dwordx2 into registers                               (embedding rows, fp16 x 4)
//   calib_store_x4     global_store_dwordx4                                             (f32 / fp16-plane epilogue stores)
//   calib_store_x2     global_store_dwordx2                                             (attention output pieces)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; profiles/summarize_calib.py divides.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr size_t BYTES = (size_t)4 << 30;      // 4 GiB per kernel
constexpr int THREADS = 256;                   // 4 waves
constexpr int PER_WAVE_ITERS = 64;             // 64 KiB per wave

// wave w of the grid owns the 64 KiB region [w * 65536, (w + 1) * 65536)
__global__ void __launch_bounds__(THREADS) calib_dma_contig(const char* __restrict__ buf, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const char* p = buf + gw * 65536 + lane * 16;
  for (int it = 0; it < PER_WAVE_ITERS; ++it)
    __builtin_amdgcn_global_load_lds((gbl_void*)(p + it * 1024), (lds_void*)(lds + wave * 4096 + (it & 3) * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 77 && sink) sink[0] = 1;
}

// 16 rows x 64 B: the 64 KiB region is seen as 16 rows of 4 KiB; instruction `it` reads the 64-B column piece it
__global__ void __launch_bounds__(THREADS) calib_dma_rows64(const char* __restrict__ buf, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const char* p = buf + gw * 65536 + (size_t)(lane >> 2) * 4096 + (lane & 3) * 16;
  for (int it = 0; it < PER_WAVE_ITERS; ++it)
    __builtin_amdgcn_global_load_lds((gbl_void*)(p + it * 64), (lds_void*)(lds + wave * 4096 + (it & 3) * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 77 && sink) sink[0] = 1;
}

// 8 rows x 128 B: 8 rows of 8 KiB
__global__ void __launch_bounds__(THREADS) calib_dma_rows128(const char* __restrict__ buf, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const char* p = buf + gw * 65536 + (size_t)(lane >> 3) * 8192 + (lane & 7) * 16;
  for (int it = 0; it < PER_WAVE_ITERS; ++it)
    __builtin_amdgcn_global_load_lds((gbl_void*)(p + it * 128), (lds_void*)(lds + wave * 4096 + (it & 3) * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 77 && sink) sink[0] = 1;
}

__global__ void __launch_bounds__(THREADS) calib_reg_x4(const char* __restrict__ buf, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const char* p = buf + gw * 65536 + lane * 16;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 8
  for (int it = 0; it < PER_WAVE_ITERS; ++it) {
    const uint4 v = *reinterpret_cast<const uint4*>(p + it * 1024);
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u && sink) sink[0] = 1;
}

__global__ void __launch_bounds__(THREADS) calib_reg_x2(const char* __restrict__ buf, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const char* p = buf + gw * 65536 + lane * 8;
  uint2 acc = make_uint2(0, 0);
#pragma unroll 8
  for (int it = 0; it < 2 * PER_WAVE_ITERS; ++it) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + it * 512);
    acc.x ^= v.x; acc.y ^= v.y;
  }
  if ((acc.x ^ acc.y) == 0x1234567u && sink) sink[0] = 1;
}

__global__ void __launch_bounds__(THREADS) calib_store_x4(char* __restrict__ buf) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  char* p = buf + gw * 65536 + lane * 16;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 3);
#pragma unroll 8
  for (int it = 0; it < PER_WAVE_ITERS; ++it) *reinterpret_cast<uint4*>(p + it * 1024) = v;
}

__global__ void __launch_bounds__(THREADS) calib_store_x2(char* __restrict__ buf) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  char* p = buf + gw * 65536 + lane * 8;
  const uint2 v = make_uint2(lane, blockIdx.x);
#pragma unroll 8
  for (int it = 0; it < 2 * PER_WAVE_ITERS; ++it) *reinterpret_cast<uint2*>(p + it * 512) = v;
}

int main() {
  char* buf; unsigned* sink;
  if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(buf, 1, BYTES);
  (void)hipDeviceSynchronize();
  const int grid = (int)(BYTES / 65536 / 4);
  calib_dma_contig<<<grid, THREADS, 16384>>>(buf, sink);
  calib_dma_rows64<<<grid, THREADS, 16384>>>(buf, sink);
  calib_dma_rows128<<<grid, THREADS, 16384>>>(buf, sink);
  calib_reg_x4<<<grid, THREADS>>>(buf, sink);
  calib_reg_x2<<<grid, THREADS>>>(buf, sink);
  calib_store_x4<<<grid, THREADS>>>(buf);
  calib_store_x2<<<grid, THREADS>>>(buf);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  printf("calib: %zu bytes per kernel\n", BYTES);
  return 0;
}
