// Diagnostic lab for the split-fp16 GEMM main loop (perf + correctness, standalone).
// Operands in K-blocked layout: element (row, k) of an [R x K] matrix lives at
//   ((k >> 4) * R + row) * 16 + (k & 15)            (halves)
// so every k16-block of a tile is one contiguous run of 32-B rows -> 1-KiB LDS-DMA instructions read
// 1 KiB of contiguous memory (full 128-B lines), and the ring can be staged at k16 granularity.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int NXCD = 8, GM = 8;
static int g_pad = 0;
#ifndef PROBE_DMA_ONCE
#define PROBE_DMA_ONCE 0
#endif
#ifndef MFMA_ORDER
#define MFMA_ORDER 0
#endif
#ifndef LO_MASK
#define LO_MASK 0xffff
#endif
#ifndef PROBE_NOBAR
#define PROBE_NOBAR 0
#endif
#ifndef PROBE_REGS
#define PROBE_REGS 0
#endif
#ifndef PROBE_ZERO
#define PROBE_ZERO 0
#endif
#ifndef PROBE_NOMFMA
#define PROBE_NOMFMA 0
#endif
#ifndef PROBE_M16
#define PROBE_M16 0
#endif
#ifndef PROBE_NOEPI
#define PROBE_NOEPI 0
#endif   // extra rows in the leading dimension of K-blocked activation buffers

__device__ __forceinline__ void tile_of_block(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  const int q = nwg / NXCD, r = nwg % NXCD;
  const int xcd = bid % NXCD, k = bid / NXCD;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  const int per_group = GM * tiles_n;
  const int g = lin / per_group, in_g = lin - g * per_group;
  const int gm = min(GM, tiles_m - g * GM);
  tm = g * GM + in_g % gm;
  tn = in_g / gm;
}

__device__ __forceinline__ void split_f16(float a, __half& hi, __half& lo) {
  asm volatile("" : "+v"(a));
  hi = __float2half(a);
  lo = __float2half(a - __half2float(hi));
}

struct Epi {
  const float* bias;
  const float* resid;   // f32 row-major [M][N] or null
  float* out_f32;       // f32 row-major or null
  __half* out_hi;       // K-blocked [N/16][M][16] or null
  __half* out_lo;
  int relu;
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N < 0, "add case");
}

// KS = k16-blocks per stage (1 or 2), ST = ring stages; lookahead ST-1 slabs.
template <int KS, int ST, int WPS, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN, WPS) lab_kernel(const __half* __restrict__ a_hi, const __half* __restrict__ a_lo,
                                                       const __half* __restrict__ w, int M, int N, int K, int tiles_m,
                                                       int tiles_n, Epi ep, int ldm) {
  constexpr int BM = 64 * WM, BN = 64 * WN, NWAVE = WM * WN;
  constexpr int A_SUB = BM * 16;                 // halves per A plane per k16-block (4 KiB)
  constexpr int W_SUB = BN * 16;                 // 8 KiB
  constexpr int SUB = 2 * A_SUB + W_SUB;         // a_hi | a_lo | w of one k16-block (16 KiB)
  constexpr int STAGE = KS * SUB;
  constexpr int PER_Q = 2 * BM / 32 + BN / 32;   // 1-KiB DMA pieces per k16-block
  static_assert((KS * PER_Q) % NWAVE == 0, "pieces must divide evenly");
  constexpr int PIECES = KS * PER_Q / NWAVE;     // DMA instructions per wave per slab
  extern __shared__ __attribute__((aligned(16))) __half smem[];

  int tm, tn;
  tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;

  // DMA pieces of a slab: KS x PER_Q instructions of 1 KiB (32 rows x 32 B); wave v owns pieces
  // v*PIECES .. v*PIECES+PIECES-1 of the list [q][a_hi rowgroups | a_lo rowgroups | w rowgroups]
  const int prow = lane >> 1;
  const __half* gp[PIECES];
  int lp[PIECES];
  size_t kstride[PIECES];
  int qof[PIECES];
  const size_t a_kb = (size_t)ldm * 16, w_kb = (size_t)N * 16;   // halves per k16-block of the whole matrix
#pragma unroll
  for (int p = 0; p < PIECES; ++p) {
    const int P = wave * PIECES + p;
    const int q = P / PER_Q, idx = P % PER_Q;
    qof[p] = q;
    if (idx < 2 * BM / 32) {
      const int plane = idx / (BM / 32), rg = idx % (BM / 32);
      const int r = rg * 32 + prow;
      const int clog = (lane & 1) ^ ((r >> 3) & 1);
      gp[p] = (plane ? a_lo : a_hi) + ((size_t)(m0 + r)) * 16 + clog * 8;
      lp[p] = q * SUB + plane * A_SUB + rg * 32 * 16;
      kstride[p] = a_kb;
    } else {
      const int rg = idx - 2 * BM / 32;
      const int r = rg * 32 + prow;
      const int clog = (lane & 1) ^ ((r >> 3) & 1);
      gp[p] = w + ((size_t)(n0 + r)) * 16 + clog * 8;
      lp[p] = q * SUB + 2 * A_SUB + rg * 32 * 16;
      kstride[p] = w_kb;
    }
  }
  auto issue = [&](int slab) {
    __half* base = smem + (slab % ST) * STAGE;
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      __builtin_amdgcn_global_load_lds((gbl_void*)(gp[p] + (size_t)(slab * KS + qof[p]) * kstride[p]), (lds_void*)(base + lp[p]), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fk = lane >> 5;
  const int fsw = (fk ^ ((frow >> 3) & 1)) * 8;   // swizzled chunk offset (halves) of this lane's fragment
  const int nslab = K / (16 * KS);
  constexpr int LA = ST - 1;
#pragma unroll
  for (int s = 0; s < LA; ++s)
    if (s < nslab) issue(s);
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc16[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 pah[2], pal[2], pbw[2];
  if (PROBE_REGS) {   // fragments of slab 0, loaded once and reused for every k-step (perf probe, wrong results)
    wait_vmcnt<0>();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      pah[i] = *reinterpret_cast<const f16x8*>(smem + (wr * 64 + i * 32 + frow) * 16 + fsw);
      pal[i] = *reinterpret_cast<const f16x8*>(smem + A_SUB + (wr * 64 + i * 32 + frow) * 16 + fsw);
      pbw[i] = *reinterpret_cast<const f16x8*>(smem + 2 * A_SUB + (wc * 64 + i * 32 + frow) * 16 + fsw);
      if (PROBE_ZERO) { pah[i] = pah[i] - pah[i]; pal[i] = pal[i] - pal[i]; pbw[i] = pbw[i] - pbw[i]; }
    }
  }
  for (int s = 0; s < nslab; ++s) {
    if (s + LA - 1 < nslab) wait_vmcnt<(LA - 1) * PIECES>(); else wait_vmcnt<0>();
    if (!PROBE_NOBAR) __builtin_amdgcn_s_barrier();   // raw: __syncthreads() would drain vmcnt (the look-ahead DMA) as well
    if (!PROBE_DMA_ONCE && s + LA < nslab) issue(s + LA);
    const __half* st = smem + (s % ST) * STAGE;
    if (PROBE_NOMFMA) continue;
    if (PROBE_M16) {   // power probe: same FLOPs per slab on v_mfma_f32_16x16x32_f16 (register operands, wrong results)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16((s & 1) ? pah[i & 1] : pal[i & 1], pbw[j & 1], acc16[i][j], 0, 0, 0);
      continue;
    }
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const __half* s_ahi = st + q * SUB;
      const __half* s_alo = s_ahi + A_SUB;
      const __half* s_w = s_ahi + 2 * A_SUB;
      f16x8 ah[2], al[2], bw[2];
      if (PROBE_REGS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { ah[i] = pah[i]; al[i] = pal[i]; bw[i] = pbw[i]; }
      } else
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wr * 64 + i * 32 + frow;
        ah[i] = *reinterpret_cast<const f16x8*>(s_ahi + row * 16 + fsw);
        al[i] = *reinterpret_cast<const f16x8*>(s_alo + row * 16 + fsw);
        const int col = wc * 64 + i * 32 + frow;
        bw[i] = *reinterpret_cast<const f16x8*>(s_w + col * 16 + fsw);
      }
      if (MFMA_ORDER == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bw[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bw[j], acc[i][j], 0, 0, 0);
          }
      } else if (MFMA_ORDER == 1) {   // A operand fixed over two MFMAs
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bw[j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bw[j], acc[i][j], 0, 0, 0);
        }
      } else {                        // all lo, then all hi
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bw[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bw[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  if (PROBE_NOEPI) {
    float t = 0.f;
    if (PROBE_M16) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc16[i][j][0] + acc16[i][j][1] + acc16[i][j][2] + acc16[i][j][3];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 12345.678f && ep.out_f32) ep.out_f32[tid] = t;
    return;
  }
  // epilogue: wave-local LDS strips [16 rows][64 cols] f32 (stride 68), as the production kernel
  constexpr int CLD = 68, CROWS = 16;
  float* s_c = reinterpret_cast<float*>(smem) + wave * CROWS * CLD;
  const int lq = lane & 31, lh = lane >> 5;
  const int erow = lane >> 3, ecol = (lane & 7) * 8;
  const int ccol = n0 + wc * 64 + ecol;
  float4 bias_a = make_float4(0.f, 0.f, 0.f, 0.f), bias_b = bias_a;
  if (ep.bias) {
    bias_a = *reinterpret_cast<const float4*>(ep.bias + ccol);
    bias_b = *reinterpret_cast<const float4*>(ep.bias + ccol + 4);
  }
  __syncthreads();
#pragma unroll
  for (int stp = 0; stp < 4; ++stp) {
    const int i = stp >> 1, half = stp & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 8; ++r)
        s_c[((r & 3) + 8 * (r >> 2) + 4 * lh) * CLD + j * 32 + lq] = acc[i][j][8 * half + r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float4 va[2], vb[2], ra[2], rb[2];
    int grow[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int srow = it * 8 + erow;
      grow[it] = m0 + wr * 64 + i * 32 + half * 16 + srow;
      va[it] = *reinterpret_cast<const float4*>(s_c + srow * CLD + ecol);
      vb[it] = *reinterpret_cast<const float4*>(s_c + srow * CLD + ecol + 4);
      ra[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      rb[it] = ra[it];
      if (ep.resid) {
        const size_t o = (size_t)grow[it] * N + ccol;
        ra[it] = *reinterpret_cast<const float4*>(ep.resid + o);
        rb[it] = *reinterpret_cast<const float4*>(ep.resid + o + 4);
      }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float x[8] = {va[it].x + bias_a.x, va[it].y + bias_a.y, va[it].z + bias_a.z, va[it].w + bias_a.w,
                    vb[it].x + bias_b.x, vb[it].y + bias_b.y, vb[it].z + bias_b.z, vb[it].w + bias_b.w};
      if (ep.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
      }
      x[0] += ra[it].x; x[1] += ra[it].y; x[2] += ra[it].z; x[3] += ra[it].w;
      x[4] += rb[it].x; x[5] += rb[it].y; x[6] += rb[it].z; x[7] += rb[it].w;
      if (ep.out_f32) {
        const size_t o = (size_t)grow[it] * N + ccol;
        *reinterpret_cast<float4*>(ep.out_f32 + o) = make_float4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<float4*>(ep.out_f32 + o + 4) = make_float4(x[4], x[5], x[6], x[7]);
      }
      if (ep.out_hi) {
        __half h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split_f16(x[e], h[e], l[e]);
        const size_t o = ((size_t)(ccol >> 4) * ldm + grow[it]) * 16 + (ccol & 15);
        *reinterpret_cast<uint4*>(ep.out_hi + o) = *reinterpret_cast<const uint4*>(h);
        *reinterpret_cast<uint4*>(ep.out_lo + o) = *reinterpret_cast<const uint4*>(l);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- helpers -----------------------------------------------------------------------------------
__global__ void fill_blocked(__half* hi, __half* lo, int R, int K, unsigned seed, float scale, int ldr) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)R * K) return;
  const int row = i / K, k = i % K;
  unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  const float v = ((x & 0xffff) / 65536.f - 0.5f) * scale;
  const size_t o = ((size_t)(k >> 4) * ldr + row) * 16 + (k & 15);
  __half h = __float2half(v);
  hi[o] = h;
  if (lo) { __half l = __float2half(v - __half2float(h)); unsigned short u = __half_as_ushort(l) & LO_MASK; lo[o] = __ushort_as_half(u); }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  p[i] = ((x & 0xffff) / 65536.f - 0.5f);
}
// reference for sampled rows: one thread per (sample row, n)
__global__ void ref_rows(const __half* hi, const __half* lo, const __half* w, const float* bias, const float* resid,
                         int M, int N, int K, int relu, const int* rows, int nrows, double* out, int ldm) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int ri = blockIdx.y;
  if (n >= N || ri >= nrows) return;
  const int m = rows[ri];
  double s = 0;
  for (int k = 0; k < K; ++k) {
    const size_t oa = ((size_t)(k >> 4) * ldm + m) * 16 + (k & 15), ow = ((size_t)(k >> 4) * N + n) * 16 + (k & 15);
    s += ((double)__half2float(hi[oa]) + (double)__half2float(lo[oa])) * (double)__half2float(w[ow]);
  }
  s += bias[n];
  if (relu) s = s > 0 ? s : 0;
  if (resid) s += resid[(size_t)m * N + n];
  out[(size_t)ri * N + n] = s;
}

struct Shape { const char* name; int N, K, split, relu, resid; double weight; };

template <int KS, int ST, int WPS, int WM, int WN>
static double run_cfg(const char* cfg, int M, const Shape& sh, __half* ahi, __half* alo, __half* w, float* bias, float* resid,
                      float* of32, __half* ohi, __half* olo, bool check) {
  const int N = sh.N, K = sh.K;
  constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
  Epi ep{bias, sh.resid ? resid : nullptr, sh.split ? nullptr : of32, sh.split ? ohi : nullptr, sh.split ? olo : nullptr, sh.relu};
  const int tiles_m = M / BM, tiles_n = N / BN;
  const size_t lds = std::max((size_t)ST * KS * (2 * BM * 16 + BN * 16) * 2, (size_t)WM * WN * 16 * 68 * 4);
  (void)hipFuncSetAttribute((const void*)lab_kernel<KS, ST, WPS, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) lab_kernel<KS, ST, WPS, WM, WN><<<tiles_m * tiles_n, NT, lds>>>(ahi, alo, w, M, N, K, tiles_m, tiles_n, ep, M + g_pad);
  const int reps = getenv("LAB_REPS") ? atoi(getenv("LAB_REPS")) : 10;
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) lab_kernel<KS, ST, WPS, WM, WN><<<tiles_m * tiles_n, NT, lds>>>(ahi, alo, w, M, N, K, tiles_m, tiles_n, ep, M + g_pad);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) { printf("launch error\n"); exit(1); }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double maxerr = -1;
  if (check) {
    const int nrows = 24;
    std::vector<int> rows(nrows);
    for (int i = 0; i < nrows; ++i) rows[i] = (int)(((long long)i * 7919 * 131 + 77 * i) % M);
    int* drows; double* dref;
    (void)hipMalloc(&drows, nrows * 4); (void)hipMalloc(&dref, (size_t)nrows * N * 8);
    (void)hipMemcpy(drows, rows.data(), nrows * 4, hipMemcpyHostToDevice);
    ref_rows<<<dim3((N + 255) / 256, nrows), 256>>>(ahi, alo, w, bias, sh.resid ? resid : nullptr, M, N, K, sh.relu, drows, nrows, dref, M + g_pad);
    std::vector<double> ref((size_t)nrows * N);
    (void)hipMemcpy(ref.data(), dref, ref.size() * 8, hipMemcpyDeviceToHost);
    maxerr = 0;
    std::vector<float> rowf(N); std::vector<__half> rh(16), rl(16);
    for (int i = 0; i < nrows; ++i) {
      if (!sh.split) {
        (void)hipMemcpy(rowf.data(), of32 + (size_t)rows[i] * N, N * 4, hipMemcpyDeviceToHost);
      } else {
        for (int cb = 0; cb < N / 16; ++cb) {
          (void)hipMemcpy(rh.data(), ohi + ((size_t)cb * (M + g_pad) + rows[i]) * 16, 32, hipMemcpyDeviceToHost);
          (void)hipMemcpy(rl.data(), olo + ((size_t)cb * (M + g_pad) + rows[i]) * 16, 32, hipMemcpyDeviceToHost);
          for (int e = 0; e < 16; ++e) rowf[cb * 16 + e] = __half2float(rh[e]) + __half2float(rl[e]);
        }
      }
      for (int n = 0; n < N; ++n) maxerr = std::max(maxerr, std::abs((double)rowf[n] - ref[(size_t)i * N + n]));
    }
    (void)hipFree(drows); (void)hipFree(dref);
  }
  const double tf = 2.0 * M * N * K / (ms * 1e-3) * 1e-12;
  printf("  %-22s %-8s N=%4d K=%4d : %7.1f us  %6.1f TF%s", cfg, sh.name, N, K, ms * 1e3, tf, check ? "" : "\n");
  if (check) printf("   max|err| %.2e\n", maxerr);
  return ms;
}

int main(int argc, char** argv) {
  const int M = 65536;
  const Shape shapes[4] = {{"qkv", 2304, 768, 1, 0, 0, 1}, {"out", 768, 768, 0, 0, 1, 1}, {"fc1", 3072, 768, 1, 1, 0, 1}, {"fc2", 768, 3072, 0, 0, 1, 1}};
  __half *ahi, *alo, *w, *ohi, *olo; float *bias, *resid, *of32;
  const size_t maxA = (size_t)(M + 4096) * 3072, maxW = (size_t)3072 * 3072;
  (void)hipMalloc(&ahi, maxA * 2); (void)hipMalloc(&alo, maxA * 2); (void)hipMalloc(&w, maxW * 2);
  (void)hipMalloc(&ohi, maxA * 2); (void)hipMalloc(&olo, maxA * 2);
  (void)hipMalloc(&bias, 3072 * 4); (void)hipMalloc(&resid, (size_t)M * 768 * 4); (void)hipMalloc(&of32, (size_t)M * 768 * 4);
  fill_f32<<<12, 256>>>(bias, 3072, 5u);
  fill_f32<<<(M * 768 + 255) / 256, 256>>>(resid, (size_t)M * 768, 9u);
  const bool check = argc > 1 && atoi(argv[1]) != 0;
  for (int pi = 2; pi < std::max(argc, 3); ++pi) {
  g_pad = argc > 2 ? atoi(argv[pi]) : 0;
  printf("pad rows %d\n", g_pad);
  // per-step scale: tokens 708,977 / 65,536 passes x 12 layers (last layer pruned: ignore)
  const double scale = 708977.0 / 65536.0 * 12.0;
  auto sweep = [&](auto runner, const char* cfg) {
    double tot = 0;
    for (const Shape& sh : shapes) {
      fill_blocked<<<(unsigned)(((size_t)M * sh.K + 255) / 256), 256>>>(ahi, alo, M, sh.K, 1u, 4.0f, M + g_pad);
      fill_blocked<<<(unsigned)(((size_t)sh.N * sh.K + 255) / 256), 256>>>(w, nullptr, sh.N, sh.K, 3u, 0.1f, sh.N);
      tot += runner(cfg, sh);
    }
    printf("%-22s => %.1f ms per bench step (4 GEMMs x 12 layers x 10.8 passes)\n", cfg, tot * scale);
  };
#define CFG(KS, ST, WPS, WM, WN) sweep([&](const char* c, const Shape& sh) { return run_cfg<KS, ST, WPS, WM, WN>(c, M, sh, ahi, alo, w, bias, resid, of32, ohi, olo, check); }, "KS" #KS " ST" #ST " wps" #WPS " " #WM "x" #WN)
  if (!getenv("LAB_ONE")) CFG(1, 3, 2, 2, 4);   // first sweep of a process runs at cold clocks: repeat
  CFG(1, 3, 2, 2, 4);
  if (!getenv("LAB_ONE")) { CFG(2, 2, 4, 4, 4); CFG(2, 3, 4, 4, 4); }
  }
  return 0;
}
