// One optimisation step of the predictor's fine-tuning loop on gfx950.
//
// Reference: train/trainer.py:122-165 - fp32 master weights (set_default_torch_dtype(float32), :99-101),
// `outputs = predictor(input_ids, attention_mask)` (HF OPTForSequenceClassification, prefill_predictor.py:76-79),
// `loss_func(outputs.view(1, -1), labels)` with listMLE / MSELoss, or CrossEntropyLoss over num_labels classes
// (:125-157), `loss.backward()`, `torch.optim.Adam(lr, weight_decay).step()` (:122,161-165).
//
// Parameters, activations, gradients and the optimiser state are f32 (the reference computes the forward under autocast
// = fp16 on its GPU; this is at least that precise, and the CPU oracle - torch autograd over the scorer's arithmetic -
// is f32).  The dense layers C = A B^T run on the fp16 matrix cores with BOTH operands split into two fp16 terms
// (A = Ah + Al, B = Bh + Bl, ~22 significant bits each; products exact, f32 accumulation): [A | A] [Bh | Bl]^T over a
// doubled K is ONE launch of the scoring path's split-fp16 GEMM (ltr_gemm.hip) = four fp16 MFMA passes instead of the
// 16x slower exact-f32 MFMA (gemm_f32_kernel; LTR_TRAIN_F32=1, and shapes the fp16 kernel does not take, keep it).  The
// backward's dX = dY W and dW = dY^T X are brought to the C = A B^T form with explicit transposes.  Attention, LayerNorm, ReLU, losses, Adam: VALU kernels below.  Training batches are
// slates of tens of prompts (trainer.py --batch-size 64), so this file is written for clarity and exactness, not for
// the ranking path's throughput.
//
// Differences to the reference's recipe, on purpose:
//  * dropout: HF OPT applies dropout(0.1) after out_proj and fc2 in train mode; its masks come from torch's global
//    RNG and cannot be reproduced by another implementation.  ltr_train_config.dropout applies the same two dropouts
//    with a counter-based hash of (seed, step, layer, site, element); 0 (the default, and what the parity fixtures
//    use) disables it.
//  * every reduction runs in a fixed order, the embedding-table gradients included (duplicate token ids / positions are
//    summed in token order by one owner workgroup, embed_bwd_kernel): a step is bit-reproducible run to run.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <unordered_map>
#include <vector>

#include "ltr_internal.h"

namespace ltr {
namespace {

constexpr int D = 64;   // head size

// ------------------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------------------
// out[c * Rp + r] = in[r * C + c]; columns R..Rp-1 of every output row are zero (the GEMM's K must be a multiple of 16)
__global__ void __launch_bounds__(256) transpose_pad_kernel(const float* __restrict__ in, int R, int C, int Rp,
                                                            float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int k = 0; k < 32; k += 8) {
    const int r = r0 + ty + k, c = c0 + tx;
    tile[ty + k][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int k = 0; k < 32; k += 8) {
    const int c = c0 + ty + k, r = r0 + tx;
    if (c < C && r < Rp) out[(size_t)c * Rp + r] = tile[tx][ty + k];
  }
}

// max |x| of a tensor into *slot (non-negative floats order like their bit patterns; the slot is zeroed per step).
// At most 128 workgroups: every workgroup ends in one atomic on the same address, and those serialise at the L2 -
// 740 of them made this kernel 14.7 us for a 12 MB tensor that streams in 2.
constexpr int AMAX_BLOCKS = 128;
__global__ void __launch_bounds__(256) amax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ slot) {
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  auto mx = [](float m, const float4 v) { return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); };
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {       // four independent loads in flight per thread
    const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    m0 = mx(m0, a); m1 = mx(m1, b); m2 = mx(m2, c); m3 = mx(m3, d);
  }
  for (; i < n4; i += stride) m0 = mx(m0, x4[i]);
  float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[n4 * 4 + threadIdx.x]));
  m = wave_max(m);
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(m));
  }
}
// f32 B [N, K] -> the split-fp16 GEMM's slab-major weight image of [Bh | Bl] ([N, 2K] with hi(B) in the first K
// columns and lo(B) = fp16(B - hi) in the last K): element (n, k') at ((k' >> 5) * N + n) * 32 + (k' & 31).  One 16-byte
// piece (8 halves) per thread.
__global__ void __launch_bounds__(256) split_pack_kernel(const float* __restrict__ B, __half* __restrict__ dst, int N, int K,
                                                         const float* __restrict__ amax) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // piece index in the destination
  const size_t pieces = (size_t)N * K / 4;                          // 2K halves per row / 8
  if (i >= pieces) return;
  const int c = (int)(i & 3);
  const size_t rn = i >> 2;                                         // slab * N + n
  const int n = (int)(rn % N), slab = (int)(rn / N);
  const int nslab = K / 32;
  const bool lo = slab >= nslab;
  const float* src = B + (size_t)n * K + (lo ? slab - nslab : slab) * 32 + c * 8;
  const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
  const float sc = split_scale(*amax);
  const float x[8] = {v0.x * sc, v0.y * sc, v0.z * sc, v0.w * sc, v1.x * sc, v1.y * sc, v1.z * sc, v1.w * sc};
  __half o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { __half h, l; split_f16(x[e], h, l); o[e] = lo ? l : h; }
  *reinterpret_cast<uint4*>(dst + i * 8) = *reinterpret_cast<const uint4*>(o);
}

// f32 A [M, K] -> the split-fp16 GEMM's slab-major operand planes of [A | A] (K' = 2K): hi' / lo' [2K / 32][M][32], the
// K / 32 slabs of A twice.  One 16-byte piece (8 halves) per thread and plane copy.
__global__ void __launch_bounds__(256) split_operand_dup_kernel(const float* __restrict__ A, __half* __restrict__ hi,
                                                                __half* __restrict__ lo, int M, int K,
                                                                const float* __restrict__ amax) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // piece of the first copy: (slab * M + row) * 4 + chunk
  const size_t pieces = (size_t)M * K / 8;
  if (i >= pieces) return;
  const int c = (int)(i & 3);
  const size_t sr = i >> 2;
  const int row = (int)(sr % M), slab = (int)(sr / M);
  const float* src = A + (size_t)row * K + slab * 32 + c * 8;
  const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
  const float sc = split_scale(*amax);
  const float x[8] = {v0.x * sc, v0.y * sc, v0.z * sc, v0.w * sc, v1.x * sc, v1.y * sc, v1.z * sc, v1.w * sc};
  __half h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) split_f16(x[e], h[e], l[e]);
  const size_t second = (size_t)M * K;                              // halves per copy
  *reinterpret_cast<uint4*>(hi + i * 8) = *reinterpret_cast<const uint4*>(h);
  *reinterpret_cast<uint4*>(hi + second + i * 8) = *reinterpret_cast<const uint4*>(h);
  *reinterpret_cast<uint4*>(lo + i * 8) = *reinterpret_cast<const uint4*>(l);
  *reinterpret_cast<uint4*>(lo + second + i * 8) = *reinterpret_cast<const uint4*>(l);
}

// The same two operand forms from a TRANSPOSED source: S [R, C] row-major f32, operand row c = column c of S, contraction
// index = row of S, zero-padded to Rp (the weight-gradient GEMMs contract over the tokens: dW = dY^T X; the data-gradient
// GEMMs need W^T).  Replaces a transpose_pad launch + a second pass over the transposed copy.  One workgroup: 32 rows
// (one K-slab) x 64 columns of S through LDS; one 16-byte piece (operand row, 8 contraction elements) per thread.
//   ROLE_A: planes hi / lo of [A | A]  (dst = hi plane, dst2 = lo plane, each [2 Rp / 32][C][32])
//   else  : image of [Bh | Bl]         (dst [2 Rp / 32][C][32]: hi in slab s, lo in slab s + Rp / 32)
template <bool ROLE_A>
__global__ void __launch_bounds__(256) split_T_kernel(const float* __restrict__ S, __half* __restrict__ dst,
                                                      __half* __restrict__ dst2, int R, int C, int Rp,
                                                      const float* __restrict__ amax) {
  __shared__ float tile[32][65];
  const int c0 = blockIdx.x * 64, slab = blockIdx.y, r0 = slab * 32;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int k = 0; k < 32; k += 16) {
    const int r = r0 + ty + k;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R) v = *reinterpret_cast<const float4*>(S + (size_t)r * C + c0 + tx * 4);
    tile[ty + k][tx * 4 + 0] = v.x; tile[ty + k][tx * 4 + 1] = v.y; tile[ty + k][tx * 4 + 2] = v.z; tile[ty + k][tx * 4 + 3] = v.w;
  }
  __syncthreads();
  const int crow = threadIdx.x >> 2, kc = threadIdx.x & 3;
  const float sc = split_scale(*amax);
  __half h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) split_f16(tile[kc * 8 + e][crow] * sc, h[e], l[e]);
  const size_t piece = ((size_t)slab * C + c0 + crow) * 32 + kc * 8;       // halves
  const size_t second = (size_t)(Rp / 32) * C * 32;
  if (ROLE_A) {
    *reinterpret_cast<uint4*>(dst + piece) = *reinterpret_cast<const uint4*>(h);
    *reinterpret_cast<uint4*>(dst + second + piece) = *reinterpret_cast<const uint4*>(h);
    *reinterpret_cast<uint4*>(dst2 + piece) = *reinterpret_cast<const uint4*>(l);
    *reinterpret_cast<uint4*>(dst2 + second + piece) = *reinterpret_cast<const uint4*>(l);
  } else {
    *reinterpret_cast<uint4*>(dst + piece) = *reinterpret_cast<const uint4*>(h);
    *reinterpret_cast<uint4*>(dst + second + piece) = *reinterpret_cast<const uint4*>(l);
  }
}

// hi | lo fp16 planes -> f32 (the MFMA attention's output for the saved activations)
__global__ void __launch_bounds__(256) planes_to_f32_kernel(const __half* __restrict__ hi, const __half* __restrict__ lo,
                                                            float* __restrict__ out, size_t n8) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const uint4 a = *reinterpret_cast<const uint4*>(hi + i * 8), b = *reinterpret_cast<const uint4*>(lo + i * 8);
  const __half* ah = reinterpret_cast<const __half*>(&a);
  const __half* bl = reinterpret_cast<const __half*>(&b);
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = __half2float(ah[e]) + __half2float(bl[e]);
  *reinterpret_cast<float4*>(out + i * 8) = *reinterpret_cast<const float4*>(o);
  *reinterpret_cast<float4*>(out + i * 8 + 4) = *reinterpret_cast<const float4*>(o + 4);
}

// out = (ReLU)((sum_p partial[p]) / (s_A s_B) + bias) + resid: the parts of a split-K GEMM added up in part order
// (deterministic), unscaled, then what the GEMM epilogue would have applied.  4 consecutive columns per thread.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ partial, int parts, size_t n4, int N,
                                                            const float* __restrict__ amax_a, const float* __restrict__ amax_b,
                                                            const float* __restrict__ bias, const float* resid, int relu,
                                                            float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float inv = 1.f / (split_scale(*amax_a) * split_scale(*amax_b));
  const float4* p4 = reinterpret_cast<const float4*>(partial);
  float4 acc = p4[i];
  for (int p = 1; p < parts; ++p) {
    const float4 v = p4[(size_t)p * n4 + i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
  if (bias) {
    const float4 b = *reinterpret_cast<const float4*>(bias + (i * 4) % N);
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
  }
  if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
  if (resid) {
    const float4 r = reinterpret_cast<const float4*>(resid)[i];
    acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
  }
  reinterpret_cast<float4*>(out)[i] = acc;
}

// column sums in two fixed-order stages: partial[b][c] = sum over rows [b * CS_ROWS, ...) ; out[c] = sum_b partial[b][c]
// (32-row blocks: a slate is a few thousand rows, 256-row blocks left the chip to 12 x 16 workgroups: 71 us per call, 98
// calls per step)
constexpr int CS_ROWS = 32;
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ x, const float* __restrict__ y /*nullable: sums x*y*/,
                                                             int M, int N, float* __restrict__ partial) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float s = 0.f;
#pragma unroll 8
  for (int r = r0; r < r1; ++r) s += y ? x[(size_t)r * N + c] * y[(size_t)r * N + c] : x[(size_t)r * N + c];
  partial[(size_t)blockIdx.y * N + c] = s;
}
// (64 columns per workgroup, the partials of a column dealt round-robin to the four waves and added up in a fixed order)
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* __restrict__ partial, int nb, int N,
                                                           float* __restrict__ out) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < N)
    for (int b = wave; b < nb; b += 4) s += partial[(size_t)b * N + c];
  sm[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < N) out[c] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// LayerNorm backward, one wave per row: dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma;
// dx_out = (base ? base : 0) + dx; xhat_dy = dy * xhat (its column sums are dgamma; dbeta = column sums of dy)
__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ dy, const float* base, int M, int H,
                                                     float* dx_out, float* __restrict__ xhat_dy) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * H;
  const float* dr = dy + (size_t)row * H;
  float s = 0.f;
  for (int c = lane; c < H; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
  for (int c = lane; c < H; c += 64) { const float d = xr[c] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + LN_EPS);
  float sg = 0.f, sgx = 0.f;
  for (int c = lane; c < H; c += 64) {
    const float xh = own_reg((xr[c] - mean) * rstd), g = dr[c] * gamma[c];     // (own_reg: isa_lint.py)
    sg += g; sgx += g * xh;
  }
  const float mg = wave_sum(sg) / (float)H, mgx = wave_sum(sgx) / (float)H;
  for (int c = lane; c < H; c += 64) {
    const float xh = (xr[c] - mean) * rstd, g = dr[c] * gamma[c];
    const float dx = rstd * (g - mg - xh * mgx);
    const size_t o = (size_t)row * H + c;
    xhat_dy[o] = dr[c] * xh;
    dx_out[o] = (base ? base[o] : 0.f) + dx;
  }
}

__global__ void __launch_bounds__(256) relu_bwd_kernel(float* __restrict__ df, const float* __restrict__ f, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    if (!(f[i] > 0.f)) df[i] = 0.f;
}

// counter-based dropout mask: keep with probability 1 - p, scale by 1 / (1 - p)
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float p) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.f / 16777216.f);
  return u < p ? 0.f : 1.f / (1.f - p);
}
// y = base + dropout(x)   (forward: base = residual; backward: x = dout, base = nullptr)
__global__ void __launch_bounds__(256) dropout_add_kernel(const float* __restrict__ x, const float* base, float* y, size_t n,
                                                          uint64_t seed, float p) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    y[i] = (base ? base[i] : 0.f) + x[i] * dropout_scale(seed, i, p);
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ cu,
                                                          int n_req, int H, float* __restrict__ dst) {
  const int r = blockIdx.x;
  if (r >= n_req) return;
  const size_t s = (size_t)(cu[r + 1] - 1) * H;
  for (int c = threadIdx.x; c < H; c += 256) dst[(size_t)r * H + c] = src[s + c];
}
__global__ void __launch_bounds__(256) scatter_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ cu,
                                                           int n_req, int H, float* __restrict__ dst) {
  const int r = blockIdx.x;
  if (r >= n_req) return;
  const size_t s = (size_t)(cu[r + 1] - 1) * H;
  for (int c = threadIdx.x; c < H; c += 256) dst[s + c] = src[(size_t)r * H + c];
}

// logits[r, j] = y[r, :] . Ws[j, :]     one wave per (row, label)
__global__ void __launch_bounds__(256) head_fwd_kernel(const float* __restrict__ y, const float* __restrict__ ws, int N, int De,
                                                       int nl, float* __restrict__ logits) {
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (long long)N * nl) return;
  const int r = (int)(w / nl), j = (int)(w % nl);
  float s = 0.f;
  for (int c = lane; c < De; c += 64) s += y[(size_t)r * De + c] * ws[(size_t)j * De + c];
  s = wave_sum(s);
  if (lane == 0) logits[w] = s;
}
// dy[r, c] = sum_j dl[r, j] Ws[j, c];  dWs[j, c] = sum_r dl[r, j] y[r, c]   (fixed order over j / r)
__global__ void __launch_bounds__(256) head_bwd_dy_kernel(const float* __restrict__ dl, const float* __restrict__ ws, int N, int De,
                                                          int nl, float* __restrict__ dy) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)N * De) return;
  const int r = (int)(i / De), c = (int)(i % De);
  float s = 0.f;
  for (int j = 0; j < nl; ++j) s += dl[(size_t)r * nl + j] * ws[(size_t)j * De + c];
  dy[i] = s;
}
__global__ void __launch_bounds__(256) head_bwd_dw_kernel(const float* __restrict__ dl, const float* __restrict__ y, int N, int De,
                                                          int nl, float* __restrict__ dws) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)nl * De) return;
  const int j = (int)(i / De), c = (int)(i % De);
  float s = 0.f;
  for (int r = 0; r < N; ++r) s += dl[(size_t)r * nl + j] * y[(size_t)r * De + c];
  dws[i] = s;
}

// MSELoss (mean over the N outputs) and its gradient; single workgroup, fixed order
__global__ void __launch_bounds__(256) mse_kernel(const float* __restrict__ o, const float* __restrict__ y, int N,
                                                  float* __restrict__ loss, float* __restrict__ dl) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) { const float d = o[i] - y[i]; s += d * d; dl[i] = 2.f * d / (float)N; }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int k = 0; k < 256; ++k) t += red[k]; *loss = t / (float)N; }
}
// CrossEntropyLoss (mean over rows): one wave per row writes row_loss and dl; a second kernel averages
__global__ void __launch_bounds__(256) ce_rows_kernel(const float* __restrict__ logits, const float* __restrict__ labels, int N,
                                                      int nl, float* __restrict__ row_loss, float* __restrict__ dl) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= N) return;
  const float* lr = logits + (size_t)r * nl;
  float mx = -INFINITY;
  for (int j = lane; j < nl; j += 64) mx = fmaxf(mx, lr[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < nl; j += 64) s += expf(lr[j] - mx);
  s = wave_sum(s);
  // a label outside [0, nl) (the reference asserts labels.max() < num_labels, trainer.py:151; the host mirror raises
  // before the call) must not index the row: the loss of the step becomes NaN instead
  const float labf = labels[r];
  const bool lab_ok = labf >= 0.f && labf < (float)nl;
  const int lab = lab_ok ? (int)labf : 0;
  for (int j = lane; j < nl; j += 64) dl[(size_t)r * nl + j] = (expf(lr[j] - mx) / s - (j == lab ? 1.f : 0.f)) / (float)N;
  if (lane == 0) row_loss[r] = lab_ok ? logf(s) + mx - lr[lab] : __builtin_nanf("");
}
__global__ void __launch_bounds__(64) mean_kernel(const float* __restrict__ v, int N, float* __restrict__ out) {
  if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < N; ++i) s += v[i]; *out = s / (float)N; }
}

// embedding backward: dE_pos[pos + 2] += dh0[t]; dE_tok[id] += dtok[t], duplicates summed in TOKEN ORDER by one owner - no
// atomics, so a step is bit-reproducible.  Workgroup t owns a table row iff no earlier token uses that row (positions: no
// earlier request is longer than t's offset; tokens: no t' < t carries the same id); the owner walks the later users in
// ascending order with the row in registers.  O(T^2 / 256) id comparisons per workgroup at worst: ~1 ms of the step at the
// trainer's largest slate (32 prompts x 2,048 tokens), microseconds at the usual 4k tokens.
constexpr int EB_COLS = 8;      // columns per thread: H, De <= 256 * 8
__device__ __forceinline__ long long clamp_id(long long id, int vocab) { return id < 0 ? 0 : (id >= vocab ? vocab - 1 : id); }
__global__ void __launch_bounds__(256) embed_bwd_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ cu, int n_req,
                                                        int T, const float* __restrict__ dh0, const float* __restrict__ dtok,
                                                        int H, int De, int vocab, int pos_rows, float* __restrict__ g_pos,
                                                        float* __restrict__ g_tok) {
  __shared__ unsigned long long masks[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  if (t >= T) return;
  const int req = find_request(cu, n_req, t);
  const int off = t - cu[req];
  const int pos = min(off + 2, pos_rows - 1);          // (ltr_train_step refuses requests longer than pos_rows - 2: never clamps)
  // ---- position row: users are the requests longer than `off`, in request order
  int earlier = 0;
  for (int r = tid; r < req; r += 256) earlier |= (cu[r + 1] - cu[r]) > off;
  if (!__syncthreads_or(earlier)) {
    float acc[EB_COLS];
#pragma unroll
    for (int k = 0; k < EB_COLS; ++k) acc[k] = 0.f;
    for (int r = req; r < n_req; ++r) {
      const int base = cu[r];
      if (cu[r + 1] - base <= off) continue;
      const float* src = dh0 + (size_t)(base + off) * H;
#pragma unroll
      for (int k = 0; k < EB_COLS; ++k) { const int c = tid + k * 256; if (c < H) acc[k] += src[c]; }
    }
#pragma unroll
    for (int k = 0; k < EB_COLS; ++k) { const int c = tid + k * 256; if (c < H) g_pos[(size_t)pos * H + c] += acc[k]; }
  }
  // ---- token row: users are the tokens with the same id, in token order
  const long long id = clamp_id(ids[t], vocab);
  int dup = 0;
  for (int u = tid; u < t; u += 256) dup |= clamp_id(ids[u], vocab) == id;
  if (__syncthreads_or(dup)) return;
  float acc[EB_COLS];
#pragma unroll
  for (int k = 0; k < EB_COLS; ++k) acc[k] = 0.f;
  for (int base = t; base < T; base += 256) {
    const int u = base + tid;
    const unsigned long long m = __ballot(u < T && clamp_id(ids[u], vocab) == id);
    if ((tid & 63) == 0) masks[tid >> 6] = m;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
      unsigned long long mm = masks[w];
      while (mm) {
        const int b = __builtin_ctzll(mm);
        mm &= mm - 1;
        const float* src = dtok + (size_t)(base + w * 64 + b) * De;
#pragma unroll
        for (int k = 0; k < EB_COLS; ++k) { const int c = tid + k * 256; if (c < De) acc[k] += src[c]; }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < EB_COLS; ++k) { const int c = tid + k * 256; if (c < De) g_tok[(size_t)id * De + c] += acc[k]; }
}

// torch.optim.Adam with L2 weight decay: g += wd p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr (m / bc1) / (sqrt(v / bc2) + eps)
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float pi = p[i];
    const float gi = g[i] + wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = pi - lr * (mi / bc1) / (sqrtf(vi) / sqrtf(bc2) + eps);
  }
}

// ------------------------------------------------------------------------------------------------------------
// attention backward (f32 VALU).  Forward (attn_f32_kernel): S2 = (q scale log2e) . k, P = exp2(S2 - lse2), O = P V.
// With D_q = sum_d dO O:  dV_k = sum_q P dO;  dS = P (dO . V_k - D_q);  dQ = scale sum_k dS K;  dK = scale sum_q dS Q.
// ------------------------------------------------------------------------------------------------------------
constexpr int BQ = 64;    // rows per block of the work list (attn_blocks_kernel with qb = 64)
constexpr int BT = 32;    // rows staged in LDS per step

// Workgroup = a block of 64 rows (lane = row) x BW waves: the waves deal the tiles of the OTHER dimension among themselves
// (round-robin), each with its own LDS staging area, and their partial sums are added up in wave order at the end.
// (Round 2 ran one wave per block: the launch then lasted as long as the serial loop of the longest prompt - 1,024 keys
// x 192 FMAs per lane - while three quarters of the chip had nothing to do: 410 + 750 us per layer for a 32-prompt slate.)
constexpr int BW = 4;

// one lane per query: dQ and D
__global__ void __launch_bounds__(64 * BW) attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                              const float* __restrict__ dout, const float* __restrict__ lse2,
                                                              const int32_t* __restrict__ blk_start, const int4* __restrict__ blk_desc,
                                                              int n_req, int H, float scale, float* __restrict__ dqkv,
                                                              float* __restrict__ Dq) {
  __shared__ __attribute__((aligned(16))) float smem[BW * 2 * BT * D];          // per wave: s_k | s_v; reused for the reduction
  const int b = blockIdx.x;
  if (b >= blk_start[n_req]) return;
  const int head = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nh = H / D;
  float* s_k = smem + wave * 2 * BT * D;
  float* s_v = s_k + BT * D;
  const int4 desc = blk_desc[b];
  const int q0 = desc.y, t0 = desc.z, L = desc.w;
  const int qi = q0 + lane;
  const bool valid = qi < L;
  const size_t ld = (size_t)3 * H;
  const float sl2 = scale * 1.4426950408889634f;
  float q[D], dO[D], dq[D];
  float dsum = 0.f, lse = 0.f;
  {
    const size_t row = (size_t)(t0 + (valid ? qi : 0));
    const float* qp = qkv + row * ld + head * D;
    const float* op = o + row * H + head * D;
    const float* dp = dout + row * H + head * D;
#pragma unroll
    for (int d = 0; d < D; ++d) { q[d] = qp[d] * sl2; dO[d] = dp[d]; dsum += dp[d] * op[d]; dq[d] = 0.f; }
    lse = lse2[row * nh + head];
    if (valid && wave == 0) Dq[row * nh + head] = dsum;
  }
  const int kend = min(L, q0 + BQ);
  for (int kt = wave * BT; kt < kend; kt += BW * BT) {        // this wave's key tiles
    __builtin_amdgcn_wave_barrier();                            // (wave-private staging: no workgroup barrier)
    for (int idx = lane; idx < BT * D / 4; idx += 64) {
      const int key = idx >> 4, d4 = (idx & 15) * 4;
      const float* base = qkv + (size_t)(t0 + min(kt + key, L - 1)) * ld + head * D + d4;
      *reinterpret_cast<float4*>(s_k + key * D + d4) = *reinterpret_cast<const float4*>(base + H);
      *reinterpret_cast<float4*>(s_v + key * D + d4) = *reinterpret_cast<const float4*>(base + 2 * H);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < BT; ++j) {
      const int kj = kt + j;
      if (kj >= kend) break;                               // uniform
      const float* kr = s_k + j * D;
      const float* vr = s_v + j * D;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s = fmaf(q[d], kr[d], s); dp = fmaf(dO[d], vr[d], dp); }
      const float p = (valid && kj <= qi) ? exp2f(s - lse) : 0.f;
      const float ds = p * (dp - dsum);
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
    }
  }
  // partial dq of the waves -> wave 0, added in wave order (fixed order: deterministic)
  __syncthreads();
  float* red = smem;                                         // [BW - 1][D][64 lanes] = 48 KiB of the 64 KiB
  if (wave > 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) red[((wave - 1) * D + d) * 64 + lane] = dq[d];
  }
  __syncthreads();
  if (wave != 0 || !valid) return;
  float* out = dqkv + (size_t)(t0 + qi) * ld + head * D;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float v = dq[d];
#pragma unroll
    for (int w = 0; w < BW - 1; ++w) v += red[(w * D + d) * 64 + lane];
    out[d] = v * scale;
  }
}

// one lane per key: dV (pass 0) and dK (pass 1); queries streamed through LDS, the query tiles dealt to the waves
__global__ void __launch_bounds__(64 * BW) attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                               const float* __restrict__ lse2, const float* __restrict__ Dq,
                                                               const int32_t* __restrict__ blk_start, const int4* __restrict__ blk_desc,
                                                               int n_req, int H, float scale, float* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) float smem[BW * (2 * BT * D + 2 * BT)];   // per wave: s_q | s_do | s_lse | s_dq
  const int b = blockIdx.x;
  if (b >= blk_start[n_req]) return;
  const int head = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nh = H / D;
  float* s_q = smem + wave * (2 * BT * D + 2 * BT);
  float* s_do = s_q + BT * D;
  float* s_lse = s_do + BT * D;
  float* s_dq = s_lse + BT;
  const int4 desc = blk_desc[b];
  const int k0 = desc.y, t0 = desc.z, L = desc.w;
  const int ki = k0 + lane;
  const bool valid = ki < L;
  const size_t ld = (size_t)3 * H;
  const float sl2 = scale * 1.4426950408889634f;
  float kk[D], vv[D], acc[D];
  {
    const float* kp = qkv + (size_t)(t0 + (valid ? ki : 0)) * ld + H + head * D;
#pragma unroll
    for (int d = 0; d < D; ++d) { kk[d] = kp[d]; vv[d] = kp[H + d]; }
  }
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    for (int qt = k0 + wave * BT; qt < L; qt += BW * BT) {   // this wave's query tiles (queries >= the first key of this block)
      __builtin_amdgcn_wave_barrier();
      for (int idx = lane; idx < BT * D / 4; idx += 64) {
        const int qq = idx >> 4, d4 = (idx & 15) * 4;
        const size_t row = (size_t)(t0 + min(qt + qq, L - 1));
        float4 qv = *reinterpret_cast<const float4*>(qkv + row * ld + head * D + d4);
        qv.x *= sl2; qv.y *= sl2; qv.z *= sl2; qv.w *= sl2;
        *reinterpret_cast<float4*>(s_q + qq * D + d4) = qv;
        *reinterpret_cast<float4*>(s_do + qq * D + d4) = *reinterpret_cast<const float4*>(dout + row * H + head * D + d4);
      }
      if (lane < BT) {
        const size_t row = (size_t)(t0 + min(qt + lane, L - 1));
        s_lse[lane] = lse2[row * nh + head];
        s_dq[lane] = Dq[row * nh + head];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      for (int j = 0; j < BT; ++j) {
        const int qj = qt + j;
        if (qj >= L) break;                                // uniform
        const float* qr = s_q + j * D;
        const float* dr = s_do + j * D;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s = fmaf(qr[d], kk[d], s); dp = fmaf(dr[d], vv[d], dp); }
        const float p = (valid && qj >= ki) ? exp2f(s - s_lse[j]) : 0.f;
        if (pass == 0) {
#pragma unroll
          for (int d = 0; d < D; ++d) acc[d] = fmaf(p, dr[d], acc[d]);
        } else {
          const float ds = p * (dp - s_dq[j]);
#pragma unroll
          for (int d = 0; d < D; ++d) acc[d] = fmaf(ds, qr[d], acc[d]);
        }
      }
    }
    // partial sums of the waves -> wave 0, in wave order
    __syncthreads();
    float* red = smem;                                       // [BW - 1][D][64] floats = 48 KiB (the staging areas are idle)
    if (wave > 0) {
#pragma unroll
      for (int d = 0; d < D; ++d) red[((wave - 1) * D + d) * 64 + lane] = acc[d];
    }
    __syncthreads();
    if (wave == 0 && valid) {
      // pass 0 -> dV; pass 1 -> dK = scale sum dS q = sum dS (q scale log2e) / log2e
      float* out = dqkv + (size_t)(t0 + ki) * ld + (pass == 0 ? 2 * H : H) + head * D;
      const float f = pass == 0 ? 1.f : 0.6931471805599453f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        float v = acc[d];
#pragma unroll
        for (int w = 0; w < BW - 1; ++w) v += red[(w * D + d) * 64 + lane];
        out[d] = v * f;
      }
    }
    __syncthreads();                                         // the reduction area is the next pass's staging area
  }
}

inline size_t up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace
}  // namespace ltr

using namespace ltr;

// ----------------------------------------------------------------------------------------------------------------
// trainer object
// ----------------------------------------------------------------------------------------------------------------
struct ltr_trainer {
  ltr_model_desc d;
  ltr_train_config cfg;
  int device = 0;
  int64_t step = 0;                    // forward/backward passes so far (seeds the dropout masks)
  int64_t adam_step = 0;               // optimizer.step() calls so far: torch.optim.Adam advances t only there
  bool use_f32 = false;                // LTR_TRAIN_F32=1: every GEMM on the exact-f32 MFMA kernel
  size_t total = 0;
  std::vector<size_t> off, cnt;        // per weight index (ltr_create's index space)
  float *P = nullptr, *G = nullptr, *M1 = nullptr, *V1 = nullptr, *wT = nullptr;
  size_t wT_elems = 0;
  ~ltr_trainer() {
    for (float* p : {P, G, M1, V1, wT}) if (p) (void)hipFree(p);
  }
  float* p(int idx) const { return P + off[idx]; }
  float* g(int idx) const { return G + off[idx]; }
  int li(int layer, int i) const { return LTR_WT_GLOBAL_COUNT + layer * LTR_WL_COUNT + i; }
};

namespace {

struct DevGuard {
  int prev = -1; bool sw = false;
  explicit DevGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) sw = hipSetDevice(dev) == hipSuccess; }
  ~DevGuard() { if (sw) (void)hipSetDevice(prev); }
};

size_t weight_count(const ltr_model_desc& d, int idx) {
  const size_t H = d.hidden_size, F = d.ffn_dim, De = d.word_embed_proj_dim;
  const bool proj = De != H;
  if (idx < LTR_WT_GLOBAL_COUNT) {
    switch (idx) {
      case LTR_WT_EMBED_TOKENS: return (size_t)d.vocab_size * De;
      case LTR_WT_EMBED_POS: return (size_t)d.pos_rows * H;
      case LTR_WT_PROJECT_IN: return proj ? H * De : 0;
      case LTR_WT_PROJECT_OUT: return proj ? De * H : 0;
      case LTR_WT_FINAL_LN_W: case LTR_WT_FINAL_LN_B: return d.pre_ln ? H : 0;
      case LTR_WT_SCORE: return (size_t)d.num_labels * De;
    }
    return 0;
  }
  switch ((idx - LTR_WT_GLOBAL_COUNT) % LTR_WL_COUNT) {
    case LTR_WL_QKV_W: return 3 * H * H;
    case LTR_WL_QKV_B: return 3 * H;
    case LTR_WL_OUT_W: return H * H;
    case LTR_WL_FC1_W: case LTR_WL_FC2_W: return F * H;
    case LTR_WL_FC1_B: return F;
    default: return H;     // out / fc2 bias, LayerNorm terms
  }
}

// workspace carve-up of one step
constexpr int SPLITK_MAX = 8;        // parts of a split-K weight-gradient GEMM
constexpr int MAX_GEMMS = 4096;      // split GEMMs per step (a 24-layer model makes ~290)
struct TrainWs {
  struct Layer { float *x0, *n1, *qkv, *ao, *lse, *mid, *n2, *f, *ao_raw, *mlp_raw; };
  std::vector<Layer> L;
  float *tok, *hfin, *hl, *z, *y, *logits, *dlogits, *row_loss, *dy, *dz, *dhl;
  float *dh, *dbig, *dsmall, *dsmall2, *xhd, *t1, *t2, *partial, *Dq;
  float *opa, *wb;         // split-fp16 GEMM operands of one call: A image hi|lo over 2K, [Bh | Bl] weight image
  float *qkvp, *aop;       // hi | lo planes of qkv [2][T, 3H] and of the attention output [2][T, H] (fp16 path)
  float* ndcg;             // scratch of the NeuralNDCG loss (ltr_neuralndcg): the unrolled Sinkhorn rounds of one slate
  size_t ndcg_bytes;
  float *gtmp, *scales;    // partial products of a split-K weight-gradient GEMM; max |x| slots of the GEMM operands of the step
  int32_t* blk;
  int32_t* blk2;     // work list of the MFMA attention forward (fp16 path)
  size_t gtmp_floats;
  size_t bytes;
};

TrainWs carve_train(const ltr_model_desc& d, int64_t T, int64_t N, void* base, bool dropout, bool ndcg) {
  const size_t H = d.hidden_size, F = d.ffn_dim, De = d.word_embed_proj_dim, nh = d.num_heads, nl = d.num_labels;
  const size_t Tp = (T + 127) / 128 * 128, Np = (N + 127) / 128 * 128;   // (contraction dims of the weight-gradient GEMMs, padded)
  const size_t big = std::max<size_t>(3 * H, F);
  char* p = (char*)base;
  size_t o = 0;
  auto take = [&](size_t n_floats) { size_t q = o; o += up(n_floats * 4); return base ? (float*)(p + q) : (float*)nullptr; };
  TrainWs w;
  w.L.resize(d.num_layers);
  for (auto& l : w.L) {
    l.x0 = take(T * H); l.n1 = take(T * H); l.qkv = take(T * 3 * H); l.ao = take(T * H); l.lse = take(T * nh);
    l.mid = take(T * H); l.n2 = take(T * H); l.f = take(T * F);
    l.ao_raw = dropout ? take(T * H) : nullptr; l.mlp_raw = dropout ? take(T * H) : nullptr;
  }
  w.tok = take(T * De); w.hfin = take(T * H);
  w.hl = take(Np * H); w.z = take(Np * H); w.y = take(Np * De); w.logits = take(N * nl); w.dlogits = take(N * nl);
  w.row_loss = take(N + 8);
  w.ndcg_bytes = ndcg ? ltr_neuralndcg_workspace_bytes(1, (int32_t)std::min<int64_t>(N, 1 << 20)) : 0;   // 0 beyond its slate limit: the step refuses
  w.ndcg = take(w.ndcg_bytes / 4);
  w.dy = take(Np * De); w.dz = take(Np * H); w.dhl = take(Np * H);
  w.dh = take(T * H); w.dbig = take(T * big); w.dsmall = take(T * H); w.dsmall2 = take(T * H); w.xhd = take(T * H);
  w.t1 = take(big * Tp); w.t2 = take(big * Tp);
  w.qkvp = take(T * 3 * H + 64); w.aop = take(T * H + 64);           // 2 planes x 2 B = one float per element
  w.gtmp_floats = std::max((size_t)SPLITK_MAX * big * std::max(H, De), (size_t)4 * Tp * H);
  w.gtmp = take(w.gtmp_floats + 64); w.scales = take(MAX_GEMMS * 2);
  w.opa = take(2 * big * Tp + 64);                                   // 2 planes x M x 2K halves = 2 M K floats
  w.wb = take(std::max(std::max(big * Tp, big * H), F * H) + 64);    // N x 2K halves = N K floats
  w.partial = take(((size_t)(T + CS_ROWS - 1) / CS_ROWS + 1) * big);
  w.Dq = take(T * nh);
  w.blk = (int32_t*)take((N + 4) + (T / 64 + N + 1) * 4);
  w.blk2 = (int32_t*)take((N + 4) + (T / 64 + N + 1) * 4);
  w.bytes = o;
  return w;
}

struct Ctx {
  ltr_trainer* t;
  hipStream_t s;
  TrainWs ws;
  int T, N;
  bool eval = false;      // predictor.model.eval() (trainer.py:171): no dropout, forward only
  int n_slot = 0;         // max|x| slots handed out so far in this step (ws.scales)
  bool attn_mfma = false; // the forward ran the split-fp16 MFMA attention (128-row work list in ws.blk2): so does the backward
  // max|x| of tensors that do not change for the rest of the step (weights, saved activations), by address: the
  // backward meets every forward operand again (X of dW = dY^T X, W of dX = dY W) and reuses its slot
  std::unordered_map<const void*, float*> amax_cache;
};

// slot holding max|x| of p[0..n) (device, filled on the stream).  `stable`: p keeps its contents until the step ends.
float* amax_of(Ctx& c, const float* p, size_t n, bool stable) {
  if (stable) {
    auto it = c.amax_cache.find(p);
    if (it != c.amax_cache.end()) return it->second;
  }
  if (c.n_slot >= 2 * MAX_GEMMS) { set_error("ltr_train_step: more than %d GEMM operands in one step", 2 * MAX_GEMMS); return nullptr; }
  float* sl = c.ws.scales + c.n_slot++;
  amax_kernel<<<(unsigned)std::min<size_t>((n + 4095) / 4096, AMAX_BLOCKS), 256, 0, c.s>>>(p, n, sl);
  if (stable) c.amax_cache[p] = sl;
  return sl;
}

// One operand of a split GEMM: `rows` x `kr` logical matrix (contraction along kr).  !trans: p is [rows, kr] row-major;
// trans: p is [kr, rows] row-major (the operand is p^T).  amax: its max|x| slot if the caller already has one.
struct Opnd { const float* p; int rows, kr; bool trans, stable; float* amax; };

int transpose_pad(const float* in, int R, int C, int Rp, float* out, hipStream_t s) {
  dim3 grid((C + 31) / 32, (Rp + 31) / 32);
  transpose_pad_kernel<<<grid, 256, 0, s>>>(in, R, C, Rp, out);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

// C[M, N] = A B^T (+ bias)(ReLU)(+ resid) with A = a (M x K), B = b (N x K), f32 in memory, on the split-fp16 MFMA kernel:
// [A | A] [Bh | Bl]^T over K' = 2K (Ah Bh + Al Bh + Ah Bl + Al Bl), each operand scaled into fp16's range by a power of
// two first (split_scale), undone in the GEMM epilogue (GemmArgs::osc_a / osc_b).  Per operand ONE preparation kernel (transposing or not).
bool split_ok(const Ctx& c, const Opnd& a, const Opnd& b) {
  if (c.t->use_f32 || b.rows % 64) return false;
  if (a.trans && a.rows % 64) return false;
  if (!a.trans && a.kr % 32) return false;
  if (!b.trans && b.kr % 32) return false;
  return true;
}
int gemm_split(Ctx& c, Opnd a, Opnd b, const float* bias, const float* resid, float* out, int relu, AOp* planes) {
  hipStream_t s = c.s;
  // (weight gradients: the token dimension is padded to 128 so that it can be cut into up to 4 x 2 equal parts of whole slabs)
  const int M = a.rows, N = b.rows, K = a.trans && b.trans ? (a.kr + 127) / 128 * 128 : (a.kr + 31) / 32 * 32;
  if (!a.amax) a.amax = amax_of(c, a.p, (size_t)a.rows * a.kr, a.stable);
  if (!b.amax) b.amax = amax_of(c, b.p, (size_t)b.rows * b.kr, b.stable);
  if (!a.amax || !b.amax) return LTR_E_INVAL;
  __half* ah = reinterpret_cast<__half*>(c.ws.opa);
  __half* al = ah + (size_t)M * 2 * K;
  if (a.trans) {
    split_T_kernel<true><<<dim3(M / 64, K / 32), 256, 0, s>>>(a.p, ah, al, a.kr, M, K, a.amax);
  } else {
    const size_t apieces = (size_t)M * K / 8;
    split_operand_dup_kernel<<<(unsigned)((apieces + 255) / 256), 256, 0, s>>>(a.p, ah, al, M, K, a.amax);
  }
  if (b.trans) {
    split_T_kernel<false><<<dim3(N / 64, K / 32), 256, 0, s>>>(b.p, reinterpret_cast<__half*>(c.ws.wb), nullptr, b.kr, N, K, b.amax);
  } else {
    const size_t pieces = (size_t)N * K / 4;
    split_pack_kernel<<<(unsigned)((pieces + 255) / 256), 256, 0, s>>>(b.p, reinterpret_cast<__half*>(c.ws.wb), N, K, b.amax);
  }
  LTR_LAUNCH_CHECK();
  // the product comes back scaled by both operand scales: the GEMM epilogue divides them out (exact powers of two)
  // before bias / ReLU / residual, and writes the row-major hi | lo planes of the result when asked (QKV for the MFMA attention)
  GemmArgs r{};
  r.a = AOp{ah, al}; r.w = c.ws.wb; r.M = M; r.N = N; r.K = 2 * K; r.a_slab = 1;
  // Few output tiles, long K (weight gradients: an [N_out, K_out] output while K = the tokens of the slate; the 768-wide
  // outputs of a 4k-token slate: 93 tiles for 512 workgroup slots): cut K into parts until the launch has about one
  // workgroup per slot, add the parts up in order and apply the epilogue there
  if (!planes) {
    static const int want = [] { const char* e = getenv("LTR_TRAIN_SPLITK_WGS"); return e ? atoi(e) : 384; }();   // A/B knob, 0 = off
    static const int narrow = [] { const char* e = getenv("LTR_TRAIN_SPLITK_NARROW"); return e ? atoi(e) : 1; }(); // 0: weight gradients only
    const int tiles = ((M + 127) / 128) * ((N + 255) / 256), slabs = 2 * K / 32;
    int parts = 1;
    if ((a.trans && b.trans) || narrow)
      while (want > 0 && parts < SPLITK_MAX && tiles * parts * 2 <= want && slabs % (parts * 2) == 0 && slabs / (parts * 2) >= 16) parts *= 2;
    if (parts > 1 && (size_t)parts * M * N <= c.ws.gtmp_floats) {
      r.out_f32 = c.ws.gtmp; r.split_k = parts;
      int rc = launch_gemm(LTR_W_F16, r, s);
      if (rc) return rc;
      const size_t n4 = (size_t)M * N / 4;
      splitk_reduce_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>(c.ws.gtmp, parts, n4, N, a.amax, b.amax, bias, resid, relu, out);
      LTR_LAUNCH_CHECK();
      return LTR_OK;
    }
  }
  r.bias = bias; r.resid = resid; r.out_f32 = out; r.relu = relu;
  if (planes) r.out_split = *planes;
  r.osc_a = a.amax; r.osc_b = b.amax;
  return launch_gemm(LTR_W_F16, r, s);
}

// C[M, N] = A[M, K] B[N, K]^T (+ bias)(ReLU)(+ resid), all f32 in memory.  Forward operands: both stable for the step.
int gemm_nt(Ctx& c, const float* A, const float* B, const float* bias, const float* resid, float* out, int M, int N, int K,
            int relu, AOp* planes = nullptr /*fp16 path: also write the result as row-major hi | lo planes; cleared if it did not*/) {
  const Opnd a{A, M, K, false, true, nullptr}, b{B, N, K, false, true, nullptr};
  if (split_ok(c, a, b)) return gemm_split(c, a, b, bias, resid, out, relu, planes);
  if (planes) *planes = AOp{nullptr, nullptr};
  GemmArgs g{};
  g.bias = bias; g.resid = resid; g.out_f32 = out; g.M = M; g.N = N; g.relu = relu;
  g.a = AOp{(void*)A, nullptr}; g.w = B; g.K = K;
  return launch_gemm(LTR_W_F32, g, c.s);
}
// dX[M, K] = (resid) + dY[M, N] W[N, K]        (dy_amax: max|dY| slot shared with the weight-gradient GEMM of the same dY)
int gemm_nn(Ctx& c, const float* dY, const float* W, const float* resid, float* dX, int M, int N, int K, float* dy_amax) {
  const Opnd a{dY, M, N, false, false, dy_amax}, b{W, K, N, true, true, nullptr};
  if (split_ok(c, a, b)) return gemm_split(c, a, b, nullptr, resid, dX, 0, nullptr);
  int rc = transpose_pad(W, N, K, N, c.t->wT, c.s);          // W^T [K, N]
  if (rc) return rc;
  GemmArgs g{};
  g.resid = resid; g.out_f32 = dX; g.M = M; g.N = K; g.K = N; g.a = AOp{(void*)dY, nullptr}; g.w = c.t->wT;
  return launch_gemm(LTR_W_F32, g, c.s);
}
// dW[N, K] = dY[M, N]^T X[M, K]
int gemm_tn(Ctx& c, const float* dY, const float* X, float* dW, int M, int N, int K, float* dy_amax) {
  const Opnd a{dY, N, M, true, false, dy_amax}, b{X, K, M, true, true, nullptr};
  if (split_ok(c, a, b)) return gemm_split(c, a, b, nullptr, nullptr, dW, 0, nullptr);
  const int Mp = (M + 31) / 32 * 32;
  int rc = transpose_pad(dY, M, N, Mp, c.ws.t1, c.s);        // [N, Mp]
  if (rc) return rc;
  if ((rc = transpose_pad(X, M, K, Mp, c.ws.t2, c.s))) return rc;   // [K, Mp]
  GemmArgs g{};
  g.out_f32 = dW; g.M = N; g.N = K; g.K = Mp; g.a = AOp{c.ws.t1, nullptr}; g.w = c.ws.t2;
  return launch_gemm(LTR_W_F32, g, c.s);
}
// max|dY| once for the two GEMMs that consume the same gradient (nullptr on the exact-f32 path: nothing is scaled there)
float* grad_amax(Ctx& c, const float* dY, size_t n) { return c.t->use_f32 ? nullptr : amax_of(c, dY, n, false); }
int colsum(Ctx& c, const float* x, const float* y, int M, int N, float* out) {
  const int nb = (M + CS_ROWS - 1) / CS_ROWS;
  dim3 grid((N + 255) / 256, nb);
  colsum_partial_kernel<<<grid, 256, 0, c.s>>>(x, y, M, N, c.ws.partial);
  LTR_LAUNCH_CHECK();
  colsum_final_kernel<<<(N + 63) / 64, 256, 0, c.s>>>(c.ws.partial, nb, N, out);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}
// LayerNorm backward of y = LN(x): dx_out = base + dLN; accumulates dgamma / dbeta into dg / db (plain stores: once per step)
int ln_bwd(Ctx& c, const float* x, const float* gamma, const float* dy, const float* base, int M, int H, float* dx_out,
           float* dg, float* db) {
  ln_bwd_kernel<<<(M + 3) / 4, 256, 0, c.s>>>(x, gamma, dy, base, M, H, dx_out, c.ws.xhd);
  LTR_LAUNCH_CHECK();
  int rc = colsum(c, c.ws.xhd, nullptr, M, H, dg);
  if (rc) return rc;
  return colsum(c, dy, nullptr, M, H, db);
}
uint64_t drop_seed(const ltr_trainer* t, int layer, int site) {
  return (t->cfg.seed * 0x100000001B3ull) ^ ((uint64_t)t->step << 20) ^ ((uint64_t)layer << 4) ^ (uint64_t)site;
}

#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

int forward(Ctx& c, const int64_t* ids, const int32_t* cu) {
  ltr_trainer* t = c.t;
  const ltr_model_desc& d = t->d;
  const int H = d.hidden_size, F = d.ffn_dim, De = d.word_embed_proj_dim, T = c.T, N = c.N, nl = d.num_labels;
  const bool proj = De != H, drop = t->cfg.dropout > 0.f && !c.eval;
  hipStream_t s = c.s;
  float* h0 = d.num_layers ? c.ws.L[0].x0 : c.ws.hfin;
  RC(launch_embed_gather(LTR_W_F32, ids, cu, N, T, 0, t->p(LTR_WT_EMBED_TOKENS), De, d.vocab_size, t->p(LTR_WT_EMBED_POS), H,
                         d.pos_rows, h0, AOp{c.ws.tok, nullptr}, nullptr, s));
  if (proj) RC(gemm_nt(c, c.ws.tok, t->p(LTR_WT_PROJECT_IN), nullptr, h0, h0, T, H, De, 0));
  for (int l = 0; l < d.num_layers; ++l) {
    auto& L = c.ws.L[l];
    float* out = l + 1 < d.num_layers ? c.ws.L[l + 1].x0 : c.ws.hfin;
    const float* qkv_in = L.x0;
    if (d.pre_ln) {
      RC(launch_layernorm(LTR_W_F32, L.x0, t->p(t->li(l, LTR_WL_LN1_W)), t->p(t->li(l, LTR_WL_LN1_B)), T, H, L.n1, AOp{nullptr, nullptr}, s));
      qkv_in = L.n1;
    }
    // QKV: f32 for the backward and - on the fp16 path - hi | lo planes for the MFMA attention kernel of the scoring
    // path (three split passes, f32 softmax; it also hands out the log-sum-exp rows the backward recomputes P from)
    __half* qp = reinterpret_cast<__half*>(c.ws.qkvp);
    AOp planes{qp, qp + (size_t)T * 3 * H};
    RC(gemm_nt(c, qkv_in, t->p(t->li(l, LTR_WL_QKV_W)), t->p(t->li(l, LTR_WL_QKV_B)), nullptr, L.qkv, T, 3 * H, H, 0, &planes));
    static const bool f32_attn = [] { const char* e = getenv("LTR_TRAIN_F32_ATTN"); return e && e[0] == '1'; }();   // A/B
    if (planes.hi && !f32_attn) {
      c.attn_mfma = true;
      __half* op = reinterpret_cast<__half*>(c.ws.aop);
      // (own work list: the MFMA kernel walks 128-query blocks, the backward kernels 64-query blocks from ws.blk)
      if (l == 0) RC(launch_attention_blocks(cu, N, 64, c.ws.blk, s));
      RC(launch_attention(LTR_W_F16, planes, cu, N, T, H, d.num_heads, c.ws.blk2, AOp{op, op + (size_t)T * H}, l == 0, s, L.lse));
      const size_t n8 = (size_t)T * H / 8;
      planes_to_f32_kernel<<<(unsigned)((n8 + 255) / 256), 256, 0, s>>>(op, op + (size_t)T * H, L.ao, n8);
      LTR_LAUNCH_CHECK();
    } else {
      RC(launch_attention(LTR_W_F32, AOp{L.qkv, nullptr}, cu, N, T, H, d.num_heads, c.ws.blk, AOp{L.ao, nullptr}, l == 0, s, L.lse));
    }
    // s1 = x0 + dropout(out_proj(ao))
    float* s1 = d.pre_ln ? L.mid : L.n1;
    if (drop) {
      RC(gemm_nt(c, L.ao, t->p(t->li(l, LTR_WL_OUT_W)), t->p(t->li(l, LTR_WL_OUT_B)), nullptr, L.ao_raw, T, H, H, 0));
      dropout_add_kernel<<<1024, 256, 0, s>>>(L.ao_raw, L.x0, s1, (size_t)T * H, drop_seed(t, l, 0), t->cfg.dropout);
      LTR_LAUNCH_CHECK();
    } else {
      RC(gemm_nt(c, L.ao, t->p(t->li(l, LTR_WL_OUT_W)), t->p(t->li(l, LTR_WL_OUT_B)), L.x0, s1, T, H, H, 0));
    }
    const float* mlp_in;
    if (d.pre_ln) {
      RC(launch_layernorm(LTR_W_F32, L.mid, t->p(t->li(l, LTR_WL_LN2_W)), t->p(t->li(l, LTR_WL_LN2_B)), T, H, L.n2, AOp{nullptr, nullptr}, s));
      mlp_in = L.n2;
    } else {
      RC(launch_layernorm(LTR_W_F32, L.n1, t->p(t->li(l, LTR_WL_LN1_W)), t->p(t->li(l, LTR_WL_LN1_B)), T, H, L.mid, AOp{nullptr, nullptr}, s));
      mlp_in = L.mid;
    }
    RC(gemm_nt(c, mlp_in, t->p(t->li(l, LTR_WL_FC1_W)), t->p(t->li(l, LTR_WL_FC1_B)), nullptr, L.f, T, F, H, 1));
    float* s2 = d.pre_ln ? out : L.n2;
    if (drop) {
      RC(gemm_nt(c, L.f, t->p(t->li(l, LTR_WL_FC2_W)), t->p(t->li(l, LTR_WL_FC2_B)), nullptr, L.mlp_raw, T, H, F, 0));
      dropout_add_kernel<<<1024, 256, 0, s>>>(L.mlp_raw, L.mid, s2, (size_t)T * H, drop_seed(t, l, 1), t->cfg.dropout);
      LTR_LAUNCH_CHECK();
    } else {
      RC(gemm_nt(c, L.f, t->p(t->li(l, LTR_WL_FC2_W)), t->p(t->li(l, LTR_WL_FC2_B)), L.mid, s2, T, H, F, 0));
    }
    if (!d.pre_ln)
      RC(launch_layernorm(LTR_W_F32, L.n2, t->p(t->li(l, LTR_WL_LN2_W)), t->p(t->li(l, LTR_WL_LN2_B)), T, H, out, AOp{nullptr, nullptr}, s));
  }
  // head: last-token rows -> (final LN) -> (project_out) -> score.weight
  const int Np = (N + 15) / 16 * 16;
  LTR_HIP_CHECK(hipMemsetAsync(c.ws.hl, 0, (size_t)Np * H * 4, s));
  gather_rows_kernel<<<N, 256, 0, s>>>(c.ws.hfin, cu, N, H, c.ws.hl);
  LTR_LAUNCH_CHECK();
  const float* z = c.ws.hl;
  if (d.pre_ln) {
    RC(launch_layernorm(LTR_W_F32, c.ws.hl, t->p(LTR_WT_FINAL_LN_W), t->p(LTR_WT_FINAL_LN_B), N, H, c.ws.z, AOp{nullptr, nullptr}, s));
    z = c.ws.z;
  }
  const float* y = z;
  if (proj) { RC(gemm_nt(c, z, t->p(LTR_WT_PROJECT_OUT), nullptr, nullptr, c.ws.y, N, De, H, 0)); y = c.ws.y; }
  head_fwd_kernel<<<(unsigned)(((size_t)N * nl + 3) / 4), 256, 0, s>>>(y, t->p(LTR_WT_SCORE), N, De, nl, c.ws.logits);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

int backward(Ctx& c, const int64_t* ids, const int32_t* cu) {
  ltr_trainer* t = c.t;
  const ltr_model_desc& d = t->d;
  const int H = d.hidden_size, F = d.ffn_dim, De = d.word_embed_proj_dim, T = c.T, N = c.N, nl = d.num_labels;
  const bool proj = De != H, drop = t->cfg.dropout > 0.f;
  hipStream_t s = c.s;
  const float scale = 0.125f;
  // head
  const float* z = d.pre_ln ? c.ws.z : c.ws.hl;
  const float* y = proj ? c.ws.y : z;
  head_bwd_dw_kernel<<<(unsigned)(((size_t)nl * De + 255) / 256), 256, 0, s>>>(c.ws.dlogits, y, N, De, nl, t->g(LTR_WT_SCORE));
  LTR_LAUNCH_CHECK();
  const int Np = (N + 15) / 16 * 16;
  LTR_HIP_CHECK(hipMemsetAsync(c.ws.dy, 0, (size_t)Np * De * 4, s));
  head_bwd_dy_kernel<<<(unsigned)(((size_t)N * De + 255) / 256), 256, 0, s>>>(c.ws.dlogits, t->p(LTR_WT_SCORE), N, De, nl, c.ws.dy);
  LTR_LAUNCH_CHECK();
  const float* dz = c.ws.dy;
  if (proj) {
    float* sdy = grad_amax(c, c.ws.dy, (size_t)N * De);
    RC(gemm_tn(c, c.ws.dy, z, t->g(LTR_WT_PROJECT_OUT), N, De, H, sdy));
    RC(gemm_nn(c, c.ws.dy, t->p(LTR_WT_PROJECT_OUT), nullptr, c.ws.dz, N, De, H, sdy));
    dz = c.ws.dz;
  }
  const float* dhl = dz;
  if (d.pre_ln) {
    RC(ln_bwd(c, c.ws.hl, t->p(LTR_WT_FINAL_LN_W), dz, nullptr, N, H, c.ws.dhl, t->g(LTR_WT_FINAL_LN_W), t->g(LTR_WT_FINAL_LN_B)));
    dhl = c.ws.dhl;
  }
  LTR_HIP_CHECK(hipMemsetAsync(c.ws.dh, 0, (size_t)T * H * 4, s));
  scatter_rows_kernel<<<N, 256, 0, s>>>(dhl, cu, N, H, c.ws.dh);
  LTR_LAUNCH_CHECK();
  float* dh = c.ws.dh;
  int4* blk_desc = reinterpret_cast<int4*>(c.ws.blk + ((N + 1 + 3) & ~3));
  for (int l = d.num_layers - 1; l >= 0; --l) {
    auto& L = c.ws.L[l];
    auto P = [&](int i) { return t->p(t->li(l, i)); };
    auto G = [&](int i) { return t->g(t->li(l, i)); };
    const float* mlp_in = d.pre_ln ? L.n2 : L.mid;
    const float* qkv_in = d.pre_ln ? L.n1 : L.x0;
    // ---- MLP half.  ds2 = gradient w.r.t. s2 = mid + dropout(mlp)
    float* ds2 = dh;
    if (!d.pre_ln) {       // out = LN2(s2)
      RC(ln_bwd(c, L.n2, P(LTR_WL_LN2_W), dh, nullptr, T, H, c.ws.dsmall, G(LTR_WL_LN2_W), G(LTR_WL_LN2_B)));
      ds2 = c.ws.dsmall;
    }
    const float* dmlp = ds2;
    if (drop) {
      dropout_add_kernel<<<1024, 256, 0, s>>>(ds2, nullptr, c.ws.dsmall2, (size_t)T * H, drop_seed(t, l, 1), t->cfg.dropout);
      LTR_LAUNCH_CHECK();
      dmlp = c.ws.dsmall2;
    }
    float* s_dmlp = grad_amax(c, dmlp, (size_t)T * H);
    RC(gemm_tn(c, dmlp, L.f, G(LTR_WL_FC2_W), T, H, F, s_dmlp));
    RC(colsum(c, dmlp, nullptr, T, H, G(LTR_WL_FC2_B)));
    RC(gemm_nn(c, dmlp, P(LTR_WL_FC2_W), nullptr, c.ws.dbig, T, H, F, s_dmlp));   // df [T, F]
    relu_bwd_kernel<<<1024, 256, 0, s>>>(c.ws.dbig, L.f, (size_t)T * F);
    LTR_LAUNCH_CHECK();
    float* s_df = grad_amax(c, c.ws.dbig, (size_t)T * F);
    RC(gemm_tn(c, c.ws.dbig, mlp_in, G(LTR_WL_FC1_W), T, F, H, s_df));
    RC(colsum(c, c.ws.dbig, nullptr, T, F, G(LTR_WL_FC1_B)));
    // ---- dmid = ds2 + d(mlp_in -> mid)
    float* dmid;
    if (d.pre_ln) {        // mlp_in = LN2(mid): dmid = ds2 + LN2_bwd(da2)
      RC(gemm_nn(c, c.ws.dbig, P(LTR_WL_FC1_W), nullptr, c.ws.dsmall, T, F, H, s_df));   // da2 (pre-LN has not used dsmall so far)
      RC(ln_bwd(c, L.mid, P(LTR_WL_LN2_W), c.ws.dsmall, ds2, T, H, dh, G(LTR_WL_LN2_W), G(LTR_WL_LN2_B)));
      dmid = dh;
    } else {               // mlp_in = mid
      RC(gemm_nn(c, c.ws.dbig, P(LTR_WL_FC1_W), ds2, dh, T, F, H, s_df));
      dmid = dh;
    }
    // ---- attention half.  ds1 = gradient w.r.t. s1 = x0 + dropout(out_proj(ao))
    float* ds1 = dmid;
    if (!d.pre_ln) {       // mid = LN1(s1)
      RC(ln_bwd(c, L.n1, P(LTR_WL_LN1_W), dmid, nullptr, T, H, c.ws.dsmall, G(LTR_WL_LN1_W), G(LTR_WL_LN1_B)));
      ds1 = c.ws.dsmall;
    }
    const float* dproj = ds1;
    if (drop) {
      dropout_add_kernel<<<1024, 256, 0, s>>>(ds1, nullptr, c.ws.dsmall2, (size_t)T * H, drop_seed(t, l, 0), t->cfg.dropout);
      LTR_LAUNCH_CHECK();
      dproj = c.ws.dsmall2;
    }
    float* s_dproj = grad_amax(c, dproj, (size_t)T * H);
    RC(gemm_tn(c, dproj, L.ao, G(LTR_WL_OUT_W), T, H, H, s_dproj));
    RC(colsum(c, dproj, nullptr, T, H, G(LTR_WL_OUT_B)));
    float* dao = c.ws.xhd;                                                           // free between LN backward calls
    RC(gemm_nn(c, dproj, P(LTR_WL_OUT_W), nullptr, dao, T, H, H, s_dproj));
    static const bool f32_attn_bwd = [] { const char* e = getenv("LTR_TRAIN_F32_ATTN"); return e && e[0] == '1'; }();   // A/B
    if (c.attn_mfma && !f32_attn_bwd) {
      // split-fp16 MFMA backward over the forward's 128-row work list (ws.blk2); dO is scaled into fp16's range from
      // max|dO| and max|qkv| (ltr_attn.hip), both taken on the stream
      float* s_dao = grad_amax(c, dao, (size_t)T * H);
      float* s_qkv = amax_of(c, L.qkv, (size_t)T * 3 * H, true);
      if (!s_dao || !s_qkv) return LTR_E_INVAL;
      RC(launch_attention_bwd(L.qkv, L.ao, dao, L.lse, s_dao, s_qkv, c.ws.blk2, N, T, H, d.num_heads, scale, c.ws.qkvp, c.ws.aop,
                              c.ws.Dq, c.ws.dbig, s));
    } else {
      dim3 grid(T / BQ + N, d.num_heads);
      attn_bwd_dq_kernel<<<grid, 64 * BW, 0, s>>>(L.qkv, L.ao, dao, L.lse, c.ws.blk, blk_desc, N, H, scale, c.ws.dbig, c.ws.Dq);
      LTR_LAUNCH_CHECK();
      attn_bwd_dkv_kernel<<<grid, 64 * BW, 0, s>>>(L.qkv, dao, L.lse, c.ws.Dq, c.ws.blk, blk_desc, N, H, scale, c.ws.dbig);
      LTR_LAUNCH_CHECK();
    }
    float* s_dqkv = grad_amax(c, c.ws.dbig, (size_t)T * 3 * H);
    RC(gemm_tn(c, c.ws.dbig, qkv_in, G(LTR_WL_QKV_W), T, 3 * H, H, s_dqkv));
    RC(colsum(c, c.ws.dbig, nullptr, T, 3 * H, G(LTR_WL_QKV_B)));
    if (d.pre_ln) {        // qkv_in = LN1(x0): dx0 = ds1 + LN1_bwd
      float* da1 = c.ws.dsmall2;
      RC(gemm_nn(c, c.ws.dbig, P(LTR_WL_QKV_W), nullptr, da1, T, 3 * H, H, s_dqkv));
      RC(ln_bwd(c, L.x0, P(LTR_WL_LN1_W), da1, ds1, T, H, dh, G(LTR_WL_LN1_W), G(LTR_WL_LN1_B)));
    } else {               // qkv_in = x0
      RC(gemm_nn(c, c.ws.dbig, P(LTR_WL_QKV_W), ds1, dh, T, 3 * H, H, s_dqkv));
    }
  }
  // embedding: h0 = project_in(tok) + pos
  const float* dtok = dh;
  if (proj) {
    float* s_dh = grad_amax(c, dh, (size_t)T * H);
    RC(gemm_tn(c, dh, c.ws.tok, t->g(LTR_WT_PROJECT_IN), T, H, De, s_dh));
    RC(gemm_nn(c, dh, t->p(LTR_WT_PROJECT_IN), nullptr, c.ws.dsmall, T, H, De, s_dh));
    dtok = c.ws.dsmall;
  }
  embed_bwd_kernel<<<T, 256, 0, s>>>(ids, cu, N, T, dh, dtok, H, De, d.vocab_size, d.pos_rows, t->g(LTR_WT_EMBED_POS),
                                     t->g(LTR_WT_EMBED_TOKENS));
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // namespace

extern "C" {

int ltr_train_create(const ltr_model_desc* desc, const void* const* weights, int32_t n_weights,
                     const ltr_train_config* cfg, void* stream, ltr_train_handle* out) {
  if (!desc || !weights || !cfg || !out) { set_error("ltr_train_create: NULL argument"); return LTR_E_INVAL; }
  const ltr_model_desc& d = *desc;
  if (d.hidden_size <= 0 || d.hidden_size != d.num_heads * 64 || d.ffn_dim % 64 || d.word_embed_proj_dim % 64 ||
      d.num_labels < 1 || d.num_layers < 0 || d.pos_rows < 3 || d.vocab_size < 1) {
    set_error("ltr_train_create: head size must be 64; F and De multiples of 64");
    return LTR_E_INVAL;
  }
  if (d.hidden_size > 256 * EB_COLS || d.word_embed_proj_dim > 256 * EB_COLS) {
    set_error("ltr_train_create: hidden size / embedding width above %d (embed_bwd_kernel keeps a table row in registers)", 256 * EB_COLS);
    return LTR_E_INVAL;
  }
  if (cfg->loss != LTR_LOSS_LISTMLE && cfg->loss != LTR_LOSS_MSE && cfg->loss != LTR_LOSS_CROSSENTROPY && cfg->loss != LTR_LOSS_NEURALNDCG) {
    set_error("ltr_train_create: unknown loss %d", cfg->loss); return LTR_E_INVAL;
  }
  if (cfg->loss != LTR_LOSS_CROSSENTROPY && d.num_labels != 1) {
    set_error("ltr_train_create: listMLE / neuralNDCG / mse train a 1-label (rank) head (prefill_predictor.py:35-36)"); return LTR_E_INVAL;
  }
  if (!(cfg->dropout >= 0.f && cfg->dropout < 1.f)) { set_error("ltr_train_create: dropout must be in [0, 1)"); return LTR_E_INVAL; }
  if (cfg->precision < LTR_TRAIN_PREC_DEFAULT || cfg->precision > LTR_TRAIN_PREC_F32) {
    set_error("ltr_train_create: precision %d is not one of LTR_TRAIN_PREC_*", cfg->precision); return LTR_E_INVAL;
  }
  const int want = LTR_WT_GLOBAL_COUNT + d.num_layers * LTR_WL_COUNT;
  if (n_weights != want) { set_error("ltr_train_create: %d weight pointers, expected %d", n_weights, want); return LTR_E_INVAL; }
  ltr_trainer* t = new (std::nothrow) ltr_trainer();
  if (!t) { set_error("ltr_train_create: out of host memory"); return LTR_E_NOMEM; }
  t->d = d; t->cfg = *cfg;
  if (cfg->precision == LTR_TRAIN_PREC_DEFAULT) { const char* e = getenv("LTR_TRAIN_F32"); t->use_f32 = e && e[0] == '1'; }
  else t->use_f32 = cfg->precision == LTR_TRAIN_PREC_F32;
  t->off.resize(want); t->cnt.resize(want);
  size_t total = 0, wmax = 0;
  for (int i = 0; i < want; ++i) {
    t->cnt[i] = weight_count(d, i);
    t->off[i] = total;
    total += (t->cnt[i] + 63) / 64 * 64;
    if (t->cnt[i] && !weights[i]) { delete t; set_error("ltr_train_create: weight pointer %d is NULL", i); return LTR_E_INVAL; }
    const int li = i < LTR_WT_GLOBAL_COUNT ? -1 : (i - LTR_WT_GLOBAL_COUNT) % LTR_WL_COUNT;
    const bool mat = i == LTR_WT_PROJECT_IN || i == LTR_WT_PROJECT_OUT || li == LTR_WL_QKV_W || li == LTR_WL_OUT_W ||
                     li == LTR_WL_FC1_W || li == LTR_WL_FC2_W;
    if (mat) wmax = std::max(wmax, t->cnt[i]);
  }
  t->total = total;
  {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, weights[LTR_WT_EMBED_TOKENS]) != hipSuccess) {
      (void)hipGetLastError(); delete t; set_error("ltr_train_create: weight pointers must be device memory"); return LTR_E_INVAL;
    }
    t->device = attr.device;
  }
  DevGuard guard(t->device);
  hipStream_t s = (hipStream_t)stream;
  t->wT_elems = wmax;
  if (hipMalloc((void**)&t->P, total * 4) != hipSuccess || hipMalloc((void**)&t->G, total * 4) != hipSuccess ||
      hipMalloc((void**)&t->M1, total * 4) != hipSuccess || hipMalloc((void**)&t->V1, total * 4) != hipSuccess ||
      hipMalloc((void**)&t->wT, std::max<size_t>(wmax, 64) * 4) != hipSuccess) {
    delete t; set_error("ltr_train_create: cannot allocate %zu parameters x 4 buffers", total); return LTR_E_NOMEM;
  }
  if (hipMemsetAsync(t->P, 0, total * 4, s) != hipSuccess || hipMemsetAsync(t->M1, 0, total * 4, s) != hipSuccess ||
      hipMemsetAsync(t->V1, 0, total * 4, s) != hipSuccess) { delete t; set_error("ltr_train_create: memset failed"); return LTR_E_HIP; }
  for (int i = 0; i < want; ++i)
    if (t->cnt[i] && hipMemcpyAsync(t->P + t->off[i], weights[i], t->cnt[i] * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      delete t; set_error("ltr_train_create: copying weight %d failed", i); return LTR_E_HIP;
    }
  if (hipStreamSynchronize(s) != hipSuccess) { delete t; set_error("ltr_train_create: copy failed"); return LTR_E_HIP; }
  *out = t;
  return LTR_OK;
}

int ltr_train_destroy(ltr_train_handle h) {
  if (h) { DevGuard guard(h->device); delete h; }
  return LTR_OK;
}

size_t ltr_train_workspace_bytes(ltr_train_handle h, int64_t N, int64_t T) {
  if (!h || N <= 0 || T <= 0) return 0;
  return carve_train(h->d, T, N, nullptr, h->cfg.dropout > 0.f, h->cfg.loss == LTR_LOSS_NEURALNDCG).bytes;
}

int ltr_train_read(ltr_train_handle h, int32_t index, int32_t what, float* dst, size_t capacity, size_t* count_out,
                   void* stream) {
  if (!h || index < 0 || index >= (int)h->off.size() || !count_out || (what != 0 && what != 1)) {
    set_error("ltr_train_read: bad argument"); return LTR_E_INVAL;
  }
  *count_out = h->cnt[index];
  if (!dst || h->cnt[index] == 0) return LTR_OK;           // size query / absent tensor
  if (capacity < h->cnt[index]) { set_error("ltr_train_read: %zu floats do not fit %zu", h->cnt[index], capacity); return LTR_E_NOMEM; }
  DevGuard guard(h->device);
  const float* src = (what == 0 ? h->P : h->G) + h->off[index];
  LTR_HIP_CHECK(hipMemcpyAsync(dst, src, h->cnt[index] * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return LTR_OK;
}

int ltr_train_step(ltr_train_handle h, const int64_t* token_ids, const int32_t* cu_seqlens, const int32_t* cu_seqlens_host,
                   int32_t N, int32_t T, const float* labels, const int32_t* shuffle, int32_t apply_update, float* loss_out,
                   float* logits_out, void* workspace, size_t ws_bytes, void* stream) {
  if (!h || !token_ids || !cu_seqlens || !cu_seqlens_host || (apply_update >= 0 && (!labels || !loss_out)) || !workspace || N <= 0 || T <= 0) {
    set_error("ltr_train_step: bad argument"); return LTR_E_INVAL;
  }
  const ltr_model_desc& d = h->d;
  if (cu_seqlens_host[0] != 0 || cu_seqlens_host[N] != T) { set_error("ltr_train_step: cu_seqlens does not span T"); return LTR_E_INVAL; }
  for (int r = 0; r < N; ++r) {
    const int L = cu_seqlens_host[r + 1] - cu_seqlens_host[r];
    if (L <= 0 || L > d.pos_rows - 2) { set_error("ltr_train_step: request %d has %d tokens (1..%d allowed)", r, L, d.pos_rows - 2); return LTR_E_INVAL; }
  }
  if (apply_update >= 0 && h->cfg.loss == LTR_LOSS_LISTMLE && (!shuffle || N > 4096)) { set_error("ltr_train_step: listMLE needs the shuffle permutation and a slate of at most 4096"); return LTR_E_INVAL; }
  if (apply_update >= 0 && h->cfg.loss == LTR_LOSS_NEURALNDCG && (N < 2 || ltr_neuralndcg_workspace_bytes(1, N) == 0)) {
    set_error("ltr_train_step: neuralNDCG takes a slate of 2..1024 prompts (one item: the reference raises IndexError, loss_utils.py:70)");
    return LTR_E_INVAL;
  }
  DevGuard guard(h->device);
  hipStream_t s = (hipStream_t)stream;
  Ctx c{h, s, carve_train(d, T, N, workspace, h->cfg.dropout > 0.f, h->cfg.loss == LTR_LOSS_NEURALNDCG), T, N};
  c.eval = apply_update < 0;
  if (!h->use_f32 && c.ws.bytes <= ws_bytes) LTR_HIP_CHECK(hipMemsetAsync(c.ws.scales, 0, MAX_GEMMS * 2 * sizeof(float), s));
  if (c.ws.bytes > ws_bytes) { set_error("ltr_train_step: workspace too small (%zu < %zu)", ws_bytes, c.ws.bytes); return LTR_E_NOMEM; }
  int rc = forward(c, token_ids, cu_seqlens);
  if (rc) return rc;
  const int nl = d.num_labels;
  if (logits_out) LTR_HIP_CHECK(hipMemcpyAsync(logits_out, c.ws.logits, (size_t)N * nl * 4, hipMemcpyDeviceToDevice, s));
  if (c.eval) return LTR_OK;                    // evaluation pass (trainer.py:171-190): the outputs are all that is wanted
  if (h->cfg.loss == LTR_LOSS_LISTMLE) {       // trainer.py:157: loss_func(outputs.view(1, -1), labels) - the batch is one slate
    rc = ltr_listmle(c.ws.logits, labels, shuffle, 1, N, h->cfg.listmle_eps, h->cfg.pad_value, loss_out, c.ws.row_loss,
                     c.ws.dlogits, stream);
    if (rc) return rc;
  } else if (h->cfg.loss == LTR_LOSS_NEURALNDCG) {   // trainer.py:127-128,157: neuralNDCG(outputs.view(1, -1), labels), every keyword at its default
    rc = ltr_neuralndcg(c.ws.logits, labels, 1, N, 1.f, 0, h->cfg.pad_value, loss_out, c.ws.row_loss, c.ws.dlogits, c.ws.ndcg,
                        c.ws.ndcg_bytes, stream);
    if (rc) return rc;
  } else if (h->cfg.loss == LTR_LOSS_MSE) {
    mse_kernel<<<1, 256, 0, s>>>(c.ws.logits, labels, N, loss_out, c.ws.dlogits);
    LTR_LAUNCH_CHECK();
  } else {
    ce_rows_kernel<<<(N + 3) / 4, 256, 0, s>>>(c.ws.logits, labels, N, nl, c.ws.row_loss, c.ws.dlogits);
    LTR_LAUNCH_CHECK();
    mean_kernel<<<1, 64, 0, s>>>(c.ws.row_loss, N, loss_out);
    LTR_LAUNCH_CHECK();
  }
  LTR_HIP_CHECK(hipMemsetAsync(h->G, 0, h->total * 4, s));        // optimizer.zero_grad() (trainer.py:165)
  if ((rc = backward(c, token_ids, cu_seqlens))) return rc;
  h->step += 1;
  if (apply_update) {                                                // optimizer.step() (trainer.py:163)
    h->adam_step += 1;                                               // gradient-only calls do not advance Adam's t
    const double b1 = h->cfg.beta1, b2 = h->cfg.beta2;
    const float bc1 = (float)(1.0 - std::pow(b1, (double)h->adam_step)), bc2 = (float)(1.0 - std::pow(b2, (double)h->adam_step));
    adam_kernel<<<2048, 256, 0, s>>>(h->P, h->G, h->M1, h->V1, h->total, h->cfg.lr, h->cfg.beta1, h->cfg.beta2, h->cfg.eps,
                                     h->cfg.weight_decay, bc1, bc2);
    LTR_LAUNCH_CHECK();
  }
  return LTR_OK;
}


namespace {
struct AttnWs { __half *qp, *op; float *lse, *Dq, *slots; int32_t* blk; size_t bytes; };
AttnWs carve_attn(int nh, int64_t N, int64_t T, void* base) {
  const size_t H = (size_t)nh * D;
  char* p = (char*)base;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t q = o; o += up(bytes); return base ? p + q : (char*)nullptr; };
  AttnWs w;
  w.qp = (__half*)take((size_t)T * 3 * H * 4);            // hi | lo planes of qkv
  w.op = (__half*)take((size_t)T * H * 4);                // hi | lo planes of out, then of dout
  w.lse = (float*)take((size_t)T * nh * 4);
  w.Dq = (float*)take((size_t)T * nh * 4);
  w.slots = (float*)take(64);
  w.blk = (int32_t*)take(((size_t)(N + 4) + ((size_t)T / 64 + N + 1) * 4) * 4);
  w.bytes = o;
  return w;
}
}  // namespace

size_t ltr_train_attention_workspace_bytes(int32_t num_heads, int64_t N, int64_t T) {
  return carve_attn(num_heads, N, T, nullptr).bytes;
}

int ltr_train_attention(int32_t num_heads, const float* qkv, const float* dout, const int32_t* cu_seqlens, int32_t N, int32_t T,
                        float* out, float* dqkv, void* workspace, size_t ws_bytes, void* stream) {
  if (num_heads <= 0 || N < 0 || T < 0) { set_error("ltr_train_attention: bad argument"); return LTR_E_INVAL; }
  if (N == 0 || T == 0) return LTR_OK;
  if (!qkv || !dout || !cu_seqlens || !out || !dqkv || !workspace) { set_error("ltr_train_attention: NULL pointer"); return LTR_E_INVAL; }
  const AttnWs w = carve_attn(num_heads, N, T, workspace);
  if (ws_bytes < w.bytes) { set_error("ltr_train_attention: workspace too small (%zu < %zu)", ws_bytes, w.bytes); return LTR_E_NOMEM; }
  hipStream_t s = (hipStream_t)stream;
  const int H = num_heads * D;
  // forward exactly as ltr_train_step runs it: planes of qkv -> MFMA attention (+ log-sum-exp rows) -> f32 out
  LTR_HIP_CHECK(hipMemsetAsync(w.slots, 0, 64, s));
  int rc = launch_attention_bwd_planes(qkv, (size_t)T * 3 * H, w.qp, s);
  if (rc) return rc;
  AOp planes{w.qp, w.qp + (size_t)T * 3 * H}, op{w.op, w.op + (size_t)T * H};
  if ((rc = launch_attention(LTR_W_F16, planes, cu_seqlens, N, T, H, num_heads, w.blk, op, 1, s, w.lse))) return rc;
  const size_t n8 = (size_t)T * H / 8;
  planes_to_f32_kernel<<<(unsigned)((n8 + 255) / 256), 256, 0, s>>>((const __half*)op.hi, (const __half*)op.lo, out, n8);
  LTR_LAUNCH_CHECK();
  amax_kernel<<<(unsigned)std::min<size_t>(((size_t)T * H + 4095) / 4096, AMAX_BLOCKS), 256, 0, s>>>(dout, (size_t)T * H, w.slots);
  amax_kernel<<<(unsigned)std::min<size_t>(((size_t)T * 3 * H + 4095) / 4096, AMAX_BLOCKS), 256, 0, s>>>(qkv, (size_t)T * 3 * H, w.slots + 1);
  LTR_LAUNCH_CHECK();
  return launch_attention_bwd(qkv, out, dout, w.lse, w.slots, w.slots + 1, w.blk, N, T, H, num_heads, 0.125f, w.qp, w.op, w.Dq,
                              dqkv, s);
}

}  // extern "C"
