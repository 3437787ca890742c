// Varlen causal self-attention of the predictor's prefill on gfx950.
//
// Reference: OPTAttention.forward (opt.py:92-102) with kv_cache=None, i.e. per-sequence
// softmax(q k^T * d^-0.5, causal) v over heads of 64 (rocm_flash_attn.py:244-290 on GPU,
// torch_sdpa.py:138-178 on CPU).  Input is the packed QKV GEMM output f32 [T, 3H]
// (q | k | v); output is the out_proj GEMM operand [T, H].
//
// Query blocks of all requests are enumerated through a device-side prefix table so the
// launch needs no host knowledge of the lengths.
//
// F16 ("split") mode - attn_f16s_kernel, the production kernel.  q, k, v arrive as fp16
// hi|lo planes written by the QKV GEMM epilogue.  A workgroup is 4 waves = 128 consecutive
// queries of one (request, head), 32 per wave; K/V tiles of 32 keys are staged through LDS
// (next tile prefetched into registers under the MFMAs).  Per 32x32 (key, query) block:
//   S^T = K Q^T      on v_mfma_f32_32x32x16_f16, 3 passes (kh*qh + kh*ql + kl*qh)
//   online softmax   lane-local: with the S^T layout a lane holds 16 keys of ONE query
//                    (col = lane & 31), so row max / sum are 15 in-lane ops + one
//                    lane<->lane^32 exchange, and the rescale factor is a per-lane scalar
//   O^T += V^T P^T   3 passes (vh*ph + vh*pl + vl*ph); the S^T accumulator registers,
//                    converted to fp16, ARE the B operand (the key <-> k-slot assignment
//                    of an MFMA is arbitrary as long as A and B agree, so V^T is staged
//                    in LDS with its keys permuted to match: no cross-lane shuffle).
// The dropped lo*lo terms are ~2^-22 relative: f32-grade, as in the GEMMs.
//
// F32 mode - attn_f32_kernel: exact f32 VALU flash kernel, one query row per lane.
#include "ltr_internal.h"

namespace ltr {
namespace {

constexpr int QB = 64;     // queries per workgroup
constexpr int KT = 32;     // keys per LDS tile
constexpr int D = 64;      // head size (OPT-125m/350m: 768/12 = 1024/16 = 64)
constexpr float NEG = -1.0e30f;

// blk_start[i] = sum_{j<i} ceil(L_j / qb), i in [0, n_req]
__global__ void __launch_bounds__(1024) attn_blocks_kernel(const int32_t* __restrict__ cu, int n_req, int qb,
                                                           int32_t* __restrict__ blk_start) {
  __shared__ int s_w[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { carry = 0; blk_start[0] = 0; }
  __syncthreads();
  for (int base = 0; base < n_req; base += 1024) {
    int i = base + tid;
    int v = (i < n_req) ? (cu[i + 1] - cu[i] + qb - 1) / qb : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    if (lane == 63) s_w[wave] = v;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; ++w) off += s_w[w];
    v += off;
    if (i < n_req) blk_start[i + 1] = v;
    __syncthreads();
    if (tid == 1023) carry = v;
    __syncthreads();
  }
}

template <bool SPLIT>
__global__ void __launch_bounds__(64) attn_f32_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                      const int32_t* __restrict__ blk_start, int n_req, int H,
                                                      float scale_log2e, void* out_hi, void* out_lo) {
  __shared__ __attribute__((aligned(16))) float s_k[KT * D];
  __shared__ __attribute__((aligned(16))) float s_v[KT * D];
  const int b = blockIdx.x;
  if (b >= blk_start[n_req]) return;
  const int head = blockIdx.y;
  const int lane = threadIdx.x;
  const int r = find_request(blk_start, n_req, b);
  const int q0 = (b - blk_start[r]) * QB;
  const int t0 = cu[r] - cu[0];
  const int L = cu[r + 1] - cu[r];
  const int qi = q0 + lane;
  const bool valid = qi < L;
  const size_t ld = (size_t)3 * H;

  float q[D], o[D];
  {
    const float* qp = qkv + (size_t)(t0 + (valid ? qi : 0)) * ld + head * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float4 v = *reinterpret_cast<const float4*>(qp + d);
      q[d] = v.x * scale_log2e; q[d + 1] = v.y * scale_log2e;
      q[d + 2] = v.z * scale_log2e; q[d + 3] = v.w * scale_log2e;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
  }
  float m = NEG, l = 0.f;
  const int kend = min(L, q0 + QB);   // causal: keys [0, kend)
  for (int kt = 0; kt < kend; kt += KT) {
    __syncthreads();
    // stage K and V rows kt..kt+KT-1 (clamped): 16 lanes x float4 cover one 256-B row
#pragma unroll
    for (int it = 0; it < KT * D / 4 / 64; ++it) {
      const int idx = it * 64 + lane;
      const int key = idx >> 4, d4 = (idx & 15) * 4;
      const int kk = min(kt + key, L - 1);
      const float* base = qkv + (size_t)(t0 + kk) * ld + head * D + d4;
      *reinterpret_cast<float4*>(s_k + key * D + d4) = *reinterpret_cast<const float4*>(base + H);
      *reinterpret_cast<float4*>(s_v + key * D + d4) = *reinterpret_cast<const float4*>(base + 2 * H);
    }
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < KT; sub += 16) {
      if (kt + sub > q0 + QB - 1 || kt + sub >= kend) break;   // wave-uniform
      float s[16];
      float mx = NEG;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float* kr = s_k + (sub + j) * D;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
          float4 kv = *reinterpret_cast<const float4*>(kr + d);
          acc = fmaf(q[d], kv.x, acc); acc = fmaf(q[d + 1], kv.y, acc);
          acc = fmaf(q[d + 2], kv.z, acc); acc = fmaf(q[d + 3], kv.w, acc);
        }
        const int kj = kt + sub + j;
        s[j] = (kj <= qi && kj < L) ? acc : NEG;
        mx = fmaxf(mx, s[j]);
      }
      const float m_new = fmaxf(m, mx);
      const float alpha = exp2f(m - m_new);
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] *= alpha;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float p = exp2f(s[j] - m_new);
        l += p;
        const float* vr = s_v + (sub + j) * D;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
          float4 vv = *reinterpret_cast<const float4*>(vr + d);
          o[d] = fmaf(p, vv.x, o[d]); o[d + 1] = fmaf(p, vv.y, o[d + 1]);
          o[d + 2] = fmaf(p, vv.z, o[d + 2]); o[d + 3] = fmaf(p, vv.w, o[d + 3]);
        }
      }
    }
  }
  if (!valid) return;
  const float inv = 1.f / l;
  const size_t ob = (size_t)(t0 + qi) * H + head * D;
  if (SPLIT) {
    __half* ph = (__half*)out_hi + ob;
    __half* pl = (__half*)out_lo + ob;
#pragma unroll
    for (int d = 0; d < D; d += 8) {
      __half hh[8], ll[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) split_f16(o[d + i] * inv, hh[i], ll[i]);
      *reinterpret_cast<uint4*>(ph + d) = *reinterpret_cast<const uint4*>(hh);
      *reinterpret_cast<uint4*>(pl + d) = *reinterpret_cast<const uint4*>(ll);
    }
  } else {
    float* po = (float*)out_hi + ob;
#pragma unroll
    for (int d = 0; d < D; d += 4)
      *reinterpret_cast<float4*>(po + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
  }
}

// ------------------------------------------------------------------------------------
// split-fp16 MFMA kernel
// ------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TK = 32;   // keys per tile = one 32x32 MFMA block

// K tile [32 keys][64 d] halves, 128-B rows of 8 16-B chunks; chunk c of row r lives at
// c ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-B slots.
__device__ __forceinline__ int k_off(int row, int c) { return row * D + ((c ^ ((row >> 1) & 7)) << 3); }
// V^T tile [64 d][32 key slots] halves, 64-B rows of 4 chunks; chunk c of row r at c ^ ((r >> 2) & 3).
__device__ __forceinline__ int v_off(int row, int c) { return row * TK + ((c ^ ((row >> 2) & 3)) << 3); }

template <int NW>
__global__ void __launch_bounds__(NW * 64) attn_f16s_kernel(
    const __half* __restrict__ qkv_hi, const __half* __restrict__ qkv_lo, const int32_t* __restrict__ cu,
    const int32_t* __restrict__ blk_start, int n_req, int H, float scale_log2e, __half* __restrict__ out_hi,
    __half* __restrict__ out_lo) {
  __shared__ __attribute__((aligned(16))) __half s_k[2][TK * D];    // hi, lo
  __shared__ __attribute__((aligned(16))) __half s_v[2][D * TK];    // hi, lo (transposed, keys permuted)
  constexpr int QBLK = 32 * NW;
  constexpr int NT = NW * 64;
  constexpr int ITERS = (TK * 8) / NT;   // 16-B chunks per plane per thread

  const int b = blockIdx.x;
  if (b >= blk_start[n_req]) return;
  const int head = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = find_request(blk_start, n_req, b);
  const int qblk0 = (b - blk_start[r]) * QBLK;
  const int t0 = cu[r] - cu[0];
  const int L = cu[r + 1] - cu[r];
  const int q0 = qblk0 + wave * 32;
  const bool wave_active = q0 < L;
  const size_t ld = (size_t)3 * H;
  const int lq = lane & 31, lh = lane >> 5;

  // Q fragments (B operand: column = query lq, k-slots = d ks*16 + 8*lh .. +7), kept in registers
  f16x8 qh[4], ql[4];
  {
    const size_t qrow = (size_t)(t0 + min(q0 + lq, L - 1)) * ld + head * D + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = *reinterpret_cast<const f16x8*>(qkv_hi + qrow + ks * 16);
      ql[ks] = *reinterpret_cast<const f16x8*>(qkv_lo + qrow + ks * 16);
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m = NEG, l = 0.f;

  const int kend = min(L, qblk0 + QBLK);      // keys needed by this block: [0, kend)
  uint4 pk[ITERS][4];                          // prefetch: k_hi, k_lo, v_hi, v_lo chunks
  auto gload = [&](int kt) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int c = it * NT + tid;
      const int key = c >> 3, dc = c & 7;
      const size_t base = (size_t)(t0 + min(kt + key, L - 1)) * ld + head * D + dc * 8;
      pk[it][0] = *reinterpret_cast<const uint4*>(qkv_hi + base + H);
      pk[it][1] = *reinterpret_cast<const uint4*>(qkv_lo + base + H);
      pk[it][2] = *reinterpret_cast<const uint4*>(qkv_hi + base + 2 * H);
      pk[it][3] = *reinterpret_cast<const uint4*>(qkv_lo + base + 2 * H);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int c = it * NT + tid;
      const int key = c >> 3, dc = c & 7;
      *reinterpret_cast<uint4*>(&s_k[0][k_off(key, dc)]) = pk[it][0];
      *reinterpret_cast<uint4*>(&s_k[1][k_off(key, dc)]) = pk[it][1];
      // key -> (chunk, element) of the V^T row so that it matches the S^T register order:
      // register e of k-step g on lane-half h holds key 16g + 8(e>>2) + 4h + (e&3)
      const int g = key >> 4, k16 = key & 15;
      const int h = (k16 >> 2) & 1, e = (k16 & 3) + 4 * (k16 >> 3);
      const int chunk = g * 2 + h;
      const __half* vh = reinterpret_cast<const __half*>(&pk[it][2]);
      const __half* vl = reinterpret_cast<const __half*>(&pk[it][3]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = dc * 8 + i;
        s_v[0][v_off(row, chunk) + e] = vh[i];
        s_v[1][v_off(row, chunk) + e] = vl[i];
      }
    }
  };

  gload(0);
  for (int kt = 0; kt < kend; kt += TK) {
    __syncthreads();
    sstore();
    __syncthreads();
    if (kt + TK < kend) gload(kt + TK);
    if (!wave_active || kt > q0) continue;       // wave-uniform: past this wave's diagonal

    // ---- S^T = K Q^T (rows = keys, cols = queries)
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + lh;
      const f16x8 kh = *reinterpret_cast<const f16x8*>(&s_k[0][k_off(lq, c)]);
      const f16x8 kl = *reinterpret_cast<const f16x8*>(&s_k[1][k_off(lq, c)]);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], sacc, 0, 0, 0);
    }
    // ---- online softmax for query lq (this lane: 16 of its 32 keys; lane^32: the others)
    const bool diag = kt == q0;
    float mx = NEG;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = sacc[i] * scale_log2e;
      const int key = (i & 3) + 8 * (i >> 2) + 4 * lh;      // relative to kt
      if (diag && key > lq) v = NEG;                          // causal
      sacc[i] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float alpha = exp2f(m - m_new);
    m = m_new;
    float psum = 0.f;
    f16x8 ph[2], pl[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = exp2f(sacc[i] - m_new);
      psum += p;
      const _Float16 hi = (_Float16)p;
      ph[i >> 3][i & 7] = hi;
      pl[i >> 3][i & 7] = (_Float16)(p - (float)hi);
    }
    l = l * alpha + psum;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
    // ---- O^T += V^T P^T  (rows = d, cols = queries)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int row = dt * 32 + lq, c = g * 2 + lh;
        const f16x8 vh = *reinterpret_cast<const f16x8*>(&s_v[0][v_off(row, c)]);
        const f16x8 vl = *reinterpret_cast<const f16x8*>(&s_v[1][v_off(row, c)]);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[g], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[g], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[g], o[dt], 0, 0, 0);
      }
    }
  }
  if (!wave_active) return;
  l += __shfl_xor(l, 32, 64);
  const int qi = q0 + lq;
  if (qi >= L) return;
  const float inv = 1.f / l;
  const size_t ob = (size_t)(t0 + qi) * H + head * D;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __half hh[4], ll[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) split_f16(o[dt][4 * j + i] * inv, hh[i], ll[i]);
      const int d = dt * 32 + 8 * j + 4 * lh;
      *reinterpret_cast<uint2*>(out_hi + ob + d) = *reinterpret_cast<const uint2*>(hh);
      *reinterpret_cast<uint2*>(out_lo + ob + d) = *reinterpret_cast<const uint2*>(ll);
    }
  }
}

}  // namespace

int launch_attention(int wdtype, AOp qkv, const int32_t* cu, int n_req, int T, int H, int n_heads,
                     int32_t* blk_start, AOp out, hipStream_t s) {
  if (n_req == 0 || T == 0) return LTR_OK;
  if (H != n_heads * D) { set_error("attention: head size must be 64 (H=%d heads=%d)", H, n_heads); return LTR_E_INVAL; }
  const float scale_log2e = 0.125f * 1.4426950408889634f;   // d^-0.5 (opt.py:73) * log2(e)
  if (wdtype == LTR_W_F16) {
    constexpr int NW = 4;
    attn_blocks_kernel<<<1, 1024, 0, s>>>(cu, n_req, 32 * NW, blk_start);
    LTR_LAUNCH_CHECK();
    dim3 grid(T / (32 * NW) + n_req, n_heads);   // sum ceil(L/qb) <= floor(T/qb) + n_req
    attn_f16s_kernel<NW><<<grid, NW * 64, 0, s>>>((const __half*)qkv.hi, (const __half*)qkv.lo, cu, blk_start, n_req,
                                                  H, scale_log2e, (__half*)out.hi, (__half*)out.lo);
  } else {
    attn_blocks_kernel<<<1, 1024, 0, s>>>(cu, n_req, QB, blk_start);
    LTR_LAUNCH_CHECK();
    dim3 grid(T / QB + n_req, n_heads);
    if (wdtype == LTR_W_F32)
      attn_f32_kernel<false><<<grid, 64, 0, s>>>((const float*)qkv.hi, cu, blk_start, n_req, H, scale_log2e, out.hi,
                                                 out.lo);
    else   // debug A/B (wdtype -1): f32 VALU attention feeding split operands
      attn_f32_kernel<true><<<grid, 64, 0, s>>>((const float*)qkv.hi, cu, blk_start, n_req, H, scale_log2e, out.hi,
                                                out.lo);
  }
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // namespace ltr
