// Varlen causal self-attention of the predictor's prefill on gfx950.
//
// Reference: OPTAttention.forward (opt.py:92-102) with kv_cache=None, i.e. per-sequence
// softmax(q k^T * d^-0.5, causal) v over heads of 64 (rocm_flash_attn.py:244-290 on GPU,
// torch_sdpa.py:138-178 on CPU).  Input is the packed QKV GEMM output f32 [T, 3H]
// (q | k | v); output is the out_proj GEMM operand [T, H].
//
// v1 kernel: exact f32, flash-style online softmax, one query row per lane.  A workgroup
// is one wave: 64 consecutive queries of one (request, head); K/V tiles of 32 keys are
// staged in LDS with coalesced 256-B row loads and read back as broadcast ds_read_b128.
// Query blocks of all requests are enumerated through a device-side prefix table so the
// launch needs no host knowledge of the lengths.
#include "ltr_internal.h"

namespace ltr {
namespace {

constexpr int QB = 64;     // queries per workgroup
constexpr int KT = 32;     // keys per LDS tile
constexpr int D = 64;      // head size (OPT-125m/350m: 768/12 = 1024/16 = 64)
constexpr float NEG = -1.0e30f;

// blk_start[i] = sum_{j<i} ceil(L_j / QB), i in [0, n_req]
__global__ void __launch_bounds__(1024) attn_blocks_kernel(const int32_t* __restrict__ cu, int n_req,
                                                           int32_t* __restrict__ blk_start) {
  __shared__ int s_w[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { carry = 0; blk_start[0] = 0; }
  __syncthreads();
  for (int base = 0; base < n_req; base += 1024) {
    int i = base + tid;
    int v = (i < n_req) ? (cu[i + 1] - cu[i] + QB - 1) / QB : 0;
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

}  // namespace

int launch_attention(int wdtype, const float* qkv, const int32_t* cu, int n_req, int T, int H, int n_heads,
                     int32_t* blk_start, AOp out, hipStream_t s) {
  if (n_req == 0 || T == 0) return LTR_OK;
  if (H != n_heads * D) { set_error("attention: head size must be 64 (H=%d heads=%d)", H, n_heads); return LTR_E_INVAL; }
  attn_blocks_kernel<<<1, 1024, 0, s>>>(cu, n_req, blk_start);
  LTR_LAUNCH_CHECK();
  const int max_blocks = T / QB + n_req;   // sum ceil(L/QB) <= floor(T/QB) + n_req
  const float scale_log2e = 0.125f * 1.4426950408889634f;   // d^-0.5 (opt.py:73) * log2(e)
  dim3 grid(max_blocks, n_heads);
  if (wdtype == LTR_W_F16) attn_f32_kernel<true><<<grid, 64, 0, s>>>(qkv, cu, blk_start, n_req, H, scale_log2e, out.hi, out.lo);
  else attn_f32_kernel<false><<<grid, 64, 0, s>>>(qkv, cu, blk_start, n_req, H, scale_log2e, out.hi, out.lo);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // namespace ltr
