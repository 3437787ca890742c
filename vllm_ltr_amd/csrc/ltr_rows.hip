// Row-wise HBM-bound kernels of the predictor forward on gfx950:
//   * embedding gather    (opt.py:241-245, 43-53; vocab_parallel_embedding.py:95-106)
//   * LayerNorm           (nn.LayerNorm in opt.py:131-133,165-167,222-224)
//   * f32 -> GEMM operand (the hi|lo fp16 split of an activation)
// One wave (64 lanes) owns one row; lanes walk the row in 16-byte pieces so every global
// access is a full 1 KiB coalesced wave transaction.  No LDS is needed: the row lives in
// registers and the two row moments come from wave shuffles.
#include <algorithm>

#include "ltr_internal.h"

namespace ltr {
namespace {

constexpr int ROWS_PER_BLOCK = 4;   // 256 threads = 4 waves = 4 rows

template <typename T> struct Vec8;   // 8 consecutive table elements -> 8 floats
template <> struct Vec8<__half> {
  static __device__ __forceinline__ void load(const __half* p, float (&v)[8]) {
    uint4 raw = *reinterpret_cast<const uint4*>(p);
    const __half2* h = reinterpret_cast<const __half2*>(&raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) { float2 f = __half22float2(h[i]); v[2 * i] = f.x; v[2 * i + 1] = f.y; }
  }
};
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
};

template <typename T> struct Vec4;   // 4 consecutive table elements -> 4 floats
template <> struct Vec4<__half> {
  static __device__ __forceinline__ void load(const __half* p, float (&v)[4]) {
    uint2 raw = *reinterpret_cast<const uint2*>(p);
    const __half2* h = reinterpret_cast<const __half2*>(&raw);
    float2 f0 = __half22float2(h[0]), f1 = __half22float2(h[1]);
    v[0] = f0.x; v[1] = f0.y; v[2] = f1.x; v[3] = f1.y;
  }
};
template <> struct Vec4<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
};

__device__ __forceinline__ void store8_f32(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8_split(__half* hi, __half* lo, const float (&v)[8]) {
  __half h[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) split_f16(v[i], h[i], l[i]);
  *reinterpret_cast<uint4*>(hi) = *reinterpret_cast<const uint4*>(h);
  *reinterpret_cast<uint4*>(lo) = *reinterpret_cast<const uint4*>(l);
}

// hidden[t,:] = E_tok[ids[t],:] + E_pos[pos(t)+2,:]                      (De == H)
// hidden[t,:] = E_pos[pos(t)+2,:];  tok_out[t,:] = E_tok[ids[t],:]       (De != H)
// One wave owns EG_ROWS consecutive token rows: ONE binary search in cu_seqlens per wave (the following rows walk
// forward from it), and the gathers of all its rows are in flight before the first store - with one row per wave a
// wave spent ~13 dependent L2 round trips on the search before it issued its first table read.
constexpr int EG_ROWS = 4;
template <typename WT, bool PROJ>
__global__ void __launch_bounds__(256) embed_gather_kernel(
    const int64_t* __restrict__ ids, const int32_t* __restrict__ cu, int n_req, int T, int tok_off,
    const WT* __restrict__ tok_table, int De, int vocab, const WT* __restrict__ pos_table, int H,
    int pos_rows, float* __restrict__ hidden, void* tok_hi, void* tok_lo, int32_t* __restrict__ err_flag) {
  const int lane = threadIdx.x & 63;
  const int t0 = (blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * EG_ROWS;   // first row of this wave inside the chunk
  if (t0 >= T) return;
  const int nrow = min(EG_ROWS, T - t0);
  int req = find_request(cu, n_req, tok_off + t0);
  int req_beg = cu[req], req_end = cu[req + 1];
  const WT* trow[EG_ROWS];
  const WT* prow[EG_ROWS];
  bool bad = false;
#pragma unroll
  for (int r = 0; r < EG_ROWS; ++r) {
    const int tg = tok_off + min(t0 + r, T - 1);                       // global token index (clamped: unused rows repeat the last)
    while (tg >= req_end) { ++req; req_beg = req_end; req_end = cu[req + 1]; }   // wave-uniform walk
    const int pos = min(tg - req_beg + 2, pos_rows - 1);               // opt.py:43-53 offset
    long long id = ids[tg];
    if (id < 0 || id >= vocab) {
      // F.embedding raises here (vocab_parallel_embedding.py:95-106).  Flag it for ltr_status and read a valid
      // row instead of faulting; the scores of this call are invalid.
      bad = true;
      id = id < 0 ? 0 : vocab - 1;
    }
    trow[r] = tok_table + (size_t)id * De;
    prow[r] = pos_table + (size_t)pos * H;
  }
  if (bad && lane == 0 && err_flag != nullptr) atomicOr(err_flag, 1);
  if (!PROJ) {
    // 4 columns per lane: every f32 store instruction writes 1 KiB of contiguous line-complete memory
    // (8 columns per lane = two float4 stores that interleave in 16-byte pieces)
    for (int c = lane * 4; c < H; c += 256) {
      float a[EG_ROWS][4], b[EG_ROWS][4];
#pragma unroll
      for (int r = 0; r < EG_ROWS; ++r) { Vec4<WT>::load(trow[r] + c, a[r]); Vec4<WT>::load(prow[r] + c, b[r]); }
#pragma unroll
      for (int r = 0; r < EG_ROWS; ++r)
        if (r < nrow)
          *reinterpret_cast<float4*>(hidden + (size_t)(t0 + r) * H + c) =
              make_float4(a[r][0] + b[r][0], a[r][1] + b[r][1], a[r][2] + b[r][2], a[r][3] + b[r][3]);
    }
  } else {
    for (int c = lane * 8; c < H; c += 512) {
      float b[EG_ROWS][8];
#pragma unroll
      for (int r = 0; r < EG_ROWS; ++r) Vec8<WT>::load(prow[r] + c, b[r]);
#pragma unroll
      for (int r = 0; r < EG_ROWS; ++r)
        if (r < nrow) store8_f32(hidden + (size_t)(t0 + r) * H + c, b[r]);
    }
    for (int c = lane * 8; c < De; c += 512) {
      float a[EG_ROWS][8];
#pragma unroll
      for (int r = 0; r < EG_ROWS; ++r) Vec8<WT>::load(trow[r] + c, a[r]);
#pragma unroll
      for (int r = 0; r < EG_ROWS; ++r) {
        if (r >= nrow) continue;
        const size_t o = (size_t)(t0 + r) * De + c;
        if (sizeof(WT) == 2) store8_split((__half*)tok_hi + o, (__half*)tok_lo + o, a[r]);   // fp16 table values: hi = value, lo = 0
        else store8_f32((float*)tok_hi + o, a[r]);
      }
    }
  }
}

// y = (x - mean) * rsqrt(var + eps) * gamma + beta, biased variance, per row.
// HALF a wave (32 lanes) owns one row, MAXV = ceil(H / 256) register chunks of 8 floats per lane:
// lanes 0-31 hold row 2w, lanes 32-63 row 2w+1, so one store instruction writes, for every
// 32-column slab it touches, the 64-B pieces of two ADJACENT rows = one full 128-B line of the
// slab-major operand image (one row per wave left half-line writes: 17.5 -> 20.5 ms per step).
template <bool SPLIT, int MAXV>
__global__ void __launch_bounds__(256) layernorm_kernel(
    const float* x /* may alias out_f32 */, const float* __restrict__ gamma, const float* __restrict__ beta, int M,
    int H, float* out_f32, void* out_hi, void* out_lo) {
  const int lane = threadIdx.x & 31;
  const int row = blockIdx.x * (2 * ROWS_PER_BLOCK) + (threadIdx.x >> 5);
  if (row >= M) return;
  const float* xr = x + (size_t)row * H;
  float v[MAXV][8];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int c = k * 256 + lane * 8;
    if (c < H) {
      Vec8<float>::load(xr + c, v[k]);
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += v[k][i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[k][i] = 0.f;
    }
  }
  const float mean = half_wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int c = k * 256 + lane * 8;
    if (c < H) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { float d = v[k][i] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(half_wave_sum(sq) / (float)H + LN_EPS);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    int c = k * 256 + lane * 8;
    if (c < H) {
      float g[8], b[8], y[8];
      Vec8<float>::load(gamma + c, g);
      Vec8<float>::load(beta + c, b);
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = (v[k][i] - mean) * rstd * g[i] + b[i];
      if (out_f32) store8_f32(out_f32 + (size_t)row * H + c, y);
      if (out_hi) {
        if (SPLIT) { const size_t o = slab_off(row, c, M); store8_split((__half*)out_hi + o, (__half*)out_lo + o, y); }
        else store8_f32((float*)out_hi + (size_t)row * H + c, y);
      }
    }
  }
}

// Last-token rows of the residual stream (f32) and of an operand buffer, compacted to [n_req, H]:
// the decoder's last layer only needs them after attention (see forward_chunk).
template <bool SPLIT>
__global__ void __launch_bounds__(256) gather_last_rows_kernel(const int32_t* __restrict__ cu, int tok_off, int n_req,
                                                               int H, int64_t plane_src, int64_t plane_dst,
                                                               const float* __restrict__ h_src, const char* a_src,
                                                               float* __restrict__ h_dst, char* a_dst) {
  const int lane = threadIdx.x & 63;
  const int req = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (req >= n_req) return;
  const size_t srow = (size_t)(cu[req + 1] - 1 - tok_off), drow = (size_t)req;
  for (int c = lane * 8; c < H; c += 512) {
    float v[8];
    Vec8<float>::load(h_src + srow * H + c, v);
    store8_f32(h_dst + drow * H + c, v);
    if (!a_src) continue;
    if (SPLIT) {   // fp16 hi | lo planes: 16 B per plane
      const __half* sh = reinterpret_cast<const __half*>(a_src);
      __half* dh = reinterpret_cast<__half*>(a_dst);
      *reinterpret_cast<uint4*>(dh + drow * H + c) = *reinterpret_cast<const uint4*>(sh + srow * H + c);
      *reinterpret_cast<uint4*>(dh + plane_dst + drow * H + c) = *reinterpret_cast<const uint4*>(sh + plane_src + srow * H + c);
    } else {
      Vec8<float>::load(reinterpret_cast<const float*>(a_src) + srow * H + c, v);
      store8_f32(reinterpret_cast<float*>(a_dst) + drow * H + c, v);
    }
  }
}

template <bool SPLIT>
__global__ void __launch_bounds__(256) to_operand_kernel(const float* __restrict__ x, int M, int H8, void* hi,
                                                         void* lo) {
  const int64_t n8 = (int64_t)M * H8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    Vec8<float>::load(x + i * 8, v);
    if (SPLIT) {
      const size_t o = slab_off((int)(i / H8), (int)(i % H8) * 8, M);
      store8_split((__half*)hi + o, (__half*)lo + o, v);
    } else {
      store8_f32((float*)hi + i * 8, v);
    }
  }
}

}  // namespace

int launch_embed_gather(int wdtype, const int64_t* ids, const int32_t* cu, int N, int T, int tok_off,
                        const void* tok_table, int De, int vocab, const void* pos_table, int H, int pos_rows,
                        float* hidden_out, AOp tok_out, int32_t* err_flag, hipStream_t s) {
  if (T == 0) return LTR_OK;
  if ((H % 8) || (De % 8)) { set_error("embed_gather: H and De must be multiples of 8"); return LTR_E_INVAL; }
  dim3 grid((T + ROWS_PER_BLOCK * EG_ROWS - 1) / (ROWS_PER_BLOCK * EG_ROWS));
  const bool proj = De != H;
  if (wdtype == LTR_W_F16) {
    if (proj) embed_gather_kernel<__half, true><<<grid, 256, 0, s>>>(ids, cu, N, T, tok_off, (const __half*)tok_table, De, vocab, (const __half*)pos_table, H, pos_rows, hidden_out, tok_out.hi, tok_out.lo, err_flag);
    else embed_gather_kernel<__half, false><<<grid, 256, 0, s>>>(ids, cu, N, T, tok_off, (const __half*)tok_table, De, vocab, (const __half*)pos_table, H, pos_rows, hidden_out, nullptr, nullptr, err_flag);
  } else {
    if (proj) embed_gather_kernel<float, true><<<grid, 256, 0, s>>>(ids, cu, N, T, tok_off, (const float*)tok_table, De, vocab, (const float*)pos_table, H, pos_rows, hidden_out, tok_out.hi, tok_out.lo, err_flag);
    else embed_gather_kernel<float, false><<<grid, 256, 0, s>>>(ids, cu, N, T, tok_off, (const float*)tok_table, De, vocab, (const float*)pos_table, H, pos_rows, hidden_out, nullptr, nullptr, err_flag);
  }
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

int launch_layernorm(int wdtype, const float* x, const float* gamma, const float* beta, int M, int H,
                     float* out_f32, AOp out_op, hipStream_t s) {
  if (M == 0) return LTR_OK;
  if (H % 32 || H > 2048) { set_error("layernorm: H must be a multiple of 32 and <= 2048"); return LTR_E_INVAL; }
  dim3 grid((M + 2 * ROWS_PER_BLOCK - 1) / (2 * ROWS_PER_BLOCK));
  const bool split = wdtype == LTR_W_F16;
  const int maxv = (H + 255) / 256;
#define LN_LAUNCH(SP, MV) layernorm_kernel<SP, MV><<<grid, 256, 0, s>>>(x, gamma, beta, M, H, out_f32, out_op.hi, out_op.lo)
#define LN_PICK(SP) \
  if (maxv <= 1) LN_LAUNCH(SP, 1); else if (maxv == 2) LN_LAUNCH(SP, 2); else if (maxv == 3) LN_LAUNCH(SP, 3); \
  else if (maxv == 4) LN_LAUNCH(SP, 4); else LN_LAUNCH(SP, 8)
  if (split) { LN_PICK(true); } else { LN_PICK(false); }
#undef LN_PICK
#undef LN_LAUNCH
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

int launch_gather_last_rows(int wdtype, const int32_t* cu, int tok_off, int n_req, int H, const float* h_src, AOp a_src,
                            float* h_dst, AOp a_dst, hipStream_t s) {
  if (n_req == 0) return LTR_OK;
  dim3 grid((n_req + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  if (wdtype == LTR_W_F16) {
    const int64_t ps = a_src.hi ? ((const __half*)a_src.lo - (const __half*)a_src.hi) : 0;
    const int64_t pd = a_src.hi ? ((__half*)a_dst.lo - (__half*)a_dst.hi) : 0;
    gather_last_rows_kernel<true><<<grid, 256, 0, s>>>(cu, tok_off, n_req, H, ps, pd, h_src, (const char*)a_src.hi, h_dst,
                                                       (char*)a_dst.hi);
  } else {
    gather_last_rows_kernel<false><<<grid, 256, 0, s>>>(cu, tok_off, n_req, H, 0, 0, h_src, (const char*)a_src.hi, h_dst,
                                                        (char*)a_dst.hi);
  }
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

int launch_to_operand(int wdtype, const float* x, int M, int H, AOp out, hipStream_t s) {
  if (M == 0) return LTR_OK;
  if (H % 32) { set_error("to_operand: H must be a multiple of 32"); return LTR_E_INVAL; }
  int64_t n8 = (int64_t)M * H / 8;
  int blocks = (int)std::min<int64_t>((n8 + 255) / 256, 256 * 8);
  if (wdtype == LTR_W_F16) to_operand_kernel<true><<<blocks, 256, 0, s>>>(x, M, H / 8, out.hi, out.lo);
  else to_operand_kernel<false><<<blocks, 256, 0, s>>>(x, M, H / 8, out.hi, out.lo);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // namespace ltr
