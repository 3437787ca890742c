// Last-token pooling + final LayerNorm / project_out + score head on gfx950.
//
// Reference: _prune_hidden_states = index_select(selected_token_indices)
// (layers/logits_processor.py:74-79, indices = cu[i+1]-1 from model_runner.py:592-593),
// the final LayerNorm (125m) or project_out (350m) of OPTDecoder.forward (opt.py:259-262)
// - both per-token maps, so applying them to the N selected rows only equals applying
// them to all T rows first - then logits = x @ score.weight^T without bias
// (opt.py:374, logits_processor.py:61-71), and for num_labels > 1 the class-mode
// argmax returned as float (opt.py:394-395).
//
// HBM-bound: per request one H-float row is read (3 KiB), everything else (LN affine,
// project_out, score.weight) is shared and stays in L2.  One 256-thread workgroup per
// request; the row is staged in LDS, reductions use wave shuffles + LDS.
#include "ltr_internal.h"

namespace ltr {
namespace {

constexpr int PH_THREADS = 256;
constexpr int PH_MAXH = 2048;

template <typename WT> __device__ __forceinline__ float wload(const WT* p);
template <> __device__ __forceinline__ float wload<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float wload<__half>(const __half* p) { return __half2float(*p); }

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <typename WT>
__global__ void __launch_bounds__(PH_THREADS) pool_head_kernel(
    const float* __restrict__ hidden, const int32_t* __restrict__ cu, int tok_off, int H, int De, int num_labels, int n_cmp,
    const float* __restrict__ ln_w, const float* __restrict__ ln_b, const WT* __restrict__ proj_out,
    const WT* __restrict__ score_w, float* __restrict__ scores, float* __restrict__ logits_out) {
  __shared__ float s_x[PH_MAXH];
  __shared__ float s_y[PH_MAXH];
  __shared__ float s_red[4];
  __shared__ float s_best;
  __shared__ int s_besti;
  const int req = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row = cu ? (size_t)(cu[req + 1] - 1 - tok_off) : (size_t)req;   // cu == nullptr: compact rows
  const float* xr = hidden + row * H;
  float part = 0.f;
  for (int c = tid; c < H; c += PH_THREADS) { float v = xr[c]; s_x[c] = v; part += v; }
  if (ln_w != nullptr) {
    const float mean = block_sum(part, s_red) / (float)H;
    float sq = 0.f;
    for (int c = tid; c < H; c += PH_THREADS) { float d = s_x[c] - mean; sq += d * d; }
    const float rstd = rsqrtf(block_sum(sq, s_red) / (float)H + LN_EPS);
    for (int c = tid; c < H; c += PH_THREADS) s_x[c] = (s_x[c] - mean) * rstd * ln_w[c] + ln_b[c];
  }
  __syncthreads();
  const float* feat = s_x;   // De-wide feature vector
  if (proj_out != nullptr) {
    // y[j] = sum_c x[c] * W_out[j, c]; one wave per output row, lanes stride the row (coalesced)
    for (int j = wave; j < De; j += 4) {
      const WT* wr = proj_out + (size_t)j * H;
      float acc = 0.f;
      for (int c = lane; c < H; c += 64) acc = fmaf(s_x[c], wload<WT>(wr + c), acc);
      acc = wave_sum(acc);
      if (lane == 0) s_y[j] = acc;
    }
    __syncthreads();
    feat = s_y;
  }
  if (tid == 0) { s_best = -INFINITY; s_besti = 0; }
  __syncthreads();
  // logits; class mode keeps the first maximum like torch.argmax
  for (int j0 = 0; j0 < num_labels; j0 += 4) {
    const int j = j0 + wave;
    float acc = 0.f;
    if (j < num_labels) {
      const WT* wr = score_w + (size_t)j * De;
      for (int c = lane; c < De; c += 64) acc = fmaf(feat[c], wload<WT>(wr + c), acc);
      acc = wave_sum(acc);
      if (lane == 0) {
        s_red[wave] = acc;
        if (logits_out) logits_out[(size_t)req * num_labels + j] = acc;
      }
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 0; w < 4 && j0 + w < n_cmp; ++w)            // labels past n_cmp do not compete (launch_pool_head)
        if (s_red[w] > s_best) { s_best = s_red[w]; s_besti = j0 + w; }
      if (j0 == 0 && num_labels == 1) scores[req] = s_red[0];
    }
    __syncthreads();
  }
  if (tid == 0 && num_labels > 1) scores[req] = (float)s_besti;
}

// Class-mode label of every row of a logits matrix (the GEMM head, ltr_api.hip): float(argmax_j logits[r, j]) over
// j < n_cmp with torch.argmax's first-maximum rule (opt.py:394-395); optionally the unpadded logits are copied out.
// One 256-thread workgroup per row; a thread walks its columns in increasing order and keeps the first maximum it
// meets, threads / waves are merged by (larger value, then smaller index).
__global__ void __launch_bounds__(256) argmax_rows_kernel(const float* __restrict__ logits, int ld, int n_cmp, int num_labels,
                                                          float* __restrict__ scores, float* __restrict__ logits_out) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (size_t)r * ld;
  constexpr int NONE = 0x7fffffff;
  float best = -INFINITY;
  int bi = NONE;
  for (int j = tid; j < num_labels; j += 256) {
    const float v = row[j];
    if (logits_out) logits_out[(size_t)r * num_labels + j] = v;
    if (j < n_cmp && (bi == NONE || v > best)) { best = v; bi = j; }   // strict >: the first maximum of this thread's columns
  }
  // merge two candidates: the larger value wins, equal values the smaller column (torch.argmax's first maximum)
  auto take = [&](float ov, int oi) {
    if (oi != NONE && (bi == NONE || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
  };
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    take(ov, oi);
  }
  if (lane == 0) { s_v[wave] = best; s_i[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) take(s_v[w], s_i[w]);
    scores[r] = (float)(bi == NONE ? 0 : bi);
  }
}

}  // namespace

int launch_argmax_rows(const float* logits, int ld, int n_rows, int n_cmp, int num_labels, float* scores_out,
                       float* logits_out, hipStream_t s) {
  if (n_rows == 0) return LTR_OK;
  argmax_rows_kernel<<<n_rows, 256, 0, s>>>(logits, ld, n_cmp, num_labels, scores_out, logits_out);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

// n_cmp: number of leading labels that compete in the class-mode argmax - min(num_labels, vocab_size): the reference's
// LogitsProcessor cuts the logits at vocab_size columns (layers/logits_processor.py:68-70) before opt.py:395
int launch_pool_head(int wdtype, const float* hidden, const int32_t* cu, int tok_off, int N, int H, int De,
                     int num_labels, int n_cmp, const float* ln_w, const float* ln_b, const void* proj_out,
                     const void* score_w, float* scores_out, float* logits_out, hipStream_t s) {
  if (N == 0) return LTR_OK;
  if (H > PH_MAXH || De > PH_MAXH) { set_error("pool_head: H/De > %d", PH_MAXH); return LTR_E_INVAL; }
  if (wdtype == LTR_W_F16)
    pool_head_kernel<__half><<<N, PH_THREADS, 0, s>>>(hidden, cu, tok_off, H, De, num_labels, n_cmp, ln_w, ln_b,
                                                      (const __half*)proj_out, (const __half*)score_w, scores_out,
                                                      logits_out);
  else
    pool_head_kernel<float><<<N, PH_THREADS, 0, s>>>(hidden, cu, tok_off, H, De, num_labels, n_cmp, ln_w, ln_b,
                                                     (const float*)proj_out, (const float*)score_w, scores_out,
                                                     logits_out);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // namespace ltr
