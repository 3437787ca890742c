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
    const float* __restrict__ hidden, const int32_t* __restrict__ cu, int tok_off, int H, int De, int num_labels,
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
      for (int w = 0; w < 4 && j0 + w < num_labels; ++w)
        if (s_red[w] > s_best) { s_best = s_red[w]; s_besti = j0 + w; }
      if (j0 == 0 && num_labels == 1) scores[req] = s_red[0];
    }
    __syncthreads();
  }
  if (tid == 0 && num_labels > 1) scores[req] = (float)s_besti;
}

}  // namespace

int launch_pool_head(int wdtype, const float* hidden, const int32_t* cu, int tok_off, int N, int H, int De,
                     int num_labels, const float* ln_w, const float* ln_b, const void* proj_out,
                     const void* score_w, float* scores_out, float* logits_out, hipStream_t s) {
  if (N == 0) return LTR_OK;
  if (H > PH_MAXH || De > PH_MAXH) { set_error("pool_head: H/De > %d", PH_MAXH); return LTR_E_INVAL; }
  if (wdtype == LTR_W_F16)
    pool_head_kernel<__half><<<N, PH_THREADS, 0, s>>>(hidden, cu, tok_off, H, De, num_labels, ln_w, ln_b,
                                                      (const __half*)proj_out, (const __half*)score_w, scores_out,
                                                      logits_out);
  else
    pool_head_kernel<float><<<N, PH_THREADS, 0, s>>>(hidden, cu, tok_off, H, De, num_labels, ln_w, ln_b,
                                                     (const float*)proj_out, (const float*)score_w, scores_out,
                                                     logits_out);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // namespace ltr
