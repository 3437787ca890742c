// ListMLE training loss of the predictor on gfx950 (the loss the reference fine-tunes the
// OPT predictors with).
//
// Reference: train/allrank/models/losses/listMLE.py:23-54, called by train/trainer.py:125-150 as
// loss_func(outputs.view(1, -1), labels): shuffle the slate, sort by true label (descending), mask padded
// items (y_true == pad), loss = mean_b sum_i [ log(sum_{j>=i} exp(p_j - max) + eps) - (p_i - max) ],
// and its gradient w.r.t. the predictions.  The shuffle permutation (random_indices, :33) is an input;
// ties keep the shuffled order (the reference's torch.sort leaves their order unspecified).
//
// One 256-thread workgroup per slate; the slate lives in LDS: rank by counting (unique keys
// (label desc, shuffled position asc)), two block scans (reversed cumsum of exp, forward cumsum of 1/cumsum).
// Latency / LDS bound: a slate is a training batch (tens to hundreds of items).
#include "ltr_internal.h"

namespace ltr {
namespace {

constexpr int LM_THREADS = 256;
constexpr int LM_MAXS = 4096;

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

// inclusive scan of a[0..S) in place, forward (rev = false) or from the end (rev = true); every thread owns a
// contiguous run of ceil(S / 256) elements
__device__ void block_scan(float* a, int S, bool rev, float* part) {
  const int tid = threadIdx.x;
  const int per = (S + LM_THREADS - 1) / LM_THREADS;
  const int lo = tid * per, hi = min(lo + per, S);
  float s = 0.f;
  for (int k = lo; k < hi; ++k) { const int i = rev ? S - 1 - k : k; s += a[i]; a[i] = s; }
  __syncthreads();
  part[tid] = s;
  __syncthreads();
  float off = 0.f;
  for (int t = 0; t < tid; ++t) off += part[t];      // 256 partials: serial prefix, deterministic order
  for (int k = lo; k < hi; ++k) { const int i = rev ? S - 1 - k : k; a[i] += off; }
  __syncthreads();
}

__global__ void __launch_bounds__(LM_THREADS) listmle_kernel(const float* __restrict__ y_pred,
                                                             const float* __restrict__ y_true,
                                                             const int32_t* __restrict__ shuffle, int B, int S,
                                                             float eps, float pad, float* __restrict__ row_loss,
                                                             float* __restrict__ grad) {
  extern __shared__ __attribute__((aligned(16))) float lm[];
  float* ts = lm;              // shuffled labels
  float* sp = lm + S;          // predictions in sorted order (-inf where padded)
  float* ce = lm + 2 * S;      // exp(p - max), then its reversed cumsum
  float* cw = lm + 3 * S;      // 1 / (cumsum + eps), then its forward cumsum
  int* rk = reinterpret_cast<int*>(lm + 4 * S);   // rank of shuffled position j
  __shared__ float red[4];
  __shared__ float part[LM_THREADS];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* pr = y_pred + (size_t)b * S;
  const float* tr = y_true + (size_t)b * S;
  for (int j = tid; j < S; j += LM_THREADS) ts[j] = tr[shuffle[j]];
  __syncthreads();
  for (int j = tid; j < S; j += LM_THREADS) {
    const float t = ts[j];
    int r = 0;
    for (int k = 0; k < S; ++k) { const float u = ts[k]; r += (u > t) || (u == t && k < j); }
    rk[j] = r;
    sp[r] = (t == pad) ? -INFINITY : pr[shuffle[j]];
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int i = tid; i < S; i += LM_THREADS) mx = fmaxf(mx, sp[i]);
  mx = block_reduce(mx, red, true);
  for (int i = tid; i < S; i += LM_THREADS) ce[i] = (sp[i] == -INFINITY) ? 0.f : expf(sp[i] - mx);
  __syncthreads();
  for (int i = tid; i < S; i += LM_THREADS) cw[i] = ce[i];          // keep exp() before it is overwritten by the scan
  __syncthreads();
  block_scan(ce, S, true, part);                                     // ce[i] = sum_{j >= i} exp
  float ls = 0.f, mp = 0.f;
  int amax = S;
  for (int i = tid; i < S; i += LM_THREADS) {
    const bool masked = sp[i] == -INFINITY;
    const float e = cw[i], c = ce[i];
    if (!masked) {
      ls += logf(c + eps) - (sp[i] - mx);
      mp += eps / (c + eps);                                         // d obs_i / d max (the path through max_pred_values)
      if (sp[i] == mx) amax = min(amax, i);
    }
    ts[i] = e;                                                       // ts is dead: exp() per sorted position
    cw[i] = masked ? 0.f : 1.f / (c + eps);
  }
  ls = block_reduce(ls, red, false);
  if (tid == 0) row_loss[b] = ls;
  if (grad == nullptr) return;
  // autograd routes d loss / d max to the arg-max item (first one in sorted order): negligible while every
  // cumsum >> eps, ~1 per trailing item once exp(p_i - max) < eps (score spread above ~23)
  mp = block_reduce(mp, red, false);
  amax = -(int)block_reduce((float)-amax, red, true);                // min over the block (S <= 4096: exact in f32)
  __syncthreads();
  block_scan(cw, S, false, part);                                    // cw[i] = sum_{k <= i} 1 / (c_k + eps)
  const float invB = 1.f / (float)B;
  for (int j = tid; j < S; j += LM_THREADS) {
    const int r = rk[j];
    const bool masked = sp[r] == -INFINITY;
    grad[(size_t)b * S + shuffle[j]] = masked ? 0.f : (ts[r] * cw[r] - 1.f + (r == amax ? mp : 0.f)) * invB;
  }
}

__global__ void __launch_bounds__(64) listmle_mean_kernel(const float* __restrict__ row_loss, int B, float* loss) {
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += row_loss[b];                    // fixed order
    *loss = s / (float)B;
  }
}

}  // namespace
}  // namespace ltr

using namespace ltr;

extern "C" int ltr_listmle(const float* y_pred, const float* y_true, const int32_t* shuffle, int32_t B, int32_t S,
                           float eps, float pad_value, float* loss_out, float* row_loss_out, float* grad_out,
                           void* stream) {
  if (B < 0 || S < 0 || (B > 0 && S > 0 && (!y_pred || !y_true || !shuffle || !row_loss_out)) || !loss_out) {
    set_error("ltr_listmle: bad argument");
    return LTR_E_INVAL;
  }
  if (S > LM_MAXS) { set_error("ltr_listmle: slate length %d > %d", S, LM_MAXS); return LTR_E_INVAL; }
  hipStream_t s = (hipStream_t)stream;
  if (B == 0 || S == 0) { LTR_HIP_CHECK(hipMemsetAsync(loss_out, 0, sizeof(float), s)); return LTR_OK; }
  const size_t lds = (size_t)5 * S * sizeof(float);
  if (lds > 48 * 1024)
    LTR_HIP_CHECK(hipFuncSetAttribute((const void*)listmle_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  listmle_kernel<<<B, LM_THREADS, lds, s>>>(y_pred, y_true, shuffle, B, S, eps, pad_value, row_loss_out, grad_out);
  LTR_LAUNCH_CHECK();
  listmle_mean_kernel<<<1, 64, 0, s>>>(row_loss_out, B, loss_out);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}
