// Hidden-state learning-to-rank head on gfx950 (next row of the scope table).
//
// Reference: vllm/model_executor/predictor.py - FCModel (:10-43): optional LayerNorm on the
// n_features-wide hidden state, then activation(Linear(x)) per layer; OutputLayer (:91-125):
// activation(w_1(x)), score = sum over d_output (or the single output); applied to the hidden
// states of the selected tokens (opt.py:250-255).
//
// The head runs on N selected rows only (one per request), so it is latency / L2 bound: one
// 256-thread workgroup per row keeps the activation vector in LDS (ping-pong buffers) and streams
// the layer weights, one wave per output neuron with lanes striding the input (coalesced 128-B
// rows, wave-shuffle reduction).  Exact f32 arithmetic on the checkpoint's weights.
#include "ltr_internal.h"

namespace ltr {
namespace {

constexpr int HD_THREADS = 256;
constexpr int HD_MAXW = 8192;     // widest vector kept in LDS

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_TANH = 3, ACT_GELU = 4, ACT_SILU = 5 };

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(x, 0.f);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    case ACT_TANH: return tanhf(x);
    case ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));   // nn.GELU default (erf)
    case ACT_SILU: return x / (1.f + expf(-x));
    default: return x;
  }
}

template <typename WT> __device__ __forceinline__ float wl(const WT* p);
template <> __device__ __forceinline__ float wl<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float wl<__half>(const __half* p) { return __half2float(*p); }

struct HeadLayers {
  const void* w[10];       // up to 8 FC layers + output layer
  const float* b[10];
  int in[10], out[10];
  int n;                   // number of dense layers including the output layer
};

template <typename WT>
__global__ void __launch_bounds__(HD_THREADS) ltr_head_kernel(const float* __restrict__ hidden,
                                                              const int32_t* __restrict__ row_index, int n_features,
                                                              const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, HeadLayers L, int act,
                                                              int out_act, float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) float s_buf[];   // 2 * maxw floats
  __shared__ float s_red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int req = blockIdx.x;
  const size_t row = row_index ? (size_t)row_index[req] : (size_t)req;
  int maxw = n_features;
  for (int i = 0; i < L.n; ++i) maxw = max(maxw, L.out[i]);
  float* cur = s_buf;
  float* nxt = s_buf + maxw;
  const float* xr = hidden + row * n_features;
  float part = 0.f;
  for (int c = tid; c < n_features; c += HD_THREADS) { const float v = xr[c]; cur[c] = v; part += v; }
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
  };
  if (ln_w != nullptr) {
    const float mean = block_sum(part) / (float)n_features;
    float sq = 0.f;
    for (int c = tid; c < n_features; c += HD_THREADS) { const float d = cur[c] - mean; sq += d * d; }
    const float rstd = rsqrtf(block_sum(sq) / (float)n_features + LN_EPS);
    for (int c = tid; c < n_features; c += HD_THREADS) cur[c] = (cur[c] - mean) * rstd * ln_w[c] + ln_b[c];
  }
  __syncthreads();
  for (int li = 0; li < L.n; ++li) {
    const int K = L.in[li], O = L.out[li];
    const WT* W = (const WT*)L.w[li];
    const int a = li == L.n - 1 ? out_act : act;
    for (int j = wave; j < O; j += 4) {
      const WT* wr = W + (size_t)j * K;
      float acc = 0.f;
      for (int c = lane; c < K; c += 64) acc = fmaf(cur[c], wl<WT>(wr + c), acc);
      acc = wave_sum(acc);
      if (lane == 0) nxt[j] = apply_act(acc + L.b[li][j], a);
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
  if (tid == 0) {   // OutputLayer.score: sum over d_output when > 1
    float sres = 0.f;
    const int O = L.out[L.n - 1];
    for (int j = 0; j < O; ++j) sres += cur[j];
    scores[req] = sres;
  }
}

}  // namespace

struct HeadModel {
  ltr_head_desc d;
  HeadLayers L;
  const float* ln_w;
  const float* ln_b;
  int maxw;
};

}  // namespace ltr

using namespace ltr;

extern "C" {

int ltr_head_create(const ltr_head_desc* d, const void* const* weights, int32_t n_weights, ltr_head_handle* out) {
  if (!d || !weights || !out) { set_error("ltr_head_create: NULL argument"); return LTR_E_INVAL; }
  if (d->n_fc < 0 || d->n_fc > 8 || d->n_features < 1 || d->n_features > HD_MAXW || d->d_output < 1) {
    set_error("ltr_head_create: bad shape (n_fc=%d n_features=%d d_output=%d)", d->n_fc, d->n_features, d->d_output);
    return LTR_E_INVAL;
  }
  const int want = 2 + 2 * (d->n_fc + 1);
  if (n_weights != want) { set_error("ltr_head_create: %d weight pointers, expected %d", n_weights, want); return LTR_E_INVAL; }
  HeadModel* m = new HeadModel();
  m->d = *d;
  m->ln_w = d->input_norm ? (const float*)weights[0] : nullptr;
  m->ln_b = d->input_norm ? (const float*)weights[1] : nullptr;
  if (d->input_norm && (!weights[0] || !weights[1])) { delete m; set_error("ltr_head_create: input_norm weights missing"); return LTR_E_INVAL; }
  int in = d->n_features;
  m->maxw = in;
  m->L.n = d->n_fc + 1;
  for (int i = 0; i <= d->n_fc; ++i) {
    const int o = i < d->n_fc ? d->fc_sizes[i] : d->d_output;
    if (o < 1 || o > HD_MAXW || !weights[2 + 2 * i] || !weights[3 + 2 * i]) {
      delete m; set_error("ltr_head_create: layer %d: bad size %d or NULL weights", i, o); return LTR_E_INVAL;
    }
    m->L.w[i] = weights[2 + 2 * i];
    m->L.b[i] = (const float*)weights[3 + 2 * i];
    m->L.in[i] = in; m->L.out[i] = o;
    in = o;
    if (o > m->maxw) m->maxw = o;
  }
  *out = (ltr_head_handle)m;
  return LTR_OK;
}

int ltr_head_destroy(ltr_head_handle h) { delete (HeadModel*)h; return LTR_OK; }

int ltr_head_score(ltr_head_handle h, const float* hidden, const int32_t* row_index, int32_t N, float* scores_out,
                   void* stream) {
  if (N < 0) { set_error("ltr_head_score: negative N"); return LTR_E_INVAL; }
  if (N == 0) return LTR_OK;
  HeadModel* m = (HeadModel*)h;
  if (!m || !hidden || !scores_out) { set_error("ltr_head_score: NULL argument"); return LTR_E_INVAL; }
  const size_t lds = (size_t)2 * m->maxw * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (lds > 48 * 1024) {   // up to 64 KiB at HD_MAXW: above the default dynamic-LDS limit of a launch
    const void* fn = m->d.weight_dtype == LTR_W_F16 ? (const void*)ltr_head_kernel<__half> : (const void*)ltr_head_kernel<float>;
    LTR_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  if (m->d.weight_dtype == LTR_W_F16)
    ltr_head_kernel<__half><<<N, HD_THREADS, lds, s>>>(hidden, row_index, m->d.n_features, m->ln_w, m->ln_b, m->L,
                                                       m->d.activation, m->d.output_activation, scores_out);
  else
    ltr_head_kernel<float><<<N, HD_THREADS, lds, s>>>(hidden, row_index, m->d.n_features, m->ln_w, m->ln_b, m->L,
                                                      m->d.activation, m->d.output_activation, scores_out);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // extern "C"
