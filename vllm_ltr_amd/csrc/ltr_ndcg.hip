// NeuralNDCG training loss of the predictor on gfx950 (`--loss neuralNDCG` of the reference's trainer) and its
// gradient w.r.t. the predictions.
//
// Reference: train/allrank/models/losses/neuralNDCG.py:27-87 (deterministic variant, the defaults train/trainer.py:127-128,157
// calls it with), on train/allrank/models/losses/loss_utils.py:24-83 (deterministic_neural_sort, sinkhorn_scaling) and
// train/allrank/models/metrics.py:89-135 (dcg, for the ideal DCG).  Per slate of n items, mask = (y_true == pad):
//   NeuralSort   logits[r, c] = s_c scaling_r - sum_k |s_c - s_k|,  scaling_r = (n_live + 1) - 2 (r + 1) for r < n_live;
//                -inf where one of item r / item c is padded, 1 where both; P = softmax_c(logits / tau)       (rank r x item c)
//   Sinkhorn     <= 50 rounds of (divide by column sums, divide by row sums), sums clamped at 1e-10; the loop ends after the
//                first round in which every row and column sum of EVERY slate of the batch is within 1e-6 of 1
//   NDCG         sum_{r < k} (sum_c P[r, c] (2^y_c - 1)) / log2(r + 2)  over  (ideal DCG@k + 1e-10);  loss = -mean over the
//                slates whose ideal DCG is not 0 (none: 0, no gradient)
// Autograd of the reference unrolls the Sinkhorn rounds; so does this backward, from the matrices the forward keeps.
//
// Shape of the work: a slate is a training batch (trainer.py --batch-size 32), the matrices are n x n, every pass over them is
// a dependent step: latency-bound VALU work, one 256-thread workgroup per slate, matrices in the caller's workspace (L2
// resident), vectors in LDS.  (Keeping the working matrix in LDS as well was tried: 1,027 -> 982 us for a slate of 32 - the time
// is the 50 x 2 chain of wave reductions, divisions and barriers, not the loads - and dropped.)  Four launches: forward (sort + all 50 rounds, each round's residual recorded), value (picks the
// batch-wide stopping round, NDCG and ideal DCG per slate), mean (loss, number of live slates), backward.
// All sums run in a fixed order: results are reproducible run to run.
#include "ltr_internal.h"

namespace ltr {
namespace {

constexpr int ND_THREADS = 256;
constexpr int ND_WAVES = ND_THREADS / 64;
constexpr int ND_MAXS = 1024;
constexpr int ND_ROUNDS = 50;          // neuralNDCG.py:57 max_iter
constexpr float ND_TOL = 1e-6f;        // neuralNDCG.py:57 tol
constexpr float ND_EPS = 1e-10f;       // allrank DEFAULT_EPS

// workspace of one slate (floats): matrices M_0 .. M_{2 ROUNDS} (M_0 = masked softmax, then one per half round), the
// softmax itself, the gradient matrix, the residual of every round
__host__ __device__ inline size_t nd_slate_floats(int S) { return (size_t)(2 * ND_ROUNDS + 3) * S * S + 64; }

struct NdWs {
  float* mats; float* soft; float* grad; float* res;
};
__device__ __forceinline__ NdWs nd_carve(float* base, int b, int S) {
  float* p = base + (size_t)b * nd_slate_floats(S);
  const size_t SS = (size_t)S * S;
  return NdWs{p, p + (2 * ND_ROUNDS + 1) * SS, p + (2 * ND_ROUNDS + 2) * SS, p + (2 * ND_ROUNDS + 3) * SS};
}

__device__ __forceinline__ float nd_block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float nd_block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// max that keeps a NaN (a poisoned slate must not look converged)
__device__ __forceinline__ float nan_max(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }

struct Slate {               // LDS vectors of one slate
  float* s;      // predictions, 0 where padded
  float* lab;    // labels, 0 where padded
  float* a;      // sum_k |s_c - s_k|
  float* v0;     // column sums / scratch
  float* v1;     // row sums / scratch
  int* pad;
};
__device__ __forceinline__ Slate slate_lds(float* lds, int S) {
  return Slate{lds, lds + S, lds + 2 * S, lds + 3 * S, lds + 4 * S, reinterpret_cast<int*>(lds + 5 * S)};
}

// loads the slate; returns the number of unpadded items
__device__ int load_slate(const Slate& L, const float* __restrict__ y_pred, const float* __restrict__ y_true, int b, int S,
                          float pad_value, float* red) {
  float live = 0.f;
  for (int c = threadIdx.x; c < S; c += ND_THREADS) {
    const float t = y_true[(size_t)b * S + c];
    const bool p = t == pad_value;
    L.pad[c] = p;
    L.lab[c] = p ? 0.f : t;
    L.s[c] = p ? 0.f : y_pred[(size_t)b * S + c];
    live += p ? 0.f : 1.f;
  }
  return (int)nd_block_sum(live, red);       // S <= 1024: exact
}

__device__ __forceinline__ float scaling_of(int r, int n_live) { return r < n_live ? (float)(n_live + 1 - 2 * (r + 1)) : 0.f; }
__device__ __forceinline__ float discount_of(int r) { return 1.f / log2f((float)r + 2.f); }

__global__ void __launch_bounds__(ND_THREADS) ndcg_forward_kernel(const float* __restrict__ y_pred, const float* __restrict__ y_true,
                                                                  int S, float pad_value, float inv_tau, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[ND_WAVES];
  const Slate L = slate_lds(lds, S);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const NdWs W = nd_carve(ws, b, S);
  const size_t SS = (size_t)S * S;
  const int n_live = load_slate(L, y_pred, y_true, b, S, pad_value, red);
  __syncthreads();
  for (int c = tid; c < S; c += ND_THREADS) {                              // loss_utils.py:63-66
    float acc = 0.f;
    if (!L.pad[c]) { const float sc = L.s[c]; for (int k = 0; k < S; ++k) acc += L.pad[k] ? 0.f : fabsf(sc - L.s[k]); }
    L.a[c] = acc;
  }
  __syncthreads();
  // softmax rows (loss_utils.py:68-83) and the matrix Sinkhorn starts from (:34-36); a wave per rank r
  float* M0 = W.mats;
  for (int r = wave; r < S; r += ND_WAVES) {
    const float sc = scaling_of(r, n_live);
    const bool pr = L.pad[r];
    float mx = -INFINITY;
    for (int c = lane; c < S; c += 64) {
      const bool pc = L.pad[c];
      const float z = (pr || pc) ? ((pr && pc) ? inv_tau : -INFINITY) : (L.s[c] * sc - L.a[c]) * inv_tau;
      W.soft[(size_t)r * S + c] = z;
      mx = fmaxf(mx, z);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < S; c += 64) { const float e = expf(W.soft[(size_t)r * S + c] - mx); W.soft[(size_t)r * S + c] = e; sum += e; }
    sum = wave_sum(sum);
    for (int c = lane; c < S; c += 64) {
      const float p = W.soft[(size_t)r * S + c] / sum;
      const bool pc = L.pad[c];
      W.soft[(size_t)r * S + c] = p;
      M0[(size_t)r * S + c] = (pr || pc) ? ((pr && pc) ? 1.f : 0.f) : p;
    }
  }
  __syncthreads();
  for (int c = tid; c < S; c += ND_THREADS) { float cs = 0.f; for (int r = 0; r < S; ++r) cs += M0[(size_t)r * S + c]; L.v0[c] = cs; }
  __syncthreads();
  for (int t = 0; t < ND_ROUNDS; ++t) {                                    // loss_utils.py:38-43
    const float* X = W.mats + (size_t)(2 * t) * SS;
    float* Y = W.mats + (size_t)(2 * t + 1) * SS;
    float* Z = W.mats + (size_t)(2 * t + 2) * SS;
    float worst = 0.f;
    for (int r = wave; r < S; r += ND_WAVES) {
      float rs = 0.f;
      for (int c = lane; c < S; c += 64) { const float y = X[(size_t)r * S + c] / fmaxf(L.v0[c], ND_EPS); Y[(size_t)r * S + c] = y; rs += y; }
      rs = wave_sum(rs);
      const float rc = fmaxf(rs, ND_EPS);
      float zs = 0.f;
      for (int c = lane; c < S; c += 64) { const float z = Y[(size_t)r * S + c] / rc; Z[(size_t)r * S + c] = z; zs += z; }
      zs = wave_sum(zs);
      worst = nan_max(worst, fabsf(zs - 1.f));
    }
    __syncthreads();
    for (int c = tid; c < S; c += ND_THREADS) {
      float cs = 0.f;
      for (int r = 0; r < S; ++r) cs += Z[(size_t)r * S + c];
      L.v0[c] = cs;
      worst = nan_max(worst, fabsf(cs - 1.f));
    }
    // NaN-keeping block max: a NaN anywhere makes the residual NaN (never "< tol", as in the reference's comparison)
    const float any_nan = nd_block_max(worst != worst ? 1.f : 0.f, red);
    const float mx = nd_block_max(worst != worst ? 0.f : worst, red);
    if (tid == 0) W.res[t] = any_nan > 0.f ? __builtin_nanf("") : mx;
    __syncthreads();
  }
}

// rounds the reference's loop runs: the first round after which every slate is within tol (loss_utils.py:42-43), else all
__device__ int rounds_run(const float* __restrict__ ws, int B, int S) {
  for (int t = 0; t < ND_ROUNDS; ++t) {
    bool ok = true;
    for (int b = 0; b < B && ok; ++b) ok = (ws + (size_t)b * nd_slate_floats(S) + (size_t)(2 * ND_ROUNDS + 3) * S * S)[t] < ND_TOL;
    if (ok) return t + 1;
  }
  return ND_ROUNDS;
}

__global__ void __launch_bounds__(ND_THREADS) ndcg_value_kernel(const float* __restrict__ y_true, int B, int S, int k, float pad_value,
                                                                const float* __restrict__ ws, float* __restrict__ ndcg_out,
                                                                float* __restrict__ idcg_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[ND_WAVES];
  const Slate L = slate_lds(lds, S);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  (void)load_slate(L, y_true, y_true, b, S, pad_value, red);               // predictions are not needed here
  __syncthreads();
  const int rounds = rounds_run(ws, B, S);
  const float* Fm = ws + (size_t)b * nd_slate_floats(S) + (size_t)(2 * rounds) * S * S;
  for (int c = tid; c < S; c += ND_THREADS) L.v0[c] = exp2f(L.lab[c]) - 1.f;         // neuralNDCG.py:62-64 (padded: 2^0 - 1)
  __syncthreads();
  float dcg = 0.f;
  for (int r = wave; r < min(S, k); r += ND_WAVES) {                       // :66-68, cut at k (:75)
    float gt = 0.f;
    if (!L.pad[r]) for (int c = lane; c < S; c += 64) gt += L.pad[c] ? 0.f : Fm[(size_t)r * S + c] * L.v0[c];
    gt = wave_sum(gt);
    if (lane == 0) dcg += gt * discount_of(r);
  }
  dcg = nd_block_sum(dcg, red);
  // ideal DCG@k (metrics.py:89-135 with y_pred = y_true): labels in descending order, padded items last with gain 0
  float ideal = 0.f;
  for (int c = tid; c < S; c += ND_THREADS) {
    const bool pc = L.pad[c];
    const float key = L.lab[c];
    int rank = 0;
    for (int j = 0; j < S; ++j) {
      const bool pj = L.pad[j];
      const float kj = L.lab[j];
      rank += pc ? (!pj || j < c) : (!pj && (kj > key || (kj == key && j < c)));
    }
    if (rank < k) ideal += L.v0[c] * discount_of(rank);
  }
  ideal = nd_block_sum(ideal, red);
  if (tid == 0) {
    idcg_out[b] = ideal;
    ndcg_out[b] = ideal == 0.f ? 0.f : dcg / (ideal + ND_EPS);             // :76-78
  }
}

// loss = -sum ndcg / #(ideal DCG != 0), or 0 when no slate has gain (:80-86); cnt_out = that count
__global__ void __launch_bounds__(64) ndcg_mean_kernel(const float* __restrict__ ndcg, const float* __restrict__ idcg, int B,
                                                       float* __restrict__ loss, float* __restrict__ cnt_out) {
  if (threadIdx.x == 0) {
    float s = 0.f, cnt = 0.f;
    for (int b = 0; b < B; ++b) { s += ndcg[b]; cnt += idcg[b] == 0.f ? 0.f : 1.f; }
    *cnt_out = cnt;
    *loss = cnt == 0.f ? 0.f : -s / cnt;
  }
}

__global__ void __launch_bounds__(ND_THREADS) ndcg_backward_kernel(const float* __restrict__ y_pred, const float* __restrict__ y_true,
                                                                   int B, int S, int k, float pad_value, float inv_tau,
                                                                   float* __restrict__ ws, const float* __restrict__ idcg_all,
                                                                   const float* __restrict__ cnt_p, float* __restrict__ grad_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[ND_WAVES];
  const Slate L = slate_lds(lds, S);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const NdWs W = nd_carve(ws, b, S);
  const size_t SS = (size_t)S * S;
  const float cnt = *cnt_p, idcg = idcg_all[b];
  if (cnt == 0.f || idcg == 0.f) {                                         // no path from this slate to the loss (:77-78, :83-84)
    for (int c = tid; c < S; c += ND_THREADS) grad_out[(size_t)b * S + c] = 0.f;
    return;
  }
  const int n_live = load_slate(L, y_pred, y_true, b, S, pad_value, red);
  __syncthreads();
  const int rounds = rounds_run(ws, B, S);
  float* G = W.grad;
  const float alpha = -1.f / (cnt * (idcg + ND_EPS));
  // d loss / d P_final[r, c] = alpha disc_r gain_c for r < k; the padded entries are constants (:60)
  for (int r = wave; r < S; r += ND_WAVES) {
    const float dr = (r < k && !L.pad[r]) ? alpha * discount_of(r) : 0.f;
    for (int c = lane; c < S; c += 64) G[(size_t)r * S + c] = L.pad[c] ? 0.f : dr * (exp2f(L.lab[c]) - 1.f);
  }
  __syncthreads();
  for (int t = rounds - 1; t >= 0; --t) {
    const float* X = W.mats + (size_t)(2 * t) * SS;
    const float* Y = W.mats + (size_t)(2 * t + 1) * SS;
    // Z = Y / clamp(rowsum Y): dY = dZ / rc - [rs >= eps] (sum_c dZ Y) / rc^2
    for (int r = wave; r < S; r += ND_WAVES) {
      float rs = 0.f, dot = 0.f;
      for (int c = lane; c < S; c += 64) { const float y = Y[(size_t)r * S + c]; rs += y; dot += G[(size_t)r * S + c] * y; }
      rs = wave_sum(rs); dot = wave_sum(dot);
      const float rc = fmaxf(rs, ND_EPS);
      const float sub = rs >= ND_EPS ? dot / (rc * rc) : 0.f;
      for (int c = lane; c < S; c += 64) G[(size_t)r * S + c] = G[(size_t)r * S + c] / rc - sub;
    }
    __syncthreads();
    // Y = X / clamp(colsum X): dX = dY / cc - [cs >= eps] (sum_r dY X) / cc^2
    for (int c = tid; c < S; c += ND_THREADS) {
      float cs = 0.f, dot = 0.f;
      for (int r = 0; r < S; ++r) { const float x = X[(size_t)r * S + c]; cs += x; dot += G[(size_t)r * S + c] * x; }
      const float cc = fmaxf(cs, ND_EPS);
      const float sub = cs >= ND_EPS ? dot / (cc * cc) : 0.f;
      for (int r = 0; r < S; ++r) G[(size_t)r * S + c] = G[(size_t)r * S + c] / cc - sub;
    }
    __syncthreads();
  }
  // loss_utils.py:34-36: the padded entries of the starting matrix are constants; then the softmax (:82-83)
  for (int r = wave; r < S; r += ND_WAVES) {
    const bool pr = L.pad[r];
    float dot = 0.f;
    for (int c = lane; c < S; c += 64) {
      const float g = (pr || L.pad[c]) ? 0.f : G[(size_t)r * S + c];
      G[(size_t)r * S + c] = g;
      dot += g * W.soft[(size_t)r * S + c];
    }
    dot = wave_sum(dot);
    for (int c = lane; c < S; c += 64) G[(size_t)r * S + c] = W.soft[(size_t)r * S + c] * (G[(size_t)r * S + c] - dot) * inv_tau;
  }
  __syncthreads();
  // logits[r, c] = s_c scaling_r - a_c:  d s_c (direct) = sum_r dZ scaling_r,  d a_c = -sum_r dZ
  for (int c = tid; c < S; c += ND_THREADS) {
    float ds = 0.f, da = 0.f;
    if (!L.pad[c]) for (int r = 0; r < S; ++r) { const float g = L.pad[r] ? 0.f : G[(size_t)r * S + c]; ds += g * scaling_of(r, n_live); da -= g; }
    L.v0[c] = ds; L.v1[c] = da;
  }
  __syncthreads();
  // a_c = sum_k |s_c - s_k|:  d s_i += sum_k sign(s_i - s_k) (d a_i + d a_k)      (sign(0) = 0, as torch.abs differentiates)
  for (int i = tid; i < S; i += ND_THREADS) {
    float g = 0.f;
    if (!L.pad[i]) {
      const float si = L.s[i], dai = L.v1[i];
      g = L.v0[i];
      for (int j = 0; j < S; ++j) {
        if (L.pad[j]) continue;
        const float d = si - L.s[j];
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : (d == 0.f ? 0.f : d));      // NaN stays NaN
        g += sg * (dai + L.v1[j]);
      }
    }
    grad_out[(size_t)b * S + i] = g;
  }
}

}  // namespace
}  // namespace ltr

using namespace ltr;

extern "C" size_t ltr_neuralndcg_workspace_bytes(int32_t B, int32_t S) {
  if (B <= 0 || S <= 0 || S > ND_MAXS) return 0;
  return ((size_t)B * nd_slate_floats(S) + 2 * (size_t)B + 64) * sizeof(float);
}

extern "C" int ltr_neuralndcg(const float* y_pred, const float* y_true, int32_t B, int32_t S, float temperature, int32_t k,
                              float pad_value, float* loss_out, float* row_ndcg_out, float* grad_out, void* workspace,
                              size_t ws_bytes, void* stream) {
  if (B < 0 || S < 0 || (B > 0 && S > 0 && (!y_pred || !y_true || !row_ndcg_out || !workspace)) || !loss_out || !(temperature > 0.f)) {
    set_error("ltr_neuralndcg: bad argument");
    return LTR_E_INVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  if (B == 0 || S == 0) { LTR_HIP_CHECK(hipMemsetAsync(loss_out, 0, sizeof(float), s)); return LTR_OK; }
  if (S == 1) {    // the reference cannot run it either: IndexError at loss_utils.py:70 (mask.squeeze(-1).sum(dim=1))
    set_error("ltr_neuralndcg: a slate of one item (the reference raises IndexError, loss_utils.py:70)");
    return LTR_E_INVAL;
  }
  if (S > ND_MAXS) { set_error("ltr_neuralndcg: slate length %d > %d", S, ND_MAXS); return LTR_E_INVAL; }
  const size_t need = ltr_neuralndcg_workspace_bytes(B, S);
  if (ws_bytes < need) { set_error("ltr_neuralndcg: workspace too small (%zu < %zu)", ws_bytes, need); return LTR_E_NOMEM; }
  if (k <= 0 || k > S) k = S;                                              // neuralNDCG.py:46-47: k = None -> the slate length
  float* ws = (float*)workspace;
  float* idcg = ws + (size_t)B * nd_slate_floats(S);
  float* cnt = idcg + B;
  const size_t lds = (size_t)6 * S * sizeof(float);
  const float inv_tau = 1.f / temperature;
  ndcg_forward_kernel<<<B, ND_THREADS, lds, s>>>(y_pred, y_true, S, pad_value, inv_tau, ws);
  LTR_LAUNCH_CHECK();
  ndcg_value_kernel<<<B, ND_THREADS, lds, s>>>(y_true, B, S, k, pad_value, ws, row_ndcg_out, idcg);
  LTR_LAUNCH_CHECK();
  ndcg_mean_kernel<<<1, 64, 0, s>>>(row_ndcg_out, idcg, B, loss_out, cnt);
  LTR_LAUNCH_CHECK();
  if (grad_out) {
    ndcg_backward_kernel<<<B, ND_THREADS, lds, s>>>(y_pred, y_true, B, S, k, pad_value, inv_tau, ws, idcg, cnt, grad_out);
    LTR_LAUNCH_CHECK();
  }
  return LTR_OK;
}
