// Internal declarations shared by the HIP translation units of libltr_hip.so.
// gfx950 (MI355X / CDNA4) only: wave64, MFMA, 160 KiB LDS.  No other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/ltr_hip.h"

namespace ltr {

constexpr int WAVE = 64;
constexpr float LN_EPS = 1e-5f;   // nn.LayerNorm default (opt.py:131-133)

void set_error(const char* fmt, ...);

#define LTR_HIP_CHECK(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      ::ltr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                       __FILE__, __LINE__);                                        \
      return LTR_E_HIP;                                                            \
    }                                                                              \
  } while (0)

#define LTR_LAUNCH_CHECK()                                                         \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      ::ltr::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),  \
                       __FILE__, __LINE__);                                        \
      return LTR_E_HIP;                                                            \
    }                                                                              \
  } while (0)

// ---- activation operand of a GEMM ------------------------------------------------
// F32 mode: `hi` is float [M, K], lo unused.
// F16 mode: hi / lo are __half [M, K] planes with a = hi + lo (|lo| <= ulp(hi)/2), so
// that a_hi*W + a_lo*W accumulated in f32 carries ~22 bits of the f32 activation.
struct AOp {
  void* hi;
  void* lo;
};

enum OutKind { OUT_F32 = 1, OUT_SPLIT = 2 };

struct GemmArgs {
  AOp a;                 // [M, K]
  const void* w;         // F32 mode: [N, K] row-major (nn.Linear layout); F16 mode: the slab-major image
                         // [K/32][N][32] made by launch_pack_weight
  const float* bias;     // [N] or nullptr
  const float* resid;    // f32 [M, N] or nullptr (may alias out_f32: in-place residual add)
  float* out_f32;        // f32 [M, N] or nullptr
  AOp out_split;         // hi/lo planes [M, N] (F16 mode) or f32 copy in .hi (F32 mode); may be null
  int M, N, K;
  int relu;
  // F16 mode only: operand images in SLAB-MAJOR order, element (row, k) at ((k >> 5) * M + row) * 32 + (k & 31)
  // halves (every 32-wide K-slab of a 16-row group is 1 KiB of contiguous memory for the LDS DMA).
  int a_slab;            // A planes are slab-major (written by launch_layernorm / launch_to_operand / a GEMM with out_slab)
  int out_slab;          // write out_split slab-major (it is the next GEMM's A operand)
  // LayerNorm folded into the GEMMs (F16 mode, pre-LN blocks; ltr_gemm.hip "LayerNorm fold").
  // Producer (the GEMM whose f32 output is the residual stream the next LayerNorm normalises):
  const float* ln_gamma = nullptr;   // [N] gamma of that LayerNorm; non-null selects the producer epilogue
  AOp ln_out{nullptr, nullptr};      // slab-major planes of out * gamma * 16: the next GEMM's A operand
  void* ln_stats_out = nullptr;      // float2 [N / 64][M]: (mean, M2) of every 64-column piece of every row
  // Consumer (A = a producer's ln_out):
  const void* ln_stats_in = nullptr; // float2 [K / 64][M]; non-null selects the consumer epilogue
  const float* ln_c = nullptr;       // [N] 16 * sum_k gamma_k W[n, k]; `bias` must then hold sum_k beta_k W[n, k] + b_n
  int ln_parts = 0;                  // K / 64
  int32_t* err_flag = nullptr;       // producer: device status word (bit 1 = operand left the fp16 range)
  // Post-LN blocks with the fold: `resid` holds the PRE-LayerNorm stream x of the previous producer and the residual to
  // add is LayerNorm(x), rebuilt in the epilogue from x, the row statistics pieces and the affine terms (F16 mode).
  const void* rln_stats = nullptr;   // float2 [N / 64][M]; non-null selects it
  const float* rln_gamma = nullptr;  // [N]
  const float* rln_beta = nullptr;   // [N]
  int rln_parts = 0;                 // N / 64
  // Scaled operands (the training step's split x split GEMMs, ltr_trainer.hip): the product is multiplied by
  // 1 / (split_scale(*osc_a) * split_scale(*osc_b)) - exact powers of two - before bias / ReLU / residual.  F16 mode,
  // not together with the LayerNorm fold.
  const float* osc_a = nullptr;      // device: max|x| of the A operand's source tensor
  const float* osc_b = nullptr;      // device: max|x| of the B operand's source tensor
  // Split-K (F16 mode, slab-major A, plain f32 output): K is cut into split_k equal parts (each a multiple of 32), part p
  // writes its partial product to out_f32 + p * M * N; the caller adds the parts up (fixed order).  Always on the 128 x 256 kernel.
  int split_k = 0;
  // Small-batch split-K (F16 mode, scoring path): scratch for the raw f32 partials [parts][M][N].  When given, launch_gemm
  // MAY cut K of a narrow-output GEMM of a small batch into parts (more workgroups, shorter operand streams) and finish
  // with splitk_epilogue_kernel, which adds the parts in part order (deterministic) and runs the epilogue.
  void* splitk_ws = nullptr;
  size_t splitk_ws_bytes = 0;
  // Row window (F16 mode): the launch computes rows [row0, row0 + M) of tensors that have `ldm` rows (0: row0 + M).  Every
  // pointer is the base of its WHOLE tensor (row-major tensors are indexed by global row, slab-major images and the
  // statistics arrays have pitch ldm).  launch_gemm uses it itself to give the rows that would start one more, mostly
  // empty, round of 128 x 256 tiles to the small-batch kernels ("tail rows").
  int row0 = 0;
  int ldm = 0;
  // LTR_F_ONE_PASS (F16 mode): multiply the hi plane of A only (the lo plane is neither streamed nor multiplied);
  // no_lo_out: do not store the lo planes of out_split / ln_out either (their only reader is another one-pass GEMM)
  // Row statistics combined ONCE per launch instead of in every tile's prologue (launch_row_stats_combine): float2 [ldm]
  // (mean, rstd) of the rows of A (consumer: ln_stats_in stays set and selects the epilogue) / of `resid` (rln_stats likewise).
  // Round 5: a [128 rows] x [12-16 pieces] gather in front of the first barrier of each of a row block's 3-16 column tiles is
  // redundant work on the critical path of every tile; large passes combine once (ltr_api.hip ChunkRun::layer).
  const void* ln_stats_comb = nullptr;
  const void* rln_stats_comb = nullptr;
  int one_pass = 0;
  int no_lo_out = 0;
  int keep_lo_out = 0;   // (caller's note to ChunkRun::gemm: this output's lo plane has a reader that is not a one-pass kernel)
  // F16 mode: store the outputs with the default cache policy instead of non-temporal.  The epilogue's nt stores stream a
  // pass's 1.4 GB of outputs past the caches (they are re-read long after they have left them: -1.4 % on the full-size call);
  // the outputs of a SMALL pass fit the 256 MiB Infinity Cache, where their reader - the next launch - finds them at twice
  // the HBM rate (ltr_api.hip ChunkRun::begin decides per pass; profiles/r06_store_policy.txt)
  int store_plain = 0;
};

// launchers (each in its own .hip file)
int launch_gemm(int wdtype, const GemmArgs& g, hipStream_t s);
// out[r] = (mean, rstd) of row r from its n_part 64-column pieces (mean, M2), parts = float2 [n_part][ldm]; rows [0, rows)
int launch_row_stats_combine(const void* parts, int n_part, int ldm, int rows, void* out, hipStream_t s);
// F16 mode: -1 = this shape runs on the 128 x 256 kernel (gemm_f16s_kernel), 0 / 1 = on a small-batch kernel (profiling)
int gemm_small_config(const GemmArgs& g);
// n_src < N: the source has n_src rows, the image is padded with zero rows up to N (N a multiple of 64 for launch_gemm)
int launch_pack_weight(const void* src_f16 /*[N,K]*/, void* dst_f16 /*[K/32][N][32]*/, int N, int K, hipStream_t s,
                       int n_src = -1);
// c[n] = 16 * sum_k gamma_k W[n,k], d[n] = sum_k beta_k W[n,k] + bias[n]  (W row-major fp16, the checkpoint layout)
int launch_ln_fold_coeff(const void* w_f16, const float* gamma, const float* beta, const float* bias, int N, int K,
                         float* c_out, float* d_out, hipStream_t s);

// out_op: F16 mode -> slab-major hi|lo image (GemmArgs::a_slab); F32 mode -> row-major f32 copy
int launch_layernorm(int wdtype, const float* x, const float* gamma, const float* beta, int M, int H,
                     float* out_f32 /*nullable, may alias x*/, AOp out_op /*nullable*/, hipStream_t s);
int launch_to_operand(int wdtype, const float* x, int M, int H, AOp out, hipStream_t s);
// compact the last-token rows (cu[i+1]-1-tok_off) of the f32 stream and of an operand buffer (a_src.hi may be
// null: the f32 rows only) to [n_req, H]
int launch_gather_last_rows(int wdtype, const int32_t* cu, int tok_off, int n_req, int H, const float* h_src, AOp a_src,
                            float* h_dst, AOp a_dst, hipStream_t s);

int launch_embed_gather(int wdtype, const int64_t* ids, const int32_t* cu, int N, int T, int tok_off,
                        const void* tok_table, int De, int vocab, const void* pos_table, int H, int pos_rows,
                        float* hidden_out, AOp tok_out /*only if De != H*/, int32_t* err_flag /*nullable*/, hipStream_t s);

// varlen causal attention over qkv [T, 3H] (q | k | v, heads of 64; f32, or fp16 hi|lo planes
// in the F16 mode), writes operand [T, H]
int launch_attention(int wdtype, AOp qkv, const int32_t* cu /*chunk-local, [n+1]*/, int n_req,
                     int T, int H, int n_heads, int32_t* blk_start /*scratch: (n_req+4)*4 + (T/64+n_req+1)*16 bytes*/, AOp out,
                     int build_blocks /*0: reuse the work list an earlier call built in blk_start for the same cu*/, hipStream_t s,
                     float* lse2 = nullptr /*[T, heads] log2-domain log-sum-exp of every query row (training), nullable*/,
                     size_t blk_bytes = 0 /*bytes behind blk_start when more than the minimum: (n_req+4)*4 + (T/32+n_req+1)*16
                                            lets small passes run the split-K/V variant (32-query blocks)*/,
                     int one_pass = 0 /*LTR_F_ONE_PASS: hi planes only, no lo output*/);

int launch_attention_bwd_planes(const float* x, size_t n /*multiple of 8*/, void* planes /*[2][n] halves*/, hipStream_t s);
// training: dqkv [T, 3H] of the attention block on the split-fp16 MFMA (ltr_attn.hip "attention BACKWARD")
int launch_attention_bwd(const float* qkv, const float* o, const float* dout, const float* lse2, const float* amax_do,
                         const float* amax_qkv, const int32_t* blk_start /*128-row work list of launch_attention*/, int n_req,
                         int T, int H, int n_heads, float scale, void* qkv_planes, void* do_planes, float* Dq, float* dqkv,
                         hipStream_t s);
int launch_attention_blocks(const int32_t* cu /*chunk-local, [n+1]*/, int n_req, int qb /*queries per block*/,
                            int32_t* blk_start /*scratch as in launch_attention*/, hipStream_t s);

// F16 mode, last layer of a scoring call: attention of the LAST query of every request only.  q f32 [n_req, H]
// (unscaled), kv = hi|lo planes [T, 2H] (k | v), out = row-major hi|lo planes [n_req, H]
int launch_attention_lastq(const float* q, AOp kv, const int32_t* cu /*chunk-local, [n+1]*/, int n_req, int H,
                           int n_heads, AOp out, hipStream_t s);

int launch_pool_head(int wdtype, const float* hidden, const int32_t* cu /*nullptr: rows are already compact*/, int tok_off,
                     int N, int H, int De,
                     int num_labels, int n_cmp /*labels that compete in the argmax: min(num_labels, vocab)*/,
                     const float* ln_w, const float* ln_b, const void* proj_out,
                     const void* score_w, float* scores_out, float* logits_out, hipStream_t s);
// class-mode label per row of logits f32 [n_rows, ld] (first maximum over the first n_cmp columns); logits_out f32
// [n_rows, num_labels] or nullptr
int launch_argmax_rows(const float* logits, int ld, int n_rows, int n_cmp, int num_labels, float* scores_out,
                       float* logits_out, hipStream_t s);

size_t rank_workspace_bytes(int64_t N);
int launch_rank_step(const float* scores, int32_t* pri, int32_t* idle, int32_t* runs, const uint32_t* tiebreak,
                     const int32_t* members, int N, int starv, int period, uint32_t flags, int32_t* perm_out, void* ws,
                     size_t ws_bytes, hipStream_t s);
int launch_queue_step(const float* scores, int32_t* pri, int32_t* idle, int32_t* runs, const uint32_t* tiebreak,
                      const int32_t* members, int N, int starv, int period, uint32_t flags, const int32_t* new_tokens,
                      const int32_t* new_seqs, const uint8_t* chunkable, int64_t token_budget, int64_t max_seqs,
                      int32_t* perm_out, int32_t* n_sel, uint8_t* ran, int32_t* granted, void* ws, size_t ws_bytes,
                      hipStream_t s);
int launch_age_update(const uint8_t* ran, const int32_t* ran_slots, int n_ran, int32_t* pri, int32_t* idle,
                      int32_t* runs, const int32_t* members, int N, hipStream_t s);
int launch_budget_prefix(const int32_t* perm, const int32_t* new_tokens, const int32_t* new_seqs,
                         const uint8_t* chunkable, int N, int64_t token_budget, int64_t max_seqs, int32_t* n_sel,
                         uint8_t* ran, int32_t* granted, hipStream_t s);
int launch_reserve_select(const int32_t* perm, const int32_t* n_sel, const uint8_t* state, const int32_t* phys,
                          const int32_t* logical, const int32_t* nrun, const int32_t* nswap, const int32_t* new_seqs,
                          int N, int64_t need_in, uint8_t* action, int32_t* n_exec, int32_t* blocks_required,
                          hipStream_t s);

// ---- device helpers ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float half_wave_sum(float v) {   // sum over the 32-lane half this lane is in
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// A value in a 32-bit register of its own.  hipcc vectorises neighbouring f32 operations into packed instructions
// (v_pk_mul_f32 ...) and, when an operand is the HIGH half of a 64-bit register (the y of a float2 that was loaded or merged as
// one value), reads it through op_sel.  One of those forms - op_sel:[0,1]: src1's lo lane from the high register - computes a
// wrong lo half in lanes 48-63 on MI355X while a library fp16 GEMM shares the CU (profiles/r06_rln_fault.txt; the build's ISA
// lint, isa_lint.py, rejects it).  Passing the scalar through here first makes the compiler broadcast it from the low half of
// a pair instead (op_sel_hi forms, which are correct).
__device__ __forceinline__ float own_reg(float v) {
  asm volatile("" : "+v"(v));
  return v;
}

// a = hi + lo with hi = fp16(a), lo = fp16(a - hi).
// The empty asm pins `a` as one materialised f32 value: without it, when `a` is a product
// x*y, hipcc contracts `a - hi` into v_fma_mix(x, y, -hi) (exact product) while the stored
// hi comes from the f32-ROUNDED product, and in near-tie cases the two disagree by one
// fp16 ulp with the wrong-signed lo (measured: isolated 2^-12 errors in the attention output).
// offset (halves) of element (row, col) in a slab-major operand image with ld rows
__device__ __forceinline__ size_t slab_off(int row, int col, int ld) {
  return ((size_t)(col >> 5) * ld + row) * 32 + (col & 31);
}

// Power-of-two scale that brings a tensor of magnitude amax to ~2^12 before it is split into fp16 hi + lo (training:
// gradients are 1e-3 ... 1e-9 in magnitude, far below fp16's normal range (6e-5), where hi would be a subnormal and lo
// flush to zero - the split would silently degrade to a few bits).  Scaling by 2^k is exact and is undone exactly in
// the GEMM epilogue (GemmArgs::osc_a / osc_b).  Elements down to 2^-15 of the tensor's maximum keep all 22 bits.
__device__ __forceinline__ float split_scale(float amax) {
  return amax > 0.f && amax < INFINITY ? ldexpf(1.f, 12 - ilogbf(amax)) : 1.f;
}
__device__ __forceinline__ void split_f16(float a, __half& hi, __half& lo) {
  asm volatile("" : "+v"(a));
  hi = __float2half_rn(a);
  lo = __float2half_rn(a - __half2float(hi));
}

// largest i in [0, n) with cu[i] <= t   (cu ascending, cu[0] <= t < cu[n])
__device__ __forceinline__ int find_request(const int32_t* __restrict__ cu, int n, int t) {
  int lo = 0, hi = n;  // answer in [lo, hi)
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (cu[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

}  // namespace ltr
