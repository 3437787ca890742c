/*
 * ltr_hip.h - C ABI of libltr_hip.so, the MI355X (gfx950) implementation of
 * vllm-ltr's per-step ranking hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / C++ types.
 * Every entry point names the reference interface it replaces (paths relative to
 * the reference checkout; see SURVEY.md section 8a/8b and INTEGRATION.md for the
 * reference-side binding).
 *
 * Conventions
 *  - All data pointers are DEVICE pointers unless the name ends in _host.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*); nothing
 *    synchronises except where stated.
 *  - Return value: 0 on success, negative LTR_E_* code on failure; never throws.
 *    ltr_last_error() returns a thread-local message for the last failure.
 *  - N = number of requests, T = total prompt tokens, L_i = cu_seqlens[i+1]-cu_seqlens[i].
 */
#ifndef LTR_HIP_H
#define LTR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LTR_ABI_VERSION 7

enum {
  LTR_OK = 0,
  LTR_E_INVAL = -22,   /* bad argument / unsupported shape                  */
  LTR_E_NOMEM = -12,   /* workspace too small                               */
  LTR_E_HIP = -5,      /* a HIP runtime call failed (see ltr_last_error)    */
  LTR_E_NODEV = -19,   /* no gfx950 device                                  */
  LTR_E_RANGE = -34    /* an activation left the fp16 range of the split path (ltr_status) */
};

/* weight element type of matrices and embedding tables */
enum { LTR_W_F32 = 0, LTR_W_F16 = 1 };

/* Shape of the predictor: HF OPTConfig fields of an OPTForSequenceClassification
 * (vllm/model_executor/models/opt.py:362-376; train/trainer.py:213-216). */
typedef struct ltr_model_desc {
  int32_t vocab_size;
  int32_t hidden_size;          /* H  */
  int32_t ffn_dim;              /* F  */
  int32_t num_layers;           /* Nl */
  int32_t num_heads;            /* head size H/num_heads must be 64 */
  int32_t word_embed_proj_dim;  /* De; != H adds project_in/out (opt.py:205-219) */
  int32_t pos_rows;             /* rows of the position table = max_position_embeddings + 2 (opt.py:43-53) */
  int32_t num_labels;           /* rows of score.weight (opt.py:374) */
  int32_t pre_ln;               /* do_layer_norm_before: 1 = 125m style, 0 = 350m style (opt.py:145-176) */
  int32_t weight_dtype;         /* LTR_W_F32: exact f32 MFMA path; LTR_W_F16: fp16 weights x (hi+lo) fp16
                                   split activations, f32 accumulate */
  int32_t flags;                /* LTR_F_* below; 0 = the production configuration (ABI 5: this word was not there before) */
} ltr_model_desc;

/* ltr_model_desc.flags (ltr_create; ltr_train_create ignores them).
 *  LTR_F_NO_LN_FOLD  the GEMMs are fed the bounded LayerNorm OUTPUT by separate LayerNorm launches instead of the folded
 *                    operand x * gamma * 16 (which must stay inside fp16: LTR_E_RANGE).  The twin handle a caller falls back
 *                    to when ltr_status reports LTR_E_RANGE for a checkpoint with massive activations (plugin.py).
 *  LTR_F_NO_LANES, LTR_F_LANES_UNPROBED   accepted and ignored since ABI 6: rounds 4-5 ran mid-sized calls as two halves on two
 *                    streams ("lanes"); measured again in round 6 (-3.8 % at k = 64 on one predictor, +8.5 % on the other,
 *                    profiles/r06_lanes_after.txt) and removed with ltr_lane_calls / ltr_lane_probe.  A scoring call uses `stream` only.
 *  LTR_F_ONE_PASS    F16 mode with ONE fp16 MFMA pass per product, in the GEMMs and in the attention (q, k, v, p as plain
 *                    fp16): activations rounded to fp16 (the `lo` plane is neither loaded nor multiplied, and not stored
 *                    where every reader runs one pass), f32 accumulate - the arithmetic of the reference's own GPU path (fp16 model,
 *                    vllm/config.py:906-943; train/trainer.py:213-216).  Scores move by ~2e-3 against the fp32 predictor:
 *                    OUTSIDE the 1e-4 contract of the default mode; opt-in, reported as its own number by bench.py.
 */
enum { LTR_F_NO_LN_FOLD = 1, LTR_F_NO_LANES = 2, LTR_F_ONE_PASS = 4, LTR_F_LANES_UNPROBED = 8 };

/* Order of the device pointers handed to ltr_create (HF tensor names in comments).
 * Matrices / tables are row-major [out, in] in `weight_dtype`; biases, LayerNorm
 * affine terms are always f32.  q,k,v are stacked in that order (opt.py:411-417). */
enum {
  LTR_WT_EMBED_TOKENS = 0,   /* model.decoder.embed_tokens.weight      [V, De]        */
  LTR_WT_EMBED_POS,          /* model.decoder.embed_positions.weight   [pos_rows, H]  */
  LTR_WT_PROJECT_IN,         /* model.decoder.project_in.weight        [H, De] | NULL */
  LTR_WT_PROJECT_OUT,        /* model.decoder.project_out.weight       [De, H] | NULL */
  LTR_WT_FINAL_LN_W,         /* model.decoder.final_layer_norm.weight  [H] f32 | NULL */
  LTR_WT_FINAL_LN_B,         /* model.decoder.final_layer_norm.bias    [H] f32 | NULL */
  LTR_WT_SCORE,              /* score.weight                           [num_labels, De] */
  LTR_WT_GLOBAL_COUNT
};
enum {
  LTR_WL_QKV_W = 0,  /* self_attn.{q,k,v}_proj.weight stacked  [3H, H] */
  LTR_WL_QKV_B,      /* ... bias stacked                        [3H] f32 */
  LTR_WL_OUT_W,      /* self_attn.out_proj.weight               [H, H]  */
  LTR_WL_OUT_B,      /*                                         [H] f32 */
  LTR_WL_LN1_W,      /* self_attn_layer_norm.weight             [H] f32 */
  LTR_WL_LN1_B,
  LTR_WL_FC1_W,      /* fc1.weight                              [F, H]  */
  LTR_WL_FC1_B,      /*                                         [F] f32 */
  LTR_WL_FC2_W,      /* fc2.weight                              [H, F]  */
  LTR_WL_FC2_B,      /*                                         [H] f32 */
  LTR_WL_LN2_W,      /* final_layer_norm.weight (per layer)     [H] f32 */
  LTR_WL_LN2_B,
  LTR_WL_COUNT
};
/* total pointers = LTR_WT_GLOBAL_COUNT + num_layers * LTR_WL_COUNT */

typedef struct ltr_model* ltr_handle;

int ltr_abi_version(void);
const char* ltr_last_error(void);

/* Replaces loading the AUX engine's model: AUXLLM(...) -> Worker.load_model ->
 * OPTForSequenceClassification.load_weights (vllm/engine/llm_engine.py:224-240,
 * vllm/model_executor/models/opt.py:411-444).  The library keeps the POINTERS (the
 * caller owns the weight memory and must keep it alive until ltr_destroy).  With
 * LTR_W_F16 the dense-layer weights (QKV / out_proj / fc1 / fc2 / project_in) are also
 * copied once into a library-owned GEMM-friendly layout: the copy kernels run on `stream`
 * (so they are ordered after the caller's own uploads on that stream) and ltr_create
 * synchronises `stream` before returning.  The handle remembers the device that owns
 * weights[0]; every handle-based call makes that device current for its launches and
 * restores the caller's device afterwards, so a handle of device 1 can be used while
 * device 0 is current.  (Handle-less entry points - ltr_rank_step etc. - launch on the
 * current device like any HIP launch: make the device of their pointers current first.) */
int ltr_create(const ltr_model_desc* desc, const void* const* weights, int32_t n_weights,
               void* stream, ltr_handle* out);
int ltr_destroy(ltr_handle h);

/* Workspace sizing.  kind: */
enum { LTR_WS_SCORE = 0, LTR_WS_RANK = 1 };
size_t ltr_workspace_bytes(ltr_handle h /* may be NULL for LTR_WS_RANK */, int32_t kind,
                           int64_t N, int64_t T);
/* Tokens processed per internal pass of ltr_score (request-aligned chunks keep the
 * activations of a pass resident in the 256 MiB Infinity Cache).  0 restores the default. */
int ltr_set_chunk_tokens(ltr_handle h, int32_t chunk_tokens);
/* The predictor forward for a flat varlen batch: replaces one drain of the AUX
 * engine, i.e. AUXLLMEngine.obtain_aux_scores' step loop (vllm/engine/
 * aux_llm_engine.py:398-405) = ModelRunner.execute_model (vllm/worker/
 * model_runner.py:827-877) = OPTForSequenceClassification.forward +
 * compute_logits + sample's logits[:,0] (opt.py:378-409).
 *   token_ids   int64 [T]   flat prompts (model_runner.py:711-716 input_tokens)
 *   cu_seqlens  int32 [N+1] prefix sums of L_i (model_runner.py:383-395 seq_start_loc);
 *               positions 0..L_i-1 and the last-token selection cu[i+1]-1
 *               (model_runner.py:592-593) are derived from it on the device
 *   cu_seqlens_host  the same array in host memory (the caller built it there);
 *               lets the library cut request-aligned chunks without a device sync.
 *               NULL: the library copies it back (one synchronising hipMemcpy).
 *   max_len     the caller's bound on L_i (ModelRunner's max_prompt_len, model_runner.py:383-395);
 *               a request longer than it is rejected with LTR_E_INVAL.  <= 0: no bound beyond
 *               the position table.
 *   scores_out  f32 [N]: rank mode logits[:,0] (opt.py:408); class mode (num_labels>1)
 *               float(argmax_j logits[:, j]) (opt.py:394-395)
 *   logits_out  f32 [N, num_labels] or NULL (raw head output, for tests/telemetry)  */
int ltr_score(ltr_handle h, const int64_t* token_ids, const int32_t* cu_seqlens,
              const int32_t* cu_seqlens_host, int32_t N, int32_t T, int32_t max_len,
              float* scores_out, float* logits_out, void* workspace, size_t ws_bytes,
              void* stream);

/* Deferred input errors.  ltr_score is asynchronous, so a token id outside [0, vocab_size) -
 * on which the reference's F.embedding raises (vocab_parallel_embedding.py:95-106) - cannot be
 * reported by the call that meets it: the embedding kernel flags it (and reads row 0 / the last
 * row instead of faulting; the scores of that call are then meaningless).  ltr_status
 * synchronises `stream`, returns LTR_E_INVAL if any forward since the previous ltr_status saw
 * such an id (LTR_OK otherwise) and clears the flag.  Call it wherever the scores are read.
 * LTR_E_RANGE: the residual stream of an F16-mode pre-LN model exceeded what the LayerNorm-fold
 * operand can carry in fp16 (|x * gamma| > 4094; the reference's own fp16 GPU path overflows at
 * |x| > 65504): the scores of that call are invalid; a handle created with LTR_F_NO_LN_FOLD
 * feeds the GEMMs the bounded LayerNorm output instead (MI355XRanker re-scores the batch on such
 * a twin handle and carries on). */
int ltr_status(ltr_handle h, void* stream);

/* Same forward stopped after `n_layers` decoder layers (n_layers < 0: all), writing the
 * f32 hidden states [T, H] (before final LN / project_out).  Test / debugging hook for
 * per-layer parity against the oracle (opt.py:145-176).  T must fit one chunk. */
int ltr_forward_hidden(ltr_handle h, const int64_t* token_ids, const int32_t* cu_seqlens,
                       const int32_t* cu_seqlens_host, int32_t N, int32_t T, int32_t max_len,
                       int32_t n_layers, float* hidden_out, void* workspace, size_t ws_bytes,
                       void* stream);

/* The two HBM-bound ends of the forward, individually addressable for roofline work.
 *
 * Embedding gather: OPTDecoder.forward's first three lines (opt.py:241-245),
 * VocabParallelEmbedding.forward (layers/vocab_parallel_embedding.py:95-106) and
 * OPTLearnedPositionalEmbedding (+2 offset, opt.py:43-53).
 *   De == H: hidden_out[t,:] = E_tok[ids[t],:] + E_pos[pos(t)+2,:]           (f32 [T,H])
 *   De != H: hidden_out[t,:] = E_pos[pos(t)+2,:]; tok_out = E_tok rows, to be multiplied
 *            by project_in (f32 [T,De] in the F32 mode; fp16 hi|lo planes [2][T,De] in F16) */
int ltr_embed_gather(ltr_handle h, const int64_t* token_ids, const int32_t* cu_seqlens,
                     int32_t N, int32_t T, float* hidden_out, void* tok_out, void* stream);

/* Pool + final LayerNorm (or project_out) + score head: _prune_hidden_states
 * (layers/logits_processor.py:74-79), the final LN / project_out of opt.py:259-262
 * applied to the N selected rows only (both are per-token maps), _get_logits
 * (logits_processor.py:61-71) with score.weight (opt.py:374), class-mode argmax
 * (opt.py:394-395).  hidden: f32 [T, H] decoder output. */
int ltr_pool_head(ltr_handle h, const float* hidden, const int32_t* cu_seqlens, int32_t N,
                  float* scores_out, float* logits_out, void* stream);

/* The varlen causal self-attention of a decoder layer alone (OPTAttention.forward, opt.py:92-102,
 * with kv_cache None: attention/backends/rocm_flash_attn.py:244-290 on the GPU, torch_sdpa.py:138-178 in the CPU oracle):
 * out[t] = softmax_{s <= t, same request}(q[t] . k[s] / 8) v[s] per head of 64, requests delimited by cu_seqlens.
 *   qkv  LTR_W_F16: fp16 hi | lo planes [2][T, 3H] (q | k | v per row; value = hi + lo), LTR_W_F32: f32 [T, 3H]
 *   out  LTR_W_F16: fp16 hi | lo planes [2][T, H] row-major,                             LTR_W_F32: f32 [T, H]
 *   workspace: (N + 4) * 4 + (T / 64 + N + 1) * 16 bytes (the work list of query blocks).
 * Test / roofline hook: ltr_score runs the same kernel inside every layer. */
int ltr_attention(ltr_handle h, const void* qkv, const int32_t* cu_seqlens, int32_t N, int32_t T, void* out,
                  void* workspace, size_t ws_bytes, void* stream);

/* One ranking step over the queued requests: the body of
 * Scheduler._get_opt_ordered_requests after scoring (vllm/core/scheduler.py:984-998).
 *
 * Device-resident queue state: scores / pri / idle / runs are SLOT arrays that persist across
 * scheduler steps (a request keeps its slot from arrival to completion; new slots start at
 * pri = idle = runs = 0, scheduler.py:372-374).  `members` int32 [N] lists the slot of every
 * queued request in the order list(waiting)+list(running)+list(swapped) (scheduler.py:985,996);
 * position i in `members` is what "input index", `tiebreak[i]` and perm_out refer to.
 * members NULL: slot == position (the arrays ARE the concatenation).
 *   1. if starv != -1, per request (in place, scheduler.py:986-993):
 *        idle >= starv            -> pri = -1, idle = 0, runs = period
 *        elif pri == -1, runs <= 0 -> pri = 0
 *   2. perm_out = stable ascending sort of the N requests by
 *        (pri [only if starv != -1 or LTR_RANK_USE_PRI], -score, tiebreak)
 *      with -0.0 == +0.0; tiebreak NULL = input index, which is exactly Python's stable
 *      sorted() over list(waiting)+list(running)+list(swapped) (scheduler.py:996,998).
 *      For the tpt/rtpt orders (scheduler.py:948,961) pass the rank of request_id under
 *      string comparison as tiebreak; LTR_RANK_ASCENDING sorts by +score (ropt/rtpt, :1015).
 *      tiebreak values must be < 2^31.  NaN scores sort after every number.
 *   scores f32, pri/idle/runs int32: slot arrays; perm_out int32 [N], perm_out[k] = position
 *   (index into members) of the k-th request to schedule.
 *   workspace: ltr_workspace_bytes(NULL, LTR_WS_RANK, N, 0) bytes (only touched above 12,288 requests). */
enum { LTR_RANK_USE_PRI = 1, LTR_RANK_ASCENDING = 2 };
int ltr_rank_step(const float* scores, int32_t* pri, int32_t* idle, int32_t* runs,
                  const uint32_t* tiebreak, const int32_t* members, int32_t N, int32_t starv,
                  int32_t period, uint32_t flags, int32_t* perm_out, void* workspace,
                  size_t ws_bytes, void* stream);

/* Post-schedule aging, the loop at the end of Scheduler._general_schedule
 * (vllm/core/scheduler.py:1358-1365) over the N queued requests (members as above):
 * ran -> (pri == -1: runs -= 1), idle = 0; else idle += 1.  Which requests ran is given either
 * as ran u8 [N] by position, or (ran NULL) as ran_slots int32 [n_ran], the ASCENDING list of the
 * slots of `running_this_step` (<= max_num_seqs entries instead of an N-byte mask). */
int ltr_age_update(const uint8_t* ran, const int32_t* ran_slots, int32_t n_ran, int32_t* pri,
                   int32_t* idle, int32_t* runs, const int32_t* members, int32_t N, void* stream);

/* A whole steady scheduler step on the device-resident queue in two launches: ltr_rank_step, then
 * ltr_budget_prefix over the resulting order (same arguments, below) and ltr_age_update with the
 * selection as `ran` - i.e. scheduler.py:984-998, :1137-1211 and :1358-1365 with nothing handed
 * back to the host in between.  new_tokens / new_seqs / chunkable are indexed by position;
 * outputs as in ltr_rank_step and ltr_budget_prefix (ran_out / granted_out nullable). */
int ltr_queue_step(const float* scores, int32_t* pri, int32_t* idle, int32_t* runs,
                   const uint32_t* tiebreak, const int32_t* members, int32_t N, int32_t starv,
                   int32_t period, uint32_t flags, const int32_t* new_tokens,
                   const int32_t* new_seqs, const uint8_t* chunkable, int64_t token_budget,
                   int64_t max_num_seqs, int32_t* perm_out, int32_t* n_selected_out,
                   uint8_t* ran_out, int32_t* granted_out, void* workspace, size_t ws_bytes,
                   void* stream);

/* Next row in scope: the hidden-state learning-to-rank head of
 * vllm/model_executor/predictor.py - FCModel (:10-43: optional input LayerNorm, then
 * activation(Linear) per layer), OutputLayer (:91-125) and LTRModel.score (:78-89, sum over
 * d_output) - applied to the backbone's hidden states at the selected tokens (opt.py:250-255;
 * loaded in model_loader/loader.py:234-241 from PredictorConfig, config_predictor.py:78-117).
 * Weight pointers: [0] input_norm.weight f32 | NULL, [1] input_norm.bias f32 | NULL, then per
 * dense layer (the n_fc FC layers, then output_layer.w_1): weight [out, in] in weight_dtype,
 * bias [out] f32. */
enum { LTR_ACT_NONE = 0, LTR_ACT_RELU = 1, LTR_ACT_SIGMOID = 2, LTR_ACT_TANH = 3, LTR_ACT_GELU = 4, LTR_ACT_SILU = 5 };
typedef struct ltr_head_desc {
  int32_t n_features;         /* ModelConfig.n_features */
  int32_t n_fc;               /* len(FCConfig.sizes), 0 when fc_model is null */
  int32_t fc_sizes[8];
  int32_t input_norm;         /* FCConfig.input_norm */
  int32_t activation;         /* FCConfig.activation -> LTR_ACT_* */
  int32_t d_output;           /* PostModelConfig.d_output */
  int32_t output_activation;  /* PostModelConfig.output_activation -> LTR_ACT_* */
  int32_t weight_dtype;       /* LTR_W_F32 | LTR_W_F16 */
} ltr_head_desc;
typedef struct ltr_head_model* ltr_head_handle;
int ltr_head_create(const ltr_head_desc* desc, const void* const* weights, int32_t n_weights,
                    ltr_head_handle* out);
int ltr_head_destroy(ltr_head_handle h);
/* hidden f32 [rows, n_features]; row_index int32 [N] selects the rows to score (NULL: rows 0..N-1,
 * the index_select of opt.py:252-254); scores_out f32 [N]. */
int ltr_head_score(ltr_head_handle h, const float* hidden, const int32_t* row_index, int32_t N,
                   float* scores_out, void* stream);

/* Per-kernel-class timing for the roofline report (bench.py).  When enabled, ltr_score /
 * ltr_forward_hidden bracket every launch with HIP events on the caller's stream (no host
 * sync); ltr_profile_read synchronises on the last event, sums event-to-event durations
 * per class and optionally resets.  work = algorithmic FLOPs (GEMM: 2*M*N*K; ATTN: causal
 * 2*L^2*H per layer and request) or algorithmic bytes (EMBED, LN, POOL) as defined in
 * DESIGN.md. */
/* LTR_K_GEMM: launches of the large-tile kernel (gemm_f16s_kernel / gemm_f32_kernel); LTR_K_GEMM_SMALL: the small-batch
 * kernels (compact last-token rows of a pass, small calls) */
enum { LTR_K_GEMM = 0, LTR_K_ATTN = 1, LTR_K_EMBED = 2, LTR_K_LN = 3, LTR_K_POOL = 4, LTR_K_GEMM_SMALL = 5, LTR_K_COUNT = 6 };
typedef struct ltr_profile_stats {
  double ms[8];        /* summed kernel time per class   */
  double work[8];      /* summed algorithmic FLOPs/bytes */
  int64_t launches[8];
} ltr_profile_stats;
int ltr_profile_enable(ltr_handle h, int32_t on);
int ltr_profile_read(ltr_handle h, ltr_profile_stats* out, int32_t reset);

/* Next row in scope: the selection the budget walk of
 * Scheduler._general_schedule makes over the ranked order (scheduler.py:1137-1211, with
 * _get_num_new_tokens :1867-1888 and SchedulingBudget.can_schedule :51-55), as one scan:
 * request k of perm is selected iff for every j <= k: new_tokens_j > 0,
 * sum_{i<j} new_tokens_i < token_budget (a chunkable group is granted
 * min(need, remaining)), sum_{i<=j} new_seqs_i <= max_num_seqs, and a group that is not
 * chunkable fits whole; the walk stops at the first k that fails.
 *   new_tokens / new_seqs  int32 [N] indexed by request (not by rank): un-chunked
 *     _get_num_new_tokens and get_max_num_running_seqs()
 *   chunkable u8 [N] or NULL: 1 iff the group has exactly one sequence in the walked status
 *     (`len(seqs) == 1`, scheduler.py:1884).  NOT the same as new_seqs == 1: a WAITING prompt with
 *     best_of > 1 has one sequence and new_seqs = best_of (sequence.py:500-504).  NULL: new_seqs <= 1.
 *   n_selected_out int32 [1]; ran_out u8 [N] or NULL (1 = selected; feeds ltr_age_update);
 *   granted_out int32 [N] or NULL (tokens granted this step, 0 if not selected). */
int ltr_budget_prefix(const int32_t* perm, const int32_t* new_tokens, const int32_t* new_seqs,
                      const uint8_t* chunkable, int32_t N, int64_t token_budget, int64_t max_num_seqs,
                      int32_t* n_selected_out, uint8_t* ran_out, int32_t* granted_out,
                      void* stream);

/* Second half of SURVEY.md 8f-1: which requests Scheduler.reserve_free_blocks
 * (scheduler.py:1376-1452) evicts when the selection does not fit the free KV blocks, as one
 * reversed prefix sum over the ranked order (perm[0 .. n_selected) = the selection, i.e.
 * `pinned_requests`; the rest = `priority_requests`):
 *   state u8 [N]: 0 waiting, 1 has RUNNING sequences, 2 has SWAPPED sequences
 *   phys / logical / nrun / nswap int32 [N]: len(_get_physical_blocks), len(logical_token_blocks),
 *     num_seqs(RUNNING), num_seqs(SWAPPED), all indexed by request
 *   n_selected int32 [1] (device; e.g. ltr_budget_prefix's n_selected_out)
 *   new_seqs NULL:   need = need_in  (num_blocks_needed - free GPU blocks + watermark, :1384-1388)
 *   new_seqs int32 [N]: the kernel accumulates gpu_block_required over the selection like
 *     :1137-1211 (running +new_seqs, swapped +phys+nswap, waiting +logical), writes it to
 *     blocks_required_out (nullable) and uses need = required - need_in, need_in = free - watermark
 *   action_out u8 [N]: 0 keep, 1 unselected running request swapped out (:1400-1420),
 *     2 selected running request put back and preempted, 3 selected swapped / waiting request put
 *     back (:1422-1447);  n_exec_out int32 [1] = selected requests that still execute. */
int ltr_reserve_select(const int32_t* perm, const int32_t* n_selected, const uint8_t* state,
                       const int32_t* phys, const int32_t* logical, const int32_t* nrun,
                       const int32_t* nswap, const int32_t* new_seqs, int32_t N, int64_t need_in,
                       uint8_t* action_out, int32_t* n_exec_out, int32_t* blocks_required_out,
                       void* stream);

/* Next row in scope: the ListMLE loss the reference fine-tunes the predictor with
 * (train/allrank/models/losses/listMLE.py:23-54, called by train/trainer.py:125-150) and its gradient
 * w.r.t. the predictions.  y_pred, y_true f32 [B, S]; shuffle int32 [S] = the random permutation of :33
 * (an input, so results are reproducible; equal labels keep the shuffled order); items with
 * y_true == pad_value are masked.  loss_out f32 [1] = mean over slates; row_loss_out f32 [B] (scratch and
 * per-slate losses); grad_out f32 [B, S] or NULL.  S <= 4096. */
int ltr_listmle(const float* y_pred, const float* y_true, const int32_t* shuffle, int32_t B, int32_t S,
                float eps, float pad_value, float* loss_out, float* row_loss_out, float* grad_out,
                void* stream);

/* The reference's other ranking loss, `--loss neuralNDCG` (train/trainer.py:127-128 -> train/allrank/models/losses/
 * neuralNDCG.py:27-87, deterministic variant = the defaults the trainer calls it with; NeuralSort and Sinkhorn scaling of
 * losses/loss_utils.py:24-83, ideal DCG of models/metrics.py:89-135) and its gradient w.r.t. the predictions (what autograd
 * yields through the unrolled Sinkhorn rounds).  y_pred, y_true f32 [B, S]; items with y_true == pad_value are masked;
 * temperature = NeuralSort's tau (1); k = rank the metric is cut at (<= 0 or > S: the slate length, the reference's None).
 * loss_out f32 [1] = -mean NDCG over the slates whose ideal DCG is not 0 (none: 0 and a zero gradient); row_ndcg_out f32 [B];
 * grad_out f32 [B, S] or NULL.  2 <= S <= 1024 (S = 1: LTR_E_INVAL - the reference raises IndexError, loss_utils.py:70).
 * The Sinkhorn loop stops, as the reference's, after the first round in which every slate OF THE BATCH is within 1e-6.
 * Gains are 2^label - 1 in f32: a label of 128 or more makes the loss NaN, here as in the reference.
 * workspace: ltr_neuralndcg_workspace_bytes(B, S) bytes (0 for an unsupported shape), device. */
size_t ltr_neuralndcg_workspace_bytes(int32_t B, int32_t S);
int ltr_neuralndcg(const float* y_pred, const float* y_true, int32_t B, int32_t S, float temperature, int32_t k,
                   float pad_value, float* loss_out, float* row_ndcg_out, float* grad_out, void* workspace,
                   size_t ws_bytes, void* stream);

/* The other half of SURVEY.md 8f-4: ONE optimisation step of the predictor's fine-tuning loop,
 * train/trainer.py:122-165 - forward of the OPTForSequenceClassification predictor on a slate of prompts
 * (prefill_predictor.py:76-79), loss_func(outputs.view(1, -1), labels) with listMLE / neuralNDCG / MSELoss or CrossEntropyLoss over
 * num_labels classes (:125-157), loss.backward(), torch.optim.Adam(lr, weight_decay).step() (:122,161-165: L2 decay added
 * to the gradient, bias-corrected moments), optimizer.zero_grad().  Parameters, activations, gradients and moments are f32
 * (fp32 master weights, :99-101); the dense layers multiply on the fp16 matrix cores with both operands split into two
 * fp16 terms (f32-grade products, f32 accumulation; LTR_TRAIN_F32=1 in the environment of ltr_train_create selects the
 * exact-f32 MFMA instead).
 * ltr_train_create COPIES the f32 weights (same pointer order as ltr_create, every tensor f32) into library-owned
 * parameter / gradient / moment buffers; ltr_train_read copies the current value of a tensor out, for evaluation and
 * for writing the fine-tuned checkpoint (trainer.py:213-216 saves it .half()). */
enum { LTR_LOSS_LISTMLE = 0, LTR_LOSS_MSE = 1, LTR_LOSS_CROSSENTROPY = 2, LTR_LOSS_NEURALNDCG = 3 };
enum { LTR_TRAIN_PREC_DEFAULT = 0, LTR_TRAIN_PREC_SPLIT = 1, LTR_TRAIN_PREC_F32 = 2 };
typedef struct ltr_train_config {
  float lr;            /* trainer.py --lr (2e-5) */
  float beta1, beta2;  /* torch.optim.Adam defaults 0.9, 0.999 */
  float eps;           /* 1e-8 */
  float weight_decay;  /* trainer.py --wc (0.01) */
  int32_t loss;        /* LTR_LOSS_* (trainer.py --loss) */
  float listmle_eps;   /* allrank DEFAULT_EPS 1e-10 */
  float pad_value;     /* allrank PADDED_Y_VALUE -1 */
  float dropout;       /* HF OPT config.dropout (0.1 in train mode) after out_proj and fc2; 0 disables.  Masks come from
                          a counter-based hash of (seed, step, layer, site, element) - torch's RNG stream cannot be matched */
  int32_t precision;   /* LTR_TRAIN_PREC_*: how the dense layers multiply.  DEFAULT (0) = split-fp16 unless the environment
                          of ltr_train_create holds LTR_TRAIN_F32=1 (the A/B switch of DESIGN.md 6.4); SPLIT / F32 select a
                          path per handle, whatever the environment says.  (Occupies what was padding in front of `seed`.) */
  uint64_t seed;
} ltr_train_config;
typedef struct ltr_trainer* ltr_train_handle;
int ltr_train_create(const ltr_model_desc* desc, const void* const* weights, int32_t n_weights,
                     const ltr_train_config* cfg, void* stream, ltr_train_handle* out);
int ltr_train_destroy(ltr_train_handle h);
size_t ltr_train_workspace_bytes(ltr_train_handle h, int64_t N, int64_t T);
/* One step on a slate of N prompts (flat ids / cu_seqlens as ltr_score; the whole slate is one pass).
 *   labels   f32 [N]: listMLE / neuralNDCG / mse targets, or class indices for crossentropy (neuralNDCG: 2 <= N <= 1024)
 *   shuffle  int32 [N]: the random permutation of listMLE.py:33 (listMLE only)
 *   apply_update 1: the full step; 0: gradients only, no Adam step; -1: evaluation forward (predictor.model.eval(),
 *            trainer.py:171-190: no dropout, no loss, no backward - labels / shuffle / loss_out may be NULL, logits_out is
 *            the result)
 *   loss_out f32 [1] (device); logits_out f32 [N, num_labels] or NULL: the outputs BEFORE the update */
int ltr_train_step(ltr_train_handle h, const int64_t* token_ids, const int32_t* cu_seqlens,
                   const int32_t* cu_seqlens_host, int32_t N, int32_t T, const float* labels,
                   const int32_t* shuffle, int32_t apply_update, float* loss_out, float* logits_out,
                   void* workspace, size_t ws_bytes, void* stream);
/* Copies parameter (what = 0) or gradient (what = 1) tensor `index` (ltr_create's index space; QKV stacked) into
 * dst (device, f32, `capacity` floats) on `stream`; *count_out = its element count (dst NULL: size query). */
int ltr_train_read(ltr_train_handle h, int32_t index, int32_t what, float* dst, size_t capacity,
                   size_t* count_out, void* stream);

/* The attention block of the training step ALONE, forward + backward, on the kernels ltr_train_step runs (split-fp16
 * MFMA; what autograd does for OPTAttention in train/trainer.py:147-159): for every request r and head,
 *   out = softmax_causal(q k^T / 8) v    and, given dout = dLoss/dout,    dqkv = (dq | dk | dv).
 *   qkv f32 [T, 3 hidden] (q | k | v per token, head-major inside each third, as QKVParallelLinear lays them out,
 *   opt.py:92-102), dout f32 [T, hidden], out f32 [T, hidden], dqkv f32 [T, 3 hidden]; all device.  head size 64.
 *   workspace: ltr_train_attention_workspace_bytes(num_heads, N, T) bytes. */
size_t ltr_train_attention_workspace_bytes(int32_t num_heads, int64_t N, int64_t T);
int ltr_train_attention(int32_t num_heads, const float* qkv, const float* dout, const int32_t* cu_seqlens, int32_t N,
                        int32_t T, float* out, float* dqkv, void* workspace, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LTR_HIP_H */
