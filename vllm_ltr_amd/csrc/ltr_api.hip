// C-ABI of libltr_hip.so (include/ltr_hip.h) and the host-side orchestration of one
// predictor forward: request-aligned chunks of the flat varlen batch are pushed through
// embed -> Nl x (LN, QKV GEMM, varlen causal attention, out_proj, LN, fc1+ReLU, fc2) ->
// pool + head, all enqueued on the caller's stream without host synchronisation.
//
// Reference control flow being replaced: AUXLLMEngine.obtain_aux_scores' step loop
// (vllm/engine/aux_llm_engine.py:398-405) -> Worker.execute_model -> ModelRunner.execute_model
// (vllm/worker/model_runner.py:827-877) -> OPTForSequenceClassification (opt.py:378-409).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "ltr_internal.h"

namespace ltr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace ltr

using namespace ltr;

struct ProfRec {
  hipEvent_t start, stop;
  int kind;
  double work;
};

struct ltr_model {
  ltr_model_desc d;
  std::vector<const void*> w;
  int chunk_tokens;
  int device = 0;                  // ordinal of the device that owns the weights (made current around every launch)
  int32_t* err_flag = nullptr;     // device word: bit 0 = a token id outside [0, vocab) was seen (ltr_status)
  bool prof_on = false;
  bool dbg_attn_valu = false;   // LTR_DEBUG_ATTN_VALU=1: f32 VALU attention inside the F16 mode (A/B for accuracy work)
  std::mutex prof_mu;              // guards prof / prof_free: calls on one handle may come from several host threads
  std::vector<ProfRec> prof;       // records in use
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;
  // F16 mode: library-owned slab-major images of the GEMM weights (launch_pack_weight), same index
  // space as w; F32 mode: wg == w (row-major, used as is)
  void* packed = nullptr;
  std::vector<const void*> wg;
  // LayerNorm fold (pre-LN, F16): per layer c / d vectors of the QKV and fc1 GEMMs (launch_ln_fold_coeff)
  bool ln_fold = false;
  float* fold = nullptr;
  std::vector<const float*> fold_c_qkv, fold_d_qkv, fold_c_fc1, fold_d_fc1;
  // F16 mode: the LAST layer's qkv_proj as two images, q rows [0, H) and k | v rows [H, 3H) - a scoring call needs
  // that layer's Q for the last token of each request only (ChunkRun::layer; LTR_NO_LASTQ=1 switches it off)
  bool lastq = false;
  const void* last_q_w = nullptr;
  const void* last_kv_w = nullptr;
  // F16 mode, head on the matrix cores (run_forward "GEMM head"): slab-major images of project_out [De, H] (350m style)
  // and of score.weight padded with zero rows to a multiple of 64 labels [head_lpad, De] (class mode with many labels:
  // train/train.sh:19-44 trains 82 / 820 / 8192-label heads)
  const void* proj_out_w = nullptr;
  const void* head_w = nullptr;
  int head_lpad = 0;
  bool one_pass = false;           // LTR_F_ONE_PASS: the GEMMs / attention multiply the hi plane only
  ~ltr_model() {
    if (packed) (void)hipFree(packed);
    if (fold) (void)hipFree(fold);
    if (err_flag) (void)hipFree(err_flag);
    for (auto& r : prof) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    for (auto& e : prof_free) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  }
  const void* gw(int i) const { return w[i]; }
  const void* lw(int layer, int i) const { return w[LTR_WT_GLOBAL_COUNT + layer * LTR_WL_COUNT + i]; }
  const void* gemm_gw(int i) const { return wg[i]; }
  const void* gemm_lw(int layer, int i) const { return wg[LTR_WT_GLOBAL_COUNT + layer * LTR_WL_COUNT + i]; }
};

namespace {

// Makes the handle's device current for the duration of a call and restores the caller's device: kernel
// launches and hipMalloc go to the CURRENT device, not to the device of their pointer arguments.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// tokens per pass: 3 x 65,536, so that every GEMM of a layer is a whole number of 512-workgroup rounds.
// Measured on the 8k-queue call: 247 ms at 64k tokens per pass, 242 at 128k, 240 at 192k-384k, 242 for a
// single 709k-token pass (LayerNorm slows down once the residual stream of a pass outgrows the 256 MiB
// Infinity Cache, attention keeps gaining from fewer, longer launches).  5.4 GB of workspace.
constexpr int DEFAULT_CHUNK_TOKENS = 196608;

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// rows of padded class logits held at a time by the GEMM head (64 MB of f32 at most, at least 256 rows)
inline int head_block_rows(int lpad) {
  static const int forced = [] { const char* e = getenv("LTR_HEAD_BLOCK_ROWS"); return e ? atoi(e) : 0; }();   // tests: small blocks
  if (forced >= 128) return forced / 128 * 128;
  const int64_t r = ((int64_t)64 << 20) / ((int64_t)(lpad > 0 ? lpad : 1) * 4);
  return (int)(r < 256 ? 256 : (r > (1 << 20) ? (1 << 20) : r)) / 128 * 128;
}

// LayerNorm fold: from this many rows on the row statistics a producer GEMM wrote are combined ONCE (one ~10-us launch) instead of
// in the prologue of every consuming tile (12-16 strided loads per row in front of the tile's first barrier, repeated by each
// of the row block's 3-16 column tiles).  LTR_STATS_COMB_MIN (lab; a huge value switches it off).  profiles/r05_rln_probe.txt
inline int64_t stats_comb_min_rows() {
  const char* e = getenv("LTR_STATS_COMB_MIN");        // (read per call: a test switches it inside one process)
  return e ? atoll(e) : 8192;
}

// passes of at most this many rows get the small-batch split-K scratch (launch_gemm decides per launch)
constexpr int64_t SPLITK_MAX_ROWS = 4800;

// workspace carve-up for one chunk of at most Tc tokens / Nc requests
struct Workspace {
  float* h;        // f32 [Tc, H]      residual stream
  AOp a;           // operand [Tc, H]  LN out / attention out / (De!=H: token rows)
  AOp qkv;         // [Tc, 3H]: f32 (F32 mode) or fp16 hi|lo planes written by the QKV GEMM epilogue
  AOp f;           // operand [Tc, F]  ReLU(fc1)
  int32_t* blk;    // attention work list: int32 [Nc + 4] prefix + int4 [Tc / 32 + Nc + 1] block descriptors
  size_t blk_bytes;
  // LayerNorm fold: second operand buffer (out_proj reads `a` while it writes the fc1 operand) and the row-piece
  // statistics written by fc2 (for the next layer's LN1) / out_proj (for LN2): float2 [H / 64][Tc] each
  AOp a2;
  void* stats1;
  void* stats2;
  void* comb1;     // float2 [Tc]: (mean, rstd) of the rows behind stats1 / stats2, combined once per producer launch (large passes)
  void* comb2;
  // GEMM head: compact last-token rows f32 [Nc, H] (only when the forward did not compact them), their operand
  // [Nc, H], the project_out result (operand or f32) [Nc, De], the padded logits f32 [Nc, lpad]
  float* head_rows;
  AOp head_op;
  char* head_feat;
  float* head_logits;
  // small-batch split-K scratch (GemmArgs::splitk_ws): raw f32 partials of out_proj / fc2 when a pass has few rows
  void* splitk;
  size_t splitk_bytes;
  size_t bytes;
};

Workspace carve(const ltr_model_desc& d, int64_t Tc, int64_t Nc, void* base, bool ln_fold = false, int head_lpad = -1) {
  const size_t H = d.hidden_size, F = d.ffn_dim;
  const size_t esz = 4;   // operand bytes per element: f32, or fp16 hi + fp16 lo
  char* p = (char*)base;
  size_t off = 0;
  Workspace ws{};
  auto take = [&](size_t n) { size_t o = off; off += align_up(n); return base ? (void*)(p + o) : (void*)nullptr; };
  ws.h = (float*)take(Tc * H * 4);
  char* a = (char*)take(Tc * H * esz);
  char* qkv = (char*)take(Tc * 3 * H * esz);
  char* f = (char*)take(Tc * F * esz);
  ws.blk_bytes = (Nc + 4) * 4 + (Tc / 32 + Nc + 1) * 16;       // (32-query blocks: the split-K/V attention of small passes)
  ws.blk = (int32_t*)take(ws.blk_bytes);
  if (ln_fold) {
    char* a2 = (char*)take(Tc * H * esz);
    ws.a2 = AOp{a2, a2 ? a2 + Tc * H * 2 : nullptr};
    ws.stats1 = take((H / 64) * Tc * 8);
    ws.stats2 = take((H / 64) * Tc * 8);
    ws.comb1 = take(Tc * 8);
    ws.comb2 = take(Tc * 8);
  }
  if (head_lpad >= 0) {        // the GEMM head is in use (head_lpad = 0: project_out only)
    const size_t De = d.word_embed_proj_dim;
    ws.head_rows = (float*)take(Nc * H * 4);
    char* ho = (char*)take(Nc * H * esz);
    ws.head_op = AOp{ho, ho ? ho + Nc * H * 2 : nullptr};
    ws.head_feat = (char*)take(Nc * De * esz);
    const size_t lrows = head_lpad > 0 ? (size_t)(Nc < head_block_rows(head_lpad) ? Nc : head_block_rows(head_lpad)) : 1;
    ws.head_logits = (float*)take(lrows * (head_lpad > 0 ? head_lpad : 1) * 4);
  }
  if (d.weight_dtype == LTR_W_F16) {
    // up to 4 parts of [rows, H] f32, rows <= SPLITK_MAX_ROWS (beyond it the GEMMs have tiles enough without)
    const size_t rows = (size_t)(Tc < SPLITK_MAX_ROWS ? Tc : SPLITK_MAX_ROWS);
    ws.splitk_bytes = 4 * rows * H * 4;
    ws.splitk = take(ws.splitk_bytes);
  }
  if (d.weight_dtype == LTR_W_F16) {
    ws.a = AOp{a, a ? a + Tc * H * 2 : nullptr};
    ws.f = AOp{f, f ? f + Tc * F * 2 : nullptr};
    ws.qkv = AOp{qkv, qkv ? qkv + Tc * 3 * H * 2 : nullptr};
  } else {
    ws.a = AOp{a, nullptr};
    ws.f = AOp{f, nullptr};
    ws.qkv = AOp{qkv, nullptr};
  }
  ws.bytes = off;
  return ws;
}

// Brackets one launch with events when profiling is on (no host sync).
struct ProfScope {
  ltr_model* m;
  hipStream_t s;
  int idx = -1;
  ProfScope(const ltr_model* cm, int kind, double work, hipStream_t st) : m(const_cast<ltr_model*>(cm)), s(st) {
    if (!m->prof_on) return;
    std::lock_guard<std::mutex> lk(m->prof_mu);
    ProfRec r{};
    if (!m->prof_free.empty()) {
      r.start = m->prof_free.back().first; r.stop = m->prof_free.back().second; m->prof_free.pop_back();
    } else if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) {
      return;
    }
    r.kind = kind; r.work = work;
    (void)hipEventRecord(r.start, s);
    m->prof.push_back(r);
    idx = (int)m->prof.size() - 1;
  }
  ~ProfScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(m->prof_mu);
    (void)hipEventRecord(m->prof[idx].stop, s);
  }
};

// One request-aligned chunk: requests [r0, r1), tokens [t0, t1) of the global batch - as a stepper (begin, then one call per
// decoder layer).  (Rounds 4-5 issued the layers of two half batches alternately onto two streams, "lanes": measured again and
// removed in round 6, profiles/r06_lanes_after.txt.)
struct ChunkRun {
  const ltr_model* m;
  const int64_t* ids;
  const int32_t* cu_dev;
  int N_total, r0, r1, t0, t1, n_layers;
  Workspace ws;
  double sum_l2;
  bool prune_last;
  hipStream_t s;
  // derived / state
  const ltr_model_desc& d;
  const int wd, H, F, De, Tc, nreq;
  int nl = 0;
  bool fold = false, ln1_folded = false;
  bool plain_all = false;      // GemmArgs::store_plain of this pass's launches (small passes: begin())
  bool comb1_valid = false, comb2_valid = false;   // ws.comb1 / comb2 hold the combined statistics of the current stats1 / stats2
  // rows / buffers of the part of a layer after attention: all Tc token rows, except in the
  // last layer of a scoring call where only the nreq last-token rows are carried on (below)
  int Mr;
  float* hb;
  AOp ab, fb;
  ChunkRun(const ltr_model* m_, const int64_t* ids_, const int32_t* cu_dev_, int N_total_, int r0_, int r1_, int t0_, int t1_,
           int n_layers_, const Workspace& ws_, double sum_l2_, bool prune_last_, hipStream_t s_)
      : m(m_), ids(ids_), cu_dev(cu_dev_), N_total(N_total_), r0(r0_), r1(r1_), t0(t0_), t1(t1_), n_layers(n_layers_), ws(ws_),
        sum_l2(sum_l2_), prune_last(prune_last_), s(s_), d(m_->d), wd(m_->d.weight_dtype), H(m_->d.hidden_size),
        F(m_->d.ffn_dim), De(m_->d.word_embed_proj_dim), Tc(t1_ - t0_), nreq(r1_ - r0_), Mr(t1_ - t0_), hb(ws_.h), ab(ws_.a),
        fb(ws_.f) {}
  int gemm(const GemmArgs& g0) {
    GemmArgs g = g0;
    if (m->one_pass) {     // LTR_F_ONE_PASS: hi plane only; lo planes that only another GEMM would read are not stored
      g.one_pass = 1;
      g.no_lo_out = g.keep_lo_out ? 0 : 1;     // every reader of a split output runs one pass too (GEMMs, the MFMA attention) - except
                                               // the last layer's K | V, which the VALU last-query kernel reads as hi + lo
    }
    // the dominant kernel (128 x 256 tiles) and the small-batch kernels are timed as separate classes
    if (plain_all) g.store_plain = 1;
    ProfScope p(m, wd == LTR_W_F16 && gemm_small_config(g) >= 0 ? LTR_K_GEMM_SMALL : LTR_K_GEMM, 2.0 * g.M * (double)g.N * g.K, s);
    return launch_gemm(wd, g, s);
  }
  int lnorm(int rows, const float* x, const float* gw_, const float* gb_, float* of, AOp oo) {
    ProfScope p(m, LTR_K_LN, (double)rows * H * (of ? 12.0 : 8.0), s);   // f32 row in, operand row (+f32 row) out
    return launch_layernorm(wd, x, gw_, gb_, rows, H, of, oo, s);
  }
  const float* h_final() const { return hb; }   // ws.h, or the compact last-token rows when the last layer was pruned
  int begin();
  int layer(int L);
};

int ChunkRun::begin() {
  int rc;

  // --- embedding (opt.py:241-245)
  const double wbytes = wd == LTR_W_F16 ? 2.0 : 4.0;
  {
    ProfScope p(m, LTR_K_EMBED, (double)Tc * (8.0 + (De + H) * wbytes + H * 4.0), s);
    rc = launch_embed_gather(wd, ids, cu_dev, N_total, Tc, t0, m->gw(LTR_WT_EMBED_TOKENS), De, d.vocab_size,
                             m->gw(LTR_WT_EMBED_POS), H, d.pos_rows, ws.h, ws.a, m->err_flag, s);
  }
  if (rc) return rc;
  if (De != H) {   // h = project_in(tok) + pos : GEMM with the position rows as residual (in place)
    GemmArgs g{};
    g.a = ws.a; g.w = m->gemm_gw(LTR_WT_PROJECT_IN); g.bias = nullptr; g.resid = ws.h; g.out_f32 = ws.h;
    g.M = Tc; g.N = H; g.K = De;
    if ((rc = gemm(g))) return rc;
  }
  if (!d.pre_ln) {   // post-LN blocks consume h itself as the first GEMM operand
    if ((rc = launch_to_operand(wd, ws.h, Tc, H, ws.a, s))) return rc;
  }
  nl = n_layers < 0 ? d.num_layers : (n_layers < d.num_layers ? n_layers : d.num_layers);
  // LayerNorm fold (ltr_gemm.hip): out_proj / fc2 emit the operand and the row statistics of the LayerNorm that
  // follows them, QKV / fc1 finish it in their epilogue.  `ln1_folded`: this layer's QKV operand (ws.a) and
  // ws.stats1 were written by the previous layer's fc2.
  fold = m->ln_fold;
  ln1_folded = false;
  // Cache policy of the GEMM epilogues' stores.  Non-temporal stores keep a full-size pass's 1.4 GB of outputs per launch from
  // sweeping the operand panels out of the L2s (-1.4 % on the 8k-queue call) - and send them to HBM, where the next launch reads
  // them at the HBM rate.  The outputs of a SMALL pass fit the 256 MiB Infinity Cache: stored with the default policy their
  // reader - the next launch - finds them there.  Measured on one box (profiles/r06_store_policy.txt): one or two arrivals
  // -5 %, eight -2 %, sixteen -1.5 ... -3 %; from ~4k tokens per pass on it is a wash (the attention reads q | k | v 10 % faster,
  // the GEMM that wrote them plainly loses as much to the lines its stores allocate in L2).  Rule: every output of the pass while
  // its widest activation (rows x F, f32-sized) is <= LTR_PLAIN_MB (40: 3,413 tokens of OPT-125m, 2,560 of OPT-350m).
  // Bit-identical either way.
  {
    const char* e = getenv("LTR_PLAIN_MB");           // (read per call: the lab moves it inside one process)
    plain_all = wd == LTR_W_F16 && (double)Tc * F * 4.0 <= (e ? atof(e) : 40.0) * 1048576.0;
  }
  return LTR_OK;
}

int ChunkRun::layer(const int L) {
  int rc;
    // --- attention half (opt.py:152-163)
    if (d.pre_ln && !ln1_folded) {
      rc = lnorm(Tc, ws.h, (const float*)m->lw(L, LTR_WL_LN1_W), (const float*)m->lw(L, LTR_WL_LN1_B), nullptr, ws.a);
      if (rc) return rc;
    }
    const bool last_pruned = prune_last && L == nl - 1;
    // F16 mode, last layer of a scoring call: Q and the attention output are needed for the last token of each
    // request only; K and V for every token.
    const bool lastq = last_pruned && m->lastq && L == d.num_layers - 1;
    if (lastq) {
      const float* qkv_b = (const float*)m->lw(L, LTR_WL_QKV_B);
      // K | V of every token: hi|lo planes [Tc, 2H] at the head of the qkv region
      AOp kv{ws.qkv.hi, (char*)ws.qkv.hi + (size_t)Tc * 2 * H * 2};
      {
        GemmArgs g{};
        g.a = ws.a; g.w = m->last_kv_w; g.bias = qkv_b + H; g.out_split = kv; g.keep_lo_out = 1;
        g.M = Tc; g.N = 2 * H; g.K = H; g.a_slab = 1;
        if (ln1_folded) { g.ln_stats_in = ws.stats1; g.ln_c = m->fold_c_qkv[L] + H; g.bias = m->fold_d_qkv[L] + H; g.ln_parts = H / 64; }
        if (ln1_folded && comb1_valid) g.ln_stats_comb = ws.comb1;
        if ((rc = gemm(g))) return rc;
      }
      // The nreq last-token rows of the residual stream go to the tail of the qkv region (Tc * H * 4 bytes are
      // free behind the K | V planes); their LayerNorm (pre-LN; computed directly, the folded operand of the
      // whole pass is not compacted) or their split copy (post-LN) is the operand of the Q GEMM.  ws.a is dead
      // once the K | V GEMM has read it: it holds the Q operand, then the attention output; ws.h is dead once the
      // rows are gathered: it holds q.
      hb = (float*)((char*)ws.qkv.hi + (size_t)Tc * 2 * H * 4);
      ab = AOp{ws.a.hi, (char*)ws.a.hi + (size_t)nreq * H * 2};
      fb = AOp{ws.f.hi, (char*)ws.f.hi + (size_t)nreq * F * 2};
      Mr = nreq;
      if ((rc = launch_gather_last_rows(wd, cu_dev + r0, t0, nreq, H, ws.h, AOp{nullptr, nullptr}, hb, AOp{nullptr, nullptr}, s)))
        return rc;
      if (d.pre_ln) rc = lnorm(nreq, hb, (const float*)m->lw(L, LTR_WL_LN1_W), (const float*)m->lw(L, LTR_WL_LN1_B), nullptr, ab);
      else if (ln1_folded)   // post-LN fold: the gathered rows are x of the previous fc2, their LayerNorm is still pending
        rc = lnorm(nreq, hb, (const float*)m->lw(L - 1, LTR_WL_LN2_W), (const float*)m->lw(L - 1, LTR_WL_LN2_B), hb, ab);
      else rc = launch_to_operand(wd, hb, nreq, H, ab, s);
      if (rc) return rc;
      float* qf = ws.h;
      {
        GemmArgs g{};
        g.a = ab; g.w = m->last_q_w; g.bias = qkv_b; g.out_f32 = qf; g.M = nreq; g.N = H; g.K = H; g.a_slab = 1;
        if ((rc = gemm(g))) return rc;
      }
      {
        ProfScope p(m, LTR_K_ATTN, 4.0 * Tc * H, s);
        rc = launch_attention_lastq(qf, kv, cu_dev + r0, nreq, H, d.num_heads, ab, s);
      }
      if (rc) return rc;
    } else {
      {
        GemmArgs g{};
        g.a = ws.a; g.w = m->gemm_lw(L, LTR_WL_QKV_W); g.bias = (const float*)m->lw(L, LTR_WL_QKV_B);
        if (wd == LTR_W_F16 && !m->dbg_attn_valu) g.out_split = ws.qkv; else g.out_f32 = (float*)ws.qkv.hi;
        g.M = Tc; g.N = 3 * H; g.K = H; g.a_slab = wd == LTR_W_F16;   // A from LayerNorm / to_operand / the fold
        if (ln1_folded) { g.ln_stats_in = ws.stats1; g.ln_c = m->fold_c_qkv[L]; g.bias = m->fold_d_qkv[L]; g.ln_parts = H / 64; }
        if (ln1_folded && comb1_valid) g.ln_stats_comb = ws.comb1;
        if ((rc = gemm(g))) return rc;
      }
      {
        ProfScope p(m, LTR_K_ATTN, 2.0 * sum_l2 * H, s);
        rc = launch_attention(m->dbg_attn_valu && wd == LTR_W_F16 ? -1 : wd, ws.qkv, cu_dev + r0, nreq, Tc, H, d.num_heads,
                              ws.blk, ws.a, L == 0, s, nullptr, ws.blk_bytes, m->one_pass && !m->dbg_attn_valu);
      }
      if (rc) return rc;
    }
    if (last_pruned && !lastq) {
      // Only the last token of each prompt is scored (logits_processor.py:74-79), and after the
      // attention of the LAST layer no token reads another token's state: out_proj, the LayerNorms
      // and the MLP are per-token maps.  Compact the nreq last-token rows of h and of the attention
      // output and run the rest of the layer on those rows only (bit-identical per row; saves
      // ~3/4 of the last layer's GEMM work).  Buffers: the qkv region (dead after attention) holds
      // the compact h and operand, the f region the compact MLP intermediate.
      char* q = (char*)ws.qkv.hi;
      hb = (float*)q;
      char* a2 = q + (((size_t)nreq * H * 4 + 255) & ~(size_t)255);
      ab = wd == LTR_W_F16 ? AOp{a2, a2 + (size_t)nreq * H * 2} : AOp{a2, nullptr};
      fb = wd == LTR_W_F16 ? AOp{ws.f.hi, (char*)ws.f.hi + (size_t)nreq * F * 2} : ws.f;
      Mr = nreq;
      if ((rc = launch_gather_last_rows(wd, cu_dev + r0, t0, nreq, H, ws.h, ws.a, hb, ab, s))) return rc;
      if (!d.pre_ln && ln1_folded) {   // post-LN fold: apply the pending LayerNorm of the previous fc2 to the compact rows
        rc = lnorm(nreq, hb, (const float*)m->lw(L - 1, LTR_WL_LN2_W), (const float*)m->lw(L - 1, LTR_WL_LN2_B), hb,
                   AOp{nullptr, nullptr});
        if (rc) return rc;
      }
    }
    // pre-LN: also on the n_req compact rows of the pruned last layer (M = n_req).  Post-LN: the compact rows of the
    // pruned last layer run unfolded (their residuals were normalised explicitly above).
    const bool fold_here = fold && (d.pre_ln || !last_pruned);
    // post-LN fold: ws.h holds x = the PRE-LayerNorm stream of the previous layer's fc2 (LN2 of layer L-1 pending)
    const bool resid_pending = !d.pre_ln && ln1_folded && !last_pruned;
    {
      GemmArgs g{};
      g.a = ab; g.w = m->gemm_lw(L, LTR_WL_OUT_W); g.bias = (const float*)m->lw(L, LTR_WL_OUT_B);
      g.resid = hb; g.out_f32 = hb; g.M = Mr; g.N = H; g.K = H;
      g.splitk_ws = ws.splitk; g.splitk_ws_bytes = ws.splitk_bytes;      // (launch_gemm uses it for small passes and for tail rows)
      // the LayerNorm that follows this residual add: pre-LN blocks LN2 (final_layer_norm), post-LN blocks LN1
      // (self_attn_layer_norm, opt.py:162-163)
      if (fold_here) {
        g.ln_gamma = (const float*)m->lw(L, d.pre_ln ? LTR_WL_LN2_W : LTR_WL_LN1_W);
        g.ln_out = ws.a2; g.ln_stats_out = ws.stats2; g.err_flag = m->err_flag;
      }
      if (resid_pending) {   // residual = LN2 of layer L-1 applied to x, rebuilt in the epilogue
        g.rln_stats = ws.stats1; g.rln_gamma = (const float*)m->lw(L - 1, LTR_WL_LN2_W);
        g.rln_beta = (const float*)m->lw(L - 1, LTR_WL_LN2_B); g.rln_parts = H / 64;
        if (comb1_valid) g.rln_stats_comb = ws.comb1;
      }
      if ((rc = gemm(g))) return rc;
      comb2_valid = false;
      if (fold_here && Mr >= stats_comb_min_rows()) {   // the pieces out_proj just wrote -> (mean, rstd) per row, once
        if ((rc = launch_row_stats_combine(ws.stats2, H / 64, Mr, Mr, ws.comb2, s))) return rc;
        comb2_valid = true;
      }
    }
    if (!d.pre_ln && !fold_here) {   // 350m: LN after the residual add; h and its operand copy
      rc = lnorm(Mr, hb, (const float*)m->lw(L, LTR_WL_LN1_W), (const float*)m->lw(L, LTR_WL_LN1_B), hb, ab);
      if (rc) return rc;
    }
    // --- feed-forward half (opt.py:165-175)
    if (d.pre_ln && !fold_here) {
      rc = lnorm(Mr, hb, (const float*)m->lw(L, LTR_WL_LN2_W), (const float*)m->lw(L, LTR_WL_LN2_B), nullptr, ab);
      if (rc) return rc;
    }
    {
      GemmArgs g{};
      g.a = fold_here ? ws.a2 : ab; g.w = m->gemm_lw(L, LTR_WL_FC1_W); g.bias = (const float*)m->lw(L, LTR_WL_FC1_B);
      g.out_split = fb; g.relu = 1; g.M = Mr; g.N = F; g.K = H; g.a_slab = g.out_slab = wd == LTR_W_F16;
      if (fold_here) { g.ln_stats_in = ws.stats2; g.ln_c = m->fold_c_fc1[L]; g.bias = m->fold_d_fc1[L]; g.ln_parts = H / 64; }
      if (fold_here && comb2_valid) g.ln_stats_comb = ws.comb2;
      if ((rc = gemm(g))) return rc;
    }
    // the LayerNorm that follows this residual add rides on fc2: pre-LN blocks the NEXT layer's LN1 (also into the pruned
    // last layer, whose K | V GEMM consumes it); post-LN blocks this layer's LN2, unless this is the last layer run (its
    // output must be the true hidden state)
    ln1_folded = fold_here && (d.pre_ln ? L + 1 < d.num_layers : L + 1 < nl);
    {
      GemmArgs g{};
      g.a = fb; g.w = m->gemm_lw(L, LTR_WL_FC2_W); g.bias = (const float*)m->lw(L, LTR_WL_FC2_B);
      g.resid = hb; g.out_f32 = hb; g.M = Mr; g.N = H; g.K = F; g.a_slab = wd == LTR_W_F16;
      g.splitk_ws = ws.splitk; g.splitk_ws_bytes = ws.splitk_bytes;
      if (ln1_folded) {
        g.ln_gamma = d.pre_ln ? (const float*)m->lw(L + 1, LTR_WL_LN1_W) : (const float*)m->lw(L, LTR_WL_LN2_W);
        g.ln_out = ws.a; g.ln_stats_out = ws.stats1; g.err_flag = m->err_flag;
      }
      if (!d.pre_ln && fold_here) {   // residual = LN1 of this layer applied to out_proj's x
        g.rln_stats = ws.stats2; g.rln_gamma = (const float*)m->lw(L, LTR_WL_LN1_W);
        g.rln_beta = (const float*)m->lw(L, LTR_WL_LN1_B); g.rln_parts = H / 64;
        if (comb2_valid) g.rln_stats_comb = ws.comb2;
      }
      if ((rc = gemm(g))) return rc;
      comb1_valid = false;
      if (ln1_folded && Mr >= stats_comb_min_rows()) {   // the pieces fc2 just wrote (for the next layer's QKV / out_proj), once
        if ((rc = launch_row_stats_combine(ws.stats1, H / 64, Mr, Mr, ws.comb1, s))) return rc;
        comb1_valid = true;
      }
    }
    if (!d.pre_ln && !ln1_folded) {
      rc = lnorm(Mr, hb, (const float*)m->lw(L, LTR_WL_LN2_W), (const float*)m->lw(L, LTR_WL_LN2_B), hb, ab);
      if (rc) return rc;
    }
  return LTR_OK;
}

int check_desc(const ltr_model_desc& d) {
  if (d.hidden_size <= 0 || d.num_heads <= 0 || d.hidden_size != d.num_heads * 64) {
    set_error("ltr_create: head size must be 64 (H=%d, heads=%d)", d.hidden_size, d.num_heads);
    return LTR_E_INVAL;
  }
  // F is the N of fc1 (the GEMM tiles N in 64-column wave strips) and the K of fc2 (32-wide slabs)
  if (d.hidden_size % 32 || d.ffn_dim % 64 || d.word_embed_proj_dim % 32 || d.hidden_size > 2048 ||
      d.word_embed_proj_dim > 2048) {
    set_error("ltr_create: H, De must be multiples of 32 (<= 2048) and F a multiple of 64");
    return LTR_E_INVAL;
  }
  if (d.weight_dtype != LTR_W_F32 && d.weight_dtype != LTR_W_F16) {
    set_error("ltr_create: unknown weight dtype %d", d.weight_dtype);
    return LTR_E_INVAL;
  }
  if (d.num_labels < 1 || d.num_layers < 0 || d.vocab_size < 1 || d.pos_rows < 3) {
    set_error("ltr_create: bad num_labels/num_layers/vocab/pos_rows");
    return LTR_E_INVAL;
  }
  if (d.flags & ~(LTR_F_NO_LN_FOLD | LTR_F_NO_LANES | LTR_F_ONE_PASS | LTR_F_LANES_UNPROBED)) {
    set_error("ltr_create: unknown flags 0x%x (a caller built against ABI 4 leaves this word uninitialised)", d.flags);
    return LTR_E_INVAL;
  }
  if ((d.flags & LTR_F_ONE_PASS) && d.weight_dtype != LTR_W_F16) {
    set_error("ltr_create: LTR_F_ONE_PASS exists in F16 mode only");
    return LTR_E_INVAL;
  }
  return LTR_OK;
}

// GEMM head (F16 mode): -1 = not used (the VALU pool_head_kernel does everything), 0 = project_out on the matrix cores
// only, > 0 = also the class logits, padded to that many labels
int head_mode(const ltr_model* m) { return m->proj_out_w || m->head_w ? m->head_lpad : -1; }

// tokens per chunk: never below the longest legal request (pos_rows - 2 positions)
int64_t chunk_cap(const ltr_model* m) {
  const int64_t maxpos = m->d.pos_rows - 2;
  return m->chunk_tokens > maxpos ? m->chunk_tokens : maxpos;
}

// host copy of cu_seqlens (caller's mirror, or a synchronising copy back)
int host_cu(const int32_t* cu_dev, const int32_t* cu_host, int N, std::vector<int32_t>& tmp, const int32_t** out,
            hipStream_t s) {
  if (cu_host) { *out = cu_host; return LTR_OK; }
  tmp.resize(N + 1);
  LTR_HIP_CHECK(hipMemcpyAsync(tmp.data(), cu_dev, (N + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  LTR_HIP_CHECK(hipStreamSynchronize(s));
  *out = tmp.data();
  return LTR_OK;
}

}  // namespace

extern "C" {

int ltr_abi_version(void) { return LTR_ABI_VERSION; }
const char* ltr_last_error(void) { return ltr::g_err; }

int ltr_create(const ltr_model_desc* desc, const void* const* weights, int32_t n_weights, void* stream,
               ltr_handle* out) {
  if (!desc || !weights || !out) { set_error("ltr_create: NULL argument"); return LTR_E_INVAL; }
  hipStream_t cs = (hipStream_t)stream;
  int rc = check_desc(*desc);
  if (rc) return rc;
  const int want = LTR_WT_GLOBAL_COUNT + desc->num_layers * LTR_WL_COUNT;
  if (n_weights != want) { set_error("ltr_create: %d weight pointers, expected %d", n_weights, want); return LTR_E_INVAL; }
  const bool proj = desc->word_embed_proj_dim != desc->hidden_size;
  for (int i = 0; i < want; ++i) {
    bool optional = (i == LTR_WT_PROJECT_IN || i == LTR_WT_PROJECT_OUT) ? !proj
                    : (i == LTR_WT_FINAL_LN_W || i == LTR_WT_FINAL_LN_B) ? true : false;
    if (!weights[i] && !optional) { set_error("ltr_create: weight pointer %d is NULL", i); return LTR_E_INVAL; }
  }
  if ((weights[LTR_WT_FINAL_LN_W] == nullptr) != (weights[LTR_WT_FINAL_LN_B] == nullptr)) {
    set_error("ltr_create: final LN weight/bias must both be set or both NULL");
    return LTR_E_INVAL;
  }
  ltr_model* m = new (std::nothrow) ltr_model();
  if (!m) { set_error("ltr_create: out of host memory"); return LTR_E_NOMEM; }
  m->d = *desc;
  m->w.assign(weights, weights + want);
  m->chunk_tokens = DEFAULT_CHUNK_TOKENS;
  {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, weights[LTR_WT_EMBED_TOKENS]) != hipSuccess) {
      (void)hipGetLastError();
      delete m;
      set_error("ltr_create: weight pointers must be device memory");
      return LTR_E_INVAL;
    }
    m->device = attr.device;
  }
  DeviceGuard guard(m->device);
  if (hipMalloc((void**)&m->err_flag, sizeof(int32_t)) != hipSuccess ||
      hipMemsetAsync(m->err_flag, 0, sizeof(int32_t), cs) != hipSuccess) {
    delete m;
    set_error("ltr_create: cannot allocate the status word");
    return LTR_E_NOMEM;
  }
  { const char* e = getenv("LTR_DEBUG_ATTN_VALU"); m->dbg_attn_valu = e && e[0] == '1'; }
  m->one_pass = (desc->flags & LTR_F_ONE_PASS) != 0;
  m->wg = m->w;
  if (desc->weight_dtype == LTR_W_F16) {
    // one-time re-layout of the dense-layer weights into the GEMM kernel's slab-major image
    const size_t H = desc->hidden_size, F = desc->ffn_dim, De = desc->word_embed_proj_dim;
    struct Item { int idx; size_t n, k; };
    std::vector<Item> items;
    if (proj) items.push_back({LTR_WT_PROJECT_IN, H, De});
    for (int L = 0; L < desc->num_layers; ++L) {
      const int b = LTR_WT_GLOBAL_COUNT + L * LTR_WL_COUNT;
      items.push_back({b + LTR_WL_QKV_W, 3 * H, H});
      items.push_back({b + LTR_WL_OUT_W, H, H});
      items.push_back({b + LTR_WL_FC1_W, F, H});
      items.push_back({b + LTR_WL_FC2_W, H, F});
    }
    { const char* e = getenv("LTR_NO_LASTQ"); m->lastq = desc->num_layers > 0 && !m->dbg_attn_valu && !(e && e[0] == '1'); }
    const size_t last_bytes = m->lastq ? 3 * H * H * 2 : 0;   // H * H and 2H * H halves: both multiples of 256 B
    // head on the matrix cores: project_out when there is one, the class logits from LTR_HEAD_GEMM_MIN_LABELS (17) labels
    int min_labels = 17;
    { const char* e = getenv("LTR_HEAD_GEMM_MIN_LABELS"); if (e) min_labels = atoi(e); }
    const bool pack_proj = proj && De % 64 == 0;
    const bool pack_head = desc->num_labels >= min_labels && min_labels > 0 && De % 64 == 0;
    const size_t lpad = pack_head ? ((size_t)desc->num_labels + 63) / 64 * 64 : 0;
    const size_t head_bytes = (pack_proj ? De * H * 2 : 0) + lpad * De * 2;        // both multiples of 256 B
    size_t total = last_bytes + head_bytes;
    for (auto& it : items) total += (it.n * it.k * 2 + 255) / 256 * 256;
    if (total) {
      if (hipMalloc(&m->packed, total) != hipSuccess) {
        delete m;
        set_error("ltr_create: cannot allocate %zu bytes for the packed weights", total);
        return LTR_E_NOMEM;
      }
      size_t off = 0;
      for (auto& it : items) {
        void* dst = (char*)m->packed + off;
        if ((rc = launch_pack_weight(m->w[it.idx], dst, (int)it.n, (int)it.k, cs))) { delete m; return rc; }
        m->wg[it.idx] = dst;
        off += (it.n * it.k * 2 + 255) / 256 * 256;
      }
      if (m->lastq) {   // nn.Linear layout [3H, H] row-major: q | k | v row blocks (opt.py:411-417)
        const char* src = (const char*)m->lw(desc->num_layers - 1, LTR_WL_QKV_W);
        char* dq = (char*)m->packed + off;
        char* dkv = dq + H * H * 2;
        if ((rc = launch_pack_weight(src, dq, (int)H, (int)H, cs)) ||
            (rc = launch_pack_weight(src + H * H * 2, dkv, (int)(2 * H), (int)H, cs))) { delete m; return rc; }
        m->last_q_w = dq; m->last_kv_w = dkv;
        off += last_bytes;
      }
      if (pack_proj) {
        void* dst = (char*)m->packed + off;
        if ((rc = launch_pack_weight(m->w[LTR_WT_PROJECT_OUT], dst, (int)De, (int)H, cs))) { delete m; return rc; }
        m->proj_out_w = dst;
        off += De * H * 2;
      }
      if (pack_head) {
        void* dst = (char*)m->packed + off;
        if ((rc = launch_pack_weight(m->w[LTR_WT_SCORE], dst, (int)lpad, (int)De, cs, desc->num_labels))) { delete m; return rc; }
        m->head_w = dst;
        m->head_lpad = (int)lpad;
        off += lpad * De * 2;
      }
    }
  }
  {
    // LayerNorm fold: pre-LN blocks in F16 mode with H a multiple of 64 (LTR_NO_LN_FOLD=1 keeps the LayerNorm
    // launches: A/B switch for measurements and debugging)
    const char* e = getenv("LTR_NO_LN_FOLD");               // (diag; callers use LTR_F_NO_LN_FOLD)
    const bool no_fold = (e && e[0] == '1') || (desc->flags & LTR_F_NO_LN_FOLD);
    const size_t H = desc->hidden_size, F = desc->ffn_dim;
    // pre-LN (125m): LN1 rides on QKV, LN2 on fc1.  Post-LN (350m): this layer's LN1 (after the attention residual)
    // rides on fc1, the PREVIOUS layer's LN2 on QKV; the residuals are rebuilt by the RLN epilogue (ChunkRun::layer).
    m->ln_fold = desc->weight_dtype == LTR_W_F16 && H % 64 == 0 && desc->num_layers > 0 && !no_fold;
    if (m->ln_fold) {
      const size_t per_layer = 2 * (3 * H + F);
      if (hipMalloc((void**)&m->fold, per_layer * desc->num_layers * sizeof(float)) != hipSuccess) {
        delete m; set_error("ltr_create: cannot allocate the LayerNorm fold vectors"); return LTR_E_NOMEM;
      }
      for (int L = 0; L < desc->num_layers; ++L) {
        float* p = m->fold + (size_t)L * per_layer;
        m->fold_c_qkv.push_back(p); m->fold_d_qkv.push_back(p + 3 * H);
        m->fold_c_fc1.push_back(p + 6 * H); m->fold_d_fc1.push_back(p + 6 * H + F);
        const bool pre = desc->pre_ln != 0;
        const int lq = pre ? L : (L > 0 ? L - 1 : 0);      // layer whose LayerNorm feeds this QKV (post-LN layer 0: unused)
        if ((rc = launch_ln_fold_coeff(m->lw(L, LTR_WL_QKV_W), (const float*)m->lw(lq, pre ? LTR_WL_LN1_W : LTR_WL_LN2_W),
                                       (const float*)m->lw(lq, pre ? LTR_WL_LN1_B : LTR_WL_LN2_B),
                                       (const float*)m->lw(L, LTR_WL_QKV_B), (int)(3 * H), (int)H, p, p + 3 * H, cs)) ||
            (rc = launch_ln_fold_coeff(m->lw(L, LTR_WL_FC1_W), (const float*)m->lw(L, pre ? LTR_WL_LN2_W : LTR_WL_LN1_W),
                                       (const float*)m->lw(L, pre ? LTR_WL_LN2_B : LTR_WL_LN1_B),
                                       (const float*)m->lw(L, LTR_WL_FC1_B), (int)F, (int)H, p + 6 * H, p + 6 * H + F, cs))) {
          delete m; return rc;
        }
      }
    }
  }
  if (hipStreamSynchronize(cs) != hipSuccess) { delete m; set_error("ltr_create: weight packing failed"); return LTR_E_HIP; }
  *out = m;
  return LTR_OK;
}

int ltr_destroy(ltr_handle h) {
  if (h) { DeviceGuard guard(h->device); delete h; }
  return LTR_OK;
}

int ltr_status(ltr_handle h, void* stream) {
  if (!h) { set_error("ltr_status: NULL handle"); return LTR_E_INVAL; }
  DeviceGuard guard(h->device);
  hipStream_t s = (hipStream_t)stream;
  int32_t flag = 0;
  LTR_HIP_CHECK(hipMemcpyAsync(&flag, h->err_flag, sizeof(flag), hipMemcpyDeviceToHost, s));
  LTR_HIP_CHECK(hipStreamSynchronize(s));
  if (flag) {
    LTR_HIP_CHECK(hipMemsetAsync(h->err_flag, 0, sizeof(int32_t), s));
    if (flag & 1)
      set_error("ltr_score: a token id outside [0, %d) was fed to the embedding (F.embedding raises on it, "
                "vocab_parallel_embedding.py:95-106); the scores of that call are invalid", h->d.vocab_size);
    else
      set_error("ltr_score: the residual stream left the fp16 range of the LayerNorm-fold operand (|x * gamma| > 4094); "
                "the scores of that call are invalid - score the batch on a handle created with LTR_F_NO_LN_FOLD");
    return (flag & 1) ? LTR_E_INVAL : LTR_E_RANGE;
  }
  return LTR_OK;
}

int ltr_set_chunk_tokens(ltr_handle h, int32_t chunk_tokens) {
  if (!h || chunk_tokens < 0) { set_error("ltr_set_chunk_tokens: bad argument"); return LTR_E_INVAL; }
  h->chunk_tokens = chunk_tokens == 0 ? DEFAULT_CHUNK_TOKENS : chunk_tokens;
  return LTR_OK;
}

size_t ltr_workspace_bytes(ltr_handle h, int32_t kind, int64_t N, int64_t T) {
  if (kind == LTR_WS_RANK) return rank_workspace_bytes(N);
  if (kind == LTR_WS_SCORE && h) {
    int64_t Tc = T < chunk_cap(h) ? T : chunk_cap(h);
    int64_t Nc = N < Tc ? N : Tc;
    const size_t one = carve(h->d, Tc > 0 ? Tc : 1, Nc > 0 ? Nc : 1, nullptr, h->ln_fold, head_mode(h)).bytes;
    return one;
  }
  return 0;
}

static int run_forward(ltr_handle h, const int64_t* token_ids, const int32_t* cu_seqlens, const int32_t* cu_host_in,
                       int32_t N, int32_t T, int32_t max_len, int32_t n_layers, float* hidden_out, float* scores_out,
                       float* logits_out, void* workspace, size_t ws_bytes, hipStream_t s) {
  if (!h) { set_error("ltr_score: NULL handle"); return LTR_E_INVAL; }
  if (N < 0 || T < 0) { set_error("ltr_score: negative size"); return LTR_E_INVAL; }
  if (N == 0) return LTR_OK;
  DeviceGuard guard(h->device);
  if (!token_ids || !cu_seqlens || !workspace) { set_error("ltr_score: NULL pointer"); return LTR_E_INVAL; }
  std::vector<int32_t> tmp;
  const int32_t* cu = nullptr;
  int rc = host_cu(cu_seqlens, cu_host_in, N, tmp, &cu, s);
  if (rc) return rc;
  if (cu[0] != 0 || cu[N] != T) { set_error("ltr_score: cu_seqlens[0]=%d cu_seqlens[N]=%d, T=%d", cu[0], cu[N], T); return LTR_E_INVAL; }
  const ltr_model_desc& d = h->d;
  const int max_pos = d.pos_rows - 2;
  if (max_len > 0) {
    for (int r = 0; r < N; ++r)
      if (cu[r + 1] - cu[r] > max_len) {
        set_error("ltr_score: request %d has %d tokens > max_len %d", r, cu[r + 1] - cu[r], max_len);
        return LTR_E_INVAL;
      }
  }
  // chunk budget from the workspace actually provided
  int64_t Tc_cap = chunk_cap(h) < T ? chunk_cap(h) : T;
  while (Tc_cap > 1 && carve(d, Tc_cap, Tc_cap < N ? Tc_cap : N, nullptr, h->ln_fold, head_mode(h)).bytes > ws_bytes) Tc_cap /= 2;
  // what follows the decoder layers of a chunk: the hidden states out, or final LayerNorm / project_out / head -> scores
  auto finish_chunk = [&](ChunkRun& c) -> int {
    const Workspace& ws = c.ws;
    hipStream_t s = c.s;
    const int r0 = c.r0, r1 = c.r1, t0 = c.t0, t1 = c.t1;
    const bool prune = c.prune_last;
    const float* h_final = c.h_final();
    int rc = LTR_OK;
    if (hidden_out) {
      LTR_HIP_CHECK(hipMemcpyAsync(hidden_out, ws.h, (size_t)(t1 - t0) * d.hidden_size * 4, hipMemcpyDeviceToDevice, s));
    } else {
      const int n = r1 - r0, H = d.hidden_size, De = d.word_embed_proj_dim, wd = d.weight_dtype;
      const int n_cmp = d.num_labels < d.vocab_size ? d.num_labels : d.vocab_size;   // logits_processor.py:68-70
      const bool proj = De != H;
      float* scores_dst = scores_out + r0;
      float* logits_dst = logits_out ? logits_out + (size_t)r0 * d.num_labels : nullptr;
      if (head_mode(h) < 0) {
        ProfScope p(h, LTR_K_POOL, (double)n * (H * 4.0 + 4.0), s);
        rc = launch_pool_head(wd, h_final, prune ? nullptr : cu_seqlens + r0, t0, n, H, De, d.num_labels, n_cmp,
                              (const float*)h->gw(LTR_WT_FINAL_LN_W), (const float*)h->gw(LTR_WT_FINAL_LN_B),
                              proj ? h->gw(LTR_WT_PROJECT_OUT) : nullptr, h->gw(LTR_WT_SCORE), scores_dst, logits_dst, s);
        if (rc) return rc;
      } else {
        // ---- GEMM head (F16 mode).  Final LayerNorm / project_out / score.weight are per-row maps of the n last-token
        // rows (logits_processor.py:74-79, opt.py:259-262,374): the LayerNorm writes the rows as a split operand,
        // project_out [n, H] x [De, H]^T and - in class mode - the logits [n, De] x [labels, De]^T run on the
        // split-fp16 MFMA kernels (8,192 labels x 8,192 requests is a 103 GFLOP GEMM, not 2,048 serial wave reductions
        // per request), then one argmax pass (first maximum, opt.py:394-395).  Rank mode keeps the VALU dot product.
        ProfScope p(h, LTR_K_POOL, (double)n * (H * 4.0 + 4.0), s);
        const float* rows = h_final;
        if (!prune) {          // the forward left all T rows: compact the last-token rows first
          if ((rc = launch_gather_last_rows(wd, cu_seqlens + r0, t0, n, H, h_final, AOp{nullptr, nullptr}, ws.head_rows,
                                            AOp{nullptr, nullptr}, s))) return rc;
          rows = ws.head_rows;
        }
        const float* ln_w = (const float*)h->gw(LTR_WT_FINAL_LN_W);
        if (ln_w) rc = launch_layernorm(wd, rows, ln_w, (const float*)h->gw(LTR_WT_FINAL_LN_B), n, H, nullptr, ws.head_op, s);
        else rc = launch_to_operand(wd, rows, n, H, ws.head_op, s);
        if (rc) return rc;
        AOp feat = ws.head_op;                       // [n, De] operand of the label GEMM
        const bool labels_gemm = h->head_w != nullptr;
        if (proj) {
          GemmArgs g{};
          g.a = ws.head_op; g.w = h->proj_out_w; g.M = n; g.N = De; g.K = H; g.a_slab = 1; g.one_pass = h->one_pass;
          if (labels_gemm) { feat = AOp{ws.head_feat, ws.head_feat + (size_t)n * De * 2}; g.out_split = feat; g.out_slab = 1; }
          else g.out_f32 = (float*)ws.head_feat;
          if ((rc = launch_gemm(wd, g, s))) return rc;
        }
        if (labels_gemm) {
          // the padded logits exist for HEAD_BLOCK_ROWS rows at a time (row windows of the label GEMM, argmax per block):
          // 8,192 labels x 8,192 requests would otherwise add 268 MB to every scoring workspace
          const int rb = head_block_rows(h->head_lpad);
          for (int rw = 0; rw < n; rw += rb) {
            const int nb = n - rw < rb ? n - rw : rb;
            GemmArgs g{};
            g.a = feat; g.w = h->head_w; g.N = h->head_lpad; g.K = De; g.a_slab = 1; g.one_pass = h->one_pass;
            g.row0 = rw; g.M = nb; g.ldm = n;
            g.out_f32 = ws.head_logits - (size_t)rw * h->head_lpad;          // (rows are indexed globally: block-local buffer)
            if ((rc = launch_gemm(wd, g, s))) return rc;
            if ((rc = launch_argmax_rows(ws.head_logits, h->head_lpad, nb, n_cmp, d.num_labels, scores_dst + rw,
                                         logits_dst ? logits_dst + (size_t)rw * d.num_labels : nullptr, s))) return rc;
          }
        } else {
          // few labels: dot products of the De-wide feature rows on the VALU (no LayerNorm, no projection left to do)
          rc = launch_pool_head(wd, (const float*)ws.head_feat, nullptr, 0, n, De, De, d.num_labels, n_cmp, nullptr, nullptr,
                                nullptr, h->gw(LTR_WT_SCORE), scores_dst, logits_dst, s);
          if (rc) return rc;
        }
      }
    }
    return LTR_OK;
  };
  int r0 = 0;
  while (r0 < N) {
    int r1 = r0;
    while (r1 < N && (int64_t)cu[r1 + 1] - cu[r0] <= Tc_cap) {
      const int L = cu[r1 + 1] - cu[r1];
      if (L <= 0) { set_error("ltr_score: request %d has length %d (empty prompts are not schedulable)", r1, L); return LTR_E_INVAL; }
      if (L > max_pos) { set_error("ltr_score: request %d has %d tokens > max positions %d (truncate first, aux_llm_engine.py:365-369)", r1, L, max_pos); return LTR_E_INVAL; }
      ++r1;
    }
    if (r1 == r0) {
      const int L = cu[r0 + 1] - cu[r0];
      if (L <= 0) { set_error("ltr_score: request %d has length %d", r0, L); return LTR_E_INVAL; }
      set_error("ltr_score: workspace of %zu bytes cannot hold request %d (%d tokens)", ws_bytes, r0, L);
      return LTR_E_NOMEM;
    }
    const int t0 = cu[r0], t1 = cu[r1];
    const bool prune = !hidden_out && d.num_layers > 0;
    auto sum_sq = [&](int a_, int b_) { double v = 0.0; for (int r = a_; r < b_; ++r) { const double L = cu[r + 1] - cu[r]; v += L * L; } return v; };
    Workspace ws = carve(d, t1 - t0, r1 - r0, workspace, h->ln_fold, head_mode(h));
    if (ws.bytes > ws_bytes) { set_error("ltr_score: workspace too small (%zu < %zu)", ws_bytes, ws.bytes); return LTR_E_NOMEM; }
    if (hidden_out && (r0 != 0 || r1 != N)) { set_error("ltr_forward_hidden: batch does not fit one chunk"); return LTR_E_NOMEM; }
    ChunkRun c(h, token_ids, cu_seqlens, N, r0, r1, t0, t1, n_layers, ws, sum_sq(r0, r1), prune, s);
    rc = c.begin();
    for (int L = 0; !rc && L < c.nl; ++L) rc = c.layer(L);
    if (rc) return rc;
    if ((rc = finish_chunk(c))) return rc;
    r0 = r1;
  }
  return LTR_OK;
}

int ltr_score(ltr_handle h, const int64_t* token_ids, const int32_t* cu_seqlens, const int32_t* cu_seqlens_host,
              int32_t N, int32_t T, int32_t max_len, float* scores_out, float* logits_out, void* workspace,
              size_t ws_bytes, void* stream) {
  if (N > 0 && !scores_out) { set_error("ltr_score: scores_out is NULL"); return LTR_E_INVAL; }
  return run_forward(h, token_ids, cu_seqlens, cu_seqlens_host, N, T, max_len, -1, nullptr, scores_out, logits_out,
                     workspace, ws_bytes, (hipStream_t)stream);
}

int ltr_forward_hidden(ltr_handle h, const int64_t* token_ids, const int32_t* cu_seqlens,
                       const int32_t* cu_seqlens_host, int32_t N, int32_t T, int32_t max_len, int32_t n_layers,
                       float* hidden_out, void* workspace, size_t ws_bytes, void* stream) {
  if (N > 0 && !hidden_out) { set_error("ltr_forward_hidden: hidden_out is NULL"); return LTR_E_INVAL; }
  return run_forward(h, token_ids, cu_seqlens, cu_seqlens_host, N, T, max_len, n_layers, hidden_out, nullptr, nullptr,
                     workspace, ws_bytes, (hipStream_t)stream);
}

int ltr_embed_gather(ltr_handle h, const int64_t* token_ids, const int32_t* cu_seqlens, int32_t N, int32_t T,
                     float* hidden_out, void* tok_out, void* stream) {
  if (!h || !token_ids || !cu_seqlens || !hidden_out) { set_error("ltr_embed_gather: NULL argument"); return LTR_E_INVAL; }
  const ltr_model_desc& d = h->d;
  const bool proj = d.word_embed_proj_dim != d.hidden_size;
  if (proj && !tok_out) { set_error("ltr_embed_gather: tok_out required when De != H"); return LTR_E_INVAL; }
  AOp tok{tok_out, nullptr};
  if (proj && d.weight_dtype == LTR_W_F16) tok.lo = (char*)tok_out + (size_t)T * d.word_embed_proj_dim * 2;
  DeviceGuard guard(h->device);
  return launch_embed_gather(d.weight_dtype, token_ids, cu_seqlens, N, T, 0, h->gw(LTR_WT_EMBED_TOKENS),
                             d.word_embed_proj_dim, d.vocab_size, h->gw(LTR_WT_EMBED_POS), d.hidden_size, d.pos_rows,
                             hidden_out, tok, h->err_flag, (hipStream_t)stream);
}

int ltr_pool_head(ltr_handle h, const float* hidden, const int32_t* cu_seqlens, int32_t N, float* scores_out,
                  float* logits_out, void* stream) {
  if (!h || !hidden || !cu_seqlens || !scores_out) { set_error("ltr_pool_head: NULL argument"); return LTR_E_INVAL; }
  const ltr_model_desc& d = h->d;
  DeviceGuard guard(h->device);
  return launch_pool_head(d.weight_dtype, hidden, cu_seqlens, 0, N, d.hidden_size, d.word_embed_proj_dim, d.num_labels,
                          d.num_labels < d.vocab_size ? d.num_labels : d.vocab_size,
                          (const float*)h->gw(LTR_WT_FINAL_LN_W), (const float*)h->gw(LTR_WT_FINAL_LN_B),
                          d.word_embed_proj_dim != d.hidden_size ? h->gw(LTR_WT_PROJECT_OUT) : nullptr,
                          h->gw(LTR_WT_SCORE), scores_out, logits_out, (hipStream_t)stream);
}

int ltr_attention(ltr_handle h, const void* qkv, const int32_t* cu_seqlens, int32_t N, int32_t T, void* out, void* workspace,
                  size_t ws_bytes, void* stream) {
  if (!h || N < 0 || T < 0) { set_error("ltr_attention: bad argument"); return LTR_E_INVAL; }
  if (N == 0 || T == 0) return LTR_OK;
  if (!qkv || !cu_seqlens || !out || !workspace) { set_error("ltr_attention: NULL pointer"); return LTR_E_INVAL; }
  const size_t need = (size_t)(N + 4) * 4 + ((size_t)T / 64 + N + 1) * 16;
  if (ws_bytes < need) { set_error("ltr_attention: workspace too small (%zu < %zu)", ws_bytes, need); return LTR_E_NOMEM; }
  const ltr_model_desc& d = h->d;
  const size_t H = d.hidden_size;
  DeviceGuard guard(h->device);
  const bool f16 = d.weight_dtype == LTR_W_F16;
  AOp in{(void*)qkv, f16 ? (void*)((char*)qkv + (size_t)T * 3 * H * 2) : nullptr};
  AOp o{out, f16 ? (void*)((char*)out + (size_t)T * H * 2) : nullptr};
  return launch_attention(d.weight_dtype, in, cu_seqlens, N, T, (int)H, d.num_heads, (int32_t*)workspace, o, 1,
                          (hipStream_t)stream, nullptr, ws_bytes);
}

int ltr_rank_step(const float* scores, int32_t* pri, int32_t* idle, int32_t* runs, const uint32_t* tiebreak,
                  const int32_t* members, int32_t N, int32_t starv, int32_t period, uint32_t flags, int32_t* perm_out,
                  void* workspace, size_t ws_bytes, void* stream) {
  if (N < 0) { set_error("ltr_rank_step: negative N"); return LTR_E_INVAL; }
  if (N == 0) return LTR_OK;
  if (!scores || !perm_out) { set_error("ltr_rank_step: NULL argument"); return LTR_E_INVAL; }
  return launch_rank_step(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, perm_out, workspace,
                          ws_bytes, (hipStream_t)stream);
}

int ltr_age_update(const uint8_t* ran, const int32_t* ran_slots, int32_t n_ran, int32_t* pri, int32_t* idle,
                   int32_t* runs, const int32_t* members, int32_t N, void* stream) {
  if (N < 0 || n_ran < 0) { set_error("ltr_age_update: negative size"); return LTR_E_INVAL; }
  if (N == 0) return LTR_OK;
  if ((!ran && n_ran > 0 && !ran_slots) || !pri || !idle || !runs) { set_error("ltr_age_update: NULL argument"); return LTR_E_INVAL; }
  return launch_age_update(ran, ran_slots, n_ran, pri, idle, runs, members, N, (hipStream_t)stream);
}

int ltr_queue_step(const float* scores, int32_t* pri, int32_t* idle, int32_t* runs, const uint32_t* tiebreak,
                   const int32_t* members, int32_t N, int32_t starv, int32_t period, uint32_t flags,
                   const int32_t* new_tokens, const int32_t* new_seqs, const uint8_t* chunkable, int64_t token_budget,
                   int64_t max_num_seqs, int32_t* perm_out, int32_t* n_selected_out, uint8_t* ran_out,
                   int32_t* granted_out, void* workspace, size_t ws_bytes, void* stream) {
  if (N < 0 || !n_selected_out || (N > 0 && (!scores || !perm_out || !new_tokens || !new_seqs))) {
    set_error("ltr_queue_step: bad argument");
    return LTR_E_INVAL;
  }
  return launch_queue_step(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, new_tokens, new_seqs,
                           chunkable, token_budget, max_num_seqs, perm_out, n_selected_out, ran_out, granted_out,
                           workspace, ws_bytes, (hipStream_t)stream);
}

int ltr_profile_enable(ltr_handle h, int32_t on) {
  if (!h) { set_error("ltr_profile_enable: NULL handle"); return LTR_E_INVAL; }
  h->prof_on = on != 0;
  return LTR_OK;
}

int ltr_profile_read(ltr_handle h, ltr_profile_stats* out, int32_t reset) {
  if (!h || !out) { set_error("ltr_profile_read: NULL argument"); return LTR_E_INVAL; }
  DeviceGuard guard(h->device);
  std::lock_guard<std::mutex> lk(h->prof_mu);
  memset(out, 0, sizeof(*out));
  for (auto& r : h->prof) {
    LTR_HIP_CHECK(hipEventSynchronize(r.stop));
    float ms = 0.f;
    LTR_HIP_CHECK(hipEventElapsedTime(&ms, r.start, r.stop));
    out->ms[r.kind] += ms;
    out->work[r.kind] += r.work;
    out->launches[r.kind] += 1;
  }
  if (reset) {
    for (auto& r : h->prof) h->prof_free.emplace_back(r.start, r.stop);
    h->prof.clear();
  }
  return LTR_OK;
}

int ltr_budget_prefix(const int32_t* perm, const int32_t* new_tokens, const int32_t* new_seqs,
                      const uint8_t* chunkable, int32_t N, int64_t token_budget, int64_t max_num_seqs,
                      int32_t* n_selected_out, uint8_t* ran_out, int32_t* granted_out, void* stream) {
  if (N < 0 || !n_selected_out || (N > 0 && (!perm || !new_tokens || !new_seqs))) {
    set_error("ltr_budget_prefix: bad argument");
    return LTR_E_INVAL;
  }
  return launch_budget_prefix(perm, new_tokens, new_seqs, chunkable, N, token_budget, max_num_seqs, n_selected_out,
                              ran_out, granted_out, (hipStream_t)stream);
}

int ltr_reserve_select(const int32_t* perm, const int32_t* n_selected, const uint8_t* state, const int32_t* phys,
                       const int32_t* logical, const int32_t* nrun, const int32_t* nswap, const int32_t* new_seqs,
                       int32_t N, int64_t need_in, uint8_t* action_out, int32_t* n_exec_out,
                       int32_t* blocks_required_out, void* stream) {
  if (N < 0 || !n_selected || !n_exec_out ||
      (N > 0 && (!perm || !state || !phys || !logical || !nrun || !nswap || !action_out))) {
    set_error("ltr_reserve_select: bad argument");
    return LTR_E_INVAL;
  }
  return launch_reserve_select(perm, n_selected, state, phys, logical, nrun, nswap, new_seqs, N, need_in, action_out,
                               n_exec_out, blocks_required_out, (hipStream_t)stream);
}

}  // extern "C"
