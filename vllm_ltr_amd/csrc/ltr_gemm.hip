// Dense layers of the predictor (QKV / out_proj / fc1 / fc2 / project_in) on gfx950 MFMA.
//
// Reference ops: QKVParallelLinear / RowParallelLinear / ColumnParallelLinear = F.linear
// (vllm/model_executor/layers/linear.py), used by OPTAttention (opt.py:92-102) and
// OPTDecoderLayer (opt.py:145-176).  C[M,N] = A[M,K] * W[N,K]^T (+bias)(ReLU)(+residual).
//
// Two arithmetic modes, selected by the checkpoint dtype:
//  * F16 ("split") - the production path.  The checkpoint is fp16 (train/trainer.py:215),
//    so W is exact in fp16.  Activations are carried as two fp16 planes a = hi + lo;
//    acc += lo*W; acc += hi*W on v_mfma_f32_16x16x32_f16 with f32 accumulation.  That keeps
//    ~22 significant bits of the f32 activation (score error vs the f32 CPU oracle ~4e-6,
//    where plain fp16 activations give 2e-3), at 2 MFMA passes instead of the 16x slower
//    f32 MFMA.
//  * F32 - exact f32 MFMA (v_mfma_f32_32x32x2_f32) for f32 checkpoints / cross-checks.
//
// Tiling: each wave owns a 64x64 sub-tile = 4x4 MFMA 16x16 accumulators (64 acc VGPRs;
// v_mfma_f32_16x16x32_f16, the more power-efficient form - see the main loop); the
// F16 kernel uses a 128 (M) x 256 (N) tile with 8 waves (activations cost 4 B/elem as hi+lo,
// weights 2, so the tile is wider in N), the F32 kernel 128x128 with 4 waves.  LDS layouts are chosen per
// instruction so the fragment reads are conflict-free (see lds_off_* below).  blockIdx is
// remapped so that consecutive tiles of one XCD share the same weight panel (8 XCDs,
// private L2s).  The F16 kernel streams operands HBM -> LDS with global_load_lds; the F32
// kernel stages through registers.
#include <cstdlib>

#include "ltr_internal.h"

namespace ltr {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;
constexpr int NXCD = 8;

// Workgroup -> tile map.  Hardware places block b on XCD b % 8 (each XCD has a private
// 4 MiB L2) and runs ~96 blocks per XCD at a time (32 CUs x 3).  Each XCD gets a contiguous
// run of the tile list, and the list is in GROUPED order: groups of GM row-tiles, M fastest
// inside a group, so the ~96 co-resident tiles of an XCD form a GM x (96/GM) patch that
// shares GM activation panels (4 B/elem: hi+lo) and 96/GM weight panels through that L2,
// instead of 96 distinct activation panels (measured 13 B/clk/CU of operand fetch was the
// limiter with the plain M-fastest order).  Bijective for any grid size.
constexpr int GM_DEFAULT = 8;
__device__ __forceinline__ void tile_of_block(int bid, int tiles_m, int tiles_n, int& tm, int& tn, int GM = GM_DEFAULT) {
  const int nwg = tiles_m * tiles_n;
  const int q = nwg / NXCD, r = nwg % NXCD;
  const int xcd = bid % NXCD, k = bid / NXCD;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  const int per_group = (GM & 0xffff) * tiles_n;
  const int g = lin / per_group, in_g = lin - g * per_group;
  const int gm = min(GM & 0xffff, tiles_m - g * (GM & 0xffff));          // last group may be short
  if (GM >> 16) {        // N fastest inside the group (narrow outputs, see launch_gemm)
    tm = g * (GM & 0xffff) + in_g / tiles_n;
    tn = in_g % tiles_n;
  } else {
    tm = g * GM + in_g % gm;
    tn = in_g / gm;
  }
}

struct Epilogue {
  const float* bias;
  const float* resid;
  float* out_f32;
  void* out_hi;
  void* out_lo;
  int M, N, relu;
  int a_slab, out_slab;   // F16 kernel: slab-major A image / slab-major split output (GemmArgs)
  // LayerNorm folded into the GEMMs (F16 kernel, GemmArgs::ln_*; see "LayerNorm fold" below)
  const float* ln_gamma;  // producer: gamma [N] of the LayerNorm that follows this output
  void* ln_hi;            // producer: slab-major planes of x * gamma * LN_FOLD_SCALE
  void* ln_lo;
  float2* stats_out;      // producer: [N / 64][M] (mean, M2) of the 64-column pieces of every output row
  const float2* stats_in; // consumer: [n_part][M] pieces of the rows of A
  const float* ln_c;      // consumer: [N] LN_FOLD_SCALE * sum_k gamma_k W[n, k]   (bias then holds beta W + b)
  int n_part;
  int32_t* err_flag;      // producer: bit 1 is set when |x gamma scale| leaves the fp16 range (ltr_status reports it)
  // residual = LayerNorm(resid) rebuilt on the fly (post-LN blocks, "RLN" below)
  const float2* r_stats;  // [r_parts][M] (mean, M2) pieces of the rows of `resid`
  const float* r_gamma;   // [N]
  const float* r_beta;    // [N]
  int r_parts;
  const float* osc_a;     // LNS: max|x| of the operands' source tensors (the product is divided by their split scales)
  const float* osc_b;
  // Row window (GemmArgs::row0 / ldm): the launch covers rows [row0, M) of tensors that have ldm rows - M is the END of
  // the window, every row index is global; ldm is the pitch of the slab-major images and of the statistics arrays.
  int row0, ldm;
  // LTR_F_ONE_PASS: one_pass - the small-batch kernels skip the lo MFMA pass (the large-tile kernel has a template
  // instance without the lo stream); no_lo_out - the lo planes of out_hi / ln_hi's twins are not stored
  int one_pass, no_lo_out;
  // (mean, rstd) per row, combined once per launch by row_stats_combine_kernel (GemmArgs::ln_stats_comb / rln_stats_comb);
  // null: the tile combines the pieces itself
  const float2* stats_comb;
  const float2* r_stats_comb;
  int st_plain;           // GemmArgs::store_plain: default-policy stores instead of non-temporal ones (small passes)
};

// LNS rides on the consumer arithmetic v = (acc - mean c_n) rstd with mean = 0, c_n = 0, rstd = the output scale
__device__ __forceinline__ float2 out_scale_stat(const Epilogue& ep) {
  return make_float2(0.f, 1.f / (split_scale(*ep.osc_a) * split_scale(*ep.osc_b)));
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
template <bool SPLIT>
__device__ __forceinline__ void store_tile(const Epilogue& e, const f32x16& acc, int row0, int col) {
  if (col >= e.N) return;
  const float b = e.bias ? e.bias[col] : 0.f;
  const int lane_hi = (threadIdx.x & 63) >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lane_hi;
    if (row < e.M) {
      float v = acc[r] + b;
      if (e.relu) v = fmaxf(v, 0.f);
      const size_t o = (size_t)row * e.N + col;
      if (e.resid) v += e.resid[o];
      if (e.out_f32) e.out_f32[o] = v;
      if (e.out_hi) {
        if (SPLIT) {
          __half h, l;
          split_f16(v, h, l);
          ((__half*)e.out_hi)[o] = h;
          ((__half*)e.out_lo)[o] = l;
        } else {
          ((float*)e.out_hi)[o] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// F16 split mode.  K-slab = 32 halves.  Operands go HBM -> LDS directly with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write): one wave instruction lands 1 KiB
// = 16 tile rows x 64 B, LDS destination linear (wave base + lane * 16).  The weights are static,
// so ltr_create re-lays them out once into the SLAB-MAJOR image [K/32][N][32] (pack_weight_kernel):
// a 16-row group of a K-slab is then 1 KiB of contiguous memory (full 128-B lines) instead of
// sixteen 64-B half lines (measured -3.5 % GEMM time; L2 -> LDS streams 32 TB/s contiguous vs
// 18-20 TB/s in 64-B row pieces, diag/dma_rate.hip).  The bank
// swizzle is applied on the SOURCE side: the lane that fills physical 16-B chunk c of row r
// fetches logical chunk c ^ ((r >> 2) & 3), and fragment reads apply the same XOR.  With it
// a ds_read_b128 lane group (rows {0-3,12-15,20-27} / ...) touches 16 distinct 16-B slots of
// the 256-B bank row -> conflict-free.  Two LDS stages (2 x 32 KiB): slab k+1 streams in
// while slab k is multiplied; one barrier per slab.  All LDS lives in ONE array (a second
// __shared__ object makes hipcc drain vmcnt before every ds_read of the pipeline).
// Epilogue: accumulators -> LDS (per-wave 16x64 f32 strips) -> 16 B per lane coalesced
// global stores with bias / ReLU / residual / fp16 hi|lo split fused.
// ------------------------------------------------------------------------------------
constexpr int BK16 = 32;
constexpr int BN16 = 256;                         // F16 kernel: 128 x 256 tile, 8 waves as 2 (M) x 4 (N)
constexpr int A_PLANE = BM * BK16;                // halves per activation plane per stage (8 KiB)
constexpr int W_PLANE = BN16 * BK16;              // 16 KiB
constexpr int STAGE = 2 * A_PLANE + W_PLANE;      // a_hi | a_lo | w = 32 KiB
constexpr int CLD = 68;                           // f32 row stride of the epilogue strip
constexpr int CROWS = 16;                         // rows per epilogue strip
// chunk swizzle of tile row `row` (a function of (row >> 2) & 3), chosen for the 16x16x32 fragments (lane: row
// l & 15, chunk l >> 4) so that the 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) hit 16 distinct 16-B
// slots of a 256-B bank row
__device__ __forceinline__ int swz(int row) {
  return (4 - ((row >> 2) & 3)) & 3;
}
__device__ __forceinline__ int lds_off_h(int row, int kc) {   // in halves
  return row * BK16 + ((kc ^ swz(row)) << 3);
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// 16-byte epilogue store.  The outputs of a launch (1.4 GB) are read back by the NEXT launch, long after they have left
// the 4 MiB L2 of the XCD, while the operand panels the co-resident tiles share live there: the stores are issued
// non-temporal so that they claim as little of it as possible.  LTR_EPI_STORE selects the cache policy (A/B knob,
// profiles/r02_ab_gemm_probes.txt): 0 plain 235.0 / 236.9 ms per call, 1 nt 231.8 (kept), 2 sc1 233.8, 3 sc0 sc1 235.1
#ifndef LTR_EPI_STORE
#define LTR_EPI_STORE 1
#endif
// LTR_EPI_NOPS=1 (A/B knob) puts 16 wait states in front of the asm stores, as rounds 2-5 shipped them: an experimental epilogue
// of round 2 "stored stale registers in one lane of 16" and the s_nops cured it.  That epilogue is long gone and the cause was
// never found; round 6 built the library without them: gemm_check clean at every shape / tile configuration / K split, the
// parity suite and the busy-GPU soak green (profiles/r06_rln_fault.txt, end) - they guard nothing in this tree and are off.
#ifndef LTR_EPI_NOPS
#define LTR_EPI_NOPS 0
#endif
#if LTR_EPI_NOPS
#define LTR_EPI_PRE "s_nop 7\n\ts_nop 7\n\t"
#else
#define LTR_EPI_PRE ""
#endif
#ifndef LTR_EPI_POST          // (-DLTR_EPI_POST="" rebuilds the stores of rounds 2-6 for the lint's own test)
#define LTR_EPI_POST "\n\ts_nop 1"
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void epi_store16(void* p, const void* v, int plain) {
  const u32x4 d = *reinterpret_cast<const u32x4*>(v);
  if (plain) { *reinterpret_cast<u32x4*>(p) = d; return; }      // (wave-uniform: a kernel argument)
#if LTR_EPI_STORE == 1
  // Inline asm: the compiler's own __builtin_nontemporal_store costs out_proj 12 % (610 vs 545 us per 196k-token launch,
  // diag/gemm_bench.hip).  The operands are ordinary "v" inputs, so the compiler's waitcnt / hazard passes see what feeds them.
  // LTR_EPI_POST: a store of more than 64 bits reads its data registers over several cycles, and gfx940-class hardware needs two
  // wait states before a VALU instruction may overwrite them (LLVM's hazard recognizer inserts them behind the stores IT emits;
  // it does not look inside an asm statement).  Without them the next VALU write can land in the data of the store before it:
  // round 2's "stale registers in one lane of 16", and - round 6 - wrong a' planes as soon as the code around the store changed
  // (profiles/r06_store_policy.txt).  The wait states belong to the asm statement; isa_lint.py checks the built code for the hazard.
  asm volatile(LTR_EPI_PRE "global_store_dwordx4 %0, %1, off nt" LTR_EPI_POST ::"v"(p), "v"(d) : "memory");
#elif LTR_EPI_STORE == 2
  asm volatile(LTR_EPI_PRE "global_store_dwordx4 %0, %1, off sc1" LTR_EPI_POST ::"v"(p), "v"(d) : "memory");
#elif LTR_EPI_STORE == 3
  asm volatile(LTR_EPI_PRE "global_store_dwordx4 %0, %1, off sc0 sc1" LTR_EPI_POST ::"v"(p), "v"(d) : "memory");
#elif LTR_EPI_STORE == 5
  __builtin_nontemporal_store(d, reinterpret_cast<u32x4*>(p));
#else
  *reinterpret_cast<u32x4*>(p) = d;
#endif
}

#ifdef LTR_GEMM_TIMELINE
__device__ unsigned long long g_timeline[8192 * 4];
__device__ unsigned long long g_waits[8192 * 2];
#endif

// LayerNorm fold (pre-LN blocks).  LN(x) W^T + b = rstd (x.gamma) W^T - rstd mean (gamma W^T) + (beta W^T + b): the GEMM that
// PRODUCES the residual stream x (out_proj, fc2: LNP) also writes a' = split(x * gamma * 16) as the next GEMM's
// slab-major operand and, per 64-column piece of every row, (mean, M2); the CONSUMER (QKV, fc1: LNC) combines the
// pieces of its 128 rows into (mean, rstd / 16) while its first operand slab is in flight and finishes
// v = (acc - mean c_n) rstd / 16 + d_n in the epilogue.  No LayerNorm launch, no second read of x.  The weights stay
// the exact fp16 checkpoint values (gamma rides on the ACTIVATION side, where the hi|lo split absorbs it); the
// power-of-two scale keeps lo = a' - hi out of the fp16 subnormals for residual streams of magnitude ~0.01.
constexpr float LN_FOLD_SCALE = 16.f;
enum { LN_NONE = 0, LNP = 1, LNC = 2, LNS = 3 };   // LNS: plain epilogue on acc * out_scale (GemmArgs::osc_a / osc_b)

// (mean, M2) of the 64-column pieces of one row -> (mean, rstd * out_scale).  Shifted-data sums around the first piece's
// mean (equal counts per piece): the loads of all pieces are independent of each other and fly together (a serial Chan
// update would put n_part dependent L2 round trips in front of the first barrier of the tile).
__device__ __forceinline__ float2 combine_row_stats(const float2* __restrict__ sp /*&stats[0][row]*/, int n_part, int M,
                                                    float out_scale) {
  const float2 s0 = sp[0];
  float s1 = 0.f, s2 = 0.f, sm = s0.y;
  auto add = [&](const float2 s) { const float d = s.x - s0.x; s1 += d; s2 = fmaf(d, d, s2); sm += s.y; };
  if (n_part == 12) {
    float2 v[11];
#pragma unroll
    for (int p = 0; p < 11; ++p) v[p] = sp[(size_t)(p + 1) * M];
#pragma unroll
    for (int p = 0; p < 11; ++p) add(v[p]);
  } else if (n_part == 16) {
    float2 v[15];
#pragma unroll
    for (int p = 0; p < 15; ++p) v[p] = sp[(size_t)(p + 1) * M];
#pragma unroll
    for (int p = 0; p < 15; ++p) add(v[p]);
  } else {
#pragma unroll 4
    for (int p = 1; p < n_part; ++p) add(sp[(size_t)p * M]);
  }
  const float np_ = (float)n_part;
  const float mean = s0.x + s1 / np_;
  const float m2 = sm + 64.f * fmaxf(s2 - s1 * s1 / np_, 0.f);
  return make_float2(mean, rsqrtf(m2 / (64.f * np_) + LN_EPS) * out_scale);
}

// (mean, rstd [/ scale]) of one row for a tile's prologue: from the per-launch combined array when there is one.
// The array's (mean, rstd) arrives as ONE 64-bit register; its components go through own_reg before anybody multiplies with
// them: as the high half of that register, rstd is read through op_sel by the packed-f32 instructions hipcc makes of the
// epilogue arithmetic, and the form it picked for splitk_epilogue_kernel<*, true> (v_pk_mul_f32 ... op_sel:[0,1]) is the one
// that computes wrong values in lanes 48-63 beside a library fp16 GEMM - round 5's "concurrency-dependent wrong scores"
// (profiles/r06_rln_fault.txt; isa_lint.py keeps the form out of the library).  LTR_RLN_FAULT_SHAPE rebuilds round 5's
// expression for the reproducer (diag/rln_fault.hip); never in the library.
template <bool RLN>
__device__ __forceinline__ float2 tile_row_stat(const Epilogue& ep, int row) {
#ifdef LTR_RLN_FAULT_SHAPE
  if (RLN) return ep.r_stats_comb ? ep.r_stats_comb[row] : combine_row_stats(ep.r_stats + row, ep.r_parts, ep.ldm, 1.f);
#else
  if (RLN) {
    if (!ep.r_stats_comb) return combine_row_stats(ep.r_stats + row, ep.r_parts, ep.ldm, 1.f);
    const float2 st = ep.r_stats_comb[row];
    return make_float2(own_reg(st.x), own_reg(st.y));
  }
#endif
  if (ep.stats_comb) { const float2 st = ep.stats_comb[row]; return make_float2(own_reg(st.x), own_reg(st.y * (1.f / LN_FOLD_SCALE))); }
  return combine_row_stats(ep.stats_in + row, ep.n_part, ep.ldm, 1.f / LN_FOLD_SCALE);
}

__global__ void __launch_bounds__(256) row_stats_combine_kernel(const float2* __restrict__ parts, int n_part, int ldm, int rows,
                                                                float2* __restrict__ out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < rows) out[r] = combine_row_stats(parts + r, n_part, ldm, 1.f);
}

// One (row, 8 columns) piece of the epilogue, shared by the large-tile and the small-tile kernel: x = accumulator
// columns ccol..ccol+3 (va) and ccol_b..ccol_b+3 (vb) of row `grow`; LayerNorm-fold consumer scaling, bias, ReLU,
// residual, f32 / split stores, LayerNorm-fold producer outputs.  The 8 lanes that hold one row's 64-column piece are
// consecutive lanes of one wave (lane & 7 = position in the piece); `piece64` = index of that piece in the row.
// RLN (post-LN blocks, LayerNorm fold): `resid` holds the PRE-LayerNorm residual stream x written by the previous
// producer; the residual to add is y = LN(x) = (x - mean) rstd gamma + beta with (mean, rstd) = rst of the row.
template <int LNM, bool RLN>
__device__ __forceinline__ void epilogue_piece(const Epilogue& ep, float4 va, float4 vb, float4 ra, float4 rb,
                                               const float4& bias_a, const float4& bias_b, const float4& lnv_a,
                                               const float4& lnv_b, const float2 st2, const float2 rst, int grow, int ccol,
                                               int ccol_b, size_t o, int ldm, int lane, int piece64) {
  if (RLN) {   // (the two halves one after the other: the affine vectors of both at once cost 8 more live registers)
    {
      const float4 g = *reinterpret_cast<const float4*>(ep.r_gamma + ccol), b = *reinterpret_cast<const float4*>(ep.r_beta + ccol);
      ra.x = (ra.x - rst.x) * rst.y * g.x + b.x; ra.y = (ra.y - rst.x) * rst.y * g.y + b.y;
      ra.z = (ra.z - rst.x) * rst.y * g.z + b.z; ra.w = (ra.w - rst.x) * rst.y * g.w + b.w;
    }
    asm volatile("" : "+v"(ra.x), "+v"(ra.y), "+v"(ra.z), "+v"(ra.w));      // keep the halves apart in the schedule
    {
      const float4 g = *reinterpret_cast<const float4*>(ep.r_gamma + ccol_b), b = *reinterpret_cast<const float4*>(ep.r_beta + ccol_b);
      rb.x = (rb.x - rst.x) * rst.y * g.x + b.x; rb.y = (rb.y - rst.x) * rst.y * g.y + b.y;
      rb.z = (rb.z - rst.x) * rst.y * g.z + b.z; rb.w = (rb.w - rst.x) * rst.y * g.w + b.w;
    }
  }
  if (LNM == LNC || LNM == LNS) {   // v = (acc - mean c_n) rstd / scale (+ d_n, held in bias)
    static_assert(!((LNM == LNC || LNM == LNS) && RLN), "consumer epilogue and LayerNorm'd residual never meet");
    va.x = (va.x - st2.x * lnv_a.x) * st2.y; va.y = (va.y - st2.x * lnv_a.y) * st2.y;
    va.z = (va.z - st2.x * lnv_a.z) * st2.y; va.w = (va.w - st2.x * lnv_a.w) * st2.y;
    vb.x = (vb.x - st2.x * lnv_b.x) * st2.y; vb.y = (vb.y - st2.x * lnv_b.y) * st2.y;
    vb.z = (vb.z - st2.x * lnv_b.z) * st2.y; vb.w = (vb.w - st2.x * lnv_b.w) * st2.y;
  }
  float x[8] = {va.x + bias_a.x, va.y + bias_a.y, va.z + bias_a.z, va.w + bias_a.w,
                vb.x + bias_b.x, vb.y + bias_b.y, vb.z + bias_b.z, vb.w + bias_b.w};
  if (ep.relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
  }
  x[0] += ra.x; x[1] += ra.y; x[2] += ra.z; x[3] += ra.w;
  x[4] += rb.x; x[5] += rb.y; x[6] += rb.z; x[7] += rb.w;
  if (ep.out_f32) {
    epi_store16(ep.out_f32 + o, x, ep.st_plain);
    epi_store16(ep.out_f32 + o + (ccol_b - ccol), x + 4, ep.st_plain);
  }
  if (ep.out_hi) {
    __half h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split_f16(x[e], h[e], l[e]);
    const size_t os = ep.out_slab ? slab_off(grow, ccol, ldm) : o;
#ifdef LTR_GEMM_NOSTORE   // diag: epilogue without its global stores (keeps the values alive through a never-true branch)
    if (h[0] == __half(12345.f) && l[7] == __half(54321.f))
#endif
    {
      epi_store16((__half*)ep.out_hi + os, h, ep.st_plain);
      if (!ep.no_lo_out) epi_store16((__half*)ep.out_lo + os, l, ep.st_plain);
    }
  }
  if (LNM == LNP) {
    // x[0..7] = columns ccol..+3 and ccol+32..+35 of the f32 row just stored (wide ownership: the 8 lanes of a
    // row hold its 64 columns in this wave).  Operand of the next GEMM: split(x gamma scale), slab-major -
    // per plane the 8 lanes x 8 rows of one store instruction cover 512 contiguous bytes of a slab.
    const float gs[8] = {lnv_a.x, lnv_a.y, lnv_a.z, lnv_a.w, lnv_b.x, lnv_b.y, lnv_b.z, lnv_b.w};
    __half h[8], l[8];
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float xs = x[e] * gs[e]; amax = fmaxf(amax, fabsf(xs)); split_f16(xs, h[e], l[e]); }
    // The operand carries the UN-normalised residual stream (times gamma, times 16): beyond fp16's 65504 its hi
    // plane is inf and the scores NaN, where separate LayerNorm launches would still work.  Flag it (NaN fails
    // the comparison too and is flagged).
    if (!(amax <= 65504.f) && ep.err_flag) atomicOr(ep.err_flag, 2);
    // lane pair (l, l ^ 1) holds columns {4k..4k+3, 32+4k..} and {4k+4..4k+7, 36+4k..}: swap halves so that the even
    // lane owns 8 consecutive columns of the first slab and the odd lane 8 of the second -> ONE 16-byte store
    // per plane and lane (8-byte stores issue at half the rate per byte)
    const bool odd = lane & 1;
    uint2 hk = *reinterpret_cast<const uint2*>(odd ? h + 4 : h), hs = *reinterpret_cast<const uint2*>(odd ? h : h + 4);
    uint2 lk = *reinterpret_cast<const uint2*>(odd ? l + 4 : l), ls = *reinterpret_cast<const uint2*>(odd ? l : l + 4);
    uint2 hr, lr;   // what the partner sends: its half that belongs to my slab
    hr.x = __shfl_xor(hs.x, 1, 64); hr.y = __shfl_xor(hs.y, 1, 64);
    lr.x = __shfl_xor(ls.x, 1, 64); lr.y = __shfl_xor(ls.y, 1, 64);
    const uint4 hv = odd ? make_uint4(hr.x, hr.y, hk.x, hk.y) : make_uint4(hk.x, hk.y, hr.x, hr.y);
    const uint4 lv = odd ? make_uint4(lr.x, lr.y, lk.x, lk.y) : make_uint4(lk.x, lk.y, lr.x, lr.y);
    const int c0 = odd ? ccol_b - 4 : ccol;               // first of my 8 consecutive columns
    const size_t oo = slab_off(grow, c0, ldm);
    epi_store16((__half*)ep.ln_hi + oo, &hv, ep.st_plain);
    if (!ep.no_lo_out) epi_store16((__half*)ep.ln_lo + oo, &lv, ep.st_plain);
    // (mean, M2) of this 64-column piece of the row (the shuffle partners lane ^ 1, 2, 4 hold the same
    // row, so they are active exactly when this lane is)
    float sm = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
    const float mu = sm * (1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float dd = x[e] - mu; q = fmaf(dd, dd, q); }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
    if ((lane & 7) == 0) ep.stats_out[(size_t)piece64 * ldm + grow] = make_float2(mu, q);
  }
}

// ONEP (LTR_F_ONE_PASS): the a_lo plane is neither streamed into LDS nor multiplied - one fp16 MFMA pass per product, the
// arithmetic of the reference's fp16 GPU predictor (the stage layout keeps its lo slot: same LDS addresses, same epilogue).
template <int LNM, bool RLN, bool ONEP = false>
__global__ void __launch_bounds__(512, 4) gemm_f16s_kernel(
    const __half* __restrict__ a_hi, const __half* __restrict__ a_lo, const __half* __restrict__ w, int M, int N,
    int K, int lda, int tiles_m, int tiles_n, int gm, Epilogue ep) {
  // 64 KiB of stages (+ 1 KiB: (mean, rstd/16) of the tile's 128 rows, LNC).  ONE array on purpose (see above).
  // (LNC: of the rows of A; RLN: of the rows of the residual - the two never meet in one GEMM)
  __shared__ __attribute__((aligned(16))) __half smem[2 * STAGE + (LNM == LNC || RLN ? 512 : 0)];

#ifdef LTR_GEMM_TIMELINE
  const unsigned long long tl0 = __builtin_readcyclecounter();
#endif
  if (LNM == LN_NONE && !RLN && gridDim.y > 1) {
    // split-K (GemmArgs::split_k; the weight-gradient GEMMs of the training step: few output tiles, K = the tokens): part
    // blockIdx.y multiplies its K columns (K = the columns of ONE part; slab-major A) into its own partial output
    // (lda = the row pitch of a row-major A: its parts are column ranges)
    const size_t part = blockIdx.y;
    const size_t aoff = ep.a_slab ? part * (size_t)K * ep.ldm : part * (size_t)K;
    a_hi += aoff; a_lo += aoff; w += part * (size_t)K * N;
    ep.out_f32 += part * (size_t)(M - ep.row0) * N;
  }
  int tm, tn;
  tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn, gm);
  const int m0 = ep.row0 + tm * BM, n0 = tn * BN16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
#ifdef LTR_GEMM_STAGGER
  // lab (profiles/r05_k768_epilogue_bound.txt): de-phase the two workgroups that share a CU.  They start together, walk the same
  // number of K-slabs and so reach their epilogues together (nobody on the CU issues MFMAs meanwhile); every later workgroup
  // inherits the phase of the one whose slot it takes.  Half of the FIRST wave of workgroups (index inside the XCD & MASK, below
  // 64) waits LTR_GEMM_STAGGER ticks of the 100 MHz clock before its first tile.
  {
    const int kx = blockIdx.x >> 3;
    if (kx < 64 && (kx & LTR_GEMM_STAGGER_MASK) && gridDim.x >= 1024) {
      const long long t0 = wall_clock64();
      while (wall_clock64() - t0 < LTR_GEMM_STAGGER) __builtin_amdgcn_s_sleep(16);
    }
  }
#endif

  // global_load_lds pieces of this lane per slab: one 16-row group of each activation plane
  // (rows wave*16 + (lane>>2)) and two 16-row groups of W (rows wave*32 + i*16 + (lane>>2))
  const __half *ga, *gl, *gw[2];
  {
    const int row = wave * 16 + (lane >> 2);
    const int c_log = (lane & 3) ^ swz(row);
    // rows past M/N: any valid row, masked later.  Row-major: row pitch K; slab-major: row pitch 32, slab pitch M*32
#ifdef LTR_GEMM_AROW_MOD   // diag (wrong results): every row tile of a K = 3072 GEMM reads the same few A rows - an L2-resident
                           // stand-in for "the fc1 -> fc2 intermediate never goes to memory" (profiles/r03_fused_and_fp8_probes.txt)
    const int arow = K == 3072 ? min(m0 + row, M - 1) % LTR_GEMM_AROW_MOD : min(m0 + row, M - 1);
    const size_t aoff = (size_t)arow * (ep.a_slab ? BK16 : lda) + c_log * 8;
#else
    const size_t aoff = (size_t)min(m0 + row, M - 1) * (ep.a_slab ? BK16 : lda) + c_log * 8;
#endif
    ga = a_hi + aoff;
    gl = a_lo + aoff;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int wrow = wave * 32 + i * 16 + (lane >> 2);
      const int wc_log = (lane & 3) ^ swz(wrow);
      gw[i] = w + (size_t)min(n0 + wrow, N - 1) * BK16 + wc_log * 8;   // slab-major weight image
    }
  }
  const size_t w_slab = (size_t)N * BK16, a_slab = ep.a_slab ? (size_t)ep.ldm * BK16 : (size_t)BK16;
  auto issue = [&](int stage, int k0) {
    __half* base = smem + stage * STAGE;
    const size_t ka = (size_t)(k0 / BK16) * a_slab, kw = (size_t)(k0 / BK16) * w_slab;
    // (default cache policy on purpose: the non-temporal hint on the activation stream costs 5 % of the call for the
    // narrow GEMMs alone and 11 % for all - the co-resident tiles share these lines through the L2)
#ifdef LTR_GEMM_A_ONCE   // diag (wrong results): a K = 3072 GEMM streams its A operand for the first two slabs only and multiplies
                         // those (valid, LDS-resident) fragments ever after - "the fc1 -> fc2 intermediate is already in LDS"
    if (K != 3072 || k0 < 2 * BK16) {
#else
    {
#endif
      __builtin_amdgcn_global_load_lds((gbl_void*)(ga + ka), (lds_void*)(base + wave * 16 * BK16), 16, 0, 0);
      if (!ONEP)
        __builtin_amdgcn_global_load_lds((gbl_void*)(gl + ka), (lds_void*)(base + A_PLANE + wave * 16 * BK16), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)(gw[i] + kw),
                                       (lds_void*)(base + 2 * A_PLANE + (wave * 32 + i * 16) * BK16), 16, 0, 0);
  };

  // v_mfma_f32_16x16x32_f16: one instruction covers the whole 32-wide K-slab of a 16x16 block.  It
  // moves half as many accumulator bytes per FLOP through the register file as the 32x32x16 form
  // and, under the package power cap that bounds this kernel, sustains 15 % more FLOP/s
  // (register-fed MFMA stream, real data: 121 vs 142 ms per bench step).
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fk = lane >> 4;
  const int nk = K / BK16;
#ifdef LTR_GEMM_TIMELINE
  unsigned long long tl_dma = 0, tl_bar = 0;
#endif
  issue(0, 0);
  if ((LNM == LNC || RLN) && tid < BM) {
    const int row = min(m0 + tid, M - 1);
    reinterpret_cast<float2*>(smem + 2 * STAGE)[tid] = tile_row_stat<RLN>(ep, row);
  }
  for (int kt = 0; kt < nk; ++kt) {
#ifdef LTR_GEMM_TIMELINE
    const unsigned long long w0 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of slab kt have landed
#ifdef LTR_GEMM_TIMELINE
    const unsigned long long w1 = __builtin_readcyclecounter();
#endif
    __syncthreads();                                    // everyone's have; slab kt-1 fully consumed
#ifdef LTR_GEMM_TIMELINE
    const unsigned long long w2 = __builtin_readcyclecounter();
    tl_dma += w1 - w0; tl_bar += w2 - w1;
#endif
    if (kt + 1 < nk) issue((kt + 1) & 1, (kt + 1) * BK16);
    const __half* s_ahi = smem + (kt & 1) * STAGE;
    const __half* s_alo = s_ahi + A_PLANE;
    const __half* s_w = s_ahi + 2 * A_PLANE;
    {
      f16x8 ah[4], al[4], bw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wr * 64 + i * 16 + frow;
        ah[i] = *reinterpret_cast<const f16x8*>(s_ahi + lds_off_h(row, fk));
        if (!ONEP) al[i] = *reinterpret_cast<const f16x8*>(s_alo + lds_off_h(row, fk));
        const int col = wc * 64 + i * 16 + frow;
        bw[i] = *reinterpret_cast<const f16x8*>(s_w + lds_off_h(col, fk));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#ifndef LTR_EXP_NO_LO    // perf probe (wrong results): what the lo pass costs
          if (!ONEP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bw[j], acc[i][j], 0, 0, 0);
#endif
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bw[j], acc[i][j], 0, 0, 0);
        }
    }
  }

#ifdef LTR_GEMM_TIMELINE
  const unsigned long long tl1 = __builtin_readcyclecounter();
#endif
#ifdef LTR_GEMM_EPI_1IN8
  // diag (wrong results; profiles/r05_k768_epilogue_bound.txt): in the two WIDE K = 768 shapes (QKV, fc1) only one tile in
  // LTR_GEMM_EPI_1IN8 runs its epilogue, the others drop their accumulators - what the K loops alone cost, i.e. the most an
  // epilogue that frees the accumulators early (and overlaps the next tile's K loop) could ever return
  if (K == 768 && N >= 2304 && (blockIdx.x % LTR_GEMM_EPI_1IN8) != 0) {
    float sink = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sink == 1.2345e-30f) ep.out_f32[0] = sink;     // (keeps the MFMAs alive; never true)
    return;
  }
#endif
  // ---- epilogue through LDS: per-wave strip [16 rows][64 cols] f32 (row stride CLD), four
  // strips per wave (its four 16-row MFMA blocks).
  // Each wave touches only its own strip, so after ONE workgroup barrier (the K-loop's LDS reads
  // are done) the strips are wave-local: LDS executes a wave's instructions in order, an
  // s_waitcnt between the writes and the transposed reads is all the synchronisation needed.
  // Read-back: a lane owns 8 consecutive columns -> 16-B stores for f32 and for each fp16 plane.
  float* s_c = reinterpret_cast<float*>(smem) + wave * CROWS * CLD;
  // Column ownership of a lane in the read-back (two float4 per row): split outputs need 8 CONSECUTIVE
  // columns (one 16-byte store per fp16 plane: 8 lanes = a 128-B row piece); f32-only outputs take columns
  // 4k..4k+3 and 32+4k..32+4k+3 instead, so that each of the two f32 stores / residual loads of the 8 lanes
  // of a row covers one full 128-byte line (with 8 consecutive columns they interleave in 16-byte pieces).
  const bool wide = ep.out_hi == nullptr;
  const int erow = lane >> 3, ecol = (lane & 7) * (wide ? 4 : 8), ecol_b = wide ? ecol + 32 : ecol + 4;
  const int ccol = n0 + wc * 64 + ecol, ccol_b = n0 + wc * 64 + ecol_b;
  float4 bias_a = make_float4(0.f, 0.f, 0.f, 0.f), bias_b = bias_a;
  if (ep.bias && ccol < N) {
    bias_a = *reinterpret_cast<const float4*>(ep.bias + ccol);
    bias_b = *reinterpret_cast<const float4*>(ep.bias + ccol_b);
  }
    float4 lnv_a = make_float4(0.f, 0.f, 0.f, 0.f), lnv_b = lnv_a;      // LNP: gamma * scale; LNC: c_n
  if ((LNM == LNP || LNM == LNC) && !RLN && ccol < N) {              // (RLN instances fetch it in their second sweep)
    const float* src = LNM == LNP ? ep.ln_gamma : ep.ln_c;
    lnv_a = *reinterpret_cast<const float4*>(src + ccol);
    lnv_b = *reinterpret_cast<const float4*>(src + ccol_b);
    if (LNM == LNP) {
      lnv_a.x *= LN_FOLD_SCALE; lnv_a.y *= LN_FOLD_SCALE; lnv_a.z *= LN_FOLD_SCALE; lnv_a.w *= LN_FOLD_SCALE;
      lnv_b.x *= LN_FOLD_SCALE; lnv_b.y *= LN_FOLD_SCALE; lnv_b.z *= LN_FOLD_SCALE; lnv_b.w *= LN_FOLD_SCALE;
    }
  }
  __syncthreads();
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    // 16x16 C layout: col = lane & 15, row = 4 * (lane >> 4) + e; strip st = the wave's 16-row block st
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) s_c[(4 * (lane >> 4) + e) * CLD + j * 16 + (lane & 15)] = acc[st][j][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (ccol < N && RLN) {
      // LayerNorm'd residual in TWO sweeps over the strip.  Everything at once - two rows of accumulator + residual in
      // flight, the affine vectors of the residual's LayerNorm, bias, the producer's gamma - does not fit the 128-VGPR
      // budget of four waves per SIMD: the compiler spilled, and the spilled LNP + RLN instance produced NaNs (its
      // inline-asm stores sit behind scratch reloads).  Sweep 1 turns the strip into x = acc + bias + LN(resid) in
      // place (wave-local LDS), sweep 2 is the ordinary epilogue on x.  (One row at a time in one sweep also fits, and
      // cost OPT-350m's producers 8 %.)
      float4 ra[2], rb[2];
      int gr[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {            // both rows' residuals in flight (global), the accumulators come from LDS
        gr[it] = m0 + wr * 64 + st * 16 + it * 8 + erow;
        ra[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        rb[it] = ra[it];
        if (gr[it] < M) {
          const size_t o = (size_t)gr[it] * N + ccol;
          ra[it] = *reinterpret_cast<const float4*>(ep.resid + o);
          rb[it] = *reinterpret_cast<const float4*>(ep.resid + o + (ecol_b - ecol));
        }
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int srow = it * 8 + erow;
        const float2 rst = reinterpret_cast<const float2*>(smem + 2 * STAGE)[min(gr[it], M - 1) - m0];
        {
          const float4 va = *reinterpret_cast<const float4*>(s_c + srow * CLD + ecol);
          const float4 g = *reinterpret_cast<const float4*>(ep.r_gamma + ccol), b = *reinterpret_cast<const float4*>(ep.r_beta + ccol);
          float4 x;
          x.x = va.x + bias_a.x + ((ra[it].x - rst.x) * rst.y * g.x + b.x);
          x.y = va.y + bias_a.y + ((ra[it].y - rst.x) * rst.y * g.y + b.y);
          x.z = va.z + bias_a.z + ((ra[it].z - rst.x) * rst.y * g.z + b.z);
          x.w = va.w + bias_a.w + ((ra[it].w - rst.x) * rst.y * g.w + b.w);
          *reinterpret_cast<float4*>(s_c + srow * CLD + ecol) = x;
        }
        {
          const float4 vb = *reinterpret_cast<const float4*>(s_c + srow * CLD + ecol_b);
          const float4 g = *reinterpret_cast<const float4*>(ep.r_gamma + ccol_b), b = *reinterpret_cast<const float4*>(ep.r_beta + ccol_b);
          float4 x;
          x.x = vb.x + bias_b.x + ((rb[it].x - rst.x) * rst.y * g.x + b.x);
          x.y = vb.y + bias_b.y + ((rb[it].y - rst.x) * rst.y * g.y + b.y);
          x.z = vb.z + bias_b.z + ((rb[it].z - rst.x) * rst.y * g.z + b.z);
          x.w = vb.w + bias_b.w + ((rb[it].w - rst.x) * rst.y * g.w + b.w);
          *reinterpret_cast<float4*>(s_c + srow * CLD + ecol_b) = x;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 ga = z4, gb = z4;                     // the producer's gamma * scale, fetched here (not held across sweep 1)
      if (LNM == LNP) {
        ga = *reinterpret_cast<const float4*>(ep.ln_gamma + ccol);
        gb = *reinterpret_cast<const float4*>(ep.ln_gamma + ccol_b);
        ga.x *= LN_FOLD_SCALE; ga.y *= LN_FOLD_SCALE; ga.z *= LN_FOLD_SCALE; ga.w *= LN_FOLD_SCALE;
        gb.x *= LN_FOLD_SCALE; gb.y *= LN_FOLD_SCALE; gb.z *= LN_FOLD_SCALE; gb.w *= LN_FOLD_SCALE;
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        if (gr[it] >= M) continue;
        const int srow = it * 8 + erow;
        const float4 xa = *reinterpret_cast<const float4*>(s_c + srow * CLD + ecol);
        const float4 xb = *reinterpret_cast<const float4*>(s_c + srow * CLD + ecol_b);
        epilogue_piece<LNM, false>(ep, xa, xb, z4, z4, z4, z4, ga, gb, make_float2(0.f, 0.f), make_float2(0.f, 0.f), gr[it],
                                   ccol, ccol_b, (size_t)gr[it] * N + ccol, ep.ldm, lane, tn * 4 + wc);
      }
    } else     if (ccol < N) {
      float4 va[2], vb[2], ra[2], rb[2];
      size_t o[2];
      int gr[2];
      bool ok[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int srow = it * 8 + erow;
        const int grow = m0 + wr * 64 + st * 16 + srow;
        ok[it] = grow < M;
        gr[it] = grow;
        o[it] = (size_t)grow * N + ccol;
        va[it] = *reinterpret_cast<const float4*>(s_c + srow * CLD + ecol);
        vb[it] = *reinterpret_cast<const float4*>(s_c + srow * CLD + ecol_b);
        ra[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        rb[it] = ra[it];
        if (ep.resid && ok[it]) {
          ra[it] = *reinterpret_cast<const float4*>(ep.resid + o[it]);
          rb[it] = *reinterpret_cast<const float4*>(ep.resid + o[it] + (ecol_b - ecol));
        }
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        if (!ok[it]) continue;
        float2 st2 = make_float2(0.f, 0.f);
        if (LNM == LNC || RLN) st2 = reinterpret_cast<const float2*>(smem + 2 * STAGE)[gr[it] - m0];
        if (LNM == LNS) st2 = out_scale_stat(ep);
        epilogue_piece<LNM, RLN>(ep, va[it], vb[it], ra[it], rb[it], bias_a, bias_b, lnv_a, lnv_b, st2, st2, gr[it], ccol,
                                   ccol_b, o[it], ep.ldm, lane, tn * 4 + wc);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#ifdef LTR_GEMM_TIMELINE
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_timeline[blockIdx.x * 4 + 0] = tl0;
    g_timeline[blockIdx.x * 4 + 1] = tl1;
    g_timeline[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();
    g_timeline[blockIdx.x * 4 + 3] = ((unsigned long long)xcc << 32) | hwid;
    g_waits[blockIdx.x * 2] = tl_dma; g_waits[blockIdx.x * 2 + 1] = tl_bar;
  }
#endif
}

// ------------------------------------------------------------------------------------
// F16 split mode, SMALL batches (the steady scheduler step: a handful of arrivals to score, SURVEY.md 8d "steady").
// With M of a few hundred rows the 128 x 256 kernel has 3-12 workgroups, each walking its K-slabs behind a
// two-stage ring at one DMA round trip per slab: 37 us per launch, 49 launches per scored request (2.1 ms for ONE
// 86-token prompt, profiles/r02_ab_gemm_probes.txt section 5).  This variant spends the chip differently:
//  * small tiles (BM x BN = 32 x 64 with 4 waves, 64 x 128 or 64 x 256 with 8) so that a 1-request batch still makes
//    100-500 workgroups and a 16-request batch fills every CU;
//  * 64-wide K stages in a RING OF FOUR (three stages in flight per workgroup, counted vmcnt, one raw barrier per
//    stage) - latency-bound, so what matters is bytes in flight per CU, not bytes per FLOP;
//  * an XCD grid of row ranges x column ranges chosen per launch, and K parts for the narrow long-K shape (launch_gemm);
//  * the same fragment layout, the same MFMA (lo pass then hi pass per 32-wide slab, slabs in order) and the same
//    epilogue arithmetic as the large-tile kernel.  Without K parts the results are bit-identical to it; with them the
//    sums associate differently: a request's score moves by <= 2e-6 with the kernel a batch size selects, a call is
//    deterministic (tests/test_gpu_small_batches.py).
// ------------------------------------------------------------------------------------
// Configuration of gemm_f16s_small_kernel: BM x BN tile, WM x WN waves (each (BM / WM) x (BN / WN)), SL 32-wide K-slabs
// per ring stage, SSTAGES stages.  The shipped instances (which one runs: choose_small, from profiles/r04_small_gemm_lab.txt):
//   32 x  64, 2 x 2 waves, 64-wide stages, ring of 4 (64 KiB)
//   64 x 128, 2 x 4 waves, 64-wide stages, ring of 4 (128 KiB)
//   64 x 256, 2 x 4 waves, 32-wide stages, ring of 4 (128 KiB)
// (128 x 256, 2 x 4 waves, 32-wide stages, ring of 4 - the large-tile kernel's tile behind a deep ring at one workgroup
//  per CU - 32 x 128 and 64 x 64 also instantiate in the lab and are parity-green, but lose at every size)
template <int BM_, int BN_, int WM_, int WN_, int SL_, int SSTAGES> struct SmallCfg {
  static constexpr int WM = WM_, WN = WN_;
  static constexpr int NW = WM * WN;
  static constexpr int TI = BM_ / WM / 16, TJ = BN_ / WN / 16;      // MFMA blocks per wave
  static constexpr int PA = BM_ / 16 * 2, PW = BN_ / 16;            // 1-KiB DMA pieces per 32-wide slab: A hi|lo, W
  static constexpr int SUB = (PA + PW) * 512;                       // halves per 32-wide slab image
  static constexpr int SL = SL_;                                    // slabs per stage
  static constexpr int STAGE_H = SL * SUB;                          // halves per stage
  static constexpr int PIECES = SL * (PA + PW) / NW;                // DMA instructions per wave and stage
  static constexpr int CLDS = BN_ + 4;                              // f32 row stride of the epilogue tile
  static constexpr int EH = ((size_t)BM_ * CLDS * 4 <= (size_t)SSTAGES * STAGE_H * 2) ? 1 : WM;   // epilogue passes (row blocks of the waves)
  static constexpr int EROWS = BM_ / EH;
  static constexpr size_t LDS_BYTES = (size_t)SSTAGES * STAGE_H * 2 + BM_ * 8;
  static_assert(SL * (PA + PW) % NW == 0 && PIECES >= 1 && PIECES <= 8, "piece map: 1-8 DMA instructions per wave and stage");
  static_assert(SSTAGES >= 3 && SSTAGES <= 8, "counted vmcnt waits cover up to 6 stages in flight");
  static_assert((size_t)EROWS * CLDS * 4 <= (size_t)SSTAGES * STAGE_H * 2, "epilogue tile must fit the ring");
  static_assert(BM_ % (WM * 16) == 0 && BN_ % (WN * 16) == 0 &&
                ((EROWS * BN_ / 8) % (NW * 64) == 0 || (NW * 64) % (EROWS * BN_ / 8) == 0), "tile / wave grid mismatch");
};

// Workgroup -> tile map of the small-tile kernels: the 8 XCDs (block b runs on XCD b % 8, private 4 MiB L2s) form an
// RX x CX grid, XCD (i, j) owns row range i and column range j of the tile grid and walks it N fastest.  A weight matrix
// is 1.2-4.7 MB: with CX column ranges an XCD keeps only its 1 / CX of it resident, while the activation rows stream
// through once per XCD column (fabric traffic CX * A + RX * W instead of 8 * W with every XCD thrashing its L2 on the
// whole matrix).  Blocks past an XCD's own tile count exit (the grid is 8 * the largest count).
struct XcdMap { int rx, cx, per_xcd; };
__host__ __device__ __forceinline__ void xcd_range(int n, int parts, int i, int& lo, int& cnt) {
  const int q = n / parts, r = n % parts;
  lo = i * q + min(i, r);
  cnt = q + (i < r ? 1 : 0);
}
__device__ __forceinline__ bool xcd_tile(int bid, int tiles_m, int tiles_n, XcdMap mp, int& tm, int& tn) {
  const int x = bid % NXCD, l = bid / NXCD;
  const int xi = x / mp.cx, xj = x % mp.cx;
  int r0, nr, c0, nc;
  xcd_range(tiles_m, mp.rx, xi, r0, nr);
  xcd_range(tiles_n, mp.cx, xj, c0, nc);
  if (l >= nr * nc) return false;
  tm = r0 + l / nc;
  tn = c0 + l % nc;
  return true;
}

template <int LNM, bool RLN, int BM_, int BN_, int WM_, int WN_, int SL_, int SSTAGES>
__global__ void __launch_bounds__((SmallCfg<BM_, BN_, WM_, WN_, SL_, SSTAGES>::NW * 64)) gemm_f16s_small_kernel(
    const __half* __restrict__ a_hi, const __half* __restrict__ a_lo, const __half* __restrict__ w, int M, int N,
    int K, int lda, int tiles_m, int tiles_n, XcdMap xmap, Epilogue ep) {
  using C = SmallCfg<BM_, BN_, WM_, WN_, SL_, SSTAGES>;
  if (LNM == LN_NONE && !RLN && gridDim.y > 1) {
    // split-K (launch_gemm "small-batch split-K"): part blockIdx.y multiplies its K columns (K = the columns of ONE part;
    // lda = the row pitch of a row-major A) into its own raw f32 partial [M, N]; splitk_epilogue_kernel adds the parts in
    // part order and runs the epilogue
    const size_t part = blockIdx.y;
    const size_t aoff = ep.a_slab ? part * (size_t)K * ep.ldm : part * (size_t)K;
    a_hi += aoff; a_lo += aoff; w += part * (size_t)K * N;
    ep.out_f32 += part * (size_t)(M - ep.row0) * N;
  }
  // dynamic LDS on purpose: with a static array hipcc tracks the LDS-DMA stores against every ds_read and drains
  // vmcnt(0) in front of the first fragment read of each stage (ltr_attn.hip has the same note)
  extern __shared__ __attribute__((aligned(16))) __half smem[];
  float2* s_stat = reinterpret_cast<float2*>(smem + SSTAGES * C::STAGE_H);
#ifdef LTR_GEMM_TIMELINE
  const unsigned long long tl0 = __builtin_readcyclecounter();
  unsigned long long tl1 = 0;
#endif
  int tm, tn;
  if (!xcd_tile(blockIdx.x, tiles_m, tiles_n, xmap, tm, tn)) return;
  const int m0 = ep.row0 + tm * BM_, n0 = tn * BN_;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / C::WN, wc = wave % C::WN;

  // this wave's DMA pieces of a stage: piece q = wave * PIECES + p of the stage's SL x (PA + PW); slab q / (PA + PW),
  // then r = q % (PA + PW): A hi groups, A lo groups, W groups - the LDS image of a slab is exactly r * 1 KiB
  constexpr int NP = C::PIECES;
  const __half* gsrc[NP];
  size_t gstep[NP];     // source advance per 32-wide slab
  int ldst[NP];         // halves from the stage base
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = wave * NP + p, sl = q / (C::PA + C::PW), r = q % (C::PA + C::PW);
    const int row16 = lane >> 2, c_log = (lane & 3) ^ swz(row16);
    ldst[p] = sl * C::SUB + r * 512;
    if (r < C::PA) {
      const int g = r % (C::PA / 2);
      const int row = min(m0 + g * 16 + row16, M - 1);
      gstep[p] = ep.a_slab ? (size_t)ep.ldm * BK16 : (size_t)BK16;
      gsrc[p] = (r < C::PA / 2 ? a_hi : a_lo) + (size_t)row * (ep.a_slab ? BK16 : lda) + c_log * 8 + sl * gstep[p];
    } else {
      const int g = r - C::PA;
      const int row = min(n0 + g * 16 + row16, N - 1);
      gstep[p] = (size_t)N * BK16;
      gsrc[p] = w + (size_t)row * BK16 + c_log * 8 + sl * gstep[p];
    }
  }
  auto issue = [&](int slot, int kt) {
    __half* base = smem + slot * C::STAGE_H;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      __builtin_amdgcn_global_load_lds((gbl_void*)(gsrc[p] + (size_t)(C::SL * kt) * gstep[p]), (lds_void*)(base + ldst[p]), 16, 0, 0);
  };

  f32x4 acc[C::TI][C::TJ];
#pragma unroll
  for (int i = 0; i < C::TI; ++i)
#pragma unroll
    for (int j = 0; j < C::TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fk = lane >> 4;
  const int nst = K / (BK16 * C::SL);
#pragma unroll
  for (int st = 0; st < SSTAGES - 1; ++st)
    if (st < nst) issue(st, st);
  if ((LNM == LNC || RLN) && tid < BM_) {     // (mean, M2) pieces of the row -> (mean, rstd [/ scale]); see the large-tile kernel
    const int row = min(m0 + tid, M - 1);
    s_stat[tid] = tile_row_stat<RLN>(ep, row);
  }
  for (int kt = 0; kt < nst; ++kt) {
    // stage kt has landed once at most the younger stages' pieces (NP per stage and wave) are outstanding
    const int ahead = min(SSTAGES - 2, nst - 1 - kt);
    switch (ahead) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * NP) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NP) : "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NP) : "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * NP) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * NP) : "memory"); break;
    }
    __builtin_amdgcn_s_barrier();                       // everyone's pieces of stage kt landed; stage kt-1 fully consumed
#ifdef LTR_GEMM_TIMELINE
    if (kt == 0) tl1 = __builtin_readcyclecounter();
#endif
    if (kt + SSTAGES - 1 < nst) issue((kt + SSTAGES - 1) % SSTAGES, kt + SSTAGES - 1);
    const __half* sb = smem + (kt % SSTAGES) * C::STAGE_H;
#pragma unroll
    for (int sl = 0; sl < C::SL; ++sl) {
      const __half* s_ahi = sb + sl * C::SUB;
      const __half* s_alo = s_ahi + BM_ * BK16;
      const __half* s_w = s_ahi + 2 * BM_ * BK16;
      f16x8 ah[C::TI], al[C::TI], bw[C::TJ];
#pragma unroll
      for (int i = 0; i < C::TI; ++i) {
        const int row = wr * (C::TI * 16) + i * 16 + frow;
        ah[i] = *reinterpret_cast<const f16x8*>(s_ahi + lds_off_h(row, fk));
        al[i] = *reinterpret_cast<const f16x8*>(s_alo + lds_off_h(row, fk));
      }
#pragma unroll
      for (int j = 0; j < C::TJ; ++j) {
        const int col = wc * (C::TJ * 16) + j * 16 + frow;
        bw[j] = *reinterpret_cast<const f16x8*>(s_w + lds_off_h(col, fk));
      }
#pragma unroll
      for (int i = 0; i < C::TI; ++i)
#pragma unroll
        for (int j = 0; j < C::TJ; ++j) {
          if (!ep.one_pass) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bw[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bw[j], acc[i][j], 0, 0, 0);
        }
    }
  }
#ifdef LTR_GEMM_TIMELINE
  const unsigned long long tl2 = __builtin_readcyclecounter();
#endif
  // ---- epilogue through LDS (the ring is idle): the tile - or, when it does not fit, one wave-row block of it after
  // the other - then one (row, 8 columns) piece per lane and pass.  (Fetching the bias / LayerNorm-fold vectors /
  // residual row of the piece BEFORE the K loop, to take them off the tail of the launch, was measured and is not
  // faster.)
  float* s_c = reinterpret_cast<float*>(smem);
  const bool wide = ep.out_hi == nullptr;
  const int l8 = lane & 7;
  const int ecol = l8 * (wide ? 4 : 8), ecol_b = wide ? ecol + 32 : ecol + 4;
  constexpr int P64 = BN_ / 64;                                      // 64-column pieces per tile row
  constexpr int WROWS = C::TI * 16;                                  // rows of a wave
#pragma unroll
  for (int eh = 0; eh < C::EH; ++eh) {
    __syncthreads();                                                 // ring / previous pass fully consumed
    const int rbase = eh * C::EROWS;                                 // first tile row of this pass
    if (C::EH == 1 || wr == eh) {
      const int lrow0 = wr * WROWS - rbase;
#pragma unroll
      for (int i = 0; i < C::TI; ++i)
#pragma unroll
        for (int j = 0; j < C::TJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            s_c[(lrow0 + i * 16 + 4 * (lane >> 4) + e) * C::CLDS + wc * (C::TJ * 16) + j * 16 + (lane & 15)] = acc[i][j][e];
    }
    __syncthreads();
    constexpr int NPIECE = C::EROWS * BN_ / 8;                         // (row, 8 columns) pieces of this pass
#pragma unroll
    for (int pi = 0; pi < (NPIECE + C::NW * 64 - 1) / (C::NW * 64); ++pi) {
      if (NPIECE < C::NW * 64 && tid >= NPIECE) continue;              // more threads than pieces (many-wave configurations)
      const int idx = (pi * (C::NW * 64) + tid) >> 3;                // (row, piece) index; the 8 lanes of a piece are consecutive
      const int lrow = idx / P64, pc = idx % P64;
      const int srow = rbase + lrow;                                 // row inside the tile
      const int grow = m0 + srow;
      const int ccol = n0 + pc * 64 + ecol, ccol_b = n0 + pc * 64 + ecol_b;
      if (grow >= M || ccol >= N) continue;
      const float4 va = *reinterpret_cast<const float4*>(s_c + lrow * C::CLDS + pc * 64 + ecol);
      const float4 vb = *reinterpret_cast<const float4*>(s_c + lrow * C::CLDS + pc * 64 + ecol_b);
      const size_t o = (size_t)grow * N + ccol;
      float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
      if (ep.resid) {
        ra = *reinterpret_cast<const float4*>(ep.resid + o);
        rb = *reinterpret_cast<const float4*>(ep.resid + o + (ecol_b - ecol));
      }
      float4 bias_a = make_float4(0.f, 0.f, 0.f, 0.f), bias_b = bias_a;
      if (ep.bias) {
        bias_a = *reinterpret_cast<const float4*>(ep.bias + ccol);
        bias_b = *reinterpret_cast<const float4*>(ep.bias + ccol_b);
      }
      float4 lnv_a = make_float4(0.f, 0.f, 0.f, 0.f), lnv_b = lnv_a;    // LNP: gamma * scale; LNC: c_n
      if (LNM == LNP || LNM == LNC) {
        const float* src = LNM == LNP ? ep.ln_gamma : ep.ln_c;
        lnv_a = *reinterpret_cast<const float4*>(src + ccol);
        lnv_b = *reinterpret_cast<const float4*>(src + ccol_b);
        if (LNM == LNP) {
          lnv_a.x *= LN_FOLD_SCALE; lnv_a.y *= LN_FOLD_SCALE; lnv_a.z *= LN_FOLD_SCALE; lnv_a.w *= LN_FOLD_SCALE;
          lnv_b.x *= LN_FOLD_SCALE; lnv_b.y *= LN_FOLD_SCALE; lnv_b.z *= LN_FOLD_SCALE; lnv_b.w *= LN_FOLD_SCALE;
        }
      }
      float2 st2 = make_float2(0.f, 0.f);
      if (LNM == LNC || RLN) st2 = s_stat[srow];
      if (LNM == LNS) st2 = out_scale_stat(ep);
      epilogue_piece<LNM, RLN>(ep, va, vb, ra, rb, bias_a, bias_b, lnv_a, lnv_b, st2, st2, grow, ccol, ccol_b, o, ep.ldm, lane,
                               tn * P64 + pc);
    }
  }
#ifdef LTR_GEMM_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the stores have left the wave)
  if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.y == 0) {
    g_timeline[blockIdx.x * 4 + 0] = tl0; g_timeline[blockIdx.x * 4 + 1] = tl1;
    g_timeline[blockIdx.x * 4 + 2] = tl2; g_timeline[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
  }
#endif
}

// Second half of a small-batch split-K GEMM: x = sum over the parts (in part order - deterministic) of the raw f32
// partials [parts][M][N], then the ordinary epilogue (bias, ReLU, residual / LayerNorm'd residual, f32 / split stores,
// LayerNorm-fold producer outputs) through the same epilogue_piece the GEMM kernels use: one (row, 8 columns) piece per
// thread, the 8 lanes of a 64-column piece consecutive.
template <int LNM, bool RLN>
__global__ void __launch_bounds__(256) splitk_epilogue_kernel(const float* __restrict__ partial, int parts, int M, int N,
                                                              Epilogue ep) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, l8 = gid & 7;
  const int p64 = N / 64;
  const int idx = gid >> 3;
  const int grow = ep.row0 + idx / p64, pc = idx % p64;
  const bool live = grow < M;            // (whole 8-lane groups are live or not: the shuffles inside epilogue_piece stay inside a group)
  const bool wide = ep.out_hi == nullptr;
  const int ecol = l8 * (wide ? 4 : 8), ecol_b = wide ? ecol + 32 : ecol + 4;
  const int ccol = pc * 64 + ecol, ccol_b = pc * 64 + ecol_b;
  if (!live) return;
  const size_t o = (size_t)grow * N + ccol;
  const size_t po = (size_t)(grow - ep.row0) * N + ccol;          // the partials hold the rows of the window only
  const size_t pstride = (size_t)(M - ep.row0) * N;
  float4 va = *reinterpret_cast<const float4*>(partial + po);
  float4 vb = *reinterpret_cast<const float4*>(partial + po + (ccol_b - ccol));
  for (int p = 1; p < parts; ++p) {
    const float4 a = *reinterpret_cast<const float4*>(partial + p * pstride + po);
    const float4 b = *reinterpret_cast<const float4*>(partial + p * pstride + po + (ccol_b - ccol));
    va.x += a.x; va.y += a.y; va.z += a.z; va.w += a.w;
    vb.x += b.x; vb.y += b.y; vb.z += b.z; vb.w += b.w;
  }
  float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
  if (ep.resid) {
    ra = *reinterpret_cast<const float4*>(ep.resid + o);
    rb = *reinterpret_cast<const float4*>(ep.resid + o + (ccol_b - ccol));
  }
  float4 bias_a = make_float4(0.f, 0.f, 0.f, 0.f), bias_b = bias_a;
  if (ep.bias) {
    bias_a = *reinterpret_cast<const float4*>(ep.bias + ccol);
    bias_b = *reinterpret_cast<const float4*>(ep.bias + ccol_b);
  }
  float4 lnv_a = make_float4(0.f, 0.f, 0.f, 0.f), lnv_b = lnv_a;
  if (LNM == LNP) {
    lnv_a = *reinterpret_cast<const float4*>(ep.ln_gamma + ccol);
    lnv_b = *reinterpret_cast<const float4*>(ep.ln_gamma + ccol_b);
    lnv_a.x *= LN_FOLD_SCALE; lnv_a.y *= LN_FOLD_SCALE; lnv_a.z *= LN_FOLD_SCALE; lnv_a.w *= LN_FOLD_SCALE;
    lnv_b.x *= LN_FOLD_SCALE; lnv_b.y *= LN_FOLD_SCALE; lnv_b.z *= LN_FOLD_SCALE; lnv_b.w *= LN_FOLD_SCALE;
  }
  float2 rst = make_float2(0.f, 0.f);
  if (RLN) rst = tile_row_stat<true>(ep, grow);
  epilogue_piece<LNM, RLN>(ep, va, vb, ra, rb, bias_a, bias_b, lnv_a, lnv_b, rst, rst, grow, ccol, ccol_b, o, ep.ldm, lane, pc);
}

// ------------------------------------------------------------------------------------
// F32 exact mode.  K-slab = 16, LDS holds the slab transposed [16 k][128 + 4 rows] so the
// one-float-per-lane fragments of v_mfma_f32_32x32x2_f32 (lane l: row l&31, k = l>>5) are
// stride-1 ds_read_b32 (conflict-free).
// ------------------------------------------------------------------------------------
constexpr int BK32 = 16;
constexpr int LDT = BM + 4;

__global__ void __launch_bounds__(256, 2) gemm_f32_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                         int M, int N, int K, int tiles_m, int tiles_n,
                                                         Epilogue ep) {
  __shared__ float s_a[BK32 * LDT];
  __shared__ float s_w[BK32 * LDT];
  int tm, tn;
  tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int srow0 = tid >> 2, skq = tid & 3, srow1 = srow0 + 64;
  const size_t ga0 = (size_t)min(m0 + srow0, M - 1) * K + skq * 4;
  const size_t ga1 = (size_t)min(m0 + srow1, M - 1) * K + skq * 4;
  const size_t gw0 = (size_t)min(n0 + srow0, N - 1) * K + skq * 4;
  const size_t gw1 = (size_t)min(n0 + srow1, N - 1) * K + skq * 4;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra0, ra1, rw0, rw1;
  auto gload = [&](int k0) {
    ra0 = *reinterpret_cast<const float4*>(a + ga0 + k0);
    ra1 = *reinterpret_cast<const float4*>(a + ga1 + k0);
    rw0 = *reinterpret_cast<const float4*>(w + gw0 + k0);
    rw1 = *reinterpret_cast<const float4*>(w + gw1 + k0);
  };
  auto sstore = [&](float* s, int row, const float4& v) {
    s[(skq * 4 + 0) * LDT + row] = v.x;
    s[(skq * 4 + 1) * LDT + row] = v.y;
    s[(skq * 4 + 2) * LDT + row] = v.z;
    s[(skq * 4 + 3) * LDT + row] = v.w;
  };
  gload(0);
  const int frow = lane & 31, fk = lane >> 5;
  for (int k0 = 0; k0 < K; k0 += BK32) {
    __syncthreads();
    sstore(s_a, srow0, ra0);
    sstore(s_a, srow1, ra1);
    sstore(s_w, srow0, rw0);
    sstore(s_w, srow1, rw1);
    __syncthreads();
    if (k0 + BK32 < K) gload(k0 + BK32);
#pragma unroll
    for (int kk = 0; kk < BK32 / 2; ++kk) {
      const int k = kk * 2 + fk;
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = s_a[k * LDT + wr * 64 + i * 32 + frow];
        bv[i] = s_w[k * LDT + wc * 64 + i * 32 + frow];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      store_tile<false>(ep, acc[i][j], m0 + wr * 64 + i * 32, n0 + wc * 64 + j * 32 + (lane & 31));
}

}  // namespace

#ifdef LTR_GEMM_TIMELINE
int gemm_waits_read(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_waits), sizeof(unsigned long long) * 8192 * 2); }
int gemm_timeline_read(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 8192 * 4); }
#endif

namespace {
// nn.Linear weight [N][K] (K contiguous) -> slab-major image [K/32][N][32]; one 16-B piece per thread
__global__ void __launch_bounds__(256) pack_weight_kernel(const __half* __restrict__ src, __half* __restrict__ dst, int N,
                                                          int K, int n_src) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // piece index in the destination
  const size_t pieces = (size_t)N * K / 8;
  if (i >= pieces) return;
  const int c = (int)(i & 3);                 // 16-B chunk inside the 64-B slab row
  const size_t rn = i >> 2;                   // slab * N + n
  const int n = (int)(rn % N), slab = (int)(rn / N);
  *reinterpret_cast<uint4*>(dst + i * 8) =
      n < n_src ? *reinterpret_cast<const uint4*>(src + (size_t)n * K + slab * BK16 + c * 8) : make_uint4(0, 0, 0, 0);
}

}  // namespace

namespace {
// c_n = scale * sum_k gamma_k W[n, k], d_n = sum_k beta_k W[n, k] + b_n for the LayerNorm fold; one wave per output
// row, double accumulation (one-time, at ltr_create)
__global__ void __launch_bounds__(256) ln_fold_coeff_kernel(const __half* __restrict__ w /*[N, K] row-major*/,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ bias, int N, int K,
                                                            float* __restrict__ c_out, float* __restrict__ d_out) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  double c = 0.0, d = 0.0;
  for (int k = lane; k < K; k += 64) {
    const double wv = (double)__half2float(w[(size_t)n * K + k]);
    c += (double)gamma[k] * wv;
    d += (double)beta[k] * wv;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); d += __shfl_xor(d, o, 64); }
  if (lane == 0) { c_out[n] = (float)(c * (double)LN_FOLD_SCALE); d_out[n] = (float)(d + (double)(bias ? bias[n] : 0.f)); }
}
}  // namespace

int launch_ln_fold_coeff(const void* w_f16, const float* gamma, const float* beta, const float* bias, int N, int K,
                         float* c_out, float* d_out, hipStream_t s) {
  ln_fold_coeff_kernel<<<(N + 3) / 4, 256, 0, s>>>((const __half*)w_f16, gamma, beta, bias, N, K, c_out, d_out);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

int launch_row_stats_combine(const void* parts, int n_part, int ldm, int rows, void* out, hipStream_t s) {
  if (rows <= 0) return LTR_OK;
  row_stats_combine_kernel<<<(rows + 255) / 256, 256, 0, s>>>((const float2*)parts, n_part, ldm, rows, (float2*)out);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

int launch_pack_weight(const void* src, void* dst, int N, int K, hipStream_t s, int n_src) {
  if (K % BK16) { set_error("pack_weight: K=%d must be a multiple of %d", K, BK16); return LTR_E_INVAL; }
  const size_t pieces = (size_t)N * K / 8;
  pack_weight_kernel<<<(unsigned)((pieces + 255) / 256), 256, 0, s>>>((const __half*)src, (__half*)dst, N, K,
                                                                      n_src < 0 ? N : n_src);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

namespace {
// LTR_GEMM_SMALL_M (diag): row count up to which the small-batch kernels are considered at all (0: never - every GEMM on
// the 128 x 256 two-stage kernel, the round-2 behaviour)
int small_m_limit() { static const int v = [] { const char* e = getenv("LTR_GEMM_SMALL_M"); return e ? atoi(e) : 4800; }(); return v; }

// Which kernel a GEMM of a small batch runs on, and in how many K parts: measured choice (profiles/r04_small_gemm_lab.txt,
// diag/small_lab.sh: every tile configuration x split x row count of a scheduler step with 1 ... 256 arrivals, OPT-125m and
// OPT-350m widths), reduced to thresholds on the row count M.  What the measurements say, in one paragraph: in this regime
// a CU fills its LDS at ~30 GB/s from lines nobody on its XCD has touched yet and ~3x faster from the XCD's L2, and the
// chip as a whole at ~6 TB/s - so the best tile is the one that (a) puts about one workgroup on every CU (a second ROUND
// of workgroups doubles the time: 264 tiles on 256 CUs is the worst case), (b) moves the fewest bytes per workgroup, and
// above ~1,700 rows the 128 x 256 tile's bytes per FLOP win although it is latency-bound (0.95 us per K-slab).
//   cfg: -1 = 128 x 256 two-stage kernel; small-batch kernel 0 = 32 x 64, 1 = 64 x 128, 5 = 64 x 256 (2: 128 x 256 behind a
//   ring of four, 3: 32 x 128, 4: 64 x 64 exist for the lab and lose everywhere)
struct SmallChoice { int cfg, parts; };
SmallChoice choose_small(const GemmArgs& g, bool can_split) {
  const int M = g.M, N = g.N, K = g.K;
  if (K % 64 || N % 64 || M > small_m_limit()) return {-1, 1};
  const bool n128 = N % 128 == 0, n256 = N % 256 == 0;
  auto fits = [&](int c) { return c == 0 || (c == 1 && n128) || (c == 5 && n256); };
  SmallChoice ch{-1, 1};
  if (!can_split) {                                  // wide outputs (QKV, fc1) and everything that is not a residual producer
    if (M <= 400) ch.cfg = ((M + 31) / 32) * (N / 64) <= 512 || !n128 ? 0 : 1;
    else if (M <= 700) ch.cfg = n128 ? 1 : 0;
    else if (M <= 1700) ch.cfg = n256 && ((M + 63) / 64) * (N / 256) <= 256 ? 5 : -1;
  } else if (K <= N + N / 2) {                       // narrow output, short K (out_proj)
    if (M <= 1100) ch.cfg = 0;
    else if (M <= 3000) ch.cfg = n128 ? 1 : 0;
    else ch.cfg = n256 ? 5 : -1;
  } else {                                           // narrow output, long K (fc2): split-K
    if (M <= 400) ch = {0, 4};
    else if (M <= 700) ch = {0, 2};
    else if (M <= 1100) ch = {n256 ? 5 : 0, n256 ? 4 : 1};
    else if (M <= 3000) ch = {-1, 4};
    else ch = {-1, 2};
  }
  if (ch.cfg >= 0 && !fits(ch.cfg)) ch.cfg = 0;
  return ch;
}
}  // namespace

// which F16 kernel a GEMM of this shape runs on: -1 the 128 x 256 kernel, >= 0 a small-batch configuration (profiling classes)
int gemm_small_config(const GemmArgs& g) {
  const bool can_split = g.splitk_ws && !g.ln_stats_in && !g.osc_a && g.out_f32 && !g.out_split.hi && !g.relu;
  return choose_small(g, can_split).cfg;
}

int launch_gemm(int wdtype, const GemmArgs& g, hipStream_t s) {
  if (g.M == 0) return LTR_OK;
  const int kmult = wdtype == LTR_W_F16 ? BK16 : BK32;
  if (g.K % kmult || g.N % 64) {
    set_error("gemm: K=%d must be a multiple of %d and N=%d of 64", g.K, kmult, g.N);
    return LTR_E_INVAL;
  }
  const int bn = wdtype == LTR_W_F16 ? BN16 : BN;
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + bn - 1) / bn;
  if ((g.row0 || g.ldm) && (wdtype != LTR_W_F16 || g.row0 < 0 || (g.ldm && g.ldm < g.row0 + g.M) || g.split_k > 1)) {
    set_error("gemm: a row window needs F16 mode, 0 <= row0, row0 + M <= ldm and no split_k");
    return LTR_E_INVAL;
  }
  const int Mend = g.row0 + g.M, ldm = g.ldm ? g.ldm : Mend;       // the kernels' M is the END row of the window
  Epilogue ep{g.bias, g.resid, g.out_f32, g.out_split.hi, g.out_split.lo, Mend, g.N, g.relu, g.a_slab, g.out_slab,
              g.ln_gamma, g.ln_out.hi, g.ln_out.lo, (float2*)g.ln_stats_out, (const float2*)g.ln_stats_in, g.ln_c,
              g.ln_parts, g.err_flag, (const float2*)g.rln_stats, g.rln_gamma, g.rln_beta, g.rln_parts, g.osc_a, g.osc_b,
              g.row0, ldm, g.one_pass, g.no_lo_out, (const float2*)g.ln_stats_comb, (const float2*)g.rln_stats_comb, g.store_plain};
  const int lnm = g.ln_gamma ? LNP : (g.ln_stats_in ? LNC : (g.osc_a ? LNS : LN_NONE));
  if (g.osc_a && (wdtype != LTR_W_F16 || !g.osc_b || g.ln_gamma || g.ln_stats_in || g.rln_stats)) {
    set_error("gemm: scaled operands (osc_a / osc_b) need F16 mode, both scales and no LayerNorm fold");
    return LTR_E_INVAL;
  }
  const bool rln = g.rln_stats != nullptr;
  if (rln && (wdtype != LTR_W_F16 || lnm == LNC || g.relu || !g.resid || !g.rln_gamma || !g.rln_beta || g.rln_parts * 64 != g.N ||
              g.rln_stats == g.ln_stats_out)) {
    set_error("gemm: a LayerNorm'd residual needs F16 mode, resid, gamma / beta [N], N / 64 statistics pieces distinct from the "
              "producer's output and no consumer epilogue");
    return LTR_E_INVAL;
  }
  if (lnm != LN_NONE) {
    if (wdtype != LTR_W_F16) { set_error("gemm: the LayerNorm fold exists in F16 mode only"); return LTR_E_INVAL; }
    if (lnm == LNP && (g.out_split.hi || !g.out_f32 || !g.ln_out.hi || !g.ln_stats_out || g.N % 64)) {
      set_error("gemm: LN producer needs an f32 output only, the operand planes, the stats buffer and N %% 64 == 0");
      return LTR_E_INVAL;
    }
    if (lnm == LNC && (!g.ln_c || g.ln_parts <= 0 || g.ln_parts * 64 != g.K || !g.a_slab)) {
      set_error("gemm: LN consumer needs c[N], K / 64 stats pieces and a slab-major A");
      return LTR_E_INVAL;
    }
  }
  if (wdtype != LTR_W_F16 && (g.a_slab || g.out_slab)) { set_error("gemm: slab-major operands exist in F16 mode only"); return LTR_E_INVAL; }
  if (g.out_slab && g.N % 32) { set_error("gemm: slab-major output needs N %% 32 == 0"); return LTR_E_INVAL; }
  dim3 grid(tiles_m * tiles_n);
  const int split = g.split_k > 1 ? g.split_k : 1;
  if (split > 1 && (wdtype != LTR_W_F16 || lnm != LN_NONE || rln || !g.a_slab || !g.out_f32 || g.out_split.hi || g.bias || g.resid ||
                    g.relu || g.K % (BK16 * split))) {
    set_error("gemm: split_k needs F16 mode, a slab-major A, K %% (32 split_k) == 0 and a plain f32 output [split_k][M][N]");
    return LTR_E_INVAL;
  }
  if (wdtype == LTR_W_F16) {
    // Order inside a group of 8 row tiles: M fastest for wide outputs (QKV, fc1: 9-12 column tiles; consecutive
    // workgroups share a weight panel), N fastest for narrow ones (out_proj, fc2: <= 4 column tiles; the column tiles of
    // a row start together and share its activation panel - at K = 3072 a panel is 1.5 MB and the L2 holds ~2 us of
    // the XCD's stream, so only tiles in lockstep share).  Measured: -2 ms per call for the narrow shapes, +2 ms if
    // applied to the wide ones (profiles/r02_ab_gemm_gm.txt).  LTR_GEMM_GM / LTR_GEMM_GM_NARROW: diag overrides
    // (group size; + 65536 = N fastest).
    static const int gm_wide = [] { const char* e = getenv("LTR_GEMM_GM"); return e ? atoi(e) : GM_DEFAULT; }();
    static const int gm_narrow = [] { const char* e = getenv("LTR_GEMM_GM_NARROW"); return e ? atoi(e) : (GM_DEFAULT | 65536); }();
    const int gm = tiles_n <= 4 ? gm_narrow : gm_wide;
    // Small batches (a scheduler step with a few arrivals): the small-tile, deep-ring kernels, chosen per launch by
    // choose_small (LTR_GEMM_SMALL_M: the row count up to which they are considered at all; LTR_GEMM_FORCE_CFG /
    // LTR_GEMM_FORCE_SPLIT: lab and tests).  (The 128 x 256 tile behind a ring of four 32-wide stages at one workgroup per
    // CU - SmallCfg<128, 256, 2, 4, 1, 4>, parity-green - was measured for 1k-23k rows and is slower than its neighbours
    // everywhere: 1,382 tokens 2.21 vs 1.42 ms per call, 5,928: 3.31 vs 3.02, 23,078: 9.58 vs 8.61 - not instantiated.)
    static const int map_mode = [] { const char* e = getenv("LTR_GEMM_SMALL_MAP"); return e ? atoi(e) : 3; }();
    static const int force_cfg = [] { const char* e = getenv("LTR_GEMM_FORCE_CFG"); return e ? atoi(e) : -2; }();       // diag: -1 big, 0, 1
    static const int force_split = [] { const char* e = getenv("LTR_GEMM_FORCE_SPLIT"); return e ? atoi(e) : 0; }();  // diag: parts (1 = off)
    // tile menu of the small-batch kernel: BM x BN (waves, K stage, ring) - see the instantiation list below
    static const int CFG_BM[9] = {32, 64, 128, 32, 64, 64, 128, 64, 32}, CFG_BN[9] = {64, 128, 256, 128, 64, 256, 256, 128, 64};
    // ---- small-batch split-K.  A narrow output (out_proj, fc2: N = H) of a small batch has few tiles (M = 262 rows:
    // 9 x 12 = 108 on 256 CUs) and, for fc2, a long K: the workgroups that exist each stream hundreds of KB through one
    // ring at the rate ONE CU pulls from the fabric.  Cutting K into `parts` multiplies the workgroups and divides the
    // stream each walks; the raw partials (parts x M x N x 4 B: small, because N is) are summed in part order by
    // splitk_epilogue_kernel.  Only producers of the residual stream qualify (no ReLU, f32 output).
    const bool can_split = split <= 1 && g.splitk_ws && lnm != LNC && lnm != LNS && g.out_f32 && !g.out_split.hi && !g.relu;
    SmallChoice ch = split > 1 ? SmallChoice{-1, 1} : choose_small(g, can_split);
    int cfg = ch.cfg;
    if (split <= 1 && (force_cfg == -1 || ((force_cfg == 0 || force_cfg == 1 || (force_cfg >= 5 && force_cfg <= 8)) && g.K % 64 == 0 &&
                                          g.N % CFG_BN[force_cfg] == 0)))
      cfg = force_cfg;
    int parts = can_split ? ch.parts : 1;
    if (can_split && force_split > 0) parts = force_split;
    while (parts > 1 && (g.K % (64 * parts) || (size_t)parts * g.M * g.N * 4 > g.splitk_ws_bytes)) parts /= 2;
    Epilogue epk = ep;                                                 // what the GEMM kernel itself stores
    int lnm_k = lnm;
    bool rln_k = rln;
    if (parts > 1) {
      epk = Epilogue{};
      // raw partials [parts][window rows][N]; the kernels index rows globally: bias the base by the window's first row
      epk.out_f32 = (float*)g.splitk_ws - (size_t)g.row0 * g.N; epk.M = Mend; epk.N = g.N; epk.a_slab = g.a_slab;
      epk.row0 = g.row0; epk.ldm = ldm; epk.one_pass = g.one_pass;
      epk.st_plain = 1;            // the partials (<= 59 MB) are read back by the reduce launch right behind this one
      lnm_k = LN_NONE; rln_k = false;
    }
    const int kpart = g.K / parts;
    auto finish_split = [&]() -> int {                                 // second half of a split launch
      const unsigned nthreads = (unsigned)g.M * (unsigned)(g.N / 8);
      const dim3 rgrid((nthreads + 255) / 256);
      const float* part = (const float*)g.splitk_ws;
      if (rln) { if (lnm == LNP) splitk_epilogue_kernel<LNP, true><<<rgrid, 256, 0, s>>>(part, parts, Mend, g.N, ep);
                 else splitk_epilogue_kernel<LN_NONE, true><<<rgrid, 256, 0, s>>>(part, parts, Mend, g.N, ep); }
      else if (lnm == LNP) splitk_epilogue_kernel<LNP, false><<<rgrid, 256, 0, s>>>(part, parts, Mend, g.N, ep);
      else splitk_epilogue_kernel<LN_NONE, false><<<rgrid, 256, 0, s>>>(part, parts, Mend, g.N, ep);
      LTR_LAUNCH_CHECK();
      return LTR_OK;
    };
    if (cfg >= 0) {
      const int bm = CFG_BM[cfg], bnn = CFG_BN[cfg];
      {
        const int tm_ = (g.M + bm - 1) / bm, tn_ = g.N / bnn;
        // XCD grid (block b runs on XCD b % 8, private 4 MiB L2s): rx row ranges x cx column ranges.  An XCD pulls the
        // activation rows of its row range and the weight rows of its column range over the fabric, so the launch moves
        // cx * A + rx * W bytes across it: take the (rx, cx) that minimises that (A = M K 4 bytes, W = N K 2 bytes);
        // map_mode 0: every XCD a row range (round 3), 1 / 2: column ranges sized from W / as many as possible.
        int cx = 1;
        if (map_mode == 1) while (cx < 8 && cx * 2 <= tn_ && (size_t)g.N * g.K * 2 / cx > (size_t)3 << 19) cx *= 2;
        if (map_mode == 2) { cx = 8; while (cx > tn_) cx /= 2; }
        if (map_mode == 3) {
          const double A = 4.0 * g.M * g.K, W = 2.0 * g.N * g.K;
          double best = 1e300;
          for (int c = 1; c <= 8; c *= 2) {
            const int r = NXCD / c;
            if (c > tn_ || r > tm_) continue;
            const double cost = c * A + r * W;
            if (cost < best) { best = cost; cx = c; }
          }
        }
        XcdMap xm{NXCD / cx, cx, 0};
        {
          int lo, a, b;
          xcd_range(tm_, xm.rx, 0, lo, a);
          xcd_range(tn_, xm.cx, 0, lo, b);
          xm.per_xcd = a * b;                                          // range 0 is never the shorter one
        }
        dim3 sgrid(xm.per_xcd * NXCD, parts);
#define LTR_SMALL_LAUNCH(LN, RL, BMv, BNv, WMv, WNv, SLv, STv)                                                             \
  do {                                                                                                                     \
    typedef SmallCfg<BMv, BNv, WMv, WNv, SLv, STv> Cfg;                                                                    \
    static const bool attr_ok = [] {                                                                                       \
      return hipFuncSetAttribute((const void*)gemm_f16s_small_kernel<LN, RL, BMv, BNv, WMv, WNv, SLv, STv>,                \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES) == hipSuccess;           \
    }();                                                                                                                   \
    (void)attr_ok;                                                                                                         \
    gemm_f16s_small_kernel<LN, RL, BMv, BNv, WMv, WNv, SLv, STv><<<sgrid, Cfg::NW * 64, Cfg::LDS_BYTES, s>>>(               \
        (const __half*)g.a.hi, (const __half*)g.a.lo, (const __half*)g.w, Mend, g.N, kpart, g.K, tm_, tn_, xm, epk);         \
  } while (0)
#define LTR_SMALL_LN(...)                                                                                                  \
  do {                                                                                                                     \
    if (rln_k) { if (lnm_k == LNP) LTR_SMALL_LAUNCH(LNP, true, __VA_ARGS__); else LTR_SMALL_LAUNCH(LN_NONE, true, __VA_ARGS__); } \
    else if (lnm_k == LNP) LTR_SMALL_LAUNCH(LNP, false, __VA_ARGS__);                                                      \
    else if (lnm_k == LNC) LTR_SMALL_LAUNCH(LNC, false, __VA_ARGS__);                                                      \
    else if (lnm_k == LNS) LTR_SMALL_LAUNCH(LNS, false, __VA_ARGS__);                                                      \
    else LTR_SMALL_LAUNCH(LN_NONE, false, __VA_ARGS__);                                                                    \
  } while (0)
        // (a ring of eight stages for grids of at most one workgroup per CU was measured and is slower: fc2 of a
        // one-request call 21.7 vs 15.7 us, profiles/r03_small_batch.txt)
        // (lab-only configurations that lost at every row count, profiles/r04_small_gemm_lab.txt: 2 = 128 x 256 behind a
        // ring of four (SmallCfg<128, 256, 2, 4, 1, 4>), 3 = 32 x 128 (<32, 128, 2, 2, 2, 4>), 4 = 64 x 64 (<64, 64, 2, 2, 2, 4>))
        if (cfg == 0) LTR_SMALL_LN(32, 64, 2, 2, 2, 4);
        else if (cfg == 1) LTR_SMALL_LN(64, 128, 2, 4, 2, 4);
        else if (cfg == 5) LTR_SMALL_LN(64, 256, 2, 4, 1, 4);
#ifdef LTR_GEMM_LAB_WAVES   // lab: the same tiles with twice the waves (does a CU's LDS-DMA rate scale with the waves that issue it?)
        else if (cfg == 6) LTR_SMALL_LN(128, 256, 4, 4, 1, 4);
        else if (cfg == 7) LTR_SMALL_LN(64, 128, 4, 4, 2, 4);
        else if (cfg == 8) LTR_SMALL_LN(32, 64, 2, 4, 2, 4);
#endif
        else { set_error("gemm: tile configuration %d is not built", cfg); return LTR_E_INVAL; }
#undef LTR_SMALL_LN
#undef LTR_SMALL_LAUNCH
        LTR_LAUNCH_CHECK();
        return parts > 1 ? finish_split() : LTR_OK;
      }
    }
    // ---- tail rows (LTR_GEMM_TAIL=1, off by default: measured, no net gain).  The 128 x 256 kernel runs 512 tiles at a time
    // (two workgroups per CU): 543 tiles - fc2 of a 23,078-row pass - look like two rounds for 6 % more work than one.  With
    // the switch on, the rows beyond the last full round go to a second launch_gemm on a row window, where the small-batch
    // kernels take them.  Measured at 23,078 rows (profiles/r04_small_gemm_lab.txt): fc2 210 -> 187 us, out_proj 86 -> 81, but
    // QKV 188 -> 193 and fc1 174 -> 190: the 31 tiles of a "second round" run alone on the chip at twice the per-tile speed of
    // a full round, so the tail costs far less than a round, and the extra launch eats what is left.
    static const int tail_on = [] { const char* e = getenv("LTR_GEMM_TAIL"); return e ? atoi(e) : 0; }();
    if (tail_on && split <= 1 && parts == 1 && force_cfg == -2) {
      const int tiles = tiles_m * tiles_n, rounds = tiles / 512;
      if (rounds >= 1 && tiles % 512) {
        const int full_row_tiles = (rounds * 512) / tiles_n;
        const int R = full_row_tiles * BM, tail = g.M - R;
        const int tail_tiles = tiles - full_row_tiles * tiles_n;
        if (R > 0 && tail > 0 && tail <= (can_split ? 4800 : 3000) && tail_tiles <= 320) {
          GemmArgs g1 = g, g2 = g;
          g1.M = R; g1.ldm = ldm;
          g2.row0 = g.row0 + R; g2.M = tail; g2.ldm = ldm;
          const int rc = launch_gemm(wdtype, g1, s);
          return rc ? rc : launch_gemm(wdtype, g2, s);
        }
      }
    }
    if (parts > 1) {                     // the 128 x 256 kernel, split: raw partials, then the epilogue kernel
      grid.y = parts;
      if (g.one_pass)
        gemm_f16s_kernel<LN_NONE, false, true><<<grid, 512, 0, s>>>((const __half*)g.a.hi, (const __half*)g.a.lo, (const __half*)g.w,
                                                                   Mend, g.N, kpart, g.K, tiles_m, tiles_n, gm, epk);
      else
        gemm_f16s_kernel<LN_NONE, false><<<grid, 512, 0, s>>>((const __half*)g.a.hi, (const __half*)g.a.lo, (const __half*)g.w, Mend, g.N,
                                                             kpart, g.K, tiles_m, tiles_n, gm, epk);
      LTR_LAUNCH_CHECK();
      return finish_split();
    }
    grid.y = split;
#define LTR_BIG_LAUNCH(LN, RL)                                                                                             \
  do {                                                                                                                     \
    if (g.one_pass)                                                                                                        \
      gemm_f16s_kernel<LN, RL, true><<<grid, 512, 0, s>>>((const __half*)g.a.hi, (const __half*)g.a.lo, (const __half*)g.w, Mend,  \
                                                          g.N, g.K / split, g.K, tiles_m, tiles_n, gm, ep);                \
    else                                                                                                                   \
      gemm_f16s_kernel<LN, RL><<<grid, 512, 0, s>>>((const __half*)g.a.hi, (const __half*)g.a.lo, (const __half*)g.w, Mend, g.N,   \
                                                    g.K / split, g.K, tiles_m, tiles_n, gm, ep);                           \
  } while (0)
    if (rln) { if (lnm == LNP) LTR_BIG_LAUNCH(LNP, true); else LTR_BIG_LAUNCH(LN_NONE, true); }
    else if (lnm == LNP) LTR_BIG_LAUNCH(LNP, false);
    else if (lnm == LNC) LTR_BIG_LAUNCH(LNC, false);
    else if (lnm == LNS) LTR_BIG_LAUNCH(LNS, false);
    else LTR_BIG_LAUNCH(LN_NONE, false);
#undef LTR_BIG_LAUNCH
  } else {
    gemm_f32_kernel<<<grid, 256, 0, s>>>((const float*)g.a.hi, (const float*)g.w, g.M, g.N, g.K, tiles_m, tiles_n,
                                         ep);
  }
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // namespace ltr
