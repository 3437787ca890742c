// Varlen causal self-attention of the predictor's prefill on gfx950.
//
// Reference: OPTAttention.forward (opt.py:92-102) with kv_cache=None, i.e. per-sequence
// softmax(q k^T * d^-0.5, causal) v over heads of 64 (rocm_flash_attn.py:244-290 on GPU,
// torch_sdpa.py:138-178 on CPU).  Input is the packed QKV GEMM output f32 [T, 3H]
// (q | k | v); output is the out_proj GEMM operand [T, H].
//
// Query blocks of all requests are enumerated through a device-side prefix table so the
// launch needs no host knowledge of the lengths.
//
// F16 ("split") mode - attn_f16s_kernel, the production kernel.  q, k, v arrive as fp16
// hi|lo planes written by the QKV GEMM epilogue.  A workgroup is 4 waves = 128 consecutive
// queries of one (request, head), 32 per wave; K/V tiles of 32 keys stream HBM -> LDS with
// global_load_lds (ring of three stages, one raw barrier per tile).  Per 32x32 (key, query) block:
//   S^T = K Q^T      on v_mfma_f32_32x32x16_f16, 3 passes (kh*qh + kh*ql + kl*qh)
//   online softmax   lane-local: with the S^T layout a lane holds 16 keys of ONE query
//                    (col = lane & 31), so row max / sum are 15 in-lane ops + one
//                    lane<->lane^32 exchange, and the rescale factor is a per-lane scalar
//   O^T += V^T P^T   3 passes (vh*ph + vh*pl + vl*ph); the S^T accumulator registers,
//                    converted to fp16, ARE the B operand (the key <-> k-slot assignment
//                    of an MFMA is arbitrary as long as A and B agree, so the V^T fragment
//                    is read from the row-major V tile in that key order with the
//                    transpose read ds_read_b64_tr_b16: no cross-lane shuffle and no
//                    transposed copy of V).
// The dropped lo*lo terms are ~2^-22 relative: f32-grade, as in the GEMMs.
//
// F32 mode - attn_f32_kernel: exact f32 VALU flash kernel, one query row per lane.
#include "ltr_internal.h"

namespace ltr {
namespace {

constexpr int QB = 64;     // queries per workgroup
constexpr int KT = 32;     // keys per LDS tile
constexpr int D = 64;      // head size (OPT-125m/350m: 768/12 = 1024/16 = 64)
constexpr float NEG = -1.0e30f;

// Work list of the attention launch: blk_start[n_req] = number of query blocks = sum_i ceil(L_i / qb) and,
// per query block b, blk_desc[b] = (request, first query of the block, first token row of the request in this
// chunk, request length).  The attention kernels read ONE 16-byte descriptor instead of binary-searching a
// prefix table and then chasing cu_seqlens (every dependent global load costs a workgroup of a short prompt
// ~1.5 us of its ~20 us life).  Blocks are listed longest first - a block's cost grows with its position k in
// the request (it streams keys 0 .. (k+1) qb) - in three classes k >= 4, k in {2, 3}, k in {0, 1}, so that the
// few long workgroups of a launch start early instead of forming its tail.
__global__ void __launch_bounds__(1024) attn_blocks_kernel(const int32_t* __restrict__ cu, int n_req, int qb,
                                                           int32_t* __restrict__ blk_start,
                                                           int4* __restrict__ blk_desc) {
  __shared__ int s_w[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int cls = 0; cls < 3; ++cls) {
    for (int base = 0; base < n_req; base += 1024) {
      const int i = base + tid;
      const int c0 = (i < n_req) ? cu[i] : 0, len = (i < n_req) ? cu[i + 1] - c0 : 0;
      const int nb = (len + qb - 1) / qb;
      const int lo = cls == 0 ? 4 : (cls == 1 ? 2 : 0);
      const int hi = cls == 0 ? nb : (cls == 1 ? min(nb, 4) : min(nb, 2));
      const int cnt = max(hi - lo, 0);
      int v = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
      if (lane == 63) s_w[wave] = v;
      __syncthreads();
      int off = carry;
      for (int w = 0; w < wave; ++w) off += s_w[w];
      v += off;
      if (cnt > 0) {
        const int t0 = c0 - cu[0];
        for (int k = lo; k < hi; ++k) blk_desc[v - cnt + (k - lo)] = make_int4(i, k * qb, t0, len);
      }
      __syncthreads();
      if (tid == 1023) carry = v;
      __syncthreads();
    }
  }
  if (tid == 0) blk_start[n_req] = carry;
}

template <bool SPLIT>
__global__ void __launch_bounds__(64) attn_f32_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                      const int32_t* __restrict__ blk_start,
                                                      const int4* __restrict__ blk_desc, int n_req, int H,
                                                      float scale_log2e, void* out_hi, void* out_lo,
                                                      float* __restrict__ lse2 /*nullable: [T, heads] m + log2(l)*/) {
  __shared__ __attribute__((aligned(16))) float s_k[KT * D];
  __shared__ __attribute__((aligned(16))) float s_v[KT * D];
  const int b = blockIdx.x;
  if (b >= blk_start[n_req]) return;
  const int head = blockIdx.y;
  const int lane = threadIdx.x;
  const int4 desc = blk_desc[b];
  const int q0 = desc.y;
  const int t0 = desc.z;
  const int L = desc.w;
  const int qi = q0 + lane;
  const bool valid = qi < L;
  const size_t ld = (size_t)3 * H;

  float q[D], o[D];
  {
    const float* qp = qkv + (size_t)(t0 + (valid ? qi : 0)) * ld + head * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float4 v = *reinterpret_cast<const float4*>(qp + d);
      q[d] = v.x * scale_log2e; q[d + 1] = v.y * scale_log2e;
      q[d + 2] = v.z * scale_log2e; q[d + 3] = v.w * scale_log2e;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
  }
  float m = NEG, l = 0.f;
  const int kend = min(L, q0 + QB);   // causal: keys [0, kend)
  for (int kt = 0; kt < kend; kt += KT) {
    __syncthreads();
    // stage K and V rows kt..kt+KT-1 (clamped): 16 lanes x float4 cover one 256-B row
#pragma unroll
    for (int it = 0; it < KT * D / 4 / 64; ++it) {
      const int idx = it * 64 + lane;
      const int key = idx >> 4, d4 = (idx & 15) * 4;
      const int kk = min(kt + key, L - 1);
      const float* base = qkv + (size_t)(t0 + kk) * ld + head * D + d4;
      *reinterpret_cast<float4*>(s_k + key * D + d4) = *reinterpret_cast<const float4*>(base + H);
      *reinterpret_cast<float4*>(s_v + key * D + d4) = *reinterpret_cast<const float4*>(base + 2 * H);
    }
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < KT; sub += 16) {
      if (kt + sub > q0 + QB - 1 || kt + sub >= kend) break;   // wave-uniform
      float s[16];
      float mx = NEG;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float* kr = s_k + (sub + j) * D;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
          float4 kv = *reinterpret_cast<const float4*>(kr + d);
          acc = fmaf(q[d], kv.x, acc); acc = fmaf(q[d + 1], kv.y, acc);
          acc = fmaf(q[d + 2], kv.z, acc); acc = fmaf(q[d + 3], kv.w, acc);
        }
        const int kj = kt + sub + j;
        s[j] = (kj <= qi && kj < L) ? acc : NEG;
        mx = fmaxf(mx, s[j]);
      }
      const float m_new = fmaxf(m, mx);
      const float alpha = exp2f(m - m_new);
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] *= alpha;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float p = exp2f(s[j] - m_new);
        l += p;
        const float* vr = s_v + (sub + j) * D;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
          float4 vv = *reinterpret_cast<const float4*>(vr + d);
          o[d] = fmaf(p, vv.x, o[d]); o[d + 1] = fmaf(p, vv.y, o[d + 1]);
          o[d + 2] = fmaf(p, vv.z, o[d + 2]); o[d + 3] = fmaf(p, vv.w, o[d + 3]);
        }
      }
    }
  }
  if (!valid) return;
  const float inv = 1.f / l;
  const size_t ob = (size_t)(t0 + qi) * H + head * D;
  if (lse2 != nullptr) lse2[(size_t)(t0 + qi) * (H / D) + head] = m + log2f(l);   // training: softmax is recomputed in the backward
  if (SPLIT) {
    __half* ph = (__half*)out_hi + ob;
    __half* pl = (__half*)out_lo + ob;
#pragma unroll
    for (int d = 0; d < D; d += 8) {
      __half hh[8], ll[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) split_f16(o[d + i] * inv, hh[i], ll[i]);
      *reinterpret_cast<uint4*>(ph + d) = *reinterpret_cast<const uint4*>(hh);
      *reinterpret_cast<uint4*>(pl + d) = *reinterpret_cast<const uint4*>(ll);
    }
  } else {
    float* po = (float*)out_hi + ob;
#pragma unroll
    for (int d = 0; d < D; d += 4)
      *reinterpret_cast<float4*>(po + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
  }
}

// ------------------------------------------------------------------------------------
// split-fp16 MFMA kernel
// ------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TK = 32;   // keys per tile = one 32x32 MFMA block
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// LDS stage: K hi | K lo | V hi | V lo, each [32 keys][64 d] halves (4 KiB), filled by
// global_load_lds (lane-linear image: one wave instruction = 8 rows x 128 B).
// K: 16-B chunk c of row r holds logical chunk c ^ ((r >> 1) & 7) (source-side swizzle) so
// the ds_read_b128 A-fragment reads are conflict-free.  V: row-major; its fragments (8 keys of
// one d column per lane) come out of ds_read_b64_tr_b16, the hardware transpose read, so no
// transposed copy of V is ever written (2-byte gathers cost 64 reads + ~40 packs per tile).
constexpr int PLANE_H = TK * D;                 // halves per plane
constexpr int ATT_STAGE = 4 * PLANE_H;          // 16 KiB
constexpr int NSTAGE = 3;
__device__ __forceinline__ int k_off(int row, int c) { return row * D + ((c ^ ((row >> 1) & 7)) << 3); }
// V: 32-B column block cb of key row r sits at block cb ^ ((r >> 1) & 1), so that the four key rows a
// transpose read touches (two 256-B bank rows) fall on distinct banks
// LTR_ATTN_VSWZ: which 16-B chunk index bits are flipped on odd row pairs (2 = 32-B blocks, round 1-2; 4 = 64-B halves;
// 6 = both).  A transpose read covers 4 key rows x 64 B per 32 lanes: rows k, k+1 share a 256-B bank row, rows k+2, k+3
// the next one, so without the 64-B flip rows k and k+2 sit on the same banks.
#ifndef LTR_ATTN_VSWZ
#define LTR_ATTN_VSWZ 6
#endif
__device__ __forceinline__ int v_off(int row, int d) {
  return row * D + ((((d >> 3) ^ (((row >> 1) & 1) * LTR_ATTN_VSWZ)) << 3) | (d & 7));
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#ifdef LTR_ATTN_TIMELINE
__device__ unsigned long long g_attn_tl[4096 * 5];
#endif
// SKV ("split K/V", the small grids of a scheduler step with a few arrivals): a workgroup owns 32 queries instead of 128 and
// its four waves share them.  Per K/V tile a wave spends ~2,300 cycles (12 + 12 MFMAs of 32 x 32 x 16, the softmax, the
// fragment reads - diag/attn_small_timeline.hip), and a lone workgroup has nothing to hide that behind: the last query
// block of a 262-token prompt walks nine tiles = 9 us.  Here wave w walks the tiles w, w + 4, ... as its OWN stream - it
// loads them itself into a private two-stage ring (no workgroup barrier in the loop) - and the four (m, l, O) states are
// merged through LDS at the end: nine tiles become three per wave.  128 KiB of LDS, one workgroup per CU.
// (Dealing the tiles to the waves while everybody still loads every tile behind a barrier gains nothing: the barrier makes
// each iteration as long as its one working wave.)
// ONEP (LTR_F_ONE_PASS): q, k, v and p as plain fp16 - the hi planes only are streamed and multiplied (one MFMA pass per
// product instead of three), the lo plane of the output is not stored (its reader, out_proj, runs one pass too): the
// arithmetic of an fp16 flash attention, as in the reference's GPU predictor (rocm_flash_attn.py:244-290 on fp16 tensors).
// NW = 8 (round 5, lab: LTR_ATTN_NW=8): 256 queries per workgroup.  A request longer than 128 tokens is walked by several
// 128-query workgroups, each of which streams the keys [0, its last query] again: 32 % (ShareGPT profile) to 54 % (LMSYS
// profile) more q|k|v bytes than the tensors hold, at a kernel that runs at the fabric's copy rate.  With eight waves one K/V
// tile in LDS serves 256 queries.  The per-wave code is unchanged (32 queries per wave); the tile loads are dealt to eight
// waves instead of four (waves 0-3 the K planes, waves 4-7 the V planes).
template <int NW, bool SKV, bool ONEP = false>
__global__ void __launch_bounds__(NW * 64, SKV ? 1 : (NW == 8 ? 4 : 3)) attn_f16s_kernel(
    const __half* __restrict__ qkv_hi, const __half* __restrict__ qkv_lo, const int32_t* __restrict__ cu,
    const int32_t* __restrict__ blk_start, const int4* __restrict__ blk_desc, int n_req, int H, float scale_log2e,
    __half* __restrict__ out_hi, __half* __restrict__ out_lo,
    float* __restrict__ lse2 /*nullable: [T, heads] m + log2(l) of every query row (training)*/) {
  // 48 KiB ring, tiles it+1 and it+2 in flight.  Dynamic LDS on purpose: with a static array hipcc tracks the
  // LDS-DMA stores against every ds_read and drains vmcnt(0) in front of the first fragment read
  extern __shared__ __attribute__((aligned(16))) __half smem[];
  constexpr int QBLK = SKV ? 32 : 32 * NW;
  static_assert(NW == 4 || (NW == 8 && !SKV), "load maps below: 4 waves (one 8-row group of each plane per wave) or 8 (K | V split)");
  constexpr int LPT = (NW == 4 ? 4 : 2) / (ONEP ? 2 : 1);      // tile loads per wave (the counted vmcnt waits below)

#ifdef LTR_ATTN_TIMELINE
  const unsigned long long tl0 = __builtin_readcyclecounter();
  unsigned long long tl1 = 0, tl2 = 0;
#endif
  const int b = blockIdx.x;
  if (b >= blk_start[n_req]) return;
  const int head = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int4 desc = blk_desc[b];
  const int qblk0 = desc.y;
  const int t0 = desc.z;
  const int L = desc.w;
  const int q0 = SKV ? qblk0 : qblk0 + wave * 32;
  const bool wave_active = q0 < L;
#ifdef LTR_ATTN_TIMELINE
  if (L > 0) tl1 = __builtin_readcyclecounter();       // the descriptor has arrived
#endif
  const size_t ld = (size_t)3 * H;
  const int lq = lane & 31, lh = lane >> 5;

  // Q fragments (B operand: column = query lq, k-slots = d ks*16 + 8*lh .. +7), kept in registers
  f16x8 qh[4], ql[4];
  {
    const size_t qrow = (size_t)(t0 + min(q0 + lq, L - 1)) * ld + head * D + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = *reinterpret_cast<const f16x8*>(qkv_hi + qrow + ks * 16);
      if (!ONEP) ql[ks] = *reinterpret_cast<const f16x8*>(qkv_lo + qrow + ks * 16);
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m = NEG, l = 0.f;

  // tile loads: NW = 4: wave w fills rows 8w..8w+7 of each of the four planes (1 KiB each); NW = 8: rows 8(w & 3).. of the
  // two K planes (w < 4) or of the two V planes (w >= 4)
  const int lw = wave & 3;
  const bool load_k = NW == 4 || wave < 4, load_v = NW == 4 || wave >= 4;      // wave-uniform
  const int lrow = lw * 8 + (lane >> 3);                        // key row inside the tile
  const int kc_log = (lane & 7) ^ ((lrow >> 1) & 7);            // K: swizzled source chunk
  const int vc = (lane & 7) ^ (((lrow >> 1) & 1) * LTR_ATTN_VSWZ);   // V: chunk bits flipped on odd row pairs (v_off)
  auto issue = [&](int stage, int kt) {
    const size_t rowoff = (size_t)(t0 + min(kt + lrow, L - 1)) * ld + head * D;
    __half* base = smem + stage * ATT_STAGE + lw * 8 * D;
    if (load_k) {
      __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_hi + rowoff + H + kc_log * 8), (lds_void*)(base), 16, 0, 0);
      if (!ONEP) __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_lo + rowoff + H + kc_log * 8), (lds_void*)(base + PLANE_H), 16, 0, 0);
    }
    if (load_v) {
      __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_hi + rowoff + 2 * H + vc * 8), (lds_void*)(base + 2 * PLANE_H), 16, 0, 0);
      if (!ONEP) __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_lo + rowoff + 2 * H + vc * 8), (lds_void*)(base + 3 * PLANE_H), 16, 0, 0);
    }
  };

  const int kend = min(L, qblk0 + QBLK);      // keys needed by this block: [0, kend)
  const int ntile = (kend + TK - 1) / TK;
  // Ring of three stages: when tile `it` is multiplied, tiles it+1 and it+2 are in flight
  // (prompts are short, so a wave's per-tile math is shorter than the HBM round trip; one
  // tile of look-ahead left the waves parked on vmcnt).  Counted waits: 4 loads per tile.
  // SKV: this wave's private stream - tile j of the wave = tile (wave + NW j) of the block, its own two-stage ring
  auto issue_own = [&](int stage, int kt) {
    __half* base = smem + (wave * 2 + stage) * ATT_STAGE;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r * 8 + (lane >> 3);
      const int kc = (lane & 7) ^ ((row >> 1) & 7), vcc = (lane & 7) ^ (((row >> 1) & 1) * LTR_ATTN_VSWZ);
      const size_t rowoff = (size_t)(t0 + min(kt + row, L - 1)) * ld + head * D;
      __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_hi + rowoff + H + kc * 8), (lds_void*)(base + r * 8 * D), 16, 0, 0);
      if (!ONEP) __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_lo + rowoff + H + kc * 8), (lds_void*)(base + PLANE_H + r * 8 * D), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_hi + rowoff + 2 * H + vcc * 8), (lds_void*)(base + 2 * PLANE_H + r * 8 * D), 16, 0, 0);
      if (!ONEP) __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_lo + rowoff + 2 * H + vcc * 8), (lds_void*)(base + 3 * PLANE_H + r * 8 * D), 16, 0, 0);
    }
  };
  const int nown = SKV ? (ntile > wave ? (ntile - wave + NW - 1) / NW : 0) : 0;      // tiles of this wave's stream
  if (SKV) {
    if (nown > 0) issue_own(0, wave * TK);
    if (nown > 1) issue_own(1, (wave + NW) * TK);
  } else {
    issue(0, 0);
    if (ntile > 1) issue(1, TK);
  }
  // Pin the Q fragments here: hipcc then waits for the (older) Q loads with a counted vmcnt BEFORE the
  // loop; left to itself it re-waits vmcnt(0) at their first use in every iteration, which also drains
  // the look-ahead tiles.
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { if (ONEP) asm volatile("" ::"v"(qh[ks])); else asm volatile("" ::"v"(qh[ks]), "v"(ql[ks])); }
  for (int it = 0; it < (SKV ? nown : ntile); ++it) {
    const int kt = SKV ? (wave + NW * it) * TK : it * TK;
    const __half* s_khi;
    if (SKV) {
      // my tile `it` has landed once only my NEXT tile's 16 loads are outstanding; no workgroup barrier: nobody else
      // touches my ring
      if (it + 1 < nown) { if (ONEP) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      s_khi = smem + (wave * 2 + (it & 1)) * ATT_STAGE;
    } else {
      if (it + 1 < ntile) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");   // tile it landed, it+1 may fly
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // raw barrier: __syncthreads() makes hipcc drain vmcnt(0) first, i.e. wait for the look-ahead tiles too
      __builtin_amdgcn_s_barrier();                       // everyone's rows landed; tile it-1 fully consumed
#ifdef LTR_ATTN_TIMELINE
      if (it == 0) tl2 = __builtin_readcyclecounter();
#endif
      if (it + 2 < ntile) issue((it + 2) % NSTAGE, kt + 2 * TK);
      if (!wave_active || kt > q0) continue;              // wave-uniform: past this wave's diagonal
      s_khi = smem + (it % NSTAGE) * ATT_STAGE;
    }
    const __half* s_klo = s_khi + PLANE_H;
    const __half* s_vhi = s_khi + 2 * PLANE_H;
    const __half* s_vlo = s_khi + 3 * PLANE_H;

    // ---- S^T = K Q^T (rows = keys, cols = queries)
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + lh;
      const f16x8 kh = *reinterpret_cast<const f16x8*>(s_khi + k_off(lq, c));
      if (!ONEP) {
        const f16x8 kl = *reinterpret_cast<const f16x8*>(s_klo + k_off(lq, c));
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], sacc, 0, 0, 0);
      }
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], sacc, 0, 0, 0);
    }
    // ---- online softmax for query lq (this lane: 16 of its 32 keys; lane^32: the others).
    // Scores stay raw in the accumulator; p = exp2(s * c - m) is one fma + one v_exp_f32.
    // The running reference m is only raised when some query of the wave sees a score more
    // than 8 (log2 units) above it, so in most tiles the O accumulators are not touched by the
    // VALU at all (p <= 2^8 is exact work for the fp16 hi|lo split and the f32 row sum).
    if (kt == q0) {                                             // diagonal tile: causal mask
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if ((i & 3) + 8 * (i >> 2) + 4 * lh > lq) sacc[i] = NEG;   // key (relative to kt) > query
    }
    float mx = sacc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sacc[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;      // scale > 0 commutes with max
    if (__any(mx > m + 8.0f)) {                                  // wave-uniform
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);     // raw v_exp_f32: argument <= 0, underflow to 0 is wanted
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
    }
    float psum = 0.f;
    f16x8 ph0, ph1, pl0, pl1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[i], scale_log2e, -m));
      const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[8 + i], scale_log2e, -m));
      psum += p0 + p1;
      const _Float16 h0 = (_Float16)p0, h1 = (_Float16)p1;
      ph0[i] = h0; ph1[i] = h1;
      if (!ONEP) {
        pl0[i] = (_Float16)(p0 - (float)h0);
        pl1[i] = (_Float16)(p1 - (float)h1);
      }
    }
    l += psum;
    // ---- O^T += V^T P^T  (rows = d, cols = queries).  A fragment of k-step g: element e of
    // lane-half lh is key 16g + 8(e>>2) + 4lh + (e&3) - the accumulator register order of S^T.
    // The V tile is row-major [key][d]; ds_read_b64_tr_b16 transposes a [4 keys][16 d] block inside
    // every 16-lane group (lane j supplies the address of key j>>2, d-chunk j&3 and receives column
    // j of the four keys), so one read yields e = 0..3 (or 4..7) of this lane's d = dt*32 + lq.
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        f16x8 vh, vl;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int key = 16 * g + 8 * hf + 4 * lh + ((lane & 15) >> 2);
          const int off = v_off(key, dt * 32 + (lane & 16) + (lane & 3) * 4);
          const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s_vhi + off));
          const f16x4 ah = __builtin_bit_cast(f16x4, a);
#pragma unroll
          for (int e = 0; e < 4; ++e) vh[4 * hf + e] = ah[e];
          if (!ONEP) {
            const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s_vlo + off));
            const f16x4 bl = __builtin_bit_cast(f16x4, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) vl[4 * hf + e] = bl[e];
          }
        }
        const f16x8 ph = g ? ph1 : ph0;
        if (!ONEP) {
          const f16x8 pl = g ? pl1 : pl0;
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[dt], 0, 0, 0);
        }
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[dt], 0, 0, 0);
      }
    }
  
    if (SKV) {
      // the stage just consumed is refilled with my tile it + 2 (the reads of this iteration are complete: the P V MFMAs
      // above depend on them)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (it + 2 < nown) issue_own(it & 1, (wave + NW * (it + 2)) * TK);
    }
  }
  // ---- output.  Lane (lq, lh) holds 4 consecutive d of query lq per (dt, j): stored directly that
  // is 32 rows x 16 B per instruction (every 128-B line written in eight pieces).  Transpose through
  // the (now idle) ring instead: per wave a [32 queries][64 d] tile per fp16 plane, row pitch 136 B
  // (conflict-free 8-byte accesses), read back so that 16 lanes cover one 128-B row.
  __syncthreads();                                      // every wave is done with the last K/V tile
#ifdef LTR_ATTN_TIMELINE
  const unsigned long long tl3 = __builtin_readcyclecounter();
#endif
  if (!wave_active) return;
  if (SKV) {
    // merge the waves' online-softmax states into wave 0 (behind its output staging area): per lane 32 accumulators + (m, l)
    constexpr int MERGE0 = 2 * 32 * 136;                // bytes: wave 0's [32 queries][64 d] hi | lo staging tile
    constexpr int MSTRIDE = 34 * 64 * 4;                // bytes per wave: 34 floats per lane
    float* mg = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + MERGE0 + (wave - 1) * MSTRIDE);
    if (wave != 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { mg[i * 64 + lane] = o[0][i]; mg[(16 + i) * 64 + lane] = o[1][i]; }
      mg[32 * 64 + lane] = m; mg[33 * 64 + lane] = l;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float* src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + MERGE0 + (w - 1) * MSTRIDE);
      const float mw = src[32 * 64 + lane], lw = src[33 * 64 + lane];
      const float m_new = fmaxf(m, mw);
      const float a = __builtin_amdgcn_exp2f(m - m_new), b = __builtin_amdgcn_exp2f(mw - m_new);   // (a wave without tiles: m = NEG, b = 0)
      m = m_new;
      l = l * a + lw * b;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[0][i] = o[0][i] * a + src[i * 64 + lane] * b;
        o[1][i] = o[1][i] * a + src[(16 + i) * 64 + lane] * b;
      }
    }
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  // training: the backward recomputes the softmax from the log-sum-exp of the row (log2 domain, like the f32 kernel;
  // m may lag the true maximum by the deferral threshold, m + log2(l) does not)
  if (lse2 != nullptr && lh == 0 && q0 + lq < L) lse2[(size_t)(t0 + q0 + lq) * (H / D) + head] = m + log2f(l);
  constexpr int OPITCH = 136;                           // bytes
  char* s_o = reinterpret_cast<char*>(smem) + wave * (2 * 32 * OPITCH);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __half hh[4], ll[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) split_f16(o[dt][4 * j + i] * inv, hh[i], ll[i]);
      const int d = dt * 32 + 8 * j + 4 * lh;
      *reinterpret_cast<uint2*>(s_o + lq * OPITCH + d * 2) = *reinterpret_cast<const uint2*>(hh);
      *reinterpret_cast<uint2*>(s_o + 32 * OPITCH + lq * OPITCH + d * 2) = *reinterpret_cast<const uint2*>(ll);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int orow = lane >> 4, opiece = lane & 15;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = p * 4 + orow;
    const uint2 vh = *reinterpret_cast<const uint2*>(s_o + row * OPITCH + opiece * 8);
    const uint2 vl = *reinterpret_cast<const uint2*>(s_o + 32 * OPITCH + row * OPITCH + opiece * 8);
    if (q0 + row < L) {
      const size_t ob = (size_t)(t0 + q0 + row) * H + head * D + opiece * 4;
      *reinterpret_cast<uint2*>(out_hi + ob) = vh;
      if (!ONEP) *reinterpret_cast<uint2*>(out_lo + ob) = vl;
    }
  }
#ifdef LTR_ATTN_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int slot = blockIdx.y * gridDim.x + blockIdx.x;
  if (tid == 0 && slot < 4096) {
    g_attn_tl[slot * 5 + 0] = tl0; g_attn_tl[slot * 5 + 1] = tl1; g_attn_tl[slot * 5 + 2] = tl2; g_attn_tl[slot * 5 + 3] = tl3;
    g_attn_tl[slot * 5 + 4] = __builtin_readcyclecounter();
  }
#endif
}


// Last layer of a scoring call: only the LAST token of each prompt is pooled (logits_processor.py:74-79), so of
// the last layer's attention only the last query of every request is needed - it reads all the request's keys
// and values (the K | V GEMM still runs over every token), but Q, the softmax and the output exist for n_req rows
// instead of T.  That is a matrix-vector problem per (request, head): ~4 B of K/V per 2 FLOP, HBM-bound, so it
// runs on the VALU in f32 straight from the hi|lo planes (k = hi + lo is exact in f32).
// Workgroup = (request, head), 4 waves; a wave instruction covers 8 keys x 64 dims (lane = key j8 x 16-byte
// piece c), groups of 8 keys are dealt round-robin to the waves, every lane runs its own online softmax over
// "its" keys for "its" 8 dims, and the (m, l, o) states are merged at the end (lanes, then waves through LDS).
constexpr int LQ_WAVES = 4;
__global__ void __launch_bounds__(LQ_WAVES * 64) attn_lastq_kernel(const float* __restrict__ q /*[n_req, H]*/,
                                                                   const __half* __restrict__ kv_hi,
                                                                   const __half* __restrict__ kv_lo /*[T, 2H]: k | v*/,
                                                                   const int32_t* __restrict__ cu, int H,
                                                                   float scale_log2e, __half* __restrict__ out_hi,
                                                                   __half* __restrict__ out_lo /*[n_req, H]*/) {
  __shared__ float s_o[LQ_WAVES][D];
  __shared__ float s_ml[LQ_WAVES][2];
  const int req = blockIdx.x, head = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 7, j8 = lane >> 3;
  const int c0 = cu[req], L = cu[req + 1] - c0;
  if (L <= 0) return;
  const size_t ld = (size_t)2 * H;
  const size_t base = (size_t)(c0 - cu[0]) * ld + head * D + c * 8;
  float qv[8];
  {
    const float* qp = q + (size_t)req * H + head * D + c * 8;
    const float4 a = *reinterpret_cast<const float4*>(qp), b = *reinterpret_cast<const float4*>(qp + 4);
    qv[0] = a.x * scale_log2e; qv[1] = a.y * scale_log2e; qv[2] = a.z * scale_log2e; qv[3] = a.w * scale_log2e;
    qv[4] = b.x * scale_log2e; qv[5] = b.y * scale_log2e; qv[6] = b.z * scale_log2e; qv[7] = b.w * scale_log2e;
  }
  float m = NEG, l = 0.f, o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0.f;
  const int ngroups = (L + 7) >> 3;
  auto widen = [](const uint4& hi, const uint4& lo, float* x) {
    const __half2* h2 = reinterpret_cast<const __half2*>(&hi);
    const __half2* l2 = reinterpret_cast<const __half2*>(&lo);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 a = __half22float2(h2[i]), b = __half22float2(l2[i]);
      x[2 * i] = a.x + b.x; x[2 * i + 1] = a.y + b.y;
    }
  };
  auto step = [&](bool valid, const uint4& kh, const uint4& kl, const uint4& vh, const uint4& vl) {
    float k[8], v[8];
    widen(kh, kl, k);
    widen(vh, vl, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s = fmaf(qv[i], k[i], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (!valid) return;
    const float mn = fmaxf(m, s);
    const float alpha = exp2f(m - mn), p = exp2f(s - mn);
    l = l * alpha + p;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaf(p, v[i], o[i] * alpha);
    m = mn;
  };
  // two groups of 8 keys per iteration: eight 16-byte loads in flight per lane
  for (int g = wave; g < ngroups; g += 2 * LQ_WAVES) {
    const int ja = g * 8 + j8, jb = (g + LQ_WAVES) * 8 + j8;
    const bool va = ja < L, vb = jb < L;
    const size_t oa = base + (size_t)(va ? ja : 0) * ld, ob = base + (size_t)(vb ? jb : 0) * ld;
    const uint4 kha = *reinterpret_cast<const uint4*>(kv_hi + oa), kla = *reinterpret_cast<const uint4*>(kv_lo + oa);
    const uint4 vha = *reinterpret_cast<const uint4*>(kv_hi + oa + H), vla = *reinterpret_cast<const uint4*>(kv_lo + oa + H);
    const uint4 khb = *reinterpret_cast<const uint4*>(kv_hi + ob), klb = *reinterpret_cast<const uint4*>(kv_lo + ob);
    const uint4 vhb = *reinterpret_cast<const uint4*>(kv_hi + ob + H), vlb = *reinterpret_cast<const uint4*>(kv_lo + ob + H);
    step(va, kha, kla, vha, vla);
    step(vb, khb, klb, vhb, vlb);
  }
  // merge the 8 key lanes of the wave (lanes that differ in j8 hold the same dims)
#pragma unroll
  for (int x = 8; x < 64; x <<= 1) {
    const float m2 = __shfl_xor(m, x, 64), l2 = __shfl_xor(l, x, 64);
    const float mn = fmaxf(m, m2);
    const float a = exp2f(m - mn), b = exp2f(m2 - mn);
    l = l * a + l2 * b;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = o[i] * a + __shfl_xor(o[i], x, 64) * b;
    m = mn;
  }
  if (j8 == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s_o[wave][c * 8 + i] = o[i];
    if (c == 0) { s_ml[wave][0] = m; s_ml[wave][1] = l; }
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    float mt = NEG;
#pragma unroll
    for (int w = 0; w < LQ_WAVES; ++w) mt = fmaxf(mt, s_ml[w][0]);
    float lt = 0.f, ot[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ot[i] = 0.f;
#pragma unroll
    for (int w = 0; w < LQ_WAVES; ++w) {
      const float a = exp2f(s_ml[w][0] - mt);
      lt += s_ml[w][1] * a;
#pragma unroll
      for (int i = 0; i < 8; ++i) ot[i] += s_o[w][c * 8 + i] * a;
    }
    const float inv = 1.0f / lt;
    __half hh[8], ll[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split_f16(ot[i] * inv, hh[i], ll[i]);
    const size_t ob = (size_t)req * H + head * D + c * 8;
    *reinterpret_cast<uint4*>(out_hi + ob) = *reinterpret_cast<const uint4*>(hh);
    *reinterpret_cast<uint4*>(out_lo + ob) = *reinterpret_cast<const uint4*>(ll);
  }
}


// ------------------------------------------------------------------------------------
// Training: attention BACKWARD on the forward kernel's machinery (the reference gets it from autograd
// through vllm's attention replaced by HF's eager attention in train/trainer.py).  Forward: S2 = (q k) scale log2e,
// P = exp2(S2 - lse2), O = P V.  With D_q = sum_d dO O:
//     dV = P^T dO          dP = dO V^T          dS = P (dP - D_q)          dQ = scale dS K          dK = scale dS^T Q
// Two kernels over the forward's work list of 128-row blocks; both keep one side's fragments in registers and stream
// 32-row tiles of the other side through a 3-stage LDS ring exactly like the forward (global_load_lds, counted vmcnt):
//   dq  kernel: block = 128 queries; resident B fragments Q, dO; streamed K (twice: row layout for S^T = K Q^T and
//               transpose-read layout for dQ^T += K^T dS^T) and V (row layout for dP^T = V dO^T).
//   dkv kernel: block = 128 keys; resident B fragments K, V; streamed Q and dO, each in both layouts:
//               S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.
// In both, the accumulator registers of the "score" products are, converted to fp16 hi | lo, directly the B operand of
// the accumulating products (the forward's P trick), and the transposed A operands come from ds_read_b64_tr_b16.
// Every product is three split passes (lo hi + hi lo + hi hi), f32 accumulate.  dO arrives scaled by a power of two
// s_O (gradients are far below fp16's range) chosen from max|dO| and max|qkv| so that |dP|, |D| <= 2^14 and |dS| <= 2^15
// stay inside fp16; everything downstream is linear in dO and the outputs are divided by s_O again (exact).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float attn_bwd_scale(float amax_do, float amax_qkv) {
  if (!(amax_do > 0.f) || !(amax_do < INFINITY)) return 1.f;
  const int eq = (amax_qkv > 0.f && amax_qkv < INFINITY) ? ilogbf(amax_qkv) + 1 : 0;
  return ldexpf(1.f, 14 - 6 - (ilogbf(amax_do) + 1) - max(eq, 0));
}

// 32 x 32 product block: rows = the 32 rows of an LDS tile in the row layout (k_off), columns = the resident fragments
__device__ __forceinline__ f32x16 bwd_score(const __half* __restrict__ s_hi, const __half* __restrict__ s_lo,
                                            const f16x8 (&bh)[4], const f16x8 (&bl)[4], int lq, int lh) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c = 2 * ks + lh;
    const f16x8 ah = *reinterpret_cast<const f16x8*>(s_hi + k_off(lq, c));
    const f16x8 al = *reinterpret_cast<const f16x8*>(s_lo + k_off(lq, c));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc, 0, 0, 0);
  }
  return acc;
}
// the 16 accumulator values of a lane (its column, 16 of the 32 tile rows) -> fp16 hi | lo B fragments of the two k-steps
__device__ __forceinline__ void bwd_split(const f32x16& x, f16x8& h0, f16x8& h1, f16x8& l0, f16x8& l1) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 a = (_Float16)x[i], b = (_Float16)x[8 + i];
    h0[i] = a; h1[i] = b;
    l0[i] = (_Float16)(x[i] - (float)a);
    l1[i] = (_Float16)(x[8 + i] - (float)b);
  }
}
// o[dt] (rows = d, cols = the lane's column) += T^T X with T the 32 x 64 LDS tile in the transpose-read layout (v_off) and
// X the split accumulator fragments (tile rows in accumulator register order) - the forward's O^T += V^T P^T
__device__ __forceinline__ void bwd_accum_T(f32x16 (&o)[2], const __half* __restrict__ t_hi, const __half* __restrict__ t_lo,
                                            const f16x8& h0, const f16x8& h1, const f16x8& l0, const f16x8& l1, int lane, int lh) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f16x8 th, tl;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int row = 16 * g + 8 * hf + 4 * lh + ((lane & 15) >> 2);
        const int off = v_off(row, dt * 32 + (lane & 16) + (lane & 3) * 4);
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(t_hi + off));
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(t_lo + off));
        const f16x4 ah = __builtin_bit_cast(f16x4, a), bl = __builtin_bit_cast(f16x4, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) { th[4 * hf + e] = ah[e]; tl[4 * hf + e] = bl[e]; }
      }
      const f16x8 xh = g ? h1 : h0, xl = g ? l1 : l0;
      o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl, xh, o[dt], 0, 0, 0);
      o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, xl, o[dt], 0, 0, 0);
      o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, xh, o[dt], 0, 0, 0);
    }
  }
}
// accumulator (rows d = dt * 32 + 8 j + 4 lh + i, column = this lane's row of the block) -> f32 row of dqkv
__device__ __forceinline__ void bwd_store(const f32x16 (&o)[2], float f, float* __restrict__ dst /*row + column base*/, int lh) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(dst + dt * 32 + 8 * j + 4 * lh) =
          make_float4(o[dt][4 * j] * f, o[dt][4 * j + 1] * f, o[dt][4 * j + 2] * f, o[dt][4 * j + 3] * f);
}

constexpr int BWD_DQ_PLANES = 6;    // K rows hi|lo, K transposable hi|lo, V rows hi|lo
constexpr int BWD_DKV_PLANES = 8;   // Q rows hi|lo, Q transposable hi|lo, dO rows hi|lo, dO transposable hi|lo
constexpr int BWD_STAT_H = 4 * 128; // halves: per wave 64 floats (lse2 | D of the tile's 32 queries), dkv kernel

__global__ void __launch_bounds__(256, 2) attn_bwd_dq_f16s_kernel(
    const __half* __restrict__ qkv_hi, const __half* __restrict__ qkv_lo, const __half* __restrict__ do_hi,
    const __half* __restrict__ do_lo, const float* __restrict__ lse2, const float* __restrict__ Dq,
    const float* __restrict__ amax_do, const float* __restrict__ amax_qkv, const int32_t* __restrict__ blk_start,
    const int4* __restrict__ blk_desc, int n_req, int H, float scale, float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) __half smem[];
  constexpr int STAGE_H = BWD_DQ_PLANES * PLANE_H;
  const int b = blockIdx.x;
  if (b >= blk_start[n_req]) return;
  const int head = blockIdx.y, nh = H / D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int4 desc = blk_desc[b];
  const int qblk0 = desc.y, t0 = desc.z, L = desc.w;
  const int q0 = qblk0 + wave * 32;
  const bool wave_active = q0 < L;
  const size_t ld = (size_t)3 * H;
  const int lq = lane & 31, lh = lane >> 5;
  const float s_o = attn_bwd_scale(*amax_do, *amax_qkv);
  const float sl2e = scale * 1.4426950408889634f;

  f16x8 qh[4], ql[4], doh[4], dol[4];
  float lse, dsum;
  {
    const size_t row = (size_t)(t0 + min(q0 + lq, L - 1));
    const size_t qo = row * ld + head * D + 8 * lh, oo = row * H + head * D + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = *reinterpret_cast<const f16x8*>(qkv_hi + qo + ks * 16);
      ql[ks] = *reinterpret_cast<const f16x8*>(qkv_lo + qo + ks * 16);
      doh[ks] = *reinterpret_cast<const f16x8*>(do_hi + oo + ks * 16);
      dol[ks] = *reinterpret_cast<const f16x8*>(do_lo + oo + ks * 16);
    }
    lse = lse2[row * nh + head];
    dsum = Dq[row * nh + head] * s_o;
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }

  const int lrow = wave * 8 + (lane >> 3);
  const int kc_log = (lane & 7) ^ ((lrow >> 1) & 7);
  const int vc = (lane & 7) ^ (((lrow >> 1) & 1) * LTR_ATTN_VSWZ);
  auto issue = [&](int stage, int kt) {
    const size_t rowoff = (size_t)(t0 + min(kt + lrow, L - 1)) * ld + head * D;
    __half* base = smem + stage * STAGE_H + wave * 8 * D;
    __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_hi + rowoff + H + kc_log * 8), (lds_void*)(base), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_lo + rowoff + H + kc_log * 8), (lds_void*)(base + PLANE_H), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_hi + rowoff + H + vc * 8), (lds_void*)(base + 2 * PLANE_H), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_lo + rowoff + H + vc * 8), (lds_void*)(base + 3 * PLANE_H), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_hi + rowoff + 2 * H + kc_log * 8), (lds_void*)(base + 4 * PLANE_H), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_lo + rowoff + 2 * H + kc_log * 8), (lds_void*)(base + 5 * PLANE_H), 16, 0, 0);
  };
  const int kend = min(L, qblk0 + 128);
  const int ntile = (kend + TK - 1) / TK;
  issue(0, 0);
  if (ntile > 1) issue(1, TK);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qh[ks]), "v"(ql[ks]), "v"(doh[ks]), "v"(dol[ks]));
  asm volatile("" ::"v"(lse), "v"(dsum));
  for (int it = 0; it < ntile; ++it) {
    const int kt = it * TK;
    if (it + 1 < ntile) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 2 < ntile) issue((it + 2) % NSTAGE, kt + 2 * TK);
    if (!wave_active || kt > q0) continue;
    const __half* st = smem + (it % NSTAGE) * STAGE_H;
    f32x16 sacc = bwd_score(st, st + PLANE_H, qh, ql, lq, lh);                        // S^T  = K Q^T
    const f32x16 dpacc = bwd_score(st + 4 * PLANE_H, st + 5 * PLANE_H, doh, dol, lq, lh);   // dP^T = V dO^T  (x s_O)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kr = (i & 3) + 8 * (i >> 2) + 4 * lh;                                 // key row inside the tile
      const float p = (kt == q0 && kr > lq) ? 0.f : __builtin_amdgcn_exp2f(fmaf(sacc[i], sl2e, -lse));
      sacc[i] = p * (dpacc[i] - dsum);                                                // dS^T (x s_O)
    }
    f16x8 h0, h1, l0, l1;
    bwd_split(sacc, h0, h1, l0, l1);
    bwd_accum_T(o, st + 2 * PLANE_H, st + 3 * PLANE_H, h0, h1, l0, l1, lane, lh);     // dQ^T += K^T dS^T
  }
  if (!wave_active || q0 + lq >= L) return;
  bwd_store(o, scale / s_o, dqkv + (size_t)(t0 + q0 + lq) * ld + head * D, lh);
}

__global__ void __launch_bounds__(256, 2) attn_bwd_dkv_f16s_kernel(
    const __half* __restrict__ qkv_hi, const __half* __restrict__ qkv_lo, const __half* __restrict__ do_hi,
    const __half* __restrict__ do_lo, const float* __restrict__ lse2, const float* __restrict__ Dq,
    const float* __restrict__ amax_do, const float* __restrict__ amax_qkv, const int32_t* __restrict__ blk_start,
    const int4* __restrict__ blk_desc, int n_req, int H, float scale, float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) __half smem[];
  constexpr int STAGE_H = BWD_DKV_PLANES * PLANE_H + BWD_STAT_H;
  const int b = blockIdx.x;
  if (b >= blk_start[n_req]) return;
  const int head = blockIdx.y, nh = H / D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int4 desc = blk_desc[b];
  const int kblk0 = desc.y, t0 = desc.z, L = desc.w;
  const int k0 = kblk0 + wave * 32;
  const bool wave_active = k0 < L;
  const size_t ld = (size_t)3 * H;
  const int lq = lane & 31, lh = lane >> 5;
  const float s_o = attn_bwd_scale(*amax_do, *amax_qkv);
  const float sl2e = scale * 1.4426950408889634f;

  f16x8 kh[4], kl[4], vh[4], vl[4];
  {
    const size_t ko = (size_t)(t0 + min(k0 + lq, L - 1)) * ld + H + head * D + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kh[ks] = *reinterpret_cast<const f16x8*>(qkv_hi + ko + ks * 16);
      kl[ks] = *reinterpret_cast<const f16x8*>(qkv_lo + ko + ks * 16);
      vh[ks] = *reinterpret_cast<const f16x8*>(qkv_hi + ko + H + ks * 16);
      vl[ks] = *reinterpret_cast<const f16x8*>(qkv_lo + ko + H + ks * 16);
    }
  }
  f32x16 ov[2], ok[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { ov[0][i] = 0.f; ov[1][i] = 0.f; ok[0][i] = 0.f; ok[1][i] = 0.f; }

  const int lrow = wave * 8 + (lane >> 3);
  const int kc_log = (lane & 7) ^ ((lrow >> 1) & 7);
  const int vc = (lane & 7) ^ (((lrow >> 1) & 1) * LTR_ATTN_VSWZ);
  auto issue = [&](int stage, int qt) {
    const size_t row = (size_t)(t0 + min(qt + lrow, L - 1));
    const size_t qo = row * ld + head * D, oo = row * H + head * D;
    __half* base = smem + stage * STAGE_H + wave * 8 * D;
    __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_hi + qo + kc_log * 8), (lds_void*)(base), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_lo + qo + kc_log * 8), (lds_void*)(base + PLANE_H), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_hi + qo + vc * 8), (lds_void*)(base + 2 * PLANE_H), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(qkv_lo + qo + vc * 8), (lds_void*)(base + 3 * PLANE_H), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(do_hi + oo + kc_log * 8), (lds_void*)(base + 4 * PLANE_H), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(do_lo + oo + kc_log * 8), (lds_void*)(base + 5 * PLANE_H), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(do_hi + oo + vc * 8), (lds_void*)(base + 6 * PLANE_H), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(do_lo + oo + vc * 8), (lds_void*)(base + 7 * PLANE_H), 16, 0, 0);
    // the tile's 32 (lse2, D) pairs, one private copy per wave (4 bytes per lane: lanes 0-31 lse2, 32-63 D)
    const size_t srow = (size_t)(t0 + min(qt + lq, L - 1)) * nh + head;
    float* sbase = reinterpret_cast<float*>(smem + stage * STAGE_H + BWD_DKV_PLANES * PLANE_H) + wave * 64;
    __builtin_amdgcn_global_load_lds((gbl_void*)((lh ? Dq : lse2) + srow), (lds_void*)(sbase), 4, 0, 0);
  };
  // queries that see this block's keys: [kblk0, L), in tiles of 32 (kblk0 is a multiple of 128)
  const int ntile = (L - kblk0 + TK - 1) / TK;
  issue(0, kblk0);
  if (ntile > 1) issue(1, kblk0 + TK);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(kh[ks]), "v"(kl[ks]), "v"(vh[ks]), "v"(vl[ks]));
  for (int it = 0; it < ntile; ++it) {
    const int qt = kblk0 + it * TK;
    if (it + 1 < ntile) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 2 < ntile) issue((it + 2) % NSTAGE, qt + 2 * TK);
    if (!wave_active || qt + TK - 1 < k0) continue;            // every query of the tile precedes this wave's keys
    const __half* st = smem + (it % NSTAGE) * STAGE_H;
    const float* s_stat = reinterpret_cast<const float*>(st + BWD_DKV_PLANES * PLANE_H) + wave * 64;
    f32x16 sacc = bwd_score(st, st + PLANE_H, kh, kl, lq, lh);                          // S  = Q K^T   (rows = queries)
    f32x16 dpacc = bwd_score(st + 4 * PLANE_H, st + 5 * PLANE_H, vh, vl, lq, lh);       // dP = dO V^T  (x s_O)
    const int key = k0 + lq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 ls = *reinterpret_cast<const float4*>(s_stat + 8 * j + 4 * lh);
      const float4 dd = *reinterpret_cast<const float4*>(s_stat + 32 + 8 * j + 4 * lh);
      const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, ddv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * j + e;
        const int q = qt + 8 * j + 4 * lh + e;
        const float p = (q < key || q >= L) ? 0.f : __builtin_amdgcn_exp2f(fmaf(sacc[i], sl2e, -lsv[e]));
        sacc[i] = p;                                                                     // P
        dpacc[i] = p * (dpacc[i] - ddv[e] * s_o);                                        // dS (x s_O)
      }
    }
    f16x8 h0, h1, l0, l1;
    bwd_split(sacc, h0, h1, l0, l1);
    bwd_accum_T(ov, st + 6 * PLANE_H, st + 7 * PLANE_H, h0, h1, l0, l1, lane, lh);      // dV^T += dO^T P
    bwd_split(dpacc, h0, h1, l0, l1);
    bwd_accum_T(ok, st + 2 * PLANE_H, st + 3 * PLANE_H, h0, h1, l0, l1, lane, lh);      // dK^T += Q^T dS
  }
  if (!wave_active || k0 + lq >= L) return;
  float* dst = dqkv + (size_t)(t0 + k0 + lq) * ld + head * D;
  bwd_store(ok, scale / s_o, dst + H, lh);
  bwd_store(ov, 1.f / s_o, dst + 2 * H, lh);
}

// D_q = sum_d dO O per (token, head), f32 (one wave per token row: lanes over the H columns)
__global__ void __launch_bounds__(256) attn_bwd_rowdot_kernel(const float* __restrict__ o, const float* __restrict__ dout,
                                                              int T, int H, float* __restrict__ Dq) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, nh = H / D;
  if (row >= T) return;
  // 16 lanes x float4 cover one head (64 columns): four heads per wave pass
  for (int h0 = 0; h0 < nh; h0 += 4) {
    const int head = h0 + (lane >> 4);
    float v = 0.f;
    if (head < nh) {
      const size_t off = (size_t)row * H + head * D + (lane & 15) * 4;
      const float4 a = *reinterpret_cast<const float4*>(o + off), b = *reinterpret_cast<const float4*>(dout + off);
      v = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    if (head < nh && (lane & 15) == 0) Dq[(size_t)row * nh + head] = v;
  }
}

// f32 [rows, cols] -> row-major fp16 hi | lo planes of x * s (s = 1, or the attention-backward scale of dO)
__global__ void __launch_bounds__(256) attn_bwd_planes_kernel(const float* __restrict__ x, size_t n8, const float* __restrict__ amax_do,
                                                              const float* __restrict__ amax_qkv, __half* __restrict__ hi,
                                                              __half* __restrict__ lo) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const float s = amax_do ? attn_bwd_scale(*amax_do, *amax_qkv) : 1.f;
  const float4 a = *reinterpret_cast<const float4*>(x + i * 8), b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
  const float v[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
  __half h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) split_f16(v[e], h[e], l[e]);
  *reinterpret_cast<uint4*>(hi + i * 8) = *reinterpret_cast<const uint4*>(h);
  *reinterpret_cast<uint4*>(lo + i * 8) = *reinterpret_cast<const uint4*>(l);
}

}  // namespace

// the work list of `qb`-query blocks alone (the training backward walks 64-query blocks whatever kernel ran the forward)
#ifdef LTR_ATTN_TIMELINE
int attn_timeline_read(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_tl), sizeof(unsigned long long) * 4096 * 5); }
#endif

int launch_attention_blocks(const int32_t* cu, int n_req, int qb, int32_t* blk_start, hipStream_t s) {
  if (n_req == 0) return LTR_OK;
  int4* blk_desc = reinterpret_cast<int4*>(blk_start + ((n_req + 1 + 3) & ~3));
  attn_blocks_kernel<<<1, 1024, 0, s>>>(cu, n_req, qb, blk_start, blk_desc);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

namespace {
// 256-query workgroups by default?  (lab rule, set from the measurements in profiles/r05_attn_wide.txt)
inline bool attn_wide_default(int T, int n_req) { (void)T; (void)n_req; return false; }
}  // namespace

int launch_attention(int wdtype, AOp qkv, const int32_t* cu, int n_req, int T, int H, int n_heads,
                     int32_t* blk_start, AOp out, int build_blocks, hipStream_t s, float* lse2, size_t blk_bytes, int one_pass) {
  if (n_req == 0 || T == 0) return LTR_OK;
  if (H != n_heads * D) { set_error("attention: head size must be 64 (H=%d heads=%d)", H, n_heads); return LTR_E_INVAL; }
  const float scale_log2e = 0.125f * 1.4426950408889634f;   // d^-0.5 (opt.py:73) * log2(e)
  // scratch layout: int32 blk_start[n_req + 1] (padded to 16 B) | int4 blk_desc[max blocks]
  int4* blk_desc = reinterpret_cast<int4*>(blk_start + ((n_req + 1 + 3) & ~3));
  if (wdtype == LTR_W_F16) {
    constexpr int NW = 4;
    // small passes (a scheduler step with a few arrivals): 32-query workgroups whose waves split the K/V tiles - when the
    // scratch holds the longer work list (T / 32 + n_req descriptors)
    // (read per call - a getenv costs nothing next to a launch - so that a test can move it; 0: never.  Measured: one-request
    // call 641 vs 676 us, no gain from ~900 tokens per pass)
    const char* skv_env = getenv("LTR_ATTN_SPLITKV_TOKENS");
    const int skv_tokens = skv_env ? atoi(skv_env) : 600;
    const bool skv = lse2 == nullptr && T <= skv_tokens &&
                     blk_bytes >= (size_t)((n_req + 1 + 3) & ~3) * 4 + ((size_t)T / 32 + n_req + 1) * 16;
    // 256-query workgroups (8 waves; the kernel's NW = 8 note).  LTR_ATTN_NW: 4 / 8 force one, unset = the measured rule
    const char* nw_s = getenv("LTR_ATTN_NW");                    // (read per call, like LTR_ATTN_SPLITKV_TOKENS: a test can move it)
    const int nw_env = nw_s ? atoi(nw_s) : 0;
    const bool wide = !skv && lse2 == nullptr && (nw_env == 8 || (nw_env == 0 && attn_wide_default(T, n_req)));
    const int qb = skv ? 32 : (wide ? 256 : 32 * NW);
    if (build_blocks) {   // the work list depends on cu_seqlens only: built once per pass, reused by every layer
      attn_blocks_kernel<<<1, 1024, 0, s>>>(cu, n_req, qb, blk_start, blk_desc);
      LTR_LAUNCH_CHECK();
    }
    dim3 grid(T / qb + n_req, n_heads);   // sum ceil(L/qb) <= floor(T/qb) + n_req
    if (one_pass && lse2 != nullptr) { set_error("attention: the one-pass variant has no training (lse2) form"); return LTR_E_INVAL; }
    if (skv) {
      constexpr int LDS = NW * 2 * ATT_STAGE * sizeof(__half);      // a private two-stage ring per wave: 128 KiB
      static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)attn_f16s_kernel<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess &&
               hipFuncSetAttribute((const void*)attn_f16s_kernel<NW, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
      }();
      (void)attr_ok;
      if (one_pass)
        attn_f16s_kernel<NW, true, true><<<grid, NW * 64, LDS, s>>>((const __half*)qkv.hi, (const __half*)qkv.lo, cu, blk_start, blk_desc,
                                                                    n_req, H, scale_log2e, (__half*)out.hi, (__half*)out.lo, lse2);
      else
        attn_f16s_kernel<NW, true><<<grid, NW * 64, LDS, s>>>((const __half*)qkv.hi, (const __half*)qkv.lo, cu, blk_start, blk_desc,
                                                              n_req, H, scale_log2e, (__half*)out.hi, (__half*)out.lo, lse2);
    } else if (wide) {
      constexpr int LDS8 = 8 * 2 * 32 * 136;                          // the output staging of eight waves (> the 48 KiB ring)
      static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)attn_f16s_kernel<8, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8) == hipSuccess &&
               hipFuncSetAttribute((const void*)attn_f16s_kernel<8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8) == hipSuccess;
      }();
      (void)attr_ok;
      if (one_pass)
        attn_f16s_kernel<8, false, true><<<grid, 512, LDS8, s>>>((const __half*)qkv.hi, (const __half*)qkv.lo, cu, blk_start, blk_desc, n_req, H,
                                                                 scale_log2e, (__half*)out.hi, (__half*)out.lo, lse2);
      else
        attn_f16s_kernel<8, false, false><<<grid, 512, LDS8, s>>>((const __half*)qkv.hi, (const __half*)qkv.lo, cu, blk_start, blk_desc, n_req, H,
                                                                  scale_log2e, (__half*)out.hi, (__half*)out.lo, lse2);
    } else if (one_pass) {
      attn_f16s_kernel<NW, false, true><<<grid, NW * 64, NSTAGE * ATT_STAGE * sizeof(__half), s>>>(
          (const __half*)qkv.hi, (const __half*)qkv.lo, cu, blk_start, blk_desc, n_req, H, scale_log2e, (__half*)out.hi, (__half*)out.lo, lse2);
    } else {
      attn_f16s_kernel<NW, false><<<grid, NW * 64, NSTAGE * ATT_STAGE * sizeof(__half), s>>>(
          (const __half*)qkv.hi, (const __half*)qkv.lo, cu, blk_start, blk_desc, n_req, H, scale_log2e, (__half*)out.hi, (__half*)out.lo, lse2);
    }
  } else {
    if (build_blocks) {
      attn_blocks_kernel<<<1, 1024, 0, s>>>(cu, n_req, QB, blk_start, blk_desc);
      LTR_LAUNCH_CHECK();
    }
    dim3 grid(T / QB + n_req, n_heads);
    if (wdtype == LTR_W_F32)
      attn_f32_kernel<false><<<grid, 64, 0, s>>>((const float*)qkv.hi, cu, blk_start, blk_desc, n_req, H, scale_log2e,
                                                 out.hi, out.lo, lse2);
    else   // debug A/B (wdtype -1): f32 VALU attention feeding split operands
      attn_f32_kernel<true><<<grid, 64, 0, s>>>((const float*)qkv.hi, cu, blk_start, blk_desc, n_req, H, scale_log2e,
                                                out.hi, out.lo, nullptr);
  }
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

int launch_attention_lastq(const float* q, AOp kv, const int32_t* cu, int n_req, int H, int n_heads, AOp out,
                           hipStream_t s) {
  if (n_req == 0) return LTR_OK;
  if (H != n_heads * D) { set_error("attention: head size must be 64 (H=%d heads=%d)", H, n_heads); return LTR_E_INVAL; }
  const float scale_log2e = 0.125f * 1.4426950408889634f;
  attn_lastq_kernel<<<dim3(n_req, n_heads), LQ_WAVES * 64, 0, s>>>(q, (const __half*)kv.hi, (const __half*)kv.lo, cu, H,
                                                                   scale_log2e, (__half*)out.hi, (__half*)out.lo);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

// f32 [n] -> row-major fp16 hi | lo planes (planes[0 .. n) = hi, planes[n .. 2n) = lo), unscaled
int launch_attention_bwd_planes(const float* x, size_t n, void* planes, hipStream_t s) {
  const size_t n8 = n / 8;
  attn_bwd_planes_kernel<<<(unsigned)((n8 + 255) / 256), 256, 0, s>>>(x, n8, nullptr, nullptr, (__half*)planes, (__half*)planes + n);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

// Attention backward of the training step (split-fp16 MFMA): dqkv [T, 3H] f32 from the saved qkv / attention output /
// log-sum-exp rows and the output gradient dout [T, H].  qkv_planes / do_planes: scratch for the fp16 hi | lo planes
// ([2][T, 3H] and [2][T, H] halves); amax_do / amax_qkv: device slots holding max|dout| and max|qkv|; blk_start: the
// forward's work list of 128-row blocks (launch_attention with F16 mode built it).
int launch_attention_bwd(const float* qkv, const float* o, const float* dout, const float* lse2, const float* amax_do,
                         const float* amax_qkv, const int32_t* blk_start, int n_req, int T, int H, int n_heads, float scale,
                         void* qkv_planes, void* do_planes, float* Dq, float* dqkv, hipStream_t s) {
  if (n_req == 0 || T == 0) return LTR_OK;
  if (H != n_heads * D) { set_error("attention backward: head size must be 64 (H=%d heads=%d)", H, n_heads); return LTR_E_INVAL; }
  const int4* blk_desc = reinterpret_cast<const int4*>(blk_start + ((n_req + 1 + 3) & ~3));
  __half* qh = (__half*)qkv_planes;
  __half* ql = qh + (size_t)T * 3 * H;
  __half* dh = (__half*)do_planes;
  __half* dl = dh + (size_t)T * H;
  const size_t nq8 = (size_t)T * 3 * H / 8, no8 = (size_t)T * H / 8;
  attn_bwd_planes_kernel<<<(unsigned)((nq8 + 255) / 256), 256, 0, s>>>(qkv, nq8, nullptr, nullptr, qh, ql);
  attn_bwd_planes_kernel<<<(unsigned)((no8 + 255) / 256), 256, 0, s>>>(dout, no8, amax_do, amax_qkv, dh, dl);
  attn_bwd_rowdot_kernel<<<(T + 3) / 4, 256, 0, s>>>(o, dout, T, H, Dq);
  LTR_LAUNCH_CHECK();
  static const bool attr_ok = [] {
    return hipFuncSetAttribute((const void*)attn_bwd_dkv_f16s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(NSTAGE * (BWD_DKV_PLANES * PLANE_H + BWD_STAT_H) * sizeof(__half))) == hipSuccess &&
           hipFuncSetAttribute((const void*)attn_bwd_dq_f16s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(NSTAGE * BWD_DQ_PLANES * PLANE_H * sizeof(__half))) == hipSuccess;
  }();
  (void)attr_ok;
  dim3 grid(T / 128 + n_req, n_heads);
  attn_bwd_dq_f16s_kernel<<<grid, 256, NSTAGE * BWD_DQ_PLANES * PLANE_H * sizeof(__half), s>>>(
      qh, ql, dh, dl, lse2, Dq, amax_do, amax_qkv, blk_start, blk_desc, n_req, H, scale, dqkv);
  attn_bwd_dkv_f16s_kernel<<<grid, 256, NSTAGE * (BWD_DKV_PLANES * PLANE_H + BWD_STAT_H) * sizeof(__half), s>>>(
      qh, ql, dh, dl, lse2, Dq, amax_do, amax_qkv, blk_start, blk_desc, n_req, H, scale, dqkv);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // namespace ltr
