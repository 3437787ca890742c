// Rank step of the scheduler queue on gfx950: starvation promote/demote, the stable
// (pri, -score, tiebreak) sort, post-schedule aging, and the budget-walk prefix.
//
// Reference semantics: vllm/core/scheduler.py:984-998 (promote/demote + sorted()),
// :1358-1365 (aging), :1137-1211 (budget walk `break` at first misfit).
//
// Sort design (integer/bit work, HBM/latency bound - no MFMA here):
//   key64 = [pri bit | order-preserving u32 image of -score | 31-bit tiebreak]
// Every key is unique (tiebreak is the input index by default), so the stable sort is a
// plain sort of unique 64-bit keys and rank(i) = #{j : key_j < key_i} is a permutation.
// For queue sizes on this path (1k..64k) a rank-by-counting sort fills all 256 CUs with
// independent work and needs no inter-workgroup hand-off: each workgroup owns 64 keys,
// streams the key array through LDS (broadcast ds_read_b64), and counts.
//
// Device-resident queue state: score / pri / idle / runs live in SLOT arrays that persist
// across scheduler steps; a step hands over `members` = the slot of every request in the order
// list(waiting)+list(running)+list(swapped) (scheduler.py:985,996).  Position i in `members` is the stable
// sort's tiebreak and what perm_out refers to; members == nullptr means slot == position.
//
// Launch count of a steady step (nothing new to score) at N <= RK_BUCKET_MIN:
//   ltr_rank_step  = rank_fused_kernel (read-only: keys are rebuilt from the state on the fly while the
//                    chunks are staged, counted, and perm is written directly) + rank_apply_kernel
//                    (promote/demote in place)                                              2 launches
//   ltr_queue_step = rank_fused_kernel + queue_tail_kernel (budget-walk scan, ran marking,
//                    promote/demote and aging in ONE single-workgroup launch)               2 launches
// (round 1: prepare, count, scatter, budget_prefix, budget_mark, age_update = 6 launches).
#include <cstdlib>

#include "ltr_internal.h"

namespace ltr {

namespace {

constexpr int RK_THREADS = 256;
constexpr int RK_ITILE = 64;      // keys ranked per workgroup (one per lane)
constexpr int RK_CHUNK = 1024;    // keys staged in LDS per iteration (8 KiB)

__device__ __forceinline__ uint32_t float_order_bits(float x) {
  uint32_t u = __float_as_uint(x);
  if (u == 0x80000000u) u = 0u;                       // -0.0 == +0.0 (Python float compare)
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending float -> ascending u32
}

// scheduler.py:986-993 on one request's counters (by value); returns the effective pri
__device__ __forceinline__ int promote_demote(int p, int& id, int& rn, int starv, int period) {
  if (id >= starv) { p = -1; id = 0; rn = period; }
  else if (p == -1 && rn <= 0) p = 0;
  return p;
}

__device__ __forceinline__ uint64_t make_key(float sc, int p, uint32_t tb, uint32_t flags) {
  const float k = (flags & LTR_RANK_ASCENDING) ? sc : -sc;
  uint64_t pbit = 0;
  if (flags & LTR_RANK_USE_PRI) pbit = (p < 0) ? 0ull : 1ull;   // pri in {-1, 0}: -1 first
  return (pbit << 63) | ((uint64_t)float_order_bits(k) << 31) | (uint64_t)(tb & 0x7fffffffu);
}

// key of position i from the (unmodified) state: promote/demote is evaluated by value
__device__ __forceinline__ uint64_t key_of(const float* __restrict__ score, const int32_t* __restrict__ pri,
                                           const int32_t* __restrict__ idle, const int32_t* __restrict__ runs,
                                           const uint32_t* __restrict__ tiebreak, const int32_t* __restrict__ members,
                                           int i, int starv, int period, uint32_t flags) {
  const int sl = members ? members[i] : i;
  int p = 0;
  if (flags & LTR_RANK_USE_PRI) {
    p = pri[sl];
    if (starv != -1) { int id = idle[sl], rn = runs[sl]; p = promote_demote(p, id, rn, starv, period); }
  }
  return make_key(score[sl], p, tiebreak ? tiebreak[i] : (uint32_t)i, flags);
}

// One launch for the whole ranking of a queue of N <= RK_BUCKET_MIN requests: a workgroup of NW waves owns 64
// positions; every thread rebuilds keys of the chunk being staged from the state (read-only, so workgroups never
// see each other's updates), the next chunk's keys are built in registers while the current one is counted
// (the two dependent L2 round trips of members -> state hide behind the compares), each wave counts 1/NW of
// the chunk for the same 64 positions, and perm is written directly.
template <int NW, int IT>
__global__ void __launch_bounds__(NW * 64) rank_fused_kernel(
    const float* __restrict__ score, const int32_t* __restrict__ pri, const int32_t* __restrict__ idle,
    const int32_t* __restrict__ runs, const uint32_t* __restrict__ tiebreak, const int32_t* __restrict__ members, int N,
    int starv, int period, uint32_t flags, int32_t* __restrict__ perm) {
  // IT = 64: a lane owns one of the workgroup's 64 positions and counts its wave's whole share of the chunk;
  // IT = 32: the workgroup owns 32 positions (twice as many workgroups: 256 at 8k requests = every CU), lane and lane ^ 32
  // own the same position and count one half of the wave's share each
  constexpr int T = NW * 64;                    // threads = keys staged per chunk
  constexpr int SHARE = 64 * 64 / IT / (64 / IT);   // keys of a chunk a wave looks at = T / NW = 64
  constexpr int PER_LANE = SHARE * IT / 64;     // ... of which one lane counts 64 (IT = 64) or 32 (IT = 32)
  __shared__ uint64_t skeys[2][T];
  __shared__ int32_t spart[T];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * IT + (lane & (IT - 1));
  const int sub = IT == 64 ? 0 : (lane >> 5) * PER_LANE;
  const uint64_t ki = (i < N) ? key_of(score, pri, idle, runs, tiebreak, members, i, starv, period, flags) : 0ull;
  int cnt = 0;
  uint64_t nxt = (threadIdx.x < N) ? key_of(score, pri, idle, runs, tiebreak, members, threadIdx.x, starv, period, flags) : ~0ull;
  int buf = 0;
  for (int j0 = 0; j0 < N; j0 += T, buf ^= 1) {
    skeys[buf][threadIdx.x] = nxt;
    __syncthreads();                            // chunk j0 is staged; the other buffer is free again
    const int jn = j0 + T + threadIdx.x;
    nxt = (jn < N) ? key_of(score, pri, idle, runs, tiebreak, members, jn, starv, period, flags) : ~0ull;
    const uint64_t* sk = skeys[buf] + wave * 64 + sub;
#pragma unroll 16
    for (int jj = 0; jj < PER_LANE; ++jj) cnt += (sk[jj] < ki) ? 1 : 0;
  }
  if (IT == 32) cnt += __shfl_xor(cnt, 32, 64);
  spart[threadIdx.x] = cnt;
  __syncthreads();
  if (wave == 0 && lane < IT && i < N) {
    int total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) total += spart[w * 64 + lane];
    perm[total] = i;
  }
}

// the same count over PRECOMPUTED keys (rank_prepare_kernel): one coalesced 8-byte load per key
template <int NW>
__global__ void __launch_bounds__(NW * 64) rank_count_direct_kernel(const uint64_t* __restrict__ keys, int N,
                                                                    int32_t* __restrict__ perm) {
  constexpr int T = NW * 64;
  __shared__ uint64_t skeys[2][T];
  __shared__ int32_t spart[T];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const uint64_t ki = (i < N) ? keys[i] : 0ull;
  int cnt = 0;
  uint64_t nxt = (threadIdx.x < N) ? keys[threadIdx.x] : ~0ull;
  int buf = 0;
  for (int j0 = 0; j0 < N; j0 += T, buf ^= 1) {
    skeys[buf][threadIdx.x] = nxt;
    __syncthreads();
    const int jn = j0 + T + threadIdx.x;
    nxt = (jn < N) ? keys[jn] : ~0ull;
    const uint64_t* sk = skeys[buf] + wave * 64;
#pragma unroll 16
    for (int jj = 0; jj < 64; ++jj) cnt += (sk[jj] < ki) ? 1 : 0;
  }
  spart[threadIdx.x] = cnt;
  __syncthreads();
  if (wave == 0 && i < N) {
    int total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) total += spart[w * 64 + lane];
    perm[total] = i;
  }
}

// promote/demote in place (scheduler.py:986-993), after rank_fused_kernel read the old state
__global__ void __launch_bounds__(256) rank_apply_kernel(int32_t* __restrict__ pri, int32_t* __restrict__ idle,
                                                         int32_t* __restrict__ runs, const int32_t* __restrict__ members,
                                                         int N, int starv, int period) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int sl = members ? members[i] : i;
  const int p0 = pri[sl], id0 = idle[sl], rn0 = runs[sl];
  int id = id0, rn = rn0;
  const int p = promote_demote(p0, id, rn, starv, period);
  if (p != p0) pri[sl] = p;
  if (id != id0) idle[sl] = id;
  if (rn != rn0) runs[sl] = rn;
}

// scheduler.py:986-993 + key build.  Also zeroes the rank accumulators.
__global__ void __launch_bounds__(256) rank_prepare_kernel(
    const float* __restrict__ score, int32_t* __restrict__ pri, int32_t* __restrict__ idle,
    int32_t* __restrict__ runs, const uint32_t* __restrict__ tiebreak, const int32_t* __restrict__ members, int N,
    int starv, int period, uint32_t flags, uint64_t* __restrict__ keys, int32_t* __restrict__ rank) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int sl = members ? members[i] : i;
  int p = 0;
  if (pri != nullptr) {
    p = pri[sl];
    if (starv != -1) {
      int id = idle[sl];
      if (id >= starv) {
        p = -1;
        pri[sl] = -1;
        idle[sl] = 0;
        runs[sl] = period;
      } else if (p == -1 && runs[sl] <= 0) {
        p = 0;
        pri[sl] = 0;
      }
    }
  }
  keys[i] = make_key(score[sl], p, tiebreak ? tiebreak[i] : (uint32_t)i, flags);
  rank[i] = 0;
}

// rank of every key among keys[0..N) by counting (the sorted sample of the bucketed path): perm[rank] = i
__global__ void __launch_bounds__(RK_THREADS) rank_count_kernel(
    const uint64_t* __restrict__ keys, int N, int32_t* __restrict__ perm) {
  __shared__ uint64_t skeys[RK_CHUNK];
  __shared__ int32_t spart[RK_THREADS];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = blockIdx.x * RK_ITILE + lane;
  const uint64_t ki = (i < N) ? keys[i] : 0ull;
  int cnt = 0;
  for (int j0 = 0; j0 < N; j0 += RK_CHUNK) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RK_CHUNK / RK_THREADS; ++r) {
      int j = j0 + r * RK_THREADS + threadIdx.x;
      skeys[r * RK_THREADS + threadIdx.x] = (j < N) ? keys[j] : ~0ull;  // pad: never < ki
    }
    __syncthreads();
    const uint64_t* sk = skeys + wave * (RK_CHUNK / 4);
#pragma unroll 16
    for (int jj = 0; jj < RK_CHUNK / 4; ++jj) cnt += (sk[jj] < ki) ? 1 : 0;
  }
  spart[threadIdx.x] = cnt;
  __syncthreads();
  if (wave == 0 && i < N) perm[spart[lane] + spart[64 + lane] + spart[128 + lane] + spart[192 + lane]] = i;
}

// ---- large queues (N > RK_BUCKET_MIN): sample-sort front end ---------------------------------
// The counting rank is O(N^2): 10 us at 8k, 0.44 ms at 64k.  Above RK_BUCKET_MIN the keys are first
// split into 256 buckets by 255 splitters taken from a sorted sample of 2048 keys (the keys are
// unique, so the buckets are balanced whatever the score distribution or the number of ties), and
// the counting rank runs inside each bucket only: O(N * N/256).  The order of keys inside a bucket
// after the atomic scatter is arbitrary, which is harmless: the in-bucket rank depends on the keys
// alone, so the permutation is the same bit-exact stable order.
constexpr int RK_BUCKET_MIN = 12288;   // measured crossover (round 1, 3-launch counting rank): 8k 45 us vs 62 us; 16k 69 vs 59 us
constexpr int RK_NBUCKET = 256;
constexpr int RK_NSAMPLE = 2048;

__global__ void __launch_bounds__(256) rank_sample_kernel(const uint64_t* __restrict__ keys, int N,
                                                          uint64_t* __restrict__ samp) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < RK_NSAMPLE) samp[k] = keys[(long long)k * N / RK_NSAMPLE];
}

// splitters: every 8th key of the sorted sample (perm_s = sorted order of samp); also zero the histogram
__global__ void __launch_bounds__(256) rank_splitters_kernel(const uint64_t* __restrict__ samp,
                                                             const int32_t* __restrict__ perm_s,
                                                             uint64_t* __restrict__ spl, int32_t* __restrict__ cnt) {
  const int k = threadIdx.x;
  if (k >= 1) spl[k - 1] = samp[perm_s[k * (RK_NSAMPLE / RK_NBUCKET)]];   // spl[0..254]
  cnt[k] = 0;
}

// bucket of key = number of splitters <= key; slot = arrival order inside the bucket.
// 4096 keys per workgroup are first counted in an LDS histogram, then ONE global atomic per
// (workgroup, non-empty bucket) reserves a range: ~16x fewer same-address atomics than one per key.
constexpr int RK_ASSIGN_THREADS = 1024, RK_ASSIGN_KEYS = 4;
__global__ void __launch_bounds__(RK_ASSIGN_THREADS) rank_assign_kernel(
    const uint64_t* __restrict__ keys, int N, const uint64_t* __restrict__ spl, int32_t* __restrict__ cnt,
    uint8_t* __restrict__ bid, int32_t* __restrict__ slot) {
  __shared__ uint64_t s_spl[RK_NBUCKET];
  __shared__ int s_cnt[RK_NBUCKET];
  __shared__ int s_base[RK_NBUCKET];
  const int tid = threadIdx.x;
  if (tid < RK_NBUCKET - 1) s_spl[tid] = spl[tid];
  if (tid < RK_NBUCKET) s_cnt[tid] = 0;
  __syncthreads();
  int myb[RK_ASSIGN_KEYS], mys[RK_ASSIGN_KEYS];
#pragma unroll
  for (int q = 0; q < RK_ASSIGN_KEYS; ++q) {
    const int i = (blockIdx.x * RK_ASSIGN_KEYS + q) * RK_ASSIGN_THREADS + tid;
    myb[q] = -1;
    if (i < N) {
      const uint64_t k = keys[i];
      int lo = 0, hi = RK_NBUCKET - 1;          // answer = #splitters <= k, in [0, 255]
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_spl[mid] <= k) lo = mid + 1; else hi = mid;
      }
      myb[q] = lo;
      mys[q] = atomicAdd(&s_cnt[lo], 1);
    }
  }
  __syncthreads();
  if (tid < RK_NBUCKET) s_base[tid] = s_cnt[tid] ? atomicAdd(&cnt[tid], s_cnt[tid]) : 0;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RK_ASSIGN_KEYS; ++q) {
    const int i = (blockIdx.x * RK_ASSIGN_KEYS + q) * RK_ASSIGN_THREADS + tid;
    if (myb[q] >= 0) { bid[i] = (uint8_t)myb[q]; slot[i] = s_base[myb[q]] + mys[q]; }
  }
}

__global__ void __launch_bounds__(256) rank_bucket_scan_kernel(const int32_t* __restrict__ cnt,
                                                               int32_t* __restrict__ off) {
  __shared__ int s[RK_NBUCKET];
  const int k = threadIdx.x;
  s[k] = cnt[k];
  __syncthreads();
  for (int d = 1; d < RK_NBUCKET; d <<= 1) {
    const int v = k >= d ? s[k - d] : 0;
    __syncthreads();
    s[k] += v;
    __syncthreads();
  }
  off[k + 1] = s[k];
  if (k == 0) off[0] = 0;
}

__global__ void __launch_bounds__(256) rank_bucket_scatter_kernel(const uint64_t* __restrict__ keys, int N,
                                                                  const uint8_t* __restrict__ bid,
                                                                  const int32_t* __restrict__ slot,
                                                                  const int32_t* __restrict__ off,
                                                                  uint64_t* __restrict__ bkeys,
                                                                  int32_t* __restrict__ bidx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int p = off[bid[i]] + slot[i];
  bkeys[p] = keys[i];
  bidx[p] = i;
}

// one workgroup per bucket: rank inside the bucket by counting, keys staged through LDS
__global__ void __launch_bounds__(256) rank_in_bucket_kernel(const uint64_t* __restrict__ bkeys,
                                                             const int32_t* __restrict__ bidx,
                                                             const int32_t* __restrict__ off,
                                                             int32_t* __restrict__ perm) {
  __shared__ uint64_t sk[RK_CHUNK];
  const int b = blockIdx.x;
  const int beg = off[b], end = off[b + 1];
  for (int e0 = beg; e0 < end; e0 += 256) {              // 256 elements per sweep (one per thread)
    const int e = e0 + threadIdx.x;
    const uint64_t ke = e < end ? bkeys[e] : 0ull;
    int cntl = 0;
    for (int j0 = beg; j0 < end; j0 += RK_CHUNK) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < RK_CHUNK / 256; ++r) {
        const int j = j0 + r * 256 + threadIdx.x;
        sk[r * 256 + threadIdx.x] = j < end ? bkeys[j] : ~0ull;
      }
      __syncthreads();
      const int lim = (min(RK_CHUNK, end - j0) + 15) & ~15;   // the tail is padded with ~0: never < ke
#pragma unroll 16
      for (int jj = 0; jj < lim; ++jj) cntl += (sk[jj] < ke) ? 1 : 0;
    }
    if (e < end) perm[beg + cntl] = bidx[e];
  }
}

// scheduler.py:1358-1365.  `ran` is a u8 flag per POSITION, or - when ran == nullptr - membership of the
// request's SLOT in the ascending list ran_slots[n_ran] (what a scheduler hands over: the <= max_num_seqs
// requests it scheduled this step; binary search instead of an N-byte mask upload).
__global__ void __launch_bounds__(256) age_update_kernel(const uint8_t* __restrict__ ran,
                                                         const int32_t* __restrict__ ran_slots, int n_ran,
                                                         int32_t* __restrict__ pri, int32_t* __restrict__ idle,
                                                         int32_t* __restrict__ runs,
                                                         const int32_t* __restrict__ members, int N) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int sl = members ? members[i] : i;
  bool r;
  if (ran != nullptr) {
    r = ran[i] != 0;
  } else {
    int lo = 0, hi = n_ran;                    // first index with ran_slots[idx] >= sl
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ran_slots[mid] < sl) lo = mid + 1; else hi = mid; }
    r = lo < n_ran && ran_slots[lo] == sl;
  }
  if (r) {
    if (pri[sl] == -1) runs[sl] -= 1;
    idle[sl] = 0;
  } else {
    idle[sl] += 1;
  }
}

// Budget-walk prefix (scheduler.py:1137-1211 with _get_num_new_tokens :1867-1888 and
// SchedulingBudget.can_schedule :51-55).  With chunking on (the walk hard-codes
// enable_chunking = True, :1128) a single-sequence request is granted
// min(need, remaining_token_budget), so the walk selects request k (in ranked order) iff
// for every j <= k:  need_j > 0, sum_{i<j} need_i < token_budget and
// sum_{i<=j} seqs_i <= max_num_seqs; it breaks at the first k that fails.  A group is chunked iff it
// has exactly ONE sequence in the walked status (:1884) - `chunkable`, which is not `seqs == 1`: a
// WAITING prompt with best_of > 1 has one sequence and new_seqs = best_of (sequence.py:500-504).
// Groups that are not chunkable must fit whole.
// Single workgroup, blocked scan over the ranked order.
constexpr int BP_THREADS = 1024;
struct BudgetShared {
  long long tok[BP_THREADS / 64], seq[BP_THREADS / 64];
  long long carry_tok, carry_seq;
  int first_bad;
};
// Blocked scan over the ranked order by one workgroup of BP_THREADS threads; returns the number of selected
// requests (uniform).  The walk stops at the first block that holds a misfit - with max_num_seqs = 256 that is the
// first block.  `own_block`: this thread keeps the grant of position own_block * BP_THREADS + tid in *own_grant
// (0 when the walk stops before it or the position is not selected); granted[] itself is written here only when
// write_granted (single-workgroup callers).
__device__ int budget_scan(const int32_t* __restrict__ perm, const int32_t* __restrict__ new_tokens,
                           const int32_t* __restrict__ new_seqs, const uint8_t* __restrict__ chunkable, int N,
                           long long token_budget, long long max_seqs, int32_t* __restrict__ granted, int own_block,
                           int* own_grant, BudgetShared& sh) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { sh.carry_tok = 0; sh.carry_seq = 0; sh.first_bad = N; }
  __syncthreads();
  int mine = 0;
  for (int base = 0; base < N; base += BP_THREADS) {
    int k = base + tid;
    long long t = 0, q = 0;
    int nt = 0, nq = 0, r = 0;
    bool chunk = true;
    if (k < N) {
      r = perm[k]; nt = new_tokens[r]; nq = new_seqs[r]; t = nt; q = nq;
      chunk = chunkable ? chunkable[r] != 0 : nq <= 1;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {   // inclusive scan inside the wave
      long long tt = __shfl_up(t, o, 64), qq = __shfl_up(q, o, 64);
      if (lane >= o) { t += tt; q += qq; }
    }
    if (lane == 63) { sh.tok[wave] = t; sh.seq[wave] = q; }
    __syncthreads();
    long long ot = sh.carry_tok, oq = sh.carry_seq;
    for (int w = 0; w < wave; ++w) { ot += sh.tok[w]; oq += sh.seq[w]; }
    t += ot; q += oq;
    const long long before = t - nt;
    bool bad = (k < N) && (nt == 0 || before >= token_budget || q > max_seqs ||
                           (!chunk && t > token_budget));
    if (bad) atomicMin(&sh.first_bad, k);
    long long g = token_budget - before;
    const int grant = (int)(chunk && g < nt ? (g > 0 ? g : 0) : nt);
    if (k < N && granted != nullptr) granted[r] = grant;
    if (base == own_block * BP_THREADS) mine = grant;
    __syncthreads();
    if (tid == BP_THREADS - 1) { sh.carry_tok = t; sh.carry_seq = q; }
    __syncthreads();
    if (sh.first_bad < N) break;   // uniform: read after the barrier
  }
  __syncthreads();
  if (own_grant) *own_grant = mine;
  return sh.first_bad;
}

__global__ void __launch_bounds__(BP_THREADS) budget_prefix_kernel(
    const int32_t* __restrict__ perm, const int32_t* __restrict__ new_tokens,
    const int32_t* __restrict__ new_seqs, const uint8_t* __restrict__ chunkable, int N, long long token_budget,
    long long max_seqs, int32_t* __restrict__ n_sel, uint8_t* __restrict__ ran, int32_t* __restrict__ granted) {
  __shared__ BudgetShared sh;
  const int nsel = budget_scan(perm, new_tokens, new_seqs, chunkable, N, token_budget, max_seqs, granted, -1, nullptr, sh);
  if (threadIdx.x == 0) *n_sel = nsel;
}

// The rest of a steady scheduler step in ONE launch, after the rank: budget-walk scan (scheduler.py:1137-1211), ran
// marking, promote/demote write-back (:986-993, when the rank kernel only read the state) and aging (:1358-1365) of
// every queued request.  Workgroup b owns positions [b * 1024, (b + 1) * 1024) of the ranked order; every
// workgroup repeats the scan (it stops in the first block unless the budget exceeds it), so no workgroup waits
// for another.
__global__ void __launch_bounds__(BP_THREADS) queue_tail_kernel(
    const int32_t* __restrict__ perm, const int32_t* __restrict__ members, const int32_t* __restrict__ new_tokens,
    const int32_t* __restrict__ new_seqs, const uint8_t* __restrict__ chunkable, int N, long long token_budget,
    long long max_seqs, int starv, int period, int apply_promote, int32_t* __restrict__ pri,
    int32_t* __restrict__ idle, int32_t* __restrict__ runs, int32_t* __restrict__ n_sel, uint8_t* __restrict__ ran,
    int32_t* __restrict__ granted) {
  __shared__ BudgetShared sh;
  int grant = 0;
  const int nsel = budget_scan(perm, new_tokens, new_seqs, chunkable, N, token_budget, max_seqs, nullptr,
                               (int)blockIdx.x, &grant, sh);
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_sel = nsel;
  const int k = blockIdx.x * BP_THREADS + threadIdx.x;
  if (k >= N) return;
  const int r = perm[k];
  const int sl = members ? members[r] : r;
  const bool rn_ = k < nsel;
  if (ran != nullptr) ran[r] = rn_ ? 1 : 0;
  if (granted != nullptr) granted[r] = rn_ ? grant : 0;
  if (pri != nullptr) {
    int p = pri[sl], id = idle[sl], ru = runs[sl];
    if (apply_promote && starv != -1) p = promote_demote(p, id, ru, starv, period);
    if (rn_) { if (p == -1) ru -= 1; id = 0; } else { id += 1; }
    pri[sl] = p; idle[sl] = id; runs[sl] = ru;
  }
}

// second phase on the whole chip: ran[perm[k]] = k < n_sel, granted = 0 past the selection
__global__ void __launch_bounds__(256) budget_mark_kernel(const int32_t* __restrict__ perm, int N,
                                                          const int32_t* __restrict__ n_sel, uint8_t* __restrict__ ran,
                                                          int32_t* __restrict__ granted) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  const int nsel = *n_sel;
  const int r = perm[k];
  if (ran != nullptr) ran[r] = (k < nsel) ? 1 : 0;
  if (granted != nullptr && k >= nsel) granted[r] = 0;
}

// Victim selection of reserve_free_blocks (scheduler.py:1376-1452) as ONE reversed exclusive prefix
// sum over the ranked order: position k (request r = perm[k]) frees w_k blocks when evicted,
//   unselected (k >= n_sel): w = phys[r] if r has RUNNING sequences else 0     (:1400-1420)
//   selected   (k <  n_sel): w = nrun + phys (running) | phys + nswap (swapped) | logical (waiting)   (:1422-1447)
// and is evicted iff the blocks freed by everything behind it are still short of `need`
// (the reference walks from the low-priority end and stops as soon as need <= 0).
// With new_seqs != nullptr the kernel first accumulates gpu_block_required of the selection
// (:1137-1211: running +new_seqs, swapped +phys+nswap, waiting +logical) and need = required - need_in.
// Single workgroup, blocked scan from the end; stops at the first block that needs no eviction.
__global__ void __launch_bounds__(BP_THREADS) reserve_select_kernel(
    const int32_t* __restrict__ perm, const int32_t* __restrict__ n_sel_p, const uint8_t* __restrict__ state,
    const int32_t* __restrict__ phys, const int32_t* __restrict__ logical, const int32_t* __restrict__ nrun,
    const int32_t* __restrict__ nswap, const int32_t* __restrict__ new_seqs, int N, long long need_in,
    uint8_t* __restrict__ action, int32_t* __restrict__ n_exec, int32_t* __restrict__ blocks_required) {
  __shared__ long long s_w[BP_THREADS / 64];
  __shared__ long long carry, s_need;
  __shared__ int s_popped;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nsel = min(*n_sel_p, N);
  // action defaults to 0 everywhere
  for (int k = tid; k < N; k += BP_THREADS) action[perm[k]] = 0;
  long long req = 0;
  if (new_seqs != nullptr) {
    for (int k = tid; k < nsel; k += BP_THREADS) {
      const int r = perm[k];
      const int st = state[r];
      req += st == 1 ? new_seqs[r] : (st == 2 ? phys[r] + nswap[r] : logical[r]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) req += __shfl_xor(req, o, 64);
    if (lane == 0) s_w[wave] = req;
  }
  if (tid == 0) { carry = 0; s_popped = 0; }
  __syncthreads();
  if (tid == 0) {
    long long need = need_in;
    if (new_seqs != nullptr) {
      long long t = 0;
      for (int w = 0; w < BP_THREADS / 64; ++w) t += s_w[w];
      if (blocks_required) *blocks_required = (int32_t)t;
      need = t - need_in;
    } else if (blocks_required) {
      *blocks_required = 0;
    }
    s_need = need;
  }
  __syncthreads();
  const long long need = s_need;
  if (need > 0) {
    for (int base = 0; base < N; base += BP_THREADS) {
      const int k = N - 1 - (base + tid);               // walk from the low-priority end
      long long w = 0;
      int r = 0, st = 0;
      bool sel = false;
      if (k >= 0) {
        r = perm[k]; st = state[r]; sel = k < nsel;
        if (sel) w = st == 1 ? (long long)nrun[r] + phys[r] : (st == 2 ? (long long)phys[r] + nswap[r] : (long long)logical[r]);
        else w = st == 1 ? phys[r] : 0;
      }
      long long t = w;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { long long tt = __shfl_up(t, o, 64); if (lane >= o) t += tt; }
      __syncthreads();                                   // previous round's s_w / carry reads are done
      if (lane == 63) s_w[wave] = t;
      __syncthreads();
      long long off = carry;
      for (int x = 0; x < wave; ++x) off += s_w[x];
      t += off;
      const long long before = t - w;                    // freed by everything behind this position
      if (k >= 0 && before < need) {
        if (sel) { action[r] = st == 1 ? 2 : 3; atomicAdd(&s_popped, 1); }
        else if (st == 1) action[r] = 1;
      }
      __syncthreads();
      if (tid == BP_THREADS - 1) carry = t;
      __syncthreads();
      if (carry >= need) break;                          // uniform
    }
  }
  __syncthreads();
  if (tid == 0) *n_exec = nsel - s_popped;
}

}  // namespace

// keys u64[n] | rank i32[n] | (bucket path) bkeys u64[n] | bidx i32[n] | slot i32[n] | bid u8[n] |
// samp u64[4096] | perm_s i32[4096] | spl u64[256] | cnt i32[256] | off i32[257]
size_t rank_workspace_bytes(int64_t N) {
  int64_t n = (N + 63) / 64 * 64;
  size_t b = (size_t)(n * sizeof(uint64_t) + n * sizeof(int32_t) + 256);
  if (N > RK_BUCKET_MIN)
    b += (size_t)(n * (8 + 4 + 4 + 1)) + RK_NSAMPLE * (8 + 4) + RK_NBUCKET * (8 + 4) + (RK_NBUCKET + 1) * 4 + 1024;
  return b;
}

namespace {

int check_rank_args(const int32_t* pri, const int32_t* idle, const int32_t* runs, int starv, uint32_t& flags) {
  if (starv != -1) flags |= LTR_RANK_USE_PRI;           // scheduler.py:996 vs :998
  if ((flags & LTR_RANK_USE_PRI) && pri == nullptr) {
    set_error("ltr_rank_step: pri is NULL but the key uses it");
    return LTR_E_INVAL;
  }
  if (starv != -1 && (idle == nullptr || runs == nullptr)) {
    set_error("ltr_rank_step: starvation control needs idle and runs");
    return LTR_E_INVAL;
  }
  return LTR_OK;
}

// N > RK_BUCKET_MIN: promote/demote in place + keys, sample-sort front end, counting rank inside the buckets
int launch_rank_bucketed(const float* scores, int32_t* pri, int32_t* idle, int32_t* runs, const uint32_t* tiebreak,
                         const int32_t* members, int N, int starv, int period, uint32_t flags, int32_t* perm_out,
                         void* ws, size_t ws_bytes, hipStream_t s) {
  if (ws == nullptr || ws_bytes < rank_workspace_bytes(N)) {
    set_error("ltr_rank_step: workspace %zu < %zu", ws_bytes, rank_workspace_bytes(N));
    return LTR_E_NOMEM;
  }
  int64_t n64 = ((int64_t)N + 63) / 64 * 64;
  uint64_t* keys = (uint64_t*)ws;
  int32_t* rank = (int32_t*)((char*)ws + n64 * sizeof(uint64_t));
  rank_prepare_kernel<<<(N + 255) / 256, 256, 0, s>>>(scores, (flags & LTR_RANK_USE_PRI) ? pri : nullptr, idle,
                                                      runs, tiebreak, members, N, starv, period, flags, keys, rank);
  LTR_LAUNCH_CHECK();
  char* p = (char*)ws + n64 * (sizeof(uint64_t) + sizeof(int32_t)) + 256;
  auto take = [&](size_t bytes) { char* q = p; p += (bytes + 255) & ~(size_t)255; return q; };
  uint64_t* bkeys = (uint64_t*)take(n64 * 8);
  int32_t* bidx = (int32_t*)take(n64 * 4);
  int32_t* slot = (int32_t*)take(n64 * 4);
  uint8_t* bid = (uint8_t*)take(n64);
  uint64_t* samp = (uint64_t*)take(RK_NSAMPLE * 8);
  int32_t* perm_s = (int32_t*)take(RK_NSAMPLE * 4);
  uint64_t* spl = (uint64_t*)take(RK_NBUCKET * 8);
  int32_t* cnt = (int32_t*)take(RK_NBUCKET * 4);
  int32_t* off = (int32_t*)take((RK_NBUCKET + 1) * 4);
  rank_sample_kernel<<<RK_NSAMPLE / 256, 256, 0, s>>>(keys, N, samp);
  LTR_LAUNCH_CHECK();
  rank_count_kernel<<<RK_NSAMPLE / RK_ITILE, RK_THREADS, 0, s>>>(samp, RK_NSAMPLE, perm_s);
  LTR_LAUNCH_CHECK();
  rank_splitters_kernel<<<1, RK_NBUCKET, 0, s>>>(samp, perm_s, spl, cnt);
  LTR_LAUNCH_CHECK();
  rank_assign_kernel<<<(N + RK_ASSIGN_THREADS * RK_ASSIGN_KEYS - 1) / (RK_ASSIGN_THREADS * RK_ASSIGN_KEYS), RK_ASSIGN_THREADS, 0, s>>>(
      keys, N, spl, cnt, bid, slot);
  LTR_LAUNCH_CHECK();
  rank_bucket_scan_kernel<<<1, RK_NBUCKET, 0, s>>>(cnt, off);
  LTR_LAUNCH_CHECK();
  rank_bucket_scatter_kernel<<<(N + 255) / 256, 256, 0, s>>>(keys, N, bid, slot, off, bkeys, bidx);
  LTR_LAUNCH_CHECK();
  rank_in_bucket_kernel<<<RK_NBUCKET, 256, 0, s>>>(bkeys, bidx, off, perm_out);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // namespace

namespace {

// experiment knobs (diag only): LTR_RANK_MODE = 0 fused keys-on-the-fly (default) | 1 prepare + count;
// LTR_RANK_NW = waves per workgroup of the counting kernel (4, 8 or 16)
int rank_mode() { static int m = [] { const char* e = getenv("LTR_RANK_MODE"); return e ? atoi(e) : 0; }(); return m; }
int rank_nw() { static int m = [] { const char* e = getenv("LTR_RANK_NW"); return e ? atoi(e) : 16; }(); return m; }

// ranking of N <= RK_BUCKET_MIN requests; returns whether promote/demote was already applied in place
int launch_rank_small(const float* scores, int32_t* pri, int32_t* idle, int32_t* runs, const uint32_t* tiebreak,
                      const int32_t* members, int N, int starv, int period, uint32_t flags, int32_t* perm_out,
                      void* ws, size_t ws_bytes, bool* applied, hipStream_t s) {
  const int nw = rank_nw();
  static const int it = [] { const char* e = getenv("LTR_RANK_IT"); return e ? atoi(e) : 32; }();   // diag: positions per workgroup
  int grid = (N + 63) / 64;
  if (rank_mode() == 1 && ws != nullptr && ws_bytes >= rank_workspace_bytes(N)) {
    int64_t n64 = ((int64_t)N + 63) / 64 * 64;
    uint64_t* keys = (uint64_t*)ws;
    int32_t* rank = (int32_t*)((char*)ws + n64 * sizeof(uint64_t));
    rank_prepare_kernel<<<(N + 255) / 256, 256, 0, s>>>(scores, (flags & LTR_RANK_USE_PRI) ? pri : nullptr, idle,
                                                        runs, tiebreak, members, N, starv, period, flags, keys, rank);
    LTR_LAUNCH_CHECK();
    if (nw == 4) rank_count_direct_kernel<4><<<grid, 256, 0, s>>>(keys, N, perm_out);
    else if (nw == 8) rank_count_direct_kernel<8><<<grid, 512, 0, s>>>(keys, N, perm_out);
    else rank_count_direct_kernel<16><<<grid, 1024, 0, s>>>(keys, N, perm_out);
    LTR_LAUNCH_CHECK();
    *applied = true;
    return LTR_OK;
  }
  if (it == 32) {
    grid = (N + 31) / 32;
    if (nw == 8) rank_fused_kernel<8, 32><<<grid, 512, 0, s>>>(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, perm_out);
    else rank_fused_kernel<16, 32><<<grid, 1024, 0, s>>>(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, perm_out);
  } else if (nw == 4) rank_fused_kernel<4, 64><<<grid, 256, 0, s>>>(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, perm_out);
  else if (nw == 8) rank_fused_kernel<8, 64><<<grid, 512, 0, s>>>(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, perm_out);
  else rank_fused_kernel<16, 64><<<grid, 1024, 0, s>>>(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, perm_out);
  LTR_LAUNCH_CHECK();
  *applied = false;
  return LTR_OK;
}

}  // namespace

int launch_rank_step(const float* scores, int32_t* pri, int32_t* idle, int32_t* runs, const uint32_t* tiebreak,
                     const int32_t* members, int N, int starv, int period, uint32_t flags, int32_t* perm_out, void* ws,
                     size_t ws_bytes, hipStream_t s) {
  if (N == 0) return LTR_OK;
  int rc = check_rank_args(pri, idle, runs, starv, flags);
  if (rc) return rc;
  if (N > RK_BUCKET_MIN)
    return launch_rank_bucketed(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, perm_out, ws,
                                ws_bytes, s);
  bool applied = false;
  if ((rc = launch_rank_small(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, perm_out, ws, ws_bytes,
                              &applied, s)))
    return rc;
  if (starv != -1 && !applied) {
    rank_apply_kernel<<<(N + 255) / 256, 256, 0, s>>>(pri, idle, runs, members, N, starv, period);
    LTR_LAUNCH_CHECK();
  }
  return LTR_OK;
}

int launch_queue_step(const float* scores, int32_t* pri, int32_t* idle, int32_t* runs, const uint32_t* tiebreak,
                      const int32_t* members, int N, int starv, int period, uint32_t flags, const int32_t* new_tokens,
                      const int32_t* new_seqs, const uint8_t* chunkable, int64_t token_budget, int64_t max_seqs,
                      int32_t* perm_out, int32_t* n_sel, uint8_t* ran, int32_t* granted, void* ws, size_t ws_bytes,
                      hipStream_t s) {
  int rc = check_rank_args(pri, idle, runs, starv, flags);
  if (rc) return rc;
  bool applied = true;
  if (N > RK_BUCKET_MIN) {
    rc = launch_rank_bucketed(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, perm_out, ws,
                              ws_bytes, s);                 // rank_prepare_kernel writes the promote/demote
    if (rc) return rc;
  } else if (N > 0) {
    if ((rc = launch_rank_small(scores, pri, idle, runs, tiebreak, members, N, starv, period, flags, perm_out, ws,
                                ws_bytes, &applied, s)))
      return rc;
  }
  const int grid = N > 0 ? (N + BP_THREADS - 1) / BP_THREADS : 1;
  queue_tail_kernel<<<grid, BP_THREADS, 0, s>>>(perm_out, members, new_tokens, new_seqs, chunkable, N,
                                                (long long)token_budget, (long long)max_seqs, starv, period,
                                                applied ? 0 : 1, (pri && idle && runs) ? pri : nullptr, idle, runs,
                                                n_sel, ran, granted);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

int launch_age_update(const uint8_t* ran, const int32_t* ran_slots, int n_ran, int32_t* pri, int32_t* idle,
                      int32_t* runs, const int32_t* members, int N, hipStream_t s) {
  if (N == 0) return LTR_OK;
  age_update_kernel<<<(N + 255) / 256, 256, 0, s>>>(ran, ran_slots, n_ran, pri, idle, runs, members, N);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

int launch_budget_prefix(const int32_t* perm, const int32_t* new_tokens, const int32_t* new_seqs,
                         const uint8_t* chunkable, int N, int64_t token_budget, int64_t max_seqs, int32_t* n_sel,
                         uint8_t* ran, int32_t* granted, hipStream_t s) {
  budget_prefix_kernel<<<1, BP_THREADS, 0, s>>>(perm, new_tokens, new_seqs, chunkable, N, (long long)token_budget,
                                                (long long)max_seqs, n_sel, ran, granted);
  LTR_LAUNCH_CHECK();
  if (N > 0 && (ran != nullptr || granted != nullptr)) {
    budget_mark_kernel<<<(N + 255) / 256, 256, 0, s>>>(perm, N, n_sel, ran, granted);
    LTR_LAUNCH_CHECK();
  }
  return LTR_OK;
}

int launch_reserve_select(const int32_t* perm, const int32_t* n_sel, const uint8_t* state, const int32_t* phys,
                          const int32_t* logical, const int32_t* nrun, const int32_t* nswap, const int32_t* new_seqs,
                          int N, int64_t need_in, uint8_t* action, int32_t* n_exec, int32_t* blocks_required,
                          hipStream_t s) {
  reserve_select_kernel<<<1, BP_THREADS, 0, s>>>(perm, n_sel, state, phys, logical, nrun, nswap, new_seqs, N,
                                                 (long long)need_in, action, n_exec, blocks_required);
  LTR_LAUNCH_CHECK();
  return LTR_OK;
}

}  // namespace ltr
