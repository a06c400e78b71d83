// sann_kernels.hip -- gfx950 kernels of the SimClusters-ANN hot path (general path + merges).
//
// Reference semantics (paths relative to /root/reference/):
//   simclusters-ann/server/src/main/scala/com/twitter/simclustersann/candidate_source/
//     ApproximateCosineSimilarity.scala:83-127   accumulate, normalise, filter, sort, take
//
// Kernels in this file
//   unit_general_kernel  one workgroup per (query, partition) unit; hash table in global memory,
//                        clusters accumulated in order with a workgroup barrier between them.
//                        Always correct for any size / duplication; it is the fallback of the
//                        LDS fast path (sann_fast.hip) and the first path that was parity-green.
//   merge_kernel         one workgroup per query: exact top-k over the units' candidates under
//                        the total order (score desc by Double.compare, tweet id asc).  While it stages
//                        the units' lists in LDS it fetches the posting of every candidate a fast unit
//                        handed over as (cluster, posting position) and computes its exact fp64 score
//                        (:92-96, :111-125).  A cut from a sorted sample (radix passes as fallback) leaves
//                        <= 512 / 1024 survivors, sorted by in-register wave sorts of 64-entry runs and
//                        pairwise merges; more candidates than the staging area holds go through it in
//                        rounds (a tournament).  Also proves the result exact: a unit that withheld
//                        candidates below its threshold is harmless iff that threshold is <= the global
//                        k-th key.
//   merge_wave_kernel    the same merge for a shard's small queries (<= 16 units, <= 512 candidates):
//                        one WAVE per query, candidates in registers, no workgroup barrier.
//   merge_shards_kernel  the owner's merge of the per-shard results the all-to-all delivered: sorted
//                        lists are ranked against each other (a lock-step binary search per list);
//                        unsorted input is selected and sorted.
//
// Compiled with -ffp-contract=off (see sann_math.h).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "sann_device.h"
#include "sann_kernels.h"
#include "sann_math.h"
#include "sann_select.h"
#include "sann_wave.h"

namespace sann {

constexpr int WG = 256;
constexpr int KMAX = 1024;       // >= MaxNumResultsUpperBound (1000), ApproximateCosineSimilarity.scala:41

// ---------------------------------------------------------------------------------------------
// normalisation, ApproximateCosineSimilarity.scala:111-119
// ---------------------------------------------------------------------------------------------
__device__ inline double normalise(int alg, double dot, double nsq, double l2norm, double lognorm) {
  switch (alg) {
    case 3: return dot / lognorm / strict_log(1 + nsq);
    case 2: return dot / l2norm / sqrt(nsq);
    case 4: return dot / sqrt(nsq);
    case 1: return dot;
    default: return __builtin_nan("");  // scala MatchError; never >= minScore
  }
}

// ---------------------------------------------------------------------------------------------
// exact top-k threshold by MSB-first radix select over the 128-bit key (score_key, id_key)
// ---------------------------------------------------------------------------------------------
__device__ inline bool prefix_match(uint64_t hi, uint64_t lo, uint64_t phi, uint64_t plo, int pass) {
  if (pass == 0) return true;
  if (pass < 8) {
    int sh = 64 - 8 * pass;
    return (hi >> sh) == (phi >> sh);
  }
  if (pass == 8) return hi == phi;
  int sh = 128 - 8 * pass;  // 56..8
  return hi == phi && (lo >> sh) == (plo >> sh);
}
__device__ inline unsigned digit_of(uint64_t hi, uint64_t lo, int pass) {
  return pass < 8 ? (unsigned)((hi >> (56 - 8 * pass)) & 0xff) : (unsigned)((lo >> (56 - 8 * (pass - 8))) & 0xff);
}
__device__ inline bool key_ge(uint64_t hi, uint64_t lo, uint64_t thi, uint64_t tlo) {
  return hi > thi || (hi == thi && lo >= tlo);
}
__device__ inline bool key_gt(uint64_t hi, uint64_t lo, uint64_t thi, uint64_t tlo) {
  return hi > thi || (hi == thi && lo > tlo);
}

// Src: int size(); bool get(int i, uint64_t& hi, uint64_t& lo)  (false = slot not a candidate)
// After the call every candidate with key >= (thr_hi, thr_lo) is in the top-k and there are
// exactly min(k, n_valid) of them.  s_hist: 256 uints, s_ctl: 4 ints, both in LDS.
template <class Src>
__device__ void wg_select_threshold(const Src &src, int k, unsigned *s_hist, int *s_ctl, uint64_t &thr_hi,
                                    uint64_t &thr_lo, int &n_valid) {
  const int tid = threadIdx.x;
  const int n = src.size();
  if (tid == 0) s_ctl[0] = 0;
  __syncthreads();
  int local = 0;
  for (int i = tid; i < n; i += WG) {
    uint64_t hi, lo;
    if (src.get(i, hi, lo)) local++;
  }
  if (local) atomicAdd(&s_ctl[0], local);
  __syncthreads();
  n_valid = s_ctl[0];
  thr_hi = 0;
  thr_lo = 0;
  if (n_valid <= k || k <= 0) {
    if (k <= 0) { thr_hi = ~0ull; thr_lo = ~0ull; }
    __syncthreads();
    return;
  }
  uint64_t phi = 0, plo = 0;
  int need = k;
  for (int pass = 0; pass < 16; pass++) {
    for (int i = tid; i < 256; i += WG) s_hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += WG) {
      uint64_t hi, lo;
      if (src.get(i, hi, lo) && prefix_match(hi, lo, phi, plo, pass)) atomicAdd(&s_hist[digit_of(hi, lo, pass)], 1u);
    }
    __syncthreads();
    if (tid < 64) wave_find_digit(s_hist, need, &s_ctl[1]);  // d, entries above d, entries in d
    __syncthreads();
    uint64_t d = (uint64_t)s_ctl[1];
    const int done = (s_ctl[3] == need - s_ctl[2]);
    need = need - s_ctl[2];
    if (pass < 8) phi |= d << (56 - 8 * pass);
    else plo |= d << (56 - 8 * (pass - 8));
    __syncthreads();
    if (done) break;
  }
  thr_hi = phi;
  thr_lo = plo;
}

// Bitonic sort, descending by (hi, lo), n a power of two, arrays in LDS.
__device__ void bitonic_sort_desc(uint64_t *hi, uint64_t *lo, int n) {
  const int tid = threadIdx.x;
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (n >> 1); t += WG) {
        int i = 2 * t - (t & (stride - 1));
        int j = i + stride;
        bool desc = ((i & size) == 0);
        uint64_t ah = hi[i], al = lo[i], bh = hi[j], bl = lo[j];
        bool a_lt_b = ah < bh || (ah == bh && al < bl);
        bool a_gt_b = ah > bh || (ah == bh && al > bl);
        if (desc ? a_lt_b : a_gt_b) {
          hi[i] = bh; lo[i] = bl;
          hi[j] = ah; lo[j] = al;
        }
      }
      __syncthreads();
    }
  }
}

// Same sort on packed 16-byte entries {x = score key, y = id key}: one ds_read_b128 / ds_write_b128
// per element instead of two 8-byte accesses.
__device__ void bitonic_sort_desc_packed(ulonglong2 *e, int n) {
  const int tid = threadIdx.x;
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (n >> 1); t += WG) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const bool desc = ((i & size) == 0);
        const ulonglong2 a = e[i], c = e[j];
        const bool a_lt_c = a.x < c.x || (a.x == c.x && a.y < c.y);
        const bool a_gt_c = a.x > c.x || (a.x == c.x && a.y > c.y);
        if (desc ? a_lt_c : a_gt_c) {
          e[i] = c;
          e[j] = a;
        }
      }
      __syncthreads();
    }
  }
}

__device__ inline int next_pow2(int x) {
  int p = 2;
  while (p < x) p <<= 1;
  return p;
}

// ---------------------------------------------------------------------------------------------
// General unit kernel: global-memory table, ordered rounds.
// ---------------------------------------------------------------------------------------------
__device__ inline int lower_bound_u32(const uint32_t *a, int n, uint32_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

struct TableSrc {
  const int64_t *keys;
  const double *dot;  // holds score_key bits after finalize
  const double *nsq;  // 1.0 = candidate, 0.0 = not
  int n;
  __device__ int size() const { return n; }
  __device__ bool get(int i, uint64_t &hi, uint64_t &lo) const {
    if (nsq[i] != 1.0) return false;
    hi = f64_bits(dot[i]);
    lo = id_key(keys[i]);
    return true;
  }
};

__global__ __launch_bounds__(WG) void unit_general_kernel(IndexView ix, BatchView b, GeneralWs ws) {
  __shared__ unsigned s_hist[256];
  __shared__ int s_ctl[4];
  __shared__ int s_special, s_cnt, s_unique;

  const int tid = threadIdx.x;
  const int blk = blockIdx.x;
  const int unit = ws.units ? ws.units[blk] : blk;
  const int q = unit >> ix.log2P;
  const int p = unit & (ix.P - 1);
  const QueryHdr h = b.hdr[q];
  const uint32_t S = ws.ws_slots[blk];
  int64_t *keys = ws.keys + ws.ws_off[blk];
  double *dot = ws.dot + ws.ws_off[blk];
  double *nsq = ws.nsq + ws.ws_off[blk];

  if (tid == 0) { s_special = 0; s_cnt = 0; s_unique = 0; }
  __syncthreads();

  // ApproximateCosineSimilarity.scala:83-100, clusters in the prepared (accumulation) order
  for (int c = 0; c < h.n_scan; c++) {
    const int row = b.scan_row[h.scan_begin + c];
    const double w = b.scan_w[h.scan_begin + c];
    const uint32_t base = ix.sub_offsets[(int64_t)row * ix.P + p];
    const uint32_t end = ix.sub_offsets[(int64_t)row * ix.P + p + 1];
    // postings with rank < M form a prefix of the sub-list (:87)
    const int len = lower_bound_u32(ix.ranks + base, (int)(end - base), (uint32_t)h.M);
    for (int j = tid; j < len; j += WG) {
      const Posting pst = ix.postings[base + j];
      const int64_t id = pst.id;
      const double s = pst.score;
      if (h.excl_enabled && id == h.src_excl) continue;   // :90
      if (id < h.earliest || id > h.latest) continue;     // :91
      const double nrm = h.use_norms ? ix.norms[base + j] : 0.0;
      if (h.use_norms && !(nrm > 0.0)) continue;          // tweets_ann.sql:14  HAVING norm > 0.0
      uint32_t slot;
      bool fresh = false;
      if (id == kEmptyKey) {
        slot = S;
        if (!s_special) { s_special = 1; fresh = true; }  // ids are unique within a list: one thread per round
      } else {
        slot = tweet_slot(mix64((uint64_t)id)) & (S - 1);
        for (;;) {
          unsigned long long old = atomicCAS((unsigned long long *)&keys[slot], (unsigned long long)kEmptyKey,
                                             (unsigned long long)id);
          if (old == (unsigned long long)kEmptyKey) { fresh = true; break; }
          if (old == (unsigned long long)id) break;
          slot = (slot + 1) & (S - 1);
        }
      }
      double d0 = fresh ? 0.0 : dot[slot];  // getOrElse(tweetId, 0.0)
      double n0 = fresh ? 0.0 : nsq[slot];
      dot[slot] = d0 + s * w;               // :92-94
      nsq[slot] = h.use_norms ? nrm : n0 + s * s;  // :95-96 ; offline forms: the tweet's full norm (tweets_ann.sql:50-51)
    }
    __syncthreads();
  }

  // normalise + filter (:105-125); turn the table into (score_key, candidate flag)
  int uniq = 0;
  for (uint32_t i = tid; i <= S; i += WG) {
    bool occ = (i < S) ? (keys[i] != kEmptyKey) : (s_special != 0);
    double flag = 0.0;
    if (occ) {
      uniq++;
      double sc = normalise(h.alg, dot[i], nsq[i], h.l2norm, h.lognorm);
      if (sc >= h.min_score) {
        dot[i] = bits_f64(score_key(sc));
        flag = 1.0;
      }
    }
    nsq[i] = flag;
  }
  if (uniq) atomicAdd(&s_unique, uniq);
  __syncthreads();

  TableSrc src{keys, dot, nsq, (int)S + 1};
  uint64_t thi, tlo;
  int n_valid;
  const int kk = h.k < b.cap2 ? h.k : b.cap2;
  wg_select_threshold(src, kk, s_hist, s_ctl, thi, tlo, n_valid);
  const int64_t obase = (int64_t)blk * b.cap2;
  for (int i = tid; i <= (int)S; i += WG) {
    uint64_t hi, lo;
    if (src.get(i, hi, lo) && key_ge(hi, lo, thi, tlo)) {
      int o = atomicAdd(&s_cnt, 1);
      if (o < b.cap2) {
        b.cand_key2[obase + o] = hi;
        b.cand_id2[obase + o] = (i == (int)S) ? kEmptyKey : keys[i];
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    b.unit_fb[unit] = blk;
    b.cand_cnt[unit] = s_cnt < b.cap2 ? s_cnt : b.cap2;
    b.unit_unique[unit] = s_unique;
    const bool trunc = n_valid > s_cnt;
    b.unit_flags[unit] = trunc ? UNIT_TRUNCATED : UNIT_OK;
    b.unit_thr[2 * (int64_t)unit] = trunc ? thi : 0;
    b.unit_thr[2 * (int64_t)unit + 1] = trunc ? tlo : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// Merge: exact top-k over P unit lists of one query.
// ---------------------------------------------------------------------------------------------
__device__ inline void unit_list(const BatchView &b, int64_t unit, const uint64_t *&key, const int64_t *&id) {
  const int fb = b.unit_fb[unit];
  if (fb < 0) {
    key = b.cand_key + unit * b.cap;
    id = b.cand_id + unit * b.cap;
  } else {
    key = b.cand_key2 + (int64_t)fb * b.cap2;
    id = b.cand_id2 + (int64_t)fb * b.cap2;
  }
}

// Select (radix) + sort (LDS bitonic) + write.  Returns the k-th key through xk (0,0 = none).
template <class Src>
__device__ void merge_select_sort_write(const Src &src, int k, int64_t *out_ids, double *out_scores, int32_t *out_count,
                                        uint64_t *s_hi, uint64_t *s_lo, unsigned *s_hist, int *s_ctl, int *s_cnt,
                                        uint64_t &xk_hi, uint64_t &xk_lo) {
  const int tid = threadIdx.x;
  int n_valid;
  wg_select_threshold(src, k, s_hist, s_ctl, xk_hi, xk_lo, n_valid);
  if (n_valid < k) { xk_hi = 0; xk_lo = 0; }
  if (tid == 0) *s_cnt = 0;
  __syncthreads();
  const int n = src.size();
  for (int i = tid; i < n; i += WG) {
    uint64_t hi, lo;
    if (k > 0 && src.get(i, hi, lo) && key_ge(hi, lo, xk_hi, xk_lo)) {
      int o = atomicAdd(s_cnt, 1);
      if (o < KMAX) { s_hi[o] = hi; s_lo[o] = lo; }
    }
  }
  __syncthreads();
  int cnt = *s_cnt < KMAX ? *s_cnt : KMAX;
  int np = next_pow2(cnt);
  for (int i = cnt + tid; i < np; i += WG) { s_hi[i] = 0; s_lo[i] = 0; }
  __syncthreads();
  bitonic_sort_desc(s_hi, s_lo, np);
  for (int i = tid; i < cnt; i += WG) {
    out_ids[i] = key_id(s_lo[i]);
    out_scores[i] = key_score(s_hi[i]);
  }
  if (tid == 0) *out_count = cnt;
}

__device__ inline uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint64_t o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ inline uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint64_t o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// 128-bit radix threshold over LDS arrays: finds thr with  need <= #{key >= thr} <= budget
// (requires n >= need; keys are unique).  Adaptive: before every digit the min and max of the
// keys still in play are recomputed and the digit is taken at their highest differing bit, so
// long common prefixes (the near-tie regime: hundreds of scores equal to the last few ulps) cost
// nothing.  Scores first; entries that tie on the whole score key are separated by the id key.
// s_ctl: 4 ints, s_mm: 2 u64, s_hist: 256 uints (all LDS).
__device__ void lds_radix_cut(const uint64_t *hi, const uint64_t *lo, int n, int need, int budget, unsigned *s_hist,
                              int *s_ctl, uint64_t *s_mm, uint64_t &thr_hi, uint64_t &thr_lo) {
  const int tid = threadIdx.x;
  // keys in play: word 0: (hi & mask) == (pre_hi & mask); word 1: hi == pre_hi && (lo & mask) == (pre_lo & mask)
  uint64_t pre_hi = 0, pre_lo = 0, mask = 0;  // mask = bits already fixed in the active word
  int word = 0;
  for (int iter = 0; iter < 40; iter++) {
    uint64_t kmin = ~0ull, kmax = 0ull;
    for (int i = tid; i < n; i += WG) {
      const uint64_t h = hi[i];
      const bool in = word == 0 ? ((h & mask) == (pre_hi & mask)) : (h == pre_hi && ((lo[i] & mask) == (pre_lo & mask)));
      const uint64_t k = word == 0 ? h : lo[i];
      kmin = (in && k < kmin) ? k : kmin;
      kmax = (in && k > kmax) ? k : kmax;
    }
    kmin = wave_min_u64(kmin);
    kmax = wave_max_u64(kmax);
    if (tid == 0) { s_mm[0] = ~0ull; s_mm[1] = 0ull; }
    __syncthreads();
    if ((tid & 63) == 0) {
      atomicMin((unsigned long long *)&s_mm[0], (unsigned long long)kmin);
      atomicMax((unsigned long long *)&s_mm[1], (unsigned long long)kmax);
    }
    __syncthreads();
    const uint64_t gmin = s_mm[0], gmax = s_mm[1];
    const uint64_t diff = gmin ^ gmax;
    __syncthreads();
    if (diff == 0) {
      // everything in play agrees on this whole word
      if (word == 0) { pre_hi = gmax; word = 1; mask = 0; continue; }
      pre_lo = gmax;
      break;
    }
    const int hbit = 63 - __clzll((long long)diff);
    const int shift = hbit - 7 < 0 ? 0 : hbit - 7;
    const int width = hbit - shift + 1;
    // all keys in play share the bits above hbit: fix them
    const uint64_t above = (hbit == 63) ? 0ull : (~0ull << (hbit + 1));
    if (word == 0) pre_hi = gmax & above; else pre_lo = gmax & above;
    mask = above;
    for (int i = tid; i < 256; i += WG) s_hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += WG) {
      const uint64_t h = hi[i];
      const bool in = word == 0 ? ((h & mask) == (pre_hi & mask)) : (h == pre_hi && ((lo[i] & mask) == (pre_lo & mask)));
      const uint64_t k = word == 0 ? h : lo[i];
      if (in) atomicAdd(&s_hist[(unsigned)((k >> shift) & ((1u << width) - 1))], 1u);
    }
    __syncthreads();
    if (tid < 64) wave_find_digit(s_hist, need, &s_ctl[1]);
    __syncthreads();
    const int d = s_ctl[1], A = s_ctl[2], B = s_ctl[3];
    const uint64_t dig = (uint64_t)d << shift;
    if (word == 0) pre_hi |= dig; else pre_lo |= dig;
    mask = (shift == 0) ? ~0ull : (~0ull << shift);
    __syncthreads();
    if (A + B <= budget) break;  // everything >= the prefix is between need and budget entries
    // digit d alone holds too many: the A entries above it are in, recurse into d
    need -= A;
    budget -= A;
    if (shift == 0) {
      // the B entries tie on this whole word
      if (word == 1) break;  // cannot happen: keys are unique
      word = 1;
      mask = 0;
    }
  }
  thr_hi = pre_hi;
  thr_lo = pre_lo;
}

// Number of entries of the descending 64-entry run L that come BEFORE the key (xh, xl): strictly greater ones, and -- when
// ties_before -- equal ones (only padding entries can be equal; the flag makes the order total).
__device__ inline int run_count_before(const ulonglong2 *L, uint64_t xh, uint64_t xl, bool ties_before) {
  int pos = 0;
#pragma unroll
  for (int step = 32; step >= 1; step >>= 1) {
    const ulonglong2 v = L[pos + step - 1];
    const bool before = v.x > xh || (v.x == xh && (v.y > xl || (v.y == xl && ties_before)));
    pos += before ? step : 0;
  }
  const ulonglong2 v = L[63];  // 64 entries = 63 reachable by the steps above, plus the last one
  const bool before = v.x > xh || (v.x == xh && (v.y > xl || (v.y == xl && ties_before)));
  return pos + ((pos == 63 && before) ? 1 : 0);
}

// ... in a descending run of L entries, L a power of two (the pairwise merge rounds of the final sort)
__device__ inline int run_count_before_n(const ulonglong2 *run, int L, uint64_t xh, uint64_t xl, bool ties_before) {
  int pos = 0;
  for (int step = L >> 1; step >= 1; step >>= 1) {
    const ulonglong2 v = run[pos + step - 1];
    const bool before = v.x > xh || (v.x == xh && (v.y > xl || (v.y == xl && ties_before)));
    pos += before ? step : 0;
  }
  const ulonglong2 v = run[L - 1];
  const bool before = v.x > xh || (v.x == xh && (v.y > xl || (v.y == xl && ties_before)));
  return pos + ((pos == L - 1 && before) ? 1 : 0);
}

// The same for all the other runs of a sorted-runs array at once: the searches advance in lock step, so that every
// step has one LDS read per run in flight instead of a chain of 7 x (runs - 1) dependent reads.
template <int RMAX>
__device__ inline int rank_among_runs(const ulonglong2 *runs, int R, int r, int lane, uint64_t xh, uint64_t xl) {
  int pos[RMAX];
#pragma unroll
  for (int r2 = 0; r2 < RMAX; r2++) pos[r2] = 0;
#pragma unroll
  for (int step = 32; step >= 1; step >>= 1) {
    ulonglong2 v[RMAX];
#pragma unroll
    for (int r2 = 0; r2 < RMAX; r2++) v[r2] = r2 < R ? runs[r2 * 64 + pos[r2] + step - 1] : make_ulonglong2(0ull, 0ull);
#pragma unroll
    for (int r2 = 0; r2 < RMAX; r2++) {
      const bool before = v[r2].x > xh || (v[r2].x == xh && (v[r2].y > xl || (v[r2].y == xl && r2 < r)));
      pos[r2] += before ? step : 0;
    }
  }
  int rank = lane;
#pragma unroll
  for (int r2 = 0; r2 < RMAX; r2++) {
    const ulonglong2 v = r2 < R ? runs[r2 * 64 + 63] : make_ulonglong2(0ull, 0ull);
    const bool before = v.x > xh || (v.x == xh && (v.y > xl || (v.y == xl && r2 < r)));
    const int p2 = pos[r2] + ((pos[r2] == 63 && before) ? 1 : 0);
    rank += (r2 < R && r2 != r) ? p2 : 0;
  }
  return rank;
}

// A threshold with need <= #{key >= thr} <= budget from a SAMPLE instead of radix passes: every thread contributes one
// staged entry, the WG = 256 samples are sorted (four in-register wave sorts + a rank merge: two barriers), and the
// sample at the position the window's middle is expected at is tried -- one counting sweep per try, the next try moved
// by the miss.  Returns false (after at most 6 tries, or when two neighbouring samples bracket the window) and leaves
// the decision to lds_radix_cut; needs n >= WG.  s_tmp: WG entries, s_ctl: 4 ints.
__device__ bool sample_cut(const uint64_t *hi, const uint64_t *lo, int n, int need, int budget, ulonglong2 *s_tmp, int *s_ctl,
                           uint64_t &thr_hi, uint64_t &thr_lo) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // A window that is wide against n (a shard's merge: need = its cut list length, budget = 256 of ~380 staged) is hit
  // from 64 samples, which one wave sorts on its own: no rank merge, and the other waves go straight to the sweep.
  const int NS = (budget - need) * 8 >= n ? 64 : WG;  // (uniform; a 112-entry window in ~800 staged: 1/7 of the range, 9 samples wide)
  if (NS == 64) {
    if (wv == 0) {
      const int idx = (int)(((long long)lane * n) / 64);
      uint64_t sh = hi[idx], sl = lo[idx];
      wave_sort_desc_k128(sh, sl);
      s_tmp[lane] = make_ulonglong2(sh, sl);
    }
    __syncthreads();
  } else {
    const int idx = (int)(((long long)tid * n) / WG);
    uint64_t sh = hi[idx], sl = lo[idx];
    wave_sort_desc_k128(sh, sl);
    s_tmp[tid] = make_ulonglong2(sh, sl);
    __syncthreads();
    const int rank = rank_among_runs<WG / 64>(s_tmp, WG / 64, wv, lane, sh, sl);
    __syncthreads();
    s_tmp[rank] = make_ulonglong2(sh, sl);  // (ranks are a permutation)
    __syncthreads();
  }
  const int target = (need + budget) / 2;
  int j = (int)(((long long)target * NS) / n) - 1;
  j = j < 0 ? 0 : (j > NS - 1 ? NS - 1 : j);
  int j_small = -1, j_big = NS;  // samples known to give too few / too many
  for (int attempt = 0; attempt < 6; attempt++) {
    const ulonglong2 t = s_tmp[j];
    int c = 0;
    for (int i = tid; i < n; i += WG) c += key_ge(hi[i], lo[i], t.x, t.y) ? 1 : 0;
    const int tot = __builtin_amdgcn_readlane(wave_incl_scan_i32(c), 63);
    if (lane == 0) s_ctl[wv] = tot;
    __syncthreads();
    const int C = s_ctl[0] + s_ctl[1] + s_ctl[2] + s_ctl[3];
    __syncthreads();
    if (C >= need && C <= budget) {
      thr_hi = t.x;
      thr_lo = t.y;
      return true;
    }
    if (C < need) j_small = j; else j_big = j;
    if (j_big - j_small <= 1) return false;
    int step = (int)(((long long)(target - C) * NS) / n);
    if (step == 0) step = C < need ? 1 : -1;
    j += step;
    j = j <= j_small ? j_small + 1 : (j >= j_big ? j_big - 1 : j);
  }
  return false;
}

// SURV = capacity of the survivor list: 512 when every k of the batch is <= 448, else 1024.  The staging
// area holds 1728 entries for SURV = 512: 39.8 KB of LDS in all, FOUR workgroups per CU, so a 1024-query batch
// merges in one round of workgroups (at 2048 entries it was 44 KB, three per CU, two rounds: twice the time).
// MERGE_LDS = entries staged per tournament round (>= SURV + any per-unit capacity): 1728 for the 32-partition
// benchmark shape; 768 when the index has <= 8 partitions per cluster and k <= 256 (sharded runs: ~500 candidates
// per query in all) -- 24 KB of LDS instead of 40, six workgroups per CU instead of four.
template <int SURV, int MERGE_LDS>
__global__ __launch_bounds__(WG, 4) void merge_kernel(IndexView ix, BatchView b, const int32_t *query_list, const int32_t *skip_done) {
  __shared__ uint64_t s_hi[MERGE_LDS], s_lo[MERGE_LDS];
  __shared__ uint8_t s_umap[MERGE_LDS];  // new entry -> unit (relative to the round's first unit; P <= 256)
  __shared__ ulonglong2 s_e2[SURV];  // survivors, packed {score key, id key}
  __shared__ uint64_t s_mm[2];
  __shared__ unsigned s_hist[256];
  __shared__ int s_off[WG + 1];
  __shared__ int s_fb[WG];
  __shared__ int s_ctl[4];

  const int tid = threadIdx.x;
  const int q = query_list ? query_list[blockIdx.x] : blockIdx.x;
  if (skip_done && skip_done[q]) return;  // (uniform) merge_wave_kernel has finished this query
  const QueryHdr h = b.hdr[q];
  const int P = ix.P;  // <= 256
  const int64_t unit0 = (int64_t)q * P;
  const int k = h.k;
  const int budget = SURV < 512 ? SURV : (k <= 448 || SURV < KMAX) ? 512 : KMAX;  // survivors kept between chunks / sorted at the end
  // debug stamps live after the units' region of the prof buffer
#define MSTAMP(i) do { if (b.prof && tid == 0) b.prof[((int64_t)b.nq * P + q) * 16 + (i)] = (unsigned long long)clock64(); } while (0)
  MSTAMP(0);

  // offsets of the unit lists in a flat index space (P <= 256: one thread per unit, wave scans)
  {
    int c = tid < P ? b.cand_cnt[unit0 + tid] : 0;
    s_fb[tid] = tid < P ? b.unit_fb[unit0 + tid] : -1;
    int incl = c;
    const int lane = tid & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) s_ctl[tid >> 6] = incl;  // wave totals (WG/64 = 4 waves)
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < (tid >> 6); w++) wbase += s_ctl[w];
    s_off[tid] = wbase + incl - c;
    if (tid == WG - 1) s_off[WG] = wbase + incl;
    __syncthreads();
  }
  const int out_chunk = q / b.out_chunk_q, ql = q - out_chunk * b.out_chunk_q;
  const int64_t out_shift = out_chunk * b.out_chunk_pitch;  // bytes
  int64_t *out_ids = (int64_t *)((char *)b.out_ids + out_shift) + (int64_t)ql * b.stride;
  double *out_scores = (double *)((char *)b.out_scores + out_shift) + (int64_t)ql * b.stride;

  // what the exactness proof at the end reads per unit (P <= WG: one unit per thread), fetched now so that the two
  // dependent trips to memory are long over when the sort is
  int pf_unique = 0;
  uint32_t pf_T = 0, pf_flags = 0;
  uint64_t pf_thi = 0, pf_tlo = 0;
  if (tid < P) {
    const int64_t unit = unit0 + tid;
    pf_unique = b.unit_unique[unit];
    pf_T = b.q_stat ? (uint32_t)b.unit_T[unit] : 0u;
    pf_flags = b.unit_flags[unit];
    pf_thi = b.unit_thr[2 * unit];
    pf_tlo = b.unit_thr[2 * unit + 1];
  }
  MSTAMP(1);  // offsets
  int best_n = 0;  // entries currently in s_e2
  int u_begin = 0;
  bool bad_handover = false;  // this thread met a handed-over candidate that points outside the index / the query
  while (u_begin < P) {
    // units [u_begin, u_end) such that best + their entries fit (a single list always fits:
    // MERGE_LDS - SURV >= any per-unit capacity)
    int u_end = u_begin;
    const int base_off = s_off[u_begin];
    while (u_end < P && best_n + (s_off[u_end + 1] - base_off) <= MERGE_LDS) u_end++;
    if (u_end == u_begin) u_end = u_begin + 1;  // cannot happen given the capacities; keeps progress
    const int n_new = s_off[u_end] - base_off;
    const int n = best_n + n_new;
    for (int i = tid; i < best_n; i += WG) { const ulonglong2 v = s_e2[i]; s_hi[i] = v.x; s_lo[i] = v.y; }
    // flat index of a new entry -> its unit: eight threads per list fill a byte map, so that the gather
    // below needs one LDS read per entry and can put all of a thread's global loads in flight together
    // (a per-entry search made each load wait for the previous one: six dependent round trips)
    for (int t = tid; t < 8 * (u_end - u_begin); t += WG) {
      const int u = u_begin + (t >> 3);
      for (int i = s_off[u] - base_off + (t & 7); i < s_off[u + 1] - base_off && i < MERGE_LDS; i += 8)
        s_umap[i] = (uint8_t)(u - u_begin);
    }
    __syncthreads();
    {
      constexpr int R = (MERGE_LDS + WG - 1) / WG;
      uint64_t kh[R];
      int64_t kid[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int i = r * WG + tid;
        kh[r] = CAND_DROPPED;
        kid[r] = 0;
        if (i < n_new && i < MERGE_LDS) {
          const int u = u_begin + s_umap[i];
          const int fb = s_fb[u];
          const int j = base_off + i - s_off[u];
          const uint64_t *key = fb < 0 ? b.cand_key + (unit0 + u) * b.cap : b.cand_key2 + (int64_t)fb * b.cap2;
          const int64_t *id = fb < 0 ? b.cand_id + (unit0 + u) * b.cap : b.cand_id2 + (int64_t)fb * b.cap2;
          kh[r] = key[j];
          kid[r] = id[j];
        }
      }
      // Entries a fast unit handed over as (cluster, posting position) (sann_device.h, CAND_DEFERRED): the posting and the
      // cluster's weight are fetched here -- all of a thread's entries in one trip -- and scored exactly as
      // ApproximateCosineSimilarity.scala:92-96,111-125 does for a tweet met in one cluster.
      // (four entries at a time: with all seven in flight the kernel needed 176 registers and lost half its occupancy)
      constexpr int RC = 4;
#pragma unroll
      for (int r0 = 0; r0 < R; r0 += RC) {
        Posting ps[RC];
        double wq[RC];
#pragma unroll
        for (int rr = 0; rr < RC; rr++) {
          const int r = r0 + rr;
          ps[rr] = Posting{0, 0.0};
          wq[rr] = 0.0;
          if (r < R && kh[r] == CAND_DEFERRED) {
            // (posting position, cluster sequence number) were written by the unit kernel: checked against the index and
            // the query before they address anything -- a corrupted hand-over becomes an unproven query (re-answered
            // exactly by the general path), never a faulting load
            if ((uint32_t)kid[r] < ix.n_postings && (uint32_t)((uint64_t)kid[r] >> 32) < (uint32_t)h.n_scan) {
              ps[rr] = ix.postings[(uint32_t)kid[r]];
              wq[rr] = b.scan_w[h.scan_begin + (int)((uint64_t)kid[r] >> 32)];
            } else {
              kh[r] = CAND_DROPPED;
              bad_handover = true;
            }
          }
        }
#pragma unroll
        for (int rr = 0; rr < RC; rr++) {
          const int r = r0 + rr;
          if (r < R && kh[r] == CAND_DEFERRED) {
            const double dot = 0.0 + ps[rr].score * wq[rr];  // getOrElse(tweetId, 0.0) + score * sourceClusterScore  (:92-94)
            double nsq = 0.0 + ps[rr].score * ps[rr].score;  // (:95-96)
            if (h.use_norms) nsq = ix.norms[(uint32_t)kid[r]];  // tweets_ann.sql:50-51 (offline forms: one more trip)
            const double v = normalise(h.alg, dot, nsq, h.l2norm, h.lognorm);
            kh[r] = v >= h.min_score ? score_key(v) : CAND_DROPPED;  // :125 (false for NaN)
            kid[r] = ps[rr].id;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int i = r * WG + tid;
        if (i < n_new && best_n + i < MERGE_LDS) {
          // (a dropped entry is staged as the all-zero key, below every real one, and never leaves the staging area)
          const bool real = kh[r] != CAND_DROPPED;
          s_hi[best_n + i] = real ? kh[r] : 0ull;
          s_lo[best_n + i] = real ? id_key(kid[r]) : 0ull;
        }
      }
    }
    __syncthreads();
    MSTAMP(2);  // staged (last round)
    uint64_t thi = 0, tlo = 0;
    if (n > budget && k > 0) {
      // (s_e2 is free here: the previous round's survivors were copied into the staging area above)
      if (!sample_cut(s_hi, s_lo, n, k, budget, s_e2, s_ctl, thi, tlo))
        lds_radix_cut(s_hi, s_lo, n, k, budget, s_hist, s_ctl, s_mm, thi, tlo);
    }
    MSTAMP(3);  // cut found (last round)
    // survivors into s_e2: per-thread counts, a DPP prefix sum per wave, wave bases through LDS -- no atomics
    {
      int c = 0;
      if (k > 0)
        for (int i = tid; i < n; i += WG) c += (s_hi[i] != 0ull && key_ge(s_hi[i], s_lo[i], thi, tlo)) ? 1 : 0;
      const int incl = wave_incl_scan_i32(c);
      if ((tid & 63) == 63) s_ctl[tid >> 6] = incl;
      __syncthreads();
      int o = incl - c;
      for (int w = 0; w < (tid >> 6); w++) o += s_ctl[w];
      const int total = s_ctl[0] + s_ctl[1] + s_ctl[2] + s_ctl[3];
      if (k > 0)
        for (int i = tid; i < n; i += WG) {
          const uint64_t a = s_hi[i], c2 = s_lo[i];
          if (a != 0ull && key_ge(a, c2, thi, tlo)) {
            if (o < SURV) s_e2[o] = make_ulonglong2(a, c2);
            o++;
          }
        }
      best_n = total < SURV ? total : SURV;
    }
    u_begin = u_end;
    __syncthreads();
  }

  MSTAMP(4);  // compacted
  // sort the survivors, keep the first k: every wave sorts runs of 64 in registers (DPP / permlane network, no barriers);
  // the runs are then merged pairwise -- an entry's place in the merged pair is its offset in its own run plus the
  // number of the partner run's entries that come before it, one binary search -- doubling the run length per round
  // (two barriers each), and the last round writes straight to the output.  (An LDS bitonic sort of 512 took 45
  // barrier-separated stages, 32 k clk; ranking every entry against ALL other runs at once was bound by LDS bandwidth,
  // 24 k clk with four queries per CU.)
  uint64_t xk_hi = 0, xk_lo = 0;
  // (the sorted entries go to the output at the very END of the kernel: a barrier behind global stores waits for them)
  constexpr int PER_OUT = SURV / WG < 1 ? 1 : SURV / WG;
  int o_at[PER_OUT], cnt_out = 0;
  uint64_t o_hi[PER_OUT], o_lo[PER_OUT];
  {
    int np = next_pow2(best_n);
    np = np < 64 ? 64 : np;
    for (int i = best_n + tid; i < np; i += WG) s_e2[i] = make_ulonglong2(0ull, 0ull);
    if (tid == 0) { s_mm[0] = 0ull; s_mm[1] = 0ull; }
    __syncthreads();
    const int R = np >> 6, lane = tid & 63, wv = tid >> 6;
    constexpr int PER = SURV / WG < 1 ? 1 : SURV / WG;
    uint64_t mh[PER], ml[PER];
    int at[PER];  // the entry's current position in s_e2
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const int r = wv + j * (WG / 64);
      mh[j] = 0ull;
      ml[j] = 0ull;
      at[j] = r * 64 + lane;
      if (r < R) {  // (uniform per wave)
        const ulonglong2 v = s_e2[at[j]];
        mh[j] = v.x;
        ml[j] = v.y;
        wave_sort_desc_k128(mh[j], ml[j]);
        s_e2[at[j]] = make_ulonglong2(mh[j], ml[j]);
      }
    }
    __syncthreads();
    MSTAMP(5);  // runs sorted
    for (int L = 64; L < np; L <<= 1) {  // (uniform)
      // the thread's entries search their partner runs in lock step: PER independent chains of LDS reads, not one after
      // the other (the chains are all this phase waits for)
      const ulonglong2 *run[PER];
      int pos[PER];
      bool second[PER];
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const int base = at[j] & ~(2 * L - 1);
        second[j] = (at[j] & L) != 0;  // in the pair's second run: equal keys of the first run come before
        run[j] = s_e2 + base + (second[j] ? 0 : L);
        pos[j] = 0;
      }
      for (int step = L >> 1; step >= 1; step >>= 1) {
        ulonglong2 v[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) v[j] = run[j][pos[j] + step - 1];
#pragma unroll
        for (int j = 0; j < PER; j++) {
          const bool before = v[j].x > mh[j] || (v[j].x == mh[j] && (v[j].y > ml[j] || (v[j].y == ml[j] && second[j])));
          pos[j] += before ? step : 0;
        }
      }
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const ulonglong2 v = run[j][L - 1];
        const bool before = v.x > mh[j] || (v.x == mh[j] && (v.y > ml[j] || (v.y == ml[j] && second[j])));
        const int cntb = pos[j] + ((pos[j] == L - 1 && before) ? 1 : 0);
        if (wv + j * (WG / 64) < R) at[j] = (at[j] & ~(2 * L - 1)) + (at[j] & (L - 1)) + cntb;
      }
      if (2 * L >= np) break;  // merged completely: at[] is the rank
      __syncthreads();
#pragma unroll
      for (int j = 0; j < PER; j++)
        if (wv + j * (WG / 64) < R) s_e2[at[j]] = make_ulonglong2(mh[j], ml[j]);
      __syncthreads();
    }
    cnt_out = best_n < k ? best_n : k;
#pragma unroll
    for (int j = 0; j < PER; j++) {
      o_at[j] = (wv + j * (WG / 64) < R && at[j] < cnt_out) ? at[j] : -1;
      o_hi[j] = mh[j];
      o_lo[j] = ml[j];
      if (o_at[j] == cnt_out - 1 && cnt_out == k) { s_mm[0] = mh[j]; s_mm[1] = ml[j]; }  // the k-th key
    }
    __syncthreads();
    if (cnt_out == k && cnt_out > 0) { xk_hi = s_mm[0]; xk_lo = s_mm[1]; }
  }
  // candidateScoresMap.size (:102) and the exactness proof: every candidate a unit withheld has
  // key < unit_thr; it cannot belong to the top-k iff unit_thr <= the k-th key.  With fewer than
  // k results there is no k-th key, so any withholding unit makes the result unproven.
  int msz = 0, inexact = 0;
  uint32_t t_max = 0, t_sum = 0;
  if (tid < P) {
    msz = pf_unique;
    t_max = pf_T;
    t_sum = pf_T;
    if (k > 0 && (pf_flags & UNIT_TRUNCATED) && key_gt(pf_thi, pf_tlo, xk_hi, xk_lo)) inexact = 1;
  }
  if (tid == 0) { s_ctl[0] = 0; s_ctl[1] = 0; s_ctl[2] = 0; s_ctl[3] = 0; }
  __syncthreads();
  if (msz) atomicAdd(&s_ctl[0], msz);
  if (inexact || bad_handover) atomicOr(&s_ctl[1], 1);
  if (t_sum) {
    atomicMax((unsigned *)&s_ctl[2], t_max);
    atomicAdd((unsigned *)&s_ctl[3], t_sum);
  }
  __syncthreads();
  if (tid == 0) {
    if (b.q_stat) b.q_stat[q] = make_uint4((unsigned)s_ctl[2], (unsigned)s_ctl[3], (unsigned)h.n_scan, 0u);
    ((int32_t *)((char *)b.out_map_sizes + out_shift))[ql] = s_ctl[0];
    if (s_ctl[1]) {
      int o = atomicAdd(&b.status[1], 1);
      b.status[2 + o] = q;  // status[2..] = list of inexact queries
    }
  }
#pragma unroll
  for (int j = 0; j < PER_OUT; j++)
    if (o_at[j] >= 0) {
      out_ids[o_at[j]] = key_id(o_lo[j]);
      out_scores[o_at[j]] = key_score(o_hi[j]);
    }
  if (tid == 0) ((int32_t *)((char *)b.out_counts + out_shift))[ql] = cnt_out;
  MSTAMP(6);  // proof + written
#undef MSTAMP
}

// The same merge for SMALL queries, one WAVE per query: a shard of an N-GPU run answers N times as many queries, each from
// a few units (P <= 8) with a couple of hundred candidates in all -- too little for 256 threads and nine barriers (the
// workgroup kernel took 131 us for an 8-GPU shard's 8192 queries, 40 us for the unsharded batch's 1024).  Here the
// candidates live in registers (E per lane), every 64 of them are sorted in registers, the runs meet in the wave's own
// 1 KB x E of LDS and every entry finds its final position by one lock-step search of the other runs: no barrier at
// all, four queries per workgroup, 32 per CU.  A query that does not fit (more than 64 E candidates, or k above that) is
// left to merge_kernel: done[q] says which.  Results, proof and statistics are merge_kernel's, entry for entry.
template <int E, int PMAX>
__global__ __launch_bounds__(WG, (E <= 4 ? 6 : 4)) void merge_wave_kernel(IndexView ix, BatchView b, int32_t *done) {
  __shared__ ulonglong2 s_run[WG / 64][E * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = blockIdx.x * (WG / 64) + wv;
  if (q >= b.nq) return;  // (whole waves; nothing below synchronises across waves)
  const QueryHdr h = b.hdr[q];
  const int P = ix.P;  // <= PMAX
  const int64_t unit0 = (int64_t)q * P;
  const int k = h.k;
  // per unit (lane u < P): list length, list location, and what the proof and the statistics read
  int cnt = 0, fb = -1, pf_unique = 0;
  uint32_t pf_T = 0, pf_flags = 0;
  uint64_t pf_thi = 0, pf_tlo = 0;
  if (lane < P) {
    const int64_t unit = unit0 + lane;
    cnt = b.cand_cnt[unit];
    fb = b.unit_fb[unit];
    pf_unique = b.unit_unique[unit];
    pf_T = b.q_stat ? (uint32_t)b.unit_T[unit] : 0u;
    pf_flags = b.unit_flags[unit];
    pf_thi = b.unit_thr[2 * unit];
    pf_tlo = b.unit_thr[2 * unit + 1];
  }
  const int incl = wave_incl_scan_i32(cnt);  // (lanes >= P hold the total)
  const int n = __builtin_amdgcn_readlane(incl, 63);
  if (n > E * 64 || k > E * 64) {  // (uniform)
    if (lane == 0) done[q] = 0;
    return;
  }
  if (lane == 0) done[q] = 1;
  const int excl = incl - cnt;
  uint64_t kh[E];
  int64_t kid[E];
#pragma unroll
  for (int r = 0; r < E; r++) {
    const int i = r * 64 + lane;
    kh[r] = CAND_DROPPED;
    kid[r] = 0;
    int u = 0;  // the unit whose list holds flat entry i: the number of lists that end at or before it
#pragma unroll
    for (int uu = 0; uu < PMAX; uu++) u += i >= __builtin_amdgcn_readlane(incl, uu) ? 1 : 0;
    u = u < PMAX ? u : PMAX - 1;  // (i >= n)
    // (the shuffles stand outside the branch: ds_bpermute returns 0 for a source lane that is switched off, and in a
    // query's last, partial run of 64 the lanes of its units may be)
    const int ex_u = __shfl(excl, u, 64), fbu = __shfl(fb, u, 64);
    if (i < n) {
      const int j = i - ex_u;
      const uint64_t *key = fbu < 0 ? b.cand_key + (unit0 + u) * b.cap : b.cand_key2 + (int64_t)fbu * b.cap2;
      const int64_t *id = fbu < 0 ? b.cand_id + (unit0 + u) * b.cap : b.cand_id2 + (int64_t)fbu * b.cap2;
      kh[r] = key[j];
      kid[r] = id[j];
    }
  }
  // candidates handed over as (cluster, posting position): fetch and score (as merge_kernel's staging does), four at a time
  bool bad_handover = false;
#pragma unroll
  for (int r0 = 0; r0 < E; r0 += 4) {
    Posting ps[4];
    double wq[4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int r = r0 + rr;
      ps[rr] = Posting{0, 0.0};
      wq[rr] = 0.0;
      if (kh[r] == CAND_DEFERRED) {
        if ((uint32_t)kid[r] < ix.n_postings && (uint32_t)((uint64_t)kid[r] >> 32) < (uint32_t)h.n_scan) {  // (as merge_kernel)
          ps[rr] = ix.postings[(uint32_t)kid[r]];
          wq[rr] = b.scan_w[h.scan_begin + (int)((uint64_t)kid[r] >> 32)];
        } else {
          kh[r] = CAND_DROPPED;
          bad_handover = true;
        }
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int r = r0 + rr;
      if (kh[r] == CAND_DEFERRED) {
        const double dot = 0.0 + ps[rr].score * wq[rr];  // :92-94
        double nsq = 0.0 + ps[rr].score * ps[rr].score;  // :95-96
        if (h.use_norms) nsq = ix.norms[(uint32_t)kid[r]];  // tweets_ann.sql:50-51
        const double v = normalise(h.alg, dot, nsq, h.l2norm, h.lognorm);
        kh[r] = v >= h.min_score ? score_key(v) : CAND_DROPPED;  // :125 (false for NaN)
        kid[r] = ps[rr].id;
      }
    }
    if (r0 + 4 < E) __builtin_amdgcn_sched_barrier(0);
  }
  // sort: runs of 64 in registers, then every entry's rank among all runs (ties -- only dropped entries tie -- by run)
  const int R = (n + 63) >> 6;  // (uniform)
  ulonglong2 *const runs = s_run[wv];
  uint64_t mh[E], ml[E];
  int n_real = 0;
#pragma unroll
  for (int r = 0; r < E; r++) {
    const bool real = kh[r] != CAND_DROPPED;
    mh[r] = real ? kh[r] : 0ull;
    ml[r] = real ? id_key(kid[r]) : 0ull;
    n_real += real ? 1 : 0;
    if (r < R) {
      wave_sort_desc_k128(mh[r], ml[r]);
      runs[r * 64 + lane] = make_ulonglong2(mh[r], ml[r]);
    }
  }
  n_real = __builtin_amdgcn_readlane(wave_incl_scan_i32(n_real), 63);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int cnt_out = n_real < k ? n_real : k;
  const int out_chunk = q / b.out_chunk_q, ql = q - out_chunk * b.out_chunk_q;
  const int64_t out_shift = out_chunk * b.out_chunk_pitch;  // bytes
  int64_t *out_ids = (int64_t *)((char *)b.out_ids + out_shift) + (int64_t)ql * b.stride;
  double *out_scores = (double *)((char *)b.out_scores + out_shift) + (int64_t)ql * b.stride;
  uint64_t xk_hi = 0, xk_lo = 0;  // the k-th key (0, 0 with fewer than k results)
#pragma unroll
  for (int r = 0; r < E; r++) {
    if (r < R) {  // (uniform)
      const int rank = R == 1 ? lane : rank_among_runs<E>(runs, R, r, lane, mh[r], ml[r]);
      if (rank < cnt_out) {
        out_ids[rank] = key_id(ml[r]);
        out_scores[rank] = key_score(mh[r]);
      }
      const unsigned long long at = __ballot(cnt_out == k && k > 0 && rank == k - 1);
      if (at != 0ull) {
        const int src = __ffsll((long long)at) - 1;
        xk_hi = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(mh[r] >> 32), src) << 32) |
                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mh[r], src);
        xk_lo = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(ml[r] >> 32), src) << 32) |
                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ml[r], src);
      }
    }
  }
  // candidateScoresMap.size (:102), the exactness proof and the statistics, as merge_kernel
  const bool inexact = lane < P && k > 0 && (pf_flags & UNIT_TRUNCATED) && key_gt(pf_thi, pf_tlo, xk_hi, xk_lo);
  const unsigned long long any_inexact = __ballot(inexact || bad_handover);
  const int msz = __builtin_amdgcn_readlane(wave_incl_scan_i32(pf_unique), 63);
  const uint32_t t_sum = (uint32_t)__builtin_amdgcn_readlane(wave_incl_scan_i32((int)pf_T), 63);
  const uint32_t t_max = wave_max_u32(pf_T);
  if (lane == 0) {
    if (b.q_stat) b.q_stat[q] = make_uint4(t_max, t_sum, (unsigned)h.n_scan, 0u);
    ((int32_t *)((char *)b.out_map_sizes + out_shift))[ql] = msz;
    ((int32_t *)((char *)b.out_counts + out_shift))[ql] = cnt_out;
    if (any_inexact != 0ull) {
      const int o = atomicAdd(&b.status[1], 1);
      b.status[2 + o] = q;  // status[2..] = list of inexact queries
    }
  }
}

// Per-shard results (already exact per shard) -> global top-k.  Shard s's arrays start pitch
// bytes after shard s-1's (pitch 0 = each array tightly packed shard-major, i.e.
// ids[n_shards][nq][stride], counts[n_shards][nq]).
struct ShardSrc {
  const int64_t *ids;
  const double *scores;
  const int32_t *counts;
  int n_shards, nq, stride, q;
  int64_t pitch_ids, pitch_cnt;  // bytes
  __device__ int size() const { return n_shards * stride; }
  __device__ bool get(int i, uint64_t &hi, uint64_t &lo) const {
    int s = i / stride, j = i - s * stride;
    const int32_t *c = (const int32_t *)((const char *)counts + s * pitch_cnt);
    if (j >= c[q]) return false;
    int64_t o = (int64_t)q * stride + j;
    hi = score_key(((const double *)((const char *)scores + s * pitch_ids))[o]);
    lo = id_key(((const int64_t *)((const char *)ids + s * pitch_ids))[o]);
    return true;
  }
};

// shard_k > 0: every shard list was cut at shard_k entries (sorted, so its last entry is its smallest).  The
// merged top-k is then exact iff no cut list could hide a better candidate, i.e. every list with shard_k
// entries ends at or below the merged k-th key; otherwise the query is counted in *inexact.
__global__ __launch_bounds__(WG) void merge_shards_kernel(int n_shards, int nq, int stride, int64_t pitch,
                                                         const int64_t *ids, const double *scores,
                                                         const int32_t *counts, const int32_t *map_sizes,
                                                         const int32_t *k, int k_all, int shard_k, int out_stride,
                                                         int64_t *out_ids, double *out_scores,
                                                         int32_t *out_counts, int32_t *out_map_sizes, int32_t *inexact) {
  __shared__ uint64_t s_keys[2 * KMAX];
  uint64_t *const s_hi = s_keys, *const s_lo = s_keys + KMAX;
  __shared__ unsigned s_hist[256];
  __shared__ int s_ctl[4];
  __shared__ int s_cnt;
  const int q = blockIdx.x;
  const int64_t pitch_ids = pitch ? pitch : (int64_t)nq * stride * 8;
  const int64_t pitch_cnt = pitch ? pitch : (int64_t)nq * 4;
  ShardSrc src{ids, scores, counts, n_shards, nq, stride, q, pitch_ids, pitch_cnt};
  int kk = k ? k[q] : k_all;
  kk = kk < out_stride ? kk : out_stride;
  uint64_t xh = 0, xl = 0;
  // The lists the library's own merge kernels deliver are SORTED (score desc, id asc): an entry's place in the merged
  // order is then its place in its own list plus, for every other list, the number of entries that come before it -- one
  // binary search per list, all in lock step -- and nothing has to be selected or sorted.  (The general path below --
  // radix passes over global memory, then an LDS bitonic sort -- took 40 us for 1024 queries x 8 lists of 104.)  Taken
  // when there are at most 8 lists with at most KMAX entries in all and every list is found in order.
  constexpr int SMAX = 8;
  __shared__ int s_off[SMAX + 1];
  __shared__ int s_unsorted;
  __shared__ uint64_t s_kth[2];  // the merged k-th key
  bool ranked = false;
  if (n_shards <= SMAX) {  // (uniform)
    const int tid = threadIdx.x;
    if (tid == 0) {
      int run = 0;
      for (int s = 0; s < n_shards; s++) {
        s_off[s] = run;
        const int c = ((const int32_t *)((const char *)counts + s * pitch_cnt))[q];
        run += c < stride ? (c > 0 ? c : 0) : stride;
      }
      for (int s = n_shards; s <= SMAX; s++) s_off[s] = run;
      s_unsorted = 0;
    }
    __syncthreads();
    const int total = s_off[n_shards];
    int off[SMAX + 1];  // (registers: the searches below read them at every step)
#pragma unroll
    for (int t = 0; t <= SMAX; t++) off[t] = s_off[t];
    if (total <= KMAX) {  // (uniform)
      ulonglong2 *const s_e = reinterpret_cast<ulonglong2 *>(s_keys);
      for (int i = tid; i < total; i += WG) {
        int s = 0;
#pragma unroll
        for (int t = 1; t < SMAX; t++) s += i >= off[t] ? 1 : 0;
        int base_s = 0;
#pragma unroll
        for (int t = 1; t < SMAX; t++) base_s = s == t ? off[t] : base_s;
        const int64_t o = (int64_t)q * stride + (i - base_s);
        s_e[i] = make_ulonglong2(score_key(((const double *)((const char *)scores + s * pitch_ids))[o]),
                                 id_key(((const int64_t *)((const char *)ids + s * pitch_ids))[o]));
      }
      __syncthreads();
      for (int i = tid; i + 1 < total; i += WG) {  // every list in order?
        int s = 0;
#pragma unroll
        for (int t = 1; t < SMAX; t++) s += i >= off[t] ? 1 : 0;
        const ulonglong2 a = s_e[i], c = s_e[i + 1];
        int end_s = off[1];
#pragma unroll
        for (int t = 1; t < SMAX; t++) end_s = s == t ? off[t + 1] : end_s;
        if (i + 1 < end_s && !key_gt(a.x, a.y, c.x, c.y)) s_unsorted = 1;
      }
      if (tid == 0) { s_ctl[0] = 0; s_ctl[1] = 0; }
      __syncthreads();
      if (!s_unsorted) {  // (uniform)
        ranked = true;
        const int cnt_out = total < kk ? total : kk;
        int max_len = 0;
#pragma unroll
        for (int t = 0; t < SMAX; t++) max_len = max(max_len, off[t + 1] - off[t]);
        const int top = max_len > 0 ? next_pow2(max_len) : 1;
        for (int i = tid; i < total; i += WG) {
          int s = 0;
#pragma unroll
          for (int t = 1; t < SMAX; t++) s += i >= off[t] ? 1 : 0;
          const ulonglong2 me = s_e[i];
          int pos[SMAX];
#pragma unroll
          for (int t = 0; t < SMAX; t++) pos[t] = 0;
          for (int st = top; st >= 1; st >>= 1) {  // (uniform trip count)
#pragma unroll
            for (int t = 0; t < SMAX; t++) {
              const int base = off[t], len = off[t + 1] - base, idx = pos[t] + st - 1;
              if (idx < len) {
                const ulonglong2 v = s_e[base + idx];
                // (entries of different lists never compare equal: tweets belong to one shard; ties by list anyway)
                const bool before = v.x > me.x || (v.x == me.x && (v.y > me.y || (v.y == me.y && t < s)));
                pos[t] += before ? st : 0;
              }
            }
          }
          int rank = i;
#pragma unroll
          for (int t = 1; t < SMAX; t++) rank = s == t ? i - off[t] : rank;
#pragma unroll
          for (int t = 0; t < SMAX; t++) rank += t != s ? pos[t] : 0;
          if (rank < cnt_out) {
            out_ids[(int64_t)q * out_stride + rank] = key_id(me.y);
            out_scores[(int64_t)q * out_stride + rank] = key_score(me.x);
          }
          if (rank == kk - 1 && cnt_out == kk) { s_kth[0] = me.x; s_kth[1] = me.y; }
        }
        __syncthreads();
        if (cnt_out == kk && kk > 0) { xh = s_kth[0]; xl = s_kth[1]; }
        if (tid == 0) out_counts[q] = cnt_out;
      }
    }
  }
  if (!ranked)
    merge_select_sort_write(src, kk, out_ids + (int64_t)q * out_stride, out_scores + (int64_t)q * out_stride, out_counts + q,
                            s_hi, s_lo, s_hist, s_ctl, &s_cnt, xh, xl);
  if (threadIdx.x == 0) {
    int m = 0, bad = 0;
    for (int s = 0; s < n_shards; s++) {
      m += ((const int32_t *)((const char *)map_sizes + s * pitch_cnt))[q];
      const int c = ((const int32_t *)((const char *)counts + s * pitch_cnt))[q];
      if (shard_k > 0 && c >= shard_k && kk > 0) {
        // the list was (possibly) cut: its smallest delivered key against the merged k-th key (0,0 = fewer than k merged)
        const int64_t o = (int64_t)q * stride + c - 1;
        const uint64_t lh = score_key(((const double *)((const char *)scores + s * pitch_ids))[o]);
        const uint64_t ll = id_key(((const int64_t *)((const char *)ids + s * pitch_ids))[o]);
        if (key_gt(lh, ll, xh, xl)) bad = 1;
      }
    }
    out_map_sizes[q] = m;
    if (bad && inexact) atomicAdd(inexact, 1);
  }
}

// Results -> the caller's pinned host arrays (see sann_kernels.h).  16-byte stores, consecutive lanes = consecutive addresses of
// one row pair; rows of the two big arrays are re-pitched from `stride` to `out_stride` entries.
__global__ __launch_bounds__(256) void copy_out_kernel(int nq, int stride, int out_stride, const int64_t *ids, const double *scores,
                                                       const int32_t *counts, const int32_t *map_sizes, int64_t *h_ids,
                                                       double *h_scores, int32_t *h_counts, int32_t *h_map_sizes) {
  const int64_t n = (int64_t)nq * stride;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (out_stride == stride && (stride & 1) == 0) {
    const ulonglong2 *a = reinterpret_cast<const ulonglong2 *>(ids), *c = reinterpret_cast<const ulonglong2 *>(scores);
    ulonglong2 *ha = reinterpret_cast<ulonglong2 *>(h_ids), *hc = reinterpret_cast<ulonglong2 *>(h_scores);
    for (int64_t i = i0; i < n / 2; i += step) {
      ha[i] = a[i];
      hc[i] = c[i];
    }
  } else {
    for (int64_t i = i0; i < n; i += step) {
      const int64_t q = i / stride, j = i - q * stride;
      h_ids[q * out_stride + j] = ids[i];
      h_scores[q * out_stride + j] = scores[i];
    }
  }
  for (int64_t i = i0; i < nq; i += step) {
    h_counts[i] = counts[i];
    h_map_sizes[i] = map_sizes[i];
  }
}
hipError_t launch_copy_out(int nq, int stride, int out_stride, const int64_t *ids, const double *scores, const int32_t *counts,
                           const int32_t *map_sizes, int64_t *h_ids, double *h_scores, int32_t *h_counts, int32_t *h_map_sizes,
                           hipStream_t stream) {
  if (nq <= 0) return hipSuccess;
  static const int wgs = [] { const char *e = getenv("SANN_COPY_WGS"); const int v = e ? atoi(e) : 16; return v >= 1 && v <= 1024 ? v : 16; }();
  hipLaunchKernelGGL(copy_out_kernel, dim3(wgs), dim3(256), 0, stream, nq, stride, out_stride, ids, scores, counts, map_sizes, h_ids,
                     h_scores, h_counts, h_map_sizes);
  return hipGetLastError();
}

// Audit hook: out[i] = normalise(alg, dot[i], nsq[i], l2norm, lognorm) -- lets a test check the
// device's fp64 division / sqrt / log bit-for-bit against the host.
__global__ void debug_normalise_kernel(int alg, int n, const double *dot, const double *nsq, double l2norm,
                                       double lognorm, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = normalise(alg, dot[i], nsq[i], l2norm, lognorm);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_debug_normalise(int alg, int n, const double *dot, const double *nsq, double l2norm, double lognorm,
                                  double *out, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(debug_normalise_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, alg, n, dot, nsq, l2norm,
                     lognorm, out);
  return hipGetLastError();
}
hipError_t launch_unit_general(const IndexView &ix, const BatchView &b, const GeneralWs &ws, int n_units,
                               hipStream_t stream) {
  if (n_units <= 0) return hipSuccess;
  hipLaunchKernelGGL(unit_general_kernel, dim3(n_units), dim3(WG), 0, stream, ix, b, ws);
  return hipGetLastError();
}
hipError_t launch_merge(const IndexView &ix, const BatchView &b, const int32_t *query_list, int n_queries,
                        hipStream_t stream) {
  if (n_queries <= 0) return hipSuccess;
  // the 512-entry survivor list serves k <= 448; cap2 is the batch's largest k
  const int32_t *none = nullptr;
  // a whole batch of small queries (a shard's): one wave per query first -- up to 256 candidates from <= 8 units, or 512
  // from <= 16 --, then the workgroups for whatever did not fit (normally nothing: they find done[q] set and leave)
  const bool waves = query_list == nullptr && b.merge_done != nullptr && ix.P <= 16 && b.cap2 <= 448 && b.cap <= 256;
  const int32_t *skip = waves ? (const int32_t *)b.merge_done : none;
  if (waves) {
    const dim3 grid((n_queries + WG / 64 - 1) / (WG / 64));
    if (ix.P <= 4 && b.cap2 <= 128)
      hipLaunchKernelGGL((merge_wave_kernel<4, 8>), grid, dim3(WG), 0, stream, ix, b, b.merge_done);
    else
      hipLaunchKernelGGL((merge_wave_kernel<8, 16>), grid, dim3(WG), 0, stream, ix, b, b.merge_done);
  }
  if (ix.P <= 8 && b.cap2 <= 256 && b.cap <= 256)  // a single list (<= 256 entries) fits beside 512 survivors
    hipLaunchKernelGGL((merge_kernel<256, 640>), dim3(n_queries), dim3(WG), 0, stream, ix, b, query_list, skip);
  else if (b.cap2 <= 448)
    hipLaunchKernelGGL((merge_kernel<512, 1728>), dim3(n_queries), dim3(WG), 0, stream, ix, b, query_list, skip);
  else
    hipLaunchKernelGGL((merge_kernel<KMAX, 2048>), dim3(n_queries), dim3(WG), 0, stream, ix, b, query_list, none);
  return hipGetLastError();
}
hipError_t launch_merge_shards(int n_shards, int nq, int stride, int64_t pitch, const int64_t *ids, const double *scores,
                               const int32_t *counts, const int32_t *map_sizes, const int32_t *k, int k_all, int shard_k,
                               int out_stride, int64_t *out_ids, double *out_scores, int32_t *out_counts,
                               int32_t *out_map_sizes, int32_t *inexact, hipStream_t stream) {
  if (nq <= 0) return hipSuccess;
  hipLaunchKernelGGL(merge_shards_kernel, dim3(nq), dim3(WG), 0, stream, n_shards, nq, stride, pitch, ids, scores, counts,
                     map_sizes, k, k_all, shard_k, out_stride, out_ids, out_scores, out_counts, out_map_sizes, inexact);
  return hipGetLastError();
}

}  // namespace sann
