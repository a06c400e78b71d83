// sann_store.hip -- the streaming side of the cluster -> top tweets store, gfx950.
//
// The reference keeps, per cluster, a Map[TweetId, ThriftDecayedValue] that a Summingbird job folds event batches
// into with TopKTweetsWithScoresMonoid.plus (src/scala/com/twitter/simclusters_v2/summingbird/common/Monoids.scala:131-158),
// i.e. TopKScoresUtils.mergeTwoTopKMapWithDecayedValues (:378-450) followed by the tweet-age filter (:154):
//
//   either side empty            -> the other side, untouched                                     (:388-394)
//   latest = max scaledTime over both sides                                                       (:396-399)
//   every value decayed to `latest` by ThriftDecayedValueMonoid.plus(v, DecayedValue(0.0, latest)) (:409-410,423-424)
//   kept only if value > threshold                                                                (:413,427)
//   the same key on both sides   -> the larger decayed value (b replaces a only if strictly larger) (:428-434)
//   more than topK * 1.2 entries -> sortBy(-value).take(topK)                                      (:441-448)
//   finally keep tweet ids >= oldestTweetId                                                        (:154)
//
// sann_topk_merge does that for a whole batch of clusters at once: one workgroup per cluster, both sides in LDS, two
// bitonic sorts (by id to pair the sides up, by value to cut and to give the result a fixed order).  A map has no
// order; results are emitted by (value desc, id asc), which is also the tie order of the cut (the reference's is
// the HashMap's iteration order, i.e. unspecified).  com.twitter.algebird is not vendored: DecayedValueMonoid is
// restated from its published definition (sann_math.h, decay_to_timestamp; zero = DecayedValue(0.0, -inf)).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <string>
#include <vector>

#include "../../include/simclusters_ann.h"
#include "sann_host.h"
#include "sann_math.h"

using namespace sann;
using sann_host::DevBuf;
using sann_host::fail;
#include "abi_guard.h"
#define ABI_CATCH catch (...) { return abi_guard::caught(sann_host::fail, SANN_ENOMEM, SANN_EINTERNAL); }

namespace {

constexpr int MERGE_MAX = 4096;  // entries of both sides of one cluster (the store keeps <= 1.2 x 1600 per side)
constexpr int MWG = 256;

struct MergeView {
  const int64_t *a_off, *b_off;  // [n_lists + 1], relative to the uploaded arrays
  const int64_t *a_id, *b_id;
  const double *a_val, *b_val, *a_t, *b_t;
  int64_t *o_id;   // [a_total + b_total]: list i writes at a_off[i] + b_off[i]
  double *o_val, *o_t;
  int32_t *o_cnt;  // [n_lists]
  int32_t top_k;
  double threshold;
  int64_t oldest;
};

// ascending by (hi, lo), payload carried along
__device__ void bitonic_asc(uint64_t *hi, uint64_t *lo, double *pay, int np) {
  const int tid = threadIdx.x;
  for (int size = 2; size <= np; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (np >> 1); t += MWG) {
        const int i = 2 * t - (t & (stride - 1)), j = i + stride;
        const bool asc = (i & size) == 0;
        const uint64_t ah = hi[i], al = lo[i], bh = hi[j], bl = lo[j];
        const bool a_gt_b = ah > bh || (ah == bh && al > bl);
        const bool a_lt_b = ah < bh || (ah == bh && al < bl);
        if (asc ? a_gt_b : a_lt_b) {
          hi[i] = bh; lo[i] = bl; hi[j] = ah; lo[j] = al;
          const double p = pay[i]; pay[i] = pay[j]; pay[j] = p;
        }
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(MWG) void topk_merge_kernel(MergeView v) {
  extern __shared__ unsigned char smem[];
  uint64_t *s_hi = reinterpret_cast<uint64_t *>(smem);           // [MERGE_MAX]
  uint64_t *s_lo = s_hi + MERGE_MAX;                             // [MERGE_MAX]
  double *s_t = reinterpret_cast<double *>(s_lo + MERGE_MAX);    // [MERGE_MAX] payload: the entry's scaledTime
  __shared__ unsigned long long s_latest;
  __shared__ int s_cnt[MWG + 1];
  __shared__ int s_m;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int64_t a0 = v.a_off[c], b0 = v.b_off[c];
  const int na = (int)(v.a_off[c + 1] - a0), nb = (int)(v.b_off[c + 1] - b0);
  const int n = na + nb;
  int np = 2;
  while (np < n) np <<= 1;
  const bool pass_through = na == 0 || nb == 0;  // :388-394 (an absent side and an empty map behave alike)
  if (tid == 0) { s_latest = 0ull; s_m = 0; }
  __syncthreads();
  // latest scaled time (:396-399), through the order-preserving key
  if (!pass_through) {
    unsigned long long mx = 0ull;
    for (int i = tid; i < n; i += MWG) {
      const double t = i < na ? v.a_t[a0 + i] : v.b_t[b0 + i - na];
      const unsigned long long k = score_key(t);
      mx = k > mx ? k : mx;
    }
    atomicMax(&s_latest, mx);
  }
  __syncthreads();
  const double latest = pass_through ? 0.0 : key_score(s_latest);
  // ---- 1. decay, threshold; sort so that equal ids are adjacent, the larger value first -----------------------------
  for (int i = tid; i < np; i += MWG) {
    uint64_t hi = ~0ull, lo = ~0ull;  // padding and dropped entries sort last
    double tt = 0.0;
    if (i < n) {
      const bool from_a = i < na;
      const int64_t id = from_a ? v.a_id[a0 + i] : v.b_id[b0 + i - na];
      double val = from_a ? v.a_val[a0 + i] : v.b_val[b0 + i - na];
      tt = from_a ? v.a_t[a0 + i] : v.b_t[b0 + i - na];
      bool keep = true;
      if (!pass_through) {
        val = decay_to_timestamp(val, tt, latest);
        tt = (val > 0.0 || val < 0.0) ? latest : -__builtin_inf();  // DecayedValueMonoid.zero = (0.0, -inf)
        keep = val > v.threshold;                                    // :413,427
      }
      if (keep) {
        hi = id_key(id);
        lo = ~score_key(val);  // larger value first within one id (equal values: the entries are identical)
      }
    }
    s_hi[i] = hi;
    s_lo[i] = lo;
    s_t[i] = tt;
  }
  __syncthreads();
  bitonic_asc(s_hi, s_lo, s_t, np);
  // ---- 2. one entry per id (the first = the larger value, :428-434); re-key by (value desc, id asc) ---------------
  uint64_t nh[MERGE_MAX / MWG], nl[MERGE_MAX / MWG];
  int mine = 0;
#pragma unroll
  for (int r = 0; r < MERGE_MAX / MWG; r++) {
    const int i = r * MWG + tid;
    nh[r] = ~0ull;
    nl[r] = ~0ull;
    if (i < np) {
      const uint64_t hi = s_hi[i], lo = s_lo[i];
      const bool valid = !(hi == ~0ull && lo == ~0ull);
      const bool first = i == 0 || s_hi[i - 1] != hi;
      if (valid && first) {
        nh[r] = lo;   // ~score_key(value): ascending = value descending
        nl[r] = ~hi;  // id_key orders ids descending (sann_math.h): its complement ascending = id ascending
        mine++;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < MERGE_MAX / MWG; r++) {
    const int i = r * MWG + tid;
    if (i < np) { s_hi[i] = nh[r]; s_lo[i] = nl[r]; }
  }
  if (mine) atomicAdd(&s_m, mine);
  __syncthreads();
  bitonic_asc(s_hi, s_lo, s_t, np);
  // ---- 3. the cut (:441-448), the age filter (:154), ordered compaction into the output ----------------------------
  const int m = s_m;
  const int kept = (!pass_through && (double)m > (double)v.top_k * 1.2) ? (v.top_k < m ? v.top_k : m) : m;
  const int per = (kept + MWG - 1) / MWG;
  const int lo_i = tid * per, hi_i = (lo_i + per < kept) ? lo_i + per : kept;
  int cnt = 0;
  for (int i = lo_i; i < hi_i; i++) cnt += key_id(~s_lo[i]) >= v.oldest ? 1 : 0;
  s_cnt[tid] = cnt;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < MWG; t++) { const int x = s_cnt[t]; s_cnt[t] = run; run += x; }
    s_cnt[MWG] = run;
  }
  __syncthreads();
  const int64_t obase = a0 + b0;
  int o = s_cnt[tid];
  for (int i = lo_i; i < hi_i; i++) {
    const int64_t id = key_id(~s_lo[i]);
    if (id >= v.oldest) {
      v.o_id[obase + o] = id;
      v.o_val[obase + o] = key_score(~s_hi[i]);
      v.o_t[obase + o] = s_t[i];
      o++;
    }
  }
  if (tid == 0) v.o_cnt[c] = s_cnt[MWG];
}

}  // namespace

extern "C" int sann_topk_merge(int32_t device, int32_t n_lists, const int64_t *a_offsets, const int64_t *a_ids,
                               const double *a_values, const double *a_scaled_times, const int64_t *b_offsets,
                               const int64_t *b_ids, const double *b_values, const double *b_scaled_times, int32_t top_k,
                               double threshold, int64_t oldest_tweet_id, int64_t out_capacity, int64_t *out_offsets,
                               int64_t *out_ids, double *out_values, double *out_scaled_times) try {
  if (n_lists < 0 || top_k < 0) return fail(SANN_EINVAL, "n_lists / top_k must not be negative");
  if (!out_offsets) return fail(SANN_EINVAL, "out_offsets is NULL");
  if (n_lists > 0 && (!a_offsets || !b_offsets)) return fail(SANN_EINVAL, "offset arrays are NULL");
  out_offsets[0] = 0;
  if (n_lists == 0) return SANN_OK;
  const int64_t a_o0 = a_offsets[0], b_o0 = b_offsets[0];
  const int64_t a_total = a_offsets[n_lists] - a_o0, b_total = b_offsets[n_lists] - b_o0;
  if (a_total < 0 || b_total < 0) return fail(SANN_EINVAL, "offsets must be non-decreasing");
  if ((a_total > 0 && (!a_ids || !a_values || !a_scaled_times)) || (b_total > 0 && (!b_ids || !b_values || !b_scaled_times)))
    return fail(SANN_EINVAL, "id / value / scaled-time arrays are NULL");
  std::vector<int64_t> ao((size_t)n_lists + 1), bo((size_t)n_lists + 1), tmp;
  for (int32_t i = 0; i <= n_lists; i++) {
    ao[(size_t)i] = a_offsets[i] - a_o0;
    bo[(size_t)i] = b_offsets[i] - b_o0;
  }
  for (int32_t i = 0; i < n_lists; i++) {
    const int64_t la = ao[(size_t)i + 1] - ao[(size_t)i], lb = bo[(size_t)i + 1] - bo[(size_t)i];
    if (la < 0 || lb < 0) return fail(SANN_EINVAL, "offsets must be non-decreasing");
    if (la + lb > MERGE_MAX) return fail(SANN_ELIMIT, "list " + std::to_string(i) + ": more than 4096 entries on both sides together");
    for (int side = 0; side < 2; side++) {  // each side is a Map: its keys are unique
      const int64_t *ids = side ? b_ids + b_o0 + bo[(size_t)i] : a_ids + a_o0 + ao[(size_t)i];
      const int64_t len = side ? lb : la;
      if (len < 2) continue;
      tmp.assign(ids, ids + len);
      std::sort(tmp.begin(), tmp.end());
      if (std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end())
        return fail(SANN_EINVAL, "list " + std::to_string(i) + ": a tweet id appears twice on one side");
    }
  }
  const int64_t total = a_total + b_total;
  HIP_TRY(hipSetDevice(device));
  DevBuf d_ao, d_bo, d_aid, d_bid, d_av, d_bv, d_at, d_bt, d_oid, d_ov, d_ot, d_cnt;
  HIP_TRY(d_ao.alloc(ao.size() * 8));
  HIP_TRY(d_bo.alloc(bo.size() * 8));
  HIP_TRY(hipMemcpy(d_ao.p, ao.data(), ao.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_bo.p, bo.data(), bo.size() * 8, hipMemcpyHostToDevice));
  auto up = [&](DevBuf &d, const void *src, int64_t n) -> hipError_t {
    hipError_t e = d.alloc((size_t)std::max<int64_t>(n, 1) * 8);
    if (e != hipSuccess || n == 0) return e;
    return hipMemcpy(d.p, src, (size_t)n * 8, hipMemcpyHostToDevice);
  };
  HIP_TRY(up(d_aid, a_ids ? a_ids + a_o0 : nullptr, a_total));
  HIP_TRY(up(d_av, a_values ? a_values + a_o0 : nullptr, a_total));
  HIP_TRY(up(d_at, a_scaled_times ? a_scaled_times + a_o0 : nullptr, a_total));
  HIP_TRY(up(d_bid, b_ids ? b_ids + b_o0 : nullptr, b_total));
  HIP_TRY(up(d_bv, b_values ? b_values + b_o0 : nullptr, b_total));
  HIP_TRY(up(d_bt, b_scaled_times ? b_scaled_times + b_o0 : nullptr, b_total));
  HIP_TRY(d_oid.alloc((size_t)std::max<int64_t>(total, 1) * 8));
  HIP_TRY(d_ov.alloc((size_t)std::max<int64_t>(total, 1) * 8));
  HIP_TRY(d_ot.alloc((size_t)std::max<int64_t>(total, 1) * 8));
  HIP_TRY(d_cnt.alloc((size_t)n_lists * 4));
  MergeView v;
  v.a_off = d_ao.as<int64_t>(); v.b_off = d_bo.as<int64_t>();
  v.a_id = d_aid.as<int64_t>(); v.b_id = d_bid.as<int64_t>();
  v.a_val = d_av.as<double>(); v.b_val = d_bv.as<double>();
  v.a_t = d_at.as<double>(); v.b_t = d_bt.as<double>();
  v.o_id = d_oid.as<int64_t>(); v.o_val = d_ov.as<double>(); v.o_t = d_ot.as<double>();
  v.o_cnt = d_cnt.as<int32_t>();
  v.top_k = top_k;
  v.threshold = threshold;
  v.oldest = oldest_tweet_id;
  const size_t lds = (size_t)MERGE_MAX * 24;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(topk_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)n_lists), dim3(MWG), lds, 0, v);
  HIP_TRY(hipGetLastError());
  std::vector<int32_t> cnt((size_t)n_lists);
  HIP_TRY(hipMemcpy(cnt.data(), d_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost));
  int64_t out_total = 0;
  for (int32_t i = 0; i < n_lists; i++) {
    out_total += cnt[(size_t)i];
    out_offsets[i + 1] = out_total;
  }
  if (out_total > out_capacity) return fail(SANN_ELIMIT, "out_capacity is smaller than the merged lists (" + std::to_string(out_total) + " entries)");
  if (out_total > 0 && (!out_ids || !out_values || !out_scaled_times)) return fail(SANN_EINVAL, "output arrays are NULL");
  std::vector<int64_t> h_id((size_t)std::max<int64_t>(total, 1));
  std::vector<double> h_v((size_t)std::max<int64_t>(total, 1)), h_t((size_t)std::max<int64_t>(total, 1));
  if (total) {
    HIP_TRY(hipMemcpy(h_id.data(), d_oid.p, (size_t)total * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_v.data(), d_ov.p, (size_t)total * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_t.data(), d_ot.p, (size_t)total * 8, hipMemcpyDeviceToHost));
  }
  for (int32_t i = 0; i < n_lists; i++) {
    const int64_t src = ao[(size_t)i] + bo[(size_t)i], dst = out_offsets[i];
    std::copy_n(h_id.data() + src, cnt[(size_t)i], out_ids + dst);
    std::copy_n(h_v.data() + src, cnt[(size_t)i], out_values + dst);
    std::copy_n(h_t.data() + src, cnt[(size_t)i], out_scaled_times + dst);
  }
  return SANN_OK;
} ABI_CATCH
