// sann_unit.h -- the (query, partition) unit kernel's arithmetic that does not depend on its structure (device only): the
// fp32 pre-filter score, the cluster-level cut, the unit's candidate quota, the blocked Bloom filter's hash bits.  Used by
// sann_fast.hip (round 3 also measured a software-pipelined kernel on top of it: profiles/r03_pipelined_unit_kernel_experiment.txt).
#pragma once
#include <hip/hip_runtime.h>

#include "sann_device.h"
#include "sann_kernels.h"
#include "sann_math.h"

namespace sann {

constexpr int NSCAN_MAX = 128;   // scanned clusters a fast unit can describe
// Per-unit side tables are sized by the unit's posting capacity (WG*U) so that small units keep
// LDS small and occupancy high:
//   blocked Bloom filter  capacity/2 64-bit words (>= 256), 4 bits per posting (two per 32-bit half): ~0.005 % false flags
//   "flagged" filter      256 words
//   match list            64 entries up to 1024 postings, 128 above
constexpr float APPROX_EPS = 4e-6f;  // bound on |approx/exact - 1| of the fp32 pre-filter (actual < 1e-6)

__device__ inline double normalise_f(int alg, double dot, double nsq, double l2norm, double lognorm) {
  switch (alg) {
    case 3: return dot / lognorm / strict_log(1 + nsq);
    case 2: return dot / l2norm / sqrt(nsq);
    case 4: return dot / sqrt(nsq);
    case 1: return dot;
    default: return __builtin_nan("");
  }
}

// The fp32 pre-filter score of phase 4.  |approx / exact - 1| <= APPROX_EPS for every input it does not flag
// (checked over a wide sweep by tests/test_sann_gpu.py::test_prefilter_error_bound through sann_debug_approx).
//   nsq64   the candidate's fp64 sum of squares, read only for LogCosine below 1e-6: the exact form is
//           log(1 + nsq) with 1 + nsq ROUNDED to fp64 (ApproximateCosineSimilarity.scala:113), so for small nsq the
//           rounding of that sum -- not nsq -- decides the score (1 + nsq == 1 below 1.1e-16: log 0, score +inf).
//           x = (1 + nsq) - 1 is that rounded excess, exactly; log(1 + x) = x - x^2/2 + O(x^3).
//   *forced the exact score is +inf (or NaN): the candidate must survive the cut whatever tau is.
__device__ inline float approx_score(int alg, float d32, float n32, double nsq64, float invl2, float invln, bool *forced) {
  *forced = false;
  switch (alg) {
    case 2: return d32 * invl2 * __builtin_amdgcn_rsqf(n32);
    case 4: return d32 * __builtin_amdgcn_rsqf(n32);
    case 3: {
      float l;
      if (n32 >= 1e-6f) {
        // log1p(x) = log(u) * x / (u - 1) with u = fl(1 + x): the quotient undoes the rounding of the sum, and log(u) is
        // the hardware's v_log_f32 (1 ulp) -- a dozen issue slots where the library's log1pf took forty.  Audited, like
        // everything here, by tests/test_sann_exactness_gpu.py::test_prefilter_error_bound.
        const float u = 1.0f + n32;
        const float dlt = u - 1.0f;  // (exact; > 0 for n32 >= 1e-6)
        l = __logf(u) * (n32 * __builtin_amdgcn_rcpf(dlt));
      } else {
        const float x = (float)((1.0 + nsq64) - 1.0);
        *forced = !(x > 0.f);
        l = x - 0.5f * x * x;
      }
      return d32 * invln * __builtin_amdgcn_rcpf(l);
    }
    case 1: return d32;
    default: return 0.f;
  }
}
constexpr uint32_t FORCED_KEY = 0xff7fffffu;  // FLT_MAX: above every key the range check below lets through

__device__ inline int lower_bound_rank(const uint32_t *a, int n, uint32_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// Cosine forms: the fp32 key of a single-cluster candidate of a cluster with weight w -- (s w) / sqrt(s s) = w, times
// 1 / l2norm for CosineSimilarity -- or 0 when it is not an ordinary positive magnitude (such clusters are keyed per posting).
__device__ inline uint32_t cosine_cluster_key(int alg, double w, float inv_l2_32) {
  const float wn = alg == 2 ? (float)w * inv_l2_32 : (float)w;
  return (wn > 1e-30f && wn < 1e30f) ? (__float_as_uint(wn) | 0x80000000u) : 0u;
}
// Entries a unit must offer before it may withhold the rest: its share of k, five sigma, and a few.  Tweets are hashed
// to partitions, so the number of a query's final top-k that sit in one unit is Binomial(k, 1/P): at k = 400, P = 32
// (mean 12.5) the bound is 34, exceeded with probability ~1e-7 per unit -- one query re-run through the general path per
// ~300 batches of 32768 units.  (Six sigma + 8 = 41 made every unit offer seven more candidates: 1760 instead of 1540 per
// query, and half of the queries needed a second staging round in the merge kernel.)
__device__ inline int unit_kl(int k, int P, int k_local_floor) {
  const float share = (float)k / (float)P;
  int kl = (int)(share + 5.0f * sqrtf(share) + 4.0f);
  if (kl < k_local_floor) kl = k_local_floor;
  if (kl > k) kl = k;
  if (kl > FAST_SCAP - 32) kl = FAST_SCAP - 32;
  return kl;
}
// Whether a query's units take the cluster-level cut: a single-cluster candidate's key is then a constant of its cluster,
// so WHERE to cut follows from the descriptors alone -- clusters by key, postings counted until kl are covered.  The
// descriptor kernels work it out per unit (unit_pre), the unit kernel only reads it.
__device__ inline bool query_has_cluster_cut(const QueryHdr &h) {
  return (h.alg == 2 || h.alg == 4) && h.n_scan <= 64 && h.use_norms == 0;
}
// the cut itself: 256 fp32 ulps (>= 3 EPS) below the key of the cluster at which the running posting count reaches kl
// (see 5a in the unit kernel for why low); 0 = the unit's postings do not add up to kl: keep everything
__device__ inline uint32_t cluster_cut_from_key(uint32_t kc) { return kc > 0x80000100u ? kc - 256u : kc; }

// FOUR bits of a 64-bit word, two in either half (20 hash bits).  Three bits anywhere in the word flagged one unit in
// five at the benchmark's shape (1264 postings over 2048 words: 0.19 false flags per unit, and a flagged unit lives 65 %
// longer); two per half flag one in seventeen, and the halves are built by 32-bit shifts (8 instructions, not 13).
__device__ inline unsigned long long bloom_bits(uint32_t hv) {
  const uint32_t lo = (1u << (hv & 31)) | (1u << ((hv >> 5) & 31));
  const uint32_t hi = (1u << ((hv >> 10) & 31)) | (1u << ((hv >> 15) & 31));
  return ((unsigned long long)hi << 32) | lo;
}
constexpr int BLOOM_POS_BITS = 20;

constexpr int bloom_log2(int capacity) { return capacity <= 512 ? 8 : capacity <= 1024 ? 9 : 11; }

enum {
  CTL_NFLAG = 0, CTL_NM, CTL_LIVE, CTL_NSURV, CTL_FOLD, CTL_SEL_D, CTL_SEL_A, CTL_SEL_B, CTL_BAD, CTL_KMIN, CTL_KMAX, CTL_PRE,
  CTL_N
};

}  // namespace sann
