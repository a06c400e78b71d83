// sann_prep.hip -- query preparation on the device, gfx950.
//
// What the reference does per request on a JVM thread before its hot loop, restated as one kernel over the whole
// batch (one workgroup per query), so that a fresh batch costs the host one packed H2D copy and no per-query work:
//
//   SimClustersEmbedding constructor     drop score <= 0, order by (score desc, cluster id asc)
//                                        src/scala/com/twitter/simclusters_v2/common/SimClustersEmbedding.scala:490-509,456-463
//   sortedClusterIds / sortedScores      the same entries by cluster id ascending                        :37-41
//   l2norm / logNorm                     left folds over sortedScores: sqrt(sum x^2), log(sum x^2 + 1)    :59-63
//                                        CosineSimilarityUtil.scala:15-17,29-31,43-45
//   fetchCandidates                      truncate(maxScanClusters).getClusterIds().toSet, iterated ascending
//                                        simclusters-ann/.../candidate_source/SimClustersANNCandidateSource.scala:72-75
//   `if sourceEmbedding.contains(c)` / getOrElse(c, 0.0)   ApproximateCosineSimilarity.scala:84 ; Experimental :62-63
//   age window, source-tweet exclusion   ApproximateCosineSimilarity.scala:65-72,90 ; Optimized :56
//
// It is the same arithmetic, in the same order, as prepare_query_host() in sann_api.hip (the path for embeddings of
// more than PREP_MAX entries, for SANN_HOST_PREP=1, and for the general-path workspace bounds): sqrt and the
// fdlibm log are bit-identical on both sides (tests/test_sann_gpu.py::test_device_fp64_division_sqrt_log_are_bit_exact),
// and tests/test_sann_reuse_gpu.py compares the two preparations field by field through their results.
#include <hip/hip_runtime.h>

#include "../../include/simclusters_ann.h"
#include "sann_device.h"
#include "sann_kernels.h"
#include "sann_math.h"

namespace sann {

namespace {

constexpr int PWG = 256;
constexpr int64_t kSnowflakeEpochMs = 1288834974657ll;  // BQGenerationUtil.scala:150-153
__device__ inline int64_t first_id_for(int64_t ms) { return (int64_t)((uint64_t)(ms - kSnowflakeEpochMs) << 22); }

__device__ inline int pow2_at_least(int x) {
  int p = 2;
  while (p < x) p <<= 1;
  return p;
}

// bitonic sorts over LDS arrays, n a power of two, PWG threads
__device__ void sort_desc_pairs(ulonglong2 *e, int n) {
  const int tid = threadIdx.x;
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (n >> 1); t += PWG) {
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
__device__ void sort_asc_u64(unsigned long long *e, int n) {
  const int tid = threadIdx.x;
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (n >> 1); t += PWG) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const bool asc = ((i & size) == 0);
        const unsigned long long a = e[i], c = e[j];
        if (asc ? a > c : a < c) {
          e[i] = c;
          e[j] = a;
        }
      }
      __syncthreads();
    }
  }
}

__device__ inline uint32_t cluster_order_key(int32_t c) { return (uint32_t)c ^ 0x80000000u; }  // unsigned order = signed order

__device__ inline int row_of(const int32_t *cluster_ids, int n_rows, int32_t cluster) {
  int lo = 0, hi = n_rows;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cluster_ids[mid] < cluster) lo = mid + 1;
    else hi = mid;
  }
  return (lo < n_rows && cluster_ids[lo] == cluster) ? lo : -1;
}

}  // namespace

__global__ __launch_bounds__(PWG) void prep_kernel(PrepView in) {
  __shared__ ulonglong2 s_e[PREP_MAX];          // (score key, id key), descending = the embedding's own order
  __shared__ unsigned long long s_b[PREP_MAX];  // (cluster order key << 32 | position in s_e), ascending = by cluster id
  __shared__ int s_n, s_base, s_wave[PWG / 64];
  __shared__ double s_sumsq;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.x;
  const sann_config_t cfg = in.configs[in.n_configs == 1 ? 0 : q];
  const int64_t eb = in.emb_offsets[q];
  const int n_raw = (int)(in.emb_offsets[q + 1] - eb);  // the host checked 0 <= n_raw <= PREP_MAX
  const int np = pow2_at_least(n_raw);
  if (tid == 0) { s_n = 0; s_base = 0; }
  __syncthreads();

  // ---- SimClustersEmbedding constructor: keep score > 0, order by (score desc, cluster id asc) ----------------
  int mine = 0;
  for (int i = tid; i < np; i += PWG) {
    ulonglong2 e = make_ulonglong2(0ull, 0ull);  // sorts behind every kept entry (a positive score's key has the top bit set)
    if (i < n_raw) {
      const double s = in.emb_scores[eb + i];
      if (s > 0.0) {
        e = make_ulonglong2(score_key(s), id_key((int64_t)in.emb_cluster_ids[eb + i]));
        mine++;
      }
    }
    s_e[i] = e;
  }
  if (mine) atomicAdd(&s_n, mine);
  __syncthreads();
  sort_desc_pairs(s_e, np);
  const int n = s_n;

  // ---- sortedClusterIds / sortedScores: stable by cluster id ------------------------------------------------------
  for (int i = tid; i < np; i += PWG)
    s_b[i] = i < n ? ((unsigned long long)cluster_order_key((int32_t)key_id(s_e[i].y)) << 32) | (unsigned)i : ~0ull;
  __syncthreads();
  sort_asc_u64(s_b, np);

  // ---- norms: one thread folds left to right over sortedScores (CosineSimilarityUtil.sumOfSquaresArray) ------------
  if (tid == 0) {
    double sumsq = 0.0;
    for (int i = 0; i < n; i++) {
      const double x = key_score(s_e[(uint32_t)s_b[i]].x);
      sumsq = sumsq + x * x;
    }
    s_sumsq = sumsq;
  }

  // ---- keys of clusterTweetsMap in accumulation order, resolved to (index row, weight) ------------------------------
  const bool explicit_keys = in.scan_offsets != nullptr;
  const int64_t so = explicit_keys ? in.scan_offsets[q] : 0;
  const int n_slots = explicit_keys ? (int)(in.scan_offsets[q + 1] - so) : n;
  // default: truncate(maxScanClusters) = the first nk entries of the embedding; their ids ascending are exactly the
  // by-id order filtered to position < nk
  const int nk = cfg.max_scan_clusters < 0 ? 0 : (n < cfg.max_scan_clusters ? n : cfg.max_scan_clusters);
  const int begin = in.scan_begin[q];
  for (int j0 = 0; j0 < n_slots; j0 += PWG) {
    const int j = j0 + tid;
    bool take = false;
    int32_t cluster = 0;
    if (j < n_slots) {
      if (explicit_keys) {
        cluster = in.scan_cluster_ids[so + j];
        take = true;
      } else {
        const unsigned long long e = s_b[j];
        cluster = (int32_t)((uint32_t)(e >> 32) ^ 0x80000000u);
        take = (int)(uint32_t)e < nk;
      }
    }
    int row = -1;
    double w = 0.0;
    if (take) {
      // by-id lookup: the first entry with cluster id >= cluster (the reference's contains / getOrElse)
      const uint32_t ck = cluster_order_key(cluster);
      int lo = 0, hi = n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((uint32_t)(s_b[mid] >> 32) < ck) lo = mid + 1;
        else hi = mid;
      }
      const bool contained = lo < n && (uint32_t)(s_b[lo] >> 32) == ck;
      if (!contained && in.variant != SANN_VARIANT_EXPERIMENTAL) {
        take = false;  // `if sourceEmbedding.contains(clusterId)`
      } else {
        w = contained ? key_score(s_e[(uint32_t)s_b[lo]].x) : 0.0;  // getOrElse(clusterId)
        row = row_of(in.cluster_ids, in.n_rows, cluster);
        take = row >= 0;  // None in clusterTweetsMap
      }
    }
    // ordered compaction of this chunk
    const unsigned long long m = __ballot(take);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int wbase = s_base;
    for (int w2 = 0; w2 < wave; w2++) wbase += s_wave[w2];
    if (take) {
      in.scan_row[begin + wbase + before] = row;
      in.scan_w[begin + wbase + before] = w;
    }
    __syncthreads();
    if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  __syncthreads();

  if (tid == 0) {
    QueryHdr h;
    const double sumsq = s_sumsq;
    h.l2norm = sqrt(sumsq);
    h.lognorm = strict_log(sumsq + 1);
    h.min_score = cfg.min_score;
    const bool legacy = in.variant == SANN_VARIANT_LEGACY;
    if (legacy) {  // SimClustersANNCandidateSource.scala:160-180 (see prepare_query_host)
      h.lognorm = h.l2norm;
      h.min_score = -__builtin_inf();
    }
    h.M = cfg.max_top_tweets_per_cluster < 0 ? 0 : cfg.max_top_tweets_per_cluster;
    const int k = cfg.max_num_results < 1000 ? cfg.max_num_results : 1000;
    h.k = k < 0 ? 0 : k;
    h.alg = cfg.ann_algorithm;
    h.use_norms = 0;
    h.reserved = 0;
    const int64_t now_ms = in.now_ms_q ? in.now_ms_q[q] : in.now_ms;  // Time.now of THIS request (micro-batched requests keep their own)
    h.earliest = (cfg.max_tweet_candidate_age_hours >= 175200 && !legacy)
                     ? 0
                     : first_id_for(now_ms - (int64_t)cfg.max_tweet_candidate_age_hours * 3600000ll);
    h.latest = first_id_for(now_ms - (int64_t)cfg.min_tweet_candidate_age_hours * 3600000ll);
    if (cfg.ann_algorithm == SANN_ALG_OFFLINE_LOG_COSINE || cfg.ann_algorithm == SANN_ALG_OFFLINE_COSINE) {
      // tweets_ann.sql:44-52 (see prepare_query_host)
      h.alg = cfg.ann_algorithm == SANN_ALG_OFFLINE_LOG_COSINE ? SANN_ALG_LOG_COSINE : SANN_ALG_COSINE_NO_SOURCE_NORM;
      h.lognorm = 1.0;
      h.use_norms = 1;
      h.earliest = (int64_t)0x8000000000000000ull;
      h.latest = 0x7fffffffffffffffll;
    }
    h.inv_l2_32 = (float)(1.0 / h.l2norm);
    h.inv_ln_32 = (float)(1.0 / h.lognorm);
    const bool has_src = in.has_source_tweet && in.has_source_tweet[q] && in.source_tweet_ids;
    if (in.variant == SANN_VARIANT_ORIGINAL || legacy) {
      h.excl_enabled = has_src ? 1 : 0;
      h.src_excl = has_src ? in.source_tweet_ids[q] : 0;
    } else {
      h.excl_enabled = 1;
      h.src_excl = has_src ? in.source_tweet_ids[q] : 0;
    }
    h.scan_begin = begin;
    h.n_scan = s_base;
    in.hdr[q] = h;
    in.d_k[q] = h.k;
  }
}

hipError_t launch_prep(const PrepView &in, hipStream_t stream) {
  if (in.nq <= 0) return hipSuccess;
  hipLaunchKernelGGL(prep_kernel, dim3((unsigned)in.nq), dim3(PWG), 0, stream, in);
  return hipGetLastError();
}

}  // namespace sann
