// sann_api.hip -- C ABI of the SimClusters-ANN engine (include/simclusters_ann.h): index build,
// query preparation (the SimClustersEmbedding / fetchCandidates semantics that stay on the
// host), launch orchestration and the general-path fallback.
//
// Host-side reference semantics restated here (paths relative to /root/reference/):
//   src/scala/com/twitter/simclusters_v2/common/SimClustersEmbedding.scala:490-509  constructor
//   .../SimClustersEmbedding.scala:377-392  truncate ; :115-125 getOrElse ; :140 contains
//   .../CosineSimilarityUtil.scala:15-17,29-31,43-45  sumOfSquares / norm / logNorm
//   simclusters-ann/.../candidate_source/SimClustersANNCandidateSource.scala:72-75  cluster choice
//   simclusters-ann/.../candidate_source/ApproximateCosineSimilarity.scala:65-72    age window
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>
#include <limits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/representation_scorer.h"
#include "../../include/simclusters_ann.h"
#include "sann_host.h"
#include "sann_kernels.h"
#include "abi_guard.h"

using namespace sann;

namespace sann_host {
thread_local std::string g_err;
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
}  // namespace sann_host
using sann_host::DevBuf;
using sann_host::fail;
using sann_host::g_err;
#define ABI_CATCH catch (...) { return abi_guard::caught(sann_host::fail, SANN_ENOMEM, SANN_EINTERNAL); }

namespace {

// Batches kept in flight -- sann_batch_run_after, and the pooled sann_get_tweet_candidates from several threads --
// run on several HIP streams.  The runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and
// two streams that land on one queue execute in order: the overlap is then lost without any error (round 1 measured
// 0.58 ms instead of 0.36 ms per batch).  The variable is read when the HIP runtime initialises, i.e. it belongs to the
// PROCESS: the launcher / JVM shim exports GPU_MAX_HW_QUEUES=8 before the process starts (INTEGRATION.md section 2;
// bench.py and the Python mirror do it before they load anything).  The library itself never touches the environment
// (setenv is not thread-safe against a running JVM's getenv, and it would change HIP for every other user of the
// process); it only reports, once, when it finds fewer queues than its streams want.
std::string g_advice;
void check_hw_queues_once() {
  static std::once_flag once;
  std::call_once(once, [] {
    const char *v = getenv("GPU_MAX_HW_QUEUES");
    const int n = v ? atoi(v) : 4;
    if (n < 8) {
      g_advice = std::string("GPU_MAX_HW_QUEUES=") + (v ? v : "(unset: 4)") +
                 ": batches in flight use several HIP streams, and streams that share a hardware queue run in order; "
                 "export GPU_MAX_HW_QUEUES=8 before the process starts";
      fprintf(stderr, "simclusters_amd: %s\n", g_advice.c_str());
    }
  });
}

constexpr int64_t kSnowflakeEpochMs = 1288834974657ll;  // BQGenerationUtil.scala:150-153
inline int64_t snowflake_first_id_for(int64_t ms) { return (int64_t)((uint64_t)(ms - kSnowflakeEpochMs) << 22); }

inline int java_double_compare(double a, double b) {
  if (a < b) return -1;
  if (a > b) return 1;
  uint64_t x = f64_bits(a), y = f64_bits(b);
  if (a != a) x = 0x7ff8000000000000ull;
  if (b != b) y = 0x7ff8000000000000ull;
  return x == y ? 0 : ((int64_t)x < (int64_t)y ? -1 : 1);
}

uint32_t next_pow2_u32(uint64_t x) {
  uint32_t p = 16;
  while (p < x) p <<= 1;
  return p;
}

}  // namespace

namespace {

// Pinned host memory that only grows.
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  PinBuf() = default;
  PinBuf(const PinBuf &) = delete;
  PinBuf &operator=(const PinBuf &) = delete;
  ~PinBuf() { if (p) (void)hipHostFree(p); }
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = n + n / 4 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
};

// One query prepared on the host: the reference's per-request work before the hot loop, in its order.  Used for
// embeddings longer than the device path takes, under SANN_HOST_PREP=1 / SANN_FORCE_GENERAL=1, and -- for the
// queries the fast path could not settle -- to size the general path's tables (unit_bound).
struct HostQuery {
  QueryHdr h{};
  std::vector<int32_t> rows;
  std::vector<double> w;
  std::vector<uint64_t> unit_bound;  // [P] postings the unit can scan, at most
  std::vector<double> unit_est, unit_var;  // [P] expected postings with rank < M, and the variance of that count
  int64_t postings_scanned = 0;
};

struct IdScore { int32_t id; double score; };

int prepare_query_host(const sann_index *ix, int variant, int64_t now_ms, const sann_config_t &cfg, const int32_t *cids,
                       const double *scores, int64_t n_emb, bool has_src, int64_t src_id, const int32_t *scan, int64_t n_scan_keys,
                       HostQuery &out, std::vector<IdScore> &emb, std::vector<IdScore> &by_id, std::vector<int32_t> &keys) {
  // SimClustersEmbedding constructor: drop score <= 0, order by (score desc, cluster id asc)
  emb.clear();
  for (int64_t i = 0; i < n_emb; i++)
    if (scores[i] > 0.0) emb.push_back({cids[i], scores[i]});
  std::sort(emb.begin(), emb.end(), [](const IdScore &x, const IdScore &y) {
    int c = java_double_compare(y.score, x.score);
    if (c) return c < 0;
    return x.id < y.id;
  });
  by_id = emb;
  std::stable_sort(by_id.begin(), by_id.end(), [](const IdScore &x, const IdScore &y) { return x.id < y.id; });
  // CosineSimilarityUtil.sumOfSquaresArray over sortedScores (left fold)
  double sumsq = 0.0;
  for (const IdScore &x : by_id) sumsq = sumsq + x.score * x.score;

  QueryHdr &h = out.h;
  h.l2norm = std::sqrt(sumsq);
  h.lognorm = strict_log(sumsq + 1);
  h.min_score = cfg.min_score;
  if (variant == SANN_VARIANT_LEGACY) {
    // SimClustersANNCandidateSource.scala:160-180: the "log" form divides by l2norm, nothing is
    // filtered by minScore, and there is no cap below maxNumResults
    h.lognorm = h.l2norm;
    h.min_score = -std::numeric_limits<double>::infinity();
  }
  h.M = cfg.max_top_tweets_per_cluster < 0 ? 0 : cfg.max_top_tweets_per_cluster;
  int k = cfg.max_num_results < 1000 ? cfg.max_num_results : 1000;
  h.k = k < 0 ? 0 : k;
  h.alg = cfg.ann_algorithm;
  h.use_norms = 0;
  h.reserved = 0;
  // age window (ApproximateCosineSimilarity.scala:65-72)
  h.earliest = (cfg.max_tweet_candidate_age_hours >= 175200 && variant != SANN_VARIANT_LEGACY)
                   ? 0
                   : snowflake_first_id_for(now_ms - (int64_t)cfg.max_tweet_candidate_age_hours * 3600000ll);
  h.latest = snowflake_first_id_for(now_ms - (int64_t)cfg.min_tweet_candidate_age_hours * 3600000ll);
  if (cfg.ann_algorithm == SANN_ALG_OFFLINE_LOG_COSINE || cfg.ann_algorithm == SANN_ALG_OFFLINE_COSINE) {
    // tweets_ann.sql:44-52: dot / LN(1 + norm) and dot / SQRT(norm) with the tweet's FULL norm and no source norm:
    // the log form with logNorm = 1, the no-source-norm cosine form, both with nsq taken from the norms column;
    // the job has no age window
    h.alg = cfg.ann_algorithm == SANN_ALG_OFFLINE_LOG_COSINE ? SANN_ALG_LOG_COSINE : SANN_ALG_COSINE_NO_SOURCE_NORM;
    h.lognorm = 1.0;
    h.use_norms = 1;
    h.earliest = std::numeric_limits<int64_t>::min();
    h.latest = std::numeric_limits<int64_t>::max();
  }
  h.inv_l2_32 = (float)(1.0 / h.l2norm);
  h.inv_ln_32 = (float)(1.0 / h.lognorm);
  // source-tweet exclusion (:90 ; Optimized :56,:67 ; Experimental :59,:70)
  if (variant == SANN_VARIANT_ORIGINAL || variant == SANN_VARIANT_LEGACY) {
    h.excl_enabled = has_src ? 1 : 0;
    h.src_excl = has_src ? src_id : 0;
  } else {
    h.excl_enabled = 1;
    h.src_excl = has_src ? src_id : 0;
  }

  // keys of clusterTweetsMap in accumulation order
  keys.clear();
  if (n_scan_keys >= 0) {
    keys.assign(scan, scan + n_scan_keys);
  } else {
    // truncate(maxScanClusters).getClusterIds().toSet -> ascending cluster id
    int64_t n = cfg.max_scan_clusters < 0 ? 0 : std::min<int64_t>((int64_t)emb.size(), cfg.max_scan_clusters);
    for (int64_t i = 0; i < n; i++) keys.push_back(emb[(size_t)i].id);
    std::sort(keys.begin(), keys.end());
  }
  const int P = ix->P;
  out.rows.clear();
  out.w.clear();
  out.unit_bound.assign((size_t)P, 0);
  out.unit_est.assign((size_t)P, 0.0);
  out.unit_var.assign((size_t)P, 0.0);
  out.postings_scanned = 0;
  for (int32_t cluster : keys) {
    auto it = std::lower_bound(by_id.begin(), by_id.end(), cluster, [](const IdScore &x, int32_t c) { return x.id < c; });
    bool contained = it != by_id.end() && it->id == cluster;
    if (!contained && variant != SANN_VARIANT_EXPERIMENTAL) continue;  // `if sourceEmbedding.contains(clusterId)`
    double w = contained ? it->score : 0.0;                             // getOrElse(clusterId)
    int row = ix->row_of(cluster);
    if (row < 0) continue;  // None in clusterTweetsMap
    out.rows.push_back(row);
    out.w.push_back(w);
    uint64_t bound_sum = 0;
    const uint64_t whole = ix->h_sub_offsets[(size_t)(row + 1) * P] - ix->h_sub_offsets[(size_t)row * P];
    // share of the list with rank < M, as if this shard held 1/n_shards of every list
    const double frac = whole == 0 ? 0.0 : std::min(1.0, (double)h.M / ((double)whole * ix->n_shards));
    for (int p = 0; p < P; p++) {
      uint64_t len = ix->h_sub_offsets[(size_t)row * P + p + 1] - ix->h_sub_offsets[(size_t)row * P + p];
      uint64_t lim = std::min<uint64_t>(len, (uint64_t)h.M);
      out.unit_bound[(size_t)p] += lim;
      out.unit_est[(size_t)p] += (double)len * frac;
      out.unit_var[(size_t)p] += (double)len * frac * (1.0 - frac);  // each posting has rank < M with prob. frac
      bound_sum += lim;
    }
    // exact when this shard holds whole lists: min(len_c, M) postings have rank < M
    out.postings_scanned += (int64_t)(ix->n_shards == 1 ? std::min<uint64_t>(whole, (uint64_t)h.M) : bound_sum);
  }
  h.n_scan = (int32_t)out.rows.size();
  return SANN_OK;
}

// smallest fast-path geometry (postings a unit holds in registers) with room for `need` postings
int geometry_for(double need) {
  static const int kCaps[] = {256, 512, 768, 1024, 1536, 2048, 3072, 4096};  // workgroup size x postings per thread
  for (int c : kCaps)
    if ((double)c - 16.0 >= need) return c;
  return 4096;
}

}  // namespace

struct sann_batch {
  sann_index *ix = nullptr;
  int nq = 0, variant = 0, cap = 0, stride = 0, n_units = 0;
  int64_t now_ms = 0;
  bool device_prep = false;  // the batch was prepared by prep_kernel; the host keeps only the packed inputs
  // ---- packed copy of the caller's arrays (pinned): what the device path uploads, and what the slow tail re-reads
  PinBuf stage;
  DevBuf d_stage;
  struct StageLayout {
    size_t emb_offsets = 0, src_ids = 0, scan_offsets = 0, emb_scores = 0, emb_cids = 0, scan_cids = 0, configs = 0,
           scan_begin = 0, has_src = 0, now_q = 0, bytes = 0;
    bool has_scan = false, has_sources = false, has_now_q = false;
    int32_t n_configs = 0;
  } lay;
  // ---- host-prepared form (host path only)
  std::vector<QueryHdr> h_hdr;
  std::vector<int32_t> h_scan_row;
  std::vector<double> h_scan_w;
  std::vector<int32_t> h_k;
  DevBuf hdr, scan_row, scan_w, scan_wq, desc, unit_T, d_k, q_stat;
  DevBuf cand_key, cand_id, cand_cnt, unit_unique, unit_flags, unit_fb, unit_thr, status, overflow_units;
  DevBuf out_ids, out_scores, out_counts, out_map_sizes, prof, merge_done;
  // caller-bound output buffers (NULL = the batch's own)
  void *bound_ids = nullptr, *bound_scores = nullptr, *bound_counts = nullptr, *bound_map_sizes = nullptr;
  int32_t bound_chunk_q = 0;  // 0 = outputs are one chunk
  int32_t bound_nq = 0, bound_stride = 0;  // the batch's shape when the outputs were bound = what the caller sized them for
  hipEvent_t ev_unit_done = nullptr;  // sann_batch_run_after: recorded behind the unit kernel(s)
  hipEvent_t last_unit_done = nullptr;
  hipEvent_t ev_all_done = nullptr;  // sann_batch_run_after: recorded behind the merge kernel
  bool unit_done_recorded = false;
  int64_t bound_chunk_pitch = 0;
  // general path: workspace and candidate lists for the units it (re)runs, grown on demand
  std::vector<uint32_t> unit_bound;  // upper bound on the postings a unit can scan (host path; device path: on demand)
  int cap2 = 1;
  DevBuf cand_key2, cand_id2, g_units, g_off, g_slots, g_keys, g_dot, g_nsq, g_queries;
  int g_cap_units = 0;        // units the g_* / cand_*2 buffers can hold
  int64_t g_cap_entries = 0;  // table entries the g_keys/dot/nsq buffers can hold
  const uint32_t *cut_ptr[4] = {nullptr, nullptr, nullptr, nullptr};  // cached cut tables for up to 4 values of M
  int32_t cut_M[4] = {-1, -1, -1, -1};
  int32_t *h_status = nullptr;  // pinned: [0] overflow units, [1] inexact queries
  PinBuf h_qstat;               // pinned copy of q_stat (device path)
  bool qstat_pending = false;
  int64_t alg_bytes_host = 0;   // host path: n_scan * 12 summed over queries
  bool use_fast = false;
  FastParams fast{};
  sann_batch_stats_t stats{};
  bool ran = false;
  bool slow_tail_ran = false;  // the last sann_batch_finish re-ran units on the general path (results changed after the first merge)
  hipStream_t own_stream = nullptr;  // pooled batches (sann_get_tweet_candidates) run on a stream of their own
  // optional HIP-event timing of the kernels, on the stream they are launched on
  bool profiling = false;
  bool prof_unit_only = false;  // profiling level 1: bracket the dominant (unit) kernel only
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool ev_pending = false;
  double desc_ms_total = 0.0, unit_ms_total = 0.0, merge_ms_total = 0.0;
  int timed_runs = 0;

  ~sann_batch() {
    if (h_status) (void)hipHostFree(h_status);
    for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    if (ev_unit_done) (void)hipEventDestroy(ev_unit_done);
    if (ev_all_done) (void)hipEventDestroy(ev_all_done);
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }

  template <class T> const T *staged(size_t off) const { return (const T *)((const char *)stage.p + off); }
  template <class T> T *d_staged(size_t off) const { return (T *)((char *)d_stage.p + off); }

  BatchView view() const {
    BatchView b;
    b.hdr = hdr.as<QueryHdr>();
    b.scan_row = scan_row.as<int32_t>();
    b.scan_w = scan_w.as<double>();
    b.q_stat = (device_prep && use_fast) ? q_stat.as<uint4>() : nullptr;
    b.desc = desc.as<uint32_t>();
    b.scan_wq = scan_wq.as<double>();
    b.desc_stride = desc_row_stride(fast.max_n_scan);
    b.unit_T = unit_T.as<int32_t>();
    b.unit_pre = unit_T.as<uint32_t>() + n_units;  // (second half of the same allocation)
    for (int j = 0; j < 4; j++) { b.cut[j] = cut_ptr[j]; b.cut_M[j] = cut_M[j]; }
    b.nq = nq;
    b.cap = cap;
    b.cap2 = cap2;
    b.cand_key = cand_key.as<uint64_t>();
    b.cand_id = cand_id.as<int64_t>();
    b.cand_key2 = cand_key2.as<uint64_t>();
    b.cand_id2 = cand_id2.as<int64_t>();
    b.unit_fb = unit_fb.as<int32_t>();
    b.unit_thr = unit_thr.as<uint64_t>();
    b.cand_cnt = cand_cnt.as<int32_t>();
    b.unit_unique = unit_unique.as<int32_t>();
    b.unit_flags = unit_flags.as<uint32_t>();
    b.status = status.as<int32_t>();
    b.overflow_units = overflow_units.as<int32_t>();
    b.out_ids = bound_ids ? (int64_t *)bound_ids : out_ids.as<int64_t>();
    b.out_scores = bound_scores ? (double *)bound_scores : out_scores.as<double>();
    b.out_counts = bound_counts ? (int32_t *)bound_counts : out_counts.as<int32_t>();
    b.out_map_sizes = bound_map_sizes ? (int32_t *)bound_map_sizes : out_map_sizes.as<int32_t>();
    b.stride = stride;
    b.out_chunk_q = bound_chunk_q > 0 ? bound_chunk_q : (nq > 0 ? nq : 1);
    b.out_chunk_pitch = bound_chunk_q > 0 ? bound_chunk_pitch : 0;
    b.prof = prof.as<unsigned long long>();
    b.merge_done = merge_done.as<int32_t>();
    return b;
  }
};

extern "C" {

const char *sann_last_error(void) { return g_err.c_str(); }
const char *sann_runtime_advice(void) { return g_advice.c_str(); }
const char *sann_version(void) { return "simclusters_amd 0.1 (gfx950, fp64 parity layout w=16)"; }

// ---------------------------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------------------------
static int index_build_impl(const sann_index_options_t *opts, int32_t n_lists, const int32_t *cluster_ids,
                            const int64_t *list_offsets, const int64_t *tweet_ids, const double *scores,
                            const double *tweet_norms, sann_index_t **out) {
  if (!out) return fail(SANN_EINVAL, "out is NULL");
  *out = nullptr;
  if (!opts) return fail(SANN_EINVAL, "opts is NULL");
  if (n_lists < 0 || (n_lists > 0 && (!cluster_ids || !list_offsets)))
    return fail(SANN_EINVAL, "bad list arrays");
  int P = opts->n_partitions == 0 ? 32 : opts->n_partitions;
  if (P < 1 || P > 256 || (P & (P - 1))) return fail(SANN_EINVAL, "n_partitions must be a power of two in [1,256]");
  int n_shards = opts->n_shards <= 0 ? 1 : opts->n_shards;
  if (opts->shard_id < 0 || opts->shard_id >= n_shards) return fail(SANN_EINVAL, "shard_id out of range");
  for (int32_t i = 1; i < n_lists; i++)
    if (cluster_ids[i] <= cluster_ids[i - 1]) return fail(SANN_EINVAL, "cluster_ids must be ascending and unique");
  for (int32_t i = 0; i < n_lists; i++)
    if (list_offsets[i + 1] < list_offsets[i]) return fail(SANN_EINVAL, "list_offsets must be non-decreasing");
  int64_t total = n_lists ? list_offsets[n_lists] - list_offsets[0] : 0;
  if (total > 0 && (!tweet_ids || !scores)) return fail(SANN_EINVAL, "tweet_ids/scores are NULL");

  sann_index *ix = new (std::nothrow) sann_index();
  if (!ix) return fail(SANN_ENOMEM, "out of host memory");
  ix->device = opts->device;
  ix->P = P;
  ix->log2P = 0;
  while ((1 << ix->log2P) < P) ix->log2P++;
  ix->shard_id = opts->shard_id;
  ix->n_shards = n_shards;
  ix->n_postings_total = total;
  ix->cluster_ids.assign(cluster_ids, cluster_ids + n_lists);

  // pass 1: count postings per (row, partition) held by this shard
  std::vector<uint64_t> counts((size_t)n_lists * P + 1, 0);
  int32_t max_len = 0;
  for (int32_t r = 0; r < n_lists; r++) {
    int64_t b = list_offsets[r], e = list_offsets[r + 1];
    if (e - b > max_len) max_len = (int32_t)std::min<int64_t>(e - b, INT32_MAX);
    for (int64_t i = b; i < e; i++) {
      uint64_t h = mix64((uint64_t)tweet_ids[i]);
      if (tweet_shard(h, (uint32_t)n_shards) != (uint32_t)ix->shard_id) continue;
      counts[(size_t)r * P + tweet_partition(h, (uint32_t)P)]++;
    }
  }
  ix->max_list_len = max_len;
  // Tweet ids are keys of a Map in the store (TopKTweetsWithScores.topTweetsByFavClusterNormalizedScore): a list
  // cannot hold one twice, and the duplicate resolver of the unit kernel relies on it (one posting per tweet and
  // scanned cluster).  Checked here rather than trusted; order and sign of the scores are the caller's business
  // (position i is kept as given: ApproximateCosineSimilarity.scala:87 reads lists as they come).
  {
    std::vector<int64_t> tmp;
    for (int32_t r = 0; r < n_lists; r++) {
      const int64_t b = list_offsets[r], e = list_offsets[r + 1];
      if (e - b < 2) continue;
      tmp.assign(tweet_ids + b, tweet_ids + e);
      std::sort(tmp.begin(), tmp.end());
      if (std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end()) {
        delete ix;
        return fail(SANN_EINVAL, "cluster " + std::to_string(cluster_ids[r]) + ": a tweet id appears twice in one list");
      }
    }
  }
  uint64_t run = 0;
  ix->h_sub_offsets.resize((size_t)n_lists * P + 1);
  for (size_t i = 0; i < (size_t)n_lists * P; i++) {
    if (run > 0xffffffffull) break;
    ix->h_sub_offsets[i] = (uint32_t)run;
    run += counts[i];
  }
  if (run > 0xfffffff0ull) {
    delete ix;
    return fail(SANN_ELIMIT, "more than 2^32 postings in one shard; use more shards");
  }
  ix->h_sub_offsets[(size_t)n_lists * P] = (uint32_t)run;
  ix->n_postings = (int64_t)run;

  // pass 2: fill, keeping list order inside every sub-list (ranks ascending)
  std::vector<Posting> h_post((size_t)run);
  std::vector<uint32_t> h_rank((size_t)run);
  std::vector<double> h_norm(tweet_norms ? (size_t)run : 0);
  std::vector<uint32_t> cursor(ix->h_sub_offsets.begin(), ix->h_sub_offsets.end() - 1);
  for (int32_t r = 0; r < n_lists; r++) {
    int64_t b = list_offsets[r], e = list_offsets[r + 1];
    for (int64_t i = b; i < e; i++) {
      uint64_t h = mix64((uint64_t)tweet_ids[i]);
      if (tweet_shard(h, (uint32_t)n_shards) != (uint32_t)ix->shard_id) continue;
      uint32_t &c = cursor[(size_t)r * P + tweet_partition(h, (uint32_t)P)];
      h_post[c].id = tweet_ids[i];
      h_post[c].score = scores[i];
      if (tweet_norms) h_norm[c] = tweet_norms[i];
      h_rank[c] = (uint32_t)std::min<int64_t>(i - b, 0xffffffffll);
      c++;
    }
  }

  hipError_t e = hipSetDevice(ix->device);
  if (e == hipSuccess) e = ix->postings.alloc(std::max<size_t>(h_post.size(), 1) * sizeof(Posting));
  if (e == hipSuccess) e = ix->ranks.alloc(std::max<size_t>(h_rank.size(), 1) * sizeof(uint32_t));
  if (e == hipSuccess) e = ix->sub_offsets.alloc(ix->h_sub_offsets.size() * sizeof(uint32_t));
  if (e == hipSuccess && !h_post.empty())
    e = hipMemcpy(ix->postings.p, h_post.data(), h_post.size() * sizeof(Posting), hipMemcpyHostToDevice);
  if (e == hipSuccess && !h_rank.empty())
    e = hipMemcpy(ix->ranks.p, h_rank.data(), h_rank.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(ix->sub_offsets.p, ix->h_sub_offsets.data(), ix->h_sub_offsets.size() * sizeof(uint32_t),
                  hipMemcpyHostToDevice);
  if (e == hipSuccess && tweet_norms) {
    e = ix->norms.alloc(std::max<size_t>(h_norm.size(), 1) * sizeof(double));
    if (e == hipSuccess && !h_norm.empty())
      e = hipMemcpy(ix->norms.p, h_norm.data(), h_norm.size() * sizeof(double), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    delete ix;
    return fail(SANN_EDEVICE, std::string("index upload: ") + hipGetErrorString(e));
  }
  *out = ix;
  return SANN_OK;
}

int sann_index_build(const sann_index_options_t *opts, int32_t n_lists, const int32_t *cluster_ids,
                     const int64_t *list_offsets, const int64_t *tweet_ids, const double *scores, sann_index_t **out) try {
  return index_build_impl(opts, n_lists, cluster_ids, list_offsets, tweet_ids, scores, nullptr, out);
} ABI_CATCH

int sann_index_build_with_norms(const sann_index_options_t *opts, int32_t n_lists, const int32_t *cluster_ids,
                                const int64_t *list_offsets, const int64_t *tweet_ids, const double *scores,
                                const double *tweet_norms, sann_index_t **out) try {
  if (!tweet_norms && n_lists > 0 && list_offsets && list_offsets[n_lists] > list_offsets[0])
    return fail(SANN_EINVAL, "tweet_norms is NULL");
  static const double none = 0.0;
  return index_build_impl(opts, n_lists, cluster_ids, list_offsets, tweet_ids, scores, tweet_norms ? tweet_norms : &none, out);
} ABI_CATCH

int sann_index_info(const sann_index_t *ix, sann_index_info_t *info) try {
  if (!ix || !info) return fail(SANN_EINVAL, "NULL argument");
  info->n_clusters = (int64_t)ix->cluster_ids.size();
  info->n_postings = ix->n_postings;
  info->n_postings_total = ix->n_postings_total;
  info->device_bytes = (int64_t)(ix->postings.bytes + ix->ranks.bytes + ix->sub_offsets.bytes + ix->norms.bytes);
  info->n_partitions = ix->P;
  info->shard_id = ix->shard_id;
  info->n_shards = ix->n_shards;
  info->max_list_len = ix->max_list_len;
  return SANN_OK;
} ABI_CATCH

int sann_index_get_list(const sann_index_t *ix, int32_t cluster_id, int64_t cap, int64_t *tweet_ids, double *scores,
                        int32_t *ranks, int64_t *n) try {
  if (!ix || !n) return fail(SANN_EINVAL, "NULL argument");
  *n = 0;
  int row = ix->row_of(cluster_id);
  if (row < 0) return SANN_OK;
  uint32_t b = ix->h_sub_offsets[(size_t)row * ix->P], e = ix->h_sub_offsets[(size_t)(row + 1) * ix->P];
  int64_t len = (int64_t)e - b;
  *n = len;
  if (len == 0 || cap <= 0) return SANN_OK;
  HIP_TRY(hipSetDevice(ix->device));
  std::vector<Posting> p((size_t)len);
  std::vector<uint32_t> r((size_t)len);
  HIP_TRY(hipMemcpy(p.data(), ix->postings.as<Posting>() + b, (size_t)len * sizeof(Posting), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(r.data(), ix->ranks.as<uint32_t>() + b, (size_t)len * sizeof(uint32_t), hipMemcpyDeviceToHost));
  // sub-lists are concatenated partition by partition; give them back in list (rank) order
  std::vector<int64_t> order((size_t)len);
  for (int64_t i = 0; i < len; i++) order[(size_t)i] = i;
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t c) { return r[(size_t)a] < r[(size_t)c]; });
  for (int64_t i = 0; i < len && i < cap; i++) {
    size_t o = (size_t)order[(size_t)i];
    if (tweet_ids) tweet_ids[i] = p[o].id;
    if (scores) scores[i] = p[o].score;
    if (ranks) ranks[i] = (int32_t)r[o];
  }
  return SANN_OK;
} ABI_CATCH

int sann_index_destroy(sann_index_t *ix) try {
  if (!ix) return SANN_OK;
  (void)hipSetDevice(ix->device);
  delete ix;
  return SANN_OK;
} ABI_CATCH

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// batch
// ---------------------------------------------------------------------------------------------
namespace {

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// (Re)prepare `b` for nq new queries, reusing its device buffers, and enqueue on `st` whatever the preparation needs on
// the device (one packed H2D copy + prep_kernel on the device path).  Asynchronous on the device path.
int batch_reset(sann_batch *b, hipStream_t st, int64_t now_ms, int32_t nq, const int64_t *emb_offsets,
                const int32_t *emb_cluster_ids, const double *emb_scores, const int64_t *source_tweet_ids,
                const uint8_t *has_source_tweet, const sann_config_t *configs, int32_t n_configs,
                const int64_t *scan_offsets, const int32_t *scan_cluster_ids, const int64_t *now_ms_q = nullptr) {
  sann_index *ix = b->ix;
  const int variant = b->variant;
  if (nq < 0) return fail(SANN_EINVAL, "nq < 0");
  if (nq > 0 && (!emb_offsets || !configs)) return fail(SANN_EINVAL, "emb_offsets/configs are NULL");
  if (n_configs != 1 && n_configs != nq) return fail(SANN_EINVAL, "n_configs must be 1 or nq");
  // the explicit scan keys come as a CSR pair: one without the other is a caller bug (silently falling back to the
  // default truncation would answer a different question with SANN_OK); an all-empty region may leave the ids NULL
  if (nq > 0 && (scan_offsets == nullptr) != (scan_cluster_ids == nullptr) &&
      !(scan_offsets && scan_offsets[nq] == scan_offsets[0]))
    return fail(SANN_EINVAL, "scan_offsets and scan_cluster_ids must be given together");
  if ((int64_t)nq * ix->P > (int64_t)INT32_MAX / 2) return fail(SANN_ELIMIT, "nq * n_partitions too large");
  // (offsets with an all-empty region and no ids: explicit keys, none of them -- every query scans nothing)
  const bool has_scan = scan_offsets != nullptr && (scan_cluster_ids != nullptr || nq == 0 || scan_offsets[nq] == scan_offsets[0]);
  const bool has_sources = has_source_tweet != nullptr && source_tweet_ids != nullptr;

  // ---- O(nq) pass over the arguments: validation, the scan regions' upper bounds, k, M ----------------------------
  int kmax = 1;
  bool any_norms = false;
  int64_t max_emb = 0, total_ub = 0;
  int max_ub = 0;
  double apriori_mean = 0.0;  // largest expected unit size over the queries (see the geometry note below)
  for (int32_t i = 0; i < n_configs; i++) {
    const sann_config_t &cfg = configs[i];
    if (variant == SANN_VARIANT_LEGACY) {
      if (cfg.ann_algorithm != SANN_ALG_DOT_PRODUCT && cfg.ann_algorithm != SANN_ALG_COSINE &&
          cfg.ann_algorithm != SANN_ALG_LOG_COSINE)
        return fail(SANN_EINVAL, "legacy variant: ann_algorithm must be dot product, cosine or log cosine");
      if (cfg.max_num_results > 1000) return fail(SANN_ELIMIT, "legacy variant: max_num_results above 1000");
    }
    int k = cfg.max_num_results < 1000 ? cfg.max_num_results : 1000;
    kmax = std::max(kmax, k);
    if (cfg.ann_algorithm == SANN_ALG_OFFLINE_LOG_COSINE || cfg.ann_algorithm == SANN_ALG_OFFLINE_COSINE) {
      if (!ix->norms.p) return fail(SANN_EINVAL, "offline scoring needs an index built with sann_index_build_with_norms");
      if (variant == SANN_VARIANT_LEGACY) return fail(SANN_EINVAL, "offline scoring is not a form of the legacy variant");
      any_norms = true;
    }
  }
  HIP_TRY(hipSetDevice(ix->device));
  // staging layout (one pinned block, one H2D copy)
  sann_batch::StageLayout L;
  L.has_scan = has_scan;
  L.has_sources = has_sources;
  L.has_now_q = now_ms_q != nullptr && nq > 0;
  L.n_configs = n_configs;
  const int64_t e0 = nq ? emb_offsets[0] : 0, e1 = nq ? emb_offsets[nq] : 0;
  const int64_t s0 = (nq && has_scan) ? scan_offsets[0] : 0, s1 = (nq && has_scan) ? scan_offsets[nq] : 0;
  if (e1 < e0 || s1 < s0) return fail(SANN_EINVAL, "offsets must be non-decreasing");
  if (e1 > e0 && (!emb_cluster_ids || !emb_scores)) return fail(SANN_EINVAL, "embedding arrays are NULL");
  {
    size_t o = 0;
    L.emb_offsets = o; o = align16(o + ((size_t)nq + 1) * 8);
    L.src_ids = o; o = align16(o + (has_sources ? (size_t)nq * 8 : 0));
    L.scan_offsets = o; o = align16(o + (has_scan ? ((size_t)nq + 1) * 8 : 0));
    L.emb_scores = o; o = align16(o + (size_t)(e1 - e0) * 8);
    L.emb_cids = o; o = align16(o + (size_t)(e1 - e0) * 4);
    L.scan_cids = o; o = align16(o + (size_t)(s1 - s0) * 4);
    L.configs = o; o = align16(o + (size_t)n_configs * sizeof(sann_config_t));
    L.scan_begin = o; o = align16(o + (size_t)nq * 4);
    L.has_src = o; o = align16(o + (has_sources ? (size_t)nq : 0));
    L.now_q = o; o = align16(o + (now_ms_q ? (size_t)nq * 8 : 0));
    L.bytes = o;
  }
  HIP_TRY(b->stage.reserve(std::max<size_t>(L.bytes, 16)));
  char *S = (char *)b->stage.p;
  int64_t *st_eo = (int64_t *)(S + L.emb_offsets);
  int64_t *st_so = (int64_t *)(S + L.scan_offsets);
  int32_t *st_sb = (int32_t *)(S + L.scan_begin);
  for (int32_t q = 0; q < nq; q++) {
    const sann_config_t &cfg = configs[n_configs == 1 ? 0 : q];
    const int64_t n_raw = emb_offsets[q + 1] - emb_offsets[q];
    if (n_raw < 0) return fail(SANN_EINVAL, "emb_offsets must be non-decreasing");
    max_emb = std::max(max_emb, n_raw);
    int64_t ub;
    if (has_scan) {
      ub = scan_offsets[q + 1] - scan_offsets[q];
      if (ub < 0) return fail(SANN_EINVAL, "scan_offsets must be non-decreasing");
    } else {
      ub = cfg.max_scan_clusters < 0 ? 0 : std::min<int64_t>(n_raw, cfg.max_scan_clusters);
    }
    if (total_ub + ub > (int64_t)INT32_MAX / 4) return fail(SANN_ELIMIT, "too many scanned clusters in one batch");
    st_eo[q] = emb_offsets[q] - e0;
    if (has_scan) st_so[q] = scan_offsets[q] - s0;
    st_sb[q] = (int32_t)total_ub;
    total_ub += ub;
    max_ub = (int)std::max<int64_t>(max_ub, std::min<int64_t>(ub, INT32_MAX));
    const double M = (double)std::max(cfg.max_top_tweets_per_cluster, 0);
    apriori_mean = std::max(apriori_mean, (double)ub * std::min(M, (double)ix->max_list_len));
  }
  if (nq) {
    st_eo[nq] = e1 - e0;
    if (has_scan) st_so[nq] = s1 - s0;
  }
  apriori_mean /= (double)ix->P * (double)ix->n_shards;

  // caller-bound output buffers were sized for the batch's shape at bind time (nq rows of `stride` entries, and for
  // the chunked form ceil(nq / queries_per_chunk) chunks): a reset must fit inside it, or the merge kernels would
  // write past the caller's buffers
  if (b->bound_ids && (nq > b->bound_nq || kmax > b->bound_stride))
    return fail(SANN_EINVAL, "the batch's outputs are bound to caller-owned buffers sized for " + std::to_string(b->bound_nq) +
                                 " queries x " + std::to_string(b->bound_stride) + " results; this reset needs " + std::to_string(nq) +
                                 " x " + std::to_string(kmax) + " (unbind or bind larger buffers first)");
  b->nq = nq;
  b->now_ms = now_ms;
  b->n_units = nq * ix->P;
  b->lay = L;
  b->stride = kmax;
  b->cap2 = kmax;
  b->cap = FAST_SCAP;  // a fast unit emits at most its survivor list
  b->ran = false;
  b->unit_done_recorded = false;
  b->stats = sann_batch_stats_t{};
  b->stats.n_units = b->n_units;
  b->unit_bound.clear();
  {
    const char *force = getenv("SANN_FORCE_GENERAL");
    b->use_fast = !(force && force[0] == '1');
    const char *hp = getenv("SANN_HOST_PREP");
    b->device_prep = b->use_fast && max_emb <= PREP_MAX && !(hp && hp[0] == '1');
  }

  // the packed inputs are kept either way: the device path uploads them, the slow tail of the host path never needs
  // them, and the slow tail of the device path re-prepares the flagged queries from them
  if (nq) {
    memcpy(S + L.emb_scores, emb_scores + e0, (size_t)(e1 - e0) * 8);
    memcpy(S + L.emb_cids, emb_cluster_ids + e0, (size_t)(e1 - e0) * 4);
    if (has_scan) memcpy(S + L.scan_cids, scan_cluster_ids + s0, (size_t)(s1 - s0) * 4);
    memcpy(S + L.configs, configs, (size_t)n_configs * sizeof(sann_config_t));
    if (has_sources) {
      memcpy(S + L.src_ids, source_tweet_ids, (size_t)nq * 8);
      memcpy(S + L.has_src, has_source_tweet, (size_t)nq);
    }
    if (L.has_now_q) memcpy(S + L.now_q, now_ms_q, (size_t)nq * 8);
  }

  // ---- device buffers (kept between resets; they only grow) -------------------------------------------------------
  const size_t nu = (size_t)std::max(b->n_units, 1), nqz = (size_t)std::max(nq, 1);
  const size_t scan_cap = (size_t)std::max<int64_t>(total_ub, 1);
  HIP_TRY(b->hdr.reserve(nqz * sizeof(QueryHdr)));
  HIP_TRY(b->scan_row.reserve(scan_cap * 4));
  HIP_TRY(b->scan_w.reserve(scan_cap * 8));
  HIP_TRY(b->desc.reserve(nu * (size_t)desc_row_stride(max_ub) * 8));
  HIP_TRY(b->scan_wq.reserve(nqz * (size_t)desc_row_stride(max_ub) * 8));
  HIP_TRY(b->unit_T.reserve(nu * 8));  // unit_T and unit_pre
  HIP_TRY(b->d_k.reserve(nqz * 4));
  HIP_TRY(b->q_stat.reserve(nqz * 16));
  HIP_TRY(b->cand_key.reserve(nu * (size_t)b->cap * 8));
  HIP_TRY(b->cand_id.reserve(nu * (size_t)b->cap * 8));
  HIP_TRY(b->cand_cnt.reserve(nu * 4));
  HIP_TRY(b->unit_unique.reserve(nu * 4));
  HIP_TRY(b->unit_flags.reserve(nu * 4));
  HIP_TRY(b->unit_fb.reserve(nu * 4));
  HIP_TRY(b->unit_thr.reserve(nu * 16));
  HIP_TRY(b->status.reserve((nqz + 2) * 4));
  HIP_TRY(b->overflow_units.reserve(nu * 4));
  HIP_TRY(b->out_ids.reserve(nqz * (size_t)b->stride * 8));
  HIP_TRY(b->out_scores.reserve(nqz * (size_t)b->stride * 8));
  HIP_TRY(b->out_counts.reserve(nqz * 4));
  HIP_TRY(b->out_map_sizes.reserve(nqz * 4));
  HIP_TRY(b->merge_done.reserve(nqz * 4));
  HIP_TRY(b->h_qstat.reserve(nqz * 16));
  if (!b->h_status) HIP_TRY(hipHostMalloc((void **)&b->h_status, 2 * 4, hipHostMallocDefault));
  b->h_status[0] = b->h_status[1] = 0;
  // the general path's workspace is sized per batch (tables depend on the units' bounds)
  b->g_cap_units = 0;
  b->g_cap_entries = 0;

  // ---- cut tables for the batch's (up to 4) distinct values of M, built once per index and M ------------------------
  for (int j = 0; j < 4; j++) { b->cut_M[j] = -1; b->cut_ptr[j] = nullptr; }
  if (b->use_fast && nq > 0) {
    int n_m = 0;
    for (int32_t i = 0; i < n_configs && n_m < 4; i++) {
      const int32_t M = std::max(configs[i].max_top_tweets_per_cluster, 0);
      bool seen = false;
      for (int j = 0; j < n_m; j++) seen = seen || b->cut_M[j] == M;
      if (seen) continue;
      const uint32_t *tab = nullptr;
      {
        std::lock_guard<std::mutex> lk(ix->cut_mu);
        for (auto &e : ix->cut_cache)
          if (e.first == M) tab = e.second->as<uint32_t>();
        if (!tab && ix->cut_cache.size() < 8) {
          std::unique_ptr<DevBuf> buf(new DevBuf());
          HIP_TRY(buf->alloc(std::max<size_t>((size_t)ix->cluster_ids.size() * ix->P, 1) * 4));
          HIP_TRY(launch_cut(ix->view(), M, buf->as<uint32_t>(), st));
          HIP_TRY(hipStreamSynchronize(st));
          tab = buf->as<uint32_t>();
          ix->cut_cache.emplace_back(M, std::move(buf));
        }
      }
      if (tab) {
        b->cut_M[n_m] = M;
        b->cut_ptr[n_m] = tab;
        n_m++;
      }
    }
  }
  b->fast.k_local = 0;
  b->fast.max_n_scan = max_ub;
  b->fast.use_norms = any_norms ? 1 : 0;

  if (b->device_prep) {
    // ---- device path: upload the packed inputs, prepare on the GPU ------------------------------------------------
    // Geometry (postings a unit holds in registers).  The host no longer sees which clusters a query scans, so the
    // choice is made from what it does know: a query scans at most ub clusters and at most min(M, longest list)
    // postings of each, and a posting falls into this (shard, partition) with probability 1 / (n_shards * P): mean
    // <= ub * min(M, max_len) / (n_shards * P), variance <= mean; five sigma of headroom keeps the expected number of
    // overflowing units of a 32k-unit batch below 0.01.  That bound is loose for queries over short lists, so the
    // index also remembers the largest unit recent batches really had (q_stat, read back with every batch) and the
    // smaller of the two geometries is used; a unit that does overflow is re-run exactly on the general path.
    int ucap = geometry_for(apriori_mean + 5.0 * std::sqrt(apriori_mean));
    const int hint = ix->unit_size_hint.load(std::memory_order_relaxed);
    if (hint > 0) ucap = std::min(ucap, geometry_for((double)hint + 2.0 * std::sqrt((double)hint)));
    if (const char *ov = getenv("SANN_UNIT_CAP")) ucap = atoi(ov);  // tuning / test override
    b->fast.unit_capacity = ucap;
    HIP_TRY(ix->ensure_device_cluster_ids());
    if (nq > 0) {
      HIP_TRY(b->d_stage.reserve(L.bytes));
      HIP_TRY(hipMemcpyAsync(b->d_stage.p, b->stage.p, L.bytes, hipMemcpyHostToDevice, st));
      PrepView pv;
      pv.emb_offsets = b->d_staged<int64_t>(L.emb_offsets);
      pv.emb_cluster_ids = b->d_staged<int32_t>(L.emb_cids);
      pv.emb_scores = b->d_staged<double>(L.emb_scores);
      pv.source_tweet_ids = has_sources ? b->d_staged<int64_t>(L.src_ids) : nullptr;
      pv.has_source_tweet = has_sources ? b->d_staged<uint8_t>(L.has_src) : nullptr;
      pv.configs = b->d_staged<sann_config_t>(L.configs);
      pv.scan_offsets = has_scan ? b->d_staged<int64_t>(L.scan_offsets) : nullptr;
      pv.scan_cluster_ids = has_scan ? b->d_staged<int32_t>(L.scan_cids) : nullptr;
      pv.scan_begin = b->d_staged<int32_t>(L.scan_begin);
      pv.cluster_ids = ix->d_cluster_ids.as<int32_t>();
      pv.hdr = b->hdr.as<QueryHdr>();
      pv.scan_row = b->scan_row.as<int32_t>();
      pv.scan_w = b->scan_w.as<double>();
      pv.d_k = b->d_k.as<int32_t>();
      pv.now_ms = now_ms;
      pv.now_ms_q = L.has_now_q ? b->d_staged<int64_t>(L.now_q) : nullptr;
      pv.n_rows = (int32_t)ix->cluster_ids.size();
      pv.n_configs = n_configs;
      pv.variant = variant;
      pv.nq = nq;
      HIP_TRY(launch_prep(pv, st));
    }
    return SANN_OK;
  }

  // ---- host path ------------------------------------------------------------------------------------------------
  b->h_hdr.resize((size_t)nq);
  b->h_k.resize((size_t)nq);
  b->h_scan_row.clear();
  b->h_scan_w.clear();
  std::vector<IdScore> emb, by_id;
  std::vector<int32_t> keys;
  HostQuery hq;
  int64_t postings_scanned = 0, alg_bytes = 0;
  std::vector<uint64_t> unit_bound((size_t)b->n_units, 0);
  std::vector<double> unit_est((size_t)b->n_units, 0.0), unit_var((size_t)b->n_units, 0.0);
  int max_n_scan = 0;
  for (int32_t q = 0; q < nq; q++) {
    const sann_config_t &cfg = configs[n_configs == 1 ? 0 : q];
    const bool has_src = has_sources && has_source_tweet[q];
    int rc = prepare_query_host(ix, variant, now_ms_q ? now_ms_q[q] : now_ms, cfg, emb_cluster_ids + emb_offsets[q], emb_scores + emb_offsets[q],
                                emb_offsets[q + 1] - emb_offsets[q], has_src, has_src ? source_tweet_ids[q] : 0,
                                has_scan ? scan_cluster_ids + scan_offsets[q] : nullptr,
                                has_scan ? scan_offsets[q + 1] - scan_offsets[q] : -1, hq, emb, by_id, keys);
    if (rc != SANN_OK) return rc;
    hq.h.scan_begin = (int32_t)b->h_scan_row.size();
    b->h_hdr[(size_t)q] = hq.h;
    b->h_k[(size_t)q] = hq.h.k;
    b->h_scan_row.insert(b->h_scan_row.end(), hq.rows.begin(), hq.rows.end());
    b->h_scan_w.insert(b->h_scan_w.end(), hq.w.begin(), hq.w.end());
    for (int p = 0; p < ix->P; p++) {
      unit_bound[(size_t)q * ix->P + p] = hq.unit_bound[(size_t)p];
      unit_est[(size_t)q * ix->P + p] = hq.unit_est[(size_t)p];
      unit_var[(size_t)q * ix->P + p] = hq.unit_var[(size_t)p];
    }
    postings_scanned += hq.postings_scanned;
    max_n_scan = std::max(max_n_scan, (int)hq.h.n_scan);
    alg_bytes += (int64_t)hq.h.n_scan * 12;
  }
  // postings_scanned = sum_c min(len_c, M) (SURVEY 8d's P_q) when the shard holds whole lists;
  // with tweet-hash shards it is the per-sub-list upper bound sum_p min(len_p, M).
  b->stats.postings_scanned = postings_scanned;
  b->stats.algorithmic_bytes = alg_bytes + postings_scanned * 16;
  b->unit_bound.resize((size_t)b->n_units);
  for (int u = 0; u < b->n_units; u++) b->unit_bound[(size_t)u] = (uint32_t)std::min<uint64_t>(unit_bound[(size_t)u], 0x7fffffffu);
  // Geometry: the smallest one under which fewer than 0.1 units of the batch are expected to overflow (a unit's count
  // of postings with rank < M is its sub-lists' lengths thinned with probability frac: mean and variance are known,
  // normal tail); a unit that does overflow goes to the general path on its own.
  {
    static const int kCaps[] = {256, 512, 768, 1024, 1536, 2048, 3072, 4096};
    int ucap = 4096;
    for (int c : kCaps) {
      double expected_overflows = 0.0;
      for (size_t u = 0; u < unit_est.size() && expected_overflows < 0.1; u++) {
        const double room = (double)c - 16.0 - unit_est[u];
        if (room <= 0.0) { expected_overflows += 1.0; continue; }
        if (unit_var[u] > 0.0) expected_overflows += 0.5 * std::erfc(room / std::sqrt(2.0 * unit_var[u]));
      }
      if (expected_overflows < 0.1) { ucap = c; break; }
    }
    if (const char *ov = getenv("SANN_UNIT_CAP")) ucap = atoi(ov);
    b->fast.unit_capacity = ucap;
    b->fast.max_n_scan = max_n_scan;
  }
  if (nq > 0) {
    HIP_TRY(hipMemcpyAsync(b->hdr.p, b->h_hdr.data(), (size_t)nq * sizeof(QueryHdr), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_k.p, b->h_k.data(), (size_t)nq * 4, hipMemcpyHostToDevice, st));
  }
  if (!b->h_scan_row.empty()) {
    HIP_TRY(hipMemcpyAsync(b->scan_row.p, b->h_scan_row.data(), b->h_scan_row.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->scan_w.p, b->h_scan_w.data(), b->h_scan_w.size() * 8, hipMemcpyHostToDevice, st));
  }
  HIP_TRY(hipStreamSynchronize(st));  // pageable sources: done with them before returning
  return SANN_OK;
}

}  // namespace

extern "C" {

int sann_batch_create(sann_index_t *ix, int32_t variant, int64_t now_ms, int32_t nq, const int64_t *emb_offsets,
                      const int32_t *emb_cluster_ids, const double *emb_scores, const int64_t *source_tweet_ids,
                      const uint8_t *has_source_tweet, const sann_config_t *configs, int32_t n_configs,
                      const int64_t *scan_offsets, const int32_t *scan_cluster_ids, sann_batch_t **out) try {
  if (!out) return fail(SANN_EINVAL, "out is NULL");
  *out = nullptr;
  if (!ix) return fail(SANN_EINVAL, "index is NULL");
  if (variant < 0 || variant > 3) return fail(SANN_EINVAL, "unknown variant");
  sann_batch *b = new (std::nothrow) sann_batch();
  if (!b) return fail(SANN_ENOMEM, "out of host memory");
  b->ix = ix;
  b->variant = variant;
  int rc = batch_reset(b, nullptr, now_ms, nq, emb_offsets, emb_cluster_ids, emb_scores, source_tweet_ids, has_source_tweet,
                       configs, n_configs, scan_offsets, scan_cluster_ids);
  // the batch may be run on any stream afterwards, non-blocking ones included: finish the preparation here
  if (rc == SANN_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(SANN_EDEVICE, "preparing the batch failed");
  if (rc != SANN_OK) {
    std::string keep = g_err;
    (void)hipSetDevice(ix->device);
    delete b;
    g_err = keep;
    return rc;
  }
  *out = b;
  return SANN_OK;
} ABI_CATCH

int sann_batch_reset(sann_batch_t *b, void *hip_stream, int64_t now_ms, int32_t nq, const int64_t *emb_offsets,
                     const int32_t *emb_cluster_ids, const double *emb_scores, const int64_t *source_tweet_ids,
                     const uint8_t *has_source_tweet, const sann_config_t *configs, int32_t n_configs,
                     const int64_t *scan_offsets, const int32_t *scan_cluster_ids) try {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  return batch_reset(b, (hipStream_t)hip_stream, now_ms, nq, emb_offsets, emb_cluster_ids, emb_scores, source_tweet_ids,
                     has_source_tweet, configs, n_configs, scan_offsets, scan_cluster_ids);
} ABI_CATCH

}  // extern "C"

namespace {

// Device-prepared batches: the host never saw which clusters the queries scan.  The slow tail needs, for the queries
// it re-runs, an upper bound on every unit's postings (the general path's table sizes): prepare just those queries
// on the host, from the packed copy of the caller's arrays.
int ensure_unit_bounds(sann_batch *b, const std::vector<int32_t> &queries) {
  if (!b->device_prep) return SANN_OK;
  const int P = b->ix->P;
  if (b->unit_bound.size() != (size_t)b->n_units) b->unit_bound.assign((size_t)b->n_units, 0u);
  const auto &L = b->lay;
  const int64_t *eo = b->staged<int64_t>(L.emb_offsets);
  const int64_t *so = b->staged<int64_t>(L.scan_offsets);
  const int32_t *ec = b->staged<int32_t>(L.emb_cids);
  const double *es = b->staged<double>(L.emb_scores);
  const int32_t *sc = b->staged<int32_t>(L.scan_cids);
  const sann_config_t *cfgs = b->staged<sann_config_t>(L.configs);
  const int64_t *src = b->staged<int64_t>(L.src_ids);
  const uint8_t *has = b->staged<uint8_t>(L.has_src);
  std::vector<IdScore> emb, by_id;
  std::vector<int32_t> keys;
  HostQuery hq;
  for (int32_t q : queries) {
    const bool has_src = L.has_sources && has[q];
    int rc = prepare_query_host(b->ix, b->variant, L.has_now_q ? b->staged<int64_t>(L.now_q)[q] : b->now_ms, cfgs[L.n_configs == 1 ? 0 : q], ec + eo[q], es + eo[q],
                                eo[q + 1] - eo[q], has_src, has_src ? src[q] : 0, L.has_scan ? sc + so[q] : nullptr,
                                L.has_scan ? so[q + 1] - so[q] : -1, hq, emb, by_id, keys);
    if (rc != SANN_OK) return rc;
    for (int p = 0; p < P; p++)
      b->unit_bound[(size_t)q * P + p] = (uint32_t)std::min<uint64_t>(hq.unit_bound[(size_t)p], 0x7fffffffu);
  }
  return SANN_OK;
}

// status[0] / status[1] as the kernels left them: counts of list entries, bounded by the batch's shape
int check_status_counts(int64_t n_over, int64_t n_inexact, int64_t n_units, int64_t nq) {
  if (n_over < 0 || n_over > n_units || n_inexact < 0 || n_inexact > nq)
    return fail(SANN_EINTERNAL, "device status block out of range: overflow units " + std::to_string(n_over) + " of " +
                                    std::to_string(n_units) + ", inexact queries " + std::to_string(n_inexact) + " of " +
                                    std::to_string(nq));
  return SANN_OK;
}

// What the slow tail re-runs, from the lists the kernels left: `over` = units the fast path could not hold, `inexact` =
// queries whose top-k the merge could not prove.  An inexact query is re-run whole, an overflowed unit alone; `queries`
// = every query that needs its merge repeated.  Pure host arithmetic on DEVICE-WRITTEN values: every id is range-checked
// (a corrupted entry is SANN_EINTERNAL, not an out-of-bounds write), duplicates are tolerated.
int plan_slow_tail(int nq, int P, const std::vector<int32_t> &over, const std::vector<int32_t> &inexact,
                   std::vector<int32_t> &units, std::vector<int32_t> &queries) {
  units.clear();
  queries.clear();
  if (nq < 0 || P < 1) return fail(SANN_EINTERNAL, "slow tail: bad batch shape");
  const int64_t n_units = (int64_t)nq * P;
  std::vector<uint8_t> q_mark((size_t)nq, 0), q_full((size_t)nq, 0);
  for (int32_t q : inexact) {
    if (q < 0 || q >= nq) return fail(SANN_EINTERNAL, "slow tail: inexact query id " + std::to_string(q) + " outside the batch");
    q_mark[(size_t)q] = 1;
    q_full[(size_t)q] = 1;
  }
  for (int32_t u : over) {
    if (u < 0 || (int64_t)u >= n_units) return fail(SANN_EINTERNAL, "slow tail: overflow unit id " + std::to_string(u) + " outside the batch");
    const int q = u / P;
    q_mark[(size_t)q] = 1;
    if (!q_full[(size_t)q]) units.push_back(u);
  }
  for (int q = 0; q < nq; q++) {
    if (q_full[(size_t)q])
      for (int p = 0; p < P; p++) units.push_back(q * P + p);
    if (q_mark[(size_t)q]) queries.push_back(q);
  }
  std::sort(units.begin(), units.end());
  units.erase(std::unique(units.begin(), units.end()), units.end());
  return SANN_OK;
}

// Run the general (global-memory table) kernel on `units` (empty = every unit of the batch) and
// leave their candidate lists in the cand_*2 buffers.  Buffers grow on demand.
int run_general(sann_batch *b, const std::vector<int32_t> &units, hipStream_t st) {
  const bool all = units.empty();
  const int n = all ? b->n_units : (int)units.size();
  if (n == 0) return SANN_OK;
  std::vector<int64_t> off((size_t)n);
  std::vector<uint32_t> slots((size_t)n);
  int64_t run = 0;
  for (int i = 0; i < n; i++) {
    const int u = all ? i : units[(size_t)i];
    uint32_t S = next_pow2_u32(2ull * b->unit_bound[(size_t)u] + 1);
    off[(size_t)i] = run;
    slots[(size_t)i] = S;
    run += (int64_t)S + 1;
  }
  HIP_TRY(b->g_units.reserve((size_t)n * 4));
  HIP_TRY(b->g_off.reserve((size_t)n * 8));
  HIP_TRY(b->g_slots.reserve((size_t)n * 4));
  HIP_TRY(b->cand_key2.reserve((size_t)n * (size_t)b->cap2 * 8));
  HIP_TRY(b->cand_id2.reserve((size_t)n * (size_t)b->cap2 * 8));
  HIP_TRY(b->g_keys.reserve((size_t)run * 8));
  HIP_TRY(b->g_dot.reserve((size_t)run * 8));
  HIP_TRY(b->g_nsq.reserve((size_t)run * 8));
  // the tiny descriptor arrays are copied synchronously (pageable host memory)
  if (!all) HIP_TRY(hipMemcpy(b->g_units.p, units.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->g_off.p, off.data(), (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->g_slots.p, slots.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemsetAsync(b->g_keys.p, 0xFF, (size_t)run * 8, st));
  GeneralWs ws;
  ws.units = all ? nullptr : b->g_units.as<int32_t>();
  ws.ws_off = b->g_off.as<int64_t>();
  ws.ws_slots = b->g_slots.as<uint32_t>();
  ws.keys = b->g_keys.as<int64_t>();
  ws.dot = b->g_dot.as<double>();
  ws.nsq = b->g_nsq.as<double>();
  HIP_TRY(launch_unit_general(b->ix->view(), b->view(), ws, n, st));
  return SANN_OK;
}

}  // namespace

extern "C" {

static int batch_run(sann_batch_t *b, void *hip_stream, bool chained, sann_batch_t *after, bool whole) {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  if (after && after->ix->device != b->ix->device) return fail(SANN_EINVAL, "batches on different devices");
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(hipSetDevice(b->ix->device));
  if (chained || b->own_stream) check_hw_queues_once();
  if (chained && !b->ev_unit_done) HIP_TRY(hipEventCreateWithFlags(&b->ev_unit_done, hipEventDisableTiming));
  if (chained && !b->ev_all_done) HIP_TRY(hipEventCreateWithFlags(&b->ev_all_done, hipEventDisableTiming));
  if (b->nq == 0) { b->ran = true; return SANN_OK; }
  if (!b->use_fast) {  // the fast path's descriptor kernel clears these itself
    HIP_TRY(hipMemsetAsync(b->status.p, 0, ((size_t)b->nq + 2) * 4, st));
    HIP_TRY(hipMemsetAsync(b->unit_fb.p, 0xFF, (size_t)b->n_units * 4, st));
  }
  {
    if (b->profiling && !(b->prof_unit_only && b->use_fast)) HIP_TRY(hipEventRecord(b->ev[0], st));
    if (b->use_fast) {
      hipError_t e = launch_desc(b->ix->view(), b->view(), b->n_units, b->fast.max_n_scan, b->fast.k_local, st);
      if (e != hipSuccess) return fail(SANN_EDEVICE, std::string("launch_desc: ") + hipGetErrorString(e));
      // the descriptor kernel may run beside anything; the dominant kernel waits for its predecessor's
      if (after && after != b && after->unit_done_recorded)
        HIP_TRY(hipStreamWaitEvent(st, whole ? after->ev_all_done : after->last_unit_done, 0));
      if (b->profiling) HIP_TRY(hipEventRecord(b->ev[3], st));
      e = launch_unit_fast(b->ix->view(), b->view(), b->fast, b->n_units, st);
      if (e != hipSuccess) return fail(SANN_EDEVICE, std::string("launch_unit_fast: ") + hipGetErrorString(e));
    } else {
      std::vector<int32_t> none;
      int rc = run_general(b, none, st);
      if (rc != SANN_OK) return rc;
    }
  }
  if (b->profiling) HIP_TRY(hipEventRecord(b->ev[1], st));
  if (chained) {  // the profiling bracket's closing event doubles as the marker (an event costs the stream ~5 us)
    if (!b->profiling) HIP_TRY(hipEventRecord(b->ev_unit_done, st));
    b->last_unit_done = b->profiling ? b->ev[1] : b->ev_unit_done;
    b->unit_done_recorded = true;
  }
  HIP_TRY(launch_merge(b->ix->view(), b->view(), nullptr, b->nq, st));
  if (b->profiling) {
    if (!b->prof_unit_only) HIP_TRY(hipEventRecord(b->ev[2], st));
    b->ev_pending = true;
  }
  if (chained) HIP_TRY(hipEventRecord(b->ev_all_done, st));
  HIP_TRY(hipMemcpyAsync(b->h_status, b->status.p, 2 * 4, hipMemcpyDeviceToHost, st));
  if (b->device_prep && b->use_fast) {
    HIP_TRY(hipMemcpyAsync(b->h_qstat.p, b->q_stat.p, (size_t)b->nq * 16, hipMemcpyDeviceToHost, st));
    b->qstat_pending = true;
  }
  b->ran = true;
  return SANN_OK;
}

int sann_batch_run(sann_batch_t *b, void *hip_stream) try {
  return batch_run(b, hip_stream, false, nullptr, false);
} ABI_CATCH

int sann_batch_run_after(sann_batch_t *b, void *hip_stream, sann_batch_t *after, int32_t after_merge) try {
  return batch_run(b, hip_stream, true, after, after_merge != 0);
} ABI_CATCH

int sann_batch_finish(sann_batch_t *b, void *hip_stream) try {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  if (!b->ran) return fail(SANN_EINVAL, "sann_batch_run was not called");
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(hipSetDevice(b->ix->device));
  HIP_TRY(hipStreamSynchronize(st));
  if (b->nq == 0) return SANN_OK;
  if (b->ev_pending) {
    float a = 0.f, c = 0.f, d = 0.f;
    if (b->prof_unit_only && b->use_fast) {
      HIP_TRY(hipEventElapsedTime(&a, b->ev[3], b->ev[1]));  // the unit kernel alone
    } else {
      HIP_TRY(hipEventElapsedTime(&a, b->ev[0], b->ev[1]));
      if (!b->prof_unit_only) HIP_TRY(hipEventElapsedTime(&c, b->ev[1], b->ev[2]));
      if (b->use_fast) {
        HIP_TRY(hipEventElapsedTime(&d, b->ev[0], b->ev[3]));  // descriptor kernel
        a -= d;
      }
    }
    b->desc_ms_total += d;
    b->unit_ms_total += a;
    b->merge_ms_total += c;
    b->timed_runs++;
    b->ev_pending = false;
  }
  if (b->qstat_pending) {
    // the batch's shape as the device found it: postings scanned, scanned clusters, largest unit
    const uint32_t *qs = (const uint32_t *)b->h_qstat.p;
    int64_t postings = 0, clusters = 0;
    uint32_t t_max = 0;
    for (int q = 0; q < b->nq; q++) {
      t_max = std::max(t_max, qs[4 * q]);
      postings += qs[4 * q + 1];
      clusters += qs[4 * q + 2];
    }
    b->stats.postings_scanned = postings;
    b->stats.algorithmic_bytes = clusters * 12 + postings * 16;
    b->stats.max_unit_postings = (int32_t)t_max;
    // geometry hint for the next batches on this index: the largest unit seen, forgotten at 2 % per batch
    sann_index *ix = b->ix;
    int cur = ix->unit_size_hint.load(std::memory_order_relaxed);
    const int decayed = cur - cur / 50;
    const int next = std::max<int>((int)t_max, std::max(decayed, 1));
    ix->unit_size_hint.store(next, std::memory_order_relaxed);
    b->qstat_pending = false;
  }
  const int n_over = b->h_status[0], n_inexact = b->h_status[1];
  b->slow_tail_ran = false;
  if (n_over == 0 && n_inexact == 0) return SANN_OK;
  b->slow_tail_ran = true;
  if (!b->use_fast) return fail(SANN_EINTERNAL, "general path reported overflow/inexact units");

  // ---- slow tail: re-run on the general path whatever the fast path could not settle ---------
  // The two counts and the two lists were written by kernels: they are range-checked before they size or index
  // anything on the host (plan_slow_tail), so a corrupted status block is an error code, never a C++ exception or a
  // wild write inside the caller's process.
  const int P = b->ix->P;
  if (int rc0 = check_status_counts(n_over, n_inexact, b->n_units, b->nq)) return rc0;
  std::vector<int32_t> over((size_t)n_over), inexact((size_t)n_inexact);
  if (n_over) HIP_TRY(hipMemcpy(over.data(), b->overflow_units.p, (size_t)n_over * 4, hipMemcpyDeviceToHost));
  if (n_inexact)
    HIP_TRY(hipMemcpy(inexact.data(), b->status.as<int32_t>() + 2, (size_t)n_inexact * 4, hipMemcpyDeviceToHost));
  std::vector<int32_t> units, queries;
  int rc = plan_slow_tail(b->nq, P, over, inexact, units, queries);
  if (rc != SANN_OK) return rc;
  rc = ensure_unit_bounds(b, queries);
  if (rc != SANN_OK) return rc;
  rc = run_general(b, units, st);
  if (rc != SANN_OK) return rc;
  HIP_TRY(b->g_queries.reserve(queries.size() * 4));
  HIP_TRY(hipMemcpy(b->g_queries.p, queries.data(), queries.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemsetAsync(b->status.p, 0, 2 * 4, st));
  HIP_TRY(launch_merge(b->ix->view(), b->view(), b->g_queries.as<int32_t>(), (int)queries.size(), st));
  HIP_TRY(hipMemcpyAsync(b->h_status, b->status.p, 2 * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  b->stats.n_fallback_units += (int32_t)units.size();
  b->stats.n_requeried += n_inexact;
  if (b->h_status[0] != 0 || b->h_status[1] != 0)
    return fail(SANN_EINTERNAL, "general path could not settle the flagged units");
  return SANN_OK;
} ABI_CATCH

// enqueue the copies of a batch's results to the host on `st` (sync = false: the caller synchronises the stream)
static int results_impl(sann_batch_t *b, hipStream_t st, int64_t *out_ids, double *out_scores, int32_t out_stride,
                        int32_t *out_counts, int32_t *out_map_sizes, bool sync = true, bool by_kernel = true) {
  if (b->nq == 0) return SANN_OK;
  if (out_stride < b->stride) return fail(SANN_EINVAL, "out_stride smaller than the batch's max k");
  if (b->bound_chunk_q > 0) return fail(SANN_EINVAL, "outputs are bound to caller-owned chunked buffers");
  HIP_TRY(hipSetDevice(b->ix->device));
  BatchView bv = b->view();
  // All four arrays in pinned memory (sann_host_alloc: what a shim keeps across calls): a small kernel writes the answer
  // straight into them over PCIe -- one launch, and indifferent to what the runtime's copy path decides.  hipMemcpyAsync into
  // the same buffers is the faster way home while at least two other batches keep the GPU busy (0.235 against 0.287 ms per
  // 1024-query call with three callers and more), and much the slower one otherwise (0.93 / 0.77 ms against 0.46 / 0.41 with
  // one / two callers: tools/e2e_probe.py) -- the engine says which regime it is in (by_kernel).  Anything else (pageable
  // arrays, a missing array) takes the runtime's copies.
  if (by_kernel && out_ids && out_scores && out_counts && out_map_sizes) {
    void *d[4] = {nullptr, nullptr, nullptr, nullptr};
    const void *h[4] = {out_ids, out_scores, out_counts, out_map_sizes};
    bool pinned = true;
    for (int i = 0; i < 4 && pinned; i++) {
      hipPointerAttribute_t at;
      pinned = hipPointerGetAttributes(&at, h[i]) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer != nullptr;
      if (pinned) d[i] = at.devicePointer;
    }
    (void)hipGetLastError();  // (an unregistered pointer is reported as an error by some runtime versions: not ours)
    if (pinned) {
      HIP_TRY(launch_copy_out(b->nq, b->stride, out_stride, bv.out_ids, bv.out_scores, bv.out_counts, bv.out_map_sizes, (int64_t *)d[0],
                              (double *)d[1], (int32_t *)d[2], (int32_t *)d[3], st));
      if (sync) HIP_TRY(hipStreamSynchronize(st));
      return SANN_OK;
    }
  }
  // rows are contiguous when the caller's stride is the batch's: one copy per array (staged by the runtime into pageable memory)
  const size_t row = (size_t)b->stride * 8;
  if (out_ids) {
    if (out_stride == b->stride) HIP_TRY(hipMemcpyAsync(out_ids, bv.out_ids, row * (size_t)b->nq, hipMemcpyDeviceToHost, st));
    else HIP_TRY(hipMemcpy2DAsync(out_ids, (size_t)out_stride * 8, bv.out_ids, row, row, (size_t)b->nq, hipMemcpyDeviceToHost, st));
  }
  if (out_scores) {
    if (out_stride == b->stride) HIP_TRY(hipMemcpyAsync(out_scores, bv.out_scores, row * (size_t)b->nq, hipMemcpyDeviceToHost, st));
    else HIP_TRY(hipMemcpy2DAsync(out_scores, (size_t)out_stride * 8, bv.out_scores, row, row, (size_t)b->nq, hipMemcpyDeviceToHost, st));
  }
  if (out_counts) HIP_TRY(hipMemcpyAsync(out_counts, bv.out_counts, (size_t)b->nq * 4, hipMemcpyDeviceToHost, st));
  if (out_map_sizes) HIP_TRY(hipMemcpyAsync(out_map_sizes, bv.out_map_sizes, (size_t)b->nq * 4, hipMemcpyDeviceToHost, st));
  if (sync) HIP_TRY(hipStreamSynchronize(st));
  return SANN_OK;
}

int sann_batch_results(sann_batch_t *b, int64_t *out_ids, double *out_scores, int32_t out_stride, int32_t *out_counts,
                       int32_t *out_map_sizes) try {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  return results_impl(b, b->own_stream, out_ids, out_scores, out_stride, out_counts, out_map_sizes);
} ABI_CATCH

int sann_batch_device_results(sann_batch_t *b, void **d_ids, void **d_scores, void **d_counts, void **d_map_sizes,
                              int32_t *stride) try {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  if (b->bound_chunk_q > 0) return fail(SANN_EINVAL, "outputs are bound to caller-owned chunked buffers");
  BatchView bv = b->view();
  if (d_ids) *d_ids = bv.out_ids;
  if (d_scores) *d_scores = bv.out_scores;
  if (d_counts) *d_counts = bv.out_counts;
  if (d_map_sizes) *d_map_sizes = bv.out_map_sizes;
  if (stride) *stride = b->stride;
  return SANN_OK;
} ABI_CATCH

int sann_batch_bind_outputs(sann_batch_t *b, void *d_ids, void *d_scores, void *d_counts, void *d_map_sizes) try {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  bool all = d_ids && d_scores && d_counts && d_map_sizes, none = !d_ids && !d_scores && !d_counts && !d_map_sizes;
  if (!all && !none) return fail(SANN_EINVAL, "bind all four output buffers or none");
  b->bound_ids = d_ids;
  b->bound_scores = d_scores;
  b->bound_counts = d_counts;
  b->bound_map_sizes = d_map_sizes;
  b->bound_chunk_q = 0;
  b->bound_chunk_pitch = 0;
  b->bound_nq = b->nq;
  b->bound_stride = b->stride;
  return SANN_OK;
} ABI_CATCH

int sann_batch_bind_outputs_chunked(sann_batch_t *b, void *d_ids, void *d_scores, void *d_counts, void *d_map_sizes,
                                    int32_t queries_per_chunk, int64_t chunk_pitch_bytes) try {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  if (!d_ids || !d_scores || !d_counts || !d_map_sizes) return fail(SANN_EINVAL, "NULL output buffer");
  if (queries_per_chunk < 1 || chunk_pitch_bytes < 0 || (chunk_pitch_bytes & 7))
    return fail(SANN_EINVAL, "queries_per_chunk >= 1 and a chunk pitch that is a multiple of 8 bytes");
  b->bound_ids = d_ids;
  b->bound_scores = d_scores;
  b->bound_counts = d_counts;
  b->bound_map_sizes = d_map_sizes;
  b->bound_chunk_q = queries_per_chunk;
  b->bound_chunk_pitch = chunk_pitch_bytes;
  b->bound_nq = b->nq;
  b->bound_stride = b->stride;
  return SANN_OK;
} ABI_CATCH

int sann_batch_device_k(sann_batch_t *b, void **d_k) try {
  if (!b || !d_k) return fail(SANN_EINVAL, "NULL argument");
  *d_k = b->d_k.p;
  return SANN_OK;
} ABI_CATCH

int sann_batch_stats(sann_batch_t *b, sann_batch_stats_t *stats) try {
  if (!b || !stats) return fail(SANN_EINVAL, "NULL argument");
  *stats = b->stats;
  return SANN_OK;
} ABI_CATCH

int sann_batch_set_profiling(sann_batch_t *b, int32_t enable) try {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  HIP_TRY(hipSetDevice(b->ix->device));
  if (enable)
    for (auto &e : b->ev)
      if (!e) HIP_TRY(hipEventCreate(&e));
  b->profiling = enable != 0;
  b->prof_unit_only = enable == 1;
  b->desc_ms_total = b->unit_ms_total = b->merge_ms_total = 0.0;
  b->timed_runs = 0;
  b->ev_pending = false;
  return SANN_OK;
} ABI_CATCH

int sann_batch_kernel_times(sann_batch_t *b, double *unit_ms_total, double *merge_ms_total, int32_t *n_runs) try {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  // unit_ms_total covers the unit kernel alone; the descriptor kernel is reported by
  // sann_batch_desc_time
  if (unit_ms_total) *unit_ms_total = b->unit_ms_total;
  if (merge_ms_total) *merge_ms_total = b->merge_ms_total;
  if (n_runs) *n_runs = b->timed_runs;
  return SANN_OK;
} ABI_CATCH

int sann_debug_phase_cycles(sann_batch_t *b, int32_t enable, double *avg16) try {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  HIP_TRY(hipSetDevice(b->ix->device));
  if (enable) {
    if (!b->prof.p) {
      HIP_TRY(b->prof.alloc((size_t)(std::max(b->n_units, 1) + std::max(b->nq, 1)) * 16 * 8));
      HIP_TRY(hipMemset(b->prof.p, 0, b->prof.bytes));
    }
    return SANN_OK;
  }
  if (!b->prof.p || !avg16) return fail(SANN_EINVAL, "phase profiling was not enabled");
  std::vector<unsigned long long> h((size_t)(b->n_units + b->nq) * 16);
  HIP_TRY(hipMemcpy(h.data(), b->prof.p, h.size() * 8, hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; i++) avg16[i] = 0.0;
  int n = 0;
  for (int u = 0; u < b->n_units; u++) {
    const unsigned long long *s = &h[(size_t)u * 16];
    if (s[0] == 0 || s[8] == 0) continue;
    for (int i = 1; i <= 8; i++) avg16[i] += (double)(s[i] - s[i - 1]);
    avg16[0] += (double)(s[8] - s[0]);
    n++;
  }
  for (int i = 0; i < 16; i++) avg16[i] = n ? avg16[i] / n : 0.0;
  avg16[15] = n;
  if (getenv("SANN_PHASE_HIST")) {  // the duplicate phase is bimodal: units that resolve duplicates, and the others
    long long n_slow = 0;
    double slow = 0.0, fast = 0.0, slow_total = 0.0, fast_total = 0.0, ph_a = 0.0, ph_b = 0.0, ph_c = 0.0;
    for (int u = 0; u < b->n_units; u++) {
      const unsigned long long *s = &h[(size_t)u * 16];
      if (s[0] == 0 || s[8] == 0) continue;
      const double dphase = (double)(s[4] - s[3]);
      if (dphase > 1500.0 && s[9] && s[10]) {
        n_slow++; slow += dphase; slow_total += (double)(s[8] - s[0]);
        ph_a += (double)(s[9] - s[3]); ph_b += (double)(s[10] - s[9]); ph_c += (double)(s[4] - s[10]);
      }
      else { fast += dphase; fast_total += (double)(s[8] - s[0]); }
    }
    fprintf(stderr, "dup phase: %lld of %d units above 1500 clk (mean %.0f clk, lifetime %.0f); the others %.0f clk (lifetime %.0f)\n",
            n_slow, n, n_slow ? slow / n_slow : 0.0, n_slow ? slow_total / n_slow : 0.0, n - n_slow ? fast / (n - n_slow) : 0.0,
            n - n_slow ? fast_total / (n - n_slow) : 0.0);
    if (n_slow) fprintf(stderr, "  flagged units: fetch + match list %.0f clk, settle groups %.0f clk, fold %.0f clk\n", ph_a / n_slow, ph_b / n_slow, ph_c / n_slow);
  }
  // merge kernel stamps (per query) are reported in slots 9..14: phases 1..6
  {
    double m[7] = {0, 0, 0, 0, 0, 0, 0};
    int nm = 0;
    for (int q = 0; q < b->nq; q++) {
      const unsigned long long *s = &h[((size_t)b->n_units + q) * 16];
      if (s[0] == 0 || s[6] == 0) continue;
      for (int i = 1; i <= 6; i++) m[i] += (double)(s[i] - s[i - 1]);
      nm++;
    }
    for (int i = 1; i <= 6; i++) avg16[8 + i] = nm ? m[i] / nm : 0.0;
  }
  HIP_TRY(b->prof.alloc(0));
  return SANN_OK;
} ABI_CATCH

int sann_debug_gather_probe(sann_batch_t *b, int32_t mode, int32_t wgs_per_cu, int32_t reps, double *ms_avg,
                            uint64_t *checksum) try {
  if (!b || !ms_avg || !checksum) return fail(SANN_EINVAL, "NULL argument");
  if (!b->ran || !b->use_fast) return fail(SANN_EINVAL, "run the batch on the fast path first");
  if (reps < 1 || wgs_per_cu < 1 || wgs_per_cu > 16) return fail(SANN_EINVAL, "bad reps / wgs_per_cu");
  HIP_TRY(hipSetDevice(b->ix->device));
  HIP_TRY(hipDeviceSynchronize());
  DevBuf out;
  HIP_TRY(out.alloc((size_t)std::max(b->n_units, 1) * 8));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  float total = 0.f;
  for (int r = 0; r <= reps; r++) {  // rep 0 = warm-up
    HIP_TRY(hipMemsetAsync(out.p, 0, out.bytes, nullptr));
    HIP_TRY(hipEventRecord(e0, nullptr));
    if (mode >= 10) HIP_TRY(launch_unit_ablation(b->ix->view(), b->view(), b->fast, mode - 10, nullptr));
    else
      HIP_TRY(launch_gather_probe(b->ix->view(), b->view(), b->fast.unit_capacity, mode, wgs_per_cu,
                                  out.as<unsigned long long>(), nullptr));
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (r) total += ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  std::vector<uint64_t> h((size_t)b->n_units);
  HIP_TRY(hipMemcpy(h.data(), out.p, h.size() * 8, hipMemcpyDeviceToHost));
  uint64_t x = 0;
  for (size_t i = 0; i < h.size(); i++) x ^= h[i] * (2 * i + 1);
  *checksum = x;
  *ms_avg = total / reps;
  return SANN_OK;
} ABI_CATCH

int sann_debug_unit_arrays(sann_batch_t *b, int32_t *unit_unique, int32_t *cand_cnt, uint32_t *unit_flags, int32_t *unit_T) try {
  if (!b) return fail(SANN_EINVAL, "batch is NULL");
  HIP_TRY(hipSetDevice(b->ix->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t n = (size_t)b->n_units * 4;
  if (unit_unique) HIP_TRY(hipMemcpy(unit_unique, b->unit_unique.p, n, hipMemcpyDeviceToHost));
  if (cand_cnt) HIP_TRY(hipMemcpy(cand_cnt, b->cand_cnt.p, n, hipMemcpyDeviceToHost));
  if (unit_flags) HIP_TRY(hipMemcpy(unit_flags, b->unit_flags.p, n, hipMemcpyDeviceToHost));
  if (unit_T) HIP_TRY(hipMemcpy(unit_T, b->unit_T.p, n, hipMemcpyDeviceToHost));
  return SANN_OK;
} ABI_CATCH

int sann_debug_overflow_reasons(sann_batch_t *b, int32_t *counts8, int32_t *n_inexact) try {
  if (!b || !counts8) return fail(SANN_EINVAL, "NULL argument");
  HIP_TRY(hipSetDevice(b->ix->device));
  std::vector<uint32_t> fl((size_t)b->n_units);
  std::vector<uint64_t> thr((size_t)b->n_units * 2);
  HIP_TRY(hipMemcpy(fl.data(), b->unit_flags.p, fl.size() * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(thr.data(), b->unit_thr.p, thr.size() * 8, hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; i++) counts8[i] = 0;
  for (int u = 0; u < b->n_units; u++)
    if (fl[(size_t)u] & UNIT_OVERFLOW) counts8[std::min<uint64_t>(thr[(size_t)u * 2 + 1], 7)]++;
  if (n_inexact) *n_inexact = b->h_status ? b->h_status[1] : 0;
  return SANN_OK;
} ABI_CATCH

int sann_debug_plan_slow_tail(int32_t nq, int32_t n_partitions, int32_t n_over, const int32_t *over_units, int32_t n_inexact,
                              const int32_t *inexact_queries, int32_t *n_units_out, int32_t *n_queries_out) try {
  if (n_units_out) *n_units_out = 0;
  if (n_queries_out) *n_queries_out = 0;
  if (nq < 0 || n_partitions < 1) return fail(SANN_EINVAL, "bad batch shape");
  int rc = check_status_counts(n_over, n_inexact, (int64_t)nq * n_partitions, nq);
  if (rc != SANN_OK) return rc;
  if ((n_over > 0 && !over_units) || (n_inexact > 0 && !inexact_queries)) return fail(SANN_EINVAL, "NULL list");
  std::vector<int32_t> over(over_units, over_units + n_over), inexact(inexact_queries, inexact_queries + n_inexact), units, queries;
  rc = plan_slow_tail(nq, n_partitions, over, inexact, units, queries);
  if (rc != SANN_OK) return rc;
  if (n_units_out) *n_units_out = (int32_t)units.size();
  if (n_queries_out) *n_queries_out = (int32_t)queries.size();
  return SANN_OK;
} ABI_CATCH

int sann_batch_desc_time(sann_batch_t *b, double *desc_ms_total) try {
  if (!b || !desc_ms_total) return fail(SANN_EINVAL, "NULL argument");
  *desc_ms_total = b->desc_ms_total;
  return SANN_OK;
} ABI_CATCH

int32_t sann_tweet_shard(int64_t tweet_id, int32_t n_shards) {
  return n_shards <= 1 ? 0 : (int32_t)tweet_shard(mix64((uint64_t)tweet_id), (uint32_t)n_shards);
}
int32_t sann_tweet_partition(int64_t tweet_id, int32_t n_partitions) {
  return n_partitions <= 1 ? 0 : (int32_t)tweet_partition(mix64((uint64_t)tweet_id), (uint32_t)n_partitions);
}

int sann_device_alloc(int32_t device, int64_t bytes, void **out) try {
  if (!out || bytes < 0) return fail(SANN_EINVAL, "bad arguments");
  *out = nullptr;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMalloc(out, (size_t)std::max<int64_t>(bytes, 16)));
  return SANN_OK;
} ABI_CATCH
int sann_device_free(int32_t device, void *p) try {
  if (!p) return SANN_OK;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipFree(p));
  return SANN_OK;
} ABI_CATCH
int sann_device_copy(int32_t device, void *dst, const void *src, int64_t bytes) try {
  if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(SANN_EINVAL, "bad arguments");
  if (bytes == 0) return SANN_OK;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDefault));
  return SANN_OK;
} ABI_CATCH

int sann_device_synchronize(int32_t device) try {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipDeviceSynchronize());
  return SANN_OK;
} ABI_CATCH

}  // extern "C"
void sann_engine_stop(sann_index *ix);
sann_index::~sann_index() {
  sann_engine_stop(this);
  for (sann_batch *b : pool) delete b;
}
extern "C" {

int sann_batch_destroy(sann_batch_t *b) try {
  if (!b) return SANN_OK;
  (void)hipSetDevice(b->ix->device);
  delete b;
  return SANN_OK;
} ABI_CATCH

// One call = (pooled batch) reset + run + finish + results: the shape a JNI stub binds.  Every concurrent caller works on
// a batch object of its own, taken from the index's pool and given back afterwards, on that object's own non-blocking
// stream: after the first few calls nothing is allocated, and callers on different threads overlap on the GPU.
}  // extern "C"
static std::atomic<double> g_trace_us[4];
static std::atomic<int64_t> g_trace_calls{0};
static inline void operator+=(std::atomic<double> &a, double v) { double o = a.load(); while (!a.compare_exchange_weak(o, o + v)) {} }
extern "C" int sann_debug_call_trace(double *us4, int64_t *calls) {
  for (int i = 0; i < 4; i++) { if (us4) us4[i] = g_trace_us[i].load(); g_trace_us[i].store(0.0); }
  if (calls) *calls = g_trace_calls.exchange(0);
  return SANN_OK;
}
// ---------------------------------------------------------------------------------------------------------------------------
// The submission engine of the boundary call.
//
// sann_get_tweet_candidates is called from many threads at once (Finagle workers; the micro-batcher's dispatchers).  Until
// round 3 every caller drove its own pooled batch on its own stream -- reset, launches, a blocking wait, the copies home,
// another blocking wait -- and the HIP runtime serialised them: with four native callers the argument pass + H2D + one kernel
// launch of a call took 377 us of wall time instead of 18 (tools/e2e_probe.py), and a 1024-query call cost 1.6 x the GPU's
// step.  Now ONE thread per index makes every HIP call of this path, and it never waits while it could be submitting:
//
//   callers       hand over a job (their argument pointers stay valid: they block until the job is done) and sleep
//   the engine    for each job: batch_reset + kernels on one of its four streams, the unit kernel chained behind the
//                 previous job's (as bench.py's replay loop does); up to three jobs run ahead.  Then, in this order: the oldest
//                 job whose kernels are still out is finished (wait for ITS kernels, slow tail if flagged) and its answer
//                 sent home by the copy kernel; jobs whose copies have arrived are retired oldest first and their callers
//                 woken; the engine blocks on a copy only when there is nothing else to do
//
// so job i's copies home overlap job i+1's kernels, job i+2's argument pass overlaps both, and nobody contends for the
// runtime's locks.  SANN_ENGINE=0 restores the per-caller path (kept: it is the simplest statement of the call).
static bool trace_on() {
  static const bool t = [] { const char *e = getenv("SANN_TRACE_CALLS"); return e && e[0] == '1'; }();
  return t;
}
static double trace_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int zero_copy_mode() {  // SANN_ZERO_COPY: 0 never, 1 (default) for a call that finds the pipeline empty, 2 always
  static const int m = [] { const char *e = getenv("SANN_ZERO_COPY"); return e ? atoi(e) : 1; }();
  return m;
}
static bool memcpy_when_busy() {  // SANN_COPY_MEMCPY=1: the runtime's copies instead of the copy kernel while the pipeline is full
  static const bool t = [] { const char *e = getenv("SANN_COPY_MEMCPY"); return e && e[0] == '1'; }();
  return t;
}
struct EngineJob {
  // arguments of sann_candidates_pooled
  int32_t variant, nq, n_configs, out_stride;
  int64_t now_ms;
  const int64_t *now_ms_q, *emb_offsets, *source_tweet_ids, *scan_offsets;
  const int32_t *emb_cluster_ids, *scan_cluster_ids;
  const double *emb_scores;
  const uint8_t *has_source_tweet;
  const sann_config_t *configs;
  int64_t *out_ids;
  double *out_scores;
  int32_t *out_counts, *out_map_sizes;
  // state
  sann_batch *b = nullptr;
  int slot = 0;
  int stage = 0;  // 1 kernels enqueued, 2 copies enqueued
  bool zero_copy = false;  // the merge kernel writes the answer straight into the caller's pinned arrays
  int rc = SANN_OK;
  std::string msg;
  std::mutex m;
  std::condition_variable cv;
  bool done = false;
};

struct sann_engine {
  sann_index *ix;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<EngineJob *> queue;
  bool stop = false;
  static constexpr int kStreams = 4;
  hipStream_t streams[kStreams] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t copied[kStreams] = {nullptr, nullptr, nullptr, nullptr};

  void finish_job(EngineJob *j) {
    std::string keep = g_err;
    {  // the batch object goes back to the pool (a device error may have left it in an unknown state: dropped then)
      if (j->b && j->zero_copy) {  // (the caller's arrays are the caller's again)
        j->b->bound_ids = j->b->bound_scores = j->b->bound_counts = j->b->bound_map_sizes = nullptr;
        j->b->bound_nq = j->b->bound_stride = 0;
      }
      bool pooled = false;
      if (j->b && (j->rc == SANN_OK || j->rc == SANN_EINVAL || j->rc == SANN_ELIMIT)) {
        std::lock_guard<std::mutex> lk(ix->pool_mu);
        if (ix->pool.size() < 16) {
          ix->pool.push_back(j->b);
          pooled = true;
        }
      }
      if (!pooled && j->b) {
        (void)hipStreamSynchronize(streams[j->slot]);
        delete j->b;
      }
      j->b = nullptr;
    }
    if (j->rc != SANN_OK) j->msg = keep;
    {
      std::lock_guard<std::mutex> lk(j->m);
      j->done = true;
    }
    j->cv.notify_one();
  }

  // reset + kernels of a job on stream `slot`, its unit kernel behind `prev`'s
  void submit(EngineJob *j, int slot, sann_batch *prev, bool pipeline_empty) {
    j->slot = slot;
    sann_batch *b = nullptr;
    {
      std::lock_guard<std::mutex> lk(ix->pool_mu);
      if (!ix->pool.empty()) {
        b = ix->pool.back();
        ix->pool.pop_back();
      }
    }
    if (!b) {
      b = new (std::nothrow) sann_batch();
      if (!b) {
        j->rc = fail(SANN_ENOMEM, "out of host memory");
        return;
      }
      b->ix = ix;
    }
    j->b = b;
    b->variant = j->variant;
    hipStream_t st = streams[slot];
    j->rc = batch_reset(b, st, j->now_ms, j->nq, j->emb_offsets, j->emb_cluster_ids, j->emb_scores, j->source_tweet_ids,
                        j->has_source_tweet, j->configs, j->n_configs, j->scan_offsets, j->scan_cluster_ids, j->now_ms_q);
    // Zero copy (round 3), for a SMALL call (<= 256 queries) that finds the pipeline EMPTY -- the latency case, a single request
    // or a thin micro-batch: when the caller's four arrays are pinned (sann_host_alloc) and laid out at the batch's own stride,
    // the merge kernel is pointed at them and writes every query's rows home as it finishes them (one query: 0.107 against
    // 0.118 ms per call).  Not for the throughput case: a merge kernel writing over PCIe holds its workgroups' slots until the
    // stores have crossed, and with calls in flight the pipeline loses more than the copy kernel cost (1024 queries: 0.30
    // against 0.27-0.28 ms per call with four or eight callers; a lone 1024-query caller would gain 10 %, 0.43 against 0.47,
    // but a burst of callers starts as a lone caller).  SANN_ZERO_COPY=0 turns it off, =2 forces it for every call (measurements).
    j->zero_copy = false;
    if (j->rc == SANN_OK && (zero_copy_mode() == 2 || (zero_copy_mode() == 1 && pipeline_empty && b->nq <= 256)) && b->nq > 0 && j->out_stride == b->stride && j->out_ids && j->out_scores && j->out_counts &&
        j->out_map_sizes) {
      void *d[4] = {nullptr, nullptr, nullptr, nullptr};
      const void *h[4] = {j->out_ids, j->out_scores, j->out_counts, j->out_map_sizes};
      bool pinned = true;
      for (int i = 0; i < 4 && pinned; i++) {
        hipPointerAttribute_t at;
        pinned = hipPointerGetAttributes(&at, h[i]) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer != nullptr;
        if (pinned) d[i] = at.devicePointer;
      }
      (void)hipGetLastError();
      if (pinned) {
        b->bound_ids = d[0];
        b->bound_scores = d[1];
        b->bound_counts = d[2];
        b->bound_map_sizes = d[3];
        b->bound_chunk_q = 0;
        b->bound_chunk_pitch = 0;
        b->bound_nq = b->nq;
        b->bound_stride = b->stride;
        j->zero_copy = true;
      }
    }
    if (j->rc == SANN_OK) j->rc = sann_batch_run_after(b, st, prev, 0);
    if (j->rc == SANN_OK) j->stage = 1;
  }

  void run() {
    (void)hipSetDevice(ix->device);
    for (int i = 0; i < kStreams; i++) {
      (void)hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking);
      (void)hipEventCreateWithFlags(&copied[i], hipEventDisableTiming);
    }
    std::deque<EngineJob *> inflight;  // oldest first; at most kStreams
    int next_slot = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      // submit what is waiting, up to two jobs ahead of the one whose answer is being collected
      while (!queue.empty() && (int)inflight.size() < kStreams) {
        EngineJob *j = queue.front();
        queue.pop_front();
        lk.unlock();
        sann_batch *prev = nullptr;
        for (auto it = inflight.rbegin(); it != inflight.rend(); ++it)
          if ((*it)->stage == 1 && (*it)->b) { prev = (*it)->b; break; }
        const double t_s = trace_on() ? trace_now() : 0.0;
        submit(j, next_slot, prev, inflight.empty());
        if (trace_on()) { g_trace_us[0] += trace_now() - t_s; g_trace_calls++; }
        if (j->rc != SANN_OK) finish_job(j);
        else {
          inflight.push_back(j);
          next_slot = (next_slot + 1) % kStreams;
        }
        lk.lock();
      }
      if (inflight.empty()) {
        if (stop) break;
        cv.wait(lk);
        continue;
      }
      const bool more_queued = !queue.empty();
      lk.unlock();
      // 1. the oldest job whose kernels are still out: wait for them, settle, send its answer home.  This comes BEFORE the wait
      //    for an older job's copy: the copy of job A (6.6 MB over PCIe, ~150 us) then flies while the engine waits for job B's
      //    kernels, instead of the two waits adding up (round 3, first form: 150 + 90 + 40 us of engine time per job = 0.28 ms
      //    per call whatever the number of callers)
      EngineJob *k1 = nullptr;
      for (EngineJob *c : inflight)
        if (c->stage == 1) { k1 = c; break; }
      if (k1) {
        hipStream_t st = streams[k1->slot];
        const double t_f = trace_on() ? trace_now() : 0.0;
        k1->rc = sann_batch_finish(k1->b, st);
        const double t_r = trace_on() ? trace_now() : 0.0;
        if (k1->rc == SANN_OK && !k1->zero_copy)
          k1->rc = results_impl(k1->b, st, k1->out_ids, k1->out_scores, k1->out_stride, k1->out_counts, k1->out_map_sizes, false,
                                /* by_kernel = */ !memcpy_when_busy() || inflight.size() < (size_t)kStreams);
        if (trace_on()) { g_trace_us[2] += t_r - t_f; g_trace_us[1] += trace_now() - t_r; }
        k1->stage = 2;  // (a failed job has nothing in flight: it retires below without a wait to speak of)
      }
      // 2. retire, in submission order, the jobs whose copies have arrived; block on the front's copy only when there is nothing
      //    else to do (no kernels out, nothing queued)
      while (!inflight.empty() && inflight.front()->stage == 2) {
        EngineJob *f = inflight.front();
        bool other_work = more_queued && (int)inflight.size() < kStreams;  // (a queued job that has a slot to go to)
        for (EngineJob *c : inflight) other_work = other_work || c->stage == 1;
        if (f->rc == SANN_OK) {
          // (the stream holds nothing but this job: a slot is reused only after its job has left `inflight`)
          if (other_work && hipStreamQuery(streams[f->slot]) == hipErrorNotReady) break;
          const double t_w = trace_on() ? trace_now() : 0.0;
          if (hipStreamSynchronize(streams[f->slot]) != hipSuccess) f->rc = fail(SANN_EDEVICE, "hipStreamSynchronize");
          if (trace_on()) g_trace_us[3] += trace_now() - t_w;
        }
        inflight.pop_front();
        finish_job(f);
      }
      lk.lock();
    }
    lk.unlock();
    for (int i = 0; i < kStreams; i++) {
      if (streams[i]) (void)hipStreamDestroy(streams[i]);
      if (copied[i]) (void)hipEventDestroy(copied[i]);
    }
  }
};

static sann_engine *engine_of(sann_index *ix) {
  std::lock_guard<std::mutex> lk(ix->engine_mu);
  if (!ix->engine) {
    sann_engine *e = new sann_engine();
    e->ix = ix;
    e->th = std::thread([e] { e->run(); });
    ix->engine = e;
  }
  return ix->engine;
}
void sann_engine_stop(sann_index *ix) {
  sann_engine *e = nullptr;
  {
    std::lock_guard<std::mutex> lk(ix->engine_mu);
    e = ix->engine;
    ix->engine = nullptr;
  }
  if (!e) return;
  {
    std::lock_guard<std::mutex> lk(e->mu);
    e->stop = true;
  }
  e->cv.notify_all();
  e->th.join();
  delete e;
}

int sann_candidates_pooled(sann_index_t *index, int32_t variant, int64_t now_ms, const int64_t *now_ms_q, int32_t nq,
                           const int64_t *emb_offsets, const int32_t *emb_cluster_ids, const double *emb_scores,
                           const int64_t *source_tweet_ids, const uint8_t *has_source_tweet,
                           const sann_config_t *configs, int32_t n_configs, const int64_t *scan_offsets,
                           const int32_t *scan_cluster_ids, int64_t *out_ids, double *out_scores,
                           int32_t out_stride, int32_t *out_counts, int32_t *out_map_sizes) {
  if (!index) return fail(SANN_EINVAL, "index is NULL");
  if (variant < 0 || variant > 3) return fail(SANN_EINVAL, "unknown variant");
  static const bool use_engine = [] { const char *e = getenv("SANN_ENGINE"); return !(e && e[0] == '0'); }();
  if (use_engine) {
    EngineJob j;
    j.variant = variant; j.nq = nq; j.n_configs = n_configs; j.out_stride = out_stride; j.now_ms = now_ms; j.now_ms_q = now_ms_q;
    j.emb_offsets = emb_offsets; j.source_tweet_ids = source_tweet_ids; j.scan_offsets = scan_offsets;
    j.emb_cluster_ids = emb_cluster_ids; j.scan_cluster_ids = scan_cluster_ids; j.emb_scores = emb_scores;
    j.has_source_tweet = has_source_tweet; j.configs = configs; j.out_ids = out_ids; j.out_scores = out_scores;
    j.out_counts = out_counts; j.out_map_sizes = out_map_sizes;
    sann_engine *e = engine_of(index);
    {
      std::lock_guard<std::mutex> lk(e->mu);
      e->queue.push_back(&j);
    }
    e->cv.notify_one();
    {
      std::unique_lock<std::mutex> lk(j.m);
      j.cv.wait(lk, [&] { return j.done; });
    }
    if (j.rc != SANN_OK) return fail(j.rc, j.msg);
    return SANN_OK;
  }
  sann_batch *b = nullptr;
  {
    std::lock_guard<std::mutex> lk(index->pool_mu);
    if (!index->pool.empty()) {
      b = index->pool.back();
      index->pool.pop_back();
    }
  }
  if (!b) {
    HIP_TRY(hipSetDevice(index->device));
    b = new (std::nothrow) sann_batch();
    if (!b) return fail(SANN_ENOMEM, "out of host memory");
    b->ix = index;
  }
  if (!b->own_stream) {
    hipError_t e = hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete b;
      return fail(SANN_EDEVICE, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e));
    }
  }
  b->variant = variant;
  hipStream_t st = b->own_stream;
  // SANN_TRACE_CALLS=1: wall time per stage of this call, summed over the process (printed by sann_debug_call_trace)
  static const bool trace = [] { const char *e = getenv("SANN_TRACE_CALLS"); return e && e[0] == '1'; }();
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_a = trace ? now_us() : 0.0;
  int rc = batch_reset(b, st, now_ms, nq, emb_offsets, emb_cluster_ids, emb_scores, source_tweet_ids, has_source_tweet,
                       configs, n_configs, scan_offsets, scan_cluster_ids, now_ms_q);
  const double t_b = trace ? now_us() : 0.0;
  if (rc == SANN_OK) rc = sann_batch_run(b, st);
  const double t_c = trace ? now_us() : 0.0;
  // (the copies of the answer are enqueued when the kernels have finished: a copy that waits for its own stream's kernels sits
  // in the copy engine's queue in front of the other callers' copies -- measured 1.16 ms instead of 0.33 ms per call, four callers)
  if (rc == SANN_OK) rc = sann_batch_finish(b, st);
  const double t_d = trace ? now_us() : 0.0;
  if (rc == SANN_OK) rc = results_impl(b, st, out_ids, out_scores, out_stride, out_counts, out_map_sizes);
  if (trace) {
    const double t_e = now_us();
    g_trace_us[0] += t_b - t_a; g_trace_us[1] += t_c - t_b; g_trace_us[2] += t_d - t_c; g_trace_us[3] += t_e - t_d;
    g_trace_calls++;
  }
  std::string keep = g_err;
  bool pooled = false;
  if (rc == SANN_OK || rc == SANN_EINVAL || rc == SANN_ELIMIT) {  // a device error may have left the object in an unknown state
    std::lock_guard<std::mutex> lk(index->pool_mu);
    if (index->pool.size() < 16) {
      index->pool.push_back(b);
      pooled = true;
    }
  }
  if (!pooled) {
    (void)hipSetDevice(index->device);
    (void)hipStreamSynchronize(st);
    delete b;
  }
  if (rc != SANN_OK) g_err = keep;
  return rc;
}
extern "C" {

int sann_get_tweet_candidates(sann_index_t *index, int32_t variant, int64_t now_ms, int32_t nq,
                              const int64_t *emb_offsets, const int32_t *emb_cluster_ids, const double *emb_scores,
                              const int64_t *source_tweet_ids, const uint8_t *has_source_tweet,
                              const sann_config_t *configs, int32_t n_configs, const int64_t *scan_offsets,
                              const int32_t *scan_cluster_ids, int64_t *out_ids, double *out_scores,
                              int32_t out_stride, int32_t *out_counts, int32_t *out_map_sizes) try {
  return sann_candidates_pooled(index, variant, now_ms, nullptr, nq, emb_offsets, emb_cluster_ids, emb_scores, source_tweet_ids,
                                has_source_tweet, configs, n_configs, scan_offsets, scan_cluster_ids, out_ids, out_scores,
                                out_stride, out_counts, out_map_sizes);
} ABI_CATCH

int sann_get_tweet_candidates_at(sann_index_t *index, int32_t variant, const int64_t *now_ms, int32_t nq,
                                 const int64_t *emb_offsets, const int32_t *emb_cluster_ids, const double *emb_scores,
                                 const int64_t *source_tweet_ids, const uint8_t *has_source_tweet,
                                 const sann_config_t *configs, int32_t n_configs, const int64_t *scan_offsets,
                                 const int32_t *scan_cluster_ids, int64_t *out_ids, double *out_scores,
                                 int32_t out_stride, int32_t *out_counts, int32_t *out_map_sizes) try {
  if (nq > 0 && !now_ms) return fail(SANN_EINVAL, "now_ms is NULL");
  return sann_candidates_pooled(index, variant, nq > 0 ? now_ms[0] : 0, now_ms, nq, emb_offsets, emb_cluster_ids, emb_scores,
                                source_tweet_ids, has_source_tweet, configs, n_configs, scan_offsets, scan_cluster_ids, out_ids,
                                out_scores, out_stride, out_counts, out_map_sizes);
} ABI_CATCH

// The legacy candidate source for a batch: light rank + (optional) heavy rank, one call (see the header).  Runs on a pooled batch
// object and stream of its own, outside the submission engine (whose jobs are plain getTweetCandidates calls).
int sann_heavy_rank(sann_index_t *index, const struct rsx_store *source_store, const struct rsx_store *tweet_store, int64_t now_ms,
                    int32_t nq, const int64_t *emb_offsets, const int32_t *emb_cluster_ids, const double *emb_scores,
                    const int64_t *source_tweet_ids, const uint8_t *has_source_tweet, const int64_t *source_internal_ids,
                    const sann_legacy_config_t *config, int64_t *out_ids, double *out_scores, int32_t out_stride,
                    int32_t *out_counts) try {
  if (!index || !config) return fail(SANN_EINVAL, "NULL argument");
  if (nq < 0) return fail(SANN_EINVAL, "nq < 0");
  const sann_legacy_config_t &c = *config;
  const bool heavy = c.enable_heavy_ranking != 0;
  if (heavy) {
    if (!source_store || !tweet_store || (nq > 0 && !source_internal_ids)) return fail(SANN_EINVAL, "heavy ranking needs the source and tweet embedding stores and the source ids");
    if (c.ranking_algorithm < 1 || c.ranking_algorithm > 7) return fail(SANN_EINVAL, "unknown ranking algorithm");
    if (c.max_reranking_candidates < 0 || c.max_reranking_candidates > 1000) return fail(SANN_ELIMIT, "max_reranking_candidates above 1000");
    int32_t d1 = -1, d2 = -1;
    if (rsx_store_device(source_store, &d1) != 0 || rsx_store_device(tweet_store, &d2) != 0 || d1 != index->device || d2 != index->device)
      return fail(SANN_EINVAL, "the embedding stores must live on the index's device");
  }
  const int k_final = c.max_num_results < 0 ? 0 : c.max_num_results;
  if (k_final > 1000) return fail(SANN_ELIMIT, "legacy variant: max_num_results above 1000");
  if (nq > 0 && (out_stride < std::max(k_final, 1) || !out_ids || !out_scores || !out_counts)) return fail(SANN_EINVAL, "output arrays / out_stride");
  if (nq == 0) return SANN_OK;
  // SimClustersANNCandidateSource.scala:147-150,163-173: no normalisation, the "log" form (over l2norm), or the cosine form
  sann_config_t light{};
  light.max_num_results = heavy ? c.max_reranking_candidates : k_final;  // candidates.take(maxReRankingCandidates) / take(maxNumResults)
  light.candidate_embedding_type = c.candidate_embedding_type;
  light.min_score = 0.0;  // (ignored by the legacy variant)
  light.max_top_tweets_per_cluster = c.max_top_tweets_per_cluster;
  light.max_scan_clusters = c.max_scan_clusters;
  light.max_tweet_candidate_age_hours = c.max_tweet_candidate_age_hours;
  light.min_tweet_candidate_age_hours = c.min_tweet_candidate_age_hours;
  light.ann_algorithm = !c.enable_partial_normalization ? SANN_ALG_DOT_PRODUCT : (c.ranking_algorithm == 6 ? SANN_ALG_LOG_COSINE : SANN_ALG_COSINE);
  sann_batch *b = nullptr;
  {
    std::lock_guard<std::mutex> lk(index->pool_mu);
    if (!index->pool.empty()) {
      b = index->pool.back();
      index->pool.pop_back();
    }
  }
  HIP_TRY(hipSetDevice(index->device));
  if (!b) {
    b = new (std::nothrow) sann_batch();
    if (!b) return fail(SANN_ENOMEM, "out of host memory");
    b->ix = index;
  }
  int rc = SANN_OK;
  if (!b->own_stream && hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking) != hipSuccess) rc = fail(SANN_EDEVICE, "hipStreamCreateWithFlags");
  hipStream_t st = b->own_stream;
  b->variant = SANN_VARIANT_LEGACY;
  DevBuf d_src, d_ids, d_sc, d_cnt;
  if (rc == SANN_OK)
    rc = batch_reset(b, st, now_ms, nq, emb_offsets, emb_cluster_ids, emb_scores, source_tweet_ids, has_source_tweet, &light, 1, nullptr, nullptr);
  if (rc == SANN_OK) rc = sann_batch_run(b, st);
  if (rc == SANN_OK) rc = sann_batch_finish(b, st);  // (exact light lists first: the slow tail, if any, has run)
  if (rc == SANN_OK && !heavy) rc = results_impl(b, st, out_ids, out_scores, out_stride, out_counts, nullptr);
  if (rc == SANN_OK && heavy) {
    const int ks = std::max(k_final, 1);
    auto dev = [&](hipError_t e, const char *what) { if (e != hipSuccess && rc == SANN_OK) rc = fail(SANN_EDEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
    dev(d_src.reserve((size_t)nq * 8), "hipMalloc");
    dev(d_ids.reserve((size_t)nq * ks * 8), "hipMalloc");
    dev(d_sc.reserve((size_t)nq * ks * 8), "hipMalloc");
    dev(d_cnt.reserve((size_t)nq * 4), "hipMalloc");
    if (rc == SANN_OK) dev(hipMemcpyAsync(d_src.p, source_internal_ids, (size_t)nq * 8, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
    if (rc == SANN_OK) {
      BatchView bv = b->view();
      if (rsx_heavy_rank_device(source_store, tweet_store, st, c.ranking_algorithm, nq, d_src.p, bv.out_ids, bv.out_counts, b->stride,
                                c.min_score, k_final, ks, d_ids.p, d_sc.p, d_cnt.p) != 0)
        rc = fail(SANN_EDEVICE, std::string("heavy rank: ") + rsx_last_error());
    }
    if (rc == SANN_OK) {
      const size_t row = (size_t)ks * 8;
      dev(hipMemcpy2DAsync(out_ids, (size_t)out_stride * 8, d_ids.p, row, row, (size_t)nq, hipMemcpyDeviceToHost, st), "hipMemcpy2DAsync");
      dev(hipMemcpy2DAsync(out_scores, (size_t)out_stride * 8, d_sc.p, row, row, (size_t)nq, hipMemcpyDeviceToHost, st), "hipMemcpy2DAsync");
      dev(hipMemcpyAsync(out_counts, d_cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
      dev(hipStreamSynchronize(st), "hipStreamSynchronize");
    }
  }
  std::string keep = g_err;
  bool pooled = false;
  if (rc == SANN_OK || rc == SANN_EINVAL || rc == SANN_ELIMIT) {
    std::lock_guard<std::mutex> lk(index->pool_mu);
    if (index->pool.size() < 16) {
      index->pool.push_back(b);
      pooled = true;
    }
  }
  if (!pooled) {
    (void)hipStreamSynchronize(st);
    delete b;
  }
  if (rc != SANN_OK) g_err = keep;
  return rc;
} ABI_CATCH

int sann_host_alloc(int64_t bytes, void **out) try {
  if (!out || bytes < 0) return fail(SANN_EINVAL, "bad arguments");
  *out = nullptr;
  if (bytes == 0) return SANN_OK;
  HIP_TRY(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
  return SANN_OK;
} ABI_CATCH
int sann_host_free(void *p) try {
  if (p) HIP_TRY(hipHostFree(p));
  return SANN_OK;
} ABI_CATCH

int sann_merge_shards(int32_t device, void *hip_stream, int32_t n_shards, int32_t nq, int32_t stride,
                      int64_t shard_pitch_bytes, const void *d_ids, const void *d_scores, const void *d_counts,
                      const void *d_map_sizes, const void *d_k,
                      void *d_out_ids, void *d_out_scores, void *d_out_counts, void *d_out_map_sizes) try {
  if (n_shards < 1 || nq < 0 || stride < 1 || stride > 1024) return fail(SANN_EINVAL, "bad merge sizes");
  if (nq == 0) return SANN_OK;
  if (!d_ids || !d_scores || !d_counts || !d_map_sizes || !d_k || !d_out_ids || !d_out_scores || !d_out_counts ||
      !d_out_map_sizes)
    return fail(SANN_EINVAL, "NULL device pointer");
  HIP_TRY(hipSetDevice(device));
  if (shard_pitch_bytes < 0 || (shard_pitch_bytes & 7)) return fail(SANN_EINVAL, "shard_pitch_bytes must be a non-negative multiple of 8");
  HIP_TRY(launch_merge_shards(n_shards, nq, stride, shard_pitch_bytes, (const int64_t *)d_ids, (const double *)d_scores,
                              (const int32_t *)d_counts, (const int32_t *)d_map_sizes, (const int32_t *)d_k, 0, 0, stride,
                              (int64_t *)d_out_ids, (double *)d_out_scores, (int32_t *)d_out_counts,
                              (int32_t *)d_out_map_sizes, nullptr, (hipStream_t)hip_stream));
  return SANN_OK;
} ABI_CATCH

int sann_merge_shards_cut(int32_t device, void *hip_stream, int32_t n_shards, int32_t nq, int32_t shard_stride,
                          int64_t shard_pitch_bytes, int32_t shard_k, int32_t k, int32_t out_stride, const void *d_ids, const void *d_scores,
                          const void *d_counts, const void *d_map_sizes, void *d_out_ids, void *d_out_scores,
                          void *d_out_counts, void *d_out_map_sizes, void *d_inexact_count) try {
  if (n_shards < 1 || nq < 0 || shard_stride < 1 || shard_stride > 1024 || out_stride < 1 || out_stride > 1024)
    return fail(SANN_EINVAL, "bad merge sizes");
  if (shard_k < 1 || shard_k > shard_stride || k < 0) return fail(SANN_EINVAL, "bad shard_k / k");
  if (shard_pitch_bytes < 0 || (shard_pitch_bytes & 7)) return fail(SANN_EINVAL, "bad shard pitch");
  if (nq == 0) return SANN_OK;
  if (!d_ids || !d_scores || !d_counts || !d_map_sizes || !d_out_ids || !d_out_scores || !d_out_counts || !d_out_map_sizes ||
      !d_inexact_count)
    return fail(SANN_EINVAL, "NULL device pointer");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(launch_merge_shards(n_shards, nq, shard_stride, shard_pitch_bytes, (const int64_t *)d_ids, (const double *)d_scores,
                              (const int32_t *)d_counts, (const int32_t *)d_map_sizes, nullptr, k, shard_k, out_stride,
                              (int64_t *)d_out_ids, (double *)d_out_scores, (int32_t *)d_out_counts,
                              (int32_t *)d_out_map_sizes, (int32_t *)d_inexact_count, (hipStream_t)hip_stream));
  return SANN_OK;
} ABI_CATCH

int sann_debug_wave_sort(int32_t device, int32_t n_waves, uint32_t *values) try {
  if (n_waves < 0 || (n_waves > 0 && !values)) return fail(SANN_EINVAL, "bad arguments");
  if (n_waves == 0) return SANN_OK;
  HIP_TRY(hipSetDevice(device));
  DevBuf d;
  HIP_TRY(d.alloc((size_t)n_waves * 64 * 4));
  HIP_TRY(hipMemcpy(d.p, values, (size_t)n_waves * 64 * 4, hipMemcpyHostToDevice));
  HIP_TRY(launch_debug_wave_sort(n_waves, d.as<uint32_t>(), nullptr));
  HIP_TRY(hipMemcpy(values, d.p, (size_t)n_waves * 64 * 4, hipMemcpyDeviceToHost));
  return SANN_OK;
} ABI_CATCH

int sann_debug_approx(int32_t device, int32_t alg, int32_t n, const double *s, const double *w, double l2norm,
                      double lognorm, float *out, uint8_t *out_forced, double *eps) try {
  if (eps) *eps = kApproxEps;
  if (n < 0 || (n > 0 && (!s || !w || !out || !out_forced))) return fail(SANN_EINVAL, "bad arguments");
  if (n == 0) return SANN_OK;
  HIP_TRY(hipSetDevice(device));
  DevBuf a, b, c, d;
  HIP_TRY(a.alloc((size_t)n * 8));
  HIP_TRY(b.alloc((size_t)n * 8));
  HIP_TRY(c.alloc((size_t)n * 4));
  HIP_TRY(d.alloc((size_t)n));
  HIP_TRY(hipMemcpy(a.p, s, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.p, w, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(launch_debug_approx(alg, n, a.as<double>(), b.as<double>(), l2norm, lognorm, c.as<float>(), d.as<uint8_t>(), nullptr));
  HIP_TRY(hipMemcpy(out, c.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_forced, d.p, (size_t)n, hipMemcpyDeviceToHost));
  return SANN_OK;
} ABI_CATCH

int sann_debug_normalise(int32_t device, int32_t alg, int32_t n, const double *dot, const double *nsq, double l2norm,
                         double lognorm, double *out) try {
  if (n < 0 || (n > 0 && (!dot || !nsq || !out))) return fail(SANN_EINVAL, "bad arguments");
  if (n == 0) return SANN_OK;
  HIP_TRY(hipSetDevice(device));
  DevBuf a, b, c;
  HIP_TRY(a.alloc((size_t)n * 8));
  HIP_TRY(b.alloc((size_t)n * 8));
  HIP_TRY(c.alloc((size_t)n * 8));
  HIP_TRY(hipMemcpy(a.p, dot, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.p, nsq, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(launch_debug_normalise(alg, n, a.as<double>(), b.as<double>(), l2norm, lognorm, c.as<double>(), nullptr));
  HIP_TRY(hipMemcpy(out, c.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return SANN_OK;
} ABI_CATCH

}  // extern "C"
