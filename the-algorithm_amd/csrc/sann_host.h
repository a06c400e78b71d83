// sann_host.h -- private host-side definitions shared by sann_api.hip and sann_corpus.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/simclusters_ann.h"
#include "sann_device.h"
#include "sann_math.h"

namespace sann_host {

int fail(int code, const std::string &msg);  // sets the thread-local error string, returns code

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return sann_host::fail(SANN_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  size_t cap = 0;
  hipError_t alloc(size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    bytes = n;
    cap = 0;
    if (n == 0) return hipSuccess;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) cap = n;
    return e;
  }
  // Make room for n bytes, keeping the allocation when it is large enough (contents are NOT kept when it grows):
  // a batch object that is reset for every request allocates only while its requests are still growing.
  hipError_t reserve(size_t n) {
    if (n <= cap) { bytes = n; return hipSuccess; }
    const size_t want = n + n / 4;
    hipError_t e = alloc(want);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <class T> T *as() const { return (T *)p; }
};

}  // namespace sann_host

// sann_get_tweet_candidates[_at] behind the ABI guard (sann_api.hip): one pooled batch object per concurrent caller; now_ms_q =
// per-query Time.now or NULL.  Also what the micro-batcher's dispatchers call (sann_batcher.hip).
int sann_candidates_pooled(sann_index_t *index, int32_t variant, int64_t now_ms, const int64_t *now_ms_q, int32_t nq,
                           const int64_t *emb_offsets, const int32_t *emb_cluster_ids, const double *emb_scores,
                           const int64_t *source_tweet_ids, const uint8_t *has_source_tweet, const sann_config_t *configs,
                           int32_t n_configs, const int64_t *scan_offsets, const int32_t *scan_cluster_ids, int64_t *out_ids,
                           double *out_scores, int32_t out_stride, int32_t *out_counts, int32_t *out_map_sizes);

struct sann_index {
  int device = 0;
  int P = 1, log2P = 0, shard_id = 0, n_shards = 1;
  std::vector<int32_t> cluster_ids;     // ascending; row = position
  std::vector<uint32_t> h_sub_offsets;  // host copy of the device CSR
  int64_t n_postings = 0, n_postings_total = 0;
  int32_t max_list_len = 0;
  sann_host::DevBuf postings, ranks, sub_offsets, norms;
  // cut cache: for a given maxTopTweetsPerCluster M, the number of postings with rank < M in every
  // (row, partition) sub-list.  M is a service-level constant in practice, so this is built once
  // (one binary search per sub-list, on the device) and reused by every batch.
  std::mutex cut_mu;
  std::vector<std::pair<int32_t, std::unique_ptr<sann_host::DevBuf>>> cut_cache;

  // device copy of cluster_ids for the query-preparation kernel (made on first use)
  sann_host::DevBuf d_cluster_ids;
  std::once_flag d_cluster_ids_once;
  hipError_t d_cluster_ids_err = hipSuccess;
  hipError_t ensure_device_cluster_ids() {
    std::call_once(d_cluster_ids_once, [this] {
      hipError_t e = d_cluster_ids.alloc(std::max<size_t>(cluster_ids.size(), 1) * 4);
      if (e == hipSuccess && !cluster_ids.empty())
        e = hipMemcpy(d_cluster_ids.p, cluster_ids.data(), cluster_ids.size() * 4, hipMemcpyHostToDevice);
      d_cluster_ids_err = e;
    });
    return d_cluster_ids_err;
  }
  // largest number of postings a (query, partition) unit of recent batches scanned (decays slowly; 0 = unknown):
  // the device-prepared batches' geometry hint (sann_api.hip, batch_reset)
  std::atomic<int> unit_size_hint{0};
  // batch objects kept for sann_get_tweet_candidates (one per concurrent caller), reset per call
  std::mutex pool_mu;
  std::vector<struct sann_batch *> pool;
  // the submission engine of sann_get_tweet_candidates (one thread that owns every HIP call of the pooled path; made on first use)
  std::mutex engine_mu;
  struct sann_engine *engine = nullptr;
  ~sann_index();

  sann::IndexView view() const {
    sann::IndexView v;
    v.postings = postings.as<sann::Posting>();
    v.ranks = ranks.as<uint32_t>();
    v.sub_offsets = sub_offsets.as<uint32_t>();
    v.norms = norms.as<double>();
    v.n_rows = (int32_t)cluster_ids.size();
    v.P = P;
    v.log2P = log2P;
    v.n_postings = (uint32_t)n_postings;
    return v;
  }
  int row_of(int32_t cluster) const {
    auto it = std::lower_bound(cluster_ids.begin(), cluster_ids.end(), cluster);
    if (it == cluster_ids.end() || *it != cluster) return -1;
    return (int)(it - cluster_ids.begin());
  }
};
