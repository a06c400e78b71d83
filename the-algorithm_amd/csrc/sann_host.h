// sann_host.h -- private host-side definitions shared by sann_api.hip and sann_corpus.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
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
  hipError_t alloc(size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    bytes = n;
    if (n == 0) return hipSuccess;
    return hipMalloc(&p, n);
  }
  template <class T> T *as() const { return (T *)p; }
};

}  // namespace sann_host

struct sann_index {
  int device = 0;
  int P = 1, log2P = 0, shard_id = 0, n_shards = 1;
  std::vector<int32_t> cluster_ids;     // ascending; row = position
  std::vector<uint32_t> h_sub_offsets;  // host copy of the device CSR
  int64_t n_postings = 0, n_postings_total = 0;
  int32_t max_list_len = 0;
  sann_host::DevBuf postings, ranks, sub_offsets;
  // cut cache: for a given maxTopTweetsPerCluster M, the number of postings with rank < M in every
  // (row, partition) sub-list.  M is a service-level constant in practice, so this is built once
  // (one binary search per sub-list, on the device) and reused by every batch.
  std::mutex cut_mu;
  std::vector<std::pair<int32_t, std::unique_ptr<sann_host::DevBuf>>> cut_cache;

  sann::IndexView view() const {
    sann::IndexView v;
    v.postings = postings.as<sann::Posting>();
    v.ranks = ranks.as<uint32_t>();
    v.sub_offsets = sub_offsets.as<uint32_t>();
    v.n_rows = (int32_t)cluster_ids.size();
    v.P = P;
    v.log2P = log2P;
    return v;
  }
  int row_of(int32_t cluster) const {
    auto it = std::lower_bound(cluster_ids.begin(), cluster_ids.end(), cluster);
    if (it == cluster_ids.end() || *it != cluster) return -1;
    return (int)(it - cluster_ids.begin());
  }
};
