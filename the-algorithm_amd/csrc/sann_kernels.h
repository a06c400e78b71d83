// sann_kernels.h -- host-callable launchers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "sann_device.h"

struct sann_config;

namespace sann {

hipError_t launch_unit_general(const IndexView &ix, const BatchView &b, const GeneralWs &ws, int n_units,
                               hipStream_t stream);
hipError_t launch_merge(const IndexView &ix, const BatchView &b, const int32_t *query_list, int n_queries,
                        hipStream_t stream);
hipError_t launch_merge_shards(int n_shards, int nq, int stride, int64_t pitch, const int64_t *ids, const double *scores,
                               const int32_t *counts, const int32_t *map_sizes, const int32_t *k, int k_all, int shard_k,
                               int out_stride, int64_t *out_ids, double *out_scores, int32_t *out_counts,
                               int32_t *out_map_sizes, int32_t *inexact, hipStream_t stream);

// The answer of a batch written straight into the caller's PINNED host arrays by a (small) kernel: rows of `stride` entries ->
// rows of `out_stride`, plus the two per-query int arrays.  A dozen workgroups keep the PCIe link busy and leave the CUs to the
// next batch's kernels; unlike hipMemcpyAsync its cost does not depend on what else the runtime finds the stream doing.
hipError_t launch_copy_out(int nq, int stride, int out_stride, const int64_t *ids, const double *scores, const int32_t *counts,
                           const int32_t *map_sizes, int64_t *h_ids, double *h_scores, int32_t *h_counts, int32_t *h_map_sizes,
                           hipStream_t stream);
hipError_t launch_debug_normalise(int alg, int n, const double *dot, const double *nsq, double l2norm, double lognorm,
                                  double *out, hipStream_t stream);

// Query preparation on the device (sann_prep.hip): raw embeddings + configs in, QueryHdr + scan rows / weights out.
constexpr int PREP_MAX = 1024;  // embedding entries a query may have on the device path (longer ones: host path)
struct PrepView {
  const int64_t *emb_offsets;       // [nq+1]
  const int32_t *emb_cluster_ids;
  const double *emb_scores;
  const int64_t *source_tweet_ids;  // [nq] or NULL
  const uint8_t *has_source_tweet;  // [nq] or NULL
  const struct ::sann_config *configs;
  const int64_t *scan_offsets;      // [nq+1] or NULL
  const int32_t *scan_cluster_ids;
  const int32_t *scan_begin;        // [nq] start of the query's region in scan_row / scan_w
  const int32_t *cluster_ids;       // [n_rows] the index's cluster ids, ascending
  QueryHdr *hdr;
  int32_t *scan_row;
  double *scan_w;
  int32_t *d_k;
  int64_t now_ms;
  const int64_t *now_ms_q;          // [nq] per-query Time.now, or NULL (every query at now_ms)
  int32_t n_rows, n_configs, variant, nq;
};
hipError_t launch_prep(const PrepView &in, hipStream_t stream);

// measurement only (sann_probe.hip): the unit kernel's gather with the arithmetic replaced by a checksum
hipError_t launch_gather_probe(const IndexView &ix, const BatchView &b, int unit_capacity, int mode, int wgs_per_cu,
                               unsigned long long *out, hipStream_t stream);
hipError_t launch_debug_wave_sort(int n_waves, uint32_t *v, hipStream_t stream);
hipError_t launch_debug_approx(int alg, int n, const double *s, const double *w, double l2norm, double lognorm, float *out,
                               uint8_t *out_forced, hipStream_t stream);
constexpr double kApproxEps = 4e-6;  // = APPROX_EPS of sann_fast.hip, reported by sann_debug_approx

// LDS fast path (sann_fast.hip).  Returns hipErrorInvalidValue when the configuration cannot
// run on the fast path at all (the caller then uses the general path for every unit).
constexpr int FAST_SCAP = 160;  // candidates a fast unit examines exactly and may emit (BatchView.cap)
struct FastParams {
  int unit_capacity;  // postings one unit holds in registers (workgroup size x postings per thread)
  int k_local;        // floor on the entries a unit must offer before it may withhold the rest
  int max_n_scan;     // largest number of scanned clusters of any query of the batch
  int use_norms;      // some query of the batch scores with the index's norms column (offline forms)
};
hipError_t launch_cut(const IndexView &ix, int M, uint32_t *out, hipStream_t stream);
hipError_t launch_desc(const IndexView &ix, const BatchView &b, int n_units, int max_n_scan, int k_local_floor, hipStream_t stream);
hipError_t launch_unit_ablation(const IndexView &ix, const BatchView &b, const FastParams &fp, int abl, hipStream_t stream);
hipError_t launch_unit_fast(const IndexView &ix, const BatchView &b, const FastParams &fp, int n_units,
                            hipStream_t stream);

}  // namespace sann
