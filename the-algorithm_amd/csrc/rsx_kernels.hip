// rsx_kernels.hip -- batched SimClusters pair scores on gfx950 (include/representation_scorer.h).
//
// Reference arithmetic (paths relative to /root/reference/src/scala/com/twitter/simclusters_v2/common/):
//   CosineSimilarityUtil.scala:224-250  sorted-merge sparse dot, `product += a*b` in ascending id
//   CosineSimilarityUtil.scala:15-59    sumOfSquares (left fold), norm, logNorm, expScaledNorm
//   CosineSimilarityUtil.scala:97-99    applyNormArray: x / norm, unchanged when norm == 0
//   SimClustersEmbedding.scala:194-224  dot / cosine / logNormCosine / expScaledCosine
//   SimClustersEmbedding.scala:235-243  jaccard ; :301-321 euclidean / manhattan
// One thread per pair: a pair is two <= ~50-entry lists, the merge is inherently sequential and the
// fp64 accumulation order is part of the semantics.  HBM-bound: (n_a + n_b) * 12 B per pair.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "../../include/representation_scorer.h"
#include "sann_math.h"
#include "abi_guard.h"
#define ABI_CATCH catch (...) { return abi_guard::caught(rsx_fail, RSX_ENOMEM, RSX_EINTERNAL); }

namespace {

thread_local std::string g_rsx_err;
int rsx_fail(int code, const std::string &m) { g_rsx_err = m; return code; }

__device__ inline double sum_sq(const double *v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; i++) s = s + v[i] * v[i];
  return s;
}

// one pair: the reference's arithmetic, operation for operation.  NaN for an unknown algorithm.
__device__ double pair_score(int alg, const int32_t *c1, const double *s1, int n1, const int32_t *c2, const double *s2,
                             int n2) {
  if (alg == 3) {  // jaccard
    if (n1 == 0 || n2 == 0) return 0.0;
    int i1 = 0, i2 = 0, inter = 0;
    while (i1 < n1 && i2 < n2) {
      const int x = c1[i1], y = c2[i2];
      inter += x == y;
      i1 += x <= y;
      i2 += y <= x;
    }
    return (double)inter / (double)(n1 + n2 - inter);
  }
  if (alg == 4 || alg == 5) {  // euclidean / manhattan over the union, ascending id
    int i1 = 0, i2 = 0;
    double sum = 0.0;
    while (i1 < n1 || i2 < n2) {
      double x = 0.0, y = 0.0;
      if (i2 >= n2 || (i1 < n1 && c1[i1] < c2[i2])) x = s1[i1++];
      else if (i1 >= n1 || c2[i2] < c1[i1]) y = s2[i2++];
      else { x = s1[i1++]; y = s2[i2++]; }
      const double d = fabs(x - y);
      sum = (alg == 4) ? sum + d * d : sum + d;
    }
    return (alg == 4) ? sqrt(sum) : sum;
  }
  double na = 1.0, nb = 1.0;
  bool norm = false;
  if (alg == 2) { na = sqrt(sum_sq(s1, n1)); nb = sqrt(sum_sq(s2, n2)); norm = true; }
  else if (alg == 6) { na = sann::strict_log(sum_sq(s1, n1) + 1); nb = sann::strict_log(sum_sq(s2, n2) + 1); norm = true; }
  else if (alg == 7) { na = pow(sum_sq(s1, n1), 0.3); nb = pow(sum_sq(s2, n2), 0.3); norm = true; }
  else if (alg != 1) return __builtin_nan("");
  const bool da = norm && na != 0, db = norm && nb != 0;  // applyNormArray leaves the array alone when norm == 0
  int i1 = 0, i2 = 0;
  double product = 0.0;
  while (i1 < n1 && i2 < n2) {
    const int x = c1[i1], y = c2[i2];
    if (x == y) {
      const double u = da ? s1[i1] / na : s1[i1];
      const double v = db ? s2[i2] / nb : s2[i2];
      product += u * v;
      i1++;
      i2++;
    } else if (x > y) {
      i2++;
    } else {
      i1++;
    }
  }
  return product;
}

__global__ __launch_bounds__(256) void rsx_pair_kernel(int alg, int n_pairs, const int64_t *ao, const int32_t *ac,
                                                       const double *as, const int64_t *bo, const int32_t *bc,
                                                       const double *bs, double *out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pairs) return;
  const int64_t a0 = ao[i], b0 = bo[i];
  out[i] = pair_score(alg, ac + a0, as + a0, (int)(ao[i + 1] - a0), bc + b0, bs + b0, (int)(bo[i + 1] - b0));
}

// ---- resident embedding stores (R1 hydration, R4 list scores, R5 engagement aggregates) ----
struct StoreView {
  const int64_t *off;
  const int32_t *cid;
  const double *sc;
};

// pair i = (row a_rows[i] of A, row b_rows[i] of B); a row < 0 is an embedding the store does not hold:
// the score is None (PairScoreStore.get, score/ScoreStore.scala:41-54)
__global__ __launch_bounds__(256) void rsx_rows_kernel(int alg, int n_pairs, StoreView A, const int32_t *a_rows, StoreView B,
                                                       const int32_t *b_rows, double *out, uint8_t *present) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pairs) return;
  const int ra = a_rows[i], rb = b_rows[i];
  if (ra < 0 || rb < 0) {
    out[i] = 0.0;
    present[i] = 0;
    return;
  }
  const int64_t a0 = A.off[ra], b0 = B.off[rb];
  out[i] = pair_score(alg, A.cid + a0, A.sc + a0, (int)(A.off[ra + 1] - a0), B.cid + b0, B.sc + b0, (int)(B.off[rb + 1] - b0));
  present[i] = 1;
}

// Scorer.computeSimilarityScoresPerTweet (representation-scorer/.../twistlyfeatures/Scorer.scala:157-369)
// for one (candidate, signal group): walk the group's signals in order, each contributing its score
// once per occurrence of its id in the hydrated id list (the groupBy(_.id) of :146-147 keeps every
// duplicate), skip the missing ones, then avg = sum / size with a left-fold sum and max folded from 0.0
// (Scorer.scala:426-429).
constexpr int RSX_MAX_MAPS = 4;
struct GroupArgs {
  StoreView cand;
  StoreView map[RSX_MAX_MAPS];
  const int32_t *cand_rows;     // [n_cand]
  const int32_t *group_map;     // [n_groups]
  const int64_t *group_off;     // [n_groups + 1]
  const int32_t *member_rows;   // row in the group's map store, < 0 = not hydrated
  const int32_t *member_mult;   // occurrences of the member's id in the map's id list
  double *out_avg, *out_max;
  int32_t *out_count;
  int n_cand, n_groups, alg;
};

__global__ __launch_bounds__(256) void rsx_group_kernel(GroupArgs g) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.n_cand * g.n_groups) return;
  const int c = i / g.n_groups, grp = i % g.n_groups;
  const int rc = g.cand_rows[c];
  double sum = 0.0, mx = 0.0;
  int cnt = 0;
  if (rc >= 0) {
    const StoreView S = g.map[g.group_map[grp]];
    const int64_t c0 = g.cand.off[rc];
    const int nc = (int)(g.cand.off[rc + 1] - c0);
    for (int64_t m = g.group_off[grp]; m < g.group_off[grp + 1]; ++m) {
      const int rs = g.member_rows[m];
      if (rs < 0) continue;
      const int64_t s0 = S.off[rs];
      const double v = pair_score(g.alg, S.cid + s0, S.sc + s0, (int)(S.off[rs + 1] - s0), g.cand.cid + c0, g.cand.sc + c0, nc);
      for (int r = 0; r < g.member_mult[m]; ++r) {
        sum = sum + v;
        mx = fmax(mx, v);
        cnt++;
      }
    }
  }
  g.out_count[i] = cnt;
  g.out_avg[i] = cnt ? sum / (double)cnt : 0.0;
  g.out_max[i] = mx;
}

// ---- the heavy-rank step of the legacy candidate source, one workgroup per query -------------------------------------------
// HeavyRanker.UniformScoreStoreRanker.rank (src/scala/com/twitter/simclusters_v2/candidate_source/HeavyRanker.scala:28-69) +
// reranking's sort and cut (SimClustersANNCandidateSource.scala:182-200): for the query's light candidates (at most 1024, already
// cut at maxReRankingCandidates) the pair score of (source embedding, candidate tweet embedding) through the resident stores --
// both hydrated by id HERE, a binary search in the store's device id column; a side the store does not hold is the
// reference's None and drops the candidate -- kept if score >= minScore (:63), sorted by score descending (ties, which the
// reference leaves to Map iteration order under a stable sort: tweet id ascending), cut at maxNumResults.
__device__ inline int store_row(const int64_t *ids, int n, int64_t id) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ids[mid] < id) lo = mid + 1;
    else hi = mid;
  }
  return (lo < n && ids[lo] == id) ? lo : -1;
}
constexpr int HR_MAX = 1024;  // light candidates per query (MaxNumResultsUpperBound is 1000)
__global__ __launch_bounds__(256) void rsx_heavy_rank_kernel(int alg, StoreView S, const int64_t *s_ids, int n_s, StoreView T,
                                                             const int64_t *t_ids, int n_t, const int64_t *source_ids,
                                                             const int64_t *light_ids, const int32_t *light_counts, int light_stride,
                                                             double min_score, int k, int out_stride, int64_t *out_ids,
                                                             double *out_scores, int32_t *out_counts) {
  __shared__ uint64_t s_hi[HR_MAX], s_lo[HR_MAX];
  __shared__ int s_n;
  const int q = blockIdx.x, tid = threadIdx.x;
  const int count = min(light_counts[q], min(light_stride, HR_MAX));
  const int rs = store_row(s_ids, n_s, source_ids[q]);  // (uniform)
  if (tid == 0) s_n = 0;
  int np = 64;
  while (np < count) np <<= 1;
  __syncthreads();
  int mine = 0;
  for (int j = tid; j < np; j += 256) {
    uint64_t hi = 0ull, lo = 0ull;
    if (j < count && rs >= 0) {
      const int64_t id = light_ids[(int64_t)q * light_stride + j];
      const int rt = store_row(t_ids, n_t, id);
      if (rt >= 0) {
        const int64_t a0 = S.off[rs], b0 = T.off[rt];
        const double v = pair_score(alg, S.cid + a0, S.sc + a0, (int)(S.off[rs + 1] - a0), T.cid + b0, T.sc + b0, (int)(T.off[rt + 1] - b0));
        if (v >= min_score) {  // HeavyRanker.scala:63 (false for NaN)
          hi = sann::score_key(v);
          lo = sann::id_key(id);
          mine++;
        }
      }
    }
    s_hi[j] = hi;
    s_lo[j] = lo;
  }
  if (mine) atomicAdd(&s_n, mine);
  __syncthreads();
  // bitonic sort of np 128-bit keys, descending (empty slots are (0, 0): below every real key)
  for (int kk = 2; kk <= np; kk <<= 1) {
    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
      for (int i = tid; i < np; i += 256) {
        const int l = i ^ jj;
        if (l > i) {
          const uint64_t ah = s_hi[i], al = s_lo[i], bh = s_hi[l], bl = s_lo[l];
          const bool a_lt_b = ah < bh || (ah == bh && al < bl);
          const bool desc = (i & kk) == 0;
          if (desc ? a_lt_b : !a_lt_b && !(ah == bh && al == bl)) {
            s_hi[i] = bh; s_lo[i] = bl; s_hi[l] = ah; s_lo[l] = al;
          }
        }
      }
      __syncthreads();
    }
  }
  const int n_out = min(s_n, max(k, 0));
  for (int i = tid; i < n_out; i += 256) {
    out_ids[(int64_t)q * out_stride + i] = sann::key_id(s_lo[i]);
    out_scores[(int64_t)q * out_stride + i] = sann::key_score(s_hi[i]);
  }
  if (tid == 0) out_counts[q] = n_out;
}

struct Buf {
  void *p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t put(const void *src, size_t bytes) {
    hipError_t r = hipMalloc(&p, bytes ? bytes : 8);
    if (r == hipSuccess && bytes && src) r = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
    return r;
  }
};

}  // namespace

struct rsx_store {
  int device = 0;
  std::vector<int64_t> ids;  // ascending; row = position
  Buf off, cid, sc;
  Buf d_ids;  // the ids on the device (hydration by id inside a kernel: rsx_heavy_rank_device)
  StoreView view() const { return StoreView{(const int64_t *)off.p, (const int32_t *)cid.p, (const double *)sc.p}; }
  int32_t row_of(int64_t id) const {
    auto it = std::lower_bound(ids.begin(), ids.end(), id);
    return (it == ids.end() || *it != id) ? -1 : (int32_t)(it - ids.begin());
  }
};

namespace {
#define RSX_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return rsx_fail(RSX_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

int rows_scores(const rsx_store *A, const rsx_store *B, int alg, int32_t n, const std::vector<int32_t> &ra,
                const std::vector<int32_t> &rb, double *out_scores, uint8_t *out_present) {
  RSX_TRY(hipSetDevice(A->device));
  Buf da, db, dout, dpres;
  RSX_TRY(da.put(ra.data(), (size_t)n * 4));
  RSX_TRY(db.put(rb.data(), (size_t)n * 4));
  RSX_TRY(dout.put(nullptr, (size_t)n * 8));
  RSX_TRY(dpres.put(nullptr, (size_t)n));
  hipLaunchKernelGGL(rsx_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, alg, n, A->view(), (const int32_t *)da.p,
                     B->view(), (const int32_t *)db.p, (double *)dout.p, (uint8_t *)dpres.p);
  RSX_TRY(hipGetLastError());
  RSX_TRY(hipMemcpy(out_scores, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  RSX_TRY(hipMemcpy(out_present, dpres.p, (size_t)n, hipMemcpyDeviceToHost));
  return RSX_OK;
}
}  // namespace

extern "C" {

const char *rsx_last_error(void) { return g_rsx_err.c_str(); }

int rsx_pair_scores_device(int32_t device, void *hip_stream, int32_t algorithm, int32_t n_pairs, const void *d_a_offsets,
                           const void *d_a_cluster_ids, const void *d_a_scores, const void *d_b_offsets,
                           const void *d_b_cluster_ids, const void *d_b_scores, void *d_out_scores) try {
  if (n_pairs < 0) return rsx_fail(RSX_EINVAL, "n_pairs < 0");
  if (algorithm < 1 || algorithm > 7) return rsx_fail(RSX_EINVAL, "unknown pair scoring algorithm");  // IllegalArgumentException in ScoreFacadeStore
  if (n_pairs == 0) return RSX_OK;
  if (!d_a_offsets || !d_b_offsets || !d_out_scores) return rsx_fail(RSX_EINVAL, "NULL device pointer");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return rsx_fail(RSX_EDEVICE, hipGetErrorString(e));
  hipLaunchKernelGGL(rsx_pair_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, algorithm, n_pairs,
                     (const int64_t *)d_a_offsets, (const int32_t *)d_a_cluster_ids, (const double *)d_a_scores,
                     (const int64_t *)d_b_offsets, (const int32_t *)d_b_cluster_ids, (const double *)d_b_scores,
                     (double *)d_out_scores);
  e = hipGetLastError();
  if (e != hipSuccess) return rsx_fail(RSX_EDEVICE, hipGetErrorString(e));
  return RSX_OK;
} ABI_CATCH

int rsx_pair_scores(int32_t device, int32_t algorithm, int32_t n_pairs, const int64_t *a_offsets,
                    const int32_t *a_cluster_ids, const double *a_scores, const int64_t *b_offsets,
                    const int32_t *b_cluster_ids, const double *b_scores, int32_t validate, double *out_scores) try {
  if (n_pairs < 0) return rsx_fail(RSX_EINVAL, "n_pairs < 0");
  if (algorithm < 1 || algorithm > 7) return rsx_fail(RSX_EINVAL, "unknown pair scoring algorithm");
  if (n_pairs == 0) return RSX_OK;
  if (!a_offsets || !b_offsets || !out_scores) return rsx_fail(RSX_EINVAL, "NULL argument");
  const int64_t na = a_offsets[n_pairs] - a_offsets[0], nb = b_offsets[n_pairs] - b_offsets[0];
  if ((na > 0 && (!a_cluster_ids || !a_scores)) || (nb > 0 && (!b_cluster_ids || !b_scores)))
    return rsx_fail(RSX_EINVAL, "NULL embedding arrays");
  if (a_offsets[0] != 0 || b_offsets[0] != 0) return rsx_fail(RSX_EINVAL, "offsets must start at 0");
  if (validate) {
    auto check = [&](const int64_t *o, const int32_t *c, const double *s) {
      for (int32_t i = 0; i < n_pairs; i++) {
        if (o[i + 1] < o[i]) return false;
        for (int64_t j = o[i]; j < o[i + 1]; j++) {
          if (!(s[j] > 0.0)) return false;
          if (j > o[i] && c[j] <= c[j - 1]) return false;
        }
      }
      return true;
    };
    if (!check(a_offsets, a_cluster_ids, a_scores) || !check(b_offsets, b_cluster_ids, b_scores))
      return rsx_fail(RSX_EINVAL, "embeddings must be sorted by cluster id, unique, with scores > 0");
  }
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return rsx_fail(RSX_EDEVICE, hipGetErrorString(e));
  Buf dao, dac, das, dbo, dbc, dbs, dout;
  auto up = [&](Buf &b, const void *src, size_t bytes) -> hipError_t {
    hipError_t r = hipMalloc(&b.p, bytes ? bytes : 8);
    if (r == hipSuccess && bytes) r = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
    return r;
  };
  if ((e = up(dao, a_offsets, ((size_t)n_pairs + 1) * 8)) != hipSuccess || (e = up(dac, a_cluster_ids, (size_t)na * 4)) != hipSuccess ||
      (e = up(das, a_scores, (size_t)na * 8)) != hipSuccess || (e = up(dbo, b_offsets, ((size_t)n_pairs + 1) * 8)) != hipSuccess ||
      (e = up(dbc, b_cluster_ids, (size_t)nb * 4)) != hipSuccess || (e = up(dbs, b_scores, (size_t)nb * 8)) != hipSuccess ||
      (e = hipMalloc(&dout.p, (size_t)n_pairs * 8)) != hipSuccess)
    return rsx_fail(RSX_EDEVICE, hipGetErrorString(e));
  int rc = rsx_pair_scores_device(device, nullptr, algorithm, n_pairs, dao.p, dac.p, das.p, dbo.p, dbc.p, dbs.p, dout.p);
  if (rc != RSX_OK) return rc;
  e = hipMemcpy(out_scores, dout.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return rsx_fail(RSX_EDEVICE, hipGetErrorString(e));
  return RSX_OK;
} ABI_CATCH

int rsx_store_build(int32_t device, int64_t n, const int64_t *ids, const int64_t *offsets, const int32_t *cluster_ids,
                    const double *scores, rsx_store_t **out) try {
  if (!out || n < 0 || (n > 0 && (!ids || !offsets))) return rsx_fail(RSX_EINVAL, "NULL argument");
  if (n >= 0x7fffffff) return rsx_fail(RSX_EINVAL, "too many embeddings");
  const int64_t total = n ? offsets[n] : 0;
  if (n && offsets[0] != 0) return rsx_fail(RSX_EINVAL, "offsets must start at 0");
  if (total > 0 && (!cluster_ids || !scores)) return rsx_fail(RSX_EINVAL, "NULL embedding arrays");
  for (int64_t i = 0; i < n; i++) {
    if (i && ids[i] <= ids[i - 1]) return rsx_fail(RSX_EINVAL, "store ids must be strictly ascending");
    if (offsets[i + 1] < offsets[i]) return rsx_fail(RSX_EINVAL, "offsets must not decrease");
    for (int64_t j = offsets[i]; j < offsets[i + 1]; j++) {
      if (!(scores[j] > 0.0) || (j > offsets[i] && cluster_ids[j] <= cluster_ids[j - 1]))
        return rsx_fail(RSX_EINVAL, "embeddings must be sorted by cluster id, unique, with scores > 0");
    }
  }
  std::unique_ptr<rsx_store> st(new rsx_store);
  st->device = device;
  st->ids.assign(ids, ids + n);
  RSX_TRY(hipSetDevice(device));
  const int64_t zero = 0;
  RSX_TRY(st->off.put(n ? offsets : &zero, ((size_t)n + 1) * 8));
  RSX_TRY(st->cid.put(cluster_ids, (size_t)total * 4));
  RSX_TRY(st->sc.put(scores, (size_t)total * 8));
  RSX_TRY(st->d_ids.put(ids, (size_t)n * 8));
  *out = st.release();
  return RSX_OK;
} ABI_CATCH

int rsx_store_destroy(rsx_store_t *store) try {
  delete store;
  return RSX_OK;
} ABI_CATCH

int rsx_store_pair_scores(const rsx_store_t *a, const rsx_store_t *b, int32_t algorithm, int32_t n_pairs,
                          const int64_t *a_ids, const int64_t *b_ids, double *out_scores, uint8_t *out_present) try {
  if (!a || !b || n_pairs < 0) return rsx_fail(RSX_EINVAL, "NULL store or n_pairs < 0");
  if (algorithm < 1 || algorithm > 7) return rsx_fail(RSX_EINVAL, "unknown pair scoring algorithm");
  if (a->device != b->device) return rsx_fail(RSX_EINVAL, "stores live on different devices");
  if (n_pairs == 0) return RSX_OK;
  if (!a_ids || !b_ids || !out_scores || !out_present) return rsx_fail(RSX_EINVAL, "NULL argument");
  std::vector<int32_t> ra((size_t)n_pairs), rb((size_t)n_pairs);
  for (int32_t i = 0; i < n_pairs; i++) {
    ra[(size_t)i] = a->row_of(a_ids[i]);
    rb[(size_t)i] = b->row_of(b_ids[i]);
  }
  return rows_scores(a, b, algorithm, n_pairs, ra, rb, out_scores, out_present);
} ABI_CATCH

int rsx_store_list_scores(const rsx_store_t *targets, const rsx_store_t *candidates, int32_t algorithm, int64_t target_id,
                          int32_t n_candidates, const int64_t *candidate_ids, double *out_scores, uint8_t *out_present) try {
  if (!targets || !candidates || n_candidates < 0) return rsx_fail(RSX_EINVAL, "NULL store or n_candidates < 0");
  if (algorithm < 1 || algorithm > 7) return rsx_fail(RSX_EINVAL, "unknown pair scoring algorithm");
  if (targets->device != candidates->device) return rsx_fail(RSX_EINVAL, "stores live on different devices");
  if (n_candidates == 0) return RSX_OK;
  if (!candidate_ids || !out_scores || !out_present) return rsx_fail(RSX_EINVAL, "NULL argument");
  std::vector<int32_t> ra((size_t)n_candidates, targets->row_of(target_id)), rb((size_t)n_candidates);
  for (int32_t i = 0; i < n_candidates; i++) rb[(size_t)i] = candidates->row_of(candidate_ids[i]);
  return rows_scores(targets, candidates, algorithm, n_candidates, ra, rb, out_scores, out_present);
} ABI_CATCH

int rsx_heavy_rank_device(const rsx_store_t *source_store, const rsx_store_t *tweet_store, void *hip_stream, int32_t algorithm,
                          int32_t nq, const void *d_source_ids, const void *d_light_ids, const void *d_light_counts,
                          int32_t light_stride, double min_score, int32_t max_num_results, int32_t out_stride, void *d_out_ids,
                          void *d_out_scores, void *d_out_counts) try {
  if (!source_store || !tweet_store || nq < 0) return rsx_fail(RSX_EINVAL, "NULL store or nq < 0");
  if (algorithm < 1 || algorithm > 7) return rsx_fail(RSX_EINVAL, "unknown pair scoring algorithm");
  if (source_store->device != tweet_store->device) return rsx_fail(RSX_EINVAL, "stores live on different devices");
  if (light_stride < 1 || out_stride < 1 || max_num_results > out_stride) return rsx_fail(RSX_EINVAL, "bad strides");
  if (nq == 0) return RSX_OK;
  if (!d_source_ids || !d_light_ids || !d_light_counts || !d_out_ids || !d_out_scores || !d_out_counts)
    return rsx_fail(RSX_EINVAL, "NULL device pointer");
  RSX_TRY(hipSetDevice(source_store->device));
  hipLaunchKernelGGL(rsx_heavy_rank_kernel, dim3(nq), dim3(256), 0, (hipStream_t)hip_stream, algorithm, source_store->view(),
                     (const int64_t *)source_store->d_ids.p, (int)source_store->ids.size(), tweet_store->view(),
                     (const int64_t *)tweet_store->d_ids.p, (int)tweet_store->ids.size(), (const int64_t *)d_source_ids,
                     (const int64_t *)d_light_ids, (const int32_t *)d_light_counts, light_stride, min_score, max_num_results, out_stride,
                     (int64_t *)d_out_ids, (double *)d_out_scores, (int32_t *)d_out_counts);
  RSX_TRY(hipGetLastError());
  return RSX_OK;
} ABI_CATCH

int rsx_store_device(const rsx_store_t *store, int32_t *device) try {
  if (!store || !device) return rsx_fail(RSX_EINVAL, "NULL argument");
  *device = store->device;
  return RSX_OK;
} ABI_CATCH

int rsx_store_group_features(const rsx_store_t *candidates, int32_t algorithm, int32_t n_candidates,
                             const int64_t *candidate_ids, int32_t n_maps, const rsx_store_t *const *map_stores,
                             const int64_t *map_id_offsets, const int64_t *map_ids, int32_t n_groups,
                             const int32_t *group_map, const int64_t *group_offsets, const int64_t *group_member_ids,
                             double *out_avg, double *out_max, int32_t *out_count) try {
  if (!candidates || n_candidates < 0 || n_groups < 0) return rsx_fail(RSX_EINVAL, "NULL store or negative size");
  if (algorithm < 1 || algorithm > 7) return rsx_fail(RSX_EINVAL, "unknown pair scoring algorithm");
  if (n_maps < 1 || n_maps > RSX_MAX_MAPS) return rsx_fail(RSX_EINVAL, "n_maps must be in 1..4");
  if (n_candidates == 0 || n_groups == 0) return RSX_OK;
  if (!candidate_ids || !map_stores || !map_id_offsets || !group_map || !group_offsets || !out_avg || !out_max || !out_count)
    return rsx_fail(RSX_EINVAL, "NULL argument");
  if ((int64_t)n_candidates * n_groups >= 0x7fffffff) return rsx_fail(RSX_EINVAL, "n_candidates * n_groups too large");
  for (int m = 0; m < n_maps; m++)
    if (!map_stores[m] || map_stores[m]->device != candidates->device) return rsx_fail(RSX_EINVAL, "bad map store");
  const int64_t n_members = group_offsets[n_groups];
  if (n_members > 0 && (!group_member_ids || !map_ids)) return rsx_fail(RSX_EINVAL, "NULL member / map ids");
  // multiplicity of every id in its map's hydrated id list (Scorer.scala:139-147: one ScoreResult per
  // occurrence, grouped by id)
  std::vector<std::vector<int64_t>> sorted((size_t)n_maps);
  for (int m = 0; m < n_maps; m++) {
    sorted[(size_t)m].assign(map_ids + map_id_offsets[m], map_ids + map_id_offsets[m + 1]);
    std::sort(sorted[(size_t)m].begin(), sorted[(size_t)m].end());
  }
  std::vector<int32_t> rows((size_t)n_members), mult((size_t)n_members), crow((size_t)n_candidates);
  for (int g = 0; g < n_groups; g++) {
    const int m = group_map[g];
    if (m < 0 || m >= n_maps) return rsx_fail(RSX_EINVAL, "group_map out of range");
    if (group_offsets[g + 1] < group_offsets[g]) return rsx_fail(RSX_EINVAL, "group offsets must not decrease");
    for (int64_t j = group_offsets[g]; j < group_offsets[g + 1]; j++) {
      const int64_t id = group_member_ids[j];
      auto r = std::equal_range(sorted[(size_t)m].begin(), sorted[(size_t)m].end(), id);
      mult[(size_t)j] = (int32_t)(r.second - r.first);  // 0: the signal's id was never scored -> contributes nothing
      rows[(size_t)j] = map_stores[m]->row_of(id);
    }
  }
  for (int32_t i = 0; i < n_candidates; i++) crow[(size_t)i] = candidates->row_of(candidate_ids[i]);
  RSX_TRY(hipSetDevice(candidates->device));
  const size_t nout = (size_t)n_candidates * n_groups;
  Buf dcrow, dgmap, dgoff, drows, dmult, davg, dmax, dcnt;
  RSX_TRY(dcrow.put(crow.data(), crow.size() * 4));
  RSX_TRY(dgmap.put(group_map, (size_t)n_groups * 4));
  RSX_TRY(dgoff.put(group_offsets, ((size_t)n_groups + 1) * 8));
  RSX_TRY(drows.put(rows.data(), rows.size() * 4));
  RSX_TRY(dmult.put(mult.data(), mult.size() * 4));
  RSX_TRY(davg.put(nullptr, nout * 8));
  RSX_TRY(dmax.put(nullptr, nout * 8));
  RSX_TRY(dcnt.put(nullptr, nout * 4));
  GroupArgs g;
  g.cand = candidates->view();
  for (int m = 0; m < RSX_MAX_MAPS; m++) g.map[m] = map_stores[m < n_maps ? m : 0]->view();
  g.cand_rows = (const int32_t *)dcrow.p;
  g.group_map = (const int32_t *)dgmap.p;
  g.group_off = (const int64_t *)dgoff.p;
  g.member_rows = (const int32_t *)drows.p;
  g.member_mult = (const int32_t *)dmult.p;
  g.out_avg = (double *)davg.p;
  g.out_max = (double *)dmax.p;
  g.out_count = (int32_t *)dcnt.p;
  g.n_cand = n_candidates;
  g.n_groups = n_groups;
  g.alg = algorithm;
  hipLaunchKernelGGL(rsx_group_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, 0, g);
  RSX_TRY(hipGetLastError());
  RSX_TRY(hipMemcpy(out_avg, davg.p, nout * 8, hipMemcpyDeviceToHost));
  RSX_TRY(hipMemcpy(out_max, dmax.p, nout * 8, hipMemcpyDeviceToHost));
  RSX_TRY(hipMemcpy(out_count, dcnt.p, nout * 4, hipMemcpyDeviceToHost));
  return RSX_OK;
} ABI_CATCH

}  // extern "C"
