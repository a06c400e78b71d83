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

#include <string>
#include <vector>

#include "../../include/representation_scorer.h"
#include "sann_math.h"

namespace {

thread_local std::string g_rsx_err;
int rsx_fail(int code, const std::string &m) { g_rsx_err = m; return code; }

__device__ inline double sum_sq(const double *v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; i++) s = s + v[i] * v[i];
  return s;
}

__global__ __launch_bounds__(256) void rsx_pair_kernel(int alg, int n_pairs, const int64_t *ao, const int32_t *ac,
                                                       const double *as, const int64_t *bo, const int32_t *bc,
                                                       const double *bs, double *out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pairs) return;
  const int64_t a0 = ao[i], b0 = bo[i];
  const int n1 = (int)(ao[i + 1] - a0), n2 = (int)(bo[i + 1] - b0);
  const int32_t *c1 = ac + a0, *c2 = bc + b0;
  const double *s1 = as + a0, *s2 = bs + b0;
  double r;
  if (alg == 3) {  // jaccard
    if (n1 == 0 || n2 == 0) {
      r = 0.0;
    } else {
      int i1 = 0, i2 = 0, inter = 0;
      while (i1 < n1 && i2 < n2) {
        const int x = c1[i1], y = c2[i2];
        inter += x == y;
        i1 += x <= y;
        i2 += y <= x;
      }
      r = (double)inter / (double)(n1 + n2 - inter);
    }
  } else if (alg == 4 || alg == 5) {  // euclidean / manhattan over the union, ascending id
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
    r = (alg == 4) ? sqrt(sum) : sum;
  } else {
    double na = 1.0, nb = 1.0;
    bool norm = false;
    if (alg == 2) { na = sqrt(sum_sq(s1, n1)); nb = sqrt(sum_sq(s2, n2)); norm = true; }
    else if (alg == 6) { na = sann::strict_log(sum_sq(s1, n1) + 1); nb = sann::strict_log(sum_sq(s2, n2) + 1); norm = true; }
    else if (alg == 7) { na = pow(sum_sq(s1, n1), 0.3); nb = pow(sum_sq(s2, n2), 0.3); norm = true; }
    else if (alg != 1) { out[i] = __builtin_nan(""); return; }
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
    r = product;
  }
  out[i] = r;
}

struct Buf {
  void *p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" {

const char *rsx_last_error(void) { return g_rsx_err.c_str(); }

int rsx_pair_scores_device(int32_t device, void *hip_stream, int32_t algorithm, int32_t n_pairs, const void *d_a_offsets,
                           const void *d_a_cluster_ids, const void *d_a_scores, const void *d_b_offsets,
                           const void *d_b_cluster_ids, const void *d_b_scores, void *d_out_scores) {
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
}

int rsx_pair_scores(int32_t device, int32_t algorithm, int32_t n_pairs, const int64_t *a_offsets,
                    const int32_t *a_cluster_ids, const double *a_scores, const int64_t *b_offsets,
                    const int32_t *b_cluster_ids, const double *b_scores, int32_t validate, double *out_scores) {
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
}

}  // extern "C"
