/*
 * dense_ann.h -- C ABI of the brute-force dense nearest-neighbour search (ann/), MI355X.
 *
 * Replaces `Queryable.queryWithDistance` of the exhaustive index (all paths relative to
 * /root/reference/ann/src/main/):
 *   scala/com/twitter/ann/brute_force/BruteForceIndex.scala:66-91   linear scan + size-k max-heap,
 *                                                                   result ascending by distance
 *   scala/com/twitter/ann/common/Api.scala:24-51                    trait Queryable[T, P, D]
 *   scala/com/twitter/ann/common/Metric.scala:88-185,263-289        L2 / Cosine (1 - cos) / InnerProduct (1 - dot)
 *   scala/com/twitter/ann/hnsw/DistanceFunctionGenerator.scala:12-30, hnsw/Hnsw.scala:149-155
 *                                                                   Cosine = L2-normalise stored vectors
 *                                                                   and queries, then InnerProduct
 *   thrift/com/twitter/ann/common/ann_common.thrift:16-19           enum DistanceMetric { L2, Cosine, InnerProduct }
 * The call shape follows the reference's JNI precedent, swig-faiss `Index.search(n, x, k, distances,
 * labels)` (java/com/twitter/ann/faiss/swig/swigfaissJNI.java:269).  It is also the exact truth
 * generator of the reference's load test (scala/com/twitter/ann/service/loadtest, KnnTruthSetGenerator).
 *
 * Arithmetic: vectors and queries are rounded to fp16, products accumulate in fp32 on the matrix
 * cores.  The reference's fp32 `EmbeddingMath` is not vendored and no test pins it: dense parity is
 * "unpinned" against the JVM and is defined against an fp32-accumulate restatement on the same
 * fp16-rounded inputs (tolerance 1e-5 on distances, tests/test_dense_gpu.py).
 */
#ifndef DENSE_ANN_H
#define DENSE_ANN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DANN_OK 0
#define DANN_EINVAL 1
#define DANN_EDEVICE 2
#define DANN_ELIMIT 3
#define DANN_ENOMEM 4    /* host allocation failed */
#define DANN_EINTERNAL 5 /* an unexpected C++ exception was caught at the ABI; the message says which */

/* ann_common.thrift:16-19 */
#define DANN_METRIC_L2 0
#define DANN_METRIC_COSINE 1
#define DANN_METRIC_INNER_PRODUCT 2

typedef struct dann_index dann_index_t;

const char *dann_last_error(void);

/* Build from host vectors (row-major fp32 [n][d]); ids NULL = 0..n-1.  d must be a multiple of 16
 * and <= 512.  Cosine stores L2-normalised vectors. */
int dann_index_build(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                     dann_index_t **out);
/* The same, keeping the fp32 rows beside the fp16 ones (4 d bytes more per vector).  Searches on such an index score
 * their survivors a second time from the fp32 rows -- fp32 operands (Cosine: normalised in fp32), fp32 accumulation, as
 * BruteForceIndex.scala:66-91 does for every vector -- and prove, per query, that no vector the fp16 pass left out can
 * reach the k-th fp32 score (else the pass repeats for that query with a lower threshold): results are those of an
 * exact fp32 scan of the ORIGINAL vectors, up to the order of fp32 summation. */
int dann_index_build_exact(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                           dann_index_t **out);
/* Synthetic index generated on the device: i.i.d. N(0,1) components (BASELINE configs[3]: 50M x 256). */
int dann_index_build_synthetic(int32_t device, int32_t metric, int64_t n, int32_t d, uint64_t seed, dann_index_t **out);
/* The stored (fp16-rounded, for Cosine normalised) vectors [i0, i0+n) as fp32: audit / oracle input. */
int dann_index_get_vectors(const dann_index_t *index, int64_t i0, int64_t n, float *out);
int dann_index_destroy(dann_index_t *index);

/* nq queries (row-major fp32 [nq][d]) -> for each the k nearest stored vectors, ascending by distance
 * (ties: id ascending): out_dist[nq*k], out_ids[nq*k], out_counts[nq] (= min(k, n)).  k <= 1024.
 * Distances: L2 = ||q - x||, Cosine = 1 - cos(q, x), InnerProduct = 1 - <q, x>. */
int dann_search(dann_index_t *index, int32_t nq, const float *queries, int32_t k, float *out_dist, int64_t *out_ids,
                int32_t *out_counts);
/* Full passes over the index the last dann_search needed: 1, plus one per round in which some query overflowed its
 * survivor buffer or (exact mode) failed its completeness proof and was re-armed with a lower threshold. */
int dann_last_rounds(const dann_index_t *index, int32_t *rounds);
/* Milliseconds spent in the two GEMM passes and the selection of the last dann_search (HIP events). */
int dann_last_timing(const dann_index_t *index, float *gemm_a_ms, float *gemm_b_ms, float *select_ms);

/* ComposedQueryable.queryWithDistance (ann/src/main/scala/com/twitter/ann/common/ShardApi.scala:71-87) for batched answers:
 * every shard (one index per GPU, each over its slice of the vectors) was asked for k_in neighbours per query; concatenate,
 * order by (distance ascending, id ascending), keep k.  Host arithmetic (the reference merges on the JVM too); exact, because a
 * distance is not a sum across shards.  Also composes hnsw_search answers.
 *   ids / dist: [n_shards][nq][k_in], counts: [n_shards][nq]; out_ids / out_dist: [nq][k], out_counts: [nq]. */
int dann_compose_shards(int32_t n_shards, int32_t nq, int32_t k_in, const int64_t *ids, const float *dist, const int32_t *counts,
                        int32_t k, int64_t *out_ids, float *out_dist, int32_t *out_counts);

#ifdef __cplusplus
}
#endif
#endif
