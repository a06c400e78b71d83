/*
 * representation_scorer.h -- C ABI of the batched SimClusters pair scorer (representation-scorer).
 *
 * Replaces the `score: (SimClustersEmbedding, SimClustersEmbedding) => Future[Option[Double]]`
 * member of the pair score stores (all paths relative to /root/reference/):
 *   src/scala/com/twitter/simclusters_v2/score/SimClustersEmbeddingPairScoreStore.scala:39-199
 *   bound to algorithm ids by src/thrift/com/twitter/simclusters_v2/score.thrift:14-22 and
 *   representation-scorer/server/src/main/scala/com/twitter/representationscorer/scorestore/ScoreStore.scala:133-164
 * and is what ScoreFacadeStore.multiGet's homogeneous-batch fast path
 * (src/scala/com/twitter/simclusters_v2/score/ScoreFacadeStore.scala:25-51) would call once per batch.
 * Embedding hydration (`PairScoreStore.multiGet`, score/ScoreStore.scala:56-69) and the
 * `None`-if-either-side-is-missing rule stay on the JVM side.
 *
 * Inputs are embeddings AS THE CLASS HOLDS THEM: `sortedClusterIds` ascending with their
 * `sortedScores`, all scores > 0, ids unique (the SimClustersEmbedding invariants,
 * src/scala/com/twitter/simclusters_v2/common/SimClustersEmbedding.scala:28-41).  With
 * validate != 0 the call checks that on the host and fails with RSX_EINVAL otherwise.
 */
#ifndef REPRESENTATION_SCORER_H
#define REPRESENTATION_SCORER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RSX_OK 0
#define RSX_EINVAL 1
#define RSX_EDEVICE 2
#define RSX_ENOMEM 3    /* host allocation failed */
#define RSX_EINTERNAL 4 /* an unexpected C++ exception was caught at the ABI; the message says which */

/* ScoringAlgorithm, score.thrift:14-22 */
#define RSX_PAIR_DOT_PRODUCT 1
#define RSX_PAIR_COSINE 2        /* merge-dot of the PRE-NORMALISED arrays, SimClustersEmbedding.scala:202-208 */
#define RSX_PAIR_JACCARD 3
#define RSX_PAIR_EUCLIDEAN 4     /* fold over the union in ascending cluster id (the reference folds a Set) */
#define RSX_PAIR_MANHATTAN 5
#define RSX_PAIR_LOG_COSINE 6
#define RSX_PAIR_EXP_SCALED 7    /* norm = pow(sum x^2, 0.3): device pow, within 2 ulp of libm */

const char *rsx_last_error(void);

/* Host arrays in, host array out (the shape a JNI stub binds).  CSR offsets int64[n_pairs+1]. */
int rsx_pair_scores(int32_t device, int32_t algorithm, int32_t n_pairs, const int64_t *a_offsets,
                    const int32_t *a_cluster_ids, const double *a_scores, const int64_t *b_offsets,
                    const int32_t *b_cluster_ids, const double *b_scores, int32_t validate, double *out_scores);

/* Device pointers in and out, enqueued on `hip_stream`; for callers that keep embeddings resident. */
int rsx_pair_scores_device(int32_t device, void *hip_stream, int32_t algorithm, int32_t n_pairs, const void *d_a_offsets,
                           const void *d_a_cluster_ids, const void *d_a_scores, const void *d_b_offsets,
                           const void *d_b_cluster_ids, const void *d_b_scores, void *d_out_scores);

/* ---- resident embedding stores: hydration by id on the library side ----
 * A store is the device-resident counterpart of one `ReadableStore[SimClustersEmbeddingId, SimClustersEmbedding]`
 * (one (embeddingType, modelVersion)); ids strictly ascending, embeddings in the class's form. */
typedef struct rsx_store rsx_store_t;
int rsx_store_build(int32_t device, int64_t n, const int64_t *ids, const int64_t *offsets, const int32_t *cluster_ids,
                    const double *scores, rsx_store_t **out);
int rsx_store_destroy(rsx_store_t *store);

/* PairScoreStore.multiGet (src/scala/com/twitter/simclusters_v2/score/ScoreStore.scala:41-69): hydrate both
 * sides, score; out_present[i] = 0 (None) when either id is not in its store. */
int rsx_store_pair_scores(const rsx_store_t *a, const rsx_store_t *b, int32_t algorithm, int32_t n_pairs,
                          const int64_t *a_ids, const int64_t *b_ids, double *out_scores, uint8_t *out_present);

/* ListScoreColumn.fetch (representation-scorer/server/src/main/scala/com/twitter/representationscorer/columns/
 * ListScoreColumn.scala:53-115): one target against a list of candidates, answers in candidate order,
 * None for a candidate (or a target) without an embedding. */
int rsx_store_list_scores(const rsx_store_t *targets, const rsx_store_t *candidates, int32_t algorithm, int64_t target_id,
                          int32_t n_candidates, const int64_t *candidate_ids, double *out_scores, uint8_t *out_present);

/* The heavy-rank step of the legacy SimClusters-ANN candidate source ON THE DEVICE, for nq queries at once (one workgroup per
 * query): HeavyRanker.UniformScoreStoreRanker.rank (src/scala/com/twitter/simclusters_v2/candidate_source/HeavyRanker.scala:28-69)
 * followed by reranking's sort and cut (SimClustersANNCandidateSource.scala:182-200).  Query q's light candidates are the first
 * d_light_counts[q] (<= 1024) tweet ids of row q of d_light_ids (rows of light_stride entries: the device results of a
 * SimClusters-ANN batch, already cut at maxReRankingCandidates); each is scored as pair(source_store[d_source_ids[q]],
 * tweet_store[candidate]) -- both sides hydrated by id inside the kernel; a missing side is the reference's None and drops the
 * candidate -- kept if score >= min_score (:63), sorted by score descending (ties: tweet id ascending), cut at
 * max_num_results (<= out_stride).  All pointers are device pointers; asynchronous on hip_stream.  sann_heavy_rank
 * (include/simclusters_ann.h) is the fused call a shim binds. */
int rsx_heavy_rank_device(const rsx_store_t *source_store, const rsx_store_t *tweet_store, void *hip_stream, int32_t algorithm,
                          int32_t nq, const void *d_source_ids, const void *d_light_ids, const void *d_light_counts,
                          int32_t light_stride, double min_score, int32_t max_num_results, int32_t out_stride, void *d_out_ids,
                          void *d_out_scores, void *d_out_counts);
/* The device a store lives on. */
int rsx_store_device(const rsx_store_t *store, int32_t *device);

/* Scorer.computeSimilarityScoresPerTweet (representation-scorer/.../twistlyfeatures/Scorer.scala:157-369) with
 * Scorer.avg / Scorer.max (:426-429), for n_candidates tweets at once.
 *   maps:   map m = the ids that were scored against every candidate through store map_stores[m]
 *           (`engagements.tweetIds` / `authorIds`, Engagements.scala:28-33; duplicates are kept and count),
 *           map_ids[map_id_offsets[m] .. map_id_offsets[m+1]).
 *   groups: group g = an ordered signal list (favs7d, favs1d, ...) looked up in map group_map[g];
 *           members group_member_ids[group_offsets[g] .. group_offsets[g+1]).
 * out_*[c * n_groups + g]: count = number of scores folded (0 => both features are None), avg = left-fold
 * sum / count, max = fold of max from 0.0.  Pair orientation: score(map embedding, candidate embedding). */
int rsx_store_group_features(const rsx_store_t *candidates, int32_t algorithm, int32_t n_candidates,
                             const int64_t *candidate_ids, int32_t n_maps, const rsx_store_t *const *map_stores,
                             const int64_t *map_id_offsets, const int64_t *map_ids, int32_t n_groups,
                             const int32_t *group_map, const int64_t *group_offsets, const int64_t *group_member_ids,
                             double *out_avg, double *out_max, int32_t *out_count);

#ifdef __cplusplus
}
#endif
#endif
