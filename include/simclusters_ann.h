/*
 * simclusters_ann.h -- C ABI of the MI355X-native SimClusters-ANN engine.
 *
 * This is the drop-in boundary for the `approximateCosineSimilarity` operator seam of
 * sagspot/the-algorithm (all paths relative to /root/reference/):
 *
 *   trait ApproximateCosineSimilarity.apply
 *     simclusters-ann/server/src/main/scala/com/twitter/simclustersann/candidate_source/ApproximateCosineSimilarity.scala:26-36
 *   selected by flag `approximate_cosine_similarity`
 *     simclusters-ann/server/src/main/scala/com/twitter/simclustersann/modules/SimClustersANNCandidateSourceModule.scala:19-38
 *   called from SimClustersANNCandidateSource.fetchCandidates
 *     simclusters-ann/server/src/main/scala/com/twitter/simclustersann/candidate_source/SimClustersANNCandidateSource.scala:66-95
 *
 * The reference has no FFI on this path; the only JNI precedent is swig-faiss
 * (ann/src/main/java/com/twitter/ann/faiss/swig/swigfaissJNI.java:13-23,269): an opaque native
 * handle (`long swigCPtr`) plus calls that take primitive arrays.  This header follows that
 * shape: opaque handles, plain pointers and sizes, int status codes, caller-owned buffers.
 * INTEGRATION.md shows the JNI stub and the 4th `ApproximateCosineSimilarity` object that a
 * maintainer would add on the Scala side.
 *
 * Threading: an index handle is immutable after build and may be shared by any number of
 * threads.  A batch handle is owned by one thread at a time.  No function throws or aborts;
 * every function returns a status and sets a thread-local message (sann_last_error).
 */
#ifndef SIMCLUSTERS_ANN_H
#define SIMCLUSTERS_ANN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SANN_OK 0
#define SANN_EINVAL 1    /* bad argument */
#define SANN_EDEVICE 2   /* HIP runtime error (message carries hipGetErrorString) */
#define SANN_ENOMEM 3
#define SANN_ELIMIT 4    /* a size exceeds what this build supports (message says which) */
#define SANN_EINTERNAL 5

/* ScoringAlgorithm, simclusters-ann/thrift/src/main/thrift/simClustersAnn.thrift:32-37 */
#define SANN_ALG_DOT_PRODUCT 1
#define SANN_ALG_COSINE 2
#define SANN_ALG_LOG_COSINE 3
#define SANN_ALG_COSINE_NO_SOURCE_NORM 4
/* The offline all-users job (src/scala/com/twitter/simclusters_v2/scio/bq_generation/sql/tweets_ann.sql:44-52,
 * tweets_ann/TweetsANNFromBQ.scala:17-21): same top-N clusters x top-M tweets x dot product, but normalised by the
 * tweet's FULL embedding norm (:10-15) and not by the source: logCosineSimilarityScore = dot / LN(1 + norm) -- the
 * job's ranking key (:57-58) -- and cosineSimilarityScore = dot / SQRT(norm); dotProductScore is SANN_ALG_DOT_PRODUCT.
 * They need an index built with sann_index_build_with_norms; tweets whose norm is not > 0 are dropped (:14); the
 * job has no age window (max / min_tweet_candidate_age_hours are ignored). */
#define SANN_ALG_OFFLINE_LOG_COSINE 5
#define SANN_ALG_OFFLINE_COSINE 6

/* Which reference implementation's edge-case behaviour to reproduce (they agree on every
 * input that SimClustersANNCandidateSource can produce):
 *   ORIGINAL     ApproximateCosineSimilarity.scala:57-128   (no source-tweet exclusion unless the
 *                source id is a tweet id; skips clusters the embedding does not contain)
 *   OPTIMIZED    OptimizedApproximateCosineSimilarity.scala:37-111 (excludes tweet id 0 when the
 *                source is not a tweet, :56)
 *   EXPERIMENTAL ExperimentalApproximateCosineSimilarity.scala:41-130 (no `contains` guard: a
 *                scanned cluster missing from the embedding has weight 0.0, :62-63) */
#define SANN_VARIANT_ORIGINAL 0
#define SANN_VARIANT_OPTIMIZED 1
#define SANN_VARIANT_EXPERIMENTAL 2
/* LEGACY  src/scala/com/twitter/simclusters_v2/candidate_source/SimClustersANNCandidateSource.scala:107-181
 *         (the in-process candidate source cr-mixer used before the service): same accumulate;
 *         no "175200 h" rule in the age window (:113); no minScore filter (min_score is ignored);
 *         ann_algorithm carries its (enablePartialNormalization, rankingAlgorithm) pair:
 *         DOT_PRODUCT = no normalisation, COSINE = partial normalisation, LOG_COSINE = partial
 *         normalisation in the "log" form, which divides by l2norm rather than logNorm (:167-169).
 *         max_num_results above 1000 is refused (SANN_ELIMIT).  The optional heavy re-rank (:182-200,
 *         HeavyRanker.scala:32-77) is this call with max_num_results = maxReRankingCandidates followed
 *         by rsx_store_list_scores; see the Python mirror LegacySimClustersANNCandidateSource. */
#define SANN_VARIANT_LEGACY 3

/* SimClustersANNConfig, simClustersAnn.thrift:18-27 -- field for field. */
typedef struct sann_config {
  int32_t max_num_results;               /* 1: maxNumResults (hard cap 1000, ApproximateCosineSimilarity.scala:41) */
  int32_t candidate_embedding_type;      /* 3: candidateEmbeddingType (carried; not used by the arithmetic) */
  double min_score;                      /* 2: minScore */
  int32_t max_top_tweets_per_cluster;    /* 4 */
  int32_t max_scan_clusters;             /* 5 */
  int32_t max_tweet_candidate_age_hours; /* 6: >= 175200 disables the lower bound (:42,:67-68) */
  int32_t min_tweet_candidate_age_hours; /* 7 */
  int32_t ann_algorithm;                 /* 8: SANN_ALG_* */
  int32_t reserved;
} sann_config_t;

typedef struct sann_index sann_index_t;
typedef struct sann_batch sann_batch_t;

typedef struct sann_index_options {
  int32_t device;       /* HIP device ordinal */
  int32_t n_partitions; /* tweet-hash partitions per cluster list inside this shard; power of two, 0 = default */
  int32_t shard_id;     /* this shard, 0 <= shard_id < n_shards */
  int32_t n_shards;     /* tweet-hash shards (one per GPU); 1 = whole corpus on this GPU */
} sann_index_options_t;

typedef struct sann_index_info {
  int64_t n_clusters;      /* cluster lists held (including empty ones) */
  int64_t n_postings;      /* postings held by this shard */
  int64_t n_postings_total;/* postings in the lists as given (all shards) */
  int64_t device_bytes;
  int32_t n_partitions, shard_id, n_shards, max_list_len;
} sann_index_info_t;

/* Per-batch counters, the native side of `candidateScoresStat` and the server's StatsReceiver
 * (ApproximateCosineSimilarity.scala:32,102). */
typedef struct sann_batch_stats {
  int64_t postings_scanned;  /* sum over queries of sum_c min(len_c, M) within this shard */
  int64_t algorithmic_bytes; /* P_q*16 + n*12 + k_out*16 summed over queries */
  int32_t n_units;           /* (query, partition) work units launched */
  int32_t n_fallback_units;  /* units re-run on the general (global-memory) path */
  int32_t n_requeried;       /* queries whose first-pass top-k could not be proven exact */
  int32_t max_unit_postings; /* largest number of postings one (query, partition) unit scanned (0 = not measured) */
} sann_batch_stats_t;

const char *sann_last_error(void);
/* Environment advice the library has for its process, "" when none.  Today: batches in flight (sann_batch_run_after,
 * sann_get_tweet_candidates from several threads) need GPU_MAX_HW_QUEUES >= 8 -- streams that share a hardware queue
 * execute in order and the overlap is silently lost.  The variable belongs to the process (the HIP runtime reads it once,
 * when it initialises): the launcher / JVM shim exports it before the process starts; the library never
 * changes the environment.  This returns (and stderr shows, once) a message when the process runs with fewer. */
const char *sann_runtime_advice(void);
/* Library self-description, e.g. "simclusters_amd 0.1 gfx950". */
const char *sann_version(void);

/*
 * Build a device-resident cluster -> top tweets index from posting lists as the reference's
 * ReadableStore[ClusterId, Seq[(TweetId, Double)]] returns them
 * (simclusters-ann/.../modules/ClusterTweetIndexProviderModule.scala:34-94 over
 *  src/scala/com/twitter/simclusters_v2/summingbird/stores/TopKTweetsForClusterReadableStore.scala:211-229):
 * each list already filtered to score > 0, sorted by score descending and capped (raw, unsorted postings go
 * through sann_index_build_from_postings, which applies that contract on the device).  The order inside a list
 * is kept as given (position i is the `i` of ApproximateCosineSimilarity.scala:87); the operator itself needs
 * neither order nor sign.  Tweet ids must be unique inside one list (they are keys of a Map in the store):
 * a repeated id is refused with SANN_EINVAL.
 *   cluster_ids[n_lists]      ascending, unique
 *   list_offsets[n_lists + 1] CSR offsets into tweet_ids / scores
 */
int sann_index_build(const sann_index_options_t *opts, int32_t n_lists, const int32_t *cluster_ids,
                     const int64_t *list_offsets, const int64_t *tweet_ids, const double *scores,
                     sann_index_t **out);
/* The same with a per-posting norms column for the offline job's scores (SANN_ALG_OFFLINE_*): tweet_norms[i] = SUM of
 * squares of the FULL embedding of the tweet of posting i (tweets_ann.sql:10-15: tweet_embeddings_norm, which the job
 * joins by tweet id -- here the caller has joined it onto the postings). */
int sann_index_build_with_norms(const sann_index_options_t *opts, int32_t n_lists, const int32_t *cluster_ids,
                                const int64_t *list_offsets, const int64_t *tweet_ids, const double *scores,
                                const double *tweet_norms, sann_index_t **out);
/*
 * The posting-list provider itself, on the device: RAW store entries in, index out.  Per cluster
 * the store holds a map tweet -> DecayedValue(value, scaledTime) (scaledTime = ms * ln 2 / halfLife, 8 h in
 * summingbird/common/Configs.scala:38); what the operator receives is
 *   decay every value to now   summingbird/stores/TopKTweetsForClusterReadableStore.scala:51-71, EntityUtil.scala:10-28,
 *                              ThriftDecayedValueMonoid.scala:33-38 (algebird DecayedValueMonoid(0.0).plus with a zero at now)
 *   keep value > 0.0, sort by value descending, take(max_results)   TopKTweetsForClusterReadableStore.scala:211-229,258-259
 * and that is what this builds -- one workgroup per cluster: decay + filter at load, LDS bitonic sort under the total
 * order (value desc, tweet id asc), cap, tweet-hash partition.  scaled_times == NULL skips the decay (the Manhattan
 * read-only store, :236-260).  Lists of more than 4096 raw entries are refused (SANN_ELIMIT; the store keeps <= 1.2 x
 * topK = 1920: summingbird/common/Monoids.scala:440-449); a tweet id twice in one list is refused (SANN_EINVAL).
 * exp is fdlibm's (java.lang.StrictMath.exp); HotSpot's Math.exp may differ from it in the last ulp.
 */
int sann_index_build_from_postings(const sann_index_options_t *opts, int32_t n_lists, const int32_t *cluster_ids,
                                   const int64_t *list_offsets, const int64_t *tweet_ids, const double *values,
                                   const double *scaled_times, int64_t now_ms, int64_t half_life_ms, int32_t max_results,
                                   sann_index_t **out);
/*
 * The streaming side of the same store: the Summingbird job folds event batches into the per-cluster
 * maps with TopKTweetsWithScoresMonoid.plus (summingbird/common/Monoids.scala:131-158), i.e.
 * TopKScoresUtils.mergeTwoTopKMapWithDecayedValues (:378-450) and the tweet-age filter (:154).  For every list i
 * (a CSR pair of sides a and b, entries = (tweet id, DecayedValue.value, DecayedValue.scaledTime); ids unique per side):
 *   a side empty -> the other side, untouched (:388-394); otherwise every value is decayed to the latest scaledTime of
 *   both sides (DecayedValueMonoid(0.0).plus with a zero at that time), kept if > threshold, the larger value wins for
 *   an id on both sides, and when more than 1.2 x top_k entries remain only the top_k largest are kept (:441-448);
 *   last, ids < oldest_tweet_id are dropped (:142,154).
 * A map has no order: results are written by (value desc, tweet id asc), which is also the tie order of the cut (the
 * reference's is its HashMap's iteration order).  One workgroup per list on the device; at most 4096 entries per list
 * on both sides together (SANN_ELIMIT).  out_offsets[n_lists + 1] is always written; SANN_ELIMIT if out_capacity is
 * too small (out_offsets then tells the size needed).  Production: top_k 1600, threshold 0.001, age 3 days / 1 hour
 * (summingbird/common/Configs.scala:41,56,63-65).
 */
int sann_topk_merge(int32_t device, int32_t n_lists, const int64_t *a_offsets, const int64_t *a_ids, const double *a_values,
                    const double *a_scaled_times, const int64_t *b_offsets, const int64_t *b_ids, const double *b_values,
                    const double *b_scaled_times, int32_t top_k, double threshold, int64_t oldest_tweet_id,
                    int64_t out_capacity, int64_t *out_offsets, int64_t *out_ids, double *out_values,
                    double *out_scaled_times);
/*
 * Generate the synthetic SimClusters corpus of SURVEY.md section 8(d) on the device and build
 * the index from it without a host round trip (per-cluster filter -> sort by score descending
 * -> cap -> partition: the device form of TopKTweetsForClusterReadableStore.scala:211-229).
 * Cluster ids are 1..n_clusters.  Deterministic in (params, seed); independent of n_partitions
 * and of the shard split (every shard generates the same corpus and keeps its tweets).
 */
typedef struct sann_synth_params {
  int64_t n_tweets;
  int64_t now_ms;                 /* tweet ids are Snowflake ids in [now - window, now) */
  uint64_t seed;
  int32_t n_clusters;             /* 144428 in the reference (SimclustersAnnWarmupHandler.scala:33) */
  int32_t index_cap;              /* tweets kept per cluster (2000: simclusters_index_generation/Config.scala:58) */
  int32_t window_hours;           /* 24 */
  int32_t max_clusters_per_tweet; /* 50 */
  float mean_clusters;            /* 25: clusters per tweet ~ min(max, 1 + Geom(1/mean)) */
  int32_t reserved;
} sann_synth_params_t;
int sann_index_build_synthetic(const sann_index_options_t *opts, const sann_synth_params_t *params, sann_index_t **out);
/* Test / audit hook: the full embeddings the generator gives tweets [t0, t0+n): counts[n] and, at a
 * fixed stride of 64 entries per tweet, cluster_ids[n*64] and scores[n*64]. */
int sann_synth_tweet_embeddings(int32_t device, const sann_synth_params_t *params, int64_t t0, int32_t n, int32_t *counts,
                                int32_t *cluster_ids, double *scores);
/* Quality truth of SURVEY.md 8(d): for nq (<= 64) source embeddings, the k tweets of the synthetic
 * corpus with the largest EXACT cosine against their full embeddings (every tweet, every cluster) --
 * the quantity SimClusters-ANN approximates (simclusters-ann/README.md:18-46).  Brute force on the
 * device; out arrays are [nq*k], sorted by cosine descending (ties: tweet id ascending). */
int sann_synth_exact_cosine_topk(int32_t device, const sann_synth_params_t *params, int32_t nq, const int64_t *emb_offsets,
                                 const int32_t *emb_cluster_ids, const double *emb_scores, int32_t k, int64_t *out_ids,
                                 double *out_cos, int32_t *out_counts);
/* Snowflake id the generator gives tweet t (0 <= t < n_tweets). */
int64_t sann_synth_tweet_id(int64_t t, int64_t n_tweets, int64_t now_ms, int32_t window_hours);
int sann_index_info(const sann_index_t *index, sann_index_info_t *info);
/* Copy one cluster's postings held by this shard back to the host, in list order
 * (cap = capacity of the out arrays; *n receives the number held).  Test / audit hook. */
int sann_index_get_list(const sann_index_t *index, int32_t cluster_id, int64_t cap, int64_t *tweet_ids,
                        double *scores, int32_t *ranks, int64_t *n);
int sann_index_destroy(sann_index_t *index);

/*
 * Prepare a batch of nq getTweetCandidates queries against `index`.
 *
 *   emb_offsets[nq+1], emb_cluster_ids, emb_scores
 *       the source SimClustersEmbedding of each query as (clusterId, score) pairs in any order;
 *       the SimClustersEmbedding constructor semantics are applied (drop score <= 0, order by
 *       score desc then cluster id asc; SimClustersEmbedding.scala:490-509).
 *   source_tweet_ids[nq], has_source_tweet[nq]   (both may be NULL = no query has a tweet source)
 *       InternalId.TweetId of sourceEmbeddingId, ApproximateCosineSimilarity.scala:48-55,90.
 *   configs[n_configs]  n_configs is 1 (shared) or nq.
 *   scan_offsets[nq+1], scan_cluster_ids   (both may be NULL)
 *       the keys of clusterTweetsMap in the iteration order the caller wants the accumulation
 *       to follow (given together or not at all: one without the other is SANN_EINVAL).
 *       NULL reproduces SimClustersANNCandidateSource.fetchCandidates:
 *       sourceEmbedding.truncate(maxScanClusters).getClusterIds().toSet, iterated in ascending
 *       cluster id (the reference's order is JVM hash order).
 *   now_ms  the value of Time.now (ApproximateCosineSimilarity.scala:65).
 */
int sann_batch_create(sann_index_t *index, int32_t variant, int64_t now_ms, int32_t nq,
                      const int64_t *emb_offsets, const int32_t *emb_cluster_ids, const double *emb_scores,
                      const int64_t *source_tweet_ids, const uint8_t *has_source_tweet,
                      const sann_config_t *configs, int32_t n_configs, const int64_t *scan_offsets,
                      const int32_t *scan_cluster_ids, sann_batch_t **out);
/*
 * Re-prepare an existing batch object for nq NEW queries (arguments as for sann_batch_create) and enqueue on
 * `hip_stream` whatever the preparation needs on the device.  The object's device buffers and pinned staging memory are
 * kept and only grow, so a front end that holds a few batch objects and resets them per request allocates nothing
 * in steady state.  The queries are prepared ON THE DEVICE (sann_prep.hip: SimClustersEmbedding constructor, norms,
 * truncate(maxScanClusters), contains / getOrElse, cluster -> index row, age window): the host does one O(nq) pass
 * over the arguments and one packed H2D copy.  (Embeddings of more than 1024 entries, SANN_HOST_PREP=1 and
 * SANN_FORCE_GENERAL=1 take the host preparation, which is the same arithmetic.)
 * Asynchronous; the caller's arrays may be reused as soon as it returns.  The batch must not be in flight
 * (sann_batch_finish has returned for its previous run), and sann_batch_run has to follow on the same stream.
 * sann_batch_stats reports postings_scanned / algorithmic_bytes of such a batch once sann_batch_finish has returned.
 */
int sann_batch_reset(sann_batch_t *batch, void *hip_stream, int64_t now_ms, int32_t nq, const int64_t *emb_offsets,
                     const int32_t *emb_cluster_ids, const double *emb_scores, const int64_t *source_tweet_ids,
                     const uint8_t *has_source_tweet, const sann_config_t *configs, int32_t n_configs,
                     const int64_t *scan_offsets, const int32_t *scan_cluster_ids);
/* Enqueue the batch on `hip_stream` (a hipStream_t, NULL = the null stream). Asynchronous;
 * may be called repeatedly (results are overwritten). */
int sann_batch_run(sann_batch_t *batch, void *hip_stream);

/* The same for batches kept in flight on different streams: the dominant (unit) kernel of `batch` does not start
 * before that of `after` (a batch enqueued earlier with this function on another stream; NULL = no predecessor)
 * has finished -- a cross-stream event, no host wait.  Everything else overlaps: this batch's descriptor kernel
 * and its unit kernel run beside `after`'s merge kernel, and the host's wait in sann_batch_finish is off the
 * GPU's critical path.  Per-launch kernel durations stay meaningful because unit kernels never run side by side. */
int sann_batch_run_after(sann_batch_t *batch, void *hip_stream, sann_batch_t *after, int32_t after_merge);
/* Wait for the batch, re-running on the general path whatever the fast path flagged. After
 * this returns SANN_OK the device results are final and exact. */
int sann_batch_finish(sann_batch_t *batch, void *hip_stream);
/* Copy results to the host.  Row q holds out_counts[q] (tweetId, score) pairs sorted by score
 * descending (ties: tweet id ascending) at out_ids + q*out_stride; out_map_sizes[q] is
 * candidateScoresMap.size.  out_stride >= min(max over queries of maxNumResults, 1000). */
int sann_batch_results(sann_batch_t *batch, int64_t *out_ids, double *out_scores, int32_t out_stride,
                       int32_t *out_counts, int32_t *out_map_sizes);
/* Device pointers of the same results (row stride *stride entries), for an on-device merge
 * across shards: d_ids int64[nq*stride], d_scores double[nq*stride], d_counts int32[nq],
 * d_map_sizes int32[nq]. */
int sann_batch_device_results(sann_batch_t *batch, void **d_ids, void **d_scores, void **d_counts,
                              void **d_map_sizes, int32_t *stride);
int sann_batch_stats(sann_batch_t *batch, sann_batch_stats_t *stats);
/* Measurement: sann_batch_run brackets kernels with HIP events on the launch stream and sann_batch_finish
 * adds the elapsed times to running totals.  enable = 1: the unit (gather + accumulate + select) kernel only --
 * two events per run, what a timed region should carry (every event costs the stream ~5 us); enable = 2: the
 * descriptor, unit and merge kernels (four events).  0 = off.  Resets the totals. */
int sann_batch_set_profiling(sann_batch_t *batch, int32_t enable);
int sann_batch_kernel_times(sann_batch_t *batch, double *unit_ms_total, double *merge_ms_total, int32_t *n_runs);
/* Total of the descriptor kernel (fast path only) over the same runs. */
int sann_batch_desc_time(sann_batch_t *batch, double *desc_ms_total);
/* Debug: enable=1 makes the fast unit kernel stamp s_memtime at its phase boundaries into a side
 * buffer; enable=0 returns the averages (avg16[0] = whole unit, avg16[i] = phase i, shader
 * clocks, avg16[15] = units counted) and frees the buffer.  Never quote a run timed this way. */
int sann_debug_phase_cycles(sann_batch_t *batch, int32_t enable, double *avg16);
/* Measurement only: the fast unit kernel's memory side on its own (same grid, descriptors and posting gather, a
 * checksum instead of the arithmetic), averaged over `reps` launches after a batch has run.  mode 0 = one workgroup
 * per unit, as the unit kernel; mode 1 = persistent workgroups (wgs_per_cu per CU) that load the next unit's postings
 * before they consume the current one.  *checksum is the same for both modes. */
int sann_debug_gather_probe(sann_batch_t *batch, int32_t mode, int32_t wgs_per_cu, int32_t reps, double *ms_avg,
                            uint64_t *checksum);
/* Debug: per-unit arrays of the last run ([nq * n_partitions] each; any may be NULL): distinct tweets accumulated,
 * candidates emitted, UNIT_* flags, postings scanned. */
int sann_debug_unit_arrays(sann_batch_t *batch, int32_t *unit_unique, int32_t *cand_cnt, uint32_t *unit_flags, int32_t *unit_T);
/* Test hook, pure host code (no HIP call): what sann_batch_finish decides from a device-written status block -- n_over
 * overflowed unit ids and n_inexact unproven query ids of a batch of nq queries x n_partitions units.  Counts and ids are
 * range-checked exactly as sann_batch_finish checks them: anything outside the batch's shape is SANN_EINTERNAL (never an
 * exception, an abort or a wild write).  *n_units_out / *n_queries_out = units the general path would re-run and queries
 * whose merge would be repeated. */
int sann_debug_plan_slow_tail(int32_t nq, int32_t n_partitions, int32_t n_over, const int32_t *over_units, int32_t n_inexact,
                              const int32_t *inexact_queries, int32_t *n_units_out, int32_t *n_queries_out);
/* Debug: after sann_batch_run + a device sync and BEFORE sann_batch_finish, histogram of why fast
 * units overflowed: [1] too many scanned clusters, [2] too many postings, [3] too many
 * multi-cluster tweets, [4] score outside the fp32 pre-filter range / hash clash, [5] tie group. */
int sann_debug_overflow_reasons(sann_batch_t *batch, int32_t *counts8, int32_t *n_inexact);
/* The shard (GPU) and the in-shard partition a tweet belongs to.  Pure host functions (no HIP call):
 * a front end can use them to route or audit, and the CPU tests use them to emulate shards. */
int32_t sann_tweet_shard(int64_t tweet_id, int32_t n_shards);
int32_t sann_tweet_partition(int64_t tweet_id, int32_t n_partitions);
/* hipDeviceSynchronize on `device` (for callers that do not link the HIP runtime themselves). */
int sann_device_synchronize(int32_t device);
int sann_batch_destroy(sann_batch_t *batch);

/* One call = reset + run + finish + results on a batch object from the index's pool (one per concurrent caller, each
 * with a stream of its own; nothing is allocated in steady state): the shape a JNI stub binds.  Thread-safe.
 * The output arrays are copied at PCIe speed when they live in pinned memory (sann_host_alloc). */
int sann_get_tweet_candidates(sann_index_t *index, int32_t variant, int64_t now_ms, int32_t nq,
                              const int64_t *emb_offsets, const int32_t *emb_cluster_ids,
                              const double *emb_scores, const int64_t *source_tweet_ids,
                              const uint8_t *has_source_tweet, const sann_config_t *configs,
                              int32_t n_configs, const int64_t *scan_offsets, const int32_t *scan_cluster_ids,
                              int64_t *out_ids, double *out_scores, int32_t out_stride, int32_t *out_counts,
                              int32_t *out_map_sizes);

/* The same with one Time.now PER QUERY (now_ms[nq]): requests that were collected into one batch over a fraction of a
 * millisecond keep the age window (ApproximateCosineSimilarity.scala:65-72) each of them would have had alone. */
int sann_get_tweet_candidates_at(sann_index_t *index, int32_t variant, const int64_t *now_ms, int32_t nq,
                                 const int64_t *emb_offsets, const int32_t *emb_cluster_ids,
                                 const double *emb_scores, const int64_t *source_tweet_ids,
                                 const uint8_t *has_source_tweet, const sann_config_t *configs,
                                 int32_t n_configs, const int64_t *scan_offsets, const int32_t *scan_cluster_ids,
                                 int64_t *out_ids, double *out_scores, int32_t out_stride, int32_t *out_counts,
                                 int32_t *out_map_sizes);

/*
 * The micro-batching queue (SURVEY 8(b) "Threading"): the reference calls the operator once per REQUEST, from many Finagle
 * worker threads (SimClustersANNCandidateSource.scala:77-94, 40 ms budget: modules/FlagsModule.scala:8-12); the GPU wants
 * ~1000 requests per launch.  sann_submit copies one request into the open batch and returns a ticket; a batch closes when
 * it holds max_batch requests or max_wait_us after its first request; dispatcher threads (each with a pooled batch object
 * and a HIP stream of its own) run closed batches through sann_get_tweet_candidates_at -- every request with ITS now_ms --
 * and hand each request its rows; sann_wait / sann_poll collect.  A request's answer is bit for bit what
 * sann_get_tweet_candidates returns for it alone.  Requests take fetchCandidates' default cluster selection (explicit scan
 * keys stay with the batch calls).  All functions are thread-safe.
 */
typedef struct sann_batcher sann_batcher_t;
typedef struct sann_batcher_options {
  int32_t variant;        /* SANN_VARIANT_*, one per queue (the service picks it by flag at start-up) */
  int32_t max_batch;      /* requests per batch; 0 = 1024 */
  int32_t max_wait_us;    /* a batch leaves at the latest this long after its first request; 0 = 500 */
  int32_t n_dispatchers;  /* batches in flight; 0 = 3 */
} sann_batcher_options_t;
typedef struct sann_batcher_stats {
  int64_t n_requests, n_batches, n_closed_full, n_closed_by_deadline, max_batch;
} sann_batcher_stats_t;
int sann_batcher_create(sann_index_t *index, const sann_batcher_options_t *options /* NULL = defaults, original variant */,
                        sann_batcher_t **out);
/* Runs whatever was submitted, then stops the dispatchers.  Tickets nobody collected are dropped. */
int sann_batcher_destroy(sann_batcher_t *batcher);
/* One getTweetCandidates request.  cluster_ids / scores: the source embedding (copied: reusable on return).  The out arrays
 * are the caller's and must stay valid until the ticket is collected; out_capacity >= min(maxNumResults, 1000).
 * Rows come sorted (score descending, tweet id ascending); *out_map_size is candidateScoresMap.size. */
int sann_submit(sann_batcher_t *batcher, int64_t now_ms, int32_t n_embedding, const int32_t *cluster_ids, const double *scores,
                int64_t source_tweet_id, int32_t has_source_tweet, const sann_config_t *config, int32_t out_capacity,
                int64_t *out_ids, double *out_scores, int32_t *out_count, int32_t *out_map_size, int64_t *ticket);
/* Block until the request is answered; returns ITS status (and sets this thread's sann_last_error).  A ticket is collected once. */
int sann_wait(sann_batcher_t *batcher, int64_t ticket);
/* *done = 1 and the ticket is collected (status returned) when the answer is there, else *done = 0. */
int sann_poll(sann_batcher_t *batcher, int64_t ticket, int32_t *done);
/* sann_submit + sann_wait: the shape of ApproximateCosineSimilarity.apply for one request. */
int sann_batcher_get_tweet_candidates(sann_batcher_t *batcher, int64_t now_ms, int32_t n_embedding, const int32_t *cluster_ids,
                                      const double *scores, int64_t source_tweet_id, int32_t has_source_tweet,
                                      const sann_config_t *config, int32_t out_capacity, int64_t *out_ids, double *out_scores,
                                      int32_t *out_count, int32_t *out_map_size);
int sann_batcher_stats(sann_batcher_t *batcher, sann_batcher_stats_t *stats);

/* Measurement: with SANN_TRACE_CALLS=1 in the environment every sann_get_tweet_candidates[_at] call adds the wall time of its
 * four stages -- argument pass + packing + H2D + preparation launch | kernel launches | wait for the kernels + status | copy
 * of the answer + wait -- to process-wide totals; this returns (us4[4], microseconds) and resets them. */
int sann_debug_call_trace(double *us4, int64_t *calls);

/*
 * The legacy in-process candidate source END TO END for a batch of queries (src/scala/com/twitter/simclusters_v2/
 * candidate_source/SimClustersANNCandidateSource.scala:107-200): the light rank (SANN_VARIANT_LEGACY) and, when
 * enable_heavy_ranking, the heavy rank fused behind it on the device -- the light top maxReRankingCandidates never leave HBM:
 *   fetchCandidates          accumulate / partial normalisation / sort (:107-181)             the batch kernels, legacy variant
 *   reranking                candidates.take(maxReRankingCandidates) -> HeavyRanker.rank -> sortBy(-score) -> take(maxNumResults)
 *                            (:182-200; HeavyRanker.scala:28-69: pair score of (source embedding id, candidate tweet) through the
 *                            uniform scoring store, kept if >= minScore)                        rsx_heavy_rank_device
 * source_store / tweet_store are representation-scorer stores (include/representation_scorer.h: rsx_store_t) on the index's
 * device; source_internal_ids[q] = the id the heavy ranker looks query q's source embedding up under.  Without heavy ranking
 * the stores may be NULL and the result is the light ranking cut at max_num_results (minScore does not apply there: :160-180).
 * Host arrays in and out, as sann_get_tweet_candidates; thread-safe.
 */
struct rsx_store;
typedef struct sann_legacy_config {      /* case class SimClustersANNConfig of the legacy source, :214-260 */
  int32_t max_num_results;
  int32_t max_tweet_candidate_age_hours, min_tweet_candidate_age_hours;
  int32_t candidate_embedding_type;       /* carried; the tweet store given here IS that type's store */
  double min_score;                       /* heavy rank only */
  int32_t enable_partial_normalization;   /* 0: dot product */
  int32_t enable_heavy_ranking;
  int32_t ranking_algorithm;              /* score.thrift ScoringAlgorithm 1..7 (RSX_PAIR_*); 6 = the "log" partial normalisation */
  int32_t max_reranking_candidates;       /* <= 1000 */
  int32_t max_top_tweets_per_cluster, max_scan_clusters;
} sann_legacy_config_t;
int sann_heavy_rank(sann_index_t *index, const struct rsx_store *source_store, const struct rsx_store *tweet_store, int64_t now_ms,
                    int32_t nq, const int64_t *emb_offsets, const int32_t *emb_cluster_ids, const double *emb_scores,
                    const int64_t *source_tweet_ids, const uint8_t *has_source_tweet, const int64_t *source_internal_ids,
                    const sann_legacy_config_t *config, int64_t *out_ids, double *out_scores, int32_t out_stride,
                    int32_t *out_counts);

/* Pinned (page-locked) host memory for request / response buffers a shim keeps across calls (e.g. behind a direct
 * ByteBuffer): device copies to and from it run at PCIe speed instead of being staged through the runtime. */
int sann_host_alloc(int64_t bytes, void **out);
int sann_host_free(void *p);

/*
 * Merge per-shard results on the device: the `ComposedQueryable` pattern
 * (ann/src/main/scala/com/twitter/ann/common/ShardApi.scala:71-87) applied to tweet-hash
 * shards, where it is exact because every tweet's postings live in one shard.
 * Inputs are the all-gathered device buffers, shard-major: d_ids[n_shards][nq][stride] ...
 * Output rows have stride `stride`; d_out_map_sizes is the sum over shards.
 */
int sann_merge_shards(int32_t device, void *hip_stream, int32_t n_shards, int32_t nq, int32_t stride,
                      int64_t shard_pitch_bytes /* 0 = each array tightly packed shard-major; else every
                      array of shard s starts s*pitch bytes after shard 0's (one packed all-gather) */,
                      const void *d_ids, const void *d_scores, const void *d_counts, const void *d_map_sizes,
                      const void *d_k /* int32[nq]: min(maxNumResults,1000) per query */, void *d_out_ids,
                      void *d_out_scores, void *d_out_counts, void *d_out_map_sizes);

/* The same merge when every shard delivers only its top shard_k (< k) -- what makes an N-way sharded run cheap:
 * a shard holds ~k/N of a query's final top-k, so shard_k = k/N + 6 sigma + 8 entries are enough except with
 * vanishing probability, and both the per-shard merges and the exchange shrink by k / shard_k.  Inputs as for
 * sann_merge_shards (shard_pitch_bytes 0 = d_ids[n_shards][nq][shard_stride] tightly packed; lists sorted, as
 * sann_batch_run leaves them); one k for all queries; outputs with stride out_stride.  Exactness is checked, not assumed: a query is exact iff every list
 * that came in full (count >= shard_k) ends at or below the merged k-th key; *d_inexact_count (device int32, not
 * reset here) is incremented per query that fails, and the caller then repeats the batch with shard_k = k. */
int sann_merge_shards_cut(int32_t device, void *hip_stream, int32_t n_shards, int32_t nq, int32_t shard_stride,
                          int64_t shard_pitch_bytes, int32_t shard_k, int32_t k, int32_t out_stride, const void *d_ids,
                          const void *d_scores, const void *d_counts, const void *d_map_sizes, void *d_out_ids,
                          void *d_out_scores, void *d_out_counts, void *d_out_map_sizes, void *d_inexact_count);

/*
 * The exchange step of the sharded path, over RCCL (xGMI inside a node): one process per GPU, GPU g holding shard g
 * (sann_index_options_t.shard_id / n_shards).  Every rank answers the whole batch on its shard with its outputs bound
 * owner-chunked (sann_batch_bind_outputs_chunked, chunk layout: sann_owner_message_layout), sann_exchange_to_owners
 * delivers chunk r of every rank's buffer to rank r -- ONE all-to-all per batch, a group of ncclSend / ncclRecv on the
 * caller's stream -- and the owner merges what it received with sann_merge_shards / sann_merge_shards_cut
 * (shard_pitch_bytes = the chunk size).  The reference's pattern: ComposedQueryable, ann/.../common/ShardApi.scala:71-87.
 * No Python or torch involved: rank 0 calls sann_comm_unique_id, ships the 128 bytes to the other ranks over any
 * control channel, every rank calls sann_comm_create (collective: all ranks must call it).
 */
typedef struct sann_comm sann_comm_t;
int sann_comm_unique_id(void *id128 /* out: 128 bytes (an ncclUniqueId) */);
int sann_comm_create(int32_t device, int32_t rank, int32_t world, const void *id128, sann_comm_t **out);
int sann_comm_info(const sann_comm_t *comm, int32_t *rank, int32_t *world);
int sann_comm_destroy(sann_comm_t *comm);
/* d_send = [world][chunk_bytes] (chunk r is for rank r), d_recv = [world][chunk_bytes] (chunk s came from rank s);
 * asynchronous on hip_stream. */
int sann_exchange_to_owners(sann_comm_t *comm, void *hip_stream, const void *d_send, void *d_recv, int64_t chunk_bytes);
/* Byte layout of one owner's message for queries_per_owner queries of `stride` entries: ids at 0, score bits at
 * *off_scores, counts at *off_counts, map sizes at *off_map_sizes; *chunk_bytes in all (a multiple of 8).  Pure host
 * arithmetic. */
int sann_owner_message_layout(int32_t queries_per_owner, int32_t stride, int64_t *chunk_bytes, int64_t *off_scores,
                              int64_t *off_counts, int64_t *off_map_sizes);

/*
 * The CLUSTER-ID-RANGE deployment north_star names: GPU g holds the WHOLE lists of the
 * clusters in its range (an ordinary index, n_shards = 1), so a candidate's score terms are spread over GPUs.  The exact way
 * to bring them together is to move postings, not partial sums: per batch every GPU sends the top-M prefix of each list the
 * batch scans to the GPU hash(tweet id) % N names, the receiver builds a temporary index of ITS tweets' postings and runs the
 * ordinary pipeline on it, and the owners merge as in the tweet-hash deployment (sann_exchange_to_owners +
 * sann_merge_shards[_cut]; ComposedQueryable, ann/.../common/ShardApi.scala:71-87).  The pieces:
 *   sann_index_export_prefix_counts      counts[cluster][dest] of the prefixes (rank < M) of the listed clusters, on the device
 *   sann_index_export_prefixes_device    the postings themselves into d_out, segment (cluster, dest) at segment_offsets (in postings,
 *                                        caller-computed from the counts: destination-major, clusters ascending), in rank order
 *   sann_exchange_postings_by_tweet_hash grouped ncclSend / ncclRecv with per-peer counts (both sides know them)
 *   sann_index_build_from_device_postings  the receiver's temporary index from device-resident lists, list order kept
 * All queries of such a batch share one maxTopTweetsPerCluster (the prefix that travels).  Clusters the index does not hold
 * count zero.  bench.py --sharding cluster-range drives them; tests/test_cluster_range_gpu.py does with logical shards.
 */
int sann_index_export_prefix_counts(sann_index_t *index, void *hip_stream, int32_t n_clusters, const int32_t *clusters, int32_t M,
                                    int32_t n_ranks, int32_t *counts /* host, [n_clusters][n_ranks] */);
int sann_index_export_prefixes_device(sann_index_t *index, void *hip_stream, int32_t n_clusters, const int32_t *clusters, int32_t M,
                                      int32_t n_ranks, const int64_t *segment_offsets /* host, [n_clusters][n_ranks] */, void *d_out);
int sann_exchange_postings_by_tweet_hash(sann_comm_t *comm, void *hip_stream, const void *d_send, const int64_t *send_counts /* [world] postings */,
                                         void *d_recv, const int64_t *recv_counts);
int sann_index_build_from_device_postings(const sann_index_options_t *opts, int32_t n_lists, const int32_t *cluster_ids,
                                          const int64_t *list_offsets /* host CSR, in postings */, const void *d_postings, sann_index_t **out);
/* Device memory for callers that do not link the HIP runtime themselves (staging buffers of the exchange above). */
int sann_device_alloc(int32_t device, int64_t bytes, void **out);
int sann_device_free(int32_t device, void *p);
/* dst / src: device or host pointers (the runtime sorts out which); synchronous. */
int sann_device_copy(int32_t device, void *dst, const void *src, int64_t bytes);

/* Make the merge kernel write the final results into caller-owned device buffers (e.g. torch
 * tensors that feed an all-gather) instead of the batch's own; pass four NULLs to unbind.
 * Sizes: int64[nq*stride], double[nq*stride], int32[nq], int32[nq], stride as reported by
 * sann_batch_device_results.  The binding survives sann_batch_reset; the batch remembers the shape (nq, stride) it had
 * when it was bound, and a reset to more queries or a larger maxNumResults is refused with SANN_EINVAL. */
int sann_batch_bind_outputs(sann_batch_t *batch, void *d_ids, void *d_scores, void *d_counts, void *d_map_sizes);

/* The same, for results that leave in one all-to-all: query q is written into chunk q / queries_per_chunk (the
 * chunk of its owner), chunk_pitch_bytes after the previous chunk, at position q % queries_per_chunk of that
 * chunk's arrays (ids / scores with the batch's stride, counts, map sizes).  Pointing the four bases into one
 * buffer -- ids at 0, scores after queries_per_chunk*stride*8 bytes, counts after twice that, map sizes
 * queries_per_chunk*4 bytes later, pitch = the sum -- makes each chunk one contiguous message, and the received
 * buffer is what sann_merge_shards / sann_merge_shards_cut read with shard_pitch_bytes = the same pitch.
 * sann_batch_results / sann_batch_device_results refuse a batch bound this way. */
int sann_batch_bind_outputs_chunked(sann_batch_t *batch, void *d_ids, void *d_scores, void *d_counts, void *d_map_sizes,
                                    int32_t queries_per_chunk, int64_t chunk_pitch_bytes);

/* Device pointer to int32[nq] holding min(max(maxNumResults,0),1000) per query. */
int sann_batch_device_k(sann_batch_t *batch, void **d_k);

/* Audit hook: out[i] = the normalisation of ApproximateCosineSimilarity.scala:111-119 evaluated on
 * the device for (dot[i], nsq[i]); host arrays in and out.  Lets a test compare the device's
 * fp64 division / sqrt / log with the host bit for bit. */
int sann_debug_normalise(int32_t device, int32_t alg, int32_t n, const double *dot, const double *nsq, double l2norm,
                         double lognorm, double *out);

/* Audit hook: values[64 * n_waves] -- every group of 64 is sorted descending in place by one wavefront with the unit
 * kernel's in-register bitonic network (quad permutes, row shifts, gfx950 row / half-wave swaps). */
int sann_debug_wave_sort(int32_t device, int32_t n_waves, uint32_t *values);
/* Audit hook: the fp32 pre-filter score the fast unit kernel gives a single-cluster candidate (posting score s[i],
 * cluster weight w[i]) under `alg`, by the same device function; out_forced[i] = 1 where the exact score is +inf / NaN
 * and the candidate is kept unconditionally.  *eps receives the bound the kernel's cut assumes on
 * |approx / exact - 1| (exact = ApproximateCosineSimilarity.scala:111-119).  Host arrays in and out. */
int sann_debug_approx(int32_t device, int32_t alg, int32_t n, const double *s, const double *w, double l2norm,
                      double lognorm, float *out, uint8_t *out_forced, double *eps);

#ifdef __cplusplus
}
#endif
#endif
