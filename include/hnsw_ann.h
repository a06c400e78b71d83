/*
 * hnsw_ann.h -- C ABI of the HNSW graph search (ann/hnsw), MI355X.
 *
 * Replaces the query side of the reference's Java HNSW (all paths relative to
 * /root/reference/ann/src/main/):
 *   java/com/twitter/ann/hnsw/HnswIndex.java:538-553    searchKnn(query, numOfNeighbours, ef)
 *   java/com/twitter/ann/hnsw/HnswIndex.java:447-475    bestEntryPointUntilLayer (greedy descent)
 *   java/com/twitter/ann/hnsw/HnswIndex.java:571-623    searchLayerForCandidates (beam search, layer 0)
 *   java/com/twitter/ann/hnsw/DistancedItemQueue.java   min / max queues = java.util.PriorityQueue ordered by
 *                                                       Float.compare on the distance (:37-43)
 *   scala/com/twitter/ann/hnsw/Hnsw.scala:95-147        queryWithDistance: ef from HnswParams, Cosine =
 *                                                       normalised vectors + InnerProduct (:139-155)
 *   scala/com/twitter/ann/hnsw/DistanceFunctionGenerator.scala:12-30
 * The graph is data: `Map<HnswNode(level, item), ImmutableList<item>>` + HnswMeta(maxLevel, entryPoint)
 * (HnswIndex.java:56-72), what HnswIndexIOUtil reads from an index directory.  hnsw_index_build takes
 * exactly that, as flat arrays.  hnsw_index_build_insert builds a graph with the reference's insertion
 * algorithm (HnswIndex.java:137-200,384-440,479-526) on the host; the reference builds offline (SURVEY 8
 * row D4) and any graph it wrote can be loaded instead.
 *
 * Search results are a function of (graph, float distances): the walk reproduces the reference step by
 * step, including java.util.PriorityQueue's sift order, so equal distances are handled as the JVM would.
 * Distances: vectors and queries rounded to fp16, products and sums in fp32 in a fixed order (8 strided
 * partial sums of 8-element chunks, then a pairwise tree) that oracle/hnsw_oracle.c repeats bit for bit.
 * The reference's own fp32 arithmetic (EmbeddingMath, un-vendored) is not pinned by any fixture: parity
 * with the JVM is "unpinned" in the float distances, exact in the walk given the distances.
 */
#ifndef HNSW_ANN_H
#define HNSW_ANN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HNSW_OK 0
#define HNSW_EINVAL 1
#define HNSW_EDEVICE 2
#define HNSW_ELIMIT 3
#define HNSW_ENOMEM 4    /* host allocation failed */
#define HNSW_EINTERNAL 5 /* an unexpected C++ exception was caught at the ABI; the message says which */

/* ann_common.thrift:16-19 */
#define HNSW_METRIC_L2 0
#define HNSW_METRIC_COSINE 1
#define HNSW_METRIC_INNER_PRODUCT 2

typedef struct hnsw_index hnsw_index_t;

const char *hnsw_last_error(void);

/* Load a graph.  Items are positions 0..n-1 of `vectors` (row-major fp32 [n][d], d <= 512); `ids` (or NULL =
 * position) are what searches return.  Graph entry e is HnswNode(entry_level[e], entry_item[e]) with
 * neighbours entry_neighbours[entry_offsets[e] .. entry_offsets[e+1]) in list order.  max_m bounds the
 * lists: 2*max_m at level 0, max_m above (HnswIndex.java:117).  entry_point < 0 = empty index. */
int hnsw_index_build(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                     int32_t max_m, int64_t entry_point, int32_t max_level, int64_t n_entries, const int32_t *entry_level,
                     const int64_t *entry_item, const int64_t *entry_offsets, const int64_t *entry_neighbours,
                     hnsw_index_t **out);

/* Build the graph with the reference's insertion algorithm (HnswIndex.insert), then load it.
 * Level draw: (int)(-ln(U) / ln(max_m)) with U from a 64-bit mixer of (seed, item) (HnswIndex.java:118,369-371).
 * n_threads = 1 inserts items 0..n-1 in order: deterministic.  With more host threads items are inserted
 * concurrently under per-item locks, as the reference's writers do (HnswIndex.java:150-200): the graph then
 * depends on the interleaving (and, as the reference notes at :376-380, may miss a few links). */
int hnsw_index_build_insert(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                            int32_t max_m, int32_t ef_construction, uint64_t seed, int32_t n_threads, hnsw_index_t **out);

/* The graph back as flat arrays (two calls: sizes, then contents) and the stored (fp16-rounded) vectors. */
/* The same, with every item's level given by the caller instead of drawn (the reference draws it from a thread-local
 * Random, HnswIndex.java:369-371): a deterministic rebuild, and the form the oracle's restatement of insert is compared
 * with (tests/test_hnsw_build_gpu.py).  levels[i] in 0..60. */
int hnsw_index_build_insert_levels(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors,
                                   const int64_t *ids, int32_t max_m, int32_t ef_construction, const int32_t *levels,
                                   int32_t n_threads, hnsw_index_t **out);
/* Build on the device, every step of it, and deterministically: two builds of one input give one graph.  The reference's
 * multi-writer insertion (HnswIndex.java:150-200; "when using concurrent writers we can miss connections", :376-380) with the
 * interleaving fixed:
 *   order   items by (level descending, position ascending); the first is the entry point and carries maxLevel
 *   rounds  the next min(batch, max(1, linked / 8)) items of the order are inserted against ONE snapshot of the graph
 *   A       per item, wireConnectionForAllLayers (:137-148): bestEntryPointUntilLayer, then per layer
 *           searchLayerForCandidates(efConstruction), selectNearestNeighboursByHeuristic(maxM), the item's own list,
 *           neighbours.get(0) as the next layer's entry; back links are recorded as (layer, neighbour, order index)
 *   B       per (layer, node), all its additions of the round in order-index order: appended while the list has room
 *           (:414-417), else one re-selection by the heuristic over old list ++ additions sorted ascending by
 *           (Float.compare distance, position) (:419-427 does that per addition)
 * Bounds the reference does not have, counted in hnsw_index_build_stats: a walk's candidate queue holds 1024 entries (when
 * full, entries beyond the current bound -- never expanded, :589-591 -- are dropped and the heap rebuilt in array order); a
 * re-selection sees the first 1024 of old list ++ additions.  oracle/hnsw_oracle.c restates exactly this
 * (oracle_hnsw_build_batched); tests/test_hnsw_gpu_build_gpu.py compares the graphs entry for entry.
 * ef_construction <= 256; batch = items per round (0 = 4096).  _levels: every item's level given (0..60) instead of drawn. */
int hnsw_index_build_insert_gpu(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                                int32_t max_m, int32_t ef_construction, uint64_t seed, int32_t batch, hnsw_index_t **out);
int hnsw_index_build_insert_gpu_levels(int32_t device, int32_t metric, int64_t n, int32_t d, const float *vectors, const int64_t *ids,
                                       int32_t max_m, int32_t ef_construction, const int32_t *levels, int32_t batch, hnsw_index_t **out);
/* Counters of the last device build of this index (any pointer may be NULL): rounds run, additions a re-selection did not see,
 * candidate-queue prunes, candidates dropped because a pruned queue was still full. */
int hnsw_index_build_stats(const hnsw_index_t *index, int64_t *rounds, int64_t *unseen_additions, int64_t *queue_prunes,
                           int64_t *dropped_candidates);
int hnsw_index_graph_size(const hnsw_index_t *index, int64_t *n_entries, int64_t *n_neighbours, int64_t *entry_point,
                          int32_t *max_level);
int hnsw_index_graph(const hnsw_index_t *index, int32_t *entry_level, int64_t *entry_item, int64_t *entry_offsets,
                     int64_t *entry_neighbours);
/* n vectors of dimension d, the metric and maxM the index was created with; the keys searches return (positions
 * when the index was given no ids) */
int hnsw_index_info(const hnsw_index_t *index, int64_t *n, int32_t *d, int32_t *metric, int32_t *max_m);
int hnsw_index_get_ids(const hnsw_index_t *index, int64_t *out);
int hnsw_index_get_vectors(const hnsw_index_t *index, int64_t i0, int64_t n, float *out);
int hnsw_index_destroy(hnsw_index_t *index);

/* searchKnn for nq queries (row-major fp32 [nq][d]): out_dist / out_ids [nq][k] ascending by distance,
 * out_counts[nq] = neighbours found (<= k).  ef as HnswParams.ef; the beam is max(ef, k) (HnswIndex.java:545).
 * max(ef, k) <= 1024. */
int hnsw_search(hnsw_index_t *index, int32_t nq, const float *queries, int32_t k, int32_t ef, float *out_dist,
                int64_t *out_ids, int32_t *out_counts);

/* Work counters of the last hnsw_search: distance evaluations, layer-0 expansions, queries that needed the
 * global-memory queues, and the kernel time (HIP events, ms). */
int hnsw_last_stats(const hnsw_index_t *index, int64_t *distance_evals, int64_t *expansions, int32_t *spilled_queries,
                    float *kernel_ms);
/* More counters of the last hnsw_search: neighbours admitted to the queues (each costs an offer to both queues and, once the
 * result queue is full, a poll), and the largest candidate queue any query of the batch reached. */
int hnsw_last_walk_counters(const hnsw_index_t *index, int64_t *admissions, int64_t *largest_candidate_queue);

#ifdef __cplusplus
}
#endif
#endif
