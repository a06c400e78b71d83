/*
 * oracle_baseline.c -- the `cpu_baseline` leg of bench.py: the reference's CPU path timed on the host cores.
 * TEST / BENCH INFRASTRUCTURE ONLY (see simclusters_oracle.c).  Labelled "C restatement of the Scala CPU
 * path (N threads)", never "JVM" (SURVEY.md 8(d)).
 *
 * Unlike oracle_sann_query (one flat table, only results matter) the two legs here keep what makes the two
 * reference implementations cost what they cost:
 *
 *   variant 0  "original"   ApproximateCosineSimilarity.scala:57-128
 *       TWO hash maps preset to 16384 (:40,78-81): per posting `getOrElse` + `put` on candidateScoresMap and again
 *       on candidateNormalizationMap (:92-96) -- four probes; then every candidate is normalised into a new
 *       sequence (:105-123), filtered (:125), FULLY sorted (:126) and cut (:127).
 *   variant 1  "optimized"  OptimizedApproximateCosineSimilarity.scala:37-111
 *       ONE map tweet -> (score, norm) (:54,69-74): one get + one put per posting; filter before materialising
 *       (:99-105); full sort + take (:108-110); tweet id 0 excluded when the source is not a tweet (:56).
 *
 * Both produce exactly oracle_sann_query's answers (tests/test_oracle_kat.py::test_baseline_legs_equal_the_oracle):
 * same accumulation order, same arithmetic, same total order.  The JVM's boxing, allocation and GC are NOT
 * modelled -- the baseline is conservative (faster than the JVM would be).
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
  int32_t max_num_results; double min_score; int32_t candidate_embedding_type;
  int32_t max_top_tweets_per_cluster, max_scan_clusters, max_tweet_candidate_age_hours,
      min_tweet_candidate_age_hours, ann_algorithm;
} oracle_sann_config;

/* shared with simclusters_oracle.c */
typedef struct oracle_embedding oracle_embedding;
oracle_embedding *oracle_embedding_build(int32_t n, const int32_t *ids, const double *scores, int32_t truncate);
void oracle_embedding_free(oracle_embedding *e);
int32_t oracle_embedding_size(const oracle_embedding *e);
void oracle_embedding_export(const oracle_embedding *e, int32_t *cluster_ids, double *scores, int32_t *sorted_ids,
                             double *sorted_scores);
double oracle_embedding_l2norm(const oracle_embedding *e);
double oracle_embedding_lognorm(const oracle_embedding *e);
double oracle_strict_log(double x);
void oracle_age_window(const oracle_sann_config *cfg, int64_t now_ms, int64_t *earliest, int64_t *latest);

/* ---- a mutable.HashMap[Long, Double] stand-in: open addressing, get and put as separate probes -------------- */
typedef struct { int64_t *key; double *val; uint8_t *used; uint32_t mask; int32_t n; int64_t *order; } lmap;
static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
static void lmap_init(lmap *m, int64_t expected) {
  uint32_t cap = 16384; /* InitialCandidateMapSize, :40 */
  while ((int64_t)cap < expected * 2) cap <<= 1;
  m->mask = cap - 1;
  m->key = malloc(8 * (size_t)cap);
  m->val = malloc(8 * (size_t)cap);
  m->used = calloc(cap, 1);
  m->order = malloc(8 * (size_t)(expected > 16 ? expected : 16)); /* insertion order = iteration order here */
  m->n = 0;
}
static void lmap_free(lmap *m) { free(m->key); free(m->val); free(m->used); free(m->order); }
static double lmap_get_or_else(const lmap *m, int64_t k, double dflt) {
  uint32_t s = (uint32_t)mix64((uint64_t)k) & m->mask;
  while (m->used[s]) {
    if (m->key[s] == k) return m->val[s];
    s = (s + 1) & m->mask;
  }
  return dflt;
}
static void lmap_put(lmap *m, int64_t k, double v) {
  uint32_t s = (uint32_t)mix64((uint64_t)k) & m->mask;
  while (m->used[s]) {
    if (m->key[s] == k) { m->val[s] = v; return; }
    s = (s + 1) & m->mask;
  }
  m->used[s] = 1; m->key[s] = k; m->val[s] = v; m->order[m->n++] = k;
}

/* ---- a java.util.HashMap[Long, (Double, Double)] stand-in: one probe finds or inserts the pair ---------------- */
typedef struct { int64_t id; double dot, nsq; } pair_ent;
typedef struct { pair_ent *e; int32_t *slot; uint32_t mask; int32_t n; } pmap;
static void pmap_init(pmap *m, int64_t expected) {
  uint32_t cap = 16384;
  while ((int64_t)cap < expected * 2) cap <<= 1;
  m->mask = cap - 1;
  m->slot = malloc(4 * (size_t)cap);
  memset(m->slot, 0xff, 4 * (size_t)cap);
  m->e = malloc(sizeof(pair_ent) * (size_t)(expected > 16 ? expected : 16));
  m->n = 0;
}
static void pmap_free(pmap *m) { free(m->slot); free(m->e); }
static pair_ent *pmap_get(pmap *m, int64_t k) { /* get: NULL when absent */
  uint32_t s = (uint32_t)mix64((uint64_t)k) & m->mask;
  while (m->slot[s] >= 0) {
    if (m->e[m->slot[s]].id == k) return &m->e[m->slot[s]];
    s = (s + 1) & m->mask;
  }
  return NULL;
}
static void pmap_put(pmap *m, int64_t k, double dot, double nsq) { /* put: a second probe, as HashMap.put is */
  uint32_t s = (uint32_t)mix64((uint64_t)k) & m->mask;
  while (m->slot[s] >= 0) {
    if (m->e[m->slot[s]].id == k) { m->e[m->slot[s]].dot = dot; m->e[m->slot[s]].nsq = nsq; return; }
    s = (s + 1) & m->mask;
  }
  m->slot[s] = m->n;
  m->e[m->n].id = k; m->e[m->n].dot = dot; m->e[m->n].nsq = nsq; m->n++;
}

typedef struct { int64_t id; double score; } scored;
static int java_double_compare(double a, double b) {
  if (a < b) return -1;
  if (a > b) return 1;
  uint64_t x, y;
  memcpy(&x, &a, 8); memcpy(&y, &b, 8);
  return x == y ? 0 : ((int64_t)x < (int64_t)y ? -1 : 1);
}
static int cmp_scored(const void *pa, const void *pb) { /* sortBy(-score), ties: tweet id ascending */
  const scored *a = pa, *b = pb;
  int c = java_double_compare(-a->score, -b->score);
  if (c) return c;
  return (a->id > b->id) - (a->id < b->id);
}
static int cmp_i32(const void *a, const void *b) { int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }
static int64_t find_list(int32_t n, const int32_t *ids, int32_t c) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (ids[mid] < c) lo = mid + 1; else hi = mid; }
  return (lo < n && ids[lo] == c) ? lo : -1;
}
static double normalise(int32_t alg, double dot, double nsq, double l2, double ln) { /* :111-119 */
  switch (alg) {
    case 3: return dot / ln / oracle_strict_log(1 + nsq);
    case 2: return dot / l2 / sqrt(nsq);
    case 4: return dot / sqrt(nsq);
    case 1: return dot;
    default: return NAN;
  }
}

/* One query through one of the two implementations.  Returns the number of results. */
int32_t oracle_baseline_query(int32_t variant, int32_t n_emb, const int32_t *emb_ids, const double *emb_scores,
                              const oracle_sann_config *cfg, int64_t now_ms, int32_t n_lists,
                              const int32_t *list_cluster_ids, const int64_t *list_offsets, const int64_t *tweet_ids,
                              const double *scores, int64_t *out_ids, double *out_scores, int32_t *map_size) {
  oracle_embedding *emb = oracle_embedding_build(n_emb, emb_ids, emb_scores, -1);
  const int32_t n = oracle_embedding_size(emb);
  int32_t *cl = malloc(4 * (size_t)(n + 1)), *scl = malloc(4 * (size_t)(n + 1));
  double *sc = malloc(8 * (size_t)(n + 1)), *ssc = malloc(8 * (size_t)(n + 1));
  oracle_embedding_export(emb, cl, sc, scl, ssc);
  const double l2 = oracle_embedding_l2norm(emb), ln = oracle_embedding_lognorm(emb);
  int64_t earliest, latest;
  oracle_age_window(cfg, now_ms, &earliest, &latest);
  /* truncate(maxScanClusters).getClusterIds().toSet, iterated ascending */
  int32_t n_scan = cfg->max_scan_clusters < 0 ? 0 : (n <= cfg->max_scan_clusters ? n : cfg->max_scan_clusters);
  int32_t *scan = malloc(4 * (size_t)(n_scan + 1));
  memcpy(scan, cl, 4 * (size_t)n_scan);
  qsort(scan, (size_t)n_scan, 4, cmp_i32);
  int64_t expected = 0;
  for (int32_t c = 0; c < n_scan; c++) {
    int64_t li = find_list(n_lists, list_cluster_ids, scan[c]);
    if (li >= 0) expected += list_offsets[li + 1] - list_offsets[li];
  }
  scored *res = NULL;
  int32_t nres = 0;
  if (variant == 0) {
    lmap score_map, norm_map;
    lmap_init(&score_map, expected);
    lmap_init(&norm_map, expected);
    for (int32_t c = 0; c < n_scan; c++) {
      int64_t li = find_list(n_lists, list_cluster_ids, scan[c]);
      if (li < 0) continue;
      int32_t pos = (int32_t)find_list(n, scl, scan[c]); /* contains + getOrElse on the sorted arrays */
      if (pos < 0) continue;
      const double w = ssc[pos];
      int64_t len = list_offsets[li + 1] - list_offsets[li];
      int64_t lim = len < cfg->max_top_tweets_per_cluster ? len : cfg->max_top_tweets_per_cluster;
      const int64_t *tid = tweet_ids + list_offsets[li];
      const double *s = scores + list_offsets[li];
      for (int64_t i = 0; i < lim; i++) {
        const int64_t t = tid[i];
        if (t >= earliest && t <= latest) { /* no tweet source in the benchmark's queries (:90) */
          lmap_put(&score_map, t, lmap_get_or_else(&score_map, t, 0.0) + s[i] * w); /* :92-94 */
          lmap_put(&norm_map, t, lmap_get_or_else(&norm_map, t, 0.0) + s[i] * s[i]); /* :95-96 */
        }
      }
    }
    *map_size = score_map.n;
    /* candidateScoresMap.map { normalise } .filter(>= minScore) .toSeq .sortBy(-score) .take */
    scored *all = malloc(sizeof(scored) * (size_t)(score_map.n + 1));
    for (int32_t i = 0; i < score_map.n; i++) {
      const int64_t t = score_map.order[i];
      all[i].id = t;
      all[i].score = normalise(cfg->ann_algorithm, lmap_get_or_else(&score_map, t, 0.0), lmap_get_or_else(&norm_map, t, 0.0), l2, ln);
    }
    res = malloc(sizeof(scored) * (size_t)(score_map.n + 1));
    for (int32_t i = 0; i < score_map.n; i++)
      if (all[i].score >= cfg->min_score) res[nres++] = all[i];
    free(all);
    lmap_free(&score_map);
    lmap_free(&norm_map);
  } else {
    pmap map;
    pmap_init(&map, expected);
    for (int32_t c = 0; c < n_scan; c++) {
      int64_t li = find_list(n_lists, list_cluster_ids, scan[c]);
      if (li < 0) continue;
      int32_t pos = (int32_t)find_list(n, scl, scan[c]);
      if (pos < 0) continue;
      const double w = ssc[pos];
      int64_t len = list_offsets[li + 1] - list_offsets[li];
      int64_t lim = len < cfg->max_top_tweets_per_cluster ? len : cfg->max_top_tweets_per_cluster;
      const int64_t *tid = tweet_ids + list_offsets[li];
      const double *s = scores + list_offsets[li];
      for (int64_t i = 0; i < lim; i++) {
        const int64_t t = tid[i];
        if (t != 0 && t >= earliest && t <= latest) { /* Optimized :56,:67: sourceTweetId defaults to 0 */
          const pair_ent *e = pmap_get(&map, t);
          const double d0 = e ? e->dot : 0.0, n0 = e ? e->nsq : 0.0;
          pmap_put(&map, t, d0 + s[i] * w, n0 + s[i] * s[i]); /* :69-74 */
        }
      }
    }
    *map_size = map.n;
    res = malloc(sizeof(scored) * (size_t)(map.n + 1));
    for (int32_t i = 0; i < map.n; i++) { /* :99-105: filter while materialising */
      const double p = normalise(cfg->ann_algorithm, map.e[i].dot, map.e[i].nsq, l2, ln);
      if (p >= cfg->min_score) { res[nres].id = map.e[i].id; res[nres].score = p; nres++; }
    }
    pmap_free(&map);
  }
  qsort(res, (size_t)nres, sizeof(scored), cmp_scored); /* the full sort both implementations pay */
  int32_t k = cfg->max_num_results < 1000 ? cfg->max_num_results : 1000;
  if (k < 0) k = 0;
  if (nres > k) nres = k;
  for (int32_t i = 0; i < nres; i++) { out_ids[i] = res[i].id; out_scores[i] = res[i].score; }
  free(res); free(scan); free(cl); free(scl); free(sc); free(ssc);
  oracle_embedding_free(emb);
  return nres;
}

typedef struct {
  int32_t variant, nq, n_lists, tid, nthreads;
  const int64_t *emb_offsets; const int32_t *emb_ids; const double *emb_scores;
  const oracle_sann_config *cfg; int64_t now_ms;
  const int32_t *list_cluster_ids; const int64_t *list_offsets; const int64_t *tweet_ids; const double *scores;
  int64_t *out_ids; double *out_scores; int32_t *out_counts; int32_t stride;
} job;

static void *worker(void *arg) {
  job *j = arg;
  for (int32_t q = j->tid; q < j->nq; q += j->nthreads) {
    int32_t msz;
    int64_t b = j->emb_offsets[q], e = j->emb_offsets[q + 1];
    j->out_counts[q] = oracle_baseline_query(j->variant, (int32_t)(e - b), j->emb_ids + b, j->emb_scores + b, j->cfg, j->now_ms,
                                             j->n_lists, j->list_cluster_ids, j->list_offsets, j->tweet_ids, j->scores,
                                             j->out_ids + (int64_t)q * j->stride, j->out_scores + (int64_t)q * j->stride, &msz);
  }
  return NULL;
}

/* variant 0 = "original" (two maps), 1 = "optimized" (one map).  out_ids/out_scores: [nq * 1000]; returns elapsed
 * wall seconds. */
double oracle_baseline_run(int32_t variant, int32_t n_threads, int32_t nq, const int64_t *emb_offsets,
                           const int32_t *emb_ids, const double *emb_scores, const oracle_sann_config *cfg,
                           int64_t now_ms, int32_t n_lists, const int32_t *list_cluster_ids, const int64_t *list_offsets,
                           const int64_t *tweet_ids, const double *scores, int64_t *out_ids, double *out_scores,
                           int32_t *out_counts) {
  if (n_threads < 1) n_threads = 1;
  pthread_t *th = malloc(sizeof(pthread_t) * (size_t)n_threads);
  job *jobs = malloc(sizeof(job) * (size_t)n_threads);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int32_t t = 0; t < n_threads; t++) {
    jobs[t] = (job){variant, nq, n_lists, t, n_threads, emb_offsets, emb_ids, emb_scores, cfg, now_ms,
                    list_cluster_ids, list_offsets, tweet_ids, scores, out_ids, out_scores, out_counts, 1000};
    pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  for (int32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(th); free(jobs);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
