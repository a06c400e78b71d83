/*
 * simclusters_oracle.c -- CPU restatement of the SimClusters-ANN / representation-scorer
 * arithmetic of sagspot/the-algorithm, in plain C (fp64, single-threaded per call).
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product path (the-algorithm_amd/) never links, imports
 * or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this path
 *  and its Scala cannot be compiled here (no JVM).  The
 * restatement is pinned by source only, plus the hand-derived known-answer vectors in
 * tests/golden/sann_kat.json (derivable from the formulas cited below).
 *
 * All citations are relative to /root/reference/.
 *   SANN  = simclusters-ann/server/src/main/scala/com/twitter/simclustersann/candidate_source/
 *   COMMON= src/scala/com/twitter/simclusters_v2/common/
 *
 * Behaviours the reference leaves to JVM hash order, fixed here (and documented in DESIGN.md):
 *  (1) accumulation order over the scanned clusters: the caller may pass an explicit order
 *      (scan_order); when it does not, ascending cluster id is used.  The reference iterates
 *      an immutable Map built from a Set (SANN/SimClustersANNCandidateSource.scala:72-80).
 *  (2) order among exactly equal scores: (score desc by java.lang.Double.compare on -score,
 *      then tweet id ascending).  The reference does a stable sortBy(-score) over a
 *      mutable.HashMap iteration order (SANN/ApproximateCosineSimilarity.scala:105-127).
 *  (3) Time.now is an explicit input (now_ms).
 *  (4) math.log is restated as the fdlibm __ieee754_log algorithm (Sun, 1993), which is what
 *      java.lang.StrictMath.log is specified to be; java.lang.Math.log may differ from it by
 *      one ulp on some JVMs (it is only specified to within 1 ulp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * fdlibm e_log.c (public algorithm, restated): log(x) for finite positive and special x.
 * java.lang.StrictMath.log is defined as this algorithm.
 * ---------------------------------------------------------------------------------------- */
static inline int32_t hi_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (int32_t)(u >> 32); }
static inline uint32_t lo_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static inline double with_hi(double x, int32_t hi) {
  uint64_t u; memcpy(&u, &x, 8);
  u = (u & 0xffffffffull) | ((uint64_t)(uint32_t)hi << 32);
  memcpy(&x, &u, 8); return x;
}

double oracle_strict_log(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, /* 3fe62e42 fee00000 */
      ln2_lo = 1.90821492927058770002e-10,                 /* 3dea39ef 35793c76 */
      two54 = 1.80143985094819840000e+16,                  /* 43500000 00000000 */
      Lg1 = 6.666666666666735130e-01,                      /* 3FE55555 55555593 */
      Lg2 = 3.999999999940941908e-01,                      /* 3FD99999 9997FA04 */
      Lg3 = 2.857142874366239149e-01,                      /* 3FD24924 94229359 */
      Lg4 = 2.222219843214978396e-01,                      /* 3FCC71C5 1D8E78AF */
      Lg5 = 1.818357216161805012e-01,                      /* 3FC74664 96CB03DE */
      Lg6 = 1.531383769920937332e-01,                      /* 3FC39A09 D078C69F */
      Lg7 = 1.479819860511658591e-01;                      /* 3FC2F112 DF3E5244 */
  static const double zero = 0.0;
  double hfsq, f, s, z, R, w, t1, t2, dk;
  int32_t k, hx, i, j;
  uint32_t lx;

  hx = hi_word(x);
  lx = lo_word(x);
  k = 0;
  if (hx < 0x00100000) { /* x < 2**-1022 */
    if (((hx & 0x7fffffff) | lx) == 0) return -two54 / zero; /* log(+-0) = -inf */
    if (hx < 0) return (x - x) / zero;                        /* log(-#) = NaN */
    k -= 54;
    x *= two54; /* subnormal, scale up */
    hx = hi_word(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  i = (hx + 0x95f64) & 0x100000;
  x = with_hi(x, hx | (i ^ 0x3ff00000)); /* normalize x or x/2 */
  k += (i >> 20);
  f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2**-20 */
    if (f == zero) {
      if (k == 0) return zero;
      dk = (double)k;
      return dk * ln2_hi + dk * ln2_lo;
    }
    R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    dk = (double)k;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  s = f / (2.0 + f);
  dk = (double)k;
  z = s * s;
  i = hx - 0x6147a;
  w = z * z;
  j = 0x6b851 - hx;
  t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  R = t2 + t1;
  if (i > 0) {
    hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* ------------------------------------------------------------------------------------------
 * SimClustersEmbedding value type -- COMMON/SimClustersEmbedding.scala:25-448,450-509
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n;
  int32_t *cluster_ids;        /* desc by score, ties asc cluster id (:37-38, :456-463) */
  double *scores;
  int32_t *sorted_cluster_ids; /* asc by cluster id (:40-41) */
  double *sorted_scores;
} oracle_embedding;

typedef struct { int32_t id; double score; } id_score;

/* SimClustersEmbedding.order, COMMON/SimClustersEmbedding.scala:456-463:
 * `b._2 compare a._2` (scala Double compare == java.lang.Double.compare), then a._1 compare b._1 */
static int java_double_compare(double a, double b) {
  if (a < b) return -1;
  if (a > b) return 1;
  int64_t x, y;
  memcpy(&x, &a, 8); memcpy(&y, &b, 8);
  /* doubleToLongBits canonicalises NaN */
  if (a != a) x = 0x7ff8000000000000ll;
  if (b != b) y = 0x7ff8000000000000ll;
  return (x == y) ? 0 : (x < y ? -1 : 1);
}
static int cmp_desc_score_asc_id(const void *pa, const void *pb) {
  const id_score *a = pa, *b = pb;
  int c = java_double_compare(b->score, a->score);
  if (c) return c;
  return (a->id > b->id) - (a->id < b->id);
}
static int cmp_asc_id_stable(const void *pa, const void *pb) {
  /* sortBy(_._1) is stable; ids are assumed unique (:470) so the index tiebreak is moot */
  const id_score *a = pa, *b = pb;
  return (a->id > b->id) - (a->id < b->id);
}

void oracle_embedding_free(oracle_embedding *e) {
  if (!e) return;
  free(e->cluster_ids); free(e->scores); free(e->sorted_cluster_ids); free(e->sorted_scores);
  free(e);
}

/* buildDefaultSimClustersEmbedding, COMMON/SimClustersEmbedding.scala:490-509:
 * filter(_._2 > 0.0), sort by `order`, optional take(truncate) (truncate < 0 = None). */
oracle_embedding *oracle_embedding_build(int32_t n, const int32_t *ids, const double *scores,
                                         int32_t truncate) {
  id_score *tmp = malloc(sizeof(id_score) * (size_t)(n > 0 ? n : 1));
  int32_t m = 0;
  for (int32_t i = 0; i < n; i++)
    if (scores[i] > 0.0) { tmp[m].id = ids[i]; tmp[m].score = scores[i]; m++; }
  /* mergesort-equivalent: the comparator is a total order on (score,id) so qsort is fine */
  qsort(tmp, (size_t)m, sizeof(id_score), cmp_desc_score_asc_id);
  if (truncate >= 0 && m > truncate) m = truncate;
  oracle_embedding *e = calloc(1, sizeof(*e));
  e->n = m;
  size_t cap = (size_t)(m > 0 ? m : 1);
  e->cluster_ids = malloc(4 * cap); e->scores = malloc(8 * cap);
  e->sorted_cluster_ids = malloc(4 * cap); e->sorted_scores = malloc(8 * cap);
  for (int32_t i = 0; i < m; i++) { e->cluster_ids[i] = tmp[i].id; e->scores[i] = tmp[i].score; }
  qsort(tmp, (size_t)m, sizeof(id_score), cmp_asc_id_stable);
  for (int32_t i = 0; i < m; i++) { e->sorted_cluster_ids[i] = tmp[i].id; e->sorted_scores[i] = tmp[i].score; }
  free(tmp);
  return e;
}

/* CosineSimilarityUtil.sumOfSquaresArray -- left fold, COMMON/CosineSimilarityUtil.scala:15-17 */
static double sum_of_squares(const double *v, int32_t n) {
  double sum = 0.0;
  for (int32_t i = 0; i < n; i++) sum = sum + v[i] * v[i];
  return sum;
}
/* :29-31 */
double oracle_embedding_l2norm(const oracle_embedding *e) { return sqrt(sum_of_squares(e->sorted_scores, e->n)); }
/* :43-45  math.log(sumOfSquares + 1) */
double oracle_embedding_lognorm(const oracle_embedding *e) { return oracle_strict_log(sum_of_squares(e->sorted_scores, e->n) + 1); }
/* :57-59  math.pow(sumOfSquares, exponent); DefaultExponent 0.3, SimClustersEmbedding.scala:454 */
double oracle_embedding_expscalednorm(const oracle_embedding *e) { return pow(sum_of_squares(e->sorted_scores, e->n), 0.3); }

/* getOrElse, COMMON/SimClustersEmbedding.scala:115-125 (linear scan over the id-sorted arrays) */
static double emb_get_or_else(const oracle_embedding *e, int32_t cluster, double dflt) {
  for (int32_t i = 0; i < e->n; i++) {
    int32_t t = e->sorted_cluster_ids[i];
    if (cluster == t) return e->sorted_scores[i];
    if (t > cluster) return dflt;
  }
  return dflt;
}
/* contains, :140 (Set membership) */
static int emb_contains(const oracle_embedding *e, int32_t cluster) {
  for (int32_t i = 0; i < e->n; i++) if (e->sorted_cluster_ids[i] == cluster) return 1;
  return 0;
}

int32_t oracle_embedding_size(const oracle_embedding *e) { return e->n; }
void oracle_embedding_export(const oracle_embedding *e, int32_t *cluster_ids, double *scores,
                             int32_t *sorted_cluster_ids, double *sorted_scores) {
  memcpy(cluster_ids, e->cluster_ids, 4 * (size_t)e->n);
  memcpy(scores, e->scores, 8 * (size_t)e->n);
  memcpy(sorted_cluster_ids, e->sorted_cluster_ids, 4 * (size_t)e->n);
  memcpy(sorted_scores, e->sorted_scores, 8 * (size_t)e->n);
}

/* ------------------------------------------------------------------------------------------
 * Sorted-merge sparse dot -- COMMON/CosineSimilarityUtil.scala:224-250
 * ---------------------------------------------------------------------------------------- */
static double dot_sorted(const int32_t *c1, const double *s1, int32_t n1, const int32_t *c2,
                         const double *s2, int32_t n2) {
  int32_t i1 = 0, i2 = 0;
  double product = 0.0;
  while (i1 < n1 && i2 < n2) {
    if (c1[i1] == c2[i2]) { product += s1[i1] * s2[i2]; i1++; i2++; }
    else if (c1[i1] > c2[i2]) i2++;
    else i1++;
  }
  return product;
}
/* applyNormArray, :97-99 -- returns the input unchanged when norm == 0 */
static void apply_norm(const double *v, int32_t n, double norm, double *out) {
  for (int32_t i = 0; i < n; i++) out[i] = (norm == 0) ? v[i] : v[i] / norm;
}

/* Pair metrics bound to ScoringAlgorithm ids 1..7 --
 * src/scala/com/twitter/simclusters_v2/score/SimClustersEmbeddingPairScoreStore.scala:39-199,
 * src/thrift/com/twitter/simclusters_v2/score.thrift:14-22.
 * Inputs are raw (id,score) lists; both sides go through the A7 constructor first. */
double oracle_pair_score(int32_t algorithm, int32_t n1, const int32_t *ids1, const double *sc1,
                         int32_t n2, const int32_t *ids2, const double *sc2) {
  oracle_embedding *a = oracle_embedding_build(n1, ids1, sc1, -1);
  oracle_embedding *b = oracle_embedding_build(n2, ids2, sc2, -1);
  double r = NAN;
  double *na = malloc(8 * (size_t)(a->n + 1)), *nb = malloc(8 * (size_t)(b->n + 1));
  switch (algorithm) {
  case 1: /* dotProduct, SimClustersEmbedding.scala:194-200 */
    r = dot_sorted(a->sorted_cluster_ids, a->sorted_scores, a->n, b->sorted_cluster_ids, b->sorted_scores, b->n);
    break;
  case 2: /* cosineSimilarity = merge-dot of the PRE-NORMALISED arrays, :202-208, :71-72 */
    apply_norm(a->sorted_scores, a->n, oracle_embedding_l2norm(a), na);
    apply_norm(b->sorted_scores, b->n, oracle_embedding_l2norm(b), nb);
    r = dot_sorted(a->sorted_cluster_ids, na, a->n, b->sorted_cluster_ids, nb, b->n);
    break;
  case 6: /* logNormCosineSimilarity, :210-216, :74-75 */
    apply_norm(a->sorted_scores, a->n, oracle_embedding_lognorm(a), na);
    apply_norm(b->sorted_scores, b->n, oracle_embedding_lognorm(b), nb);
    r = dot_sorted(a->sorted_cluster_ids, na, a->n, b->sorted_cluster_ids, nb, b->n);
    break;
  case 7: /* expScaledCosineSimilarity, :218-224, :77-78 */
    apply_norm(a->sorted_scores, a->n, oracle_embedding_expscalednorm(a), na);
    apply_norm(b->sorted_scores, b->n, oracle_embedding_expscalednorm(b), nb);
    r = dot_sorted(a->sorted_cluster_ids, na, a->n, b->sorted_cluster_ids, nb, b->n);
    break;
  case 3: { /* jaccardSimilarity, :235-243 */
    if (a->n == 0 || b->n == 0) { r = 0.0; break; }
    int32_t i1 = 0, i2 = 0, inter = 0;
    while (i1 < a->n && i2 < b->n) {
      if (a->sorted_cluster_ids[i1] == b->sorted_cluster_ids[i2]) { inter++; i1++; i2++; }
      else if (a->sorted_cluster_ids[i1] > b->sorted_cluster_ids[i2]) i2++;
      else i1++;
    }
    int32_t uni = a->n + b->n - inter;
    r = (double)inter / uni;
    break;
  }
  case 4:   /* euclideanDistance, :301-309: fold over the UNION set; sum + d*d, d = |x - y| */
  case 5: { /* manhattanDistance, :315-321 */
    /* The fold order over a scala Set union is hash order (unspecified); ascending cluster
     * id is used here.  Floating-point sums of non-negative terms: order changes only ulps. */
    int32_t i1 = 0, i2 = 0;
    double sum = 0.0;
    while (i1 < a->n || i2 < b->n) {
      double x = 0.0, y = 0.0;
      if (i2 >= b->n || (i1 < a->n && a->sorted_cluster_ids[i1] < b->sorted_cluster_ids[i2])) x = a->sorted_scores[i1++];
      else if (i1 >= a->n || b->sorted_cluster_ids[i2] < a->sorted_cluster_ids[i1]) y = b->sorted_scores[i2++];
      else { x = a->sorted_scores[i1++]; y = b->sorted_scores[i2++]; }
      double d = fabs(x - y);
      sum = (algorithm == 4) ? sum + d * d : sum + d;
    }
    r = (algorithm == 4) ? sqrt(sum) : sum;
    break;
  }
  default: break; /* ScoreFacadeStore: unknown algorithm -> IllegalArgumentException; NaN here */
  }
  free(na); free(nb);
  oracle_embedding_free(a); oracle_embedding_free(b);
  return r;
}

/* ------------------------------------------------------------------------------------------
 * Snowflake window -- SANN/ApproximateCosineSimilarity.scala:65-72.  SnowflakeId itself is
 * un-vendored; the bit layout is pinned in-repo by
 * src/scala/com/twitter/simclusters_v2/scio/bq_generation/common/BQGenerationUtil.scala:150-153
 * (ms = 1288834974657 + (id >> 22)).
 * ---------------------------------------------------------------------------------------- */
#define SNOWFLAKE_EPOCH_MS 1288834974657ll
int64_t oracle_snowflake_first_id_for(int64_t time_ms) {
  return (int64_t)((uint64_t)(time_ms - SNOWFLAKE_EPOCH_MS) << 22);
}

typedef struct {
  int32_t max_num_results;               /* simClustersAnn.thrift:19 */
  double min_score;                      /* :20 */
  int32_t candidate_embedding_type;      /* :21 (carried, unused by the arithmetic) */
  int32_t max_top_tweets_per_cluster;    /* :22 */
  int32_t max_scan_clusters;             /* :23 */
  int32_t max_tweet_candidate_age_hours; /* :24 */
  int32_t min_tweet_candidate_age_hours; /* :25 */
  int32_t ann_algorithm;                 /* :26; 1 Dot, 2 Cosine, 3 LogCosine, 4 CosineNoSrcNorm (:32-37) */
} oracle_sann_config;

void oracle_age_window(const oracle_sann_config *cfg, int64_t now_ms, int64_t *earliest, int64_t *latest) {
  /* MaxTweetCandidateAgeUpperBound = 175200, :42,:66-72 */
  *earliest = (cfg->max_tweet_candidate_age_hours >= 175200)
                  ? 0
                  : oracle_snowflake_first_id_for(now_ms - (int64_t)cfg->max_tweet_candidate_age_hours * 3600000ll);
  *latest = oracle_snowflake_first_id_for(now_ms - (int64_t)cfg->min_tweet_candidate_age_hours * 3600000ll);
}

/* ------------------------------------------------------------------------------------------
 * Candidate map: open addressing keyed by tweet id, insertion-ordered entries.
 * "original" keeps two maps (scores, normalisation); the arithmetic is identical, so a single
 * entry array holds both accumulators: here only results matter.  (bench.py's cpu_baseline does NOT time
 * this function: oracle_baseline.c restates the two-map "original" and the one-map "optimized" with their
 * own probe counts, and a test checks that both give exactly this function's answers.)
 * ---------------------------------------------------------------------------------------- */
typedef struct { int64_t id; double dot; double nsq; } cand;
typedef struct { cand *e; int32_t n, cap_e; int32_t *slot; uint32_t mask; } candmap;

static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
static void candmap_init(candmap *m, int32_t expected) {
  uint32_t cap = 16384; /* InitialCandidateMapSize, :40 */
  while (cap < (uint32_t)expected * 2u) cap <<= 1;
  m->mask = cap - 1;
  m->slot = malloc(4 * (size_t)cap);
  memset(m->slot, 0xff, 4 * (size_t)cap);
  m->cap_e = expected > 16 ? expected : 16;
  m->e = malloc(sizeof(cand) * (size_t)m->cap_e);
  m->n = 0;
}
static cand *candmap_get_or_insert(candmap *m, int64_t id) {
  uint32_t h = (uint32_t)mix64((uint64_t)id) & m->mask;
  for (;;) {
    int32_t s = m->slot[h];
    if (s < 0) {
      m->slot[h] = m->n;
      cand *c = &m->e[m->n++];
      c->id = id; c->dot = 0.0; c->nsq = 0.0; /* getOrElse(tweetId, 0.0), :92-96 */
      return c;
    }
    if (m->e[s].id == id) return &m->e[s];
    h = (h + 1) & m->mask;
  }
}
static void candmap_free(candmap *m) { free(m->slot); free(m->e); }

typedef struct { int64_t id; double score; } scored;
static int cmp_scored(const void *pa, const void *pb) {
  const scored *a = pa, *b = pb;
  int c = java_double_compare(-a->score, -b->score); /* sortBy(-_._2), :126 */
  if (c) return c;
  return (a->id > b->id) - (a->id < b->id);           /* documented tie-break (2) */
}

static int cmp_i32(const void *a, const void *b) { int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }

/* Find the posting list of a cluster in the CSR (list_cluster_ids ascending). -1 = None. */
static int64_t find_list(int32_t n_lists, const int32_t *list_cluster_ids, int32_t cluster) {
  int64_t lo = 0, hi = (int64_t)n_lists - 1;
  while (lo <= hi) {
    int64_t mid = (lo + hi) >> 1;
    if (list_cluster_ids[mid] == cluster) return mid;
    if (list_cluster_ids[mid] < cluster) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

/*
 * ApproximateCosineSimilarity.apply, three variants:
 *   variant 0 "original"     SANN/ApproximateCosineSimilarity.scala:57-128
 *   variant 1 "optimized"    SANN/OptimizedApproximateCosineSimilarity.scala:37-111
 *   variant 2 "experimental" SANN/ExperimentalApproximateCosineSimilarity.scala:41-130
 *   variant 3 "legacy"       src/scala/com/twitter/simclusters_v2/candidate_source/
 *                            SimClustersANNCandidateSource.scala:107-181 (fetchCandidates): the age
 *                            window has no "175200 h = unbounded" rule (:113-114); the config's
 *                            (enablePartialNormalization, rankingAlgorithm) arrive folded into
 *                            ann_algorithm -- 1 = no normalisation (raw dot), 2 = partial
 *                            normalisation, 3 = partial normalisation with the "log" form, which
 *                            divides by l2norm, NOT logNorm (:167-169); no minScore filter and no
 *                            1000 cap before the sort (:177-180, reranking :195 takes maxNumResults).
 * preceded by the cluster selection of SimClustersANNCandidateSource.fetchCandidates
 * (SANN/SimClustersANNCandidateSource.scala:72-80) when scan_order == NULL.
 *
 * The posting lists are what the ReadableStore[ClusterId, Seq[(TweetId, Double)]] returns
 * (already filtered > 0, sorted desc, capped), passed as a CSR over ascending
 * list_cluster_ids.  A cluster absent from the CSR is a `None` value in clusterTweetsMap.
 *
 * has_source_tweet: 1 when sourceEmbeddingId.internalId is InternalId.TweetId (parseTweetId).
 * Returns the number of results written (<= min(maxNumResults, 1000)); *map_size receives
 * candidateScoresMap.size (the candidateScoresStat callback argument, :102).
 */
int32_t oracle_sann_query(int32_t variant, int32_t n_emb, const int32_t *emb_ids, const double *emb_scores,
                          int32_t has_source_tweet, int64_t source_tweet_id,
                          const oracle_sann_config *cfg, int64_t now_ms,
                          int32_t n_lists, const int32_t *list_cluster_ids, const int64_t *list_offsets,
                          const int64_t *tweet_ids, const double *scores,
                          int32_t n_scan_order, const int32_t *scan_order,
                          int64_t *out_ids, double *out_scores, int32_t *map_size) {
  oracle_embedding *emb = oracle_embedding_build(n_emb, emb_ids, emb_scores, -1);
  int64_t earliest, latest;
  oracle_age_window(cfg, now_ms, &earliest, &latest);
  if (variant == 3) /* legacy :113: always SnowflakeId.firstIdFor(now - maxTweetCandidateAge) */
    earliest = oracle_snowflake_first_id_for(now_ms - (int64_t)cfg->max_tweet_candidate_age_hours * 3600000ll);

  /* keys of clusterTweetsMap, in iteration order */
  int32_t n_scan;
  int32_t *scan;
  if (scan_order) {
    n_scan = n_scan_order;
    scan = malloc(4 * (size_t)(n_scan + 1));
    memcpy(scan, scan_order, 4 * (size_t)n_scan);
  } else {
    /* truncate(maxScanClusters).getClusterIds().toSet, COMMON/SimClustersEmbedding.scala:377-392:
     * the first `size` ids of the desc-by-score array; scala take(n<=0) is empty. */
    n_scan = cfg->max_scan_clusters < 0 ? 0 : (emb->n <= cfg->max_scan_clusters ? emb->n : cfg->max_scan_clusters);
    scan = malloc(4 * (size_t)(n_scan + 1));
    memcpy(scan, emb->cluster_ids, 4 * (size_t)n_scan);
    qsort(scan, (size_t)n_scan, 4, cmp_i32); /* documented order (1): ascending cluster id */
  }

  int64_t expected = 0;
  for (int32_t c = 0; c < n_scan; c++) {
    int64_t li = find_list(n_lists, list_cluster_ids, scan[c]);
    if (li >= 0) expected += list_offsets[li + 1] - list_offsets[li];
  }
  candmap map;
  candmap_init(&map, (int32_t)(expected < (1 << 30) ? expected : (1 << 30)));

  /* OptimizedApproximateCosineSimilarity.scala:56 / Experimental :59: getOrElse(0L) */
  int64_t src_excl = has_source_tweet ? source_tweet_id : 0;

  for (int32_t c = 0; c < n_scan; c++) {
    int32_t cluster = scan[c];
    int64_t li = find_list(n_lists, list_cluster_ids, cluster);
    if (li < 0) continue; /* case _ => () for None */
    double w;
    if (variant == 2) {
      w = emb_get_or_else(emb, cluster, 0.0); /* Experimental :62-63, no contains guard */
    } else {
      if (!emb_contains(emb, cluster)) continue; /* :84 / Optimized :59 */
      w = emb_get_or_else(emb, cluster, 0.0);    /* :85 */
    }
    int64_t len = list_offsets[li + 1] - list_offsets[li];
    int64_t lim = len < cfg->max_top_tweets_per_cluster ? len : cfg->max_top_tweets_per_cluster; /* :87 */
    const int64_t *tid = tweet_ids + list_offsets[li];
    const double *sc = scores + list_offsets[li];
    for (int64_t i = 0; i < lim; i++) {
      int64_t t = tid[i];
      double s = sc[i];
      int excl = (variant == 0 || variant == 3) ? (has_source_tweet && t == source_tweet_id) /* :90; legacy :139 */
                                : (t == src_excl);                            /* Optimized :67 */
      if (!excl && t >= earliest && t <= latest) { /* :90-91 */
        cand *e = candmap_get_or_insert(&map, t);
        e->dot = e->dot + s * w; /* :92-94 */
        e->nsq = e->nsq + s * s; /* :95-96 */
      }
    }
  }
  *map_size = map.n; /* :102 */

  double l2 = oracle_embedding_l2norm(emb), ln = oracle_embedding_lognorm(emb);
  scored *res = malloc(sizeof(scored) * (size_t)(map.n + 1));
  int32_t nres = 0;
  for (int32_t i = 0; i < map.n; i++) {
    double score = map.e[i].dot, nsq = map.e[i].nsq, p;
    if (variant == 3) { /* legacy :160-174 */
      switch (cfg->ann_algorithm) {
      case 3: p = score / l2 / oracle_strict_log(1 + nsq); break;
      case 2: p = score / l2 / sqrt(nsq); break;
      case 1: p = score; break;
      default: p = NAN; break;
      }
      res[nres].id = map.e[i].id; res[nres].score = p; nres++;
      continue;
    }
    switch (cfg->ann_algorithm) { /* :111-119 */
    case 3: p = score / ln / oracle_strict_log(1 + nsq); break;
    case 2: p = score / l2 / sqrt(nsq); break;
    case 4: p = score / sqrt(nsq); break;
    case 1: p = score; break;
    default: p = NAN; break; /* scala MatchError -> controller rescues to an empty response */
    }
    if (p >= cfg->min_score) { res[nres].id = map.e[i].id; res[nres].score = p; nres++; } /* :125 */
  }
  qsort(res, (size_t)nres, sizeof(scored), cmp_scored);
  int32_t k = cfg->max_num_results < 1000 ? cfg->max_num_results : 1000; /* :41,:127 */
  if (variant == 3) k = cfg->max_num_results; /* legacy: plain take(maxNumResults); the caller sizes the outputs */
  if (k < 0) k = 0;
  if (nres > k) nres = k;
  for (int32_t i = 0; i < nres; i++) { out_ids[i] = res[i].id; out_scores[i] = res[i].score; }
  free(res); candmap_free(&map); free(scan); oracle_embedding_free(emb);
  return nres;
}

/* ------------------------------------------------------------------------------------------
 * Posting-list materialisation: what the cluster -> top tweets store hands the operator.
 *   decay to now     summingbird/stores/TopKTweetsForClusterReadableStore.scala:51-71 -> summingbird/common/EntityUtil.scala:10-28
 *                    -> ThriftDecayedValueMonoid.decayToTimestamp (summingbird/common/ThriftDecayedValueMonoid.scala:33-38)
 *                    = algebird DecayedValueMonoid(eps 0.0).plus(v, DecayedValue(0.0, now * ln2 / halfLife))
 *                    (com.twitter.algebird: not vendored, no version pinned in the tree; restated from its published
 *                    DecayedValue: scaledPlus(newer, older) = newer.value + exp(older.t - newer.t) * older.value)
 *   filter / sort    TopKTweetsForClusterReadableStore.scala:211-229: value > 0.0, .toSeq.sortBy(-_._2)
 *                    (ties: Map iteration order under a stable sort -- fixed here as tweet id ascending)
 *   take             :258-259, :281-282  .take(maxResults)
 * math.exp is restated as fdlibm's __ieee754_exp (= StrictMath.exp); HotSpot's Math.exp may differ by 1 ulp.
 * ---------------------------------------------------------------------------------------- */
double oracle_strict_exp(double x) {
  static const double halF[2] = {0.5, -0.5}, huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302,
                      o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
                      ln2HI[2] = {6.93147180369123816490e-01, -6.93147180369123816490e-01},
                      ln2LO[2] = {1.90821492927058770002e-10, -1.90821492927058770002e-10},
                      invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                      P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                      P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  double y, hi = 0.0, lo = 0.0, c, t;
  int32_t k = 0, xsb;
  uint32_t hx = (uint32_t)hi_word(x);
  xsb = (int32_t)((hx >> 31) & 1);
  hx &= 0x7fffffff;
  if (hx >= 0x40862E42) {
    if (hx >= 0x7ff00000) {
      if (((hx & 0xfffff) | lo_word(x)) != 0) return x + x;
      return (xsb == 0) ? x : 0.0;
    }
    if (x > o_threshold) return huge * huge;
    if (x < u_threshold) return twom1000 * twom1000;
  }
  if (hx > 0x3fd62e42) {
    if (hx < 0x3FF0A2B2) {
      hi = x - ln2HI[xsb];
      lo = ln2LO[xsb];
      k = 1 - xsb - xsb;
    } else {
      k = (int32_t)(invln2 * x + halF[xsb]);
      t = k;
      hi = x - t * ln2HI[0];
      lo = t * ln2LO[0];
    }
    x = hi - lo;
  } else if (hx < 0x3e300000) {
    return 1.0 + x;
  }
  t = x * x;
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) return with_hi(y, hi_word(y) + (k << 20));
  y = with_hi(y, hi_word(y) + ((k + 1000) << 20));
  return y * twom1000;
}

double oracle_decay_to_timestamp(double value, double scaled_time, double now_scaled) {
  double nv = scaled_time < now_scaled ? 0.0 + oracle_strict_exp(scaled_time - now_scaled) * value
                                       : value + oracle_strict_exp(now_scaled - scaled_time) * 0.0;
  return fabs(nv) > 0.0 ? nv : 0.0;
}

/* One cluster's list as the store returns it.  scaled_times may be NULL (a store that does not decay: the
 * Manhattan read-only path, :236-260).  Returns the number of (tweet, score) pairs written (<= max_results). */
int32_t oracle_store_list(int32_t n, const int64_t *tweet_ids, const double *values, const double *scaled_times,
                          double now_scaled, int32_t max_results, int64_t *out_ids, double *out_scores) {
  scored *v = malloc(sizeof(scored) * (size_t)(n + 1));
  int32_t m = 0;
  for (int32_t i = 0; i < n; i++) {
    double x = scaled_times ? oracle_decay_to_timestamp(values[i], scaled_times[i], now_scaled) : values[i];
    if (x > 0.0) { v[m].id = tweet_ids[i]; v[m].score = x; m++; }
  }
  qsort(v, (size_t)m, sizeof(scored), cmp_scored); /* sortBy(-score), ties tweet id ascending */
  if (max_results < 0) max_results = 0;
  if (m > max_results) m = max_results;
  for (int32_t i = 0; i < m; i++) { out_ids[i] = v[i].id; out_scores[i] = v[i].score; }
  free(v);
  return m;
}

/* Exact cosine of the source embedding against a tweet's FULL embedding -- what SANN
 * approximates (simclusters-ann/README.md:18-46); used for the quality recall@k only. */
double oracle_full_cosine(int32_t n1, const int32_t *ids1, const double *sc1, int32_t n2,
                          const int32_t *ids2, const double *sc2) {
  return oracle_pair_score(2, n1, ids1, sc1, n2, ids2, sc2);
}
