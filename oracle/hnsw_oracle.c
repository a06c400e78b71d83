/*
 * hnsw_oracle.c -- CPU restatement of the reference's HNSW query walk.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows (paths relative to /root/reference/ann/src/main/java/com/twitter/ann/hnsw/):
 *   HnswIndex.java:538-553   searchKnn: descend to layer 0, beam search with max(ef, k), dequeueAll, reverse, cut
 *   HnswIndex.java:447-475   bestEntryPointUntilLayer
 *   HnswIndex.java:571-623   searchLayerForCandidates (isUpdate = false)
 *   DistancedItemQueue.java:37-43,101-108   queues: java.util.PriorityQueue with Float.compare on distance
 * java.util.PriorityQueue itself is JDK code, absent from the tree: offer = siftUp from the end, poll = move the
 * last element to the root and siftDown choosing the smaller child (left when equal) -- the published OpenJDK
 * algorithm.  Ties in distance therefore resolve as on the JVM.
 *
 * PARITY UNPINNED in the float distances: the reference computes them with the un-vendored
 * com.twitter.ml.api.embedding.EmbeddingMath and ships no fixture.  The arithmetic here is the one
 * include/hnsw_ann.h documents (fp16-rounded operands, fp32, 8 strided partial sums of 8-element chunks, pairwise
 * tree); the walk is exact given the distances.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float d; int64_t item; } qitem;
typedef struct { qitem *a; int n, cap, min; } jpq;

static int jfloat_compare(float x, float y) { /* java.lang.Float.compare */
  if (x < y) return -1;
  if (x > y) return 1;
  int32_t bx, by;
  if (x != x) bx = 0x7fc00000; else memcpy(&bx, &x, 4);
  if (y != y) by = 0x7fc00000; else memcpy(&by, &y, 4);
  return bx == by ? 0 : (bx < by ? -1 : 1);
}
static int jcmp(const jpq *q, qitem x, qitem y) { return q->min ? jfloat_compare(x.d, y.d) : jfloat_compare(y.d, x.d); }
static void jpq_init(jpq *q, int min) { q->n = 0; q->cap = 64; q->min = min; q->a = malloc(sizeof(qitem) * 64); }
static void jpq_offer(jpq *q, qitem x) {
  if (q->n == q->cap) { q->cap *= 2; q->a = realloc(q->a, sizeof(qitem) * (size_t)q->cap); }
  int k = q->n++;
  while (k > 0) {
    int parent = (k - 1) >> 1;
    if (jcmp(q, x, q->a[parent]) >= 0) break;
    q->a[k] = q->a[parent];
    k = parent;
  }
  q->a[k] = x;
}
static qitem jpq_poll(jpq *q) {
  qitem result = q->a[0];
  int s = --q->n;
  if (s > 0) {
    qitem x = q->a[s];
    int k = 0, half = s >> 1;
    while (k < half) {
      int child = 2 * k + 1, right = child + 1;
      qitem c = q->a[child];
      if (right < s && jcmp(q, c, q->a[right]) > 0) c = q->a[child = right];
      if (jcmp(q, x, c) <= 0) break;
      q->a[k] = c;
      k = child;
    }
    q->a[k] = x;
  }
  return result;
}

typedef struct {
  int metric, d;
  const float *x, *q;
} dist_ctx;

static float hdistance(const dist_ctx *c, int64_t item) {
  const float *y = c->x + (size_t)item * c->d;
  const int dpad = (c->d + 63) / 64 * 64;
  float p[8];
  for (int j = 0; j < 8; j++) {
    float acc = 0.0f;
    for (int ch = j; ch < dpad / 8; ch += 8)
      for (int e = 0; e < 8; e++) {
        int k = ch * 8 + e;
        float qv = k < c->d ? c->q[k] : 0.0f, yv = k < c->d ? y[k] : 0.0f;
        if (c->metric == 0) { float t = qv - yv; acc = acc + t * t; }
        else acc = acc + qv * yv;
      }
    p[j] = acc;
  }
  float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  return c->metric == 0 ? sqrtf(s) : 1.0f - s;
}

/* graph: entry e = HnswNode(level[e], item[e]) -> neighbours[offsets[e] .. offsets[e+1]) */
int32_t oracle_hnsw_search(int32_t metric, int64_t n, int32_t d, const float *x, const float *query, int64_t entry_point,
                           int32_t max_level, int64_t n_entries, const int32_t *entry_level, const int64_t *entry_item,
                           const int64_t *entry_offsets, const int64_t *entry_neighbours, int32_t k, int32_t ef,
                           int64_t *out_items, float *out_dist, int64_t *out_distance_evals) {
  if (entry_point < 0) return 0; /* :550-552 */
  dist_ctx ctx = {metric, d, x, query};
  int64_t evals = 0;
  /* lookup[level][item] -> entry index, -1 = getConnectionListForRead's empty list */
  int64_t **lookup = malloc(sizeof(int64_t *) * (size_t)(max_level + 1));
  for (int l = 0; l <= max_level; l++) {
    lookup[l] = malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) lookup[l][i] = -1;
  }
  for (int64_t e = 0; e < n_entries; e++) lookup[entry_level[e]][entry_item[e]] = e;

  /* bestEntryPointUntilLayer(entry, query, maxLevel, 0) */
  int64_t cur = entry_point;
  if (0 < max_level) {
    float cur_dist = hdistance(&ctx, cur); evals++;
    for (int level = max_level; level > 0; level--) {
      int changed = 1;
      while (changed) {
        changed = 0;
        int64_t e = lookup[level][cur];
        if (e < 0) continue;
        int64_t b = entry_offsets[e], en = entry_offsets[e + 1];
        for (int64_t j = b; j < en; j++) {
          int64_t nn = entry_neighbours[j];
          float t = hdistance(&ctx, nn); evals++;
          if (t < cur_dist) { cur_dist = t; cur = nn; changed = 1; }
        }
      }
    }
  }

  /* searchLayerForCandidates(query, cur, max(ef, k), 0) */
  int beam = ef > k ? ef : k;
  jpq cq, wq;
  jpq_init(&cq, 1);
  jpq_init(&wq, 0);
  qitem first = {hdistance(&ctx, cur), cur}; evals++;
  jpq_offer(&cq, first);
  jpq_offer(&wq, first); /* cQueue.reverse() of a one-element queue */
  uint8_t *visited = calloc((size_t)(n > 0 ? n : 1), 1);
  visited[cur] = 1;
  float lower = wq.a[0].d;
  while (cq.n > 0) {
    qitem cand = cq.a[0];
    if (cand.d > lower) break;
    jpq_poll(&cq);
    int64_t e = lookup[0][cand.item];
    if (e < 0) continue;
    for (int64_t j = entry_offsets[e]; j < entry_offsets[e + 1]; j++) {
      int64_t nn = entry_neighbours[j];
      if (visited[nn]) continue;
      visited[nn] = 1;
      float dist = hdistance(&ctx, nn); evals++;
      if (wq.n < beam || dist < wq.a[0].d) {
        qitem it = {dist, nn};
        jpq_offer(&cq, it);
        jpq_offer(&wq, it);
        if (wq.n > beam) jpq_poll(&wq);
        lower = wq.a[0].d;
      }
    }
  }
  /* dequeueAll + Collections.reverse + subList(0, k) */
  int found = wq.n, m = found < k ? found : k;
  for (int pos = found - 1; pos >= 0; pos--) {
    qitem it = jpq_poll(&wq);
    if (pos < m) { out_items[pos] = it.item; out_dist[pos] = it.d; }
  }
  if (out_distance_evals) *out_distance_evals = evals;
  free(visited); free(cq.a); free(wq.a);
  for (int l = 0; l <= max_level; l++) free(lookup[l]);
  free(lookup);
  return m;
}
