/*
 * hnsw_oracle.c -- CPU restatement of the reference's HNSW query walk and of its index construction.  TEST INFRASTRUCTURE ONLY.
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

/* ---------------------------------------------------------------------------------------------------------------------
 * Index construction: HnswIndex.insert restated (single writer: the locks of the Java code order nothing then).
 *   HnswIndex.java:150-199   insert: duplicate check, level, entry point / max level, wire, entry-point update
 *   HnswIndex.java:137-148   wireConnectionForAllLayers
 *   HnswIndex.java:571-623   searchLayerForCandidates (isUpdate = false), with distFnIndex (item to item)
 *   HnswIndex.java:384-440   mutuallyConnectNewElement
 *   HnswIndex.java:479-526   selectNearestNeighboursByHeuristic
 *   DistancedItemQueue.java:111-118,176-189   toListWithItem / reverse: both walk the PriorityQueue's ARRAY order
 * The level of every item is an input (the reference draws (int)(-ln U / ln maxM) from a thread-local Random, :369-371).
 * Item-to-item distances use the arithmetic of hdistance() with the first item's stored row as the query.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int64_t n;
  int metric, d, max_m, max_m0, efc, n_levels, cap;
  const float *x;
  int32_t **cnt;   /* [level][item]: list length, -1 = the graph has no HnswNode(level, item) */
  int64_t **nb;    /* [level][item * cap + i] */
  int64_t entry;   /* -1 = Optional.empty() */
  int max_level;   /* HnswMeta starts at (-1, empty), :97 */
} hbuild;

static float item_distance(const hbuild *g, int64_t a, int64_t b) { /* distFnIndex.distance(a, b) */
  dist_ctx c = {g->metric, g->d, g->x, g->x + (size_t)a * g->d};
  return hdistance(&c, b);
}
static int list_of(const hbuild *g, int level, int64_t item, const int64_t **out) { /* getConnectionListForRead / getOrDefault */
  if (level < 0 || level >= g->n_levels || g->cnt[level][item] < 0) { *out = NULL; return 0; }
  *out = g->nb[level] + (size_t)item * g->cap;
  return g->cnt[level][item];
}
static void put_list(hbuild *g, int level, int64_t item, const int64_t *list, int n) { /* graph.put(HnswNode.from(level, item), ...) */
  memcpy(g->nb[level] + (size_t)item * g->cap, list, sizeof(int64_t) * (size_t)n);
  g->cnt[level][item] = n;
}

/* candidates: a MAX queue whose origin is `base`; returns the number of neighbours written to out (<= max_conn) */
static int select_by_heuristic(const hbuild *g, const jpq *cand, int64_t base, int max_conn, int64_t *out) {
  int m = 0;
  if (cand->n <= max_conn) { /* :488-491 toListWithItem (array order), remove the first occurrence of the base */
    int removed = 0;
    for (int i = 0; i < cand->n; i++) {
      if (!removed && cand->a[i].item == base) { removed = 1; continue; }
      out[m++] = cand->a[i].item;
    }
    return m;
  }
  jpq minq; /* :495 candidates.reverse(): re-offer in array order under the reversed comparator */
  jpq_init(&minq, 1);
  for (int i = 0; i < cand->n; i++) jpq_offer(&minq, cand->a[i]);
  while (minq.n > 0) {
    if (m >= max_conn) break;
    qitem c = jpq_poll(&minq);
    if (c.item == base) continue; /* :505-507 */
    int include = 1;
    for (int i = 0; i < m; i++) {
      float dist = item_distance(g, out[i], c.item); /* distFnIndex.distance(e, candidate) */
      if (dist < c.d) { include = 0; break; }
    }
    if (include) out[m++] = c.item;
  }
  free(minq.a);
  return m;
}

static void search_layer(const hbuild *g, int64_t item, int64_t entry, int ef, int level, uint8_t *visited, jpq *wq) {
  jpq cq;
  jpq_init(&cq, 1);
  jpq_init(wq, 0);
  qitem first = {item_distance(g, item, entry), entry};
  jpq_offer(&cq, first);
  jpq_offer(wq, first);
  memset(visited, 0, (size_t)g->n);
  visited[entry] = 1;
  float lower = wq->a[0].d;
  while (cq.n > 0) {
    qitem cand = cq.a[0];
    if (cand.d > lower) break;
    jpq_poll(&cq);
    const int64_t *list;
    int ln = list_of(g, level, cand.item, &list);
    for (int j = 0; j < ln; j++) {
      int64_t nn = list[j];
      if (visited[nn]) continue;
      visited[nn] = 1;
      float dist = item_distance(g, item, nn);
      if (wq->n < ef || dist < wq->a[0].d) {
        qitem it = {dist, nn};
        jpq_offer(&cq, it);
        jpq_offer(wq, it);
        if (wq->n > ef) jpq_poll(wq);
        lower = wq->a[0].d;
      }
    }
  }
  free(cq.a);
}

static int64_t mutually_connect(hbuild *g, int64_t item, const jpq *cand, int level) {
  int64_t *neigh = malloc(sizeof(int64_t) * (size_t)(g->cap + 1));
  int64_t *upd = malloc(sizeof(int64_t) * (size_t)(g->cap + 1));
  int nn = select_by_heuristic(g, cand, item, g->max_m, neigh); /* :392 maxM on every level */
  put_list(g, level, item, neigh, nn);                            /* :393 */
  const int M = level == 0 ? g->max_m0 : g->max_m;
  for (int i = 0; i < nn; i++) {
    int64_t other = neigh[i];
    if (other == item) continue;
    const int64_t *conn;
    int cn = list_of(g, level, other, &conn);
    if (cn < M) { /* :414-417 append */
      memcpy(upd, conn, sizeof(int64_t) * (size_t)cn);
      upd[cn] = item;
      put_list(g, level, other, upd, cn + 1);
    } else { /* :419-427 max queue around `other` holding its connections, plus the new item */
      jpq q;
      jpq_init(&q, 0);
      for (int j = 0; j < cn; j++) { qitem it = {item_distance(g, other, conn[j]), conn[j]}; jpq_offer(&q, it); }
      qitem it = {item_distance(g, other, item), item};
      jpq_offer(&q, it);
      int un = select_by_heuristic(g, &q, other, M, upd);
      put_list(g, level, other, upd, un);
      free(q.a);
    }
  }
  int64_t first = neigh[0]; /* :439 neighbours.get(0) */
  free(neigh);
  free(upd);
  return first;
}

/* Inserts items 0 .. n-1 in order.  Output: the graph's entries sorted by (level, item), as oracle_hnsw_search reads
 * them; returns the number of entries (or -1 if the output arrays are too small). */
int64_t oracle_hnsw_build(int32_t metric, int64_t n, int32_t d, const float *x, const int32_t *levels, int32_t max_m,
                          int32_t ef_construction, int64_t cap_entries, int64_t cap_neighbours, int32_t *entry_level,
                          int64_t *entry_item, int64_t *entry_offsets, int64_t *entry_neighbours, int64_t *entry_point,
                          int32_t *max_level) {
  hbuild g;
  g.n = n; g.metric = metric; g.d = d; g.max_m = max_m; g.max_m0 = 2 * max_m; g.efc = ef_construction; g.x = x;
  g.cap = 2 * max_m + 1;
  g.entry = -1; g.max_level = -1;
  int top = 0;
  for (int64_t i = 0; i < n; i++) if (levels[i] > top) top = levels[i];
  g.n_levels = top + 1;
  g.cnt = malloc(sizeof(int32_t *) * (size_t)g.n_levels);
  g.nb = malloc(sizeof(int64_t *) * (size_t)g.n_levels);
  for (int l = 0; l < g.n_levels; l++) {
    g.cnt[l] = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) g.cnt[l][i] = -1;
    g.nb[l] = malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1) * (size_t)g.cap);
  }
  uint8_t *visited = malloc((size_t)(n > 0 ? n : 1));
  for (int64_t item = 0; item < n; item++) {
    const int cur_level = levels[item];      /* :165 */
    const int64_t entry = g.entry;           /* :166 */
    const int max_copy = g.max_level;        /* :170 */
    if (entry >= 0) {                        /* :184-186 wireConnectionForAllLayers(entry, item, curLevel, maxLevelCopy) */
      int64_t cur = entry;
      if (cur_level < max_copy) {            /* bestEntryPointUntilLayer(cur, item, maxLayer, itemLevel), :447-475 */
        float cur_dist = item_distance(&g, item, cur);
        for (int level = max_copy; level > cur_level; level--) {
          int changed = 1;
          while (changed) {
            changed = 0;
            const int64_t *list;
            int ln = list_of(&g, level, cur, &list);
            for (int j = 0; j < ln; j++) {
              float t = item_distance(&g, item, list[j]);
              if (t < cur_dist) { cur_dist = t; cur = list[j]; changed = 1; }
            }
          }
        }
      }
      for (int level = cur_level < max_copy ? cur_level : max_copy; level >= 0; level--) {
        jpq wq;
        search_layer(&g, item, cur, g.efc, level, visited, &wq);
        cur = mutually_connect(&g, item, &wq, level);
        free(wq.a);
      }
    }
    if (cur_level > max_copy) { g.max_level = cur_level; g.entry = item; } /* :188-193 */
  }
  int64_t ne = 0, nnb = 0;
  int fits = 1;
  for (int l = 0; l < g.n_levels && fits; l++)
    for (int64_t i = 0; i < n; i++) {
      if (g.cnt[l][i] < 0) continue;
      if (ne >= cap_entries || nnb + g.cnt[l][i] > cap_neighbours) { fits = 0; break; }
      entry_level[ne] = l;
      entry_item[ne] = i;
      entry_offsets[ne] = nnb;
      memcpy(entry_neighbours + nnb, g.nb[l] + (size_t)i * g.cap, sizeof(int64_t) * (size_t)g.cnt[l][i]);
      nnb += g.cnt[l][i];
      ne++;
    }
  if (fits) entry_offsets[ne] = nnb;
  *entry_point = g.entry;
  *max_level = g.max_level;
  for (int l = 0; l < g.n_levels; l++) { free(g.cnt[l]); free(g.nb[l]); }
  free(g.cnt); free(g.nb); free(visited);
  return fits ? ne : -1;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * The device builder's batched insertion, restated on the CPU (include/hnsw_ann.h, hnsw_index_build_insert_gpu): the
 * reference's multi-writer mode (HnswIndex.java:150-200,376-380) with the interleaving fixed.
 *   order   items by (level descending, position ascending); order[0] is the entry point, maxLevel = its level
 *   rounds  the next min(batch, max(1, linked / 8)) items are wired against one snapshot of the graph
 *   A       wireConnectionForAllLayers (:137-148) per item with the functions above, except that the item's back links are
 *           only recorded; the walk's candidate queue holds ccap entries -- when full, entries beyond the current bound
 *           (never expanded, :589-591) are dropped and the heap is rebuilt by re-offering the survivors in array order
 *   B       per (layer, node), additions in order-index order: appended while there is room (:414-417), else ONE
 *           re-selection by the heuristic (:495-523) over the first link_cap of old list ++ additions, ascending by
 *           (Float.compare distance, position)
 * ------------------------------------------------------------------------------------------------------------------ */
static int64_t g_batched_prunes = 0; /* of the last oracle_hnsw_build_batched */
int64_t oracle_hnsw_batched_prunes(void) { return g_batched_prunes; }
static void search_layer_bounded(const hbuild *g, int64_t item, int64_t entry, int ef, int level, uint8_t *visited, jpq *wq, int ccap) {
  jpq cq;
  jpq_init(&cq, 1);
  jpq_init(wq, 0);
  qitem first = {item_distance(g, item, entry), entry};
  jpq_offer(&cq, first);
  jpq_offer(wq, first);
  memset(visited, 0, (size_t)g->n);
  visited[entry] = 1;
  float lower = wq->a[0].d;
  while (cq.n > 0) {
    qitem cand = cq.a[0];
    if (cand.d > lower) break;
    jpq_poll(&cq);
    const int64_t *list;
    int ln = list_of(g, level, cand.item, &list);
    for (int j = 0; j < ln; j++) {
      int64_t nn = list[j];
      if (visited[nn]) continue;
      visited[nn] = 1;
      float dist = item_distance(g, item, nn);
      if (wq->n < ef || dist < wq->a[0].d) {
        qitem it = {dist, nn};
        if (cq.n >= ccap) { /* prune */
          g_batched_prunes++;
          int had = cq.n;
          qitem *old = malloc(sizeof(qitem) * (size_t)had);
          memcpy(old, cq.a, sizeof(qitem) * (size_t)had);
          cq.n = 0;
          for (int s = 0; s < had; s++) if (!(old[s].d > lower)) jpq_offer(&cq, old[s]);
          free(old);
        }
        if (cq.n < ccap) jpq_offer(&cq, it);
        jpq_offer(wq, it);
        if (wq->n > ef) jpq_poll(wq);
        lower = wq->a[0].d;
      }
    }
  }
  free(cq.a);
}

typedef struct { int level; int64_t target; int64_t t; } backlink;
static int backlink_cmp(const void *pa, const void *pb) {
  const backlink *a = pa, *b = pb;
  if (a->level != b->level) return a->level < b->level ? -1 : 1;
  if (a->target != b->target) return a->target < b->target ? -1 : 1;
  return a->t < b->t ? -1 : (a->t > b->t ? 1 : 0);
}

int64_t oracle_hnsw_build_batched(int32_t metric, int64_t n, int32_t d, const float *x, const int32_t *levels, int32_t max_m,
                                  int32_t ef_construction, int32_t batch, int32_t ccap, int32_t link_cap, int64_t cap_entries,
                                  int64_t cap_neighbours, int32_t *entry_level, int64_t *entry_item, int64_t *entry_offsets,
                                  int64_t *entry_neighbours, int64_t *entry_point, int32_t *max_level) {
  hbuild g;
  g_batched_prunes = 0;
  g.n = n; g.metric = metric; g.d = d; g.max_m = max_m; g.max_m0 = 2 * max_m; g.efc = ef_construction; g.x = x;
  g.cap = 2 * max_m + 1;
  g.entry = -1; g.max_level = -1;
  int top = 0;
  for (int64_t i = 0; i < n; i++) if (levels[i] > top) top = levels[i];
  g.n_levels = top + 1;
  g.cnt = malloc(sizeof(int32_t *) * (size_t)g.n_levels);
  g.nb = malloc(sizeof(int64_t *) * (size_t)g.n_levels);
  for (int l = 0; l < g.n_levels; l++) {
    g.cnt[l] = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) g.cnt[l][i] = -1;
    g.nb[l] = malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1) * (size_t)g.cap);
  }
  uint8_t *visited = malloc((size_t)(n > 0 ? n : 1));
  /* order: level descending, position ascending (a counting sort by level) */
  int64_t *order = malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  {
    int64_t at = 0;
    for (int l = top; l >= 0; l--)
      for (int64_t i = 0; i < n; i++) if (levels[i] == l) order[at++] = i;
  }
  if (n > 0) { g.entry = order[0]; g.max_level = levels[order[0]]; }
  backlink *links = malloc(sizeof(backlink) * (size_t)(batch > 0 ? batch : 1) * (size_t)(top + 1) * (size_t)max_m);
  int64_t *neigh = malloc(sizeof(int64_t) * (size_t)(g.cap + 1));
  int64_t *tid = malloc(sizeof(int64_t) * (size_t)link_cap), *cid = malloc(sizeof(int64_t) * (size_t)link_cap);
  float *td = malloc(sizeof(float) * (size_t)link_cap), *cd = malloc(sizeof(float) * (size_t)link_cap);
  int64_t at = 1, linked = 1;
  while (at < n) {
    int64_t m = linked / 8;
    if (m < 1) m = 1;
    if (m > batch) m = batch;
    if (m > n - at) m = n - at;
    int64_t nl = 0;
    /* ---- A: every item of the round against the snapshot (the new items' own lists are not reachable from it) ---- */
    for (int64_t t = 0; t < m; t++) {
      const int64_t item = order[at + t];
      const int cur_level = levels[item];
      int64_t cur = g.entry;
      if (cur_level < g.max_level) {
        float cur_dist = item_distance(&g, item, cur);
        for (int level = g.max_level; level > cur_level; level--) {
          int changed = 1;
          while (changed) {
            changed = 0;
            const int64_t *list;
            int ln = list_of(&g, level, cur, &list);
            for (int j = 0; j < ln; j++) {
              float dd = item_distance(&g, item, list[j]);
              if (dd < cur_dist) { cur_dist = dd; cur = list[j]; changed = 1; }
            }
          }
        }
      }
      for (int level = cur_level < g.max_level ? cur_level : g.max_level; level >= 0; level--) {
        jpq wq;
        search_layer_bounded(&g, item, cur, g.efc, level, visited, &wq, ccap);
        int nn = select_by_heuristic(&g, &wq, item, g.max_m, neigh);
        put_list(&g, level, item, neigh, nn);
        for (int e = 0; e < nn; e++) { links[nl].level = level; links[nl].target = neigh[e]; links[nl].t = t; nl++; }
        cur = neigh[0];
        free(wq.a);
      }
    }
    /* (the new items' lists must not be visible to the walks of the same round: they are not, since no old list points
     * to a new item before B runs and the entry point is old) */
    /* ---- B ---- */
    qsort(links, (size_t)nl, sizeof(backlink), backlink_cmp);
    for (int64_t i0 = 0; i0 < nl;) {
      int64_t i1 = i0;
      while (i1 < nl && links[i1].level == links[i0].level && links[i1].target == links[i0].target) i1++;
      const int level = links[i0].level;
      const int64_t base = links[i0].target;
      const int M = level == 0 ? g.max_m0 : g.max_m;
      const int64_t *conn;
      int old_n = list_of(&g, level, base, &conn);
      const int add_n = (int)(i1 - i0);
      if (old_n + add_n <= M) {
        int64_t *upd = malloc(sizeof(int64_t) * (size_t)(M + 1));
        if (old_n > 0) memcpy(upd, conn, sizeof(int64_t) * (size_t)old_n);
        for (int e = 0; e < add_n; e++) upd[old_n + e] = order[at + links[i0 + e].t];
        put_list(&g, level, base, upd, old_n + add_n);
        free(upd);
      } else {
        int cn = old_n + add_n;
        if (cn > link_cap) cn = link_cap;
        for (int i = 0; i < cn; i++) {
          tid[i] = i < old_n ? conn[i] : order[at + links[i0 + (i - old_n)].t];
          td[i] = item_distance(&g, base, tid[i]);
        }
        for (int i = 0; i < cn; i++) { /* ascending by (Float.compare, position) */
          int rank = 0;
          for (int e = 0; e < cn; e++) {
            int c = jfloat_compare(td[e], td[i]);
            rank += (c < 0 || (c == 0 && e < i)) ? 1 : 0;
          }
          cid[rank] = tid[i];
          cd[rank] = td[i];
        }
        int64_t *upd = malloc(sizeof(int64_t) * (size_t)(M + 1));
        int nk = 0;
        for (int i = 0; i < cn && nk < M; i++) {
          if (cid[i] == base) continue;
          int include = 1;
          for (int k = 0; k < nk; k++)
            if (item_distance(&g, upd[k], cid[i]) < cd[i]) { include = 0; break; }
          if (include) upd[nk++] = cid[i];
        }
        put_list(&g, level, base, upd, nk);
        free(upd);
      }
      i0 = i1;
    }
    at += m;
    linked += m;
  }
  int64_t ne = 0, nnb = 0;
  int fits = 1;
  for (int l = 0; l < g.n_levels && fits; l++)
    for (int64_t i = 0; i < n; i++) {
      if (g.cnt[l][i] < 0) continue;
      if (ne >= cap_entries || nnb + g.cnt[l][i] > cap_neighbours) { fits = 0; break; }
      entry_level[ne] = l;
      entry_item[ne] = i;
      entry_offsets[ne] = nnb;
      memcpy(entry_neighbours + nnb, g.nb[l] + (size_t)i * g.cap, sizeof(int64_t) * (size_t)g.cnt[l][i]);
      nnb += g.cnt[l][i];
      ne++;
    }
  if (fits) entry_offsets[ne] = nnb;
  *entry_point = g.entry;
  *max_level = g.max_level;
  for (int l = 0; l < g.n_levels; l++) { free(g.cnt[l]); free(g.nb[l]); }
  free(g.cnt); free(g.nb); free(visited); free(order); free(links); free(neigh); free(tid); free(cid); free(td); free(cd);
  return fits ? ne : -1;
}
