/* batcher_load.c -- native load generator for the micro-batching queue (measurement tooling, not product code).
 *
 * The reference's callers are Finagle worker threads, each holding many outstanding single-request Futures
 * (SimClustersANNCandidateSource.scala:77-94).  A Python harness cannot offer that load (one interpreter lock): this file
 * does, with a few pthreads that each keep a window of requests in flight, through the C ABI only (sann_submit / sann_wait).  Built by __graft_entry__.build() into
 * tools/micro/libbatcher_load.so and driven by bench.py / tests through ctypes; the index, the batcher and the query
 * arrays are made by the caller. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/simclusters_ann.h"

typedef struct {
  sann_batcher_t *mb;
  const int64_t *offs;
  const int32_t *cids;
  const double *scs;
  const sann_config_t *cfg;
  int64_t now_ms;
  int32_t nq, first, step, n_req, k, window;
  double *lat_us;    /* [n_req] */
  int64_t candidates, checksum;
  int rc;
} worker_t;

static double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

/* One caller thread with `window` requests in flight (a Finagle worker thread holds many outstanding Futures, it does not
 * block on each): submit until the window is full, collect the oldest, submit the next. */
static void *run(void *p) {
  worker_t *w = (worker_t *)p;
  const int32_t W = w->window, k = w->k;
  int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * (size_t)k * (size_t)W);
  double *sc = (double *)malloc(sizeof(double) * (size_t)k * (size_t)W);
  int32_t *cnt = (int32_t *)calloc((size_t)W * 2, sizeof(int32_t));
  int64_t *ticket = (int64_t *)calloc((size_t)W, sizeof(int64_t));
  double *t0 = (double *)calloc((size_t)W, sizeof(double));
  int32_t *qq = (int32_t *)calloc((size_t)W, sizeof(int32_t));
  if (!ids || !sc || !cnt || !ticket || !t0 || !qq) { w->rc = SANN_ENOMEM; return NULL; }
  int32_t q = w->first % w->nq, submitted = 0, collected = 0;
  while (collected < w->n_req) {
    while (submitted < w->n_req && submitted - collected < W) {
      const int32_t s = submitted % W;
      t0[s] = now_us();
      qq[s] = q;
      const int rc = sann_submit(w->mb, w->now_ms, (int32_t)(w->offs[q + 1] - w->offs[q]), w->cids + w->offs[q], w->scs + w->offs[q], 0, 0,
                                 w->cfg, k, ids + (size_t)s * k, sc + (size_t)s * k, &cnt[2 * s], &cnt[2 * s + 1], &ticket[s]);
      if (rc != SANN_OK) { w->rc = rc; goto out; }
      submitted++;
      q = (q + w->step) % w->nq;
    }
    {
      const int32_t s = collected % W;
      const int rc = sann_wait(w->mb, ticket[s]);
      w->lat_us[collected] = now_us() - t0[s];
      if (rc != SANN_OK) { w->rc = rc; goto out; }
      w->candidates += cnt[2 * s];
      for (int32_t j = 0; j < cnt[2 * s]; j += 97) w->checksum ^= ids[(size_t)s * k + j] + qq[s];
      collected++;
    }
  }
out:
  free(ids); free(sc); free(cnt); free(ticket); free(t0); free(qq);
  return NULL;
}

static int cmp_d(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }

/* n_threads callers x n_req single requests each, `window` of them in flight per caller, walking the nq queries (CSR
 * offs / cids / scs) with stride n_threads.  out[0] = seconds, out[1] = requests, out[2] = candidates, out[3..5] = latency
 * p50 / p99 / max in us (submit -> collected).  Returns the first non-zero status of any request. */
int batcher_load_run(sann_batcher_t *mb, int32_t n_threads, int32_t window, int32_t n_req, int32_t nq, const int64_t *offs,
                     const int32_t *cids, const double *scs, const sann_config_t *cfg, int64_t now_ms, double *out) {
  if (n_threads < 1 || n_req < 1 || nq < 1 || window < 1) return SANN_EINVAL;
  const int32_t k = cfg->max_num_results < 1000 ? (cfg->max_num_results < 1 ? 1 : cfg->max_num_results) : 1000;
  worker_t *w = (worker_t *)calloc((size_t)n_threads, sizeof(worker_t));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  double *lat = (double *)malloc(sizeof(double) * (size_t)n_threads * (size_t)n_req);
  if (!w || !th || !lat) return SANN_ENOMEM;
  for (int32_t t = 0; t < n_threads; t++)
    w[t] = (worker_t){mb, offs, cids, scs, cfg, now_ms, nq, t, n_threads, n_req, k, window, lat + (size_t)t * n_req, 0, 0, 0};
  const double t0 = now_us();
  for (int32_t t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, run, &w[t]);
  int rc = SANN_OK;
  int64_t cand = 0;
  for (int32_t t = 0; t < n_threads; t++) {
    pthread_join(th[t], NULL);
    if (w[t].rc && !rc) rc = w[t].rc;
    cand += w[t].candidates;
  }
  const double sec = (now_us() - t0) * 1e-6;
  const size_t n = (size_t)n_threads * (size_t)n_req;
  qsort(lat, n, sizeof(double), cmp_d);
  out[0] = sec;
  out[1] = (double)n;
  out[2] = (double)cand;
  out[3] = lat[n / 2];
  out[4] = lat[(size_t)((double)n * 0.99)];
  out[5] = lat[n - 1];
  free(lat);
  free(th);
  free(w);
  return rc;
}

/* ---- the batched boundary call under native callers (bench.py's end_to_end leg) -------------------------------------------
 * n_threads callers, each making n_calls blocking sann_get_tweet_candidates calls of nq queries; call i of thread t takes
 * query set (t + i * n_threads) % n_sets.  Outputs go to per-thread buffers from sann_host_alloc (pinned).  Python threads
 * cannot do this: their argument marshalling holds the interpreter lock for ~0.1 ms per call, which is the GPU's whole
 * step.  out[0] = seconds, out[1] = calls, out[2] = candidates. */
typedef struct {
  sann_index_t *ix;
  const sann_config_t *cfg;
  int64_t now_ms;
  int32_t nq, n_sets, t, n_threads, n_calls, k;
  const int64_t *const *offs;
  const int32_t *const *cids;
  const double *const *scs;
  int64_t candidates;
  int rc;
  void *buf[4]; /* pinned response buffers (made before the clock starts: hipHostMalloc takes milliseconds) */
} e2e_worker_t;

static void *e2e_run(void *p) {
  e2e_worker_t *w = (e2e_worker_t *)p;
  void *ids = w->buf[0], *sc = w->buf[1], *cnt = w->buf[2], *msz = w->buf[3];
  for (int32_t i = 0; i < w->n_calls; i++) {
    const int32_t s = (w->t + i * w->n_threads) % w->n_sets;
    const int rc = sann_get_tweet_candidates(w->ix, 0, w->now_ms, w->nq, w->offs[s], w->cids[s], w->scs[s], NULL, NULL, w->cfg, 1, NULL, NULL,
                                             (int64_t *)ids, (double *)sc, w->k, (int32_t *)cnt, (int32_t *)msz);
    if (rc != SANN_OK) { w->rc = rc; break; }
    for (int32_t q = 0; q < w->nq; q++) w->candidates += ((int32_t *)cnt)[q];
  }
  return NULL;
}

int e2e_load_run(sann_index_t *ix, int32_t n_threads, int32_t n_calls_total, int32_t nq, int32_t n_sets, const int64_t *const *offs,
                 const int32_t *const *cids, const double *const *scs, const sann_config_t *cfg, int64_t now_ms, double *out) {
  if (n_threads < 1 || n_calls_total < 1 || nq < 1 || n_sets < 1) return SANN_EINVAL;
  const int32_t k = cfg->max_num_results < 1000 ? (cfg->max_num_results < 1 ? 1 : cfg->max_num_results) : 1000;
  e2e_worker_t *w = (e2e_worker_t *)calloc((size_t)n_threads, sizeof(e2e_worker_t));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  if (!w || !th) return SANN_ENOMEM;
  int32_t total = 0;
  for (int32_t t = 0; t < n_threads; t++) {
    const int32_t n = n_calls_total / n_threads + (t < n_calls_total % n_threads ? 1 : 0);
    w[t] = (e2e_worker_t){ix, cfg, now_ms, nq, n_sets, t, n_threads, n, k, offs, cids, scs, 0, 0, {NULL, NULL, NULL, NULL}};
    if (sann_host_alloc((int64_t)nq * k * 8, &w[t].buf[0]) || sann_host_alloc((int64_t)nq * k * 8, &w[t].buf[1]) ||
        sann_host_alloc((int64_t)nq * 4, &w[t].buf[2]) || sann_host_alloc((int64_t)nq * 4, &w[t].buf[3]))
      return SANN_ENOMEM;
    total += n;
  }
  const double t0 = now_us();
  for (int32_t t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, e2e_run, &w[t]);
  int rc = SANN_OK;
  int64_t cand = 0;
  for (int32_t t = 0; t < n_threads; t++) {
    pthread_join(th[t], NULL);
    if (w[t].rc && !rc) rc = w[t].rc;
    cand += w[t].candidates;
  }
  out[0] = (now_us() - t0) * 1e-6;
  for (int32_t t = 0; t < n_threads; t++)
    for (int i = 0; i < 4; i++) sann_host_free(w[t].buf[i]);
  out[1] = (double)total;
  out[2] = (double)cand;
  free(th);
  free(w);
  return rc;
}
