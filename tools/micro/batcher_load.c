/* batcher_load.c -- native load generator for the micro-batching queue (measurement tooling, not product code).
 *
 * The reference's callers are Finagle worker threads that each make ONE getTweetCandidates request at a time
 * (SimClustersANNCandidateSource.scala:77-94).  A Python harness cannot offer that load (one interpreter lock): this file
 * does, with pthreads, through the C ABI only (sann_batcher_get_tweet_candidates).  Built by __graft_entry__.build() into
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
  int32_t nq, first, step, n_req, k;
  int64_t *ids;      /* [k] scratch of this thread */
  double *sc;
  double *lat_us;    /* [n_req] */
  int64_t candidates, checksum;
  int rc;
} worker_t;

static double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static void *run(void *p) {
  worker_t *w = (worker_t *)p;
  int32_t q = w->first % w->nq;
  for (int32_t i = 0; i < w->n_req; i++) {
    int32_t cnt = 0, msz = 0;
    const double t0 = now_us();
    const int rc = sann_batcher_get_tweet_candidates(w->mb, w->now_ms, (int32_t)(w->offs[q + 1] - w->offs[q]), w->cids + w->offs[q],
                                                     w->scs + w->offs[q], 0, 0, w->cfg, w->k, w->ids, w->sc, &cnt, &msz);
    w->lat_us[i] = now_us() - t0;
    if (rc != SANN_OK) { w->rc = rc; return NULL; }
    w->candidates += cnt;
    for (int32_t j = 0; j < cnt; j += 97) w->checksum ^= w->ids[j] + q;
    q = (q + w->step) % w->nq;
  }
  return NULL;
}

static int cmp_d(const void *a, const void *b) { return (*(const double *)a > *(const double *)b) - (*(const double *)a < *(const double *)b); }

/* n_threads callers x n_req blocking single requests each, walking the nq queries (CSR offs / cids / scs) with stride
 * n_threads.  out[0] = seconds, out[1] = requests, out[2] = candidates, out[3..5] = latency p50 / p99 / max in us.
 * Returns the first non-zero status of any request. */
int batcher_load_run(sann_batcher_t *mb, int32_t n_threads, int32_t n_req, int32_t nq, const int64_t *offs, const int32_t *cids,
                     const double *scs, const sann_config_t *cfg, int64_t now_ms, double *out) {
  if (n_threads < 1 || n_req < 1 || nq < 1) return SANN_EINVAL;
  const int32_t k = cfg->max_num_results < 1000 ? (cfg->max_num_results < 1 ? 1 : cfg->max_num_results) : 1000;
  worker_t *w = (worker_t *)calloc((size_t)n_threads, sizeof(worker_t));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  double *lat = (double *)malloc(sizeof(double) * (size_t)n_threads * (size_t)n_req);
  if (!w || !th || !lat) return SANN_ENOMEM;
  for (int32_t t = 0; t < n_threads; t++) {
    w[t] = (worker_t){mb, offs, cids, scs, cfg, now_ms, nq, t, n_threads, n_req, k, NULL, NULL, lat + (size_t)t * n_req, 0, 0, 0};
    w[t].ids = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
    w[t].sc = (double *)malloc(sizeof(double) * (size_t)k);
    if (!w[t].ids || !w[t].sc) return SANN_ENOMEM;
  }
  const double t0 = now_us();
  for (int32_t t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, run, &w[t]);
  int rc = SANN_OK;
  int64_t cand = 0;
  for (int32_t t = 0; t < n_threads; t++) {
    pthread_join(th[t], NULL);
    if (w[t].rc && !rc) rc = w[t].rc;
    cand += w[t].candidates;
    free(w[t].ids);
    free(w[t].sc);
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
