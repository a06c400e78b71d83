/*
 * oracle_baseline.c -- multi-threaded driver that times the restated Scala CPU path
 * (oracle_sann_query) over a batch of queries.  TEST / BENCH INFRASTRUCTURE ONLY (see
 * simclusters_oracle.c).  It is the `cpu_baseline` leg of bench.py, labelled
 * "C restatement of the Scala CPU path (N threads)", never "JVM" (BASELINE.md section 2).
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

typedef struct {
  int32_t max_num_results; double min_score; int32_t candidate_embedding_type;
  int32_t max_top_tweets_per_cluster, max_scan_clusters, max_tweet_candidate_age_hours,
      min_tweet_candidate_age_hours, ann_algorithm;
} oracle_sann_config;

int32_t oracle_sann_query(int32_t variant, int32_t n_emb, const int32_t *emb_ids, const double *emb_scores,
                          int32_t has_source_tweet, int64_t source_tweet_id, const oracle_sann_config *cfg,
                          int64_t now_ms, int32_t n_lists, const int32_t *list_cluster_ids,
                          const int64_t *list_offsets, const int64_t *tweet_ids, const double *scores,
                          int32_t n_scan_order, const int32_t *scan_order, int64_t *out_ids, double *out_scores,
                          int32_t *map_size);

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
    j->out_counts[q] = oracle_sann_query(j->variant, (int32_t)(e - b), j->emb_ids + b, j->emb_scores + b, 0, 0, j->cfg,
                                         j->now_ms, j->n_lists, j->list_cluster_ids, j->list_offsets, j->tweet_ids,
                                         j->scores, 0, NULL, j->out_ids + (int64_t)q * j->stride,
                                         j->out_scores + (int64_t)q * j->stride, &msz);
  }
  return NULL;
}

/* out_ids/out_scores: [nq * 1000]; returns elapsed wall seconds. */
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
