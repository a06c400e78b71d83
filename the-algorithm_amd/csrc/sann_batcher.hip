// sann_batcher.hip -- the native micro-batching queue in front of the batched operator (host code only).
//
// The reference calls ApproximateCosineSimilarity.apply once per request, from Finagle worker threads, inside a Future.map
// (simclusters-ann/server/src/main/scala/com/twitter/simclustersann/candidate_source/SimClustersANNCandidateSource.scala:77-94)
// with a 40 ms budget (modules/FlagsModule.scala:8-12).  One request is 32 units of work for a 256-CU GPU (0.16 ms of
// latency, 6 k requests/s per caller thread); the GPU wants ~1000 requests per launch.  This queue folds concurrent
// single-request calls into batches:
//
//   sann_submit           copies the request (embedding, config, source tweet, Time.now) into the OPEN batch and returns a ticket
//   a batch closes        when it holds max_batch requests, or max_wait_us after its first request arrived
//   dispatcher threads    (n_dispatchers, each with a pooled batch object and HIP stream of its own) take closed batches and
//                         run them through the same call a batched caller makes (sann_get_tweet_candidates_at: every
//                         request keeps ITS OWN now_ms, so the age window of ApproximateCosineSimilarity.scala:65-72 is the
//                         one the request would have seen alone), then hand every request its rows
//   sann_wait / sann_poll the caller collects its answer (sann_batcher_get_tweet_candidates = submit + wait)
//
// Results are bit for bit those of the request run alone (tests/test_batcher_gpu.py).  Requests use the default cluster
// selection of fetchCandidates (truncate(maxScanClusters), SimClustersANNCandidateSource.scala:72-75); explicit scan keys
// stay with the batch API.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/simclusters_ann.h"
#include "sann_host.h"
#include "abi_guard.h"

using sann_host::fail;
#define ABI_CATCH catch (...) { return abi_guard::caught(sann_host::fail, SANN_ENOMEM, SANN_EINTERNAL); }

namespace {

using Clock = std::chrono::steady_clock;

// completion of one batch: its requests' callers sleep here, not on a queue-wide condition (a finished batch would wake
// every waiting caller of every batch)
struct BatchSync {
  std::mutex m;
  std::condition_variable cv;
  bool done = false;
};

struct Request {
  int64_t ticket = 0;
  std::shared_ptr<BatchSync> sync;
  int32_t out_cap = 0;
  int64_t *out_ids = nullptr;
  double *out_scores = nullptr;
  int32_t *out_count = nullptr, *out_map_size = nullptr;
  int status = SANN_OK;
  std::string message;
};

// The embeddings of a batch's requests, back to back.  Plain buffers, not std::vector: a submitter RESERVES its stretch under the
// queue's lock and copies into it after letting the lock go (`pending` counts the copies still running; the dispatcher waits
// for zero before it reads), so growing must never move the buffer under a writer -- a batch that would outgrow its buffer is
// closed instead, and only an EMPTY batch grows.
struct EmbBuf {
  std::unique_ptr<int32_t[]> cids;
  std::unique_ptr<double[]> scores;
  size_t size = 0, cap = 0;
  void reserve_empty(size_t n) {  // (size == 0: nobody holds a pointer into the old buffers)
    if (n <= cap) return;
    cids.reset(new int32_t[n]);
    scores.reset(new double[n]);
    cap = n;
  }
};

struct Batch {
  std::shared_ptr<BatchSync> sync{new BatchSync()};
  std::vector<int64_t> emb_offsets{0};
  EmbBuf emb;
  std::atomic<int> pending{0};
  std::vector<int64_t> src, now;
  std::vector<uint8_t> has_src;
  std::vector<sann_config_t> cfgs;
  std::vector<std::shared_ptr<Request>> reqs;
  Clock::time_point deadline;
  int kmax = 1;
  void clear() {
    emb_offsets.assign(1, 0);
    emb.size = 0;
    src.clear(); now.clear(); has_src.clear(); cfgs.clear(); reqs.clear();
    kmax = 1;
    sync.reset(new BatchSync());
  }
};

// pinned response buffers of one dispatcher, grown on demand
struct OutBuf {
  void *p = nullptr;
  size_t cap = 0;
  ~OutBuf() { if (p) (void)hipHostFree(p); }
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipHostMalloc(&p, n + n / 4, hipHostMallocDefault);
    if (e == hipSuccess) cap = n + n / 4;
    return e;
  }
};

}  // namespace

struct sann_batcher {
  sann_index *ix = nullptr;
  sann_batcher_options_t opt{};
  std::mutex mu;
  std::condition_variable cv_work;
  std::unique_ptr<Batch> open;                 // accepting requests (never null)
  std::deque<std::unique_ptr<Batch>> ready;    // closed, waiting for a dispatcher
  std::vector<std::unique_ptr<Batch>> spare;   // recycled
  // the ticket table has locks of its own (collecting answers does not hold up submissions), sixteen of them: a ticket's
  // shard is its low bits, so callers that submit and collect at a million requests a second rarely meet
  static constexpr int TK_SHARDS = 16;
  struct TicketShard {
    std::mutex mu;
    std::unordered_map<int64_t, std::shared_ptr<Request>> map;
  };
  TicketShard tk[TK_SHARDS];
  TicketShard &shard_of(int64_t ticket) { return tk[(size_t)ticket & (TK_SHARDS - 1)]; }
  std::vector<std::thread> workers;
  std::atomic<int64_t> next_ticket{1};
  bool stop = false;
  // statistics
  int64_t n_requests = 0, n_batches = 0, n_full = 0, n_timeout = 0, max_batch_seen = 0;

  void close_open_locked() {  // open -> ready (mu held)
    if (open->reqs.empty()) return;
    ready.push_back(std::move(open));
    if (!spare.empty()) { open = std::move(spare.back()); spare.pop_back(); }
    else {
      open.reset(new Batch());
      open->reqs.reserve((size_t)opt.max_batch);
    }
    open->emb.reserve_empty((size_t)opt.max_batch * 64);
    cv_work.notify_one();
  }

  void run_batch(Batch &bt, OutBuf &ob) {
    const int nq = (int)bt.reqs.size();
    const int stride = bt.kmax;
    int rc = SANN_OK;
    std::string msg;
    int64_t *ids = nullptr;
    double *sc = nullptr;
    int32_t *cnt = nullptr, *msz = nullptr;
    const size_t row = (size_t)stride * 8, need = 2 * row * (size_t)nq + 8 * (size_t)nq;
    if (hipSetDevice(ix->device) != hipSuccess || ob.reserve(need) != hipSuccess) {
      rc = SANN_EDEVICE;
      msg = "micro-batcher: pinned response buffer";
    } else {
      ids = (int64_t *)ob.p;
      sc = (double *)((char *)ob.p + row * (size_t)nq);
      cnt = (int32_t *)((char *)ob.p + 2 * row * (size_t)nq);
      msz = cnt + nq;
      rc = sann_candidates_pooled(ix, opt.variant, bt.now[0], bt.now.data(), nq, bt.emb_offsets.data(), bt.emb.cids.get(),
                                  bt.emb.scores.get(), bt.src.data(), bt.has_src.data(), bt.cfgs.data(), nq, nullptr, nullptr, ids, sc,
                                  stride, cnt, msz);
      if (rc != SANN_OK) msg = sann_last_error();
    }
    // hand every request its rows (its own thread may be anywhere: the copy is ours)
    for (int q = 0; q < nq; q++) {
      Request &r = *bt.reqs[(size_t)q];
      int st = rc;
      if (rc == SANN_OK) {
        const int c = cnt[q];
        if (c > r.out_cap) {
          st = SANN_EINVAL;
          r.message = "out_capacity smaller than the number of results";
        } else {
          if (c > 0) {
            memcpy(r.out_ids, ids + (size_t)q * stride, (size_t)c * 8);
            memcpy(r.out_scores, sc + (size_t)q * stride, (size_t)c * 8);
          }
          if (r.out_count) *r.out_count = c;
          if (r.out_map_size) *r.out_map_size = msz[q];
        }
      } else {
        r.message = msg;
      }
      r.status = st;
    }
    {
      std::lock_guard<std::mutex> lk(bt.sync->m);
      bt.sync->done = true;
    }
    bt.sync->cv.notify_all();
  }

  void worker() {
    OutBuf ob;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      // a closed batch, or the open batch's deadline, or the end
      while (!stop && ready.empty()) {
        if (!open->reqs.empty()) {
          if (Clock::now() >= open->deadline) {
            n_timeout++;
            close_open_locked();
            break;
          }
          cv_work.wait_until(lk, open->deadline);
        } else {
          cv_work.wait(lk);
        }
      }
      if (ready.empty()) {
        if (stop) {
          if (open->reqs.empty()) return;
          close_open_locked();  // drain what was submitted before the stop
        }
        if (ready.empty()) continue;
      }
      std::unique_ptr<Batch> bt = std::move(ready.front());
      ready.pop_front();
      n_batches++;
      max_batch_seen = std::max<int64_t>(max_batch_seen, (int64_t)bt->reqs.size());
      lk.unlock();
      while (bt->pending.load(std::memory_order_acquire) != 0) std::this_thread::yield();  // (submitters still copying their embeddings in)
      run_batch(*bt, ob);
      bt->clear();
      lk.lock();
      if (spare.size() < 8) spare.push_back(std::move(bt));
    }
  }
};

extern "C" {

int sann_batcher_create(sann_index_t *index, const sann_batcher_options_t *options, sann_batcher_t **out) try {
  if (!out) return fail(SANN_EINVAL, "out is NULL");
  *out = nullptr;
  if (!index) return fail(SANN_EINVAL, "index is NULL");
  sann_batcher_options_t o{};
  if (options) o = *options;
  if (o.variant < 0 || o.variant > 3) return fail(SANN_EINVAL, "unknown variant");
  if (o.max_batch == 0) o.max_batch = 1024;
  if (o.max_wait_us == 0) o.max_wait_us = 500;
  if (o.n_dispatchers == 0) o.n_dispatchers = 3;
  if (o.max_batch < 1 || o.max_batch > 65536 || o.max_wait_us < 0 || o.n_dispatchers < 1 || o.n_dispatchers > 16)
    return fail(SANN_EINVAL, "max_batch in 1..65536, max_wait_us >= 0, n_dispatchers in 1..16");
  std::unique_ptr<sann_batcher> b(new sann_batcher());
  b->ix = index;
  b->opt = o;
  b->open.reset(new Batch());
  b->open->emb.reserve_empty((size_t)o.max_batch * 64);
  for (int i = 0; i < o.n_dispatchers; i++) b->workers.emplace_back([p = b.get()] { p->worker(); });
  *out = b.release();
  return SANN_OK;
} ABI_CATCH

int sann_submit(sann_batcher_t *b, int64_t now_ms, int32_t n_embedding, const int32_t *cluster_ids, const double *scores,
                int64_t source_tweet_id, int32_t has_source_tweet, const sann_config_t *config, int32_t out_capacity,
                int64_t *out_ids, double *out_scores, int32_t *out_count, int32_t *out_map_size, int64_t *ticket) try {
  if (!b || !config || !ticket) return fail(SANN_EINVAL, "NULL argument");
  if (n_embedding < 0 || (n_embedding > 0 && (!cluster_ids || !scores))) return fail(SANN_EINVAL, "bad embedding");
  if (b->opt.variant == SANN_VARIANT_LEGACY) {
    if (config->ann_algorithm != SANN_ALG_DOT_PRODUCT && config->ann_algorithm != SANN_ALG_COSINE && config->ann_algorithm != SANN_ALG_LOG_COSINE)
      return fail(SANN_EINVAL, "legacy variant: ann_algorithm must be dot product, cosine or log cosine");
    if (config->max_num_results > 1000) return fail(SANN_ELIMIT, "legacy variant: max_num_results above 1000");
  }
  const int k = config->max_num_results < 1000 ? (config->max_num_results < 0 ? 0 : config->max_num_results) : 1000;
  if (out_capacity < k || (k > 0 && (!out_ids || !out_scores))) return fail(SANN_EINVAL, "out_capacity must hold min(maxNumResults, 1000) results");
  std::shared_ptr<Request> r = std::make_shared<Request>();
  r->ticket = b->next_ticket.fetch_add(1, std::memory_order_relaxed);
  r->out_cap = out_capacity;
  r->out_ids = out_ids;
  r->out_scores = out_scores;
  r->out_count = out_count;
  r->out_map_size = out_map_size;
  {
    auto &sh = b->shard_of(r->ticket);
    std::lock_guard<std::mutex> lk(sh.mu);
    sh.map.emplace(r->ticket, r);
  }
  Batch *copy_to = nullptr;
  size_t copy_at = 0;
  {
    std::lock_guard<std::mutex> lk(b->mu);
    if (b->stop) {
      auto &sh = b->shard_of(r->ticket);
      std::lock_guard<std::mutex> lk2(sh.mu);
      sh.map.erase(r->ticket);
      return fail(SANN_EINVAL, "the batcher is shutting down");
    }
    if (b->open->emb.size + (size_t)n_embedding > b->open->emb.cap) {  // would outgrow the buffer: close, and grow the (empty) next one
      if (!b->open->reqs.empty()) {
        b->n_full++;
        b->close_open_locked();
      }
      b->open->emb.reserve_empty(std::max((size_t)n_embedding * 2, (size_t)b->opt.max_batch * 64));
    }
    Batch &bt = *b->open;
    if (bt.reqs.empty()) bt.deadline = Clock::now() + std::chrono::microseconds(b->opt.max_wait_us);
    copy_to = &bt;
    copy_at = bt.emb.size;
    bt.emb.size += (size_t)n_embedding;
    bt.pending.fetch_add(1, std::memory_order_relaxed);
    bt.emb_offsets.push_back((int64_t)bt.emb.size);
    bt.src.push_back(has_source_tweet ? source_tweet_id : 0);
    bt.has_src.push_back(has_source_tweet ? 1 : 0);
    bt.now.push_back(now_ms);
    bt.cfgs.push_back(*config);
    bt.kmax = std::max(bt.kmax, k);
    r->sync = bt.sync;
    bt.reqs.push_back(r);
    b->n_requests++;
    const bool first = bt.reqs.size() == 1;
    if ((int)bt.reqs.size() >= b->opt.max_batch) {
      b->n_full++;
      b->close_open_locked();
    } else if (first) {
      b->cv_work.notify_one();  // a dispatcher now has a deadline to sleep towards
    }
  }
  // the embedding itself travels outside the lock, into the stretch reserved above
  if (n_embedding > 0) {
    memcpy(copy_to->emb.cids.get() + copy_at, cluster_ids, (size_t)n_embedding * sizeof(int32_t));
    memcpy(copy_to->emb.scores.get() + copy_at, scores, (size_t)n_embedding * sizeof(double));
  }
  copy_to->pending.fetch_sub(1, std::memory_order_release);
  *ticket = r->ticket;
  return SANN_OK;
} ABI_CATCH

static int collect(sann_batcher *b, int64_t ticket, bool block, int32_t *done) {
  std::shared_ptr<Request> r;
  auto &sh = b->shard_of(ticket);
  {
    std::lock_guard<std::mutex> lk(sh.mu);
    auto it = sh.map.find(ticket);
    if (it == sh.map.end()) return fail(SANN_EINVAL, "unknown ticket (already collected?)");
    r = it->second;
  }
  bool is_done;
  {
    std::unique_lock<std::mutex> lk(r->sync->m);
    if (block) r->sync->cv.wait(lk, [&] { return r->sync->done; });
    is_done = r->sync->done;
  }
  if (done) *done = is_done ? 1 : 0;
  if (!is_done) return SANN_OK;
  {
    std::lock_guard<std::mutex> lk(sh.mu);
    if (sh.map.erase(ticket) == 0) return fail(SANN_EINVAL, "unknown ticket (already collected?)");
  }
  if (r->status != SANN_OK) return fail(r->status, r->message);
  return SANN_OK;
}

int sann_wait(sann_batcher_t *b, int64_t ticket) try {
  if (!b) return fail(SANN_EINVAL, "batcher is NULL");
  return collect(b, ticket, true, nullptr);
} ABI_CATCH

int sann_poll(sann_batcher_t *b, int64_t ticket, int32_t *done) try {
  if (!b || !done) return fail(SANN_EINVAL, "NULL argument");
  return collect(b, ticket, false, done);
} ABI_CATCH

int sann_batcher_get_tweet_candidates(sann_batcher_t *b, int64_t now_ms, int32_t n_embedding, const int32_t *cluster_ids,
                                      const double *scores, int64_t source_tweet_id, int32_t has_source_tweet,
                                      const sann_config_t *config, int32_t out_capacity, int64_t *out_ids, double *out_scores,
                                      int32_t *out_count, int32_t *out_map_size) try {
  int64_t t = 0;
  int rc = sann_submit(b, now_ms, n_embedding, cluster_ids, scores, source_tweet_id, has_source_tweet, config, out_capacity, out_ids,
                       out_scores, out_count, out_map_size, &t);
  if (rc != SANN_OK) return rc;
  return collect(b, t, true, nullptr);
} ABI_CATCH

int sann_batcher_stats(sann_batcher_t *b, sann_batcher_stats_t *stats) try {
  if (!b || !stats) return fail(SANN_EINVAL, "NULL argument");
  std::lock_guard<std::mutex> lk(b->mu);
  stats->n_requests = b->n_requests;
  stats->n_batches = b->n_batches;
  stats->n_closed_full = b->n_full;
  stats->n_closed_by_deadline = b->n_timeout;
  stats->max_batch = b->max_batch_seen;
  return SANN_OK;
} ABI_CATCH

int sann_batcher_destroy(sann_batcher_t *b) try {
  if (!b) return SANN_OK;
  {
    std::lock_guard<std::mutex> lk(b->mu);
    b->stop = true;
  }
  b->cv_work.notify_all();
  for (auto &t : b->workers) t.join();
  // (requests submitted before the stop were run; tickets nobody collected are dropped with the object)
  delete b;
  return SANN_OK;
} ABI_CATCH

}  // extern "C"
