"""The native micro-batching queue (include/simclusters_ann.h: sann_batcher_*, sann_submit / sann_wait / sann_poll): single
requests from many threads -- the reference's calling pattern, SimClustersANNCandidateSource.scala:77-94 -- are folded into
batches, and every request gets, bit for bit, the answer it would have had alone: the oracle's (every request keeps its own
Time.now, its own config, its own source tweet)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _requests(pkg, co, n, seed):
    rng = np.random.default_rng(seed)
    offs, cids, scs = pkg.corpus.make_queries(n, len(co.cluster_ids), seed=seed, clusters_per_user=30)
    SA = pkg.ScoringAlgorithm
    reqs = []
    for q in range(n):
        cfg = pkg.SimClustersANNConfig(
            maxNumResults=int(rng.choice([5, 100, 400, 1000])), minScore=float(rng.choice([0.0, 0.02])),
            maxTopTweetsPerCluster=int(rng.choice([50, 800])), maxScanClusters=int(rng.choice([10, 50])),
            maxTweetCandidateAgeHours=int(rng.choice([24, 6])), minTweetCandidateAgeHours=int(rng.choice([0, 1])),
            annAlgorithm=SA(int(rng.choice([1, 2, 3, 4]))))
        src = int(rng.choice(co.tweet_ids)) if rng.random() < 0.3 else None
        now = co.now_ms - int(rng.integers(0, 3_600_000))  # requests arrive over an hour: the age windows differ
        reqs.append((cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], cfg, src, now))
    return reqs


def _oracle_answer(oracle, co, r):
    e_c, e_s, cfg, src, now = r
    return oracle.sann_query(e_c, e_s, src, cfg, now, co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)


def test_sixty_four_threads_of_single_requests_get_the_batched_answer(pkg, oracle):
    co = pkg.corpus.make_corpus(60_000, 1500, seed=21, index_cap=1000)
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
    n_thr, per = 64, 12
    reqs = _requests(pkg, co, n_thr * per, 5)
    mb = pkg.MicroBatcher(index, max_batch=256, max_wait_us=300, n_dispatchers=3)
    got = [None] * len(reqs)
    errs = []

    def caller(t):
        try:
            for i in range(per):
                j = t * per + i
                e_c, e_s, cfg, src, now = reqs[j]
                got[j] = mb.get_tweet_candidates(e_c, e_s, cfg, now_ms=now, source_tweet_id=src)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=caller, args=(t,)) for t in range(n_thr)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs[0]
    st = mb.stats()
    assert st.n_requests == len(reqs) and 1 <= st.n_batches < len(reqs), "requests were folded into batches"
    assert st.max_batch > 1
    # the same requests as ONE batch with per-query Time.now: the batch call the dispatchers make
    offs = np.zeros(len(reqs) + 1, np.int64)
    offs[1:] = np.cumsum([len(r[0]) for r in reqs])
    b_ids, b_sc, b_cnt, b_msz = pkg.simclusters_ann.get_tweet_candidates(
        index, offs, np.concatenate([r[0] for r in reqs]), np.concatenate([r[1] for r in reqs]), [r[2] for r in reqs],
        now_ms=np.array([r[4] for r in reqs], np.int64), source_tweet_ids=np.array([r[3] or 0 for r in reqs], np.int64),
        has_source_tweet=np.array([r[3] is not None for r in reqs], np.uint8))
    for j, r in enumerate(reqs):
        ids, sc, msz = got[j]
        assert len(ids) == b_cnt[j] and msz == b_msz[j]
        assert np.array_equal(ids, b_ids[j, :b_cnt[j]]) and np.array_equal(sc.view(np.int64), b_sc[j, :b_cnt[j]].view(np.int64))
        if j % 7 == 0:  # and the oracle's, for a sample
            o_ids, o_sc, o_msz = _oracle_answer(oracle, co, r)
            assert np.array_equal(ids, o_ids) and np.array_equal(sc.view(np.int64), o_sc.view(np.int64)) and msz == o_msz
    mb.close()
    index.close()


def test_async_tickets_deadline_and_errors(pkg, oracle):
    co = pkg.corpus.make_corpus(20_000, 600, seed=22, index_cap=400)
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
    reqs = _requests(pkg, co, 40, 6)
    mb = pkg.MicroBatcher(index, max_batch=1000, max_wait_us=2000)  # never full: every batch leaves by its deadline
    tickets = [mb.submit(r[0], r[1], r[2], now_ms=r[4], source_tweet_id=r[3]) for r in reqs]
    for (t, out), r in zip(tickets, reqs):
        mb.wait(t)
        o_ids, o_sc, o_msz = _oracle_answer(oracle, co, r)
        n = int(out[2][0])
        assert n == len(o_ids) and int(out[3][0]) == o_msz
        assert np.array_equal(out[0][:n], o_ids) and np.array_equal(out[1][:n].view(np.int64), o_sc.view(np.int64))
    st = mb.stats()
    assert st.n_closed_by_deadline >= 1 and st.n_closed_full == 0
    # a ticket is collected once
    with pytest.raises(pkg.simclusters_ann.SannError):
        mb.wait(tickets[0][0])
    # poll: eventually done
    t, out = mb.submit(reqs[0][0], reqs[0][1], reqs[0][2], now_ms=reqs[0][4], source_tweet_id=reqs[0][3])
    import time
    for _ in range(2000):
        if mb.poll(t):
            break
        time.sleep(0.001)
    else:
        raise AssertionError("poll never reported the request done")
    # an empty embedding is a request like any other (no candidates)
    ids, sc, msz = mb.get_tweet_candidates(np.empty(0, np.int32), np.empty(0), reqs[0][2], now_ms=co.now_ms)
    assert len(ids) == 0 and msz == 0
    # destroying a batcher with requests in flight runs them first
    t2, out2 = mb.submit(reqs[1][0], reqs[1][1], reqs[1][2], now_ms=reqs[1][4], source_tweet_id=reqs[1][3])
    mb.close()
    o_ids, _, _ = _oracle_answer(oracle, co, reqs[1])
    assert int(out2[2][0]) == len(o_ids) and np.array_equal(out2[0][:len(o_ids)], o_ids)
    index.close()


def test_embedding_buffer_limits_close_and_grow_the_open_batch(pkg, oracle):
    """A batch's embeddings live in one buffer that submitters copy into after releasing the queue's lock, so it may never
    move under them: a batch that would outgrow it is closed early, and a single request larger than the whole buffer makes
    the next (empty) batch grow.  max_batch = 4 gives a 256-entry buffer: 30-entry embeddings close batches by size, not by
    count; then a 700-entry embedding; then 24 threads race on the same tiny buffer.  Every answer is the oracle's."""
    co = pkg.corpus.make_corpus(30_000, 900, seed=31, index_cap=600)
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
    mb = pkg.MicroBatcher(index, max_batch=4, max_wait_us=2000, n_dispatchers=2)
    cfg = pkg.SimClustersANNConfig(maxNumResults=50, maxScanClusters=50)
    try:
        offs, cids, scs = pkg.corpus.make_queries(40, 900, seed=32, clusters_per_user=30)
        reqs = [(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], cfg, None, co.now_ms) for q in range(40)]
        big_c = np.arange(1, 701, dtype=np.int32)
        big_s = np.random.default_rng(3).random(700) + 0.01
        reqs.insert(17, (big_c, big_s, cfg, None, co.now_ms))
        tickets = [mb.submit(r[0], r[1], r[2], now_ms=r[4]) for r in reqs]
        for r, (t, out) in zip(reqs, tickets):
            mb.wait(t)
            o_ids, o_sc, o_m = _oracle_answer(oracle, co, r)
            n = int(out[2][0])
            assert n == len(o_ids) and int(out[3][0]) == o_m
            assert np.array_equal(out[0][:n], o_ids) and np.array_equal(out[1][:n].view(np.int64), o_sc.view(np.int64))
        got, errs = {}, []

        def caller(t):
            try:
                for i in range(10):
                    j = (t * 10 + i) % len(reqs)
                    got[(t, i)] = (j, mb.get_tweet_candidates(reqs[j][0], reqs[j][1], cfg, now_ms=co.now_ms))
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        ths = [threading.Thread(target=caller, args=(t,)) for t in range(24)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        assert not errs, errs[0]
        for (t, i), (j, ans) in got.items():
            o_ids, o_sc, o_m = _oracle_answer(oracle, co, reqs[j])
            assert np.array_equal(ans[0], o_ids) and np.array_equal(ans[1].view(np.int64), o_sc.view(np.int64)) and ans[2] == o_m
    finally:
        mb.close()
        index.close()
