"""The boundary call as a front end uses it: batch objects that are reset per request (sann_batch_reset), queries
prepared on the device (sann_prep.hip) against the host preparation of the same arithmetic, and the pooled one-call
form sann_get_tweet_candidates from several threads -- the seam of
simclusters-ann/.../candidate_source/SimClustersANNCandidateSource.scala:66-95."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(pkg):
    co = pkg.corpus.make_corpus(30000, 1500, seed=21, index_cap=400)
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=16)
    return co, index


def _queries(pkg, n, seed, per_user=50):
    return pkg.corpus.make_queries(n, 1500, seed=seed, clusters_per_user=per_user)


def _fresh(pkg, index, co, q, cfg, **kw):
    qb = pkg.QueryBatch(index, *q, cfg, now_ms=co.now_ms, **kw)
    qb.run()
    qb.finish()
    out = qb.results()
    qb.close()
    return out


def _same(a, b):
    ids_a, sc_a, cnt_a, msz_a = a
    ids_b, sc_b, cnt_b, msz_b = b
    assert np.array_equal(cnt_a, cnt_b) and np.array_equal(msz_a, msz_b)
    for q in range(len(cnt_a)):
        n = cnt_a[q]
        assert np.array_equal(ids_a[q, :n], ids_b[q, :n]), q
        assert np.array_equal(sc_a[q, :n].view(np.int64), sc_b[q, :n].view(np.int64)), q


def test_consecutive_resets_equal_fresh_batches(pkg, small):
    """Two (and more) consecutive resets of one batch object -- different batch sizes, k, M, algorithms, growing and
    shrinking -- give what a freshly created batch gives for the same queries."""
    co, index = small
    SA = pkg.ScoringAlgorithm
    shapes = [(24, 31, dict(maxNumResults=400, maxTopTweetsPerCluster=300)),
              (7, 32, dict(maxNumResults=10, maxTopTweetsPerCluster=50, annAlgorithm=SA.LogCosineSimilarity)),
              (64, 33, dict(maxNumResults=1000, maxTopTweetsPerCluster=400, annAlgorithm=SA.DotProduct)),
              (24, 31, dict(maxNumResults=400, maxTopTweetsPerCluster=300)),
              (1, 34, dict(maxNumResults=200, maxScanClusters=5)),
              (0, 35, dict()),
              (40, 36, dict(maxNumResults=37, minScore=0.1, annAlgorithm=SA.CosineSimilarityNoSourceEmbeddingNormalization))]
    first = _queries(pkg, 3, 30)
    qb = pkg.QueryBatch(index, *first, pkg.SimClustersANNConfig(), now_ms=co.now_ms)
    for nq, seed, kw in shapes:
        q = _queries(pkg, nq, seed) if nq else (np.zeros(1, np.int64), np.empty(0, np.int32), np.empty(0))
        cfg = pkg.SimClustersANNConfig(**kw)
        qb.reset(*q, cfg, now_ms=co.now_ms)
        qb.run()
        qb.finish()
        got = qb.results()
        if nq:
            _same(got, _fresh(pkg, index, co, q, cfg))
            assert qb.stats().postings_scanned > 0 and qb.stats().max_unit_postings > 0
        # the same batch replayed without a reset still gives the same answer
        qb.run()
        qb.finish()
        if nq:
            _same(qb.results(), got)
    qb.close()


def _awkward_queries(rng, nq):
    """Embeddings the constructor has work to do on: non-positive scores, exactly tied scores, clusters missing from
    the index, empty embeddings, 1 .. 300 entries."""
    offs, cids, scs = [0], [], []
    for q in range(nq):
        n = int(rng.choice([0, 1, 2, 17, 50, 64, 65, 130, 300]))
        c = rng.choice(np.arange(1, 2200), size=n, replace=False).astype(np.int32)  # ids above 1500 are not in the index
        s = np.round(np.exp(rng.normal(0, 1, n)), 1)  # one decimal: many exact ties
        s[rng.random(n) < 0.1] = 0.0
        s[rng.random(n) < 0.05] = -1.0
        cids.append(c)
        scs.append(s)
        offs.append(offs[-1] + n)
    return np.array(offs, np.int64), np.concatenate(cids), np.concatenate(scs)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_device_preparation_equals_host_preparation(pkg, oracle, small, variant, monkeypatch):
    co, index = small
    rng = np.random.default_rng(50 + variant)
    nq = 48
    q = _awkward_queries(rng, nq)
    cfgs = [pkg.SimClustersANNConfig(maxNumResults=int(rng.choice([1, 50, 400, 1000])), minScore=float(rng.choice([0.0, 0.05])),
                                     maxTopTweetsPerCluster=int(rng.choice([1, 100, 400])),
                                     maxScanClusters=int(rng.choice([-1, 0, 1, 10, 50, 200])),
                                     maxTweetCandidateAgeHours=int(rng.choice([12, 24, 175200])),
                                     minTweetCandidateAgeHours=int(rng.choice([0, 2])),
                                     annAlgorithm=pkg.ScoringAlgorithm(int(rng.integers(1, 4 if variant == 3 else 5))))
            for _ in range(nq)]
    sources = [int(co.tweet_ids[rng.integers(0, len(co.tweet_ids))]) if i % 3 == 0 else None for i in range(nq)]
    src = np.array([0 if s is None else s for s in sources], np.int64)
    has = np.array([0 if s is None else 1 for s in sources], np.uint8)
    kw = dict(variant=pkg.Variant(variant), source_tweet_ids=src, has_source_tweet=has)
    dev = _fresh(pkg, index, co, q, cfgs, **kw)
    monkeypatch.setenv("SANN_HOST_PREP", "1")
    host = _fresh(pkg, index, co, q, cfgs, **kw)
    monkeypatch.delenv("SANN_HOST_PREP")
    _same(dev, host)
    # and both equal the oracle
    offs, cids, scs = q
    ids, scores, counts, msz = dev
    for i in range(nq):
        o_ids, o_sc, o_msz = oracle.sann_query(cids[offs[i]:offs[i + 1]], scs[offs[i]:offs[i + 1]], sources[i], cfgs[i], co.now_ms,
                                               co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, variant=variant)
        assert counts[i] == len(o_ids) and msz[i] == o_msz, i
        assert np.array_equal(ids[i, :counts[i]], o_ids) and np.array_equal(scores[i, :counts[i]].view(np.int64), o_sc.view(np.int64))


def test_explicit_scan_keys_prepared_on_the_device(pkg, oracle, small, monkeypatch):
    """clusterTweetsMap keys in a caller-given order, including clusters the embedding lacks (experimental: weight 0)
    and more keys than one chunk of the preparation kernel (300 > 256)."""
    co, index = small
    rng = np.random.default_rng(77)
    nq = 12
    q = _queries(pkg, nq, 78)
    offs, cids, scs = q
    keys = [rng.permutation(np.concatenate([cids[offs[i]:offs[i + 1]], rng.integers(1, 1500, int(rng.choice([0, 5, 250])))]))
            .astype(np.int32) for i in range(nq)]
    keys = [np.array(list(dict.fromkeys(k.tolist())), np.int32) for k in keys]  # a Map's keys are unique
    so = np.zeros(nq + 1, np.int64)
    so[1:] = np.cumsum([len(k) for k in keys])
    sc = np.concatenate(keys)
    cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=200, annAlgorithm=pkg.ScoringAlgorithm.DotProduct)
    for variant in (0, 2):
        kw = dict(variant=pkg.Variant(variant), scan_offsets=so, scan_cluster_ids=sc)
        dev = _fresh(pkg, index, co, q, cfg, **kw)
        monkeypatch.setenv("SANN_HOST_PREP", "1")
        host = _fresh(pkg, index, co, q, cfg, **kw)
        monkeypatch.delenv("SANN_HOST_PREP")
        _same(dev, host)
        ids, scores, counts, msz = dev
        for i in range(nq):
            o_ids, o_sc, o_msz = oracle.sann_query(cids[offs[i]:offs[i + 1]], scs[offs[i]:offs[i + 1]], None, cfg, co.now_ms,
                                                   co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, variant=variant,
                                                   scan_order=keys[i])
            assert counts[i] == len(o_ids) and msz[i] == o_msz, (variant, i)
            assert np.array_equal(ids[i, :counts[i]], o_ids)


def test_long_embeddings_take_the_host_preparation(pkg, oracle, small):
    """More than 1024 entries in one embedding: the batch falls back to the host preparation, same answers."""
    co, index = small
    rng = np.random.default_rng(91)
    n = 1300
    offs = np.array([0, n, n + 40], np.int64)
    cids = np.concatenate([rng.permutation(np.arange(1, 1501))[:n], rng.permutation(np.arange(1, 1501))[:40]]).astype(np.int32)
    scs = np.exp(rng.normal(0, 1, n + 40))
    cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=100, maxScanClusters=60)
    ids, scores, counts, msz = _fresh(pkg, index, co, (offs, cids, scs), cfg)
    for i in range(2):
        o_ids, o_sc, o_msz = oracle.sann_query(cids[offs[i]:offs[i + 1]], scs[offs[i]:offs[i + 1]], None, cfg, co.now_ms,
                                               co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
        assert counts[i] == len(o_ids) and msz[i] == o_msz
        assert np.array_equal(ids[i, :counts[i]], o_ids) and np.array_equal(scores[i, :counts[i]].view(np.int64), o_sc.view(np.int64))


def test_get_tweet_candidates_from_concurrent_callers(pkg, oracle, small):
    """The one-call form from 4 threads at once (Finagle workers): pooled batch objects, own streams; every answer
    equals the oracle's; response buffers in pinned memory are reused across calls."""
    co, index = small
    sa = pkg.simclusters_ann
    cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=300)
    errors = []

    def worker(t):
        try:
            out = (sa.pinned_array((32, 400), np.int64), sa.pinned_array((32, 400), np.float64),
                   sa.pinned_array((32,), np.int32), sa.pinned_array((32,), np.int32))
            for it in range(6):
                nq = [32, 5, 17][it % 3]
                offs, cids, scs = _queries(pkg, nq, 1000 + 10 * t + it)
                ids, scores, counts, msz = sa.get_tweet_candidates(index, offs, cids, scs, cfg, now_ms=co.now_ms, out=out)
                for i in range(nq):
                    o_ids, o_sc, o_msz = oracle.sann_query(cids[offs[i]:offs[i + 1]], scs[offs[i]:offs[i + 1]], None, cfg, co.now_ms,
                                                           co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
                    assert counts[i] == len(o_ids) and msz[i] == o_msz
                    assert np.array_equal(ids[i, :counts[i]], o_ids)
                    assert np.array_equal(scores[i, :counts[i]].view(np.int64), o_sc.view(np.int64))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_get_tweet_candidates_reports_bad_arguments(pkg, small):
    co, index = small
    sa = pkg.simclusters_ann
    offs, cids, scs = _queries(pkg, 4, 5)
    cfgs = [pkg.SimClustersANNConfig()] * 3  # neither 1 nor nq
    with pytest.raises(sa.SannError) as e:
        sa.get_tweet_candidates(index, offs, cids, scs, cfgs, now_ms=co.now_ms)
    assert e.value.code == 1
    # the pool survives an argument error
    sa.get_tweet_candidates(index, offs, cids, scs, pkg.SimClustersANNConfig(), now_ms=co.now_ms)
