"""the cluster -> top tweets provider on the device (sann_index_build_from_postings: decay
to now, keep > 0, sort descending, take, partition) against the oracle's restatement of
TopKTweetsForClusterReadableStore.scala:51-71,211-229 -- list by list, bit for bit -- and queries on the index it builds."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HALF_LIFE = 8 * 3600 * 1000
NOW = 1_700_000_000_000


def _raw_store(seed, n_clusters=300, max_len=1900):
    rng = np.random.default_rng(seed)
    cids = np.sort(rng.choice(np.arange(1, 5000), n_clusters, replace=False)).astype(np.int32)
    lens = rng.integers(0, max_len, n_clusters)
    lens[:3] = [0, 1, max_len]
    offs = np.zeros(n_clusters + 1, np.int64)
    offs[1:] = np.cumsum(lens)
    n = int(offs[-1])
    ms = NOW - rng.integers(-3600_000, 3 * 24 * 3600_000, n)  # up to 3 days old, a few written "in the future"
    tid = ((ms - 1288834974657) << 22) | rng.integers(0, 1 << 22, n)
    for i in range(n_clusters):  # unique inside a list
        b, e = offs[i], offs[i + 1]
        tid[b:e] = np.unique(tid[b:e])[: e - b] if len(np.unique(tid[b:e])) == e - b else np.arange(b, e) + (1 << 40)
    vals = np.exp(rng.normal(-2, 1.5, n))
    vals[rng.random(n) < 0.03] = 0.0
    vals[rng.random(n) < 0.02] *= -1.0
    vals[rng.random(n) < 0.05] = 0.25  # exact ties
    scaled = ms.astype(np.float64) * math.log(2.0) / HALF_LIFE
    return cids, offs, tid.astype(np.int64), vals, scaled


@pytest.mark.parametrize("decay", [True, False])
@pytest.mark.parametrize("P,max_results", [(1, 2000), (32, 800), (8, 5)])
def test_built_lists_equal_the_store(pkg, oracle, P, max_results, decay):
    cids, offs, tid, vals, scaled = _raw_store(7)
    now_scaled = NOW * math.log(2.0) / HALF_LIFE
    index = pkg.ClusterTweetIndex.from_raw_postings(cids, offs, tid, vals, scaled if decay else None, now_ms=NOW,
                                                    half_life_ms=HALF_LIFE, max_results=max_results, n_partitions=P)
    total = 0
    for i, c in enumerate(cids):
        b, e = offs[i], offs[i + 1]
        w_i, w_s = oracle.store_list(tid[b:e], vals[b:e], scaled[b:e] if decay else None, now_scaled, max_results)
        g_i, g_s, g_r = index.get_list(int(c))
        assert np.array_equal(g_i, w_i), c
        assert np.array_equal(g_s.view(np.int64), w_s.view(np.int64)), c
        assert np.array_equal(g_r, np.arange(len(w_i)))
        total += len(w_i)
    info = index.info()
    assert info.n_postings == total and info.n_clusters == len(cids)
    index.close()


def test_queries_on_a_provider_built_index(pkg, oracle):
    """End to end: raw store -> device provider -> operator, against oracle lists -> oracle operator."""
    cids, offs, tid, vals, scaled = _raw_store(9, n_clusters=400, max_len=1200)
    now_scaled = NOW * math.log(2.0) / HALF_LIFE
    index = pkg.ClusterTweetIndex.from_raw_postings(cids, offs, tid, vals, scaled, now_ms=NOW, half_life_ms=HALF_LIFE,
                                                    max_results=1000, n_partitions=16)
    l_off, l_t, l_s = [0], [], []
    for i in range(len(cids)):
        b, e = offs[i], offs[i + 1]
        w_i, w_s = oracle.store_list(tid[b:e], vals[b:e], scaled[b:e], now_scaled, 1000)
        l_t.append(w_i); l_s.append(w_s); l_off.append(l_off[-1] + len(w_i))
    l_off, l_t, l_s = np.array(l_off, np.int64), np.concatenate(l_t), np.concatenate(l_s)
    rng = np.random.default_rng(10)
    nq = 16
    e_offs = np.arange(0, 30 * nq + 1, 30, dtype=np.int64)
    e_c = np.concatenate([rng.choice(cids, 30, replace=False) for _ in range(nq)]).astype(np.int32)
    e_s = np.exp(rng.normal(0, 1, 30 * nq))
    for alg in (2, 3):
        cfg = pkg.SimClustersANNConfig(maxNumResults=200, maxTopTweetsPerCluster=400, maxScanClusters=25,
                                       maxTweetCandidateAgeHours=48, annAlgorithm=pkg.ScoringAlgorithm(alg))
        ids, scores, counts, msz = pkg.simclusters_ann.get_tweet_candidates(index, e_offs, e_c, e_s, cfg, now_ms=NOW)
        for q in range(nq):
            o_i, o_s, o_m = oracle.sann_query(e_c[e_offs[q]:e_offs[q + 1]], e_s[e_offs[q]:e_offs[q + 1]], None, cfg, NOW, cids, l_off, l_t, l_s)
            assert counts[q] == len(o_i) and msz[q] == o_m
            assert np.array_equal(ids[q, :counts[q]], o_i) and np.array_equal(scores[q, :counts[q]].view(np.int64), o_s.view(np.int64))
    index.close()


def test_provider_refuses_bad_input(pkg):
    E = pkg.simclusters_ann.SannError
    cids = np.array([1], np.int32)
    with pytest.raises(E) as e:  # more raw entries than a cluster's store can hold
        pkg.ClusterTweetIndex.from_raw_postings(cids, np.array([0, 5000], np.int64), np.arange(5000, dtype=np.int64), np.ones(5000), None, now_ms=NOW)
    assert e.value.code == 4
    with pytest.raises(E) as e:  # a tweet twice in one list
        pkg.ClusterTweetIndex.from_raw_postings(cids, np.array([0, 2], np.int64), np.array([7, 7], np.int64), np.ones(2), None, now_ms=NOW)
    assert e.value.code == 1
