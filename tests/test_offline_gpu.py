"""the offline all-users job (scio/bq_generation/sql/tweets_ann.sql) on the same kernels, against a
plain-Python restatement of the SQL (oracle.tweets_ann_sql) -- ids, order and all three score columns bit for bit --
and the SQL against the operator oracle on the one quantity both define (the dot product)."""
import numpy as np
import pytest

N_CL = 60


def _make(seed, n_tweets=3000, n_users=24):
    rng = np.random.default_rng(seed)
    tweets = {}
    for t in range(n_tweets):
        k = int(rng.integers(1, 9))
        cl = rng.choice(N_CL, size=k, replace=False) + 1
        tweets[int(1000 + 7 * t)] = [(int(c), float(np.exp(rng.normal(-2, 1)))) for c in cl]
    tweets[999] = [(3, 0.0)]  # a zero-norm tweet: HAVING norm > 0 drops it
    users = {}
    for u in range(n_users):
        k = int(rng.integers(1, 15))
        cl = rng.choice(N_CL + 5, size=k, replace=False) + 1  # a few clusters no tweet has
        users[u] = [(int(c), float(np.exp(rng.normal(0, 1)))) for c in cl]
    return users, tweets


def _lists(tweets):
    """cluster -> full posting list (tweetScore DESC, tweet id ASC) + per-posting full norms, CSR."""
    norm = {}
    for t, emb in tweets.items():
        total = 0.0
        for _c, s in sorted(emb):
            total = total + s * s
        norm[t] = total
    by_cluster = {}
    for t, emb in tweets.items():
        for c, s in emb:
            by_cluster.setdefault(c, []).append((t, s))
    cids = sorted(by_cluster)
    offs, tid, sc, nr = [0], [], [], []
    for c in cids:
        for t, s in sorted(by_cluster[c], key=lambda ts: (-ts[1], ts[0])):
            tid.append(t); sc.append(s); nr.append(norm[t])
        offs.append(len(tid))
    return (np.array(cids, np.int32), np.array(offs, np.int64), np.array(tid, np.int64), np.array(sc), np.array(nr))


def test_sql_and_operator_oracles_agree_on_the_dot_product(oracle):
    """The cross-check the SQL buys (CPU only): every (user, tweet) row of the job carries the dot product the operator
    oracle accumulates for the same user over the same top-N x top-M postings."""
    users, tweets = _make(5)
    cids, offs, tid, sc, _nr = _lists(tweets)
    N, M = 6, 25
    res = oracle.tweets_ann_sql(users, tweets, N, M, 10 ** 6)

    class Cfg:
        maxNumResults, minScore, candidateEmbeddingType = 1000, -1e300, 0
        maxTopTweetsPerCluster, maxScanClusters, maxTweetCandidateAgeHours, minTweetCandidateAgeHours, annAlgorithm = M, N, 175200, 0, 1
    for u, emb in users.items():
        e_c = np.array([c for c, _ in emb], np.int32)
        e_s = np.array([s for _, s in emb])
        o_ids, o_sc, _ = oracle.sann_query(e_c, e_s, None, Cfg, 1_700_000_000_000, cids, offs, tid, sc)
        op = dict(zip(o_ids.tolist(), o_sc.tolist()))
        rows = res[u]
        assert {r[0] for r in rows} | {999} >= set(op) - {999} and len(rows) >= len(op) - 1
        for t, d, _cos, _lc in rows:
            assert op[t] == d


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 8, 32])
def test_offline_job_matches_the_sql(pkg, oracle, P):
    users, tweets = _make(11)
    cids, offs, tid, sc, nr = _lists(tweets)
    index = pkg.ClusterTweetIndex(cids, offs, tid, sc, n_partitions=P, tweet_norms=nr)
    SA = pkg.ScoringAlgorithm
    uids = sorted(users)
    e_offs = np.zeros(len(uids) + 1, np.int64)
    e_offs[1:] = np.cumsum([len(users[u]) for u in uids])
    e_c = np.array([c for u in uids for c, _ in users[u]], np.int32)
    e_s = np.array([s for u in uids for _, s in users[u]])
    for N, M, K in ((8, 40, 50), (25, 100, 150), (3, 5, 1000)):
        want = oracle.tweets_ann_sql(users, tweets, N, M, K)
        got = {}
        for alg in (SA.OfflineLogCosineSimilarity, SA.OfflineCosineSimilarity, SA.DotProduct):
            k = K if alg == SA.OfflineLogCosineSimilarity else 1000
            cfg = pkg.SimClustersANNConfig(maxNumResults=k, minScore=-1e300, maxTopTweetsPerCluster=M, maxScanClusters=N,
                                           maxTweetCandidateAgeHours=175200, annAlgorithm=alg)
            qb = pkg.QueryBatch(index, e_offs, e_c, e_s, cfg, now_ms=1_700_000_000_000)
            qb.run(); qb.finish()
            got[alg] = qb.results()
            qb.close()
        ids, lc, cnt, _ = got[SA.OfflineLogCosineSimilarity]
        for qi, u in enumerate(uids):
            rows = want[u]
            assert cnt[qi] == len(rows), (N, M, K, u, cnt[qi], len(rows))
            assert ids[qi, :cnt[qi]].tolist() == [r[0] for r in rows]
            assert np.array_equal(lc[qi, :cnt[qi]].view(np.int64), np.array([r[3] for r in rows]).view(np.int64))
            # the other two columns of the job's rows, looked up in the exhaustive (k = 1000) runs
            cos = dict(zip(got[SA.OfflineCosineSimilarity][0][qi, :got[SA.OfflineCosineSimilarity][2][qi]].tolist(),
                           got[SA.OfflineCosineSimilarity][1][qi].tolist()))
            dot = dict(zip(got[SA.DotProduct][0][qi, :got[SA.DotProduct][2][qi]].tolist(), got[SA.DotProduct][1][qi].tolist()))
            if max(len(cos), len(dot)) < 1000:  # (complete candidate sets)
                for t, d, c, _l in rows:
                    assert cos[t] == c and dot[t] == d
    index.close()


@pytest.mark.gpu
def test_offline_scores_need_an_index_with_norms(pkg):
    users, tweets = _make(3, 200, 2)
    cids, offs, tid, sc, nr = _lists(tweets)
    index = pkg.ClusterTweetIndex(cids, offs, tid, sc)
    cfg = pkg.SimClustersANNConfig(annAlgorithm=pkg.ScoringAlgorithm.OfflineLogCosineSimilarity)
    with pytest.raises(pkg.simclusters_ann.SannError) as e:
        pkg.QueryBatch(index, np.array([0, 1], np.int64), np.array([1], np.int32), np.array([1.0]), cfg, now_ms=0)
    assert e.value.code == 1 and "norms" in str(e.value)
    index.close()
