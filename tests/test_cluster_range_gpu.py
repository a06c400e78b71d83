"""The partitioning north_star names -- cluster-id ranges -- with the exact merge (sharding.ClusterRangeDeployment): every
shard sends the scanned lists' top-M prefixes to the GPU hash(tweetId) % N names, which runs the ordinary pipeline on its
tweets' postings; the merged answer must equal the unsharded one bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(pkg):
    co = pkg.corpus.make_corpus(60_000, 3000, seed=77)
    offs, cids, scs = pkg.corpus.make_queries(24, 3000, seed=78, clusters_per_user=40)
    return co, offs, cids, scs


@pytest.mark.parametrize("n_shards", [2, 3, 8])
@pytest.mark.parametrize("alg", ["CosineSimilarity", "LogCosineSimilarity", "DotProduct"])
def test_cluster_range_shards_with_exact_merge_equal_the_unsharded_index(pkg, small, n_shards, alg):
    co, offs, cids, scs = small
    cfg = pkg.SimClustersANNConfig(maxNumResults=200, maxTopTweetsPerCluster=150, maxScanClusters=30,
                                   annAlgorithm=getattr(pkg.ScoringAlgorithm, alg))
    full = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
    qb = pkg.QueryBatch(full, offs, cids, scs, cfg, now_ms=co.now_ms)
    qb.run(); qb.finish()
    w_ids, w_sc, w_cnt, w_msz = qb.results()
    qb.close(); full.close()
    dep = pkg.sharding.ClusterRangeDeployment(pkg, co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_shards, n_partitions=8)
    try:
        ids, sc, cnt, msz, stats = dep.get_tweet_candidates(offs, cids, scs, cfg, now_ms=co.now_ms)
    finally:
        dep.close()
    assert np.array_equal(cnt, w_cnt) and np.array_equal(msz, w_msz)
    for q in range(len(cnt)):
        assert np.array_equal(ids[q, :cnt[q]], w_ids[q, :cnt[q]]), q
        assert np.array_equal(sc[q, :cnt[q]].view(np.int64), w_sc[q, :cnt[q]].view(np.int64)), q
    # the ranges are balanced by posting mass and every scanned posting is regrouped exactly once
    assert stats["postings_regrouped"] > 0 and 0 < stats["bytes_moved"] <= stats["postings_regrouped"] * 16
    assert abs(stats["bytes_moved"] / (stats["postings_regrouped"] * 16) - (n_shards - 1) / n_shards) < 0.1


def test_range_bounds_balance_posting_mass(pkg):
    lens = np.array([1000, 10, 10, 10, 500, 500, 5, 5, 960], np.int64)
    b = pkg.sharding.cluster_range_bounds(lens, 3)
    assert b[0] == 0 and b[-1] == len(lens) and np.all(np.diff(b) >= 0)
    mass = [int(lens[b[g]:b[g + 1]].sum()) for g in range(3)]
    assert max(mass) <= 1.6 * sum(mass) / 3
