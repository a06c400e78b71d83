"""Regression tests for the fast unit kernel's "every answer is exact" claim (round-1 review):

  * the fp32 pre-filter against the exact fp64 normalisation (ApproximateCosineSimilarity.scala:111-119) over a
    wide sweep of posting scores, including LogCosine's regime where `1 + nsq` rounds in fp64;
  * corpora with posting scores down to 1e-12 (decayed production scores are arbitrarily small:
    summingbird/common/Configs.scala:38, 8 h half-life), for every variant, bit-exact against the oracle;
  * quantised / tied scores with unit sizes in (keep_all, 160], where a non-zero cut can keep every candidate;
  * sann_index_build refusing a tweet id that appears twice in one list.
"""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _exact(alg, s, w, l2, ln):
    dot = 0.0 + s * w
    nsq = 0.0 + s * s
    with np.errstate(divide="ignore", invalid="ignore"):
        if alg == 1:
            return dot
        if alg == 2:
            return dot / l2 / np.sqrt(nsq)
        if alg == 3:
            return dot / ln / np.log(1 + nsq)
        return dot / np.sqrt(nsq)


@pytest.mark.parametrize("alg", [1, 2, 3, 4])
def test_prefilter_error_bound(pkg, alg):
    """max |approx / exact - 1| over posting scores 1e-12 .. 1e3 and weights 1e-3 .. 1e3 stays below the bound the
    kernel's cut assumes (APPROX_EPS); candidates whose exact score is +inf are flagged `forced`, never bounded."""
    lib = pkg.load_library()
    rng = np.random.default_rng(100 + alg)
    n = 400_000
    s = np.exp(rng.uniform(math.log(1e-12), math.log(1e3), n))
    # the awkward region for LogCosine: 1 + s^2 a few ulps above 1
    s[: n // 4] = np.sqrt(rng.uniform(0.3, 40.0, n // 4) * 2.0 ** -52)
    w = np.exp(rng.uniform(math.log(1e-3), math.log(1e3), n))
    l2, ln = 3.7, math.log(3.7 * 3.7 + 1)
    out = np.zeros(n, np.float32)
    forced = np.zeros(n, np.uint8)
    eps = C.c_double()
    rc = lib.sann_debug_approx(0, alg, n, s.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), l2, ln,
                               out.ctypes.data_as(C.c_void_p), forced.ctypes.data_as(C.c_void_p), C.byref(eps))
    assert rc == 0, lib.sann_last_error()
    exact = _exact(alg, s, w, l2, ln)
    inf = ~np.isfinite(exact)
    assert np.array_equal(forced.astype(bool), inf)  # forced <=> the exact score is +inf (1 + nsq == 1)
    if alg != 3:
        assert not inf.any()
    # what the kernel trusts: finite approximations in (1e-30, 1e30) with nsq in fp32 range; the rest goes to the general path
    n32 = (s.astype(np.float32) ** 2).astype(np.float32)
    trusted = ~inf & (out > 1e-30) & (out < 1e30) & (n32 > 1e-30) & (n32 < 1e30)
    assert trusted.sum() > n // 3
    rel = np.abs(out[trusted].astype(np.float64) / exact[trusted] - 1.0)
    assert rel.max() < eps.value, (alg, rel.max(), eps.value)
    assert rel.max() < 1.5e-6  # the actual error, well inside the 4e-6 the cut assumes


def test_wave_sort_network(pkg):
    """The in-register 64-lane bitonic network of the cut (DPP quad permutes / row shifts, v_permlane16/32_swap)."""
    lib = pkg.load_library()
    rng = np.random.default_rng(9)
    v = rng.integers(0, 2 ** 32, size=(500, 64), dtype=np.uint64).astype(np.uint32)
    v[:50] = rng.integers(0, 5, size=(50, 64)).astype(np.uint32)  # heavy ties
    v[50] = 0
    v[51] = np.arange(64, dtype=np.uint32)
    want = -np.sort(-v.astype(np.int64), axis=1)
    got = np.ascontiguousarray(v.copy())
    assert lib.sann_debug_wave_sort(0, got.shape[0], got.ctypes.data_as(C.c_void_p)) == 0, lib.sann_last_error()
    assert np.array_equal(got.astype(np.int64), want)


def _tiny_score_corpus(pkg, seed, lo_exp):
    """The small test corpus with every posting score scaled by 10^-U(0, lo_exp) (decay), lists re-sorted by score
    descending as the store would return them."""
    co = pkg.corpus.make_corpus(30000, 1500, seed=seed, index_cap=400)
    rng = np.random.default_rng(seed + 1)
    sc = co.scores * 10.0 ** (-rng.uniform(0.0, lo_exp, len(co.scores)))
    # a few exact zeros of 1 + s^2 - 1: scores below 1e-8 make the LogCosine score +inf
    sc[rng.integers(0, len(sc), 200)] = 10.0 ** (-rng.uniform(8.0, 12.0, 200))
    tid = co.tweet_ids.copy()
    for i in range(len(co.cluster_ids)):
        b, e = co.list_offsets[i], co.list_offsets[i + 1]
        order = np.lexsort((tid[b:e], -sc[b:e]))
        tid[b:e], sc[b:e] = tid[b:e][order], sc[b:e][order]
    co.tweet_ids, co.scores = tid, sc
    return co


def _check(pkg, oracle, co, offs, cids, scs, cfg, variant, P):
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=P)
    qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=co.now_ms, variant=pkg.Variant(variant))
    qb.run()
    qb.finish()
    ids, scores, counts, msz = qb.results()
    st = qb.stats()
    qb.close()
    index.close()
    for q in range(len(offs) - 1):
        o_ids, o_sc, o_msz = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg, co.now_ms,
                                               co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, variant=variant)
        assert counts[q] == len(o_ids) and msz[q] == o_msz, (q, counts[q], len(o_ids), msz[q], o_msz)
        assert np.array_equal(ids[q, :counts[q]], o_ids), f"query {q}: ids differ"
        assert np.array_equal(scores[q, :counts[q]].view(np.int64), o_sc.view(np.int64)), f"query {q}: scores differ"
    return st


@pytest.mark.parametrize("P", [1, 8, 32])
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("lo_exp", [3.0, 6.0, 9.0])
def test_tiny_posting_scores_logcosine(pkg, oracle, variant, P, lo_exp):
    """LogCosine (and the legacy log form) with posting scores spanning 1e-3 .. 1e-9 and a sprinkle below 1e-8
    (+inf scores): the fast path must neither cut a candidate by a wrong bound nor report a wrong theta."""
    co = _tiny_score_corpus(pkg, 31, lo_exp)
    offs, cids, scs = pkg.corpus.make_queries(16, 1500, seed=32, clusters_per_user=50)
    for k in (10, 400):
        cfg = pkg.SimClustersANNConfig(maxNumResults=k, maxTopTweetsPerCluster=300, maxScanClusters=50,
                                       maxTweetCandidateAgeHours=175200, annAlgorithm=pkg.ScoringAlgorithm(3))
        _check(pkg, oracle, co, offs, cids, scs, cfg, variant, P)


@pytest.mark.parametrize("alg", [1, 2, 4])
def test_tiny_posting_scores_other_algorithms(pkg, oracle, alg):
    co = _tiny_score_corpus(pkg, 41, 9.0)
    offs, cids, scs = pkg.corpus.make_queries(16, 1500, seed=42, clusters_per_user=50)
    cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=300, maxScanClusters=50,
                                   maxTweetCandidateAgeHours=175200, annAlgorithm=pkg.ScoringAlgorithm(alg))
    for P in (4, 32):
        _check(pkg, oracle, co, offs, cids, scs, cfg, 0, P)


@pytest.mark.parametrize("levels", [1, 2, 5])
@pytest.mark.parametrize("alg", [1, 2, 3])
def test_quantised_scores_every_candidate_above_a_nonzero_cut(pkg, oracle, alg, levels):
    """Near-tied scores with unit sizes in (keep_all, 160]: the need-th key then sits in the lowest occupied
    histogram digit, the cut tau is non-zero and yet every live candidate is above it.  Such a unit withholds
    nothing and must emit everything (round 1 dropped the candidates below tau * (1 + 2 eps) without saying so)."""
    rng = np.random.default_rng(7 + levels)
    n_clusters, per_list = 40, 150
    cids = np.arange(1, n_clusters + 1, dtype=np.int32)
    tid = (np.arange(n_clusters * per_list, dtype=np.int64) * 7919 + 12345) << 22
    base = 0.25
    # `levels` distinct scores a few fp32 ulps apart: their approximate keys differ only in the last bits
    sc = base * (1.0 + rng.integers(0, levels, n_clusters * per_list) * 3e-7)
    offs_l = np.arange(0, n_clusters * per_list + 1, per_list, dtype=np.int64)
    for i in range(n_clusters):
        b, e = offs_l[i], offs_l[i + 1]
        order = np.lexsort((tid[b:e], -sc[b:e]))
        tid[b:e], sc[b:e] = tid[b:e][order], sc[b:e][order]

    class Co:
        pass
    co = Co()
    co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, co.now_ms = cids, offs_l, tid, sc, 1_700_000_000_000
    # queries of ONE cluster with weight 1: every candidate of a unit has (nearly) the same score; k large enough
    # that a unit must offer more than keep_all of its ~130 candidates
    nq = 12
    e_offs = np.arange(nq + 1, dtype=np.int64)
    e_cids = cids[:nq].copy()
    e_scs = np.ones(nq)
    # P = 1, k = 80: kl = 80, keep_all = 136 < 150 live candidates <= 160: the radix cut runs with need = 80
    for k, P in ((30, 1), (80, 1), (100, 1), (140, 1), (1000, 1), (160, 2)):
        cfg = pkg.SimClustersANNConfig(maxNumResults=k, maxTopTweetsPerCluster=per_list, maxScanClusters=50,
                                       maxTweetCandidateAgeHours=175200, annAlgorithm=pkg.ScoringAlgorithm(alg))
        _check(pkg, oracle, co, e_offs, e_cids, e_scs, cfg, 0, P)


def test_index_build_refuses_a_repeated_tweet_id(pkg):
    """a tweet id twice in one list reached the duplicate resolver with equal sequence numbers.  The store
    cannot produce it (Map keys); the builder now says so instead of trusting it."""
    cids = np.array([1, 2], np.int32)
    offs = np.array([0, 3, 5], np.int64)
    tids = np.array([10, 11, 10, 10, 12], np.int64)
    scs = np.array([3.0, 2.0, 1.0, 1.0, 0.5])
    with pytest.raises(pkg.simclusters_ann.SannError) as e:
        pkg.ClusterTweetIndex(cids, offs, tids, scs)
    assert e.value.code == 1 and "twice" in str(e.value)
    # the same id in two different lists is the normal multi-cluster tweet
    tids[2] = 13
    pkg.ClusterTweetIndex(cids, offs, tids, scs).close()
