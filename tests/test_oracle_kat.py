"""The CPU oracle against the hand-derived golden vectors (tests/golden/sann_kat.json).
No GPU.  These pin the oracle before anything is compared with it."""
import json
import math
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "sann_kat.json")))


class Cfg:
    def __init__(self, d):
        self.__dict__.update(d)


def lists_to_csr(lists):
    cids = sorted(int(c) for c in lists)
    offs, tids, scs = [0], [], []
    for c in cids:
        for t, s in lists[str(c)]:
            tids.append(t)
            scs.append(s)
        offs.append(len(tids))
    return (np.array(cids, np.int32), np.array(offs, np.int64), np.array(tids, np.int64), np.array(scs, np.float64))


def ulps(a, b):
    if a == b:
        return 0
    ia = struct.unpack("<q", struct.pack("<d", a))[0]
    ib = struct.unpack("<q", struct.pack("<d", b))[0]
    return abs(ia - ib)


@pytest.mark.parametrize("case", KAT["sann"], ids=[c["name"] for c in KAT["sann"]])
def test_sann_kat(oracle, case):
    cids, offs, tids, scs = lists_to_csr(case["lists"])
    emb = case["emb"]
    ids, scores, msz = oracle.sann_query([e[0] for e in emb], [e[1] for e in emb], case["source"], Cfg(case["config"]),
                                         case["now_ms"], cids, offs, tids, scs, variant=case["variant"],
                                         scan_order=case.get("scan_keys"))
    exp = case["expect"]
    assert msz == case["map_size"]
    assert [int(i) for i in ids] == [e[0] for e in exp]
    for got, e in zip(scores, exp):
        assert ulps(float(got), float.fromhex(e[1])) <= case["ulp"], (got, float.fromhex(e[1]))


@pytest.mark.parametrize("case", KAT["pairs"], ids=[c["name"] for c in KAT["pairs"]])
def test_pair_kat(oracle, case):
    got = oracle.pair_score(case["alg"], [x[0] for x in case["a"]], [x[1] for x in case["a"]],
                            [x[0] for x in case["b"]], [x[1] for x in case["b"]])
    assert ulps(got, float.fromhex(case["expect"])) <= case["ulp"]


def test_snowflake(oracle):
    for s in KAT["snowflake"]:
        assert oracle.lib().oracle_snowflake_first_id_for(s["ms"]) == s["id"]
    # layout pinned in-repo: ms = 1288834974657 + (id >> 22)  (BQGenerationUtil.scala:150-153)
    i = oracle.lib().oracle_snowflake_first_id_for(1_700_000_000_000)
    assert 1288834974657 + (i >> 22) == 1_700_000_000_000


def test_strict_log_within_one_ulp_of_libm(oracle):
    rng = np.random.default_rng(3)
    xs = np.concatenate([np.exp(rng.normal(0, 4, 20000)), 1 + rng.uniform(-1e-7, 1e-7, 5000), [1.0, 2.0, 26.0, 1e-320]])
    L = oracle.lib()
    for x in xs:
        assert ulps(L.oracle_strict_log(float(x)), math.log(float(x))) <= 1
    assert L.oracle_strict_log(1.0) == 0.0
    assert L.oracle_strict_log(0.0) == -math.inf
    assert math.isnan(L.oracle_strict_log(-1.0))


def test_embedding_constructor(oracle):
    # SimClustersEmbedding.scala:490-509: drop <= 0, sort (score desc, id asc); sorted arrays by id
    e = oracle.embedding([7, 3, 9, 1, 5], [2.0, 5.0, 2.0, -1.0, 0.0])
    assert list(e["clusterIds"]) == [3, 7, 9]
    assert list(e["scores"]) == [5.0, 2.0, 2.0]
    assert list(e["sortedClusterIds"]) == [3, 7, 9]
    assert e["l2norm"] == math.sqrt(5.0 * 5.0 + 2.0 * 2.0 + 2.0 * 2.0)
    t = oracle.embedding([7, 3, 9], [2.0, 5.0, 2.0], truncate=2)
    assert list(t["clusterIds"]) == [3, 7]


def test_baseline_legs_equal_the_oracle(oracle):
    """bench.py's cpu_baseline legs (two-map "original", one-map "optimized", oracle/oracle_baseline.c) give exactly the
    oracle's answers: what is timed is the reference's algorithm, not something cheaper."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from _pkg import load_package
    pkg = load_package()
    co = pkg.corpus.make_corpus(20000, 800, seed=3, index_cap=300)
    co.tweet_ids[5] = 0  # a tweet id 0: "optimized" drops it, "original" keeps it
    offs, cids, scs = pkg.corpus.make_queries(12, 800, seed=4, clusters_per_user=40)

    class Cfg:
        maxNumResults, minScore, candidateEmbeddingType = 200, 0.0, 0
        maxTopTweetsPerCluster, maxScanClusters, maxTweetCandidateAgeHours, minTweetCandidateAgeHours = 150, 30, 175200, 0
    for alg in (1, 2, 3, 4):
        Cfg.annAlgorithm = alg
        for variant in (0, 1):
            for q in range(12):
                e_i, e_s = cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]]
                a = oracle.baseline_query(variant, e_i, e_s, Cfg, co.now_ms, co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
                b = oracle.sann_query(e_i, e_s, None, Cfg, co.now_ms, co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, variant=variant)
                assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.int64), b[1].view(np.int64))


def test_strict_exp_is_within_an_ulp_of_libm(oracle):
    import math
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-40, 5, 20000), rng.uniform(-1e-3, 1e-3, 2000), [0.0, -0.0, 1.0, -1.0, 709.0, -745.0, -750.0, 710.0]])
    for x in xs:
        got, want = oracle.strict_exp(float(x)), math.exp(float(x)) if x < 709.7 else float("inf")
        assert got == want or abs(got - want) <= math.ulp(want), (x, got, want)
    assert oracle.strict_exp(float("-inf")) == 0.0 and oracle.strict_exp(float("inf")) == float("inf")


def test_store_list_hand_cases(oracle):
    """TopKTweetsForClusterReadableStore + provider on values a reader can check: half-life decay, the > 0 filter,
    score-descending order with id-ascending ties, take."""
    import math
    half_life = 8 * 3600 * 1000
    now = 1_700_000_000_000
    sc = lambda ms: ms * math.log(2.0) / half_life  # DecayedValue.build's scaledTime
    ids = np.array([5, 6, 7, 8, 9, 10], np.int64)
    vals = np.array([1.0, 4.0, 2.0, 0.0, -3.0, 2.0])
    #        8 h old -> 0.5 | 16 h old -> 1.0 | now -> 2.0 | zero | negative | written 1 h in the FUTURE: kept as is
    st = np.array([sc(now - half_life), sc(now - 2 * half_life), sc(now), sc(now), sc(now), sc(now + 3600_000)])
    got_i, got_s = oracle.store_list(ids, vals, st, sc(now), 10)
    assert got_i.tolist() == [7, 10, 6, 5]  # 2.0 (id 7) and 2.0 (id 10) tie: id ascending
    assert got_s[0] == 2.0 and got_s[1] == 2.0
    # (scaledTime ~ 4e4 carries ~7e-12 of rounding: the decay factor is exact to about 1e-11, as in the reference)
    assert abs(got_s[2] - 1.0) <= 1e-10 and abs(got_s[3] - 0.5) <= 1e-10
    assert oracle.store_list(ids, vals, st, sc(now), 2)[0].tolist() == [7, 10]          # take(2)
    assert oracle.store_list(ids, vals, None, 0.0, 10)[0].tolist() == [6, 7, 10, 5]    # no decay: 4, 2, 2, 1


def test_topk_merge_hand_cases(oracle):
    """TopKTweetsWithScoresMonoid.plus by hand: one half-life = ln 2 in scaled time."""
    ln2 = math.log(2.0)
    a = {1: (1.0, 0.0), 2: (0.3, 0.0), 3: (0.0015, 0.0)}
    b = {1: (0.4, ln2), 4: (0.25, ln2), 2: (0.2, ln2)}
    got = oracle.topk_merge(a, b, 10, 0.001, 0)
    # a decays by exactly 1/2: tweet 1 keeps a's 0.5 (> b's 0.4), tweet 2 takes b's 0.2 (> 0.15), tweet 3 falls to
    # 0.00075 < threshold, tweet 4 is b's alone; every survivor sits at the latest time
    assert set(got) == {1, 2, 4}
    assert got[1] == (pytest.approx(0.5, rel=1e-15), ln2) and got[2] == (0.2, ln2) and got[4] == (0.25, ln2)
    # an empty side returns the other untouched (no decay, no threshold), but the age filter still applies
    assert oracle.topk_merge({}, {5: (1e-9, 3.0), 9: (2.0, 1.0)}, 1, 0.5, 6) == {9: (2.0, 1.0)}
    assert oracle.topk_merge(None, None, 1, 0.5, 0) is None
    # the cut fires only above 1.2 x topK entries: 12 entries, topK 10 -> kept; 13 -> the 10 largest
    many = {i: (float(i), 0.0) for i in range(1, 13)}
    assert len(oracle.topk_merge(many, {100: (0.5, 0.0)}, 10, 0.0, 0)) == 10
    assert len(oracle.topk_merge({i: (float(i), 0.0) for i in range(1, 12)}, {100: (0.5, 0.0)}, 10, 0.0, 0)) == 12
