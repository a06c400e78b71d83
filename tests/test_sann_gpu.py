"""GPU parity: the HIP path through the C ABI against the CPU oracle on the same inputs.
Bar: tweet ids bit-exact (same set, same order), scores bit-exact (fp64; no tolerance needed
because both sides do unfused IEEE arithmetic in the same order), candidateScoresMap.size equal."""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

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


@pytest.mark.parametrize("P", [1, 4, 32])
@pytest.mark.parametrize("case", KAT["sann"], ids=[c["name"] for c in KAT["sann"]])
def test_kat_through_c_abi(pkg, case, P):
    cids, offs, tids, scs = lists_to_csr(case["lists"])
    index = pkg.ClusterTweetIndex(cids, offs, tids, scs, n_partitions=P)
    c = case["config"]
    cfg = pkg.SimClustersANNConfig(**{**c, "annAlgorithm": pkg.ScoringAlgorithm(c["annAlgorithm"])})
    op = pkg.ApproximateCosineSimilarity(index, pkg.Variant(case["variant"]), now_ms=case["now_ms"])
    seen = []
    got = op.apply([(e[0], e[1]) for e in case["emb"]], case["source"], cfg, seen.append, case.get("scan_keys"))
    assert seen == [case["map_size"]]
    assert [g[0] for g in got] == [e[0] for e in case["expect"]]
    for g, e in zip(got, case["expect"]):
        exp = float.fromhex(e[1])
        if case["ulp"] == 0:
            assert g[1] == exp
        else:
            assert abs(g[1] - exp) <= case["ulp"] * math.ulp(exp)
    index.close()


def run_batch(pkg, index, co, offs, cids, scs, cfg, variant=0, **kw):
    qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=co.now_ms, variant=pkg.Variant(variant), **kw)
    qb.run()
    qb.finish()
    out = qb.results()
    st = qb.stats()
    qb.close()
    return out, st


def check_against_oracle(pkg, oracle, co, offs, cids, scs, cfg, out, variant=0, sources=None, scan=None):
    ids, scores, counts, msz = out
    nq = len(offs) - 1
    for q in range(nq):
        cq = cfg[q] if isinstance(cfg, list) else cfg
        src = None if sources is None else sources[q]
        so = None if scan is None else scan[q]
        o_ids, o_sc, o_msz = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], src, cq, co.now_ms,
                                               co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores,
                                               variant=variant, scan_order=so)
        assert counts[q] == len(o_ids), (q, counts[q], len(o_ids))
        assert msz[q] == o_msz, (q, msz[q], o_msz)
        assert np.array_equal(ids[q, :counts[q]], o_ids), f"query {q}: id order differs"
        assert np.array_equal(scores[q, :counts[q]].view(np.int64), o_sc.view(np.int64)), f"query {q}: scores differ"


@pytest.fixture(scope="module")
def small(pkg):
    co = pkg.corpus.make_corpus(30000, 1500, seed=21, index_cap=400)
    offs, cids, scs = pkg.corpus.make_queries(24, 1500, seed=22, clusters_per_user=50)
    return co, offs, cids, scs


@pytest.mark.parametrize("P", [1, 8, 32])
@pytest.mark.parametrize("alg", [1, 2, 3, 4])
def test_random_corpus_bit_exact(pkg, oracle, small, alg, P):
    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=P)
    cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=300, maxScanClusters=50,
                                   annAlgorithm=pkg.ScoringAlgorithm(alg))
    out, st = run_batch(pkg, index, co, offs, cids, scs, cfg)
    check_against_oracle(pkg, oracle, co, offs, cids, scs, cfg, out)
    assert st.postings_scanned > 0
    index.close()


def test_per_query_configs_sources_and_windows(pkg, oracle, small):
    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=16)
    rng = np.random.default_rng(5)
    nq = len(offs) - 1
    cfgs, sources = [], []
    for q in range(nq):
        cfgs.append(pkg.SimClustersANNConfig(
            maxNumResults=int(rng.choice([1, 10, 400, 1000, 5000])), minScore=float(rng.choice([0.0, 0.05, 0.3])),
            maxTopTweetsPerCluster=int(rng.choice([1, 50, 400, 10000])), maxScanClusters=int(rng.choice([1, 5, 50, 200])),
            maxTweetCandidateAgeHours=int(rng.choice([6, 12, 24, 175200])), minTweetCandidateAgeHours=int(rng.choice([0, 1, 3])),
            annAlgorithm=pkg.ScoringAlgorithm(int(rng.integers(1, 5)))))
        # half the queries are tweet-sourced with an id that exists in the corpus
        sources.append(int(co.tweet_ids[rng.integers(0, len(co.tweet_ids))]) if q % 2 else None)
    src = np.array([0 if s is None else s for s in sources], np.int64)
    has = np.array([0 if s is None else 1 for s in sources], np.uint8)
    for variant in (0, 1, 2):
        out, _ = run_batch(pkg, index, co, offs, cids, scs, cfgs, variant=variant, source_tweet_ids=src,
                           has_source_tweet=has)
        check_against_oracle(pkg, oracle, co, offs, cids, scs, cfgs, out, variant=variant, sources=sources)
    index.close()


@pytest.mark.parametrize("P", [2, 8])
@pytest.mark.parametrize("k", [1, 37, 104, 256, 257])
def test_few_partitions_small_k_merge(pkg, oracle, small, P, k):
    """Shape of a sharded run (<= 8 partitions, short result lists): the merge kernel's small-LDS variant
    (256 survivors, 640 staged entries, several tournament rounds when the units' lists are long) and, at
    k = 257, the hand-over to the 512-survivor variant."""
    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=P)
    for alg in (1, 3):
        cfg = pkg.SimClustersANNConfig(maxNumResults=k, maxTopTweetsPerCluster=400, annAlgorithm=pkg.ScoringAlgorithm(alg),
                                       maxTweetCandidateAgeHours=175200)
        out, st = run_batch(pkg, index, co, offs, cids, scs, cfg)
        check_against_oracle(pkg, oracle, co, offs, cids, scs, cfg, out)
    index.close()


@pytest.mark.parametrize("P", [1, 4, 16])
def test_wave_merge_beside_workgroup_merge(pkg, oracle, small, P):
    """merge_wave_kernel (one wave per query: a shard's small queries) beside merge_kernel: per-query k from 3 to 448 and
    list caps from 5 to 400 give queries with a handful of candidates, queries whose candidates end in a partial run of
    64, and queries that do not fit a wave's registers and are left to the workgroup kernel; minScore drops handed-over
    candidates inside either kernel."""
    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=P)
    rng = np.random.default_rng(77 + P)
    nq = len(offs) - 1
    for rep in range(2):
        cfgs = [pkg.SimClustersANNConfig(
            maxNumResults=int(rng.choice([3, 64, 65, 200, 256, 448])), minScore=float(rng.choice([0.0, 0.0, 0.2])),
            maxTopTweetsPerCluster=int(rng.choice([5, 40, 400])), maxScanClusters=int(rng.choice([3, 20, 50])),
            maxTweetCandidateAgeHours=175200, annAlgorithm=pkg.ScoringAlgorithm(int(rng.integers(1, 5))))
            for _ in range(nq)]
        out, _ = run_batch(pkg, index, co, offs, cids, scs, cfgs)
        check_against_oracle(pkg, oracle, co, offs, cids, scs, cfgs, out)
    index.close()


@pytest.mark.parametrize("P", [8, 32])
def test_more_cut_values_than_the_cache_holds(pkg, oracle, small, P):
    """Six distinct maxTopTweetsPerCluster values in one batch: the index caches cut tables for four, the other
    queries' descriptors are found by binary search in the rank column (both descriptor kernels: one wave per
    unit at P = 8, one workgroup per query at P = 32)."""
    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=P)
    nq = len(offs) - 1
    Ms = [3, 17, 60, 150, 333, 10000]
    cfgs = [pkg.SimClustersANNConfig(maxNumResults=100, maxTopTweetsPerCluster=Ms[q % len(Ms)]) for q in range(nq)]
    out, _ = run_batch(pkg, index, co, offs, cids, scs, cfgs)
    check_against_oracle(pkg, oracle, co, offs, cids, scs, cfgs, out)
    index.close()


def test_heavy_duplication_and_explicit_order(pkg, oracle):
    """Every cluster lists the SAME tweets: each candidate is a 40-term ordered fp64 sum, and the
    explicit key order changes last-ulp results (accumulation order is the caller's)."""
    rng = np.random.default_rng(9)
    n_c, n_t = 40, 700
    base = (np.arange(n_t, dtype=np.int64) * 7919 + 12345) << 22
    lists = {}
    for c in range(1, n_c + 1):
        s = np.sort(np.exp(rng.normal(-2, 1, n_t)))[::-1]
        t = rng.permutation(base)
        lists[c] = list(zip(t.tolist(), s.tolist()))
    index = pkg.ClusterTweetIndex.from_map(lists, n_partitions=8)
    cids_csr = np.array(sorted(lists), np.int32)
    offs_csr = np.arange(0, (n_c + 1) * n_t, n_t, dtype=np.int64)
    t_csr = np.concatenate([np.array([x[0] for x in lists[c]], np.int64) for c in sorted(lists)])
    s_csr = np.concatenate([np.array([x[1] for x in lists[c]], np.float64) for c in sorted(lists)])

    class Co:
        now_ms = 1_700_000_000_000
        cluster_ids, list_offsets, tweet_ids, scores = cids_csr, offs_csr, t_csr, s_csr

    emb_c = np.arange(1, n_c + 1, dtype=np.int32)
    emb_s = np.exp(rng.normal(0, 1, n_c))
    offs = np.array([0, n_c, 2 * n_c], np.int64)
    cids = np.concatenate([emb_c, emb_c])
    scs = np.concatenate([emb_s, emb_s])
    order_a = list(range(1, n_c + 1))
    order_b = list(range(n_c, 0, -1))
    so = np.array([0, n_c, 2 * n_c], np.int64)
    sc = np.array(order_a + order_b, np.int32)
    for alg in (1, 2, 3):
        cfg = pkg.SimClustersANNConfig(maxNumResults=1000, maxTopTweetsPerCluster=n_t, maxScanClusters=n_c,
                                       maxTweetCandidateAgeHours=175200, annAlgorithm=pkg.ScoringAlgorithm(alg))
        out, _ = run_batch(pkg, index, Co, offs, cids, scs, cfg, scan_offsets=so, scan_cluster_ids=sc)
        check_against_oracle(pkg, oracle, Co, offs, cids, scs, cfg, out, scan=[order_a, order_b])
        assert out[3][0] == n_t  # candidateScoresMap.size: every tweet once
    # the two orders give different last bits for at least one candidate (so order really matters)
    cfg = pkg.SimClustersANNConfig(maxNumResults=1000, maxTopTweetsPerCluster=n_t, maxScanClusters=n_c,
                                   maxTweetCandidateAgeHours=175200, annAlgorithm=pkg.ScoringAlgorithm.DotProduct)
    out, _ = run_batch(pkg, index, Co, offs, cids, scs, cfg, scan_offsets=so, scan_cluster_ids=sc)
    a = dict(zip(out[0][0, :out[2][0]].tolist(), out[1][0, :out[2][0]].tolist()))
    b = dict(zip(out[0][1, :out[2][1]].tolist(), out[1][1, :out[2][1]].tolist()))
    assert any(a[t] != b[t] for t in a if t in b)
    index.close()


@pytest.mark.parametrize("P", [4, 16])
def test_moderate_duplication_sorted_match_list(pkg, oracle, P):
    """A few dozen match-list entries per unit -- the regime of a duplicate-heavy corpus (1M tweets under 144k clusters), where
    the unit kernel sorts the list by (tweet id, cluster sequence) in registers and settles the groups run by run: 24 popular
    tweets sit in 2 to 5 of the 12 scanned clusters each (groups of every size, ~15-45 entries per unit at P = 4), everything else
    in one."""
    rng = np.random.default_rng(31)
    n_c, per = 12, 220
    popular = ((np.arange(24, dtype=np.int64) * 104729 + 777) << 22) + 5
    lists, nxt = {}, 1
    member = {int(t): set(rng.choice(np.arange(1, n_c + 1), size=int(rng.integers(2, 6)), replace=False).tolist()) for t in popular}
    for c in range(1, n_c + 1):
        mine = [t for t in popular.tolist() if c in member[t]]
        own = ((np.arange(nxt, nxt + per - len(mine), dtype=np.int64) * 15485863) << 22) + 9
        nxt += per
        t = rng.permutation(np.concatenate([np.array(mine, np.int64), own]))
        s = np.sort(np.exp(rng.normal(-2, 1, len(t))))[::-1]
        lists[c] = list(zip(t.tolist(), s.tolist()))
    index = pkg.ClusterTweetIndex.from_map(lists, n_partitions=P)
    cs = sorted(lists)

    class Co:
        now_ms = 1_700_000_000_000
        cluster_ids = np.array(cs, np.int32)
        list_offsets = np.concatenate([[0], np.cumsum([len(lists[c]) for c in cs])]).astype(np.int64)
        tweet_ids = np.concatenate([np.array([x[0] for x in lists[c]], np.int64) for c in cs])
        scores = np.concatenate([np.array([x[1] for x in lists[c]], np.float64) for c in cs])

    nq = 6
    offs = np.arange(0, (nq + 1) * n_c, n_c, dtype=np.int64)
    cids = np.tile(np.arange(1, n_c + 1, dtype=np.int32), nq)
    scs = np.exp(rng.normal(0, 1, nq * n_c))
    for alg in (1, 2, 3, 4):
        cfg = pkg.SimClustersANNConfig(maxNumResults=100, maxTopTweetsPerCluster=per, maxScanClusters=n_c,
                                       maxTweetCandidateAgeHours=175200, annAlgorithm=pkg.ScoringAlgorithm(alg))
        out, st = run_batch(pkg, index, Co, offs, cids, scs, cfg)
        check_against_oracle(pkg, oracle, Co, offs, cids, scs, cfg, out)
        # (the fast path's own duplicate resolution, not the general path's table)
        assert st.n_fallback_units == 0, (st.n_fallback_units, st.n_units)
    index.close()


def test_edge_cases(pkg, oracle, small):
    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=4)
    # empty batch
    qb = pkg.QueryBatch(index, np.zeros(1, np.int64), np.empty(0, np.int32), np.empty(0), pkg.SimClustersANNConfig(),
                        now_ms=co.now_ms)
    qb.run(); qb.finish(); qb.close()
    # empty embeddings, k = 0, negative k / M, tweet id -1 and INT64 extremes in the index
    e_offs = np.array([0, 0, 3, 6, 9], np.int64)
    e_c = np.array([5, 6, 7] * 3, np.int32)
    e_s = np.array([1.0, 2.0, 3.0] * 3)
    cfgs = [pkg.SimClustersANNConfig(), pkg.SimClustersANNConfig(maxNumResults=0),
            pkg.SimClustersANNConfig(maxNumResults=-3), pkg.SimClustersANNConfig(maxTopTweetsPerCluster=-1)]
    out, _ = run_batch(pkg, index, co, e_offs, e_c, e_s, cfgs)
    check_against_oracle(pkg, oracle, co, e_offs, e_c, e_s, cfgs, out)
    assert list(out[2]) == [0, 0, 0, 0]
    index.close()

    lists = {1: [(-1, 3.0), (2**63 - 1, 2.0), (-(2**63), 1.5), (0, 1.0)], 2: [(-1, 0.5), (7, 0.25)]}
    ix2 = pkg.ClusterTweetIndex.from_map(lists, n_partitions=2)
    # one second after the Snowflake epoch with a 24 h window: earliest is negative, so tweet id
    # -1 (the hash table's sentinel value) is a legal candidate and must take the special slot
    now = 1288834974657 + 1000
    op = pkg.ApproximateCosineSimilarity(ix2, pkg.Variant.original, now_ms=now)
    cfg = pkg.SimClustersANNConfig(maxTweetCandidateAgeHours=24, annAlgorithm=pkg.ScoringAlgorithm.DotProduct)
    got = op.apply([(1, 2.0), (2, 4.0)], None, cfg)
    cids_c = np.array([1, 2], np.int32); offs_c = np.array([0, 4, 6], np.int64)
    t_c = np.array([-1, 2**63 - 1, -(2**63), 0, -1, 7], np.int64); s_c = np.array([3.0, 2.0, 1.5, 1.0, 0.5, 0.25])
    o_ids, o_sc, _ = oracle.sann_query([1, 2], [2.0, 4.0], None, cfg, now, cids_c, offs_c, t_c, s_c)
    assert [g[0] for g in got] == o_ids.tolist() and [g[1] for g in got] == o_sc.tolist()
    assert got[0] == (-1, 3.0 * 2.0 + 0.5 * 4.0) and {t for t, _ in got} == {-1, 0, 7}
    ix2.close()


@pytest.mark.parametrize("S,P", [(3, 8), (8, 4)])
def test_tweet_hash_shards_merge_exactly(pkg, oracle, small, S, P):
    """S tweet-hash shards on one GPU + sann_merge_shards == the unsharded answer (the
    ComposedQueryable pattern, ShardApi.scala:71-87); (8, 4) is the partitioning bench.py gives 8 GPUs."""
    import ctypes as C
    import torch

    co, offs, cids, scs = small
    cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=300)
    nq = len(offs) - 1
    batches, bufs = [], []
    for s in range(S):
        ix = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=P, shard_id=s,
                                   n_shards=S)
        qb = pkg.QueryBatch(ix, offs, cids, scs, cfg, now_ms=co.now_ms)
        qb.run(); qb.finish()
        batches.append((ix, qb))
    assert sum(b[0].info().n_postings for b in batches) == batches[0][0].info().n_postings_total
    stride = batches[0][1].stride
    dev = torch.device("cuda:0")
    g_ids = torch.zeros((S, nq, stride), dtype=torch.int64, device=dev)
    g_sc = torch.zeros((S, nq, stride), dtype=torch.float64, device=dev)
    g_cnt = torch.zeros((S, nq), dtype=torch.int32, device=dev)
    g_msz = torch.zeros((S, nq), dtype=torch.int32, device=dev)
    hip = C.CDLL("libamdhip64.so")
    for s, (ix, qb) in enumerate(batches):
        (p_ids, p_sc, p_cnt, p_msz), st = qb.device_results()
        assert st == stride
        for dst, src, nbytes in ((g_ids[s], p_ids, nq * stride * 8), (g_sc[s], p_sc, nq * stride * 8),
                                 (g_cnt[s], p_cnt, nq * 4), (g_msz[s], p_msz, nq * 4)):
            assert hip.hipMemcpy(C.c_void_p(dst.data_ptr()), C.c_void_p(src), C.c_size_t(nbytes), 3) == 0
    o_ids = torch.zeros((nq, stride), dtype=torch.int64, device=dev)
    o_sc = torch.zeros((nq, stride), dtype=torch.float64, device=dev)
    o_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    o_msz = torch.zeros(nq, dtype=torch.int32, device=dev)
    lib = pkg.load_library()
    rc = lib.sann_merge_shards(0, None, S, nq, stride, 0, g_ids.data_ptr(), g_sc.data_ptr(), g_cnt.data_ptr(),
                               g_msz.data_ptr(), batches[0][1].device_k(), o_ids.data_ptr(), o_sc.data_ptr(),
                               o_cnt.data_ptr(), o_msz.data_ptr())
    assert rc == 0, lib.sann_last_error()
    torch.cuda.synchronize()
    out = (o_ids.cpu().numpy(), o_sc.cpu().numpy(), o_cnt.cpu().numpy(), o_msz.cpu().numpy())
    check_against_oracle(pkg, oracle, co, offs, cids, scs, cfg, out)
    # the same lists in another order (not what the library's own merge kernels leave): the merge ranks sorted lists
    # against each other, but it checks that they are sorted and selects + sorts when they are not
    cnt = g_cnt.cpu().numpy()
    gi, gs = g_ids.cpu().numpy().copy(), g_sc.cpu().numpy().copy()
    rng = np.random.default_rng(3)
    for s in range(S):
        for q in range(nq):
            c = int(cnt[s, q])
            if q % 3 == 0 or c < 2:
                continue  # (some queries keep their sorted lists)
            perm = rng.permutation(c)
            gi[s, q, :c] = gi[s, q, :c][perm]
            gs[s, q, :c] = gs[s, q, :c][perm]
    g_ids.copy_(torch.from_numpy(gi)); g_sc.copy_(torch.from_numpy(gs))
    o_ids.zero_(); o_sc.zero_(); o_cnt.zero_(); o_msz.zero_()
    rc = lib.sann_merge_shards(0, None, S, nq, stride, 0, g_ids.data_ptr(), g_sc.data_ptr(), g_cnt.data_ptr(),
                               g_msz.data_ptr(), batches[0][1].device_k(), o_ids.data_ptr(), o_sc.data_ptr(),
                               o_cnt.data_ptr(), o_msz.data_ptr())
    assert rc == 0, lib.sann_last_error()
    torch.cuda.synchronize()
    out = (o_ids.cpu().numpy(), o_sc.cpu().numpy(), o_cnt.cpu().numpy(), o_msz.cpu().numpy())
    check_against_oracle(pkg, oracle, co, offs, cids, scs, cfg, out)
    for ix, qb in batches:
        qb.close(); ix.close()


def test_cut_shard_lists_merge_exactly_or_say_so(pkg, oracle, small):
    """sann_merge_shards_cut: every shard delivers only its top shard_k; the owner's merged top-k equals the
    unsharded answer whenever the kernel's proof holds (no cut list ends above the merged k-th key), and the
    queries where it does not hold are counted -- the count must match the same rule evaluated on the host."""
    import ctypes as C
    import torch

    co, offs, cids, scs = small
    K, S = 120, 3
    cfg = pkg.SimClustersANNConfig(maxNumResults=K, maxTopTweetsPerCluster=300)
    nq = len(offs) - 1
    dev = torch.device("cuda:0")
    hip = C.CDLL("libamdhip64.so")
    lib = pkg.load_library()
    shards = [pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=8, shard_id=s,
                                    n_shards=S) for s in range(S)]
    seen_exact = seen_inexact = False
    for shard_k in (K, 72, 56, 8):
        cfg_s = pkg.SimClustersANNConfig(maxNumResults=shard_k, maxTopTweetsPerCluster=300)
        g_ids = torch.zeros((S, nq, shard_k), dtype=torch.int64, device=dev)
        g_sc = torch.zeros((S, nq, shard_k), dtype=torch.float64, device=dev)
        g_cnt = torch.zeros((S, nq), dtype=torch.int32, device=dev)
        g_msz = torch.zeros((S, nq), dtype=torch.int32, device=dev)
        for s, ix in enumerate(shards):
            qb = pkg.QueryBatch(ix, offs, cids, scs, cfg_s, now_ms=co.now_ms)
            qb.run(); qb.finish()
            (p_ids, p_sc, p_cnt, p_msz), st = qb.device_results()
            assert st == shard_k
            for dst, src, nbytes in ((g_ids[s], p_ids, nq * st * 8), (g_sc[s], p_sc, nq * st * 8),
                                     (g_cnt[s], p_cnt, nq * 4), (g_msz[s], p_msz, nq * 4)):
                assert hip.hipMemcpy(C.c_void_p(dst.data_ptr()), C.c_void_p(src), C.c_size_t(nbytes), 3) == 0
            qb.close()
        o_ids = torch.zeros((nq, K), dtype=torch.int64, device=dev)
        o_sc = torch.zeros((nq, K), dtype=torch.float64, device=dev)
        o_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
        o_msz = torch.zeros(nq, dtype=torch.int32, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = lib.sann_merge_shards_cut(0, None, S, nq, shard_k, 0, shard_k, K, K, g_ids.data_ptr(), g_sc.data_ptr(),
                                       g_cnt.data_ptr(), g_msz.data_ptr(), o_ids.data_ptr(), o_sc.data_ptr(),
                                       o_cnt.data_ptr(), o_msz.data_ptr(), bad.data_ptr())
        assert rc == 0, lib.sann_last_error()
        torch.cuda.synchronize()
        # the same rule on the host
        h_ids, h_sc, h_cnt = g_ids.cpu().numpy(), g_sc.cpu().numpy(), g_cnt.cpu().numpy()
        flagged = np.zeros(nq, bool)
        for q in range(nq):
            ent = sorted((-float(h_sc[s, q, j]), int(h_ids[s, q, j])) for s in range(S) for j in range(h_cnt[s, q]))
            kth = ent[K - 1] if len(ent) >= K else None
            for s in range(S):
                c = h_cnt[s, q]
                if c >= shard_k:
                    last = (-float(h_sc[s, q, c - 1]), int(h_ids[s, q, c - 1]))
                    if kth is None or last < kth:
                        flagged[q] = True
        assert int(bad.item()) == int(flagged.sum()), (shard_k, int(bad.item()), int(flagged.sum()))
        out = (o_ids.cpu().numpy(), o_sc.cpu().numpy(), o_cnt.cpu().numpy(), o_msz.cpu().numpy())
        keep = np.flatnonzero(~flagged)
        seen_exact |= shard_k < K and len(keep) == nq
        seen_inexact |= bool(flagged.any())
        # every query the proof covers equals the unsharded oracle answer, bit for bit
        for q in keep:
            e_ids, e_sc, e_msz = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg, co.now_ms,
                                                   co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
            n = out[2][q]
            assert n == len(e_ids) and out[3][q] == e_msz, (shard_k, q)
            assert np.array_equal(out[0][q, :n], e_ids) and np.array_equal(out[1][q, :n].view(np.int64), e_sc.view(np.int64))
        if shard_k == K:
            assert not flagged.any()  # full-length lists can never be flagged
    assert seen_exact and seen_inexact  # the sweep exercised both outcomes
    for ix in shards:
        ix.close()


def test_outputs_bound_in_owner_chunks(pkg, small):
    """sann_batch_bind_outputs_chunked: query q's results land in chunk q // nql of one packed buffer (the message
    to its owner), identical to the batch's own outputs; a batch bound this way refuses sann_batch_results."""
    import torch

    co, offs, cids, scs = small
    cfg = pkg.SimClustersANNConfig(maxNumResults=50, maxTopTweetsPerCluster=300)
    nq = len(offs) - 1
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=8)
    qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=co.now_ms)
    qb.run(); qb.finish()
    ids, sc, cnt, msz = qb.results()
    stride = qb.stride
    for nql in (nq // 3, 5, nq):  # 5 does not divide 24: the last chunk is ragged
        n_chunks = -(-nq // nql)
        arr = nql * stride * 8
        chunk = 2 * arr + 8 * nql + 64  # slack between chunks must stay untouched
        buf = torch.full((n_chunks * chunk,), 0xAB, dtype=torch.uint8, device="cuda:0")
        p = buf.data_ptr()
        qb.bind_outputs_chunked(p, p + arr, p + 2 * arr, p + 2 * arr + 4 * nql, nql, chunk)
        qb.run(); qb.finish()
        with pytest.raises(pkg.simclusters_ann.SannError):
            qb.results()
        h = buf.cpu().numpy()
        for q in range(nq):
            c, ql = divmod(q, nql)
            base = h[c * chunk:(c + 1) * chunk]
            g_ids = base[:arr].view(np.int64).reshape(nql, stride)[ql]
            g_sc = base[arr:2 * arr].view(np.int64).reshape(nql, stride)[ql]
            g_cnt = base[2 * arr:2 * arr + 4 * nql].view(np.int32)[ql]
            g_msz = base[2 * arr + 4 * nql:2 * arr + 8 * nql].view(np.int32)[ql]
            assert g_cnt == cnt[q] and g_msz == msz[q]
            assert np.array_equal(g_ids[:g_cnt], ids[q, :g_cnt]) and np.array_equal(g_sc[:g_cnt], sc[q, :g_cnt].view(np.int64))
        for c in range(n_chunks):
            assert (h[c * chunk + 2 * arr + 8 * nql:(c + 1) * chunk] == 0xAB).all()
    qb.bind_outputs(0, 0, 0, 0)  # unbind: the batch's own buffers again
    qb.run(); qb.finish()
    again = qb.results()
    assert np.array_equal(again[0], ids) and np.array_equal(again[2], cnt)
    qb.close(); index.close()


def same_results(got, want):
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
    for q, n in enumerate(want[2]):
        assert np.array_equal(got[0][q, :n], want[0][q, :n])
        assert np.array_equal(got[1][q, :n].view(np.int64), want[1][q, :n].view(np.int64))


@pytest.mark.parametrize("after_merge", [False, True])
def test_batches_in_flight_on_two_streams(pkg, small, after_merge):
    """sann_batch_run_after: two batches alternate on two non-blocking streams, each unit kernel ordered behind
    the other batch's by a cross-stream event; every pass gives exactly the results of a plain run."""
    import ctypes as C

    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=8)
    cfgs = [pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=300),
            pkg.SimClustersANNConfig(maxNumResults=37, maxTopTweetsPerCluster=120, annAlgorithm=pkg.ScoringAlgorithm.DotProduct)]
    want = []
    for cfg in cfgs:
        out, _ = run_batch(pkg, index, co, offs, cids, scs, cfg)
        want.append(out)
    hip = C.CDLL("libamdhip64.so")
    streams = []
    for _ in range(2):
        h = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0
        streams.append(h.value)
    qbs = [pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=co.now_ms) for cfg in cfgs]
    launched = []
    for i in range(9):
        j = i & 1
        qbs[j].run_after(streams[j], qbs[1 - j], after_merge=after_merge)
        launched.append(j)
        if len(launched) > 1:
            d = launched.pop(0)
            qbs[d].finish(streams[d])
            same_results(qbs[d].results(), want[d])
    d = launched.pop(0)
    qbs[d].finish(streams[d])
    same_results(qbs[d].results(), want[d])
    for qb in qbs:
        qb.close()
    for st in streams:
        assert hip.hipStreamDestroy(C.c_void_p(st)) == 0
    index.close()


def test_device_fp64_division_sqrt_log_are_bit_exact(pkg, oracle):
    """The normalisation (ApproximateCosineSimilarity.scala:111-119) evaluated on the device equals
    the host's IEEE result bit for bit: correctly rounded / and sqrt, and the fdlibm log."""
    import ctypes as C

    rng = np.random.default_rng(17)
    n = 400_000
    dot = np.exp(rng.normal(0, 3, n))
    nsq = np.exp(rng.normal(0, 4, n))
    nsq[:1000] = rng.uniform(1e-300, 1e-290, 1000)
    nsq[1000:2000] = rng.uniform(1e-320, 1e-310, 1000)  # subnormal
    l2, ln = 3.3721, 2.1234
    lib = pkg.load_library()
    out = np.zeros(n)
    L = oracle.lib()
    for alg in (1, 2, 3, 4):
        rc = lib.sann_debug_normalise(0, alg, n, dot.ctypes.data_as(C.c_void_p), nsq.ctypes.data_as(C.c_void_p),
                                      l2, ln, out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        if alg == 1:
            ref = dot
        elif alg == 2:
            ref = dot / l2 / np.sqrt(nsq)
        elif alg == 4:
            ref = dot / np.sqrt(nsq)
        else:
            lg = np.array([L.oracle_strict_log(float(1 + x)) for x in nsq[:50000]])
            with np.errstate(divide="ignore"):
                ref = dot[:50000] / ln / lg
        m = len(ref)
        bad = np.nonzero(out[:m].view(np.int64) != np.asarray(ref).view(np.int64))[0]
        assert len(bad) == 0, (alg, len(bad), out[bad[:3]], np.asarray(ref)[bad[:3]])


def _mix64(x):
    m = (1 << 64) - 1
    x &= m
    x ^= x >> 33
    x = (x * 0xff51afd7ed558ccd) & m
    x ^= x >> 33
    x = (x * 0xc4ceb9fe1a85ec53) & m
    x ^= x >> 33
    return x


def test_forced_general_path_matches(pkg, oracle, small, monkeypatch):
    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=8)
    cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=300)
    monkeypatch.setenv("SANN_FORCE_GENERAL", "1")
    out, st = run_batch(pkg, index, co, offs, cids, scs, cfg)
    monkeypatch.delenv("SANN_FORCE_GENERAL")
    check_against_oracle(pkg, oracle, co, offs, cids, scs, cfg, out)
    out2, st2 = run_batch(pkg, index, co, offs, cids, scs, cfg)
    for a, b in zip(out, out2):
        assert np.array_equal(a, b)
    index.close()


def test_fast_path_falls_back_when_units_do_not_fit(pkg, oracle, small):
    """P = 1: a unit is a whole query (thousands of postings): the LDS path flags overflow and the
    general kernel settles it; results stay exact and the stats say so."""
    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=1)
    cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=400)
    out, st = run_batch(pkg, index, co, offs, cids, scs, cfg)
    check_against_oracle(pkg, oracle, co, offs, cids, scs, cfg, out)
    assert st.n_fallback_units > 0
    index.close()


def test_skewed_partition_forces_requery(pkg, oracle, monkeypatch):
    """All high scorers hash to ONE partition: that unit truncates its list, the merge cannot
    prove the top-k exact, the query is re-run on the general path.  Still bit-exact.
    (The unit geometry is pinned to 1024 postings: a batch prepared on the device sizes its units for uniformly hashed
    tweets, and this adversarial unit -- 600 postings of one list in one partition -- would simply overflow to the
    general path on its own, which the second half checks.)"""
    P = 32
    rng = np.random.default_rng(31)
    hot, cold = [], []
    x = 1 << 40
    while len(hot) < 600 or len(cold) < 3000:
        x += int(rng.integers(1, 1000))
        (hot if (_mix64(x) & (P - 1)) == 5 else cold).append(x)
    hot, cold = hot[:600], cold[:3000]
    # DotProduct scoring: cluster 1 (weight 10) lists only partition-5 tweets with the highest scores
    lists = {1: [(t, 5.0 - 0.001 * i) for i, t in enumerate(hot)],
             2: [(t, 1.0 - 0.0001 * i) for i, t in enumerate(cold[:1500])],
             3: [(t, 0.9 - 0.0001 * i) for i, t in enumerate(cold[1500:])]}
    index = pkg.ClusterTweetIndex.from_map(lists, n_partitions=P)
    cids_c = np.array([1, 2, 3], np.int32)
    offs_c = np.array([0, 600, 2100, 3600], np.int64)
    t_c = np.array([t for c in (1, 2, 3) for t, _ in lists[c]], np.int64)
    s_c = np.array([s for c in (1, 2, 3) for _, s in lists[c]], np.float64)

    class Co:
        now_ms = 1_700_000_000_000
        cluster_ids, list_offsets, tweet_ids, scores = cids_c, offs_c, t_c, s_c

    offs = np.array([0, 3], np.int64)
    cids = np.array([1, 2, 3], np.int32)
    scs = np.array([10.0, 1.0, 1.0])
    cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=2000, maxTweetCandidateAgeHours=175200,
                                   annAlgorithm=pkg.ScoringAlgorithm.DotProduct)
    monkeypatch.setenv("SANN_UNIT_CAP", "1024")
    out, st = run_batch(pkg, index, Co, offs, cids, scs, cfg)
    check_against_oracle(pkg, oracle, Co, offs, cids, scs, cfg, out)
    assert set(out[0][0, :400].tolist()) <= set(hot)
    assert st.n_requeried == 1 and st.n_fallback_units == P
    monkeypatch.delenv("SANN_UNIT_CAP")
    out, st = run_batch(pkg, index, Co, offs, cids, scs, cfg)
    check_against_oracle(pkg, oracle, Co, offs, cids, scs, cfg, out)
    assert st.n_requeried == 0 and st.n_fallback_units == 1 and st.max_unit_postings >= 600
    index.close()


def test_many_scan_clusters_overflow_to_general(pkg, oracle, small):
    """More scanned clusters than the fast path's descriptor table (128) -> general path."""
    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=16)
    rng = np.random.default_rng(2)
    e_c = rng.choice(co.cluster_ids, size=300, replace=False).astype(np.int32)
    e_s = np.exp(rng.normal(0, 1, 300))
    e_offs = np.array([0, 300], np.int64)
    cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=50, maxScanClusters=300)
    out, st = run_batch(pkg, index, co, e_offs, e_c, e_s, cfg)
    check_against_oracle(pkg, oracle, co, e_offs, e_c, e_s, cfg, out)
    assert st.n_fallback_units == 16
    index.close()


@pytest.mark.parametrize("alg", [1, 2, 3])
def test_legacy_variant_bit_exact(pkg, oracle, small, alg):
    """SANN_VARIANT_LEGACY (simclusters_v2/candidate_source/SimClustersANNCandidateSource.scala:107-181):
    no minScore filter, real lower bound of the age window, 'log' form over l2norm, source exclusion."""
    co, offs, cids, scs = small
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=8)
    nq = len(offs) - 1
    rng = np.random.default_rng(alg)
    sources = [int(co.tweet_ids[rng.integers(len(co.tweet_ids))]) if q % 3 == 0 else None for q in range(nq)]
    cfgs = [pkg.SimClustersANNConfig(maxNumResults=150 + 10 * q, minScore=0.5, maxTopTweetsPerCluster=200, maxScanClusters=40,
                                     maxTweetCandidateAgeHours=[175200, 12, 24][q % 3], minTweetCandidateAgeHours=q % 2,
                                     annAlgorithm=pkg.ScoringAlgorithm(alg)) for q in range(nq)]
    src = np.array([0 if s is None else s for s in sources], np.int64)
    has = np.array([0 if s is None else 1 for s in sources], np.uint8)
    out, _ = run_batch(pkg, index, co, offs, cids, scs, cfgs, variant=3, source_tweet_ids=src, has_source_tweet=has)
    check_against_oracle(pkg, oracle, co, offs, cids, scs, cfgs, out, variant=3, sources=sources)
    assert out[2].max() > 100, "minScore 0.5 would have emptied a cosine result under the service variants"
    # refused inputs
    bad = pkg.SimClustersANNConfig(maxNumResults=10, annAlgorithm=pkg.ScoringAlgorithm(4))
    with pytest.raises(pkg.simclusters_ann.SannError):
        pkg.QueryBatch(index, offs, cids, scs, bad, now_ms=co.now_ms, variant=pkg.Variant.legacy)
    big = pkg.SimClustersANNConfig(maxNumResults=1001)
    with pytest.raises(pkg.simclusters_ann.SannError):
        pkg.QueryBatch(index, offs, cids, scs, big, now_ms=co.now_ms, variant=pkg.Variant.legacy)
    index.close()


def test_legacy_source_with_heavy_ranking(pkg, oracle, small):
    """fetchCandidates -> take(maxReRankingCandidates) -> HeavyRanker (pair score >= minScore) -> sort -> take
    (SimClustersANNCandidateSource.scala:182-200, HeavyRanker.scala:32-77), device light rank + device pair scorer,
    against the same composition of the two oracles."""
    co, offs, cids, scs = small
    rs = pkg.representation_scorer
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=8)
    rng = np.random.default_rng(5)
    emb = [(int(c), float(s)) for c, s in zip(cids[offs[2]:offs[3]], scs[offs[2]:offs[3]])]
    lcfg = pkg.LegacySimClustersANNConfig(maxNumResults=60, maxTweetCandidateAgeHours=175200, minScore=0.02, enableHeavyRanking=True,
                                          rankingAlgorithm=6, maxReRankingCandidates=300, maxTopTweetsPerCluster=200, maxScanClusters=40)
    # light ranking by the oracle (legacy log form), then tweet embeddings for ~90 % of those candidates
    ocfg = pkg.SimClustersANNConfig(maxNumResults=300, maxTopTweetsPerCluster=200, maxScanClusters=40, maxTweetCandidateAgeHours=175200,
                                    annAlgorithm=pkg.ScoringAlgorithm.LogCosineSimilarity)
    l_ids, l_sc, _ = oracle.sann_query([c for c, _ in emb], [s for _, s in emb], None, ocfg, co.now_ms, co.cluster_ids,
                                       co.list_offsets, co.tweet_ids, co.scores, variant=3)
    assert len(l_ids) == 300
    tweets = {}
    for t in l_ids.tolist():
        if rng.random() < 0.9:
            n = int(rng.integers(1, 12))
            tweets[t] = [(int(c), float(s)) for c, s in zip(rng.choice(cids[offs[2]:offs[3]], n, replace=False), rng.random(n) + 0.05)]
    src_store, tw_store = rs.EmbeddingStore({77: emb}), rs.EmbeddingStore(tweets)
    source = pkg.LegacySimClustersANNCandidateSource(index, src_store, tw_store, now_ms=co.now_ms)
    got = source.get(emb, None, lcfg, source_internal_id=77)
    se = rs.simclusters_embedding(emb)
    want = []
    for t in l_ids.tolist():
        if t in tweets:
            te = rs.simclusters_embedding(tweets[t])
            s = oracle.pair_score(6, se[0], se[1], te[0], te[1])
            if s >= lcfg.minScore:
                want.append((t, s))
    want.sort(key=lambda x: (-x[1], x[0]))
    want = want[:60]
    assert len(got) == len(want) == 60
    assert [t for t, _ in got] == [t for t, _ in want]
    assert all(abs(a[1] - b[1]) <= 1e-15 * abs(b[1]) for a, b in zip(got, want))
    # without heavy ranking: the light ranking itself, cut at maxNumResults
    lcfg2 = pkg.LegacySimClustersANNConfig(maxNumResults=60, maxTweetCandidateAgeHours=175200, rankingAlgorithm=6,
                                           maxTopTweetsPerCluster=200, maxScanClusters=40)
    light = source.get(emb, None, lcfg2)
    assert [t for t, _ in light] == l_ids[:60].tolist()
    assert np.array_equal(np.array([s for _, s in light]).view(np.int64), l_sc[:60].view(np.int64))
    src_store.close(); tw_store.close(); index.close()


def test_heavy_rank_batch_behind_the_c_abi(pkg, oracle, small):
    """sann_heavy_rank: the legacy source for a BATCH in one C-ABI call -- light rank, then HeavyRanker.rank +
    sortBy(-score) + take(maxNumResults) (SimClustersANNCandidateSource.scala:182-200, HeavyRanker.scala:28-69) fused behind it
    on the device -- against the composition of the operator oracle and the pair-score oracle: several queries, a tweet source,
    candidates without an embedding (None: dropped), a source id the store does not hold (every pair None: empty result),
    three ranking algorithms, and the light ranking alone."""
    co, offs, cids, scs = small
    rs = pkg.representation_scorer
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=8)
    rng = np.random.default_rng(9)
    nq = 5
    embs = [[(int(c), float(s)) for c, s in zip(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]])] for q in range(nq)]
    src_tweets = [None, int(co.tweet_ids[5]), None, None, int(co.tweet_ids[77])]
    sids = [100, 101, 102, 999, 104]  # 999 is not in the source store
    src_store = rs.EmbeddingStore({sids[q]: embs[q] for q in range(nq) if sids[q] != 999})
    for ranking, heavy_min in ((2, 0.05), (6, 0.01), (1, 0.0)):
        lcfg = pkg.LegacySimClustersANNConfig(maxNumResults=40, maxTweetCandidateAgeHours=175200, minScore=heavy_min, enableHeavyRanking=True,
                                              rankingAlgorithm=ranking, maxReRankingCandidates=250, maxTopTweetsPerCluster=200, maxScanClusters=40)
        light_alg = pkg.ScoringAlgorithm.LogCosineSimilarity if ranking == 6 else pkg.ScoringAlgorithm.CosineSimilarity
        ocfg = pkg.SimClustersANNConfig(maxNumResults=250, maxTopTweetsPerCluster=200, maxScanClusters=40, maxTweetCandidateAgeHours=175200,
                                        annAlgorithm=light_alg)
        lights, tweets = [], {}
        for q in range(nq):
            l_ids, _, _ = oracle.sann_query([c for c, _ in embs[q]], [s for _, s in embs[q]], src_tweets[q], ocfg, co.now_ms, co.cluster_ids,
                                            co.list_offsets, co.tweet_ids, co.scores, variant=3)
            lights.append(l_ids)
            for t in l_ids.tolist():
                if t not in tweets and rng.random() < 0.85:
                    n = int(rng.integers(1, 12))
                    pool = cids[offs[q]:offs[q + 1]]
                    tweets[t] = [(int(c), float(s)) for c, s in zip(rng.choice(pool, min(n, len(pool)), replace=False), rng.random(n) + 0.05)]
        tw_store = rs.EmbeddingStore(tweets)
        source = pkg.LegacySimClustersANNCandidateSource(index, src_store, tw_store, now_ms=co.now_ms)
        e_o = np.zeros(nq + 1, np.int64)
        e_o[1:] = np.cumsum([len(e) for e in embs])
        ids, sc, cnt = source.get_batch(e_o, np.concatenate([[c for c, _ in e] for e in embs]).astype(np.int32),
                                        np.concatenate([[s for _, s in e] for e in embs]), lcfg, source_tweet_ids=src_tweets, source_internal_ids=sids)
        for q in range(nq):
            want = []
            if sids[q] != 999:
                se = rs.simclusters_embedding(embs[q])
                for t in lights[q].tolist():
                    if t in tweets:
                        te = rs.simclusters_embedding(tweets[t])
                        s = oracle.pair_score(ranking, se[0], se[1], te[0], te[1])
                        if s >= lcfg.minScore:
                            want.append((t, s))
            want.sort(key=lambda x: (-x[1], x[0]))
            want = want[:40]
            assert cnt[q] == len(want), (ranking, q, cnt[q], len(want))
            assert ids[q, :cnt[q]].tolist() == [t for t, _ in want]
            assert np.array_equal(sc[q, :cnt[q]].view(np.int64), np.array([s for _, s in want]).view(np.int64))
        assert cnt[3] == 0 and cnt[0] == 40
        tw_store.close()
    # without heavy ranking the stores are not needed: the light ranking of every query, cut at maxNumResults
    lcfg2 = pkg.LegacySimClustersANNConfig(maxNumResults=60, maxTweetCandidateAgeHours=175200, rankingAlgorithm=6, maxTopTweetsPerCluster=200,
                                           maxScanClusters=40)
    ocfg2 = pkg.SimClustersANNConfig(maxNumResults=60, maxTopTweetsPerCluster=200, maxScanClusters=40, maxTweetCandidateAgeHours=175200,
                                     annAlgorithm=pkg.ScoringAlgorithm.LogCosineSimilarity)
    ids, sc, cnt = pkg.LegacySimClustersANNCandidateSource(index, now_ms=co.now_ms).get_batch(
        e_o, np.concatenate([[c for c, _ in e] for e in embs]).astype(np.int32), np.concatenate([[s for _, s in e] for e in embs]), lcfg2,
        source_tweet_ids=src_tweets)
    for q in range(nq):
        l_ids, l_sc, _ = oracle.sann_query([c for c, _ in embs[q]], [s for _, s in embs[q]], src_tweets[q], ocfg2, co.now_ms, co.cluster_ids,
                                           co.list_offsets, co.tweet_ids, co.scores, variant=3)
        assert cnt[q] == len(l_ids) and np.array_equal(ids[q, :cnt[q]], l_ids) and np.array_equal(sc[q, :cnt[q]].view(np.int64), l_sc.view(np.int64))
    # argument errors
    with pytest.raises(ValueError):
        pkg.LegacySimClustersANNCandidateSource(index, now_ms=co.now_ms).get_batch(e_o, np.zeros(e_o[-1], np.int32), np.ones(e_o[-1]),
                                                                                   pkg.LegacySimClustersANNConfig(enableHeavyRanking=True))
    src_store.close(); index.close()
