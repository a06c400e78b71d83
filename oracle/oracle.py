"""ctypes loader of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("ORACLE_LIB_PATH") or os.path.join(_HERE, "liboracle.so")  # (override: a sanitizer build of the same sources)


class oracle_sann_config(C.Structure):
    _fields_ = [
        ("max_num_results", C.c_int32),
        ("min_score", C.c_double),
        ("candidate_embedding_type", C.c_int32),
        ("max_top_tweets_per_cluster", C.c_int32),
        ("max_scan_clusters", C.c_int32),
        ("max_tweet_candidate_age_hours", C.c_int32),
        ("min_tweet_candidate_age_hours", C.c_int32),
        ("ann_algorithm", C.c_int32),
    ]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        src_newer = (not os.path.exists(LIB)) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(LIB)
            for f in ("simclusters_oracle.c", "oracle_baseline.c", "hnsw_oracle.c", "Makefile"))
        if src_newer:
            build()
        L = C.CDLL(LIB)
        L.oracle_strict_log.restype = C.c_double
        L.oracle_strict_log.argtypes = [C.c_double]
        L.oracle_snowflake_first_id_for.restype = C.c_int64
        L.oracle_snowflake_first_id_for.argtypes = [C.c_int64]
        L.oracle_pair_score.restype = C.c_double
        L.oracle_pair_score.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.oracle_sann_query.restype = C.c_int32
        L.oracle_sann_query.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                        C.POINTER(oracle_sann_config), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_int32)]
        L.oracle_embedding_build.restype = C.c_void_p
        L.oracle_embedding_build.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        L.oracle_embedding_free.argtypes = [C.c_void_p]
        L.oracle_embedding_size.restype = C.c_int32
        L.oracle_embedding_size.argtypes = [C.c_void_p]
        L.oracle_embedding_export.argtypes = [C.c_void_p] * 5
        for f in ("oracle_embedding_l2norm", "oracle_embedding_lognorm", "oracle_embedding_expscalednorm"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_void_p]
        L.oracle_baseline_run.restype = C.c_double
        L.oracle_baseline_run.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(oracle_sann_config), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_baseline_query.restype = C.c_int32
        L.oracle_baseline_query.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(oracle_sann_config), C.c_int64,
                                            C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_int32)]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_config(cfg) -> oracle_sann_config:
    """cfg: any object with the SimClustersANNConfig thrift field names."""
    return oracle_sann_config(int(cfg.maxNumResults), float(cfg.minScore), int(cfg.candidateEmbeddingType),
                              int(cfg.maxTopTweetsPerCluster), int(cfg.maxScanClusters),
                              int(cfg.maxTweetCandidateAgeHours), int(cfg.minTweetCandidateAgeHours),
                              int(cfg.annAlgorithm))


def sann_query(emb_ids, emb_scores, source_tweet_id, cfg, now_ms, cluster_ids, list_offsets, tweet_ids, scores,
               variant=0, scan_order=None):
    """Returns (ids int64[n], scores float64[n], map_size)."""
    L = lib()
    emb_ids = np.ascontiguousarray(emb_ids, np.int32)
    emb_scores = np.ascontiguousarray(emb_scores, np.float64)
    cluster_ids = np.ascontiguousarray(cluster_ids, np.int32)
    list_offsets = np.ascontiguousarray(list_offsets, np.int64)
    tweet_ids = np.ascontiguousarray(tweet_ids, np.int64)
    scores = np.ascontiguousarray(scores, np.float64)
    so = None if scan_order is None else np.ascontiguousarray(scan_order, np.int32)
    cap = max(1000, int(getattr(cfg, "maxNumResults", 1000)) if variant == 3 else 1000)
    out_ids = np.zeros(cap, np.int64)
    out_scores = np.zeros(cap, np.float64)
    msz = C.c_int32()
    c = make_config(cfg)
    n = L.oracle_sann_query(int(variant), len(emb_ids), _p(emb_ids), _p(emb_scores),
                            0 if source_tweet_id is None else 1, 0 if source_tweet_id is None else int(source_tweet_id),
                            C.byref(c), int(now_ms), len(cluster_ids), _p(cluster_ids), _p(list_offsets), _p(tweet_ids),
                            _p(scores), 0 if so is None else len(so), _p(so), _p(out_ids), _p(out_scores), C.byref(msz))
    return out_ids[:n].copy(), out_scores[:n].copy(), msz.value


def pair_score(algorithm, ids1, sc1, ids2, sc2) -> float:
    a = np.ascontiguousarray(ids1, np.int32)
    b = np.ascontiguousarray(sc1, np.float64)
    c = np.ascontiguousarray(ids2, np.int32)
    d = np.ascontiguousarray(sc2, np.float64)
    return lib().oracle_pair_score(int(algorithm), len(a), _p(a), _p(b), len(c), _p(c), _p(d))


def embedding(ids, scores, truncate=-1):
    """Returns dict with clusterIds/scores (desc) and sortedClusterIds/sortedScores (asc id), norms."""
    L = lib()
    a = np.ascontiguousarray(ids, np.int32)
    b = np.ascontiguousarray(scores, np.float64)
    e = L.oracle_embedding_build(len(a), _p(a), _p(b), int(truncate))
    try:
        n = L.oracle_embedding_size(e)
        ci = np.zeros(n, np.int32); cs = np.zeros(n); si = np.zeros(n, np.int32); ss = np.zeros(n)
        L.oracle_embedding_export(e, _p(ci), _p(cs), _p(si), _p(ss))
        return dict(clusterIds=ci, scores=cs, sortedClusterIds=si, sortedScores=ss,
                    l2norm=L.oracle_embedding_l2norm(e), logNorm=L.oracle_embedding_lognorm(e),
                    expScaledNorm=L.oracle_embedding_expscalednorm(e))
    finally:
        L.oracle_embedding_free(e)


def baseline_query(variant, emb_ids, emb_scores, cfg, now_ms, cluster_ids, list_offsets, tweet_ids, scores):
    """One query through a cpu_baseline leg (0 = "original", two maps; 1 = "optimized", one map)."""
    L = lib()
    emb_ids = np.ascontiguousarray(emb_ids, np.int32)
    emb_scores = np.ascontiguousarray(emb_scores, np.float64)
    out_ids = np.zeros(1000, np.int64)
    out_scores = np.zeros(1000, np.float64)
    msz = C.c_int32()
    c = make_config(cfg)
    n = L.oracle_baseline_query(int(variant), len(emb_ids), _p(emb_ids), _p(emb_scores), C.byref(c), int(now_ms),
                                len(cluster_ids), _p(np.ascontiguousarray(cluster_ids, np.int32)),
                                _p(np.ascontiguousarray(list_offsets, np.int64)), _p(np.ascontiguousarray(tweet_ids, np.int64)),
                                _p(np.ascontiguousarray(scores, np.float64)), _p(out_ids), _p(out_scores), C.byref(msz))
    return out_ids[:n].copy(), out_scores[:n].copy(), msz.value


def baseline_run(variant, n_threads, emb_offsets, emb_ids, emb_scores, cfg, now_ms, cluster_ids, list_offsets,
                 tweet_ids, scores, out_ids, out_scores, out_counts) -> float:
    """Run every query through the restated Scala path on n_threads threads; returns seconds."""
    c = make_config(cfg)
    nq = len(emb_offsets) - 1
    return lib().oracle_baseline_run(int(variant), int(n_threads), nq, _p(emb_offsets), _p(emb_ids), _p(emb_scores),
                                     C.byref(c), int(now_ms), len(cluster_ids), _p(cluster_ids), _p(list_offsets),
                                     _p(tweet_ids), _p(scores), _p(out_ids), _p(out_scores), _p(out_counts))


# ---------------------------------------------------------------------------------------------
# dense exhaustive search.  PARITY UNPINNED: the reference's arithmetic lives in the un-vendored
# com.twitter.ml.api.embedding.EmbeddingMath (ann/src/main/scala/com/twitter/ann/common/Api.scala:4-12)
# and the tree holds no test or fixture for it; this restates the published definitions the call
# sites name (Metric.scala:88-185,263-289; BruteForceIndex.scala:66-91) in float64.
# ---------------------------------------------------------------------------------------------
def dense_prepare(metric: int, x: np.ndarray) -> np.ndarray:
    """What the index stores / the query becomes: Cosine rows are L2-normalised first
    (DistanceFunctionGenerator.scala:12-30, Hnsw.scala:149-155), then everything is rounded to fp16."""
    x = np.asarray(x, np.float32)
    if metric == 1:
        norm = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
        norm[~(norm > 0)] = 1.0
        x = x / norm[:, None]
    return x.astype(np.float16).astype(np.float32)


def dense_distances(metric: int, stored: np.ndarray, q: np.ndarray) -> np.ndarray:
    """Distances of one prepared query to prepared stored vectors (Metric.scala: L2 :88-97,
    Cosine :119-126 = 1 - cos, InnerProduct :150-158 = 1 - dot), float64."""
    s, q = stored.astype(np.float64), q.astype(np.float64)
    if metric == 0:
        return np.sqrt(((s - q[None, :]) ** 2).sum(axis=1))
    return 1.0 - s @ q


def dense_bruteforce(metric: int, stored: np.ndarray, ids, queries: np.ndarray, k: int):
    """BruteForceIndex.queryWithDistance (BruteForceIndex.scala:66-91) per prepared query: the k smallest
    distances ascending; ties (which the reference's heap leaves unspecified) by id ascending."""
    ids = np.arange(len(stored), dtype=np.int64) if ids is None else np.asarray(ids, np.int64)
    out = []
    for q in queries:
        dist = dense_distances(metric, stored, q)
        order = np.lexsort((ids, dist))[:k]
        out.append((ids[order], dist[order]))
    return out


# ---------------------------------------------------------------------------------------------
# representation-scorer list / aggregate columns, restated literally (test infrastructure)
# ---------------------------------------------------------------------------------------------
def rsx_list_scores(algorithm, target, candidates):
    """ListScoreColumn.fetch (representation-scorer/.../columns/ListScoreColumn.scala:53-115): `target` and
    each candidate are (ids, scores) embeddings or None (not hydrated) -> Option[Double] per candidate."""
    out = []
    for c in candidates:
        out.append(None if target is None or c is None else pair_score(algorithm, target[0], target[1], c[0], c[1]))
    return out


def rsx_engagement_features(algorithm, cand, map_ids, map_embeddings, groups):
    """Scorer.computeSimilarityScoresPerTweet + Scorer.avg / max (twistlyfeatures/Scorer.scala:157-369,426-429)
    for one candidate embedding `cand` (or None).  map_ids[m]: the id list scored through map m (duplicates
    kept); map_embeddings[m]: id -> embedding; groups: (map, [signal ids in order]).  Returns (avg, max) per
    group with None for an empty fold."""
    results = []  # per map: list of ScoreResult(id, Option[score]) -- getTweetScores / getUserScores
    for ids, emb in zip(map_ids, map_embeddings):
        rs = []
        for i in ids:
            e = emb.get(i)
            rs.append((i, None if cand is None or e is None else pair_score(algorithm, e[0], e[1], cand[0], cand[1])))
        results.append(rs)
    out = []
    for m, signal_ids in groups:
        by_id = {}
        for i, s in results[m]:  # groupBy(_.id)
            by_id.setdefault(i, []).append(s)
        vals = [s for i in signal_ids for s in by_id.get(i, []) if s is not None]
        if not vals:
            out.append((None, None))
            continue
        total = 0.0
        for v in vals:
            total = total + v
        mx = 0.0
        for v in vals:
            mx = max(mx, v)
        out.append((total / len(vals), mx))
    return out



# ---------------------------------------------------------------------------------------------
# HNSW walk (oracle/hnsw_oracle.c).  `graph` = (entry_level int32[], entry_item int64[], entry_offsets int64[],
# entry_neighbours int64[], entry_point, max_level): the reference's Map<HnswNode, List> + HnswMeta, flattened.
# ---------------------------------------------------------------------------------------------
def hnsw_search(metric: int, stored: np.ndarray, graph, query: np.ndarray, k: int, ef: int):
    """searchKnn for one prepared query (dense_prepare) over fp16-rounded stored vectors.
    Returns (items int64[m], distances float32[m], distance evaluations)."""
    L = lib()
    L.oracle_hnsw_search.restype = C.c_int32
    lv, it, off, nb, entry, max_level = graph
    x = np.ascontiguousarray(stored, np.float32)
    q = np.ascontiguousarray(query, np.float32)
    lv = np.ascontiguousarray(lv, np.int32); it = np.ascontiguousarray(it, np.int64)
    off = np.ascontiguousarray(off, np.int64); nb = np.ascontiguousarray(nb, np.int64)
    out_i = np.zeros(k, np.int64); out_d = np.zeros(k, np.float32)
    evals = C.c_int64()
    m = L.oracle_hnsw_search(C.c_int32(metric), C.c_int64(x.shape[0]), C.c_int32(x.shape[1]), _p(x), _p(q), C.c_int64(entry),
                             C.c_int32(max_level), C.c_int64(len(lv)), _p(lv), _p(it), _p(off), _p(nb), C.c_int32(k), C.c_int32(ef),
                             _p(out_i), _p(out_d), C.byref(evals))
    return out_i[:m].copy(), out_d[:m].copy(), evals.value


def hnsw_build(metric: int, stored: np.ndarray, levels, max_m: int, ef_construction: int):
    """HnswIndex.insert for items 0 .. n-1 in order, each at its given level (oracle/hnsw_oracle.c, oracle_hnsw_build),
    over fp16-rounded stored vectors.  Returns the graph in the form hnsw_search takes, entries sorted by (level, item)."""
    L = lib()
    L.oracle_hnsw_build.restype = C.c_int64
    x = np.ascontiguousarray(stored, np.float32)
    lv_in = np.ascontiguousarray(levels, np.int32)
    n = x.shape[0]
    cap_e = n * (int(lv_in.max(initial=0)) + 1) + 1
    cap_n = cap_e * (2 * max_m + 1)
    e_lv = np.zeros(cap_e, np.int32); e_it = np.zeros(cap_e, np.int64)
    e_off = np.zeros(cap_e + 1, np.int64); e_nb = np.zeros(cap_n, np.int64)
    entry = C.c_int64(); ml = C.c_int32()
    ne = L.oracle_hnsw_build(C.c_int32(metric), C.c_int64(n), C.c_int32(x.shape[1]), _p(x), _p(lv_in), C.c_int32(max_m),
                             C.c_int32(ef_construction), C.c_int64(cap_e), C.c_int64(cap_n), _p(e_lv), _p(e_it), _p(e_off),
                             _p(e_nb), C.byref(entry), C.byref(ml))
    assert ne >= 0
    return e_lv[:ne].copy(), e_it[:ne].copy(), e_off[:ne + 1].copy(), e_nb[:e_off[ne]].copy(), entry.value, ml.value


def hnsw_build_batched(metric: int, stored: np.ndarray, levels, max_m: int, ef_construction: int, batch: int = 4096,
                       ccap: int = 1024, link_cap: int = 1024):
    """The device builder's batched insertion restated (oracle/hnsw_oracle.c, oracle_hnsw_build_batched): the graph
    hnsw_index_build_insert_gpu[_levels] must produce, in the form hnsw_search takes."""
    L = lib()
    L.oracle_hnsw_build_batched.restype = C.c_int64
    x = np.ascontiguousarray(stored, np.float32)
    lv_in = np.ascontiguousarray(levels, np.int32)
    n = x.shape[0]
    cap_e = n * (int(lv_in.max(initial=0)) + 1) + 1
    cap_n = cap_e * (2 * max_m + 1)
    e_lv = np.zeros(cap_e, np.int32); e_it = np.zeros(cap_e, np.int64)
    e_off = np.zeros(cap_e + 1, np.int64); e_nb = np.zeros(cap_n, np.int64)
    entry = C.c_int64(); ml = C.c_int32()
    ne = L.oracle_hnsw_build_batched(C.c_int32(metric), C.c_int64(n), C.c_int32(x.shape[1]), _p(x), _p(lv_in), C.c_int32(max_m),
                                     C.c_int32(ef_construction), C.c_int32(batch), C.c_int32(ccap), C.c_int32(link_cap),
                                     C.c_int64(cap_e), C.c_int64(cap_n), _p(e_lv), _p(e_it), _p(e_off), _p(e_nb), C.byref(entry),
                                     C.byref(ml))
    assert ne >= 0
    return e_lv[:ne].copy(), e_it[:ne].copy(), e_off[:ne + 1].copy(), e_nb[:e_off[ne]].copy(), entry.value, ml.value


# ---------------------------------------------------------------------------------------------
# The offline all-users job: src/scala/com/twitter/simclusters_v2/scio/bq_generation/sql/tweets_ann.sql:1-64, restated
# step by step in plain Python (small inputs only).  It is the reference's only INDEPENDENT statement of the
# algorithm (a SQL text, not the Scala operator), so besides pinning the offline scores it cross-checks the
# operator oracle: both must give a (user, tweet) pair the same dot product.
# Orders BigQuery leaves open are fixed as everywhere else: ORDER BY ... DESC ties by id ascending; SUM adds in
# ascending cluster id.
# ---------------------------------------------------------------------------------------------
def tweets_ann_sql(consumer_embeddings, tweet_embeddings, top_n, top_m, top_k):
    """consumer_embeddings: {userId: [(clusterId, userScore)]}; tweet_embeddings: {tweetId: [(clusterId, tweetScore)]}.
    Returns {userId: [(tweetId, dotProductScore, cosineSimilarityScore, logCosineSimilarityScore)]} ordered by
    logCosineSimilarityScore DESC, at most top_k entries (:55-60)."""
    import math

    # :10-15 tweet_embeddings_norm: SUM(tweetScore * tweetScore) ... HAVING norm > 0.0
    norm = {}
    for t, emb in tweet_embeddings.items():
        total = 0.0
        for _c, s in sorted(emb):
            total = total + s * s
        if total > 0.0:
            norm[t] = total
    # :17-21 top N clusters per consumer embedding, ORDER BY userScore DESC LIMIT N
    top_clusters = {u: sorted(emb, key=lambda cs: (-cs[1], cs[0]))[:max(top_n, 0)] for u, emb in consumer_embeddings.items()}
    # :23-27 top M tweets per cluster, ORDER BY tweetScore DESC LIMIT M
    by_cluster = {}
    for t, emb in tweet_embeddings.items():
        for c, s in emb:
            by_cluster.setdefault(c, []).append((t, s))
    cluster_tweets = {c: sorted(v, key=lambda ts: (-ts[1], ts[0]))[:max(top_m, 0)] for c, v in by_cluster.items()}
    out = {}
    for u, clusters in top_clusters.items():
        # :29-37 join, :39-46 SUM(userScore * tweetScore) GROUP BY userId, tweetId
        dot = {}
        for c, user_score in sorted(clusters):  # ascending cluster id = the summation order
            for t, tweet_score in cluster_tweets.get(c, []):
                dot[t] = dot.get(t, 0.0) + user_score * tweet_score
        # :47-54 similarity scores; the JOIN drops tweets without a norm row
        rows = []
        for t, d in dot.items():
            if t not in norm:
                continue
            rows.append((t, d, d / math.sqrt(norm[t]), d / strict_log(1 + norm[t])))
        # :55-60 top K ORDER BY logCosineSimilarityScore DESC
        rows.sort(key=lambda r: (-r[3], r[0]))
        out[u] = rows[:max(top_k, 0)]
    return out


def strict_exp(x: float) -> float:
    L = lib()
    L.oracle_strict_exp.restype = C.c_double
    L.oracle_strict_exp.argtypes = [C.c_double]
    return float(L.oracle_strict_exp(float(x)))


def store_list(tweet_ids, values, scaled_times, now_scaled, max_results):
    """One cluster's posting list as TopKTweetsForClusterReadableStore + the provider return it: decay to now,
    keep > 0, sort by score descending (ties tweet id ascending), take."""
    L = lib()
    L.oracle_store_list.restype = C.c_int32
    L.oracle_store_list.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]
    t = np.ascontiguousarray(tweet_ids, np.int64)
    v = np.ascontiguousarray(values, np.float64)
    st = None if scaled_times is None else np.ascontiguousarray(scaled_times, np.float64)
    out_i = np.zeros(len(t) + 1, np.int64)
    out_s = np.zeros(len(t) + 1, np.float64)
    m = L.oracle_store_list(len(t), _p(t), _p(v), _p(st), float(now_scaled), int(max_results), _p(out_i), _p(out_s))
    return out_i[:m].copy(), out_s[:m].copy()


def topk_merge(a, b, top_k: int, threshold: float, oldest_tweet_id: int):
    """TopKTweetsWithScoresMonoid.plus for ONE cluster, restated literally with dicts (small inputs only):
    src/scala/com/twitter/simclusters_v2/summingbird/common/Monoids.scala:131-158 (plus, age filter :142,154) and
    :378-450 (TopKScoresUtils.mergeTwoTopKMapWithDecayedValues).  a, b: {tweetId: (value, scaledTime)} or None.
    ThriftDecayedValueMonoid.plus(v, DecayedValue(0.0, latest)) is algebird's DecayedValueMonoid(0.0).plus (un-vendored;
    restated in oracle_decay_to_timestamp, zero = DecayedValue(0.0, -inf)).  The cut's order among equal values is the
    HashMap's in the reference; here (value desc, tweet id asc), as everywhere."""
    L = lib()
    L.oracle_decay_to_timestamp.restype = C.c_double
    L.oracle_decay_to_timestamp.argtypes = [C.c_double, C.c_double, C.c_double]

    def age(m):
        return None if m is None else {k: v for k, v in m.items() if k >= oldest_tweet_id}  # :154

    if a is None or len(a) == 0:        # :388-390
        return age(b)
    if b is None or len(b) == 0:        # :392-394
        return age(a)
    latest = max(t for _v, t in list(a.values()) + list(b.values()))  # :396-399

    def decayed(value, t):              # :409-410
        nv = L.oracle_decay_to_timestamp(float(value), float(t), float(latest))
        return (nv, latest) if nv != 0.0 else (0.0, float("-inf"))

    merged = {}
    for k, (v, t) in a.items():         # :405-417
        d = decayed(v, t)
        if d[0] > threshold:
            merged[k] = d
    for k, (v, t) in b.items():         # :419-438
        d = decayed(v, t)
        if d[0] > threshold:
            if k not in merged:
                merged[k] = d
            elif d[0] > merged[k][0]:
                merged[k] = d
    if len(merged) > top_k * 1.2:       # :441-448
        merged = dict(sorted(merged.items(), key=lambda kv: (-kv[1][0], kv[0]))[:top_k])
    return age(merged)


def strict_log(x: float) -> float:
    return float(lib().oracle_strict_log(float(x)))


# ---------------------------------------------------------------------------------------------
# representation-scorer `simClustersRecentEngagementSimilarity`: the whole feature computation from RAW user signals,
# restated literally and independently of the product's Engagements mirror (round-1 review: the test fed the oracle the
# product's own window grouping).  Paths relative to /root/reference/representation-scorer/server/src/main/scala/com/
# twitter/representationscorer/twistlyfeatures/.
# ---------------------------------------------------------------------------------------------
_DAY_MS = 86_400_000
# UserSignalServiceRecentEngagementsClient.scala:38-52: Engagements field <- (SignalType, earliest valid = now - days)
_USS_FIELDS = (("favs7d", "TweetFavorite", 7), ("retweets7d", "Retweet", 7), ("follows30d", "AccountFollowWithDelay", 30),
               ("shares7d", "TweetShareV1", 7), ("replies7d", "Reply", 7), ("originalTweets7d", "OriginalTweet", 7),
               ("videoPlaybacks7d", "VideoView90dPlayback50V1", 7), ("block30d", "AccountBlock", 30),
               ("mute30d", "AccountMute", 30), ("report30d", "TweetReport", 30), ("dontlike30d", "TweetDontLike", 30),
               ("seeFewer30d", "TweetSeeFewer", 30))
# Scorer.scala:306-369: SimClustersRecentEngagementSimilarities field prefix <- (Engagements value, score map), in the
# constructor's order.  block* / mute* read the TWEET score map although their targets are authors (:232-260): restated
# as written.
_FEATURES = (("fav1d", "favs1d", "tweet"), ("fav7d", "favs7d", "tweet"), ("retweet1d", "retweets1d", "tweet"),
             ("retweet7d", "retweets7d", "tweet"), ("follow7d", "follows7d", "author"), ("follow30d", "follows30d", "author"),
             ("share1d", "shares1d", "tweet"), ("share7d", "shares7d", "tweet"), ("reply1d", "replies1d", "tweet"),
             ("reply7d", "replies7d", "tweet"), ("originalTweet1d", "originalTweets1d", "tweet"),
             ("originalTweet7d", "originalTweets7d", "tweet"), ("videoPlayback1d", "videoPlaybacks1d", "tweet"),
             ("videoPlayback7d", "videoPlaybacks7d", "tweet"), ("block1d", "block1d", "tweet"), ("block7d", "block7d", "tweet"),
             ("block30d", "block30d", "tweet"), ("mute1d", "mute1d", "tweet"), ("mute7d", "mute7d", "tweet"),
             ("mute30d", "mute30d", "tweet"), ("report1d", "report1d", "tweet"), ("report7d", "report7d", "tweet"),
             ("report30d", "report30d", "tweet"), ("dontlike1d", "dontlike1d", "tweet"), ("dontlike7d", "dontlike7d", "tweet"),
             ("dontlike30d", "dontlike30d", "tweet"), ("seeFewer1d", "seeFewer1d", "tweet"), ("seeFewer7d", "seeFewer7d", "tweet"),
             ("seeFewer30d", "seeFewer30d", "tweet"))


def rsx_engagements_from_signals(signal_response, now_ms):
    """UserSignalServiceRecentEngagementsClient.get / getUserSignals (:30-71) followed by the Engagements constructor
    (Engagements.scala:21-56).  signal_response: {SignalType name: [(targetId or None, timestamp ms), ...]} in the
    order the signal service returned them; a None target stands for a non-Long internal id (:66-67).
    Returns {value name: [(targetId, timestamp)]} for the 12 fields and every derived window."""
    e = {}
    for field, signal_type, days in _USS_FIELDS:
        earliest = now_ms - days * _DAY_MS
        kept = [(t, ts) for t, ts in signal_response.get(signal_type, []) if ts > earliest and t is not None]  # :63-67
        e[field] = kept[:10]  # .take(EngagementsToScore), :68 / :128
    one_day_ago, seven_days_ago = now_ms - _DAY_MS, now_ms - 7 * _DAY_MS  # Engagements.scala:23-25

    def since(xs, cut):
        return [s for s in xs if s[1] > cut]
    e["dontlike7d"] = since(e["dontlike30d"], seven_days_ago)  # :36-37
    e["seeFewer7d"] = since(e["seeFewer30d"], seven_days_ago)
    for a, b in (("favs1d", "favs7d"), ("retweets1d", "retweets7d"), ("shares1d", "shares7d"), ("replies1d", "replies7d"),
                 ("originalTweets1d", "originalTweets7d"), ("videoPlaybacks1d", "videoPlaybacks7d"), ("dontlike1d", "dontlike7d"),
                 ("seeFewer1d", "seeFewer7d")):  # :39-46
        e[a] = since(e[b], one_day_ago)
    for a, b in (("follows7d", "follows30d"), ("block7d", "block30d"), ("mute7d", "mute30d"), ("report7d", "report30d")):  # :49-52
        e[a] = since(e[b], seven_days_ago)
    for a, b in (("block1d", "block7d"), ("mute1d", "mute7d"), ("report1d", "report7d")):  # :54-56
        e[a] = since(e[b], one_day_ago)
    # :28-33
    e["tweetIds"] = [t for f in ("favs7d", "retweets7d", "shares7d", "replies7d", "originalTweets7d", "videoPlaybacks7d",
                                 "report30d", "dontlike30d", "seeFewer30d") for t, _ in e[f]]
    e["authorIds"] = [t for f in ("follows30d", "block30d", "mute30d") for t, _ in e[f]]
    return e


def rsx_scorer_features(signal_response, now_ms, cand_emb, tweet_embeddings, author_embeddings, algorithm=2):
    """Scorer.get for ONE candidate tweet (Scorer.scala:125-149,157-369,426-429): {feature name: Optional[float]}, 58 names.
    cand_emb: the candidate's (ids, scores) embedding or None; *_embeddings: id -> embedding."""
    e = rsx_engagements_from_signals(signal_response, now_ms)

    def score_results(ids, embs):  # getTweetScores / getUserScores: one ScoreResult(id, Option[score]) per requested id
        out = []
        for i in ids:
            s = embs.get(i)
            out.append((i, None if cand_emb is None or s is None else pair_score(algorithm, s[0], s[1], cand_emb[0], cand_emb[1])))
        return out
    maps = {}
    for name, ids, embs in (("tweet", e["tweetIds"], tweet_embeddings), ("author", e["authorIds"], author_embeddings)):
        grouped = {}
        for i, s in score_results(ids, embs):  # .groupBy(_.id), :141-142
            grouped.setdefault(i, []).append(s)
        maps[name] = grouped
    feats = {}
    for prefix, value, which in _FEATURES:
        # engagements.<value>.view.flatMap(s => scores.get(s.targetId)).flatten.flatMap(_.score).force
        vals = [s for t, _ts in e[value] for s in maps[which].get(t, []) if s is not None]
        if not vals:
            feats[prefix + "Last10Max"] = feats[prefix + "Last10Avg"] = None
            continue
        total = 0.0
        for v in vals:  # s.sum / s.size, :427
            total = total + v
        mx = 0.0
        for v in vals:  # foldLeft(0.0)(math.max), :429
            mx = max(mx, v)
        feats[prefix + "Last10Avg"] = total / len(vals)
        feats[prefix + "Last10Max"] = mx
    return feats
