#!/usr/bin/env python3
"""Writes tests/golden/sann_kat.json: hand-derived known-answer vectors for the SANN and RSX
arithmetic.  The reference ships no fixtures for this path, so every
expected value below is an explicit expression a reader can check against the cited formula;
nothing here calls the oracle or the product.

Formulas (paths relative to /root/reference/):
  accumulate   dot[t] += s*w_c ; nsq[t] += s*s        ApproximateCosineSimilarity.scala:92-96
  Dot          dot                                      :118
  Cosine       dot / l2norm / sqrt(nsq)                 :114-115
  LogCosine    dot / logNorm / log(1 + nsq)             :112-113
  NoSrcNorm    dot / sqrt(nsq)                          :116-117
  l2norm = sqrt(sum x^2), logNorm = log(sum x^2 + 1)    CosineSimilarityUtil.scala:29-31,43-45
  window       earliest = firstIdFor(now - maxAge h) (0 when maxAge >= 175200), latest = firstIdFor(now - minAge h)
               firstIdFor(ms) = (ms - 1288834974657) << 22                :65-72, BQGenerationUtil.scala:150-153

Doubles are stored as C99 hex strings (float.hex) so the fixture is bit-exact.  Values that go
through math.log carry "ulp": 1 (java.lang.Math.log is only specified to 1 ulp).
"""
import json
import math
import os

EPOCH = 1288834974657
NOW = 1_700_000_000_000


def first_id(ms):
    return (ms - EPOCH) << 22


def hx(x):
    return float(x).hex()


def cfg(k=10, min_score=0.0, M=800, N=50, max_age=175200, min_age=0, alg=2):
    return dict(maxNumResults=k, minScore=min_score, candidateEmbeddingType=0, maxTopTweetsPerCluster=M,
                maxScanClusters=N, maxTweetCandidateAgeHours=max_age, minTweetCandidateAgeHours=min_age, annAlgorithm=alg)


cases = []

# --- 1. the SURVEY 8(c) example: source {1:3, 2:4}; lists 1->[(10,2),(11,1)], 2->[(10,1),(12,5)]
emb = [[1, 3.0], [2, 4.0]]
lists = {"1": [[10, 2.0], [11, 1.0]], "2": [[10, 1.0], [12, 5.0]]}
l2 = math.sqrt(3.0 * 3.0 + 4.0 * 4.0)           # 5
ln = math.log(3.0 * 3.0 + 4.0 * 4.0 + 1)        # ln 26
d10, n10 = 2.0 * 3.0 + 1.0 * 4.0, 2.0 * 2.0 + 1.0 * 1.0   # 10, 5
d11, n11 = 1.0 * 3.0, 1.0 * 1.0                           # 3, 1
d12, n12 = 5.0 * 4.0, 5.0 * 5.0                           # 20, 25
cases.append(dict(name="survey_dot", emb=emb, lists=lists, source=None, variant=0, now_ms=NOW, config=cfg(alg=1),
                  expect=[[12, hx(d12)], [10, hx(d10)], [11, hx(d11)]], map_size=3, ulp=0))
cases.append(dict(name="survey_cosine", emb=emb, lists=lists, source=None, variant=0, now_ms=NOW, config=cfg(alg=2),
                  expect=[[10, hx(d10 / l2 / math.sqrt(n10))], [12, hx(d12 / l2 / math.sqrt(n12))],
                          [11, hx(d11 / l2 / math.sqrt(n11))]], map_size=3, ulp=0))
cases.append(dict(name="survey_logcosine", emb=emb, lists=lists, source=None, variant=0, now_ms=NOW, config=cfg(alg=3),
                  # 20/ln26/ln26 = 1.884 > 10/ln26/ln6 = 1.713 > 3/ln26/ln2 = 1.328
                  expect=[[12, hx(d12 / ln / math.log(1 + n12))], [10, hx(d10 / ln / math.log(1 + n10))],
                          [11, hx(d11 / ln / math.log(1 + n11))]], map_size=3, ulp=4))
cases.append(dict(name="survey_nosrcnorm", emb=emb, lists=lists, source=None, variant=0, now_ms=NOW, config=cfg(alg=4),
                  expect=[[10, hx(d10 / math.sqrt(n10))], [12, hx(d12 / math.sqrt(n12))], [11, hx(d11 / math.sqrt(n11))]],
                  map_size=3, ulp=0))

# --- 2. maxTopTweetsPerCluster = 1: only the first posting of every list is read (:87)
cases.append(dict(name="m_cut", emb=emb, lists=lists, source=None, variant=0, now_ms=NOW, config=cfg(alg=1, M=1),
                  expect=[[10, hx(2.0 * 3.0 + 1.0 * 4.0)]], map_size=1, ulp=0))

# --- 3. maxNumResults = 2 and minScore (:125-127)
cases.append(dict(name="take_k", emb=emb, lists=lists, source=None, variant=0, now_ms=NOW, config=cfg(alg=1, k=2),
                  expect=[[12, hx(20.0)], [10, hx(10.0)]], map_size=3, ulp=0))
cases.append(dict(name="min_score", emb=emb, lists=lists, source=None, variant=0, now_ms=NOW,
                  config=cfg(alg=1, min_score=10.0),
                  expect=[[12, hx(20.0)], [10, hx(10.0)]], map_size=3, ulp=0))

# --- 4. source tweet exclusion (:90): source id 10 is a tweet id
cases.append(dict(name="source_excluded", emb=emb, lists=lists, source=10, variant=0, now_ms=NOW, config=cfg(alg=1),
                  expect=[[12, hx(20.0)], [11, hx(3.0)]], map_size=2, ulp=0))

# --- 5. tweet id 0 with a non-tweet source: kept by "original", dropped by "optimized"/"experimental"
lists0 = {"1": [[0, 2.0], [11, 1.0]]}
cases.append(dict(name="id0_original", emb=[[1, 3.0]], lists=lists0, source=None, variant=0, now_ms=NOW, config=cfg(alg=1),
                  expect=[[0, hx(6.0)], [11, hx(3.0)]], map_size=2, ulp=0))
cases.append(dict(name="id0_optimized", emb=[[1, 3.0]], lists=lists0, source=None, variant=1, now_ms=NOW, config=cfg(alg=1),
                  expect=[[11, hx(3.0)]], map_size=1, ulp=0))
cases.append(dict(name="id0_experimental", emb=[[1, 3.0]], lists=lists0, source=None, variant=2, now_ms=NOW, config=cfg(alg=1),
                  expect=[[11, hx(3.0)]], map_size=1, ulp=0))

# --- 6. age window: now = NOW, maxAge 24 h, minAge 1 h
t_old = first_id(NOW - 25 * 3600_000) + 5        # older than 24 h  -> dropped
t_edge_lo = first_id(NOW - 24 * 3600_000)        # == earliest      -> kept (>=)
t_mid = first_id(NOW - 2 * 3600_000) + 123       # inside
t_edge_hi = first_id(NOW - 1 * 3600_000)         # == latest        -> kept (<=)
t_new = first_id(NOW - 1 * 3600_000) + 1         # newer than latest-> dropped
cases.append(dict(name="age_window", emb=[[7, 2.0]], source=None, variant=0, now_ms=NOW,
                  lists={"7": [[t_old, 5.0], [t_edge_lo, 4.0], [t_mid, 3.0], [t_edge_hi, 2.0], [t_new, 1.0]]},
                  config=cfg(alg=1, max_age=24, min_age=1),
                  expect=[[t_edge_lo, hx(8.0)], [t_mid, hx(6.0)], [t_edge_hi, hx(4.0)]], map_size=3, ulp=0))

# --- 7. exact ties: all single-cluster candidates of one cluster under Cosine score w/l2/1 * ... ;
#        with s = 1, 2, 4 (powers of two) the three scores are exactly equal -> tweet id ascending
cases.append(dict(name="tie_break_id_asc", emb=[[3, 1.0]], source=None, variant=0, now_ms=NOW,
                  lists={"3": [[33, 4.0], [22, 2.0], [11, 1.0]]}, config=cfg(alg=2, k=2),
                  expect=[[11, hx(1.0)], [22, hx(1.0)]], map_size=3, ulp=0))

# --- 8. constructor semantics: non-positive scores dropped, clusters missing from the index ignored
cases.append(dict(name="ctor_and_missing", emb=[[1, 3.0], [2, 0.0], [9, -1.0], [5, 4.0]], source=None, variant=0,
                  now_ms=NOW, lists={"1": [[10, 2.0]], "2": [[77, 9.0]]}, config=cfg(alg=2),
                  # embedding = {1:3, 5:4}; l2norm 5; cluster 5 has no list (None); cluster 2 not in embedding
                  expect=[[10, hx(2.0 * 3.0 / 5.0 / math.sqrt(4.0))]], map_size=1, ulp=0))
cases.append(dict(name="empty_embedding", emb=[], source=None, variant=0, now_ms=NOW, lists=lists, config=cfg(alg=2),
                  expect=[], map_size=0, ulp=0))

# --- 9. maxScanClusters: top-N clusters by score, ties by cluster id asc (SimClustersEmbedding.scala:456-463,377-392)
#        embedding {4:1, 2:1, 9:5}: desc order = 9, 2, 4 ; N = 2 scans {9, 2}; norms use the FULL embedding
l2f = math.sqrt(1.0 * 1.0 + 1.0 * 1.0 + 5.0 * 5.0)  # sortedScores order by cluster id: 2,4,9
cases.append(dict(name="max_scan_clusters", emb=[[4, 1.0], [2, 1.0], [9, 5.0]], source=None, variant=0, now_ms=NOW,
                  lists={"2": [[20, 1.0]], "4": [[40, 1.0]], "9": [[90, 1.0]]}, config=cfg(alg=2, N=2),
                  expect=[[90, hx(5.0 / l2f / 1.0)], [20, hx(1.0 / l2f / 1.0)]], map_size=2, ulp=0))

# --- 10. explicit clusterTweetsMap keys incl. a cluster the embedding lacks:
#         original skips it (:84); experimental gives it weight 0.0 (Experimental :62-63)
cases.append(dict(name="explicit_keys_original", emb=[[1, 3.0]], source=None, variant=0, now_ms=NOW, scan_keys=[2, 1],
                  lists=lists, config=cfg(alg=1), expect=[[10, hx(6.0)], [11, hx(3.0)]], map_size=2, ulp=0))
cases.append(dict(name="explicit_keys_experimental", emb=[[1, 3.0]], source=None, variant=2, now_ms=NOW, scan_keys=[2, 1],
                  lists=lists, config=cfg(alg=4),
                  # t10: dot = 1*0 + 2*3 = 6, nsq = 1 + 4 = 5 ; t12: dot = 0, nsq = 25 ; t11: dot 3, nsq 1
                  expect=[[11, hx(3.0 / math.sqrt(1.0))], [10, hx((1.0 * 0.0 + 2.0 * 3.0) / math.sqrt(1.0 * 1.0 + 2.0 * 2.0))],
                          [12, hx(0.0 / math.sqrt(25.0))]], map_size=3, ulp=0))

# --- 11. accumulation order matters in the last ulp: three clusters, explicit order
a, b, c = 0.1 * 3.0, 0.2 * 3.0, 0.3 * 3.0
cases.append(dict(name="accum_order_123", emb=[[1, 3.0], [2, 3.0], [3, 3.0]], source=None, variant=0, now_ms=NOW,
                  scan_keys=[1, 2, 3], lists={"1": [[5, 0.1]], "2": [[5, 0.2]], "3": [[5, 0.3]]}, config=cfg(alg=1),
                  expect=[[5, hx((0.0 + a + b) + c)]], map_size=1, ulp=0))
cases.append(dict(name="accum_order_321", emb=[[1, 3.0], [2, 3.0], [3, 3.0]], source=None, variant=0, now_ms=NOW,
                  scan_keys=[3, 2, 1], lists={"1": [[5, 0.1]], "2": [[5, 0.2]], "3": [[5, 0.3]]}, config=cfg(alg=1),
                  expect=[[5, hx((0.0 + c + b) + a)]], map_size=1, ulp=0))

# --- 12. legacy candidate source (variant 3), simclusters_v2/candidate_source/SimClustersANNCandidateSource.scala:107-181
#     ann_algorithm carries (enablePartialNormalization, rankingAlgorithm): 1 = raw dot, 2 = dot/l2norm/sqrt(nsq),
#     3 = dot / l2norm / log(1 + nsq)  -- l2norm, not logNorm (:167-169); no minScore filter (:177-180)
emb = [[1, 3.0], [2, 4.0]]
lists = {"1": [[10, 2.0], [11, 1.0]], "2": [[10, 1.0], [12, 5.0]]}
cases.append(dict(name="legacy_log_divides_by_l2norm", emb=emb, lists=lists, source=None, variant=3, now_ms=NOW,
                  config=cfg(alg=3),  # default maxAge 175200 h: now - 20 y is before the Snowflake epoch, lower bound < 0
                  # 20/5/ln26 = 1.2277 > 10/5/ln6 = 1.1162 > 3/5/ln2 = 0.8656
                  expect=[[12, hx(d12 / l2 / math.log(1 + n12))], [10, hx(d10 / l2 / math.log(1 + n10))],
                          [11, hx(d11 / l2 / math.log(1 + n11))]], map_size=3, ulp=4))
cases.append(dict(name="legacy_ignores_min_score", emb=emb, lists=lists, source=None, variant=3, now_ms=NOW,
                  config=cfg(alg=2, min_score=0.95),
                  expect=[[10, hx(d10 / l2 / math.sqrt(n10))], [12, hx(d12 / l2 / math.sqrt(n12))],
                          [11, hx(d11 / l2 / math.sqrt(n11))]], map_size=3, ulp=0))
cases.append(dict(name="legacy_no_normalisation", emb=emb, lists=lists, source=11, variant=3, now_ms=NOW,
                  config=cfg(alg=1, k=1),
                  expect=[[12, hx(d12)]], map_size=2, ulp=0))   # source tweet 11 excluded (:139), take(1)
# no "175200 h = unbounded" rule (:113): with now far enough in the future the lower bound is a real id.
# now - 175200 h = EPOCH + 1000 ms -> earliest = 1000 << 22; the same case under variant 0 keeps both tweets.
FUTURE = EPOCH + 175200 * 3600_000 + 1000
old_t, new_t = (999 << 22) + 5, (1000 << 22)
lists_w = {"1": [[old_t, 2.0], [new_t, 1.0]]}
cases.append(dict(name="legacy_window_has_a_lower_bound", emb=[[1, 3.0]], lists=lists_w, source=None, variant=3, now_ms=FUTURE,
                  config=cfg(alg=1, max_age=175200), expect=[[new_t, hx(1.0 * 3.0)]], map_size=1, ulp=0))
cases.append(dict(name="original_175200h_is_unbounded", emb=[[1, 3.0]], lists=lists_w, source=None, variant=0, now_ms=FUTURE,
                  config=cfg(alg=1, max_age=175200), expect=[[old_t, hx(2.0 * 3.0)], [new_t, hx(1.0 * 3.0)]], map_size=2, ulp=0))

# --- RSX pair scores (SimClustersEmbedding.scala:194-224,235-243,301-321; score.thrift:14-22)
A = [[1, 3.0], [2, 4.0]]
B = [[2, 1.0], [3, 2.0], [1, 2.0]]
nA = math.sqrt(9.0 + 16.0)
nB = math.sqrt(2.0 * 2.0 + 1.0 * 1.0 + 2.0 * 2.0)        # sortedScores by id: 1:2, 2:1, 3:2
lA, lB = math.log(25.0 + 1), math.log(9.0 + 1)
eA, eB = math.pow(25.0, 0.3), math.pow(9.0, 0.3)
pairs = [
    dict(name="pair_dot", alg=1, a=A, b=B, expect=hx(3.0 * 2.0 + 4.0 * 1.0), ulp=0),
    # cosine = merge-dot of PRE-NORMALISED arrays (:202-208), not dot/(|a||b|)
    dict(name="pair_cosine", alg=2, a=A, b=B, expect=hx((3.0 / nA) * (2.0 / nB) + (4.0 / nA) * (1.0 / nB)), ulp=0),
    dict(name="pair_jaccard", alg=3, a=A, b=B, expect=hx(2 / 3), ulp=0),
    dict(name="pair_euclid", alg=4, a=A, b=B, expect=hx(math.sqrt((3.0 - 2.0) ** 2 + (4.0 - 1.0) ** 2 + (0.0 - 2.0) ** 2)), ulp=0),
    dict(name="pair_manhattan", alg=5, a=A, b=B, expect=hx(abs(3.0 - 2.0) + abs(4.0 - 1.0) + abs(0.0 - 2.0)), ulp=0),
    dict(name="pair_logcosine", alg=6, a=A, b=B, expect=hx((3.0 / lA) * (2.0 / lB) + (4.0 / lA) * (1.0 / lB)), ulp=8),
    dict(name="pair_expscaled", alg=7, a=A, b=B, expect=hx((3.0 / eA) * (2.0 / eB) + (4.0 / eA) * (1.0 / eB)), ulp=8),
    dict(name="pair_empty_jaccard", alg=3, a=[], b=B, expect=hx(0.0), ulp=0),
    dict(name="pair_disjoint_cosine", alg=2, a=[[1, 1.0]], b=[[2, 1.0]], expect=hx(0.0), ulp=0),
]

snowflake = [dict(ms=NOW, id=first_id(NOW)), dict(ms=EPOCH, id=0), dict(ms=EPOCH + 1, id=1 << 22)]

out = dict(comment="hand-derived; see make_sann_kat.py", sann=cases, pairs=pairs, snowflake=snowflake)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sann_kat.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path, len(cases), "sann cases,", len(pairs), "pair cases")
