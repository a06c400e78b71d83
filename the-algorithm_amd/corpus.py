"""Synthetic SimClusters corpus and query generator, numpy, deterministic.

Shapes and constants come from the reference (paths relative to /root/reference/):
  144,428 clusters                 simclusters-ann/.../SimclustersAnnWarmupHandler.scala:33
  ~25 clusters per tweet, cap 50   src/scala/com/twitter/simclusters_v2/summingbird/storm/TweetJob.scala:74,
                                   .../scio/bq_generation/simclusters_index_generation/Config.scala:54
  index cap 2000 tweets/cluster    .../simclusters_index_generation/Config.scala:58
  50 clusters per user embedding   .../summingbird/common/Configs.scala:43
  Snowflake id layout              .../scio/bq_generation/common/BQGenerationUtil.scala:150-153
  posting-list materialisation     .../summingbird/stores/TopKTweetsForClusterReadableStore.scala:211-229
                                   (filter > 0, sort by score descending, take)

Cluster popularity is Zipf(s=1) realised as the log-uniform law P(rank r) = ln(1+1/r)/ln(C+1),
r = floor((C+1)^u); ranks map to cluster ids through a fixed permutation so that hot clusters
are spread over the id range (matters for cluster-range sharding).

This generator serves tests and small benchmarks (<= a few million tweets).  The 100M-tweet
corpus is generated on the device by the library (sann_corpus_*), following the same law.
"""
from __future__ import annotations

import dataclasses
from typing import Tuple

import numpy as np

N_CLUSTERS = 144_428
NOW_MS = 1_700_000_000_000
SNOWFLAKE_EPOCH_MS = 1_288_834_974_657
CORPUS_SEED = 20260104
QUERY_SEED = 20260105


def zipf_ranks(rng: np.random.Generator, n: int, n_clusters: int) -> np.ndarray:
    u = rng.random(n)
    r = np.floor(np.exp(u * np.log(n_clusters + 1.0))).astype(np.int64)
    return np.clip(r, 1, n_clusters)  # 1-based rank


def perm_multiplier(n_clusters: int) -> int:
    """Multiplier A of the rank -> cluster map `cluster = 1 + (rank * A) % C`, gcd(A, C) = 1.
    The same rule is used by the device generator (csrc/sann_corpus.hip)."""
    import math

    a = 2654435761 % n_clusters
    if a == 0:
        a = 1
    while math.gcd(a, n_clusters) != 1:
        a += 1
    return a


def cluster_permutation(n_clusters: int, seed: int = 7) -> np.ndarray:
    """rank (1-based, index r-1) -> cluster id in [1, n_clusters]; spreads the popular clusters over
    the id range.  (`seed` is kept for call compatibility; the map is the fixed multiplicative one
    so that host- and device-generated corpora and the query generator agree on popularity.)"""
    r = np.arange(1, n_clusters + 1, dtype=np.int64)
    return (1 + (r * perm_multiplier(n_clusters)) % n_clusters).astype(np.int32)


@dataclasses.dataclass
class Corpus:
    n_tweets: int
    n_clusters: int
    now_ms: int
    # CSR posting lists: the ReadableStore[ClusterId, Seq[(TweetId, Double)]] contents
    cluster_ids: np.ndarray   # int32 ascending (only clusters with a non-empty list)
    list_offsets: np.ndarray  # int64 [n+1]
    tweet_ids: np.ndarray     # int64
    scores: np.ndarray        # float64
    # full tweet embeddings (for quality recall): CSR by tweet
    tweet_id_of: np.ndarray       # int64 [T] snowflake id of tweet t
    tweet_emb_offsets: np.ndarray  # int64 [T+1]
    tweet_emb_clusters: np.ndarray  # int32
    tweet_emb_scores: np.ndarray    # float64

    def list_of(self, cluster: int) -> Tuple[np.ndarray, np.ndarray]:
        i = int(np.searchsorted(self.cluster_ids, cluster))
        if i >= len(self.cluster_ids) or self.cluster_ids[i] != cluster:
            return np.empty(0, np.int64), np.empty(0, np.float64)
        b, e = self.list_offsets[i], self.list_offsets[i + 1]
        return self.tweet_ids[b:e], self.scores[b:e]


def topic_cluster_ranks(rng: np.random.Generator, topics: np.ndarray, n_clusters: int, n_topics: int, topic_size: int,
                        affinity: float) -> np.ndarray:
    """Popularity ranks (1-based) of cluster draws under the TOPIC-MIXTURE variant: draw j belongs to an entity whose
    topic is topics[j]; with probability `affinity` the cluster comes from that topic's own `topic_size` clusters (its
    i-th cluster has global rank topic + (i - 1) * n_topics + 1: every topic owns popular and rare clusters alike;
    i is Zipf within the topic), else from the global Zipf law.  SURVEY 8(d)'s corpus draws a tweet's clusters
    independently of each other, which makes the operator's partial cosine unrelated to the full cosine
    (recall_at_400_quality 0.05); real SimClusters embeddings are topical, and this variant restores that."""
    n = len(topics)
    own = rng.random(n) < affinity
    i = np.clip(np.floor(np.exp(rng.random(n) * np.log(topic_size + 1.0))).astype(np.int64), 1, topic_size)
    in_topic = (topics + (i - 1) * n_topics) % n_clusters + 1
    return np.where(own, in_topic, zipf_ranks(rng, n, n_clusters))


def make_corpus(n_tweets: int, n_clusters: int = N_CLUSTERS, *, seed: int = CORPUS_SEED, index_cap: int = 2000,
                now_ms: int = NOW_MS, window_hours: int = 24, mean_clusters: float = 25.0,
                max_clusters_per_tweet: int = 50, n_topics: int = 0, topic_size: int = 64, affinity: float = 0.9) -> Corpus:
    """n_topics = 0: SURVEY 8(d)'s corpus (independent Zipf draws).  n_topics > 0: the topic-mixture variant
    (topic_cluster_ranks), every tweet in one Zipf-drawn topic."""
    rng = np.random.default_rng(seed)
    perm = cluster_permutation(n_clusters)
    # tweet ids: unique Snowflake ids uniform over the window before now_ms
    span = window_hours * 3_600_000
    ms = now_ms - 1 - rng.integers(0, span, size=n_tweets, dtype=np.int64)
    tid = ((ms - SNOWFLAKE_EPOCH_MS) << 22) | rng.integers(0, 1 << 22, size=n_tweets, dtype=np.int64)
    tid = np.unique(tid)
    while len(tid) < n_tweets:  # astronomically rare; keep ids unique
        extra = ((now_ms - 1 - rng.integers(0, span, size=n_tweets - len(tid), dtype=np.int64) - SNOWFLAKE_EPOCH_MS) << 22) | \
            rng.integers(0, 1 << 22, size=n_tweets - len(tid), dtype=np.int64)
        tid = np.unique(np.concatenate([tid, extra]))
    tid = rng.permutation(tid)
    # clusters per tweet: min(cap, 1 + Geom(p)) with mean ~ mean_clusters
    n_t = np.minimum(max_clusters_per_tweet, rng.geometric(1.0 / mean_clusters, size=n_tweets)).astype(np.int64)
    tw = np.repeat(np.arange(n_tweets, dtype=np.int64), n_t)
    if n_topics > 0:
        t_topic = zipf_ranks(rng, n_tweets, n_topics) - 1
        cl = perm[topic_cluster_ranks(rng, t_topic[tw], n_clusters, n_topics, topic_size, affinity) - 1]
    else:
        cl = perm[zipf_ranks(rng, len(tw), n_clusters) - 1]
    sc = np.maximum(np.exp(rng.normal(-2.0, 1.0, size=len(tw))), 0.001)
    # distinct clusters per tweet: drop repeated (tweet, cluster) draws
    key = tw * np.int64(n_clusters + 1) + cl
    _, first = np.unique(key, return_index=True)
    first.sort()
    tw, cl, sc = tw[first], cl[first], sc[first]
    # full tweet embeddings (CSR by tweet; `tw` is non-decreasing after the sort above)
    counts = np.bincount(tw, minlength=n_tweets)
    t_off = np.zeros(n_tweets + 1, np.int64)
    np.cumsum(counts, out=t_off[1:])
    # posting lists: per cluster sort by score desc (ties tweet id asc), cap
    t_ids = tid[tw]
    order = np.lexsort((t_ids, -sc, cl))
    cl_s, tid_s, sc_s = cl[order], t_ids[order], sc[order]
    uniq, start, cnt = np.unique(cl_s, return_index=True, return_counts=True)
    pos = np.arange(len(cl_s)) - np.repeat(start, cnt)
    keep = pos < index_cap
    kept_cnt = np.minimum(cnt, index_cap)
    offs = np.zeros(len(uniq) + 1, np.int64)
    np.cumsum(kept_cnt, out=offs[1:])
    return Corpus(n_tweets, n_clusters, now_ms, uniq.astype(np.int32), offs, np.ascontiguousarray(tid_s[keep]),
                  np.ascontiguousarray(sc_s[keep]), tid, t_off, cl.astype(np.int32), sc)


def make_queries(n_queries: int, n_clusters: int = N_CLUSTERS, *, seed: int = QUERY_SEED, clusters_per_user: int = 50,
                 n_topics: int = 0, topic_size: int = 64, affinity: float = 0.9, topics_per_user: int = 2):
    """User embeddings: `clusters_per_user` distinct Zipf-drawn clusters, scores exp(N(0,1)).
    Returns CSR (offsets int64[nq+1], cluster ids int32, scores float64).
    n_topics > 0: the topic-mixture variant -- a user follows `topics_per_user` Zipf-drawn topics."""
    rng = np.random.default_rng(seed)
    perm = cluster_permutation(n_clusters)
    offs = np.zeros(n_queries + 1, np.int64)
    cids, scs = [], []
    for q in range(n_queries):
        got: list = []
        seen = set()
        mine = zipf_ranks(rng, topics_per_user, n_topics) - 1 if n_topics > 0 else None
        while len(got) < min(clusters_per_user, n_clusters):
            if mine is not None:
                ranks = topic_cluster_ranks(rng, mine[rng.integers(0, len(mine), clusters_per_user)], n_clusters, n_topics,
                                            topic_size, affinity)
            else:
                ranks = zipf_ranks(rng, clusters_per_user, n_clusters)
            for c in perm[ranks - 1]:
                if int(c) not in seen and len(got) < clusters_per_user:
                    seen.add(int(c))
                    got.append(int(c))
        cids.append(np.array(got, np.int32))
        scs.append(np.exp(rng.normal(0.0, 1.0, size=len(got))))
        offs[q + 1] = offs[q] + len(got)
    return offs, np.concatenate(cids) if cids else np.empty(0, np.int32), np.concatenate(scs) if scs else np.empty(0)
