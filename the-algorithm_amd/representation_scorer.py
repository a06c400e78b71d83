"""ctypes binding of include/representation_scorer.h and a mirror of the reference's pair scorer.

Reference (paths relative to /root/reference/):
  src/scala/com/twitter/simclusters_v2/score/SimClustersEmbeddingPairScoreStore.scala:39-199
  src/thrift/com/twitter/simclusters_v2/score.thrift:14-22   ScoringAlgorithm ids
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .simclusters_ann import load_library


class ScoringAlgorithm(enum.IntEnum):
    """score.thrift:14-22 (the pairwise block)."""

    PairEmbeddingDotProduct = 1
    PairEmbeddingCosineSimilarity = 2
    PairEmbeddingJaccardSimilarity = 3
    PairEmbeddingEuclideanDistance = 4
    PairEmbeddingManhattanDistance = 5
    PairEmbeddingLogCosineSimilarity = 6
    PairEmbeddingExpScaledCosineSimilarity = 7


PROTOS = {
    "rsx_last_error": (C.c_char_p, []),
    "rsx_pair_scores": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 6 + [C.c_int32, C.c_void_p]),
    "rsx_pair_scores_device": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 7),
    "rsx_store_build": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rsx_store_destroy": (C.c_int, [C.c_void_p]),
    "rsx_store_pair_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rsx_store_list_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rsx_heavy_rank_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_double, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rsx_store_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "rsx_store_group_features": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}


def _lib():
    lib = load_library()
    if not getattr(lib, "_rsx_ready", False):
        for name, (res, args) in PROTOS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib._rsx_ready = True
    return lib


def simclusters_embedding(pairs: Sequence[Tuple[int, float]]):
    """The SimClustersEmbedding constructor's view (SimClustersEmbedding.scala:490-509): drop score <= 0,
    return (sortedClusterIds int32, sortedScores float64) ascending by cluster id."""
    kept = sorted((int(c), float(s)) for c, s in pairs if s > 0.0)
    return np.array([c for c, _ in kept], np.int32), np.array([s for _, s in kept], np.float64)


def pair_scores(algorithm: ScoringAlgorithm, a_offsets, a_ids, a_scores, b_offsets, b_ids, b_scores, *, device: int = 0,
                validate: bool = True) -> np.ndarray:
    """Scores of n pairs; side A and side B as CSR over already-sorted embeddings."""
    lib = _lib()
    ao = np.ascontiguousarray(a_offsets, np.int64); bo = np.ascontiguousarray(b_offsets, np.int64)
    ai = np.ascontiguousarray(a_ids, np.int32); bi = np.ascontiguousarray(b_ids, np.int32)
    asx = np.ascontiguousarray(a_scores, np.float64); bsx = np.ascontiguousarray(b_scores, np.float64)
    n = len(ao) - 1
    out = np.zeros(n, np.float64)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    rc = lib.rsx_pair_scores(device, int(algorithm), n, p(ao), p(ai), p(asx), p(bo), p(bi), p(bsx), 1 if validate else 0, p(out))
    if rc != 0:
        raise RuntimeError(f"representation_scorer error {rc}: {lib.rsx_last_error().decode()}")
    return out


def _check(lib, rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"representation_scorer error {rc}: {lib.rsx_last_error().decode()}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class EmbeddingStore:
    """Device-resident `ReadableStore[SimClustersEmbeddingId, SimClustersEmbedding]` of one
    (embeddingType, modelVersion): `embeddings` maps an internal id to its (clusterId, score) pairs."""

    def __init__(self, embeddings: Dict[int, Sequence[Tuple[int, float]]], *, device: int = 0):
        lib = _lib()
        ids = np.array(sorted(embeddings), np.int64)
        offs, cl, sc = [0], [], []
        for i in ids.tolist():
            c, v = simclusters_embedding(embeddings[i])
            cl.append(c); sc.append(v); offs.append(offs[-1] + len(c))
        self.ids = ids
        o = np.array(offs, np.int64)
        c = np.concatenate(cl) if cl else np.zeros(0, np.int32)
        v = np.concatenate(sc) if sc else np.zeros(0, np.float64)
        self._h = C.c_void_p()
        _check(lib, lib.rsx_store_build(device, len(ids), _p(ids), _p(o), _p(c), _p(v), C.byref(self._h)))

    def close(self):
        if self._h:
            _lib().rsx_store_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def multi_get(algorithm: ScoringAlgorithm, store_a: EmbeddingStore, store_b: EmbeddingStore, pairs: Sequence[Tuple[int, int]]) -> List[Optional[float]]:
    """PairScoreStore.multiGet (score/ScoreStore.scala:41-69): None when either embedding is missing."""
    lib = _lib()
    n = len(pairs)
    a = np.array([x for x, _ in pairs], np.int64); b = np.array([y for _, y in pairs], np.int64)
    out = np.zeros(n, np.float64); pres = np.zeros(n, np.uint8)
    _check(lib, lib.rsx_store_pair_scores(store_a._h, store_b._h, int(algorithm), n, _p(a), _p(b), _p(out), _p(pres)))
    return [float(out[i]) if pres[i] else None for i in range(n)]


def list_scores(algorithm: ScoringAlgorithm, targets: EmbeddingStore, candidates: EmbeddingStore, target_id: int,
                candidate_ids: Sequence[int]) -> List[Optional[float]]:
    """ListScoreColumn.fetch (columns/ListScoreColumn.scala:53-115): ordered as requested, None where a
    score cannot be produced."""
    lib = _lib()
    n = len(candidate_ids)
    c = np.ascontiguousarray(candidate_ids, np.int64)
    out = np.zeros(n, np.float64); pres = np.zeros(n, np.uint8)
    _check(lib, lib.rsx_store_list_scores(targets._h, candidates._h, int(algorithm), int(target_id), n, _p(c), _p(out), _p(pres)))
    return [float(out[i]) if pres[i] else None for i in range(n)]


@dataclass
class UserSignal:
    """twistlyfeatures/UserSignalServiceRecentEngagementsClient: (targetId, timestamp ms)."""
    targetId: int
    timestamp: int


_TWEET_SIGNALS = ("favs", "retweets", "shares", "replies", "originalTweets", "videoPlaybacks")


@dataclass
class Engagements:
    """twistlyfeatures/Engagements.scala:7-56; `now_ms` replaces the constructor's Time.now."""
    now_ms: int
    favs7d: List[UserSignal] = field(default_factory=list)
    retweets7d: List[UserSignal] = field(default_factory=list)
    follows30d: List[UserSignal] = field(default_factory=list)
    shares7d: List[UserSignal] = field(default_factory=list)
    replies7d: List[UserSignal] = field(default_factory=list)
    originalTweets7d: List[UserSignal] = field(default_factory=list)
    videoPlaybacks7d: List[UserSignal] = field(default_factory=list)
    block30d: List[UserSignal] = field(default_factory=list)
    mute30d: List[UserSignal] = field(default_factory=list)
    report30d: List[UserSignal] = field(default_factory=list)
    dontlike30d: List[UserSignal] = field(default_factory=list)
    seeFewer30d: List[UserSignal] = field(default_factory=list)

    # UserSignalServiceRecentEngagementsClient.scala:38-52: field <- (SignalType, days of validity)
    USS_FIELDS = (("favs7d", "TweetFavorite", 7), ("retweets7d", "Retweet", 7), ("follows30d", "AccountFollowWithDelay", 30),
                  ("shares7d", "TweetShareV1", 7), ("replies7d", "Reply", 7), ("originalTweets7d", "OriginalTweet", 7),
                  ("videoPlaybacks7d", "VideoView90dPlayback50V1", 7), ("block30d", "AccountBlock", 30),
                  ("mute30d", "AccountMute", 30), ("report30d", "TweetReport", 30), ("dontlike30d", "TweetDontLike", 30),
                  ("seeFewer30d", "TweetSeeFewer", 30))
    ENGAGEMENTS_TO_SCORE = 10  # :128

    @classmethod
    def from_signal_response(cls, signal_response, now_ms: int) -> "Engagements":
        """UserSignalServiceRecentEngagementsClient.get / getUserSignals (:30-71): per signal type keep the signals newer
        than the field's window whose target is a Long id, take the first 10.  signal_response: {SignalType name:
        [(targetId or None, timestamp ms)]} as the signal service returned them."""
        kw = {}
        for fld, signal_type, days in cls.USS_FIELDS:
            earliest = now_ms - days * 86_400_000
            kept = [UserSignal(t, ts) for t, ts in signal_response.get(signal_type, []) if ts > earliest and t is not None]
            kw[fld] = kept[:cls.ENGAGEMENTS_TO_SCORE]
        return cls(now_ms=now_ms, **kw)

    def _since(self, xs, days):
        cut = self.now_ms - days * 86_400_000
        return [s for s in xs if s.timestamp > cut]

    @property
    def tweetIds(self) -> List[int]:  # Engagements.scala:28-32
        xs = (self.favs7d + self.retweets7d + self.shares7d + self.replies7d + self.originalTweets7d
              + self.videoPlaybacks7d + self.report30d + self.dontlike30d + self.seeFewer30d)
        return [s.targetId for s in xs]

    @property
    def authorIds(self) -> List[int]:  # Engagements.scala:33
        return [s.targetId for s in self.follows30d + self.block30d + self.mute30d]

    def groups(self) -> List[Tuple[str, int, List[int]]]:
        """(feature prefix, map: 0 = tweetScores / 1 = authorScores, signal ids in order), in the order of
        the SimClustersRecentEngagementSimilarities constructor (Scorer.scala:306-369).  block* / mute* are
        looked up in tweetScores, exactly as Scorer.scala:232-260 does."""
        g: List[Tuple[str, int, List[UserSignal]]] = []
        for name, attr in (("fav", "favs7d"), ("retweet", "retweets7d")):
            g += [(name + "1d", 0, self._since(getattr(self, attr), 1)), (name + "7d", 0, getattr(self, attr))]
        g += [("follow7d", 1, self._since(self.follows30d, 7)), ("follow30d", 1, self.follows30d)]
        for name, attr in (("share", "shares7d"), ("reply", "replies7d"), ("originalTweet", "originalTweets7d"),
                           ("videoPlayback", "videoPlaybacks7d")):
            g += [(name + "1d", 0, self._since(getattr(self, attr), 1)), (name + "7d", 0, getattr(self, attr))]
        for name, attr in (("block", "block30d"), ("mute", "mute30d"), ("report", "report30d"),
                           ("dontlike", "dontlike30d"), ("seeFewer", "seeFewer30d")):
            x30 = getattr(self, attr)
            x7 = self._since(x30, 7)
            g += [(name + "1d", 0, self._since(x7, 1)), (name + "7d", 0, x7), (name + "30d", 0, x30)]
        return [(n, m, [s.targetId for s in xs]) for n, m, xs in g]


class Scorer:
    """twistlyfeatures/Scorer.scala: per (user, tweet) similarity features between the tweet and the user's
    recent engagements.  `tweets` / `authors` are the embedding stores behind getTweetScoreId /
    getAuthorScoreId (Scorer.scala:431-470); the algorithm is PairEmbeddingCosineSimilarity as there."""

    def __init__(self, tweets: EmbeddingStore, authors: EmbeddingStore,
                 algorithm: ScoringAlgorithm = ScoringAlgorithm.PairEmbeddingCosineSimilarity):
        self.tweets, self.authors, self.algorithm = tweets, authors, algorithm

    def get(self, engagements: Engagements, tweet_ids: Sequence[int]) -> List[Dict[str, Optional[float]]]:
        """Scorer.get (:125-149): one feature dict per requested tweet, same number / order;
        `<prefix>Last10Max` / `<prefix>Last10Avg`, None where the reference returns None."""
        lib = _lib()
        groups = engagements.groups()
        cand = np.ascontiguousarray(tweet_ids, np.int64)
        map_ids = [np.array(engagements.tweetIds, np.int64), np.array(engagements.authorIds, np.int64)]
        map_off = np.array([0, len(map_ids[0]), len(map_ids[0]) + len(map_ids[1])], np.int64)
        all_ids = np.concatenate(map_ids) if map_off[-1] else np.zeros(0, np.int64)
        gmap = np.array([m for _, m, _ in groups], np.int32)
        goff = np.cumsum([0] + [len(x) for _, _, x in groups]).astype(np.int64)
        members = np.array([i for _, _, x in groups for i in x], np.int64)
        stores = (C.c_void_p * 2)(self.tweets._h, self.authors._h)
        n, ng = len(cand), len(groups)
        avg = np.zeros((n, ng)); mx = np.zeros((n, ng)); cnt = np.zeros((n, ng), np.int32)
        if n:
            _check(lib, lib.rsx_store_group_features(self.tweets._h, int(self.algorithm), n, _p(cand), 2, stores, _p(map_off),
                                                     _p(all_ids), ng, _p(gmap), _p(goff), _p(members), _p(avg), _p(mx), _p(cnt)))
        out = []
        for i in range(n):
            d: Dict[str, Optional[float]] = {}
            for j, (name, _, _) in enumerate(groups):
                d[name + "Last10Max"] = float(mx[i, j]) if cnt[i, j] else None
                d[name + "Last10Avg"] = float(avg[i, j]) if cnt[i, j] else None
            out.append(d)
        return out
