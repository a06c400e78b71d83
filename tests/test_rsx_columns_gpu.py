"""R1 (hydrating multiGet), R4 (ListScoreColumn) and R5 (Scorer aggregates) through the C ABI against
the literal restatements in oracle/oracle.py.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _emb(rng, n_clusters=300, k=20):
    n = int(rng.integers(1, k))
    ids = rng.choice(n_clusters, n, replace=False)
    return [(int(c), float(s)) for c, s in zip(ids, rng.random(n) + 0.01)]


@pytest.fixture(scope="module")
def world(pkg):
    rs = pkg.representation_scorer
    rng = np.random.default_rng(11)
    tweets = {int(i): _emb(rng) for i in rng.choice(10_000, 400, replace=False)}
    authors = {int(i): _emb(rng) for i in rng.choice(10_000, 60, replace=False)}
    ts, au = rs.EmbeddingStore(tweets), rs.EmbeddingStore(authors)
    yield rs, rng, tweets, authors, ts, au
    ts.close(); au.close()


def _e(rs, d, i):
    return rs.simclusters_embedding(d[i]) if i in d else None


@pytest.mark.parametrize("alg", [1, 2, 3, 4, 5, 6, 7])
def test_multi_get_and_list_scores(world, oracle, alg):
    rs, rng, tweets, authors, ts, au = world
    t_ids = list(tweets); a_ids = list(authors)
    pairs = [(int(rng.choice(a_ids)) if rng.random() < 0.9 else 123456789, int(rng.choice(t_ids)) if rng.random() < 0.9 else -5)
             for _ in range(300)]
    got = rs.multi_get(rs.ScoringAlgorithm(alg), au, ts, pairs)
    for (a, b), g in zip(pairs, got):
        ea, eb = _e(rs, authors, a), _e(rs, tweets, b)
        want = None if ea is None or eb is None else oracle.pair_score(alg, ea[0], ea[1], eb[0], eb[1])
        assert (g is None) == (want is None)
        if g is not None:
            assert np.float64(g).view(np.int64) == np.float64(want).view(np.int64) or (alg == 7 and abs(g - want) <= 4e-16 * abs(want))
    target = a_ids[3]
    cands = [int(rng.choice(t_ids)) for _ in range(50)] + [424242, t_ids[0], t_ids[0]]
    got = rs.list_scores(rs.ScoringAlgorithm(alg), au, ts, target, cands)
    want = oracle.rsx_list_scores(alg, _e(rs, authors, target), [_e(rs, tweets, c) for c in cands])
    assert len(got) == len(cands) and got[50] is None
    for g, w in zip(got, want):
        assert (g is None) == (w is None)
        if g is not None:
            assert g == w or (alg == 7 and abs(g - w) <= 4e-16 * abs(w))
    # a target without an embedding: every score is None, order and length kept
    assert rs.list_scores(rs.ScoringAlgorithm(alg), au, ts, 999_999_999, cands) == [None] * len(cands)


def test_scorer_features_match_the_reference_fold(world, oracle):
    rs, rng, tweets, authors, ts, au = world
    now = 1_700_000_000_000
    t_ids = list(tweets); a_ids = list(authors)

    def sig(pool, n, span_days, missing=0.15):
        out = []
        for _ in range(n):
            i = int(rng.choice(pool)) if rng.random() > missing else int(rng.integers(20_000, 30_000))
            out.append(rs.UserSignal(i, now - int(rng.random() * span_days * 86_400_000)))
        return out

    shared = sig(t_ids, 3, 7, 0.0)  # the same tweet faved AND retweeted: counted twice in each fold
    eng = rs.Engagements(
        now_ms=now, favs7d=sig(t_ids, 8, 7) + shared, retweets7d=shared + sig(t_ids, 4, 7), follows30d=sig(a_ids, 7, 30),
        shares7d=sig(t_ids, 2, 7), replies7d=[], originalTweets7d=sig(t_ids, 10, 7), videoPlaybacks7d=sig(t_ids, 5, 7),
        block30d=sig(a_ids + t_ids[:5], 6, 30), mute30d=sig(a_ids, 3, 30), report30d=sig(t_ids, 4, 30),
        dontlike30d=sig(t_ids, 6, 30), seeFewer30d=sig(t_ids, 1, 30))
    cand_ids = [int(rng.choice(t_ids)) for _ in range(40)] + [31337]
    got = rs.Scorer(ts, au).get(eng, cand_ids)
    assert len(got) == len(cand_ids) and len(got[0]) == 58
    groups = eng.groups()
    emb_t = {i: rs.simclusters_embedding(v) for i, v in tweets.items()}
    emb_a = {i: rs.simclusters_embedding(v) for i, v in authors.items()}
    some_avg_differs_from_max = False
    for c, feats in zip(cand_ids, got):
        want = oracle.rsx_engagement_features(2, emb_t.get(c), [eng.tweetIds, eng.authorIds], [emb_t, emb_a],
                                              [(m, ids) for _, m, ids in groups])
        for (name, _, _), (w_avg, w_max) in zip(groups, want):
            assert feats[name + "Last10Avg"] == w_avg, (c, name)
            assert feats[name + "Last10Max"] == w_max, (c, name)
            some_avg_differs_from_max |= w_avg is not None and w_avg != w_max
    assert some_avg_differs_from_max
    assert all(v is None for v in got[-1].values())          # candidate without an embedding
    assert got[0]["reply7dLast10Avg"] is None                # empty signal list
    by_name = {g[0]: g[2] for g in groups}
    assert set(by_name["fav1d"]) <= set(by_name["fav7d"]) and set(by_name["block1d"]) <= set(by_name["block7d"]) <= set(by_name["block30d"])


def test_scorer_features_from_raw_signals(world, oracle):
    """R5 end to end from RAW user signals: the product (Engagements.from_signal_response -> windows -> device folds)
    against oracle.rsx_scorer_features, which restates UserSignalServiceRecentEngagementsClient.getUserSignals (window
    filter, Long ids only, take 10), Engagements.scala:21-56 and Scorer.scala:157-369 on its own -- including signals
    exactly on a window edge (`>` is strict), more than 10 signals per type, and the block / mute map quirk."""
    rs, rng, tweets, authors, ts, au = world
    now = 1_700_000_000_000
    day = 86_400_000
    t_ids = list(tweets); a_ids = list(authors)

    def raw(pool, n, span_days, edges=()):
        out = []
        for _ in range(n):
            i = int(rng.choice(pool)) if rng.random() > 0.1 else int(rng.integers(20_000, 30_000))
            out.append((i if rng.random() > 0.05 else None, now - int(rng.random() * span_days * day)))
        out += [(int(rng.choice(pool)), now - d * day) for d in edges]          # exactly on an edge: excluded
        out += [(int(rng.choice(pool)), now - d * day + 1) for d in edges]      # one ms inside: included
        order = rng.permutation(len(out))
        return [out[i] for i in order]

    emb_t = {i: rs.simclusters_embedding(v) for i, v in tweets.items()}
    emb_a = {i: rs.simclusters_embedding(v) for i, v in authors.items()}
    for trial in range(4):
        response = {"TweetFavorite": raw(t_ids, 18, 9, (1, 7)), "Retweet": raw(t_ids, 6, 9, (1,)),
                    "AccountFollowWithDelay": raw(a_ids, 14, 40, (7, 30)), "TweetShareV1": raw(t_ids, 3, 8),
                    "Reply": [] if trial % 2 else raw(t_ids, 2, 3), "OriginalTweet": raw(t_ids, 12, 8, (1,)),
                    "VideoView90dPlayback50V1": raw(t_ids, 5, 8), "AccountBlock": raw(a_ids + t_ids[:6], 9, 35, (1, 7, 30)),
                    "AccountMute": raw(a_ids, 4, 35), "TweetReport": raw(t_ids, 5, 35, (7,)),
                    "TweetDontLike": raw(t_ids, 11, 35, (1, 7, 30)), "TweetSeeFewer": raw(t_ids, 2, 35)}
        eng = rs.Engagements.from_signal_response(response, now)
        cand_ids = [int(rng.choice(t_ids)) for _ in range(12)] + [31337]
        got = rs.Scorer(ts, au).get(eng, cand_ids)
        for c, feats in zip(cand_ids, got):
            want = oracle.rsx_scorer_features(response, now, emb_t.get(c), emb_t, emb_a, algorithm=2)
            assert set(feats) == set(want) and len(want) == 58
            for name, w in want.items():
                assert feats[name] == w, (trial, c, name, feats[name], w)


def test_store_rejects_malformed_input(pkg):
    rs = pkg.representation_scorer
    lib = rs._lib()
    import ctypes as C
    h = C.c_void_p()
    ids = np.array([5, 5], np.int64); off = np.array([0, 1, 2], np.int64)
    cl = np.array([1, 2], np.int32); sc = np.array([0.5, 0.5])
    assert lib.rsx_store_build(0, 2, ids.ctypes.data, off.ctypes.data, cl.ctypes.data, sc.ctypes.data, C.byref(h)) == 1
    assert b"ascending" in lib.rsx_last_error()
