"""sann_topk_merge (the streaming side of the cluster -> top tweets store) against the oracle's literal
restatement of TopKTweetsWithScoresMonoid.plus (summingbird/common/Monoids.scala:131-158,378-450): same keys, bit-equal
decayed values and scaled times, results ordered by (value desc, tweet id asc)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HALF_LIFE_MS = 8 * 3600 * 1000
NOW_MS = 1_700_000_000_000


def _scaled(ms):
    return ms * math.log(2.0) / HALF_LIFE_MS


def _side(rng, n, id_pool, t_lo_ms, t_hi_ms, lo_exp=-4.0, hi_exp=1.0):
    ids = rng.choice(id_pool, size=n, replace=False)
    vals = np.exp(rng.uniform(lo_exp, hi_exp, n) * math.log(10.0))
    times = _scaled(rng.integers(t_lo_ms, t_hi_ms, n).astype(np.float64))
    return {int(i): (float(v), float(t)) for i, v, t in zip(ids, vals, times)}


def _csr(sides):
    off, ids, vals, ts = [0], [], [], []
    for s in sides:
        for k, (v, t) in (s or {}).items():
            ids.append(k); vals.append(v); ts.append(t)
        off.append(len(ids))
    return (np.array(off, np.int64), np.array(ids, np.int64), np.array(vals, np.float64), np.array(ts, np.float64))


def _check(pkg, oracle, A, B, top_k, threshold, oldest):
    oo, oi, ov, ot = pkg.simclusters_ann.topk_merge(_csr(A), _csr(B), top_k=top_k, threshold=threshold, oldest_tweet_id=oldest)
    assert oo[0] == 0 and len(oo) == len(A) + 1
    for c, (a, b) in enumerate(zip(A, B)):
        want = oracle.topk_merge(a, b, top_k, threshold, oldest) or {}
        lo, hi = int(oo[c]), int(oo[c + 1])
        got = {int(i): (float(v), float(t)) for i, v, t in zip(oi[lo:hi], ov[lo:hi], ot[lo:hi])}
        assert got.keys() == want.keys(), (c, len(got), len(want))
        for k in got:
            assert np.float64(got[k][0]).view(np.int64) == np.float64(want[k][0]).view(np.int64), (c, k, got[k], want[k])
            assert got[k][1] == want[k][1], (c, k, got[k], want[k])
        order = sorted(got.items(), key=lambda kv: (-kv[1][0], kv[0]))
        assert [k for k, _ in order] == [int(i) for i in oi[lo:hi]], f"list {c}: not in (value desc, id asc) order"


def test_merge_matches_the_monoid(pkg, oracle):
    rng = np.random.default_rng(11)
    pool = (np.arange(1, 6000, dtype=np.int64) << 22) + ((NOW_MS - 1288834974657) << 22) - (3 * 86_400_000 << 22)
    A, B = [], []
    for c in range(40):
        na, nb = int(rng.integers(0, 900)), int(rng.integers(0, 900))
        if c % 9 == 0: na = 0
        if c % 13 == 0: nb = 0
        A.append(_side(rng, na, pool, NOW_MS - 3 * 86_400_000, NOW_MS - 3_600_000))
        B.append(_side(rng, nb, pool, NOW_MS - 3_600_000, NOW_MS))
    A[5], B[5] = None, None
    oldest = int(np.sort(pool)[600])
    _check(pkg, oracle, A, B, 400, 0.001, oldest)       # cut fires for the long lists (> 480 survivors)
    _check(pkg, oracle, A, B, 1600, 0.001, -(1 << 63))   # production top_k: no cut, no age filter
    _check(pkg, oracle, A, B, 10, -1.0, oldest)          # negative threshold: exact zeros survive with time -inf


def test_merge_ties_and_identical_times(pkg, oracle):
    """Quantised values (ties at the cut), equal scaled times on both sides, the same tweet on both sides with equal and
    with different values."""
    rng = np.random.default_rng(5)
    t0 = _scaled(NOW_MS)
    A, B = [], []
    for c in range(12):
        ids = rng.permutation(np.arange(1, 400, dtype=np.int64))
        a = {int(i): (float(rng.integers(1, 6)) / 4.0, t0 if c % 2 else _scaled(NOW_MS - int(rng.integers(0, 4)) * HALF_LIFE_MS)) for i in ids[:250]}
        b = {int(i): (float(rng.integers(1, 6)) / 4.0, t0) for i in ids[150:380]}
        A.append(a); B.append(b)
    _check(pkg, oracle, A, B, 100, 0.2, 0)
    _check(pkg, oracle, A, B, 100, 0.0, 50)


def test_merge_full_size_lists(pkg, oracle):
    """Both sides at the store's maximum (1.2 x 1600) -- 3840 entries in one workgroup."""
    rng = np.random.default_rng(8)
    pool = np.arange(1, 5000, dtype=np.int64)
    A = [_side(rng, 1920, pool, NOW_MS - 86_400_000, NOW_MS) for _ in range(3)]
    B = [_side(rng, 1920, pool, NOW_MS - 86_400_000, NOW_MS) for _ in range(3)]
    _check(pkg, oracle, A, B, 1600, 0.001, 0)


def test_merge_argument_errors(pkg):
    sa = pkg.simclusters_ann
    one = (np.array([0, 2], np.int64), np.array([7, 7], np.int64), np.ones(2), np.ones(2))
    other = (np.array([0, 1], np.int64), np.array([3], np.int64), np.ones(1), np.ones(1))
    with pytest.raises(sa.SannError):
        sa.topk_merge(one, other)  # a tweet twice on one side
    big = (np.array([0, 4000], np.int64), np.arange(4000, dtype=np.int64), np.ones(4000), np.ones(4000))
    big2 = (np.array([0, 200], np.int64), np.arange(5000, 5200, dtype=np.int64), np.ones(200), np.ones(200))
    with pytest.raises(sa.SannError):
        sa.topk_merge(big, big2)  # 4200 entries in one list
    with pytest.raises(ValueError):
        sa.topk_merge(one, (np.array([0, 1, 1], np.int64), np.array([3], np.int64), np.ones(1), np.ones(1)))
