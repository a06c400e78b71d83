"""Exact mode of the dense search (dann_index_build_exact): survivors of the fp16 GEMM are scored again from the fp32
rows and the result is proven complete, so it must equal an exact scan of the ORIGINAL (unrounded) vectors -- what
BruteForceIndex does (ann/src/main/scala/com/twitter/ann/brute_force/BruteForceIndex.scala:66-91).  The check is a
float64 numpy scan of the original inputs: distances within 2e-6, ids equal wherever the float64 distances are further
apart than that.  PARITY UNPINNED against the JVM's fp32 summation order (EmbeddingMath is not in the tree)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scan(metric_name, x, q, k):
    x64, q64 = x.astype(np.float64), q.astype(np.float64)
    if metric_name == "Cosine":
        x64 = x64 / np.linalg.norm(x64, axis=1, keepdims=True)
        q64 = q64 / np.linalg.norm(q64, axis=1, keepdims=True)
    if metric_name == "L2":
        d = np.sqrt(np.maximum(0.0, (q64 ** 2).sum(1)[:, None] + (x64 ** 2).sum(1)[None, :] - 2.0 * q64 @ x64.T))
    else:
        d = 1.0 - q64 @ x64.T
    order = np.argsort(d, axis=1, kind="stable")[:, :k]
    return order, np.take_along_axis(d, order, 1)


def _check(ids, dist, cnt, want_ids, want_d, tol):
    for qi in range(len(ids)):
        m = cnt[qi]
        assert m == want_ids.shape[1]
        assert np.allclose(dist[qi, :m], want_d[qi], atol=tol, rtol=0), qi
        gaps_ok = np.ones(m, bool)
        gaps_ok[1:] &= np.diff(want_d[qi]) > 2 * tol
        gaps_ok[:-1] &= np.diff(want_d[qi]) > 2 * tol
        assert np.array_equal(ids[qi, :m][gaps_ok], want_ids[qi][gaps_ok]), qi


@pytest.mark.parametrize("metric", ["Cosine", "InnerProduct", "L2"])
@pytest.mark.parametrize("d", [64, 256])
def test_exact_mode_equals_a_scan_of_the_original_vectors(pkg, metric, d):
    m = getattr(pkg.dense_ann.DistanceMetric, metric)
    rng = np.random.default_rng(d + len(metric))
    x = rng.standard_normal((30_000, d)).astype(np.float32)
    q = rng.standard_normal((48, d)).astype(np.float32)
    if metric != "Cosine":
        x *= 0.25
    ix = pkg.dense_ann.BruteForceIndex.build(m, x, exact=True)
    try:
        ids, dist, cnt = ix.search(q, 50)
    finally:
        ix.close()
    want_ids, want_d = _scan(metric, x, q, 50)
    scale = 1.0 if metric == "Cosine" else float(np.abs(want_d).max())
    _check(ids, dist, cnt, want_ids, want_d, 3e-6 * max(scale, 1.0))


def test_exact_mode_separates_what_fp16_cannot(pkg):
    """Near-duplicates that differ below fp16's resolution: the fp16 search cannot order them, the exact mode must."""
    m = pkg.dense_ann.DistanceMetric.Cosine
    rng = np.random.default_rng(3)
    d = 128
    base = rng.standard_normal(d).astype(np.float32)
    base /= np.linalg.norm(base)
    other = rng.standard_normal((20_000, d)).astype(np.float32)
    # 40 copies of `base`, each nudged by a different, tiny amount towards an orthogonal direction
    orth = rng.standard_normal(d).astype(np.float32)
    orth -= orth.dot(base) * base
    orth /= np.linalg.norm(orth)
    eps = (np.arange(1, 41, dtype=np.float32) * 2e-4)[rng.permutation(40)]
    near = base[None, :] + eps[:, None] * orth[None, :]
    x = np.concatenate([other, near.astype(np.float32)])
    q = base[None, :].copy()
    ix = pkg.dense_ann.BruteForceIndex.build(m, x, exact=True)
    try:
        ids, dist, cnt = ix.search(q, 40)
    finally:
        ix.close()
    want_ids, want_d = _scan("Cosine", x, q, 40)
    # cosine distances of the 40: eps^2 / 2 ~ 2e-8 .. 3.2e-5 -- far below what fp16 operands resolve (1e-3)
    assert set(ids[0, :40].tolist()) == set(range(20_000, 20_040))
    big = np.diff(want_d[0]) > 3e-7  # (fp32 arithmetic itself resolves ~1e-7 here)
    got_rank = {int(v): i for i, v in enumerate(ids[0, :40])}
    for i in np.nonzero(big)[0]:
        assert got_rank[int(want_ids[0, i])] < got_rank[int(want_ids[0, i + 1])], i


def test_proof_rearms_a_query_and_still_gets_it_right(pkg, monkeypatch):
    """An inflated rounding bound (test knob) makes the proof fail: the pass repeats with lower thresholds for those
    queries and the answer is still the exact one."""
    m = pkg.dense_ann.DistanceMetric.InnerProduct
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((20_000, 64)) * 0.3).astype(np.float32)
    q = rng.standard_normal((16, 64)).astype(np.float32)
    want_ids, want_d = _scan("InnerProduct", x, q, 20)
    ix = pkg.dense_ann.BruteForceIndex.build(m, x, exact=True)
    try:
        a = ix.search(q, 20)
        assert ix.last_rounds() == 1
        monkeypatch.setenv("DANN_DEBUG_DELTA_SCALE", "20")
        b = ix.search(q, 20)
        assert ix.last_rounds() >= 2, "the inflated bound should have re-armed some query"
    finally:
        ix.close()
    for res in (a, b):
        _check(res[0], res[1], res[2], want_ids, want_d, 3e-6 * float(np.abs(want_d).max()))
