"""representation-scorer pair kernel (include/representation_scorer.h) against the oracle and the KATs.
BASELINE configs[0]: 10,000 (user, tweet) SimClusters embedding pairs, PairEmbeddingCosineSimilarity."""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "sann_kat.json")))


def csr(embs):
    offs = np.zeros(len(embs) + 1, np.int64)
    for i, (c, _) in enumerate(embs):
        offs[i + 1] = offs[i] + len(c)
    return offs, np.concatenate([c for c, _ in embs]) if embs else np.empty(0, np.int32), \
        np.concatenate([s for _, s in embs]) if embs else np.empty(0)


@pytest.mark.parametrize("case", KAT["pairs"], ids=[c["name"] for c in KAT["pairs"]])
def test_pair_kat(pkg, case):
    rs = pkg.representation_scorer
    a = rs.simclusters_embedding(case["a"]); b = rs.simclusters_embedding(case["b"])
    got = rs.pair_scores(rs.ScoringAlgorithm(case["alg"]), *csr([a]), *csr([b]))[0]
    exp = float.fromhex(case["expect"])
    assert abs(got - exp) <= case["ulp"] * math.ulp(exp) if case["ulp"] else got == exp


def test_10k_pairs_all_algorithms_match_oracle(pkg, oracle):
    rs = pkg.representation_scorer
    n = 10_000
    rng = np.random.default_rng(20260105)
    perm = pkg.corpus.cluster_permutation(pkg.corpus.N_CLUSTERS)
    users, tweets = [], []
    for _ in range(n):
        uc = np.unique(perm[pkg.corpus.zipf_ranks(rng, 50, pkg.corpus.N_CLUSTERS) - 1])
        nt = min(50, int(rng.geometric(1 / 25)))
        tc = np.unique(perm[pkg.corpus.zipf_ranks(rng, nt, pkg.corpus.N_CLUSTERS) - 1])
        users.append((uc.astype(np.int32), np.exp(rng.normal(0, 1, len(uc)))))
        tweets.append((tc.astype(np.int32), np.maximum(np.exp(rng.normal(-2, 1, len(tc))), 0.001)))
    A, B = csr(users), csr(tweets)
    overlap = 0
    for alg in range(1, 8):
        got = rs.pair_scores(rs.ScoringAlgorithm(alg), *A, *B)
        step = 1 if alg == 2 else 7  # every pair for the headline algorithm, a sample for the others
        for i in range(0, n, step):
            exp = oracle.pair_score(alg, users[i][0], users[i][1], tweets[i][0], tweets[i][1])
            if alg == 7:
                assert abs(got[i] - exp) <= 4 * math.ulp(exp) + 1e-300, (alg, i, got[i], exp)
            else:
                assert got[i] == exp or (math.isnan(got[i]) and math.isnan(exp)), (alg, i, got[i], exp)
            overlap += alg == 2 and exp > 0
    assert overlap > 100, "the synthetic pairs should overlap often enough to exercise the merge"


def test_rejects_bad_input(pkg):
    rs = pkg.representation_scorer
    offs = np.array([0, 2], np.int64)
    good = (offs, np.array([1, 2], np.int32), np.array([1.0, 2.0]))
    with pytest.raises(RuntimeError):
        rs.pair_scores(rs.ScoringAlgorithm.PairEmbeddingCosineSimilarity, offs, np.array([2, 1], np.int32), np.array([1.0, 2.0]), *good)
    with pytest.raises(RuntimeError):
        rs.pair_scores(rs.ScoringAlgorithm.PairEmbeddingCosineSimilarity, offs, np.array([1, 2], np.int32), np.array([1.0, 0.0]), *good)
    lib = pkg.load_library()
    assert lib.rsx_pair_scores(0, 1000, 1, None, None, None, None, None, None, 0, None) == 1  # TagSpaceCosineSimilarity: not a pair metric here
