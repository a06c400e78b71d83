"""Randomised parity run: the HIP path against the oracle over random corpora, partitionings, configurations, variants,
source tweets and windows -- far more shapes than tests/ enumerates.  Stops at the first difference and prints the case's
seed.  `python tools/fuzz_parity.py [--seconds 300] [--seed 1]`   (GPU; the oracle is the checker, as in tests/)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg  # noqa: E402
import oracle  # noqa: E402


def one_case(pkg, seed, wild=False):
    """wild: lists the store's contract (score > 0, sorted descending: TopKTweetsForClusterReadableStore.scala:211-229)
    rules out but the operator takes as they come (ApproximateCosineSimilarity.scala:87 reads position i of whatever it is
    given): shuffled order, negative and zero scores.  sann_index_build keeps positions as given, so must the result."""
    rng = np.random.default_rng(seed)
    n_tweets = int(rng.choice([3000, 20_000, 80_000, 250_000]))
    n_clusters = int(rng.choice([40, 300, 1500, 6000]))
    cap = int(rng.choice([50, 400, 2000]))
    topics = int(rng.choice([0, 0, 25]))
    co = pkg.corpus.make_corpus(n_tweets, n_clusters, seed=seed, index_cap=cap, mean_clusters=float(rng.choice([3, 10, 25])),
                                n_topics=topics)
    nq = int(rng.choice([1, 7, 40]))
    offs, cids, scs = pkg.corpus.make_queries(nq, n_clusters, seed=seed + 1, clusters_per_user=int(rng.choice([1, 8, 50, 90])),
                                              n_topics=topics)
    if rng.random() < 0.3:  # quantised scores: ties everywhere
        co.scores[:] = np.maximum(np.round(co.scores * 8) / 8, 0.125)
        # (lists must stay sorted by score desc, ties id asc)
        for i in range(len(co.cluster_ids)):
            b, e = co.list_offsets[i], co.list_offsets[i + 1]
            o = np.lexsort((co.tweet_ids[b:e], -co.scores[b:e]))
            co.tweet_ids[b:e] = co.tweet_ids[b:e][o]; co.scores[b:e] = co.scores[b:e][o]
    if wild:
        flip = rng.random(len(co.scores))
        co.scores[flip < 0.10] *= -1.0
        co.scores[flip > 0.96] = 0.0
        for i in range(len(co.cluster_ids)):
            b, e = co.list_offsets[i], co.list_offsets[i + 1]
            o = rng.permutation(e - b)
            co.tweet_ids[b:e] = co.tweet_ids[b:e][o]; co.scores[b:e] = co.scores[b:e][o]
    P = int(rng.choice([1, 2, 8, 32, 64]))
    variant = int(rng.choice([0, 0, 1, 2]))
    cfgs, sources = [], []
    for q in range(nq):
        cfgs.append(pkg.SimClustersANNConfig(
            maxNumResults=int(rng.choice([1, 10, 200, 400, 1000])), minScore=float(rng.choice([0.0, 0.0, 0.05, -1.0])),
            maxTopTweetsPerCluster=int(rng.choice([1, 20, 300, 800, 3000])), maxScanClusters=int(rng.choice([1, 5, 50, 80])),
            maxTweetCandidateAgeHours=int(rng.choice([24, 24, 6, 175200])), minTweetCandidateAgeHours=int(rng.choice([0, 0, 2])),
            annAlgorithm=pkg.ScoringAlgorithm(int(rng.choice([1, 2, 3, 4])))))
        sources.append(int(rng.choice(co.tweet_ids)) if rng.random() < 0.3 else None)
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=P)
    src_ids = np.array([s if s is not None else 0 for s in sources], np.int64)
    has_src = np.array([s is not None for s in sources], np.uint8)
    qb = pkg.QueryBatch(index, offs, cids, scs, cfgs, now_ms=co.now_ms, variant=pkg.Variant(variant), source_tweet_ids=src_ids,
                        has_source_tweet=has_src)
    qb.run(); qb.finish()
    ids, scores, counts, msz = qb.results()
    for q in range(nq):
        o_ids, o_sc, o_msz = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], sources[q], cfgs[q], co.now_ms,
                                               co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, variant=variant)
        ok = (counts[q] == len(o_ids) and msz[q] == o_msz and np.array_equal(ids[q, :counts[q]], o_ids)
              and np.array_equal(scores[q, :counts[q]].view(np.int64), o_sc.view(np.int64)))
        if not ok:
            return f"seed {seed} query {q}: P={P} variant={variant} cfg={cfgs[q]} tweets={n_tweets} clusters={n_clusters} " \
                   f"counts {counts[q]} vs {len(o_ids)} msz {msz[q]} vs {o_msz}"
    qb.close(); index.close()
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--wild", action="store_true", help="unsorted lists with negative and zero scores")
    a = ap.parse_args()
    pkg = _pkg.load_package()
    t0, n, seed = time.time(), 0, a.seed * 100_000
    while time.time() - t0 < a.seconds:
        bad = one_case(pkg, seed, wild=a.wild)
        if bad:
            print("MISMATCH", bad, flush=True)
            sys.exit(1)
        n += 1; seed += 1
        if n % 20 == 0:
            print(f"{n} cases ok, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz ok: {n} cases in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
