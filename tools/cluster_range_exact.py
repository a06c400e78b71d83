"""The cluster-id-range partitioning north_star names, with the EXACT merge (sharding.ClusterRangeDeployment), reported
beside the tweet-hash deployment: what it moves per batch and what it costs.

    python tools/cluster_range_exact.py [--tweets 1000000] [--queries 1024] [--shards 8]

Runs N logical cluster-range shards on one GPU over the host generator's corpus, checks the merged answer against the
unsharded index bit for bit, and prints one JSON line.  The 100M-tweet figure is the same count with every scanned
list at least M long (true of that corpus: the scanned clusters are the popular ones)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tweets", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--shards", type=int, default=8)
    a = ap.parse_args()
    pkg = _pkg.load_package()
    co = pkg.corpus.make_corpus(a.tweets)
    offs, cids, scs = pkg.corpus.make_queries(a.queries)
    cfg = pkg.SimClustersANNConfig(maxNumResults=400)
    full = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
    qb = pkg.QueryBatch(full, offs, cids, scs, cfg, now_ms=co.now_ms)
    qb.run(); qb.finish()
    w = qb.results()
    qb.close(); full.close()
    dep = pkg.sharding.ClusterRangeDeployment(pkg, co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, a.shards)
    t0 = time.time()
    ids, sc, cnt, msz, stats = dep.get_tweet_candidates(offs, cids, scs, cfg, now_ms=co.now_ms)
    wall = time.time() - t0
    dep.close()
    same = bool(np.array_equal(cnt, w[2]) and np.array_equal(msz, w[3]) and
                all(np.array_equal(ids[q, :cnt[q]], w[0][q, :cnt[q]]) and
                    np.array_equal(sc[q, :cnt[q]].view(np.int64), w[1][q, :cnt[q]].view(np.int64)) for q in range(len(cnt))))
    N, M = a.shards, cfg.maxTopTweetsPerCluster
    shard_k = pkg.sharding.shard_list_length(400, N)
    out = {"tweets": a.tweets, "queries": a.queries, "cluster_range_shards": N, "equals_unsharded_bit_for_bit": same,
           **stats, "seconds_per_batch_host_orchestrated": round(wall, 2),
           "at_100M_tweets_bytes_moved_per_gpu": int(stats["scanned_clusters"] * M * 16 * (N - 1) / N / N),
           "tweet_hash_exchange_bytes_per_gpu": int((N - 1) / N * a.queries * N / N * shard_k * 16 + 0) if N > 1 else 0,
           "note": "cluster-range re-partitions the scanned lists' top-M prefixes on every batch; tweet-hash exchanges only the cut "
                   "per-shard result lists (shard_k entries per query)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
