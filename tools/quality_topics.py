"""What `recall_at_400_quality` looks like when the corpus is topical.

SURVEY 8(d)'s synthetic corpus draws a tweet's clusters independently of each other; the operator's partial cosine
(over at most N x M postings) is then unrelated to the full cosine and the bench's quality recall reads 0.05.  Real
SimClusters embeddings are topical.  This tool builds the host generator's corpus twice -- as specified, and as the
topic-mixture variant (corpus.topic_cluster_ranks) -- runs the operator on the GPU for a few queries and compares its
top-k with the exact full-cosine top-k (scipy sparse product over every tweet's full embedding).

    python tools/quality_topics.py [--tweets 1000000] [--queries 64] [--topics 2000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _pkg  # noqa: E402


def run(pkg, n_tweets, n_queries, n_topics, k, alg):
    C = pkg.corpus.N_CLUSTERS
    t0 = time.time()
    co = pkg.corpus.make_corpus(n_tweets, C, n_topics=n_topics)
    offs, cids, scs = pkg.corpus.make_queries(n_queries, C, n_topics=n_topics)
    gen_s = time.time() - t0
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
    cfg = pkg.SimClustersANNConfig(maxNumResults=k, minScore=0.0, maxTopTweetsPerCluster=800, maxScanClusters=50,
                                   maxTweetCandidateAgeHours=24, minTweetCandidateAgeHours=0, annAlgorithm=alg)
    batch = pkg.QueryBatch(index, offs, cids, scs, [cfg], now_ms=co.now_ms)
    batch.run(); batch.finish()
    ids, scores, counts, _ = batch.results()
    # exact full cosine: T [tweets x clusters] . u / (|T_row| |u|)
    T = sp.csr_matrix((co.tweet_emb_scores, co.tweet_emb_clusters.astype(np.int64), co.tweet_emb_offsets), shape=(n_tweets, C + 1))
    t_norm = np.sqrt(np.asarray(T.multiply(T).sum(axis=1)).ravel())
    rec = []
    for q in range(n_queries):
        b, e = offs[q], offs[q + 1]
        u = sp.csr_matrix((scs[b:e], (np.zeros(e - b, np.int64), cids[b:e].astype(np.int64))), shape=(1, C + 1))
        dots = np.asarray((T @ u.T).todense()).ravel()
        cos = np.where(t_norm > 0, dots / np.maximum(t_norm, 1e-300), 0.0)
        top = np.argpartition(-cos, k)[:k]
        top = top[cos[top] > 0]
        exact = set(int(x) for x in co.tweet_id_of[top])
        got = set(int(x) for x in ids[q, :counts[q]])
        rec.append(len(exact & got) / max(len(exact), 1))
    batch.close(); index.close()
    return {"n_topics": n_topics, "tweets": n_tweets, "queries": n_queries, "k": k, "algorithm": alg.name,
            "recall_at_k_quality": float(np.mean(rec)), "min": float(np.min(rec)), "max": float(np.max(rec)),
            "postings": int(len(co.tweet_ids)), "generate_s": round(gen_s, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tweets", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--topics", type=int, default=2000)
    ap.add_argument("--k", type=int, default=400)
    a = ap.parse_args()
    pkg = _pkg.load_package()
    for nt in (0, a.topics):
        print(json.dumps(run(pkg, a.tweets, a.queries, nt, a.k, pkg.ScoringAlgorithm.CosineSimilarity)), flush=True)


if __name__ == "__main__":
    main()
