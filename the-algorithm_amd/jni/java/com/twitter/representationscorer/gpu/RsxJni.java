package com.twitter.representationscorer.gpu;

/**
 * Native binding of the representation-scorer pair scores (include/representation_scorer.h); C side:
 * the-algorithm_amd/jni/representation_scorer_jni.c.  Serves PairScoreStore.get / multiGet
 * (src/scala/com/twitter/simclusters_v2/score/ScoreStore.scala:41-69) and ListScoreColumn.fetch
 * (representation-scorer/.../columns/ListScoreColumn.scala:53-115) with both embedding tables resident on the GPU.
 * An unknown algorithm id is an IllegalArgumentException, as ScoreFacadeStore.scala:25-51 throws.
 */
public final class RsxJni {
  static {
    System.loadLibrary("representation_scorer_jni");
  }

  private RsxJni() {}

  /** ids strictly ascending; embedding i = (clusterIds, scores)[offsets[i] .. offsets[i+1]) in SimClustersEmbedding's form. */
  public static native long storeBuild(int device, long[] ids, long[] offsets, int[] clusterIds, double[] scores);

  public static native void storeDestroy(long store);

  /** PairScoreStore.multiGet: outPresent[i] == 0 is the reference's None (either side missing). */
  public static native void pairScores(long storeA, long storeB, int algorithm, long[] aIds, long[] bIds, double[] outScores,
                                       byte[] outPresent);

  /** ListScoreColumn.fetch: answers in candidate order, None for a candidate (or the target) without an embedding. */
  public static native void listScores(long targets, long candidates, int algorithm, long targetId, long[] candidateIds,
                                       double[] outScores, byte[] outPresent);
}
