package com.twitter.simclustersann.gpu;

import java.nio.ByteBuffer;

/**
 * Native binding of the simclusters-ann hot path (include/simclusters_ann.h) -- the class the JVM shim of INTEGRATION.md loads.
 * The C side is the-algorithm_amd/jni/simclusters_ann_jni.c; tests/test_jni_cpu.py holds every method below against the symbol
 * and the parameter list the compiled glue exports.  Shape as the reference's own JNI precedent, swig-faiss: a raw native
 * handle in a long, primitive arrays and direct buffers (ann/src/main/java/com/twitter/ann/faiss/swig/swigfaissJNI.java:13-23,269).
 * Direct buffers are little-endian and caller-owned.  A native failure surfaces as a RuntimeException, which
 * SimClustersANNController.scala:70-74 already turns into an empty response and a failures/<class> counter.
 */
public final class SannJni {
  static {
    System.loadLibrary("simclusters_ann_jni");  // links libsimclusters_amd.so
  }

  private SannJni() {}

  /** ClusterTweetIndexProviderModule.scala:34-94: the lists as the store returns them, CSR. Returns sann_index_t*. */
  public static native long indexBuild(int device, int partitions, int shardId, int nShards, int[] clusterIds, long[] listOffsets,
                                       long[] tweetIds, double[] scores);

  public static native void indexDestroy(long index);

  /** Pinned memory for request / response buffers kept across calls (results then arrive without a bounce copy). */
  public static native ByteBuffer hostAlloc(long bytes);

  public static native void hostFree(ByteBuffer buffer);

  /** ApproximateCosineSimilarity.apply for nq micro-batched requests (ApproximateCosineSimilarity.scala:26-36). */
  public static native int getTweetCandidates0(long index, int variant, long nowMs, int nq, int nConfigs, ByteBuffer embOffsets,
                                               ByteBuffer embClusterIds, ByteBuffer embScores, ByteBuffer sourceTweetIds,
                                               ByteBuffer hasSourceTweet, ByteBuffer configs, ByteBuffer scanOffsets,
                                               ByteBuffer scanClusterIds, ByteBuffer outIds, ByteBuffer outScores, int outStride,
                                               ByteBuffer outCounts, ByteBuffer outMapSizes);

  /** The legacy in-process source with its heavy ranker (SimClustersANNCandidateSource.scala:107-200, HeavyRanker.scala:28-69). */
  public static native int heavyRank0(long index, long sourceStore, long tweetStore, long nowMs, int nq, ByteBuffer embOffsets,
                                      ByteBuffer embClusterIds, ByteBuffer embScores, ByteBuffer sourceTweetIds,
                                      ByteBuffer hasSourceTweet, ByteBuffer sourceInternalIds, ByteBuffer legacyConfig,
                                      ByteBuffer outIds, ByteBuffer outScores, int outStride, ByteBuffer outCounts);

  /** The native micro-batching queue over an index (0 = the library's defaults). Returns sann_batcher_t*. */
  public static native long batcherCreate(long index, int variant, int maxBatch, int maxWaitUs, int nDispatchers);

  public static native void batcherDestroy(long batcher);

  /**
   * ONE request, from a Finagle worker thread (SimClustersANNCandidateSource.scala:77-94): blocks for the batching window plus
   * one batch's GPU time. outCountAndMapSize = {number of results, candidateScoresMap.size}.
   */
  public static native int request0(long batcher, long nowMs, int[] clusterIds, double[] scores, long sourceTweetId,
                                    boolean hasSourceTweet, ByteBuffer config, long[] outIds, double[] outScores,
                                    int[] outCountAndMapSize);
}
