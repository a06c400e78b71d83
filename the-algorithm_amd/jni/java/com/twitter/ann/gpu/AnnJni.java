package com.twitter.ann.gpu;

import java.nio.ByteBuffer;

/**
 * Native binding of the ann/ dense exhaustive search (include/dense_ann.h; BruteForceIndex.scala:66-91) and the HNSW walk
 * (include/hnsw_ann.h; HnswIndex.java:538-623); C side: the-algorithm_amd/jni/ann_jni.c.  The search entries have the shape of
 * the reference's own JNI call, faiss Index_search(ptr, n, x, k, distances, labels) (swigfaissJNI.java:269), batched over n
 * queries; a Scala adapter in the mould of QueryableIndexAdapter.scala:139-195 turns rows back into NeighborWithDistance.
 * Buffers are direct, little-endian and caller-owned.
 */
public final class AnnJni {
  static {
    System.loadLibrary("ann_jni");
  }

  private AnnJni() {}

  /** vectors: float[n][d]; ids: long[n] or null; exact: keep the fp32 rows and prove every result (dann_index_build_exact). */
  public static native long denseIndexBuild(int device, int metric, long n, int d, ByteBuffer vectors, ByteBuffer ids, boolean exact);

  public static native void denseIndexDestroy(long index);

  /** x: float[nq][d]; distances: float[nq][k]; labels: long[nq][k]; counts: int[nq]. */
  public static native void denseSearch(long index, int nq, int d, ByteBuffer x, int k, ByteBuffer distances, ByteBuffer labels,
                                        ByteBuffer counts);

  /** HnswIndex.insert for every row (TypedHnswIndex.index / Hnsw.append); nThreads = 0 builds on the device. */
  public static native long hnswIndexBuildInsert(int device, int metric, long n, int d, ByteBuffer vectors, ByteBuffer ids, int maxM,
                                                 int efConstruction, long seed, int nThreads);

  /** The files a reference index directory holds (hnsw_index_metadata, hnsw_internal_index/...). */
  public static native long hnswIndexLoadDirectory(int device, int metric, long n, int d, ByteBuffer vectors, ByteBuffer ids,
                                                   String directory);

  public static native void hnswIndexDestroy(long index);

  /**
   * ComposedQueryable.queryWithDistance (ShardApi.scala:71-87) over the batched answers of one index per GPU: ids long[nShards][nq][kIn],
   * distances float[nShards][nq][kIn], counts int[nShards][nq] in; the k nearest per query, by (distance, id), out.
   */
  public static native void composeShards(int nShards, int nq, int kIn, ByteBuffer ids, ByteBuffer distances, ByteBuffer counts, int k,
                                          ByteBuffer outIds, ByteBuffer outDistances, ByteBuffer outCounts);

  /** Hnsw.queryWithDistance for nq queries (HnswParams.ef; Hnsw.scala:125-147). */
  public static native void hnswSearch(long index, int nq, int d, ByteBuffer x, int k, int ef, ByteBuffer distances, ByteBuffer labels,
                                       ByteBuffer counts);
}
