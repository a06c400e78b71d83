/*
 * simclusters_ann_jni.c -- JNI glue between the JVM shim of INTEGRATION.md (class
 * com.twitter.simclustersann.gpu.SannJni) and the C ABI of include/simclusters_ann.h.
 *
 * The reference's only JNI precedent is swig-faiss: a raw native handle held as a Java long and calls that take
 * primitive arrays (ann/src/main/java/com/twitter/ann/faiss/swig/swigfaissJNI.java:13-23,269; Index.java:11-37).
 * This file follows that shape.  Per-request buffers are direct ByteBuffers owned by the caller, which avoids the
 * lifetime hazard QueryableIndexAdapter.scala:128-132 documents (the JVM freeing `distances` during the call).
 *
 * All logic lives behind the C ABI (which the test-suite exercises); this file only converts argument types and
 * turns a non-zero status into a RuntimeException -- which the controller already maps to an empty response and a
 * failures/<class> counter (simclusters-ann/.../controllers/SimClustersANNController.scala:70-74).
 *
 * Build (deployment): cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *                        simclusters_ann_jni.c -L.. -lsimclusters_amd -o libsimclusters_ann_jni.so
 * Compile check (this image has no JDK): gcc -fsyntax-only -DSANN_JNI_MINIMAL ... (tests/test_abi_cpu.py).
 */
#ifdef SANN_JNI_MINIMAL
#include "jni_min.h"
#else
#include <jni.h>
#endif
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/simclusters_ann.h"

static void throw_runtime(JNIEnv *env, const char *msg) {
  jclass cls = (*env)->FindClass(env, "java/lang/RuntimeException");
  if (cls) (*env)->ThrowNew(env, cls, msg ? msg : "simclusters_amd: native failure");
}

/* long indexBuild(int device, int partitions, int shardId, int nShards,
 *                 int[] clusterIds, long[] listOffsets, long[] tweetIds, double[] scores)
 * The lists as ClusterTweetIndexProviderModule's store returns them (ClusterTweetIndexProviderModule.scala:34-94). */
JNIEXPORT jlong JNICALL Java_com_twitter_simclustersann_gpu_SannJni_indexBuild(JNIEnv *env, jclass cls, jint device,
                                                                                jint partitions, jint shardId, jint nShards,
                                                                                jintArray clusterIds, jlongArray listOffsets,
                                                                                jlongArray tweetIds, jdoubleArray scores) {
  (void)cls;
  sann_index_options_t o = {device, partitions, shardId, nShards};
  if (!clusterIds || !listOffsets || !tweetIds || !scores) {
    throw_runtime(env, "a list array is null");
    return 0;
  }
  const jsize n = (*env)->GetArrayLength(env, clusterIds);
  if ((*env)->GetArrayLength(env, listOffsets) != n + 1) {
    throw_runtime(env, "listOffsets must have clusterIds.length + 1 entries");
    return 0;
  }
  const jsize n_t = (*env)->GetArrayLength(env, tweetIds), n_s = (*env)->GetArrayLength(env, scores);
  /* pinned, no copy; nothing between Get and Release may call back into the JVM */
  void *c = (*env)->GetPrimitiveArrayCritical(env, clusterIds, NULL);
  void *of = (*env)->GetPrimitiveArrayCritical(env, listOffsets, NULL);
  void *t = (*env)->GetPrimitiveArrayCritical(env, tweetIds, NULL);
  void *s = (*env)->GetPrimitiveArrayCritical(env, scores, NULL);
  sann_index_t *ix = NULL;
  int rc = SANN_ENOMEM;
  const char *bad = NULL;
  if (c && of && t && s) {
    /* sann_index_build reads tweetIds / scores up to listOffsets[n]: the Java arrays must reach that far (a short array
     * would be a native read past the end of a Java heap object, not an exception) */
    const int64_t *lo = (const int64_t *)of;
    if (lo[0] < 0 || lo[n] < lo[0] || lo[n] > (int64_t)n_t || lo[n] > (int64_t)n_s) bad = "tweetIds / scores are shorter than listOffsets says";
    else rc = sann_index_build(&o, n, (const int32_t *)c, lo, (const int64_t *)t, (const double *)s, &ix);
  }
  if (s) (*env)->ReleasePrimitiveArrayCritical(env, scores, s, JNI_ABORT);
  if (t) (*env)->ReleasePrimitiveArrayCritical(env, tweetIds, t, JNI_ABORT);
  if (of) (*env)->ReleasePrimitiveArrayCritical(env, listOffsets, of, JNI_ABORT);
  if (c) (*env)->ReleasePrimitiveArrayCritical(env, clusterIds, c, JNI_ABORT);
  if (bad) {
    throw_runtime(env, bad);
    return 0;
  }
  if (rc != SANN_OK) {
    throw_runtime(env, rc == SANN_ENOMEM && !(c && of && t && s) ? "could not pin the list arrays" : sann_last_error());
    return 0;
  }
  return (jlong)(intptr_t)ix;
}

JNIEXPORT void JNICALL Java_com_twitter_simclustersann_gpu_SannJni_indexDestroy(JNIEnv *env, jclass cls, jlong index) {
  (void)env;
  (void)cls;
  sann_index_destroy((sann_index_t *)(intptr_t)index);
}

/* ByteBuffer hostAlloc(long bytes): pinned memory for the request / response buffers the shim keeps across calls
 * (device copies then run at PCIe speed); freed with hostFree(buffer). */
JNIEXPORT jobject JNICALL Java_com_twitter_simclustersann_gpu_SannJni_hostAlloc(JNIEnv *env, jclass cls, jlong bytes) {
  (void)cls;
  void *p = NULL;
  if (sann_host_alloc(bytes, &p) != SANN_OK) {
    throw_runtime(env, sann_last_error());
    return NULL;
  }
  return (*env)->NewDirectByteBuffer(env, p, bytes);
}
JNIEXPORT void JNICALL Java_com_twitter_simclustersann_gpu_SannJni_hostFree(JNIEnv *env, jclass cls, jobject buffer) {
  (void)cls;
  if (buffer) sann_host_free((*env)->GetDirectBufferAddress(env, buffer));
}

/* int getTweetCandidates0(long index, int variant, long nowMs, int nq, int nConfigs,
 *                         ByteBuffer embOffsets, embClusterIds, embScores, sourceTweetIds, hasSourceTweet, configs,
 *                         scanOffsets, scanClusterIds, outIds, outScores, int outStride, outCounts, outMapSizes)
 * = ApproximateCosineSimilarity.apply for nq micro-batched requests (ApproximateCosineSimilarity.scala:26-36).
 * All buffers are direct, little-endian, caller-owned; sourceTweetIds / hasSourceTweet / scanOffsets /
 * scanClusterIds may be null.  `configs` holds nConfigs (1 or nq) sann_config_t records of 40 bytes. */
JNIEXPORT jint JNICALL Java_com_twitter_simclustersann_gpu_SannJni_getTweetCandidates0(
    JNIEnv *env, jclass cls, jlong index, jint variant, jlong nowMs, jint nq, jint nConfigs, jobject embOffsets,
    jobject embClusterIds, jobject embScores, jobject sourceTweetIds, jobject hasSourceTweet, jobject configs,
    jobject scanOffsets, jobject scanClusterIds, jobject outIds, jobject outScores, jint outStride, jobject outCounts,
    jobject outMapSizes) {
  (void)cls;
#define BUF(x) ((x) ? (*env)->GetDirectBufferAddress(env, (x)) : NULL)
  if (!embOffsets || !configs || !outIds || !outScores || !outCounts || !outMapSizes) {
    throw_runtime(env, "a required direct buffer is null");
    return SANN_EINVAL;
  }
  /* sizes first (a negative nq or stride would make every product below negative and every capacity "large enough") */
  if (nq < 0 || outStride < 1 || (nConfigs != 1 && nConfigs != nq)) {
    throw_runtime(env, "nq >= 0, outStride >= 1 and nConfigs in {1, nq}");
    return SANN_EINVAL;
  }
#define CAP(x) ((*env)->GetDirectBufferCapacity(env, (x)))
  if (CAP(configs) < (jlong)nConfigs * (jlong)sizeof(sann_config_t) || CAP(outIds) < (jlong)nq * outStride * 8 ||
      CAP(outScores) < (jlong)nq * outStride * 8 || CAP(outCounts) < (jlong)nq * 4 || CAP(outMapSizes) < (jlong)nq * 4 ||
      CAP(embOffsets) < ((jlong)nq + 1) * 8 || (sourceTweetIds && CAP(sourceTweetIds) < (jlong)nq * 8) ||
      (hasSourceTweet && CAP(hasSourceTweet) < (jlong)nq) || (scanOffsets && CAP(scanOffsets) < ((jlong)nq + 1) * 8)) {
    throw_runtime(env, "a direct buffer is smaller than the batch needs");
    return SANN_EINVAL;
  }
  if ((sourceTweetIds == NULL) != (hasSourceTweet == NULL) || (scanOffsets == NULL) != (scanClusterIds == NULL)) {
    throw_runtime(env, "sourceTweetIds / hasSourceTweet and scanOffsets / scanClusterIds come in pairs");
    return SANN_EINVAL;
  }
  {
    /* the CSR regions the offsets name must lie inside their buffers */
    const int64_t *eo = (const int64_t *)BUF(embOffsets);
    const int64_t e0 = nq ? eo[0] : 0, e1 = nq ? eo[nq] : 0;
    if (e0 < 0 || e1 < e0 ||
        (e1 > e0 && (!embClusterIds || !embScores || CAP(embClusterIds) < e1 * 4 || CAP(embScores) < e1 * 8))) {
      throw_runtime(env, "embClusterIds / embScores are smaller than embOffsets says");
      return SANN_EINVAL;
    }
    if (scanOffsets && nq) {
      const int64_t *so = (const int64_t *)BUF(scanOffsets);
      if (so[0] < 0 || so[nq] < so[0] || CAP(scanClusterIds) < so[nq] * 4) {
        throw_runtime(env, "scanClusterIds is smaller than scanOffsets says");
        return SANN_EINVAL;
      }
    }
  }
#undef CAP
  const int rc = sann_get_tweet_candidates(
      (sann_index_t *)(intptr_t)index, variant, nowMs, nq, (const int64_t *)BUF(embOffsets), (const int32_t *)BUF(embClusterIds),
      (const double *)BUF(embScores), (const int64_t *)BUF(sourceTweetIds), (const uint8_t *)BUF(hasSourceTweet),
      (const sann_config_t *)BUF(configs), nConfigs, (const int64_t *)BUF(scanOffsets), (const int32_t *)BUF(scanClusterIds),
      (int64_t *)BUF(outIds), (double *)BUF(outScores), outStride, (int32_t *)BUF(outCounts), (int32_t *)BUF(outMapSizes));
#undef BUF
  if (rc != SANN_OK) throw_runtime(env, sann_last_error());
  return rc;
}


/* int heavyRank0(long index, long sourceStore, long tweetStore, long nowMs, int nq, ByteBuffer embOffsets, embClusterIds, embScores,
 *                sourceTweetIds, hasSourceTweet, sourceInternalIds, legacyConfig /+ one sann_legacy_config_t, 48 bytes +/,
 *                outIds, outScores, int outStride, outCounts)
 * = the legacy in-process source for nq queries: fetchCandidates + reranking with the heavy rank fused behind the light one
 * (SimClustersANNCandidateSource.scala:107-200, HeavyRanker.scala:28-69).  sourceStore / tweetStore are RsxJni store handles
 * (0 when the config has no heavy ranking). */
JNIEXPORT jint JNICALL Java_com_twitter_simclustersann_gpu_SannJni_heavyRank0(
    JNIEnv *env, jclass cls, jlong index, jlong sourceStore, jlong tweetStore, jlong nowMs, jint nq, jobject embOffsets, jobject embClusterIds,
    jobject embScores, jobject sourceTweetIds, jobject hasSourceTweet, jobject sourceInternalIds, jobject legacyConfig, jobject outIds,
    jobject outScores, jint outStride, jobject outCounts) {
  (void)cls;
#define BUF(x) ((x) ? (*env)->GetDirectBufferAddress(env, (x)) : NULL)
#define CAP(x) ((*env)->GetDirectBufferCapacity(env, (x)))
  if (!embOffsets || !legacyConfig || !outIds || !outScores || !outCounts) {
    throw_runtime(env, "a required direct buffer is null");
    return SANN_EINVAL;
  }
  if (nq < 0 || outStride < 1) {
    throw_runtime(env, "nq >= 0 and outStride >= 1");
    return SANN_EINVAL;
  }
  if (CAP(legacyConfig) < (jlong)sizeof(sann_legacy_config_t) || CAP(outIds) < (jlong)nq * outStride * 8 || CAP(outScores) < (jlong)nq * outStride * 8 ||
      CAP(outCounts) < (jlong)nq * 4 || CAP(embOffsets) < ((jlong)nq + 1) * 8 || (sourceTweetIds && CAP(sourceTweetIds) < (jlong)nq * 8) ||
      (hasSourceTweet && CAP(hasSourceTweet) < (jlong)nq) || (sourceInternalIds && CAP(sourceInternalIds) < (jlong)nq * 8)) {
    throw_runtime(env, "a direct buffer is smaller than the batch needs");
    return SANN_EINVAL;
  }
  if ((sourceTweetIds == NULL) != (hasSourceTweet == NULL)) {
    throw_runtime(env, "sourceTweetIds / hasSourceTweet come as a pair");
    return SANN_EINVAL;
  }
  {
    const int64_t *eo = (const int64_t *)BUF(embOffsets);
    const int64_t e0 = nq ? eo[0] : 0, e1 = nq ? eo[nq] : 0;
    if (e0 < 0 || e1 < e0 || (e1 > e0 && (!embClusterIds || !embScores || CAP(embClusterIds) < e1 * 4 || CAP(embScores) < e1 * 8))) {
      throw_runtime(env, "embClusterIds / embScores are smaller than embOffsets says");
      return SANN_EINVAL;
    }
  }
  const int rc = sann_heavy_rank((sann_index_t *)(intptr_t)index, (const struct rsx_store *)(intptr_t)sourceStore,
                                 (const struct rsx_store *)(intptr_t)tweetStore, nowMs, nq, (const int64_t *)BUF(embOffsets),
                                 (const int32_t *)BUF(embClusterIds), (const double *)BUF(embScores), (const int64_t *)BUF(sourceTweetIds),
                                 (const uint8_t *)BUF(hasSourceTweet), (const int64_t *)BUF(sourceInternalIds),
                                 (const sann_legacy_config_t *)BUF(legacyConfig), (int64_t *)BUF(outIds), (double *)BUF(outScores), outStride,
                                 (int32_t *)BUF(outCounts));
#undef CAP
#undef BUF
  if (rc != SANN_OK) throw_runtime(env, sann_last_error());
  return rc;
}

/* ---- the native micro-batching queue: what a Finagle worker thread calls, once per request -------------------------------
 * long batcherCreate(long index, int variant, int maxBatch, int maxWaitUs, int nDispatchers)   (0 = the library's defaults)
 * void batcherDestroy(long batcher)
 * int  request0(long batcher, long nowMs, int[] clusterIds, double[] scores, long sourceTweetId, boolean hasSourceTweet,
 *               ByteBuffer config /+ one sann_config_t, 40 bytes +/, long[] outIds, double[] outScores, int[] outCountAndMapSize)
 * = ApproximateCosineSimilarity.apply for ONE request (ApproximateCosineSimilarity.scala:26-36), blocking the calling thread for
 * the batching window + one batch's GPU time, as the reference's call blocks it for its CPU time
 * (SimClustersANNCandidateSource.scala:77-94).  The Java arrays are copied in and out (GetArrayRegion / SetArrayRegion, no
 * pinning across the wait); outIds / outScores must hold min(maxNumResults, 1000) entries; outCountAndMapSize = {count, map size}. */
JNIEXPORT jlong JNICALL Java_com_twitter_simclustersann_gpu_SannJni_batcherCreate(JNIEnv *env, jclass cls, jlong index, jint variant,
                                                                                   jint maxBatch, jint maxWaitUs, jint nDispatchers) {
  (void)cls;
  sann_batcher_options_t o = {variant, maxBatch, maxWaitUs, nDispatchers};
  sann_batcher_t *b = NULL;
  if (sann_batcher_create((sann_index_t *)(intptr_t)index, &o, &b) != SANN_OK) {
    throw_runtime(env, sann_last_error());
    return 0;
  }
  return (jlong)(intptr_t)b;
}
JNIEXPORT void JNICALL Java_com_twitter_simclustersann_gpu_SannJni_batcherDestroy(JNIEnv *env, jclass cls, jlong batcher) {
  (void)env;
  (void)cls;
  sann_batcher_destroy((sann_batcher_t *)(intptr_t)batcher);
}
JNIEXPORT jint JNICALL Java_com_twitter_simclustersann_gpu_SannJni_request0(JNIEnv *env, jclass cls, jlong batcher, jlong nowMs,
                                                                             jintArray clusterIds, jdoubleArray scores, jlong sourceTweetId,
                                                                             jboolean hasSourceTweet, jobject config, jlongArray outIds,
                                                                             jdoubleArray outScores, jintArray outCountAndMapSize) {
  (void)cls;
  enum { MAX_EMB = 4096, MAX_OUT = 1000 };
  int32_t c[MAX_EMB];
  double s[MAX_EMB];
  int64_t ids[MAX_OUT];
  double sc[MAX_OUT];
  if (!batcher || !clusterIds || !scores || !config || !outIds || !outScores || !outCountAndMapSize) {
    throw_runtime(env, "a required argument is null");
    return SANN_EINVAL;
  }
  const jsize n = (*env)->GetArrayLength(env, clusterIds);
  if ((*env)->GetArrayLength(env, scores) != n || n > MAX_EMB) {
    throw_runtime(env, "clusterIds / scores differ in length or hold more than 4096 entries");
    return SANN_EINVAL;
  }
  if ((*env)->GetDirectBufferCapacity(env, config) < (jlong)sizeof(sann_config_t) || (*env)->GetArrayLength(env, outCountAndMapSize) < 2) {
    throw_runtime(env, "config must hold one sann_config_t and outCountAndMapSize two ints");
    return SANN_EINVAL;
  }
  const sann_config_t *cfg = (const sann_config_t *)(*env)->GetDirectBufferAddress(env, config);
  int32_t cap = cfg->max_num_results < MAX_OUT ? cfg->max_num_results : MAX_OUT;
  if (cap < 0) cap = 0;
  if ((*env)->GetArrayLength(env, outIds) < cap || (*env)->GetArrayLength(env, outScores) < cap) {
    throw_runtime(env, "outIds / outScores are shorter than min(maxNumResults, 1000)");
    return SANN_EINVAL;
  }
  { /* copy in: pinned only for the copy, never across the wait below */
    void *pc = (*env)->GetPrimitiveArrayCritical(env, clusterIds, NULL);
    void *ps = (*env)->GetPrimitiveArrayCritical(env, scores, NULL);
    if (pc && ps) {
      memcpy(c, pc, (size_t)n * 4);
      memcpy(s, ps, (size_t)n * 8);
    }
    if (ps) (*env)->ReleasePrimitiveArrayCritical(env, scores, ps, JNI_ABORT);
    if (pc) (*env)->ReleasePrimitiveArrayCritical(env, clusterIds, pc, JNI_ABORT);
    if (!pc || !ps) {
      throw_runtime(env, "could not pin the embedding arrays");
      return SANN_ENOMEM;
    }
  }
  int32_t cm[2] = {0, 0};
  const int rc = sann_batcher_get_tweet_candidates((sann_batcher_t *)(intptr_t)batcher, nowMs, n, c, s, sourceTweetId, hasSourceTweet ? 1 : 0,
                                                   cfg, cap > 0 ? cap : 1, ids, sc, &cm[0], &cm[1]);
  if (rc != SANN_OK) {
    throw_runtime(env, sann_last_error());
    return rc;
  }
  { /* copy out */
    void *pi = (*env)->GetPrimitiveArrayCritical(env, outIds, NULL);
    void *ps = (*env)->GetPrimitiveArrayCritical(env, outScores, NULL);
    void *pm = (*env)->GetPrimitiveArrayCritical(env, outCountAndMapSize, NULL);
    const int ok = pi && ps && pm;
    if (ok) {
      memcpy(pi, ids, (size_t)cm[0] * 8);
      memcpy(ps, sc, (size_t)cm[0] * 8);
      memcpy(pm, cm, 8);
    }
    if (pm) (*env)->ReleasePrimitiveArrayCritical(env, outCountAndMapSize, pm, 0);
    if (ps) (*env)->ReleasePrimitiveArrayCritical(env, outScores, ps, 0);
    if (pi) (*env)->ReleasePrimitiveArrayCritical(env, outIds, pi, 0);
    if (!ok) {
      throw_runtime(env, "could not pin the output arrays");
      return SANN_ENOMEM;
    }
  }
  return rc;
}
