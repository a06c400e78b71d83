/*
 * representation_scorer_jni.c -- JNI glue between a JVM shim (class com.twitter.representationscorer.gpu.RsxJni, sketched in
 * INTEGRATION.md section 6) and the C ABI of include/representation_scorer.h.
 *
 * What it serves: PairScoreStore.get / multiGet (src/scala/com/twitter/simclusters_v2/score/ScoreStore.scala:41-69) with both
 * sides hydrated from device-resident stores -- the `score: (V1, V2) => Future[Option[Double]]` member replaced by one
 * batched call -- and ListScoreColumn.fetch (representation-scorer/server/src/main/scala/com/twitter/representationscorer/
 * columns/ListScoreColumn.scala:53-115).  Shape as the reference's only JNI precedent (swig-faiss: a raw native handle in a
 * Java long, primitive arrays; ann/src/main/java/com/twitter/ann/faiss/swig/swigfaissJNI.java:13-23,269).
 * All sizes are checked here before the library reads through a pinned Java array.
 */
#ifdef SANN_JNI_MINIMAL
#include "jni_min.h"
#else
#include <jni.h>
#endif
#include <stddef.h>
#include <stdint.h>

#include "../../include/representation_scorer.h"

static void throw_runtime(JNIEnv *env, const char *msg) {
  jclass cls = (*env)->FindClass(env, "java/lang/RuntimeException");
  if (cls) (*env)->ThrowNew(env, cls, msg ? msg : "representation scorer: native failure");
}
static void throw_illegal_argument(JNIEnv *env, const char *msg) {
  /* ScoreFacadeStore throws IllegalArgumentException for an unknown algorithm (ScoreFacadeStore.scala:25-51) */
  jclass cls = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
  if (cls) (*env)->ThrowNew(env, cls, msg);
}

/* long storeBuild(int device, long[] ids, long[] offsets, int[] clusterIds, double[] scores)
 * ids strictly ascending; embedding i = (clusterIds, scores)[offsets[i] .. offsets[i+1]) in SimClustersEmbedding's form. */
JNIEXPORT jlong JNICALL Java_com_twitter_representationscorer_gpu_RsxJni_storeBuild(JNIEnv *env, jclass cls, jint device, jlongArray ids,
                                                                                    jlongArray offsets, jintArray clusterIds,
                                                                                    jdoubleArray scores) {
  (void)cls;
  if (!ids || !offsets || !clusterIds || !scores) {
    throw_runtime(env, "an array is null");
    return 0;
  }
  const jsize n = (*env)->GetArrayLength(env, ids);
  if ((*env)->GetArrayLength(env, offsets) != n + 1) {
    throw_runtime(env, "offsets must have ids.length + 1 entries");
    return 0;
  }
  const jsize n_c = (*env)->GetArrayLength(env, clusterIds), n_s = (*env)->GetArrayLength(env, scores);
  void *pi = (*env)->GetPrimitiveArrayCritical(env, ids, NULL);
  void *po = (*env)->GetPrimitiveArrayCritical(env, offsets, NULL);
  void *pc = (*env)->GetPrimitiveArrayCritical(env, clusterIds, NULL);
  void *ps = (*env)->GetPrimitiveArrayCritical(env, scores, NULL);
  rsx_store_t *st = NULL;
  int rc = RSX_ENOMEM;
  const char *bad = NULL;
  if (pi && po && pc && ps) {
    const int64_t *o = (const int64_t *)po;
    if (o[0] != 0 || o[n] < 0 || o[n] > (int64_t)n_c || o[n] > (int64_t)n_s) bad = "clusterIds / scores are shorter than offsets says";
    else rc = rsx_store_build(device, n, (const int64_t *)pi, o, (const int32_t *)pc, (const double *)ps, &st);
  }
  if (ps) (*env)->ReleasePrimitiveArrayCritical(env, scores, ps, JNI_ABORT);
  if (pc) (*env)->ReleasePrimitiveArrayCritical(env, clusterIds, pc, JNI_ABORT);
  if (po) (*env)->ReleasePrimitiveArrayCritical(env, offsets, po, JNI_ABORT);
  if (pi) (*env)->ReleasePrimitiveArrayCritical(env, ids, pi, JNI_ABORT);
  if (bad) {
    throw_runtime(env, bad);
    return 0;
  }
  if (rc != RSX_OK) {
    throw_runtime(env, (pi && po && pc && ps) ? rsx_last_error() : "could not pin the arrays");
    return 0;
  }
  return (jlong)(intptr_t)st;
}

JNIEXPORT void JNICALL Java_com_twitter_representationscorer_gpu_RsxJni_storeDestroy(JNIEnv *env, jclass cls, jlong store) {
  (void)env;
  (void)cls;
  rsx_store_destroy((rsx_store_t *)(intptr_t)store);
}

/* void pairScores(long storeA, long storeB, int algorithm, long[] aIds, long[] bIds, double[] outScores, byte[] outPresent)
 * = PairScoreStore.multiGet: outPresent[i] == 0 is the reference's None (either side missing). */
JNIEXPORT void JNICALL Java_com_twitter_representationscorer_gpu_RsxJni_pairScores(JNIEnv *env, jclass cls, jlong storeA, jlong storeB,
                                                                                   jint algorithm, jlongArray aIds, jlongArray bIds,
                                                                                   jdoubleArray outScores, jbyteArray outPresent) {
  (void)cls;
  if (algorithm < 1 || algorithm > 7) {
    throw_illegal_argument(env, "unknown pair scoring algorithm");
    return;
  }
  if (!storeA || !storeB || !aIds || !bIds || !outScores || !outPresent) {
    throw_runtime(env, "a store or an array is null");
    return;
  }
  const jsize n = (*env)->GetArrayLength(env, aIds);
  if ((*env)->GetArrayLength(env, bIds) != n || (*env)->GetArrayLength(env, outScores) < n || (*env)->GetArrayLength(env, outPresent) < n) {
    throw_runtime(env, "bIds / outScores / outPresent must hold aIds.length entries");
    return;
  }
  void *pa = (*env)->GetPrimitiveArrayCritical(env, aIds, NULL);
  void *pb = (*env)->GetPrimitiveArrayCritical(env, bIds, NULL);
  void *po = (*env)->GetPrimitiveArrayCritical(env, outScores, NULL);
  void *pp = (*env)->GetPrimitiveArrayCritical(env, outPresent, NULL);
  int rc = RSX_ENOMEM;
  if (pa && pb && po && pp)
    rc = rsx_store_pair_scores((const rsx_store_t *)(intptr_t)storeA, (const rsx_store_t *)(intptr_t)storeB, algorithm, n, (const int64_t *)pa,
                               (const int64_t *)pb, (double *)po, (uint8_t *)pp);
  if (pp) (*env)->ReleasePrimitiveArrayCritical(env, outPresent, pp, 0);
  if (po) (*env)->ReleasePrimitiveArrayCritical(env, outScores, po, 0);
  if (pb) (*env)->ReleasePrimitiveArrayCritical(env, bIds, pb, JNI_ABORT);
  if (pa) (*env)->ReleasePrimitiveArrayCritical(env, aIds, pa, JNI_ABORT);
  if (rc != RSX_OK) throw_runtime(env, (pa && pb && po && pp) ? rsx_last_error() : "could not pin the arrays");
}

/* void listScores(long targets, long candidates, int algorithm, long targetId, long[] candidateIds, double[] outScores, byte[] outPresent)
 * = ListScoreColumn.fetch: answers in candidate order, None for a candidate (or the target) without an embedding. */
JNIEXPORT void JNICALL Java_com_twitter_representationscorer_gpu_RsxJni_listScores(JNIEnv *env, jclass cls, jlong targets, jlong candidates,
                                                                                   jint algorithm, jlong targetId, jlongArray candidateIds,
                                                                                   jdoubleArray outScores, jbyteArray outPresent) {
  (void)cls;
  if (algorithm < 1 || algorithm > 7) {
    throw_illegal_argument(env, "unknown pair scoring algorithm");
    return;
  }
  if (!targets || !candidates || !candidateIds || !outScores || !outPresent) {
    throw_runtime(env, "a store or an array is null");
    return;
  }
  const jsize n = (*env)->GetArrayLength(env, candidateIds);
  if ((*env)->GetArrayLength(env, outScores) < n || (*env)->GetArrayLength(env, outPresent) < n) {
    throw_runtime(env, "outScores / outPresent must hold candidateIds.length entries");
    return;
  }
  void *pc = (*env)->GetPrimitiveArrayCritical(env, candidateIds, NULL);
  void *po = (*env)->GetPrimitiveArrayCritical(env, outScores, NULL);
  void *pp = (*env)->GetPrimitiveArrayCritical(env, outPresent, NULL);
  int rc = RSX_ENOMEM;
  if (pc && po && pp)
    rc = rsx_store_list_scores((const rsx_store_t *)(intptr_t)targets, (const rsx_store_t *)(intptr_t)candidates, algorithm, targetId, n,
                               (const int64_t *)pc, (double *)po, (uint8_t *)pp);
  if (pp) (*env)->ReleasePrimitiveArrayCritical(env, outPresent, pp, 0);
  if (po) (*env)->ReleasePrimitiveArrayCritical(env, outScores, po, 0);
  if (pc) (*env)->ReleasePrimitiveArrayCritical(env, candidateIds, pc, JNI_ABORT);
  if (rc != RSX_OK) throw_runtime(env, (pc && po && pp) ? rsx_last_error() : "could not pin the arrays");
}
