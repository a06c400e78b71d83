/*
 * ann_jni.c -- JNI glue between a JVM shim (class com.twitter.ann.gpu.AnnJni, sketched in INTEGRATION.md section 6) and the C ABIs
 * of include/dense_ann.h (exhaustive search: BruteForceIndex.scala:66-91) and include/hnsw_ann.h (HnswIndex.searchKnn:
 * HnswIndex.java:538-623), plus the index directory loader of include/ann_codec.h (SerializableHnsw.scala:170-190).
 *
 * The search entries have the shape of the reference's own JNI call, faiss `Index_search(ptr, n, x*, k, distances*, labels*)`
 * (ann/src/main/java/com/twitter/ann/faiss/swig/swigfaissJNI.java:269; Index.java:99-101), batched over n queries as faiss is;
 * a Scala adapter in the mould of QueryableIndexAdapter (ann/src/main/scala/com/twitter/ann/faiss/QueryableIndexAdapter.scala:
 * 139-195) turns rows back into NeighborWithDistance.  Buffers are direct ByteBuffers owned by the caller (little-endian): the
 * lifetime hazard QueryableIndexAdapter.scala:128-132 documents -- the JVM freeing `distances` during the call -- cannot occur.
 * All capacities are checked here, before the library writes.
 */
#ifdef SANN_JNI_MINIMAL
#include "jni_min.h"
#else
#include <jni.h>
#endif
#include <stddef.h>
#include <stdint.h>

#include "../../include/ann_codec.h"
#include "../../include/dense_ann.h"
#include "../../include/hnsw_ann.h"

static void throw_runtime(JNIEnv *env, const char *msg) {
  jclass cls = (*env)->FindClass(env, "java/lang/RuntimeException");
  if (cls) (*env)->ThrowNew(env, cls, msg ? msg : "ann: native failure");
}
#define BUF(x) ((x) ? (*env)->GetDirectBufferAddress(env, (x)) : NULL)
#define CAP(x) ((x) ? (*env)->GetDirectBufferCapacity(env, (x)) : 0)

/* ---- exhaustive search ------------------------------------------------------------------------------------------------ */
/* long denseIndexBuild(int device, int metric, long n, int d, ByteBuffer vectors /+ float[n][d] +/, ByteBuffer ids /+ long[n] or null +/, boolean exact) */
JNIEXPORT jlong JNICALL Java_com_twitter_ann_gpu_AnnJni_denseIndexBuild(JNIEnv *env, jclass cls, jint device, jint metric, jlong n, jint d,
                                                                        jobject vectors, jobject ids, jboolean exact) {
  (void)cls;
  if (n < 1 || d < 1 || !vectors || CAP(vectors) / 4 / d < n || (ids && CAP(ids) / 8 < n)) {
    throw_runtime(env, "vectors must hold n x d floats (and ids n longs)");
    return 0;
  }
  dann_index_t *ix = NULL;
  const int rc = exact ? dann_index_build_exact(device, metric, n, d, (const float *)BUF(vectors), (const int64_t *)BUF(ids), &ix)
                       : dann_index_build(device, metric, n, d, (const float *)BUF(vectors), (const int64_t *)BUF(ids), &ix);
  if (rc != DANN_OK) {
    throw_runtime(env, dann_last_error());
    return 0;
  }
  return (jlong)(intptr_t)ix;
}
JNIEXPORT void JNICALL Java_com_twitter_ann_gpu_AnnJni_denseIndexDestroy(JNIEnv *env, jclass cls, jlong index) {
  (void)env;
  (void)cls;
  dann_index_destroy((dann_index_t *)(intptr_t)index);
}
/* void denseSearch(long index, int nq, int d, ByteBuffer x /+ float[nq][d] +/, int k, ByteBuffer distances /+ float[nq][k] +/,
 *                  ByteBuffer labels /+ long[nq][k] +/, ByteBuffer counts /+ int[nq] +/) */
JNIEXPORT void JNICALL Java_com_twitter_ann_gpu_AnnJni_denseSearch(JNIEnv *env, jclass cls, jlong index, jint nq, jint d, jobject x, jint k,
                                                                   jobject distances, jobject labels, jobject counts) {
  (void)cls;
  if (!index || nq < 1 || d < 1 || k < 1 || !x || !distances || !labels || !counts) {
    throw_runtime(env, "index, nq >= 1, k >= 1 and four direct buffers");
    return;
  }
  if (CAP(x) / 4 / d < nq || CAP(distances) / 4 / k < nq || CAP(labels) / 8 / k < nq || CAP(counts) / 4 < nq) {
    throw_runtime(env, "a direct buffer is smaller than nq x k (nq x d) entries");
    return;
  }
  if (dann_search((dann_index_t *)(intptr_t)index, nq, (const float *)BUF(x), k, (float *)BUF(distances), (int64_t *)BUF(labels),
                  (int32_t *)BUF(counts)) != DANN_OK)
    throw_runtime(env, dann_last_error());
}

/* ---- HNSW --------------------------------------------------------------------------------------------------------------- */
/* long hnswIndexBuildInsert(int device, int metric, long n, int d, ByteBuffer vectors, ByteBuffer ids, int maxM, int efConstruction,
 *                           long seed, int nThreads)   HnswIndex.insert for every row (TypedHnswIndex.index / Hnsw.append);
 * nThreads >= 1: on that many host threads (1 = the sequential reference graph); nThreads == 0: on the device, deterministic
 * (hnsw_index_build_insert_gpu: 1M x 256 in 3.7 s, 50M in 149 s) */
JNIEXPORT jlong JNICALL Java_com_twitter_ann_gpu_AnnJni_hnswIndexBuildInsert(JNIEnv *env, jclass cls, jint device, jint metric, jlong n, jint d,
                                                                             jobject vectors, jobject ids, jint maxM, jint efConstruction,
                                                                             jlong seed, jint nThreads) {
  (void)cls;
  if (n < 0 || d < 1 || (n > 0 && (!vectors || CAP(vectors) / 4 / d < n)) || (ids && CAP(ids) / 8 < n)) {
    throw_runtime(env, "vectors must hold n x d floats (and ids n longs)");
    return 0;
  }
  hnsw_index_t *ix = NULL;
  const int rc = nThreads == 0 ? hnsw_index_build_insert_gpu(device, metric, n, d, (const float *)BUF(vectors), (const int64_t *)BUF(ids), maxM,
                                                           efConstruction, (uint64_t)seed, 0, &ix)
                               : hnsw_index_build_insert(device, metric, n, d, (const float *)BUF(vectors), (const int64_t *)BUF(ids), maxM,
                                                         efConstruction, (uint64_t)seed, nThreads, &ix);
  if (rc != HNSW_OK) {
    throw_runtime(env, hnsw_last_error());
    return 0;
  }
  return (jlong)(intptr_t)ix;
}
/* long hnswIndexLoadDirectory(int device, int metric, long n, int d, ByteBuffer vectors, ByteBuffer ids, String directory)
 * the files a reference index directory holds (hnsw_index_metadata, hnsw_internal_index/...), the vectors as a flat array */
JNIEXPORT jlong JNICALL Java_com_twitter_ann_gpu_AnnJni_hnswIndexLoadDirectory(JNIEnv *env, jclass cls, jint device, jint metric, jlong n,
                                                                               jint d, jobject vectors, jobject ids, jstring directory) {
  (void)cls;
  if (n < 0 || d < 1 || !directory || (n > 0 && (!vectors || CAP(vectors) / 4 / d < n)) || (ids && CAP(ids) / 8 < n)) {
    throw_runtime(env, "vectors must hold n x d floats (and ids n longs); directory must not be null");
    return 0;
  }
  const char *dir = (*env)->GetStringUTFChars(env, directory, NULL);
  if (!dir) return 0; /* OutOfMemoryError is pending */
  hnsw_index_t *ix = NULL;
  const int rc = hnsw_index_load_directory(device, metric, n, d, (const float *)BUF(vectors), (const int64_t *)BUF(ids), dir, &ix);
  (*env)->ReleaseStringUTFChars(env, directory, dir);
  if (rc != 0) {
    throw_runtime(env, ann_codec_last_error());
    return 0;
  }
  return (jlong)(intptr_t)ix;
}
JNIEXPORT void JNICALL Java_com_twitter_ann_gpu_AnnJni_hnswIndexDestroy(JNIEnv *env, jclass cls, jlong index) {
  (void)env;
  (void)cls;
  hnsw_index_destroy((hnsw_index_t *)(intptr_t)index);
}
/* void hnswSearch(long index, int nq, int d, ByteBuffer x, int k, int ef, ByteBuffer distances, ByteBuffer labels, ByteBuffer counts)
 * = Hnsw.queryWithDistance for nq queries (HnswParams.ef; Hnsw.scala:125-147) */
JNIEXPORT void JNICALL Java_com_twitter_ann_gpu_AnnJni_hnswSearch(JNIEnv *env, jclass cls, jlong index, jint nq, jint d, jobject x, jint k, jint ef,
                                                                  jobject distances, jobject labels, jobject counts) {
  (void)cls;
  if (!index || nq < 1 || d < 1 || k < 1 || !x || !distances || !labels || !counts) {
    throw_runtime(env, "index, nq >= 1, k >= 1 and four direct buffers");
    return;
  }
  if (CAP(x) / 4 / d < nq || CAP(distances) / 4 / k < nq || CAP(labels) / 8 / k < nq || CAP(counts) / 4 < nq) {
    throw_runtime(env, "a direct buffer is smaller than nq x k (nq x d) entries");
    return;
  }
  int64_t n_ix = 0;
  int32_t d_ix = 0, metric = 0, max_m = 0;
  if (hnsw_index_info((hnsw_index_t *)(intptr_t)index, &n_ix, &d_ix, &metric, &max_m) != HNSW_OK || d_ix != d) {
    throw_runtime(env, "query dimension differs from the index's");
    return;
  }
  if (hnsw_search((hnsw_index_t *)(intptr_t)index, nq, (const float *)BUF(x), k, ef, (float *)BUF(distances), (int64_t *)BUF(labels),
                  (int32_t *)BUF(counts)) != HNSW_OK)
    throw_runtime(env, hnsw_last_error());
}

/* void composeShards(int nShards, int nq, int kIn, ByteBuffer ids /+ long[nShards][nq][kIn] +/, ByteBuffer distances /+ float[..] +/,
 *                    ByteBuffer counts /+ int[nShards][nq] +/, int k, ByteBuffer outIds, ByteBuffer outDistances, ByteBuffer outCounts)
 * = ComposedQueryable.queryWithDistance (ShardApi.scala:71-87) over the batched answers of one index per GPU */
JNIEXPORT void JNICALL Java_com_twitter_ann_gpu_AnnJni_composeShards(JNIEnv *env, jclass cls, jint nShards, jint nq, jint kIn, jobject ids,
                                                                     jobject distances, jobject counts, jint k, jobject outIds,
                                                                     jobject outDistances, jobject outCounts) {
  (void)cls;
  if (nShards < 1 || nq < 0 || kIn < 0 || k < 0 || !counts || !outCounts || (kIn > 0 && (!ids || !distances)) || (k > 0 && (!outIds || !outDistances))) {
    throw_runtime(env, "nShards >= 1, nq / kIn / k >= 0 and the direct buffers they need");
    return;
  }
  const jlong rows = (jlong)nShards * nq;
  if (CAP(counts) / 4 < rows || CAP(outCounts) / 4 < nq || (kIn > 0 && (CAP(ids) / 8 / kIn < rows || CAP(distances) / 4 / kIn < rows)) ||
      (k > 0 && (CAP(outIds) / 8 / k < nq || CAP(outDistances) / 4 / k < nq))) {
    throw_runtime(env, "a direct buffer is smaller than nShards x nq x kIn (nq x k) entries");
    return;
  }
  if (dann_compose_shards(nShards, nq, kIn, (const int64_t *)BUF(ids), (const float *)BUF(distances), (const int32_t *)BUF(counts), k,
                          (int64_t *)BUF(outIds), (float *)BUF(outDistances), (int32_t *)BUF(outCounts)) != DANN_OK)
    throw_runtime(env, dann_last_error());
}
