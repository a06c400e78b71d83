/* jni_harness.c -- a JNIEnv the tests can hand to the JNI glue of the-algorithm_amd/jni/ (test infrastructure).
 *
 * This image has no JDK: the glue is compiled against jni/jni_min.h, whose JNINativeInterface_ lists exactly the entries
 * the glue calls.  This file IMPLEMENTS that table over plain C objects -- an "array" is (data, length), a "direct buffer"
 * is (address, capacity), a "string" is a char pointer -- so that the glue functions can be EXECUTED (argument checks on
 * the CPU; whole calls against the oracle on the GPU) and not merely compiled.  ThrowNew records the message; the tests
 * read it back with jh_exception(). */
#include <stdlib.h>
#include <string.h>

#include "jni_min.h"

typedef struct jh_obj {
  int kind; /* 1 array, 2 direct buffer, 3 string, 4 class */
  void *data;
  jlong n; /* elements (array) / bytes (buffer) */
} jh_obj;

static char g_exception[512];
static char g_exception_class[128];
static jh_obj g_class = {4, NULL, 0};

static jclass jh_FindClass(JNIEnv *env, const char *name) {
  (void)env;
  strncpy(g_exception_class, name, sizeof(g_exception_class) - 1);
  return (jclass)&g_class;
}
static jint jh_ThrowNew(JNIEnv *env, jclass c, const char *msg) {
  (void)env;
  (void)c;
  strncpy(g_exception, msg ? msg : "", sizeof(g_exception) - 1);
  return 0;
}
static jsize jh_GetArrayLength(JNIEnv *env, jarray a) { (void)env; return (jsize)((jh_obj *)a)->n; }
static void *jh_GetPrimitiveArrayCritical(JNIEnv *env, jarray a, jboolean *is_copy) {
  (void)env;
  if (is_copy) *is_copy = 0;
  return ((jh_obj *)a)->data;
}
static void jh_ReleasePrimitiveArrayCritical(JNIEnv *env, jarray a, void *p, jint mode) { (void)env; (void)a; (void)p; (void)mode; }
static void *jh_GetDirectBufferAddress(JNIEnv *env, jobject b) { (void)env; return ((jh_obj *)b)->data; }
static jlong jh_GetDirectBufferCapacity(JNIEnv *env, jobject b) { (void)env; return b ? ((jh_obj *)b)->n : -1; }
static jobject jh_NewDirectByteBuffer(JNIEnv *env, void *addr, jlong cap) {
  (void)env;
  jh_obj *o = (jh_obj *)malloc(sizeof(jh_obj));
  o->kind = 2; o->data = addr; o->n = cap;
  return (jobject)o;
}
static const char *jh_GetStringUTFChars(JNIEnv *env, jstring s, jboolean *is_copy) {
  (void)env;
  if (is_copy) *is_copy = 0;
  return (const char *)((jh_obj *)s)->data;
}
static void jh_ReleaseStringUTFChars(JNIEnv *env, jstring s, const char *c) { (void)env; (void)s; (void)c; }

static const struct JNINativeInterface_ g_table = {
    jh_FindClass, jh_ThrowNew, jh_GetArrayLength, jh_GetPrimitiveArrayCritical, jh_ReleasePrimitiveArrayCritical,
    jh_GetDirectBufferAddress, jh_GetDirectBufferCapacity, jh_NewDirectByteBuffer, jh_GetStringUTFChars, jh_ReleaseStringUTFChars};
static JNIEnv g_env = &g_table;

/* ---- what the tests call (ctypes) ---- */
JNIEnv *jh_env(void) { return &g_env; }
void *jh_array(void *data, long long n_elements) {
  jh_obj *o = (jh_obj *)malloc(sizeof(jh_obj));
  o->kind = 1; o->data = data; o->n = n_elements;
  return o;
}
void *jh_buffer(void *addr, long long capacity_bytes) { return (void *)jh_NewDirectByteBuffer(&g_env, addr, capacity_bytes); }
void *jh_string(const char *s) { jh_obj *o = (jh_obj *)malloc(sizeof(jh_obj)); o->kind = 3; o->data = (void *)s; o->n = (jlong)strlen(s); return o; }
void jh_free(void *o) { free(o); }
const char *jh_exception(void) { return g_exception; }
const char *jh_exception_class(void) { return g_exception_class; }
void jh_clear(void) { g_exception[0] = 0; g_exception_class[0] = 0; }
void *jh_buffer_address(void *b) { return ((jh_obj *)b)->data; }
