/*
 * jni_min.h -- the subset of the Java Native Interface that the glue files of this directory use, declared by hand so that
 * the glue can be COMPILE-CHECKED in an image without a JDK (tests/test_abi_cpu.py runs `gcc -fsyntax-only`).
 * Types and function names follow the JNI specification (jni.h of any JDK >= 8); the function table below lists only
 * the entries the glue calls, so it is NOT layout-compatible with the real JNINativeInterface_: a deployment build
 * defines nothing and includes the JDK's <jni.h> instead (cc -I$JAVA_HOME/include -I$JAVA_HOME/include/linux).
 */
#ifndef SANN_JNI_MIN_H
#define SANN_JNI_MIN_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef float jfloat;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef jint jsize;
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;
typedef jarray jfloatArray;
typedef jarray jbyteArray;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv *env, const char *name);
  jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
  jsize (*GetArrayLength)(JNIEnv *env, jarray array);
  void *(*GetPrimitiveArrayCritical)(JNIEnv *env, jarray array, jboolean *isCopy);
  void (*ReleasePrimitiveArrayCritical)(JNIEnv *env, jarray array, void *carray, jint mode);
  void *(*GetDirectBufferAddress)(JNIEnv *env, jobject buf);
  jlong (*GetDirectBufferCapacity)(JNIEnv *env, jobject buf);
  jobject (*NewDirectByteBuffer)(JNIEnv *env, void *address, jlong capacity);
  const char *(*GetStringUTFChars)(JNIEnv *env, jstring str, jboolean *isCopy);
  void (*ReleaseStringUTFChars)(JNIEnv *env, jstring str, const char *chars);
};
#endif
