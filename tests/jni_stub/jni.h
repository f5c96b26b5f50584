/* tests/jni_stub/jni.h — TEST-ONLY declarations of the part of the JNI the glue in bindings/jni/fmx_jni.c uses (types and
 * function-table entries as the JNI specification defines them), so that `gcc -fsyntax-only` type-checks the glue against
 * include/fmx.h on a machine without a JDK: a change of fmx.h that the glue does not follow then breaks a test
 * (tests/test_abi.py).  Not a JDK header, never shipped, never linked: a real build uses $JAVA_HOME/include/jni.h
 * (bindings/build.sh). */
#ifndef FMX_TEST_JNI_STUB_H
#define FMX_TEST_JNI_STUB_H
#include <stdint.h>

typedef uint8_t jboolean;
#define JNI_FALSE 0
#define JNI_TRUE 1
typedef int8_t jbyte;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef int32_t jint;
typedef int64_t jlong;
typedef jint jsize;

struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jbyteArray;
typedef jarray jcharArray;
typedef jarray jintArray;
typedef jarray jlongArray;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *env, const char *name);
    jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jsize (*GetArrayLength)(JNIEnv *env, jarray array);
    jbyteArray (*NewByteArray)(JNIEnv *env, jsize len);
    jlongArray (*NewLongArray)(JNIEnv *env, jsize len);
    jbyte *(*GetByteArrayElements)(JNIEnv *env, jbyteArray array, jboolean *isCopy);
    jchar *(*GetCharArrayElements)(JNIEnv *env, jcharArray array, jboolean *isCopy);
    jint *(*GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
    jlong *(*GetLongArrayElements)(JNIEnv *env, jlongArray array, jboolean *isCopy);
    void (*ReleaseByteArrayElements)(JNIEnv *env, jbyteArray array, jbyte *elems, jint mode);
    void (*ReleaseCharArrayElements)(JNIEnv *env, jcharArray array, jchar *elems, jint mode);
    void (*ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
    void (*ReleaseLongArrayElements)(JNIEnv *env, jlongArray array, jlong *elems, jint mode);
    void (*SetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, const jbyte *buf);
};
#endif
