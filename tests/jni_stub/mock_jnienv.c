/* tests/jni_stub/mock_jnienv.c — TEST-ONLY stand-in for the part of a JVM that bindings/jni/fmx_jni.c talks to, so that the glue
 * RUNS (not only type-checks) on a machine without a JDK: tests/test_jni_glue.py, tests/test_gpu_jni_glue.py build this file and
 * the glue against tests/jni_stub/jni.h into one shared object, link it to libfmx.so and call the Java_* entry points the way a
 * JVM would — arrays as objects, results read back out of them, exceptions as a pending {class, message}.
 *
 * NOT a JVM and no claim about one: the function table has the stub header's layout (a real JNIEnv has ~230 slots), there is no
 * GC, no class loading, no threads.  What it does hold the glue to is the CONTRACT of the calls it uses, in its strictest legal
 * form: Get<T>ArrayElements always hands out a COPY (so a result released with JNI_ABORT is lost: the tests pre-fill every
 * output array and compare all of it), every Get must be matched by exactly one Release
 * (mock_outstanding), no JNI call other than the exception ones may be made while an exception is pending (mock_violations).
 * The Java class, compiled and run under a JVM, stays SURVEY §8's row f2: blocked by the image. */
#include <jni.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { KIND_CLASS = 1, KIND_BYTES = 2, KIND_CHARS = 3, KIND_INTS = 4, KIND_LONGS = 5 };

struct _jobject {
    int kind;
    jsize len;
    void *data;      /* the array's storage ("the Java heap") */
    char name[96];   /* classes */
};

static int g_outstanding;       /* Get<T>ArrayElements without their Release so far */
static int g_violations;        /* contract breaches seen (see below) */
static char g_violation[160];
static int g_pending;           /* an exception is pending */
static char g_pending_class[96];
static char g_pending_message[512];
static int g_gets, g_copy_backs, g_aborts;

static void violation(const char *what) {
    if (!g_violations) snprintf(g_violation, sizeof g_violation, "%s", what);
    ++g_violations;
}
static size_t elem_size(int kind) {
    return kind == KIND_BYTES ? 1 : kind == KIND_CHARS ? 2 : kind == KIND_INTS ? 4 : kind == KIND_LONGS ? 8 : 0;
}
/* JNI forbids most calls while an exception is pending */
static void no_pending(const char *fn) {
    if (g_pending) {
        char b[128];
        snprintf(b, sizeof b, "%s called with an exception pending", fn);
        violation(b);
    }
}

static jobject new_object(int kind, jsize len) {
    jobject o = (jobject)calloc(1, sizeof *o);
    if (!o) return NULL;
    o->kind = kind;
    o->len = len;
    if (kind != KIND_CLASS) {
        o->data = calloc((size_t)(len > 0 ? len : 1), elem_size(kind));  /* Java arrays start zeroed */
        if (!o->data) {
            free(o);
            return NULL;
        }
    }
    return o;
}

static jclass m_FindClass(JNIEnv *env, const char *name) {
    (void)env;
    no_pending("FindClass");
    jobject o = new_object(KIND_CLASS, 0);  /* leaked on purpose: a handful of bytes per thrown exception in a test */
    if (o) snprintf(o->name, sizeof o->name, "%s", name ? name : "");
    return o;
}
static jint m_ThrowNew(JNIEnv *env, jclass clazz, const char *msg) {
    (void)env;
    if (!clazz || clazz->kind != KIND_CLASS) {
        violation("ThrowNew without a class");
        return -1;
    }
    if (g_pending) violation("ThrowNew with an exception pending");
    g_pending = 1;
    snprintf(g_pending_class, sizeof g_pending_class, "%s", clazz->name);
    snprintf(g_pending_message, sizeof g_pending_message, "%s", msg ? msg : "");
    return 0;
}
static jsize m_GetArrayLength(JNIEnv *env, jarray a) {
    (void)env;
    no_pending("GetArrayLength");
    if (!a || a->kind == KIND_CLASS) {
        violation("GetArrayLength of something that is no array");
        return 0;
    }
    return a->len;
}
static jbyteArray m_NewByteArray(JNIEnv *env, jsize len) {
    (void)env;
    no_pending("NewByteArray");
    return len < 0 ? NULL : new_object(KIND_BYTES, len);
}
static jlongArray m_NewLongArray(JNIEnv *env, jsize len) {
    (void)env;
    no_pending("NewLongArray");
    return len < 0 ? NULL : new_object(KIND_LONGS, len);
}

/* always a copy, with a small header in front that remembers which array it came from */
struct copy_head {
    jobject from;
    uint64_t magic;
};
#define COPY_MAGIC 0x6a6e69636f707921ull
static void *get_elements(jarray a, int kind, jboolean *is_copy, const char *fn) {
    no_pending(fn);
    if (!a || a->kind != kind) {
        violation("Get<T>ArrayElements on an array of another type");
        return NULL;
    }
    const size_t bytes = (size_t)a->len * elem_size(kind);
    struct copy_head *h = (struct copy_head *)malloc(sizeof *h + (bytes ? bytes : 1));
    if (!h) return NULL;
    h->from = a;
    h->magic = COPY_MAGIC;
    memcpy(h + 1, a->data, bytes);
    if (is_copy) *is_copy = JNI_TRUE;
    ++g_outstanding;
    ++g_gets;
    return h + 1;
}
/* Release<T>ArrayElements may be called with an exception pending (the specification lists it among the safe ones) */
static void release_elements(jarray a, void *elems, jint mode, int kind) {
    if (!elems) {
        violation("Release<T>ArrayElements of a null pointer");
        return;
    }
    struct copy_head *h = (struct copy_head *)elems - 1;
    if (h->magic != COPY_MAGIC || h->from != a || !a || a->kind != kind) {
        violation("Release<T>ArrayElements of a pointer this array did not hand out");
        return;
    }
    if (mode == 0) {
        memcpy(a->data, elems, (size_t)a->len * elem_size(kind));
        ++g_copy_backs;
    } else if (mode == JNI_ABORT) {
        ++g_aborts;
    } else {
        violation("Release<T>ArrayElements with a mode the glue is not expected to use");
    }
    h->magic = 0;
    free(h);
    --g_outstanding;
}
static jbyte *m_GetByteArrayElements(JNIEnv *e, jbyteArray a, jboolean *c) { (void)e; return (jbyte *)get_elements(a, KIND_BYTES, c, "GetByteArrayElements"); }
static jchar *m_GetCharArrayElements(JNIEnv *e, jcharArray a, jboolean *c) { (void)e; return (jchar *)get_elements(a, KIND_CHARS, c, "GetCharArrayElements"); }
static jint *m_GetIntArrayElements(JNIEnv *e, jintArray a, jboolean *c) { (void)e; return (jint *)get_elements(a, KIND_INTS, c, "GetIntArrayElements"); }
static jlong *m_GetLongArrayElements(JNIEnv *e, jlongArray a, jboolean *c) { (void)e; return (jlong *)get_elements(a, KIND_LONGS, c, "GetLongArrayElements"); }
static void m_ReleaseByteArrayElements(JNIEnv *e, jbyteArray a, jbyte *p, jint m) { (void)e; release_elements(a, p, m, KIND_BYTES); }
static void m_ReleaseCharArrayElements(JNIEnv *e, jcharArray a, jchar *p, jint m) { (void)e; release_elements(a, p, m, KIND_CHARS); }
static void m_ReleaseIntArrayElements(JNIEnv *e, jintArray a, jint *p, jint m) { (void)e; release_elements(a, p, m, KIND_INTS); }
static void m_ReleaseLongArrayElements(JNIEnv *e, jlongArray a, jlong *p, jint m) { (void)e; release_elements(a, p, m, KIND_LONGS); }
static void m_SetByteArrayRegion(JNIEnv *env, jbyteArray a, jsize start, jsize len, const jbyte *buf) {
    (void)env;
    no_pending("SetByteArrayRegion");
    if (!a || a->kind != KIND_BYTES || start < 0 || len < 0 || (int64_t)start + len > a->len) {
        violation("SetByteArrayRegion outside the array");  /* a JVM throws ArrayIndexOutOfBoundsException */
        return;
    }
    memcpy((jbyte *)a->data + start, buf, (size_t)len);
}

static const struct JNINativeInterface_ g_table = {
    m_FindClass,
    m_ThrowNew,
    m_GetArrayLength,
    m_NewByteArray,
    m_NewLongArray,
    m_GetByteArrayElements,
    m_GetCharArrayElements,
    m_GetIntArrayElements,
    m_GetLongArrayElements,
    m_ReleaseByteArrayElements,
    m_ReleaseCharArrayElements,
    m_ReleaseIntArrayElements,
    m_ReleaseLongArrayElements,
    m_SetByteArrayRegion,
};
static JNIEnv g_env = &g_table;

/* ---- what the tests call (ctypes) ---------------------------------------------------------------------------------------- */
#define API __attribute__((visibility("default")))

API JNIEnv *mock_env(void) { return &g_env; }
/* kind: 2 byte[], 3 char[], 4 int[], 5 long[] */
API jobject mock_new_array(int kind, jsize len) { return (kind >= KIND_BYTES && kind <= KIND_LONGS && len >= 0) ? new_object(kind, len) : NULL; }
API void *mock_array_data(jobject a) { return a ? a->data : NULL; }  /* the array's own storage: what Java code would read */
API jsize mock_array_length(jobject a) { return a ? a->len : -1; }
API int mock_array_kind(jobject a) { return a ? a->kind : 0; }
API void mock_free_array(jobject a) {
    if (a) {
        free(a->data);
        free(a);
    }
}
API int mock_exception_pending(void) { return g_pending; }
API const char *mock_exception_class(void) { return g_pending ? g_pending_class : ""; }
API const char *mock_exception_message(void) { return g_pending ? g_pending_message : ""; }
API void mock_exception_clear(void) { g_pending = 0; }
API int mock_outstanding(void) { return g_outstanding; }
API int mock_violations(void) { return g_violations; }
API const char *mock_first_violation(void) { return g_violations ? g_violation : ""; }
API void mock_counters(int *gets, int *copy_backs, int *aborts) {
    if (gets) *gets = g_gets;
    if (copy_backs) *copy_backs = g_copy_backs;
    if (aborts) *aborts = g_aborts;
}
API void mock_reset(void) {
    g_pending = g_violations = 0;
    g_gets = g_copy_backs = g_aborts = 0;
    g_violation[0] = 0;
}
