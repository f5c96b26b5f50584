/*
 * fmx_jni.c — JNI glue between com.dynatrace.fm.gpu.GpuFmIndex and libfmx.so (include/fmx.h).
 * Never compiled against a JDK here (none in the build image).  What the test suites do instead: a syntax pass against a
 * test-local declaration of the JNI entries used (tests/jni_stub/jni.h, tests/test_abi.py), and every entry point below RUN
 * against a mock JNIEnv with copy-always array semantics (tests/jni_stub/mock_jnienv.c; tests/test_jni_glue.py on the CPU,
 * tests/test_gpu_jni_glue.py on the GPU against the oracle).  Build on a JDK host with bindings/build.sh, or by hand:
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *      bindings/jni/fmx_jni.c -Lindex4j_amd -lfmx -o libfmx_jni.so
 * Java `char` is an unsigned 16-bit UTF-16 code unit == the uint16_t the C ABI takes; `int` == int32_t.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>

#include "fmx.h"

#define CLS "com/dynatrace/fm/gpu/GpuFmIndex"

static void throw_lib_error(JNIEnv *env, int rc) {
    const char *msg = fmx_last_error();
    const char *cls = "java/lang/RuntimeException";
    if (rc == FMX_E_VERSION || rc == FMX_E_FORMAT) cls = "java/io/IOException";           /* SER:46-56 */
    if (rc == FMX_E_ALPHABET) {                                                             /* FM:423-426 */
        cls = "java/lang/IllegalArgumentException";
        msg = "Input has more than 32767 different symbols";
    }
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls), msg ? msg : "libfmx error");
}

/* raw pointers only cross into libfmx after the Java arrays have been measured against what the call will touch */
static int bad_args(JNIEnv *env, const char *what) {
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), what);
    return 1;
}
static int too_short(JNIEnv *env, jarray a, jlong need, const char *what) {
    if (a == NULL || (jlong)(*env)->GetArrayLength(env, a) < need) return bad_args(env, what);
    return 0;
}
static jlong null_argument(JNIEnv *env, const char *what) {
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/NullPointerException"), what);
    return 0;
}
/* offsets has n + 1 entries, starts at 0, never decreases and ends inside chars */
static int bad_patterns(JNIEnv *env, jcharArray chars, jintArray offsets, jint n) {
    if (n < 0) return bad_args(env, "negative batch size");
    if (chars == NULL) return bad_args(env, "chars is null");
    if (too_short(env, offsets, (jlong)n + 1, "offsets shorter than n + 1")) return 1;
    jint *po = (*env)->GetIntArrayElements(env, offsets, NULL);
    jsize n_chars = (*env)->GetArrayLength(env, chars);
    int bad = po[0] != 0 || po[n] > n_chars;
    for (jint i = 0; i < n && !bad; ++i) bad = po[i + 1] < po[i];
    (*env)->ReleaseIntArrayElements(env, offsets, po, JNI_ABORT);
    return bad ? bad_args(env, "pattern offsets are not a partition of chars") : 0;
}

JNIEXPORT jlong JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeLoad(JNIEnv *env, jclass c, jbyteArray ser, jint device) {
    if (ser == NULL) return null_argument(env, "serialized");
    jsize len = (*env)->GetArrayLength(env, ser);
    jbyte *p = (*env)->GetByteArrayElements(env, ser, NULL);
    fmx_index *idx = NULL;
    int rc = fmx_load((const uint8_t *)p, (size_t)len, &idx);
    (*env)->ReleaseByteArrayElements(env, ser, p, JNI_ABORT);
    if (rc == FMX_OK) rc = fmx_to_device(idx, device);
    if (rc != FMX_OK) {
        fmx_free(idx);
        throw_lib_error(env, rc);
        return 0;
    }
    return (jlong)(intptr_t)idx;
}

/* FmIndex.write bytes of the index (fmx_save), framed like Serialization.writeToByteArray or bare */
JNIEXPORT jbyteArray JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeSave(JNIEnv *env, jclass c, jlong h, jboolean framed) {
    uint8_t *buf = NULL;
    size_t len = 0;
    int rc = fmx_save((const fmx_index *)(intptr_t)h, framed ? 1 : 0, &buf, &len);
    if (rc != FMX_OK) {
        throw_lib_error(env, rc);
        return NULL;
    }
    jbyteArray out = len <= 0x7fffffff ? (*env)->NewByteArray(env, (jsize)len) : NULL;
    if (out) (*env)->SetByteArrayRegion(env, out, 0, (jsize)len, (const jbyte *)buf);
    fmx_free_buffer(buf);
    return out;
}

/* whether toSerialized()'s character-map order is the replayed HashMap order (fmx.h fmx_save_key_order_modelled) */
JNIEXPORT jboolean JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeSavedOrderModelled(JNIEnv *env, jclass c, jlong h) {
    int rc = fmx_save_key_order_modelled((const fmx_index *)(intptr_t)h);
    if (rc < 0) {
        throw_lib_error(env, rc);
        return JNI_FALSE;
    }
    return rc == 1 ? JNI_TRUE : JNI_FALSE;
}

/* buildOnGpu: the constructor's suffix-array stage (FM:329-394) runs on `device` — same index, byte for byte */
JNIEXPORT jlong JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeBuild(JNIEnv *env, jclass c, jcharArray text,
                                                                        jint sampleRate, jboolean extract, jint device,
                                                                        jboolean buildOnGpu) {
    if (text == NULL) return null_argument(env, "text");
    jsize n = (*env)->GetArrayLength(env, text);
    jchar *p = (*env)->GetCharArrayElements(env, text, NULL);
    fmx_index *idx = NULL;
    int rc = buildOnGpu ? fmx_build_on_device((const uint16_t *)p, n, sampleRate, extract ? 1 : 0, device, &idx, NULL,
                                              NULL, NULL)
                        : fmx_build((const uint16_t *)p, n, sampleRate, extract ? 1 : 0, &idx);
    (*env)->ReleaseCharArrayElements(env, text, p, JNI_ABORT);
    if (rc == FMX_OK) rc = fmx_to_device(idx, device);
    if (rc != FMX_OK) {
        fmx_free(idx);
        throw_lib_error(env, rc);
        return 0;
    }
    return (jlong)(intptr_t)idx;
}

JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeFree(JNIEnv *env, jclass c, jlong h) {
    fmx_free((fmx_index *)(intptr_t)h);
}
JNIEXPORT jint JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeInputLength(JNIEnv *env, jclass c, jlong h) {
    return fmx_input_length((const fmx_index *)(intptr_t)h);
}
JNIEXPORT jint JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeAlphabetLength(JNIEnv *env, jclass c, jlong h) {
    return fmx_alphabet_length((const fmx_index *)(intptr_t)h);
}

JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeCountBatch(JNIEnv *env, jclass c, jlong h, jcharArray chars,
                                                                            jintArray offsets, jint n, jintArray counts,
                                                                            jintArray status) {
    if (bad_patterns(env, chars, offsets, n) || too_short(env, counts, n, "counts shorter than n") ||
        too_short(env, status, n, "status shorter than n"))
        return;
    jchar *pc = (*env)->GetCharArrayElements(env, chars, NULL);
    jint *po = (*env)->GetIntArrayElements(env, offsets, NULL);
    jint *pn = (*env)->GetIntArrayElements(env, counts, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    int rc = fmx_count_batch((const fmx_index *)(intptr_t)h, (const uint16_t *)pc, (const int32_t *)po, n, (int32_t *)pn, NULL,
                             (int32_t *)ps);
    (*env)->ReleaseCharArrayElements(env, chars, pc, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, offsets, po, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, counts, pn, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}

JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeLocateBatch(JNIEnv *env, jclass c, jlong h, jcharArray chars,
                                                                             jintArray offsets, jint n, jint maxMatches,
                                                                             jintArray locations, jint locCap,
                                                                             jintArray found, jintArray status) {
    if (bad_patterns(env, chars, offsets, n) || locCap < 0 ||
        too_short(env, locations, (jlong)n * locCap, "locations shorter than n * locCap") ||
        too_short(env, found, n, "found shorter than n") || too_short(env, status, n, "status shorter than n"))
        return;
    jchar *pc = (*env)->GetCharArrayElements(env, chars, NULL);
    jint *po = (*env)->GetIntArrayElements(env, offsets, NULL);
    jint *pl = (*env)->GetIntArrayElements(env, locations, NULL);
    jint *pf = (*env)->GetIntArrayElements(env, found, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    int rc = fmx_locate_batch((const fmx_index *)(intptr_t)h, (const uint16_t *)pc, (const int32_t *)po, n, maxMatches,
                              (int32_t *)pl, locCap, (int32_t *)pf, NULL, (int32_t *)ps);
    (*env)->ReleaseCharArrayElements(env, chars, pc, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, offsets, po, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, locations, pl, 0);
    (*env)->ReleaseIntArrayElements(env, found, pf, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}

JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeExtractBatch(JNIEnv *env, jclass c, jlong h, jintArray start,
                                                                              jintArray stop, jint n, jcharArray dst,
                                                                              jint dstLen, jint offset, jintArray outLen,
                                                                              jintArray status) {
    if (n < 0 || dstLen < 0 || too_short(env, start, n, "start shorter than n") || too_short(env, stop, n, "stop shorter than n") ||
        too_short(env, dst, (jlong)n * dstLen, "dst shorter than n * dstLen") ||
        too_short(env, outLen, n, "outLen shorter than n") || too_short(env, status, n, "status shorter than n"))
        return;
    jint *pa = (*env)->GetIntArrayElements(env, start, NULL);
    jint *pb = (*env)->GetIntArrayElements(env, stop, NULL);
    jchar *pd = (*env)->GetCharArrayElements(env, dst, NULL);
    jint *pl = (*env)->GetIntArrayElements(env, outLen, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    int rc = fmx_extract_batch((const fmx_index *)(intptr_t)h, (const int32_t *)pa, (const int32_t *)pb, n, (uint16_t *)pd,
                               dstLen, offset, (int32_t *)pl, NULL, (int32_t *)ps);
    (*env)->ReleaseIntArrayElements(env, start, pa, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, stop, pb, JNI_ABORT);
    (*env)->ReleaseCharArrayElements(env, dst, pd, 0);
    (*env)->ReleaseIntArrayElements(env, outLen, pl, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}

JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeExtractBoundaryBatch(
    JNIEnv *env, jclass c, jlong h, jintArray from, jint n, jchar boundary, jint mode, jcharArray dst, jint dstLen,
    jint offset, jintArray outLen, jintArray status, jintArray aux) {
    if (n < 0 || dstLen < 0 || too_short(env, from, n, "from shorter than n") ||
        too_short(env, dst, (jlong)n * dstLen, "dst shorter than n * dstLen") || too_short(env, outLen, n, "outLen shorter than n") ||
        too_short(env, status, n, "status shorter than n") || too_short(env, aux, n, "aux shorter than n"))
        return;
    jint *pa = (*env)->GetIntArrayElements(env, from, NULL);
    jchar *pd = (*env)->GetCharArrayElements(env, dst, NULL);
    jint *pl = (*env)->GetIntArrayElements(env, outLen, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    jint *px = (*env)->GetIntArrayElements(env, aux, NULL);
    int rc = fmx_extract_boundary_batch((const fmx_index *)(intptr_t)h, (const int32_t *)pa, n, (uint16_t)boundary, mode,
                                        (uint16_t *)pd, dstLen, offset, (int32_t *)pl, NULL, (int32_t *)ps, (int32_t *)px);
    (*env)->ReleaseIntArrayElements(env, from, pa, JNI_ABORT);
    (*env)->ReleaseCharArrayElements(env, dst, pd, 0);
    (*env)->ReleaseIntArrayElements(env, outLen, pl, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    (*env)->ReleaseIntArrayElements(env, aux, px, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}

/* locate -> extract (mode < 0, rowLength = extraction length) or locate -> extractUntilBoundary{,Left,Right}
 * (mode 0/1/2) with the hit table kept on the device: fmx_locate_extract_batch / fmx_locate_lines_batch */
JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeLocatePipeline(
    JNIEnv *env, jclass c, jlong h, jcharArray chars, jintArray offsets, jint n, jint maxMatches, jint mode, jchar boundary,
    jint rowLength, jintArray locations, jintArray found, jcharArray rows, jintArray outLen, jintArray status,
    jintArray hitStatus, jintArray hitAux) {
    const jlong slots = (jlong)n * maxMatches;
    if (bad_patterns(env, chars, offsets, n) || maxMatches < 1 || rowLength < 0 ||
        too_short(env, locations, slots, "locations shorter than n * maxMatches") || too_short(env, found, n, "found shorter than n") ||
        too_short(env, rows, slots * rowLength, "rows shorter than n * maxMatches * rowLength") ||
        too_short(env, outLen, slots, "outLen shorter than n * maxMatches") || too_short(env, status, n, "status shorter than n") ||
        too_short(env, hitStatus, slots, "hitStatus shorter than n * maxMatches") ||
        too_short(env, hitAux, slots, "hitAux shorter than n * maxMatches"))
        return;
    jchar *pc = (*env)->GetCharArrayElements(env, chars, NULL);
    jint *po = (*env)->GetIntArrayElements(env, offsets, NULL);
    jint *pl = (*env)->GetIntArrayElements(env, locations, NULL);
    jint *pf = (*env)->GetIntArrayElements(env, found, NULL);
    jchar *pr = (*env)->GetCharArrayElements(env, rows, NULL);
    jint *pn = (*env)->GetIntArrayElements(env, outLen, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    jint *ph = (*env)->GetIntArrayElements(env, hitStatus, NULL);
    jint *pa = (*env)->GetIntArrayElements(env, hitAux, NULL);
    const fmx_index *idx = (const fmx_index *)(intptr_t)h;
    int rc = mode < 0 ? fmx_locate_extract_batch(idx, (const uint16_t *)pc, (const int32_t *)po, n, maxMatches, rowLength,
                                                 (int32_t *)pl, (int32_t *)pf, (uint16_t *)pr, (int32_t *)pn, NULL,
                                                 (int32_t *)ps, (int32_t *)ph)
                      : fmx_locate_lines_batch(idx, (const uint16_t *)pc, (const int32_t *)po, n, maxMatches, boundary, mode,
                                               rowLength, (int32_t *)pl, (int32_t *)pf, (uint16_t *)pr, (int32_t *)pn, NULL,
                                               (int32_t *)ps, (int32_t *)ph, (int32_t *)pa);
    (*env)->ReleaseCharArrayElements(env, chars, pc, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, offsets, po, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, locations, pl, 0);
    (*env)->ReleaseIntArrayElements(env, found, pf, 0);
    (*env)->ReleaseCharArrayElements(env, rows, pr, 0);
    (*env)->ReleaseIntArrayElements(env, outLen, pn, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    (*env)->ReleaseIntArrayElements(env, hitStatus, ph, 0);
    (*env)->ReleaseIntArrayElements(env, hitAux, pa, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}

/* K indexes of one long text (texts of 2^31 chars or more): summed counts, fmx_count_segments */
JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeCountSegments(JNIEnv *env, jclass c, jlongArray handles,
                                                                               jcharArray chars, jintArray offsets, jint n,
                                                                               jlongArray counts, jintArray status) {
    if (handles == NULL || bad_patterns(env, chars, offsets, n) || too_short(env, counts, n, "counts shorter than n") ||
        too_short(env, status, n, "status shorter than n"))
        return;
    jsize k = (*env)->GetArrayLength(env, handles);
    if (k < 1) {
        bad_args(env, "no segments");
        return;
    }
    const fmx_index **segs = (const fmx_index **)malloc((size_t)k * sizeof *segs);  /* any number of segments */
    if (!segs) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/OutOfMemoryError"), "segment table");
        return;
    }
    jlong *ph = (*env)->GetLongArrayElements(env, handles, NULL);
    for (jsize i = 0; i < k; ++i) segs[i] = (const fmx_index *)(intptr_t)ph[i];
    jchar *pc = (*env)->GetCharArrayElements(env, chars, NULL);
    jint *po = (*env)->GetIntArrayElements(env, offsets, NULL);
    jlong *pn = (*env)->GetLongArrayElements(env, counts, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    int rc = fmx_count_segments(segs, (int32_t)k, (const uint16_t *)pc, (const int32_t *)po, n, (int64_t *)pn, NULL,
                                (int32_t *)ps);
    free(segs);
    (*env)->ReleaseLongArrayElements(env, handles, ph, JNI_ABORT);
    (*env)->ReleaseCharArrayElements(env, chars, pc, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, offsets, po, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, counts, pn, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}

/* ---- replicas: one index on several GPUs (fmx.h "replicas") ---------------------------------------------------------- */

/* a long[] of handles as the array of fmx_index pointers the *_multi calls take (free() it); NULL + exception on failure */
static const fmx_index **handle_table(JNIEnv *env, jlongArray handles, jlong need) {
    if (handles == NULL || (*env)->GetArrayLength(env, handles) < 1 || (need > 0 && (jlong)(*env)->GetArrayLength(env, handles) != need)) {
        bad_args(env, "no replicas, or not replicas x segments handles");
        return NULL;
    }
    jsize k = (*env)->GetArrayLength(env, handles);
    const fmx_index **t = (const fmx_index **)malloc((size_t)k * sizeof *t);
    if (!t) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/OutOfMemoryError"), "replica table");
        return NULL;
    }
    jlong *ph = (*env)->GetLongArrayElements(env, handles, NULL);
    for (jsize i = 0; i < k; ++i) t[i] = (const fmx_index *)(intptr_t)ph[i];
    (*env)->ReleaseLongArrayElements(env, handles, ph, JNI_ABORT);
    return t;
}

JNIEXPORT jlongArray JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeReplicate(JNIEnv *env, jclass c, jlong h, jintArray devices) {
    if (devices == NULL || (*env)->GetArrayLength(env, devices) < 1) {
        bad_args(env, "no devices");
        return NULL;
    }
    jsize k = (*env)->GetArrayLength(env, devices);
    fmx_index **made = (fmx_index **)calloc((size_t)k, sizeof *made);
    if (!made) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/OutOfMemoryError"), "replica table");
        return NULL;
    }
    jint *pd = (*env)->GetIntArrayElements(env, devices, NULL);
    int rc = fmx_replicate((const fmx_index *)(intptr_t)h, (const int32_t *)pd, (int32_t)k, made);
    (*env)->ReleaseIntArrayElements(env, devices, pd, JNI_ABORT);
    jlongArray out = NULL;
    if (rc != FMX_OK) {
        throw_lib_error(env, rc);
    } else if ((out = (*env)->NewLongArray(env, k)) != NULL) {
        jlong *po = (*env)->GetLongArrayElements(env, out, NULL);
        for (jsize i = 0; i < k; ++i) po[i] = (jlong)(intptr_t)made[i];
        (*env)->ReleaseLongArrayElements(env, out, po, 0);
    } else {
        for (jsize i = 0; i < k; ++i) fmx_free(made[i]);
    }
    free(made);
    return out;
}

JNIEXPORT jint JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeDeviceOf(JNIEnv *env, jclass c, jlong h) {
    return fmx_device_of((const fmx_index *)(intptr_t)h);
}

JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeCountBatchMulti(JNIEnv *env, jclass c, jlongArray handles,
                                                                                 jcharArray chars, jintArray offsets, jint n,
                                                                                 jintArray counts, jintArray status) {
    if (bad_patterns(env, chars, offsets, n) || too_short(env, counts, n, "counts shorter than n") ||
        too_short(env, status, n, "status shorter than n"))
        return;
    const fmx_index **reps = handle_table(env, handles, 0);
    if (!reps) return;
    jsize k = (*env)->GetArrayLength(env, handles);
    jchar *pc = (*env)->GetCharArrayElements(env, chars, NULL);
    jint *po = (*env)->GetIntArrayElements(env, offsets, NULL);
    jint *pn = (*env)->GetIntArrayElements(env, counts, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    int rc = fmx_count_batch_multi(reps, (int32_t)k, (const uint16_t *)pc, (const int32_t *)po, n, (int32_t *)pn, NULL, (int32_t *)ps);
    free(reps);
    (*env)->ReleaseCharArrayElements(env, chars, pc, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, offsets, po, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, counts, pn, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}

JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeLocateBatchMulti(JNIEnv *env, jclass c, jlongArray handles,
                                                                                  jcharArray chars, jintArray offsets, jint n,
                                                                                  jint maxMatches, jintArray locations, jint locCap,
                                                                                  jintArray found, jintArray status) {
    if (bad_patterns(env, chars, offsets, n) || locCap < 0 ||
        too_short(env, locations, (jlong)n * locCap, "locations shorter than n * locCap") ||
        too_short(env, found, n, "found shorter than n") || too_short(env, status, n, "status shorter than n"))
        return;
    const fmx_index **reps = handle_table(env, handles, 0);
    if (!reps) return;
    jsize k = (*env)->GetArrayLength(env, handles);
    jchar *pc = (*env)->GetCharArrayElements(env, chars, NULL);
    jint *po = (*env)->GetIntArrayElements(env, offsets, NULL);
    jint *pl = (*env)->GetIntArrayElements(env, locations, NULL);
    jint *pf = (*env)->GetIntArrayElements(env, found, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    int rc = fmx_locate_batch_multi(reps, (int32_t)k, (const uint16_t *)pc, (const int32_t *)po, n, maxMatches, (int32_t *)pl, locCap,
                                    (int32_t *)pf, NULL, (int32_t *)ps);
    free(reps);
    (*env)->ReleaseCharArrayElements(env, chars, pc, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, offsets, po, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, locations, pl, 0);
    (*env)->ReleaseIntArrayElements(env, found, pf, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}

JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeExtractBatchMulti(JNIEnv *env, jclass c, jlongArray handles,
                                                                                   jintArray start, jintArray stop, jint n,
                                                                                   jcharArray dst, jint dstLen, jint offset,
                                                                                   jintArray outLen, jintArray status) {
    if (n < 0 || dstLen < 0 || too_short(env, start, n, "start shorter than n") || too_short(env, stop, n, "stop shorter than n") ||
        too_short(env, dst, (jlong)n * dstLen, "dst shorter than n * dstLen") ||
        too_short(env, outLen, n, "outLen shorter than n") || too_short(env, status, n, "status shorter than n"))
        return;
    const fmx_index **reps = handle_table(env, handles, 0);
    if (!reps) return;
    jsize k = (*env)->GetArrayLength(env, handles);
    jint *pa = (*env)->GetIntArrayElements(env, start, NULL);
    jint *pb = (*env)->GetIntArrayElements(env, stop, NULL);
    jchar *pd = (*env)->GetCharArrayElements(env, dst, NULL);
    jint *pl = (*env)->GetIntArrayElements(env, outLen, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    int rc = fmx_extract_batch_multi(reps, (int32_t)k, (const int32_t *)pa, (const int32_t *)pb, n, (uint16_t *)pd, dstLen, offset,
                                     (int32_t *)pl, NULL, (int32_t *)ps);
    free(reps);
    (*env)->ReleaseIntArrayElements(env, start, pa, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, stop, pb, JNI_ABORT);
    (*env)->ReleaseCharArrayElements(env, dst, pd, 0);
    (*env)->ReleaseIntArrayElements(env, outLen, pl, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}

JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeExtractBoundaryBatchMulti(
    JNIEnv *env, jclass c, jlongArray handles, jintArray from, jint n, jchar boundary, jint mode, jcharArray dst, jint dstLen,
    jint offset, jintArray outLen, jintArray status, jintArray aux) {
    if (n < 0 || dstLen < 0 || too_short(env, from, n, "from shorter than n") ||
        too_short(env, dst, (jlong)n * dstLen, "dst shorter than n * dstLen") || too_short(env, outLen, n, "outLen shorter than n") ||
        too_short(env, status, n, "status shorter than n") || too_short(env, aux, n, "aux shorter than n"))
        return;
    const fmx_index **reps = handle_table(env, handles, 0);
    if (!reps) return;
    jsize k = (*env)->GetArrayLength(env, handles);
    jint *pa = (*env)->GetIntArrayElements(env, from, NULL);
    jchar *pd = (*env)->GetCharArrayElements(env, dst, NULL);
    jint *pl = (*env)->GetIntArrayElements(env, outLen, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    jint *px = (*env)->GetIntArrayElements(env, aux, NULL);
    int rc = fmx_extract_boundary_batch_multi(reps, (int32_t)k, (const int32_t *)pa, n, (uint16_t)boundary, mode, (uint16_t *)pd, dstLen,
                                              offset, (int32_t *)pl, NULL, (int32_t *)ps, (int32_t *)px);
    free(reps);
    (*env)->ReleaseIntArrayElements(env, from, pa, JNI_ABORT);
    (*env)->ReleaseCharArrayElements(env, dst, pd, 0);
    (*env)->ReleaseIntArrayElements(env, outLen, pl, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    (*env)->ReleaseIntArrayElements(env, aux, px, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}

/* BASELINE configs[4] from a Java host: handles = replicas x segments, replica-major (fmx_count_locate_segments_multi) */
JNIEXPORT void JNICALL Java_com_dynatrace_fm_gpu_GpuFmIndex_nativeCountLocateSegmentsMulti(
    JNIEnv *env, jclass c, jlongArray handles, jint replicas, jint segments, jlongArray segmentBase, jcharArray chars,
    jintArray offsets, jint n, jint maxMatches, jlongArray counts, jlongArray locations, jintArray found, jintArray status) {
    if (replicas < 1 || segments < 1 || maxMatches < 1 || bad_patterns(env, chars, offsets, n) ||
        too_short(env, segmentBase, segments, "segmentBase shorter than the number of segments") ||
        too_short(env, counts, n, "counts shorter than n") ||
        too_short(env, locations, (jlong)n * maxMatches, "locations shorter than n * maxMatches") ||
        too_short(env, found, n, "found shorter than n") || too_short(env, status, n, "status shorter than n"))
        return;
    const fmx_index **segs = handle_table(env, handles, (jlong)replicas * segments);
    if (!segs) return;
    jlong *pb = (*env)->GetLongArrayElements(env, segmentBase, NULL);
    jchar *pc = (*env)->GetCharArrayElements(env, chars, NULL);
    jint *po = (*env)->GetIntArrayElements(env, offsets, NULL);
    jlong *pn = (*env)->GetLongArrayElements(env, counts, NULL);
    jlong *pl = (*env)->GetLongArrayElements(env, locations, NULL);
    jint *pf = (*env)->GetIntArrayElements(env, found, NULL);
    jint *ps = (*env)->GetIntArrayElements(env, status, NULL);
    int rc = fmx_count_locate_segments_multi(segs, replicas, segments, (const int64_t *)pb, (const uint16_t *)pc, (const int32_t *)po, n,
                                             maxMatches, (int64_t *)pn, NULL, (int64_t *)pl, (int32_t *)pf, (int32_t *)ps);
    free(segs);
    (*env)->ReleaseLongArrayElements(env, segmentBase, pb, JNI_ABORT);
    (*env)->ReleaseCharArrayElements(env, chars, pc, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, offsets, po, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, counts, pn, 0);
    (*env)->ReleaseLongArrayElements(env, locations, pl, 0);
    (*env)->ReleaseIntArrayElements(env, found, pf, 0);
    (*env)->ReleaseIntArrayElements(env, status, ps, 0);
    if (rc != FMX_OK) throw_lib_error(env, rc);
}
