/*
 * GpuFmIndex — Java host class that keeps index4j's FmIndex query API and forwards the backward-search
 * path to libfmx.so (MI355X) through JNI.  SOURCE ONLY: the build image has no JDK, so this file has
 * never been compiled; it documents the binding a maintainer adds (see INTEGRATION.md).
 *
 * Drop-in recipe: build (or deserialize) a com.dynatrace.fm.FmIndex exactly as today, then
 *     GpuFmIndex gpu = GpuFmIndex.fromFmIndex(fmIndex, 0);          // FmIndex.write bytes -> fmx_load -> HBM
 * and call count / locate / extract / extractUntilBoundary* with the reference's signatures, or the
 * batch variants.  Exceptions carry the reference's types and messages (FmIndex.java:566-576, 591-593,
 * 619-625, 659-661, 732-737).
 */
package com.dynatrace.fm.gpu;

import com.dynatrace.fm.FmIndex;
import com.dynatrace.serialization.Serialization;
import java.io.IOException;

public final class GpuFmIndex implements AutoCloseable {

    static {
        System.loadLibrary("fmx_jni"); // bindings/jni/fmx_jni.c, linked against libfmx.so
    }

    private long handle; // fmx_index*

    private GpuFmIndex(long handle) {
        this.handle = handle;
    }

    /** Ships FmIndex.write(...) bytes (Serialization.writeToByteArray, SER:67-79) to the GPU. */
    public static GpuFmIndex fromFmIndex(FmIndex index, int device) throws IOException {
        byte[] bytes = Serialization.writeToByteArray(FmIndex::write, index);
        return fromSerialized(bytes, device);
    }

    public static GpuFmIndex fromSerialized(byte[] bytes, int device) throws IOException {
        long h = nativeLoad(bytes, device);
        return new GpuFmIndex(h);
    }

    /** new FmIndexBuilder().setSampleRate(s).setEnableExtraction(e).build(text) built natively (FMB:34-62). */
    public static GpuFmIndex build(char[] text, int sampleRate, boolean enableExtraction, int device) {
        return new GpuFmIndex(nativeBuild(text, sampleRate, enableExtraction, device, false));
    }

    /** The same index with the constructor's suffix-array stage (FM:329-394) computed on the GPU. */
    public static GpuFmIndex buildOnGpu(char[] text, int sampleRate, boolean enableExtraction, int device) {
        return new GpuFmIndex(nativeBuild(text, sampleRate, enableExtraction, device, true));
    }

    // ---- the reference's scalar signatures (FmIndex.java:443-941) ----
    public int count(char[] pattern) {
        return count(pattern, 0, pattern.length);
    }

    public int count(char[] pattern, int offset, int length) {
        int[] counts = new int[1];
        int[] status = new int[1];
        nativeCountBatch(handle, slice(pattern, offset, length), new int[] {0, length}, 1, counts, status);
        rethrow(status[0], 0);
        return counts[0];
    }

    public int locate(char[] pattern, int[] locations) {
        return locate(pattern, 0, pattern.length, locations, -1);
    }

    public int locate(char[] pattern, int offset, int length, int[] locations, int maxMatches) {
        int[] found = new int[1];
        int[] status = new int[1];
        nativeLocateBatch(handle, slice(pattern, offset, length), new int[] {0, length}, 1, maxMatches, locations,
                locations.length, found, status);
        rethrow(status[0], 0);
        return found[0];
    }

    public int extract(int start, int stop, char[] destination, int offset) {
        int[] len = new int[1];
        int[] status = new int[1];
        nativeExtractBatch(handle, new int[] {start}, new int[] {stop}, 1, destination, destination.length, offset, len,
                status);
        rethrow(status[0], 0);
        return len[0];
    }

    public int extractUntilBoundary(int from, char[] destination, int offset, char boundary) {
        return boundary(0, from, destination, offset, boundary);
    }

    public int extractUntilBoundaryLeft(int from, char[] destination, int offset, char boundary) {
        return boundary(1, from, destination, offset, boundary);
    }

    public int extractUntilBoundaryRight(int from, char[] destination, int offset, char boundary) {
        return boundary(2, from, destination, offset, boundary);
    }

    public int getInputLength() {
        return nativeInputLength(handle);
    }

    public int getAlphabetLength() {
        return nativeAlphabetLength(handle);
    }

    // ---- batch surface: one kernel launch for the whole batch ----
    /** patterns concatenated in {@code chars}; pattern i is chars[offsets[i] .. offsets[i+1]). */
    public int[] countBatch(char[] chars, int[] offsets) {
        int n = offsets.length - 1;
        int[] counts = new int[n];
        int[] status = new int[n];
        nativeCountBatch(handle, chars, offsets, n, counts, status);
        for (int s : status) {
            rethrow(s, 0);
        }
        return counts;
    }

    /** locations: n rows of maxMatches ints; returns the number located per pattern. */
    public int[] locateBatch(char[] chars, int[] offsets, int maxMatches, int[] locations) {
        int n = offsets.length - 1;
        int[] found = new int[n];
        int[] status = new int[n];
        nativeLocateBatch(handle, chars, offsets, n, maxMatches, locations, maxMatches, found, status);
        for (int s : status) {
            rethrow(s, 0);
        }
        return found;
    }

    /**
     * locate, then extract(loc, min(getInputLength(), loc + extractLength), row, 0) for every hit, in one device
     * call (what locateAndExtractBenchmark does hit by hit). rows: n * maxMatches rows of extractLength chars;
     * locations / outLen / hitStatus: n * maxMatches slots; returns the number located per pattern.
     */
    public int[] locateExtractBatch(char[] chars, int[] offsets, int maxMatches, int extractLength, int[] locations,
            char[] rows, int[] outLen, int[] hitStatus) {
        return pipeline(chars, offsets, maxMatches, -1, (char) 0, extractLength, locations, rows, outLen, hitStatus);
    }

    /** locate, then extractUntilBoundary (mode 0) / Left (1) / Right (2) for every hit. */
    public int[] locateLinesBatch(char[] chars, int[] offsets, int maxMatches, char boundary, int mode, int rowLength,
            int[] locations, char[] rows, int[] outLen, int[] hitStatus) {
        return pipeline(chars, offsets, maxMatches, mode, boundary, rowLength, locations, rows, outLen, hitStatus);
    }

    private int[] pipeline(char[] chars, int[] offsets, int maxMatches, int mode, char boundary, int rowLength,
            int[] locations, char[] rows, int[] outLen, int[] hitStatus) {
        int n = offsets.length - 1;
        int[] found = new int[n];
        int[] status = new int[n];
        int[] hitAux = new int[n * maxMatches];
        nativeLocatePipeline(handle, chars, offsets, n, maxMatches, mode, boundary, rowLength, locations, found, rows,
                outLen, status, hitStatus, hitAux);
        for (int s : status) {
            rethrow(s, 0);
        }
        return found;
    }

    /** count() summed over K indexes of one long text (a Java int cannot address 2^31 chars, FmIndex.java:131). */
    public static long[] countSegments(GpuFmIndex[] segments, char[] chars, int[] offsets) {
        int n = offsets.length - 1;
        long[] handles = new long[segments.length];
        for (int i = 0; i < segments.length; i++) {
            handles[i] = segments[i].handle;
        }
        long[] counts = new long[n];
        int[] status = new int[n];
        nativeCountSegments(handles, chars, offsets, n, counts, status);
        for (int s : status) {
            rethrow(s, 0);
        }
        return counts;
    }

    /**
     * This index on several GPUs of the node (fmx_replicate): FmIndex is immutable and @ThreadSafe (FmIndex.java:82) — index4j's
     * own throughput benchmark gives every thread an index of its own (FmIndexThroughputState.java:30) — so the image is copied
     * to every device named (peer copies out of this index's HBM over xGMI, all destinations at once) and a batch is cut into
     * contiguous shards, one per replica: no exchange on the query path.  A device may be named more than once.  This index stays
     * usable and is NOT one of the replicas; close both.
     */
    public Replicas replicate(int[] devices) {
        return new Replicas(nativeReplicate(handle, devices));
    }

    /** The device ordinal this index is resident on (fmx_device_of). */
    public int device() {
        return nativeDeviceOf(handle);
    }

    /**
     * A replica set: the batch surface of GpuFmIndex with the batch sharded over the replicas by the library
     * (fmx_*_multi: shard r = fmx_shard_range(n, replicas, r) runs on replica r from a host thread of the library's own and
     * stores into its slice of the caller's arrays).  Results equal those of the single index, entry by entry.
     */
    public static final class Replicas implements AutoCloseable {
        private long[] handles; // fmx_index* per replica

        private Replicas(long[] handles) {
            this.handles = handles;
        }

        public int size() {
            return handles.length;
        }

        public int[] countBatch(char[] chars, int[] offsets) {
            int n = offsets.length - 1;
            int[] counts = new int[n];
            int[] status = new int[n];
            nativeCountBatchMulti(handles, chars, offsets, n, counts, status);
            for (int s : status) {
                rethrow(s, 0);
            }
            return counts;
        }

        /** locations: n rows of maxMatches ints; returns the number located per pattern. */
        public int[] locateBatch(char[] chars, int[] offsets, int maxMatches, int[] locations) {
            int n = offsets.length - 1;
            int[] found = new int[n];
            int[] status = new int[n];
            nativeLocateBatchMulti(handles, chars, offsets, n, maxMatches, locations, maxMatches, found, status);
            for (int s : status) {
                rethrow(s, 0);
            }
            return found;
        }

        /** extract(start[i], stop[i], row i, offset) for every i; rows: n rows of rowLength chars; returns the lengths. */
        public int[] extractBatch(int[] start, int[] stop, char[] rows, int rowLength, int offset) {
            int n = start.length;
            int[] len = new int[n];
            int[] status = new int[n];
            nativeExtractBatchMulti(handles, start, stop, n, rows, rowLength, offset, len, status);
            for (int s : status) {
                rethrow(s, 0);
            }
            return len;
        }

        /** extractUntilBoundary (mode 0) / Left (1) / Right (2) of every from[i] into row i. */
        public int[] extractUntilBoundaryBatch(int[] from, char boundary, int mode, char[] rows, int rowLength, int offset) {
            int n = from.length;
            int[] len = new int[n];
            int[] status = new int[n];
            int[] aux = new int[n];
            nativeExtractBoundaryBatchMulti(handles, from, n, boundary, mode, rows, rowLength, offset, len, status, aux);
            for (int i = 0; i < n; i++) {
                rethrow(status[i], aux[i]);
            }
            return len;
        }

        /**
         * count() and locate() of one batch over K indexes of one long text (a Java int cannot address 2^31 chars,
         * FmIndex.java:131), every index replicated on the same devices: segments[s] = replicas of segment s, segmentBase[s] = its
         * first char in the text.  counts: sums over the segments; locations: n rows of maxMatches longs, segment 0's hits first
         * (fmx_count_locate_segments_multi).  Returns the number located per pattern.
         */
        public static int[] countLocateSegments(Replicas[] segments, long[] segmentBase, char[] chars, int[] offsets,
                int maxMatches, long[] counts, long[] locations) {
            int n = offsets.length - 1;
            int replicas = segments[0].handles.length;
            long[] flat = new long[replicas * segments.length]; // replica-major
            for (int r = 0; r < replicas; r++) {
                for (int s = 0; s < segments.length; s++) {
                    flat[r * segments.length + s] = segments[s].handles[r];
                }
            }
            int[] found = new int[n];
            int[] status = new int[n];
            nativeCountLocateSegmentsMulti(flat, replicas, segments.length, segmentBase, chars, offsets, n, maxMatches, counts,
                    locations, found, status);
            for (int s : status) {
                rethrow(s, 0);
            }
            return found;
        }

        @Override
        public void close() {
            if (handles != null) {
                for (long h : handles) {
                    nativeFree(h);
                }
                handles = null;
            }
        }
    }

    /**
     * FmIndex.write(...) of this index as libfmx emits it (fmx_save): with {@code framed} the ObjectOutputStream form
     * Serialization.writeToByteArray produces (SER:67-79), else the bare DataOutput stream.  Readable by
     * Serialization.readFromByteArray(FmIndex::read, bytes).
     */
    public byte[] toSerialized(boolean framed) {
        return nativeSave(handle, framed);
    }

    /**
     * FmIndex.write iterates a HashMap's keySet() (FM:956-960); toSerialized reproduces that order by replaying the map's puts.
     * {@code false}: a JVM would have turned one of the map's buckets into a tree bin (9 keys in one slot of a table of 64 slots
     * or more), whose iteration order is not modelled — the stream still loads with FmIndex.read (its reader does not depend on
     * the order) but may differ from index4j's own bytes inside that bucket (fmx.h: fmx_save_key_order_modelled).
     */
    public boolean isSerializedFormVerified() {
        return nativeSavedOrderModelled(handle);
    }

    @Override
    public void close() {
        if (handle != 0) {
            nativeFree(handle);
            handle = 0;
        }
    }

    private int boundary(int mode, int from, char[] destination, int offset, char boundary) {
        int[] len = new int[1];
        int[] status = new int[1];
        int[] aux = new int[1];
        nativeExtractBoundaryBatch(handle, new int[] {from}, 1, boundary, mode, destination, destination.length, offset,
                len, status, aux);
        rethrow(status[0], aux[0]);
        return len[0];
    }

    private static char[] slice(char[] pattern, int offset, int length) {
        if (offset == 0 && length == pattern.length) {
            return pattern;
        }
        char[] s = new char[Math.max(length, 0)];
        System.arraycopy(pattern, offset, s, 0, s.length); // throws like pattern[i] would (FM:456-457)
        return s;
    }

    /** status codes of include/fmx.h -> the reference's exception types and messages. */
    private static void rethrow(int status, int aux) {
        switch (status) {
            case 0:
                return;
            case 1:
                throw new RuntimeException("Text recovery not enabled at build time");
            case 2:
                throw new RuntimeException("Requested position less than 0");
            case 3:
                throw new RuntimeException("Stop position longer than index string");
            case 4:
                throw new RuntimeException("Supplied destination is not large enough");
            case 5:
                throw new RuntimeException("Requested position longer than index string");
            case 6:
                throw new IllegalArgumentException("Supplied destination for extraction has size zero");
            case 7:
                throw new IllegalArgumentException("Boundary does not exist");
            case 8:
                throw new RuntimeException(
                        "Extraction does not fit in the supplied destination. Currently extracted: " + aux);
            default:
                throw new ArrayIndexOutOfBoundsException();
        }
    }

    private static native long nativeLoad(byte[] serialized, int device) throws IOException;

    private static native byte[] nativeSave(long handle, boolean framed);

    private static native boolean nativeSavedOrderModelled(long handle);

    private static native long nativeBuild(char[] text, int sampleRate, boolean enableExtraction, int device,
            boolean buildOnGpu);

    private static native void nativeLocatePipeline(long handle, char[] chars, int[] offsets, int n, int maxMatches,
            int mode, char boundary, int rowLength, int[] locations, int[] found, char[] rows, int[] outLen, int[] status,
            int[] hitStatus, int[] hitAux);

    private static native void nativeCountSegments(long[] handles, char[] chars, int[] offsets, int n, long[] counts,
            int[] status);

    private static native void nativeFree(long handle);

    private static native long[] nativeReplicate(long handle, int[] devices);

    private static native int nativeDeviceOf(long handle);

    private static native void nativeCountBatchMulti(long[] handles, char[] chars, int[] offsets, int n, int[] counts,
            int[] status);

    private static native void nativeLocateBatchMulti(long[] handles, char[] chars, int[] offsets, int n, int maxMatches,
            int[] locations, int locCap, int[] found, int[] status);

    private static native void nativeExtractBatchMulti(long[] handles, int[] start, int[] stop, int n, char[] destination,
            int dstLen, int offset, int[] outLen, int[] status);

    private static native void nativeExtractBoundaryBatchMulti(long[] handles, int[] from, int n, char boundary, int mode,
            char[] destination, int dstLen, int offset, int[] outLen, int[] status, int[] aux);

    private static native void nativeCountLocateSegmentsMulti(long[] handles, int replicas, int segments, long[] segmentBase,
            char[] chars, int[] offsets, int n, int maxMatches, long[] counts, long[] locations, int[] found, int[] status);

    private static native int nativeInputLength(long handle);

    private static native int nativeAlphabetLength(long handle);

    private static native void nativeCountBatch(long handle, char[] chars, int[] offsets, int n, int[] counts, int[] status);

    private static native void nativeLocateBatch(long handle, char[] chars, int[] offsets, int n, int maxMatches,
            int[] locations, int locCap, int[] found, int[] status);

    private static native void nativeExtractBatch(long handle, int[] start, int[] stop, int n, char[] destination,
            int dstLen, int offset, int[] outLen, int[] status);

    private static native void nativeExtractBoundaryBatch(long handle, int[] from, int n, char boundary, int mode,
            char[] destination, int dstLen, int offset, int[] outLen, int[] status, int[] aux);
}
