/*
 * GpuFmIndexParityTest — JDK-side verification kit (SURVEY §8 f1 / f2).  SOURCE ONLY: the build image has no JDK,
 * so this has never been compiled or run; bindings/build.sh builds and runs it on a host that has a JDK 17+,
 * index4j's jar, JUnit 5's console launcher and an MI355X.
 *
 * What it pins, against the REAL reference (com.dynatrace.fm.FmIndex), not against this repository's oracle:
 *   (f1) serialized bytes — FmIndex.write -> fmx_load -> fmx_save(framed) must reproduce the reference's stream byte
 *        for byte (ObjectOutputStream framing, HashMap key order FM:956-960, every field), and the native builders
 *        must emit what `new FmIndexBuilder()...build(text)` emits (Huffman tie-breaks WFBB:1684-1707, the size
 *        estimator WFBB:853-987);
 *   (f2) the shim — every query of FmIndexTest's kind through GpuFmIndex equals FmIndex on the same inputs:
 *        counts, located positions IN ORDER, extracted chars, whole destination arrays, exception types and messages.
 */
package com.dynatrace.fm.gpu;

import static org.junit.jupiter.api.Assertions.assertArrayEquals;
import static org.junit.jupiter.api.Assertions.assertEquals;
import static org.junit.jupiter.api.Assertions.assertThrows;
import static org.junit.jupiter.api.Assertions.assertTrue;

import com.dynatrace.fm.FmIndex;
import com.dynatrace.fm.FmIndexBuilder;
import com.dynatrace.serialization.Serialization;
import java.io.IOException;
import java.nio.charset.StandardCharsets;
import java.nio.file.Files;
import java.nio.file.Path;
import java.util.Arrays;
import java.util.Random;
import org.junit.jupiter.api.Test;
import org.junit.jupiter.params.ParameterizedTest;
import org.junit.jupiter.params.provider.ValueSource;

class GpuFmIndexParityTest {

    /** the reference's own fixture (indices/src/test/resources/HDFS_2k_multichar.log; a copy is tests/golden/) */
    private static final char[] TEXT = load();

    private static final int N_TESTS = 500;
    private static final char NUL = (char) 0;
    private static final char NOT_IN_TEXT = (char) 0xC774; // the Hangul syllable FmIndexTest uses as an absent boundary

    private static char[] load() {
        try {
            String p = System.getProperty("fmx.fixture", "tests/golden/HDFS_2k_multichar.log");
            return new String(Files.readAllBytes(Path.of(p)), StandardCharsets.UTF_8).toCharArray();
        } catch (IOException e) {
            throw new IllegalStateException(e);
        }
    }

    private static char[] withNul(String s) {
        return s.replace('#', NUL).toCharArray();
    }

    // ---- (f1) serialized layout -------------------------------------------------------------------------------

    @ParameterizedTest
    @ValueSource(ints = {1, 2, 4, 8, 16, 32, 64})
    void readerAndWriterReproduceTheReferenceStream(int sampleRate) throws IOException {
        for (boolean extract : new boolean[] {true, false}) {
            FmIndex ref = new FmIndexBuilder().setSampleRate(sampleRate).setEnableExtraction(extract).build(TEXT);
            byte[] expected = Serialization.writeToByteArray(FmIndex::write, ref); // SER:67-79 over FM:948-975
            try (GpuFmIndex gpu = GpuFmIndex.fromSerialized(expected, 0)) {
                // (byte identity is only claimed where the character map's HashMap order is the replayed one: no tree bins —
                // and THIS text's map has none, so the comparison below cannot be skipped silently: ADVICE r5)
                assertTrue(gpu.isSerializedFormVerified(), "the fixture's 763-key map makes no tree bin");
                if (gpu.isSerializedFormVerified()) {
                    assertArrayEquals(expected, gpu.toSerialized(true), "fmx_load -> fmx_save(framed)");
                }
            }
        }
    }

    @ParameterizedTest
    @ValueSource(ints = {1, 3, 32, 64})
    void nativeBuildersEmitTheReferenceStream(int sampleRate) throws IOException {
        char[][] texts = {
            TEXT,
            withNul("This is a long string#"),
            withNul("This #is a #long string#"),
            "a".toCharArray(),
            "zqzqzqzqzqzqzqzqzqzqzqzqzqzqzqzqzqzq".toCharArray()
        };
        for (char[] text : texts) {
            FmIndex ref = new FmIndexBuilder().setSampleRate(sampleRate).build(text);
            byte[] expected = Serialization.writeToByteArray(FmIndex::write, ref);
            try (GpuFmIndex host = GpuFmIndex.build(text, sampleRate, true, 0);
                    GpuFmIndex dev = GpuFmIndex.buildOnGpu(text, sampleRate, true, 0)) {
                assertArrayEquals(expected, host.toSerialized(true), "fmx_build");
                assertArrayEquals(expected, dev.toSerialized(true), "fmx_build_on_device");
            }
            // and the reference reads what libfmx wrote
            try (GpuFmIndex host = GpuFmIndex.build(text, sampleRate, true, 0)) {
                FmIndex back = Serialization.readFromByteArray(FmIndex::read, host.toSerialized(true));
                assertEquals(ref.hashCode(), back.hashCode());
                assertEquals(ref.getInputLength(), back.getInputLength());
            }
        }
    }

    // ---- (f2) the shim against the reference, query by query ---------------------------------------------------

    @ParameterizedTest
    @ValueSource(ints = {1, 2, 4, 8, 16, 32, 64})
    void countLocateExtractAgreeWithTheReference(int sampleRate) throws IOException {
        Random random = new Random(42);
        FmIndex ref = new FmIndexBuilder().setSampleRate(sampleRate).build(TEXT);
        try (GpuFmIndex gpu = GpuFmIndex.fromFmIndex(ref, 0)) {
            assertEquals(ref.getInputLength(), gpu.getInputLength()); // T-FM:564-578
            assertEquals(ref.getAlphabetLength(), gpu.getAlphabetLength());
            int[] a = new int[100_000];
            int[] b = new int[100_000];
            for (int i = 0; i < N_TESTS; i++) {
                int start = random.nextInt(TEXT.length - 32);
                char[] p = Arrays.copyOfRange(TEXT, start, start + 1 + random.nextInt(31));
                assertEquals(ref.count(p), gpu.count(p));
                if (p.length > 1) {
                    assertEquals(ref.count(p, 1, p.length - 1), gpu.count(p, 1, p.length - 1));
                }
                for (int max : new int[] {-1, 1, 16}) {
                    Arrays.fill(a, -7);
                    Arrays.fill(b, -7);
                    int na = ref.locate(p, 0, p.length, a, max);
                    int nb = gpu.locate(p, 0, p.length, b, max);
                    assertEquals(na, nb);
                    assertArrayEquals(a, b, "located positions, in the reference's (suffix-array) order");
                }
                int stop = Math.min(TEXT.length, start + 1 + random.nextInt(200));
                char[] da = new char[256];
                char[] db = new char[256];
                assertEquals(ref.extract(start, stop, da, 3), gpu.extract(start, stop, db, 3));
                assertArrayEquals(da, db);
                for (int cap : new int[] {2048, 40}) {
                    compareBoundary(ref, gpu, start, cap, 0);
                    compareBoundary(ref, gpu, start, cap, 5);
                }
            }
            assertEquals(0, gpu.count("baaazz".toCharArray()));
            assertEquals(ref.count(new char[] {NUL}), gpu.count(new char[] {NUL}));
        }
    }

    private interface Call {
        int run(char[] dest);
    }

    private static void same(Call ref, Call gpu, int cap) {
        char[] da = new char[cap];
        char[] db = new char[cap];
        RuntimeException ea = null;
        RuntimeException eb = null;
        int ra = 0;
        int rb = 0;
        try {
            ra = ref.run(da);
        } catch (RuntimeException e) {
            ea = e;
        }
        try {
            rb = gpu.run(db);
        } catch (RuntimeException e) {
            eb = e;
        }
        if (ea != null || eb != null) {
            assertEquals(ea == null ? null : ea.getClass(), eb == null ? null : eb.getClass());
            assertEquals(ea.getMessage(), eb.getMessage()); // incl. "Currently extracted: N" (T-FM:447-474)
        } else {
            assertEquals(ra, rb);
        }
        assertArrayEquals(da, db, "whole destination array (the left part is written from the top, FM:655)");
    }

    private static void compareBoundary(FmIndex ref, GpuFmIndex gpu, int from, int cap, int offset) {
        same(d -> ref.extractUntilBoundary(from, d, offset, '\n'), d -> gpu.extractUntilBoundary(from, d, offset, '\n'), cap);
        same(
                d -> ref.extractUntilBoundaryLeft(from, d, offset, '\n'),
                d -> gpu.extractUntilBoundaryLeft(from, d, offset, '\n'),
                cap);
        same(
                d -> ref.extractUntilBoundaryRight(from, d, offset, '\n'),
                d -> gpu.extractUntilBoundaryRight(from, d, offset, '\n'),
                cap);
    }

    @Test
    void exceptionsCarryTheReferenceTypesAndMessages() throws IOException {
        FmIndex noExtract = new FmIndexBuilder().setEnableExtraction(false).build(TEXT);
        FmIndex ref = new FmIndexBuilder().build(TEXT);
        final int len = TEXT.length;
        try (GpuFmIndex g0 = GpuFmIndex.fromFmIndex(noExtract, 0);
                GpuFmIndex gpu = GpuFmIndex.fromFmIndex(ref, 0)) {
            same(d -> noExtract.extract(5, 10, d, 0), d -> g0.extract(5, 10, d, 0), 50); // FM:566-568
            same(d -> ref.extract(-5, 100, d, 0), d -> gpu.extract(-5, 100, d, 0), 50); // FM:570-572
            same(d -> ref.extract(len + 1, len + 51, d, 0), d -> gpu.extract(len + 1, len + 51, d, 0), 50); // FM:574-576
            same(d -> ref.extract(50, 100, d, 0), d -> gpu.extract(50, 100, d, 0), 10); // FM:591-593
            same(
                    d -> ref.extractUntilBoundary(len + 1, d, 0, '\n'),
                    d -> gpu.extractUntilBoundary(len + 1, d, 0, '\n'),
                    50); // FM:619-621
            same(
                    d -> ref.extractUntilBoundary(50, d, 0, NOT_IN_TEXT),
                    d -> gpu.extractUntilBoundary(50, d, 0, NOT_IN_TEXT),
                    50); // FM:659-661
            same(d -> ref.extractUntilBoundary(50, d, 0, '\n'), d -> gpu.extractUntilBoundary(50, d, 0, '\n'), 0); // FM:623-625
            same(d -> ref.extractUntilBoundary(50, d, 0, '\n'), d -> gpu.extractUntilBoundary(50, d, 0, '\n'), 10); // ": 13"
            same(d -> ref.extractUntilBoundaryLeft(50, d, 0, '\n'), d -> gpu.extractUntilBoundaryLeft(50, d, 0, '\n'), 10); // ": 10"
            same(d -> ref.extractUntilBoundaryRight(50, d, 0, '\n'), d -> gpu.extractUntilBoundaryRight(50, d, 0, '\n'), 10); // ": 11"
            assertThrows(ArrayIndexOutOfBoundsException.class, () -> gpu.count(new char[0]));
            assertThrows(ArrayIndexOutOfBoundsException.class, () -> ref.count(new char[0])); // FM:456-457
        }
    }

    @Test
    void batchCallsEqualTheScalarOnes() throws IOException {
        Random random = new Random(7);
        FmIndex ref = new FmIndexBuilder().build(TEXT);
        try (GpuFmIndex gpu = GpuFmIndex.fromFmIndex(ref, 0)) {
            int n = 20_000; // large enough for the planned (suffix-ordered) path
            StringBuilder chars = new StringBuilder();
            int[] offsets = new int[n + 1];
            for (int i = 0; i < n; i++) {
                int start = random.nextInt(TEXT.length - 16);
                chars.append(TEXT, start, 1 + random.nextInt(15));
                offsets[i + 1] = chars.length();
            }
            char[] all = chars.toString().toCharArray();
            int[] counts = gpu.countBatch(all, offsets);
            for (int i = 0; i < n; i += 37) {
                assertEquals(ref.count(all, offsets[i], offsets[i + 1] - offsets[i]), counts[i]);
            }
            // the same batch sharded over replicas of the index (here: three replicas on device 0; on a node: one per GPU):
            // FmIndex is immutable and @ThreadSafe (FmIndex.java:82), every shard stores into its slice of the arrays
            try (GpuFmIndex.Replicas replicas = gpu.replicate(new int[] {0, 0, 0})) {
                assertEquals(3, replicas.size());
                assertArrayEquals(counts, replicas.countBatch(all, offsets));
                int[] rows = new int[n * 4];
                int[] rowsOne = new int[n * 4];
                assertArrayEquals(gpu.locateBatch(all, offsets, 4, rowsOne), replicas.locateBatch(all, offsets, 4, rows));
                assertArrayEquals(rowsOne, rows);
            }
        }
    }
}
