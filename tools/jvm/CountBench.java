/*
 * CountBench.java — index4j's own count() timed on the host cores, for bench.py's cpu_baseline leg
 * ("index4j JVM"; cf. countBenchmark, indices/src/jmh/java/com/dynatrace/fm/FmIndexThroughputBenchmark.java:191-199).
 *
 * Runs only where a JDK (11+: single-file source launch) and an index4j jar exist; bench.py probes for both:
 *   java -cp $INDEX4J_JAR tools/jvm/CountBench.java <index.ser> <patterns.bin> <threads>
 *     index.ser     what Serialization.writeToByteArray(FmIndex::write, index) produces (here: fmx_save, framed)
 *     patterns.bin  int32 n, int32 m (little endian), then n*m UTF-16 code units (little endian)
 * 3 warm-up + 5 timed passes over all patterns; `threads` workers take contiguous slices (FmIndex is @ThreadSafe).
 * Prints one JSON line.  NOT compiled or run in the build container (no JDK there).
 */
import com.dynatrace.fm.FmIndex;
import com.dynatrace.serialization.Serialization;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.file.Files;
import java.nio.file.Path;
import java.util.ArrayList;
import java.util.List;
import java.util.Locale;

public final class CountBench {
    public static void main(String[] args) throws Exception {
        final byte[] ser = Files.readAllBytes(Path.of(args[0]));
        final FmIndex index = Serialization.readFromByteArray(FmIndex::read, ser);
        final ByteBuffer pb = ByteBuffer.wrap(Files.readAllBytes(Path.of(args[1]))).order(ByteOrder.LITTLE_ENDIAN);
        final int n = pb.getInt();
        final int m = pb.getInt();
        final char[][] patterns = new char[n][m];
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < m; j++) {
                patterns[i][j] = (char) (pb.getShort() & 0xFFFF);
            }
        }
        final int threads = Math.max(1, Integer.parseInt(args[2]));
        final long[] sums = new long[threads];
        double best = Double.MAX_VALUE;
        double total = 0;
        final int warm = 3;
        final int timed = 5;
        for (int pass = 0; pass < warm + timed; pass++) {
            final long t0 = System.nanoTime();
            final List<Thread> workers = new ArrayList<>();
            for (int t = 0; t < threads; t++) {
                final int id = t;
                final int lo = (int) ((long) n * t / threads);
                final int hi = (int) ((long) n * (t + 1) / threads);
                final Thread w = new Thread(() -> {
                    long s = 0;
                    for (int i = lo; i < hi; i++) {
                        s += index.count(patterns[i]);
                    }
                    sums[id] = s;
                });
                workers.add(w);
                w.start();
            }
            for (Thread w : workers) {
                w.join();
            }
            final double sec = (System.nanoTime() - t0) * 1e-9;
            if (pass >= warm) {
                best = Math.min(best, sec);
                total += sec;
            }
        }
        long checksum = 0;
        for (long s : sums) {
            checksum += s;
        }
        System.out.println(String.format(Locale.ROOT,
                "{\"kind\": \"index4j-jvm\", \"threads\": %d, \"patterns\": %d, \"value\": %.1f, \"unit\": \"patterns/s\", "
                        + "\"best_value\": %.1f, \"count_checksum\": %d, \"java\": \"%s\"}",
                threads, n, n * (double) timed / total, n / best, checksum, System.getProperty("java.version")));
    }
}
