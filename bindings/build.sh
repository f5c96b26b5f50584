#!/bin/bash
# bindings/build.sh — build and run the JDK-side verification kit (NOT runnable in the build container: no JDK).
# Needs: JAVA_HOME (JDK 17+), INDEX4J_JAR (the reference's jar, e.g. indices/build/libs/indices-*.jar),
#        JUNIT_JAR (junit-platform-console-standalone-1.x.jar), ROCm + an MI355X.
# It builds libfmx.so, compiles bindings/jni/fmx_jni.c -> libfmx_jni.so and GpuFmIndex + GpuFmIndexParityTest, and
# runs the tests: every query through the shim against com.dynatrace.fm.FmIndex itself, and the serialized
# streams byte for byte.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
: "${JAVA_HOME:?set JAVA_HOME to a JDK 17+}" "${INDEX4J_JAR:?set INDEX4J_JAR}" "${JUNIT_JAR:?set JUNIT_JAR}"
OUT=$ROOT/bindings/out
mkdir -p "$OUT/classes"
make -C "$ROOT/index4j_amd/csrc" -j
cc -O2 -shared -fPIC -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -I"$ROOT/include" \
   "$ROOT/bindings/jni/fmx_jni.c" -L"$ROOT/index4j_amd" -lfmx -Wl,-rpath,"$ROOT/index4j_amd" -o "$OUT/libfmx_jni.so"
"$JAVA_HOME/bin/javac" -cp "$INDEX4J_JAR:$JUNIT_JAR" -d "$OUT/classes" \
   "$ROOT/bindings/java/com/dynatrace/fm/gpu/GpuFmIndex.java" \
   "$ROOT/bindings/java/test/com/dynatrace/fm/gpu/GpuFmIndexParityTest.java"
cd "$ROOT"
"$JAVA_HOME/bin/java" -Djava.library.path="$OUT" -Dfmx.fixture="$ROOT/tests/golden/HDFS_2k_multichar.log" \
   -jar "$JUNIT_JAR" execute -cp "$OUT/classes:$INDEX4J_JAR" --select-class com.dynatrace.fm.gpu.GpuFmIndexParityTest
# the JVM CPU baseline bench.py looks for:  INDEX4J_JAR=$INDEX4J_JAR python bench.py
