// fmx_synth.cpp — deterministic synthetic workload generator (host tooling for bench.py and tests).
//
// BASELINE.md §2.3 / SURVEY.md §8(d): ASCII log text, '\n'-terminated lines shaped like the
// reference fixture's lines (HDFS_2k_multichar.log:1-4)
//     "<yymmdd> <hhmmss> <pid> <LEVEL> <component>: <message with ints, IPs, blk_ ids>"
// PRNG SplitMix64 (seed 42 for text), truncated to exactly n chars; patterns are substrings at
// next() % (n - m) (seed 43), mirroring the reference's JMH state which samples substrings of the
// indexed text (indices/src/jmh/java/com/dynatrace/fm/FmIndexThroughputState.java:76-83).
#include "../../include/fmx.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
};

const char *kLevels[] = {"INFO", "INFO", "INFO", "INFO", "INFO", "WARN", "WARN", "ERROR", "DEBUG"};
const char *kComponents[] = {"dfs.DataNode$PacketResponder", "dfs.FSNamesystem", "dfs.DataNode$DataXceiver",
                             "dfs.DataBlockScanner",         "dfs.DataNode",     "dfs.FSDataset",
                             "net.Server$Handler",           "sched.TaskTracker", "auth.TokenCache",
                             "io.CompactionQueue"};

void put(std::string &line, const char *s) { line += s; }
void put_uint(std::string &line, uint64_t v, int min_digits = 1) {
    char buf[32];
    snprintf(buf, sizeof buf, "%0*llu", min_digits, (unsigned long long)v);
    line += buf;
}
void put_ip(std::string &line, SplitMix64 &r) {
    put(line, "10.");
    put_uint(line, 250 + r.below(2));
    put(line, ".");
    put_uint(line, r.below(32));
    put(line, ".");
    put_uint(line, r.below(256));
}
void put_blk(std::string &line, SplitMix64 &r) {
    put(line, "blk_");
    if (r.below(2)) put(line, "-");
    put_uint(line, r.next() % 9000000000000000000ULL);
}

void message(std::string &line, SplitMix64 &r) {
    switch (r.below(10)) {
        case 0:
            put(line, "PacketResponder ");
            put_uint(line, r.below(3));
            put(line, " for block ");
            put_blk(line, r);
            put(line, " terminating");
            break;
        case 1:
            put(line, "BLOCK* NameSystem.addStoredBlock: blockMap updated: ");
            put_ip(line, r);
            put(line, ":50010 is added to ");
            put_blk(line, r);
            put(line, " size ");
            put_uint(line, r.below(67108865));
            break;
        case 2:
            put(line, "Receiving block ");
            put_blk(line, r);
            put(line, " src: /");
            put_ip(line, r);
            put(line, ":");
            put_uint(line, 30000 + r.below(30000));
            put(line, " dest: /");
            put_ip(line, r);
            put(line, ":50010");
            break;
        case 3:
            put(line, "Received block ");
            put_blk(line, r);
            put(line, " of size ");
            put_uint(line, r.below(67108865));
            put(line, " from /");
            put_ip(line, r);
            break;
        case 4:
            put(line, "Verification succeeded for ");
            put_blk(line, r);
            break;
        case 5:
            put(line, "BLOCK* NameSystem.allocateBlock: /user/root/rand/_temporary/_task_200811092030_0001_m_");
            put_uint(line, r.below(2000), 6);
            put(line, "_0/part-");
            put_uint(line, r.below(2000), 5);
            put(line, ". ");
            put_blk(line, r);
            break;
        case 6:
            put(line, "Deleting block ");
            put_blk(line, r);
            put(line, " file /mnt/hadoop/dfs/data/current/subdir");
            put_uint(line, r.below(64));
            put(line, "/");
            put_blk(line, r);
            break;
        case 7:
            put(line, "request id=");
            put_uint(line, r.next() % 100000000ULL);
            put(line, " user=svc");
            put_uint(line, r.below(40));
            put(line, " latency_ms=");
            put_uint(line, r.below(5000));
            put(line, " status=");
            put_uint(line, (r.below(20) == 0) ? 500 + r.below(4) : 200);
            break;
        case 8:
            put(line, "heartbeat from /");
            put_ip(line, r);
            put(line, " load=");
            put_uint(line, r.below(100));
            put(line, "% queue=");
            put_uint(line, r.below(512));
            put(line, " [ok]");
            break;
        default:
            put(line, "Served block ");
            put_blk(line, r);
            put(line, " to /");
            put_ip(line, r);
            break;
    }
}
}  // namespace

// One log line (ASCII, '\n'-terminated) into `line`.
static void log_line(std::string &line, SplitMix64 &r, uint32_t &sec) {
    line.clear();
    sec += r.below(3);
    uint32_t day = 9 + sec / 86400, s = sec % 86400 + 20 * 3600 + 35 * 60;
    if (s >= 86400) {
        s -= 86400;
        ++day;
    }
    put(line, "0811");
    put_uint(line, day % 100, 2);
    put(line, " ");
    put_uint(line, s / 3600, 2);
    put_uint(line, (s / 60) % 60, 2);
    put_uint(line, s % 60, 2);
    put(line, " ");
    put_uint(line, 1 + r.below(4000));
    put(line, " ");
    put(line, kLevels[r.below(9)]);
    put(line, " ");
    put(line, kComponents[r.below(10)]);
    put(line, ": ");
    message(line, r);
    put(line, "\n");
}

extern "C" int fmx_synth_log(uint64_t seed, int32_t n, uint16_t *out) {
    if (n < 0 || !out) return FMX_E_ARG;
    SplitMix64 r(seed);
    int64_t pos = 0;
    uint32_t sec = 0;  // seconds since 2008-11-09 20:35:00
    std::string line;
    while (pos < n) {
        log_line(line, r, sec);
        for (size_t i = 0; i < line.size() && pos < n; ++i) out[pos++] = (uint16_t)(unsigned char)line[i];
    }
    return FMX_OK;
}

// The same log with a LARGE alphabet, shaped like the reference's own fixture (HDFS_2k_multichar.log: ASCII log lines
// with runs of 2-9 consecutive multi-byte characters dropped in at word boundaries, ~6 % of the characters) and like
// the data set its published numbers are quoted on (loghub Android.log, "> 1,000 distinct symbols", README.md:291-292).
// `symbols` = distinct characters wanted (the ASCII part brings ~70; the rest comes from a pool of katakana, Thai and
// CJK code points, all inside the BMP like the fixture's).  A run = consecutive pool entries from a start drawn with a
// quadratic skew, so that a few symbols are frequent and most are rare, as in real logs.
extern "C" int fmx_synth_log_multichar(uint64_t seed, int32_t n, int32_t symbols, uint16_t *out) {
    if (n < 0 || !out || symbols < 0 || symbols > 30000) return FMX_E_ARG;
    const int32_t extra = symbols > 70 ? symbols - 70 : 0;
    std::vector<uint16_t> pool((size_t)extra);
    for (int32_t i = 0; i < extra; ++i) {
        if (i < 86)
            pool[(size_t)i] = (uint16_t)(0x30A1 + i);  // katakana
        else if (i < 86 + 58)
            pool[(size_t)i] = (uint16_t)(0x0E01 + (i - 86));  // Thai
        else
            pool[(size_t)i] = (uint16_t)(0x4E00 + (uint32_t)(i - 144) * 20000u / (uint32_t)(extra > 144 ? extra - 144 : 1));  // CJK, spread
    }
    SplitMix64 r(seed);
    SplitMix64 r2(seed ^ 0x5bd1e995u);  // the insertions draw from their own stream: the ASCII part equals fmx_synth_log's lines
    int64_t pos = 0;
    uint32_t sec = 0;
    std::string line;
    while (pos < n) {
        log_line(line, r, sec);
        for (size_t i = 0; i < line.size() && pos < n; ++i) {
            out[pos++] = (uint16_t)(unsigned char)line[i];
            if (line[i] == ' ' && extra > 0 && r2.below(8) == 0) {
                const uint32_t len = 2 + r2.below(8);
                const double u = (double)(r2.next() >> 11) / 9007199254740992.0;
                uint32_t start = (uint32_t)(u * u * (double)extra);
                if (start + len > (uint32_t)extra) start = (uint32_t)extra > len ? (uint32_t)extra - len : 0;
                for (uint32_t k = 0; k < len && start + k < (uint32_t)extra && pos < n; ++k) out[pos++] = pool[start + k];
                if (pos < n) out[pos++] = ' ';
            }
        }
    }
    return FMX_OK;
}

// patterns: `count` substrings of length m at next() % (n - m); written back to back, pat_off[i] = i*m
extern "C" int fmx_synth_patterns(uint64_t seed, const uint16_t *text, int32_t n, int32_t m, int32_t count,
                                  uint16_t *pat, int32_t *pat_off, int32_t *positions /*nullable*/) {
    if (!text || !pat || !pat_off || n <= m || m <= 0 || count < 0) return FMX_E_ARG;
    SplitMix64 r(seed);
    for (int32_t i = 0; i < count; ++i) {
        int32_t p = (int32_t)(r.next() % (uint64_t)(n - m));
        if (positions) positions[i] = p;
        memcpy(pat + (size_t)i * m, text + p, sizeof(uint16_t) * (size_t)m);
        pat_off[i] = i * m;
    }
    pat_off[count] = count * m;
    return FMX_OK;
}
