// test_host_mirror.cpp — the C++ host mirror (include/index4j/FmIndex.hpp) exercised the way the
// reference's JUnit tests exercise FmIndex (FmIndexTest.java).  Mode "host": builder / serialization /
// accessors only (no GPU).  Mode "gpu": the query tests.  Exit code 0 = all assertions held.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>

#include "../../include/index4j/FmIndex.hpp"

using index4j::FmIndex;
using index4j::FmIndexBuilder;
using index4j::FmIndexReplicas;

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                    \
        }                                                                  \
    } while (0)
template <typename E, typename F>
static bool throws_with(F f, const char *msg) {
    try {
        f();
    } catch (const E &e) {
        return std::string(e.what()) == msg;
    } catch (...) {
        return false;
    }
    return false;
}

static std::u16string utf8_to_u16(const std::string &s) {  // BMP only (the fixture has no astral chars)
    std::u16string out;
    for (size_t i = 0; i < s.size();) {
        unsigned c = (unsigned char)s[i];
        if (c < 0x80) {
            out.push_back((char16_t)c);
            i += 1;
        } else if ((c >> 5) == 6) {
            out.push_back((char16_t)(((c & 0x1f) << 6) | (s[i + 1] & 0x3f)));
            i += 2;
        } else {
            out.push_back((char16_t)(((c & 0x0f) << 12) | ((s[i + 1] & 0x3f) << 6) | (s[i + 2] & 0x3f)));
            i += 3;
        }
    }
    return out;
}

static size_t occurrences(const std::u16string &text, const std::u16string &pat) {
    size_t n = 0, pos = text.find(pat);
    while (pos != std::u16string::npos) {
        ++n;
        pos = text.find(pat, pos + 1);
    }
    return n;
}

int main(int argc, char **argv) {
    const bool gpu = argc > 1 && !std::strcmp(argv[1], "gpu");
    const char *fixture = argc > 2 ? argv[2] : "tests/golden/HDFS_2k_multichar.log";
    std::ifstream in(fixture, std::ios::binary);
    const std::string raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    const std::u16string hdfs = utf8_to_u16(raw);
    CHECK(hdfs.size() == 315118);
    const int dev = gpu ? 0 : -1;

    // shouldTestConvenienceMethods (T-FM:564-578)
    FmIndex fmi = FmIndexBuilder().setDevice(dev).build(hdfs);
    CHECK(fmi.getInputLength() == (int)hdfs.size() + 1);
    CHECK(fmi.getAlphabetLength() == 763);
    CHECK(fmi.toString() == "FMIndex-sampleRate:32-extract:true");
    // shouldSerialize... (T-FM:219-242): write -> read -> identical bytes
    {
        const auto framed = fmi.write(true), plain = fmi.write(false);
        CHECK(framed.size() > plain.size() && framed[0] == 0xAC && framed[1] == 0xED);
        FmIndex again = FmIndex::read(framed, -1);
        CHECK(again.write(false) == plain);
        auto bad = plain;
        bad[0] = 7;
        CHECK((throws_with<index4j::IoError>([&] { FmIndex::read(bad, -1); },
                                             "Incompatible serial versions! Expected version 0 but was 7.")));  // util/UtilTest.java:36-49
    }
    // shouldExceedCharsetLimit (T-FM:165-179)
    {
        std::u16string many;
        for (int i = 0; i < 32768; ++i) many.push_back((char16_t)i);
        CHECK((throws_with<std::invalid_argument>([&] { FmIndex f(many, 32, true, -1); },
                                                  "Input has more than 32767 different symbols")));
    }
    // shouldComplainFromTooBigChar (T-FM:146-163)
    {
        const uint8_t p[6] = {'a', 0xF7, 0xB8, 0xB8, 0xB8, 'c'};
        char16_t d[3];
        CHECK((throws_with<std::runtime_error>([&] { FmIndex::convertBytePatternToCharPattern(p, 0, 6, d); },
                                               "Found a character that exceeds (32767): it was 2068024")));
    }
    if (!gpu) {
        // no CPU query path: a query on a host-only index must fail loudly
        bool threw = false;
        try {
            fmi.count(u"INFO");
        } catch (const std::runtime_error &) {
            threw = true;
        }
        CHECK(threw);
        std::printf("host mirror (host mode): %d failure(s)\n", failures);
        return failures ? 1 : 0;
    }

    // shouldCount / shouldCountPartialString / shouldCountSlicedString (T-FM:43-102)
    {
        const std::u16string text(u"This is a long string\0", 22);
        FmIndex f = FmIndexBuilder().setEnableExtraction(false).build(text);
        CHECK(f.count(u"is") == 2);
        CHECK(f.count(u"is a long", 0, 2) == 2);
        CHECK(f.count(u"is a long", 2, 1) == 4);
        CHECK(f.count(u"baaa") == 0);
        std::vector<int32_t> none(1);
        CHECK(f.locate(u"does not exist here", none) == 0);
        std::u16string d(50, u'\0');
        CHECK((throws_with<std::runtime_error>([&] { f.extract(5, 10, d, 0); }, "Text recovery not enabled at build time")));
    }
    // shouldLocateMaxNumberOfMatches (T-FM:195-200)
    {
        std::vector<int32_t> locs(100);
        CHECK(fmi.locate(u"INFO", 0, 4, locs, 100) == 100);
        for (int32_t p : locs) CHECK(hdfs.compare((size_t)p, 4, u"INFO") == 0);
        CHECK(fmi.count(u"INFO") == (int)occurrences(hdfs, u"INFO"));
    }
    // shouldTestOutOfBoundsExtraction / shouldAttemptExtraction... (T-FM:284-348, 402-475)
    {
        std::u16string d50(50, u'\0'), d10(10, u'\0'), d0;
        CHECK((throws_with<std::runtime_error>([&] { fmi.extract(-5, 100, d50, 0); }, "Requested position less than 0")));
        CHECK((throws_with<std::runtime_error>([&] { fmi.extract((int)hdfs.size() + 1, (int)hdfs.size() + 51, d50, 0); },
                                               "Stop position longer than index string")));
        CHECK((throws_with<std::runtime_error>([&] { fmi.extract(50, 100, d10, 0); },
                                               "Supplied destination is not large enough")));
        CHECK((throws_with<std::runtime_error>([&] { fmi.extractUntilBoundary((int)hdfs.size() + 1, d50, 0, u'\n'); },
                                               "Requested position longer than index string")));
        CHECK((throws_with<std::invalid_argument>([&] { fmi.extractUntilBoundary(50, d50, 0, u'\xC774'); },
                                                  "Boundary does not exist")));
        CHECK((throws_with<std::invalid_argument>([&] { fmi.extractUntilBoundary(50, d0, 0, u'\n'); },
                                                  "Supplied destination for extraction has size zero")));
        CHECK((throws_with<std::runtime_error>(
            [&] { fmi.extractUntilBoundary(50, d10, 0, u'\n'); },
            "Extraction does not fit in the supplied destination. Currently extracted: 13")));
        CHECK((throws_with<std::runtime_error>(
            [&] { fmi.extractUntilBoundaryLeft(50, d10, 0, u'\n'); },
            "Extraction does not fit in the supplied destination. Currently extracted: 10")));
        CHECK((throws_with<std::runtime_error>(
            [&] { fmi.extractUntilBoundaryRight(50, d10, 0, u'\n'); },
            "Extraction does not fit in the supplied destination. Currently extracted: 11")));
    }
    // shouldExtractTwoFirstLogLines (T-FM:477-496)
    {
        std::u16string dest(300, u'\0');
        int n = fmi.extractUntilBoundary(5, dest, 0, u'\n');
        dest[n++] = u'\n';
        n += fmi.extractUntilBoundary(n + 2, dest, n, u'\n');
        CHECK(dest.substr(0, (size_t)n) == hdfs.substr(0, (size_t)n));
        CHECK(hdfs[(size_t)n] == u'\n');
    }
    // batch surface
    {
        const auto counts = fmi.countBatch({u"INFO", u"WARN", u"blk_", u"zzzzzz"});
        CHECK(counts[0] == 1920 && counts[1] == (int)occurrences(hdfs, u"WARN") &&
              counts[2] == (int)occurrences(hdfs, u"blk_") && counts[3] == 0);
    }
    // the same batches sharded over replicas of the index (fmx_replicate + fmx_*_multi: three replicas on device 0)
    {
        FmIndexReplicas reps(fmi, {0, 0, 0});
        CHECK(reps.size() == 3 && reps.deviceOf(2) == 0);
        const std::vector<std::u16string> pats = {u"INFO", u"WARN", u"blk_", u"zzzzzz", u"dfs.DataNode", u"e", u"src: /10."};
        CHECK(reps.countBatch(pats) == fmi.countBatch(pats));
        std::vector<int32_t> one, many;
        CHECK(reps.locateBatch(pats, 6, many) == fmi.locateBatch(pats, 6, one));
        CHECK(one == many);
    }
    // the index built with its suffix-array stage on the GPU is the same bytes
    {
        FmIndex onGpu = FmIndexBuilder().setDevice(-1).setBuildDevice(0).build(hdfs);
        CHECK(onGpu.write(false) == fmi.write(false));
    }
    // locate -> extract / extractUntilBoundary pipelines (locateAndExtractBenchmark, J-FM:231-249)
    {
        const std::vector<std::u16string> pats = {u"INFO", u"blk_", u"zzzzzz"};
        const int mm = 5, xl = 12;
        const auto r = fmi.locateExtractBatch(pats, mm, xl);
        CHECK(r.found[0] == mm && r.found[1] == mm && r.found[2] == 0);
        for (size_t i = 0; i < pats.size(); ++i)
            for (int k = 0; k < mm; ++k) {
                const size_t q = i * mm + (size_t)k;
                if (k >= r.found[i]) {
                    CHECK(r.locations[q] == -1 && r.outLen[q] == -1);
                    continue;
                }
                CHECK(r.hitStatus[q] == 0 && r.outLen[q] == xl);
                CHECK(r.rows[q] == hdfs.substr((size_t)r.locations[q], (size_t)xl));
                CHECK(r.rows[q].compare(0, pats[i].size(), pats[i]) == 0);
            }
        const auto lines = fmi.locateLinesBatch({u"WARN"}, 3, u'\n', 600);
        CHECK(lines.found[0] == 3);
        for (int k = 0; k < 3; ++k) {
            const size_t loc = (size_t)lines.locations[(size_t)k];
            const size_t lo = hdfs.rfind(u'\n', loc) + 1, hi = hdfs.find(u'\n', loc);
            CHECK(lines.hitStatus[(size_t)k] == 0 && (size_t)lines.outLen[(size_t)k] == hi - lo);
            CHECK(lines.rows[(size_t)k].substr(0, hi - lo) == hdfs.substr(lo, hi - lo));
        }
    }
    // one text as three segment indexes cut at line ends
    {
        index4j::SegmentedFmIndex seg;
        size_t a = 0;
        for (int piece = 0; piece < 3; ++piece) {
            size_t b = piece == 2 ? hdfs.size() : hdfs.find(u'\n', (piece + 1) * hdfs.size() / 3) + 1;
            seg.add(FmIndexBuilder().setSampleRate(16).build(hdfs.substr(a, b - a)), (int64_t)a);
            a = b;
        }
        const std::vector<std::u16string> pats = {u"INFO", u"WARN", u"zzzzzz"};
        const auto counts = seg.countBatch(pats);
        CHECK(counts[0] == 1920 && counts[1] == (int64_t)occurrences(hdfs, u"WARN") && counts[2] == 0);
        std::vector<int64_t> locs;
        const auto found = seg.locateBatch(pats, 4, locs);
        CHECK(found[0] == 4 && found[2] == 0);
        for (int k = 0; k < found[1]; ++k) CHECK(hdfs.compare((size_t)locs[4 + (size_t)k], 4, u"WARN") == 0);
    }
    std::printf("host mirror (gpu mode): %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
