// index4j/FmIndex.hpp — host-side C++ mirror of com.dynatrace.fm.FmIndex / FmIndexBuilder over the
// C ABI of libfmx.so (include/fmx.h).  Header-only.
//
// The reference's host language is Java; this image has no JDK, so the host layer above the C ABI is
// written in C++ with the reference's method names, argument meaning and error behaviour
// (fm/FmIndex.java:443-941, fm/FmIndexBuilder.java:21-62).  Java exceptions map to:
//   RuntimeException              -> std::runtime_error      (same message)
//   IllegalArgumentException      -> std::invalid_argument   (same message)
//   ArrayIndexOutOfBoundsException-> std::out_of_range
//   IOException                   -> index4j::IoError
// Java `char[]` is std::u16string / char16_t*.  Scalar calls are batches of one on the GPU; the
// *Batch methods are the intended production surface.  bindings/java holds the equivalent JNI shim.
#pragma once

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../fmx.h"

namespace index4j {

struct IoError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

namespace detail {
inline void check(int rc, const char *where) {
    if (rc == FMX_OK) return;
    const std::string msg = fmx_last_error() ? fmx_last_error() : "";
    if (rc == FMX_E_ALPHABET) throw std::invalid_argument("Input has more than 32767 different symbols");  // FM:423-426
    if (rc == FMX_E_VERSION) throw IoError(msg);                                                           // SER:46-56
    if (rc == FMX_E_FORMAT) throw IoError(std::string(where) + ": " + msg);
    throw std::runtime_error(std::string(where) + " failed (" + std::to_string(rc) + "): " + msg);
}
// re-throw the reference's exception for a per-query status code
inline void raise_for_status(int status, int aux = 0) {
    if (status == FMX_ST_OK) return;
    char buf[256];
    std::snprintf(buf, sizeof buf, fmx_status_message(status), aux);
    switch (fmx_status_kind(status)) {
        case 1: throw std::invalid_argument(buf);
        case 2: throw std::out_of_range(buf);
        default: throw std::runtime_error(buf);
    }
}
}  // namespace detail

class FmIndex {
public:
    // new FmIndex(char[] input, int sampleRate, boolean enableExtract)  FM:155-174
    FmIndex(const std::u16string &input, int sampleRate, bool enableExtract = true, int device = 0) {
        detail::check(fmx_build(reinterpret_cast<const uint16_t *>(input.data()), (int32_t)input.size(), sampleRate,
                                enableExtract ? 1 : 0, &h_),
                      "fmx_build");
        if (device >= 0) toDevice(device);
    }
    // FmIndex.read(ObjectInput) FM:983-1025 (raw DataOutput stream or ObjectOutputStream-framed, SER:89-100)
    static FmIndex read(const std::vector<uint8_t> &bytes, int device = 0) {
        fmx_index *h = nullptr;
        detail::check(fmx_load(bytes.data(), bytes.size(), &h), "fmx_load");
        FmIndex f(h);
        if (device >= 0) f.toDevice(device);
        return f;
    }
    FmIndex(FmIndex &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    FmIndex &operator=(FmIndex &&o) noexcept {
        if (this != &o) {
            fmx_free(h_);
            h_ = o.h_;
            o.h_ = nullptr;
        }
        return *this;
    }
    FmIndex(const FmIndex &) = delete;
    FmIndex &operator=(const FmIndex &) = delete;
    ~FmIndex() { fmx_free(h_); }

    void toDevice(int device) { detail::check(fmx_to_device(h_, device), "fmx_to_device"); }
    fmx_index *handle() const { return h_; }

    // FmIndex.write(ObjectOutput) FM:948-975; framed = through Serialization.writeToByteArray SER:67-79
    std::vector<uint8_t> write(bool framed = true) const {
        uint8_t *buf = nullptr;
        size_t len = 0;
        detail::check(fmx_save(h_, framed ? 1 : 0, &buf, &len), "fmx_save");
        std::vector<uint8_t> out(buf, buf + len);
        fmx_free_buffer(buf);
        return out;
    }

    int getInputLength() const { return fmx_input_length(h_); }        // FM:929
    int getAlphabetLength() const { return fmx_alphabet_length(h_); }  // FM:939
    std::string toString() const {                                     // FM:1044-1046
        return "FMIndex-sampleRate:" + std::to_string(fmx_sample_rate(h_)) +
               "-extract:" + (fmx_extract_enabled(h_) ? "true" : "false");
    }

    // ---- batched queries (one GPU launch per call) ----
    std::vector<int32_t> countBatch(const std::vector<std::u16string> &patterns) const {
        std::vector<uint16_t> chars;
        std::vector<int32_t> off;
        pack(patterns, chars, off);
        std::vector<int32_t> counts(patterns.size()), status(patterns.size());
        detail::check(fmx_count_batch(h_, chars.data(), off.data(), (int32_t)patterns.size(), counts.data(), nullptr,
                                      status.data()),
                      "fmx_count_batch");
        for (int s : status) detail::raise_for_status(s);
        return counts;
    }
    // returns found[i]; locations is patterns.size() rows of maxMatches ints
    std::vector<int32_t> locateBatch(const std::vector<std::u16string> &patterns, int maxMatches,
                                     std::vector<int32_t> &locations) const {
        std::vector<uint16_t> chars;
        std::vector<int32_t> off;
        pack(patterns, chars, off);
        const int32_t n = (int32_t)patterns.size();
        locations.assign((size_t)n * (size_t)maxMatches, 0);
        std::vector<int32_t> found(patterns.size()), status(patterns.size());
        detail::check(fmx_locate_batch(h_, chars.data(), off.data(), n, maxMatches, locations.data(), maxMatches,
                                       found.data(), nullptr, status.data()),
                      "fmx_locate_batch");
        for (int s : status) detail::raise_for_status(s);
        return found;
    }

    // ---- scalar API, as in the reference ----
    int count(const std::u16string &pattern) const { return count(pattern, 0, (int)pattern.size()); }  // FM:443-445
    int count(const std::u16string &pattern, int offset, int length) const {                            // FM:455-474
        if (length <= 0 || offset < 0 || (size_t)(offset + length) > pattern.size())
            throw std::out_of_range("ArrayIndexOutOfBoundsException");
        const int32_t off[2] = {0, length};
        int32_t c = 0, st = 0;
        detail::check(fmx_count_batch(h_, reinterpret_cast<const uint16_t *>(pattern.data()) + offset, off, 1, &c,
                                      nullptr, &st),
                      "fmx_count_batch");
        detail::raise_for_status(st);
        return c;
    }
    int locate(const std::u16string &pattern, std::vector<int32_t> &locations) const {  // FM:487-489
        return locate(pattern, 0, (int)pattern.size(), locations, -1);
    }
    int locate(const std::u16string &pattern, int offset, int length, std::vector<int32_t> &locations,
               int maxMatches) const {  // FM:504-552: `locations` is the caller's pre-sized array
        if (length <= 0 || offset < 0 || (size_t)(offset + length) > pattern.size())
            throw std::out_of_range("ArrayIndexOutOfBoundsException");
        const int32_t off[2] = {0, length};
        int32_t found = 0, st = 0;
        detail::check(fmx_locate_batch(h_, reinterpret_cast<const uint16_t *>(pattern.data()) + offset, off, 1,
                                       maxMatches, locations.data(), (int32_t)locations.size(), &found, nullptr, &st),
                      "fmx_locate_batch");
        detail::raise_for_status(st);
        return found;
    }
    int extract(int start, int stop, std::u16string &destination, int offset) const {  // FM:564-608
        int32_t len = 0, st = 0;
        detail::check(fmx_extract_batch(h_, &start, &stop, 1, reinterpret_cast<uint16_t *>(&destination[0]),
                                        (int32_t)destination.size(), offset, &len, nullptr, &st),
                      "fmx_extract_batch");
        detail::raise_for_status(st);
        return len;
    }
    int extractUntilBoundary(int from, std::u16string &destination, int offset, char16_t boundary) const {  // FM:640-759
        return boundaryCall(0, from, destination, offset, boundary);
    }
    int extractUntilBoundaryLeft(int from, std::u16string &destination, int offset, char16_t boundary) const {  // FM:772-831
        return boundaryCall(1, from, destination, offset, boundary);
    }
    int extractUntilBoundaryRight(int from, std::u16string &destination, int offset, char16_t boundary) const {  // FM:844-922
        return boundaryCall(2, from, destination, offset, boundary);
    }

    // FM:239-298
    static int convertBytePatternToCharPattern(const uint8_t *pattern, int offset, int length, char16_t *destination) {
        int32_t bad = 0;
        const int n = fmx_convert_byte_pattern(pattern, offset, length, reinterpret_cast<uint16_t *>(destination), &bad);
        if (n < 0)
            throw std::runtime_error("Found a character that exceeds (32767): it was " + std::to_string(bad));
        return n;
    }

private:
    explicit FmIndex(fmx_index *h) : h_(h) {}
    static void pack(const std::vector<std::u16string> &patterns, std::vector<uint16_t> &chars,
                     std::vector<int32_t> &off) {
        off.assign(1, 0);
        for (const auto &p : patterns) {
            chars.insert(chars.end(), p.begin(), p.end());
            off.push_back((int32_t)chars.size());
        }
        if (chars.empty()) chars.push_back(0);
    }
    int boundaryCall(int mode, int from, std::u16string &destination, int offset, char16_t boundary) const {
        int32_t len = 0, st = 0, aux = 0;
        detail::check(fmx_extract_boundary_batch(h_, &from, 1, (uint16_t)boundary, mode,
                                                 reinterpret_cast<uint16_t *>(&destination[0]),
                                                 (int32_t)destination.size(), offset, &len, nullptr, &st, &aux),
                      "fmx_extract_boundary_batch");
        detail::raise_for_status(st, aux);
        return len;
    }
    fmx_index *h_ = nullptr;
};

// fm/FmIndexBuilder.java: defaults sampleRate = 32, enableExtraction = true (FMB:21-22)
class FmIndexBuilder {
public:
    FmIndexBuilder &setSampleRate(int sampleRate) {  // FMB:34-37
        sampleRate_ = sampleRate;
        return *this;
    }
    FmIndexBuilder &setEnableExtraction(bool enable) {  // FMB:46-49
        enableExtraction_ = enable;
        return *this;
    }
    FmIndexBuilder &setDevice(int device) {  // -1 keeps the index on the host (build / save only)
        device_ = device;
        return *this;
    }
    FmIndex build(const std::u16string &input) const {  // FMB:59-61
        return FmIndex(input, sampleRate_, enableExtraction_, device_);
    }

private:
    int sampleRate_ = 32;
    bool enableExtraction_ = true;
    int device_ = 0;
};

}  // namespace index4j
