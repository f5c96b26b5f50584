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
    // new FmIndex(char[] input, int sampleRate, boolean enableExtract)  FM:155-174.
    // buildDevice >= 0: the constructor's suffix-array stage (FM:329-394) runs on that GPU — same index, byte for byte
    FmIndex(const std::u16string &input, int sampleRate, bool enableExtract = true, int device = 0,
            int buildDevice = -1) {
        const uint16_t *chars = reinterpret_cast<const uint16_t *>(input.data());
        if (buildDevice >= 0)
            detail::check(fmx_build_on_device(chars, (int32_t)input.size(), sampleRate, enableExtract ? 1 : 0, buildDevice,
                                              &h_, nullptr, nullptr, nullptr),
                          "fmx_build_on_device");
        else
            detail::check(fmx_build(chars, (int32_t)input.size(), sampleRate, enableExtract ? 1 : 0, &h_), "fmx_build");
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

    // locate, then extract(loc, min(getInputLength(), loc + extractLength), row, 0) per hit, both on the device:
    // the composite the reference times in locateAndExtractBenchmark (FmIndexThroughputBenchmark.java:231-249).
    // Hit k of pattern i: locations[i*maxMatches+k], rows[i*maxMatches+k] (extractLength chars, only the first
    // outLen are the text), hitStatus = status of that extract call (not thrown: one hit near the end of the text
    // fails with "Stop position longer than index string" without hiding the others).
    struct HitRows {
        std::vector<int32_t> found, locations, outLen, hitStatus, hitAux;
        std::vector<std::u16string> rows;
    };
    HitRows locateExtractBatch(const std::vector<std::u16string> &patterns, int maxMatches, int extractLength) const {
        return pipeline(patterns, maxMatches, extractLength, -1, 0);
    }
    // locate, then extractUntilBoundary (mode 0) / ...Left (1) / ...Right (2) per hit, FM:640-922
    HitRows locateLinesBatch(const std::vector<std::u16string> &patterns, int maxMatches, char16_t boundary,
                             int rowLength, int mode = 0) const {
        return pipeline(patterns, maxMatches, rowLength, mode, boundary);
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

    static void packPatterns(const std::vector<std::u16string> &patterns, std::vector<uint16_t> &chars,
                             std::vector<int32_t> &off) {
        pack(patterns, chars, off);
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
    HitRows pipeline(const std::vector<std::u16string> &patterns, int maxMatches, int rowLength, int mode,
                     char16_t boundary) const {
        std::vector<uint16_t> chars;
        std::vector<int32_t> off;
        pack(patterns, chars, off);
        const int32_t n = (int32_t)patterns.size();
        const size_t slots = (size_t)n * (size_t)maxMatches;
        HitRows r;
        r.found.assign(patterns.size(), 0);
        r.locations.assign(slots, -1);
        r.outLen.assign(slots, -1);
        r.hitStatus.assign(slots, 0);
        r.hitAux.assign(slots, 0);
        std::vector<uint16_t> dst(slots * (size_t)rowLength, 0);
        std::vector<int32_t> status(patterns.size());
        if (mode < 0)
            detail::check(fmx_locate_extract_batch(h_, chars.data(), off.data(), n, maxMatches, rowLength,
                                                   r.locations.data(), r.found.data(), dst.data(), r.outLen.data(), nullptr,
                                                   status.data(), r.hitStatus.data()),
                          "fmx_locate_extract_batch");
        else
            detail::check(fmx_locate_lines_batch(h_, chars.data(), off.data(), n, maxMatches, (uint16_t)boundary, mode,
                                                 rowLength, r.locations.data(), r.found.data(), dst.data(), r.outLen.data(),
                                                 nullptr, status.data(), r.hitStatus.data(), r.hitAux.data()),
                          "fmx_locate_lines_batch");
        for (int st : status) detail::raise_for_status(st);
        r.rows.resize(slots);
        for (size_t q = 0; q < slots; ++q)
            r.rows[q].assign(reinterpret_cast<const char16_t *>(dst.data()) + q * (size_t)rowLength, (size_t)rowLength);
        return r;
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
    FmIndexBuilder &setBuildDevice(int device) {  // extension: suffix-array stage of the build on this GPU
        buildDevice_ = device;
        return *this;
    }
    FmIndex build(const std::u16string &input) const {  // FMB:59-61
        return FmIndex(input, sampleRate_, enableExtraction_, device_, buildDevice_);
    }

private:
    int sampleRate_ = 32;
    bool enableExtraction_ = true;
    int device_ = 0;
    int buildDevice_ = -1;
};

// One FmIndex on several GPUs of a node (fmx.h "replicas"): FmIndex is immutable and @ThreadSafe (FM:82), index4j's throughput
// benchmark gives every thread an index of its own (FmIndexThroughputState.java:30) — the image is copied to every device named
// (fmx_replicate: peer copies, all destinations at once) and a batch is cut into contiguous shards, one per replica, each stored
// into its own slice of the caller's arrays (fmx_*_multi): no exchange on the query path.  A device may be named more than once.
class FmIndexReplicas {
public:
    FmIndexReplicas(const FmIndex &source, const std::vector<int32_t> &devices) : handles_(devices.size(), nullptr) {
        detail::check(fmx_replicate(source.handle(), devices.data(), (int32_t)devices.size(), handles_.data()), "fmx_replicate");
    }
    FmIndexReplicas(const FmIndexReplicas &) = delete;
    FmIndexReplicas &operator=(const FmIndexReplicas &) = delete;
    ~FmIndexReplicas() {
        for (fmx_index *h : handles_) fmx_free(h);
    }
    size_t size() const { return handles_.size(); }
    int deviceOf(size_t replica) const { return fmx_device_of(handles_[replica]); }
    std::vector<int32_t> countBatch(const std::vector<std::u16string> &patterns) const {
        std::vector<uint16_t> chars;
        std::vector<int32_t> off;
        FmIndex::packPatterns(patterns, chars, off);
        std::vector<int32_t> counts(patterns.size()), status(patterns.size());
        detail::check(fmx_count_batch_multi(handles_.data(), (int32_t)handles_.size(), chars.data(), off.data(), (int32_t)patterns.size(),
                                            counts.data(), nullptr, status.data()),
                      "fmx_count_batch_multi");
        for (int s : status) detail::raise_for_status(s);
        return counts;
    }
    // returns found[i]; locations is patterns.size() rows of maxMatches ints
    std::vector<int32_t> locateBatch(const std::vector<std::u16string> &patterns, int maxMatches, std::vector<int32_t> &locations) const {
        std::vector<uint16_t> chars;
        std::vector<int32_t> off;
        FmIndex::packPatterns(patterns, chars, off);
        const int32_t n = (int32_t)patterns.size();
        locations.assign((size_t)n * (size_t)maxMatches, 0);
        std::vector<int32_t> found(patterns.size()), status(patterns.size());
        detail::check(fmx_locate_batch_multi(handles_.data(), (int32_t)handles_.size(), chars.data(), off.data(), n, maxMatches,
                                             locations.data(), maxMatches, found.data(), nullptr, status.data()),
                      "fmx_locate_batch_multi");
        for (int s : status) detail::raise_for_status(s);
        return found;
    }

private:
    std::vector<fmx_index *> handles_;
};

// One long text as K FmIndex objects over consecutive pieces (a Java int cannot address 2^31 chars, FM:131):
// count = sum over the pieces, hits = piece start + local position, in piece order — what a caller's loop over K
// indexes computes, as one device call (fmx_count_segments / fmx_locate_segments).  All pieces on one GPU.
class SegmentedFmIndex {
public:
    void add(FmIndex &&segment, int64_t textOffset) {
        segments_.push_back(std::move(segment));
        bases_.push_back(textOffset);
        handles_.push_back(segments_.back().handle());
    }
    size_t size() const { return segments_.size(); }
    std::vector<int64_t> countBatch(const std::vector<std::u16string> &patterns) const {
        std::vector<uint16_t> chars;
        std::vector<int32_t> off;
        FmIndex::packPatterns(patterns, chars, off);
        std::vector<int64_t> counts(patterns.size());
        std::vector<int32_t> status(patterns.size());
        detail::check(fmx_count_segments(handles_.data(), (int32_t)handles_.size(), chars.data(), off.data(),
                                         (int32_t)patterns.size(), counts.data(), nullptr, status.data()),
                      "fmx_count_segments");
        for (int st : status) detail::raise_for_status(st);
        return counts;
    }
    // returns found[i]; locations = patterns.size() rows of maxMatches global text positions
    std::vector<int32_t> locateBatch(const std::vector<std::u16string> &patterns, int maxMatches,
                                     std::vector<int64_t> &locations) const {
        std::vector<uint16_t> chars;
        std::vector<int32_t> off;
        FmIndex::packPatterns(patterns, chars, off);
        locations.assign(patterns.size() * (size_t)maxMatches, -1);
        std::vector<int32_t> found(patterns.size()), status(patterns.size());
        detail::check(fmx_locate_segments(handles_.data(), (int32_t)handles_.size(), bases_.data(), chars.data(),
                                          off.data(), (int32_t)patterns.size(), maxMatches, locations.data(), found.data(),
                                          status.data()),
                      "fmx_locate_segments");
        for (int st : status) detail::raise_for_status(st);
        return found;
    }

private:
    std::vector<FmIndex> segments_;
    std::vector<int64_t> bases_;
    std::vector<const fmx_index *> handles_;
};

}  // namespace index4j
