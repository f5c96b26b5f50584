// fmx_model.hpp — host-side model of an index4j FM-index: the arrays FmIndex.write serializes
// (FM:948-975), held as plain vectors.  It exists only to (a) parse / emit the serialized layout and
// (b) be flattened into the HBM blob (fmx_blob.hpp).  No query code lives on the host.
//
// Citations: FM = fm/FmIndex.java, WFBB = wavelet/WaveletFixedBlockBoosting.java,
// RRR = bitsequence/RrrVector.java, IV = intsequence/IntVector.java,
// VIV = intsequence/VariableWidthIntVector.java, CMN = intsequence/Common.java
// (under /root/reference/indices/src/main/java/com/dynatrace/).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace fmx {

inline uint64_t low_bits(int n) { return n >= 64 ? ~0ULL : ((1ULL << n) - 1ULL); }

// CMN:169-175: 1 for 0, else floor(log2 v) + 1
inline int min_bits(uint64_t v) { return v == 0 ? 1 : 64 - __builtin_clzll(v); }

inline int64_t words_for_bits(int64_t bits) { return (bits % 64 == 0) ? bits / 64 : bits / 64 + 1; }

// IV: `length` elements of `width` bits, LSB-first in 64-bit words, element i at bit i*width.
struct PackedVec {
    int32_t length = 0;
    int32_t width = 1;
    std::vector<uint64_t> words;

    void init(int32_t len, int32_t w) {  // IV:46-55
        length = len;
        width = w;
        words.assign((size_t)words_for_bits((int64_t)len * w), 0);
    }
    // IV:91-119 for a zero-initialised vector (every element written at most once by the builder)
    void set(int64_t index, uint64_t value) {
        put_bits(index * width, value, width);
    }
    void put_bits(int64_t bit, uint64_t value, int nbits) {  // also VIV:94-118
        value &= low_bits(nbits);
        size_t w = (size_t)(bit >> 6);
        int off = (int)(bit & 63);
        words[w] |= value << off;
        if (off + nbits > 64) words[w + 1] |= value >> (64 - off);
    }
    uint64_t get(int64_t index) const { return get_bits(index * width, width); }  // IV:129-143
    uint64_t get_bits(int64_t bit, int nbits) const {                             // VIV:127-140
        size_t w = (size_t)(bit >> 6);
        int off = (int)(bit & 63);
        uint64_t v = words[w] >> off;
        if (off + nbits > 64) v |= words[w + 1] << (64 - off);
        return v & low_bits(nbits);
    }
};

// RRR:92-103
struct RrrModel {
    int32_t sample_size = 32;  // counted in 15-bit blocks (RRR:278, 326, 368)
    int32_t length = 0;
    int32_t total_ones = 0;
    int32_t bits_per_offset_pos = 1;
    PackedVec classes;          // width 4
    std::vector<uint64_t> offsets;  // VIV words
    PackedVec sampled_offsets;  // lengthOfSampledOffsets
    PackedVec prefix_sums;
};

// WFBB:1589-1595
struct BlockHeader {
    int32_t bv_rank, bv_offset, var_off;
    int16_t sigma, tree_height;
};

// WFBB:1621-1629
struct SuperBlockModel {
    int16_t sigma = 0;
    int16_t block_size_log = 0;
    RrrModel rank_support;
    std::vector<BlockHeader> block_headers;
    std::vector<uint8_t> var;
    std::vector<int16_t> mapping;
};

// WFBB:105-112
struct WfbbModel {
    int64_t size = 0;
    int32_t alphabet_size = 0;
    int32_t sampling_rate = 64;
    std::vector<int64_t> count;
    std::vector<int64_t> hyper_rank;
    std::vector<int32_t> super_rank;
    std::vector<int16_t> global_mapping;
    std::vector<SuperBlockModel> sb;
};

// FM:93-131
struct FmModel {
    int32_t sample_rate = 32;
    bool enable_extract = true;
    int32_t bw_suffixes = 0, bw_positions = 0;
    int32_t length = 0;
    std::vector<int32_t> map_keys;   // monotonicMap in insertion order
    std::vector<int16_t> map_vals;
    std::vector<int32_t> C;          // cumulativeCounts
    std::vector<int32_t> look_up;    // monotonicLookUp
    PackedVec suffixes, positions;
    RrrModel sampled;
    WfbbModel wt;
};

// fmx_build.cpp
// build_device >= 0: the suffix-array stage (FM:329-394) runs on that GPU (fmx_sa_gpu.hip) and, with device_wavelet,
// the wavelet-tree encode as well (fmx_wt_gpu.hip); else everything on the host
struct SaStageStats;
int build_model(const uint16_t *text, int32_t n, int32_t sample_rate, bool enable_extract, FmModel &out,
                std::string &err, int build_device = -1, SaStageStats *stats = nullptr, bool device_wavelet = true);
void build_wavelet(const int16_t *bwt, int64_t n, int sampling_rate, WfbbModel &w);
int pick_block_size_log(const int64_t *hdr_sum, const int64_t *unc_sum, int64_t sb_size, int64_t sb_sigma, int sampling_rate);
void build_rrr(const uint64_t *bits, int64_t nbits, int sample_size, RrrModel &r, int threads = 1);
const uint16_t *rrr_offset_of_value();  // 32768 entries
const uint16_t *rrr_value_of_offset();  // 32768 entries
const uint16_t *rrr_class_base();       // 16 entries  (CARDINALITY_OFFSETS, RRR:105)
const uint8_t *rrr_bits_needed();       // 16 entries  (BITS_NEEDED_BINOMIAL_COEFFICIENTS, RRR:109-129)

// fmx_serial.cpp
int parse_model(const uint8_t *buf, size_t len, FmModel &out, std::string &err);
void emit_model(const FmModel &m, bool framed, std::vector<uint8_t> &out);
// false: a JVM's HashMap would have turned one of the key map's buckets into a tree bin, whose iteration order emit_model does
// not model (fmx_serial.cpp hashmap_order): the stream is valid, its key order inside that bucket unverified
bool key_order_is_modelled(const FmModel &m);
int validate_model(const FmModel &m, std::string &err);  // 0 ok, -3 malformed: run on every parsed stream (fmx_load)

// fmx_blob.cpp
int flatten_model(const FmModel &m, std::vector<uint8_t> &blob, std::string &err);
void set_map_by_symbol(int mode);  // -1 auto, 0 rows by superblock code, 1 rows by global symbol
void set_image_compact(int on);    // 1: images flattened from now on keep their bit vectors as RRR records (BlobHeader.compact)
void set_split_blocks(int64_t blocks);  // bit vectors above this many 15-bit blocks are decoded in chunks
void set_inv_fast(int on);         // 0: inverseSelect takes the reference's own route in every block (tests)
void set_map_fast(int on);         // 0: every present mapping entry takes the reference's own route (tests)
uint64_t blob_checksum(const uint8_t *p, size_t len);
uint64_t image_checksum(const uint8_t *blob, size_t len);  // body + header (checksum field taken as zero)
int validate_blob(const uint8_t *blob, size_t len, std::string &err);  // 0 ok, -3 malformed
int flatten_rrr_only(const RrrModel &r, std::vector<uint8_t> &blob, std::string &err);

}  // namespace fmx
