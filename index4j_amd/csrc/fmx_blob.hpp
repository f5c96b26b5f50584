// fmx_blob.hpp — the flat HBM image of an index ("blob"): one contiguous, pointer-free, relocatable
// allocation that the kernels walk and that RCCL broadcasts to the other GPUs.  Shared by the host
// flattener (fmx_blob.cpp) and the device code (fmx_device.hpp).
//
// Layout rules
//   * little-endian; every section starts on a 64-byte boundary; section offsets are stored as
//     (byte offset / 8) in a uint32 (blobs up to 32 GiB).
//   * top-level tables that every LF-step touches are small and dense so they stay in L2:
//       SbcEntry[(n_sb+1) * sigma]   {rank at superblock start incl. hyperblock rank, superblock code}
//                                    = superBlockRank + hyperBlockRank + globalMapping (WFBB:108-110)
//                                    fused into one 8-byte load; row n_sb holds count[] (WFBB:1063-1069)
//       SbDesc[n_sb]                 one 64-byte sector per superblock header (WFBB:1621-1629)
//   * per superblock: mapping (int16), BlockHeader[] (16 B, as WFBB:1589-1595), variable headers
//     (bytes as written at WFBB:742-809, + 8 guard bytes), and its RRR vector.
//   * an RRR vector (RRR:92-103) is stored array-of-records instead of the reference's four separate
//     bit-packed vectors: record k (one per `sample` 15-bit blocks) = { u32 prefix sum (RRR:101),
//     u32 offset bit pointer (RRR:99-100), the `sample` 4-bit classes of its blocks (RRR:96) },
//     padded to a power-of-two stride so a record never straddles a 64-byte sector for sample <= 64:
//     a rank touches one record sector + one sector of the offsets bit stream (RRR:97-98).
//   * the 64 KiB value-of-offset table (RRR:106) travels in the blob and is staged into LDS by
//     every workgroup.
#pragma once

#include <cstdint>

namespace fmx {

constexpr uint32_t kBlobMagic = 0x31584D46u;  // "FMX1"
constexpr uint32_t kBlobVersion = 1;

struct RrrDesc {           // 32 bytes
    uint32_t off_rec;      // records, stride 1 << rec_shift bytes
    uint32_t off_bits;     // offsets bit stream, 64-bit words LSB-first (+2 guard words)
    int32_t length;        // RRR:94
    int32_t total_ones;    // RRR:95
    int32_t rec_shift;
    int32_t n_rec;
    int32_t n_blocks;
    int32_t sample;        // RRR:93 sampleSize, in 15-bit blocks
};

struct SbDesc {            // 64 bytes
    int16_t sigma;         // WFBB:1623 (superblock alphabet size - 1)
    int16_t bsl;           // WFBB:1624 blockSizeLog
    int32_t n_blocks;
    uint32_t off_mapping;  // int16[(sigma+1) << (20 - bsl)]
    uint32_t off_bh;       // BlockHdr[n_blocks]
    uint32_t off_var;      // variable-size block headers
    int32_t var_len;
    int32_t mapping_len;
    int32_t pad;
    RrrDesc rrr;
};

struct BlockHdr {          // 16 bytes, WFBB:1589-1595
    int32_t bv_rank, bv_offset, var_off;
    int16_t sigma, tree_height;
};

struct SbcEntry {          // 8 bytes
    int32_t rank;          // hyperBlockRank + superBlockRank of the symbol at the superblock start
    int16_t sbc;           // globalMapping: superblock-local code, sigma-1 = absent
    int16_t pad;
};

struct BlobHeader {        // 256 bytes
    uint32_t magic, version;
    uint64_t total_bytes;
    int32_t sample_rate, enable_extract, length, n_keys;
    int32_t bw_suffixes, bw_positions, n_c, n_look;
    int32_t wt_sigma, n_sb, n_suffixes, n_positions;
    int64_t wt_size;
    uint32_t off_c;          // int32 cumulativeCounts[n_c]           FM:103
    uint32_t off_lookup;     // int32 monotonicLookUp[n_look]          FM:105
    uint32_t off_char2code;  // int16[65536]: monotonicMap.getOrDefault(ch, 0)  FM:97
    uint32_t off_suffixes;   // packed words of `suffixes`             FM:108
    uint32_t off_positions;  // packed words of `positions`            FM:112
    uint32_t off_sbc;        // SbcEntry[(n_sb + 1) * wt_sigma]
    uint32_t off_sbdesc;     // SbDesc[n_sb]
    uint32_t off_inv;        // uint16[32768] value-of-offset table    RRR:106
    RrrDesc sampled;         // sampledSuffixes                        FM:123
    uint8_t reserved[256 - 8 - 8 - 12 * 4 - 8 - 8 * 4 - 32];
};
static_assert(sizeof(RrrDesc) == 32, "RrrDesc");
static_assert(sizeof(SbDesc) == 64, "SbDesc");
static_assert(sizeof(BlockHdr) == 16, "BlockHdr");
static_assert(sizeof(SbcEntry) == 8, "SbcEntry");
static_assert(sizeof(BlobHeader) == 256, "BlobHeader");

// record stride (bytes, log2) for an RRR vector sampled every `sample` blocks:
// 8 header bytes + ceil(sample / 16) class words, rounded up to a power of two
inline int rrr_rec_shift(int sample) {
    int bytes = 8 + 8 * ((sample + 15) / 16);
    int sh = 4;
    while ((1 << sh) < bytes) ++sh;
    return sh;
}

}  // namespace fmx
