// fmx_blob.hpp — the flat HBM image of an index ("blob"): one contiguous, pointer-free, relocatable
// allocation that the kernels walk and that RCCL broadcasts to the other GPUs.  Shared by the host
// flattener (fmx_blob.cpp) and the device code (fmx_device.hpp).
//
// Layout rules
//   * little-endian; every section starts on a 64-byte boundary; section offsets are stored as
//     (byte offset / 8) in a uint32 (blobs up to 32 GiB).
//   * top-level tables that every LF-step touches are small and dense so they stay in L2:
//       SbcEntry[(n_sb+1) * sigma]   {C[symbol] + rank at superblock start incl. hyperblock rank, superblock code}
//                                    = superBlockRank + hyperBlockRank + globalMapping (WFBB:108-110)
//                                    fused into one 8-byte load; row n_sb holds count[] (WFBB:1063-1069)
//       SbDesc[n_sb]                 one 64-byte sector per superblock header (WFBB:1621-1629)
//   * per superblock: mapping (MapEntry, 16 B), BlockHeader[] (16 B, as WFBB:1589-1595, the root node's one-count
//     in the spare top bytes), variable headers (bytes as written at WFBB:742-809, + 16 guard bytes), its bit
//     vector as 96-bit cells (BvCell), and the inverseSelect section (InvHdr per block, NodeRec per internal node).
//   * mapping (WFBB:1628): a present entry holds what the block's header tells about the symbol and where its
//     path records are (MapEntry below); an ABSENT entry (alphabetSize-1 in the reference) holds d = distance to
//     the closest block to the right that holds the symbol (or to the end of the superblock).  The reference
//     finds that block with a linear scan (WFBB:1051-1059: 27 dependent reads on average on log text); the skip
//     pointer returns the same block in one read.  Rows are indexed by the global symbol
//     (BlobHeader.map_by_symbol) or, for very large alphabets, by the superblock code as in the reference.
//   * bit vectors (RrrVector in the reference: WFBB:116, FM:123) are EXPANDED into BvCell arrays (below).
//   * a stand-alone RrrVector (fmx_rrr_build) keeps the compressed form: 16-byte records, one per 16 blocks of 15
//     bits, whatever the sampleSize: { u32 ones before the record (the role of prefixSums, RRR:101), u32 bit
//     pointer into the offsets stream (lengthOfSampledOffsets, RRR:99-100), u64 = the 16 4-bit classes (RRR:96) }
//     + the offsets bit stream (RRR:97-98) + the value-of-offset table (RRR:106, 64 KiB in the reference): only
//     classes 0..7 are stored (16,384 entries, 32 KiB) — class 15-k is the bitwise complement in reverse offset
//     order: value(15-k, off) = ~value(k, C(15,k)-1-off) & 0x7fff — staged into LDS by the RRR kernels.
#pragma once

#include <cstdint>

namespace fmx {

constexpr uint32_t kBlobMagic = 0x31584D46u;  // "FMX1"
constexpr uint32_t kBlobVersion = 15;  // 15: a run block's mapping entry carries the next-block path's word (MapEntry.w); 14: RRR records address their offsets relative to the records; BlobHeader.compact

struct RrrRecord {         // 16 bytes: 16 blocks of 15 bits
    uint32_t ones_before;  // 1-bits in all earlier blocks
    uint32_t offset_bit;   // bit position of this record's first offset, counted from the vector's FIRST RECORD: the
                           // offsets stream lies behind the records (RrrDesc.off_rec addresses both; off_bits is free)
    uint64_t classes;      // block j's class in bits [4j, 4j+4)
};
// Records are the form of the stand-alone RrrVector handles and of every bit vector of a COMPACT image
// (BlobHeader.compact, option image_compact): the reference's own compression (15-bit blocks as class + offset,
// RRR:225-286) with a sample — ones before, offset pointer — every 16 blocks instead of every sampleRate.

// The wavelet tree's per-superblock bit vectors (WFBB:116, an RrrVector in the reference) are EXPANDED when the
// index is flattened for the GPU: Huffman-shaped levels are already near the entropy of the BWT, RRR saves little
// on them, and HBM is not the scarce resource here — the rank is.  16-byte cells:
//   {u32 ones in all earlier cells, 96 bits}
// so rankOnes(p) is ONE aligned 16-byte load + three masked popcounts (the RRR form: a record, a dependent load
// of the offset bits, a 32 KiB table lookup in LDS, ~85 VALU).  Costs 1/3 more space than the raw bits.
struct BvCell {
    uint32_t ones_before;
    uint32_t bits[3];  // bit i of the cell = bits[i >> 5] >> (i & 31)
};
constexpr uint32_t kBvCellBits = 96;

struct RrrDesc {           // 32 bytes; the first 16 are what a rank needs (one dwordx4 load)
    uint32_t off_rec;      // expanded vectors: BvCell[n_rec]; compressed ones: RrrRecord[n_rec], 64 bytes, then the offsets
                           // bit stream (64-bit words LSB-first, +2 guard words)
    uint32_t off_bits;     // a superblock's vector: offset of its inverseSelect section — InvHdr[n_blocks] + NodeRec[],
                           // below; any other vector: 0
    int32_t length;        // RRR:94
    int32_t total_ones;    // RRR:95
    int32_t n_rec;
    int32_t n_blocks;
    int32_t sample;        // RRR:93 sampleSize of the source index (informational)
    int32_t node_len;      // a superblock's vector: 16-byte units in its inverseSelect section (else 0)
};

// One entry of the superblock's symbol -> block mapping (WFBB:461-471), widened from the reference's int16 to
// everything rank() needs about (symbol, block), so that the common path of rank() is
//   {superblock entry, mapping entry} -> first cell [+ the leaf's path records] -> next cell -> ...
// and reads neither the block header nor the leaf entry, the level table or the cumulative counts:
//   x = tag[7:0] | value[31:8]
//         tag 0..16   PRESENT, fast: tag = canonical code length (0 = run block), value = occurrences of the symbol
//                     in the superblock before the block (the leaf's u24, WFBB:774-788)
//         kMapAbsent  the block does not hold the symbol (alphabetSize-1 in the reference, WFBB:383-387); value =
//                     distance to the closest block to the right that holds it, or to the end of the row
//         kMapSlow    present, "take the reference's own route": value = the reference's int16 (min(sigma-2, leaf
//                     index), WFBB:466-471).  Used for codes longer than 16 bits and for clamped entries.
//   y = A0[23:0] | code[7:0]  << 24        A0 = bit position of the root node in the superblock's bit vector
//   z = B0[23:0] | code[15:8] << 24        B0 = one-bits before it       (BlockHeaderItem, WFBB:450-454)
//   w = offset of the leaf's path records, in 8-byte units from the start of the superblock's mapping table; in a RUN block's
//       entry (tag 0: no records) the word the next-block path reads when it lands on this block (WFBB:1096-1108 at tree
//       height 0, Q11): kMapRunNext | u24, or kMapRunNextOutside = the reference's read leaves the byte array (blob v15)
// Path records (PathRec, 8 bytes, one per level d = 1 .. length-1 of the leaf's code, contiguous): the node the
// walk visits at depth d starts at bit A_d of the superblock's bit vector, with B_d one-bits before it.  In the
// reference these are `blockVectorOffset + leftTotalBvSize` and `blockVectorRank + leftOnes` of WFBB:1187-1278,
// rebuilt for every query from the level table and the cumulative counts; they depend on (block, code prefix)
// only, never on the position, so the flattener evaluates that arithmetic once per leaf (with the reference's own
// integer widths).  rank1 inside the node = rankOnes(A_d + rank in node) - B_d  (WFBB:1216-1218).
struct MapEntry {
    uint32_t x, y, z, w;
};
struct PathRec {
    uint32_t a, b;
};
// The inverseSelect section of a superblock (blob v12): the walk of WFBB:1386-1493 evaluated at flatten time.
// inverseSelect does not know the symbol in advance, so it cannot use a mapping entry; in the reference (and in
// blob <= v11) it reads the block header, the level table, and per level two cumulative-count entries before it can
// address the next node's bits, then the leaf entry, then superBlockRank[symbol].  Everything but the bits themselves
// depends on (block, code prefix) only.  Per block one InvHdr, per internal node of its tree one NodeRec (both 16
// bytes, one aligned load); the walk is  InvHdr -> {root cell, root NodeRec} -> {next cell, next NodeRec} -> ...
//   InvHdr  x = A0[23:0] | flags[31:24]   A0 = bit position of the root node in the superblock's bit vector
//           y = B0                        one-bits before it
//           z = index of the block's first NodeRec (root), in 16-byte units from the start of the section
//           w = number of NodeRecs of the block (internal nodes = leaves - 1)
//     run block (kInvRun; tree height 0, WFBB:1329-1355):  y = symbol as the reference reports it (masked to 8 bits,
//           WFBB:1332; kInvMasked is set when that changed it), z = folded superblock rank of that symbol + the block's
//           rank entry;  the answer is {y, z + index in block}
//     kInvSlow: the block's header did not pass the flattener's checks (or option inv_fast = 0): the reference's own
//           route over BlockHdr and the variable-size header bytes
//   NodeRec = two u64 halves, [0] = child taken on a 0 bit, [1] = on a 1 bit (WFBB:1235-1244):
//           internal child: idx[15:0] (its NodeRec, relative to the block's root; > the parent's, never 0)
//                           | A[39:16] | B[63:40]    rank1 inside the node = rankOnes(A + rank in node) - B (WFBB:1389-1393)
//           leaf child:     0[15:0] | symbol[31:16] | (folded superblock rank of the symbol + the leaf's u24)[63:32]
//                           (what WFBB:1495-1533 reads from the leaf entry and superBlockRank / hyperBlockRank)
struct InvHdr {
    uint32_t x, y, z, w;
};
struct NodeRec {
    uint64_t child[2];
};
constexpr uint32_t kInvRun = 0x80000000u;
constexpr uint32_t kInvMasked = 0x40000000u;
constexpr uint32_t kInvSlow = 0x20000000u;

constexpr uint32_t kMapSlow = 0xffu;
constexpr uint32_t kMapAbsent = 0xfeu;
constexpr uint32_t kMapMaxLen = 16;
constexpr uint32_t kMapRunNext = 0x01000000u;         // MapEntry.w of a run block's entry: the low 24 bits are the next-block path's u24
constexpr uint32_t kMapRunNextOutside = 0x02000000u;  // ... that read falls outside the header bytes (the JVM raises AIOOBE)

struct SbDesc {            // 64 bytes; bytes 0..15 = header of every rank, bytes 32..47 = its RRR vector
    int16_t sigma;         // WFBB:1623 (superblock alphabet size - 1)
    int16_t bsl;           // WFBB:1624 blockSizeLog
    uint32_t off_mapping;  // MapEntry[mapping_len = rows << (20 - bsl)] followed by PathRec[path_len];
                           // rows = the index's alphabet size (BlobHeader.map_by_symbol: the row of a symbol is known
                           // without the superblock's code table, so the entry is requested one load earlier) or sigma+1
    uint32_t off_bh;       // BlockHdr[n_blocks]
    uint32_t off_var;      // variable-size block headers
    int32_t n_blocks;
    int32_t var_len;
    int32_t mapping_len;
    int32_t path_len;
    RrrDesc rrr;
};

struct BlockHdr {          // 16 bytes, WFBB:1589-1595; bv_rank / bv_offset: 24 bits + a byte each of the root's one-count
    int32_t bv_rank, bv_offset, var_off;
    int16_t sigma, tree_height;
};

struct SbcEntry {          // 8 bytes
    int32_t rank;          // cumulativeCounts[symbol] (FM:103) + hyperBlockRank + superBlockRank at the superblock start
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
    uint32_t off_inv;        // uint16[16384] value-of-offset table, classes 0..7  RRR:106
    RrrDesc sampled;         // sampledSuffixes                        FM:123
    int32_t map_by_symbol;   // 1: a superblock's mapping rows are indexed by the global symbol, 0: by its superblock code
    int32_t kind;            // 0 = FM-index image, 1 = stand-alone RrrVector (fmx_rrr_build)
    uint64_t checksum;       // image_checksum(): body and header (this field taken as zero) — an image that travelled
                             // (RCCL broadcast, fmx_attach_device_blob) is the one the flattener wrote
    int32_t compact;         // 1: the image's bit vectors are RrrRecords (+ offsets streams, + the value table at off_inv),
                             // not BvCells: 0.47 instead of 0.64 bytes per text byte on the 256 MiB log, a rank costs a second
                             // dependent load and a table lookup in LDS (kernels of namespace fmxc)
    uint8_t reserved[256 - 8 - 8 - 12 * 4 - 8 - 8 * 4 - 32 - 4 - 4 - 8 - 4];
};
static_assert(sizeof(RrrDesc) == 32, "RrrDesc");
static_assert(sizeof(RrrRecord) == 16, "RrrRecord");
static_assert(sizeof(BvCell) == 16, "BvCell");
static_assert(sizeof(MapEntry) == 16, "MapEntry");
static_assert(sizeof(PathRec) == 8, "PathRec");
static_assert(sizeof(InvHdr) == 16, "InvHdr");
static_assert(sizeof(NodeRec) == 16, "NodeRec");
static_assert(sizeof(SbDesc) == 64, "SbDesc");
static_assert(sizeof(BlockHdr) == 16, "BlockHdr");
static_assert(sizeof(SbcEntry) == 8, "SbcEntry");
static_assert(sizeof(BlobHeader) == 256, "BlobHeader");

constexpr int kInvEntries = 16384;  // sum of C(15,k), k = 0..7

}  // namespace fmx
