// fmx_device.hpp — per-lane device functions of the backward-search path, written for gfx950.
//
//   bv_rank1* / bv_access                   RrrVector.rankOnes RRR:358-396, access RRR:314-349 over expanded cells
//   rrr_decode / rrr_rank1 / rrr_access     the same over the compressed form (stand-alone RrrVector handles)
//   wt_rank                                 WaveletFixedBlockBoosting.rank WFBB:1010-1285
//   wt_inverse_select                       WaveletFixedBlockBoosting.inverseSelect WFBB:1305-1537
//   fm_* helpers                            FmIndex FM:455-922
//
// Results are bit-exact with the reference, including its quirks (SURVEY.md §7 Q1-Q3, Q9 and the
// unguarded (treeHeight-1)*4 of WFBB:1081).  The arithmetic is 32-bit where Java's long provably
// fits (positions < 2^31, ranks < 2^31); the class scan of RRR:376-380 is done with SWAR sums over
// the 4-bit classes of a sample record instead of a <=sample-1 iteration loop.
//
// The file compiles under hipcc (device) and under g++ (tests/hostsim.cpp — a test-only host
// simulation of the same source used to debug on machines without a GPU; the shipped library never
// builds or calls the host variant).
#pragma once

#include "fmx_blob.hpp"

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define FMX_HD __device__ inline __attribute__((always_inline))
// A COLD route: ONE real function per code object instead of a copy at every call site of the LF-walk kernels (the
// reference's own routes through the wavelet tree, met on block boundaries and quirk paths: a fraction of a percent of the
// LF-steps, but 80 % of the kernels' instructions when inlined at each of their ten-odd call sites).  It takes the index as a
// pointer to the resident copy of the launch-independent DevIndex (DevIndex.self), not a stack copy of the kernel's.
#define FMX_COLD static __device__ __attribute__((noinline, unused))
#define FMX_SELF(ix) ((ix).self)
#else
#define FMX_HD inline
#define FMX_COLD inline
#define FMX_SELF(ix) (&(ix))
#endif

namespace fmx {

// per-query status codes (mirror include/fmx.h FMX_ST_*)
enum : int {
    ST_OK = 0,
    ST_NOT_ENABLED = 1,
    ST_POS_NEGATIVE = 2,
    ST_STOP_TOO_LONG = 3,
    ST_DEST_TOO_SMALL = 4,
    ST_POS_TOO_LONG = 5,
    ST_DEST_SIZE_ZERO = 6,
    ST_NO_BOUNDARY = 7,
    ST_DOES_NOT_FIT = 8,
    ST_JAVA_AIOOBE = 9
};

// kernel-argument view of a blob resident in HBM (all pointers into the one allocation)
struct DevIndex {
    const uint8_t *base;
    const int32_t *C;            // cumulativeCounts            FM:103
    const int32_t *look_up;      // monotonicLookUp             FM:105
    const int16_t *char2code;    // monotonicMap as a 64Ki LUT  FM:97
    const uint32_t *suffix_words;
    const uint32_t *pos_words;
    const SbcEntry *sbc;
    const SbDesc *sbd;
    const uint16_t *inv_global;  // value-of-offset table (staged into LDS by the kernels)
    RrrDesc sampled;             // sampledSuffixes             FM:123
    int32_t length, sample_rate, enable_extract;
    int32_t wt_sigma, n_sb, bw_suffixes, bw_positions, n_positions, n_c;
    int32_t map_by_symbol;       // mapping rows indexed by global symbol (BlobHeader.map_by_symbol)
    // {header quad, bit-vector view quad} of every superblock, staged in LDS by the kernel (nullptr: read from HBM)
    const struct Quad *sb_cache;
    int32_t sb_cache_limit;      // stage the cache only for indexes with at most this many superblocks (0 = never)
    uint32_t wt_size;
    // Suffix table (nullptr: none): the SA interval of every string of 2 .. `suffix_chars` codes THAT OCCURS IN THE TEXT,
    // i.e. the state of FM:455-474 after a pattern's last characters — an open-addressing hash table of 16-byte
    // slots {key, start, end}; key = the codes, the LAST character in the low bits, `suffix_key_bits` (8, 12 or 16, by this
    // index's alphabet) each: the low bits of a plan record's code word as they stand (codes beyond the string: 0).  A string
    // that is not in the table (it does not occur, holds a code of 0, or its search raised a status) is simply looked for
    // with the loop.
    const struct SuffixSlot *suffix_table;
    int32_t suffix_chars;
    int32_t suffix_key_bits;
    uint32_t suffix_shift;  // group of a key = (its hashed bits * kSuffixHashMul) >> suffix_shift (fm_suffix_home)
    uint32_t suffix_mask;   // slots - 1 (slots: a power of two, >= 1024)
    // Order-1 statistics of the table's two-character strings (nullptr: none — alphabets above kOrder1MaxSigma codes): entry
    // [x * wt_sigma + y] = {where the rows of "xy" start inside x's rows, and how much of them they take} as fractions
    // ((s(xy) - C[x]) / n_x, |xy| / n_x; {0, 0}: "xy" does not occur).  k_plan_codes stages it in LDS and estimates from it
    // the SA row a pattern's search starts at: its sort key.  Results never depend on it.
    const float *suffix_order1;
    // Window directory (nullptr: none; grown on the device when an index becomes resident, like the suffix table — "win_*"
    // below): one 64-byte SECTOR per kWinW consecutive BWT positions that answers rank(c, position) for the window's three
    // most frequent symbols and inverseSelect(position) (+ the sampled-row bit) for every position that holds one of them —
    // one sector fill instead of {mapping entry, path records, a cell per tree level}.  Whatever it does not hold takes the
    // path above.  Results never depend on it: every count in it was checked against rank() when it was grown.
    const struct Quad *win;
    const uint16_t *win_other;  // the entries of the positions no class of their window holds (nullptr iff win is): six bytes each,
                                // or — win_entry4 — four (win_other_load)
    const uint64_t *win_full;   // four-byte entries: the few answers that are more than a row (a status, `suspect`, a symbol the row does
                                // not give away), eight bytes each, pointed at by their entries
    int32_t win_entry4;         // 1: four-byte entries
    int32_t win_flat;           // 1: `win` is not cells but the FLAT form — one 32-bit word per BWT position (win_step): a step of a walk
                                // is ONE sector for every position, at 4 bytes per text byte (option window_cells = 3)
    const int32_t *c_lds;       // cumulativeCounts staged in LDS by the kernel (nullptr: read C) — win_symbol_of_row
    const uint16_t *c_lut;      // ... and, beside them, where to start looking: kWinLutBuckets + 1 symbols (nullptr: the whole range)
    int32_t c_lut_shift;        // row >> c_lut_shift = the row's bucket
    // this index's DevIndex as the API layer keeps it in HBM (no LDS cache, no launch option applied): what a cold route reads
    const DevIndex *self;
};
// width of one alphabet code in a plan record's word and in a suffix-table key: 8 bits while the alphabet fits, 12 for alphabets
// of up to 4,096 codes (five codes per 64-bit word, keys of five characters), else 16
inline int fmx_code_bits_for(int32_t sigma, bool allow_12 = true) { return sigma <= 256 ? 8 : ((allow_12 && sigma <= 4096) ? 12 : 16); }
constexpr int kOrder1MaxSigma = 90;  // hard cap of the order-1 table (90^2 pairs of two floats = 64,800 bytes)
// The plan kernels stage that table in LDS BEHIND their histogram (`bins` words): it is only used — and only built — where both
// fit the kernels' dynamic LDS: sigma <= 78 at the default 4,096 bins (between 79 and 90 the table used to be built
// and never read).
constexpr size_t kPlanCodesLdsMax = 64 << 10;
#if defined(__HIPCC__)
__host__ __device__
#endif
constexpr size_t order1_lds_bytes(int bins, int sigma) {
    return (size_t)bins * 4 + (size_t)sigma * sigma * 8 + ((size_t)sigma + 1) * 4;
}
constexpr int kPlanDefaultBins = 1 << 12;
struct SuffixSlot {
    uint64_t key;  // kSuffixEmpty: free
    uint32_t start, end;
};
constexpr uint64_t kSuffixEmpty = ~0ull;
constexpr uint64_t kSuffixHashMul = 0x9E3779B97F4A7C15ull;
// Slots come in groups of 64 — 1 KiB — and probing moves by whole groups (fm_suffix_home).
constexpr int kSuffixGroupLog2 = 6;
constexpr uint32_t kSuffixGroup = 1u << kSuffixGroupLog2;

#if !defined(__HIPCC__)
inline int fmx_popc(uint32_t v) { return __builtin_popcount(v); }
inline int fmx_popcll(uint64_t v) { return __builtin_popcountll(v); }
#else
FMX_HD int fmx_popc(uint32_t v) { return __builtin_popcount(v); }
FMX_HD int fmx_popcll(uint64_t v) { return __builtin_popcountll(v); }
#endif

// keeps the compiler from narrowing a wide load whose value is masked afterwards (a 24-bit field
// fetched as one dword must not become a byte + a short load: every load instruction is a trip through
// the texture addresser, the scarce resource of these kernels)
#if defined(__HIPCC__)
#define FMX_NO_UNROLL _Pragma("clang loop unroll(disable)")
#define FMX_OPAQUE32(v) asm volatile("" : "+v"(v))
#define FMX_OPAQUE64(v) asm volatile("" : "+v"(v))
// pins a 16-byte load: one dwordx4, complete at this point (used to request independent loads together)
#define FMX_PIN_QUAD(q) asm volatile("" : "+v"((q).x), "+v"((q).y), "+v"((q).z), "+v"((q).w))
#else
#define FMX_NO_UNROLL
#define FMX_OPAQUE32(v) (void)0
#define FMX_OPAQUE64(v) (void)0
#define FMX_PIN_QUAD(q) (void)0
#endif

// 16 bytes of descriptor in one load
struct Quad {
    uint32_t x, y, z, w;
};
#if defined(FMX_DIAG_LINES) && defined(__HIPCC__)
// Diagnostic build only (tools/k_count_lines.py): which 128-byte lines of the index image each XCD's loads touch.
__device__ unsigned long long g_diag_base;    // first byte of the image (set by launch_count)
__device__ unsigned g_diag_lines[8][1 << 19];  // per XCD: one bit per line of the first 2 GiB
#endif
FMX_HD Quad ld_quad(const void *p) {
#if defined(FMX_DIAG_LINES) && defined(__HIP_DEVICE_COMPILE__)
    {
        const unsigned long long d = (unsigned long long)p - g_diag_base;
        if (d < (1ull << 31)) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            atomicOr(&g_diag_lines[xcc & 7][d >> 12], 1u << ((d >> 7) & 31));
        }
    }
#endif
    Quad q;
    memcpy(&q, p, 16);
    return q;
}
// unaligned 4 / 8 bytes from the variable-size block headers as ONE load
FMX_HD uint32_t ld32u(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    FMX_OPAQUE32(v);
    return v;
}
FMX_HD uint64_t ld64u(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    FMX_OPAQUE64(v);
    return v;
}

// unaligned little-endian reads from the variable-size block headers (WFBB:240, 1107)
FMX_HD uint32_t ld16(const uint8_t *p) {
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}
FMX_HD uint32_t ld24(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);  // the byte after the field is inside the header array or its guard bytes
    return v & 0xffffffu;
}
// `nbits` (<= 32) bits at bit position `bit` of a little-endian bit stream addressed as dwords
FMX_HD uint32_t ld_bits(const uint32_t *words, uint64_t bit, int nbits) {
    const uint32_t *p = words + (bit >> 5);
    uint64_t v = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
    v >>= (bit & 31);
    return (uint32_t)(v & ((nbits >= 32) ? 0xffffffffull : ((1ull << nbits) - 1ull)));
}

// BITS_NEEDED_BINOMIAL_COEFFICIENTS (RRR:109-129) = {1,4,7,9,11,12,13,13,13,13,12,11,9,7,4,1}, one nibble each
constexpr uint64_t kBitsNeededLut = 0x1479BCDDDDCB9741ull;
FMX_HD int rrr_bits_needed(int cls) { return (int)((kBitsNeededLut >> (4 * cls)) & 15); }
// index into the half value-of-offset table (classes 0..7 stored; class 15-k = complement in reverse
// offset order).  For cls < 8 the entry is base[cls] + off with base = CARDINALITY_OFFSETS (RRR:105,
// literals RRR:8692-8697: 0,1,16,121,576,1941,4944,9949); for cls >= 8 it is base[16-cls] - 1 - off
// and the looked-up value is complemented.  Four u16 per constant word.
FMX_HD uint32_t rrr_inv_lookup(const uint16_t *inv, int cls, uint32_t off) {
    const uint64_t w0 = 0x0079001000010000ull;  // cls 0..3 : 0, 1, 16, 121
    const uint64_t w1 = 0x26DD135007950240ull;  // cls 4..7 : 576, 1941, 4944, 9949
    const uint64_t w2 = 0x0794134F26DC3FFFull;  // cls 8..11: 16383, 9948, 4943, 1940
    const uint64_t w3 = 0x0000000F0078023Full;  // cls 12..15: 575, 120, 15, 0
    const uint64_t lo = (cls & 4) ? w1 : w0, hi = (cls & 4) ? w3 : w2;
    const uint64_t w = (cls & 8) ? hi : lo;
    const uint32_t b = (uint32_t)((w >> (16 * (cls & 3))) & 0xffff);
    const bool comp = (cls & 8) != 0;
    const uint32_t v = inv[(comp ? b - off : b + off) & (uint32_t)(kInvEntries - 1)];  // (a damaged offset stays inside the table)
    return comp ? (~v & 0x7fffu) : v;
}

// sums over the low `n` nibbles (0 <= n <= 16) of a class word:
//   ones += sum of classes, obits += sum of BITS_NEEDED[class]   (the loop body of RRR:376-380)
// BITS_NEEDED[k] with m = min(k, 15-k): 1 + 3[m>=1] + 3[m>=2] + 2[m>=3] + 2[m>=4] + [m>=5] + [m>=6]
FMX_HD void rrr_scan_word(uint64_t w, int n, uint32_t &ones, uint32_t &obits) {
    if (n <= 0) return;
    const uint64_t keep = (n >= 16) ? ~0ull : ((1ull << (4 * n)) - 1ull);
    w &= keep;
    // nibble sum
    uint64_t b = (w & 0x0f0f0f0f0f0f0f0full) + ((w >> 4) & 0x0f0f0f0f0f0f0f0full);
    ones += (uint32_t)((b * 0x0101010101010101ull) >> 56);
    // m = k ^ (k >= 8 ? 15 : 0), kept in 3 bits per nibble
    const uint64_t hi = (w >> 3) & 0x1111111111111111ull;
    const uint64_t m = (w ^ (hi * 15)) & 0x7777777777777777ull;
    const uint64_t top = 0x8888888888888888ull & keep;
    uint32_t acc = (uint32_t)n;
    acc += 3u * (uint32_t)fmx_popcll((m + 0x7777777777777777ull) & top);  // m >= 1
    acc += 3u * (uint32_t)fmx_popcll((m + 0x6666666666666666ull) & top);  // m >= 2
    acc += 2u * (uint32_t)fmx_popcll((m + 0x5555555555555555ull) & top);  // m >= 3
    acc += 2u * (uint32_t)fmx_popcll((m + 0x4444444444444444ull) & top);  // m >= 4
    acc += (uint32_t)fmx_popcll((m + 0x3333333333333333ull) & top);       // m >= 5
    acc += (uint32_t)fmx_popcll((m + 0x2222222222222222ull) & top);       // m >= 6
    obits += acc;
}

// FMX_COMPACT = 1: this translation unit serves COMPACT images (BlobHeader.compact): the bit vectors of the FM path are
// RrrRecords, and every bv_* function below decodes them (fmx_kernels.hip is compiled twice: namespace fmx for expanded
// images, fmxc for compact ones; the API picks by the image's flag)
#if !defined(FMX_COMPACT)
#define FMX_COMPACT 0
#endif
// the four fields of an RRR vector a query needs (the first 16 bytes of RrrDesc)
struct RrrView {
    uint32_t off_rec, off_bits;
    int32_t length, total_ones;
#if FMX_COMPACT
    const uint8_t *base;  // the image and the value-of-offset table: what decoding a record needs beside the record
    const uint16_t *inv;  // (bv_bind)
#endif
};
FMX_HD RrrView rrr_view_from(const Quad &q) {
    RrrView v;
    v.off_rec = q.x;
    v.off_bits = q.y;
    v.length = (int32_t)q.z;
    v.total_ones = (int32_t)q.w;
#if FMX_COMPACT
    v.base = nullptr;
    v.inv = nullptr;
#endif
    return v;
}
FMX_HD RrrView rrr_view(const RrrDesc &d) { return rrr_view_from(ld_quad(&d)); }  // one 16-byte load

// the 16-block record that covers bit `position` (0 <= position < length)
FMX_HD const RrrRecord *rrr_record_ptr(const uint8_t *base, const RrrView &d, uint32_t position) {
    return reinterpret_cast<const RrrRecord *>(base + ((uint64_t)d.off_rec << 3)) + ((position / 15u) >> 4);
}
FMX_HD RrrRecord rrr_record_from(const Quad &q) {
    RrrRecord r;
    r.ones_before = q.x;
    r.offset_bit = q.y;
    r.classes = (uint64_t)q.z | ((uint64_t)q.w << 32);
    return r;
}
FMX_HD RrrRecord rrr_load_record(const uint8_t *base, const RrrView &d, uint32_t position) {
    return rrr_record_from(ld_quad(rrr_record_ptr(base, d, position)));
}

// decode the 15-bit block that holds bit `position` (0 <= position < length) from its record:
// returns the block value; prefix = ones before the block (RRR:367-390 over the 16-block records).
FMX_HD uint32_t rrr_decode_record(const uint8_t *base, const RrrView &d, const uint16_t *inv, const RrrRecord &rec,
                                  uint32_t position, uint32_t &prefix) {
    const uint32_t block_id = position / 15u;  // RRR:367
    const uint32_t j = block_id & 15u;
    uint32_t ones = rec.ones_before;                          // RRR:370
    uint32_t obits = rec.offset_bit;                          // RRR:371-372
    rrr_scan_word(rec.classes, (int)j, ones, obits);          // RRR:376-380
    const int cls = (int)((rec.classes >> (4 * j)) & 15);     // RRR:382
    const int nb = rrr_bits_needed(cls);                      // RRR:383
    // (the offsets stream lies behind the vector's records; offset_bit counts from the first record)
    const uint32_t *bits = reinterpret_cast<const uint32_t *>(base + ((uint64_t)d.off_rec << 3));
    const uint32_t off = ld_bits(bits, obits, nb);            // RRR:386
    prefix = ones;
    return rrr_inv_lookup(inv, cls, off);                     // RRR:387-390
}
FMX_HD uint32_t rrr_decode(const uint8_t *base, const RrrView &d, const uint16_t *inv, uint32_t position,
                           uint32_t &prefix) {
    const RrrRecord rec = rrr_load_record(base, d, position);
    return rrr_decode_record(base, d, inv, rec, position, prefix);
}

// rankOnes(position) with the record fetched ahead of time by the caller (only when 0 <= position < length;
// otherwise `rec` is not looked at)
FMX_HD bool rrr_in_range(const RrrView &d, int32_t position) { return position >= 0 && position < d.length; }
FMX_HD int32_t rrr_rank1_record(const uint8_t *base, const RrrView &d, const uint16_t *inv, const RrrRecord &rec,
                                int32_t position) {
    if (position < 0) return 0;
    if (position >= d.length) return d.total_ones;
    uint32_t prefix;
    const uint32_t block = rrr_decode_record(base, d, inv, rec, (uint32_t)position, prefix);
    const uint32_t t = (uint32_t)position % 15u;
    return (int32_t)(prefix + (uint32_t)fmx_popc(block & ((1u << t) - 1u)));  // RRR:393-395
}

// RRR:358-396
FMX_HD int32_t rrr_rank1(const uint8_t *base, const RrrView &d, const uint16_t *inv, int32_t position) {
    if (position < 0) return 0;
    if (position >= d.length) return d.total_ones;
    uint32_t prefix;
    const uint32_t block = rrr_decode(base, d, inv, (uint32_t)position, prefix);
    const uint32_t t = (uint32_t)position % 15u;
    return (int32_t)(prefix + (uint32_t)fmx_popc(block & ((1u << t) - 1u)));  // RRR:393-395
}

// rankOnes(p) and access(p) from a record fetched ahead of time (see rrr_rank1_access)
FMX_HD int32_t rrr_rank1_access_record(const uint8_t *base, const RrrView &d, const uint16_t *inv, const RrrRecord &rec,
                                       int32_t position, bool &bit) {
    if (position >= d.length || position < 0) {
        bit = false;
        return position < 0 ? 0 : d.total_ones;
    }
    uint32_t prefix;
    const uint32_t block = rrr_decode_record(base, d, inv, rec, (uint32_t)position, prefix);
    const uint32_t t = (uint32_t)position % 15u;
    bit = (block >> t) & 1u;
    return (int32_t)(prefix + (uint32_t)fmx_popc(block & ((1u << t) - 1u)));
}

// RRR:314-349; out-of-range is reported through *status (IllegalArgumentException in the reference)
FMX_HD bool rrr_access(const uint8_t *base, const RrrView &d, const uint16_t *inv, int32_t position, int &status) {
    if (position < 0 || position >= d.length) {
        status = ST_JAVA_AIOOBE;
        return true;  // stops any walk that polls this bit
    }
    uint32_t prefix;
    const uint32_t block = rrr_decode(base, d, inv, (uint32_t)position, prefix);
    return (block >> ((uint32_t)position % 15u)) & 1u;
}

// rankOnes(p) and access(p) at the same position p < length: one decode (WFBB:1389-1393)
FMX_HD int32_t rrr_rank1_access(const uint8_t *base, const RrrView &d, const uint16_t *inv, int32_t position,
                                bool &bit) {
    if (position >= d.length || position < 0) {  // rankOnes saturates; access would throw — unreachable for a
        bit = false;                             // well-formed tree (the node bit always exists)
        return position < 0 ? 0 : d.total_ones;
    }
    uint32_t prefix;
    const uint32_t block = rrr_decode(base, d, inv, (uint32_t)position, prefix);
    const uint32_t t = (uint32_t)position % 15u;
    bit = (block >> t) & 1u;
    return (int32_t)(prefix + (uint32_t)fmx_popc(block & ((1u << t) - 1u)));
}

#if FMX_COMPACT
// ---- the FM path's bit vectors in a COMPACT image: 16-block RrrRecords + offsets stream -------------------------------
// Same interface as the expanded form below: the "cell" of a position is its RECORD (one aligned 16-byte load, requested
// as early as the expanded form's cell); the rank then takes a second, dependent load (the block's offset bits) and the
// value-of-offset lookup in LDS.
struct DevIndex;
FMX_HD uint32_t bv_clamp(const RrrView &d, int32_t position) {
    return (uint32_t)(position < 0 ? 0 : (position > d.length ? d.length : position));
}
FMX_HD const RrrRecord *bv_cell_ptr(const uint8_t *base, const RrrView &d, uint32_t position) {
    return rrr_record_ptr(base, d, position);  // (position <= length: n_rec = n_blocks / 16 + 1 records)
}
FMX_HD Quad bv_load_cell(const uint8_t *base, const RrrView &d, int32_t position) {
    return ld_quad(bv_cell_ptr(base, d, bv_clamp(d, position)));
}
FMX_HD int32_t bv_rank1_cell(const RrrView &d, const Quad &cell, int32_t position) {
    return rrr_rank1_record(d.base, d, d.inv, rrr_record_from(cell), position);  // saturates outside [0, length) itself
}
FMX_HD int32_t bv_rank1_access_cell(const RrrView &d, const Quad &cell, int32_t position, bool &bit) {
    return rrr_rank1_access_record(d.base, d, d.inv, rrr_record_from(cell), position, bit);
}
// rankOnes(position + 1) from the record of `position` when bit `position` is known to be set (FM:541 after the poll of FM:531)
FMX_HD int32_t bv_rank1_after_set_bit(const RrrView &d, const Quad &cell, int32_t position) {
    return bv_rank1_cell(d, cell, position) + 1;
}
#else
// ---- the wavelet tree's bit vectors: expanded 96-bit cells (fmx_blob.hpp BvCell) -------------------------------
// rankOnes / access with RrrVector's semantics (RRR:358-396, 314-349): rank saturates outside [0, length)
FMX_HD const BvCell *bv_cell_ptr(const uint8_t *base, const RrrView &d, uint32_t position) {
    return reinterpret_cast<const BvCell *>(base + ((uint64_t)d.off_rec << 3)) + position / kBvCellBits;
}
// ones among the first r (0 <= r < 96) bits of a cell
FMX_HD uint32_t bv_cell_prefix(const Quad &cell, uint32_t r) {
    const uint64_t lo = (uint64_t)cell.y | ((uint64_t)cell.z << 32);  // bits 0..63
    const uint32_t r_lo = r < 64u ? r : 64u, r_hi = r < 64u ? 0u : r - 64u;
    const uint64_t m_lo = r_lo >= 64u ? ~0ull : ((1ull << r_lo) - 1ull);
    return (uint32_t)fmx_popcll(lo & m_lo) + (uint32_t)fmx_popc(cell.w & ((1u << r_hi) - 1u));
}
// bit r (0 <= r < 96) of a cell — with shifts, not a select over the three words: the compiler turns such a select
// into an indexed read of a stack copy of the cell (scratch traffic in every LF-walk kernel: extractUntilBoundary
// 2.54 -> 2.04 ms, locate 0.61 -> 0.59 ms)
FMX_HD bool bv_cell_bit(const Quad &cell, uint32_t r) {
    const uint64_t lo = (uint64_t)cell.y | ((uint64_t)cell.z << 32);
    const uint32_t from_lo = (uint32_t)(lo >> (r & 63u)), from_hi = cell.w >> (r & 31u);
    return ((r < 64u ? from_lo : from_hi) & 1u) != 0;
}
// RrrVector.rankOnes saturates outside [0, length) (RRR:360-365).  Here that costs no branch: the position is clamped
// into [0, length], whose cell always exists (length / 96 + 2 cells), and the rank at a clamped position IS the
// saturated value — 0 at position 0, totalOnes at `length` (the bits past `length` are zero, and the flattener
// verified that the cells add up to totalOnes).
FMX_HD uint32_t bv_clamp(const RrrView &d, int32_t position) {
    return (uint32_t)(position < 0 ? 0 : (position > d.length ? d.length : position));
}
FMX_HD Quad bv_load_cell(const uint8_t *base, const RrrView &d, int32_t position) {
    return ld_quad(bv_cell_ptr(base, d, bv_clamp(d, position)));
}
// `cell` = bv_load_cell(position)
FMX_HD int32_t bv_rank1_cell(const RrrView &d, const Quad &cell, int32_t position) {
    return (int32_t)(cell.x + bv_cell_prefix(cell, bv_clamp(d, position) % kBvCellBits));
}
FMX_HD int32_t bv_rank1_access_cell(const RrrView &d, const Quad &cell, int32_t position, bool &bit) {
    // access would throw outside [0, length) — unreachable for a well-formed tree (the node bit always exists);
    // reported as a 0 bit: beyond `length` the cells hold zeros, below 0 the position is clamped onto bit 0
    const uint32_t r = bv_clamp(d, position) % kBvCellBits;
    bit = position >= 0 && bv_cell_bit(cell, r);
    return (int32_t)(cell.x + bv_cell_prefix(cell, r));
}
// rankOnes(position + 1) from the cell of `position` (0 <= position < length) when bit `position` is known to be set
FMX_HD int32_t bv_rank1_after_set_bit(const RrrView &d, const Quad &cell, int32_t position) {
    (void)d;
    return (int32_t)(cell.x + bv_cell_prefix(cell, (uint32_t)position % kBvCellBits)) + 1;
}

#endif  // FMX_COMPACT
// what a compact image's views need beside their four fields (nothing in an expanded one)
template <class Index>
FMX_HD void bv_bind(RrrView &v, const Index &ix, const uint16_t *inv) {
#if FMX_COMPACT
    v.base = ix.base;
    v.inv = inv ? inv : ix.inv_global;
#else
    (void)v;
    (void)ix;
    (void)inv;
#endif
}

// stand-alone forms (one load each); access reports an out-of-range position through *status like rrr_access
FMX_HD int32_t bv_rank1(const uint8_t *base, const RrrView &d, int32_t position) {
    if (position < 0) return 0;
    if (position >= d.length) return d.total_ones;
    return bv_rank1_cell(d, ld_quad(bv_cell_ptr(base, d, (uint32_t)position)), position);
}
FMX_HD bool bv_access(const uint8_t *base, const RrrView &d, int32_t position, int &status) {
    if (position < 0 || position >= d.length) {
        status = ST_JAVA_AIOOBE;
        return true;  // stops any walk that polls this bit
    }
    bool bit;
    (void)bv_rank1_access_cell(d, ld_quad(bv_cell_ptr(base, d, (uint32_t)position)), position, bit);
    return bit;
}

// WFBB:250-278: block-local leaf index -> canonical (code, length).  The per-level leaf counts are the
// u16 at stride 4 of the level table; up to four levels come from one 16-byte load.
FMX_HD uint32_t quad_entry(const Quad &q, int i) {
    const uint32_t lo = (i & 1) ? q.y : q.x, hi = (i & 1) ? q.w : q.z;
    return (i & 2) ? hi : lo;
}
// `chunk` = level entries 0..3 (the first 16 bytes of the header), fetched by the caller
FMX_HD void wt_restore_code(uint32_t block_c, const uint8_t *hdr, int32_t tree_height, Quad chunk, uint32_t &code,
                            int32_t &code_length) {
    code = 0;
    code_length = 1;
    uint32_t leaf_count = 0;
    int32_t lvl = 0;  // index of the level entry under inspection
    while (code_length < tree_height) {
        if (lvl != 0 && (lvl & 3) == 0) {
            chunk = ld_quad(hdr + 4 * lvl);  // entries lvl..lvl+3 (guard bytes cover the tail)
            FMX_OPAQUE32(chunk.x);
        }
        const uint32_t e = quad_entry(chunk, lvl);
        const uint32_t level_leaf_count = e & 0xffffu;
        code <<= 1;
        if (leaf_count + level_leaf_count > block_c) {
            code += block_c - leaf_count;
            break;
        }
        code += level_leaf_count;
        ++code_length;
        leaf_count += level_leaf_count;
        ++lvl;
    }
    if (code_length == tree_height) {
        code <<= 1;
        code += block_c - leaf_count;
    }
}

// rank + superblock code of a symbol as one 8-byte load
FMX_HD SbcEntry sbc_from(uint64_t v) {
    SbcEntry e;
    e.rank = (int32_t)(uint32_t)v;
    e.sbc = (int16_t)(uint16_t)(v >> 32);
    e.pad = 0;
    return e;
}
FMX_HD SbcEntry ld_sbc(const SbcEntry *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return sbc_from(v);
}

// header fields of a superblock that every rank needs: the first 16 bytes of SbDesc in one load
struct SbHead {
    int32_t sigma, bsl;
    uint32_t off_mapping, off_bh, off_var;
};
FMX_HD SbHead sb_head_from(const Quad &q) {
    SbHead h;
    h.sigma = (int32_t)(int16_t)(q.x & 0xffffu);
    h.bsl = (int32_t)(int16_t)(q.x >> 16);
    h.off_mapping = q.y;
    h.off_bh = q.z;
    h.off_var = q.w;
    return h;
}
FMX_HD SbHead sb_head(const SbDesc &sd) { return sb_head_from(ld_quad(&sd)); }
// (the image packs the root node's one-count into the spare top bytes of the two 24-bit bit-vector fields)
FMX_HD uint32_t block_hdr_root_ones(const Quad &q) { return (q.x >> 24) | ((q.y >> 24) << 8); }
FMX_HD BlockHdr block_hdr_from(const Quad &q) {
    BlockHdr b;
    b.bv_rank = (int32_t)(q.x & 0xffffffu);
    b.bv_offset = (int32_t)(q.y & 0xffffffu);
    b.var_off = (int32_t)q.z;
    b.sigma = (int16_t)(q.w & 0xffffu);
    b.tree_height = (int16_t)(q.w >> 16);
    return b;
}
FMX_HD BlockHdr ld_block_hdr(const BlockHdr *p) { return block_hdr_from(ld_quad(p)); }

// one level of the tree walk shared by rank and inverseSelect (WFBB:1187-1278 / 1388-1489):
// the cumulative one-counts of the level (u16 each): [left-1] and [left] come from one 4-byte load
struct TreeWalk {
    int32_t bv_rank, bv_offset, internal_nodes, left_siblings, left_total_bv, node_bv_size, depth_total_bv, node_rank;
    const uint8_t *hdr;     // variable-size header of the block
    uint32_t second;        // offset of the cumulative one-counts of the current level
    uint32_t level;         // offset of the next entry of the level table
    uint32_t limit;         // bytes of the superblock's header array from `hdr` on (16 guard bytes follow them)
};
// Offsets inside a header come out of the header's own entries: on a damaged index they can be anything.  Every read at
// a computed offset is clamped into the array (a well-formed header never reaches the clamp).
FMX_HD const uint8_t *hdr_bytes(const uint8_t *hdr, uint32_t limit, uint32_t off) { return hdr + (off < limit ? off : limit); }
FMX_HD const uint8_t *tree_bytes(const TreeWalk &t, uint32_t off) { return hdr_bytes(t.hdr, t.limit, off); }
// split in two so that a caller can request these together with other loads and wait once
FMX_HD void tree_level_counts_load(const TreeWalk &t, uint32_t &raw_pair, uint32_t &raw_level) {
    // entries [left-1, left], or [0, 1] for the leftmost node (entry 1 is then not looked at; the bytes exist:
    // the header ends in guard bytes) — one unconditional 4-byte load
    const int32_t first = t.left_siblings > 0 ? t.left_siblings - 1 : 0;
    memcpy(&raw_pair, tree_bytes(t, t.second + 2u * (uint32_t)first), 4);
    raw_level = ld16(tree_bytes(t, t.second + 2u * (uint32_t)(t.internal_nodes - 1)));
}
FMX_HD void tree_level_counts_decode(const TreeWalk &t, uint32_t raw_pair, uint32_t raw_level, int32_t &left_ones,
                                     int32_t &node_ones, int32_t &level_ones) {
    if (t.left_siblings > 0) {
        left_ones = (int32_t)(raw_pair & 0xffffu);                    // WFBB:1193-1206
        node_ones = (int32_t)(raw_pair >> 16) - left_ones;            // WFBB:1210-1214
    } else {
        left_ones = 0;
        node_ones = (int32_t)(raw_pair & 0xffffu);
    }
    level_ones = (int32_t)raw_level;  // WFBB:1220-1229
}
FMX_HD void tree_level_counts(const TreeWalk &t, int32_t &left_ones, int32_t &node_ones, int32_t &level_ones) {
    uint32_t raw_pair, raw_level;
    tree_level_counts_load(t, raw_pair, raw_level);
    FMX_OPAQUE32(raw_pair);
    tree_level_counts_decode(t, raw_pair, raw_level, left_ones, node_ones, level_ones);
}
FMX_HD void tree_descend(TreeWalk &t, bool bit, int32_t rank1, int32_t node_ones) {
    const int32_t node_zeros = t.node_bv_size - node_ones;
    const int32_t rank0 = t.node_rank - rank1;
    t.second += 2 * t.internal_nodes;
    t.left_siblings <<= 1;
    if (bit) {  // WFBB:1235-1244
        t.node_rank = rank1;
        t.node_bv_size = node_ones;
        ++t.left_siblings;
        t.left_total_bv += node_zeros;
    } else {
        t.node_rank = rank0;
        t.node_bv_size = node_zeros;
    }
}
// WFBB:1247-1278: next level's leaf count and total bitvector size (one 4-byte load); returns the leaf count
FMX_HD int32_t tree_next_level_entry(TreeWalk &t, uint32_t e);
FMX_HD int32_t tree_next_level(TreeWalk &t) { return tree_next_level_entry(t, ld32u(tree_bytes(t, t.level))); }
// the same step with the level entry already at hand (entries 0..3 travel with the header's first 16 bytes)
FMX_HD int32_t tree_next_level_entry(TreeWalk &t, uint32_t e) {
    const int32_t next_leaf_count = (int32_t)(e & 0xffffu);
    const int32_t next_total_bv = (int32_t)(e >> 16) + 1;
    t.level += 4;
    t.left_total_bv -= (t.depth_total_bv - next_total_bv);
    t.bv_offset += t.depth_total_bv;
    t.depth_total_bv = next_total_bv;
    t.internal_nodes = (t.internal_nodes << 1) - next_leaf_count;
    return next_leaf_count;
}

// WFBB:1010-1285.  position <= 2^31-1, symbol is a mapped code.
// `suspect` is set when the value came through a path where the reference is known to misbehave (next-block
// path onto a run block or through a clamped mapping entry, Q2/Q11; out-of-range superblock, Q3): callers
// that replace the reference's access pattern by an equivalent one only do so on un-suspect results.
//
// The SbcEntry table stores rank + cumulativeCounts[symbol] ("folded"): every LF-step adds C[c] to the rank
// (FM:469-470, 534-535), and with the sum in the table that is one load less per step.  wt_rank_folded returns
// C[symbol] + rank; wt_rank (the reference's value) subtracts it again for callers outside the FM-index.
// monotonicLookUp[c] (FM:105) for a symbol inverseSelect reported: the table is zero-padded to wt_sigma + 1 entries in the
// image and the symbol clamped to that (no well-formed tree reports such a symbol; the reference would throw)
FMX_HD uint16_t fm_char_of(const DevIndex &ix, int32_t c) {
    const uint32_t u = (uint32_t)c < (uint32_t)ix.wt_sigma ? (uint32_t)c : (uint32_t)ix.wt_sigma;
    return (uint16_t)ix.look_up[u];
}
FMX_HD int32_t fm_c_or_zero(const DevIndex &ix, int32_t symbol) {
    return (symbol >= 0 && symbol < ix.n_c) ? ix.C[symbol] : 0;
}
// ---- window directory (DevIndex.win, DevIndex.win_other): an LF directory --------------------------------------
// What a step of an LF-walk costs is the tree walk: a loop over the levels of the symbol's code that a 64-lane wave runs to the
// DEEPEST code among its 64 positions (10+ levels with 1,000 symbols; the average position needs 1.8) — rocprofv3, round 5:
// k_extract issued 25.8 load instructions and 620 VALU per wave-step.  The directory holds the step itself — for row j, the
// symbol c = BWT[j - 1] and the row C[c] + rank(c, j) the reference's two calls (FM:532-535) arrive at — in sectors addressed
// by the position p = j - 1 alone.  One 64-byte cell per kWinW = 112 positions:
//   words 0..2   folded rank (C[c] + occurrences of c in BWT[0, window start)) of class 0, 1, 2
//   word  3      symbol of class 0 | class 1 << 16            (kWinNone: the class is not used)
//   word  4      index of the window's first entry in win_other
//   word  5      symbol of class 2 | the planes' first 16 bits
//   words 5..15  three bit planes of 112 bits each, from bit 16 of word 5 on, back to back: plane 0 / 1 = low / high bit of the
//                position's class (3 = "none of the three"), plane 2 = the position's bit in sampledSuffixes (FM:123: what
//                locate polls before every step, FM:531)
// and one 6-byte entry in win_other per position of class 3, in position order: {the row the step arrives at, symbol, status,
// suspect} — everything fm_lf_step hands back for that row.  The step from row p + 1: a class position -> {symbol, count + the
// class's positions before p + 1}: ONE sector; class 3 -> the entry (index = the cell's first + the class-3 positions before p):
// a second, dependent load.  No tree walk either way.  The classes are the window's three most frequent symbols among the
// positions whose step is CLEAN (no status, no `suspect`, a symbol the int16 cast of FM:532 leaves alone): ~80 % of the
// positions of log text.
// Every step in the directory is the one fm_lf_step took over the tree — all of the reference's routes and quirks — when the
// directory was grown (win_build_cell, win_build_other): a class position's step must have arrived where the count says, and an
// entry simply IS the function's answer for its row, status and `suspect` included (a masked run block Q1, a next-block path
// Q2 / Q11, ...): results never depend on the directory, and a walk over an index that has one needs no tree-walk code at all.
constexpr uint32_t kWinW = 112;
constexpr uint32_t kWinNone = 0xffffu;
struct WinCell {
    Quad q0, q1, q2, q3;
};
FMX_HD size_t win_cells_for(uint32_t wt_size) { return (size_t)(wt_size / kWinW) + 1; }
FMX_HD WinCell win_load(const DevIndex &ix, uint32_t position, uint32_t &r) {
    const uint32_t w = position / kWinW;
    r = position - w * kWinW;
    const Quad *p = ix.win + 4 * (uint64_t)w;
    WinCell c;
    c.q0 = ld_quad(p);
    c.q1 = ld_quad(p + 1);
    c.q2 = ld_quad(p + 2);
    c.q3 = ld_quad(p + 3);
    FMX_PIN_QUAD(c.q0);
    FMX_PIN_QUAD(c.q1);
    FMX_PIN_QUAD(c.q2);
    FMX_PIN_QUAD(c.q3);
    return c;
}
// the two class planes, positions 0..63 / 64..111 (+ 16 bits of whatever follows)
struct WinPlanes {
    uint64_t a_lo, a_hi, b_lo, b_hi;
};
FMX_HD WinPlanes win_planes(const WinCell &c) {
    WinPlanes p;  // words: q1 = {4, 5, 6, 7}, q2 = {8 .. 11}, q3 = {12 .. 15}; plane 0 from bit 16 of word 5, plane 1 = words 9 .. 12
    p.a_lo = (uint64_t)((c.q1.y >> 16) | (c.q1.z << 16)) | ((uint64_t)((c.q1.z >> 16) | (c.q1.w << 16)) << 32);
    p.a_hi = (uint64_t)((c.q1.w >> 16) | (c.q2.x << 16)) | ((uint64_t)((c.q2.x >> 16) | (c.q2.y << 16)) << 32);
    p.b_lo = (uint64_t)c.q2.y | ((uint64_t)c.q2.z << 32);
    p.b_hi = (uint64_t)c.q2.w | ((uint64_t)c.q3.x << 32);
    return p;
}
// positions of class k (0 .. 3) among the window's first r
FMX_HD int32_t win_class_before(const WinPlanes &p, uint32_t k, uint32_t r) {
    const uint64_t fa = (k & 1u) ? 0ull : ~0ull, fb = (k & 2u) ? 0ull : ~0ull;
    const uint64_t m_lo = (p.a_lo ^ fa) & (p.b_lo ^ fb), m_hi = (p.a_hi ^ fa) & (p.b_hi ^ fb);
    const uint32_t r_lo = r < 64u ? r : 64u, r_hi = r < 64u ? 0u : r - 64u;  // r <= kWinW: r_hi <= 48
    const uint64_t k_lo = r_lo >= 64u ? ~0ull : ((1ull << r_lo) - 1ull);
    return fmx_popcll(m_lo & k_lo) + fmx_popcll(m_hi & ((1ull << r_hi) - 1ull));
}
// The step from row position + 1 (position < wt_size) out of its window: true = {symbol, the row the step arrives at}; false =
// the position is of class 3 and other_out is the index of its entry.  sampled_out = the position's bit in sampledSuffixes
// either way (positions beyond that vector: 0 — callers check the range first, as FM:531 would throw)
FMX_HD bool win_inv_from(const WinCell &c, uint32_t r, int32_t &symbol_out, int32_t &row_out, bool &sampled_out, uint32_t &other_out) {
    const WinPlanes p = win_planes(c);
    const uint32_t sh = r & 63u;
    const uint32_t b0 = (uint32_t)((r < 64u ? p.a_lo : p.a_hi) >> sh) & 1u, b1 = (uint32_t)((r < 64u ? p.b_lo : p.b_hi) >> sh) & 1u;
    // plane 2 from bit 16 of word 12: positions 0..63 / 64..111
    const uint64_t s_lo = (uint64_t)((c.q3.x >> 16) | (c.q3.y << 16)) | ((uint64_t)((c.q3.y >> 16) | (c.q3.z << 16)) << 32);
    const uint64_t s_hi = (uint64_t)((c.q3.z >> 16) | (c.q3.w << 16)) | ((uint64_t)(c.q3.w >> 16) << 32);
    sampled_out = (((r < 64u ? s_lo : s_hi) >> sh) & 1ull) != 0;
    const uint32_t k = b0 | (b1 << 1);
    const int32_t before = win_class_before(p, k, r);
    if (k == 3u) {
        other_out = c.q1.x + (uint32_t)before;
        return false;
    }
    symbol_out = (int32_t)(k == 0u ? (c.q0.w & 0xffffu) : (k == 1u ? (c.q0.w >> 16) : (c.q1.y & 0xffffu)));
    row_out = (int32_t)(k == 0u ? c.q0.x : (k == 1u ? c.q0.y : c.q0.z)) + before + 1;
    return true;
}
// the entry of a class-3 position = what fm_lf_step hands back for its row.  SIX bytes (round 6; eight before): the row it
// arrives at (31 bits: rows are Java ints >= 0) with `suspect` in bit 31, then the symbol (15 bits: an alphabet has at most 32,767
// codes, FM:423-426, and validate_blob holds every leaf symbol below wt_sigma) with "the step raised a status" in bit 15 — the only
// status a step can raise is the JVM's ArrayIndexOutOfBounds (every `status =` on the path of fm_lf_step).
// FOUR bytes (DevIndex.win_entry4, alphabets whose cumulativeCounts fit LDS: kWinSymbolSearchMax): the row alone.  A clean step
// with symbol c arrives at C[c] + rank, a row in (C[c], C[c + 1]]: the symbol is the largest c with C[c] < row — a search over
// cumulativeCounts (win_symbol_of_row; the kernels that want symbols stage C in LDS, locate never asks).  An answer that is more
// than that — a status, `suspect`, a row that does not give its symbol away (none on a well-formed index outside the quirk rows) —
// lives as a six-byte answer in an eight-byte slot of DevIndex.win_full, and its entry is bit 31 + the slot's index.
// 0.18 x 6 (4) bytes per position: the directory takes 1.65 (1.29) instead of 2.0 bytes per text byte of log text.
constexpr uint32_t kWinEntryWords = 3;  // 16-bit words per six-byte entry
constexpr int32_t kWinSymbolSearchMax = 2050;  // entries of cumulativeCounts up to which the four-byte form is offered (8 KB of LDS)
constexpr uint32_t kWinEntryEscape = 0x80000000u;
FMX_HD uint64_t win_other_make(int32_t row, int32_t c, int status, bool suspect) {
    return (uint64_t)(((uint32_t)row & 0x7fffffffu) | (suspect ? 0x80000000u : 0u)) |
           ((uint64_t)(((uint32_t)c & 0x7fffu) | (status != ST_OK ? 0x8000u : 0u)) << 32);
}
// false: this step's answer does not fit an entry (a row below 0, a symbol outside 15 bits, a status other than the one above —
// none of which a validated index produces): the directory is not offered for such an index
FMX_HD bool win_other_fits(int32_t row, int32_t c, int status) {
    return row >= 0 && c >= 0 && c <= 0x7fff && (status == ST_OK || status == ST_JAVA_AIOOBE);
}
FMX_HD void win_other_unpack(uint64_t entry, int32_t &symbol_out, int32_t &row_out, int &status, bool &suspect) {
    const uint32_t lo = (uint32_t)entry, hi = (uint32_t)(entry >> 32);
    symbol_out = (int32_t)(hi & 0x7fffu);
    row_out = (int32_t)(lo & 0x7fffffffu);
    if (hi & 0x8000u) status = ST_JAVA_AIOOBE;
    if (lo >> 31) suspect = true;
}
// the largest c with C[c] < row: the symbol of a clean step that arrives at `row` (C = cumulativeCounts: n_c entries, ascending)
// (a kernel that has C in LDS also has a table of kWinLutBuckets + 1 symbols there: entry b = the largest c with C[c] < b << shift,
// so that a row of bucket b has its symbol in [lut[b], lut[b + 1]] — one symbol, or two, for most rows of any text: the search
// that is left takes no step, or one)
constexpr int32_t kWinLutBuckets = 256;
FMX_HD int32_t win_lut_shift(int32_t length) {
    int32_t shift = 0;
    while ((length >> shift) >= kWinLutBuckets) ++shift;
    return shift;
}
FMX_HD int32_t win_symbol_of_row(const DevIndex &ix, int32_t row) {
    const int32_t *C = ix.c_lds ? ix.c_lds : ix.C;
    int32_t lo = 0, hi = ix.n_c - 1;
    if (ix.c_lut) {
        const int32_t b = row >> ix.c_lut_shift;  // (0 <= row <= length: an entry's row is a row of the index)
        if ((uint32_t)b < (uint32_t)kWinLutBuckets) {
            lo = ix.c_lut[b];
            hi = ix.c_lut[b + 1];
        }
    }
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (C[mid] < row)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}
FMX_HD void win_other_store(uint16_t *entries, uint32_t index, uint64_t v) {
    uint16_t *p = entries + (uint64_t)index * kWinEntryWords;
    p[0] = (uint16_t)v;
    p[1] = (uint16_t)(v >> 16);
    p[2] = (uint16_t)(v >> 32);
}
// an entry as stored (four-byte form: in the low word)
FMX_HD uint64_t win_other_load(const DevIndex &ix, uint32_t index) {
    if (ix.win_entry4) {
        uint32_t e = reinterpret_cast<const uint32_t *>(ix.win_other)[index];
        FMX_OPAQUE32(e);
        return e;
    }
    const uint16_t *p = ix.win_other + (uint64_t)index * kWinEntryWords;
    uint32_t a = p[0], b = p[1], c = p[2];  // three 2-byte loads (an entry is 2-byte aligned), one wait
    FMX_OPAQUE32(a);
    FMX_OPAQUE32(b);
    FMX_OPAQUE32(c);
    return (uint64_t)(a | (b << 16)) | ((uint64_t)c << 32);
}
// ... and what it says.  kSymbol false: the caller does not read symbol_out (locate), so nothing is searched for.
template <bool kSymbol = true>
FMX_HD void win_other_from(const DevIndex &ix, uint64_t entry, int32_t &symbol_out, int32_t &row_out, int &status, bool &suspect) {
    if (ix.win_entry4) {
        const uint32_t e = (uint32_t)entry;
        if (e & kWinEntryEscape) {
            win_other_unpack(ix.win_full[e & ~kWinEntryEscape], symbol_out, row_out, status, suspect);
            return;
        }
        row_out = (int32_t)e;
        if (kSymbol) symbol_out = win_symbol_of_row(ix, (int32_t)e);
        return;
    }
    win_other_unpack(entry, symbol_out, row_out, status, suspect);
}

// THE FLAT FORM of the directory (DevIndex.win_flat; option window_cells = 3, never picked by itself): no cells — one 32-bit word
// per BWT position p: the row the step from row p + 1 arrives at (30 bits: texts below 2^30 characters), bit 30 = the position's bit
// in sampledSuffixes, bit 31 = "more than a row": the low bits are then the index of an eight-byte slot in win_full, as with the
// four-byte entries.  A step of a walk is ONE random sector for EVERY position (the cells' form: 1.18 on log text, the second
// one dependent on the first) at 4 bytes per text byte instead of 1.26: the fast end of the space / time trade.
constexpr uint32_t kWinFlatEscape = 0x80000000u, kWinFlatSampled = 0x40000000u, kWinFlatRow = 0x3fffffffu;
FMX_HD uint32_t win_flat_load(const DevIndex &ix, uint32_t p) {
    uint32_t e = reinterpret_cast<const uint32_t *>(ix.win)[p];
    FMX_OPAQUE32(e);
    return e;
}
template <bool kSymbol = true>
FMX_HD void win_flat_from(const DevIndex &ix, uint32_t e, int32_t &symbol_out, int32_t &row_out, bool &sampled_out, int &status,
                          bool &suspect) {
    sampled_out = (e & kWinFlatSampled) != 0;
    if (e & kWinFlatEscape) {
        win_other_unpack(ix.win_full[e & kWinFlatRow], symbol_out, row_out, status, suspect);
        return;
    }
    row_out = (int32_t)(e & kWinFlatRow);
    if (kSymbol) symbol_out = win_symbol_of_row(ix, row_out);
}
// The step from row p + 1 (p < wt_size) out of the directory, whichever form it has: {symbol, the row it arrives at, the position's
// bit in sampledSuffixes} and whatever status / `suspect` the step carries.  kSymbol false: the caller does not read symbol_out.
// kStopAtSampled (locate, FM:531): a sampled position's step is not looked at — true is returned before the second load.
// kForm: what the caller knows about the directory's form at COMPILE time — kFormAsk = look at ix.win_flat, kFormCells / kFormFlat =
// that form and no code for the other (the walk kernels of locate / extract are instantiated per form: a body that carries both
// measured 2.5 % slower over the cells).
enum : int { kFormAsk = 0, kFormCells = 1, kFormFlat = 2 };
template <bool kSymbol = true, bool kStopAtSampled = false, int kForm = kFormAsk>
FMX_HD bool win_step(const DevIndex &ix, uint32_t p, int32_t &symbol_out, int32_t &row_out, bool &sampled_out, int &status,
                     bool &suspect) {
    if (kForm == kFormFlat || (kForm == kFormAsk && ix.win_flat)) {
        const uint32_t e = win_flat_load(ix, p);
        if (kStopAtSampled && (e & kWinFlatSampled)) {  // (whatever the step carries is not looked at)
            sampled_out = true;
            return true;
        }
        win_flat_from<kSymbol>(ix, e, symbol_out, row_out, sampled_out, status, suspect);
        return false;
    }
    uint32_t r, other = 0;
    const WinCell cell = win_load(ix, p, r);
    const bool answered = win_inv_from(cell, r, symbol_out, row_out, sampled_out, other);
    if (kStopAtSampled && sampled_out) return true;
    if (!answered) win_other_from<kSymbol>(ix, win_other_load(ix, other), symbol_out, row_out, status, suspect);
    return false;
}

// With the superblock's header at hand (ix.sb_cache, staged in LDS by the kernel) and the mapping rows indexed by
// the global symbol, the mapping entry and the block header are requested together with the superblock entry:
//   {superblock entry, mapping entry, block header} -> first cell -> ...
// what a cold route hands back: its value and, packed, the status / flags it would have set
struct ColdOut {
    int32_t value, aux;
};
FMX_COLD ColdOut wt_rank_folded_cold(const DevIndex *self, uint32_t position, int32_t symbol);
// kHot: the copy inlined into a kernel's loop — the common path and the cheap exits here, everything else (the reference's own
// route, the literal next-block arithmetic) by ONE call of the cold copy, which is this very function with kHot = false.
template <bool kHot>
FMX_HD int32_t wt_rank_folded_t(const DevIndex &ix, const uint16_t *inv, uint32_t position, int32_t symbol, int &status,
                                bool &suspect) {
    const Quad *sb_cache = ix.sb_cache;
    if (position == 0) return fm_c_or_zero(ix, symbol);                 // WFBB:1012-1014
    if (position > ix.wt_size) position = ix.wt_size;                   // WFBB:1015-1017
    if (symbol >= ix.wt_sigma) return fm_c_or_zero(ix, symbol);         // WFBB:1018-1020
    const uint32_t sb_id = position >> 20;                              // WFBB:1023
    if (sb_id >= (uint32_t)ix.n_sb || symbol < 0) {  // Q3: the JVM raises ArrayIndexOutOfBounds here
        status = ST_JAVA_AIOOBE;
        suspect = true;
        return fm_c_or_zero(ix, symbol);
    }
    // The loads of one rank form a dependent chain (superblock -> mapping -> block header -> leaf -> levels);
    // what a stage needs is requested as soon as its address is known, so that the chain is
    //   {superblock entry, superblock header, RRR view} -> {mapping entry, block header}
    //   -> {leaf, level table, first level's counts, first RRR record} -> offset bits -> ...
    uint64_t sbc_raw;
    memcpy(&sbc_raw, ix.sbc + (uint64_t)sb_id * (uint32_t)ix.wt_sigma + (uint32_t)symbol, 8);  // WFBB:1024, 1034-1037
    const SbDesc &sd = ix.sbd[sb_id];  // WFBB:1026
    Quad head_q, view_q;
    if (sb_cache) {
        head_q = sb_cache[2 * sb_id];
        view_q = sb_cache[2 * sb_id + 1];
    } else {
        head_q = ld_quad(&sd);
        view_q = ld_quad(&sd.rrr);  // same 64-byte line
        FMX_PIN_QUAD(head_q);
        FMX_PIN_QUAD(view_q);
    }
    const SbHead sh = sb_head_from(head_q);
    RrrView rv = rrr_view_from(view_q);
    bv_bind(rv, ix, inv);
    const int32_t bsl = sh.bsl;
    const uint32_t block_size = 1u << bsl;
    const int32_t blocks_log = 20 - bsl;
    const uint32_t block_index = position & (block_size - 1);
    uint32_t block_id = (position & 0xfffffu) >> bsl;
    const MapEntry *mapping = reinterpret_cast<const MapEntry *>(ix.base + ((uint64_t)sh.off_mapping << 3));
    const BlockHdr *bhs = reinterpret_cast<const BlockHdr *>(ix.base + ((uint64_t)sh.off_bh << 3));
    const uint8_t *var = ix.base + ((uint64_t)sh.off_var << 3);
    Quad mq = {0, 0, 0, 0};
    uint32_t map_row = (uint32_t)symbol << blocks_log;
    if (ix.map_by_symbol)  // the row does not depend on the superblock entry: ask for both at once
        mq = ld_quad(mapping + map_row + block_id);  // WFBB:1044-1046 (+ what the symbol's leaf would tell)
    FMX_OPAQUE64(sbc_raw);
    const SbcEntry e = sbc_from(sbc_raw);
    if (ix.map_by_symbol) FMX_PIN_QUAD(mq);
    if ((int32_t)e.sbc >= sh.sigma + 1) return e.rank;  // WFBB:1040-1042
    if (!ix.map_by_symbol) {
        map_row = (uint32_t)e.sbc << blocks_log;
        mq = ld_quad(mapping + map_row + block_id);
        FMX_PIN_QUAD(mq);
    }
    const uint32_t tag = mq.x & 0xffu;

    if (tag == kMapAbsent) {  // WFBB:1048-1110: the entry holds the distance to the closest block to the right that
                              // holds the symbol (what the scan of WFBB:1051-1059 finds), or to the superblock end
        block_id += mq.x >> 8;
        if (block_id >= (1u << blocks_log))  // WFBB:1060-1069 (row n_sb of the table holds count[])
            return ix.sbc[(uint64_t)(sb_id + 1) * (uint32_t)ix.wt_sigma + (uint32_t)symbol].rank;
        const Quad nq = ld_quad(mapping + map_row + block_id);
        const uint32_t ntag = nq.x & 0xffu;
        // WFBB:1096-1108 reads the u24 of leaf `mapping value` of that block.  A fast entry (never clamped, see
        // fmx_blob.hpp) of a block with a tree IS that leaf's u24, the symbol's own: not suspect.
        if (ntag >= 1u && ntag <= kMapMaxLen) return e.rank + (int32_t)(nq.x >> 8);
        if constexpr (kHot) {
            const ColdOut r = wt_rank_folded_cold(FMX_SELF(ix), position, symbol);
            if (r.aux & 0xff) status = r.aux & 0xff;
            if (r.aux >> 8) suspect = true;
            return r.value;
        }
        // run block (tree height 0: the reference reads 4 bytes early, Q11) or an entry on the reference's route
        // (clamped, no fix-up here: Q2): the literal address arithmetic of WFBB:1080-1081
        const int32_t block_c = ntag == kMapSlow ? (int32_t)(nq.x >> 8) : 0;
        const BlockHdr bh = ld_block_hdr(bhs + block_id);
        const int32_t p = bh.var_off + ((int32_t)bh.tree_height - 1) * 4 + block_c * 5 + 2;
        if (p < 0 || p + 2 >= sd.var_len) {
            status = ST_JAVA_AIOOBE;
            suspect = true;
            return fm_c_or_zero(ix, symbol);
        }
        // the result is only trustworthy if the entry read is the symbol's own
        if (bh.tree_height == 0 || p < 2 || (int32_t)ld16(var + p - 2) != symbol) suspect = true;
        return e.rank + (int32_t)(ld32u(var + p) & 0xffffffu);  // WFBB:1096-1108
    }

    if (tag <= kMapMaxLen) {
        // The common path.  The mapping entry holds the leaf's rank at the block start, its canonical code and the
        // root node {A0 = bit position, B0 = ones before it}; the nodes below the root come from the leaf's path
        // records, whose address is known now — so everything but the cells is requested at once, and each level
        // is ONE dependent 16-byte load: rank1 in the node = rankOnes(A_d + rank in node) - B_d (WFBB:1216-1218).
        const int32_t code_length = (int32_t)tag;
        const int32_t rank_block = (int32_t)(mq.x >> 8);
        if (code_length == 0) return e.rank + rank_block + (int32_t)block_index;  // run block, WFBB:1141-1146
        const uint32_t code = (mq.y >> 24) | ((mq.z >> 24) << 8);
        uint32_t node_a = mq.y & 0xffffffu, node_b = mq.z & 0xffffffu;
        const PathRec *path = reinterpret_cast<const PathRec *>(mapping) + mq.w;
        int32_t pos = (int32_t)node_a + (int32_t)block_index;
        Quad pq = {0, 0, 0, 0}, cell = {0, 0, 0, 0};
        if (code_length > 1) pq = ld_quad(path);  // records of levels 1 and 2
        cell = bv_load_cell(ix.base, rv, pos);
        FMX_PIN_QUAD(pq);
        FMX_PIN_QUAD(cell);
        int32_t node_rank = (int32_t)block_index;
        FMX_NO_UNROLL  // one copy of the level code: peeled copies only add instruction-cache pressure
        for (int32_t depth = 0; depth < code_length; ++depth) {
            const int32_t rank1 = bv_rank1_cell(rv, cell, pos) - (int32_t)node_b;
            node_rank = (code >> (code_length - depth - 1)) & 1u ? rank1 : node_rank - rank1;  // WFBB:1235-1244
            if (depth + 1 != code_length) {
                node_a = (depth & 1) ? pq.z : pq.x;  // record `depth` = the node at level depth + 1
                node_b = (depth & 1) ? pq.w : pq.y;
                pos = (int32_t)node_a + node_rank;
                cell = bv_load_cell(ix.base, rv, pos);
                if ((depth & 1) && depth + 2 < code_length) pq = ld_quad(path + depth + 1);  // the next two levels
                FMX_PIN_QUAD(cell);
                FMX_PIN_QUAD(pq);
            }
        }
        return e.rank + rank_block + node_rank;  // WFBB:1281-1284
    }

    // The reference's own route (codes longer than 16 bits, clamped mapping entries): block header, leaf entry,
    // clamped-mapping fix-up, code rebuilt from the level table, walk over the level table and cumulative counts.
    if constexpr (kHot) {
        const ColdOut r = wt_rank_folded_cold(FMX_SELF(ix), position, symbol);
        if (r.aux & 0xff) status = r.aux & 0xff;
        if (r.aux >> 8) suspect = true;
        return r.value;
    }
    int32_t block_c = (int32_t)(mq.x >> 8);
    Quad bhq = ld_quad(bhs + block_id);  // WFBB:1113
    FMX_PIN_QUAD(bhq);
    const BlockHdr bh = block_hdr_from(bhq);
    const int32_t tree_height = bh.tree_height;
    const uint8_t *hdr = var + bh.var_off;
    const uint32_t hdr_limit = (uint32_t)(sd.var_len - bh.var_off);  // (>= the header's own size: validate_model / validate_blob)
    const uint32_t second0 = (uint32_t)((tree_height - 1) * 4 + ((int32_t)bh.sigma + 1) * 5);  // WFBB:1177-1182
    int32_t rank_block, code_length, position0;
    uint32_t code, counts0 = 0;
    Quad chunk = {0, 0, 0, 0}, rec_q = {0, 0, 0, 0};
    {
        const uint32_t leaves = (uint32_t)(tree_height > 0 ? (tree_height - 1) * 4 : 0);  // WFBB:1119-1121
        position0 = bh.bv_offset + (int32_t)block_index;
        uint64_t leaf;
        // {u16 symbol, u24 rank at block start} + 3 bytes of the next entry
        memcpy(&leaf, hdr_bytes(hdr, hdr_limit, leaves + 5u * (uint32_t)block_c), 8);
        if (tree_height > 0) {
            chunk = ld_quad(hdr);  // level entries 0..3 (guard bytes cover the tail)
            counts0 = block_hdr_root_ones(bhq);
            rec_q = bv_load_cell(ix.base, rv, position0);
        }
        FMX_OPAQUE64(leaf);
        FMX_PIN_QUAD(chunk);
        FMX_PIN_QUAD(rec_q);
        if ((int32_t)(leaf & 0xffffu) != symbol) {  // WFBB:1123-1130: clamped mapping entry
            ++block_c;
            leaf = ld64u(hdr_bytes(hdr, hdr_limit, leaves + 5u * (uint32_t)block_c));
        }
        rank_block = (int32_t)((leaf >> 16) & 0xffffffu);                         // WFBB:1132-1138
        if (tree_height == 0) return e.rank + rank_block + (int32_t)block_index;  // WFBB:1141-1146
        wt_restore_code((uint32_t)block_c, hdr, tree_height, chunk, code, code_length);  // WFBB:1148-1156
    }

    const uint32_t cur_block_size = (ix.wt_size - (position - block_index)) < block_size
                                        ? (ix.wt_size - (position - block_index))
                                        : block_size;  // WFBB:1032
    TreeWalk t;
    t.bv_rank = bh.bv_rank;      // WFBB:1158
    t.bv_offset = bh.bv_offset;  // WFBB:1161
    t.internal_nodes = 1;
    t.left_siblings = 0;
    t.left_total_bv = 0;
    t.node_bv_size = (int32_t)cur_block_size;
    t.depth_total_bv = t.node_bv_size;
    t.node_rank = (int32_t)block_index;
    t.hdr = hdr;
    t.second = second0;
    t.level = 0;
    t.limit = hdr_limit;

    // WFBB:1185-1279.  Level 0's counts are the single u16 at `second` (no left sibling, one internal node);
    // the counts and the record of level d+1 are requested at the end of level d.
    int32_t left_ones = 0, node_ones = (int32_t)counts0, level_ones = (int32_t)counts0;
    int32_t rrr_position = position0;
    FMX_NO_UNROLL  // one copy of the level code: peeled copies only add instruction-cache pressure
    for (int32_t depth = 0; depth < code_length; ++depth) {
        int32_t rank1 = bv_rank1_cell(rv, rec_q, rrr_position);
        rank1 -= t.bv_rank + left_ones;
        t.bv_rank += level_ones;
        tree_descend(t, (code & (1u << (code_length - depth - 1))) != 0, rank1, node_ones);
        if (depth + 1 != code_length) {
            const uint32_t entry = depth < 4 ? quad_entry(chunk, depth) : ld32u(tree_bytes(t, t.level));
            t.left_siblings -= tree_next_level_entry(t, entry);
            uint32_t raw_pair, raw_level;
            tree_level_counts_load(t, raw_pair, raw_level);
            rrr_position = t.bv_offset + t.left_total_bv + t.node_rank;
            rec_q = bv_load_cell(ix.base, rv, rrr_position);
            FMX_OPAQUE32(raw_pair);
            FMX_OPAQUE32(raw_level);
            FMX_PIN_QUAD(rec_q);
            tree_level_counts_decode(t, raw_pair, raw_level, left_ones, node_ones, level_ones);
        }
    }
    return e.rank + rank_block + t.node_rank;  // WFBB:1281-1284
}

FMX_COLD ColdOut wt_rank_folded_cold(const DevIndex *self, uint32_t position, int32_t symbol) {
    int status = ST_OK;
    bool suspect = false;
    ColdOut r;
    r.value = wt_rank_folded_t<false>(*self, nullptr, position, symbol, status, suspect);
    r.aux = status | (suspect ? 0x100 : 0);
    return r;
}
// rank inside a kernel's loop (k_count: two of these per pattern character): every route inlined — a call in that loop, however
// rarely taken, costs the kernel 10-15 spilled VGPRs at its 64-register budget and 1.5 % of the headline (measured, round 5;
// FMX_COUNT_COLD_ROUTE = 1 builds the other form)
#if !defined(FMX_COUNT_COLD_ROUTE)
#define FMX_COUNT_COLD_ROUTE 0
#endif
FMX_HD int32_t wt_rank_folded(const DevIndex &ix, const uint16_t *inv, uint32_t position, int32_t symbol, int &status,
                              bool &suspect) {
    return wt_rank_folded_t<FMX_COUNT_COLD_ROUTE != 0>(ix, inv, position, symbol, status, suspect);
}
FMX_HD int32_t wt_rank_folded(const DevIndex &ix, const uint16_t *inv, uint32_t position, int32_t symbol, int &status) {
    bool suspect = false;
    return wt_rank_folded_t<FMX_COUNT_COLD_ROUTE != 0>(ix, inv, position, symbol, status, suspect);
}
// the whole of it by one call: the LF-walks need rank() only where a step crosses a block boundary or meets a quirk
FMX_HD int32_t wt_rank_folded_rare(const DevIndex &ix, uint32_t position, int32_t symbol, int &status, bool &suspect) {
    const ColdOut r = wt_rank_folded_cold(FMX_SELF(ix), position, symbol);
    if (r.aux & 0xff) status = r.aux & 0xff;
    if (r.aux >> 8) suspect = true;
    return r.value;
}
// WaveletFixedBlockBoosting.rank as the reference returns it
FMX_HD int32_t wt_rank(const DevIndex &ix, const uint16_t *inv, uint32_t position, int32_t symbol, int &status,
                       bool &suspect) {
    return wt_rank_folded(ix, inv, position, symbol, status, suspect) - fm_c_or_zero(ix, symbol);
}
FMX_HD int32_t wt_rank(const DevIndex &ix, const uint16_t *inv, uint32_t position, int32_t symbol, int &status) {
    bool suspect = false;
    return wt_rank(ix, inv, position, symbol, status, suspect);
}

// WFBB:1305-1537: returns the symbol at `position` (< size); *rank = occurrences before it
// (the reference packs (rank << 32) | symbol and returns the bare symbol when position == 0).
// (rank_out is FOLDED: C[symbol] + occurrences before `position`, see wt_rank_folded)
// exact_out = false only for a run block whose symbol code is >= 256 (WFBB:1332 masks it to 8 bits, Q1).
//
// The block's InvHdr (fmx_blob.hpp) can be requested by the caller ahead of time (`inv_hdr_ptr` depends on the
// position only): the LF-walks ask for it together with the sampled-row cell they poll before every step.
struct InvView {
    RrrView rv;
    int32_t bsl;
};
FMX_HD InvView wt_inv_view(const DevIndex &ix, uint32_t sb_id, const uint16_t *inv = nullptr) {
    Quad head_q, view_q;
    if (ix.sb_cache) {
        head_q = ix.sb_cache[2 * sb_id];
        view_q = ix.sb_cache[2 * sb_id + 1];
    } else {
        const SbDesc &sd = ix.sbd[sb_id];
        head_q = ld_quad(&sd);
        view_q = ld_quad(&sd.rrr);
        FMX_PIN_QUAD(head_q);
        FMX_PIN_QUAD(view_q);
    }
    InvView v;
    v.rv = rrr_view_from(view_q);
    bv_bind(v.rv, ix, inv);
    v.bsl = (int32_t)(int16_t)(head_q.x >> 16);
    return v;
}
FMX_HD const InvHdr *wt_inv_hdr_ptr(const DevIndex &ix, const InvView &v, uint32_t position) {
    return reinterpret_cast<const InvHdr *>(ix.base + ((uint64_t)v.rv.off_bits << 3)) + ((position & 0xfffffu) >> v.bsl);
}

// the reference's own route (blocks whose InvHdr says kInvSlow): block header, level table, cumulative counts per
// level, leaf entry, superBlockRank[symbol]
FMX_HD int32_t wt_inverse_select_reference_route(const DevIndex &ix, uint32_t position, const InvView &v,
                                                 int32_t &rank_out, bool &exact_out) {
    const uint32_t sb_id = position >> 20;
    const SbHead sh = sb_head(ix.sbd[sb_id]);
    const RrrView &rv = v.rv;
    const uint32_t block_size = 1u << v.bsl;
    const uint32_t block_index = position & (block_size - 1);
    const uint32_t block_id = (position & 0xfffffu) >> v.bsl;
    const BlockHdr *bhs = reinterpret_cast<const BlockHdr *>(ix.base + ((uint64_t)sh.off_bh << 3));
    const uint8_t *var = ix.base + ((uint64_t)sh.off_var << 3);
    const Quad bhq = ld_quad(bhs + block_id);
    const BlockHdr bh = block_hdr_from(bhq);
    const int32_t tree_height = bh.tree_height;
    const uint8_t *hdr = var + bh.var_off;
    const uint32_t hdr_limit = (uint32_t)(ix.sbd[sb_id].var_len - bh.var_off);
    const uint8_t *leaves = hdr + (tree_height > 0 ? (tree_height - 1) * 4 : 0);  // WFBB:1324-1327
    const SbcEntry *row = ix.sbc + (uint64_t)sb_id * (uint32_t)ix.wt_sigma;
    exact_out = true;
    if (tree_height == 0) {  // WFBB:1329-1355
        const uint64_t leaf = ld64u(leaves);
        const int32_t c = (int32_t)(leaf & 0xffu);  // WFBB:1332 (Q1)
        exact_out = (int32_t)(leaf & 0xffffu) == c;
        rank_out = (c < ix.wt_sigma ? row[c].rank : 0) + (int32_t)((leaf >> 16) & 0xffffffu) + (int32_t)block_index;
        return c;
    }

    const uint32_t second0 = (uint32_t)((tree_height - 1) * 4 + ((int32_t)bh.sigma + 1) * 5);
    int32_t rrr_position = bh.bv_offset + (int32_t)block_index;
    Quad chunk = ld_quad(hdr);  // level entries 0..3 (guard bytes cover the tail)
    const uint32_t counts0 = block_hdr_root_ones(bhq);  // = the u16 at hdr + second0 (WFBB:793-809)
    Quad rec_q = {0, 0, 0, 0};
    rec_q = bv_load_cell(ix.base, rv, rrr_position);
    FMX_PIN_QUAD(chunk);
    FMX_PIN_QUAD(rec_q);

    const uint32_t cur_block_size = (ix.wt_size - (position - block_index)) < block_size
                                        ? (ix.wt_size - (position - block_index))
                                        : block_size;
    uint32_t code = 0;
    int32_t code_length = 0;
    TreeWalk t;
    t.bv_rank = bh.bv_rank;
    t.bv_offset = bh.bv_offset;
    t.internal_nodes = 1;
    t.left_siblings = 0;
    t.left_total_bv = 0;
    t.node_bv_size = (int32_t)cur_block_size;
    t.depth_total_bv = t.node_bv_size;
    t.node_rank = (int32_t)block_index;
    t.hdr = hdr;
    t.second = second0;
    t.level = 0;
    t.limit = hdr_limit;

    // WFBB:1386-1493; level 0's counts are the single u16 at `second`, the loads of level d+1 are requested at
    // the end of level d
    int32_t left_ones = 0, node_ones = (int32_t)counts0, level_ones = (int32_t)counts0;
    FMX_NO_UNROLL
    for (int32_t depth = 0;; ++depth) {
        bool next_bit;
        int32_t rank1 = bv_rank1_access_cell(rv, rec_q, rrr_position, next_bit);
        rank1 -= t.bv_rank + left_ones;
        t.bv_rank += level_ones;
        code = (code << 1) | (next_bit ? 1u : 0u);
        ++code_length;
        tree_descend(t, next_bit, rank1, node_ones);
        if (depth + 1 < tree_height) {
            const uint32_t entry = depth < 4 ? quad_entry(chunk, depth) : ld32u(tree_bytes(t, t.level));
            const int32_t next_leaf_count = tree_next_level_entry(t, entry);
            if (t.left_siblings >= next_leaf_count)  // WFBB:1485-1489
                t.left_siblings -= next_leaf_count;
            else
                break;
            uint32_t raw_pair, raw_level;
            tree_level_counts_load(t, raw_pair, raw_level);
            rrr_position = t.bv_offset + t.left_total_bv + t.node_rank;
            rec_q = bv_load_cell(ix.base, rv, rrr_position);
            FMX_OPAQUE32(raw_pair);
            FMX_OPAQUE32(raw_level);
            FMX_PIN_QUAD(rec_q);
            tree_level_counts_decode(t, raw_pair, raw_level, left_ones, node_ones, level_ones);
        } else {
            break;
        }
    }
    // WFBB:232-248 computeSymbolFromBlockHeader (level entries 0..3 are already at hand)
    uint32_t block_c = 0, temp_code = 0;
    for (int32_t i = 1; i < code_length; ++i) {
        const uint32_t level_leaf_count = (i <= 4 ? quad_entry(chunk, i - 1) : ld32u(hdr + 4 * (i - 1))) & 0xffffu;
        temp_code += level_leaf_count;
        block_c += level_leaf_count;
        temp_code <<= 1;
    }
    block_c += code - temp_code;
    // {u16 symbol, u24 rank at block start}, WFBB:1495-1520
    const uint64_t leaf = ld64u(hdr_bytes(hdr, hdr_limit, (uint32_t)(tree_height - 1) * 4u + 5u * block_c));
    const int32_t c = (int32_t)(leaf & 0xffffu);
    rank_out = (c < ix.wt_sigma ? row[c].rank : 0) + (int32_t)((leaf >> 16) & 0xffffffu) + t.node_rank;  // WFBB:1521-1533
    return c;
}

// ... as ONE real function per code object (FMX_COLD): blocks on that route are rare, its code is long
FMX_COLD ColdOut wt_inverse_select_route_cold(const DevIndex *self, uint32_t position) {
    const InvView v = wt_inv_view(*self, position >> 20, nullptr);
    int32_t rank = 0;
    bool exact = true;
    const int32_t c = wt_inverse_select_reference_route(*self, position, v, rank, exact);
    ColdOut r;
    r.value = rank;
    r.aux = (c & 0xffff) | (exact ? 0x10000 : 0);
    return r;
}
FMX_HD int32_t wt_inverse_select_route_rare(const DevIndex &ix, uint32_t position, int32_t &rank_out, bool &exact_out) {
    const ColdOut r = wt_inverse_select_route_cold(FMX_SELF(ix), position);
    rank_out = r.value;
    exact_out = (r.aux & 0x10000) != 0;
    return r.aux & 0xffff;
}

// the walk with the block's InvHdr at hand
// kCold: the rare routes of an LF-step by a call (extractUntilBoundary: 15,800 -> 3,700 instructions, 114 -> 94 VGPRs) or inlined
// (locate / extract at their 64-register budgets: the call's spills cost them 1 %)
template <bool kCold = true>
FMX_HD int32_t wt_inverse_select_from(const DevIndex &ix, uint32_t position, const InvView &v, const Quad &ihq,
                                      int32_t &rank_out, bool &exact_out) {
    const uint32_t block_index = position & ((1u << v.bsl) - 1u);
    exact_out = true;
    if (ihq.x & kInvRun) {  // WFBB:1329-1355: the stored symbol is already masked to 8 bits (WFBB:1332, Q1)
        exact_out = (ihq.x & kInvMasked) == 0;
        rank_out = (int32_t)ihq.z + (int32_t)block_index;
        return (int32_t)ihq.y;
    }
    if (ihq.x & kInvSlow)
        return kCold ? wt_inverse_select_route_rare(ix, position, rank_out, exact_out)
                     : wt_inverse_select_reference_route(ix, position, v, rank_out, exact_out);
    const RrrView &rv = v.rv;
    const NodeRec *nodes = reinterpret_cast<const NodeRec *>(ix.base + ((uint64_t)rv.off_bits << 3)) + ihq.z;
    uint32_t node_b = ihq.y;
    int32_t pos = (int32_t)(ihq.x & 0xffffffu) + (int32_t)block_index;
    int32_t node_rank = (int32_t)block_index;
    Quad nq = ld_quad(nodes), cell = {0, 0, 0, 0};
    cell = bv_load_cell(ix.base, rv, pos);
    FMX_PIN_QUAD(nq);
    FMX_PIN_QUAD(cell);
    FMX_NO_UNROLL
    for (;;) {  // ends: the flattener / validate_blob guarantee children lie behind their parents
        bool bit;
        const int32_t rank1 = bv_rank1_access_cell(rv, cell, pos, bit) - (int32_t)node_b;  // WFBB:1389-1393
        node_rank = bit ? rank1 : node_rank - rank1;                                        // WFBB:1435-1470
        const uint32_t lo = bit ? nq.z : nq.x, hi = bit ? nq.w : nq.y;
        const uint32_t idx = lo & 0xffffu;
        if (idx == 0) {  // leaf: {symbol, folded superblock rank + rank at block start} (WFBB:1495-1533)
            rank_out = (int32_t)hi + node_rank;
            return (int32_t)(lo >> 16);
        }
        node_b = hi >> 8;
        pos = (int32_t)((lo >> 16) | ((hi & 0xffu) << 16)) + node_rank;
        nq = ld_quad(nodes + idx);
        cell = bv_load_cell(ix.base, rv, pos);
        FMX_PIN_QUAD(nq);
        FMX_PIN_QUAD(cell);
    }
}

template <bool kCold = true>
FMX_HD int32_t wt_inverse_select_folded(const DevIndex &ix, const uint16_t *inv, uint32_t position, int32_t &rank_out,
                                        int32_t &bsl_out, bool &exact_out) {
    (void)inv;
    const InvView v = wt_inv_view(ix, position >> 20, inv);
    bsl_out = v.bsl;
    const Quad ihq = ld_quad(wt_inv_hdr_ptr(ix, v, position));
    return wt_inverse_select_from<kCold>(ix, position, v, ihq, rank_out, exact_out);
}

FMX_HD int32_t wt_inverse_select(const DevIndex &ix, const uint16_t *inv, uint32_t position, int32_t &rank_out) {
    int32_t bsl;
    bool exact;
    const int32_t c = wt_inverse_select_folded(ix, inv, position, rank_out, bsl, exact);
    rank_out -= fm_c_or_zero(ix, c);
    return c;
}

// ---- FmIndex helpers -----------------------------------------------------------------------

FMX_HD int32_t fm_map(const DevIndex &ix, uint16_t ch) { return ix.char2code[ch]; }  // getOrDefault(ch, 0) FM:457

// one LF-step of locate / extract: c = BWT[row-1]; row' = C[c] + rank_c(BWT, row)  (FM:532-535, 597-599).
// The reference calls inverseSelect(row-1) and then rank(row, c).  inverseSelect already yields
// rank_c(row-1) (WFBB:1529-1535), and rank(row, c) == rank_c(row-1) + 1 whenever rank() takes its main path
// through the SAME block with the symbol it actually holds: i.e. row-1 and row share a block (row is not a
// block boundary), the position is in range, and the symbol was not altered by the 8-bit mask of run blocks
// (Q1, WFBB:1332).  Only then is the second call skipped; every other case runs rank() as the reference
// does, so all of its quirks (next-block path, Q3) are preserved.  tests/test_fused_lf.py checks the
// equivalence exhaustively on quirk-heavy inputs.
template <bool kCold = true>
FMX_HD int32_t fm_lf_finish(const DevIndex &ix, const uint16_t *inv, int32_t row, int32_t c, int32_t rank_before,
                            int32_t bsl, bool exact_symbol, int &status, bool &suspect) {
    const bool same_block = ((uint32_t)row & ((1u << bsl) - 1u)) != 0 && (uint32_t)row <= ix.wt_size;
    // a run block whose symbol is >= 256 reports a masked symbol: rank(row, masked c) must really be evaluated
    if (!exact_symbol) suspect = true;  // Q1
    if (same_block && exact_symbol) return rank_before + 1;
    if (kCold) return wt_rank_folded_rare(ix, (uint32_t)row, c, status, suspect);
    return wt_rank_folded_t<false>(ix, inv, (uint32_t)row, c, status, suspect);
}
// kWin — what a walk's code knows about the window directory at COMPILE time: kWinAsk = look at ix.win (both routes in the body:
// the boundary kernels, the host simulation's default), kWinNever = an index without one (the tree walk alone: the kernels'
// bodies of round 4), kWinAlways = an index with one: window cell / entry and nothing else — no tree walk, no call, none of
// their registers (k_locate_walk / k_extract are instantiated for kWinNever and kWinAlways and the launcher picks by ix.win).
enum : int { kWinAsk = 0, kWinNever = 1, kWinAlways = 2, kWinFlat = 3 };  // (kWinFlat: an index whose directory has the flat form)
// the directory's form a walk instantiated for kWin may meet
#define FMX_FORM_OF(KWIN) ((KWIN) == kWinFlat ? kFormFlat : ((KWIN) == kWinAlways ? kFormCells : kFormAsk))
template <bool kCold = true, int kWin = kWinAsk>
FMX_HD int32_t fm_lf_step(const DevIndex &ix, const uint16_t *inv, int32_t row, int32_t &c_out, int &status,
                          bool &suspect) {
    const uint32_t p = (uint32_t)(row - 1);
    if (p >= ix.wt_size) {  // never on a well-formed index (its rows are 1..length); superBlockHeaderItems[superBlockId] would throw (WFBB:1310)
        status = ST_JAVA_AIOOBE;
        c_out = 0;
        return 0;
    }
    int32_t rank_before;
    int32_t bsl_i;
    bool exact_symbol;
    if (kWin == kWinAlways || kWin == kWinFlat || (kWin == kWinAsk && ix.win)) {
        // the window of p: {symbol, next row} from one sector, or from the position's entry behind it — no tree walk
        int32_t wc = 0, next = 0;
        bool sampled;
        (void)win_step<true, false, FMX_FORM_OF(kWin)>(ix, p, wc, next, sampled, status, suspect);
        c_out = wc;
        return next;
    }
    const int32_t c = (int32_t)(int16_t)wt_inverse_select_folded<kCold>(ix, inv, p, rank_before, bsl_i, exact_symbol);  // C[c] + rank
    c_out = c;
    return fm_lf_finish<kCold>(ix, inv, row, c, rank_before, bsl_i, exact_symbol, status, suspect);
}
template <bool kCold = true, int kWin = kWinAsk>
FMX_HD int32_t fm_lf_step(const DevIndex &ix, const uint16_t *inv, int32_t row, int32_t &c_out, int &status) {
    bool suspect = false;
    return fm_lf_step<kCold, kWin>(ix, inv, row, c_out, status, suspect);
}
// TWO independent LF-steps of one lane, their loads in flight together (round 3: extractUntilBoundary fetches the sample
// intervals left and right of a position — two walks that know nothing of each other).  A chain whose flag is off is left
// alone.  Both chains take the common path jointly — InvHdr, then per level {NodeRec, cell} — so that a level costs ONE round
// trip for the two of them; whatever leaves that path (a block on the reference's route, a step that crosses a block
// boundary or meets a quirk) finishes with the very functions fm_lf_step calls.  Same rows, symbols, statuses as two calls.
struct LfChain {
    int32_t row;   // in: SA row; out: the row before it in text order
    int32_t c;     // out: the symbol
    bool on;
};
template <int kWin = kWinAsk>
FMX_HD void fm_lf_step2(const DevIndex &ix, const uint16_t *inv, LfChain &a, LfChain &b, int &status, bool &suspect) {
    // per chain: p = row - 1, the block's view and InvHdr
    const uint32_t pa = (uint32_t)(a.row - 1), pb = (uint32_t)(b.row - 1);
    bool la = a.on, lb = b.on;
    if (la && pa >= ix.wt_size) {  // as fm_lf_step: a row no well-formed index produces
        status = ST_JAVA_AIOOBE;
        a.row = 0;
        a.c = 0;
        la = false;
    }
    if (lb && pb >= ix.wt_size) {
        status = ST_JAVA_AIOOBE;
        b.row = 0;
        b.c = 0;
        lb = false;
    }
    if (!la && !lb) return;
    if (kWin == kWinAlways || kWin == kWinFlat || (kWin == kWinAsk && ix.win)) {  // the windows of both positions first (one sector each, requested together), then the entries of class-3
                   // positions (together as well); a chain they answer is done
        int32_t wca_c = 0, wcb_c = 0, nexta = 0, nextb = 0;
        bool sampled;
        if (kWin == kWinFlat || (kWin != kWinAlways && ix.win_flat)) {  // (the flat form: both words requested together)
            const uint32_t fa = win_flat_load(ix, la ? pa : 0u), fb = win_flat_load(ix, lb ? pb : 0u);
            if (la) {
                win_flat_from(ix, fa, wca_c, nexta, sampled, status, suspect);
                a.c = wca_c;
                a.row = nexta;
            }
            if (lb) {
                win_flat_from(ix, fb, wcb_c, nextb, sampled, status, suspect);
                b.c = wcb_c;
                b.row = nextb;
            }
            return;
        }
        uint32_t ra = 0, rb = 0, oa = 0, ob = 0;
        const WinCell wca = win_load(ix, la ? pa : 0u, ra), wcb = win_load(ix, lb ? pb : 0u, rb);
        const bool ha = win_inv_from(wca, ra, wca_c, nexta, sampled, oa), hb = win_inv_from(wcb, rb, wcb_c, nextb, sampled, ob);
        const bool ea = la && !ha, eb = lb && !hb;
        if (ea || eb) {
            const uint64_t va = win_other_load(ix, ea ? oa : 0u), vb = win_other_load(ix, eb ? ob : 0u);
            if (ea) win_other_from(ix, va, wca_c, nexta, status, suspect);
            if (eb) win_other_from(ix, vb, wcb_c, nextb, status, suspect);
        }
        if (la) {
            a.c = wca_c;
            a.row = nexta;
        }
        if (lb) {
            b.c = wcb_c;
            b.row = nextb;
        }
        return;
    }
    const InvView va = wt_inv_view(ix, la ? pa >> 20 : 0u, inv), vb = wt_inv_view(ix, lb ? pb >> 20 : 0u, inv);
    Quad iha = {0, 0, 0, 0}, ihb = {0, 0, 0, 0};
    if (la) iha = ld_quad(wt_inv_hdr_ptr(ix, va, pa));
    if (lb) ihb = ld_quad(wt_inv_hdr_ptr(ix, vb, pb));
    FMX_PIN_QUAD(iha);
    FMX_PIN_QUAD(ihb);
    int32_t rank_a = 0, rank_b = 0, ca = 0, cb = 0;
    bool exact_a = true, exact_b = true;
    // which chains walk node records (the others are answered by their InvHdr, or take the reference's route below)
    bool wa = la && !(iha.x & (kInvRun | kInvSlow)), wb = lb && !(ihb.x & (kInvRun | kInvSlow));
    const uint32_t bia = pa & ((1u << va.bsl) - 1u), bib = pb & ((1u << vb.bsl) - 1u);
    // (a chain that does not walk — its InvHdr holds other things in these fields — reads the section's first record and
    // the vector's first cell: loads that are issued, never used)
    const NodeRec *na = reinterpret_cast<const NodeRec *>(ix.base + ((uint64_t)va.rv.off_bits << 3)) + (wa ? iha.z : 0u);
    const NodeRec *nb = reinterpret_cast<const NodeRec *>(ix.base + ((uint64_t)vb.rv.off_bits << 3)) + (wb ? ihb.z : 0u);
    uint32_t node_b_a = iha.y, node_b_b = ihb.y;
    int32_t pos_a = wa ? (int32_t)(iha.x & 0xffffffu) + (int32_t)bia : 0, pos_b = wb ? (int32_t)(ihb.x & 0xffffffu) + (int32_t)bib : 0;
    int32_t nr_a = (int32_t)bia, nr_b = (int32_t)bib;
    // Lockstep: the loads of a level — {NodeRec, cell} of both chains — are issued back to back and waited for ONCE; all the
    // arithmetic of a level sits between that wait and the next four loads.  (Interleaving "use A, load A, use B, load B"
    // made every use wait for the other chain's fresh loads as well: the two walks ran one after the other.)
    Quad nqa = {0, 0, 0, 0}, cella = {0, 0, 0, 0}, nqb = {0, 0, 0, 0}, cellb = {0, 0, 0, 0};
    if (wa) {
        nqa = ld_quad(na);
        cella = bv_load_cell(ix.base, va.rv, pos_a);
    }
    if (wb) {
        nqb = ld_quad(nb);
        cellb = bv_load_cell(ix.base, vb.rv, pos_b);
    }
    FMX_NO_UNROLL
    while (wa || wb) {  // ends: children lie behind their parents (flattener / validate_blob), as in wt_inverse_select_from
        FMX_PIN_QUAD(nqa);
        FMX_PIN_QUAD(cella);
        FMX_PIN_QUAD(nqb);
        FMX_PIN_QUAD(cellb);
        uint32_t idx_a = 0, idx_b = 0;
        if (wa) {
            bool bit;
            const int32_t rank1 = bv_rank1_access_cell(va.rv, cella, pos_a, bit) - (int32_t)node_b_a;  // WFBB:1389-1393
            nr_a = bit ? rank1 : nr_a - rank1;                                                      // WFBB:1435-1470
            const uint32_t lo = bit ? nqa.z : nqa.x, hi = bit ? nqa.w : nqa.y;
            idx_a = lo & 0xffffu;
            if (idx_a == 0) {  // leaf (WFBB:1495-1533)
                rank_a = (int32_t)hi + nr_a;
                ca = (int32_t)(lo >> 16);
                wa = false;
            } else {
                node_b_a = hi >> 8;
                pos_a = (int32_t)((lo >> 16) | ((hi & 0xffu) << 16)) + nr_a;
            }
        }
        if (wb) {
            bool bit;
            const int32_t rank1 = bv_rank1_access_cell(vb.rv, cellb, pos_b, bit) - (int32_t)node_b_b;
            nr_b = bit ? rank1 : nr_b - rank1;
            const uint32_t lo = bit ? nqb.z : nqb.x, hi = bit ? nqb.w : nqb.y;
            idx_b = lo & 0xffffu;
            if (idx_b == 0) {
                rank_b = (int32_t)hi + nr_b;
                cb = (int32_t)(lo >> 16);
                wb = false;
            } else {
                node_b_b = hi >> 8;
                pos_b = (int32_t)((lo >> 16) | ((hi & 0xffu) << 16)) + nr_b;
            }
        }
        if (wa) {
            nqa = ld_quad(na + idx_a);
            cella = bv_load_cell(ix.base, va.rv, pos_a);
        }
        if (wb) {
            nqb = ld_quad(nb + idx_b);
            cellb = bv_load_cell(ix.base, vb.rv, pos_b);
        }
    }
    // run blocks (the InvHdr holds the answer; WFBB:1329-1355, the symbol masked to 8 bits: Q1) and blocks on the reference's route
    if (la && (iha.x & kInvRun)) {
        exact_a = (iha.x & kInvMasked) == 0;
        rank_a = (int32_t)iha.z + (int32_t)bia;
        ca = (int32_t)iha.y;
    } else if (la && (iha.x & kInvSlow)) {
        ca = wt_inverse_select_route_rare(ix, pa, rank_a, exact_a);
    }
    if (lb && (ihb.x & kInvRun)) {
        exact_b = (ihb.x & kInvMasked) == 0;
        rank_b = (int32_t)ihb.z + (int32_t)bib;
        cb = (int32_t)ihb.y;
    } else if (lb && (ihb.x & kInvSlow)) {
        cb = wt_inverse_select_route_rare(ix, pb, rank_b, exact_b);
    }
    if (la) {
        a.c = (int32_t)(int16_t)ca;
        a.row = fm_lf_finish(ix, inv, a.row, a.c, rank_a, va.bsl, exact_a, status, suspect);  // FM:532-535
    }
    if (lb) {
        b.c = (int32_t)(int16_t)cb;
        b.row = fm_lf_finish(ix, inv, b.row, b.c, rank_b, vb.bsl, exact_b, status, suspect);
    }
}

// One cell of the window directory (layout: "window directory" above), made from the index's OWN steps: `ix` must not carry a
// directory itself (every value below comes from fm_lf_step over the tree, the reference's routes included).
//   1. the step from row p + 1 for every position p of the window; a position is a candidate if the step raised no status and
//      no `suspect` and its symbol (behind the int16 cast of FM:532) is in [0, kWinNone);
//   2. the three most frequent candidate symbols become the classes (ties: the one met first);
//   3. a class's count is the row its FIRST position's step arrived at, less one, and every further position of the class must
//      have arrived at that plus its number in the window — else the class is dropped (its positions become class 3).
// `out` = the 16 words of the cell, word 4 (the first win_other entry) left 0; returns the number of class-3 positions (entries).
FMX_HD uint32_t win_build_cell(const DevIndex &ix, uint32_t w, uint32_t *out) {
    const uint64_t ws64 = (uint64_t)w * kWinW;
    const uint32_t ws = (uint32_t)ws64;
    const uint32_t n = ws64 >= ix.wt_size ? 0u : (ix.wt_size - ws < kWinW ? ix.wt_size - ws : kWinW);
    uint16_t sym[kWinW];
    int32_t next[kWinW];
    uint32_t plane[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    RrrView sv = rrr_view_from(Quad{ix.sampled.off_rec, ix.sampled.off_bits, (uint32_t)ix.sampled.length, (uint32_t)ix.sampled.total_ones});
    bv_bind(sv, ix, nullptr);
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t p = ws + j;
        int status = ST_OK;
        bool suspect = false;
        int32_t c = 0;
        next[j] = fm_lf_step<true, kWinNever>(ix, nullptr, (int32_t)(p + 1u), c, status, suspect);
        sym[j] = (status == ST_OK && !suspect && c >= 0 && (uint32_t)c < kWinNone) ? (uint16_t)c : (uint16_t)kWinNone;
        if ((int32_t)p < sv.length) {
            int st = ST_OK;
            if (bv_access(ix.base, sv, (int32_t)p, st) && st == ST_OK) plane[2][j >> 5] |= 1u << (j & 31u);
        }
    }
    uint32_t best_sym[3] = {kWinNone, kWinNone, kWinNone}, best_cnt[3] = {0, 0, 0};
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t c = sym[j];
        if (c == kWinNone) continue;
        bool seen = false;
        for (uint32_t i = 0; i < j && !seen; ++i) seen = sym[i] == c;
        if (seen) continue;
        uint32_t cnt = 1;
        for (uint32_t i = j + 1; i < n; ++i) cnt += sym[i] == c ? 1u : 0u;
        for (int k = 0; k < 3; ++k) {  // insertion, descending; a tie stays behind the earlier symbol
            if (cnt > best_cnt[k]) {
                for (int t = 2; t > k; --t) {
                    best_cnt[t] = best_cnt[t - 1];
                    best_sym[t] = best_sym[t - 1];
                }
                best_cnt[k] = cnt;
                best_sym[k] = c;
                break;
            }
        }
    }
    uint32_t base[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        if (best_sym[k] == kWinNone) continue;
        bool ok = true, first = true;
        int32_t running = 0;
        for (uint32_t j = 0; j < n && ok; ++j) {
            if (sym[j] != (uint16_t)best_sym[k]) continue;
            if (first) {
                running = next[j] - 1;
                base[k] = (uint32_t)running;
                first = false;
            }
            ok = next[j] == running + 1;
            ++running;
        }
        if (!ok) {
            best_sym[k] = kWinNone;
            base[k] = 0;
        }
    }
    uint32_t others = 0;
    for (uint32_t j = 0; j < kWinW; ++j) {
        uint32_t k = 3;
        if (j < n && sym[j] != kWinNone)
            for (uint32_t t = 0; t < 3; ++t)
                if (best_sym[t] == sym[j]) k = t;
        if (k & 1u) plane[0][j >> 5] |= 1u << (j & 31u);
        if (k & 2u) plane[1][j >> 5] |= 1u << (j & 31u);
        if (k == 3u && j < n) ++others;
    }
    // words 5..15 as one string: plane t from bit 16 + 112 t on
    uint32_t words[16] = {base[0], base[1], base[2], best_sym[0] | (best_sym[1] << 16), 0, best_sym[2], 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t t = 0; t < 3; ++t)
        for (uint32_t j = 0; j < kWinW; ++j)
            if ((plane[t][j >> 5] >> (j & 31u)) & 1u) {
                const uint32_t bit = 16u + kWinW * t + j;
                words[5 + (bit >> 5)] |= 1u << (bit & 31u);
            }
    for (int i = 0; i < 16; ++i) out[i] = words[i];
    return others;
}
// The win_other entries of window w (cell = its 16 words as win_build_cell made them): one per position of class 3, in position
// order, from entry `first` on — what fm_lf_step hands back for the position's row (win_other_make).  Writes `first` into the
// cell's word 4; returns the number of entries that carry a status or `suspect` (statistics: a handful per index), bit 31 set if
// some step's answer does not fit an entry (win_other_fits) or — four-byte form — the slots for the answers that are more than a
// row have run out.
// Four-byte form (entry4): full / full_cap / full_count = the eight-byte slots, how many there are, how many are taken (an atomic
// counter on the device; FMX_WIN_TAKE_SLOT).  An entry is the row alone exactly when the search gives the symbol back.
#ifndef FMX_WIN_TAKE_SLOT
#if defined(__HIP_DEVICE_COMPILE__)
#define FMX_WIN_TAKE_SLOT(counter) atomicAdd((counter), 1u)
#else
#define FMX_WIN_TAKE_SLOT(counter) ((*(counter))++)
#endif
#endif
FMX_HD uint32_t win_build_other(const DevIndex &ix, uint32_t w, uint32_t *cell_words, uint32_t first, uint16_t *entries,
                                bool entry4 = false, uint64_t *full = nullptr, uint32_t full_cap = 0, uint32_t *full_count = nullptr) {
    const uint64_t ws64 = (uint64_t)w * kWinW;
    const uint32_t ws = (uint32_t)ws64;
    const uint32_t n = ws64 >= ix.wt_size ? 0u : (ix.wt_size - ws < kWinW ? ix.wt_size - ws : kWinW);
    cell_words[4] = first;
    WinCell cell;
    memcpy(&cell, cell_words, 64);
    uint32_t at = first, unclean = 0;
    for (uint32_t j = 0; j < n; ++j) {
        int32_t wc, wnext;
        bool sampled;
        uint32_t other;
        if (win_inv_from(cell, j, wc, wnext, sampled, other)) continue;
        int status = ST_OK;
        bool suspect = false;
        int32_t c = 0;
        const int32_t next = fm_lf_step<true, kWinNever>(ix, nullptr, (int32_t)(ws + j + 1u), c, status, suspect);
        if (status != ST_OK || suspect) ++unclean;
        if (!win_other_fits(next, c, status)) unclean |= 0x80000000u;  // (never on a validated index: the caller drops the directory)
        if (!entry4) {
            win_other_store(entries, at++, win_other_make(next, c, status, suspect));
            continue;
        }
        uint32_t e = (uint32_t)next;
        if (status != ST_OK || suspect || next < 0 || win_symbol_of_row(ix, next) != c) {
            const uint32_t slot = FMX_WIN_TAKE_SLOT(full_count);
            if (slot < full_cap)
                full[slot] = win_other_make(next, c, status, suspect);
            else
                unclean |= 0x80000000u;
            e = kWinEntryEscape | slot;
        }
        reinterpret_cast<uint32_t *>(entries)[at++] = e;
    }
    return unclean;
}

// The flat form's word of position p (win_step): the step fm_lf_step takes from row p + 1 over the tree, the position's bit of
// sampledSuffixes, and — where the answer is more than a row, or the row does not fit 30 bits — an eight-byte slot (full /
// full_cap / full_count as in win_build_other).  Returns what win_build_other returns: 1 for an unclean step, bit 31 if the
// answer fits nowhere.
FMX_HD uint32_t win_build_flat(const DevIndex &ix, uint32_t p, uint32_t *flat, uint64_t *full, uint32_t full_cap, uint32_t *full_count) {
    int status = ST_OK;
    bool suspect = false;
    int32_t c = 0;
    const int32_t next = fm_lf_step<true, kWinNever>(ix, nullptr, (int32_t)(p + 1u), c, status, suspect);
    uint32_t unclean = (status != ST_OK || suspect) ? 1u : 0u;
    if (!win_other_fits(next, c, status)) unclean |= 0x80000000u;
    uint32_t e = (uint32_t)next;
    if (status != ST_OK || suspect || next < 0 || (uint32_t)next > kWinFlatRow || win_symbol_of_row(ix, next) != c) {
        const uint32_t slot = FMX_WIN_TAKE_SLOT(full_count);
        if (slot < full_cap && slot <= kWinFlatRow)
            full[slot] = win_other_make(next, c, status, suspect);
        else
            unclean |= 0x80000000u;
        e = kWinFlatEscape | (slot & kWinFlatRow);
    }
    RrrView sv = rrr_view_from(Quad{ix.sampled.off_rec, ix.sampled.off_bits, (uint32_t)ix.sampled.length, (uint32_t)ix.sampled.total_ones});
    bv_bind(sv, ix, nullptr);
    if ((int32_t)p < sv.length) {
        int st = ST_OK;
        if (bv_access(ix.base, sv, (int32_t)p, st) && st == ST_OK) e |= kWinFlatSampled;
    }
    flat[p] = e;
    return unclean;
}

// IntVector.getValue on the packed `suffixes` / `positions` words (IV:129-143)
FMX_HD int32_t fm_packed_get(const uint32_t *words, int64_t index, int width) {
    return (int32_t)ld_bits(words, (uint64_t)index * (uint32_t)width, width);
}

// positions.getValue((x / sampleRate) + 1) + 1 and the skip count (FM:579-587, 645-653, 705-710)
FMX_HD void fm_seek_after(const DevIndex &ix, int32_t x, int32_t &row, int32_t &skip) {
    const int32_t s = ix.sample_rate;
    const int32_t q = x / s;
    row = fm_packed_get(ix.pos_words, (int64_t)q + 1, ix.bw_positions) + 1;
    skip = s - x % s;
    if (q == ix.n_positions - 2) skip = ix.length - x;
    // (the sample after x lies inside the text: on a well-formed index skip <= length - x already; a damaged sample count
    // must not turn the walk back to x into one of 2^31 steps)
    if (skip > ix.length - x) skip = ix.length - x;
}

// ---- suffix table (DevIndex.suffix_table) ---------------------------------------------------------------------
// The table is grown level by level when an index becomes resident: level 1 = the characters (cumulativeCounts), level
// j + 1 = every string of level j with every character put in front of it — by the very step k_count runs (FM:469-470 over
// the index's own rank(), quirks included) — keeping what has a non-empty interval and raised no status.  A slot IS the
// state of the backward search after a pattern's last k characters.
// one string of level j (`parent`: its key holds j codes) with code c in front: false = not kept
FMX_HD bool fm_suffix_extend(const DevIndex &ix, const SuffixSlot &parent, int depth, int32_t c, int key_bits, SuffixSlot &child) {
    int status = ST_OK;
    const int32_t s2 = wt_rank_folded(ix, ix.inv_global, parent.start, c, status);  // FM:469
    const int32_t e2 = wt_rank_folded(ix, ix.inv_global, parent.end, c, status);    // FM:470
    if (status != ST_OK || s2 >= e2) return false;
    child.key = parent.key | ((uint64_t)(uint32_t)c << (depth * key_bits));
    child.start = (uint32_t)s2;
    child.end = (uint32_t)e2;
    return true;
}
// Home slot of a key of `len` codes (2 .. suffix_chars; shorter strings leave the key's upper codes 0, which no tabulated
// string has anywhere).  Strings that differ only in the low 6 bits of their FIRST character's code (the one a pattern
// consumes last of the table's characters) share a group of 64 consecutive slots — the lanes of a wave, neighbours in suffix
// order, read the same few lines.  Inside its group a string sits at (those 6 bits) XOR (6 further bits of the hash of the
// rest): a bijection per group, so the strings of one group never collide with each other, and a character that is frequent
// in the text does not fill "its" slot column of every group (round 3's table, addressed by the bare 6 bits, had to be made
// 20 times larger than its strings to keep that column half empty).
FMX_HD uint32_t fm_suffix_home(const DevIndex &ix, uint64_t key, int len) {
    const int top = (len - 1) * ix.suffix_key_bits;  // where the first character's code sits
    const uint64_t low6 = (key >> top) & (kSuffixGroup - 1);
    const uint64_t rest = key & ~((uint64_t)(kSuffixGroup - 1) << top);
    const uint64_t hash = rest * kSuffixHashMul;
    const uint32_t group = (uint32_t)(hash >> ix.suffix_shift);
    const uint32_t turn = (uint32_t)(hash >> (ix.suffix_shift - kSuffixGroupLog2)) & (kSuffixGroup - 1);
    return (group * kSuffixGroup + ((uint32_t)low6 ^ turn)) & ix.suffix_mask;
}
// A pattern whose trailing `len` codes spell `key`: where the search stands after them, if the table says so.
// Returns false when the loop has to run from the first character.
FMX_HD bool fm_suffix_lookup(const DevIndex &ix, uint64_t key, int len, int32_t &start, int32_t &end, int32_t &back) {
    uint32_t h = fm_suffix_home(ix, key, len);
    Quad q = ld_quad(ix.suffix_table + h);  // the home slot answers nearly every lookup (the table is at most 0.7 full)
    FMX_PIN_QUAD(q);
    uint64_t k = (uint64_t)q.x | ((uint64_t)q.y << 32);
    // (ends at a free slot: the table always has free slots in every column; the bound is for a damaged table)
    for (uint32_t probe = 0; k != key && k != kSuffixEmpty && probe < ix.suffix_mask / kSuffixGroup; ++probe) {
        h = (h + kSuffixGroup) & ix.suffix_mask;
        q = ld_quad(ix.suffix_table + h);
        k = (uint64_t)q.x | ((uint64_t)q.y << 32);
    }
    if (k != key) return false;
    start = (int32_t)q.z;
    end = (int32_t)q.w;
    back = len - 1;
    return true;
}
// the key of a pattern's last `len` characters from its codes (code_at(0) = the last character); false: a code of 0
template <class CodeAt>
FMX_HD bool fm_suffix_key(const DevIndex &ix, CodeAt code_at, int len, uint64_t &key) {
    key = 0;
    bool known = true;
    for (int j = 0; j < len; ++j) {
        const uint32_t cj = (uint32_t)code_at(j);
        known = known && cj != 0;
        key |= (uint64_t)cj << (j * ix.suffix_key_bits);
    }
    // (eight codes of 255 / four of 65,535 spell the free slot's mark: that one string takes the loop)
    return known && key != kSuffixEmpty;
}
// how many of a pattern's m trailing characters the table may answer: every length from 2 to suffix_chars is tabulated
FMX_HD int fm_suffix_len(const DevIndex &ix, int32_t m) { return m < ix.suffix_chars ? (int)m : ix.suffix_chars; }

// FM:526-548 for one hit: SA row i = start + 1 + k; LF-walk until a sampled row.
// Returns the text position; *distance = number of LF-steps walked.
template <int kWin = kWinAsk>
FMX_HD int32_t fm_locate_hit(const DevIndex &ix, const uint16_t *inv, int32_t start, int32_t k, int32_t &distance,
                             int &status) {
    int32_t j = start + 1 + k;  // FM:527-529
    distance = 0;
    RrrView sv = rrr_view_from(Quad{ix.sampled.off_rec, ix.sampled.off_bits, (uint32_t)ix.sampled.length, (uint32_t)ix.sampled.total_ones});
    bv_bind(sv, ix, inv);
    // Every round polls sampledSuffixes.access(j - 1) (FM:531) and, if the row is not sampled, runs
    // inverseSelect(j - 1) (FM:532): the bitmap cell and the block's InvHdr depend on j alone and are requested together.
    Quad scell = {0, 0, 0, 0};
    // Every sample_rate-th text position is sampled, so a walk takes < sampleRate steps — unless a quirk of the reference
    // derails it (Q1: a run block's symbol masked to 8 bits sends the walk to another row, from where it goes on to THAT
    // row's next sample; the reference has no bound at all).  256 such stretches are out of reach for a well-formed index
    // and still end a walk over a damaged one in milliseconds instead of `length` steps.
    const int64_t stretches = (int64_t)ix.sample_rate * 256;
    const int32_t walk_limit = (int32_t)(stretches < 4096 ? 4096 : (stretches < (int64_t)ix.length ? stretches : (int64_t)ix.length));
    for (;;) {
        const int32_t p = j - 1;
        if (p < 0 || p >= sv.length) {  // RrrVector.access throws (RRR:316-323)
            status = ST_JAVA_AIOOBE;
            break;
        }
        if (kWin == kWinAlways || kWin == kWinFlat || (kWin == kWinAsk && ix.win != nullptr)) {
            // the window of p holds the row's sampled bit as well: a step of a walk is ONE sector where the symbol is one of the
            // window's classes, or two (the bitmap's own cell is fetched once, for the rank behind the loop)
            if ((uint32_t)p >= ix.wt_size) {  // (p < the bitmap's length == the tree's size on every image validate_blob lets through)
                status = ST_JAVA_AIOOBE;
                break;
            }
            int32_t c = 0, next = 0;
            bool sampled_row, suspect = false;
            if (win_step<false, true, FMX_FORM_OF(kWin)>(ix, (uint32_t)p, c, next, sampled_row, status, suspect)) {  // (locate never reads the symbol)
                scell = ld_quad(bv_cell_ptr(ix.base, sv, (uint32_t)p));
                FMX_PIN_QUAD(scell);
                break;
            }
            j = next;  // (the step fm_lf_step took over the tree when the directory was grown)
        } else {
            // (p < length == the wavelet tree's size: validate_model / validate_blob)
            const InvView v = wt_inv_view(ix, (uint32_t)p >> 20, inv);
            Quad ihq = ld_quad(wt_inv_hdr_ptr(ix, v, (uint32_t)p));
            scell = ld_quad(bv_cell_ptr(ix.base, sv, (uint32_t)p));
            FMX_PIN_QUAD(ihq);
            FMX_PIN_QUAD(scell);
            bool sampled_row;
            (void)bv_rank1_access_cell(sv, scell, p, sampled_row);
            if (sampled_row) break;
            int32_t rank_before;
            bool exact, suspect = false;
            const int32_t c = (int32_t)(int16_t)wt_inverse_select_from<false>(ix, (uint32_t)p, v, ihq, rank_before, exact);
            j = fm_lf_finish<false>(ix, inv, j, c, rank_before, v.bsl, exact, status, suspect);  // FM:532-535
        }
        ++distance;
        if (distance > walk_limit) {  // bounds the walk on a damaged index (see walk_limit above)
            status = ST_JAVA_AIOOBE;
            break;
        }
    }
    // FM:541 sampledSuffixes.rankOnes(j): row j - 1 is sampled and its cell is at hand — rankOnes(j) = rankOnes(j - 1) + 1
    // (also at j == length, where rankOnes saturates to the total)
    int32_t r;
    if (status == ST_OK)
        r = bv_rank1_after_set_bit(sv, scell, j - 1) - 1;
    else {
        // (a walk that ended in a status — the reference throws — reports no position: the read only has to stay inside
        // `suffixes`, whose total_ones entries validate_model / validate_blob guarantee; rankOnes saturates at that total)
        r = bv_rank1(ix.base, sv, j) - 1;
        if (r < 0) r = 0;
    }
    return fm_packed_get(ix.suffix_words, r, ix.bw_suffixes) + distance;  // FM:538-542
}

// One hit's walk over the window directory in INSTALMENTS (k_locate_walk_c: a wave walks until its longest walk meets a sampled
// row — twice the average — so the workgroup packs the walks still under way into fewer waves now and then; the state that
// travels is this).  fm_locate_steps_win: up to `budget` steps; true = the walk is over (row j - 1 is sampled, or a status ended
// it).  fm_locate_finish_win: FM:538-542 for a finished walk.  Together: fm_locate_hit<kWinAlways>, step for step.
struct WalkState {
    int32_t j, distance;
    int status;
};
// (fm_locate_hit: what bounds a walk over a damaged index)
FMX_HD int32_t fm_walk_limit(const DevIndex &ix) {
    const int64_t stretches = (int64_t)ix.sample_rate * 256;
    return (int32_t)(stretches < 4096 ? 4096 : (stretches < (int64_t)ix.length ? stretches : (int64_t)ix.length));
}
template <int kForm = kFormAsk>
FMX_HD bool fm_locate_steps_win(const DevIndex &ix, WalkState &w, int32_t budget, int32_t walk_limit) {
    FMX_NO_UNROLL
    for (int32_t n = 0; n < budget; ++n) {
        const int32_t p = w.j - 1;
        if (p < 0 || p >= ix.sampled.length || (uint32_t)p >= ix.wt_size) {  // RrrVector.access throws (RRR:316-323)
            w.status = ST_JAVA_AIOOBE;
            return true;
        }
        int32_t c = 0, next = 0;
        bool sampled_row, suspect = false;
        if (win_step<false, true, kForm>(ix, (uint32_t)p, c, next, sampled_row, w.status, suspect)) return true;  // FM:531 (locate never reads the symbol)
        w.j = next;  // (the step fm_lf_step took over the tree when the directory was grown)
        if (++w.distance > walk_limit) {  // bounds the walk on a damaged index
            w.status = ST_JAVA_AIOOBE;
            return true;
        }
    }
    return false;
}
FMX_HD int32_t fm_locate_finish_win(const DevIndex &ix, const uint16_t *inv, const WalkState &w) {
    RrrView sv = rrr_view_from(Quad{ix.sampled.off_rec, ix.sampled.off_bits, (uint32_t)ix.sampled.length, (uint32_t)ix.sampled.total_ones});
    bv_bind(sv, ix, inv);
    int32_t r;
    if (w.status == ST_OK) {  // row j - 1 is sampled: rankOnes(j) = rankOnes(j - 1) + 1 (FM:541)
        const Quad scell = ld_quad(bv_cell_ptr(ix.base, sv, (uint32_t)(w.j - 1)));
        r = bv_rank1_after_set_bit(sv, scell, w.j - 1) - 1;
    } else {  // (a walk that ended in a status reports no position: the read only has to stay inside `suffixes`)
        r = bv_rank1(ix.base, sv, w.j) - 1;
        if (r < 0) r = 0;
    }
    return fm_packed_get(ix.suffix_words, r, ix.bw_suffixes) + w.distance;  // FM:538-542
}

// up to four characters of one aligned 8-byte group of a destination row (mask: which of them): one store when all four are there
FMX_HD void fm_flush_chars(uint16_t *group_at, uint64_t group, uint32_t mask) {
    if (mask == 0xfu) {
        memcpy(__builtin_assume_aligned(group_at, 8), &group, 8);
        return;
    }
    for (uint32_t k = 0; k < 4; ++k)
        if (mask & (1u << k)) group_at[k] = (uint16_t)(group >> (16u * k));
}
// FM:564-608.  Returns the reference's return value (0 when an exception status is set).
template <int kWin = kWinAsk>
FMX_HD int32_t fm_extract(const DevIndex &ix, const uint16_t *inv, int32_t start, int32_t stop, uint16_t *dest,
                          int32_t dst_len, int32_t offset, int32_t &steps, int &status) {
    steps = 0;
    if (!ix.enable_extract) {
        status = ST_NOT_ENABLED;  // FM:566-568
        return 0;
    }
    if (start < 0) {
        status = ST_POS_NEGATIVE;  // FM:570-572
        return 0;
    }
    if (stop >= ix.length) {
        status = ST_STOP_TOO_LONG;  // FM:574-576
        return 0;
    }
    if (stop / ix.sample_rate + 1 < 0) {
        status = ST_JAVA_AIOOBE;  // negative IntVector index
        return 0;
    }
    int32_t row, skip;
    fm_seek_after(ix, stop, row, skip);
    const int32_t range = stop - start;
    if (dst_len - offset < range) {
        status = ST_DEST_TOO_SMALL;  // FM:591-593
        return 0;
    }
    int32_t remaining = range, distance = 0;
    // The characters arrive last to first.  A 2-byte store per character was a third of the kernel's memory requests (round 5:
    // with the window directory a step is ~1.2 sector reads, and the chip serves ~56 G requests/s, reads and writes alike): four
    // characters are gathered per aligned 8 bytes of the row and leave as ONE store; a group the range covers only in part (the
    // row's other characters are the caller's) leaves character by character.
    uint64_t group = 0;
    uint32_t group_mask = 0;
    uint16_t *group_at = dest;  // the group's first character
    while (remaining > 0) {  // FM:596-606
        int32_t c;
        row = fm_lf_step<false, kWin>(ix, inv, row, c, status);
        ++steps;
        if (distance >= skip) {
            const int32_t idx = remaining - 1 + offset;
            if (idx < 0 || idx >= dst_len) {
                status = ST_JAVA_AIOOBE;
                fm_flush_chars(group_at, group, group_mask);
                return 0;
            }
            uint16_t *at = dest + idx;
            const uint32_t slot = (uint32_t)(reinterpret_cast<uintptr_t>(at) >> 1) & 3u;
            group |= (uint64_t)fm_char_of(ix, c) << (16u * slot);
            group_mask |= 1u << slot;
            group_at = at - slot;
            --remaining;
            if (slot == 0u || remaining == 0) {
                fm_flush_chars(group_at, group, group_mask);
                group = 0;
                group_mask = 0;
            }
        }
        ++distance;
    }
    return range;
}

// ---- right part of extractUntilBoundary / extractUntilBoundaryRight (FM:692-758 / FM:853-921) ----------
// The reference fetches the text right of `from` in +4-char chunks; chunk t re-seeks the ISA sample after
// from+4t and walks back skip+4 LF-steps, i.e. ~(s/2+4)/4 steps per character.  The per-character control
// logic (boundary detection, capacity exception, write position, return value) only depends on the
// characters themselves, so it is kept verbatim in `boundary_chunk_char` and fed either
//   * literally (fm_boundary_right_literal: the reference's own seek + walk per chunk), or
//   * from a per-lane buffer holding one whole sample interval fetched with ONE walk of <= s steps
//     (fm_boundary_right_blocks) — same characters, ~s/8 times fewer LF-steps.  If any step of such a walk
//     reports `suspect` (a quirk path of the wavelet tree, where the reference's result depends on its own
//     access pattern), the query is redone literally.
struct RightState {
    int32_t final_pos, up_pos, aux, ret;
    bool done;  // returned 0 or threw
};
// body of the inner loop for one emitted character with code c (FM:721-741 / FM:882-904)
FMX_HD void boundary_chunk_char(const DevIndex &ix, int mode, int32_t c, int32_t mapped_boundary, uint16_t *dest,
                                int32_t dst_len, int32_t offset, int32_t down_len, RightState &r, int &status,
                                bool writer = true) {
    if (c == mapped_boundary) {
        if (r.up_pos == 0) {  // the first char was a boundary: return 0 (FM:725-728)
            r.ret = 0;
            r.done = true;
            return;
        }
        r.final_pos = r.up_pos;
    }
    const int32_t w = (mode == 0) ? offset + down_len + r.up_pos : offset + r.up_pos;
    if (w >= dst_len) {  // FM:732-737 / FM:893-898
        status = ST_DOES_NOT_FIT;
        r.aux = w;
        r.done = true;
        return;
    }
    if (mode == 0) {
        if (w < 0) {
            status = ST_JAVA_AIOOBE;
            r.done = true;
            return;
        }
        if (writer) dest[w] = fm_char_of(ix, c);  // FM:738-739
        --r.up_pos;
    } else if (r.up_pos > 0) {  // range is (from, boundary], FM:899-902
        if (w - 1 < 0) {
            status = ST_JAVA_AIOOBE;
            r.done = true;
            return;
        }
        if (writer) dest[w - 1] = fm_char_of(ix, c);
        --r.up_pos;
    }
}

FMX_HD int32_t fm_boundary_right_literal(const DevIndex &ix, const uint16_t *inv, int mode, int32_t from,
                                         int32_t mapped_boundary, uint16_t *dest, int32_t dst_len, int32_t offset,
                                         int32_t down_len, int32_t &steps, int &status, int32_t &aux) {
    const int32_t step = 4;
    RightState r = {-1, 0, 0, 0, false};
    int32_t times_up = 1;
    while (r.final_pos == -1) {
        const int32_t prev_from = from;
        from += step;
        if (from > ix.length - 1) from = ix.length - 1;
        int32_t remaining = from - prev_from;
        r.up_pos = (times_up - 1) * step + remaining - 1;
        int32_t row, skip;
        fm_seek_after(ix, from, row, skip);
        int32_t distance = 0;
        while (remaining > 0) {
            int32_t c;
            row = fm_lf_step(ix, inv, row, c, status);
            ++steps;
            if (distance >= skip) {
                boundary_chunk_char(ix, mode, c, mapped_boundary, dest, dst_len, offset, down_len, r, status);
                if (r.done) {
                    aux = r.aux;
                    return r.ret;
                }
                --remaining;
            }
            ++distance;
        }
        if (from == ix.length - 1) {  // FM:745-752 / FM:908-915
            r.final_pos = (mode == 0) ? ((r.up_pos < 0) ? 1 : r.up_pos + from - prev_from) : r.up_pos + from - prev_from;
            break;
        }
        ++times_up;
    }
    return (mode == 0) ? down_len + r.final_pos : r.final_pos - 1;  // FM:758 / FM:921
}

// codes of text positions [k*s, min((k+1)*s, length)) into buf[(pos - k*s) * stride] with one walk from the
// ISA sample k+1 (the wrap entry for the last interval, FM:367-369); returns false if any step is suspect
// Where an interval holds the boundary character / the sentinel code 0 (bit j = offset j of the interval; sample rates up to
// 64): noted by the walk that fetches the interval, so that the replay can FIND a line's two ends instead of scanning for them
// character by character (fm_extract_boundary_group's fast path).
struct IntervalMarks {
    uint64_t boundary, zero;
};
FMX_HD void marks_note(IntervalMarks &m, int32_t offset, int32_t c, int32_t mapped_boundary) {
    const uint64_t bit = 1ull << (offset & 63);
    if (c == mapped_boundary) m.boundary |= bit;
    if (c == 0) m.zero |= bit;
}
template <int kWin = kWinAsk>
FMX_HD bool fm_fetch_interval(const DevIndex &ix, const uint16_t *inv, int32_t k, uint16_t *buf, int64_t stride,
                              int32_t &steps, int &status, IntervalMarks *marks = nullptr, int32_t mapped_boundary = -1) {
    const int32_t s = ix.sample_rate;
    const int64_t top64 = (int64_t)(k + 1) * s;
    const int32_t top = top64 < ix.length ? (int32_t)top64 : ix.length;
    int32_t row = fm_packed_get(ix.pos_words, (int64_t)k + 1, ix.bw_positions) + 1;  // FM:705-706
    bool suspect = false;
    for (int32_t pos = top - 1; pos >= k * s; --pos) {
        int32_t c;
        row = fm_lf_step<true, kWin>(ix, inv, row, c, status, suspect);
        ++steps;
        buf[(pos - k * s) * stride] = (uint16_t)c;
        if (marks) marks_note(*marks, pos - k * s, (int32_t)(uint16_t)c, mapped_boundary);
    }
    return !suspect && status == ST_OK;
}

// two intervals with the two walks interleaved (fm_lf_step2): ka / kb < 0 = nothing to fetch on that side
template <int kWin = kWinAsk>
FMX_HD bool fm_fetch_interval2(const DevIndex &ix, const uint16_t *inv, int32_t ka, uint16_t *bufa, int32_t kb, uint16_t *bufb,
                               int64_t stride, int32_t &steps_a, int32_t &steps_b, int &status, IntervalMarks *marks_a = nullptr,
                               IntervalMarks *marks_b = nullptr, int32_t mapped_boundary = -1) {
    const int32_t s = ix.sample_rate;
    int32_t na = 0, nb = 0;
    LfChain a = {0, 0, false}, b = {0, 0, false};
    if (ka >= 0) {
        const int64_t top64 = (int64_t)(ka + 1) * s;
        na = (top64 < ix.length ? (int32_t)top64 : ix.length) - ka * s;
        a.row = fm_packed_get(ix.pos_words, (int64_t)ka + 1, ix.bw_positions) + 1;  // FM:705-706
    }
    if (kb >= 0) {
        const int64_t top64 = (int64_t)(kb + 1) * s;
        nb = (top64 < ix.length ? (int32_t)top64 : ix.length) - kb * s;
        b.row = fm_packed_get(ix.pos_words, (int64_t)kb + 1, ix.bw_positions) + 1;
    }
    bool suspect = false;
    const int32_t n = na > nb ? na : nb;
    for (int32_t i = 0; i < n; ++i) {  // step i writes offset n_x - 1 - i of its interval
        a.on = i < na;
        b.on = i < nb;
        fm_lf_step2<kWin>(ix, inv, a, b, status, suspect);
        if (a.on) bufa[(int64_t)(na - 1 - i) * stride] = (uint16_t)a.c;
        if (b.on) bufb[(int64_t)(nb - 1 - i) * stride] = (uint16_t)b.c;
        if (marks_a && a.on) marks_note(*marks_a, na - 1 - i, (int32_t)(uint16_t)a.c, mapped_boundary);
        if (marks_b && b.on) marks_note(*marks_b, nb - 1 - i, (int32_t)(uint16_t)b.c, mapped_boundary);
    }
    steps_a += na;
    steps_b += nb;
    return !suspect && status == ST_OK;
}

// same result as fm_boundary_right_literal; `buf` holds sample_rate codes per lane (element i at buf[i*stride])
FMX_HD int32_t fm_boundary_right_blocks(const DevIndex &ix, const uint16_t *inv, int mode, int32_t from,
                                        int32_t mapped_boundary, uint16_t *dest, int32_t dst_len, int32_t offset,
                                        int32_t down_len, int32_t &steps, int &status, int32_t &aux, uint16_t *buf,
                                        int64_t stride, bool &clean) {
    const int32_t s = ix.sample_rate;
    const int32_t step = 4;
    RightState r = {-1, 0, 0, 0, false};
    int32_t times_up = 1;
    int32_t cur_k = -1;
    clean = true;
    while (r.final_pos == -1) {
        const int32_t prev_from = from;
        from += step;
        if (from > ix.length - 1) from = ix.length - 1;
        const int32_t remaining = from - prev_from;
        r.up_pos = (times_up - 1) * step + remaining - 1;
        // the chunk's codes in position order: prev_from .. from-1 (<= 4; may straddle two sample intervals)
        int32_t c4[4] = {0, 0, 0, 0};
        for (int32_t i = 0; i < remaining; ++i) {
            const int32_t pos = prev_from + i;
            const int32_t k = pos / s;
            if (k != cur_k) {
                if (!fm_fetch_interval(ix, inv, k, buf, stride, steps, status)) {
                    clean = false;
                    return 0;
                }
                cur_k = k;
            }
            c4[i] = buf[(pos - k * s) * stride];
        }
        for (int32_t i = remaining - 1; i >= 0; --i) {  // the reference emits from-1 first, prev_from last
            const int32_t c = (i == 3) ? c4[3] : (i == 2) ? c4[2] : (i == 1) ? c4[1] : c4[0];
            boundary_chunk_char(ix, mode, c, mapped_boundary, dest, dst_len, offset, down_len, r, status);
            if (r.done) {
                aux = r.aux;
                return r.ret;
            }
        }
        if (from == ix.length - 1) {
            r.final_pos = (mode == 0) ? ((r.up_pos < 0) ? 1 : r.up_pos + from - prev_from) : r.up_pos + from - prev_from;
            break;
        }
        ++times_up;
    }
    return (mode == 0) ? down_len + r.final_pos : r.final_pos - 1;
}

// scratch == nullptr (or sample_rate > scratch capacity): literal form only
FMX_HD int32_t fm_boundary_right(const DevIndex &ix, const uint16_t *inv, int mode, int32_t from,
                                 int32_t mapped_boundary, uint16_t *dest, int32_t dst_len, int32_t offset,
                                 int32_t down_len, int32_t &steps, int &status, int32_t &aux, uint16_t *scratch,
                                 int64_t scratch_stride) {
    if (scratch) {
        bool clean;
        int32_t steps2 = 0, aux2 = 0;
        int status2 = ST_OK;
        const int32_t ret = fm_boundary_right_blocks(ix, inv, mode, from, mapped_boundary, dest, dst_len, offset,
                                                     down_len, steps2, status2, aux2, scratch, scratch_stride, clean);
        steps += steps2;
        if (clean) {
            status = status2;
            aux = aux2;
            return ret;
        }
    }
    return fm_boundary_right_literal(ix, inv, mode, from, mapped_boundary, dest, dst_len, offset, down_len, steps,
                                     status, aux);
}

// FM:640-759 (mode 0), FM:772-831 (mode 1), FM:844-922 (mode 2).  `mapped_boundary` = code of the
// boundary char (FM:658).  *aux = N of "Currently extracted: N".
// scratch: per-lane buffer of sample_rate codes for the accelerated right part (nullptr = literal form).
FMX_HD int32_t fm_extract_boundary(const DevIndex &ix, const uint16_t *inv, int mode, int32_t from,
                                   int32_t mapped_boundary, uint16_t *dest, int32_t dst_len, int32_t offset,
                                   int32_t &steps, int &status, int32_t &aux, uint16_t *scratch = nullptr,
                                   int64_t scratch_stride = 1) {
    steps = 0;
    aux = 0;
    if (mode == 1) ++from;  // FM:774
    // checkBoundsForExtraction FM:610-626
    if (!ix.enable_extract) {
        status = ST_NOT_ENABLED;
        return 0;
    }
    if (from < 0) {
        status = ST_POS_NEGATIVE;
        return 0;
    }
    if (from >= ix.length) {
        status = ST_POS_TOO_LONG;
        return 0;
    }
    if (dst_len == 0) {
        status = ST_DEST_SIZE_ZERO;
        return 0;
    }
    if (mapped_boundary == 0) {
        status = ST_NO_BOUNDARY;  // FM:659-661, 792-794, 849-851
        return 0;
    }
    int32_t down_len = 0;
    if (mode != 2) {  // left part, FM:645-690 / FM:777-828
        int32_t row, skip;
        fm_seek_after(ix, from, row, skip);
        int32_t down_pos = dst_len - 1;
        int32_t remaining = dst_len, distance = 0;
        while (mode == 1 || remaining > 0) {
            int32_t c;
            row = fm_lf_step(ix, inv, row, c, status);
            ++steps;
            // unreachable on a well-formed index: at most `skip` <= sampleRate uncounted steps (they may wrap around
            // a text shorter than the sample rate), then the walk ends at the boundary or the sentinel
            if (steps > ix.length + ix.sample_rate) {
                status = ST_JAVA_AIOOBE;
                return 0;
            }
            if (distance >= skip) {
                if (c == mapped_boundary || c == 0) break;  // FM:674-680
                if (down_pos < 0) {                          // destination[-1]
                    status = ST_JAVA_AIOOBE;
                    return 0;
                }
                dest[down_pos--] = fm_char_of(ix, c);  // FM:682
                --remaining;
                if (mode == 1 && down_pos == offset) {  // FM:816-821
                    status = ST_DOES_NOT_FIT;
                    aux = dst_len - offset;
                    return 0;
                }
            }
            ++distance;
        }
        down_len = dst_len - (down_pos + 1);  // FM:689
        // System.arraycopy(dest, downPos+1, dest, offset, downLen) — memmove semantics, FM:690
        if (offset < 0 || offset + down_len > dst_len) {
            status = ST_JAVA_AIOOBE;
            return 0;
        }
        if (down_len > 0 && offset != down_pos + 1) {
            if (offset < down_pos + 1)
                for (int32_t t = 0; t < down_len; ++t) dest[offset + t] = dest[down_pos + 1 + t];
            else
                for (int32_t t = down_len - 1; t >= 0; --t) dest[offset + t] = dest[down_pos + 1 + t];
        }
        if (mode == 1) return down_len;  // FM:830
    }
    return fm_boundary_right(ix, inv, mode, from, mapped_boundary, dest, dst_len, offset, down_len, steps, status, aux,
                             scratch, scratch_stride);
}

// ---- group-cooperative extractUntilBoundary ------------------------------------------------------------
// G lanes serve one query.  All of them run the same replay of the reference's control flow (cheap, and it
// keeps the group converged); the text comes from a window of G sample intervals in LDS, refilled on demand
// with each lane walking a DIFFERENT interval (the walks are independent: every interval starts at its own
// ISA sample).  Lane 0 of the group is the only writer of `dest`.  One serial chain of hundreds of LF-steps
// per query becomes a few rounds of <= sample_rate steps, with G times more lanes in flight.
// Host build: G = 1, g = 0 (no cross-lane traffic).
template <int G>
struct TextWindow {
    uint16_t *buf;    // slot i (interval k_lo + i), offset j at buf[j * row_stride + i * slot_stride]
    int64_t row_stride, slot_stride;
    int32_t s;        // sample rate
    int32_t k_lo;     // first interval in the window (-1: empty)
    int32_t n;        // intervals in the window: G after a refill, G / 2 after the first fill of both windows
    int32_t g;        // this lane's index in the group
    int32_t steps;    // LF-steps this lane walked
    bool suspect;     // some walk of the group touched a quirk path -> redo the query literally
    // what the lane's walk of the window's FIRST fill saw: interval marks_k (-1: none) holds the boundary / the sentinel where
    // `marks` says (later refills do not maintain them: the fast replay runs right after the first fill)
    int32_t mapped_boundary;
    int32_t marks_k;
    IntervalMarks marks;
    // ... and of the SECOND round of a fill in rounds (window_fill_round): the lane's second interval of this window
    int32_t marks2_k = -1;
    IntervalMarks marks2 = {0, 0};
};

template <int G>
FMX_HD bool group_any(bool v) {
#if defined(__HIPCC__)
    int x = v ? 1 : 0;
    for (int m = 1; m < G; m <<= 1) x |= __shfl_xor(x, m);
    return x != 0;
#else
    return v;
#endif
}
template <int G>
FMX_HD int32_t group_max(int32_t v) {
#if defined(__HIPCC__)
    for (int m = 1; m < G; m <<= 1) {
        const int32_t o = __shfl_xor(v, m);
        v = o > v ? o : v;
    }
#endif
    return v;
}
template <int G>
FMX_HD int32_t group_min(int32_t v) {
#if defined(__HIPCC__)
    for (int m = 1; m < G; m <<= 1) {
        const int32_t o = __shfl_xor(v, m);
        v = o < v ? o : v;
    }
#endif
    return v;
}
template <int G>
FMX_HD int32_t group_sum(int32_t v) {
#if defined(__HIPCC__)
    for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m);
#endif
    return v;
}

// Interval k lives in slot k mod G of the window's buffer (a ring), so that a window can GROW by the slots it does not
// use yet without moving what it holds.
template <int G>
FMX_HD uint16_t *window_slot(const TextWindow<G> &w, int32_t k) {
    return w.buf + (int64_t)(k & (G - 1)) * w.slot_stride;
}
// make interval k resident.  Adjacent to a window that still has free slots: the window grows towards k by as many
// intervals as it has free slots (what it holds — the interval of `from` above all — stays).  Otherwise the window is
// replaced; dir > 0: [k, k+G), dir < 0: [k-G+1, k].  Every lane that fetches walks ONE interval.
template <int G, int kWin = kWinAsk>
FMX_HD void window_refill(const DevIndex &ix, const uint16_t *inv, TextWindow<G> &w, int32_t k, int dir) {
    const int32_t k_max = (ix.length - 1) / w.s;  // last interval that holds text (incl. the sentinel)
    int32_t mine = -1;
    if (w.k_lo >= 0 && w.n < G && dir < 0 && k == w.k_lo - 1) {  // grow to the left
        int32_t e = G - w.n;
        if (e > k + 1) e = k + 1;  // (not below interval 0)
        if (w.g < e) mine = k - w.g;
        w.k_lo -= e;
        w.n += e;
    } else if (w.k_lo >= 0 && w.n < G && dir > 0 && k == w.k_lo + w.n) {  // grow to the right
        const int32_t e = G - w.n;
        if (w.g < e) mine = k + w.g;
        w.n += e;
    } else {
        int32_t lo = dir > 0 ? k : k - (G - 1);
        if (lo < 0) lo = 0;
        w.k_lo = lo;
        w.n = G;
        mine = lo + w.g;
        w.marks_k = (mine >= 0 && mine <= k_max) ? mine : -1;
        w.marks.boundary = w.marks.zero = 0;
    }
    bool ok = true;
    if (mine >= 0 && mine <= k_max) {
        int status = ST_OK;
        ok = fm_fetch_interval<kWin>(ix, inv, mine, window_slot<G>(w, mine), w.row_stride, w.steps, status,
                               mine == w.marks_k ? &w.marks : nullptr, w.mapped_boundary);
    }
    if (group_any<G>(!ok)) w.suspect = true;
}

// The default first fill — G intervals on each side — with every lane's two walks interleaved (fm_fetch_interval2): what
// window_refill(wl, k0, -1) followed by window_refill(wr, k0 + 1, +1) would fetch, in half the round trips.
template <int G, int kWin = kWinAsk>
FMX_HD void window_fill_pairs(const DevIndex &ix, const uint16_t *inv, TextWindow<G> &wl, TextWindow<G> &wr, int32_t k0,
                              bool want_right) {
    const int32_t k_max = (ix.length - 1) / wl.s;
    int32_t lo = k0 - (G - 1);
    if (lo < 0) lo = 0;
    wl.k_lo = lo;
    wl.n = G;
    int32_t mine_l = lo + wl.g, mine_r = -1;
    if (mine_l > k_max) mine_l = -1;
    if (want_right) {
        wr.k_lo = k0 + 1;
        wr.n = G;
        mine_r = k0 + 1 + wr.g;
        if (mine_r > k_max) mine_r = -1;
    }
    int status = ST_OK;
    wl.marks_k = mine_l;
    wr.marks_k = mine_r;
    wl.marks.boundary = wl.marks.zero = wr.marks.boundary = wr.marks.zero = 0;
    const bool ok = fm_fetch_interval2<kWin>(ix, inv, mine_l, window_slot<G>(wl, mine_l < 0 ? 0 : mine_l), mine_r,
                                       window_slot<G>(wr, mine_r < 0 ? 0 : mine_r), wl.row_stride, wl.steps, wr.steps, status,
                                       &wl.marks, &wr.marks, wl.mapped_boundary);
    if (group_any<G>(!ok)) {
        wl.suspect = true;
        if (want_right) wr.suspect = true;
    }
}

// THE FILL IN ROUNDS (round 6; G = 4, extractUntilBoundary both ways): a line of log text needs 3.1 of the 8 sample intervals the
// first fill above fetches, and three quarters of the lines end inside the TWO intervals on each side of `from`.  Round 0 fetches
// those four — ONE walk per lane: lanes 0, 1 the intervals k0 - 1, k0 of the left window, lanes 2, 3 the intervals k0 + 1, k0 + 2
// of the right one — and the marked replay is tried; a group whose line does not end inside them runs round 1: the two intervals
// further out on each side, into the windows' free slots, their marks beside the first round's (TextWindow.marks2).  After
// round 1 the windows hold what window_fill_pairs fetches, marks of all eight intervals included: the same replay, the same rows.
template <int G, int kWin = kWinAsk>
FMX_HD void window_fill_round(const DevIndex &ix, const uint16_t *inv, TextWindow<G> &wl, TextWindow<G> &wr, int32_t k0, int round) {
    static_assert(G == 4, "the fill in rounds is the group of four's");
    const int32_t k_max = (ix.length - 1) / wl.s;
    const bool left_lane = wl.g < 2;
    const int32_t j = wl.g & 1;
    int32_t mine;
    if (round == 0) {
        const int32_t lo = k0 >= 1 ? k0 - 1 : 0;
        wl.k_lo = lo;
        wl.n = k0 - lo + 1;  // [lo, k0]: two intervals, or one at the text's start
        wr.k_lo = k0 + 1;
        wr.n = 2;
        mine = left_lane ? (lo + j <= k0 ? lo + j : -1) : k0 + 1 + j;
    } else {
        const int32_t e = wl.k_lo < 2 ? wl.k_lo : 2;  // the left window grows by two intervals (not below interval 0)
        const int32_t new_lo = wl.k_lo - e;
        mine = left_lane ? (j < e ? new_lo + j : -1) : wr.k_lo + wr.n + j;
        wl.k_lo = new_lo;
        wl.n += e;
        wr.n += 2;
    }
    if (mine > k_max) mine = -1;
    IntervalMarks m = {0, 0};
    int status = ST_OK;
    int32_t walked = 0;
    bool ok = true;
    if (mine >= 0) {
        uint16_t *slot = (left_lane ? wl.buf : wr.buf) + (int64_t)(mine & (G - 1)) * wl.slot_stride;
        ok = fm_fetch_interval<kWin>(ix, inv, mine, slot, wl.row_stride, walked, status, &m, wl.mapped_boundary);
    }
    if (left_lane)
        wl.steps += walked;
    else
        wr.steps += walked;
    const IntervalMarks none = {0, 0};
    if (round == 0) {
        wl.marks_k = left_lane ? mine : -1;
        wl.marks = left_lane ? m : none;
        wr.marks_k = left_lane ? -1 : mine;
        wr.marks = left_lane ? none : m;
        wl.marks2_k = wr.marks2_k = -1;
        wl.marks2 = wr.marks2 = none;
    } else {
        wl.marks2_k = left_lane ? mine : -1;
        wl.marks2 = left_lane ? m : none;
        wr.marks2_k = left_lane ? -1 : mine;
        wr.marks2 = left_lane ? none : m;
    }
    if (group_any<G>(!ok)) wl.suspect = wr.suspect = true;
}

template <int G, int kWin = kWinAsk>
FMX_HD int32_t window_code_at(const DevIndex &ix, const uint16_t *inv, TextWindow<G> &w, int32_t pos, int dir) {
    const int32_t k = pos / w.s;
    if (w.k_lo < 0 || k < w.k_lo || k >= w.k_lo + w.n) window_refill<G, kWin>(ix, inv, w, k, dir);
    return window_slot<G>(w, k)[(int64_t)(pos - k * w.s) * w.row_stride];
}

#if !defined(__HIPCC__)
inline long g_marked_replays[3] = {0, 0, 0};
#endif
FMX_HD int fmx_clzll(uint64_t v) { return __builtin_clzll(v); }  // (v != 0)
FMX_HD int fmx_ctzll(uint64_t v) { return __builtin_ctzll(v); }

// The replay of extractUntilBoundary / ...Left / ...Right (FM:640-922) for the common case, WITHOUT walking the text character
// by character: the first fill's walks noted where their intervals hold the boundary and the sentinel (TextWindow.marks), so
// the line's two ends are two bit searches and a group reduction, and what the reference's loops leave in `destination` is
// known in closed form —
//   left part (modes 0 and 1, FM:655-690 / FM:788-828): text(lb, from) is written at the END of destination going down, then
//     moved to `offset`: both copies stay (the temporary one is part of the row the caller sees); mode 1 starts one position
//     further right (FM:774) and returns downLen;
//   right part (modes 0 and 2, FM:692-758 / FM:853-921): whole +4 chunks text[from, E), E = the end of the chunk that holds the
//     first boundary rb >= from (the characters behind rb inside that chunk are written too); mode 0 writes them at offset +
//     downLen and returns downLen + (rb - from); mode 2 writes text(from, E) one slot lower (the character AT `from` is not
//     written, FM:899-902) and returns rb - from - 1.
// The G lanes of the group write those characters side by side (lane g: every G-th one) instead of ONE lane writing while
// all G replay the same loops.  Anything else — an end outside the fetched windows, `from` on a boundary (FM:725-728), a line
// that reaches the text's end (FM:745-752), a destination the line does not fit with room to spare (FM:732-737, 816-821,
// 893-898, and overlapping copies) — returns false and the literal replay below runs as before.  The result of the test is the
// same in every lane of the group.  `from` is the position AFTER mode 1's increment.
template <int G>
FMX_HD bool fm_boundary_replay_marked(const DevIndex &ix, int mode, const TextWindow<G> &wl, const TextWindow<G> &wr, int32_t from,
                                      int32_t k0, uint16_t *dest, int32_t dst_len, int32_t offset, int32_t &ret) {
    const int32_t s = wl.s;
    const int32_t o = from - k0 * s;  // 0 <= o < s <= 64
    const uint64_t below = (1ull << o) - 1ull;
    if (offset < 0) return false;
    int32_t a = 0;  // downLen
    int32_t lb = -1;
    if (mode != 2) {
        // lb: the nearest position below `from` that holds the boundary or the sentinel (FM:674-680); -1 = the text's start
        int32_t cand = -1;
        if (wl.marks_k >= 0 && wl.marks_k <= k0) {
            uint64_t m = wl.marks.boundary | wl.marks.zero;
            if (wl.marks_k == k0) m &= below;
            if (m) cand = wl.marks_k * s + 63 - fmx_clzll(m);
        }
        if (wl.marks2_k >= 0 && wl.marks2_k < k0) {  // (a fill in rounds: the lane's second interval, further left)
            const uint64_t m = wl.marks2.boundary | wl.marks2.zero;
            if (m) {
                const int32_t c2 = wl.marks2_k * s + 63 - fmx_clzll(m);
                cand = c2 > cand ? c2 : cand;
            }
        }
        lb = group_max<G>(cand);
        if (lb < 0 && wl.k_lo != 0) return false;  // the line starts left of the window
        a = from - lb - 1;
        // mode 0 stops at dst_len characters (FM:664), mode 1 raises when the write position reaches `offset` (FM:816-821)
        if (mode == 0 ? a >= dst_len : a >= dst_len - 1 - offset) return false;
    }
    int64_t right_len = 0;
    int32_t rb = 0;
    if (mode != 1) {
        // rb: the first position >= from that holds the boundary
        int32_t rc = INT32_MAX;
        if (wl.marks_k == k0) {
            const uint64_t m = wl.marks.boundary & ~below;
            if (m) rc = k0 * s + fmx_ctzll(m);
        }
        if (wr.marks_k > k0 && wr.marks.boundary) {
            const int32_t r2 = wr.marks_k * s + fmx_ctzll(wr.marks.boundary);
            rc = r2 < rc ? r2 : rc;
        }
        if (wr.marks2_k > k0 && wr.marks2.boundary) {  // (a fill in rounds: the lane's second interval, further right)
            const int32_t r2 = wr.marks2_k * s + fmx_ctzll(wr.marks2.boundary);
            rc = r2 < rc ? r2 : rc;
        }
        rb = group_min<G>(rc);
        if (rb == INT32_MAX || rb <= from) return false;  // not in the windows / `from` itself is the boundary
        const int64_t e = (int64_t)from + 4 * (int64_t)((rb - from) / 4 + 1);  // end of rb's chunk
        if (e >= (int64_t)ix.length - 1) return false;  // the last chunk is clamped / ends the loop by itself (FM:745-752)
        const int32_t k_last = (int32_t)((e - 1) / s);
        if (k_last > k0 && (wr.k_lo < 0 || k_last >= wr.k_lo + wr.n)) return false;
        right_len = e - from;
    }
    // everything fits, and the final characters stay below the left part's temporary copy at the row's end
    // (the capacity check of FM:732-737 / 893-898 looks at offset + downLen + r for every character r of the chunks, also in
    // mode 2, which writes one slot lower)
    const int64_t used = (int64_t)a + right_len;
    if ((int64_t)offset + used > (int64_t)dst_len - a) return false;
    if (mode != 2) {
        uint16_t *tmp = dest + (dst_len - a);
        for (int32_t i = wl.g; i < a; i += G) {
            const int32_t pos = lb + 1 + i, k = pos / s;
            const uint16_t ch = fm_char_of(ix, (int32_t)window_slot<G>(wl, k)[(int64_t)(pos - k * s) * wl.row_stride]);
            tmp[i] = ch;            // FM:682 / 810
            dest[offset + i] = ch;  // FM:690 / 828
        }
    }
    if (mode != 1) {
        // mode 0: text[from + i] at offset + a + i; mode 2: text[from + i], i >= 1, at offset + i - 1
        uint16_t *up = mode == 0 ? dest + offset + a : dest + offset - 1;
        for (int32_t i = wl.g; i < (int32_t)right_len; i += G) {
            if (mode == 2 && i == 0) continue;  // the character AT `from` is not part of (from, boundary]
            const int32_t pos = from + i, k = pos / s;
            const TextWindow<G> &w = k <= k0 ? wl : wr;
            up[i] = fm_char_of(ix, (int32_t)window_slot<G>(w, k)[(int64_t)(pos - k * s) * w.row_stride]);  // FM:738-739 / 899-902
        }
    }
    ret = mode == 0 ? a + (rb - from) : mode == 1 ? a : rb - from - 1;  // FM:758 / 830 / 921
#if !defined(__HIPCC__)
    ++g_marked_replays[mode];  // (host simulation only: the tests make sure this route is the one they exercise)
#endif
    return true;
}

// Same results as fm_extract_boundary (FM:640-922).  `clean` = false: a walk was suspect, nothing can be
// trusted — the caller reruns the query with the literal form.  steps: LF-steps walked by the whole group.
// buf: two windows of G intervals per group (left window, then right window `win_stride` elements further).
// Both windows are fetched up front, at ONE program point, so that all lanes of a wave walk their intervals
// together; later refills (lines longer than a window) happen wherever the replay needs them.
// kMode >= 0: the mode as a compile-time constant (the kernels: one instance per mode, each without the other two's code)
// kDefer: the NARROW first round of a batch (G = 2: two sample intervals on each side of `from`, what most lines need) — a query the
// marked replay cannot finish from those windows is not replayed literally here (its refills would hold the whole wave up) but
// handed back like a suspect one (`clean` false): the caller puts it on a list that the wide form (G = 4) works off.
template <int G, int kMode = -1, int kWin = kWinAsk, bool kDefer = false, bool kRounds = false>
FMX_HD int32_t fm_extract_boundary_group(const DevIndex &ix, const uint16_t *inv, int mode, int32_t from,
                                         int32_t mapped_boundary, uint16_t *dest, int32_t dst_len, int32_t offset,
                                         int32_t &steps, int &status, int32_t &aux, uint16_t *buf, int64_t row_stride,
                                         int64_t slot_stride, int64_t win_stride, int32_t g, bool &clean,
                                         bool pair_walks = false) {
    if (kMode >= 0) mode = kMode;
    steps = 0;
    aux = 0;
    clean = true;
    const bool writer = (g == 0);
    if (mode == 1) ++from;  // FM:774
    if (!ix.enable_extract) {  // checkBoundsForExtraction FM:610-626
        status = ST_NOT_ENABLED;
        return 0;
    }
    if (from < 0) {
        status = ST_POS_NEGATIVE;
        return 0;
    }
    if (from >= ix.length) {
        status = ST_POS_TOO_LONG;
        return 0;
    }
    if (dst_len == 0) {
        status = ST_DEST_SIZE_ZERO;
        return 0;
    }
    if (mapped_boundary == 0) {
        status = ST_NO_BOUNDARY;  // FM:659-661, 792-794, 849-851
        return 0;
    }
    const int32_t s = ix.sample_rate;
    const int32_t k0 = from / s;
    TextWindow<G> wl = {buf, row_stride, slot_stride, s, -1, 0, g, 0, false, mapped_boundary, -1, {0, 0}};               // intervals <= k0
    TextWindow<G> wr = {buf + win_stride, row_stride, slot_stride, s, -1, 0, g, 0, false, mapped_boundary, -1, {0, 0}};  // intervals > k0
    bool filled = false;
    if constexpr (kRounds && G == 4) {
        if (mode == 0 && s <= 64) {
            // the fill in rounds (window_fill_round): the four intervals next to `from` first; the replay decides who needs the rest
            int32_t ret0 = 0;
            window_fill_round<G, kWin>(ix, inv, wl, wr, k0, 0);
            if (!wl.suspect && !wr.suspect && fm_boundary_replay_marked<G>(ix, mode, wl, wr, from, k0, dest, dst_len, offset, ret0)) {
                steps = group_sum<G>(wl.steps + wr.steps);
                return ret0;
            }
            if (!wl.suspect && !wr.suspect) window_fill_round<G, kWin>(ix, inv, wl, wr, k0, 1);
            filled = true;
        }
    }
    if (filled) {
    } else if (pair_walks) {
        window_fill_pairs<G, kWin>(ix, inv, wl, wr, k0, mode != 1);       // the same two windows, a lane's two walks interleaved
    } else {
        window_refill<G, kWin>(ix, inv, wl, k0, -1);                      // [k0-G+1, k0]: the left part and text[from..]
        if (mode != 1) window_refill<G, kWin>(ix, inv, wr, k0 + 1, +1);   // [k0+1, k0+G]
    }
    int32_t ret = 0;
    if (s <= 64 && !wl.suspect && !wr.suspect &&
        fm_boundary_replay_marked<G>(ix, mode, wl, wr, from, k0, dest, dst_len, offset, ret)) {
        steps = group_sum<G>(wl.steps + wr.steps);
        return ret;
    }
    if (kDefer) {
        steps = group_sum<G>(wl.steps + wr.steps);
        clean = false;
        return 0;
    }
    bool finished = false;
    int32_t down_len = 0;
    if (mode != 2) {  // left part (FM:655-690 / FM:788-828): text[from-1], text[from-2], ... until boundary / start
        int32_t down_pos = dst_len - 1;
        int32_t remaining = dst_len;
        for (int32_t pos = from - 1; mode == 1 || remaining > 0; --pos) {
            const int32_t c = pos < 0 ? 0 : window_code_at<G, kWin>(ix, inv, wl, pos, -1);  // before text[0] the walk meets the sentinel
            if (wl.suspect) break;
            if (c == mapped_boundary || c == 0) break;  // FM:674-680
            if (down_pos < 0) {                          // destination[-1]
                status = ST_JAVA_AIOOBE;
                finished = true;
                break;
            }
            if (writer) dest[down_pos] = fm_char_of(ix, c);  // FM:682
            --down_pos;
            --remaining;
            if (mode == 1 && down_pos == offset) {  // FM:816-821
                status = ST_DOES_NOT_FIT;
                aux = dst_len - offset;
                finished = true;
                break;
            }
        }
        if (!finished && !wl.suspect) {
            down_len = dst_len - (down_pos + 1);  // FM:689
            if (offset < 0 || offset + down_len > dst_len) {  // System.arraycopy range check, FM:690
                status = ST_JAVA_AIOOBE;
                finished = true;
            } else {
                if (writer && down_len > 0 && offset != down_pos + 1) {
                    if (offset < down_pos + 1)
                        for (int32_t t = 0; t < down_len; ++t) dest[offset + t] = dest[down_pos + 1 + t];
                    else
                        for (int32_t t = down_len - 1; t >= 0; --t) dest[offset + t] = dest[down_pos + 1 + t];
                }
                if (mode == 1) {
                    ret = down_len;  // FM:830
                    finished = true;
                }
            }
        }
    }
    if (!finished && !wl.suspect && !wr.suspect) {  // right part in +4 chunks (FM:692-758 / FM:853-921)
        const int32_t step = 4;
        RightState r = {-1, 0, 0, 0, false};
        int32_t times_up = 1;
        while (r.final_pos == -1 && !r.done && !wl.suspect && !wr.suspect) {
            const int32_t prev_from = from;
            from += step;
            if (from > ix.length - 1) from = ix.length - 1;
            const int32_t remaining = from - prev_from;
            r.up_pos = (times_up - 1) * step + remaining - 1;
            int32_t c4[4] = {0, 0, 0, 0};
            for (int32_t i = 0; i < remaining; ++i) {
                const int32_t pos = prev_from + i;
                c4[i] = (pos / s <= k0) ? window_code_at<G, kWin>(ix, inv, wl, pos, +1) : window_code_at<G, kWin>(ix, inv, wr, pos, +1);
            }
            if (wl.suspect || wr.suspect) break;
            for (int32_t i = remaining - 1; i >= 0 && !r.done; --i) {  // the reference emits from-1 first
                const int32_t c = (i == 3) ? c4[3] : (i == 2) ? c4[2] : (i == 1) ? c4[1] : c4[0];
                boundary_chunk_char(ix, mode, c, mapped_boundary, dest, dst_len, offset, down_len, r, status, writer);
            }
            if (r.done) break;
            if (from == ix.length - 1) {  // FM:745-752 / FM:908-915
                r.final_pos = (mode == 0) ? ((r.up_pos < 0) ? 1 : r.up_pos + from - prev_from) : r.up_pos + from - prev_from;
                break;
            }
            ++times_up;
        }
        if (r.done) {
            aux = r.aux;
            ret = r.ret;
        } else if (!wl.suspect && !wr.suspect) {
            ret = (mode == 0) ? down_len + r.final_pos : r.final_pos - 1;  // FM:758 / FM:921
        }
    }
    steps = group_sum<G>(wl.steps + wr.steps);
    if (wl.suspect || wr.suspect) {
        clean = false;
        status = ST_OK;
        return 0;
    }
    return ret;
}

}  // namespace fmx
