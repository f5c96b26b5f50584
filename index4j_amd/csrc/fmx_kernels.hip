// fmx_kernels.hip — HIP kernels for gfx950 (MI355X) and their launchers.
//
//   k_count            FmIndex.count            FM:455-474   two lanes per pattern (start / end of the SA interval)
//   k_locate_walk      FmIndex.locate           FM:526-548   one lane per (pattern, hit): LF-walk to a sampled row
//   k_extract          FmIndex.extract          FM:564-608   one lane per query
//   k_extract_boundary extractUntilBoundary{,Left,Right} FM:640-922  one lane per query
//
//   k_order_*          the plan stage of a batch: code words, suffix order (bucket pass + tile-local radix sort)
//   k_wt_*, k_rrr_*    WaveletFixedBlockBoosting / RrrVector as stand-alone structures
//   k_segment_*        merging the answers of a segment set
//
// Workgroups grid-stride over queries; the FM kernels stage the superblock headers (10 KiB) in LDS.  The work is
// bit-level integer gather (no MFMA): throughput comes from tens of thousands of independent dependent-load
// chains in flight, the two lanes of a pattern sharing their sectors (start and end of an interval usually
// fall in the same blocks).
#include <atomic>
#include <cstring>

#include <hip/hip_runtime.h>

#include "fmx_device.hpp"
#include "fmx_plan.hpp"

// This file is compiled twice (Makefile): as it stands for expanded images (kernels and launchers in namespace fmx), and
// with -DFMX_COMPACT=1 -DFMX_KNS=fmxc for COMPACT images (fmx_device.hpp: the bv_* functions decode RrrRecords, every FM
// kernel stages the value-of-offset table in LDS).  The C-ABI layer picks the namespace by the image's flag.
#if !defined(FMX_KNS)
#define FMX_KNS fmx
#endif
// the value-of-offset table of an FM kernel: none in an expanded image, 32 KiB of LDS in a compact one
#if FMX_COMPACT
#define FMX_FM_INV(IX)                                \
    __shared__ uint16_t s_inv_lds[kInvEntries];       \
    stage_inverse_table(s_inv_lds, (IX).inv_global);  \
    const uint16_t *s_inv = s_inv_lds
#else
#define FMX_FM_INV(IX) const uint16_t *s_inv = nullptr /* no RRR vector on an expanded image's path */
#endif

namespace FMX_KNS {
using namespace fmx;
static std::atomic<int> g_code_bits_12{1};  // option "code_bits_12" (plan_code_bits)

// Workgroup size is a template parameter (512 / 1024 threads).  Only the stand-alone RrrVector kernels stage the
// 32 KiB value-of-offset table in LDS.
// FMX_WAVES_PER_EU asks the register allocator for 8 waves per SIMD (<= 64 VGPRs, <= 80 SGPRs): the
// kernels are latency-bound chains of dependent loads, so resident waves are what hides latency.
// (the kernels over compact images carry the record decode: 76-79 VGPRs, 6 waves — no target is forced on them)
#if FMX_COMPACT
#define FMX_KERNEL(BLOCK) __global__ __launch_bounds__(BLOCK)
#else
#define FMX_KERNEL(BLOCK) __global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
// The LF-walk kernels (locate / extract / extractUntilBoundary) carry more state per lane; FMX_WALK_WAVES is the
// occupancy their register budget is sized for (512 / FMX_WALK_WAVES VGPRs per lane).
#ifndef FMX_WALK_WAVES
#define FMX_WALK_WAVES 8
#endif
#if FMX_COMPACT
#define FMX_WALK_KERNEL(BLOCK) __global__ __launch_bounds__(BLOCK)
#else
#define FMX_WALK_KERNEL(BLOCK) \
    __global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FMX_WALK_WAVES, 8)))
#endif
// k_extract: 49.8 ms at a budget for 8 waves, 47.7 ms at 6 (locate -> extract pipeline, tools/bench_pipeline.py)
#ifndef FMX_EXTRACT_WAVES
#define FMX_EXTRACT_WAVES 6
#endif
#define FMX_EXTRACT_KERNEL(BLOCK) \
    __global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FMX_EXTRACT_WAVES, 8)))
// extractUntilBoundary keeps two text windows, the replay state and a walk alive: measured 4.74 / 4.42 / 3.56 ms
// (configs[3]) at budgets for 8 / 6 / 4 waves per SIMD — spilling costs more than the lost occupancy
#ifndef FMX_BOUNDARY_WAVES
#define FMX_BOUNDARY_WAVES 4
#endif
#define FMX_BOUNDARY_KERNEL(BLOCK) \
    __global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FMX_BOUNDARY_WAVES, 8)))

__device__ __forceinline__ void stage_inverse_table(uint16_t *s_inv, const uint16_t *g_inv) {
    const uint4 *src = reinterpret_cast<const uint4 *>(g_inv);
    uint4 *dst = reinterpret_cast<uint4 *>(s_inv);
    for (int i = threadIdx.x; i < kInvEntries * 2 / 16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// FM:455-474 (also the first half of locate, FM:506-523).  Lane 2p computes `start`, lane 2p+1
// computes `end`; they swap results with one DPP-class shuffle per pattern character.
// range_out (nullable): 2 ints per pattern {start, end} for k_locate_walk.
// perm (nullable): processing order — slot q of the grid handles pattern perm[q].  The launcher sorts
// the batch by the patterns' last characters so that the lanes of a wave start their backward search
// in the same SA intervals (same sectors, broadcast loads); results land at the original index.
// The plan stage hands k_count one 64-bit word per pattern with the codes of its trailing characters, the last
// character in the low bits: 8 codes of 8 bits when the alphabet fits (sigma <= 256), else 4 codes of 16 bits.
// 12 bits for alphabets of 257 .. 4,096 codes (round 6): FIVE codes per word instead of four and a suffix table one character
// deeper (60-bit keys) — the shape of the data set the reference's published numbers are quoted on (> 1,000 symbols,
// README.md:291-292).  fmx_code_bits_for (fmx_device.hpp) is the one rule; option "code_bits_12" = 0 gives 16 bits there (A/B).
inline int plan_code_bits(int32_t sigma) { return fmx_code_bits_for(sigma, g_code_bits_12.load() != 0); }

// The header quad and the bit-vector view quad of every superblock (32 bytes each) are staged in LDS when the
// index has at most kSbCacheMax superblocks (335 M symbols): the first stage of every rank / inverseSelect then
// reads LDS instead of HBM, and what depends only on the header is requested one round trip earlier.
constexpr int kSbCacheMax = 320;
__device__ __forceinline__ const Quad *stage_sb_cache(Quad *s_sb, const DevIndex &ix) {
    if (ix.n_sb > kSbCacheMax || ix.n_sb > ix.sb_cache_limit) return nullptr;
    const Quad *src = reinterpret_cast<const Quad *>(ix.sbd);
    for (int i = threadIdx.x; i < 2 * ix.n_sb; i += blockDim.x) s_sb[i] = src[(i >> 1) * 4 + (i & 1) * 2];
    __syncthreads();
    return s_sb;
}
#define FMX_WITH_SB_CACHE(GLOBAL_IX, LOCAL_IX)      \
    __shared__ Quad s_sb[2 * kSbCacheMax];          \
    DevIndex LOCAL_IX = GLOBAL_IX;                  \
    LOCAL_IX.sb_cache = stage_sb_cache(s_sb, GLOBAL_IX)

// cumulativeCounts in LDS for the kernels that read SYMBOLS out of a window directory with four-byte entries (win_symbol_of_row:
// an entry is the row a step arrives at, its symbol the largest c with C[c] < row) — extract and extractUntilBoundary; locate
// never asks.  Alphabets beyond kWinSymbolSearchMax entries keep six-byte entries (or, forced to four, search C where it lies).
__device__ __forceinline__ void stage_c_lds(int32_t *s_c, uint16_t *s_lut, DevIndex &ix) {
    ix.c_lds = nullptr;
    ix.c_lut = nullptr;
    ix.c_lut_shift = 0;
    if (!ix.win || !ix.win_entry4 || ix.n_c > kWinSymbolSearchMax) return;
    for (int i = threadIdx.x; i < ix.n_c; i += blockDim.x) s_c[i] = ix.C[i];
    __syncthreads();
    ix.c_lds = s_c;
    // where a row's search starts: entry b = the largest c with C[c] < b << shift (win_symbol_of_row)
    const int32_t shift = win_lut_shift(ix.length);
    for (int b = threadIdx.x; b <= kWinLutBuckets; b += blockDim.x) {
        const int64_t row = (int64_t)b << shift;
        s_lut[b] = (uint16_t)win_symbol_of_row(ix, row > 0x7fffffff ? 0x7fffffff : (int32_t)row);
    }
    __syncthreads();
    ix.c_lut = s_lut;
    ix.c_lut_shift = shift;
}
#define FMX_WITH_C_LDS(IX, KWIN)                                                  \
    __shared__ int32_t s_c_lds[(KWIN) == kWinNever ? 1 : kWinSymbolSearchMax];    \
    __shared__ uint16_t s_c_lut[(KWIN) == kWinNever ? 1 : kWinLutBuckets + 2];    \
    if ((KWIN) != kWinNever) stage_c_lds(s_c_lds, s_c_lut, IX)

// geometry of the plan kernels' workgroups, measured on configs[1] (codes / scatter kernel, us): 512 x 8: 19.3 / 19.9,
// 256 x 4: 27.3 / 41.2 (four times the workgroups, each zeroing, flushing and scanning all bins), 1024 x 4: 15.7 / 17.5
#ifndef FMX_TILE_THREADS
#define FMX_TILE_THREADS 1024
#endif
#ifndef FMX_TILE_ITEMS
#define FMX_TILE_ITEMS 4
#endif
constexpr int kTileThreads = FMX_TILE_THREADS;
constexpr int kTileItems = FMX_TILE_ITEMS;          // patterns per thread
constexpr int kTile = kTileThreads * kTileItems;    // patterns per workgroup of the plan kernels
constexpr int kCoarseBitsMax = 14;                  // 16,384 LDS bins (64 KiB)

// The (up to) 8 trailing characters of a pattern.  Patterns of >= 8 characters: the 16 bytes [beg + m - 8, beg + m) are
// fetched as four or five ALIGNED dwords (every dword holds at least one byte of the pattern, so nothing outside
// the caller's pages is touched) instead of eight 2-byte loads.  Split in load / decode so that a thread can have
// the loads of several patterns in flight.
struct TailWords {
    uint32_t d0, d1, d2, d3, d4;
    bool odd;
};
__device__ __forceinline__ TailWords pattern_tail_load(const uint16_t *pat, int32_t beg, int32_t m) {
    TailWords t = {0, 0, 0, 0, 0, false};
    if (m >= 8) {
        // (pointer arithmetic on `pat`, not on an integer: the loads stay global_load instead of flat_load)
        const int32_t first = beg + m - 8;
        t.odd = (((uint32_t)(reinterpret_cast<uintptr_t>(pat) >> 1) + (uint32_t)first) & 1u) != 0;
        const uint32_t *w = reinterpret_cast<const uint32_t *>(pat + first - (t.odd ? 1 : 0));
        t.d0 = w[0];
        t.d1 = w[1];
        t.d2 = w[2];
        t.d3 = w[3];
        if (t.odd) t.d4 = w[4];  // starts in the upper half of d0: the eighth character is the lower half of a fifth dword
    }
    return t;
}
// ch[0] = the LAST character
__device__ __forceinline__ void pattern_tail_chars(TailWords t, const uint16_t *pat, int32_t beg, int32_t m,
                                                   uint32_t ch[8]) {
    if (m >= 8) {
        FMX_OPAQUE32(t.d0);  // (keeps the compiler from turning the dwords back into eight 2-byte loads)
        FMX_OPAQUE32(t.d1);
        FMX_OPAQUE32(t.d2);
        FMX_OPAQUE32(t.d3);
        FMX_OPAQUE32(t.d4);
        if (t.odd) {
            t.d0 = (t.d0 >> 16) | (t.d1 << 16);
            t.d1 = (t.d1 >> 16) | (t.d2 << 16);
            t.d2 = (t.d2 >> 16) | (t.d3 << 16);
            t.d3 = (t.d3 >> 16) | (t.d4 << 16);
        }
        ch[7] = t.d0 & 0xffffu;
        ch[6] = t.d0 >> 16;
        ch[5] = t.d1 & 0xffffu;
        ch[4] = t.d1 >> 16;
        ch[3] = t.d2 & 0xffffu;
        ch[2] = t.d2 >> 16;
        ch[1] = t.d3 & 0xffffu;
        ch[0] = t.d3 >> 16;
    } else {
        for (int j = 0; j < 8; ++j) ch[j] = j < m ? (uint32_t)pat[beg + m - 1 - j] : 0u;
    }
}
// the plan's code word of a pattern: codes of its trailing characters, the LAST character in the low bits
// (s_map = LDS copy of the first 256 entries of the character map)
template <int kCodeBits>
__device__ __forceinline__ uint64_t pattern_code_word(const DevIndex &ix, const int16_t *s_map, const uint32_t ch[8],
                                                      int32_t m) {
    constexpr int n_codes = 64 / kCodeBits;
    uint64_t w = 0;
#pragma unroll
    for (int j = 0; j < n_codes; ++j)
        if (j < m) {
            const int32_t c = ch[j] < 256u ? (int32_t)s_map[ch[j]] : fm_map(ix, (uint16_t)ch[j]);
            w |= (uint64_t)(uint32_t)c << (j * kCodeBits);
        }
    return w;
}
// sort key = the first `chars` codes of the word, the last character most significant
__device__ __forceinline__ uint32_t suffix_key(uint64_t word, int code_bits, int chars, int bits) {
    const uint32_t mask = (1u << code_bits) - 1u;
    uint32_t key = 0;
    for (int j = 0; j < chars; ++j) key = (key << bits) | ((uint32_t)(word >> (j * code_bits)) & mask);
    return key;
}

// One record per pattern.  k_plan_codes writes it at the pattern's index with {code word, sort key, length | fine
// bin}; k_plan_fine writes the final order, where the second dword pair is {pattern index, length}.
struct PlanRec {
    uint64_t cw;   // codes of the trailing characters, the LAST character in the low bits
    uint32_t a;    // k_plan_codes: suffix key (first characters of the code word, last character most significant);
                   // final order: index of the pattern in the caller's batch
    uint32_t m;    // the pattern's length in bits 0..21 (kPlanLongPattern: longer — read it from the offsets); from
                   // k_plan_codes also, in bits 22..31, the bin k_plan_fine ranks a window by: the key bits around the
                   // lower end of the coarse bits (two coarse bits, so that neighbouring buckets stay apart, + 8 below)
};
static_assert(sizeof(PlanRec) == 16, "PlanRec");
constexpr uint32_t kPlanLongPattern = 0x3fffffu;
constexpr int kFineBits = 10;  // width of the window-local order of k_plan_fine (counting sort in LDS)
constexpr int kFineThreads = 512;
constexpr int kFineItems = 2;
constexpr int kFineWindow = kFineThreads * kFineItems;  // 1,024 patterns: the window k_plan_fine orders

// recs (nullable): the plan — the batch's records in processing order; lane pair q takes record q (one coalesced
// 16-byte load, no gather).  plan_look_up (mode 2): code of the plan's alphabet -> character, to translate code
// words made with ANOTHER index of a segment set (0 = character absent there).
// kMode: 0 = no plan (the caller's order, characters mapped here), 1 = plan made with this index, 2 = plan made with
// another index of a segment set (code words translated), 3 = such a plan with 16-bit codes (order only)
// The codes of a pattern's characters reach the loop in CHUNKS of up to 8 consecutive characters (counted from the
// pattern's end), code j of the chunk at bits [j * code_bits, (j + 1) * code_bits) of a 128-bit pair.  The first chunk
// of a planned batch is the record's code word; every further chunk is ONE fetch of the pattern's next 16 bytes (aligned
// dwords, pattern_tail_load) mapped through an LDS copy of the character map's first 256 entries — instead of a
// character load and a map lookup (two dependent loads) in front of every pair of ranks.  Patterns of the
// reference's benchmark shape are 8..31 characters long (FmIndexThroughputState.java:76-83).
struct CodeChunk {
    uint64_t lo, hi;
    int32_t base;  // `back` index of the chunk's first code
    int32_t n;     // codes in the chunk
};
template <int kCodeBits>
__device__ __forceinline__ int32_t chunk_code(const CodeChunk &ck, int32_t j) {
    constexpr uint32_t code_mask = (1u << kCodeBits) - 1u;
    const uint32_t pos = (uint32_t)j * (uint32_t)kCodeBits;
    if (kCodeBits == 8) return (int32_t)((uint32_t)(ck.lo >> pos) & code_mask);  // 8 codes of 8 bits: one word
    if (kCodeBits == 12) {  // five codes per word (no code straddles the two)
        const uint32_t jj = (uint32_t)j;
        return (int32_t)((uint32_t)((jj < 5u ? ck.lo : ck.hi) >> ((jj < 5u ? jj : jj - 5u) * 12u)) & code_mask);
    }
    const uint64_t w = pos < 64u ? ck.lo : ck.hi;
    return (int32_t)((uint32_t)(w >> (pos & 63u)) & code_mask);
}
// the chunk that starts `back` characters before the pattern's last one (`beg` = the pattern's first character).
// NOT inlined: a refill happens once per 8 characters, and with its forty-odd instructions and five loads folded into
// k_count the register allocation of the whole kernel got worse (the headline batch, whose patterns never refill, ran
// 1.2 % slower with the code merely present: profiles/r03_experiments.txt).
template <int kCodeBits>
__device__ __forceinline__ CodeChunk chunk_refill_body(const int16_t *__restrict__ char2code, const int16_t *s_map,
                                                       const uint16_t *__restrict__ pat, int32_t beg, int32_t m, int32_t back) {
    const int32_t left = m - back;  // characters not yet consumed (>= 1)
    const TailWords t = pattern_tail_load(pat, beg, left);
    uint32_t ch[8];
    pattern_tail_chars(t, pat, beg, left, ch);
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t c = 0;
        if (j < left) c = (uint32_t)(uint16_t)(ch[j] < 256u ? s_map[ch[j]] : char2code[ch[j]]);
        if (kCodeBits == 12) {  // (chunk_code: five codes per word)
            if (j < 5)
                lo |= (uint64_t)c << (j * 12);
            else
                hi |= (uint64_t)c << ((j - 5) * 12);
            continue;
        }
        const uint32_t pos = (uint32_t)(j * kCodeBits);
        if (pos < 64u)
            lo |= (uint64_t)c << pos;
        else
            hi |= (uint64_t)c << (pos - 64u);
    }
    CodeChunk ck;
    ck.lo = lo;
    ck.hi = hi;
    ck.base = back;
    ck.n = left < 8 ? left : 8;
    return ck;
}
template <int kCodeBits>
__device__ __attribute__((noinline)) CodeChunk chunk_refill(const int16_t *__restrict__ char2code, const int16_t *s_map,
                                                            const uint16_t *__restrict__ pat, int32_t beg, int32_t m,
                                                            int32_t back) {
    return chunk_refill_body<kCodeBits>(char2code, s_map, pat, beg, m, back);
}

// The backward search of ONE pattern by a lane pair (FM:455-474): lane `role` 0 computes `start`, lane 1 `end`.
// kChunks = false: every character the loop will consume has its code in the record's word `ck.lo` (m <= ck.n; the
// common case of a planned batch of short patterns: nothing but shifts in front of a rank).  kChunks = true: further
// chunks are fetched and mapped on the way (chunk_refill), or — kMode 0 / 3 — all of them.
template <int kMode, int kCodeBits, bool kChunks>
__device__ __forceinline__ void count_one(const DevIndex &ix, const uint16_t *s_inv, const int16_t *s_map, const int16_t *s_xlat,
                                          const uint16_t *__restrict__ pat, const int32_t *__restrict__ pat_off, int32_t p,
                                          int32_t m, CodeChunk ck, int role, int32_t &start, int32_t &end, int32_t &back,
                                          int32_t &tabled, int &status) {
    constexpr uint32_t code_mask = (1u << kCodeBits) - 1u;
    constexpr bool translate = kMode == 2;
    // where the pattern starts: only patterns longer than the record's code word need it (one load, requested here,
    // used by the first refill)
    int32_t beg = 0;
    if (kChunks) {
        if (m > ck.n) beg = pat_off[p];
        if (kMode == 3) ck = chunk_refill<kCodeBits>(ix.char2code, s_map, pat, beg, m, 0);  // no code word: the first chunk
    }
    // the plan stage left the codes of the trailing characters (no character load and map lookup in front of every rank)
    int32_t c = chunk_code<kCodeBits>(ck, 0);
    // a foreign code of 0 only says "not in the plan's alphabet": the character itself decides here
    if (translate) c = c ? s_xlat[c] : fm_map(ix, pat[pat_off[p] + m - 1]);
    if (c == 0) return;  // FM:458-460
    start = ix.C[c];
    end = ix.C[c + 1];
    int len = ix.suffix_table ? fm_suffix_len(ix, m) : 0;
    if (ck.n < len) len = ck.n;
    if (len >= 2) {
        // the interval after the last `len` characters is tabulated (the table was grown by this very loop over every string
        // of 2 .. suffix_chars codes that occurs): one 16-byte slot instead of 2 * (len - 1) ranks.  The key is the low bits of
        // the code word where that is written in this index's alphabet; a segment of a set makes it from the translated codes
        // (a code its alphabet lacks is 0: not tabulated, the loop runs and lets the character decide).
        uint64_t key;
        bool known = true;
        if (translate || kCodeBits != ix.suffix_key_bits) {
            known = fm_suffix_key(
                ix,
                [&](int j) {
                    const uint32_t cj = (uint32_t)chunk_code<kCodeBits>(ck, j);
                    return translate ? (uint32_t)(uint16_t)s_xlat[cj] : cj;
                },
                len, key);
        } else {
            const int bits = len * kCodeBits;
            key = bits >= 64 ? ck.lo : (ck.lo & ((1ull << bits) - 1ull));
            // The key's TOP code must be a real one: a 0 there (an unknown character) would spell the tabulated string that
            // is one character SHORTER — whose interval is not this pattern's, which ends at that character (FM:466-468).
            // (A 0 further down finds nothing: no tabulated string has one below its top.)
            known = (uint32_t)(key >> (bits - kCodeBits)) != 0u && key != kSuffixEmpty;
        }
        if (known) (void)fm_suffix_lookup(ix, key, len, start, end, back);
        tabled = back;  // characters whose rank evaluations the table answered
    }
    bool first_chunk = true;  // the record's word (its codes are the PLAN's: translated in mode 2)
    while (start < end && back + 1 < m) {  // FM:464
        ++back;
        if (kChunks) {
            if (back - ck.base >= ck.n) {
                ck = chunk_refill<kCodeBits>(ix.char2code, s_map, pat, beg, m, back);
                first_chunk = false;
            }
            c = chunk_code<kCodeBits>(ck, back - ck.base);
        } else {
            c = (int32_t)((uint32_t)(ck.lo >> (back * kCodeBits)) & code_mask);
        }
        if (translate && first_chunk) c = c ? s_xlat[c] : fm_map(ix, pat[pat_off[p] + m - 1 - back]);
        if (c == 0) {  // FM:466-468: ends the search before this character's two ranks
            start = end = 0;
            --back;
            break;
        }
        // (the window directory — an LF directory: fmx_device.hpp — has nothing for a rank of a GIVEN symbol; a form of it that had
        // was measured: a wave's lanes split into those it answers and those that walk the tree, and the wave pays for both —
        // profiles/r05_experiments.txt 7)
        const int32_t mine = wt_rank_folded(ix, s_inv, (uint32_t)(role ? end : start), c, status);  // C[c] + rank
        const int32_t other = __shfl_xor(mine, 1);
        start = role ? other : mine;  // FM:469
        end = role ? mine : other;    // FM:470
    }
}

#if defined(FMX_DIAG_TIMELINE)
// Diagnostic build only (make EXTRA_DEFS=-DFMX_DIAG_TIMELINE; tools/k_count_timeline.py): when every workgroup of the
// last k_count launch started and ended (100 MHz wall clock), and where it ran (XCC id, CU id).
constexpr int kDiagGroups = 8192;
__device__ unsigned long long g_diag_timeline[3 * kDiagGroups];
__device__ __forceinline__ void diag_begin() {
    if (threadIdx.x == 0 && blockIdx.x < kDiagGroups) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_diag_timeline[3 * blockIdx.x] = wall_clock64();
        g_diag_timeline[3 * blockIdx.x + 1] = 0;
        g_diag_timeline[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
    }
}
__device__ __forceinline__ void diag_end() {
    if ((threadIdx.x & 63) == 0 && blockIdx.x < kDiagGroups) atomicMax(&g_diag_timeline[3 * blockIdx.x + 1], wall_clock64());
}
#else
__device__ __forceinline__ void diag_begin() {}
__device__ __forceinline__ void diag_end() {}
#endif

// The records {code word, pattern index (~0: no pattern), length} of a workgroup's lane pairs handed out again by length (k_count):
// a counting sort in LDS on min(length, 63), longest first.  Both lanes of a pair pass the same record and get the same one
// back.  A real call (as chunk_refill): inlined, its registers cost the batches of one length — which never get here — 2 %.
template <int kBlock>
__device__ __forceinline__ Quad regroup_records_body(uint32_t *s_bin, Quad *s_rec, Quad mine) {
    const int32_t m = (int32_t)mine.w;
    const int bin = mine.z == 0xffffffffu ? 0 : (m < 0 ? 0 : (m > 63 ? 63 : m));
    if (threadIdx.x < 64) s_bin[threadIdx.x] = 0;
    __syncthreads();
    uint32_t rank_in_bin = 0;
    if ((threadIdx.x & 1) == 0) rank_in_bin = atomicAdd(&s_bin[bin], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the 64 bins by the first wave, the longest patterns first
        const uint32_t v = s_bin[63 - threadIdx.x];
        uint32_t incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if ((int)threadIdx.x >= d) incl += t;
        }
        s_bin[63 - threadIdx.x] = incl - v;
    }
    __syncthreads();
    if ((threadIdx.x & 1) == 0) s_rec[s_bin[bin] + rank_in_bin] = mine;
    __syncthreads();
    return s_rec[threadIdx.x >> 1];
}
template <int kBlock>
__device__ __noinline__ Quad regroup_records(uint32_t *s_bin, Quad *s_rec, Quad mine) {
    return regroup_records_body<kBlock>(s_bin, s_rec, mine);
}

#ifndef FMX_COUNT_WAVES
#define FMX_COUNT_WAVES 8
#endif
#if FMX_COMPACT
#define FMX_COUNT_KERNEL(BLOCK) __global__ __launch_bounds__(BLOCK)
#else
#define FMX_COUNT_KERNEL(BLOCK) __global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FMX_COUNT_WAVES, 8)))
#endif
template <int kBlock, int kMode, int kCodeBits>
FMX_COUNT_KERNEL(kBlock) void k_count(DevIndex ix_global, const uint16_t *__restrict__ pat,
                                                  const int32_t *__restrict__ pat_off,
                                                  const PlanRec *__restrict__ recs, int32_t n,
                                                  int32_t *__restrict__ counts, int32_t *__restrict__ lf_steps,
                                                  int32_t *__restrict__ status_out, int32_t *__restrict__ range_out,
                                                  const int32_t *__restrict__ plan_look_up, int32_t plan_sigma,
                                                  int steps_mode, int regroup, const uint32_t *__restrict__ plan_mixed,
                                                  uint32_t plan_epoch, int halve_uniform, const int32_t *__restrict__ redo_list,
                                                  uint32_t *__restrict__ redo_count) {
    // kMode 4 = LIST mode: the patterns are redo_list[0 .. *redo_count) — those k_count_lean met on a route it does not carry —
    // searched as in mode 0 (the caller's order, every route inlined).  The list is usually empty: nothing is staged then.  The last
    // workgroup to finish leaves the counters zero for the next launch.
    if (kMode == 4) {
        n = (int32_t)*reinterpret_cast<volatile uint32_t *>(redo_count);
        if (n == 0) return;
    }
    // kCodeBits: width of a code in the record's word and in the chunks — 8 when the alphabet fits (8 codes per word),
    // else 16 (the plan's alphabet in modes 1 / 2, this index's own in modes 0 / 3)
    constexpr int kPairs = kBlock / 2;
    // A PLANNED batch of ONE length runs on HALF the grid (halve_uniform; the plan's flag says which kind it is, so the decision is
    // taken here, on the device): every workgroup's start-up — character map, superblock headers, a barrier — is paid half as often
    // (headline -4 %: 8 workgroups per CU x 2 tiles instead of 16 x 1), while a batch of MIXED lengths keeps the full grid — its
    // tiles take different times, and on the smaller grid the reference-shaped series lost 16 % (round 5, tools/ab_options_rows.sh).
    // The upper half of the grid leaves before it stages anything.
    uint32_t grid_x = gridDim.x;
    if (halve_uniform && plan_mixed && *plan_mixed != plan_epoch && grid_x >= 2) {
        grid_x = (grid_x + 1) / 2;
        if (blockIdx.x >= grid_x) return;
    }
    diag_begin();
    __shared__ int16_t s_xlat[256];
    __shared__ int16_t s_map[256];  // this index's character map, characters below 256
    __shared__ uint32_t s_len_bin[64];     // regrouping by length: bins, then their first slots
    __shared__ Quad s_len_rec[kBlock / 2];  // ... and the workgroup's records in length order
    constexpr bool planned = kMode != 0 && kMode != 4;
    constexpr bool translate = kMode == 2;  // only offered for 8-bit code words (plan_sigma <= 256)
    // (staged in front of the superblock cache so that both share ONE barrier: a workgroup lives for a few hundred
    // patterns, its start-up is not free)
    for (int c = threadIdx.x; c < 256; c += kBlock) {
        s_map[c] = ix_global.char2code[c];
        if (translate)
            s_xlat[c] = (c > 0 && c < plan_sigma) ? (int16_t)fm_map(ix_global, (uint16_t)plan_look_up[c]) : (int16_t)0;
    }
    FMX_FM_INV(ix_global);
    FMX_WITH_SB_CACHE(ix_global, ix);
    if (!ix.sb_cache) __syncthreads();
    const int role = threadIdx.x & 1;
    // codes the record's word carries (mode 3: a foreign plan with 16-bit codes gives the order only)
    constexpr int n_codes = kMode != 3 ? 64 / kCodeBits : 0;
    const int32_t pairs_per_grid = (int32_t)grid_x * kPairs;  // 32-bit indices: n < 2^31, fewer live registers
    // (an XCD-aware block order — a contiguous eighth of the sorted batch per XCD — was measured slower:
    // profiles/r01_i_xcd_remap.txt)
#if defined(FMX_DIAG_REVERSE)
    for (int32_t q0 = (int32_t)(grid_x - 1 - blockIdx.x) * kPairs; q0 < n; q0 += pairs_per_grid) {
#else
    for (int32_t q0 = (int32_t)blockIdx.x * kPairs; q0 < n; q0 += pairs_per_grid) {
#endif
        const int32_t q = q0 + (int32_t)(threadIdx.x >> 1);
        bool live = q < n;
        int32_t p = q, m = 0;
        CodeChunk ck = {0ull, 0ull, 0, 0};
        if (live) {
            if (planned) {
                Quad rq = ld_quad(recs + q);
                FMX_PIN_QUAD(rq);
                ck.lo = (uint64_t)rq.x | ((uint64_t)rq.y << 32);
                p = (int32_t)rq.z;
                m = (int32_t)(rq.w & kPlanLongPattern);
                if (m == (int32_t)kPlanLongPattern) m = pat_off[p + 1] - pat_off[p];
            } else {
                if (kMode == 4) p = redo_list[q];
                const int32_t beg0 = pat_off[p];
                m = pat_off[p + 1] - beg0;
                if ((kMode == 0 || kMode == 4) && m > 0) {
                    // the caller's order (no plan stage): the code word of the trailing characters is made here, as k_plan_codes
                    // makes it — one 16-byte fetch of the pattern's tail, characters mapped through the LDS copy of the map —
                    // so that short patterns take the lean loop of a planned batch
                    const TailWords tw = pattern_tail_load(pat, beg0, m);
                    uint32_t ch[8];
                    pattern_tail_chars(tw, pat, beg0, m, ch);
                    ck.lo = pattern_code_word<kCodeBits>(ix, s_map, ch, m);
                }
            }
        }
        // Patterns of MIXED LENGTHS (the reference's own benchmark draws 8..31 characters): a wave runs as long as its longest
        // pattern, so a workgroup in which some wave holds different lengths hands its kPairs records out again by length — a
        // counting sort in LDS on min(m, 63) — and every wave gets patterns of (nearly) one length.  Which pair runs which
        // record is free (results go to the pattern's index); the batch's order survives at the workgroup's granularity.
        // A batch of one length pays one barrier, a PLANNED one not even that (CountPlan.mixed).
        if (regroup && (!plan_mixed || *plan_mixed == plan_epoch)) {  // (a plan knows whether its batch holds two lengths)
            const bool wave_mixed = __any(live && m != __shfl(m, 0)) != 0;
            if (__syncthreads_or(wave_mixed ? 1 : 0)) {
                Quad mine;
                mine.x = (uint32_t)ck.lo;
                mine.y = (uint32_t)(ck.lo >> 32);
                mine.z = live ? (uint32_t)p : 0xffffffffu;
                mine.w = (uint32_t)m;
                const Quad r = regroup_records<kBlock>(s_len_bin, s_len_rec, mine);
                ck.lo = (uint64_t)r.x | ((uint64_t)r.y << 32);
                p = (int32_t)r.z;
                m = (int32_t)r.w;
                live = r.z != 0xffffffffu;
            }
        }
        ck.n = m < n_codes ? m : n_codes;
        int status = ST_OK;
        int32_t start = 0, end = 0;
        int32_t back = 0;  // characters consumed so far, counted from the pattern's end (FM:456: i = m - 1 - back)
        int32_t tabled = 0;
        // one decision per wave: does any of its patterns run past the record's code word?
#if defined(FMX_EXPERIMENT_NO_CHUNKS)
        const bool chunks = kMode == 3;
#else
        const bool chunks = n_codes == 0 || __any(m > n_codes);
#endif
        if (live && m <= 0) {
            status = ST_JAVA_AIOOBE;  // pattern[-1], FM:456-457
        } else if (live) {
            if (chunks)
                count_one<kMode, kCodeBits, true>(ix, s_inv, s_map, s_xlat, pat, pat_off, p, m, ck, role, start, end, back, tabled, status);
            else
                count_one<kMode, kCodeBits, false>(ix, s_inv, s_map, s_xlat, pat, pat_off, p, m, ck, role, start, end, back, tabled, status);
        }
        // LF-steps of the pattern: two ranks per character after the first (steps_mode 1: only those evaluated here, without
        // the ones the suffix table answered — what bench.py counts as executed work)
        const int32_t steps = 2 * (steps_mode ? back - tabled : back);
        status |= __shfl_xor(status, 1);
        if (live && role == 0) {  // (whole lane pairs: q is the same for both lanes of a pair)
            const int32_t d = end - start;
#if defined(FMX_EXPERIMENT_STORE_Q)
            counts[q] = d > 0 ? d : 0;  // (experiment: results in PROCESSING order — coalesced stores; what do the scattered ones cost?)
#else
            counts[p] = d > 0 ? d : 0;  // FM:473
#endif
            if (lf_steps) lf_steps[p] = steps;
            if (status_out) status_out[p] = status;
            if (range_out) {
                range_out[2 * (int64_t)p] = start;
                range_out[2 * (int64_t)p + 1] = end;
            }
        }
    }
    if (kMode == 4) {
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(redo_count + 1, 1u) == gridDim.x - 1) {
            redo_count[0] = 0;
            redo_count[1] = 0;
        }
    }
    diag_end();
}

#if !FMX_COMPACT
// ---- k_count_lean: the backward search of a PLANNED batch over an expanded image, written for instruction issue ---------------
// Round 5 found k_count bound by VALU issue, not by memory: 331 vector instructions per wave-step for ONE rank, 77 spilled SGPRs
// (every spill and reload is a v_writelane / v_readlane: vector issue slots), both arms of every uniform branch and the whole of
// the reference's own route (block header, level table, cumulative counts: WFBB:1113-1279) inside the loop.  This kernel is the
// same search (FM:455-474 over the fast route of WFBB.rank, fmx_device.hpp wt_rank_folded_t) with
//   * its arguments as ONE small struct of what the loop reads — no DevIndex by value;
//   * the image's shape in the template arguments (mapping rows by symbol or by superblock code), so no dead arm is live;
//   * the superblock headers staged in LDS as ONE quad per superblock {sigma | blockSizeLog, mapping table, cells, vector length};
//   * nothing but the fast routes inlined: symbol absent from the superblock (WFBB:1040-1042), run block (WFBB:1141-1146), next
//     block to the right that holds the symbol (WFBB:1048-1110 through a fast entry), and the walk over cells and path records
//     (WFBB:1187-1278 evaluated at flatten time).  Anything else — an entry on the reference's own route (codes longer than 16
//     bits, clamped entries, Q2), position 0, Q3's superblock — makes the rank answer -1; the pair then leaves the loop and puts its
//     pattern on a REDO LIST (an index and an atomic, in the plan's workspace) that k_count's list mode — every route inlined, as
//     ever — works off in a second, small launch right behind this one (usually it finds the list empty and leaves).  No real call
//     in this kernel: a call's caller-saved registers cost it 47 spilled VGPRs and ~25 scratch accesses per pattern when the cold
//     route was a function (measured: 0.142 ms against k_count's 0.091).
// Same counts, statuses, LF-step counts and ranges as k_count: the fast routes are its own, bit for bit.
// What the loop over a pattern's characters reads stays in SGPRs (CountLeanHot); what a pattern needs once — where its record,
// its first interval and its results live — is read from an LDS copy (CountLeanTile) at the point of use: as kernel arguments
// those twenty-odd SGPRs were live across the whole kernel, and with its three real calls (chunk_refill, regroup_records, the cold
// continuation) the allocator spilled ALL of them into VGPR lanes and reloaded them in the loop (311 v_readlane: vector
// issue slots, the very thing this kernel is short of).
struct CountLeanHot {
    const uint8_t *base;  // the image
    const SbcEntry *sbc;
    int32_t wt_sigma, n_sb;
    uint32_t wt_size;
    int32_t n;
};
struct CountLeanTile {
    const int32_t *C;
    const SuffixSlot *suffix_table;  // nullptr: none (or launches told to ignore it)
    const int16_t *char2code;
    int32_t *redo_list;     // patterns that met a route this kernel does not carry (at most n of them) ...
    uint32_t *redo_count;   // ... and how many: k_count's list mode works them off and leaves the counter zero again
    const uint16_t *pat;
    const int32_t *pat_off;
    const PlanRec *recs;
    int32_t *counts, *lf_steps, *status_out, *range_out;
    int32_t suffix_chars;
    uint32_t suffix_shift, suffix_mask;
    int32_t steps_mode;
    int32_t mixed;  // set by the kernel: the workgroups regroup their records by length (option on, batch of two lengths or more)
};
// the tile in LDS, addressed as LDS: a generic pointer to it made every field's address a VGPR PAIR kept across the kernel (32 VGPRs
// for addresses of constants, 26-47 spilled registers); its offset is laundered once per pattern so that the loads stay where
// they are used instead of being hoisted out of the loop into registers
using CountLeanTileLds = const __attribute__((address_space(3))) CountLeanTile;
__device__ __forceinline__ CountLeanTileLds *tile_lds(const CountLeanTile *generic) {
    uint32_t off = (uint32_t)(uintptr_t)(CountLeanTileLds *)generic;
    asm volatile("" : "+v"(off));
    return (CountLeanTileLds *)(uintptr_t)off;
}
struct CountLeanArgs {
    CountLeanHot hot;
    CountLeanTile tile;
    const SbDesc *sbd;
    const uint32_t *plan_mixed;
    uint32_t plan_epoch;
    int32_t regroup, halve_uniform;
};

// C[c] + rank(c, position) over the fast routes, or -1
template <bool kMapBySymbol>
__device__ __forceinline__ int32_t rank_lean(const CountLeanHot &A, const Quad *s_sbl, uint32_t position, uint32_t c) {
    const uint32_t sb_id = position >> 20;  // WFBB:1023
    // position 0 (WFBB:1012-1014), beyond the tree (WFBB:1015-1017), Q3's superblock, a symbol outside the tree: cold
    if (position - 1u >= A.wt_size || sb_id >= (uint32_t)A.n_sb || c >= (uint32_t)A.wt_sigma) return -1;
    uint64_t sbc_raw;
    memcpy(&sbc_raw, A.sbc + (uint64_t)sb_id * (uint32_t)A.wt_sigma + c, 8);  // WFBB:1024, 1034-1037
    const Quad hq = s_sbl[sb_id];
    const uint32_t bsl = hq.x >> 16;
    const int32_t sb_sigma = (int32_t)(int16_t)(hq.x & 0xffffu);
    const uint32_t blocks_log = 20u - bsl;
    const uint32_t block_index = position & ((1u << bsl) - 1u);
    uint32_t block_id = (position & 0xfffffu) >> bsl;
    const MapEntry *mapping = reinterpret_cast<const MapEntry *>(A.base + ((uint64_t)hq.y << 3));
    Quad mq;
    uint32_t map_row = c << blocks_log;
    if (kMapBySymbol) mq = ld_quad(mapping + map_row + block_id);  // WFBB:1044-1046: asked for together with the superblock entry
    FMX_OPAQUE64(sbc_raw);
    const int32_t e_rank = (int32_t)(uint32_t)sbc_raw;
    const int32_t e_sbc = (int32_t)(int16_t)(uint16_t)(sbc_raw >> 32);
    if (e_sbc >= sb_sigma + 1) return e_rank;  // WFBB:1040-1042
    if (!kMapBySymbol) {
        map_row = (uint32_t)e_sbc << blocks_log;
        mq = ld_quad(mapping + map_row + block_id);
    }
    FMX_PIN_QUAD(mq);
    const uint32_t tag = mq.x & 0xffu, value = mq.x >> 8;
    if (tag == kMapAbsent) {  // WFBB:1048-1110
        block_id += value;
        if (block_id >= (1u << blocks_log)) {  // WFBB:1060-1069
            int32_t next;
            memcpy(&next, A.sbc + (uint64_t)(sb_id + 1u) * (uint32_t)A.wt_sigma + c, 4);
            return next;
        }
        const Quad nq = ld_quad(mapping + map_row + block_id);
        const uint32_t ntag = nq.x & 0xffu;
        // WFBB:1096-1108 reads the u24 of leaf `mapping value` of that block: a fast entry of a block with a tree IS that leaf's
        // u24; landing on a RUN block the reference reads 4 bytes early (Q11) — the word the flattener left in the entry
        if (ntag >= 1u && ntag <= kMapMaxLen) return e_rank + (int32_t)(nq.x >> 8);
        if (ntag == 0u && (nq.w >> 24) == (kMapRunNext >> 24)) return e_rank + (int32_t)(nq.w & 0xffffffu);
        return -1;
    }
    if (tag > kMapMaxLen) return -1;                                       // the reference's own route
    if (tag == 0u) return e_rank + (int32_t)value + (int32_t)block_index;  // run block, WFBB:1141-1146
    const uint32_t code_length = tag;
    const uint32_t code = (mq.y >> 24) | ((mq.z >> 24) << 8);
    uint32_t node_a = mq.y & 0xffffffu, node_b = mq.z & 0xffffffu;
    const PathRec *path = reinterpret_cast<const PathRec *>(mapping) + mq.w;
    const uint8_t *cells = A.base + ((uint64_t)hq.z << 3);
    const uint32_t bv_len = hq.w;
    uint32_t pos = node_a + block_index;
    pos = pos > bv_len ? bv_len : pos;  // (RRR:360-365: the rank at the clamped position IS the saturated value)
    Quad pq = {0, 0, 0, 0};
    if (code_length > 1u) pq = ld_quad(path);  // records of levels 1 and 2
    Quad cell = ld_quad(cells + (uint64_t)(pos / kBvCellBits) * 16u);
    FMX_PIN_QUAD(pq);
    FMX_PIN_QUAD(cell);
    uint32_t node_rank = block_index;
    FMX_NO_UNROLL
    for (uint32_t depth = 0; depth < code_length; ++depth) {
        const uint32_t rank1 = cell.x + bv_cell_prefix(cell, pos % kBvCellBits) - node_b;  // WFBB:1216-1218
        node_rank = (code >> (code_length - depth - 1u)) & 1u ? rank1 : node_rank - rank1;  // WFBB:1235-1244
        if (depth + 1u != code_length) {
            node_a = (depth & 1u) ? pq.z : pq.x;  // record `depth` = the node at level depth + 1
            node_b = (depth & 1u) ? pq.w : pq.y;
            pos = node_a + node_rank;
            pos = pos > bv_len ? bv_len : pos;
            cell = ld_quad(cells + (uint64_t)(pos / kBvCellBits) * 16u);
            if ((depth & 1u) && depth + 2u < code_length) pq = ld_quad(path + depth + 1u);  // the next two levels
            FMX_PIN_QUAD(cell);
            FMX_PIN_QUAD(pq);
        }
    }
    return e_rank + (int32_t)value + (int32_t)node_rank;  // WFBB:1281-1284
}

template <int kCodeBits, bool kChunks, bool kMapBySymbol>
__device__ __forceinline__ void count_one_lean(const CountLeanHot &H, CountLeanTileLds *T, const Quad *s_sbl,
                                               const int16_t *s_map, int32_t p, int32_t m, CodeChunk ck, int role, int32_t &start,
                                               int32_t &end, int32_t &back, int32_t &tabled, bool &cold) {
    constexpr uint32_t code_mask = (1u << kCodeBits) - 1u;
    int32_t beg = 0;
    if (kChunks && m > ck.n) beg = T->pat_off[p];
    int32_t c = chunk_code<kCodeBits>(ck, 0);
    if (c == 0) return;  // FM:458-460
    {
        uint64_t cc;
        memcpy(&cc, T->C + c, 8);  // cumulativeCounts[c], [c + 1]
        start = (int32_t)(uint32_t)cc;
        end = (int32_t)(uint32_t)(cc >> 32);
    }
    const SuffixSlot *table = T->suffix_table;
    int len = 0;
    if (table) {
        const int32_t chars = T->suffix_chars;
        len = m < chars ? (int)m : (int)chars;
    }
    if (ck.n < len) len = ck.n;
    if (len >= 2) {  // the interval after the last `len` characters is tabulated (count_one has the whole story)
        const int bits = len * kCodeBits;
        const uint64_t key = bits >= 64 ? ck.lo : (ck.lo & ((1ull << bits) - 1ull));
        if ((uint32_t)(key >> (bits - kCodeBits)) != 0u && key != kSuffixEmpty) {
            const uint32_t shift = T->suffix_shift, mask = T->suffix_mask;
            const int top = bits - kCodeBits;  // fm_suffix_home
            const uint64_t low6 = (key >> top) & (kSuffixGroup - 1);
            const uint64_t hash = (key & ~((uint64_t)(kSuffixGroup - 1) << top)) * kSuffixHashMul;
            const uint32_t group = (uint32_t)(hash >> shift);
            const uint32_t turn = (uint32_t)(hash >> (shift - kSuffixGroupLog2)) & (kSuffixGroup - 1);
            uint32_t h = (group * kSuffixGroup + ((uint32_t)low6 ^ turn)) & mask;
            Quad q = ld_quad(table + h);
            FMX_PIN_QUAD(q);
            uint64_t k = (uint64_t)q.x | ((uint64_t)q.y << 32);
            for (uint32_t probe = 0; k != key && k != kSuffixEmpty && probe < mask / kSuffixGroup; ++probe) {
                h = (h + kSuffixGroup) & mask;
                q = ld_quad(table + h);
                k = (uint64_t)q.x | ((uint64_t)q.y << 32);
            }
            if (k == key) {
                start = (int32_t)q.z;
                end = (int32_t)q.w;
                back = len - 1;
            }
        }
        tabled = back;
    }
    while (start < end && back + 1 < m) {  // FM:464
        ++back;
        if (kChunks) {
            if (back - ck.base >= ck.n) ck = chunk_refill_body<kCodeBits>(T->char2code, s_map, T->pat, beg, m, back);
            c = chunk_code<kCodeBits>(ck, back - ck.base);
        } else {
            c = (int32_t)((uint32_t)(ck.lo >> (back * kCodeBits)) & code_mask);
        }
        if (c == 0) {  // FM:466-468
            start = end = 0;
            --back;
            break;
        }
        const int32_t mine = rank_lean<kMapBySymbol>(H, s_sbl, (uint32_t)(role ? end : start), (uint32_t)c);
        const int32_t other = __shfl_xor(mine, 1);
        if ((mine | other) < 0) {  // one of the two ranks is off the fast routes: this character and the rest go to the cold continuation
            --back;
            cold = true;
            break;
        }
        start = role ? other : mine;  // FM:469
        end = role ? mine : other;    // FM:470
    }
}

template <int kBlock, int kCodeBits, bool kMapBySymbol>
FMX_COUNT_KERNEL(kBlock) void k_count_lean(CountLeanArgs A) {
    constexpr int kPairs = kBlock / 2;
    uint32_t grid_x = gridDim.x;
    const bool batch_mixed = !A.plan_mixed || *A.plan_mixed == A.plan_epoch;
    if (A.halve_uniform && !batch_mixed && grid_x >= 2) {  // (k_count: a batch of ONE length runs on half the grid)
        grid_x = (grid_x + 1) / 2;
        if (blockIdx.x >= grid_x) return;
    }
    __shared__ int16_t s_map[256];
    __shared__ uint32_t s_len_bin[64];
    __shared__ Quad s_len_rec[kBlock / 2];
    __shared__ Quad s_sbl[kSbCacheMax];
    __shared__ CountLeanTile s_tile;
    if (threadIdx.x == 0) {
        s_tile = A.tile;
        s_tile.mixed = (A.regroup && batch_mixed) ? 1 : 0;
    }
    for (int c = threadIdx.x; c < 256; c += kBlock) s_map[c] = A.tile.char2code[c];
    for (int i = threadIdx.x; i < A.hot.n_sb; i += kBlock) {
        const Quad h = ld_quad(&A.sbd[i]), v = ld_quad(&A.sbd[i].rrr);
        Quad q;
        q.x = h.x;  // sigma | blockSizeLog << 16
        q.y = h.y;  // off_mapping
        q.z = v.x;  // off_rec: the superblock's cells
        q.w = v.z;  // length of its bit vector
        s_sbl[i] = q;
    }
    __syncthreads();
    const CountLeanHot H = A.hot;
    const int role = threadIdx.x & 1;
    constexpr int n_codes = 64 / kCodeBits;
    const int32_t n = H.n;
    const int32_t pairs_per_grid = (int32_t)grid_x * kPairs;
    for (int32_t q0 = (int32_t)blockIdx.x * kPairs; q0 < n; q0 += pairs_per_grid) {
        const int32_t q = q0 + (int32_t)(threadIdx.x >> 1);
        CountLeanTileLds *T = tile_lds(&s_tile);
        bool live = q < n;
        int32_t p = q, m = 0;
        CodeChunk ck = {0ull, 0ull, 0, 0};
        if (live) {
            Quad rq = ld_quad(T->recs + q);
            FMX_PIN_QUAD(rq);
            ck.lo = (uint64_t)rq.x | ((uint64_t)rq.y << 32);
            p = (int32_t)rq.z;
            m = (int32_t)(rq.w & kPlanLongPattern);
            if (m == (int32_t)kPlanLongPattern) {
                const int32_t *off = T->pat_off;
                m = off[p + 1] - off[p];
            }
        }
        if (T->mixed) {  // mixed lengths: every wave gets patterns of (nearly) one (k_count)
            const bool wave_mixed = __any(live && m != __shfl(m, 0)) != 0;
            if (__syncthreads_or(wave_mixed ? 1 : 0)) {
                Quad mine;
                mine.x = (uint32_t)ck.lo;
                mine.y = (uint32_t)(ck.lo >> 32);
                mine.z = live ? (uint32_t)p : 0xffffffffu;
                mine.w = (uint32_t)m;
                const Quad r = regroup_records_body<kBlock>(s_len_bin, s_len_rec, mine);
                ck.lo = (uint64_t)r.x | ((uint64_t)r.y << 32);
                p = (int32_t)r.z;
                m = (int32_t)r.w;
                live = r.z != 0xffffffffu;
            }
        }
        ck.n = m < n_codes ? m : n_codes;
        int status = ST_OK;
        int32_t start = 0, end = 0, back = 0, tabled = 0;
        bool cold = false;
        const bool chunks = __any(m > n_codes);  // one decision per wave
        if (live && m <= 0) {
            status = ST_JAVA_AIOOBE;  // pattern[-1], FM:456-457
        } else if (live) {
            if (chunks)
                count_one_lean<kCodeBits, true, kMapBySymbol>(H, T, s_sbl, s_map, p, m, ck, role, start, end, back, tabled, cold);
            else
                count_one_lean<kCodeBits, false, kMapBySymbol>(H, T, s_sbl, s_map, p, m, ck, role, start, end, back, tabled, cold);
        }
        if (cold && role == 0) {  // the whole pattern goes on the redo list: k_count's list mode searches it over every route
            uint32_t *redo_count = T->redo_count;
            const uint32_t at = atomicAdd(redo_count, 1u);
            T->redo_list[at] = p;
        }
        if (live && role == 0 && !cold) {
            const int32_t d = end - start;
            T->counts[p] = d > 0 ? d : 0;  // FM:473
            int32_t *lf = T->lf_steps, *sto = T->status_out, *rng = T->range_out;
            if (lf) lf[p] = 2 * (T->steps_mode ? back - tabled : back);
            if (sto) sto[p] = status;
            if (rng) {
                rng[2 * (int64_t)p] = start;
                rng[2 * (int64_t)p + 1] = end;
            }
        }
    }
}
#endif  // !FMX_COMPACT

// ---- growing the suffix table (fmx_device.hpp: fm_suffix_extend) when an index becomes resident ----
// level 1: every character of the alphabet that occurs (its interval is cumulativeCounts' own)
__global__ __launch_bounds__(256) void k_suffix_level1(DevIndex ix, SuffixSlot *__restrict__ out, uint32_t *__restrict__ count,
                                                       uint32_t cap) {
    const int32_t c = (int32_t)(blockIdx.x * 256 + threadIdx.x) + 1;  // code 0 is never tabulated
    if (c + 1 >= ix.n_c || c >= ix.wt_sigma) return;
    const int32_t lo = ix.C[c], hi = ix.C[c + 1];
    if (lo >= hi) return;
    const uint32_t at = atomicAdd(count, 1u);
    if (at < cap) out[at] = SuffixSlot{(uint64_t)(uint32_t)c, (uint32_t)lo, (uint32_t)hi};
}
// level depth + 1: every string of level `depth` with every character in front of it
__global__ __launch_bounds__(256) void k_suffix_expand(DevIndex ix, const SuffixSlot *__restrict__ in, uint32_t n_in, int depth,
                                                       int key_bits, SuffixSlot *__restrict__ out, uint32_t *__restrict__ count,
                                                       uint32_t cap) {
    const uint32_t sigma1 = (uint32_t)ix.wt_sigma - 1u;  // codes 1 .. sigma - 1
    const uint64_t work = (uint64_t)n_in * sigma1;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < work; t += (uint64_t)gridDim.x * 256) {
        if (*reinterpret_cast<volatile uint32_t *>(count) > cap) return;  // the level no longer fits: it will be dropped
        const uint32_t i = (uint32_t)(t / sigma1);
        const int32_t c = (int32_t)(t - (uint64_t)i * sigma1) + 1;
        if (c + 1 >= ix.n_c) continue;
        SuffixSlot child;
        if (!fm_suffix_extend(ix, in[i], depth, c, key_bits, child)) continue;
        const uint32_t at = atomicAdd(count, 1u);
        if (at < cap) out[at] = child;
    }
}
// one level (strings of `len` codes) into the hash table (slots preset to kSuffixEmpty); ix carries the table's geometry (fm_suffix_home)
__global__ __launch_bounds__(256) void k_suffix_insert(DevIndex ix, const SuffixSlot *__restrict__ in, uint32_t n_in, int len,
                                                       SuffixSlot *__restrict__ slots) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_in) return;
    const SuffixSlot me = in[i];
    if (me.key == kSuffixEmpty) return;  // the one key that reads as a free slot is never tabulated (fm_suffix_key)
    uint32_t h = fm_suffix_home(ix, me.key, len);
    for (uint32_t probe = 0; probe <= ix.suffix_mask / kSuffixGroup; ++probe, h = (h + kSuffixGroup) & ix.suffix_mask) {
        unsigned long long *key = reinterpret_cast<unsigned long long *>(&slots[h].key);
        if (atomicCAS(key, (unsigned long long)kSuffixEmpty, (unsigned long long)me.key) == (unsigned long long)kSuffixEmpty) {
            slots[h].start = me.start;
            slots[h].end = me.end;
            return;
        }
    }
    // (every slot of this string's probe sequence is taken: cannot happen below 0.7 load spread by the hash; the string is then simply not found)
}
int launch_suffix_level1(const DevIndex &ix, SuffixSlot *out, uint32_t *count, uint32_t cap, hipStream_t st) {
    DevIndex plain = ix;
    plain.sb_cache = nullptr;
    plain.suffix_table = nullptr;
    hipLaunchKernelGGL(k_suffix_level1, dim3((unsigned)((ix.wt_sigma + 255) / 256)), dim3(256), 0, st, plain, out, count, cap);
    return (int)hipGetLastError();
}
int launch_suffix_expand(const DevIndex &ix, int n_cu, const SuffixSlot *in, uint32_t n_in, int depth, int key_bits, SuffixSlot *out,
                         uint32_t *count, uint32_t cap, hipStream_t st) {
    DevIndex plain = ix;
    plain.sb_cache = nullptr;
    plain.suffix_table = nullptr;
    const uint64_t work = (uint64_t)n_in * (uint64_t)(ix.wt_sigma > 1 ? ix.wt_sigma - 1 : 1);
    uint64_t blocks = (work + 255) / 256;
    const uint64_t cap_blocks = (uint64_t)n_cu * 64;
    if (blocks > cap_blocks) blocks = cap_blocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_suffix_expand, dim3((unsigned)blocks), dim3(256), 0, st, plain, in, n_in, depth, key_bits, out, count, cap);
    return (int)hipGetLastError();
}
int launch_suffix_insert(const DevIndex &geometry, const SuffixSlot *in, uint32_t n_in, int len, SuffixSlot *slots, hipStream_t st) {
    if (n_in == 0) return 0;
    hipLaunchKernelGGL(k_suffix_insert, dim3((n_in + 255) / 256), dim3(256), 0, st, geometry, in, n_in, len, slots);
    return (int)hipGetLastError();
}

// DevIndex.suffix_order1 from the finished table: one thread per pair of codes (x, y)
__global__ __launch_bounds__(256) void k_suffix_order1(DevIndex ix, float *__restrict__ out) {
    const int sigma = ix.wt_sigma;
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (i >= sigma * sigma) return;
    const int x = i / sigma, y = i - x * sigma;
    float f = 0.0f, p = 0.0f;
    if (x != 0 && y != 0 && x + 1 < ix.n_c) {
        int32_t s2 = 0, e2 = 0, back = 0;
        const int32_t cx = ix.C[x], nx = ix.C[x + 1] - cx;
        if (nx > 0 && fm_suffix_lookup(ix, (uint64_t)y | ((uint64_t)x << ix.suffix_key_bits), 2, s2, e2, back) && e2 > s2) {
            f = (float)(s2 - cx) / (float)nx;
            p = (float)(e2 - s2) / (float)nx;
        }
    }
    out[2 * i] = f;
    out[2 * i + 1] = p;
}
int launch_suffix_order1(const DevIndex &ix, float *out, hipStream_t st) {
    const int pairs = ix.wt_sigma * ix.wt_sigma;
    if (pairs <= 0 || ix.wt_sigma > kOrder1MaxSigma || !ix.suffix_table) return 0;
    hipLaunchKernelGGL(k_suffix_order1, dim3((pairs + 255) / 256), dim3(256), 0, st, ix, out);
    return (int)hipGetLastError();
}

// ---- growing the window directory (fmx_device.hpp: win_build_cell / win_build_other) when an index becomes resident: a lane
// per window; first the cells and every window's number of class-3 positions, then — the caller has turned those counts into
// each window's first entry — the entries ----
__global__ __launch_bounds__(256) void k_win_build(DevIndex ix, uint32_t n_win, Quad *__restrict__ out, uint32_t *__restrict__ others) {
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < n_win; w += (uint64_t)gridDim.x * 256) {
        uint32_t words[16];
        others[w] = win_build_cell(ix, (uint32_t)w, words);
        for (int i = 0; i < 4; ++i) out[4 * w + i] = Quad{words[4 * i], words[4 * i + 1], words[4 * i + 2], words[4 * i + 3]};
    }
}
__global__ __launch_bounds__(256) void k_win_other(DevIndex ix, uint32_t n_win, Quad *__restrict__ cells, const uint32_t *__restrict__ first,
                                                   uint16_t *__restrict__ entries, uint32_t *__restrict__ open_entries, int entry4,
                                                   uint64_t *__restrict__ full, uint32_t full_cap) {
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < n_win; w += (uint64_t)gridDim.x * 256) {
        uint32_t words[16];
        for (int i = 0; i < 4; ++i) {
            const Quad q = cells[4 * w + i];
            words[4 * i] = q.x;
            words[4 * i + 1] = q.y;
            words[4 * i + 2] = q.z;
            words[4 * i + 3] = q.w;
        }
        const uint32_t open = win_build_other(ix, (uint32_t)w, words, first[w], entries, entry4 != 0, full, full_cap, open_entries + 2);
        cells[4 * w + 1].x = words[4];
        if (open & 0x7fffffffu) atomicAdd(open_entries, open & 0x7fffffffu);  // (entries that carry a status or `suspect`: statistics)
        if (open >> 31) atomicOr(open_entries + 1, 1u);  // an answer that does not fit an entry: the caller drops the directory
    }
}
// the flat form (fmx_device.hpp win_build_flat): a lane per position
__global__ __launch_bounds__(256) void k_win_flat(DevIndex ix, uint32_t n_pos, uint32_t *__restrict__ flat, uint32_t *__restrict__ tail,
                                                  uint64_t *__restrict__ full, uint32_t full_cap) {
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < n_pos; p += (uint64_t)gridDim.x * 256) {
        const uint32_t open = win_build_flat(ix, (uint32_t)p, flat, full, full_cap, tail + 2);
        if (open & 0x7fffffffu) atomicAdd(tail, open & 0x7fffffffu);
        if (open >> 31) atomicOr(tail + 1, 1u);
    }
}
static DevIndex win_plain_index(const DevIndex &ix) {
    DevIndex plain = ix;  // the directory is made from the tree walk's own answers
    plain.sb_cache = nullptr;
    plain.suffix_table = nullptr;
    plain.win = nullptr;
    plain.win_other = nullptr;
    plain.win_full = nullptr;
    plain.win_entry4 = 0;
    plain.win_flat = 0;
    plain.c_lds = nullptr;
    plain.c_lut = nullptr;
    plain.c_lut_shift = 0;
    return plain;
}
static unsigned win_blocks(int n_cu, uint32_t n_win) {
    uint64_t blocks = ((uint64_t)n_win + 255) / 256;
    const uint64_t cap_blocks = (uint64_t)n_cu * 64;
    return (unsigned)(blocks > cap_blocks ? cap_blocks : blocks);
}
int launch_win_build(const DevIndex &ix, int n_cu, uint32_t n_win, Quad *out, uint32_t *others, hipStream_t st) {
    if (n_win == 0) return 0;
    hipLaunchKernelGGL(k_win_build, dim3(win_blocks(n_cu, n_win)), dim3(256), 0, st, win_plain_index(ix), n_win, out, others);
    return (int)hipGetLastError();
}
int launch_win_other(const DevIndex &ix, int n_cu, uint32_t n_win, Quad *cells, const uint32_t *first, uint16_t *entries,
                     uint32_t *open_entries, int entry4, uint64_t *full, uint32_t full_cap, hipStream_t st) {
    if (n_win == 0) return 0;
    hipLaunchKernelGGL(k_win_other, dim3(win_blocks(n_cu, n_win)), dim3(256), 0, st, win_plain_index(ix), n_win, cells, first, entries,
                       open_entries, entry4, full, full_cap);
    return (int)hipGetLastError();
}

int launch_win_flat(const DevIndex &ix, int n_cu, uint32_t n_pos, uint32_t *flat, uint32_t *tail, uint64_t *full, uint32_t full_cap,
                    hipStream_t st) {
    if (n_pos == 0) return 0;
    hipLaunchKernelGGL(k_win_flat, dim3(win_blocks(n_cu, n_pos)), dim3(256), 0, st, win_plain_index(ix), n_pos, flat, tail, full, full_cap);
    return (int)hipGetLastError();
}

// FM:526-548: hit k of pattern p is SA row i = start + 1 + k; walk LF until a sampled row.
constexpr int kRedoHead = 4;   // ints in front of a redo list's entries ({count, 0, 0, 0}: 16 bytes)
constexpr int kWalkLanes = 128;  // lanes of k_locate_walk per pattern (at most): two waves
template <int kBlock, int kWin>
FMX_WALK_KERNEL(kBlock) void k_locate_walk(DevIndex ix_global, const int32_t *__restrict__ range, int32_t n,
                                                        int32_t max_matches, int32_t *__restrict__ locs,
                                                        int32_t loc_cap, int32_t slots, int32_t *__restrict__ found,
                                                        int32_t *__restrict__ lf_steps,
                                                        int32_t *__restrict__ status_out,
                                                        const int32_t *__restrict__ taken,
                                                        const PlanRec *__restrict__ order,
                                                        const uint32_t *__restrict__ order_idle,
                                                        int64_t *__restrict__ set_locs, int64_t set_base) {
    // set_locs (nullable; segment sets): the hits go straight into the SET's rows — int64 text positions moved by this
    // segment's start, behind the taken[p] hits of the earlier segments (none: the first segment) — instead of into `locs` for a kernel that appends
    // them (8 x 0.5 ms per step of configs[4]); k_segment_commit then advances the set's `found` by what was located here.
    FMX_FM_INV(ix_global);
    FMX_WITH_SB_CACHE(ix_global, ix);
    // order (nullable): the batch's records {start, end, pattern} by the first row of their ranges (k_walk_hist); its first
    // *order_idle records have nothing to locate.  Whole windows of the fine pass below that mark get one lane per pattern
    // (a lane that finds hits there after all walks them one after the other); every other pattern gets `slots` lanes, at most
    // two waves (with maxMatches 1000 most slots stay empty, and a lane per slot spent its time finding that out: 1 M queries
    // at sampleRate 1 3.3 -> 2.5 ms; a cap of one wave costs maxMatches 100 its second round: +3..+8 %): lane g walks hits g,
    // g + lanes, ... — adjacent lanes walk adjacent SA rows.
    const int32_t lanes = slots < kWalkLanes ? slots : kWalkLanes;
    const int64_t idle = order ? (int64_t)(*order_idle / (uint32_t)kFineWindow) * kFineWindow : 0;
    const int64_t total = idle + ((int64_t)n - idle) * lanes;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        int64_t rec = t;
        int32_t k = 0, step = 1;
        if (t >= idle) {
            rec = idle + (t - idle) / lanes;
            k = (int32_t)((t - idle) - (rec - idle) * lanes);
            step = lanes;
        }
        int32_t p = (int32_t)rec;
        int32_t start, end;
        if (order) {
            const Quad r = ld_quad(order + rec);
            start = (int32_t)r.x;
            end = (int32_t)r.y;
            p = (int32_t)r.z;
        } else {
            start = range[2 * p];
            end = range[2 * p + 1];
        }
        int32_t hits = start < end ? end - start : 0;
        // segment sets: `taken[p]` hits came from earlier segments, the caller's loop passes maxMatches - taken
        int32_t limit = max_matches;
        if (taken) {
            limit = max_matches - taken[p];
            if (limit <= 0) hits = 0;
        }
        // the reference stops at maxMatches (FM:544-546) and overruns `locations` beyond its length (Java AIOOBE)
        const int32_t wanted = (limit > 0 && hits > limit) ? limit : hits;
        const int32_t located = wanted < loc_cap ? wanted : loc_cap;
        if (k == 0) {
            found[p] = located;
            if (wanted > loc_cap && status_out) atomicOr(&status_out[p], ST_JAVA_AIOOBE);
        }
        for (; k < located; k += step) {
            int status = ST_OK;
            int32_t distance;
            const int32_t at = fm_locate_hit<kWin>(ix, s_inv, start, k, distance, status);
            if (set_locs)
                set_locs[(int64_t)p * loc_cap + (taken ? taken[p] : 0) + k] = set_base + at;
            else
                locs[(int64_t)p * loc_cap + k] = at;
            if (lf_steps && distance) atomicAdd(&lf_steps[p], distance);
            if (status && status_out) atomicOr(&status_out[p], status);
        }
    }
}

// k_locate_walk over an index with a window directory, with the walks still under way PACKED into fewer waves twice on their
// way (after sample_rate / 2 and sample_rate * 3 / 4 steps).  The kernel is bound by VALU issue, and a wave walks until the longest
// of its 64 walks meets a sampled row — sample_rate - 1 steps where the average walk takes half of that: half of the wave-steps ran
// on lanes whose walk was over (profiles/r05_experiments.txt 12).  After an instalment every lane still walking writes its state
// {row, distance, status, where the position goes} to LDS at its rank among them, the workgroup's first lanes take the states
// over, and the waves behind them fall through the next instalment.  Same tickets, same stores as k_locate_walk.
template <int kBlock, int kForm>
FMX_WALK_KERNEL(kBlock) void k_locate_walk_c(DevIndex ix_global, const int32_t *__restrict__ range, int32_t n,
                                             int32_t max_matches, int32_t *__restrict__ locs, int32_t loc_cap, int32_t slots,
                                             int32_t *__restrict__ found, int32_t *__restrict__ lf_steps,
                                             int32_t *__restrict__ status_out, const int32_t *__restrict__ taken,
                                             const PlanRec *__restrict__ order, const uint32_t *__restrict__ order_idle,
                                             int64_t *__restrict__ set_locs, int64_t set_base, int packings) {
    FMX_FM_INV(ix_global);
    FMX_WITH_SB_CACHE(ix_global, ix);
    __shared__ Quad s_state[kBlock];        // {row, distance, status, pattern}
    __shared__ int64_t s_dest[kBlock];      // index of the hit's position in locs / set_locs
    __shared__ uint32_t s_walking[kBlock / 64];
    const int32_t lanes = slots < kWalkLanes ? slots : kWalkLanes;
    const int64_t idle = order ? (int64_t)(*order_idle / (uint32_t)kFineWindow) * kFineWindow : 0;
    const int64_t total = idle + ((int64_t)n - idle) * lanes;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int32_t walk_limit = fm_walk_limit(ix);
    // instalments: two packings — after sample_rate / 2 and 3 sample_rate / 4 steps — or three, after every quarter (option walk_pack)
    const int32_t first = packings >= 3 ? ix.sample_rate / 4 : ix.sample_rate / 2, second = ix.sample_rate / 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (every lane of the workgroup runs the same number of rounds: the barriers below are the workgroup's)
    for (int64_t t0 = (int64_t)blockIdx.x * kBlock; t0 < total; t0 += stride) {
        const int64_t t = t0 + threadIdx.x;
        const bool have = t < total;
        int64_t rec = have ? t : 0;
        int32_t k = 0, step = 1;
        if (have && t >= idle) {
            rec = idle + (t - idle) / lanes;
            k = (int32_t)((t - idle) - (rec - idle) * lanes);
            step = lanes;
        }
        int32_t p = (int32_t)rec, start = 0, end = 0, taken_p = 0, located = 0;
        if (have) {
            if (order) {
                const Quad r = ld_quad(order + rec);
                start = (int32_t)r.x;
                end = (int32_t)r.y;
                p = (int32_t)r.z;
            } else {
                start = range[2 * p];
                end = range[2 * p + 1];
            }
            int32_t hits = start < end ? end - start : 0;
            int32_t limit = max_matches;
            if (taken) {
                taken_p = taken[p];
                limit = max_matches - taken_p;
                if (limit <= 0) hits = 0;
            }
            // the reference stops at maxMatches (FM:544-546) and overruns `locations` beyond its length (Java AIOOBE)
            const int32_t wanted = (limit > 0 && hits > limit) ? limit : hits;
            located = wanted < loc_cap ? wanted : loc_cap;
            if (k == 0) {
                found[p] = located;
                if (wanted > loc_cap && status_out) atomicOr(&status_out[p], ST_JAVA_AIOOBE);
            }
        }
        for (;; k += step) {  // this ticket's hits k, k + step, ...: a round of the workgroup per hit
            const bool hit = have && k < located;
            if (!__syncthreads_or(hit ? 1 : 0)) break;
            // the walk this lane carries (after a packing: another lane's)
            WalkState w = {start + 1 + k, 0, ST_OK};  // FM:527-529
            int32_t wp = p;
            int64_t dest = (int64_t)p * loc_cap + (set_locs ? taken_p : 0) + k;
            bool walking = hit;
            for (int phase = 0; phase <= packings; ++phase) {
                const int32_t budget = phase == packings ? 0x7fffffff : (phase == 0 ? first : second);
                if (walking && fm_locate_steps_win<kForm>(ix, w, budget, walk_limit)) {
                    const int32_t at = fm_locate_finish_win(ix, s_inv, w);
                    if (set_locs)
                        set_locs[dest] = set_base + at;
                    else
                        locs[dest] = at;
                    if (lf_steps && w.distance) atomicAdd(&lf_steps[wp], w.distance);
                    if (w.status && status_out) atomicOr(&status_out[wp], w.status);
                    walking = false;
                }
                if (phase == packings) break;
                // pack the walks still under way into the workgroup's first lanes
                const unsigned long long ball = __ballot(walking ? 1 : 0);
                if (lane == 0) s_walking[wave] = (uint32_t)__popcll(ball);
                __syncthreads();
                uint32_t before = 0, all = 0;
                for (int i = 0; i < kBlock / 64; ++i) {
                    const uint32_t c = s_walking[i];
                    before += i < wave ? c : 0u;
                    all += c;
                }
                if (walking) {
                    const uint32_t at = before + (uint32_t)__popcll(ball & ((1ull << lane) - 1ull));
                    s_state[at] = Quad{(uint32_t)w.j, (uint32_t)w.distance, (uint32_t)w.status, (uint32_t)wp};
                    s_dest[at] = dest;
                }
                __syncthreads();
                walking = threadIdx.x < all;
                if (walking) {
                    const Quad q = s_state[threadIdx.x];
                    w.j = (int32_t)q.x;
                    w.distance = (int32_t)q.y;
                    w.status = (int)q.z;
                    wp = (int32_t)q.w;
                    dest = s_dest[threadIdx.x];
                }
                __syncthreads();  // (the arrays are written again in the next packing)
            }
        }
    }
}

// k_locate_walk over an index with a window directory as a TICKET QUEUE per wave (round 6).  A walk ends where it meets a sampled
// row: after 0 .. sample_rate - 1 steps, so a wave that gives every lane ONE hit walks sample_rate - 1 steps with half of its lanes
// done on average (k_locate_walk_c packs the unfinished walks of a workgroup through LDS twice on the way: three barriers a
// packing).  Here a wave owns a run of `chunk` consecutive tickets and hands them out itself: every `burst` steps the lanes whose
// walk is over take the next hit of their ticket, or the next tickets of the run (a ballot and a prefix count: no LDS, no
// barrier), and walk on beside the lanes still under way.  A lane idles for half a burst per walk instead of half a walk.
// Same tickets, same stores as k_locate_walk: which lane walks which hit is free (results go to the hit's own slot).
template <int kBlock, int kForm>
FMX_WALK_KERNEL(kBlock) void k_locate_walk_q(DevIndex ix, const int32_t *__restrict__ range, int32_t n, int32_t max_matches,
                                             int32_t *__restrict__ locs, int32_t loc_cap, int32_t slots,
                                             int32_t *__restrict__ found, int32_t *__restrict__ lf_steps,
                                             int32_t *__restrict__ status_out, const int32_t *__restrict__ taken,
                                             const PlanRec *__restrict__ order, const uint32_t *__restrict__ order_idle,
                                             int64_t *__restrict__ set_locs, int64_t set_base, int32_t chunk, int32_t burst) {
    FMX_FM_INV(ix);
    const int32_t lanes = slots < kWalkLanes ? slots : kWalkLanes;
    const int64_t idle = order ? (int64_t)(*order_idle / (uint32_t)kFineWindow) * kFineWindow : 0;
    const int64_t total = idle + ((int64_t)n - idle) * lanes;
    const int32_t walk_limit = fm_walk_limit(ix);
    const int64_t waves = (int64_t)gridDim.x * (kBlock / 64);
    const int64_t wave_id = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
    for (int64_t run = wave_id * chunk; run < total; run += waves * chunk) {
        int64_t next = run;  // wave-uniform: the run's first ticket not handed out yet
        const int64_t run_end = run + chunk < total ? run + chunk : total;
        // the ticket a lane holds: hits k, k + step, ... < located of pattern p; and the walk it carries
        int32_t p = 0, start = 0, k = 0, step = 1, located = 0, taken_p = 0;
        bool walking = false;
        WalkState w = {0, 0, ST_OK};
        for (;;) {
            // the hand-out (every lane votes: `next` stays the wave's): a lane whose walk is over moves on to the next hit of its
            // ticket, and takes a new ticket when that one is done (or there was none yet)
            bool need = false;
            if (!walking) {
                k += step;
                need = k >= located;
            }
            const unsigned long long ball = __ballot(need ? 1 : 0);
            const int64_t mine = next + (int64_t)__popcll(ball & below);
            next += (int64_t)__popcll(ball);
            if (need) {
                located = 0;  // (no ticket left for this lane: it idles until the run is walked)
                k = 0;
                if (mine < run_end) {
                    int64_t rec = mine;
                    step = 1;
                    if (mine >= idle) {
                        rec = idle + (mine - idle) / lanes;
                        k = (int32_t)((mine - idle) - (rec - idle) * lanes);
                        step = lanes;
                    }
                    p = (int32_t)rec;
                    int32_t end;
                    if (order) {
                        const Quad r = ld_quad(order + rec);
                        start = (int32_t)r.x;
                        end = (int32_t)r.y;
                        p = (int32_t)r.z;
                    } else {
                        start = range[2 * p];
                        end = range[2 * p + 1];
                    }
                    int32_t hits = start < end ? end - start : 0;
                    int32_t limit = max_matches;
                    taken_p = 0;
                    if (taken) {  // segment sets: `taken[p]` hits came from earlier segments (the caller's loop passes maxMatches - taken)
                        taken_p = taken[p];
                        limit = max_matches - taken_p;
                        if (limit <= 0) hits = 0;
                    }
                    // the reference stops at maxMatches (FM:544-546) and overruns `locations` beyond its length (Java AIOOBE)
                    const int32_t wanted = (limit > 0 && hits > limit) ? limit : hits;
                    located = wanted < loc_cap ? wanted : loc_cap;
                    if (k == 0) {
                        found[p] = located;
                        if (wanted > loc_cap && status_out) atomicOr(&status_out[p], ST_JAVA_AIOOBE);
                    }
                }
            }
            if (!walking && k < located) {
                w.j = start + 1 + k;  // FM:527-529
                w.distance = 0;
                w.status = ST_OK;
                walking = true;
            }
            if (!__any(walking ? 1 : 0)) {
                if (next >= run_end) break;
                continue;
            }
            if (walking && fm_locate_steps_win<kForm>(ix, w, burst, walk_limit)) {
                const int32_t at = fm_locate_finish_win(ix, s_inv, w);
                const int64_t dest = (int64_t)p * loc_cap + (set_locs ? taken_p : 0) + k;
                if (set_locs)
                    set_locs[dest] = set_base + at;
                else
                    locs[dest] = at;
                if (lf_steps && w.distance) atomicAdd(&lf_steps[p], w.distance);
                if (w.status && status_out) atomicOr(&status_out[p], w.status);
                walking = false;
            }
        }
    }
}

// FM:564-608.  Pipeline form (slot_found != nullptr): query q is hit (q % slots) of pattern (q / slots) and
// runs only if that hit exists; with stops == nullptr the stop position is min(inputLength, start + fixed_len)
// (the reference's locateAndExtractBenchmark, FmIndexThroughputBenchmark.java:231-249).
template <int kBlock, int kWin>
FMX_EXTRACT_KERNEL(kBlock) void k_extract(DevIndex ix_global, const int32_t *__restrict__ starts, const int32_t *__restrict__ stops,
                                  int64_t n, uint16_t *__restrict__ dst, int32_t dst_len, int32_t offset,
                                  int32_t *__restrict__ out_len, int32_t *__restrict__ lf_steps,
                                  int32_t *__restrict__ status_out, const int32_t *__restrict__ slot_found,
                                  int32_t slots, int32_t fixed_len, const PlanRec *__restrict__ order) {
    FMX_FM_INV(ix_global);
    FMX_WITH_SB_CACHE(ix_global, ix);
    FMX_WITH_C_LDS(ix, kWin);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += stride) {
        // order (nullable; the pipeline form): the hits by text position — hits of equal patterns are equal positions, and equal
        // extractions walked by neighbouring lanes read the same lines (launch_extract)
        const int64_t q = order ? (int64_t)ld_quad(order + t).z : t;
        if (slot_found && (int32_t)(q % slots) >= slot_found[q / slots]) continue;
        int status = ST_OK;
        int32_t steps;
        const int32_t start = starts[q];
        int32_t stop;
        if (stops)
            stop = stops[q];
        else {
            const int64_t e = (int64_t)start + fixed_len;
            stop = e < ix.length ? (int32_t)e : ix.length;
        }
        const int32_t ret = fm_extract<kWin>(ix, s_inv, start, stop, dst + q * (int64_t)dst_len, dst_len, offset, steps, status);
        out_len[q] = status ? 0 : ret;
        if (lf_steps) lf_steps[q] = steps;
        if (status_out) status_out[q] = status;
    }
}

// FM:640-759 (mode 0), FM:772-831 (mode 1), FM:844-922 (mode 2), one lane per query.  `scratch` (nullable) holds
// sample_rate codes per lane of the grid (element j of lane t at scratch[j * lanes + t]) for the
// interval-buffered right walk (fm_boundary_right_blocks); without it the literal form runs.
// redo (nullable): {count, 0, 0, 0, queries...} — the queries the group kernel could not answer (a walk met a quirk path of the
// wavelet tree): only those are run, literally, and their LF-steps are ADDED to what the group kernel already walked for them.
template <int kBlock>
FMX_BOUNDARY_KERNEL(kBlock) void k_extract_boundary(DevIndex ix_global, const int32_t *__restrict__ froms, int64_t n, uint16_t boundary,
                                           int mode, uint16_t *__restrict__ dst, int32_t dst_len, int32_t offset,
                                           int32_t *__restrict__ out_len, int32_t *__restrict__ lf_steps,
                                           int32_t *__restrict__ status_out, int32_t *__restrict__ aux_out,
                                           uint16_t *__restrict__ scratch, const int32_t *__restrict__ slot_found,
                                           int32_t slots, const PlanRec *__restrict__ order,
                                           const int32_t *__restrict__ redo) {
    if (redo) {
        n = redo[0];
        if (n <= 0) return;  // (nearly every launch: no superblock header is staged for nothing)
    }
    FMX_FM_INV(ix_global);
    FMX_WITH_SB_CACHE(ix_global, ix);
    FMX_WITH_C_LDS(ix, kWinAsk);
    const int64_t lanes = (int64_t)gridDim.x * kBlock;
    const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int32_t mapped_boundary = fm_map(ix, boundary);  // FM:658
    for (int64_t t = lane; t < n; t += lanes) {
        const int64_t q = redo ? (int64_t)redo[kRedoHead + t] : order ? (int64_t)ld_quad(order + t).z : t;  // (the queries by text position: launch_extract_boundary)
        if (slot_found && (int32_t)(q % slots) >= slot_found[q / slots]) continue;
        int status = ST_OK;
        int32_t steps, aux;
        const int32_t ret = fm_extract_boundary(ix, s_inv, mode, froms[q], mapped_boundary, dst + q * (int64_t)dst_len,
                                                dst_len, offset, steps, status, aux, scratch ? scratch + lane : nullptr,
                                                lanes);
        out_len[q] = status ? 0 : ret;
        if (lf_steps) lf_steps[q] = redo ? lf_steps[q] + steps : steps;
        if (status_out) status_out[q] = status;
        if (aux_out) aux_out[q] = aux;
    }
}

// Group-cooperative extractUntilBoundary: G lanes per query (fm_extract_boundary_group); the window of a group
// is G consecutive lane columns of `scratch` (left window in the first half, right window in the second).
// kMode: the mode (0 / 1 / 2) as a compile-time constant — each instance carries one mode's replay.  A query whose walks met a
// quirk path (`clean` false) is not answered here: its index goes onto the `redo` list ({count, 0, 0, 0, queries...}), which a
// launch of the literal k_extract_boundary behind this kernel works off — the literal form is not part of this kernel's body.
// kDefer (the NARROW first round, G = 2): a query whose line does not lie inside the two intervals on each side of `from` goes onto
// `redo` as well — which then is the `todo` list of a launch of the wide form (G = 4, kDefer false) behind this one.
// todo (nullable): {count, 0, 0, 0, queries...} — only those queries are run, and their LF-steps are ADDED to what is there.
template <int kBlock, int G, int kMode, int kWin, bool kDefer = false, bool kRounds = false>
FMX_BOUNDARY_KERNEL(kBlock) void k_extract_boundary_group(DevIndex ix_global, const int32_t *__restrict__ froms, int64_t n,
                                                 uint16_t boundary, uint16_t *__restrict__ dst,
                                                 int32_t dst_len, int32_t offset, int32_t *__restrict__ out_len,
                                                 int32_t *__restrict__ lf_steps, int32_t *__restrict__ status_out,
                                                 int32_t *__restrict__ aux_out, uint16_t *__restrict__ scratch,
                                                 const int32_t *__restrict__ slot_found, int32_t slots, int pair_walks,
                                                 const PlanRec *__restrict__ order, int32_t *__restrict__ redo,
                                                 const int32_t *__restrict__ todo) {
    if (todo) {
        n = todo[0];
        if (n <= 0) return;
    }
    FMX_FM_INV(ix_global);
    FMX_WITH_SB_CACHE(ix_global, ix);
    FMX_WITH_C_LDS(ix, kWin);
    const int64_t lanes = (int64_t)gridDim.x * kBlock;
    const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int g = threadIdx.x % G;
    const int64_t groups = lanes / G;
    const int32_t mapped_boundary = fm_map(ix, boundary);  // FM:658
    for (int64_t t = lane / G; t < n; t += groups) {
        // order (nullable): the queries by text position — equal and neighbouring `from` fetch the same sample intervals, and
        // walked by neighbouring groups those walks read the same lines (launch_extract_boundary)
        const int64_t q = todo ? (int64_t)todo[kRedoHead + t] : order ? (int64_t)ld_quad(order + t).z : t;
        if (slot_found && (int32_t)(q % slots) >= slot_found[q / slots]) continue;  // group-uniform
        int status = ST_OK;
        int32_t steps, aux;
        bool clean;
        uint16_t *dest = dst + q * (int64_t)dst_len;
        const int32_t ret = fm_extract_boundary_group<G, kMode, kWin, kDefer, kRounds>(ix, s_inv, kMode, froms[q], mapped_boundary, dest, dst_len, offset,
                                                                        steps, status, aux, scratch + (lane - g), lanes, 1,
                                                                        lanes * (int64_t)ix.sample_rate, g, clean, pair_walks != 0);
        if (g == 0) {
            if (lf_steps) lf_steps[q] = todo ? lf_steps[q] + steps : steps;
            if (!clean) {  // (rare) the literal form answers it: k_extract_boundary over the redo list
                redo[kRedoHead + atomicAdd(&redo[0], 1)] = (int32_t)q;
            } else {
                out_len[q] = status ? 0 : ret;
                if (status_out) status_out[q] = status;
                if (aux_out) aux_out[q] = aux;
            }
        }
    }
}

// WaveletFixedBlockBoosting.rank(position, symbol) WFBB:1010-1285, one lane per query
template <int kBlock>
FMX_KERNEL(kBlock) void k_wt_rank(DevIndex ix, const int64_t *__restrict__ positions, const int32_t *__restrict__ symbols,
                                  int32_t n, int64_t *__restrict__ out, int32_t *__restrict__ status_out) {
    FMX_FM_INV(ix);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride) {
        int status = ST_OK;
        int64_t pos = positions[q];
        const int32_t sym = symbols[q];
        int64_t r = 0;
        if (pos < 0 || sym < 0)
            status = ST_JAVA_AIOOBE;  // negative array index in the reference
        else {
            if (pos > (int64_t)ix.wt_size) pos = ix.wt_size;  // WFBB:1015-1017 (also keeps the position in 32 bits)
            r = wt_rank(ix, s_inv, (uint32_t)pos, sym, status);
        }
        out[q] = r;
        if (status_out) status_out[q] = status;
    }
}

// WaveletFixedBlockBoosting.inverseSelect(position) WFBB:1305-1537: the reference's packed long
// (rank << 32) | symbol, the bare symbol for position 0 (WFBB:1334-1335, 1508-1509)
template <int kBlock>
FMX_KERNEL(kBlock) void k_wt_inverse_select(DevIndex ix, const int64_t *__restrict__ positions, int32_t n,
                                            int64_t *__restrict__ out, int32_t *__restrict__ status_out) {
    FMX_FM_INV(ix);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride) {
        const int64_t pos = positions[q];
        int status = ST_OK;
        int64_t v = 0;
        if (pos < 0 || pos >= (int64_t)ix.wt_size)
            status = ST_JAVA_AIOOBE;
        else {
            int32_t rank;
            const int32_t c = wt_inverse_select(ix, s_inv, (uint32_t)pos, rank);
            v = (pos == 0) ? (int64_t)c : (int64_t)(((uint64_t)(uint32_t)rank << 32) | (uint32_t)c);
        }
        out[q] = v;
        if (status_out) status_out[q] = status;
    }
}

// RrrVector as a stand-alone structure (fmx_rrr_*): the compressed form — 16-block records, offset bit stream,
// and the halved value-of-offset table staged in LDS (32 KiB per workgroup).
// rankOnes(position) RRR:358-396
template <int kBlock>
FMX_KERNEL(kBlock) void k_rrr_rank_ones(DevIndex ix, const int32_t *__restrict__ positions, int32_t n,
                                        int32_t *__restrict__ out) {
    __shared__ uint16_t s_inv[kInvEntries];
    stage_inverse_table(s_inv, ix.inv_global);
    const RrrView v = rrr_view_from(Quad{ix.sampled.off_rec, ix.sampled.off_bits, (uint32_t)ix.sampled.length, (uint32_t)ix.sampled.total_ones});
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride)
        out[q] = rrr_rank1(ix.base, v, s_inv, positions[q]);
}
// access(position) RRR:314-349; an out-of-range position is the reference's exception (status)
template <int kBlock>
FMX_KERNEL(kBlock) void k_rrr_access(DevIndex ix, const int32_t *__restrict__ positions, int32_t n,
                                     uint8_t *__restrict__ out, int32_t *__restrict__ status_out) {
    __shared__ uint16_t s_inv[kInvEntries];
    stage_inverse_table(s_inv, ix.inv_global);
    const RrrView v = rrr_view_from(Quad{ix.sampled.off_rec, ix.sampled.off_bits, (uint32_t)ix.sampled.length, (uint32_t)ix.sampled.total_ones});
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride) {
        int status = ST_OK;
        const bool bit = rrr_access(ix.base, v, s_inv, positions[q], status);
        out[q] = status ? 0 : (bit ? 1 : 0);
        if (status_out) status_out[q] = status;
    }
}

// ---- processing order of a batch (the plan stage) -----------------------------------------------------
// Patterns are processed in (approximate) order of their trailing characters, the LAST character most
// significant (it is consumed first, FM:456-457): lanes of a wave then start their backward search in the
// same SA intervals.  Any grouping works — results are written at the original index — so instead of a
// general device sort (rocPRIM falls back to a 21-launch merge sort at 1 M keys) the order is built by
//   1. k_plan_codes: one pass over the patterns — the 16 bytes of a pattern's tail as aligned dwords, characters
//      mapped through an LDS copy of the map's first 256 entries — leaving a record {code word, key, length, fine
//      bin} per pattern, its coarse key (the top <= 13 key bits), and a histogram of the coarse keys: per-workgroup
//      LDS histograms, ONE global atomic per (workgroup, non-empty bin), no hot-bin serialisation;
//   2. k_plan_scatter: every workgroup scans the histogram itself (no scan kernel), reserves its share of each
//      bin with one atomic and writes the pattern indices in bucket order; the last workgroup to finish zeroes
//      the histogram and the cursors for the next plan on this stream (no memset launch);
//   3. the fine order where it matters — which 32 patterns share a wave — is made inside k_count: a counting
//      sort of each workgroup's 256 records on 10 key bits, in LDS.
// Two short kernels (round 1: memset + four kernels with a tile-local radix sort, 76 us at 1 M patterns).

// (dynamic LDS of k_plan_codes with the order-1 table staged behind the histogram: order1_lds_bytes / kPlanCodesLdsMax in
// fmx_device.hpp — the launcher, the kernel and the table's builder decide alike)

// what the plan kernels keep in LDS beside their histogram (k_plan_codes, k_plan_fused)
struct PlanTables {
    const int16_t *s_map;  // the character map's first 256 entries
    const float2 *s_o1;    // the order-1 table (o1) ...
    const int32_t *s_c;    // ... and cumulativeCounts
    bool o1;
    int sigma, below, fine_shift, bins;
};
__device__ __forceinline__ bool plan_uses_order1(const DevIndex &ix, const SortShape &sh, int code_bits, int bins, size_t extra_lds) {
    return code_bits == 8 && sh.sa_key == 2 && ix.suffix_order1 != nullptr && ix.wt_sigma <= kOrder1MaxSigma &&
           order1_lds_bytes(bins, ix.wt_sigma) + extra_lds <= kPlanCodesLdsMax;
}
// stages the tables behind `after` (the first free LDS word); the caller's barrier follows
__device__ __forceinline__ PlanTables plan_tables_stage(const DevIndex &ix, const SortShape &sh, int16_t *s_map, uint32_t *after, bool o1) {
    PlanTables t;
    t.sigma = ix.wt_sigma;
    t.bins = 1 << sh.coarse_bits;
    t.below = sh.total_bits - sh.coarse_bits;
    t.fine_shift = t.below > 8 ? t.below - 8 : 0;  // the fine bin of a record: kFineBits key bits ending 8 bits below the coarse bits
    t.o1 = o1;
    for (int i = threadIdx.x; i < 256; i += kTileThreads) s_map[i] = ix.char2code[i];
    float2 *s_o1 = reinterpret_cast<float2 *>(after);
    int32_t *s_c = reinterpret_cast<int32_t *>(s_o1 + (o1 ? t.sigma * t.sigma : 0));
    if (o1) {
        const float2 *src = reinterpret_cast<const float2 *>(ix.suffix_order1);
        for (int i = threadIdx.x; i < t.sigma * t.sigma; i += kTileThreads) s_o1[i] = src[i];
        for (int i = threadIdx.x; i <= t.sigma; i += kTileThreads) s_c[i] = i < ix.n_c ? ix.C[i] : ix.length;
    }
    t.s_map = s_map;
    t.s_o1 = s_o1;
    t.s_c = s_c;
    return t;
}

// the record of ONE pattern — {code word, key, length | fine bin} — and its bin of the bucket pass
template <int kCodeBits>
__device__ __forceinline__ Quad plan_record(const DevIndex &ix, const PlanTables &L, const SortShape &sh, const uint16_t *__restrict__ pat,
                                            int32_t beg, int32_t m, const TailWords &tail, uint32_t &bin) {
    const int sigma = L.sigma;
    uint32_t ch[8];
    pattern_tail_chars(tail, pat, beg, m, ch);
    const uint64_t word = pattern_code_word<kCodeBits>(ix, L.s_map, ch, m);
    uint32_t key;
    if (sh.sa_key) {
        // where the pattern's backward search stands when k_count takes it up: the first SA row of its tabulated suffix
        // — 0 for a pattern that ends at once.  sa_key 1: the suffix table's own answer (one lookup at a random slot of
        // the table per pattern: 14 us of the pass); sa_key 2: an ESTIMATE of that row from the table's two-character
        // strings alone (a few hundred slots: cache-resident) — the row range of the suffix's first two characters,
        // narrowed character by character by the share the next character has after its predecessor (an order-1 chain:
        // row ~ s(xy) + |xy| * (F(z|y) + P(z|y) * (F(u|z) + ...)), F / P = where the two-character string yz starts inside
        // y's rows and how much of them it takes).  The estimate is monotone in the suffix's lexicographic order, which
        // is all the bucket pass needs; results never depend on it.
        key = 0;
        constexpr uint32_t cmask = (1u << kCodeBits) - 1u;
        const int32_t c_last = (int32_t)(word & cmask);
        if (m > 0 && c_last != 0) {
            int32_t start = ix.C[c_last], end = 0, back = 0;
            int len = fm_suffix_len(ix, m);
            if (len > 64 / kCodeBits) len = 64 / kCodeBits;
            if (len >= 2 && kCodeBits == ix.suffix_key_bits && sh.sa_key == 1) {
                const int bits = len * kCodeBits;
                const uint64_t tk = bits >= 64 ? word : (word & ((1ull << bits) - 1ull));
                if ((uint32_t)(tk >> (bits - kCodeBits)) != 0u && tk != kSuffixEmpty) (void)fm_suffix_lookup(ix, tk, len, start, end, back);
            } else if (len >= 2 && L.o1) {
                // the same estimate as below from the LDS copy of the order-1 table
                auto code_at = [&](int j) { return (uint32_t)(word >> (j * kCodeBits)) & cmask; };
                uint32_t x = code_at(len - 1), y = code_at(len - 2);
                if (x >= (uint32_t)sigma) x = 0;
                if (y >= (uint32_t)sigma) y = 0;
                float lo = (float)L.s_c[x ? x : ((uint32_t)c_last < (uint32_t)sigma ? c_last : 0)], width = 0.0f;
                if (x != 0 && y != 0) {
                    float2 fp = L.s_o1[x * sigma + y];
                    if (fp.y > 0.0f) {
                        const float nx = (float)(L.s_c[x + 1] - L.s_c[x]);
                        lo = (float)L.s_c[x] + fp.x * nx;
                        width = fp.y * nx;
                        for (int j = len - 3; j >= 0 && width >= 1.0f; --j) {
                            x = y;
                            y = code_at(j);
                            if (y == 0 || y >= (uint32_t)sigma) break;
                            fp = L.s_o1[x * sigma + y];
                            if (fp.y <= 0.0f) break;
                            lo += width * fp.x;
                            width *= fp.y;
                        }
                    }
                }
                start = (int32_t)lo;
                if (start < 0) start = 0;
                if (start > ix.length) start = ix.length;
            } else if (len >= 2 && kCodeBits == ix.suffix_key_bits) {
                // codes of the suffix, w[j] = j characters before the pattern's end; its first character is w[len - 1]
                auto code_at = [&](int j) { return (uint32_t)(word >> (j * kCodeBits)) & cmask; };
                uint32_t x = code_at(len - 1), y = code_at(len - 2);
                float lo = (float)ix.C[x ? x : c_last], width = 0.0f;
                int32_t s2 = 0, e2 = 0;
                if (x != 0 && y != 0 && fm_suffix_lookup(ix, (uint64_t)y | ((uint64_t)x << kCodeBits), 2, s2, e2, back)) {
                    lo = (float)s2;
                    width = (float)(e2 - s2);
                    for (int j = len - 3; j >= 0 && width >= 1.0f; --j) {
                        x = y;
                        y = code_at(j);
                        if (y == 0 || !fm_suffix_lookup(ix, (uint64_t)y | ((uint64_t)x << kCodeBits), 2, s2, e2, back)) break;
                        const float cx = (float)ix.C[x], nx = (float)(ix.C[x + 1] - ix.C[x]);
                        lo += width * ((float)s2 - cx) / nx;
                        width *= (float)(e2 - s2) / nx;
                    }
                }
                start = (int32_t)lo;
                if (start < 0) start = 0;
                if (start > ix.length) start = ix.length;
            }
            key = (uint32_t)start;
        }
    } else {
        key = suffix_key(word, kCodeBits, sh.chars, sh.bits);
    }
    const uint32_t lenf = m < 0 ? 0u : ((uint32_t)m < kPlanLongPattern ? (uint32_t)m : kPlanLongPattern);
    Quad q;
    q.x = (uint32_t)word;
    q.y = (uint32_t)(word >> 32);
    q.z = key;
    q.w = lenf | (((key >> L.fine_shift) & ((1u << kFineBits) - 1u)) << 22);  // (PlanRec.a / .m)
    uint32_t c = key >> L.below;
    if (c >= (uint32_t)L.bins) c = (uint32_t)L.bins - 1u;  // cannot happen for a validated index (codes < 2^bits)
    bin = c;
    return q;
}

// pass 1: records, coarse keys, global histogram
template <int kCodeBits>
__global__ __launch_bounds__(kTileThreads) void k_plan_codes(DevIndex ix, const uint16_t *__restrict__ pat,
                                                             const int32_t *__restrict__ pat_off, int32_t n,
                                                             SortShape sh, PlanRec *__restrict__ recs,
                                                             uint32_t *__restrict__ ghist, uint32_t *__restrict__ mixed,
                                                             uint32_t epoch) {
    extern __shared__ uint32_t s_hist[];
    __shared__ int16_t s_map[256];
    const int bins = 1 << sh.coarse_bits;
    for (int i = threadIdx.x; i < bins; i += kTileThreads) s_hist[i] = 0;
    // sa_key 2 with the order-1 table (small alphabets): the table and cumulativeCounts behind the histogram
    const PlanTables L = plan_tables_stage(ix, sh, s_map, s_hist + bins, plan_uses_order1(ix, sh, kCodeBits, bins, 0));
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
    const int32_t m_first = pat_off[1] - pat_off[0];  // (n >= 1) the batch is of ONE length if every pattern has this one
    bool differs = false;
    constexpr int kGroup = kTileItems < 4 ? kTileItems : 4;  // patterns whose loads a thread keeps in flight together
    for (int k0 = 0; k0 < kTileItems; k0 += kGroup) {
        int32_t beg[kGroup], len[kGroup];
        TailWords tail[kGroup];
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int64_t p = base + (int64_t)(k0 + k) * kTileThreads + threadIdx.x;
            beg[k] = 0;
            len[k] = -1;
            if (p < n) {
                beg[k] = pat_off[p];
                len[k] = pat_off[p + 1] - beg[k];
            }
        }
#pragma unroll
        for (int k = 0; k < kGroup; ++k) tail[k] = pattern_tail_load(pat, beg[k], len[k]);
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int64_t p = base + (int64_t)(k0 + k) * kTileThreads + threadIdx.x;
            if (p >= n) continue;
            differs |= len[k] != m_first;
            uint32_t c;
            const Quad q = plan_record<kCodeBits>(ix, L, sh, pat, beg[k], len[k], tail[k], c);
            *reinterpret_cast<Quad *>(recs + p) = q;
            atomicAdd(&s_hist[c], 1u);
        }
    }
    // (one store per workgroup at most, and none once the word holds the epoch: thousands of stores to one address queue up)
    if (__syncthreads_or(differs ? 1 : 0) && threadIdx.x == 0) {
        volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(mixed);
        if (*flag != epoch) *flag = epoch;
    }
    for (int i = threadIdx.x; i < bins; i += kTileThreads) {
        const uint32_t v = s_hist[i];
        if (v) atomicAdd(&ghist[i], v);
    }
}

// The plan stage as ONE launch (round 5; k_plan_codes + k_plan_scatter: 22 + 21 us at 1 M patterns, two thirds of it
// start-up, drain and the records' round trip through memory).  A workgroup keeps its tile's records in REGISTERS, publishes
// the tile's histogram with RETURNING atomic adds — what an add returns is this tile's first slot inside the bin: the cursor
// pass of k_plan_scatter is gone —, meets the other workgroups at a grid barrier, scans the complete histogram (agent-scope
// loads: the counts live where the atomics executed, not in this XCD's L2) and scatters its records.
// The barrier is BOUNDED: launch_count_plan offers this kernel only to batches of at most one tile per CU, but what else
// runs on the device (another stream's k_count, another process) decides whether all tiles are resident at once.  A workgroup
// that has polled `spin_limit` times raises ABORT in the barrier word; from then on every workgroup writes its records at
// their own indices instead: a valid plan in the caller's order (k_count runs as planned, only unsorted).  The word decides
// for all (see the barrier below).  Histogram, barrier word and ticket are zero again when the last workgroup leaves (as
// k_plan_scatter leaves them).
constexpr uint32_t kPlanAbort = 0x80000000u;
template <int kCodeBits>
__global__ __launch_bounds__(kTileThreads) void k_plan_fused(DevIndex ix, const uint16_t *__restrict__ pat,
                                                             const int32_t *__restrict__ pat_off, int32_t n, SortShape sh,
                                                             PlanRec *__restrict__ recs_out, uint32_t *__restrict__ ghist,
                                                             uint32_t *__restrict__ ticket, uint32_t *__restrict__ mixed,
                                                             uint32_t epoch, uint32_t spin_limit) {
    // [bins] this tile's counts, then its first slots | the tables of the records' keys; behind the barrier, when the records
    // are made: [bins] scan of the histogram
    extern __shared__ uint32_t s_mem[];
    __shared__ int16_t s_map[256];
    __shared__ uint32_t s_wave[kTileThreads / 64];
    __shared__ uint32_t s_go;
    const int bins = 1 << sh.coarse_bits;
    uint32_t *s_cnt = s_mem, *s_scan = s_mem + bins;
    uint32_t *bar = ticket + 3;
    for (int i = threadIdx.x; i < bins; i += kTileThreads) s_cnt[i] = 0;
    const PlanTables L = plan_tables_stage(ix, sh, s_map, s_mem + bins, plan_uses_order1(ix, sh, kCodeBits, bins, 0));
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
    const int32_t m_first = pat_off[1] - pat_off[0];
    bool differs = false;
    Quad rec[kTileItems];
    uint32_t bin[kTileItems], rank[kTileItems];
    {
        int32_t beg[kTileItems], len[kTileItems];
        TailWords tail[kTileItems];
#pragma unroll
        for (int k = 0; k < kTileItems; ++k) {
            const int64_t p = base + (int64_t)k * kTileThreads + threadIdx.x;
            beg[k] = 0;
            len[k] = -1;
            if (p < n) {
                beg[k] = pat_off[p];
                len[k] = pat_off[p + 1] - beg[k];
            }
        }
#pragma unroll
        for (int k = 0; k < kTileItems; ++k) tail[k] = pattern_tail_load(pat, beg[k], len[k]);
#pragma unroll
        for (int k = 0; k < kTileItems; ++k) {
            const int64_t p = base + (int64_t)k * kTileThreads + threadIdx.x;
            bin[k] = 0xffffffffu;
            rank[k] = 0;
            if (p >= n) continue;
            differs |= len[k] != m_first;
            rec[k] = plan_record<kCodeBits>(ix, L, sh, pat, beg[k], len[k], tail[k], bin[k]);
            rec[k].z = (uint32_t)p;  // PlanRec.a of the final order: the pattern's index
            rank[k] = atomicAdd(&s_cnt[bin[k]], 1u);
        }
    }
    if (__syncthreads_or(differs ? 1 : 0) && threadIdx.x == 0) {
        volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(mixed);
        if (*flag != epoch) *flag = epoch;
    }
    // publish: one returning add per (tile, non-empty bin); all of a thread's adds are issued before the first result is used
    for (int i0 = 0; i0 < bins; i0 += kTileThreads * 8) {
        uint32_t got[8], cnt[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * kTileThreads + (int)threadIdx.x;
            cnt[k] = i < bins ? s_cnt[i] : 0u;
            got[k] = cnt[k] ? atomicAdd(&ghist[i], cnt[k]) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * kTileThreads + (int)threadIdx.x;
            if (i < bins) s_cnt[i] = got[k];  // from here on: this tile's first slot inside the bin
        }
    }
    __syncthreads();  // every add of this tile has returned: it is part of the histogram
    if (threadIdx.x == 0) {
        // arrive with ONE add (a compare-and-swap loop here serialised the 256 arrivals: 0.5 ms).  The abort bit can only be
        // set by a compare-and-swap against an INCOMPLETE count, so "bit set" and "count complete without the bit" exclude each
        // other for good: whoever sees the bit gives up (arrivals after it included, whatever their adds do to the count),
        // whoever sees the complete count without it passes.
        uint32_t go = 0;
        uint32_t v = atomicAdd(bar, 1u) + 1u;
        for (uint32_t polls = 0; !go; ++polls) {
            if (v & kPlanAbort)
                go = 2;
            else if (v == gridDim.x)
                go = 1;
            else if (polls >= spin_limit) {
                const uint32_t seen = atomicCAS(bar, v, v | kPlanAbort);
                v = seen == v ? (v | kPlanAbort) : seen;
            } else {
                __builtin_amdgcn_s_sleep(4);
                v = __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        s_go = go;
    }
    __syncthreads();
    if (s_go == 1) {
        // exclusive scan of the complete histogram: coalesced agent-scope loads into LDS, per-thread chunks of consecutive
        // bins, wave scan, wave totals through LDS (k_plan_scatter's scan)
        for (int i = threadIdx.x; i < bins; i += kTileThreads) s_scan[i] = __hip_atomic_load(&ghist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int per = (bins + kTileThreads - 1) / kTileThreads;
        const int lo = (int)threadIdx.x * per;
        uint32_t sum = 0;
        for (int i = lo; i < lo + per && i < bins; ++i) sum += s_scan[i];
        uint32_t incl = sum;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        uint32_t run = before + incl - sum;
        for (int i = lo; i < lo + per && i < bins; ++i) {
            const uint32_t c = s_scan[i];
            s_scan[i] = run;
            run += c;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kTileItems; ++k)
            if (bin[k] != 0xffffffffu)
                *reinterpret_cast<Quad *>(recs_out + (s_scan[bin[k]] + s_cnt[bin[k]] + rank[k])) = rec[k];
    } else {
        // aborted: the records at their own indices — the caller's order, with code words
#pragma unroll
        for (int k = 0; k < kTileItems; ++k)
            if (bin[k] != 0xffffffffu) *reinterpret_cast<Quad *>(recs_out + (base + (int64_t)k * kTileThreads + threadIdx.x)) = rec[k];
    }
    // the last workgroup to leave zeroes histogram, barrier word and ticket for the next plan in this workspace
    __syncthreads();
    if (threadIdx.x == 0) s_go = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_go) {
        for (int i = threadIdx.x; i < bins; i += kTileThreads) ghist[i] = 0;
        if (threadIdx.x == 0) {
            *bar = 0;
            *ticket = 0;
        }
    }
}

// The walk order of locate: a pattern's hits are the rows start..end-1 of its SA range, and the walks of neighbouring rows
// read neighbouring lines for as long as the rows are preceded by the same characters (LF-mapping keeps such rows in order)
// — so k_locate_walk takes the patterns by the first row of their ranges.  Key of a pattern: start (0 = no hits).
// A pattern with nothing to locate — no hits, or (segment sets) its maxMatches already taken from earlier segments — has key
// 0 and goes into a bin of its own in front of the others (bin = 1 + the key's top bits): k_locate_walk gives such a pattern
// one lane instead of one per slot.
__device__ __forceinline__ uint32_t walk_key(int32_t start, int32_t end, const int32_t *__restrict__ taken, int32_t max_matches,
                                             int64_t p) {
    if (start >= end) return 0u;
    if (taken && max_matches - taken[p] <= 0) return 0u;
    return start > 0 ? (uint32_t)start : 1u;
}
__device__ __forceinline__ uint32_t walk_bin(uint32_t key, int below, int bins) {
    if (key == 0u) return 0u;
    const uint32_t c = 1u + (key >> below);
    return c < (uint32_t)bins ? c : (uint32_t)bins - 1u;
}

// what a record of the walk order is made from: the SA range of pattern p (locate) — or, range == nullptr, query p of
// extractUntilBoundary: {from, from + 1}, the order being the TEXT position (queries at equal and neighbouring positions fetch
// the same sample intervals: the same walks side by side); an absent slot of the pipeline form has nothing to do: {0, 0}
__device__ __forceinline__ int2 walk_pair(const int32_t *__restrict__ range, const int32_t *__restrict__ froms,
                                          const int32_t *__restrict__ slot_found, int32_t slots, int64_t p) {
    if (range) return *reinterpret_cast<const int2 *>(range + 2 * p);
    if (slot_found && (int32_t)(p % slots) >= slot_found[p / slots]) return make_int2(0, 0);
    const int32_t f = froms[p];
    return make_int2(f, f < INT32_MAX ? f + 1 : f);
}

// A slot of an LDS counter for every lane that asks (`take`), ONE atomic per wave: the zero bin of the walk order takes most of a
// batch's records in the later segments of a set (patterns whose maxMatches are already found), and 4,096 single adds to one LDS
// word serialised (k_walk_hist 0.7-1.0 ms per 8 M patterns, 5.6 ms per step of configs[4]: profiles/r05_*segments*).
__device__ __forceinline__ uint32_t wave_ticket(uint32_t *ctr, bool take) {
    const uint64_t m = __ballot(take);
    if (!m) return 0;
    const int lane = threadIdx.x & 63, leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(ctr, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// bins: the zero bin + 2^coarse_bits
__global__ __launch_bounds__(kTileThreads) void k_walk_hist(const int32_t *__restrict__ range, int32_t n, int bins, int below,
                                                            const int32_t *__restrict__ taken, int32_t max_matches,
                                                            uint32_t *__restrict__ ghist, const int32_t *__restrict__ froms,
                                                            const int32_t *__restrict__ slot_found, int32_t slots) {
    extern __shared__ uint32_t s_hist[];
    for (int i = threadIdx.x; i < bins; i += kTileThreads) s_hist[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
#pragma unroll
    for (int k = 0; k < kTileItems; ++k) {
        const int64_t p = base + (int64_t)k * kTileThreads + threadIdx.x;
        uint32_t b = 0xffffffffu;  // (no `continue`: every lane of the wave reaches the ballot)
        if (p < n) {
            const int2 r = walk_pair(range, froms, slot_found, slots, p);
            b = walk_bin(walk_key(r.x, r.y, taken, max_matches, p), below, bins);
        }
        (void)wave_ticket(&s_hist[0], b == 0u);  // (the zero bin: one add per wave)
        if (b != 0u && b != 0xffffffffu) atomicAdd(&s_hist[b], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += kTileThreads) {
        const uint32_t v = s_hist[i];
        if (v) atomicAdd(&ghist[i], v);
    }
}

// pass 2: records into bucket order (16-byte scattered writes; gathering them later instead costs a 64-byte line
// through the fabric per record).  ghist = the batch's histogram, cursor = running fill of each bin (zero at entry),
// ticket = workgroups done (zero at entry); all three are zero again when the kernel has finished.
// kFromRange (the walk order of locate, k_walk_hist): the records are made here from the batch's SA ranges — {start, end,
// pattern index, 0}, keyed by start — instead of being read from recs_in.
template <bool kFromRange>
__global__ __launch_bounds__(kTileThreads) void k_plan_scatter(const PlanRec *__restrict__ recs_in,
                                                               const int32_t *__restrict__ range,
                                                               const int32_t *__restrict__ taken, int32_t max_matches,
                                                               int32_t n, int bins, int below,
                                                               uint32_t *__restrict__ ghist,
                                                               uint32_t *__restrict__ cursor,
                                                               uint32_t *__restrict__ ticket,
                                                               PlanRec *__restrict__ recs_out,
                                                               const int32_t *__restrict__ froms = nullptr,
                                                               const int32_t *__restrict__ slot_found = nullptr,
                                                               int32_t slots = 1) {
    const int fine_shift = below > 8 ? below - 8 : 0;  // (kFromRange: the fine bin of a record, as k_plan_codes leaves it)
    extern __shared__ uint32_t s_mem[];  // [bins] exclusive scan of ghist, [bins] this workgroup's counts / slots
    __shared__ uint32_t s_wave[kTileThreads / 64];
    __shared__ uint32_t s_last;
    uint32_t *s_scan = s_mem, *s_cnt = s_mem + bins;
    // (the walk order: how many patterns have nothing to locate = the zero bin, kept for k_locate_walk in the word behind the
    // ticket; the histogram is complete, and stays until the last workgroup has taken its ticket)
    if (kFromRange && blockIdx.x == 0 && threadIdx.x == 0) ticket[1] = ghist[0];
    // this workgroup's items per bin (the loads of the scan below overlap these)
    for (int i = threadIdx.x; i < bins; i += kTileThreads) s_cnt[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
    Quad mine[kTileItems];
    uint32_t bin[kTileItems];
#pragma unroll
    for (int k = 0; k < kTileItems; ++k) {
        const int64_t p = base + (int64_t)k * kTileThreads + threadIdx.x;
        bin[k] = 0xffffffffu;
        if (p < n) {
            if constexpr (kFromRange) {
                const int2 r = walk_pair(range, froms, slot_found, slots, p);
                mine[k].x = (uint32_t)r.x;
                mine[k].y = (uint32_t)r.y;
                mine[k].z = walk_key(r.x, r.y, taken, max_matches, p);
                mine[k].w = ((mine[k].z >> fine_shift) & ((1u << kFineBits) - 1u)) << 22;
                bin[k] = walk_bin(mine[k].z, below, bins);
            } else {
                mine[k] = ld_quad(recs_in + p);
                uint32_t c = mine[k].z >> below;  // PlanRec.a = the key
                if (c >= (uint32_t)bins) c = (uint32_t)bins - 1u;
                bin[k] = c;
            }
            mine[k].z = (uint32_t)p;  // from here on PlanRec.a = the pattern's index
        }
    }
#pragma unroll
    for (int k = 0; k < kTileItems; ++k) {
        if constexpr (kFromRange) (void)wave_ticket(&s_cnt[0], bin[k] == 0u);  // (the walk order's zero bin: one add per wave)
        if (bin[k] != 0xffffffffu && !(kFromRange && bin[k] == 0u)) atomicAdd(&s_cnt[bin[k]], 1u);
    }
    // exclusive scan of the histogram: per-thread chunks of consecutive bins, wave scan, wave totals through LDS
    const int per = (bins + kTileThreads - 1) / kTileThreads;
    const int lo = (int)threadIdx.x * per;
    uint32_t sum = 0;
    for (int i = lo; i < lo + per && i < bins; ++i) sum += ghist[i];
    uint32_t incl = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    uint32_t run = before + incl - sum;
    for (int i = lo; i < lo + per && i < bins; ++i) {
        s_scan[i] = run;
        run += ghist[i];
    }
    __syncthreads();
    // first slot of this workgroup's share of each bin it holds: all of a thread's atomics are issued before the
    // first result is consumed (one round trip to the L2 instead of one per bin)
    for (int i0 = 0; i0 < bins; i0 += kTileThreads * 8) {
        uint32_t got[8], cnt[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * kTileThreads + (int)threadIdx.x;
            cnt[k] = i < bins ? s_cnt[i] : 0u;
            got[k] = cnt[k] ? atomicAdd(&cursor[i], cnt[k]) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * kTileThreads + (int)threadIdx.x;
            if (cnt[k]) s_cnt[i] = s_scan[i] + got[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTileItems; ++k) {
        uint32_t slot = 0;
        if constexpr (kFromRange) slot = wave_ticket(&s_cnt[0], bin[k] == 0u);
        if (bin[k] != 0xffffffffu && !(kFromRange && bin[k] == 0u)) slot = atomicAdd(&s_cnt[bin[k]], 1u);
        if (bin[k] != 0xffffffffu) *reinterpret_cast<Quad *>(recs_out + slot) = mine[k];
    }
    // The last workgroup leaves histogram, cursors and ticket zeroed for the next plan in this workspace.  No fence:
    // every workgroup's histogram loads and cursor atomics have returned before it takes its ticket, and the zeroes
    // only have to be visible to the NEXT kernel on the stream (a __threadfence here writes the L2 back: 75 us).
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last) {
        for (int i = threadIdx.x; i < bins; i += kTileThreads) {
            ghist[i] = 0;
            cursor[i] = 0;
        }
        if (threadIdx.x == 0) *ticket = 0;
    }
}

// pass 3: the fine order, in place.  A workgroup takes a window of kFineWindow consecutive records of the bucket order
// and ranks them by their fine bins with a counting sort in LDS (histogram, scan by one wave, ranked copy), then
// writes the window back in that order: k_count reads the records with coalesced 16-byte loads.  Which patterns
// share a wave is what matters, not their order inside it, so 10 key bits per window are as good as a full sort.
__global__ __launch_bounds__(kFineThreads) void k_plan_fine(PlanRec *__restrict__ recs, int32_t n) {
    constexpr int n_bins = 1 << kFineBits;
    __shared__ uint32_t s_bin[n_bins];
    __shared__ Quad s_rec[kFineWindow];
    for (int i = threadIdx.x; i < n_bins; i += kFineThreads) s_bin[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kFineWindow;
    Quad r[kFineItems];
    uint32_t bin[kFineItems];
#pragma unroll
    for (int k = 0; k < kFineItems; ++k) {
        const int64_t q = base + (int64_t)k * kFineThreads + threadIdx.x;
        bin[k] = 0xffffffffu;
        if (q < n) {
            r[k] = ld_quad(recs + q);
            bin[k] = r[k].w >> 22;
            r[k].w &= kPlanLongPattern;
        }
    }
#pragma unroll
    for (int k = 0; k < kFineItems; ++k)
        if (bin[k] != 0xffffffffu) atomicAdd(&s_bin[bin[k]], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the bins by one wave, n_bins / 64 consecutive bins per lane
        constexpr int kPer = n_bins / 64;
        uint32_t v[kPer], sum = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            v[k] = s_bin[(int)threadIdx.x * kPer + k];
            sum += v[k];
        }
        uint32_t incl = sum;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if ((int)threadIdx.x >= d) incl += t;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            s_bin[(int)threadIdx.x * kPer + k] = run;
            run += v[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kFineItems; ++k)
        if (bin[k] != 0xffffffffu) s_rec[atomicAdd(&s_bin[bin[k]], 1u)] = r[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kFineItems; ++k) {
        const int64_t q = base + (int64_t)k * kFineThreads + threadIdx.x;
        if (q < n) *reinterpret_cast<Quad *>(recs + q) = s_rec[k * kFineThreads + threadIdx.x];
    }
}

// ---- segment sets: one logical text as K independent indexes (a Java int cannot address >= 2^31 chars) ----
// counts add up; a segment's hits are appended after those of the earlier segments, moved by its base
__global__ __launch_bounds__(256) void k_segment_add_counts(int64_t *__restrict__ total, int64_t *__restrict__ lf_total,
                                                           int32_t *__restrict__ status_total,
                                                           const int32_t *__restrict__ counts,
                                                           const int32_t *__restrict__ lf,
                                                           const int32_t *__restrict__ status, int32_t n, int first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    total[i] = (first ? 0 : total[i]) + counts[i];
    if (lf_total) lf_total[i] = (first ? 0 : lf_total[i]) + lf[i];
    if (status_total) {
        const int32_t prev = first ? 0 : status_total[i];
        status_total[i] = prev ? prev : status[i];
    }
}

__global__ __launch_bounds__(256) void k_segment_append_hits(int64_t *__restrict__ locs, int32_t *__restrict__ found,
                                                            int32_t *__restrict__ status_total,
                                                            const int32_t *__restrict__ seg_locs,
                                                            const int32_t *__restrict__ seg_found,
                                                            const int32_t *__restrict__ seg_status, int32_t n,
                                                            int32_t cap, int64_t base, int first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t have = first ? 0 : found[i];
    const int32_t st = seg_status[i];
    const int32_t add = st ? 0 : seg_found[i];
    int32_t k = 0;
    for (; k < add && have + k < cap; ++k) locs[i * (int64_t)cap + have + k] = base + seg_locs[i * (int64_t)cap + k];
    found[i] = have + k;
    if (status_total) {
        const int32_t prev = first ? 0 : status_total[i];
        status_total[i] = prev ? prev : st;
    }
}

// a segment's hits are already in the set's rows (k_locate_walk, set_locs): advance `found` by them — unless the segment raised a
// status for the pattern, whose hits then do not count (as k_segment_append_hits would not have appended them)
__global__ __launch_bounds__(256) void k_segment_commit(int32_t *__restrict__ found, int32_t *__restrict__ status_total,
                                                       const int32_t *__restrict__ seg_found,
                                                       const int32_t *__restrict__ seg_status, int32_t n, int32_t cap, int first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t have = first ? 0 : found[i];
    const int32_t st = seg_status[i];
    int32_t add = st ? 0 : seg_found[i];
    if (add > cap - have) add = cap - have;
    found[i] = have + (add > 0 ? add : 0);
    if (status_total) {
        const int32_t prev = first ? 0 : status_total[i];
        status_total[i] = prev ? prev : st;
    }
}

// ---- launchers (called from fmx_api.cpp) -----------------------------------------------------

// tunables (fmx_set_option): workgroup size and how many workgroups per CU the grid is capped at.  Atomics: a
// launch on one host thread may read them while another thread sets one (results are identical for every
// setting, so a launch that sees a mix of old and new values is still correct).
static std::atomic<int> g_block{512};
static std::atomic<int> g_groups_per_cu{16};
static std::atomic<int> g_walk_queue{8};  // option "walk_queue": locate over a window directory hands its tickets out per wave, this many per lane and run (0: the packed form)
static std::atomic<int> g_walk_queue_min_slots{32};  // option "walk_queue_min_slots": ... for calls with at least this many hit slots per pattern
static std::atomic<int> g_walk_burst{0};  // option "walk_burst": LF-steps between two hand-outs (0 = sample_rate / 4, at least 2)
static std::atomic<int> g_walk_pack{1};  // option "walk_pack": locate over a window directory packs the walks still under way into fewer waves (0: A/B)
static std::atomic<int> g_boundary_accel{1};  // 0 = literal right walk of extractUntilBoundary (A/B and fallback)
static std::atomic<int> g_boundary_group{4};  // lanes per query of extractUntilBoundary (0 = one lane per query)
// first fill of extractUntilBoundary's two text windows: 0 = G intervals on each side, a lane walks one after the other;
// 2 = the same with a lane's two walks interleaved (fm_lf_step2)
static std::atomic<int> g_boundary_first_fill{2};
// option "boundary_narrow" (default OFF, by measurement — round 6, profiles/r06_experiments.txt 2): the narrow first round walks 37.8 M
// LF-steps where the wide form walks 50.9 M on configs[3] and takes 0.855 ms against 0.808: every pass costs a wave's lifetime (64
// dependent steps of one or two HBM round trips: ~0.35 ms whatever the batch), and the second pass pays it again for a quarter of the queries
static std::atomic<int> g_boundary_narrow{0};
static std::atomic<int> g_boundary_narrow_min{4096};   // option "boundary_narrow_min": ... for batches at least this large
static std::atomic<int> g_regroup_by_length{1};  // k_count: workgroups with mixed pattern lengths hand their records out again by length (0: A/B)
static std::atomic<int> g_steps_executed_only{0};  // 1 = d_lf_steps of count() leave out what the suffix table answered
static std::atomic<int> g_suffix_table_use{1};  // 0 = k_count ignores the index's suffix table (A/B)
static std::atomic<int> g_lds_pad_kb{0};   // experiment knob: extra dynamic LDS per workgroup (lowers occupancy)
static std::atomic<int> g_sort_min{16384};  // batches at least this large are processed in suffix-sorted order (0 = never)
// bins of the bucket pass = 2^coarse_bits (<= 14: they live in LDS).  Measured on configs[1] (tools/tune_coarse.py):
// 14 bits: plan 0.091 ms, step 0.304 ms; 12 bits: 0.075 / 0.286 ms; 10 bits: 0.071 / 0.286 ms; 8 bits: 0.069 / 0.294 ms
static std::atomic<int> g_coarse_bits{12};
static std::atomic<int> g_plan_fine{1};  // 0 = skip the window-local fine order (A/B)
// 0 = order by the trailing characters' codes even where a suffix table exists; 1 = by the SA row the table answers; 2 = by an
// estimate of that row from the table's two-character strings (SortShape.sa_key)
static std::atomic<int> g_plan_sa_key{2};
// planned k_count: a batch of one length runs on half the grid (decided on the device from the plan's flag; 0: A/B).
// Tried first (round 5): tiles taken from a counter on a grid of 8 workgroups per CU — the barrier that hands a tile to a
// workgroup's eight waves ties them together: headline 0.138 -> 0.151 ms, series count +3 %.  Dropped.
static std::atomic<int> g_count_halve_uniform{1};
// option "count_lean": planned batches over expanded images run k_count_lean + k_count's list mode instead of k_count.  Default
// OFF, by measurement (round 6, profiles/r06_experiments.txt 1): the lean kernel issues 20 % fewer vector instructions (26.0 M against
// 32.5 M per headline launch, no spill at all) and takes the SAME time (89.0 against 90.9 us; + 4.6 us for the list pass that finds
// its list empty) — k_count is not bound by instruction issue.  Kept as the A/B that showed it.
static std::atomic<int> g_count_lean{0};
// option "boundary_rounds": extractUntilBoundary (both ways, the group of four) fetches the four sample intervals next to `from`
// first and the four further out only for the groups whose line does not end inside those (fmx_device.hpp window_fill_round)
static std::atomic<int> g_boundary_rounds{1};
// 1 = the plan stage of a batch of at most one tile per CU is ONE launch (k_plan_fused); 0 (default) = k_plan_codes +
// k_plan_scatter.  Measured (round 5, configs[1]): step 0.1365 -> 0.1339 ms (-2 %), with two batches in flight 0.109 -> 0.117
// (+7 %: workgroups waiting at the barrier hold their CUs) — not worth a spinning kernel by default.
static std::atomic<int> g_plan_fused{0};
static std::atomic<int> g_plan_spin_limit{4096};  // polls of k_plan_fused's barrier before a workgroup aborts the order (~1 us each)
// locate: batches at least this large walk their hits by the first row of the patterns' SA ranges (0 = always in the caller's
// order).  Measured on configs[1]'s index, <= 16 hits per pattern (tools/locate_order_probe.py): 16,384 patterns +8 % (the two
// or three short kernels in front), 32,768 -5 %, 100,000 -24 %, 1,048,576 -43 %.
static std::atomic<int> g_walk_order_min{32768};
// extractUntilBoundary: batches at least this large take their queries by text position (0 = always the caller's order):
// 100,000 hit locations of configs[3] (40,024 distinct) 1.93 -> 1.74 ms sorted on the host (tools/boundary_order_probe.py)
static std::atomic<int> g_boundary_order_min{32768};
static std::atomic<int> g_walk_fine{1};  // the window-local fine order on top of the buckets (k_plan_fine; 0: A/B)
static std::atomic<int> g_sort_bits{28};    // full key width: floor(sort_bits / bits-per-code) trailing characters

int set_option(const char *name, int value) {
    if (!strcmp(name, "block")) {
        if (value != 512 && value != 1024) return -1;
        g_block = value;
        return 0;
    }
    if (!strcmp(name, "walk_pack")) {  // 0: k_locate_walk; 1 / 2: two packings; 3: three
        if (value < 0 || value > 3) return -1;
        g_walk_pack = value;
        return 0;
    }
    if (!strcmp(name, "walk_queue")) {
        if (value < 0 || value > 64) return -1;
        g_walk_queue = value;
        return 0;
    }
    if (!strcmp(name, "walk_queue_min_slots")) {
        if (value < 0) return -1;
        g_walk_queue_min_slots = value;
        return 0;
    }
    if (!strcmp(name, "walk_burst")) {
        if (value < 0 || value > 1024) return -1;
        g_walk_burst = value;
        return 0;
    }
    if (!strcmp(name, "groups_per_cu")) {
        if (value < 1 || value > 64) return -1;
        g_groups_per_cu = value;
        return 0;
    }
    if (!strcmp(name, "boundary_accel")) {
        g_boundary_accel = value != 0;
        return 0;
    }
    if (!strcmp(name, "boundary_group")) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16) return -1;
        g_boundary_group = value;
        return 0;
    }
    if (!strcmp(name, "regroup_by_length")) {
        g_regroup_by_length = value != 0;
        return 0;
    }
    if (!strcmp(name, "lf_steps_executed_only")) {
        g_steps_executed_only = value != 0;
        return 0;
    }
    if (!strcmp(name, "boundary_first_fill")) {
        if (value != 0 && value != 2) return -1;  // (1, half-width windows, was measured slower in round 3 and is gone)
        g_boundary_first_fill = value;
        return 0;
    }
    if (!strcmp(name, "boundary_narrow")) {
        g_boundary_narrow = value != 0;
        return 0;
    }
    if (!strcmp(name, "boundary_narrow_min")) {
        if (value < 0) return -1;
        g_boundary_narrow_min = value;
        return 0;
    }
    if (!strcmp(name, "suffix_table")) {
        g_suffix_table_use = value != 0;
        return 0;
    }
    if (!strcmp(name, "coarse_bits")) {
        if (value < 4 || value > kCoarseBitsMax - 1) return -1;  // k_plan_scatter keeps two arrays of 2^bits words in LDS
        g_coarse_bits = value;
        return 0;
    }
    if (!strcmp(name, "plan_sa_key")) {
        if (value < 0 || value > 2) return -1;
        g_plan_sa_key = value;
        return 0;
    }
    if (!strcmp(name, "count_halve_uniform")) {
        g_count_halve_uniform = value != 0;
        return 0;
    }
    if (!strcmp(name, "code_bits_12")) {
        g_code_bits_12 = value != 0;
        return 0;
    }
    if (!strcmp(name, "boundary_rounds")) {
        g_boundary_rounds = value != 0;
        return 0;
    }
    if (!strcmp(name, "count_lean")) {
        g_count_lean = value != 0;
        return 0;
    }
    if (!strcmp(name, "plan_fused")) {
        g_plan_fused = value != 0;
        return 0;
    }
    if (!strcmp(name, "plan_spin_limit")) {
        if (value < 0) return -1;
        g_plan_spin_limit = value;
        return 0;
    }
    if (!strcmp(name, "plan_fine")) {
        if (value < 0 || value > 2) return -1;
        g_plan_fine = value;
        return 0;
    }
    if (!strcmp(name, "lds_pad_kb")) {
        if (value < 0 || value > 96) return -1;
        g_lds_pad_kb = value;
        return 0;
    }
    if (!strcmp(name, "sort_min")) {
        if (value < 0) return -1;
        g_sort_min = value;
        return 0;
    }
    if (!strcmp(name, "walk_fine")) {
        g_walk_fine = value != 0;
        return 0;
    }
    if (!strcmp(name, "boundary_order_min")) {
        if (value < 0) return -1;
        g_boundary_order_min = value;
        return 0;
    }
    if (!strcmp(name, "walk_order_min")) {
        if (value < 0) return -1;
        g_walk_order_min = value;
        return 0;
    }
    if (!strcmp(name, "sort_bits")) {
        if (value < 1 || value > 32) return -1;
        g_sort_bits = value;
        return 0;
    }
    return -1;
}

static int grid_for(int64_t lanes, int block, int n_cu) {
    int64_t blocks = (lanes + block - 1) / block;
    const int64_t cap = (int64_t)n_cu * g_groups_per_cu;  // a few rounds of workgroups per CU, grid-stride the rest
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

#define FMX_DISPATCH(KERNEL, LANES, ...)                                                                     \
    do {                                                                                                     \
        const int blk__ = g_block;                                                                           \
        const dim3 grid__(grid_for((LANES), blk__, n_cu));                                                   \
        if (blk__ == 1024)                                                                                   \
            hipLaunchKernelGGL(KERNEL<1024>, grid__, dim3(1024), (size_t)g_lds_pad_kb * 1024, st, __VA_ARGS__);                        \
        else                                                                                                 \
            hipLaunchKernelGGL(KERNEL<512>, grid__, dim3(512), (size_t)g_lds_pad_kb * 1024, st, __VA_ARGS__);                          \
    } while (0)

// ... of a kernel that only runs over a window directory, instantiated per form of it (fmx_device.hpp: kFormCells / kFormFlat)
#define FMX_DISPATCH_FORM(KERNEL, IX, LANES, ...)                                                                      \
    do {                                                                                                               \
        const int blk__ = g_block;                                                                                     \
        const dim3 grid__(grid_for((LANES), blk__, n_cu));                                                             \
        const size_t lds__ = (size_t)g_lds_pad_kb * 1024;                                                              \
        if ((IX).win_flat) {                                                                                           \
            if (blk__ == 1024)                                                                                         \
                hipLaunchKernelGGL((KERNEL<1024, kFormFlat>), grid__, dim3(1024), lds__, st, __VA_ARGS__);            \
            else                                                                                                       \
                hipLaunchKernelGGL((KERNEL<512, kFormFlat>), grid__, dim3(512), lds__, st, __VA_ARGS__);              \
        } else {                                                                                                       \
            if (blk__ == 1024)                                                                                         \
                hipLaunchKernelGGL((KERNEL<1024, kFormCells>), grid__, dim3(1024), lds__, st, __VA_ARGS__);           \
            else                                                                                                       \
                hipLaunchKernelGGL((KERNEL<512, kFormCells>), grid__, dim3(512), lds__, st, __VA_ARGS__);             \
        }                                                                                                              \
    } while (0)

// ... of a walk kernel instantiated for indexes with a window directory (in either form) and without one (fmx_device.hpp: kWinAlways /
// kWinFlat / kWinNever)
#define FMX_DISPATCH_WIN(KERNEL, IX, LANES, ...)                                                                       \
    do {                                                                                                               \
        const int blk__ = g_block;                                                                                     \
        const dim3 grid__(grid_for((LANES), blk__, n_cu));                                                             \
        const size_t lds__ = (size_t)g_lds_pad_kb * 1024;                                                              \
        if ((IX).win && (IX).win_flat) {                                                                               \
            if (blk__ == 1024)                                                                                         \
                hipLaunchKernelGGL((KERNEL<1024, kWinFlat>), grid__, dim3(1024), lds__, st, __VA_ARGS__);             \
            else                                                                                                       \
                hipLaunchKernelGGL((KERNEL<512, kWinFlat>), grid__, dim3(512), lds__, st, __VA_ARGS__);               \
        } else if ((IX).win) {                                                                                         \
            if (blk__ == 1024)                                                                                         \
                hipLaunchKernelGGL((KERNEL<1024, kWinAlways>), grid__, dim3(1024), lds__, st, __VA_ARGS__);           \
            else                                                                                                       \
                hipLaunchKernelGGL((KERNEL<512, kWinAlways>), grid__, dim3(512), lds__, st, __VA_ARGS__);             \
        } else {                                                                                                       \
            if (blk__ == 1024)                                                                                         \
                hipLaunchKernelGGL((KERNEL<1024, kWinNever>), grid__, dim3(1024), lds__, st, __VA_ARGS__);            \
            else                                                                                                       \
                hipLaunchKernelGGL((KERNEL<512, kWinNever>), grid__, dim3(512), lds__, st, __VA_ARGS__);              \
        }                                                                                                              \
    } while (0)

static SortShape sort_shape(const DevIndex &ix) {
    SortShape sh;
    sh.bits = 1;
    while ((1 << sh.bits) < ix.wt_sigma && sh.bits < 15) ++sh.bits;
    const int sort_bits = g_sort_bits, coarse_bits = g_coarse_bits;
    sh.chars = sort_bits / sh.bits;
    if (sh.chars < 1) sh.chars = 1;
    if (sh.chars > 64 / plan_code_bits(ix.wt_sigma)) sh.chars = 64 / plan_code_bits(ix.wt_sigma);
    sh.total_bits = sh.chars * sh.bits;
    sh.sa_key = 0;
    if (g_plan_sa_key && ix.suffix_table && g_suffix_table_use) {  // order by SA row (SortShape.sa_key)
        sh.sa_key = g_plan_sa_key;
        sh.total_bits = 1;
        while (sh.total_bits < 32 && (1ll << sh.total_bits) <= (long long)ix.length) ++sh.total_bits;
    }
    sh.coarse_bits = sh.total_bits < coarse_bits ? sh.total_bits : coarse_bits;
    return sh;
}

// workspace layout: head (ghist[2^14] cursor[2^14] ticket, kPlanHeadBytes, all zero between plans) | records by pattern
// [n] (16 B) | records in processing order [n] (16 B)
// bytes of scratch needed to order a batch of n patterns (0 = the batch is not sorted)
size_t count_workspace_bytes(const DevIndex &ix, int32_t n) {
    const int sort_min = g_sort_min;
    if (sort_min <= 0 || n < sort_min) return 0;
    return kPlanHeadBytes + (size_t)n * 2 * sizeof(PlanRec) + 64;
}

// Orders a batch: plan->recs = the patterns' records in processing order (nullptr: batch too small, or no
// workspace).  workspace: count_workspace_bytes(ix, n); head_is_zero: the workspace's head is known to be zero
// (a per-stream workspace keeps that invariant itself), else it is cleared first.  Returns a hipError_t value.
int launch_count_plan(const DevIndex &ix, int n_cu, const uint16_t *pat, const int32_t *off, int32_t n, void *workspace,
                      size_t workspace_bytes, bool head_is_zero, CountPlan *plan, hipStream_t st) {
    *plan = CountPlan();
    const size_t need = count_workspace_bytes(ix, n);
    if (n <= 0 || !workspace || need == 0 || workspace_bytes < need) return 0;
    const SortShape sh = sort_shape(ix);
    const int bins = 1 << sh.coarse_bits;
    uint8_t *wsb = static_cast<uint8_t *>(workspace);
    uint32_t *ghist = reinterpret_cast<uint32_t *>(wsb);
    uint32_t *cursor = ghist + (1 << kCoarseBitsMax);
    uint32_t *ticket = cursor + (1 << kCoarseBitsMax);
    PlanRec *recs = reinterpret_cast<PlanRec *>(wsb + kPlanHeadBytes);
    PlanRec *ordered = recs + n;
    if (!head_is_zero) {
        hipError_t e = hipMemsetAsync(workspace, 0, kPlanHeadBytes, st);
        if (e != hipSuccess) return (int)e;
    }
    const int tiles = (n + kTile - 1) / kTile;
    const int code_bits = plan_code_bits(ix.wt_sigma);
    static std::atomic<uint32_t> plan_epoch{0};
    uint32_t epoch = ++plan_epoch;
    if (epoch == 0) epoch = ++plan_epoch;  // (0 = the value of a fresh workspace)
    // (the order-1 table and cumulativeCounts behind the histogram: k_plan_codes' `o1`)
    const bool o1 = code_bits == 8 && sh.sa_key == 2 && ix.suffix_order1 && ix.wt_sigma <= kOrder1MaxSigma &&
                    order1_lds_bytes(bins, ix.wt_sigma) <= kPlanCodesLdsMax;
    const size_t lds_codes = o1 ? order1_lds_bytes(bins, ix.wt_sigma) : (size_t)bins * 4;
    // ONE launch (k_plan_fused) while every tile can be resident at once — a tile per CU at most — and no fine pass follows (its
    // window order needs the bucket order complete); else k_plan_codes + k_plan_scatter
    const bool fine_pass = g_plan_fine == 2 || (g_plan_fine == 1 && !sh.sa_key);
    if (g_plan_fused && !fine_pass && n_cu > 0 && tiles <= n_cu) {
        const size_t lds_fused = std::max(lds_codes, (size_t)bins * 8);
        const uint32_t spin_limit = (uint32_t)g_plan_spin_limit.load();
        if (code_bits == 8)
            hipLaunchKernelGGL(k_plan_fused<8>, dim3(tiles), dim3(kTileThreads), lds_fused, st, ix, pat, off, n, sh, ordered, ghist, ticket,
                               ticket + 2, epoch, spin_limit);
        else if (code_bits == 12)
            hipLaunchKernelGGL(k_plan_fused<12>, dim3(tiles), dim3(kTileThreads), (size_t)bins * 8, st, ix, pat, off, n, sh, ordered, ghist,
                               ticket, ticket + 2, epoch, spin_limit);
        else
            hipLaunchKernelGGL(k_plan_fused<16>, dim3(tiles), dim3(kTileThreads), (size_t)bins * 8, st, ix, pat, off, n, sh, ordered, ghist,
                               ticket, ticket + 2, epoch, spin_limit);
    } else {
    if (code_bits == 8)
        hipLaunchKernelGGL(k_plan_codes<8>, dim3(tiles), dim3(kTileThreads), lds_codes, st, ix, pat, off, n, sh, recs,
                           ghist, ticket + 2, epoch);
    else if (code_bits == 12)
        hipLaunchKernelGGL(k_plan_codes<12>, dim3(tiles), dim3(kTileThreads), (size_t)bins * 4, st, ix, pat, off, n, sh, recs,
                           ghist, ticket + 2, epoch);
    else
        hipLaunchKernelGGL(k_plan_codes<16>, dim3(tiles), dim3(kTileThreads), (size_t)bins * 4, st, ix, pat, off, n, sh, recs,
                           ghist, ticket + 2, epoch);
    // (a failed launch here would leave the histogram filled: the caller then clears the workspace's head)
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_plan_scatter<false>, dim3(tiles), dim3(kTileThreads), (size_t)bins * 8, st, recs, nullptr, nullptr, 0, n, bins,
                       sh.total_bits - sh.coarse_bits, ghist, cursor, ticket, ordered);
    }
    // (the fine pass: for the code key; with the SA-row key 4,096 buckets already are what a full sort gives within 5 %: option 2 forces it)
    if (g_plan_fine == 2 || (g_plan_fine == 1 && !sh.sa_key))
        hipLaunchKernelGGL(k_plan_fine, dim3((n + kFineWindow - 1) / kFineWindow), dim3(kFineThreads), 0, st, ordered, n);
    plan->recs = ordered;
    // (the records by pattern are dead once the order is made: their place serves k_count_lean's redo list; its two counters sit
    // in the head, zero between launches like the rest of it)
    plan->redo_list = reinterpret_cast<int32_t *>(recs);
    plan->redo_count = ticket + 8;
    plan->mixed = ticket + 2;
    plan->epoch = epoch;
    plan->n = n;
    plan->code_bits = code_bits;
    plan->shape = sh;
    plan->look_up = ix.look_up;
    plan->sigma = ix.wt_sigma;
    return (int)hipGetLastError();
}

// plan (nullable): the batch's order and code words; plan_is_foreign: it was made with ANOTHER index of a segment
// set, so its code words are in that index's alphabet (translated in the kernel, or ignored for 16-bit codes)
int launch_count(const DevIndex &ix, int n_cu, const uint16_t *pat, const int32_t *off, const CountPlan *plan,
                 bool plan_is_foreign, int32_t n, int32_t *counts, int32_t *lf, int32_t *status, int32_t *range,
                 hipStream_t st) {
    if (n <= 0) return 0;
    const PlanRec *recs = (plan && plan->recs && plan->n == n) ? static_cast<const PlanRec *>(plan->recs) : nullptr;
    CountPlan none;
    const CountPlan &pl = recs ? *plan : none;
    // a foreign plan's 8-bit code words can be translated only if THIS index's codes fit 8 bits as well: refilled chunks are
    // made with this index's own alphabet at the kernel's code width (a 300-symbol segment beside an ASCII segment 0)
    const bool translate = recs && plan_is_foreign && pl.code_bits == 8 && plan_code_bits(ix.wt_sigma) == 8;
    const int mode = !recs ? 0 : (!plan_is_foreign ? 1 : (translate ? 2 : 3));
    DevIndex ix_launch = ix;
    if (!g_suffix_table_use) ix_launch.suffix_table = nullptr;  // (A/B: the same index without its table)
    // code width: the plan's in modes 1 / 2 (its record words), this index's own where the kernel makes the chunks itself
    const int bits = (mode == 1 || mode == 2) ? pl.code_bits : plan_code_bits(ix.wt_sigma);
#define FMX_COUNT_LAUNCH(BLOCK, MODE, BITS)                                                                           \
    hipLaunchKernelGGL((k_count<BLOCK, MODE, BITS>), grid__, dim3(BLOCK), (size_t)g_lds_pad_kb * 1024, st, ix_launch, pat, \
                       off, recs, n, counts, lf, status, range, pl.look_up, pl.sigma, (int)g_steps_executed_only, (int)g_regroup_by_length, recs ? pl.mixed : nullptr, pl.epoch, (int)g_count_halve_uniform, (const int32_t *)nullptr, (uint32_t *)nullptr)
#define FMX_COUNT_MODE(MODE)                                                                                       \
    do {                                                                                                           \
        const int blk__ = g_block;                                                                                 \
        const dim3 grid__(grid_for(2 * (int64_t)n, blk__, n_cu));                                                  \
        if (blk__ == 1024 && bits == 8)                                                                            \
            FMX_COUNT_LAUNCH(1024, MODE, 8);                                                                       \
        else if (blk__ == 1024 && bits == 12)                                                                      \
            FMX_COUNT_LAUNCH(1024, MODE, 12);                                                                      \
        else if (blk__ == 1024)                                                                                    \
            FMX_COUNT_LAUNCH(1024, MODE, 16);                                                                      \
        else if (bits == 8)                                                                                        \
            FMX_COUNT_LAUNCH(512, MODE, 8);                                                                        \
        else if (bits == 12)                                                                                       \
            FMX_COUNT_LAUNCH(512, MODE, 12);                                                                       \
        else                                                                                                       \
            FMX_COUNT_LAUNCH(512, MODE, 16);                                                                       \
    } while (0)
#if defined(FMX_DIAG_LINES)
    {
        const unsigned long long base = (unsigned long long)ix.base;
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_diag_base), &base, 8, 0, hipMemcpyHostToDevice, st);
    }
#endif
#if !FMX_COMPACT
    // a planned batch over an expanded image whose superblock headers fit LDS: the lean kernel (option "count_lean" = 0: A/B)
    if (mode == 1 && g_count_lean && pl.redo_list && pl.redo_count && ix.n_sb <= kSbCacheMax && ix.n_sb <= ix.sb_cache_limit &&
        (!ix_launch.suffix_table || pl.code_bits == ix.suffix_key_bits)) {
        CountLeanArgs a;
        a.hot.base = ix.base;
        a.hot.sbc = ix.sbc;
        a.hot.wt_sigma = ix.wt_sigma;
        a.hot.n_sb = ix.n_sb;
        a.hot.wt_size = ix.wt_size;
        a.hot.n = n;
        a.tile.C = ix.C;
        a.tile.suffix_table = ix_launch.suffix_table;
        a.tile.char2code = ix.char2code;
        a.tile.redo_list = pl.redo_list;
        a.tile.redo_count = pl.redo_count;
        a.tile.pat = pat;
        a.tile.pat_off = off;
        a.tile.recs = recs;
        a.tile.counts = counts;
        a.tile.lf_steps = lf;
        a.tile.status_out = status;
        a.tile.range_out = range;
        a.tile.suffix_chars = ix.suffix_chars;
        a.tile.suffix_shift = ix.suffix_shift;
        a.tile.suffix_mask = ix.suffix_mask;
        a.tile.steps_mode = (int)g_steps_executed_only;
        a.tile.mixed = 0;
        a.sbd = ix.sbd;
        a.plan_mixed = pl.mixed;
        a.plan_epoch = pl.epoch;
        a.regroup = (int)g_regroup_by_length;
        a.halve_uniform = (int)g_count_halve_uniform;
        const int blk = g_block;
        const dim3 grid(grid_for(2 * (int64_t)n, blk, n_cu));
        const size_t lds = (size_t)g_lds_pad_kb * 1024;
#define FMX_LEAN_LAUNCH(BLOCK, BITS, BYSYM) hipLaunchKernelGGL((k_count_lean<BLOCK, BITS, BYSYM>), grid, dim3(BLOCK), lds, st, a)
#define FMX_LEAN_SHAPE(BLOCK)                                    \
    do {                                                         \
        if (bits == 8 && ix.map_by_symbol)                       \
            FMX_LEAN_LAUNCH(BLOCK, 8, true);                     \
        else if (bits == 8)                                      \
            FMX_LEAN_LAUNCH(BLOCK, 8, false);                    \
        else if (bits == 12 && ix.map_by_symbol)                 \
            FMX_LEAN_LAUNCH(BLOCK, 12, true);                    \
        else if (bits == 12)                                     \
            FMX_LEAN_LAUNCH(BLOCK, 12, false);                   \
        else if (ix.map_by_symbol)                               \
            FMX_LEAN_LAUNCH(BLOCK, 16, true);                    \
        else                                                     \
            FMX_LEAN_LAUNCH(BLOCK, 16, false);                   \
    } while (0)
        if (blk == 1024)
            FMX_LEAN_SHAPE(1024);
        else
            FMX_LEAN_SHAPE(512);
#undef FMX_LEAN_SHAPE
#undef FMX_LEAN_LAUNCH
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
        // the redo list: k_count in list mode on a small grid (a workgroup per CU: the list is short, and mostly empty)
        const int own_bits = plan_code_bits(ix.wt_sigma);
        const dim3 redo_grid(n_cu > 0 ? (unsigned)n_cu : 256u);
#define FMX_REDO_LAUNCH(BLOCK, BITS)                                                                                              \
    hipLaunchKernelGGL((k_count<BLOCK, 4, BITS>), redo_grid, dim3(BLOCK), lds, st, ix_launch, pat, off, (const PlanRec *)nullptr, n, \
                       counts, lf, status, range, (const int32_t *)nullptr, 0, (int)g_steps_executed_only, 0, (const uint32_t *)nullptr, \
                       0u, 0, (const int32_t *)pl.redo_list, pl.redo_count)
        if (blk == 1024 && own_bits == 8)
            FMX_REDO_LAUNCH(1024, 8);
        else if (blk == 1024 && own_bits == 12)
            FMX_REDO_LAUNCH(1024, 12);
        else if (blk == 1024)
            FMX_REDO_LAUNCH(1024, 16);
        else if (own_bits == 8)
            FMX_REDO_LAUNCH(512, 8);
        else if (own_bits == 12)
            FMX_REDO_LAUNCH(512, 12);
        else
            FMX_REDO_LAUNCH(512, 16);
#undef FMX_REDO_LAUNCH
        return (int)hipGetLastError();
    }
#endif
    if (mode == 0)
        FMX_COUNT_MODE(0);
    else if (mode == 1)
        FMX_COUNT_MODE(1);
    else if (mode == 2)
        FMX_COUNT_MODE(2);
    else
        FMX_COUNT_MODE(3);
#undef FMX_COUNT_LAUNCH
#undef FMX_COUNT_MODE
    return (int)hipGetLastError();
}

// bytes of scratch for the walk order of a batch of n patterns (0 = the batch is walked in the caller's order):
// head (as a plan's, kPlanHeadBytes, zero between uses) | records in walk order [n]
size_t walk_workspace_bytes(const DevIndex &ix, int32_t n) {
    const int walk_min = g_walk_order_min;
    if (walk_min <= 0 || n < walk_min) return 0;
    return kPlanHeadBytes + (size_t)n * sizeof(PlanRec) + 64;
}

// workspace (nullable): walk_workspace_bytes(ix, n) — the patterns are then walked by the first row of their ranges
int launch_locate_walk(const DevIndex &ix, int n_cu, const int32_t *range, int32_t n, int32_t max_matches,
                       int32_t *locs, int32_t loc_cap, int32_t *found, int32_t *lf, int32_t *status,
                       const int32_t *taken, void *workspace, size_t workspace_bytes, bool head_is_zero, hipStream_t st,
                       int64_t *set_locs, int64_t set_base) {
    if (n <= 0) return 0;
    int32_t slots = (max_matches > 0 && max_matches < loc_cap) ? max_matches : loc_cap;
    if (slots < 1) slots = 1;
    const PlanRec *order = nullptr;
    const uint32_t *order_idle = nullptr;
    const size_t need = walk_workspace_bytes(ix, n);
    if (workspace && need != 0 && workspace_bytes >= need && loc_cap > 0) {
        uint8_t *wsb = static_cast<uint8_t *>(workspace);
        uint32_t *ghist = reinterpret_cast<uint32_t *>(wsb);
        uint32_t *cursor = ghist + (1 << kCoarseBitsMax);
        uint32_t *ticket = cursor + (1 << kCoarseBitsMax);
        PlanRec *ordered = reinterpret_cast<PlanRec *>(wsb + kPlanHeadBytes);
        if (!head_is_zero) {
            hipError_t e = hipMemsetAsync(workspace, 0, kPlanHeadBytes, st);
            if (e != hipSuccess) return (int)e;
        }
        int total_bits = 1;
        while (total_bits < 32 && (1ll << total_bits) <= (long long)ix.length) ++total_bits;
        const int coarse_bits = total_bits < g_coarse_bits ? total_bits : (int)g_coarse_bits;
        const int bins = (1 << coarse_bits) + 1, below = total_bits - coarse_bits;  // (+ the bin of patterns with nothing to locate)
        const int tiles = (n + kTile - 1) / kTile;
        hipLaunchKernelGGL(k_walk_hist, dim3(tiles), dim3(kTileThreads), (size_t)bins * 4, st, range, n, bins, below, taken,
                           max_matches, ghist, nullptr, nullptr, 1);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k_plan_scatter<true>, dim3(tiles), dim3(kTileThreads), (size_t)bins * 8, st, nullptr, range, taken,
                           max_matches, n, bins, below, ghist, cursor, ticket, ordered);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
        if (g_walk_fine)
            hipLaunchKernelGGL(k_plan_fine, dim3((n + kFineWindow - 1) / kFineWindow), dim3(kFineThreads), 0, st, ordered, n);
        order = ordered;
        order_idle = ticket + 1;
    }
    const int64_t tickets = (int64_t)n * (slots < kWalkLanes ? slots : kWalkLanes);
    // a window directory and many hits per pattern: the ticket-queue form (k_locate_walk_q).  Measured (round 6, profiles/r06_experiments.txt
    // 3): locate(100) of the reference-shaped series 2.47 -> 2.14-2.26 ms; with <= 16 hits per pattern the hand-out's record loads cost
    // more than the idle lanes they save (configs[2] 0.247 -> 0.34 ms, locate(1) 1.09 -> 1.14): those keep the packed form below
    if (ix.win && g_walk_queue.load() && ix.sample_rate >= 4 && slots >= g_walk_queue_min_slots.load()) {
        const int64_t waves = (tickets + 63) / 64;  // a lane per ticket would need this many waves: give each wave `per` lanes' worth
        int32_t per = g_walk_queue.load();          // tickets per lane and run (option "walk_queue": 0 = off)
        const int32_t chunk = 64 * per;
        int32_t burst = g_walk_burst.load();
        if (burst <= 0) burst = ix.sample_rate / 4 > 2 ? ix.sample_rate / 4 : 2;
        (void)waves;
        FMX_DISPATCH_FORM(k_locate_walk_q, ix, (tickets + per - 1) / per, ix, range, n, max_matches, locs, loc_cap, slots, found, lf, status, taken, order,
                     order_idle, set_locs, set_base, chunk, burst);
        return (int)hipGetLastError();
    }
    if (ix.win && g_walk_pack.load() && ix.sample_rate >= 8) {  // a window directory: the packed form (k_locate_walk_c)
        const int packings = g_walk_pack.load() >= 3 ? 3 : 2;
        FMX_DISPATCH_FORM(k_locate_walk_c, ix, tickets, ix, range, n, max_matches, locs, loc_cap, slots, found, lf, status, taken, order, order_idle,
                     set_locs, set_base, packings);
        return (int)hipGetLastError();
    }
    FMX_DISPATCH_WIN(k_locate_walk, ix, tickets, ix, range, n, max_matches, locs, loc_cap,
                     slots, found, lf, status, taken, order, order_idle, set_locs, set_base);
    return (int)hipGetLastError();
}

// pat_off of a run of equal-length patterns, made on the device (the host-buffer pipeline then does not ship it)
__global__ __launch_bounds__(256) void k_fill_offsets(int32_t *__restrict__ off, int32_t first, int32_t m, int32_t count) {
    const int32_t i = (int32_t)blockIdx.x * 256 + (int32_t)threadIdx.x;
    if (i < count) off[i] = first + i * m;
}
int launch_fill_offsets(int32_t *off, int32_t first, int32_t m, int32_t count, hipStream_t st) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_fill_offsets, dim3((count + 255) / 256), dim3(256), 0, st, off, first, m, count);
    return (int)hipGetLastError();
}

int launch_segment_add_counts(int64_t *total, int64_t *lf_total, int32_t *status_total, const int32_t *counts,
                              const int32_t *lf, const int32_t *status, int32_t n, int first, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_segment_add_counts, dim3((n + 255) / 256), dim3(256), 0, st, total, lf_total, status_total, counts,
                       lf, status, n, first);
    return (int)hipGetLastError();
}

int launch_segment_append_hits(int64_t *locs, int32_t *found, int32_t *status_total, const int32_t *seg_locs,
                               const int32_t *seg_found, const int32_t *seg_status, int32_t n, int32_t cap, int64_t base,
                               int first, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_segment_append_hits, dim3((n + 255) / 256), dim3(256), 0, st, locs, found, status_total, seg_locs,
                       seg_found, seg_status, n, cap, base, first);
    return (int)hipGetLastError();
}

int launch_segment_commit(int32_t *found, int32_t *status_total, const int32_t *seg_found, const int32_t *seg_status, int32_t n,
                          int32_t cap, int first, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_segment_commit, dim3((n + 255) / 256), dim3(256), 0, st, found, status_total, seg_found, seg_status, n, cap,
                       first);
    return (int)hipGetLastError();
}

int launch_wt_rank(const DevIndex &ix, int n_cu, const int64_t *pos, const int32_t *sym, int32_t n, int64_t *out,
                   int32_t *status, hipStream_t st) {
    if (n <= 0) return 0;
    FMX_DISPATCH(k_wt_rank, (int64_t)n, ix, pos, sym, n, out, status);
    return (int)hipGetLastError();
}

int launch_wt_inverse_select(const DevIndex &ix, int n_cu, const int64_t *pos, int32_t n, int64_t *out, int32_t *status,
                             hipStream_t st) {
    if (n <= 0) return 0;
    FMX_DISPATCH(k_wt_inverse_select, (int64_t)n, ix, pos, n, out, status);
    return (int)hipGetLastError();
}

int launch_rrr_rank_ones(const DevIndex &ix, int n_cu, const int32_t *pos, int32_t n, int32_t *out, hipStream_t st) {
    if (n <= 0) return 0;
    FMX_DISPATCH(k_rrr_rank_ones, (int64_t)n, ix, pos, n, out);
    return (int)hipGetLastError();
}
int launch_rrr_access(const DevIndex &ix, int n_cu, const int32_t *pos, int32_t n, uint8_t *out, int32_t *status,
                      hipStream_t st) {
    if (n <= 0) return 0;
    FMX_DISPATCH(k_rrr_access, (int64_t)n, ix, pos, n, out, status);
    return (int)hipGetLastError();
}

// bytes of scratch for taking n extractUntilBoundary queries by text position (0 = the caller's order): head | records [n]
size_t boundary_order_bytes(const DevIndex &ix, int64_t n) {
    const int order_min = g_boundary_order_min;
    if (order_min <= 0 || n < order_min || n > INT32_MAX) return 0;
    return kPlanHeadBytes + (size_t)n * sizeof(PlanRec) + 64;
}

// the three passes of a walk order keyed by text position: records {position, position + 1, query} into `ordered`
static int launch_position_order(const DevIndex &ix, const int32_t *positions, int32_t n, const int32_t *slot_found, int32_t slots,
                                 void *order_ws, bool head_is_zero, PlanRec *ordered, hipStream_t st) {
    uint8_t *wsb = static_cast<uint8_t *>(order_ws);
    uint32_t *ghist = reinterpret_cast<uint32_t *>(wsb);
    uint32_t *cursor = ghist + (1 << kCoarseBitsMax);
    uint32_t *ticket = cursor + (1 << kCoarseBitsMax);
    if (!head_is_zero) {
        hipError_t e = hipMemsetAsync(order_ws, 0, kPlanHeadBytes, st);
        if (e != hipSuccess) return (int)e;
    }
    int total_bits = 1;
    while (total_bits < 32 && (1ll << total_bits) <= (long long)ix.length) ++total_bits;
    const int coarse_bits = total_bits < g_coarse_bits ? total_bits : (int)g_coarse_bits;
    const int bins = (1 << coarse_bits) + 1, below = total_bits - coarse_bits;
    const int tiles = (n + kTile - 1) / kTile;
    hipLaunchKernelGGL(k_walk_hist, dim3(tiles), dim3(kTileThreads), (size_t)bins * 4, st, nullptr, n, bins, below, nullptr, 0, ghist,
                       positions, slot_found, slots < 1 ? 1 : slots);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_plan_scatter<true>, dim3(tiles), dim3(kTileThreads), (size_t)bins * 8, st, nullptr, nullptr, nullptr, 0, n, bins,
                       below, ghist, cursor, ticket, ordered, positions, slot_found, slots < 1 ? 1 : slots);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_plan_fine, dim3((n + kFineWindow - 1) / kFineWindow), dim3(kFineThreads), 0, st, ordered, n);
    return (int)hipGetLastError();
}

// order_ws (nullable): boundary_order_bytes(ix, n) — only offered for the pipeline form (slot_found != nullptr), whose
// positions are hits and repeat; extractions at random positions gain nothing from an order (profiles/r04_experiments.txt 10)
int launch_extract(const DevIndex &ix, int n_cu, const int32_t *start, const int32_t *stop, int64_t n, uint16_t *dst,
                   int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf, int32_t *status,
                   const int32_t *slot_found, int32_t slots, int32_t fixed_len, void *order_ws, size_t order_ws_bytes,
                   bool head_is_zero, hipStream_t st) {
    if (n <= 0) return 0;
    const PlanRec *order = nullptr;
    const size_t order_need = slot_found ? boundary_order_bytes(ix, n) : 0;
    if (order_ws && order_need != 0 && order_ws_bytes >= order_need) {
        PlanRec *ordered = reinterpret_cast<PlanRec *>(static_cast<uint8_t *>(order_ws) + kPlanHeadBytes);
        if (int e = launch_position_order(ix, start, (int32_t)n, slot_found, slots, order_ws, head_is_zero, ordered, st)) return e;
        order = ordered;
    }
    FMX_DISPATCH_WIN(k_extract, ix, n, ix, start, stop, n, dst, dst_len, offset, out_len, lf, status, slot_found, slots, fixed_len, order);
    return (int)hipGetLastError();
}

// lanes the extractUntilBoundary grid will run with, and the scratch it needs (sample_rate codes per lane).
// The options are read ONCE per decision (BoundaryShape) so that a concurrent fmx_set_option cannot make the
// launch disagree with the workspace it was sized for.
struct BoundaryShape {
    int block, group, accel;
};
static BoundaryShape boundary_shape() {
    BoundaryShape b;
    b.block = g_block;
    b.accel = g_boundary_accel;
    b.group = b.accel ? (int)g_boundary_group : 0;  // 0 = one lane, literal/serial forms
    return b;
}
static size_t boundary_bytes_for_grid(const DevIndex &ix, int blocks, const BoundaryShape &b) {
    return (size_t)blocks * (size_t)b.block * (size_t)ix.sample_rate * sizeof(uint16_t) * 2 + 256;  // two windows
}
// (behind the windows: the redo list of the group kernel — {count, 0, 0, 0} + one int per query)
static size_t boundary_redo_bytes(int64_t n) { return ((size_t)kRedoHead + (size_t)n) * sizeof(int32_t) + 64; }
static size_t boundary_bytes_for(const DevIndex &ix, int64_t n, int n_cu, const BoundaryShape &b) {
    if (!b.accel || n <= 0) return 0;
    // (two lists: the narrow round's `todo` for the wide form, and the wide form's redo for the literal one)
    return boundary_bytes_for_grid(ix, grid_for(n * (b.group ? b.group : 1), b.block, n_cu), b) + 2 * boundary_redo_bytes(n) + 32;
}
size_t boundary_workspace_bytes(const DevIndex &ix, int64_t n, int n_cu) {
    // upper bound over the workgroup sizes: whatever shape the launch snapshots fits
    BoundaryShape b = boundary_shape();
    size_t need = 0;
    for (int blk : {512, 1024}) {
        b.block = blk;
        const size_t v = boundary_bytes_for(ix, n, n_cu, b);
        if (v > need) need = v;
    }
    return need;  // (a racing change of groups_per_cu / boundary_group at worst makes the launch take the literal form)
}

// order_ws (nullable): boundary_order_bytes(ix, n) — the queries are then taken by their text position
int launch_extract_boundary(const DevIndex &ix, int n_cu, const int32_t *from, int64_t n, uint16_t boundary, int mode,
                            uint16_t *dst, int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf,
                            int32_t *status, int32_t *aux, void *workspace, size_t workspace_bytes,
                            const int32_t *slot_found, int32_t slots, void *order_ws, size_t order_ws_bytes, bool head_is_zero,
                            hipStream_t st) {
    if (n <= 0) return 0;
    const PlanRec *order = nullptr;
    const size_t order_need = boundary_order_bytes(ix, n);
    if (order_ws && order_need != 0 && order_ws_bytes >= order_need) {
        PlanRec *ordered = reinterpret_cast<PlanRec *>(static_cast<uint8_t *>(order_ws) + kPlanHeadBytes);
        if (int e = launch_position_order(ix, from, (int32_t)n, slot_found, slots, order_ws, head_is_zero, ordered, st)) return e;
        order = ordered;
    }
    const BoundaryShape shape = boundary_shape();
    const int blk = shape.block;
    const int blocks_accel = grid_for(n * (shape.group ? shape.group : 1), blk, n_cu);  // the grid the scratch is sized for
    const size_t windows_bytes = boundary_bytes_for_grid(ix, blocks_accel, shape);
    uint16_t *scratch = (workspace && shape.accel && workspace_bytes >= windows_bytes + boundary_redo_bytes(n))
                            ? static_cast<uint16_t *>(workspace)
                            : nullptr;
    const int G = scratch ? shape.group : 0;
    const int pair_walks = g_boundary_first_fill != 0;
    const dim3 grid(scratch ? blocks_accel : grid_for(n, blk, n_cu));
    // the group kernel's redo list lives behind the windows; its count is cleared in front of every launch
    int32_t *redo = (scratch && G > 0) ? reinterpret_cast<int32_t *>(static_cast<uint8_t *>(workspace) + ((windows_bytes + 15) & ~(size_t)15)) : nullptr;
    // The NARROW first round (option "boundary_narrow", default 1; the default group of 4, sample rates the marked replay serves,
    // batches of "boundary_narrow_min" queries or more): the walks are what this costs (one sector per LF-step, at the chip's
    // random-sector rate) and the wide form fetches 8 sample intervals per query where a line needs 3.1 — so every query first
    // gets the two intervals on each side of `from` (G = 2, a lane's two walks interleaved: HALF the LF-steps), and only a query
    // whose line does not end inside them (about one in seven of configs[3]) takes the wide form, off a list, in a launch behind.
    int32_t *todo = nullptr;
    if (redo && G == 4 && pair_walks && g_boundary_narrow && ix.sample_rate <= 64 && n >= g_boundary_narrow_min &&
        workspace_bytes >= windows_bytes + 2 * boundary_redo_bytes(n) + 16) {
        todo = redo;
        redo = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(todo) + ((boundary_redo_bytes(n) + 15) & ~(size_t)15));
        if (hipError_t e = hipMemsetAsync(todo, 0, kRedoHead * sizeof(int32_t), st); e != hipSuccess) return (int)e;
    }
    if (redo)
        if (hipError_t e = hipMemsetAsync(redo, 0, kRedoHead * sizeof(int32_t), st); e != hipSuccess) return (int)e;
    // (kWinAsk: this kernel looks at ix.win itself.  An instantiation without the tree walk — kWinAlways, as k_locate_walk and
    // k_extract have — was measured SLOWER here: 94 instead of 112 VGPRs, five waves per SIMD instead of four, 20 bytes of scratch
    // in the walk's loop: 0.98 vs 0.78 ms on configs[3], round 5)
#define FMX_LAUNCH_NARROW_MODE(BLK, MODE)                                                                                \
    hipLaunchKernelGGL((k_extract_boundary_group<BLK, 2, MODE, kWinAsk, true>), narrow_grid, dim3(BLK), 0, st, ix, from, n, boundary, dst, \
                       dst_len, offset, out_len, lf, status, aux, scratch, slot_found, slots, pair_walks, order, todo,          \
                       (const int32_t *)nullptr)
    if (todo) {
        const dim3 narrow_grid(grid_for(n * 2, blk, n_cu));  // (never more lanes than the windows were sized for: the wide form's grid)
        if (blk == 1024) {
            if (mode == 0) FMX_LAUNCH_NARROW_MODE(1024, 0);
            else if (mode == 1) FMX_LAUNCH_NARROW_MODE(1024, 1);
            else FMX_LAUNCH_NARROW_MODE(1024, 2);
        } else {
            if (mode == 0) FMX_LAUNCH_NARROW_MODE(512, 0);
            else if (mode == 1) FMX_LAUNCH_NARROW_MODE(512, 1);
            else FMX_LAUNCH_NARROW_MODE(512, 2);
        }
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
    }
#undef FMX_LAUNCH_NARROW_MODE
#define FMX_LAUNCH_GROUP_MODE_WIN(BLK, GG, MODE, KWIN)                                                                   \
    hipLaunchKernelGGL((k_extract_boundary_group<BLK, GG, MODE, KWIN>), grid, dim3(BLK), 0, st, ix, from, n, boundary, dst, dst_len, \
                       offset, out_len, lf, status, aux, scratch, slot_found, slots, pair_walks, todo ? nullptr : order, redo,  \
                       (const int32_t *)todo)
// (over the directory's FLAT form extractUntilBoundaryLeft — mode 1, the default group of 4 — runs the instantiation that has no tree
// walk in its body: 0.464 -> 0.392 ms per 100,000 queries; modes 0 and 2 measured slower / equal that way, 0.720 -> 0.776 / 0.656 ->
// 0.660, and keep the one that looks at ix.win itself: tools/boundary_probe.py, round 6)
#define FMX_LAUNCH_GROUP_MODE(BLK, GG, MODE)                                                                             \
    do {                                                                                                                \
        if (GG == 4 && MODE == 0 && g_boundary_rounds && pair_walks && !todo)                                            \
            hipLaunchKernelGGL((k_extract_boundary_group<BLK, ((GG == 4 && MODE == 0) ? 4 : 1), 0, kWinAsk, false,       \
                                                         (GG == 4 && MODE == 0)>),                                      \
                               grid, dim3(BLK), 0, st, ix, from, n, boundary, dst, dst_len, offset, out_len, lf, status, aux, scratch, \
                               slot_found, slots, pair_walks, order, redo, (const int32_t *)nullptr);                   \
        else if (GG == 4 && MODE == 1 && ix.win_flat)                                                                    \
            FMX_LAUNCH_GROUP_MODE_WIN(BLK, ((GG == 4 && MODE == 1) ? 4 : 1), ((GG == 4 && MODE == 1) ? 1 : 0),          \
                                      ((GG == 4 && MODE == 1) ? kWinFlat : kWinAsk));                                  \
        else                                                                                                            \
            FMX_LAUNCH_GROUP_MODE_WIN(BLK, GG, MODE, kWinAsk);                                                          \
    } while (0)
#define FMX_LAUNCH_GROUP(GG)                                                                                            \
    do {                                                                                                                \
        if (blk == 1024) {                                                                                              \
            if (mode == 0) FMX_LAUNCH_GROUP_MODE(1024, GG, 0);                                                          \
            else if (mode == 1) FMX_LAUNCH_GROUP_MODE(1024, GG, 1);                                                     \
            else FMX_LAUNCH_GROUP_MODE(1024, GG, 2);                                                                    \
        } else {                                                                                                        \
            if (mode == 0) FMX_LAUNCH_GROUP_MODE(512, GG, 0);                                                           \
            else if (mode == 1) FMX_LAUNCH_GROUP_MODE(512, GG, 1);                                                      \
            else FMX_LAUNCH_GROUP_MODE(512, GG, 2);                                                                     \
        }                                                                                                               \
    } while (0)
    if (G == 1)
        FMX_LAUNCH_GROUP(1);
    else if (G == 2)
        FMX_LAUNCH_GROUP(2);
    else if (G == 4)
        FMX_LAUNCH_GROUP(4);
    else if (G == 8)
        FMX_LAUNCH_GROUP(8);
    else if (G == 16)
        FMX_LAUNCH_GROUP(16);
    else if (blk == 1024)
        hipLaunchKernelGGL(k_extract_boundary<1024>, grid, dim3(1024), 0, st, ix, from, n, boundary, mode, dst, dst_len, offset,
                           out_len, lf, status, aux, scratch, slot_found, slots, order, nullptr);
    else
        hipLaunchKernelGGL(k_extract_boundary<512>, grid, dim3(512), 0, st, ix, from, n, boundary, mode, dst, dst_len, offset,
                           out_len, lf, status, aux, scratch, slot_found, slots, order, nullptr);
#undef FMX_LAUNCH_GROUP
#undef FMX_LAUNCH_GROUP_MODE
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
    if (redo) {
        // what the group kernel left on its redo list (usually nothing: the launch then ends at its first instruction), literally;
        // a few workgroups grid-stride over the list
        const dim3 redo_grid(n < 64 * 512 ? (unsigned)((n + 511) / 512) : 64u);
        hipLaunchKernelGGL(k_extract_boundary<512>, redo_grid, dim3(512), 0, st, ix, from, n, boundary, mode, dst, dst_len, offset,
                           out_len, lf, status, aux, nullptr, slot_found, slots, nullptr, redo);
    }
    return (int)hipGetLastError();
}

}  // namespace FMX_KNS

#if defined(FMX_DIAG_TIMELINE) && !FMX_COMPACT
// (diagnostic builds only; not declared in include/fmx.h) copies {start, end, xcc << 32 | hw id} of the first `groups`
// workgroups of the last k_count launch
extern "C" __attribute__((visibility("default"))) int fmx_diag_timeline(unsigned long long *out, int groups) {
    if (groups > fmx::kDiagGroups) groups = fmx::kDiagGroups;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fmx::g_diag_timeline), (size_t)groups * 24, 0, hipMemcpyDeviceToHost);
}
#endif

#if defined(FMX_DIAG_LINES) && !FMX_COMPACT
// (diagnostic builds only) out == NULL: clear the bitmaps; else copy them out (8 x 2 MiB)
extern "C" __attribute__((visibility("default"))) int fmx_diag_lines(unsigned *out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (!out) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(fmx::g_diag_lines)) != hipSuccess) return -1;
        return (int)hipMemset(p, 0, sizeof(unsigned) * 8 * (1 << 19));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fmx::g_diag_lines), sizeof(unsigned) * 8 * (1 << 19), 0, hipMemcpyDeviceToHost);
}
#endif
