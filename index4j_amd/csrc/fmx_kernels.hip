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
#include <rocprim/block/block_radix_sort.hpp>

#include "fmx_device.hpp"

namespace fmx {

// Workgroup size is a template parameter (512 / 1024 threads).  Only the stand-alone RrrVector kernels stage the
// 32 KiB value-of-offset table in LDS.
// FMX_WAVES_PER_EU asks the register allocator for 8 waves per SIMD (<= 64 VGPRs, <= 80 SGPRs): the
// kernels are latency-bound chains of dependent loads, so resident waves are what hides latency.
#define FMX_KERNEL(BLOCK) __global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))
// The LF-walk kernels (locate / extract / extractUntilBoundary) carry more state per lane; FMX_WALK_WAVES is the
// occupancy their register budget is sized for (512 / FMX_WALK_WAVES VGPRs per lane).
#ifndef FMX_WALK_WAVES
#define FMX_WALK_WAVES 8
#endif
#define FMX_WALK_KERNEL(BLOCK) \
    __global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FMX_WALK_WAVES, 8)))
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
__host__ __device__ inline int plan_code_bits(int32_t sigma) { return sigma <= 256 ? 8 : 16; }

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

template <int kBlock>
FMX_KERNEL(kBlock) void k_count(DevIndex ix_global, const uint16_t *__restrict__ pat,
                                                  const int32_t *__restrict__ pat_off,
                                                  const uint32_t *__restrict__ perm, int32_t n,
                                                  int32_t *__restrict__ counts, int32_t *__restrict__ lf_steps,
                                                  int32_t *__restrict__ status_out, int32_t *__restrict__ range_out,
                                                  const uint64_t *__restrict__ codes) {
    const uint16_t *s_inv = nullptr;  // no RRR vector on this kernel's path (the wavelet tree's are expanded)
    FMX_WITH_SB_CACHE(ix_global, ix);
    const int role = threadIdx.x & 1;
    const int code_bits = plan_code_bits(ix.wt_sigma), n_codes = codes ? 64 / code_bits : 0;
    const uint32_t code_mask = (1u << code_bits) - 1u;
    const int32_t pairs_per_grid = (int32_t)gridDim.x * (kBlock / 2);  // 32-bit indices: n < 2^31, fewer live registers
    // (an XCD-aware block order — a contiguous eighth of the sorted batch per XCD — was measured slower:
    // profiles/r01_i_xcd_remap.txt)
    for (int32_t q = (int32_t)blockIdx.x * (kBlock / 2) + (int32_t)(threadIdx.x >> 1); q < n; q += pairs_per_grid) {
        const int32_t p = perm ? (int32_t)perm[q] : q;
        const int32_t m = pat_off[p + 1] - pat_off[p];
        int status = ST_OK;
        int32_t start = 0, end = 0;
        int32_t back = 0;  // characters consumed so far, counted from the pattern's end (FM:456: i = m - 1 - back)
        if (m <= 0) {
            status = ST_JAVA_AIOOBE;  // pattern[-1], FM:456-457
        } else {
            // the plan stage left the codes of the trailing characters (one 8-byte load per pattern instead of a
            // character load and a map lookup in front of every rank)
            const uint64_t cw = codes ? codes[p] : 0ull;
            int32_t c = codes ? (int32_t)((uint32_t)cw & code_mask) : fm_map(ix, pat[pat_off[p] + m - 1]);
            if (c != 0) {  // FM:458-460
                start = ix.C[c];
                end = ix.C[c + 1];
                while (start < end && back + 1 < m) {  // FM:464
                    ++back;
                    if (back < n_codes) {
                        c = (int32_t)((uint32_t)(cw >> (back * code_bits)) & code_mask);
                    } else {
                        c = fm_map(ix, pat[pat_off[p] + m - 1 - back]);
                    }
                    if (c == 0) {  // FM:466-468: ends the search before this character's two ranks
                        start = end = 0;
                        --back;
                        break;
                    }
                    const int32_t mine = wt_rank_folded(ix, s_inv, (uint32_t)(role ? end : start), c, status);  // C[c] + rank
                    const int32_t other = __shfl_xor(mine, 1);
                    start = role ? other : mine;  // FM:469
                    end = role ? mine : other;    // FM:470
                }
            }
        }
        const int32_t steps = 2 * back;  // LF-steps executed: two ranks per character after the first
        status |= __shfl_xor(status, 1);
        if (role == 0) {
            const int32_t d = end - start;
            counts[p] = d > 0 ? d : 0;  // FM:473
            if (lf_steps) lf_steps[p] = steps;
            if (status_out) status_out[p] = status;
            if (range_out) {
                range_out[2 * (int64_t)p] = start;
                range_out[2 * (int64_t)p + 1] = end;
            }
        }
    }
}

// FM:526-548: hit k of pattern p is SA row i = start + 1 + k; walk LF until a sampled row.
template <int kBlock>
FMX_WALK_KERNEL(kBlock) void k_locate_walk(DevIndex ix_global, const int32_t *__restrict__ range, int32_t n,
                                                        int32_t max_matches, int32_t *__restrict__ locs,
                                                        int32_t loc_cap, int32_t slots, int32_t *__restrict__ found,
                                                        int32_t *__restrict__ lf_steps,
                                                        int32_t *__restrict__ status_out,
                                                        const int32_t *__restrict__ taken) {
    const uint16_t *s_inv = nullptr;  // no RRR vector on this kernel's path (the sampled-row bitmap is expanded)
    FMX_WITH_SB_CACHE(ix_global, ix);
    const int64_t total = (int64_t)n * slots;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        const int32_t p = (int32_t)(t / slots);
        const int32_t k = (int32_t)(t - (int64_t)p * slots);
        const int32_t start = range[2 * p], end = range[2 * p + 1];
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
        if (k >= located) continue;
        int status = ST_OK;
        int32_t distance;
        locs[(int64_t)p * loc_cap + k] = fm_locate_hit(ix, s_inv, start, k, distance, status);
        if (lf_steps && distance) atomicAdd(&lf_steps[p], distance);
        if (status && status_out) atomicOr(&status_out[p], status);
    }
}

// FM:564-608.  Pipeline form (slot_found != nullptr): query q is hit (q % slots) of pattern (q / slots) and
// runs only if that hit exists; with stops == nullptr the stop position is min(inputLength, start + fixed_len)
// (the reference's locateAndExtractBenchmark, FmIndexThroughputBenchmark.java:231-249).
template <int kBlock>
FMX_EXTRACT_KERNEL(kBlock) void k_extract(DevIndex ix_global, const int32_t *__restrict__ starts, const int32_t *__restrict__ stops,
                                  int64_t n, uint16_t *__restrict__ dst, int32_t dst_len, int32_t offset,
                                  int32_t *__restrict__ out_len, int32_t *__restrict__ lf_steps,
                                  int32_t *__restrict__ status_out, const int32_t *__restrict__ slot_found,
                                  int32_t slots, int32_t fixed_len) {
    const uint16_t *s_inv = nullptr;  // no RRR vector on this kernel's path (the wavelet tree's are expanded)
    FMX_WITH_SB_CACHE(ix_global, ix);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride) {
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
        const int32_t ret = fm_extract(ix, s_inv, start, stop, dst + q * (int64_t)dst_len, dst_len, offset, steps, status);
        out_len[q] = status ? 0 : ret;
        if (lf_steps) lf_steps[q] = steps;
        if (status_out) status_out[q] = status;
    }
}

// FM:640-759 (mode 0), FM:772-831 (mode 1), FM:844-922 (mode 2), one lane per query.  `scratch` (nullable) holds
// sample_rate codes per lane of the grid (element j of lane t at scratch[j * lanes + t]) for the
// interval-buffered right walk (fm_boundary_right_blocks); without it the literal form runs.
template <int kBlock>
FMX_BOUNDARY_KERNEL(kBlock) void k_extract_boundary(DevIndex ix_global, const int32_t *__restrict__ froms, int64_t n, uint16_t boundary,
                                           int mode, uint16_t *__restrict__ dst, int32_t dst_len, int32_t offset,
                                           int32_t *__restrict__ out_len, int32_t *__restrict__ lf_steps,
                                           int32_t *__restrict__ status_out, int32_t *__restrict__ aux_out,
                                           uint16_t *__restrict__ scratch, const int32_t *__restrict__ slot_found,
                                           int32_t slots) {
    const uint16_t *s_inv = nullptr;  // no RRR vector on this kernel's path (the wavelet tree's are expanded)
    FMX_WITH_SB_CACHE(ix_global, ix);
    const int64_t lanes = (int64_t)gridDim.x * kBlock;
    const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int32_t mapped_boundary = fm_map(ix, boundary);  // FM:658
    for (int64_t q = lane; q < n; q += lanes) {
        if (slot_found && (int32_t)(q % slots) >= slot_found[q / slots]) continue;
        int status = ST_OK;
        int32_t steps, aux;
        const int32_t ret = fm_extract_boundary(ix, s_inv, mode, froms[q], mapped_boundary, dst + q * (int64_t)dst_len,
                                                dst_len, offset, steps, status, aux, scratch ? scratch + lane : nullptr,
                                                lanes);
        out_len[q] = status ? 0 : ret;
        if (lf_steps) lf_steps[q] = steps;
        if (status_out) status_out[q] = status;
        if (aux_out) aux_out[q] = aux;
    }
}

// Group-cooperative extractUntilBoundary: G lanes per query (fm_extract_boundary_group); the window of a group
// is G consecutive lane columns of `scratch` (left window in the first half, right window in the second).
template <int kBlock, int G>
FMX_BOUNDARY_KERNEL(kBlock) void k_extract_boundary_group(DevIndex ix_global, const int32_t *__restrict__ froms, int64_t n,
                                                 uint16_t boundary, int mode, uint16_t *__restrict__ dst,
                                                 int32_t dst_len, int32_t offset, int32_t *__restrict__ out_len,
                                                 int32_t *__restrict__ lf_steps, int32_t *__restrict__ status_out,
                                                 int32_t *__restrict__ aux_out, uint16_t *__restrict__ scratch,
                                                 const int32_t *__restrict__ slot_found, int32_t slots) {
    const uint16_t *s_inv = nullptr;  // no RRR vector on this kernel's path (the wavelet tree's are expanded)
    FMX_WITH_SB_CACHE(ix_global, ix);
    const int64_t lanes = (int64_t)gridDim.x * kBlock;
    const int64_t lane = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int g = threadIdx.x % G;
    const int64_t groups = lanes / G;
    const int32_t mapped_boundary = fm_map(ix, boundary);  // FM:658
    for (int64_t q = lane / G; q < n; q += groups) {
        if (slot_found && (int32_t)(q % slots) >= slot_found[q / slots]) continue;  // group-uniform
        int status = ST_OK;
        int32_t steps, aux;
        bool clean;
        uint16_t *dest = dst + q * (int64_t)dst_len;
        int32_t ret = fm_extract_boundary_group<G>(ix, s_inv, mode, froms[q], mapped_boundary, dest, dst_len, offset, steps,
                                                   status, aux, scratch + (lane - g), lanes, 1, lanes * (int64_t)ix.sample_rate, g, clean);
        if (!clean && g == 0) {  // a walk met a quirk path of the wavelet tree: literal form (rare)
            int32_t steps2;
            status = ST_OK;
            ret = fm_extract_boundary(ix, s_inv, mode, froms[q], mapped_boundary, dest, dst_len, offset, steps2, status, aux);
            steps += steps2;
        }
        if (g == 0) {
            out_len[q] = status ? 0 : ret;
            if (lf_steps) lf_steps[q] = steps;
            if (status_out) status_out[q] = status;
            if (aux_out) aux_out[q] = aux;
        }
    }
}

// WaveletFixedBlockBoosting.rank(position, symbol) WFBB:1010-1285, one lane per query
template <int kBlock>
FMX_KERNEL(kBlock) void k_wt_rank(DevIndex ix, const int64_t *__restrict__ positions, const int32_t *__restrict__ symbols,
                                  int32_t n, int64_t *__restrict__ out, int32_t *__restrict__ status_out) {
    const uint16_t *s_inv = nullptr;  // no RRR vector on this kernel's path (the wavelet tree's are expanded)
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
    const uint16_t *s_inv = nullptr;  // no RRR vector on this kernel's path (the wavelet tree's are expanded)
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
    const RrrView v = {ix.sampled.off_rec, ix.sampled.off_bits, ix.sampled.length, ix.sampled.total_ones};
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
    const RrrView v = {ix.sampled.off_rec, ix.sampled.off_bits, ix.sampled.length, ix.sampled.total_ones};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += stride) {
        int status = ST_OK;
        const bool bit = rrr_access(ix.base, v, s_inv, positions[q], status);
        out[q] = status ? 0 : (bit ? 1 : 0);
        if (status_out) status_out[q] = status;
    }
}

// ---- processing order of a batch -------------------------------------------------------------------
// Patterns are processed in (approximate) order of their trailing characters, the LAST character most
// significant (it is consumed first, FM:456-457): lanes of a wave then start their backward search in the
// same SA intervals.  Any grouping works — results are written at the original index — so instead of a
// general device sort (rocPRIM falls back to a 21-launch merge sort at 1 M keys) the order is built by
//   1. a coarse bucket pass on the top <=14 key bits: per-workgroup LDS histograms, ONE global atomic per
//      (workgroup, non-empty bin) — no hot-bin serialisation — then a single-workgroup scan and a scatter
//      whose slots come from LDS cursors;
//   2. a tile-local LDS radix sort (4,096 patterns per workgroup) on the full key, which restores the fine
//      order where it matters: inside a wave / a CU.
constexpr int kTileThreads = 512;
constexpr int kTileItems = 8;                       // patterns per thread
constexpr int kTile = kTileThreads * kTileItems;    // 4,096 patterns per workgroup
constexpr int kCoarseBitsMax = 14;                  // 16,384 LDS bins (64 KiB)

struct SortShape {
    int bits;         // bits per alphabet code
    int chars;        // trailing characters in the full key
    int total_bits;   // chars * bits (<= 32)
    int coarse_bits;  // top bits used by the bucket pass
};

// the plan's code word of a pattern: codes of its trailing characters, the LAST character in the low bits
__device__ __forceinline__ uint64_t pattern_code_word(const DevIndex &ix, const uint16_t *pat, int32_t beg, int32_t m,
                                                      int code_bits) {
    const int n_codes = 64 / code_bits;
    uint64_t w = 0;
    for (int j = 0; j < n_codes && j < m; ++j) w |= (uint64_t)(uint32_t)fm_map(ix, pat[beg + m - 1 - j]) << (j * code_bits);
    return w;
}
// sort key = the first `chars` codes of the word, the last character most significant
__device__ __forceinline__ uint32_t suffix_key(uint64_t word, int code_bits, int chars, int bits) {
    const uint32_t mask = (1u << code_bits) - 1u;
    uint32_t key = 0;
    for (int j = 0; j < chars; ++j) key = (key << bits) | ((uint32_t)(word >> (j * code_bits)) & mask);
    return key;
}

// pass 1a: coarse keys + global histogram (LDS-privatised)
__global__ __launch_bounds__(kTileThreads) void k_order_hist(DevIndex ix, const uint16_t *__restrict__ pat,
                                                             const int32_t *__restrict__ pat_off, int32_t n,
                                                             SortShape sh, uint32_t *__restrict__ coarse,
                                                             uint32_t *__restrict__ ghist,
                                                             uint64_t *__restrict__ codes) {
    extern __shared__ uint32_t s_hist[];
    const int code_bits = plan_code_bits(ix.wt_sigma);
    const int bins = 1 << sh.coarse_bits;
    for (int i = threadIdx.x; i < bins; i += kTileThreads) s_hist[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
    for (int k = 0; k < kTileItems; ++k) {
        const int64_t p = base + (int64_t)k * kTileThreads + threadIdx.x;
        if (p < n) {
            const int32_t beg = pat_off[p];
            const uint64_t word = pattern_code_word(ix, pat, beg, pat_off[p + 1] - beg, code_bits);
            codes[p] = word;  // kept for the tile sort and for k_count
            const uint32_t key = suffix_key(word, code_bits, sh.chars, sh.bits);
            uint32_t c = key >> (sh.total_bits - sh.coarse_bits);
            if (c >= (uint32_t)bins) c = (uint32_t)bins - 1u;  // cannot happen for a validated index (codes < 2^bits)
            coarse[p] = c;
            atomicAdd(&s_hist[c], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += kTileThreads) {
        const uint32_t v = s_hist[i];
        if (v) atomicAdd(&ghist[i], v);
    }
}

// pass 1b: exclusive scan of <= 16,384 bins by one workgroup (in place): coalesced load into LDS, per-thread
// chunks, Hillis-Steele over the 1,024 partial sums, coalesced store
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void k_order_scan(uint32_t *__restrict__ ghist, int bins) {
    __shared__ uint32_t s_val[(1 << kCoarseBitsMax) + 64];
    __shared__ uint32_t s_part[kScanThreads];
    for (int i = threadIdx.x; i < bins; i += kScanThreads) s_val[i + (i >> 8)] = ghist[i];  // +1 pad per 256: no 16-way conflicts
    __syncthreads();
    const int per = (bins + kScanThreads - 1) / kScanThreads;
    const int lo = threadIdx.x * per;
    uint32_t sum = 0;
    for (int i = lo; i < lo + per && i < bins; ++i) sum += s_val[i + (i >> 8)];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < kScanThreads; d <<= 1) {
        const uint32_t v = (threadIdx.x >= (unsigned)d) ? s_part[threadIdx.x - d] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
    for (int i = lo; i < lo + per && i < bins; ++i) {
        const uint32_t v = s_val[i + (i >> 8)];
        s_val[i + (i >> 8)] = run;
        run += v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += kScanThreads) ghist[i] = s_val[i + (i >> 8)];
}

// pass 1c: scatter into bucket order; each workgroup reserves its share of a bin with one atomic
__global__ __launch_bounds__(kTileThreads) void k_order_scatter(const uint32_t *__restrict__ coarse, int32_t n,
                                                                SortShape sh, uint32_t *__restrict__ cursor,
                                                                uint32_t *__restrict__ perm) {
    extern __shared__ uint32_t s_hist[];
    const int bins = 1 << sh.coarse_bits;
    for (int i = threadIdx.x; i < bins; i += kTileThreads) s_hist[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
    uint32_t mine[kTileItems];
    for (int k = 0; k < kTileItems; ++k) {
        const int64_t p = base + (int64_t)k * kTileThreads + threadIdx.x;
        mine[k] = (p < n) ? coarse[p] : 0xffffffffu;
        if (p < n) atomicAdd(&s_hist[mine[k]], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += kTileThreads) {
        const uint32_t v = s_hist[i];
        if (v) s_hist[i] = atomicAdd(&cursor[i], v);  // first slot of this workgroup's share
    }
    __syncthreads();
    for (int k = 0; k < kTileItems; ++k) {
        const int64_t p = base + (int64_t)k * kTileThreads + threadIdx.x;
        if (p < n) perm[atomicAdd(&s_hist[mine[k]], 1u)] = (uint32_t)p;
    }
}

// pass 2: tile-local radix sort of the bucket order on the full key
__global__ __launch_bounds__(kTileThreads) void k_order_tile_sort(DevIndex ix, const uint64_t *__restrict__ codes,
                                                                  int32_t n, SortShape sh,
                                                                  const uint32_t *__restrict__ perm_in,
                                                                  uint32_t *__restrict__ perm_out) {
    const int code_bits = plan_code_bits(ix.wt_sigma);
    using Sort = rocprim::block_radix_sort<uint32_t, kTileThreads, kTileItems, uint32_t>;
    __shared__ typename Sort::storage_type storage;
    __shared__ uint32_t s_first, s_last;
    const int64_t base = (int64_t)blockIdx.x * kTile;
    const int low_bits = sh.total_bits - sh.coarse_bits;
    uint32_t keys[kTileItems], vals[kTileItems];
    for (int k = 0; k < kTileItems; ++k) {
        const int64_t i = base + (int64_t)threadIdx.x * kTileItems + k;
        if (i < n) {
            const uint32_t p = perm_in[i];
            keys[k] = suffix_key(codes[p], code_bits, sh.chars, sh.bits);
            vals[k] = p;
            // the input is in bucket order: the tile's first item has its smallest coarse key, its last item the largest
            if (i == base) s_first = keys[k] >> low_bits;
            if (i == n - 1 || i == base + kTile - 1) s_last = keys[k] >> low_bits;
        } else {
            keys[k] = 0;
            vals[k] = 0xffffffffu;
        }
    }
    __syncthreads();
    // keys relative to the tile's first bucket: only the bits that can differ inside the tile are sorted
    // (a tile inside one hot bucket sorts 16 bits instead of 28)
    const uint32_t first = s_first, span = s_last - s_first;
    int end_bit = low_bits + 1;
    while ((span + 1) >> (end_bit - low_bits)) ++end_bit;  // room for span + 1: the padding sorts last
    for (int k = 0; k < kTileItems; ++k) {
        const int64_t i = base + (int64_t)threadIdx.x * kTileItems + k;
        keys[k] = i < n ? keys[k] - (first << low_bits) : (span + 1) << low_bits;
    }
    Sort().sort(keys, vals, storage, 0, end_bit);
    for (int k = 0; k < kTileItems; ++k) {
        const int64_t i = base + (int64_t)threadIdx.x * kTileItems + k;
        if (i < n) perm_out[i] = vals[k];
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

// ---- launchers (called from fmx_api.cpp) -----------------------------------------------------

// tunables (fmx_set_option): workgroup size and how many workgroups per CU the grid is capped at.  Atomics: a
// launch on one host thread may read them while another thread sets one (results are identical for every
// setting, so a launch that sees a mix of old and new values is still correct).
static std::atomic<int> g_block{512};
static std::atomic<int> g_groups_per_cu{16};
static std::atomic<int> g_boundary_accel{1};  // 0 = literal right walk of extractUntilBoundary (A/B and fallback)
static std::atomic<int> g_boundary_group{4};  // lanes per query of extractUntilBoundary (0 = one lane per query)
static std::atomic<int> g_lds_pad_kb{0};   // experiment knob: extra dynamic LDS per workgroup (lowers occupancy)
static std::atomic<int> g_sort_min{16384};  // batches at least this large are processed in suffix-sorted order (0 = never)
// bins of the bucket pass = 2^coarse_bits (<= 14: they live in LDS).  Measured on configs[1] (tools/tune_coarse.py):
// 14 bits: plan 0.091 ms, step 0.304 ms; 12 bits: 0.075 / 0.286 ms; 10 bits: 0.071 / 0.286 ms; 8 bits: 0.069 / 0.294 ms
static std::atomic<int> g_coarse_bits{12};
static std::atomic<int> g_sort_bits{28};    // full key width: floor(sort_bits / bits-per-code) trailing characters

int set_option(const char *name, int value) {
    if (!strcmp(name, "block")) {
        if (value != 512 && value != 1024) return -1;
        g_block = value;
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
    if (!strcmp(name, "coarse_bits")) {
        if (value < 4 || value > kCoarseBitsMax) return -1;
        g_coarse_bits = value;
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

static SortShape sort_shape(const DevIndex &ix) {
    SortShape sh;
    sh.bits = 1;
    while ((1 << sh.bits) < ix.wt_sigma && sh.bits < 15) ++sh.bits;
    const int sort_bits = g_sort_bits, coarse_bits = g_coarse_bits;
    sh.chars = sort_bits / sh.bits;
    if (sh.chars < 1) sh.chars = 1;
    if (sh.chars > 64 / plan_code_bits(ix.wt_sigma)) sh.chars = 64 / plan_code_bits(ix.wt_sigma);
    sh.total_bits = sh.chars * sh.bits;
    sh.coarse_bits = sh.total_bits < coarse_bits ? sh.total_bits : coarse_bits;
    return sh;
}

// workspace layout: coarse[n] perm1[n] perm2[n] ghist[bins] | codes[n] (8 bytes each, 16-byte aligned)
static size_t plan_codes_offset(int32_t n) {
    return (((size_t)n * 12 + ((size_t)4 << kCoarseBitsMax) + 512) + 15) & ~(size_t)15;
}
// bytes of scratch needed to order a batch of n patterns (0 = the batch is not sorted)
size_t count_workspace_bytes(const DevIndex &ix, int32_t n) {
    const int sort_min = g_sort_min;
    if (sort_min <= 0 || n < sort_min) return 0;
    return plan_codes_offset(n) + (size_t)n * sizeof(uint64_t);
}

// perm_out[q] = index of the q-th pattern in processing order.  workspace: count_workspace_bytes(ix, n).
// Returns a hipError_t value.
int launch_count_plan(const DevIndex &ix, const uint16_t *pat, const int32_t *off, int32_t n, void *workspace,
                      size_t workspace_bytes, const uint32_t **perm_out, const void **codes_out, hipStream_t st) {
    *perm_out = nullptr;
    *codes_out = nullptr;
    const size_t need = count_workspace_bytes(ix, n);
    if (n <= 0 || !workspace || need == 0 || workspace_bytes < need) return 0;
    const SortShape sh = sort_shape(ix);
    const int bins = 1 << sh.coarse_bits;
    uint32_t *coarse = static_cast<uint32_t *>(workspace);
    uint32_t *perm1 = coarse + n;
    uint32_t *perm2 = perm1 + n;
    uint32_t *ghist = perm2 + n;
    const int tiles = (n + kTile - 1) / kTile;
    hipError_t e = hipMemsetAsync(ghist, 0, (size_t)bins * 4, st);
    if (e != hipSuccess) return (int)e;
    uint64_t *codes = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(workspace) + plan_codes_offset(n));
    hipLaunchKernelGGL(k_order_hist, dim3(tiles), dim3(kTileThreads), (size_t)bins * 4, st, ix, pat, off, n, sh, coarse, ghist, codes);
    hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(kScanThreads), 0, st, ghist, bins);
    hipLaunchKernelGGL(k_order_scatter, dim3(tiles), dim3(kTileThreads), (size_t)bins * 4, st, coarse, n, sh, ghist, perm1);
    if (sh.total_bits > sh.coarse_bits) {
        hipLaunchKernelGGL(k_order_tile_sort, dim3(tiles), dim3(kTileThreads), 0, st, ix, codes, n, sh, perm1, perm2);
        *perm_out = perm2;
    } else {
        *perm_out = perm1;
    }
    *codes_out = codes;
    return (int)hipGetLastError();
}

// `codes` = the plan's per-pattern code words for THIS index's alphabet (nullptr: characters are mapped in the kernel)
int launch_count(const DevIndex &ix, int n_cu, const uint16_t *pat, const int32_t *off, const uint32_t *perm,
                 const void *codes, int32_t n, int32_t *counts, int32_t *lf, int32_t *status, int32_t *range,
                 hipStream_t st) {
    if (n <= 0) return 0;
    FMX_DISPATCH(k_count, 2 * (int64_t)n, ix, pat, off, perm, n, counts, lf, status, range,
                 perm ? static_cast<const uint64_t *>(codes) : nullptr);
    return (int)hipGetLastError();
}

int launch_locate_walk(const DevIndex &ix, int n_cu, const int32_t *range, int32_t n, int32_t max_matches,
                       int32_t *locs, int32_t loc_cap, int32_t *found, int32_t *lf, int32_t *status,
                       const int32_t *taken, hipStream_t st) {
    if (n <= 0) return 0;
    int32_t slots = (max_matches > 0 && max_matches < loc_cap) ? max_matches : loc_cap;
    if (slots < 1) slots = 1;
    FMX_DISPATCH(k_locate_walk, (int64_t)n * slots, ix, range, n, max_matches, locs, loc_cap, slots, found, lf, status, taken);
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

int launch_extract(const DevIndex &ix, int n_cu, const int32_t *start, const int32_t *stop, int64_t n, uint16_t *dst,
                   int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf, int32_t *status,
                   const int32_t *slot_found, int32_t slots, int32_t fixed_len, hipStream_t st) {
    if (n <= 0) return 0;
    FMX_DISPATCH(k_extract, n, ix, start, stop, n, dst, dst_len, offset, out_len, lf, status, slot_found, slots, fixed_len);
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
static size_t boundary_bytes_for(const DevIndex &ix, int64_t n, int n_cu, const BoundaryShape &b) {
    if (!b.accel || n <= 0) return 0;
    return boundary_bytes_for_grid(ix, grid_for(n * (b.group ? b.group : 1), b.block, n_cu), b);
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

int launch_extract_boundary(const DevIndex &ix, int n_cu, const int32_t *from, int64_t n, uint16_t boundary, int mode,
                            uint16_t *dst, int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf,
                            int32_t *status, int32_t *aux, void *workspace, size_t workspace_bytes,
                            const int32_t *slot_found, int32_t slots, hipStream_t st) {
    if (n <= 0) return 0;
    const BoundaryShape shape = boundary_shape();
    const int blk = shape.block;
    const int blocks_accel = grid_for(n * (shape.group ? shape.group : 1), blk, n_cu);  // the grid the scratch is sized for
    uint16_t *scratch = (workspace && shape.accel && workspace_bytes >= boundary_bytes_for_grid(ix, blocks_accel, shape))
                            ? static_cast<uint16_t *>(workspace)
                            : nullptr;
    const int G = scratch ? shape.group : 0;
    const dim3 grid(scratch ? blocks_accel : grid_for(n, blk, n_cu));
#define FMX_LAUNCH_GROUP(GG)                                                                                            \
    do {                                                                                                                \
        if (blk == 1024)                                                                                                \
            hipLaunchKernelGGL((k_extract_boundary_group<1024, GG>), grid, dim3(1024), 0, st, ix, from, n, boundary, mode, dst, \
                               dst_len, offset, out_len, lf, status, aux, scratch, slot_found, slots);                  \
        else                                                                                                            \
            hipLaunchKernelGGL((k_extract_boundary_group<512, GG>), grid, dim3(512), 0, st, ix, from, n, boundary, mode, dst,  \
                               dst_len, offset, out_len, lf, status, aux, scratch, slot_found, slots);                  \
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
                           out_len, lf, status, aux, scratch, slot_found, slots);
    else
        hipLaunchKernelGGL(k_extract_boundary<512>, grid, dim3(512), 0, st, ix, from, n, boundary, mode, dst, dst_len, offset,
                           out_len, lf, status, aux, scratch, slot_found, slots);
#undef FMX_LAUNCH_GROUP
    return (int)hipGetLastError();
}

}  // namespace fmx
