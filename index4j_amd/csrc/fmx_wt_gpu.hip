// Index construction, device stage 2: the fixed-block-boosting wavelet tree (WFBB:130-154, 362-535, 570-991) and
// the RRR vector of every superblock (RRR:225-286) encoded in HBM from the BWT the suffix-array stage left there
// (fmx_sa_gpu.hip).  The structures are the ones the host encoder (fmx_build.cpp) and the reference build — the
// serialized index is byte-identical (tests/test_gpu_parity.py, tools/fuzz_gpu.py).
//
// One WAVE per block, for every phase that looks at a block:
//   histogram of the block in LDS  ->  Huffman code lengths with the reference's tie order (WFBB:334-360, comparator
//   WFBB:1684-1707: (frequency, first symbol of the merged list)) as a wave-cooperative merge loop — every lane owns
//   the queue entries of its symbols, the two minima of a round are two wave-wide min reductions over 32-bit keys
//   (frequency << 15 | first symbol), the merge bumps the depth of every leaf of the two groups  ->  canonical codes
//   in (length, symbol) order (WFBB:549-554)  ->  header bytes (WFBB:713-809)  ->  node bit vectors (WFBB:606-708):
//   each lane takes one symbol of the text, and for every level the lanes that sit in the same tree node rank
//   themselves with one ballot (position in the node = symbols routed through it so far + rank in the wave).
// Phases (kernels): per-superblock symbol counts; the size estimate of WFBB:853-987 for block sizes 2^9..2^16 (the
// double-precision chain runs on the host from the per-level sums); per-block plan (sizes, one-counts, per-symbol
// frequencies); scans over the blocks of a superblock (header offsets, bit-vector offsets, ranks at block starts);
// encode; RRR classes and group sums; RRR offsets and samples.
// Alphabet: every per-block phase works on the SUPERBLOCK-LOCAL codes of WFBB:838-846 (globalMapping: the symbols present in
// the superblock, numbered in ascending global order) — the BWT is rewritten into them once (k_wt_localize).  The numbering
// is order-preserving, so the Huffman tie order (first symbol of a merged list) and the canonical (length, symbol) order are
// the global alphabet's; only the leaves' u16 symbol is translated back (inverse table per superblock).  LDS and the dense
// per-block tables are then sized by the largest superblock alphabet, not by the text's: the reference's own data-set shape
// (1,099 symbols) encodes in HBM; only a superblock holding more than kWtMaxSigma distinct symbols goes to the host encoder.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "fmx_build_stage.hpp"
#include "fmx_model.hpp"

namespace fmx {
namespace {

constexpr int kSbLog = 20;                 // WFBB:93
constexpr int64_t kSbs = 1ll << kSbLog;
constexpr int kMinLog = 9, kMaxLog = 16;   // WFBB:856, 867-869
constexpr int kLevels = kMaxLog - kMinLog + 1;
constexpr uint32_t kInfKey = 0xffffffffu;

#define WT_TRY(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            return -6;                                                                       \
        }                                                                                    \
    } while (0)

struct DevMem {
    std::vector<void *> ptrs;
    ~DevMem() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T>
    hipError_t alloc(T **out, size_t count, bool zero = false) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) {
            ptrs.push_back(p);
            if (zero) e = hipMemset(p, 0, (count ? count : 1) * sizeof(T));
        }
        *out = static_cast<T *>(p);
        return e;
    }
};

// ---- wave helpers (workgroups are ONE wave of 64 lanes; LDS traffic of a wave is in program order) -------------
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = __shfl_xor(v, m);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = __shfl_xor(v, m);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ uint64_t lanes_below() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// per-wave scratch carved out of dynamic LDS; sigma_pad = alphabet size rounded up to 64
struct WaveLds {
    uint32_t *hist;    // [sigma_pad] occurrences of every symbol in the block
    uint32_t *hkey;    // [sigma_pad] queue keys of the compacted symbol list (slot i belongs to lane i % 64)
    uint32_t *code;    // [sigma_pad] canonical code, by symbol
    uint32_t *nsize;   // [sigma_pad] internal nodes in BFS order: bits, first bit, bits written, ones
    uint32_t *noff, *nfill, *nones;
    uint16_t *hsym;    // [sigma_pad] compacted list: symbol of slot i (ascending)
    uint16_t *hgrp;    // [sigma_pad] group (= first symbol of the merged list) the leaf of slot i belongs to
    uint16_t *idx;     // [sigma_pad] index of the symbol in (code length, symbol) order, by symbol
    uint16_t *sorted;  // [sigma_pad] symbol at that index
    uint8_t *hdep;     // [sigma_pad] depth of the leaf of slot i
    uint8_t *len;      // [sigma_pad] code length, by symbol
    uint32_t *cnt;     // [40] leaves per code length
    uint32_t *first;   // [40] first canonical code of a length
    uint32_t *lenbase; // [40] leaves with a shorter code
    uint32_t *running; // [40]
    uint32_t *lvl_base;   // [40] BFS index of the first internal node of a level
    uint32_t *lvl_first;  // [40] prefix value of the first internal node of a level
    uint32_t *lf;         // [40] symbols (with multiplicity) per code length
};
__host__ __device__ inline size_t wave_lds_bytes(int sigma_pad) {
    return (size_t)sigma_pad * (7 * 4 + 4 * 2 + 2 * 1) + 7 * 40 * 4 + 64;
}
__device__ __forceinline__ WaveLds carve(uint8_t *p, int sigma_pad) {
    WaveLds w;
    uint32_t *u = reinterpret_cast<uint32_t *>(p);
    w.hist = u; u += sigma_pad;
    w.hkey = u; u += sigma_pad;
    w.code = u; u += sigma_pad;
    w.nsize = u; u += sigma_pad;
    w.noff = u; u += sigma_pad;
    w.nfill = u; u += sigma_pad;
    w.nones = u; u += sigma_pad;
    w.cnt = u; u += 40;
    w.first = u; u += 40;
    w.lenbase = u; u += 40;
    w.running = u; u += 40;
    w.lvl_base = u; u += 40;
    w.lvl_first = u; u += 40;
    w.lf = u; u += 40;
    uint16_t *h = reinterpret_cast<uint16_t *>(u);
    w.hsym = h; h += sigma_pad;
    w.hgrp = h; h += sigma_pad;
    w.idx = h; h += sigma_pad;
    w.sorted = h; h += sigma_pad;
    uint8_t *b = reinterpret_cast<uint8_t *>(h);
    w.hdep = b; b += sigma_pad;
    w.len = b;
    return w;
}

// occurrences of every symbol in text[0, count)
__device__ __forceinline__ void block_histogram(const WaveLds &w, int sigma_pad, const int16_t *__restrict__ text,
                                                uint32_t count) {
    const int lane = threadIdx.x;
    for (int i = lane; i < sigma_pad; i += 64) w.hist[i] = 0;
    __syncthreads();
    for (uint32_t i = lane; i < count; i += 64) atomicAdd(&w.hist[(uint16_t)text[i]], 1u);
    __syncthreads();
}

// Huffman code lengths of the block's symbols (WFBB:334-360).  Returns the number of distinct symbols; th = longest
// code (0 for a single symbol), unc = sum of frequency * length.  Leaves: w.len[symbol]; the compacted list
// (hsym / hdep, ascending symbols) stays valid for the caller.
__device__ __forceinline__ int wave_huffman(const WaveLds &w, int sigma, int &th, uint64_t &unc) {
    const int lane = threadIdx.x;
    int s = 0;
    for (int base = 0; base < sigma; base += 64) {
        const int sym = base + lane;
        const uint32_t f = sym < sigma ? w.hist[sym] : 0u;
        const uint64_t m = __ballot(f > 0);
        if (f > 0) {
            const int pos = s + __popcll(m & lanes_below());
            w.hsym[pos] = (uint16_t)sym;
            w.hkey[pos] = (f << 15) | (uint32_t)sym;
            w.hgrp[pos] = (uint16_t)sym;
            w.hdep[pos] = 0;
        }
        s += __popcll(m);
    }
    __syncthreads();
    // every lane owns slots lane, lane + 64, ...: from here on it only touches its own slots
    for (int round = 1; round < s; ++round) {
        uint32_t local = kInfKey;
        for (int i = lane; i < s; i += 64) local = min(local, w.hkey[i]);
        const uint32_t m1 = wave_min_u32(local);  // the queue's head (WFBB:340): smallest (frequency, first symbol)
        int slot1 = -1;
        local = kInfKey;
        for (int i = lane; i < s; i += 64) {
            const uint32_t k = w.hkey[i];
            if (k == m1) {
                slot1 = i;
                w.hkey[i] = kInfKey;
            } else {
                local = min(local, k);
            }
        }
        const uint32_t m2 = wave_min_u32(local);  // the second poll (WFBB:341)
        const uint32_t g1 = m1 & 0x7fffu, g2 = m2 & 0x7fffu;
        for (int i = lane; i < s; i += 64) {
            if (w.hkey[i] == m2) w.hkey[i] = kInfKey;
            const uint32_t g = w.hgrp[i];
            if (g == g1 || g == g2) {  // every leaf below the new node sinks one level (WFBB:343-352)
                w.hdep[i] = (uint8_t)(w.hdep[i] + 1);
                w.hgrp[i] = (uint16_t)g1;
            }
        }
        if (slot1 >= 0) w.hkey[slot1] = (((m1 >> 15) + (m2 >> 15)) << 15) | g1;  // merged list starts with x's first symbol
    }
    uint32_t deepest = 0;
    uint64_t bits = 0;
    for (int i = lane; i < s; i += 64) {
        const uint32_t d = w.hdep[i];
        const uint32_t sym = w.hsym[i];
        w.len[sym] = (uint8_t)d;
        deepest = max(deepest, d);
        bits += (uint64_t)w.hist[sym] * d;
    }
    th = (int)wave_max_u32(deepest);
    unc = wave_sum_u64(bits);
    __syncthreads();
    return s;
}

// canonical codes in (length, symbol) order (WFBB:549-554) and the level geometry of the tree.  th <= 31.
__device__ __forceinline__ void wave_canonical(const WaveLds &w, int s, int th) {
    const int lane = threadIdx.x;
    if (lane < 40) {
        w.cnt[lane] = 0;
        w.running[lane] = 0;
        w.lf[lane] = 0;
    }
    __syncthreads();
    for (int i = lane; i < s; i += 64) {
        atomicAdd(&w.cnt[w.hdep[i]], 1u);
        atomicAdd(&w.lf[w.hdep[i]], w.hist[w.hsym[i]]);
    }
    __syncthreads();
    if (lane == 0) {
        uint32_t c = 0, before = 0, base = 0;
        w.first[0] = 0;
        w.lenbase[0] = 0;
        for (int L = 0; L <= th; ++L) {
            // level L: leaves take the prefixes [first, first + cnt), internal nodes the rest up to 2^L - 1
            w.first[L] = c;
            w.lenbase[L] = before;
            w.lvl_first[L] = c + w.cnt[L];
            w.lvl_base[L] = base;
            base += (1u << L) - (c + w.cnt[L]);
            before += w.cnt[L];
            c = (c + w.cnt[L]) << 1;
        }
    }
    __syncthreads();
    // rank inside a length in symbol order: the compacted list is ascending, a chunk of 64 slots ranks itself
    for (int base = 0; base < s; base += 64) {
        const int i = base + lane;
        const bool act = i < s;
        const uint32_t L = act ? w.hdep[i] : 0u;
        uint64_t todo = __ballot(act);
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const uint32_t k = __shfl(L, leader);
            const uint64_t m = __ballot(act && L == k);
            if (act && L == k) {
                const uint32_t r = w.running[k] + (uint32_t)__popcll(m & lanes_below());
                const uint32_t sym = w.hsym[i];
                w.code[sym] = w.first[k] + r;
                w.idx[sym] = (uint16_t)(w.lenbase[k] + r);
                w.sorted[w.lenbase[k] + r] = (uint16_t)sym;
            }
            __syncthreads();
            if (lane == leader) w.running[k] += (uint32_t)__popcll(m);
            __syncthreads();
            todo &= ~m;
        }
    }
    __syncthreads();
}

// ---- kernels ---------------------------------------------------------------------------------------------------

// symbol counts of every superblock (the running count[] of WFBB:833-837 is their prefix sum)
// (use_lds = 0 for alphabets whose histogram does not fit 48 KiB of LDS: atomics on the zeroed output row instead)
__global__ void k_wt_sb_counts(const int16_t *__restrict__ bwt, int64_t n, int sigma, int use_lds, uint32_t *__restrict__ out) {
    extern __shared__ uint32_t s_hist[];
    const int64_t sb = blockIdx.x;
    const int64_t beg = sb << kSbLog, end = min(beg + kSbs, n);
    if (!use_lds) {
        for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) atomicAdd(&out[sb * sigma + (uint16_t)bwt[i]], 1u);
        return;
    }
    for (int i = threadIdx.x; i < sigma; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) atomicAdd(&s_hist[(uint16_t)bwt[i]], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < sigma; i += blockDim.x) out[sb * sigma + i] = s_hist[i];
}
// the BWT in superblock-local codes: globalMapping[superblock][symbol] (WFBB:838-846)
__global__ void k_wt_localize(const int16_t *__restrict__ bwt, int64_t n, int sigma, const int16_t *__restrict__ gmap,
                              int16_t *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = gmap[(i >> kSbLog) * sigma + (uint16_t)bwt[i]];
}

struct LevelSums {
    unsigned long long hdr;  // sum over blocks of sigma*4 + (sigma-1)*2 + (mcl > 1 ? (mcl-1)*3 : 0)   WFBB:924-947
    unsigned long long unc;  // sum over blocks of frequency * code length
};

// WFBB:853-987, per block size 2^level (blockIdx.y = level - 9): the estimate's per-block terms, summed per superblock
__global__ __launch_bounds__(64) void k_wt_estimate(const int16_t *__restrict__ bwt, int64_t n, const int32_t *__restrict__ sb_sigma,
                              int sigma_pad, LevelSums *__restrict__ sums, int *__restrict__ too_deep) {
    extern __shared__ uint8_t s_raw[];
    const WaveLds w = carve(s_raw, sigma_pad);
    const int level = kMinLog + (int)blockIdx.y;
    const int64_t n_sb = (n + kSbs - 1) >> kSbLog;
    const int per_sb = 1 << (kSbLog - level);
    const int64_t total = n_sb * per_sb;
    for (int64_t g = blockIdx.x; g < total; g += gridDim.x) {
        const int64_t sb = g >> (kSbLog - level);
        const int64_t beg = (sb << kSbLog) + ((g & (per_sb - 1)) << level);
        const int64_t sb_end = min((sb + 1) << kSbLog, n);
        if (beg >= sb_end) continue;
        const int64_t end = min(beg + ((int64_t)1 << level), sb_end);
        block_histogram(w, sigma_pad, bwt + beg, (uint32_t)(end - beg));
        int th;
        uint64_t unc;
        const int s = wave_huffman(w, sb_sigma[sb], th, unc);
        if (threadIdx.x == 0) {
            unsigned long long hdr = (unsigned long long)s * 4 + (unsigned long long)(s - 1) * 2;
            if (th > 1) hdr += (unsigned long long)(th - 1) * 3;
            atomicAdd(&sums[sb * kLevels + (level - kMinLog)].hdr, hdr);
            atomicAdd(&sums[sb * kLevels + (level - kMinLog)].unc, (unsigned long long)unc);
            if (th > 31) *too_deep = 1;
        }
        __syncthreads();
    }
}

struct BlockInfo {
    uint32_t bv_size;   // bits of the block's node bit vectors (0 for a run block)
    uint32_t var_size;  // bytes of its variable-size header
    uint32_t ones;      // one-bits among them
    uint32_t shape;     // (sigma - 1) | tree height << 16
};
struct SbPlan {         // per superblock of the batch
    int64_t beg, end;   // symbols of the BWT
    int32_t bsl;
    int32_t n_blocks;
    int64_t blk_base;   // first block in the batch's dense per-block arrays
    int64_t var_base;   // first byte in the batch's header arena
    int64_t bv_base;    // first 32-bit word in the batch's bit-vector arena
    int64_t map_base;   // first entry in the batch's mapping arena
    int64_t inv_row;    // first entry of the superblock's row in the inverse table (local code -> global symbol)
    int32_t sigma;      // symbols present in the superblock = its local alphabet
    int32_t pad;
};

// pass 1 of WFBB:400-484 for every block at its superblock's chosen size
// (sigma = row length of the dense per-block tables: the largest local alphabet of the batch's superblocks)
__global__ __launch_bounds__(64) void k_wt_plan(const int16_t *__restrict__ bwt, int sigma, int sigma_pad, const SbPlan *__restrict__ plans,
                          BlockInfo *__restrict__ info, uint32_t *__restrict__ freq_dense) {
    extern __shared__ uint8_t s_raw[];
    const WaveLds w = carve(s_raw, sigma_pad);
    const SbPlan p = plans[blockIdx.y];
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < p.n_blocks; b += gridDim.x) {
        const int64_t beg = p.beg + ((int64_t)b << p.bsl), end = min(beg + ((int64_t)1 << p.bsl), p.end);
        block_histogram(w, sigma_pad, bwt + beg, (uint32_t)(end - beg));
        int th;
        uint64_t unc;
        const int s = wave_huffman(w, p.sigma, th, unc);
        wave_canonical(w, s, th > 31 ? 31 : th);
        uint64_t ones = 0;
        for (int i = lane; i < s; i += 64) ones += (uint64_t)w.hist[w.hsym[i]] * (uint32_t)__popc(w.code[w.hsym[i]]);
        ones = wave_sum_u64(ones);
        uint32_t *row = freq_dense + (p.blk_base + b) * (int64_t)sigma;
        for (int i = lane; i < sigma; i += 64) row[i] = w.hist[i];
        if (lane == 0) {
            BlockInfo bi;
            bi.bv_size = s > 1 ? (uint32_t)unc : 0u;
            bi.var_size = (uint32_t)((th > 1 ? (th - 1) * 4 : 0) + s * 5 + (s - 1) * 2);  // WFBB:479-483
            bi.ones = s > 1 ? (uint32_t)ones : 0u;
            bi.shape = (uint32_t)(s - 1) | ((uint32_t)th << 16);
            info[p.blk_base + b] = bi;
        }
        __syncthreads();
    }
}

struct SbTotals {
    uint32_t bv_bits, var_bytes, ones, pad;
};
// scans over the blocks of a superblock: block headers (WFBB:446-454 + bv_rank of WFBB:569) and, per symbol, the
// occurrences before every block (the leaves' u24, WFBB:774-788) — freq_dense becomes rank_dense in place
__global__ void k_wt_scan(int sigma, const SbPlan *__restrict__ plans, const BlockInfo *__restrict__ info,
                          uint32_t *__restrict__ dense, BlockHeader *__restrict__ headers, SbTotals *__restrict__ totals) {
    const SbPlan p = plans[blockIdx.x];
    for (int sym = threadIdx.x; sym < sigma; sym += blockDim.x) {
        uint32_t acc = 0;
        uint32_t *col = dense + p.blk_base * (int64_t)sigma + sym;
        for (int b = 0; b < p.n_blocks; ++b) {
            const uint32_t f = col[(int64_t)b * sigma];
            col[(int64_t)b * sigma] = acc;
            acc += f;
        }
    }
    if (threadIdx.x == 0) {
        uint32_t bv = 0, var = 0, ones = 0;
        for (int b = 0; b < p.n_blocks; ++b) {
            const BlockInfo bi = info[p.blk_base + b];
            BlockHeader h;
            h.bv_rank = (int32_t)ones;
            h.bv_offset = (int32_t)bv;
            h.var_off = (int32_t)var;
            h.sigma = (int16_t)(bi.shape & 0xffffu);
            h.tree_height = (int16_t)(bi.shape >> 16);
            headers[p.blk_base + b] = h;
            bv += bi.bv_size;
            var += bi.var_size;
            ones += bi.ones;
        }
        SbTotals t = {bv, var, ones, 0};
        totals[blockIdx.x] = t;
    }
}

__device__ __forceinline__ void wr16(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v & 0xffu);
    p[1] = (uint8_t)((v >> 8) & 0xffu);
}

// pass 2 (WFBB:499-531 + 570-810): variable-size headers, mapping entries and node bit vectors of every block
__global__ __launch_bounds__(64) void k_wt_encode(const int16_t *__restrict__ bwt, int sigma, int sigma_pad, int sigma_global,
                            const SbPlan *__restrict__ plans, const BlockHeader *__restrict__ headers,
                            const uint32_t *__restrict__ rank_dense, const uint16_t *__restrict__ inv_table,
                            uint8_t *__restrict__ var_arena, uint32_t *__restrict__ bv_arena,
                            int16_t *__restrict__ map_arena) {
    extern __shared__ uint8_t s_raw[];
    const WaveLds w = carve(s_raw, sigma_pad);
    const SbPlan p = plans[blockIdx.y];
    const int lane = threadIdx.x;
    const int64_t blocks_per_sb = kSbs >> p.bsl;
    for (int b = blockIdx.x; b < p.n_blocks; b += gridDim.x) {
        const int64_t beg = p.beg + ((int64_t)b << p.bsl), end = min(beg + ((int64_t)1 << p.bsl), p.end);
        const uint32_t count = (uint32_t)(end - beg);
        block_histogram(w, sigma_pad, bwt + beg, count);
        int th;
        uint64_t unc;
        const int s = wave_huffman(w, p.sigma, th, unc);
        wave_canonical(w, s, th);
        const BlockHeader bh = headers[p.blk_base + b];
        uint8_t *hdr = var_arena + p.var_base + bh.var_off;
        const uint32_t *ranks = rank_dense + (p.blk_base + b) * (int64_t)sigma;
        const int lvl_bytes = th > 1 ? (th - 1) * 4 : 0;
        // level table (WFBB:713-760): leaves per level, bits per level - 1
        if (lane == 0) {
            uint32_t deeper = 0;  // symbols (with multiplicity) whose code is longer than d
            for (int d = th; d >= 1; --d) {
                if (d < th) {
                    wr16(hdr + (d - 1) * 4, w.cnt[d] & 0xffffu);
                    wr16(hdr + (d - 1) * 4 + 2, (deeper - 1u) & 0xffffu);
                }
                deeper += w.lf[d];
            }
        }
        // leaves (WFBB:774-788) in (length, symbol) order and the superblock's mapping entries (WFBB:461-471)
        for (int i = lane; i < s; i += 64) {
            const uint32_t sym = w.hsym[i];
            const uint32_t ix = w.idx[sym];
            uint8_t *lp = hdr + lvl_bytes + ix * 5;
            const uint32_t rv = ranks[sym];
            wr16(lp, inv_table[p.inv_row + sym]);  // the leaf names the GLOBAL symbol (WFBB:774-788)
            lp[2] = (uint8_t)(rv & 0xffu);
            lp[3] = (uint8_t)((rv >> 8) & 0xffu);
            lp[4] = (uint8_t)((rv >> 16) & 0xffu);
            const int64_t row = sym;  // = globalMapping[superblock][symbol]: the local code itself
            const int clamp = sigma_global - 2;
            map_arena[p.map_base + row * blocks_per_sb + b] = (int16_t)((int)ix < clamp ? (int)ix : clamp);
        }
        if (s > 1) {
            const int n_nodes = s - 1;
            for (int k = lane; k < n_nodes; k += 64) {
                w.nsize[k] = 0;
                w.nfill[k] = 0;
                w.nones[k] = 0;
            }
            __syncthreads();
            // bits per internal node: every symbol passes through one node per level above its leaf
            for (int i = lane; i < s; i += 64) {
                const uint32_t sym = w.hsym[i];
                const uint32_t L = w.hdep[i], c = w.code[sym], f = w.hist[sym];
                for (uint32_t d = 0; d < L; ++d) atomicAdd(&w.nsize[w.lvl_base[d] + (c >> (L - d)) - w.lvl_first[d]], f);
            }
            __syncthreads();
            if (lane == 0) {
                uint32_t acc = (uint32_t)bh.bv_offset;
                for (int k = 0; k < n_nodes; ++k) {
                    w.noff[k] = acc;
                    acc += w.nsize[k];
                }
            }
            __syncthreads();
            // WFBB:606-708: every symbol appends one bit to each internal node on its path; its position in the node
            // is the number of earlier symbols routed through that node
            uint32_t *bv = bv_arena + p.bv_base;
            for (uint32_t base = 0; base < count; base += 64) {
                const uint32_t i = base + lane;
                const bool valid = i < count;
                const uint32_t sym = valid ? (uint32_t)(uint16_t)bwt[beg + i] : 0u;
                const uint32_t L = valid ? w.len[sym] : 0u;
                const uint32_t c = valid ? w.code[sym] : 0u;
                const uint32_t deepest = wave_max_u32(L);
                for (uint32_t d = 0; d < deepest; ++d) {
                    const bool act = valid && d < L;
                    const uint32_t node = act ? w.lvl_base[d] + (c >> (L - d)) - w.lvl_first[d] : 0u;
                    const bool bit = act && ((c >> (L - d - 1)) & 1u);
                    uint64_t todo = __ballot(act);
                    while (todo) {
                        const int leader = __ffsll((unsigned long long)todo) - 1;
                        const uint32_t k = __shfl(node, leader);
                        const bool mine = act && node == k;
                        const uint64_t m = __ballot(mine);
                        const uint64_t m1 = __ballot(mine && bit);
                        const uint32_t fill = w.nfill[k];
                        if (mine && bit) {
                            const uint32_t pos = w.noff[k] + fill + (uint32_t)__popcll(m & lanes_below());
                            atomicOr(&bv[pos >> 5], 1u << (pos & 31u));
                        }
                        __syncthreads();
                        if (lane == leader) {
                            w.nfill[k] = fill + (uint32_t)__popcll(m);
                            w.nones[k] += (uint32_t)__popcll(m1);
                        }
                        __syncthreads();
                        todo &= ~m;
                    }
                }
            }
            __syncthreads();
            // cumulative one-counts per level, WFBB:793-809 (u16 wrap at WFBB:798)
            if (lane == 0) {
                uint8_t *cp = hdr + lvl_bytes + s * 5;
                int ptr = 0;
                for (int d = 0; d < th; ++d) {
                    const int n_internal = (int)((1u << d) - w.lvl_first[d]);
                    uint32_t level_ones = 0;
                    for (int j = 0; j < n_internal; ++j) {
                        level_ones += w.nones[ptr++];
                        wr16(cp, level_ones & 0xffffu);
                        cp += 2;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- RRR of every superblock's bit vector (RRR:225-286) --------------------------------------------------------
struct RrrPlan {        // per superblock of the batch
    int64_t bv_base;    // first 32-bit word of the bit vector in the arena
    uint32_t nbits;
    uint32_t n_blocks;  // 15-bit blocks (RRR:232)
    uint32_t n_groups;  // sample groups of sample_size blocks
    int64_t grp_base;   // first group in the dense group arrays
    int64_t cls_base;   // first 32-bit word of the class vector in its arena
    int64_t off_base;   // first 64-bit word of the offsets stream
    int64_t so_base;    // ... of lengthOfSampledOffsets
    int64_t ps_base;    // ... of prefixSums
    int32_t w_so, w_ps; // their widths
};
__constant__ uint8_t c_bits_needed[16] = {1, 4, 7, 9, 11, 12, 13, 13, 13, 13, 12, 11, 9, 7, 4, 1};

__device__ __forceinline__ uint32_t bits15_at(const uint32_t *__restrict__ bv, uint32_t from, uint32_t nbits) {
    const uint32_t len = nbits - from < 15u ? nbits - from : 15u;
    const uint32_t wi = from >> 5, sh = from & 31u;
    uint64_t v = (uint64_t)bv[wi] | ((uint64_t)bv[wi + 1] << 32);
    return (uint32_t)(v >> sh) & ((1u << len) - 1u);
}
__device__ __forceinline__ void or_bits64(unsigned long long *words, uint64_t bit, uint64_t value, int nbits) {
    if (nbits <= 0) return;
    value &= nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
    const uint64_t wi = bit >> 6;
    const int sh = (int)(bit & 63);
    if (value << sh) atomicOr(&words[wi], (unsigned long long)(value << sh));
    if (sh + nbits > 64 && (value >> (64 - sh))) atomicOr(&words[wi + 1], (unsigned long long)(value >> (64 - sh)));
}

// one thread per sample group: classes (4 bits per block, RRR:252-258) and the group's offset bits / ones
__global__ void k_wt_rrr_classes(const uint32_t *__restrict__ bv_arena, const RrrPlan *__restrict__ plans, int sample,
                                 uint32_t *__restrict__ cls_arena, uint32_t *__restrict__ grp_bits,
                                 uint32_t *__restrict__ grp_ones) {
    const RrrPlan p = plans[blockIdx.y];
    const uint32_t *bv = bv_arena + p.bv_base;
    uint32_t *cls = cls_arena + p.cls_base;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < p.n_groups; g += gridDim.x * blockDim.x) {
        const uint32_t lo = g * (uint32_t)sample, hi = min(lo + (uint32_t)sample, p.n_blocks);
        uint32_t ob = 0, ones = 0;
        for (uint32_t b = lo; b < hi; ++b) {
            const uint32_t k = (uint32_t)__popc(bits15_at(bv, b * 15u, p.nbits));
            atomicOr(&cls[b >> 3], k << (4u * (b & 7u)));
            ob += c_bits_needed[k];
            ones += k;
        }
        grp_bits[p.grp_base + g] = ob;
        grp_ones[p.grp_base + g] = ones;
    }
}
// exclusive scan over the groups of a superblock, in place; totals per superblock
__global__ void k_wt_rrr_scan(const RrrPlan *__restrict__ plans, uint32_t *__restrict__ grp_bits,
                              uint32_t *__restrict__ grp_ones, uint32_t *__restrict__ totals) {
    __shared__ uint32_t s_bits[256], s_ones[256];
    const RrrPlan p = plans[blockIdx.x];
    const uint32_t per = (p.n_groups + 255u) / 256u;
    const uint32_t lo = min(threadIdx.x * per, p.n_groups), hi = min(lo + per, p.n_groups);
    uint32_t sb = 0, so = 0;
    for (uint32_t g = lo; g < hi; ++g) {
        sb += grp_bits[p.grp_base + g];
        so += grp_ones[p.grp_base + g];
    }
    s_bits[threadIdx.x] = sb;
    s_ones[threadIdx.x] = so;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ab = 0, ao = 0;
        for (int t = 0; t < 256; ++t) {
            const uint32_t b = s_bits[t], o = s_ones[t];
            s_bits[t] = ab;
            s_ones[t] = ao;
            ab += b;
            ao += o;
        }
        totals[2 * blockIdx.x] = ab;
        totals[2 * blockIdx.x + 1] = ao;
    }
    __syncthreads();
    uint32_t ab = s_bits[threadIdx.x], ao = s_ones[threadIdx.x];
    for (uint32_t g = lo; g < hi; ++g) {
        const uint32_t b = grp_bits[p.grp_base + g], o = grp_ones[p.grp_base + g];
        grp_bits[p.grp_base + g] = ab;
        grp_ones[p.grp_base + g] = ao;
        ab += b;
        ao += o;
    }
}
// one thread per sample group: offsets (RRR:266-276) and the group's sample (RRR:277-283)
__global__ void k_wt_rrr_offsets(const uint32_t *__restrict__ bv_arena, const RrrPlan *__restrict__ plans, int sample,
                                 const uint16_t *__restrict__ offset_of_value, const uint32_t *__restrict__ grp_bits,
                                 const uint32_t *__restrict__ grp_ones, unsigned long long *__restrict__ off_arena,
                                 unsigned long long *__restrict__ so_arena, unsigned long long *__restrict__ ps_arena) {
    const RrrPlan p = plans[blockIdx.y];
    const uint32_t *bv = bv_arena + p.bv_base;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < p.n_groups; g += gridDim.x * blockDim.x) {
        const uint32_t lo = g * (uint32_t)sample, hi = min(lo + (uint32_t)sample, p.n_blocks);
        uint64_t cur = grp_bits[p.grp_base + g];
        or_bits64(so_arena + p.so_base, (uint64_t)g * (uint32_t)p.w_so, cur, p.w_so);
        or_bits64(ps_arena + p.ps_base, (uint64_t)g * (uint32_t)p.w_ps, grp_ones[p.grp_base + g], p.w_ps);
        for (uint32_t b = lo; b < hi; ++b) {
            const uint32_t v = bits15_at(bv, b * 15u, p.nbits);
            const int nb = c_bits_needed[__popc(v)];
            or_bits64(off_arena + p.off_base, cur, offset_of_value[v], nb);
            cur += (uint64_t)nb;
        }
    }
}

template <class T>
hipError_t upload(DevMem &mem, T **d, const std::vector<T> &h) {
    hipError_t e = mem.alloc(d, h.size());
    if (e != hipSuccess) return e;
    return h.empty() ? hipSuccess : hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace

int device_wavelet_stage(const int16_t *d_bwt_global, int64_t n, int sampling_rate, int alphabet, WfbbModel &w, std::string &err) {
    const int sigma = alphabet;
    if (sigma < 1 || sigma > 32768) return 1;  // not handled here: the caller encodes on the host
    const int64_t n_sb = (n + kSbs - 1) >> kSbLog;
    const int64_t n_hb = (n + (1ll << 32) - 1) >> 32;
    DevMem mem;

    // symbol counts per superblock -> count[], hyperBlockRank, superBlockRank, globalMapping, sigma (WFBB:812-851)
    uint32_t *d_counts = nullptr;
    const int counts_in_lds = (size_t)sigma * 4 <= (48u << 10) ? 1 : 0;
    WT_TRY(mem.alloc(&d_counts, (size_t)(n_sb * sigma), !counts_in_lds));
    hipLaunchKernelGGL(k_wt_sb_counts, dim3((unsigned)n_sb), dim3(1024), counts_in_lds ? (size_t)sigma * 4 : 0, 0, d_bwt_global, n, sigma,
                       counts_in_lds, d_counts);
    std::vector<uint32_t> sb_counts((size_t)(n_sb * sigma));
    WT_TRY(hipMemcpy(sb_counts.data(), d_counts, sb_counts.size() * 4, hipMemcpyDeviceToHost));
    w.size = n;
    w.sampling_rate = sampling_rate;
    w.alphabet_size = sigma;
    w.count.assign((size_t)sigma, 0);
    w.hyper_rank.assign((size_t)(n_hb * sigma), 0);
    w.super_rank.assign((size_t)(n_sb * sigma), 0);
    w.global_mapping.assign((size_t)(n_sb * sigma), (int16_t)(sigma - 1));
    w.sb.assign((size_t)n_sb, SuperBlockModel());
    std::vector<int64_t> running((size_t)sigma, 0);
    std::vector<int64_t> sb_sigma((size_t)n_sb, 0);
    for (int64_t s = 0; s < n_sb; ++s) {
        const int64_t hb = (s << kSbLog) >> 32;
        if (((s << kSbLog) & ((1ll << 32) - 1)) == 0)  // WFBB:819-825
            for (int i = 0; i < sigma; ++i) w.hyper_rank[(size_t)(hb * sigma + i)] = running[(size_t)i];
        int64_t present = 0;
        for (int i = 0; i < sigma; ++i) {
            w.super_rank[(size_t)(s * sigma + i)] = (int32_t)(running[(size_t)i] - w.hyper_rank[(size_t)(hb * sigma + i)]);
            if (sb_counts[(size_t)(s * sigma + i)]) w.global_mapping[(size_t)(s * sigma + i)] = (int16_t)present++;
            running[(size_t)i] += sb_counts[(size_t)(s * sigma + i)];
        }
        sb_sigma[(size_t)s] = present;
        w.sb[(size_t)s].sigma = (int16_t)(present - 1);
    }
    for (int i = 0; i < sigma; ++i) w.count[(size_t)i] = running[(size_t)i];

    // from here on the BWT holds superblock-local codes; sigma_l = the largest local alphabet (row length of the inverse
    // table and of the dense per-block tables, LDS per wave)
    int64_t sigma_l64 = 1;
    for (int64_t s = 0; s < n_sb; ++s) sigma_l64 = std::max(sigma_l64, sb_sigma[(size_t)s]);
    if (sigma_l64 > kWtMaxSigma) return 1;  // a superblock with more distinct symbols than a wave's LDS holds: host encoder
    const int sigma_l = (int)sigma_l64;
    const int sigma_pad = (sigma_l + 63) & ~63;
    const size_t lds = wave_lds_bytes(sigma_pad);
    int16_t *d_gmap = nullptr;
    uint16_t *d_inv = nullptr;
    int32_t *d_sbsig = nullptr;
    {
        std::vector<uint16_t> inv((size_t)(n_sb * sigma_l), 0);
        std::vector<int32_t> sbsig((size_t)n_sb);
        for (int64_t s = 0; s < n_sb; ++s) {
            sbsig[(size_t)s] = (int32_t)sb_sigma[(size_t)s];
            for (int i = 0; i < sigma; ++i)
                if (sb_counts[(size_t)(s * sigma + i)]) inv[(size_t)(s * sigma_l + w.global_mapping[(size_t)(s * sigma + i)])] = (uint16_t)i;
        }
        WT_TRY(upload(mem, &d_gmap, w.global_mapping));
        WT_TRY(upload(mem, &d_inv, inv));
        WT_TRY(upload(mem, &d_sbsig, sbsig));
    }
    int16_t *d_bwt = nullptr;  // (a copy: the caller's BWT stays as it is for the host encoder, should this stage hand the text back)
    WT_TRY(mem.alloc(&d_bwt, (size_t)n));
    hipLaunchKernelGGL(k_wt_localize, dim3(4096), dim3(256), 0, 0, d_bwt_global, n, sigma, d_gmap, d_bwt);

    // block size of every superblock (WFBB:853-987)
    LevelSums *d_sums = nullptr;
    int *d_flag = nullptr;
    WT_TRY(mem.alloc(&d_sums, (size_t)(n_sb * kLevels), true));
    WT_TRY(mem.alloc(&d_flag, 1, true));
    {
        const int64_t most = n_sb << (kSbLog - kMinLog);
        const unsigned gx = (unsigned)std::min<int64_t>(most, 256 * 64);
        hipLaunchKernelGGL(k_wt_estimate, dim3(gx, kLevels), dim3(64), lds, 0, d_bwt, n, d_sbsig, sigma_pad, d_sums, d_flag);
    }
    std::vector<LevelSums> sums((size_t)(n_sb * kLevels));
    int too_deep = 0;
    WT_TRY(hipMemcpy(sums.data(), d_sums, sums.size() * sizeof(LevelSums), hipMemcpyDeviceToHost));
    WT_TRY(hipMemcpy(&too_deep, d_flag, 4, hipMemcpyDeviceToHost));
    if (too_deep) return 1;  // a code longer than 31 bits somewhere: the host encoder takes it
    std::vector<int> bsl((size_t)n_sb);
    for (int64_t s = 0; s < n_sb; ++s) {
        int64_t hdr[kLevels], unc[kLevels];
        for (int l = 0; l < kLevels; ++l) {
            hdr[l] = (int64_t)sums[(size_t)(s * kLevels + l)].hdr;
            unc[l] = (int64_t)sums[(size_t)(s * kLevels + l)].unc;
        }
        const int64_t sb_size = std::min(kSbs, n - (s << kSbLog));
        bsl[(size_t)s] = pick_block_size_log(hdr, unc, sb_size, sb_sigma[(size_t)s], sampling_rate);
        w.sb[(size_t)s].block_size_log = (int16_t)bsl[(size_t)s];
    }

    uint16_t *d_oov = nullptr;
    WT_TRY(mem.alloc(&d_oov, 32768));
    WT_TRY(hipMemcpy(d_oov, rrr_offset_of_value(), 32768 * 2, hipMemcpyHostToDevice));

    // superblocks are encoded in batches whose dense per-block tables stay below ~1 GiB
    const int64_t dense_budget = 1ll << 30;
    int64_t s0 = 0;
    while (s0 < n_sb) {
        DevMem bm;
        std::vector<SbPlan> plans;
        int64_t blocks = 0, map_entries = 0;
        int64_t s1 = s0;
        while (s1 < n_sb) {
            const int64_t beg = s1 << kSbLog, end = std::min(beg + kSbs, n);
            const int b = bsl[(size_t)s1];
            const int64_t nb = (end - beg + (1ll << b) - 1) >> b;
            if (!plans.empty() && (blocks + nb) * sigma_l * 4 > dense_budget) break;
            SbPlan p;
            p.beg = beg;
            p.end = end;
            p.bsl = b;
            p.n_blocks = (int32_t)nb;
            p.blk_base = blocks;
            p.var_base = p.bv_base = 0;
            p.map_base = map_entries;
            p.inv_row = s1 * sigma_l;
            p.sigma = (int32_t)sb_sigma[(size_t)s1];
            p.pad = 0;
            plans.push_back(p);
            blocks += nb;
            map_entries += sb_sigma[(size_t)s1] * (kSbs >> b);
            ++s1;
        }
        const int nsb = (int)plans.size();
        SbPlan *d_plans = nullptr;
        BlockInfo *d_info = nullptr;
        uint32_t *d_dense = nullptr;
        BlockHeader *d_headers = nullptr;
        SbTotals *d_totals = nullptr;
        WT_TRY(upload(bm, &d_plans, plans));
        WT_TRY(bm.alloc(&d_info, (size_t)blocks));
        WT_TRY(bm.alloc(&d_dense, (size_t)(blocks * sigma_l)));
        WT_TRY(bm.alloc(&d_headers, (size_t)blocks));
        WT_TRY(bm.alloc(&d_totals, (size_t)nsb));
        hipLaunchKernelGGL(k_wt_plan, dim3(64, (unsigned)nsb), dim3(64), lds, 0, d_bwt, sigma_l, sigma_pad, d_plans, d_info,
                           d_dense);
        hipLaunchKernelGGL(k_wt_scan, dim3((unsigned)nsb), dim3(256), 0, 0, sigma_l, d_plans, d_info, d_dense, d_headers,
                           d_totals);
        std::vector<SbTotals> totals((size_t)nsb);
        WT_TRY(hipMemcpy(totals.data(), d_totals, totals.size() * sizeof(SbTotals), hipMemcpyDeviceToHost));
        int64_t var_bytes = 0, bv_words = 0;
        for (int k = 0; k < nsb; ++k) {
            plans[(size_t)k].var_base = var_bytes;
            plans[(size_t)k].bv_base = bv_words;
            var_bytes += ((int64_t)totals[(size_t)k].var_bytes + 15) & ~15ll;
            bv_words += (((int64_t)totals[(size_t)k].bv_bits + 63) / 64 + 2) * 2;
        }
        WT_TRY(hipMemcpy(d_plans, plans.data(), plans.size() * sizeof(SbPlan), hipMemcpyHostToDevice));
        uint8_t *d_var = nullptr;
        uint32_t *d_bv = nullptr;
        int16_t *d_map = nullptr;
        WT_TRY(bm.alloc(&d_var, (size_t)var_bytes, true));
        WT_TRY(bm.alloc(&d_bv, (size_t)bv_words, true));
        WT_TRY(bm.alloc(&d_map, (size_t)map_entries));
        {  // absent marker alphabetSize - 1 (WFBB:377-387): 16-bit fill
            const int16_t absent = (int16_t)(sigma - 1);
            if (map_entries) WT_TRY(hipMemsetD16((hipDeviceptr_t)d_map, (unsigned short)absent, (size_t)map_entries));
        }
        hipLaunchKernelGGL(k_wt_encode, dim3(64, (unsigned)nsb), dim3(64), lds, 0, d_bwt, sigma_l, sigma_pad, sigma, d_plans,
                           d_headers, d_dense, d_inv, d_var, d_bv, d_map);

        // RRR vectors
        std::vector<RrrPlan> rp((size_t)nsb);
        int64_t groups = 0, cls_words = 0;
        for (int k = 0; k < nsb; ++k) {
            RrrPlan &r = rp[(size_t)k];
            memset(&r, 0, sizeof r);
            r.bv_base = plans[(size_t)k].bv_base;
            r.nbits = totals[(size_t)k].bv_bits;
            r.n_blocks = r.nbits / 15 + (r.nbits % 15 ? 1 : 0);
            r.n_groups = (r.n_blocks + (uint32_t)sampling_rate - 1) / (uint32_t)sampling_rate;
            r.grp_base = groups;
            r.cls_base = cls_words;
            groups += r.n_groups;
            cls_words += words_for_bits((int64_t)r.n_blocks * 4) * 2;
        }
        RrrPlan *d_rp = nullptr;
        uint32_t *d_cls = nullptr, *d_gbits = nullptr, *d_gones = nullptr, *d_rtot = nullptr;
        WT_TRY(upload(bm, &d_rp, rp));
        WT_TRY(bm.alloc(&d_cls, (size_t)cls_words, true));
        WT_TRY(bm.alloc(&d_gbits, (size_t)groups));
        WT_TRY(bm.alloc(&d_gones, (size_t)groups));
        WT_TRY(bm.alloc(&d_rtot, (size_t)nsb * 2));
        hipLaunchKernelGGL(k_wt_rrr_classes, dim3(32, (unsigned)nsb), dim3(256), 0, 0, d_bv, d_rp, sampling_rate, d_cls,
                           d_gbits, d_gones);
        hipLaunchKernelGGL(k_wt_rrr_scan, dim3((unsigned)nsb), dim3(256), 0, 0, d_rp, d_gbits, d_gones, d_rtot);
        std::vector<uint32_t> rtot((size_t)nsb * 2);
        WT_TRY(hipMemcpy(rtot.data(), d_rtot, rtot.size() * 4, hipMemcpyDeviceToHost));
        int64_t off_words = 0, so_words = 0, ps_words = 0;
        for (int k = 0; k < nsb; ++k) {
            RrrPlan &r = rp[(size_t)k];
            RrrModel &m = w.sb[(size_t)(s0 + k)].rank_support;
            const int64_t total_bits = rtot[(size_t)k * 2], total_ones = rtot[(size_t)k * 2 + 1];
            m.sample_size = sampling_rate;
            m.length = (int32_t)r.nbits;
            m.total_ones = (int32_t)total_ones;
            m.bits_per_offset_pos = min_bits((uint64_t)total_bits);                                     // RRR:262
            m.classes.init((int32_t)r.n_blocks, 4);
            m.sampled_offsets.init((int32_t)(r.n_blocks / (uint32_t)sampling_rate + 1), m.bits_per_offset_pos);  // RRR:263
            m.prefix_sums.init((int32_t)(r.n_blocks / (uint32_t)sampling_rate + 2), min_bits((uint64_t)total_ones));  // RRR:264
            m.offsets.assign((size_t)words_for_bits(total_bits), 0);
            r.w_so = m.sampled_offsets.width;
            r.w_ps = m.prefix_sums.width;
            r.off_base = off_words;
            r.so_base = so_words;
            r.ps_base = ps_words;
            off_words += (int64_t)m.offsets.size() + 1;
            so_words += (int64_t)m.sampled_offsets.words.size() + 1;
            ps_words += (int64_t)m.prefix_sums.words.size() + 1;
        }
        WT_TRY(hipMemcpy(d_rp, rp.data(), rp.size() * sizeof(RrrPlan), hipMemcpyHostToDevice));
        unsigned long long *d_off = nullptr, *d_so = nullptr, *d_ps = nullptr;
        WT_TRY(bm.alloc(&d_off, (size_t)off_words, true));
        WT_TRY(bm.alloc(&d_so, (size_t)so_words, true));
        WT_TRY(bm.alloc(&d_ps, (size_t)ps_words, true));
        hipLaunchKernelGGL(k_wt_rrr_offsets, dim3(32, (unsigned)nsb), dim3(256), 0, 0, d_bv, d_rp, sampling_rate, d_oov,
                           d_gbits, d_gones, d_off, d_so, d_ps);
        WT_TRY(hipGetLastError());

        // results -> host model
        std::vector<uint8_t> h_var((size_t)var_bytes);
        std::vector<int16_t> h_map((size_t)map_entries);
        std::vector<BlockHeader> h_headers((size_t)blocks);
        std::vector<uint32_t> h_cls((size_t)cls_words);
        std::vector<uint64_t> h_off((size_t)off_words), h_so((size_t)so_words), h_ps((size_t)ps_words);
        if (var_bytes) WT_TRY(hipMemcpy(h_var.data(), d_var, h_var.size(), hipMemcpyDeviceToHost));
        if (map_entries) WT_TRY(hipMemcpy(h_map.data(), d_map, h_map.size() * 2, hipMemcpyDeviceToHost));
        WT_TRY(hipMemcpy(h_headers.data(), d_headers, h_headers.size() * sizeof(BlockHeader), hipMemcpyDeviceToHost));
        if (cls_words) WT_TRY(hipMemcpy(h_cls.data(), d_cls, h_cls.size() * 4, hipMemcpyDeviceToHost));
        if (off_words) WT_TRY(hipMemcpy(h_off.data(), d_off, h_off.size() * 8, hipMemcpyDeviceToHost));
        if (so_words) WT_TRY(hipMemcpy(h_so.data(), d_so, h_so.size() * 8, hipMemcpyDeviceToHost));
        if (ps_words) WT_TRY(hipMemcpy(h_ps.data(), d_ps, h_ps.size() * 8, hipMemcpyDeviceToHost));
        for (int k = 0; k < nsb; ++k) {
            const SbPlan &p = plans[(size_t)k];
            const RrrPlan &r = rp[(size_t)k];
            SuperBlockModel &sb = w.sb[(size_t)(s0 + k)];
            sb.block_headers.assign(h_headers.begin() + p.blk_base, h_headers.begin() + p.blk_base + p.n_blocks);
            sb.var.assign(h_var.begin() + p.var_base, h_var.begin() + p.var_base + totals[(size_t)k].var_bytes);
            const int64_t n_map = sb_sigma[(size_t)(s0 + k)] * (kSbs >> p.bsl);
            sb.mapping.assign(h_map.begin() + p.map_base, h_map.begin() + p.map_base + n_map);
            RrrModel &m = sb.rank_support;
            if (!m.classes.words.empty()) memcpy(m.classes.words.data(), h_cls.data() + r.cls_base, m.classes.words.size() * 8);
            if (!m.offsets.empty()) memcpy(m.offsets.data(), h_off.data() + r.off_base, m.offsets.size() * 8);
            if (!m.sampled_offsets.words.empty())
                memcpy(m.sampled_offsets.words.data(), h_so.data() + r.so_base, m.sampled_offsets.words.size() * 8);
            if (!m.prefix_sums.words.empty())
                memcpy(m.prefix_sums.words.data(), h_ps.data() + r.ps_base, m.prefix_sums.words.size() * 8);
            // RRR:285: one more prefix sum after the last sampled block
            const int64_t n_sampled = r.n_blocks == 0 ? 0 : (int64_t)(r.n_blocks - 1) / sampling_rate + 1;
            m.prefix_sums.set(n_sampled, (uint64_t)(uint32_t)m.total_ones);
        }
        s0 = s1;
    }
    return 0;
}


// ---- the FM-index's own vectors (FM:343-370) -------------------------------------------------------------------
namespace {
// IntVector packing (IV:91-119): element k = vals[k] (k < n_vals; k == wrap_index: vals[0], the wrap entry of
// FM:367-369; else 0) at bit k * width; one thread per 64-bit word
__global__ void k_pack_values(const uint32_t *__restrict__ vals, int64_t n_vals, int64_t length, int width,
                              int64_t wrap_index, unsigned long long *__restrict__ words, int64_t n_words) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_words) return;
    const int64_t lo_bit = j * 64, hi_bit = lo_bit + 64;
    unsigned long long acc = 0;
    for (int64_t k = lo_bit / width; k < length && k * width < hi_bit; ++k) {
        const uint64_t v = k < n_vals ? vals[k] : (k == wrap_index ? vals[0] : 0u);
        const int64_t at = k * width;
        const uint64_t masked = width >= 64 ? v : (v & ((1ull << width) - 1ull));
        if (at >= lo_bit)
            acc |= masked << (at - lo_bit);
        else
            acc |= masked >> (lo_bit - at);
    }
    words[j] = acc;
}
}  // namespace

int device_pack_values(const uint32_t *d_vals, int64_t n_vals, int64_t length, int width, int64_t wrap_index,
                       PackedVec &out, std::string &err) {
    out.init((int32_t)length, width);
    const int64_t n_words = (int64_t)out.words.size();
    if (n_words == 0) return 0;
    DevMem mem;
    unsigned long long *d_words = nullptr;
    WT_TRY(mem.alloc(&d_words, (size_t)n_words));
    hipLaunchKernelGGL(k_pack_values, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, 0, d_vals, n_vals, length, width,
                       wrap_index, d_words, n_words);
    WT_TRY(hipGetLastError());
    WT_TRY(hipMemcpy(out.words.data(), d_words, (size_t)n_words * 8, hipMemcpyDeviceToHost));
    return 0;
}

// RRR:225-286 of one bit vector in HBM (d_bits: LSB-first 64-bit words, two guard words behind the last bit)
int device_rrr_of_bits(const uint64_t *d_bits, int64_t nbits, int sample, RrrModel &m, std::string &err) {
    DevMem mem;
    RrrPlan r;
    memset(&r, 0, sizeof r);
    r.nbits = (uint32_t)nbits;
    r.n_blocks = (uint32_t)(nbits / 15 + (nbits % 15 ? 1 : 0));
    r.n_groups = (r.n_blocks + (uint32_t)sample - 1) / (uint32_t)sample;
    const uint32_t *d_bv = reinterpret_cast<const uint32_t *>(d_bits);
    RrrPlan *d_rp = nullptr;
    uint32_t *d_cls = nullptr, *d_gbits = nullptr, *d_gones = nullptr, *d_rtot = nullptr;
    uint16_t *d_oov = nullptr;
    const int64_t cls_words = words_for_bits((int64_t)r.n_blocks * 4) * 2;
    WT_TRY(mem.alloc(&d_rp, 1));
    WT_TRY(hipMemcpy(d_rp, &r, sizeof r, hipMemcpyHostToDevice));
    WT_TRY(mem.alloc(&d_cls, (size_t)cls_words, true));
    WT_TRY(mem.alloc(&d_gbits, (size_t)r.n_groups));
    WT_TRY(mem.alloc(&d_gones, (size_t)r.n_groups));
    WT_TRY(mem.alloc(&d_rtot, 2));
    WT_TRY(mem.alloc(&d_oov, 32768));
    WT_TRY(hipMemcpy(d_oov, rrr_offset_of_value(), 32768 * 2, hipMemcpyHostToDevice));
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(((int64_t)r.n_groups + 255) / 256, 4096));
    hipLaunchKernelGGL(k_wt_rrr_classes, dim3(gx, 1), dim3(256), 0, 0, d_bv, d_rp, sample, d_cls, d_gbits, d_gones);
    hipLaunchKernelGGL(k_wt_rrr_scan, dim3(1), dim3(256), 0, 0, d_rp, d_gbits, d_gones, d_rtot);
    uint32_t rtot[2] = {0, 0};
    WT_TRY(hipMemcpy(rtot, d_rtot, 8, hipMemcpyDeviceToHost));
    m.sample_size = sample;
    m.length = (int32_t)nbits;
    m.total_ones = (int32_t)rtot[1];
    m.bits_per_offset_pos = min_bits((uint64_t)rtot[0]);
    m.classes.init((int32_t)r.n_blocks, 4);
    m.sampled_offsets.init((int32_t)(r.n_blocks / (uint32_t)sample + 1), m.bits_per_offset_pos);
    m.prefix_sums.init((int32_t)(r.n_blocks / (uint32_t)sample + 2), min_bits((uint64_t)rtot[1]));
    m.offsets.assign((size_t)words_for_bits((int64_t)rtot[0]), 0);
    r.w_so = m.sampled_offsets.width;
    r.w_ps = m.prefix_sums.width;
    WT_TRY(hipMemcpy(d_rp, &r, sizeof r, hipMemcpyHostToDevice));
    unsigned long long *d_off = nullptr, *d_so = nullptr, *d_ps = nullptr;
    WT_TRY(mem.alloc(&d_off, m.offsets.size() + 1, true));
    WT_TRY(mem.alloc(&d_so, m.sampled_offsets.words.size() + 1, true));
    WT_TRY(mem.alloc(&d_ps, m.prefix_sums.words.size() + 1, true));
    hipLaunchKernelGGL(k_wt_rrr_offsets, dim3(gx, 1), dim3(256), 0, 0, d_bv, d_rp, sample, d_oov, d_gbits, d_gones, d_off,
                       d_so, d_ps);
    WT_TRY(hipGetLastError());
    if (!m.classes.words.empty())
        WT_TRY(hipMemcpy(m.classes.words.data(), d_cls, m.classes.words.size() * 8, hipMemcpyDeviceToHost));
    if (!m.offsets.empty()) WT_TRY(hipMemcpy(m.offsets.data(), d_off, m.offsets.size() * 8, hipMemcpyDeviceToHost));
    if (!m.sampled_offsets.words.empty())
        WT_TRY(hipMemcpy(m.sampled_offsets.words.data(), d_so, m.sampled_offsets.words.size() * 8, hipMemcpyDeviceToHost));
    if (!m.prefix_sums.words.empty())
        WT_TRY(hipMemcpy(m.prefix_sums.words.data(), d_ps, m.prefix_sums.words.size() * 8, hipMemcpyDeviceToHost));
    const int64_t n_sampled = r.n_blocks == 0 ? 0 : (int64_t)(r.n_blocks - 1) / sample + 1;  // RRR:285
    m.prefix_sums.set(n_sampled, (uint64_t)(uint32_t)m.total_ones);
    return 0;
}

}  // namespace fmx
