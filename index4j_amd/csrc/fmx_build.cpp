// fmx_build.cpp — host-side construction of an index4j-identical FM-index (product code, not the
// oracle): alphabet remap, SA-IS suffix array, sampling, BWT, and the fixed-block-boosting wavelet
// tree with one RRR vector per superblock, encoded superblock-parallel on the host cores.
//
// The structures produced are the ones `new FmIndex(char[], sampleRate, enableExtract)` builds
// (FM:155-174), so that fmx_save emits what FmIndex.write would and a JVM can load it.  Algorithms
// are re-designed for speed (sparse per-block symbol lists, O(s log s) Huffman with the reference's
// tie order, single-pass node bitvector emission, word-level RRR packing, one thread per superblock);
// tests/test_builder_parity.py checks the serialized bytes against the oracle's.
//
// Citations as in fmx_model.hpp.
#include "fmx_model.hpp"
#include "fmx_build_stage.hpp"
#include "fmx_sais.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace fmx {

// ---------------------------------------------------------------------------------------------
// RRR tables (RRR:104-129; literal tables RRR:488-16900 are generated, not copied: within class k
// the offset of a 15-bit value is the lexicographic rank of its set-bit positions among the
// k-subsets of {0..14}; tools/check_rrr_tables.py verified this against the literals)
// ---------------------------------------------------------------------------------------------
namespace {
struct RrrTables {
    uint16_t offset_of_value[32768];
    uint16_t value_of_offset[32768];
    uint16_t class_base[16];
    uint8_t bits_needed[16];
    RrrTables() {
        int binom[16][16];
        for (int n = 0; n < 16; ++n) {
            binom[n][0] = 1;
            for (int k = 1; k < 16; ++k) binom[n][k] = n == 0 ? 0 : binom[n - 1][k - 1] + binom[n - 1][k];
        }
        int base = 0;
        for (int k = 0; k < 16; ++k) {
            class_base[k] = (uint16_t)base;
            base += binom[15][k];
            bits_needed[k] = (uint8_t)min_bits((uint64_t)binom[15][k]);
        }
        // enumerate k-subsets in lexicographic order with the "next combination" step
        for (int k = 0; k < 16; ++k) {
            int pos[16];
            for (int i = 0; i < k; ++i) pos[i] = i;
            int idx = 0;
            while (true) {
                unsigned v = 0;
                for (int i = 0; i < k; ++i) v |= 1u << pos[i];
                offset_of_value[v] = (uint16_t)idx;
                value_of_offset[class_base[k] + idx] = (uint16_t)v;
                ++idx;
                int i = k - 1;
                while (i >= 0 && pos[i] == 15 - k + i) --i;
                if (i < 0) break;
                ++pos[i];
                for (int j = i + 1; j < k; ++j) pos[j] = pos[j - 1] + 1;
            }
        }
    }
};
const RrrTables &tables() {
    static const RrrTables t;
    return t;
}
}  // namespace

const uint16_t *rrr_offset_of_value() { return tables().offset_of_value; }
const uint16_t *rrr_value_of_offset() { return tables().value_of_offset; }
const uint16_t *rrr_class_base() { return tables().class_base; }
const uint8_t *rrr_bits_needed() { return tables().bits_needed; }

// ---------------------------------------------------------------------------------------------
// RRR construction from an LSB-first bit array (RRR:225-286), two passes over 15-bit blocks
// ---------------------------------------------------------------------------------------------
static inline unsigned bits15(const uint64_t *w, int64_t from, int64_t nbits) {
    int64_t len = std::min<int64_t>(15, nbits - from);
    size_t wi = (size_t)(from >> 6);
    int off = (int)(from & 63);
    uint64_t v = w[wi] >> off;
    if (off + len > 64) v |= w[wi + 1] << (64 - off);
    return (unsigned)(v & low_bits((int)len));
}

// atomically OR `nbits` of `value` into a word array shared by several writer threads (neighbouring chunks meet
// inside a word)
static inline void put_bits_shared(uint64_t *words, int64_t bit, uint64_t value, int nbits) {
    if (nbits <= 0) return;
    value &= low_bits(nbits);
    const size_t w = (size_t)(bit >> 6);
    const int off = (int)(bit & 63);
    __atomic_fetch_or(&words[w], value << off, __ATOMIC_RELAXED);
    if (off + nbits > 64) __atomic_fetch_or(&words[w + 1], value >> (64 - off), __ATOMIC_RELAXED);
}

// RRR:225-286.  `threads` > 1: the blocks are cut into chunks (multiples of 16 * sample_size blocks); one pass
// counts every chunk's offset bits and ones, a prefix over the chunks gives each its start, a second pass writes.
void build_rrr(const uint64_t *bits, int64_t nbits, int sample_size, RrrModel &r, int threads) {
    const RrrTables &T = tables();
    r.sample_size = sample_size;
    r.length = (int32_t)nbits;
    const int64_t num_blocks = nbits / 15 + ((nbits % 15 > 0) ? 1 : 0);  // RRR:232
    r.classes.init((int32_t)num_blocks, 4);
    int64_t chunk = (num_blocks + threads - 1) / (threads > 0 ? threads : 1);
    const int64_t grain = 16 * (int64_t)sample_size;  // chunk starts: whole class words and whole samples
    chunk = (chunk + grain - 1) / grain * grain;
    if (chunk <= 0) chunk = grain;
    const int64_t n_chunks = num_blocks == 0 ? 0 : (num_blocks + chunk - 1) / chunk;
    std::vector<int64_t> chunk_bits((size_t)n_chunks + 1, 0), chunk_ones((size_t)n_chunks + 1, 0);
    auto run = [&](const std::function<void(int64_t)> &fn) {
        if (threads <= 1 || n_chunks <= 1) {
            for (int64_t c = 0; c < n_chunks; ++c) fn(c);
            return;
        }
        std::atomic<int64_t> next{0};
        auto worker = [&]() {
            for (;;) {
                const int64_t c = next.fetch_add(1);
                if (c >= n_chunks) return;
                fn(c);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads && t < n_chunks; ++t) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
    };
    run([&](int64_t c) {  // classes (chunks start on class-word boundaries) and per-chunk totals
        const int64_t lo = c * chunk, hi = std::min(num_blocks, lo + chunk);
        int64_t ob = 0, ones = 0;
        for (int64_t b = lo; b < hi; ++b) {
            const unsigned v = bits15(bits, b * 15, nbits);
            const int k = __builtin_popcount(v);
            r.classes.set(b, (uint64_t)k);
            ob += T.bits_needed[k];
            ones += k;
        }
        chunk_bits[(size_t)c + 1] = ob;
        chunk_ones[(size_t)c + 1] = ones;
    });
    for (int64_t c = 0; c < n_chunks; ++c) {
        chunk_bits[(size_t)c + 1] += chunk_bits[(size_t)c];
        chunk_ones[(size_t)c + 1] += chunk_ones[(size_t)c];
    }
    const int64_t total_offset_bits = chunk_bits[(size_t)n_chunks], total_ones = chunk_ones[(size_t)n_chunks];
    r.total_ones = (int32_t)total_ones;
    r.bits_per_offset_pos = min_bits((uint64_t)total_offset_bits);                         // RRR:262
    r.sampled_offsets.init((int32_t)(num_blocks / sample_size + 1), r.bits_per_offset_pos);  // RRR:263
    r.prefix_sums.init((int32_t)(num_blocks / sample_size + 2), min_bits((uint64_t)total_ones));  // RRR:264
    std::vector<uint64_t> off((size_t)words_for_bits(total_offset_bits) + 1, 0);  // VIV:41-47 (+1 scratch word, trimmed below)
    run([&](int64_t c) {
        const int64_t lo = c * chunk, hi = std::min(num_blocks, lo + chunk);
        int64_t cur_bits = chunk_bits[(size_t)c], prefix = chunk_ones[(size_t)c];
        for (int64_t b = lo; b < hi; ++b) {
            const unsigned v = bits15(bits, b * 15, nbits);
            const int k = __builtin_popcount(v);
            const int nb = T.bits_needed[k];
            put_bits_shared(off.data(), cur_bits, T.offset_of_value[v], nb);
            if (b % sample_size == 0) {
                const int64_t sampled = b / sample_size;
                put_bits_shared(r.sampled_offsets.words.data(), sampled * r.sampled_offsets.width, (uint64_t)cur_bits,
                                r.sampled_offsets.width);
                put_bits_shared(r.prefix_sums.words.data(), sampled * r.prefix_sums.width, (uint64_t)prefix,
                                r.prefix_sums.width);
            }
            cur_bits += nb;
            prefix += k;
        }
    });
    // RRR:285: one more prefix sum after the last sampled block
    const int64_t n_sampled = num_blocks == 0 ? 0 : (num_blocks - 1) / sample_size + 1;
    r.prefix_sums.set(n_sampled, (uint64_t)total_ones);
    off.resize((size_t)words_for_bits(total_offset_bits));
    r.offsets.swap(off);
}

// all-zero RRR of n bits: getEstimatedMemoryUsage() (WFBB:961-965 -> RRR:418-423), closed form
static int64_t rrr_all_zero_estimate(int64_t n, int sample) {
    const int64_t num_blocks = n / 15 + ((n % 15 > 0) ? 1 : 0);
    const int64_t total_bits = num_blocks * tables().bits_needed[0];
    int64_t words = words_for_bits(num_blocks * 4) + words_for_bits(total_bits) +
                    words_for_bits((num_blocks / sample + 1) * (int64_t)min_bits((uint64_t)total_bits)) +
                    words_for_bits((num_blocks / sample + 2) * (int64_t)min_bits(0));
    return (int64_t)(int32_t)(words * 8);  // the Java method returns int
}

// ---------------------------------------------------------------------------------------------
// Huffman code lengths with the reference's tie order (WFBB:334-360 + comparator WFBB:1684-1707)
// ---------------------------------------------------------------------------------------------
// Queue items are (frequency, symbol list); ties compare the lists element-wise.  The lists are
// disjoint, so the first elements already differ, and merging x then y keeps x's first element:
// the order is (frequency, first symbol of the list).
struct SymFreq {
    int16_t sym;
    int32_t freq;
};

struct HuffScratch {
    struct Node {
        int64_t freq;
        int32_t first;
        int32_t parent;
    };
    std::vector<Node> nodes;
    std::vector<int32_t> heap;
    std::vector<int32_t> depth;
};

// lens[i] = code length of syms[i]; returns max code length (0 for a single symbol)
static int huffman_lengths(const SymFreq *syms, int n, int *lens, HuffScratch &S) {
    if (n <= 1) {
        if (n == 1) lens[0] = 0;
        return 0;
    }
    S.nodes.resize((size_t)(2 * n - 1));
    S.heap.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        S.nodes[i] = {syms[i].freq, syms[i].sym, -1};
        S.heap[i] = i;
    }
    auto less = [&](int a, int b) {  // strict "a after b" for a min-heap built with std::*_heap
        const auto &x = S.nodes[a], &y = S.nodes[b];
        return x.freq != y.freq ? x.freq > y.freq : x.first > y.first;
    };
    std::make_heap(S.heap.begin(), S.heap.end(), less);
    int next = n;
    int hs = n;
    while (hs > 1) {
        std::pop_heap(S.heap.begin(), S.heap.begin() + hs, less);
        int x = S.heap[--hs];
        std::pop_heap(S.heap.begin(), S.heap.begin() + hs, less);
        int y = S.heap[--hs];
        S.nodes[next] = {S.nodes[x].freq + S.nodes[y].freq, S.nodes[x].first, -1};
        S.nodes[x].parent = next;
        S.nodes[y].parent = next;
        S.heap[hs++] = next;
        std::push_heap(S.heap.begin(), S.heap.begin() + hs, less);
        ++next;
    }
    S.depth.assign((size_t)next, 0);
    int mx = 0;
    for (int i = next - 2; i >= 0; --i) S.depth[i] = S.depth[S.nodes[i].parent] + 1;
    for (int i = 0; i < n; ++i) {
        lens[i] = S.depth[i];
        mx = std::max(mx, lens[i]);
    }
    return mx;
}

// ---------------------------------------------------------------------------------------------
// wavelet tree: per-superblock encoding (WFBB:362-535, 570-810, 812-991)
// ---------------------------------------------------------------------------------------------
static const int SBS_LOG = 20;  // WFBB:93,97
static const int64_t SBS = 1LL << SBS_LOG;
static const int64_t HBS = 1LL << 32;       // WFBB:99
static const int BLOCK_HEADER_ITEM_SIZE = 14;  // WFBB:95 (estimator constant)

struct SbScratch {
    std::vector<int32_t> dense;                // alphabet-sized counter scratch
    std::vector<std::vector<SymFreq>> lists;   // per-block sorted (sym, freq) lists
    std::vector<std::vector<SymFreq>> lists2;
    std::vector<int> lens;
    HuffScratch huff;
    std::vector<uint64_t> bv;                  // superblock bitvector under construction
};

static void block_symbol_list(const int16_t *text, int64_t len, std::vector<int32_t> &dense,
                              std::vector<SymFreq> &out) {
    out.clear();
    for (int64_t i = 0; i < len; ++i) {
        int16_t c = text[i];
        if (dense[c]++ == 0) out.push_back({c, 0});
    }
    std::sort(out.begin(), out.end(), [](const SymFreq &a, const SymFreq &b) { return a.sym < b.sym; });
    for (auto &e : out) {
        e.freq = dense[e.sym];
        dense[e.sym] = 0;
    }
}

static void merge_lists(const std::vector<SymFreq> &a, const std::vector<SymFreq> *b, std::vector<SymFreq> &out) {
    out.clear();
    if (!b) {
        out = a;
        return;
    }
    size_t i = 0, j = 0;
    while (i < a.size() || j < b->size()) {
        if (j >= b->size() || (i < a.size() && a[i].sym < (*b)[j].sym))
            out.push_back(a[i++]);
        else if (i >= a.size() || (*b)[j].sym < a[i].sym)
            out.push_back((*b)[j++]);
        else {
            out.push_back({a[i].sym, a[i].freq + (*b)[j].freq});
            ++i;
            ++j;
        }
    }
}

// WFBB:853-987 given, per candidate block size 2^9 .. 2^16, the sums over the superblock's blocks of the header
// terms (WFBB:924-947) and of frequency * code length: the size estimate with its double-precision chain, and the
// choice.  Shared by the host encoder (below) and the device encoder (fmx_wt_gpu.hip), which produce the sums.
int pick_block_size_log(const int64_t *hdr_sum, const int64_t *unc_sum, int64_t sb_size, int64_t sb_sigma, int sampling_rate) {
    const int smallest_log = std::max(0, std::min(SBS_LOG, 16) - 7);
    const int top_log = std::min(SBS_LOG, 16);
    int best_log = 0;
    int64_t best_size = 0;
    int64_t compressed = 0, prev_uncompressed = 0;
    for (int bsl = smallest_log; bsl <= top_log; ++bsl) {
        const int64_t block_size = 1LL << bsl;
        const int64_t n_blocks = (sb_size + block_size - 1) / block_size;
        int64_t enc = (int64_t)BLOCK_HEADER_ITEM_SIZE * n_blocks + sb_sigma * (SBS / block_size) + hdr_sum[bsl - smallest_log];
        const int64_t uncompressed = unc_sum[bsl - smallest_log];
        if (uncompressed > 0) {
            if (bsl == smallest_log) {
                compressed = rrr_all_zero_estimate(uncompressed, sampling_rate);
            } else {
                // (long)((double)compressed * ((double)u / (double)prev)) with Java's saturating cast
                double prod = (double)compressed * ((double)uncompressed / (double)prev_uncompressed);
                if (std::isnan(prod))
                    compressed = 0;
                else if (prod >= 9223372036854775807.0)
                    compressed = INT64_MAX;
                else if (prod <= -9223372036854775808.0)
                    compressed = INT64_MIN;
                else
                    compressed = (int64_t)prod;
            }
            enc += compressed;
        }
        prev_uncompressed = uncompressed;
        if (bsl == smallest_log || enc < best_size) {
            best_log = bsl;
            best_size = enc;
        }
    }
    return best_log;
}

// WFBB:853-987: choose blockSizeLog in 9..16 by estimated encoding size
static int choose_block_size_log(const int16_t *text, int64_t sb_size, int64_t sb_sigma, int alphabet,
                                 int sampling_rate, SbScratch &S) {
    const int smallest_log = std::max(0, std::min(SBS_LOG, 16) - 7);
    const int top_log = std::min(SBS_LOG, 16);
    int64_t hdr_sum[8] = {0}, unc_sum[8] = {0};
    std::vector<std::vector<SymFreq>> *cur = &S.lists, *nxt = &S.lists2;
    for (int bsl = smallest_log; bsl <= top_log; ++bsl) {
        const int64_t block_size = 1LL << bsl;
        const int64_t n_blocks = (sb_size + block_size - 1) / block_size;
        if (bsl == smallest_log) {
            cur->resize((size_t)n_blocks);
            for (int64_t b = 0; b < n_blocks; ++b) {
                int64_t beg = b * block_size, end = std::min(beg + block_size, sb_size);
                block_symbol_list(text + beg, end - beg, S.dense, (*cur)[(size_t)b]);
            }
        } else {
            const int64_t prev_blocks = (sb_size + (block_size / 2) - 1) / (block_size / 2);
            nxt->resize((size_t)n_blocks);
            for (int64_t b = 0; b < prev_blocks; b += 2)
                merge_lists((*cur)[(size_t)b], (b + 1 < prev_blocks) ? &(*cur)[(size_t)b + 1] : nullptr,
                            (*nxt)[(size_t)(b >> 1)]);
            std::swap(cur, nxt);
        }
        int64_t uncompressed = 0, headers = 0;
        for (int64_t b = 0; b < n_blocks; ++b) {
            const auto &L = (*cur)[(size_t)b];
            const int64_t block_sigma = (int64_t)L.size();
            headers += block_sigma * 4;          // WFBB:924
            headers += (block_sigma - 1) * 2;    // WFBB:925
            S.lens.resize(L.size());
            int mcl = huffman_lengths(L.data(), (int)L.size(), S.lens.data(), S.huff);
            if (L.empty()) mcl = -1;
            if (mcl > 1) headers += (int64_t)(mcl - 1) * 3;  // WFBB:945-947
            for (size_t i = 0; i < L.size(); ++i) uncompressed += (int64_t)L[i].freq * S.lens[i];
        }
        hdr_sum[bsl - smallest_log] = headers;
        unc_sum[bsl - smallest_log] = uncompressed;
    }
    (void)alphabet;
    return pick_block_size_log(hdr_sum, unc_sum, sb_size, sb_sigma, sampling_rate);
}

static inline void wr16(uint8_t *p, unsigned v) {
    p[0] = (uint8_t)(v & 0xff);
    p[1] = (uint8_t)((v >> 8) & 0xff);
}

// one superblock: WFBB:812-851 (given the running counts), then 362-535 / 570-810 at the chosen size
static void encode_superblock(const int16_t *bwt, WfbbModel &w, int64_t sb_id, const int64_t *count_before,
                              const int64_t *count_after, SbScratch &S) {
    const int sigma_g = w.alphabet_size;
    const int64_t sb_beg = sb_id * SBS;
    const int64_t sb_end = std::min(sb_beg + SBS, w.size);
    const int64_t sb_size = sb_end - sb_beg;
    const int16_t *text = bwt + sb_beg;
    SuperBlockModel &sb = w.sb[(size_t)sb_id];
    const int64_t hb_id = (sb_id * SBS) / HBS;

    for (int i = 0; i < sigma_g; ++i)
        w.super_rank[(size_t)(sb_id * sigma_g + i)] =
            (int32_t)(count_before[i] - w.hyper_rank[(size_t)(hb_id * sigma_g + i)]);
    int64_t sb_sigma = 0;
    int16_t *gmap = &w.global_mapping[(size_t)(sb_id * sigma_g)];
    for (int i = 0; i < sigma_g; ++i)
        if (count_before[i] != count_after[i]) gmap[i] = (int16_t)sb_sigma++;
    sb.sigma = (int16_t)(sb_sigma - 1);

    if (S.dense.size() < (size_t)sigma_g) S.dense.assign((size_t)sigma_g, 0);
    const int bsl = choose_block_size_log(text, sb_size, sb_sigma, sigma_g, w.sampling_rate, S);
    sb.block_size_log = (int16_t)bsl;
    const int64_t block_size = 1LL << bsl;
    const int64_t blocks_per_sb = SBS / block_size;
    const int64_t n_blocks = (sb_size + block_size - 1) / block_size;

    sb.mapping.assign((size_t)(sb_sigma * blocks_per_sb), (int16_t)(sigma_g - 1));  // WFBB:377-387
    sb.block_headers.assign((size_t)n_blocks, BlockHeader{0, 0, 0, 0, 0});

    // pass 1 (WFBB:400-484): per-block code lengths -> header offsets, bitvector offsets, mapping
    struct BlockPlan {
        std::vector<SymFreq> syms;   // sorted by (code length, symbol)
        std::vector<int> lens;
        int tree_height;
    };
    std::vector<BlockPlan> plan((size_t)n_blocks);
    int64_t sb_bv_size = 0, var_size = 0;
    std::vector<SymFreq> list;
    std::vector<int> order;
    for (int64_t b = 0; b < n_blocks; ++b) {
        const int64_t beg = b * block_size, end = std::min(beg + block_size, sb_size);
        block_symbol_list(text + beg, end - beg, S.dense, list);
        S.lens.resize(list.size());
        const int th = huffman_lengths(list.data(), (int)list.size(), S.lens.data(), S.huff);
        order.resize(list.size());
        for (size_t i = 0; i < list.size(); ++i) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int a, int c) {  // WFBB:1709-1718: (codeLength, symbol)
            return S.lens[a] != S.lens[c] ? S.lens[a] < S.lens[c] : list[a].sym < list[c].sym;
        });
        BlockPlan &P = plan[(size_t)b];
        P.tree_height = th;
        P.syms.resize(list.size());
        P.lens.resize(list.size());
        int64_t bv_size = 0;
        const int64_t sigma = (int64_t)list.size();
        for (size_t i = 0; i < list.size(); ++i) {
            P.syms[i] = list[order[i]];
            P.lens[i] = S.lens[order[i]];
            if (sigma > 1) bv_size += (int64_t)P.syms[i].freq * P.lens[i];
            const int16_t sb_char = gmap[P.syms[i].sym];
            const int64_t address = (int64_t)sb_char * blocks_per_sb + b;
            sb.mapping[(size_t)address] = (int16_t)std::min<int>(sigma_g - 2, (int)i);  // WFBB:466-471
        }
        BlockHeader &bh = sb.block_headers[(size_t)b];
        bh.bv_offset = (int32_t)sb_bv_size;
        bh.var_off = (int32_t)var_size;
        bh.tree_height = (int16_t)th;
        bh.sigma = (int16_t)(sigma - 1);
        sb_bv_size += bv_size;
        if (th > 1) var_size += (int64_t)(th - 1) * 4;  // WFBB:479-483
        var_size += sigma * 5;
        var_size += (sigma - 1) * 2;
    }

    sb.var.assign((size_t)var_size, 0);
    S.bv.assign((size_t)(sb_bv_size / 64 + 2), 0);
    std::vector<int64_t> block_rank((size_t)sigma_g, 0);  // occurrences before the block, per symbol
    int64_t bv_rank = 0;

    // pass 2 (WFBB:499-531 + 570-810): node bitvectors and variable-size headers
    std::vector<uint32_t> code_of((size_t)sigma_g), len_of((size_t)sigma_g);
    std::vector<int64_t> node_size, node_off, node_fill, node_ones;
    std::vector<int32_t> node_index;  // node id -> bitvector index (BFS order), -1 for leaves / absent
    std::vector<int32_t> ids;
    for (int64_t b = 0; b < n_blocks; ++b) {
        const int64_t beg = b * block_size, end = std::min(beg + block_size, sb_size);
        const BlockPlan &P = plan[(size_t)b];
        BlockHeader &bh = sb.block_headers[(size_t)b];
        const int th = P.tree_height;
        const int sigma = (int)P.syms.size();
        uint8_t *hdr = sb.var.data() + bh.var_off;
        int64_t ones_count = 0;

        // canonical codes in (length, symbol) order, WFBB:549-554
        uint32_t c = 0;
        for (int i = 0; i < sigma; ++i) {
            if (i != 0) c = (c + 1) << (P.lens[i] - P.lens[i - 1]);
            code_of[(size_t)P.syms[i].sym] = c;
            len_of[(size_t)P.syms[i].sym] = (uint32_t)P.lens[i];
        }
        // level tables, WFBB:713-760
        std::vector<int64_t> clf((size_t)th + 1, 0), ltf((size_t)th + 1, 0);
        for (int i = 0; i < sigma; ++i) {
            if (P.lens[i] < th) clf[(size_t)P.lens[i]] += 1;
            for (int d = 1; d < P.lens[i]; ++d) ltf[(size_t)d] += P.syms[i].freq;
        }
        int64_t bp = 0;
        for (int d = 1; d < th; ++d) {
            wr16(hdr + bp, (unsigned)(clf[(size_t)d] & 0xffff));
            wr16(hdr + bp + 2, (unsigned)((ltf[(size_t)d] - 1) & 0xffff));
            bp += 4;
        }
        // leaves, WFBB:774-788
        for (int i = 0; i < sigma; ++i) {
            const int16_t symbol = P.syms[i].sym;
            const int64_t rv = block_rank[(size_t)symbol];
            wr16(hdr + bp, (unsigned)(uint16_t)symbol);
            hdr[bp + 2] = (uint8_t)(rv & 0xff);
            hdr[bp + 3] = (uint8_t)((rv >> 8) & 0xff);
            hdr[bp + 4] = (uint8_t)((rv >> 16) & 0xff);
            bp += 5;
        }

        if (sigma > 1) {
            // internal nodes: id = (1 << depth) | code prefix, BFS order = ascending id (WFBB:606-637)
            const size_t id_space = (size_t)1 << th;
            node_index.assign(id_space, -1);
            ids.clear();
            for (int i = 0; i < sigma; ++i) {
                const uint32_t len = (uint32_t)P.lens[i], code = code_of[(size_t)P.syms[i].sym];
                for (uint32_t d = 0; d < len; ++d) {
                    const uint32_t id = ((1u << len) | code) >> (len - d);
                    if (node_index[id] < 0) {
                        node_index[id] = 0;
                        ids.push_back((int32_t)id);
                    }
                }
            }
            std::sort(ids.begin(), ids.end());
            const size_t n_nodes = ids.size();
            node_size.assign(n_nodes, 0);
            node_fill.assign(n_nodes, 0);
            node_ones.assign(n_nodes, 0);
            node_off.assign(n_nodes, 0);
            for (size_t i = 0; i < n_nodes; ++i) node_index[(size_t)ids[i]] = (int32_t)i;
            for (int i = 0; i < sigma; ++i) {
                const uint32_t len = (uint32_t)P.lens[i], code = code_of[(size_t)P.syms[i].sym];
                for (uint32_t d = 0; d < len; ++d)
                    node_size[(size_t)node_index[((1u << len) | code) >> (len - d)]] += P.syms[i].freq;
            }
            int64_t acc = bh.bv_offset;
            for (size_t i = 0; i < n_nodes; ++i) {
                node_off[i] = acc;
                acc += node_size[i];
            }
            // every symbol appends one bit to each internal node on its path; its position inside
            // the node is the number of earlier symbols routed through that node (== WFBB:674-700)
            for (int64_t i = beg; i < end; ++i) {
                const int16_t sym = text[i];
                const uint32_t len = len_of[(size_t)sym], code = code_of[(size_t)sym];
                for (uint32_t d = 0; d < len; ++d) {
                    const size_t ni = (size_t)node_index[((1u << len) | code) >> (len - d)];
                    const int64_t pos = node_off[ni] + node_fill[ni]++;
                    if (code & (1u << (len - d - 1))) {
                        S.bv[(size_t)(pos >> 6)] |= 1ULL << (pos & 63);
                        ++node_ones[ni];
                        ++ones_count;
                    }
                }
            }
            // cumulative one-counts per level, WFBB:793-809 (u16 wrap at WFBB:798)
            int64_t n_internal = 1;
            size_t ptr = 0;
            for (int d = 0; d < th; ++d) {
                int64_t level_ones = 0;
                for (int64_t j = 0; j < n_internal; ++j) {
                    level_ones += node_ones[ptr++];
                    wr16(hdr + bp, (unsigned)(level_ones & 0xffff));
                    bp += 2;
                }
                if (d + 1 != th) {
                    n_internal <<= 1;
                    n_internal -= clf[(size_t)d + 1];
                }
            }
        }
        bh.bv_rank = (int32_t)bv_rank;
        bv_rank += ones_count;
        for (int i = 0; i < sigma; ++i) block_rank[(size_t)P.syms[i].sym] += P.syms[i].freq;
    }
    build_rrr(S.bv.data(), sb_bv_size, w.sampling_rate, sb.rank_support);  // WFBB:534
}

// WFBB:130-154
void build_wavelet(const int16_t *bwt, int64_t n, int sampling_rate, WfbbModel &w) {
    w.size = n;
    w.sampling_rate = sampling_rate;
    int mx = INT32_MIN;
    for (int64_t i = 0; i < n; ++i) mx = std::max<int>(mx, bwt[i]);
    w.alphabet_size = mx + 1;
    const int sigma = w.alphabet_size;
    const int64_t n_sb = (n + SBS - 1) / SBS;
    const int64_t n_hb = (n + HBS - 1) / HBS;
    w.count.assign((size_t)sigma, 0);
    w.hyper_rank.assign((size_t)(n_hb * sigma), 0);
    w.super_rank.assign((size_t)(n_sb * sigma), 0);
    w.global_mapping.assign((size_t)(n_sb * sigma), (int16_t)(sigma - 1));
    w.sb.assign((size_t)n_sb, SuperBlockModel());

    // running symbol counts at every superblock boundary (replaces the serial count[] of WFBB:833-837)
    std::vector<int64_t> counts((size_t)((n_sb + 1) * sigma), 0);
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    {
        std::atomic<int64_t> next{0};
        auto work = [&]() {
            for (;;) {
                const int64_t s = next.fetch_add(1);
                if (s >= n_sb) break;
                int64_t *row = &counts[(size_t)((s + 1) * sigma)];
                const int64_t beg = s * SBS, end = std::min(beg + SBS, n);
                for (int64_t i = beg; i < end; ++i) ++row[bwt[i]];
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 0; t < std::min<unsigned>(hw, (unsigned)n_sb); ++t) th.emplace_back(work);
        for (auto &t : th) t.join();
    }
    for (int64_t s = 0; s < n_sb; ++s)
        for (int i = 0; i < sigma; ++i) counts[(size_t)((s + 1) * sigma + i)] += counts[(size_t)(s * sigma + i)];
    for (int64_t s = 0; s < n_sb; ++s)
        if ((s * SBS) % HBS == 0) {  // WFBB:819-825
            const int64_t hb = (s * SBS) / HBS;
            for (int i = 0; i < sigma; ++i) w.hyper_rank[(size_t)(hb * sigma + i)] = counts[(size_t)(s * sigma + i)];
        }
    for (int i = 0; i < sigma; ++i) w.count[(size_t)i] = counts[(size_t)(n_sb * sigma + i)];

    std::atomic<int64_t> next{0};
    auto work = [&]() {
        SbScratch S;
        for (;;) {
            const int64_t s = next.fetch_add(1);
            if (s >= n_sb) break;
            encode_superblock(bwt, w, s, &counts[(size_t)(s * sigma)], &counts[(size_t)((s + 1) * sigma)], S);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < std::min<unsigned>(hw, (unsigned)n_sb); ++t) th.emplace_back(work);
    for (auto &t : th) t.join();
}

// ---------------------------------------------------------------------------------------------
// FmIndex constructor (FM:155-174)
// ---------------------------------------------------------------------------------------------
int host_sa_stage(const int16_t *seq, int32_t n, int alphabet, int sample_rate, bool extract, SaStage &out) {
    std::vector<int32_t> sa((size_t)n);
    suffix_array(seq, n, alphabet, sa.data());
    out.which.assign((size_t)(n / 64 + 2), 0);
    out.suffix_vals.clear();
    out.suffix_vals.reserve((size_t)n / (size_t)sample_rate + 2);
    out.position_vals.clear();
    if (extract) out.position_vals.assign((size_t)n / (size_t)sample_rate + 2, 0);
    for (int32_t i = 0; i < n; ++i)  // FM:341-366
        if (sa[(size_t)i] % sample_rate == 0) {
            out.suffix_vals.push_back((uint32_t)sa[(size_t)i]);
            out.which[(size_t)(i >> 6)] |= 1ULL << (i & 63);
            if (extract) out.position_vals[(size_t)(sa[(size_t)i] / sample_rate)] = (uint32_t)i;
        }
    out.bwt.resize((size_t)n);  // FM:385-392
    for (int32_t i = 0; i < n; ++i) out.bwt[(size_t)i] = sa[(size_t)i] == 0 ? seq[(size_t)n - 1] : seq[(size_t)sa[(size_t)i] - 1];
    return 0;
}

// FMX_BUILD_TIMING=1 prints the wall time of the constructor's phases to stderr
struct PhaseTimer {
    const bool on = getenv("FMX_BUILD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fmx build] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

int build_model(const uint16_t *input, int32_t n_in, int32_t sample_rate, bool enable_extract, FmModel &m,
                std::string &err, int build_device, SaStageStats *stats, bool device_wavelet) {
    PhaseTimer timer;
    if (n_in < 0 || sample_rate <= 0 || n_in == INT32_MAX) {
        err = "bad arguments";
        return -1;
    }
    m = FmModel();
    m.sample_rate = sample_rate;
    m.enable_extract = enable_extract;
    const int32_t n = n_in + 1;  // FM:300-305: terminating '\0' appended
    m.length = n;

    // FM:396-435: codes in order of first appearance; the appended sentinel is code 0; an embedded
    // '\0' gets code 1.  One parallel pass collects, per character, its first position and its count; the codes
    // follow from the first positions, the mapped text and cumulativeCounts (FM:307-327) from a second pass.
    unsigned n_threads = std::thread::hardware_concurrency();
    if (n_threads == 0) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (n_in < (1 << 20)) n_threads = 1;
    auto for_chunks = [&](const std::function<void(unsigned, int32_t, int32_t)> &fn) {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t) {
            const int32_t lo = (int32_t)((int64_t)n_in * t / n_threads), hi = (int32_t)((int64_t)n_in * (t + 1) / n_threads);
            if (t + 1 == n_threads)
                fn(t, lo, hi);
            else
                pool.emplace_back(fn, t, lo, hi);
        }
        for (auto &th : pool) th.join();
    };
    std::vector<int32_t> first(65536, -1);
    std::vector<int64_t> raw_count(65536, 0);
    void *d_text = nullptr;  // the text in HBM, when the device stages take it from there
    if (build_device >= 0 && device_wavelet && device_alphabet_stage && n_in >= (1 << 16)) {
        const int rc = device_alphabet_stage(input, n_in, build_device, first, raw_count, &d_text, err);
        if (rc) return rc;
    } else {
        std::vector<std::vector<int32_t>> first_seen(n_threads), seen_count(n_threads);
        for_chunks([&](unsigned t, int32_t lo, int32_t hi) {
            std::vector<int32_t> &fs = first_seen[t], &count = seen_count[t];
            fs.assign(65536, -1);
            count.assign(65536, 0);
            for (int32_t i = lo; i < hi; ++i) {
                const uint16_t ch = input[i];
                if (fs[ch] < 0) fs[ch] = i;
                ++count[ch];
            }
        });
        for (unsigned t = 0; t < n_threads; ++t)  // chunks are in text order: the first chunk that saw it wins
            for (int ch = 0; ch < 65536; ++ch) {
                if (first[(size_t)ch] < 0 && first_seen[t][(size_t)ch] >= 0) first[(size_t)ch] = first_seen[t][(size_t)ch];
                raw_count[(size_t)ch] += seen_count[t][(size_t)ch];
            }
    }
    struct DeviceTextGuard {  // device_sa_stage takes the buffer over; freed here only if an early return skips that stage
        void *&p;
        ~DeviceTextGuard() {
            if (p && device_release) device_release(p);
        }
    } text_guard{d_text};
    std::vector<int32_t> code_of(65536, -1);
    const int64_t zeros = 1 + raw_count[0];
    int mapped = (zeros != 1) ? 1 : 0;
    m.map_keys.push_back(0);
    m.map_vals.push_back((int16_t)mapped);
    code_of[0] = mapped;
    ++mapped;
    std::vector<std::pair<int32_t, int32_t>> order;  // (first position, character)
    for (int ch = 1; ch < 65536; ++ch)
        if (first[(size_t)ch] >= 0) order.emplace_back(first[(size_t)ch], ch);
    std::sort(order.begin(), order.end());
    if (order.size() + 1 > 32767) {  // FM:423-426
        err = "Input has more than 32767 different symbols";
        return -2;
    }
    for (const auto &fc : order) {
        code_of[(size_t)fc.second] = mapped;
        m.map_keys.push_back(fc.second);
        m.map_vals.push_back((int16_t)mapped);
        ++mapped;
    }
    const int distinct = (int)m.map_keys.size();  // == alphabet.size() of FM:397-404
    m.look_up.assign((size_t)distinct + 1, 0);    // FM:411
    for (size_t i = 0; i < m.map_keys.size(); ++i) m.look_up[(size_t)m.map_vals[i]] = m.map_keys[i];

    std::vector<int16_t> seq;
    if (!d_text) {  // (else the device maps its own copy of the text)
        seq.resize((size_t)n);
        for_chunks([&](unsigned, int32_t lo, int32_t hi) {
            for (int32_t i = lo; i < hi; ++i) seq[(size_t)i] = (int16_t)code_of[input[i]];
        });
        seq[(size_t)n - 1] = 0;  // FM:433
    }

    timer.mark("alphabet + mapped text");
    // FM:307-327
    const int n_look = (int)m.look_up.size();
    std::vector<int32_t> cc(65536, 0);
    for (int ch = 0; ch < 65536; ++ch)
        if (raw_count[(size_t)ch]) cc[(size_t)code_of[(size_t)ch]] += (int32_t)raw_count[(size_t)ch];
    ++cc[0];  // the terminator (code 0 at n-1 only)
    int32_t off = cc[0];
    cc[0] = 0;
    for (int i = 1; i < n_look; ++i) {
        const int32_t prev = cc[(size_t)i];
        cc[(size_t)i] = cc[(size_t)i - 1] + off;
        off = prev;
    }
    m.C.assign(cc.begin(), cc.begin() + n_look);
    m.C.push_back(m.length);

    timer.mark("cumulative counts");
    // FM:329-394: suffix array -> sampled rows, inverse samples, BWT (on the host, or in HBM)
    SaStage st;
    int rc;
    if (build_device >= 0) {
        if (!device_sa_stage) {  // host-only link of this file (sanitizer builds of the host code)
            err = "this build has no device construction stage";
            return -8;
        }
        std::vector<int16_t> codes16(65536, 0);  // monotonicMap.getOrDefault(ch, 0) as a table (absent characters never occur)
        for (int ch = 0; ch < 65536; ++ch)
            if (code_of[(size_t)ch] >= 0) codes16[(size_t)ch] = (int16_t)code_of[(size_t)ch];
        void *text_for_stage = d_text;
        d_text = nullptr;  // the stage frees it
        rc = device_sa_stage(seq.data(), n, sample_rate, enable_extract, build_device, st, stats, err,
                             device_wavelet ? &m.wt : nullptr, mapped, text_for_stage, codes16.data());
    } else {
        rc = host_sa_stage(seq.data(), n, n_look + 1, sample_rate, enable_extract, st);
    }
    if (rc) return rc;
    timer.mark("suffix-array stage");
    std::vector<int16_t>().swap(seq);
    m.bw_suffixes = min_bits((uint64_t)n);
    if (enable_extract) m.bw_positions = m.bw_suffixes;
    if (st.vectors_done) {  // packed and RRR-encoded in HBM
        m.suffixes = std::move(st.suffixes);
        if (enable_extract) m.positions = std::move(st.positions);
        m.sampled = std::move(st.sampled);
        timer.mark("(sample vectors came packed)");
    } else {
        m.suffixes.init(n / sample_rate + 1, m.bw_suffixes);
        for (size_t k = 0; k < st.suffix_vals.size(); ++k) m.suffixes.set((int64_t)k, st.suffix_vals[k]);
        timer.mark("pack suffix samples");
        build_rrr(st.which.data(), n, sample_rate, m.sampled, n >= (1 << 22) ? (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency())) : 1);
        timer.mark("RRR of the sample bitmap");
        if (enable_extract) {
            m.positions.init(n / sample_rate + 2, m.bw_positions);
            const int64_t n_pos = (int64_t)(n - 1) / sample_rate + 1;  // slots 0 .. (n-1)/s hold samples
            for (int64_t k = 0; k < n_pos; ++k) m.positions.set(k, st.position_vals[(size_t)k]);
            m.positions.set((n - 1) / sample_rate + 1, m.positions.get(0));  // FM:367-369
        }
        timer.mark("pack inverse samples");
    }
    if (!st.wavelet_done) {
        std::vector<int16_t> &bwt = st.bwt;
        build_wavelet(bwt.data(), n, sample_rate, m.wt);  // FM:173
        timer.mark("wavelet tree (host)");
    }
    return 0;
}

}  // namespace fmx
