// fmx_blob.cpp — flatten the host model into the HBM image described in fmx_blob.hpp.
#include "fmx_blob.hpp"
#include "fmx_device.hpp"  // the tree-walk arithmetic of rank(), evaluated once per leaf at flatten time
#include "fmx_model.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

namespace fmx {
namespace {

struct Arena {
    std::vector<uint8_t> &b;
    std::function<void(size_t)> before_growth;  // told the new size before every resize
    // reserve `bytes` at a 64-byte boundary, zero-filled; returns the byte offset
    size_t alloc(size_t bytes) {
        size_t off = (b.size() + 63) & ~(size_t)63;
        if (before_growth) before_growth(off + bytes);
        b.resize(off + bytes, 0);
        return off;
    }
    template <typename T>
    T *at(size_t off) {
        return reinterpret_cast<T *>(b.data() + off);
    }
};

inline uint32_t off8(size_t byte_off) { return (uint32_t)(byte_off >> 3); }

// RRR:92-103 -> 16-block records, and behind them the offsets bit stream (RrrRecord.offset_bit counts from the first record)
bool flatten_rrr(Arena &A, const RrrModel &r, RrrDesc &d, std::string &err) {
    if (r.sample_size <= 0 || r.classes.width != 4) {
        err = "unsupported RRR parameters";
        return false;
    }
    const uint8_t *bits_needed = rrr_bits_needed();
    const int64_t n_blocks = r.classes.length;
    const int64_t n_rec = n_blocks / 16 + 1;
    d.n_rec = (int32_t)n_rec;
    d.n_blocks = (int32_t)n_blocks;
    d.length = r.length;
    d.total_ones = r.total_ones;
    d.sample = r.sample_size;
    d.node_len = 0;
    d.off_bits = 0;
    const size_t rec_bytes = (size_t)n_rec * sizeof(RrrRecord) + 64;
    const size_t rec_off = A.alloc(rec_bytes + (r.offsets.size() + 2) * 8);
    d.off_rec = off8(rec_off);
    uint64_t ones = 0, obits = (uint64_t)rec_bytes * 8;
    for (int64_t k = 0; k < n_rec; ++k) {
        RrrRecord rec;
        rec.ones_before = (uint32_t)ones;
        rec.offset_bit = (uint32_t)obits;
        rec.classes = 0;
        for (int64_t j = 0; j < 16 && k * 16 + j < n_blocks; ++j) {
            const uint64_t cls = r.classes.get(k * 16 + j);
            rec.classes |= cls << (4 * j);
            ones += cls;
            obits += bits_needed[cls];
        }
        A.at<RrrRecord>(rec_off)[k] = rec;
    }
    if (obits - (uint64_t)rec_bytes * 8 > (uint64_t)r.offsets.size() * 64 || obits > 0xffffffffull) {
        err = "RRR offsets stream shorter than its classes imply (or records + stream beyond 512 MiB)";
        return false;
    }
    if (!r.offsets.empty()) memcpy(A.at<uint8_t>(rec_off + rec_bytes), r.offsets.data(), r.offsets.size() * 8);
    return true;
}

// RRR:92-103 -> expanded 96-bit cells with running one-counts (see fmx_blob.hpp).  Two steps so that the
// decoding of many vectors can run on many threads once all regions are allocated: reserve, then fill.
struct ExpandJob {
    const RrrModel *r;
    size_t cell_off;
    int64_t n_cells;
    // A long vector (the sampled-row bitmap: one bit per text character) is decoded by several jobs.  A job takes the
    // 15-bit blocks [b0, b1); b0 is a multiple of the sample size — where RrrVector keeps the offset pointer and the
    // prefix sum (RRR:277-283) — and of 32, so that its first bit, 15 * b0, starts a 96-bit cell.
    int64_t b0 = 0, b1 = -1;  // b1 < 0: the whole vector
};
bool expanded_reserve(Arena &A, const RrrModel &r, RrrDesc &d, ExpandJob &job, std::string &err) {
    if (r.classes.width != 4) {
        err = "unsupported RRR parameters";
        return false;
    }
    const int64_t n_cells = (int64_t)r.length / kBvCellBits + 2;  // + a cell for position == length and a guard
    d.n_rec = (int32_t)n_cells;
    d.n_blocks = r.classes.length;
    d.length = r.length;
    d.total_ones = r.total_ones;
    d.sample = r.sample_size;
    d.node_len = 0;
    d.off_bits = 0;
    job.r = &r;
    job.n_cells = n_cells;
    job.cell_off = A.alloc((size_t)n_cells * sizeof(BvCell));
    d.off_rec = off8(job.cell_off);
    return true;
}
// the jobs of one vector: one, or — above kSplitBlocks blocks — chunks of whole sample groups that start on cell boundaries
std::atomic<int64_t> g_split_blocks{1 << 20};
void expanded_jobs(const ExpandJob &whole, std::vector<ExpandJob> &out) {
    const RrrModel &r = *whole.r;
    const int64_t n_blocks = r.classes.length;
    const int64_t sample = r.sample_size;
    const int64_t kSplitBlocks = g_split_blocks;
    if (n_blocks <= kSplitBlocks || sample <= 0 || r.sampled_offsets.length <= 0 || r.prefix_sums.length <= 0) {
        out.push_back(whole);
        return;
    }
    int64_t unit = sample;  // lcm(sample, 32)
    while (unit % 32) unit += sample;
    const int64_t chunk = (kSplitBlocks / 4 + unit - 1) / unit * unit;
    for (int64_t b0 = 0; b0 < n_blocks; b0 += chunk) {
        ExpandJob j = whole;
        j.b0 = b0;
        j.b1 = b0 + chunk < n_blocks ? b0 + chunk : n_blocks;
        // (a chunk needs the vector's own samples at its first block: a model whose sample vectors are too short is
        // decoded in one piece, where the stream itself is the only authority)
        if (b0 / sample >= r.sampled_offsets.length || b0 / sample >= r.prefix_sums.length) {
            out.resize(out.size() - (size_t)(b0 / chunk));
            out.push_back(whole);
            return;
        }
        out.push_back(j);
    }
}
bool expanded_fill(BvCell *cells, const ExpandJob &job, std::string &err) {
    const RrrModel &r = *job.r;
    const uint8_t *bits_needed = rrr_bits_needed();
    const uint16_t *value_of = rrr_value_of_offset(), *class_base = rrr_class_base();
    const int64_t n_blocks = r.classes.length;
    const bool whole = job.b1 < 0;
    const int64_t b0 = whole ? 0 : job.b0, b1 = whole ? n_blocks : job.b1;
    const bool last = b1 >= n_blocks;
    // the cells this job writes: from its first bit up to its last block's end — the last job also the tail cells
    const int64_t c0 = b0 * 15 / kBvCellBits;
    const int64_t c1 = last ? job.n_cells : b1 * 15 / kBvCellBits;
    // RRR:382-390 for every block: (class, offset) -> 15 bits, into a plain LSB-first bit array ...
    std::vector<uint64_t> plain((size_t)((c1 - c0) * kBvCellBits / 64 + 2), 0);
    uint64_t obits = whole ? 0 : r.sampled_offsets.get(b0 / r.sample_size);  // RRR:371-372
    uint64_t ones = whole ? 0 : r.prefix_sums.get(b0 / r.sample_size);        // RRR:370
    const uint64_t avail = (uint64_t)r.offsets.size() * 64;
    const int64_t bit0 = c0 * (int64_t)kBvCellBits;
    for (int64_t b = b0; b < b1; ++b) {
        const int cls = (int)r.classes.get(b);
        const int nb = bits_needed[cls];
        if (obits + (uint64_t)nb > avail) {
            err = "RRR offsets stream shorter than its classes imply";
            return false;
        }
        const size_t w = (size_t)(obits >> 6);
        const int sh = (int)(obits & 63);
        uint64_t off = r.offsets[w] >> sh;
        if (sh + nb > 64) off |= r.offsets[w + 1] << (64 - sh);
        off &= (1ull << nb) - 1ull;
        obits += (uint64_t)nb;
        // (an offset beyond its class's C(15, class) values: the reference would read a neighbouring class's entry, or
        // past VALUE_OF_OFFSET — no encoder writes it)
        const size_t class_end = cls < 15 ? (size_t)class_base[cls + 1] : (size_t)32768;
        if ((size_t)class_base[cls] + (size_t)off >= class_end) {
            err = "RRR offset outside its class";
            return false;
        }
        const uint64_t value = value_of[(size_t)class_base[cls] + (size_t)off];
        const int64_t pos = b * 15;
        if (pos + 15 > job.n_cells * (int64_t)kBvCellBits) {
            err = "RRR vector has more blocks than its length";
            return false;
        }
        const size_t pw = (size_t)((pos - bit0) >> 6);
        const int ps = (int)((pos - bit0) & 63);
        plain[pw] |= value << ps;
        if (ps + 15 > 64) plain[pw + 1] |= value >> (64 - ps);
    }
    // ... cut into 96-bit cells with running one-counts (bits past `length` are zero: RRR pads its last block)
    const uint32_t *plain32 = reinterpret_cast<const uint32_t *>(plain.data());
    for (int64_t c = c0; c < c1; ++c) {
        BvCell cell;
        cell.ones_before = (uint32_t)ones;
        for (int k = 0; k < 3; ++k) {
            cell.bits[k] = plain32[(size_t)(c - c0) * 3 + (size_t)k];
            ones += (uint64_t)__builtin_popcount(cell.bits[k]);
        }
        cells[c] = cell;
    }
    // the stream and the vector's own counters must agree: with the next chunk's prefix sum, or with totalOnes at the end
    const uint64_t expect = last ? (uint64_t)(uint32_t)r.total_ones
                                 : (b1 / r.sample_size < r.prefix_sums.length ? r.prefix_sums.get(b1 / r.sample_size) : ones);
    if (ones != expect) {
        err = "RRR vector decodes to a different number of ones than it declares";
        return false;
    }
    return true;
}

size_t put_packed(Arena &A, const PackedVec &v) {
    const size_t off = A.alloc((v.words.size() + 2) * 8);
    if (!v.words.empty()) memcpy(A.at<uint8_t>(off), v.words.data(), v.words.size() * 8);
    return off;
}


// ---- inverseSelect section (fmx_blob.hpp: InvHdr, NodeRec) --------------------------------------------------------
// One block's tree: every leaf's canonical code is walked with the arithmetic of WFBB:1386-1493 (the device header's
// TreeWalk functions, run here on the host), the bits coming from the code instead of the bit vector.  Every node on
// the way gets its {A, B}; every (node, bit) its child.  Returns false when the header does not describe a walk this
// table can stand for (the block then keeps the reference's own route).
struct InvBlockTable {
    std::vector<NodeRec> nodes;
    std::vector<uint8_t> seen;  // bit 0 / 1: child[0] / child[1] filled
};
inline uint64_t inv_internal_half(uint32_t idx, uint32_t a, uint32_t b) {
    return (uint64_t)idx | ((uint64_t)a << 16) | ((uint64_t)b << 40);
}
inline uint64_t inv_leaf_half(uint32_t symbol, uint32_t folded) { return ((uint64_t)symbol << 16) | ((uint64_t)folded << 32); }

bool build_inverse_block(const uint8_t *var, int64_t var_len, const BlockHeader &bh, int32_t cur_block_size, int sigma,
                         const SbcEntry *sbc_row, InvBlockTable &out, uint32_t &root_a, uint32_t &root_b) {
    const int h = bh.tree_height, n_leaves = (int)bh.sigma + 1;
    if (h < 1 || h > 30 || n_leaves < 2 || bh.var_off < 0) return false;
    const int64_t hdr = bh.var_off, leaves = hdr + (int64_t)(h - 1) * 4;
    const int64_t second0 = (int64_t)(h - 1) * 4 + (int64_t)n_leaves * 5;
    if (hdr + second0 + (int64_t)(n_leaves - 1) * 2 > var_len) return false;
    if (((uint32_t)bh.bv_offset | (uint32_t)bh.bv_rank) >> 24) return false;
    const uint32_t counts0 = ld16(var + hdr + second0);
    const int n_nodes = n_leaves - 1;
    out.nodes.assign((size_t)n_nodes, NodeRec{{0, 0}});
    out.seen.assign((size_t)n_nodes, 0);
    root_a = (uint32_t)bh.bv_offset;
    root_b = (uint32_t)bh.bv_rank;
    for (int i = 0; i < n_leaves; ++i) {
        const uint8_t *lp = var + leaves + (int64_t)i * 5;
        const uint32_t symbol = ld16(lp), rank_block = ld24(lp + 2);
        if ((int)symbol >= sigma) return false;
        uint32_t code = 0;
        int32_t len = 0;
        wt_restore_code((uint32_t)i, var + hdr, h, ld_quad(var + hdr), code, len);  // WFBB:250-278
        if (len < 1 || len > h || len > 31) return false;
        TreeWalk t;
        t.bv_rank = bh.bv_rank;
        t.bv_offset = bh.bv_offset;
        t.internal_nodes = 1;
        t.left_siblings = 0;
        t.left_total_bv = 0;
        t.node_bv_size = cur_block_size;
        t.depth_total_bv = t.node_bv_size;
        t.node_rank = 0;
        t.hdr = var + hdr;
        t.second = (uint32_t)second0;
        t.level = 0;
        t.limit = (uint32_t)(var_len - hdr);
        int32_t left_ones = 0, node_ones = (int32_t)counts0, level_ones = (int32_t)counts0;
        int64_t level_base = 0;  // index of the first internal node of the current level
        int64_t cur = 0;         // the node the walk stands on
        uint32_t walked = 0;     // the bits read so far (inverseSelect's `code`)
        for (int32_t depth = 0;; ++depth) {
            if (depth >= len) return false;  // the walk goes deeper than the leaf's code
            const bool bit = (code >> (len - depth - 1)) & 1u;
            const int32_t internal_here = t.internal_nodes;
            t.bv_rank += level_ones;
            walked = (walked << 1) | (bit ? 1u : 0u);
            tree_descend(t, bit, 0, node_ones);
            bool leaf_child = true;
            if (depth + 1 < h) {  // WFBB:1480-1489
                if ((int64_t)t.level + 4 > (int64_t)(h - 1) * 4) return false;
                const int32_t next_leaf_count = tree_next_level_entry(t, ld32u(t.hdr + t.level));
                if (t.left_siblings >= next_leaf_count) {
                    t.left_siblings -= next_leaf_count;
                    leaf_child = false;
                }
            }
            uint64_t half;
            int64_t child = -1;
            if (leaf_child) {
                if (depth + 1 != len) return false;
                // WFBB:232-248 computeSymbolFromBlockHeader must land on this very leaf
                uint32_t block_c = 0, temp_code = 0;
                for (int32_t k = 1; k < len; ++k) {
                    const uint32_t level_leaf_count = ld32u(var + hdr + 4 * (k - 1)) & 0xffffu;
                    temp_code += level_leaf_count;
                    block_c += level_leaf_count;
                    temp_code <<= 1;
                }
                block_c += walked - temp_code;
                if (block_c != (uint32_t)i) return false;
                half = inv_leaf_half(symbol, (uint32_t)sbc_row[symbol].rank + rank_block);
            } else {
                if (depth + 1 >= len) return false;
                if (t.internal_nodes <= 0 || t.left_siblings < 0 || t.left_siblings >= t.internal_nodes ||
                    hdr + (int64_t)t.second + 2 * (int64_t)t.internal_nodes > var_len)
                    return false;
                tree_level_counts(t, left_ones, node_ones, level_ones);
                child = level_base + internal_here + t.left_siblings;
                level_base += internal_here;
                const int64_t a = (int64_t)t.bv_offset + t.left_total_bv, b = (int64_t)t.bv_rank + left_ones;
                if (child <= cur || child >= n_nodes || a < 0 || b < 0 || (a >> 24) || (b >> 24)) return false;
                half = inv_internal_half((uint32_t)child, (uint32_t)a, (uint32_t)b);
            }
            NodeRec &rec = out.nodes[(size_t)cur];
            const uint8_t mask = bit ? 2 : 1;
            if ((out.seen[(size_t)cur] & mask) && rec.child[bit ? 1 : 0] != half) return false;  // two leaves disagree
            rec.child[bit ? 1 : 0] = half;
            out.seen[(size_t)cur] |= mask;
            if (leaf_child) break;
            cur = child;
        }
    }
    for (int k = 0; k < n_nodes; ++k)
        if (out.seen[(size_t)k] != 3) return false;  // not a full binary tree
    return true;
}

}  // namespace

// a stand-alone RrrVector (fmx_rrr_build): header + value-of-offset table + the vector in its compressed form
int flatten_rrr_only(const RrrModel &r, std::vector<uint8_t> &blob, std::string &err) {
    blob.clear();
    Arena A{blob, {}};
    const size_t hdr_off = A.alloc(sizeof(BlobHeader));
    BlobHeader h;
    memset(&h, 0, sizeof h);
    h.magic = kBlobMagic;
    h.version = kBlobVersion;
    h.length = r.length;
    h.sample_rate = r.sample_size;
    const size_t inv_off = A.alloc((size_t)kInvEntries * 2);
    memcpy(A.at<uint8_t>(inv_off), rrr_value_of_offset(), (size_t)kInvEntries * 2);  // classes 0..7 come first
    h.off_inv = off8(inv_off);
    if (!flatten_rrr(A, r, h.sampled, err)) return -8;
    A.alloc(64);
    h.total_bytes = blob.size();
    h.kind = 1;
    h.checksum = 0;
    memcpy(A.at<uint8_t>(hdr_off), &h, sizeof h);
    A.at<BlobHeader>(hdr_off)->checksum = image_checksum(blob.data(), blob.size());
    return 0;
}

// blocks above which a bit vector is decoded in chunks (tests lower it: the image must not depend on it)
void set_split_blocks(int64_t blocks) { g_split_blocks = blocks < 64 ? 64 : blocks; }
// -1 = by alphabet size; 0 / 1 force the row layout of the mapping tables (tests exercise both)
static std::atomic<int> g_map_by_symbol{-1};
void set_map_by_symbol(int mode) { g_map_by_symbol = mode; }
// 1: images flattened from now on keep their bit vectors compressed (BlobHeader.compact)
static std::atomic<int> g_image_compact{0};
void set_image_compact(int on) { g_image_compact = on ? 1 : 0; }
// 0: every present mapping entry says "take the reference's own route" (tests: the slow path alone must give
// the same answers)
static std::atomic<int> g_map_fast{1};
void set_map_fast(int on) { g_map_fast = on; }
// 0: every block's InvHdr says "take the reference's own route" (tests)
static std::atomic<int> g_inv_fast{1};
void set_inv_fast(int on) { g_inv_fast = on; }

namespace {
struct FlattenTimer {  // FMX_BUILD_TIMING=1 prints the wall time of the flattener's phases to stderr
    const bool on = getenv("FMX_BUILD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fmx flatten] %-26s %.3f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};
}  // namespace

int flatten_model(const FmModel &m, std::vector<uint8_t> &blob, std::string &err) {
    FlattenTimer timer;
    const WfbbModel &w = m.wt;
    const int sigma = w.alphabet_size;
    const int64_t n_sb = (int64_t)w.sb.size();
    if (sigma <= 0 || (int64_t)w.hyper_rank.size() != sigma || w.size != m.length ||
        (int64_t)w.super_rank.size() != n_sb * sigma || (int64_t)w.global_mapping.size() != n_sb * sigma ||
        (int64_t)w.count.size() != sigma || n_sb != (w.size + (1 << 20) - 1) / (1 << 20)) {
        err = "wavelet tree shape not supported (expects one 2^32 hyperblock, size == length)";
        return -8;
    }
    if (m.bw_suffixes > 32 || m.bw_positions > 32 || m.C.empty()) {
        err = "unsupported bit widths";
        return -8;
    }
    blob.clear();
    // every superblock's own table shape first (WFBB:461-471: superBlockSigma x blocks per superblock entries): what is laid
    // out and reserved below is then bounded by what the model itself holds, whatever a damaged header field says
    for (int64_t s = 0; s < n_sb; ++s) {
        const SuperBlockModel &sb = w.sb[(size_t)s];
        const int bsl = sb.block_size_log;
        if (bsl < 0 || bsl > 20 || sb.sigma < -1 || (int64_t)sb.mapping.size() != ((int64_t)sb.sigma + 1) << (20 - bsl)) {
            err = "superblock mapping shape mismatch";
            return -3;
        }
        if (sb.rank_support.length < 0 || sb.rank_support.length > (int64_t)15 * sb.rank_support.classes.length + 15) {
            err = "superblock bit vector longer than its RRR blocks";
            return -3;
        }
    }
    if (m.sampled.length < 0 || m.sampled.length > (int64_t)15 * m.sampled.classes.length + 15) {
        err = "sampled-row bitmap longer than its RRR blocks";
        return -3;
    }
    // mapping rows by global symbol (rows of symbols a superblock does not hold are pure skip pointers)
    const int map_mode = g_map_by_symbol;
    bool by_symbol = map_mode > 0;
    if (map_mode < 0) {  // automatic: rows by symbol unless that more than doubles the tables
        int64_t rows_by_code = 0, rows_by_symbol = 0;
        for (int64_t s = 0; s < n_sb; ++s) {
            const int bsl = w.sb[(size_t)s].block_size_log;
            rows_by_code += ((int64_t)w.sb[(size_t)s].sigma + 1) << (20 - bsl);
            rows_by_symbol += (int64_t)sigma << (20 - bsl);
        }
        by_symbol = rows_by_symbol <= 2 * rows_by_code;
    }
    {
        // ONE allocation for the whole image (an upper bound): the arena never moves — the cell decoders below run while
        // this thread still appends tables — and no page is touched twice by a reallocation
        auto cells_bytes = [](const RrrModel &r) { return ((size_t)r.length / kBvCellBits + 2) * sizeof(BvCell) + 64; };
        size_t bound = (4u << 20) + 65536 * 2 + (m.suffixes.words.size() + m.positions.words.size() + 8) * 8 +
                       (m.C.size() + m.look_up.size()) * 4 + cells_bytes(m.sampled) +
                       (size_t)(n_sb + 1) * (size_t)sigma * sizeof(SbcEntry) + (size_t)n_sb * (sizeof(SbDesc) + 1024);
        for (int64_t s = 0; s < n_sb; ++s) {
            const SuperBlockModel &sb = w.sb[(size_t)s];
            const int bsl = sb.block_size_log;
            const size_t per_row = (size_t)1 << (20 - bsl);
            const size_t rows = by_symbol ? (size_t)sigma : (size_t)((int64_t)sb.sigma + 1);
            bound += rows * per_row * sizeof(MapEntry) + sb.var.size() * 32 + sb.block_headers.size() * 64 +
                     cells_bytes(sb.rank_support) + 4096;
        }
        blob.reserve(bound);
    }
    Arena A{blob, {}};
    const size_t hdr_off = A.alloc(sizeof(BlobHeader));
    BlobHeader h;
    memset(&h, 0, sizeof h);
    h.magic = kBlobMagic;
    h.version = kBlobVersion;
    h.sample_rate = m.sample_rate;
    h.enable_extract = m.enable_extract ? 1 : 0;
    h.length = m.length;
    h.n_keys = (int32_t)m.map_keys.size();
    h.bw_suffixes = m.bw_suffixes;
    h.bw_positions = m.bw_positions;
    h.n_c = (int32_t)m.C.size();
    h.n_look = (int32_t)m.look_up.size();
    h.wt_sigma = sigma;
    h.n_sb = (int32_t)n_sb;
    h.n_suffixes = m.suffixes.length;
    h.n_positions = m.enable_extract ? m.positions.length : 0;
    h.wt_size = w.size;

    size_t off = A.alloc(m.C.size() * 4 + 8);
    h.off_c = off8(off);
    memcpy(A.at<uint8_t>(off), m.C.data(), m.C.size() * 4);
    // (zero-padded to the wavelet tree's alphabet + 1: the kernels clamp a symbol to that before they look its character
    // up, so that a symbol no well-formed tree holds still reads inside the table — fm_char_of)
    off = A.alloc(std::max(m.look_up.size(), (size_t)w.alphabet_size + 1) * 4 + 8);
    h.off_lookup = off8(off);
    memcpy(A.at<uint8_t>(off), m.look_up.data(), m.look_up.size() * 4);
    off = A.alloc(65536 * 2);
    h.off_char2code = off8(off);
    for (size_t i = 0; i < m.map_keys.size(); ++i)
        if (m.map_keys[i] >= 0 && m.map_keys[i] < 65536) A.at<int16_t>(off)[m.map_keys[i]] = m.map_vals[i];
    h.off_suffixes = off8(put_packed(A, m.suffixes));
    h.off_positions = m.enable_extract ? off8(put_packed(A, m.positions)) : 0;
    const bool compact = g_image_compact != 0;
    std::vector<ExpandJob> jobs((size_t)n_sb + 1);
    h.off_inv = 0;  // (an expanded image holds no compressed RRR vector: no value table)
    if (compact) {
        // the vectors stay what RRR:225-286 made of them — classes and offsets — regrouped into 16-block records; the kernels
        // decode a block through the value-of-offset table (classes 0..7; staged in LDS)
        const size_t inv_off = A.alloc((size_t)kInvEntries * 2);
        memcpy(A.at<uint8_t>(inv_off), rrr_value_of_offset(), (size_t)kInvEntries * 2);
        h.off_inv = off8(inv_off);
        if (!flatten_rrr(A, m.sampled, h.sampled, err)) return -8;
    } else if (!expanded_reserve(A, m.sampled, h.sampled, jobs[(size_t)n_sb], err)) {
        return -8;
    }
    h.compact = compact ? 1 : 0;

    // fused (rank, superblock code) table; row n_sb = total counts (WFBB:1063-1069)
    off = A.alloc((size_t)(n_sb + 1) * sigma * sizeof(SbcEntry));
    h.off_sbc = off8(off);
    for (int64_t s = 0; s <= n_sb; ++s)
        for (int c = 0; c < sigma; ++c) {
            SbcEntry e;
            if (s < n_sb) {
                e.rank = (int32_t)(w.hyper_rank[(size_t)c] + w.super_rank[(size_t)(s * sigma + c)]);
                e.sbc = w.global_mapping[(size_t)(s * sigma + c)];
            } else {
                e.rank = (int32_t)w.count[(size_t)c];
                e.sbc = (int16_t)(sigma - 1);
            }
            e.pad = 0;
            // folded: + cumulativeCounts[c] (FM:103), so that an LF-step's C[c] + rank is one table entry
            if ((size_t)c < m.C.size()) e.rank += m.C[(size_t)c];
            A.at<SbcEntry>(off)[s * sigma + c] = e;
        }

    h.map_by_symbol = by_symbol ? 1 : 0;
    const size_t sbd_off = A.alloc((size_t)n_sb * sizeof(SbDesc));
    h.off_sbdesc = off8(sbd_off);
    // Every bit vector's cells are reserved now and decoded on the other cores WHILE this thread builds the
    // per-superblock tables below.  The arena must not move under the decoders: its capacity was reserved for an upper
    // bound of the whole image (if that bound should ever fall short, the tables wait for the decoders).
    std::vector<RrrDesc> sb_rrr((size_t)n_sb);
    for (int64_t s = 0; s < n_sb; ++s)
        if (compact ? !flatten_rrr(A, w.sb[(size_t)s].rank_support, sb_rrr[(size_t)s], err)
                    : !expanded_reserve(A, w.sb[(size_t)s].rank_support, sb_rrr[(size_t)s], jobs[(size_t)s], err))
            return -8;
    std::vector<ExpandJob> work;  // the longest vector first, in chunks (none for a compact image: nothing is decoded)
    if (!compact) {
        expanded_jobs(jobs[(size_t)n_sb], work);
        for (int64_t s = 0; s < n_sb; ++s) expanded_jobs(jobs[(size_t)s], work);
    }
    std::atomic<size_t> next_job{0};
    std::mutex err_mutex;
    bool failed = false;
    uint8_t *const arena_base = blob.data();
    auto worker = [&]() {
        for (;;) {
            const size_t j = next_job.fetch_add(1);
            if (j >= work.size()) return;
            std::string local;
            if (!expanded_fill(reinterpret_cast<BvCell *>(arena_base + work[j].cell_off), work[j], local)) {
                std::lock_guard<std::mutex> lock(err_mutex);
                failed = true;
                err = local;
            }
        }
    };
    std::vector<std::thread> pool;
    {
        unsigned n_threads = std::thread::hardware_concurrency();
        if (n_threads == 0) n_threads = 1;
        if (n_threads > work.size()) n_threads = (unsigned)work.size();
        for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    }
    auto join_decoders = [&]() {
        for (auto &t : pool) t.join();
        pool.clear();
    };
    struct JoinGuard {  // (an early return below must not leave threads running into a dying arena)
        std::function<void()> f;
        ~JoinGuard() { f(); }
    } join_guard{join_decoders};
    A.before_growth = [&](size_t need) {
        if (need > blob.capacity()) join_decoders();  // the bound fell short: no decoder may run while the arena moves
    };
    timer.mark("reserve + start decoders");
    const bool fast = g_map_fast != 0;
    std::vector<MapEntry> map_scratch;
    std::vector<PathRec> path_scratch;
    std::vector<uint8_t> var_scratch;
    std::vector<InvHdr> inv_hdr_scratch;
    std::vector<NodeRec> inv_node_scratch;
    InvBlockTable inv_table;
    for (int64_t s = 0; s < n_sb; ++s) {
        const SuperBlockModel &sb = w.sb[(size_t)s];
        SbDesc d;
        memset(&d, 0, sizeof d);
        d.sigma = sb.sigma;
        d.bsl = sb.block_size_log;
        d.n_blocks = (int32_t)sb.block_headers.size();
        if (d.bsl < 0 || d.bsl > 20 || (int64_t)sb.mapping.size() != ((int64_t)sb.sigma + 1) << (20 - d.bsl)) {
            err = "superblock mapping shape mismatch";
            return -3;
        }
        const int64_t per_row = (int64_t)1 << (20 - d.bsl);
        const int64_t n_rows = by_symbol ? sigma : (int64_t)sb.sigma + 1;
        d.mapping_len = (int32_t)(n_rows * per_row);
        {
            std::vector<MapEntry> &ents = map_scratch;
            std::vector<PathRec> &path = path_scratch;
            ents.assign((size_t)(n_rows * per_row), MapEntry{0, 0, 0, 0});
            path.clear();
            // absent entries (alphabetSize - 1, WFBB:383-387) become skip pointers: the distance to the next
            // block to the right holding the symbol, or to the end of the superblock
            const int16_t absent = (int16_t)(sigma - 1);
            for (int64_t row = 0; row < n_rows; ++row) {
                // the reference's row of this device row: by superblock code (WFBB:461-465)
                const int64_t src_row = by_symbol ? (int64_t)w.global_mapping[(size_t)(s * sigma + row)] : row;
                const bool in_superblock = src_row >= 0 && src_row <= sb.sigma;
                int64_t next_present = per_row;
                for (int64_t blk = per_row - 1; blk >= 0; --blk) {
                    const int16_t v = in_superblock ? sb.mapping[(size_t)(src_row * per_row + blk)] : absent;
                    MapEntry &e = ents[(size_t)(row * per_row + blk)];
                    if (v != absent) {
                        if (v < 0) {
                            err = "negative mapping entry";
                            return -3;
                        }
                        next_present = blk;
                        // until the block's header has been read (below) a present entry takes the reference's route
                        e.x = kMapSlow | ((uint32_t)(uint16_t)v << 8);
                    } else {
                        e.x = kMapAbsent | ((uint32_t)(next_present - blk) << 8);
                    }
                }
            }
            // what rank() needs about every (symbol, block) that occurs, from the block's own header: the leaf's rank
            // at the block start, its canonical code, and the nodes its walk visits (PathRec)
            std::vector<uint8_t> &var = var_scratch;  // + guard bytes: fields are fetched as whole dwords
            var.assign(sb.var.begin(), sb.var.end());
            var.resize(sb.var.size() + 16, 0);
            const int64_t var_len = (int64_t)sb.var.size();
            const int64_t sb_start = s << 20;
            for (int64_t blk = 0; fast && blk < (int64_t)sb.block_headers.size() && blk < per_row; ++blk) {
                const BlockHeader &bh = sb.block_headers[(size_t)blk];
                const int h = bh.tree_height, n_leaves = (int)bh.sigma + 1;
                if (h < 0 || h > 30 || n_leaves <= 0 || bh.var_off < 0) continue;
                const int64_t hdr = bh.var_off, leaves = hdr + (h > 0 ? (int64_t)(h - 1) * 4 : 0);
                const int64_t second0 = (int64_t)(h - 1) * 4 + (int64_t)n_leaves * 5;
                // the whole header the reference would touch (WFBB:479-483) must lie inside the byte array
                if (hdr + (h > 1 ? (int64_t)(h - 1) * 4 : 0) + (int64_t)n_leaves * 5 + (int64_t)(n_leaves - 1) * 2 > var_len) continue;
                if (((uint32_t)bh.bv_offset | (uint32_t)bh.bv_rank) >> 24) continue;
                const uint32_t counts0 = h > 0 ? ld16(var.data() + hdr + second0) : 0u;
                const int64_t block_start = sb_start + (blk << d.bsl);
                const int64_t left = w.size - block_start;
                const int32_t cur_block_size = (int32_t)(left < ((int64_t)1 << d.bsl) ? left : ((int64_t)1 << d.bsl));  // WFBB:1032
                for (int i = 0; i < n_leaves; ++i) {
                    const uint8_t *lp = var.data() + leaves + (int64_t)i * 5;
                    const int symbol = (int)ld16(lp);
                    const uint32_t rank_block = ld24(lp + 2);
                    if (symbol >= sigma) continue;
                    const int code_row = w.global_mapping[(size_t)(s * sigma + symbol)];
                    if (code_row < 0 || code_row > sb.sigma) continue;
                    MapEntry &e = ents[(size_t)((int64_t)(by_symbol ? symbol : code_row) * per_row + blk)];
                    // only where the mapping holds the leaf's own index: an absent entry stays absent, a clamped one
                    // (min(sigma-2, index), WFBB:466-471) keeps the reference's route with its fix-up (WFBB:1123-1130)
                    if ((e.x & 0xffu) != kMapSlow || (e.x >> 8) != (uint32_t)i) continue;
                    uint32_t code = 0;
                    int32_t len = 0;
                    if (h > 0) {
                        Quad chunk = ld_quad(var.data() + hdr);
                        wt_restore_code((uint32_t)i, var.data() + hdr, h, chunk, code, len);  // WFBB:250-278
                    }
                    if (len > (int32_t)kMapMaxLen || (code >> 16)) continue;  // stays on the reference's route
                    // the walk of WFBB:1185-1279 along this code, position-independent part
                    TreeWalk t;
                    t.bv_rank = bh.bv_rank;
                    t.bv_offset = bh.bv_offset;
                    t.internal_nodes = 1;
                    t.left_siblings = 0;
                    t.left_total_bv = 0;
                    t.node_bv_size = cur_block_size;
                    t.depth_total_bv = t.node_bv_size;
                    t.node_rank = 0;
                    t.hdr = var.data() + hdr;
                    t.second = (uint32_t)second0;
                    t.level = 0;
                    t.limit = (uint32_t)(var_len - hdr);
                    int32_t left_ones = 0, node_ones = (int32_t)counts0, level_ones = (int32_t)counts0;
                    const size_t first_rec = path.size();
                    bool ok = true;
                    for (int32_t depth = 0; depth < len; ++depth) {
                        const int64_t a = (int64_t)t.bv_offset + t.left_total_bv, b = (int64_t)t.bv_rank + left_ones;
                        if (a < 0 || b < 0 || (a >> 24) || (b >> 24)) {
                            ok = false;
                            break;
                        }
                        if (depth > 0) path.push_back(PathRec{(uint32_t)a, (uint32_t)b});
                        t.bv_rank += level_ones;
                        tree_descend(t, (code & (1u << (len - depth - 1))) != 0, 0, node_ones);
                        if (depth + 1 != len) {
                            if ((int64_t)t.level + 4 > (int64_t)(h - 1) * 4) {  // level table exhausted: malformed header
                                ok = false;
                                break;
                            }
                            t.left_siblings -= tree_next_level_entry(t, ld32u(t.hdr + t.level));
                            if (t.left_siblings < 0 || t.internal_nodes <= 0 || t.left_siblings >= t.internal_nodes ||
                                hdr + t.second + 2 * (int64_t)t.internal_nodes > var_len) {
                                ok = false;
                                break;
                            }
                            tree_level_counts(t, left_ones, node_ones, level_ones);
                        }
                    }
                    if (!ok || (rank_block >> 24)) {
                        path.resize(first_rec);
                        continue;
                    }
                    e.x = (uint32_t)len | (rank_block << 8);
                    e.y = (uint32_t)bh.bv_offset | ((code & 0xffu) << 24);
                    e.z = (uint32_t)bh.bv_rank | ((code >> 8) << 24);
                    e.w = (uint32_t)(2 * (uint64_t)ents.size() + first_rec);
                    if (len == 0) {
                        // A run block has no path records.  Its entry's last word holds what the NEXT-BLOCK path of rank() reads
                        // when it lands on this block from an absent entry to the left (WFBB:1072-1108 with tree height 0: the
                        // u24 at var_off + (0 - 1) * 4 + 0 * 5 + 2 = var_off - 2, i.e. two bytes of the header BEFORE this
                        // block's and one of its own: Q11) — a property of the block alone, evaluated here once instead of by
                        // two dependent loads (block header, header bytes) per rank: kMapRunNext | u24, or kMapRunNextOutside
                        // where the reference's read leaves the byte array (ArrayIndexOutOfBounds).
                        const int64_t q = hdr - 2;
                        if (q < 0 || q + 2 >= var_len)
                            e.w = kMapRunNextOutside;
                        else
                            e.w = kMapRunNext | ld24(var.data() + q);
                    }
                }
            }
            d.path_len = (int32_t)path.size();
            off = A.alloc(ents.size() * sizeof(MapEntry) + path.size() * sizeof(PathRec) + 32);
            d.off_mapping = off8(off);
            if (!ents.empty()) memcpy(A.at<uint8_t>(off), ents.data(), ents.size() * sizeof(MapEntry));
            if (!path.empty())
                memcpy(A.at<uint8_t>(off) + ents.size() * sizeof(MapEntry), path.data(), path.size() * sizeof(PathRec));
        }
        off = A.alloc(sb.block_headers.size() * sizeof(BlockHdr) + 16);
        d.off_bh = off8(off);
        static_assert(sizeof(BlockHeader) == sizeof(BlockHdr), "block header layout");
        if (!sb.block_headers.empty())
            memcpy(A.at<uint8_t>(off), sb.block_headers.data(), sb.block_headers.size() * sizeof(BlockHdr));
        {
            // the root node's one-count (first u16 of the cumulative counts, WFBB:793-809) rides in the spare top
            // bytes of bv_rank / bv_offset (both < 2^24: at most 2^20 symbols of < 16 code bits per superblock)
            BlockHdr *bhd = A.at<BlockHdr>(off);
            for (size_t b = 0; b < sb.block_headers.size(); ++b) {
                if (((uint32_t)bhd[b].bv_rank | (uint32_t)bhd[b].bv_offset) >> 24) {
                    err = "block bit-vector fields exceed 24 bits";
                    return -8;
                }
                const int h = bhd[b].tree_height;
                const int64_t at = (int64_t)bhd[b].var_off + (int64_t)(h - 1) * 4 + ((int64_t)bhd[b].sigma + 1) * 5;
                if (h <= 0 || bhd[b].var_off < 0 || at + 2 > (int64_t)sb.var.size()) continue;
                const uint32_t root_ones = (uint32_t)sb.var[(size_t)at] | ((uint32_t)sb.var[(size_t)at + 1] << 8);
                bhd[b].bv_rank |= (int32_t)((root_ones & 0xffu) << 24);
                bhd[b].bv_offset |= (int32_t)((root_ones >> 8) << 24);
            }
        }
        // 16 guard bytes behind the array: 24-bit fields are fetched as one 32-bit load
        off = A.alloc(sb.var.size() + 16);
        d.off_var = off8(off);
        d.var_len = (int32_t)sb.var.size();
        if (!sb.var.empty()) memcpy(A.at<uint8_t>(off), sb.var.data(), sb.var.size());
        d.rrr = sb_rrr[(size_t)s];
        {
            // inverseSelect section: InvHdr per block, NodeRec per internal node (fmx_blob.hpp)
            const SbcEntry *sbc_row = A.at<SbcEntry>((size_t)h.off_sbc << 3) + (size_t)s * (size_t)sigma;
            const uint8_t *var = var_scratch.data();  // sb.var + 16 guard bytes (filled above)
            const int64_t var_len = (int64_t)sb.var.size();
            const size_t n_blk = sb.block_headers.size();
            std::vector<InvHdr> &hdrs = inv_hdr_scratch;
            std::vector<NodeRec> &nodes = inv_node_scratch;
            hdrs.assign(n_blk, InvHdr{kInvSlow, 0, 0, 0});
            nodes.clear();
            const bool inv_fast = g_inv_fast != 0;
            for (size_t b = 0; inv_fast && b < n_blk; ++b) {
                const BlockHeader &bh = sb.block_headers[b];
                const int hgt = bh.tree_height, n_leaves = (int)bh.sigma + 1;
                if (hgt < 0 || n_leaves <= 0 || bh.var_off < 0) continue;
                if (hgt == 0) {  // WFBB:1329-1355
                    if ((int64_t)bh.var_off + 5 > var_len) continue;
                    const uint8_t *lp = var + bh.var_off;
                    const uint32_t symbol = ld16(lp), masked = symbol & 0xffu, rank_block = ld24(lp + 2);  // Q1
                    if ((int)masked >= sigma) continue;
                    hdrs[b] = InvHdr{kInvRun | (masked != symbol ? kInvMasked : 0u), masked,
                                     (uint32_t)sbc_row[masked].rank + rank_block, 0};
                    continue;
                }
                const int64_t block_start = ((int64_t)s << 20) + ((int64_t)b << d.bsl);
                const int64_t left = w.size - block_start;
                const int32_t cur_block_size = (int32_t)(left < ((int64_t)1 << d.bsl) ? left : ((int64_t)1 << d.bsl));
                uint32_t root_a, root_b;
                if (!build_inverse_block(var, var_len, bh, cur_block_size, sigma, sbc_row, inv_table, root_a, root_b)) continue;
                hdrs[b] = InvHdr{root_a, root_b, (uint32_t)(n_blk + nodes.size()), (uint32_t)inv_table.nodes.size()};
                nodes.insert(nodes.end(), inv_table.nodes.begin(), inv_table.nodes.end());
            }
            d.rrr.node_len = (int32_t)(n_blk + nodes.size());
            const size_t inv_off = A.alloc((n_blk + nodes.size()) * 16 + 64);
            d.rrr.off_bits = off8(inv_off);
            if (n_blk) memcpy(A.at<uint8_t>(inv_off), hdrs.data(), n_blk * sizeof(InvHdr));
            if (!nodes.empty()) memcpy(A.at<uint8_t>(inv_off) + n_blk * sizeof(InvHdr), nodes.data(), nodes.size() * sizeof(NodeRec));
        }
        *A.at<SbDesc>(sbd_off + (size_t)s * sizeof(SbDesc)) = d;
    }
    A.alloc(64);  // tail guard
    timer.mark("tables (this thread)");
    worker();  // help with what is left, then wait for the others
    join_decoders();
    if (failed) return -8;
    timer.mark("cells (all cores)");
    h.total_bytes = blob.size();
    if (blob.size() >= ((uint64_t)1 << 35)) {
        err = "blob exceeds 32 GiB";
        return -8;
    }
    h.kind = 0;
    h.checksum = 0;
    *A.at<BlobHeader>(hdr_off) = h;
    A.at<BlobHeader>(hdr_off)->checksum = image_checksum(blob.data(), blob.size());
    timer.mark("checksum");
    return 0;
}

// checksum of a whole image: its body and its header (with the checksum field itself taken as zero)
uint64_t image_checksum(const uint8_t *blob, size_t len) {
    BlobHeader h;
    memcpy(&h, blob, sizeof h);
    h.checksum = 0;
    const uint64_t head = blob_checksum(reinterpret_cast<const uint8_t *>(&h), sizeof h);
    return blob_checksum(blob + sizeof(BlobHeader), len - sizeof(BlobHeader)) + head * 0x9E3779B97F4A7C15ull;
}

// Order-sensitive 64-bit checksum of an image body: sum over its 8-byte words of mix(word + index * K)
// (SplitMix64's finalizer), a sum so that threads can take slices.
uint64_t blob_checksum(const uint8_t *p, size_t len) {
    const size_t n_words = len / 8;
    auto slice = [p](size_t lo, size_t hi) {
        uint64_t acc = 0;
        for (size_t i = lo; i < hi; ++i) {
            uint64_t w;
            memcpy(&w, p + 8 * i, 8);
            uint64_t z = w + (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            acc += z ^ (z >> 31);
        }
        return acc;
    };
    unsigned n_threads = std::thread::hardware_concurrency();
    if (n_threads == 0) n_threads = 1;
    if (n_threads > 16) n_threads = 16;
    if (n_words < ((size_t)1 << 20)) n_threads = 1;
    std::vector<uint64_t> part(n_threads, 0);
    std::vector<std::thread> pool;
    const size_t per = (n_words + n_threads - 1) / n_threads;
    for (unsigned t = 0; t < n_threads; ++t) {
        const size_t lo = (size_t)t * per, hi = lo + per < n_words ? lo + per : n_words;
        if (lo >= hi) break;
        if (t + 1 == n_threads || hi == n_words) {
            part[t] = slice(lo, hi);
            break;
        }
        pool.emplace_back([&part, &slice, t, lo, hi]() { part[t] = slice(lo, hi); });
    }
    for (auto &t : pool) t.join();
    uint64_t acc = 0x5851F42D4C957F2Dull ^ (uint64_t)len;
    for (uint64_t v : part) acc += v;
    for (size_t i = n_words * 8; i < len; ++i) acc = acc * 0x100000001B3ull + p[i];  // tail bytes (images are 8-aligned)
    return acc;
}

// Structural validation of an image that did not come out of this process's flattener (fmx_attach_device_blob):
// every section and per-superblock table lies inside the image, the shapes agree with each other, every block
// header addresses bytes inside its superblock's arrays, and the body checksum matches — the same guarantee
// validate_model gives a serialized stream before any kernel may walk it.
int validate_blob(const uint8_t *b, size_t len, std::string &err) {
    auto bad = [&](const char *what) {
        err = std::string("device image fails validation: ") + what;
        return -3;
    };
    if (len < sizeof(BlobHeader) + 64 || (len & 7)) return bad("size");
    BlobHeader h;
    memcpy(&h, b, sizeof h);
    if (h.magic != kBlobMagic) return bad("magic");
    if (h.version != kBlobVersion) return bad("version");
    if (h.total_bytes != len) return bad("total_bytes");
    auto inside = [&](uint32_t off, uint64_t bytes) {
        const uint64_t lo = (uint64_t)off << 3;
        return lo >= sizeof(BlobHeader) && bytes <= len && lo <= len - bytes;
    };
    if (h.checksum != image_checksum(b, len)) return bad("checksum");
    // a vector in record form: shape, section inside the image, every record's offset pointer inside the stream behind the
    // records, the running one-counts what the classes say (rankOnes of the sampled-row bitmap indexes `suffixes`)
    auto records_ok = [&](const RrrDesc &r, bool count_ones) {
        if (r.length < 0 || r.n_blocks < 0 || r.n_rec != r.n_blocks / 16 + 1 || (int64_t)r.n_blocks < ((int64_t)r.length + 14) / 15 ||
            (int64_t)r.n_blocks > ((int64_t)r.length + 14) / 15 + 1)
            return false;
        const uint64_t rec_bytes = (uint64_t)r.n_rec * sizeof(RrrRecord) + 64;
        if (!inside(r.off_rec, rec_bytes + 16)) return false;
        const RrrRecord *rec = reinterpret_cast<const RrrRecord *>(b + ((uint64_t)r.off_rec << 3));
        const uint64_t avail_bits = (len - ((uint64_t)r.off_rec << 3)) * 8;
        const uint8_t *bits_needed = rrr_bits_needed();
        uint64_t ones = 0, obits = rec_bytes * 8;
        for (int32_t k = 0; k < r.n_rec; ++k) {
            if ((uint64_t)rec[k].offset_bit != obits || obits + 16 * 13 + 64 > avail_bits) return false;
            if (count_ones && rec[k].ones_before != ones) return false;
            for (int j = 0; j < 16; ++j) {
                const int cls = (int)((rec[k].classes >> (4 * j)) & 15);
                if ((int64_t)k * 16 + j < r.n_blocks) {
                    ones += (uint64_t)cls;
                    obits += bits_needed[cls];
                } else if (cls) {
                    return false;
                }
            }
        }
        return ones == (uint64_t)(uint32_t)r.total_ones;
    };
    if (h.kind == 1) {  // stand-alone RrrVector: value table + records + offsets stream
        if (!inside(h.off_inv, (uint64_t)kInvEntries * 2) || !records_ok(h.sampled, true)) return bad("RrrVector");
        return 0;
    }
    if (h.kind != 0) return bad("kind");
    if (h.sample_rate <= 0 || h.length <= 0 || h.wt_size != h.length) return bad("sampleRate / length");
    if (h.wt_sigma < 1 || h.wt_sigma > 32768 || h.n_c < 2 || h.n_look < 1) return bad("alphabet");
    if (h.n_sb != (int32_t)(((int64_t)h.length + (1 << 20) - 1) >> 20)) return bad("superblock count");
    if (h.bw_suffixes < 1 || h.bw_suffixes > 32 || (h.enable_extract && (h.bw_positions < 1 || h.bw_positions > 32)))
        return bad("bit widths");
    if (h.map_by_symbol != 0 && h.map_by_symbol != 1) return bad("map_by_symbol");
    if (h.compact != 0 && h.compact != 1) return bad("compact");
    if (h.compact && !inside(h.off_inv, (uint64_t)kInvEntries * 2)) return bad("value-of-offset table");
    const int64_t n_samples = (int64_t)h.length / h.sample_rate;
    if (h.n_suffixes < n_samples + 1 || (h.enable_extract && h.n_positions < n_samples + 2)) return bad("sample counts");
    auto packed_bytes = [](int64_t n, int width) { return (uint64_t)(words_for_bits(n * width) + 2) * 8; };
    if (!inside(h.off_c, (uint64_t)h.n_c * 4) || !inside(h.off_lookup, (uint64_t)std::max(h.n_look, h.wt_sigma + 1) * 4) ||
        !inside(h.off_char2code, 65536 * 2) || !inside(h.off_suffixes, packed_bytes(h.n_suffixes, h.bw_suffixes)) ||
        (h.enable_extract && !inside(h.off_positions, packed_bytes(h.n_positions, h.bw_positions))) ||
        !inside(h.off_sbc, (uint64_t)(h.n_sb + 1) * h.wt_sigma * sizeof(SbcEntry)) ||
        !inside(h.off_sbdesc, (uint64_t)h.n_sb * sizeof(SbDesc)))
        return bad("section outside the image");
    const int16_t *c2c = reinterpret_cast<const int16_t *>(b + ((uint64_t)h.off_char2code << 3));
    for (int i = 0; i < 65536; ++i)
        if (c2c[i] < 0 || c2c[i] >= h.wt_sigma || c2c[i] + 1 >= h.n_c || c2c[i] >= h.n_look) return bad("char -> code table");
    const int32_t *C = reinterpret_cast<const int32_t *>(b + ((uint64_t)h.off_c << 3));
    for (int i = 0; i < h.n_c; ++i)
        if (C[i] < 0 || C[i] > h.length) return bad("cumulativeCounts");
    auto cells_ok = [&](const RrrDesc &r) {
        if (h.compact) return records_ok(r, true);
        return r.length >= 0 && r.n_rec == (int32_t)((int64_t)r.length / kBvCellBits + 2) &&
               inside(r.off_rec, (uint64_t)r.n_rec * sizeof(BvCell));
    };
    if (h.sampled.length != h.length || !cells_ok(h.sampled)) return bad("sampled-row bitmap");
    if (h.compact) {
        if (h.sampled.total_ones < 1 || h.sampled.total_ones > h.n_suffixes) return bad("sampled rows vs suffix samples");
    } else {  // rankOnes over this bitmap indexes `suffixes` (FM:541): its running counts must be what its bits say
        const BvCell *cells = reinterpret_cast<const BvCell *>(b + ((uint64_t)h.sampled.off_rec << 3));
        uint64_t ones = 0;
        for (int32_t c = 0; c < h.sampled.n_rec; ++c) {
            if (cells[c].ones_before != ones) return bad("sampled-row bitmap counts");
            for (int k = 0; k < 3; ++k) ones += (uint64_t)__builtin_popcount(cells[c].bits[k]);
        }
        if (ones != (uint64_t)(uint32_t)h.sampled.total_ones || h.sampled.total_ones < 1 || h.sampled.total_ones > h.n_suffixes)
            return bad("sampled rows vs suffix samples");
    }
    if (h.enable_extract) {  // inverse samples are SA rows: the walks start there (FM:579-587)
        const uint32_t *pw = reinterpret_cast<const uint32_t *>(b + ((uint64_t)h.off_positions << 3));
        for (int64_t i = 0; i < h.n_positions; ++i)
            if ((int64_t)ld_bits(pw, (uint64_t)i * (uint32_t)h.bw_positions, h.bw_positions) >= h.length)
                return bad("inverse sample outside the text");
    }
    const SbcEntry *sbc = reinterpret_cast<const SbcEntry *>(b + ((uint64_t)h.off_sbc << 3));
    const SbDesc *sbd = reinterpret_cast<const SbDesc *>(b + ((uint64_t)h.off_sbdesc << 3));
    for (int32_t s = 0; s < h.n_sb; ++s) {
        const SbDesc &d = sbd[s];
        if (d.bsl < 0 || d.bsl > 20 || d.sigma < -1 || d.sigma >= h.wt_sigma) return bad("superblock header");
        const int64_t per_row = (int64_t)1 << (20 - d.bsl);
        const int64_t symbols = (int64_t)h.length - ((int64_t)s << 20) < (1 << 20) ? (int64_t)h.length - ((int64_t)s << 20) : (1 << 20);
        const int64_t n_blocks = (symbols + ((int64_t)1 << d.bsl) - 1) >> d.bsl;
        if (d.n_blocks != n_blocks || d.n_blocks > per_row) return bad("block count");
        const int64_t rows = h.map_by_symbol ? h.wt_sigma : (int64_t)d.sigma + 1;
        if (d.mapping_len != rows * per_row || d.path_len < 0 || d.var_len < 0) return bad("mapping shape");
        if (!inside(d.off_mapping, (uint64_t)d.mapping_len * sizeof(MapEntry) + (uint64_t)d.path_len * sizeof(PathRec) + 32) ||
            !inside(d.off_bh, (uint64_t)d.n_blocks * sizeof(BlockHdr) + 16) || !inside(d.off_var, (uint64_t)d.var_len + 16) ||
            !cells_ok(d.rrr) || d.rrr.length >= (1 << 24) || d.rrr.node_len < d.n_blocks ||
            !inside(d.rrr.off_bits, (uint64_t)d.rrr.node_len * 16 + 64))
            return bad("superblock table outside the image");
        for (int c = 0; c < h.wt_sigma; ++c) {
            const int16_t code = sbc[(int64_t)s * h.wt_sigma + c].sbc;
            if (code < 0 || (code > d.sigma && code != h.wt_sigma - 1)) return bad("superblock code");
        }
        const BlockHdr *bh = reinterpret_cast<const BlockHdr *>(b + ((uint64_t)d.off_bh << 3));
        for (int32_t k = 0; k < d.n_blocks; ++k) {
            const int32_t height = bh[k].tree_height, leaves = (int32_t)bh[k].sigma + 1;
            if (height < 0 || height > 30 || leaves < 1 || bh[k].var_off < 0) return bad("block header");
            const int64_t need = (height > 1 ? (int64_t)(height - 1) * 4 : 0) + (int64_t)leaves * 5 + (int64_t)(leaves - 1) * 2;
            if ((int64_t)bh[k].var_off + need > d.var_len) return bad("block header outside the byte array");
            if ((bh[k].bv_offset & 0xffffff) > d.rrr.length) return bad("block bit-vector offset");
        }
        const MapEntry *me = reinterpret_cast<const MapEntry *>(b + ((uint64_t)d.off_mapping << 3));
        for (int64_t row = 0; row < rows; ++row)
            for (int64_t blk = 0; blk < per_row; ++blk) {
                const MapEntry &e = me[row * per_row + blk];
                const uint32_t tag = e.x & 0xffu, value = e.x >> 8;
                if (tag == kMapAbsent) {
                    if (value < 1 || blk + (int64_t)value > per_row) return bad("skip pointer");
                    if (blk + (int64_t)value < per_row && (me[row * per_row + blk + value].x & 0xffu) == kMapAbsent)
                        return bad("skip pointer onto an absent entry");
                } else if (blk >= d.n_blocks) {
                    return bad("mapping entry beyond the last block");
                } else if (tag == kMapSlow) {
                    if (value > 0x7fffu) return bad("leaf index");
                } else if (tag <= kMapMaxLen) {
                    if ((e.y & 0xffffffu) > (uint32_t)d.rrr.length) return bad("root node position");
                    if (tag > 1) {
                        const uint64_t first = e.w, last = first + (tag - 1);
                        if (first < 2 * (uint64_t)d.mapping_len || last > 2 * (uint64_t)d.mapping_len + (uint64_t)d.path_len)
                            return bad("path records outside the table");
                    }
                } else {
                    return bad("mapping tag");
                }
            }
        // inverseSelect section: every walk stays inside its block's records and ends (children lie behind parents)
        const InvHdr *ih = reinterpret_cast<const InvHdr *>(b + ((uint64_t)d.rrr.off_bits << 3));
        for (int32_t k = 0; k < d.n_blocks; ++k) {
            const InvHdr &q = ih[k];
            if (q.x & kInvSlow) continue;
            if (q.x & kInvRun) {
                if (bh[k].tree_height != 0 || q.y >= (uint32_t)h.wt_sigma) return bad("run block record");
                continue;
            }
            if (bh[k].tree_height < 1 || (q.x >> 24) || (q.x & 0xffffffu) > (uint32_t)d.rrr.length || (q.y >> 24))
                return bad("root record");
            if (q.w < 1 || q.z < (uint32_t)d.n_blocks || (uint64_t)q.z + q.w > (uint64_t)d.rrr.node_len)
                return bad("node records outside the section");
            const NodeRec *nr = reinterpret_cast<const NodeRec *>(ih) + q.z;
            for (uint32_t n = 0; n < q.w; ++n)
                for (int c = 0; c < 2; ++c) {
                    const uint64_t half = nr[n].child[c];
                    const uint32_t idx = (uint32_t)(half & 0xffffu);
                    if (idx == 0) {
                        if (((half >> 16) & 0xffffu) >= (uint64_t)h.wt_sigma) return bad("leaf symbol");
                    } else if (idx <= n || idx >= q.w || ((half >> 16) & 0xffffffu) > (uint64_t)d.rrr.length) {
                        return bad("node record");
                    }
                }
        }
        const PathRec *pr = reinterpret_cast<const PathRec *>(me + d.mapping_len);
        for (int32_t k = 0; k < d.path_len; ++k)
            if (pr[k].a > (uint32_t)d.rrr.length || (pr[k].b >> 24)) return bad("path record");
    }
    return 0;
}

}  // namespace fmx
