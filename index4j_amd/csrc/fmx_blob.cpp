// fmx_blob.cpp — flatten the host model into the HBM image described in fmx_blob.hpp.
#include "fmx_blob.hpp"
#include "fmx_model.hpp"

#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>

namespace fmx {
namespace {

struct Arena {
    std::vector<uint8_t> &b;
    // reserve `bytes` at a 64-byte boundary, zero-filled; returns the byte offset
    size_t alloc(size_t bytes) {
        size_t off = (b.size() + 63) & ~(size_t)63;
        b.resize(off + bytes, 0);
        return off;
    }
    template <typename T>
    T *at(size_t off) {
        return reinterpret_cast<T *>(b.data() + off);
    }
};

inline uint32_t off8(size_t byte_off) { return (uint32_t)(byte_off >> 3); }

// RRR:92-103 -> 16-block records + offsets bit stream
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
    d.pad = 0;
    const size_t rec_off = A.alloc((size_t)n_rec * sizeof(RrrRecord) + 64);
    d.off_rec = off8(rec_off);
    uint64_t ones = 0, obits = 0;
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
    if (obits > (uint64_t)r.offsets.size() * 64 || obits > 0xffffffffull) {
        err = "RRR offsets stream shorter than its classes imply";
        return false;
    }
    const size_t bits_off = A.alloc((r.offsets.size() + 2) * 8);
    d.off_bits = off8(bits_off);
    if (!r.offsets.empty()) memcpy(A.at<uint8_t>(bits_off), r.offsets.data(), r.offsets.size() * 8);
    return true;
}

// RRR:92-103 -> expanded 96-bit cells with running one-counts (see fmx_blob.hpp).  Two steps so that the
// decoding of many vectors can run on many threads once all regions are allocated: reserve, then fill.
struct ExpandJob {
    const RrrModel *r;
    size_t cell_off;
    int64_t n_cells;
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
    d.pad = 0;
    d.off_bits = 0;
    job.r = &r;
    job.n_cells = n_cells;
    job.cell_off = A.alloc((size_t)n_cells * sizeof(BvCell));
    d.off_rec = off8(job.cell_off);
    return true;
}
bool expanded_fill(BvCell *cells, const ExpandJob &job, std::string &err) {
    const RrrModel &r = *job.r;
    const uint8_t *bits_needed = rrr_bits_needed();
    const uint16_t *value_of = rrr_value_of_offset(), *class_base = rrr_class_base();
    const int64_t n_blocks = r.classes.length;
    // RRR:382-390 for every block: (class, offset) -> 15 bits, into a plain LSB-first bit array ...
    std::vector<uint64_t> plain((size_t)(job.n_cells * kBvCellBits / 64 + 2), 0);
    uint64_t obits = 0;
    const uint64_t avail = (uint64_t)r.offsets.size() * 64;
    for (int64_t b = 0; b < n_blocks; ++b) {
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
        const uint64_t value = value_of[(size_t)class_base[cls] + (size_t)off];
        const int64_t pos = b * 15;
        if (pos + 15 > job.n_cells * (int64_t)kBvCellBits) {
            err = "RRR vector has more blocks than its length";
            return false;
        }
        const size_t pw = (size_t)(pos >> 6);
        const int ps = (int)(pos & 63);
        plain[pw] |= value << ps;
        if (ps + 15 > 64) plain[pw + 1] |= value >> (64 - ps);
    }
    // ... cut into 96-bit cells with running one-counts (bits past `length` are zero: RRR pads its last block)
    const uint32_t *plain32 = reinterpret_cast<const uint32_t *>(plain.data());
    uint64_t ones = 0;
    for (int64_t c = 0; c < job.n_cells; ++c) {
        BvCell cell;
        cell.ones_before = (uint32_t)ones;
        for (int k = 0; k < 3; ++k) {
            cell.bits[k] = plain32[(size_t)c * 3 + (size_t)k];
            ones += (uint64_t)__builtin_popcount(cell.bits[k]);
        }
        cells[c] = cell;
    }
    if (ones != (uint64_t)(uint32_t)r.total_ones) {
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

}  // namespace

// a stand-alone RrrVector (fmx_rrr_build): header + value-of-offset table + the vector in its compressed form
int flatten_rrr_only(const RrrModel &r, std::vector<uint8_t> &blob, std::string &err) {
    blob.clear();
    Arena A{blob};
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
    memcpy(A.at<uint8_t>(hdr_off), &h, sizeof h);
    return 0;
}

// -1 = by alphabet size; 0 / 1 force the row layout of the mapping tables (tests exercise both)
static int g_map_by_symbol = -1;
void set_map_by_symbol(int mode) { g_map_by_symbol = mode; }

int flatten_model(const FmModel &m, std::vector<uint8_t> &blob, std::string &err) {
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
    Arena A{blob};
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
    off = A.alloc(m.look_up.size() * 4 + 8);
    h.off_lookup = off8(off);
    memcpy(A.at<uint8_t>(off), m.look_up.data(), m.look_up.size() * 4);
    off = A.alloc(65536 * 2);
    h.off_char2code = off8(off);
    for (size_t i = 0; i < m.map_keys.size(); ++i)
        if (m.map_keys[i] >= 0 && m.map_keys[i] < 65536) A.at<int16_t>(off)[m.map_keys[i]] = m.map_vals[i];
    h.off_suffixes = off8(put_packed(A, m.suffixes));
    h.off_positions = m.enable_extract ? off8(put_packed(A, m.positions)) : 0;
    std::vector<ExpandJob> jobs((size_t)n_sb + 1);
    if (!expanded_reserve(A, m.sampled, h.sampled, jobs[(size_t)n_sb], err)) return -8;

    h.off_inv = 0;  // no compressed RRR vector in an FM-index image (all bit vectors are expanded)

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

    // mapping rows by global symbol (rows of symbols a superblock does not hold are pure skip pointers)
    bool by_symbol = g_map_by_symbol > 0;
    if (g_map_by_symbol < 0) {  // automatic: rows by symbol unless that more than doubles the tables
        int64_t rows_by_code = 0, rows_by_symbol = 0;
        for (int64_t s = 0; s < n_sb; ++s) {
            const int bsl = w.sb[(size_t)s].block_size_log;
            if (bsl < 0 || bsl > 20) continue;
            rows_by_code += ((int64_t)w.sb[(size_t)s].sigma + 1) << (20 - bsl);
            rows_by_symbol += (int64_t)sigma << (20 - bsl);
        }
        by_symbol = rows_by_symbol <= 2 * rows_by_code;
    }
    h.map_by_symbol = by_symbol ? 1 : 0;
    const size_t sbd_off = A.alloc((size_t)n_sb * sizeof(SbDesc));
    h.off_sbdesc = off8(sbd_off);
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
        off = A.alloc((size_t)(n_rows * per_row) * sizeof(MapEntry) + 16);
        d.off_mapping = off8(off);
        d.mapping_len = (int32_t)(n_rows * per_row);
        {
            // absent entries (alphabetSize - 1, WFBB:383-387) become skip pointers: -(distance to the next
            // block to the right holding the symbol, or to the end of the superblock)
            MapEntry *dst = A.at<MapEntry>(off);
            const int16_t absent = (int16_t)(sigma - 1);
            for (int64_t row = 0; row < n_rows; ++row) {
                // the reference's row of this device row: by superblock code (WFBB:461-465)
                const int64_t src_row = by_symbol ? (int64_t)w.global_mapping[(size_t)(s * sigma + row)] : row;
                const bool in_superblock = src_row >= 0 && src_row <= sb.sigma;
                int64_t next_present = per_row;
                for (int64_t blk = per_row - 1; blk >= 0; --blk) {
                    const int16_t v = in_superblock ? sb.mapping[(size_t)(src_row * per_row + blk)] : absent;
                    int16_t raw;
                    if (v != absent) {
                        if (v < 0) {
                            err = "negative mapping entry";
                            return -3;
                        }
                        next_present = blk;
                        raw = v;
                    } else {
                        raw = (int16_t)(-(next_present - blk));
                    }
                    // until the block's header has been read (below) a present entry takes the reference's route
                    dst[row * per_row + blk] = MapEntry{(uint32_t)(uint16_t)raw | (kMapSlow << 16), 0, 0, 0};
                }
            }
            // what rank() needs about every (symbol, block) that occurs, from the block's own header
            const uint8_t *var = sb.var.data();
            const int64_t var_len = (int64_t)sb.var.size();
            for (int64_t blk = 0; blk < (int64_t)sb.block_headers.size() && blk < per_row; ++blk) {
                const BlockHeader &bh = sb.block_headers[(size_t)blk];
                const int h = bh.tree_height, n_leaves = (int)bh.sigma + 1;
                if (h < 0 || n_leaves <= 0 || bh.var_off < 0) continue;
                const int64_t hdr = bh.var_off, leaves = hdr + (h > 0 ? (int64_t)(h - 1) * 4 : 0);
                const int64_t second0 = (int64_t)(h - 1) * 4 + (int64_t)n_leaves * 5;
                if (leaves + (int64_t)n_leaves * 5 > var_len || (h > 0 && hdr + second0 + 2 > var_len)) continue;
                const uint32_t counts0 = h > 0 ? (uint32_t)var[hdr + second0] | ((uint32_t)var[hdr + second0 + 1] << 8) : 0u;
                for (int i = 0; i < n_leaves; ++i) {
                    const uint8_t *lp = var + leaves + (int64_t)i * 5;
                    const int symbol = (int)lp[0] | ((int)lp[1] << 8);
                    const uint32_t rank_block = (uint32_t)lp[2] | ((uint32_t)lp[3] << 8) | ((uint32_t)lp[4] << 16);
                    if (symbol >= sigma) continue;
                    const int code_row = w.global_mapping[(size_t)(s * sigma + symbol)];
                    if (code_row < 0 || code_row > sb.sigma) continue;
                    MapEntry &e = dst[(int64_t)(by_symbol ? symbol : code_row) * per_row + blk];
                    if ((int16_t)(e.x & 0xffffu) < 0) continue;  // the mapping says absent: leave it to the reference's route
                    // WFBB:250-278 restoreCodeFromBlockHeader for leaf i
                    uint32_t code = 0, leaf_count = 0;
                    int len = h > 0 ? 1 : 0;
                    for (int lvl = 0; len < h && len > 0; ++lvl) {
                        const uint32_t level_leaves = (uint32_t)var[hdr + 4 * lvl] | ((uint32_t)var[hdr + 4 * lvl + 1] << 8);
                        code <<= 1;
                        if (leaf_count + level_leaves > (uint32_t)i) {
                            code += (uint32_t)i - leaf_count;
                            break;
                        }
                        code += level_leaves;
                        ++len;
                        leaf_count += level_leaves;
                    }
                    if (h > 0 && len == h) {
                        code <<= 1;
                        code += (uint32_t)i - leaf_count;
                    }
                    if (len > 16 || (code >> 16)) continue;  // stays on the reference's route
                    e.x = (e.x & 0xffffu) | ((uint32_t)len << 16) | ((counts0 >> 8) << 24);
                    e.y = rank_block | ((counts0 & 0xffu) << 24);
                    e.z = (uint32_t)bh.bv_offset | ((code & 0xffu) << 24);
                    e.w = (uint32_t)bh.bv_rank | ((code >> 8) << 24);
                }
            }
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
        if (!expanded_reserve(A, sb.rank_support, d.rrr, jobs[(size_t)s], err)) return -8;
        {
            // leaf section for inverseSelect: per leaf {symbol as reported (run blocks: masked to 8 bits, WFBB:1332),
            // folded superblock rank of that symbol + the leaf's rank at block start}
            const size_t leaf_off = A.alloc(4 * sb.var.size() + 64);
            d.rrr.off_bits = off8(leaf_off);
            uint8_t *dst = A.at<uint8_t>(leaf_off);
            const SbcEntry *sbc_row = A.at<SbcEntry>((size_t)h.off_sbc << 3) + (size_t)s * (size_t)sigma;
            const uint8_t *var = sb.var.data();
            const int64_t var_len = (int64_t)sb.var.size();
            for (size_t b = 0; b < sb.block_headers.size(); ++b) {
                const BlockHeader &bh = sb.block_headers[b];
                const int hgt = bh.tree_height, n_leaves = (int)bh.sigma + 1;
                if (hgt < 0 || n_leaves <= 0 || bh.var_off < 0) continue;
                const int64_t leaves = (int64_t)bh.var_off + (hgt > 0 ? (int64_t)(hgt - 1) * 4 : 0);
                if (leaves + (int64_t)n_leaves * 5 > var_len) continue;
                if (4 * (int64_t)bh.var_off + 8 * (int64_t)n_leaves > 4 * var_len + 56) continue;
                for (int i = 0; i < n_leaves; ++i) {
                    const uint8_t *lp = var + leaves + (int64_t)i * 5;
                    int symbol = (int)lp[0] | ((int)lp[1] << 8);
                    if (hgt == 0) symbol &= 0xff;  // Q1
                    const uint32_t rank_block = (uint32_t)lp[2] | ((uint32_t)lp[3] << 8) | ((uint32_t)lp[4] << 16);
                    const uint32_t folded = (symbol < sigma ? (uint32_t)sbc_row[symbol].rank : 0u) + rank_block;
                    const uint64_t v = ((uint64_t)folded << 32) | (uint64_t)(uint32_t)symbol;
                    memcpy(dst + 4 * (size_t)bh.var_off + 8 * (size_t)i, &v, 8);
                }
            }
        }
        *A.at<SbDesc>(sbd_off + (size_t)s * sizeof(SbDesc)) = d;
    }
    A.alloc(64);  // tail guard
    {
        // all regions exist (the arena no longer moves): decode the bit vectors on all cores
        std::atomic<size_t> next{0};
        std::mutex err_mutex;
        bool failed = false;
        auto worker = [&]() {
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= jobs.size()) return;
                std::string local;
                if (!expanded_fill(A.at<BvCell>(jobs[j].cell_off), jobs[j], local)) {
                    std::lock_guard<std::mutex> lock(err_mutex);
                    failed = true;
                    err = local;
                }
            }
        };
        unsigned n_threads = std::thread::hardware_concurrency();
        if (n_threads == 0) n_threads = 1;
        if (n_threads > jobs.size()) n_threads = (unsigned)jobs.size();
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
        if (failed) return -8;
    }
    h.total_bytes = blob.size();
    if (blob.size() >= ((uint64_t)1 << 35)) {
        err = "blob exceeds 32 GiB";
        return -8;
    }
    *A.at<BlobHeader>(hdr_off) = h;
    return 0;
}

}  // namespace fmx
