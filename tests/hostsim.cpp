// hostsim.cpp — TEST-ONLY host simulation of the device code.
//
// Compiles index4j_amd/csrc/fmx_device.hpp with g++ (its FMX_HD functions become plain inline C++)
// and runs them over a HOST copy of the blob, one query at a time.  It exists so that the exact
// source the GPU executes can be checked against the oracle on machines without a GPU (this build
// container).  It is never linked into libfmx.so and never used by the package: the product has no
// CPU query path.
#include "../index4j_amd/csrc/fmx_device.hpp"

#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

using namespace fmx;

// window directories attached to blobs (sim_win_attach): what fmx_to_device grows beside a resident image
static std::map<const uint8_t *, std::vector<uint32_t>> g_windows;
static std::map<const uint8_t *, std::vector<uint16_t>> g_window_entries;
static std::map<const uint8_t *, uint64_t> g_window_unclean;
static std::map<const uint8_t *, std::vector<uint64_t>> g_window_full;  // four-byte entries: the answers that are more than a row
static std::map<const uint8_t *, int> g_window_entry4;
static std::map<const uint8_t *, int> g_window_flat;  // the directory of this blob is the flat form (sim_set_entry_bytes(-1))
static std::map<const uint8_t *, std::vector<uint16_t>> g_window_lut;  // ... and where a symbol search starts (the kernels make it in LDS)
static int g_entry_bytes = 0;  // sim_set_entry_bytes: 0 = by the alphabet (as fmx_to_device), 4, 6
static int g_pack = 1;  // locate over a window directory: the instalment form (k_locate_walk_c) or fm_locate_hit<kWinAlways> (option walk_pack 0)

static DevIndex make_index(const uint8_t *b) {
    BlobHeader h;
    memcpy(&h, b, sizeof h);
    DevIndex d;
    auto at = [&](uint32_t off) { return b + ((uint64_t)off << 3); };
    d.base = b;
    d.C = reinterpret_cast<const int32_t *>(at(h.off_c));
    d.look_up = reinterpret_cast<const int32_t *>(at(h.off_lookup));
    d.char2code = reinterpret_cast<const int16_t *>(at(h.off_char2code));
    d.suffix_words = reinterpret_cast<const uint32_t *>(at(h.off_suffixes));
    d.pos_words = reinterpret_cast<const uint32_t *>(at(h.off_positions));
    d.sbc = reinterpret_cast<const SbcEntry *>(at(h.off_sbc));
    d.sbd = reinterpret_cast<const SbDesc *>(at(h.off_sbdesc));
    d.inv_global = reinterpret_cast<const uint16_t *>(at(h.off_inv));
    d.sampled = h.sampled;
    d.length = h.length;
    d.sample_rate = h.sample_rate;
    d.enable_extract = h.enable_extract;
    d.wt_sigma = h.wt_sigma;
    d.n_sb = h.n_sb;
    d.bw_suffixes = h.bw_suffixes;
    d.bw_positions = h.bw_positions;
    d.n_positions = h.n_positions;
    d.n_c = h.n_c;
    d.map_by_symbol = h.map_by_symbol;
    d.sb_cache = nullptr;
    d.sb_cache_limit = 0;
    d.wt_size = (uint32_t)h.wt_size;
    d.suffix_table = nullptr;
    {
        auto it = g_windows.find(b);
        d.win = it == g_windows.end() ? nullptr : reinterpret_cast<const Quad *>(it->second.data());
        auto e = g_window_entries.find(b);
        d.win_other = e == g_window_entries.end() ? nullptr : e->second.data();
        auto f = g_window_full.find(b);
        d.win_full = f == g_window_full.end() ? nullptr : f->second.data();
        auto k = g_window_entry4.find(b);
        d.win_entry4 = k == g_window_entry4.end() ? 0 : k->second;
        auto fl = g_window_flat.find(b);
        d.win_flat = fl == g_window_flat.end() ? 0 : fl->second;
        d.c_lds = nullptr;
        auto l = g_window_lut.find(b);
        d.c_lut = l == g_window_lut.end() ? nullptr : l->second.data();
        d.c_lut_shift = win_lut_shift(h.length);
    }
    d.self = nullptr;
    d.suffix_chars = 0;
    d.suffix_key_bits = h.wt_sigma <= 256 ? 8 : 16;
    d.suffix_shift = 0;
    d.suffix_mask = 0;
    return d;
}

extern "C" {

// grows the window directory of a blob with the very function k_win_build runs (win_build_cell) and attaches it: every sim_*
// call on that blob then takes the windows first, as the kernels do on a resident index.  Returns the number of cells;
// stats (nullable, 6 slots): {positions with a class, positions, classes in use, positions with an entry, entries with a status or
// suspect, four-byte form: entries that point at an eight-byte slot (-1: six-byte form)}.
int64_t sim_win_attach(const uint8_t *blob, int64_t *stats) {
    g_windows.erase(blob);
    g_window_entries.erase(blob);
    g_window_full.erase(blob);
    g_window_entry4.erase(blob);
    g_window_lut.erase(blob);
    g_window_flat.erase(blob);
    DevIndex ix = make_index(blob);  // (no directory: it is made from the tree walk's own answers)
    if (g_entry_bytes == -1) {  // the flat form (fmx_to_device under window_cells = 3): a word per position, win_build_flat
        std::vector<uint32_t> flat((size_t)ix.wt_size + 4);
        std::vector<uint64_t> full((size_t)ix.wt_size + 1);
        uint32_t full_count = 0;
        uint64_t open_entries = 0;
        for (uint32_t p = 0; p < ix.wt_size; ++p)
            open_entries += win_build_flat(ix, p, flat.data(), full.data(), (uint32_t)full.size(), &full_count) & 0x7fffffffu;
        g_window_unclean[blob] = open_entries;
        if (stats) {
            stats[0] = stats[2] = 0;
            stats[1] = stats[3] = (int64_t)ix.wt_size;
            stats[4] = (int64_t)open_entries;
            stats[5] = (int64_t)full_count;
        }
        std::vector<uint16_t> lut(kWinLutBuckets + 2);
        const int32_t shift = win_lut_shift(ix.length);
        for (int32_t b = 0; b <= kWinLutBuckets; ++b) {
            const int64_t row = (int64_t)b << shift;
            lut[(size_t)b] = (uint16_t)win_symbol_of_row(ix, row > 0x7fffffff ? 0x7fffffff : (int32_t)row);
        }
        g_windows[blob] = std::move(flat);
        g_window_entries[blob] = std::vector<uint16_t>(4);
        g_window_full[blob] = std::move(full);
        g_window_entry4[blob] = 1;
        g_window_flat[blob] = 1;
        g_window_lut[blob] = std::move(lut);
        return (int64_t)ix.wt_size;
    }
    const size_t cells = win_cells_for(ix.wt_size);
    std::vector<uint32_t> words(cells * 16 + 4), first(cells + 1);
    uint64_t total = 0;
    for (size_t w = 0; w < cells; ++w) {
        first[w] = (uint32_t)total;
        total += win_build_cell(ix, (uint32_t)w, words.data() + 16 * w);
    }
    std::vector<uint16_t> entries((total + 1) * kWinEntryWords);
    const bool entry4 = g_entry_bytes == 4 || (g_entry_bytes == 0 && ix.n_c <= kWinSymbolSearchMax);
    std::vector<uint64_t> full(entry4 ? total + 1 : 1);  // (room for every entry: the host simulation never runs out of slots)
    uint32_t full_count = 0;
    uint64_t open_entries = 0;
    for (size_t w = 0; w < cells; ++w)
        open_entries += win_build_other(ix, (uint32_t)w, words.data() + 16 * w, first[w], entries.data(), entry4, full.data(),
                                        (uint32_t)full.size(), &full_count) & 0x7fffffffu;
    g_window_unclean[blob] = open_entries;
    g_window_entry4[blob] = entry4 ? 1 : 0;
    if (stats) stats[5] = entry4 ? (int64_t)full_count : -1;
    if (stats) {
        stats[0] = stats[1] = stats[2] = stats[3] = 0;
        stats[4] = (int64_t)open_entries;
        for (size_t w = 0; w < cells; ++w) {
            const uint32_t *c = words.data() + 16 * w;
            WinCell cell;
            memcpy(&cell, c, 64);
            const uint32_t ids[3] = {c[3] & 0xffffu, c[3] >> 16, c[5] & 0xffffu};
            for (int k = 0; k < 3; ++k) stats[2] += ids[k] != kWinNone;
            for (uint32_t r = 0; r < kWinW && (uint64_t)w * kWinW + r < ix.wt_size; ++r) {
                int32_t sym, rank;
                bool sampled;
                uint32_t other = 0;
                if (win_inv_from(cell, r, sym, rank, sampled, other))
                    ++stats[0];
                else
                    ++stats[3];
                ++stats[1];
            }
        }
    }
    g_windows[blob] = std::move(words);
    g_window_entries[blob] = std::move(entries);
    g_window_full[blob] = std::move(full);
    if (entry4) {  // as stage_c_lds does per workgroup
        std::vector<uint16_t> lut(kWinLutBuckets + 2);
        const int32_t shift = win_lut_shift(ix.length);
        for (int32_t b = 0; b <= kWinLutBuckets; ++b) {
            const int64_t row = (int64_t)b << shift;
            lut[(size_t)b] = (uint16_t)win_symbol_of_row(ix, row > 0x7fffffff ? 0x7fffffff : (int32_t)row);
        }
        g_window_lut[blob] = std::move(lut);
    }
    return (int64_t)cells;
}
void sim_set_pack(int on) { g_pack = on; }
void sim_set_entry_bytes(int bytes) { g_entry_bytes = bytes; }
void sim_win_detach(const uint8_t *blob) {
    g_windows.erase(blob);
    g_window_entries.erase(blob);
    g_window_unclean.erase(blob);
    g_window_full.erase(blob);
    g_window_entry4.erase(blob);
    g_window_lut.erase(blob);
    g_window_flat.erase(blob);
}

int32_t sim_wt_rank(const uint8_t *blob, uint32_t position, int32_t symbol, int32_t *status) {
    DevIndex ix = make_index(blob);
    int st = 0;
    int32_t r = wt_rank(ix, ix.inv_global, position, symbol, st);
    *status = st;
    return r;
}

int32_t sim_wt_inverse_select(const uint8_t *blob, uint32_t position, int32_t *rank) {
    DevIndex ix = make_index(blob);
    return wt_inverse_select(ix, ix.inv_global, position, *rank);
}

// fused LF-step (fm_lf_step) next to the reference's literal two-call form, for tests/test_fused_lf.py
void sim_lf_step_both(const uint8_t *blob, int32_t row, int32_t *out /* fused_row, fused_c, ref_row, ref_c, fused_st, ref_st */) {
    DevIndex ix = make_index(blob);
    int st = 0, c = 0;
    out[0] = fm_lf_step(ix, ix.inv_global, row, c, st);
    out[1] = c;
    out[4] = st;
    int st2 = 0;
    int32_t unused;
    const int32_t c2 = (int32_t)(int16_t)wt_inverse_select(ix, ix.inv_global, (uint32_t)(row - 1), unused);  // FM:532-533
    out[2] = ix.C[c2] + wt_rank(ix, ix.inv_global, (uint32_t)row, c2, st2);                                   // FM:534-535
    out[3] = c2;
    out[5] = st2;
}

// mirrors k_count with both roles of a pair evaluated in turn
void sim_count(const uint8_t *blob, const uint16_t *pat, const int32_t *off, int32_t n, int32_t *counts,
               int32_t *lf, int32_t *status_out, int32_t *range) {
    DevIndex ix = make_index(blob);
    for (int32_t p = 0; p < n; ++p) {
        const int32_t beg = off[p], m = off[p + 1] - beg;
        int status = ST_OK;
        int32_t start = 0, end = 0, steps = 0;
        if (m <= 0) {
            status = ST_JAVA_AIOOBE;
        } else {
            int32_t i = m - 1;
            int32_t c = fm_map(ix, pat[beg + i]);
            if (c != 0) {
                start = ix.C[c];
                end = ix.C[c + 1];
                while (start < end && i >= 1) {
                    c = fm_map(ix, pat[beg + --i]);
                    if (c == 0) {
                        start = end = 0;
                        break;
                    }
                    const int32_t s2 = ix.C[c] + wt_rank(ix, ix.inv_global, (uint32_t)start, c, status);
                    const int32_t e2 = ix.C[c] + wt_rank(ix, ix.inv_global, (uint32_t)end, c, status);
                    start = s2;
                    end = e2;
                    steps += 2;
                }
            }
        }
        const int32_t d = end - start;
        counts[p] = d > 0 ? d : 0;
        if (lf) lf[p] = steps;
        if (status_out) status_out[p] = status;
        if (range) {
            range[2 * p] = start;
            range[2 * p + 1] = end;
        }
    }
}

// mirrors k_count with a suffix table of up to `chars` characters: the table is grown level by level with fm_suffix_extend
// (what k_suffix_level1 / k_suffix_expand run), every level of 2 .. chars codes hashed as k_suffix_insert does, and consulted
// by fm_suffix_key + fm_suffix_lookup (what k_count runs per pattern: a pattern shorter than `chars` looks its whole self
// up).  Returns the number of strings in the table.
int64_t sim_count_table(const uint8_t *blob, int32_t chars, const uint16_t *pat, const int32_t *off, int32_t n,
                        int32_t *counts, int32_t *lf, int32_t *status_out, int64_t *answered_steps) {
    DevIndex ix = make_index(blob);
    const int key_bits = ix.wt_sigma <= 256 ? 8 : 16;
    std::vector<SuffixSlot> level;
    std::vector<std::vector<SuffixSlot>> levels;  // levels[i]: strings of i + 2 codes
    for (int32_t c = 1; c + 1 < ix.n_c && c < ix.wt_sigma; ++c)
        if (ix.C[c] < ix.C[c + 1]) level.push_back(SuffixSlot{(uint64_t)(uint32_t)c, (uint32_t)ix.C[c], (uint32_t)ix.C[c + 1]});
    for (int depth = 1; depth < chars; ++depth) {
        std::vector<SuffixSlot> next;
        for (const SuffixSlot &parent : level)
            for (int32_t c = 1; c < ix.wt_sigma; ++c) {
                if (c + 1 >= ix.n_c) continue;
                SuffixSlot child;
                if (fm_suffix_extend(ix, parent, depth, c, key_bits, child)) next.push_back(child);
            }
        level.swap(next);
        levels.push_back(level);
    }
    // sized as the library does when nothing limits it: a power of two, half full at most
    uint64_t total = 0;
    for (const auto &l : levels) total += l.size();
    uint32_t slots = 1024;
    while (slots < 2 * total) slots <<= 1;
    int log2_slots = 0;
    while ((1u << log2_slots) < slots) ++log2_slots;
    std::vector<SuffixSlot> table(slots, SuffixSlot{kSuffixEmpty, 0, 0});
    ix.suffix_key_bits = key_bits;
    ix.suffix_chars = chars;
    ix.suffix_shift = (uint32_t)(64 - (log2_slots - kSuffixGroupLog2));
    ix.suffix_mask = slots - 1;
    for (size_t i = 0; i < levels.size(); ++i)
        for (const SuffixSlot &e : levels[i]) {
            if (e.key == kSuffixEmpty) continue;
            uint32_t h = fm_suffix_home(ix, e.key, (int)i + 2);
            while (table[h].key != kSuffixEmpty) h = (h + kSuffixGroup) & ix.suffix_mask;
            table[h] = e;
        }
    ix.suffix_table = table.data();
    const int64_t entries = (int64_t)total;
    int64_t answered = 0;
    for (int32_t p = 0; p < n; ++p) {
        const int32_t beg = off[p], m = off[p + 1] - beg;
        int status = ST_OK;
        int32_t start = 0, end = 0, back = 0;
        if (m <= 0) {
            status = ST_JAVA_AIOOBE;
        } else {
            int32_t c = fm_map(ix, pat[beg + m - 1]);
            if (c != 0) {
                start = ix.C[c];
                end = ix.C[c + 1];
                uint64_t key;
                const int len = fm_suffix_len(ix, m);
                if (len >= 2 && fm_suffix_key(ix, [&](int j) { return (uint32_t)fm_map(ix, pat[beg + m - 1 - j]); }, len, key) &&
                    fm_suffix_lookup(ix, key, len, start, end, back))
                    answered += 2 * back;
                while (start < end && back + 1 < m) {
                    ++back;
                    c = fm_map(ix, pat[beg + m - 1 - back]);
                    if (c == 0) {
                        start = end = 0;
                        --back;
                        break;
                    }
                    const int32_t s2 = wt_rank_folded(ix, ix.inv_global, (uint32_t)start, c, status);
                    const int32_t e2 = wt_rank_folded(ix, ix.inv_global, (uint32_t)end, c, status);
                    start = s2;
                    end = e2;
                }
            }
        }
        const int32_t d = end - start;
        counts[p] = d > 0 ? d : 0;
        if (lf) lf[p] = 2 * back;
        if (status_out) status_out[p] = status;
    }
    if (answered_steps) *answered_steps = answered;
    return (int64_t)entries;
}

// mirrors k_locate_walk
void sim_locate_walk(const uint8_t *blob, const int32_t *range, int32_t n, int32_t max_matches, int32_t *locs,
                     int32_t loc_cap, int32_t *found, int32_t *lf, int32_t *status_out) {
    DevIndex ix = make_index(blob);
    int32_t slots = (max_matches > 0 && max_matches < loc_cap) ? max_matches : loc_cap;
    if (slots < 1) slots = 1;
    for (int32_t p = 0; p < n; ++p)
        for (int32_t k = 0; k < slots; ++k) {
            const int32_t start = range[2 * p], end = range[2 * p + 1];
            const int32_t hits = start < end ? end - start : 0;
            const int32_t wanted = (max_matches > 0 && hits > max_matches) ? max_matches : hits;
            const int32_t located = wanted < loc_cap ? wanted : loc_cap;
            if (k == 0) {
                found[p] = located;
                if (wanted > loc_cap && status_out) status_out[p] |= ST_JAVA_AIOOBE;
            }
            if (k >= located) continue;
            int status = ST_OK;
            int32_t distance;
            if (ix.win && g_pack && ix.sample_rate >= 8) {
                // (as launch_locate_walk picks over a window directory: k_locate_walk_c — the walk in instalments of sample_rate / 2,
                // sample_rate / 4 steps and the rest, as the kernel takes them between its packings)
                WalkState w = {start + 1 + k, 0, ST_OK};
                const int32_t limit = fm_walk_limit(ix);
                // (the kernel's instantiation for the directory's form: FMX_DISPATCH_FORM)
                auto steps = [&](int32_t budget) {
                    return ix.win_flat ? fm_locate_steps_win<kFormFlat>(ix, w, budget, limit) : fm_locate_steps_win<kFormCells>(ix, w, budget, limit);
                };
                if (!steps(ix.sample_rate / 2) && !steps(ix.sample_rate / 4)) (void)steps(0x7fffffff);
                locs[(int64_t)p * loc_cap + k] = fm_locate_finish_win(ix, ix.inv_global, w);
                distance = w.distance;
                status = w.status;
            } else
            locs[(int64_t)p * loc_cap + k] = ix.win_flat ? fm_locate_hit<kWinFlat>(ix, ix.inv_global, start, k, distance, status)  // (as launch_locate_walk picks)
                                             : ix.win    ? fm_locate_hit<kWinAlways>(ix, ix.inv_global, start, k, distance, status)
                                                         : fm_locate_hit<kWinNever>(ix, ix.inv_global, start, k, distance, status);
            if (lf) lf[p] += distance;
            if (status && status_out) status_out[p] |= status;
        }
}

void sim_extract(const uint8_t *blob, const int32_t *starts, const int32_t *stops, int32_t n, uint16_t *dst,
                 int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf, int32_t *status_out) {
    DevIndex ix = make_index(blob);
    for (int32_t q = 0; q < n; ++q) {
        int status = ST_OK;
        int32_t steps;
        uint16_t *row = dst + (int64_t)q * dst_len;
        const int32_t ret = ix.win_flat ? fm_extract<kWinFlat>(ix, ix.inv_global, starts[q], stops[q], row, dst_len, offset, steps, status)  // (as launch_extract picks)
                            : ix.win    ? fm_extract<kWinAlways>(ix, ix.inv_global, starts[q], stops[q], row, dst_len, offset, steps, status)
                                        : fm_extract<kWinNever>(ix, ix.inv_global, starts[q], stops[q], row, dst_len, offset, steps, status);
        out_len[q] = status ? 0 : ret;
        if (lf) lf[q] = steps;
        if (status_out) status_out[q] = status;
    }
}

void sim_extract_boundary(const uint8_t *blob, const int32_t *froms, int32_t n, uint16_t boundary, int mode,
                          uint16_t *dst, int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf,
                          int32_t *status_out, int32_t *aux_out, int accelerate) {
    DevIndex ix = make_index(blob);
    const int32_t mapped_boundary = fm_map(ix, boundary);
    uint16_t *scratch = accelerate ? new uint16_t[2 * (size_t)ix.sample_rate + 2] : nullptr;
    for (int32_t q = 0; q < n; ++q) {
        int status = ST_OK;
        int32_t steps, aux;
        int32_t ret;
        if (accelerate >= 2) {  // group-cooperative form with a group of one lane (3: the lane's two first walks interleaved)
            bool clean;
            // (over a complete window directory: the instantiation without tree-walk code, as launch_extract_boundary picks)
            ret = ix.win_flat
                      ? fm_extract_boundary_group<1, -1, kWinFlat>(ix, ix.inv_global, mode, froms[q], mapped_boundary, dst + (int64_t)q * dst_len,
                                                                   dst_len, offset, steps, status, aux, scratch, 1, 1, ix.sample_rate + 1, 0,
                                                                   clean, accelerate == 3)
                  : ix.win  // (the instantiation without the tree walk: not what the kernel runs, kept covered)
                      ? fm_extract_boundary_group<1, -1, kWinAlways>(ix, ix.inv_global, mode, froms[q], mapped_boundary, dst + (int64_t)q * dst_len,
                                                                     dst_len, offset, steps, status, aux, scratch, 1, 1, ix.sample_rate + 1, 0,
                                                                     clean, accelerate == 3)
                      : fm_extract_boundary_group<1>(ix, ix.inv_global, mode, froms[q], mapped_boundary, dst + (int64_t)q * dst_len,
                                                     dst_len, offset, steps, status, aux, scratch, 1, 1, ix.sample_rate + 1, 0, clean,
                                                     accelerate == 3);
            if (!clean) {
                status = ST_OK;
                ret = fm_extract_boundary(ix, ix.inv_global, mode, froms[q], mapped_boundary, dst + (int64_t)q * dst_len,
                                          dst_len, offset, steps, status, aux);
            }
        } else {
            ret = fm_extract_boundary(ix, ix.inv_global, mode, froms[q], mapped_boundary, dst + (int64_t)q * dst_len,
                                      dst_len, offset, steps, status, aux, scratch, 1);
        }
        out_len[q] = status ? 0 : ret;
        if (lf) lf[q] = steps;
        if (status_out) status_out[q] = status;
        if (aux_out) aux_out[q] = aux;
    }
    delete[] scratch;
}
}

// how often the marked replay of extractUntilBoundary answered (per mode) since the library was loaded
extern "C" long sim_marked_replays(int mode) { return (mode >= 0 && mode < 3) ? fmx::g_marked_replays[mode] : -1; }
