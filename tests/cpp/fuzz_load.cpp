// fuzz_load.cpp — TEST-ONLY mutation fuzz of the two doors an index comes through:
//   A. a serialized index4j stream:  parse_model (FM:983-1025) -> validate_model -> flatten_model -> validate_blob
//      (= fmx_load + fmx_to_device)
//   B. a flat image as fmx_attach_device_blob receives it: validate_blob (checksum recomputed by the "attacker")
// Whatever the validators ACCEPT is then queried with the device code itself, compiled for the host
// (tests/hostsim.cpp), under AddressSanitizer: an out-of-bounds read or an endless loop here is an out-of-bounds read or
// a hung wave on the GPU.  Usage: fuzz_load <iterations> <seed>; prints one summary line, exit code 0 = clean.
// Built a second time with -DFMX_COMPACT=1: the same campaign over COMPACT images (option image_compact) and the device
// header's record-decoding form (what the kernels of namespace fmxc run).
#include "../hostsim.cpp"

#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unistd.h>
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#endif

#include "fmx_model.hpp"

extern "C" int fmx_synth_log(uint64_t seed, int32_t n, uint16_t *out);
extern "C" int fmx_synth_log_multichar(uint64_t seed, int32_t n, int32_t symbols, uint16_t *out);

namespace {

struct Rng {
    uint64_t s;
    uint64_t next() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return s;
    }
    uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
};

const char *g_where = "start";
long g_iter = -1;
void on_alarm(int) {
    fprintf(stderr, "HANG in %s (iteration %ld)\n", g_where, g_iter);
#if defined(__SANITIZE_ADDRESS__)
    __sanitizer_print_stack_trace();
#endif
    _exit(3);
}

const int32_t kSpecial[] = {0, 1, -1, 2, 0x7fffffff, (int32_t)0x80000000, 0x7ffffffe, 255, 256, 65535, 65536, 1 << 20, 1 << 24, -2};

// a few bytes of damage; `len` stays
void mutate(std::vector<uint8_t> &b, Rng &r, bool big_endian) {
    const int n_mut = 1 + (int)r.below(3);
    for (int k = 0; k < n_mut; ++k) {
        size_t at = r.below(4) == 0 ? r.below((uint32_t)std::min<size_t>(b.size(), 4096)) : r.below((uint32_t)b.size());
        switch (r.below(5)) {
        case 0: b[at] ^= (uint8_t)(1u << r.below(8)); break;
        case 1: b[at] = (uint8_t)r.next(); break;
        case 2: {  // a special 32-bit value at an aligned place
            at &= ~(size_t)3;
            if (at + 4 > b.size()) break;
            int32_t v = kSpecial[r.below(sizeof kSpecial / sizeof *kSpecial)];
            if (r.below(3) == 0) {  // the old value +- a little
                uint32_t old = 0;
                for (int i = 0; i < 4; ++i) old |= (uint32_t)b[at + i] << (big_endian ? 24 - 8 * i : 8 * i);
                v = (int32_t)(old + (uint32_t)(r.below(2) ? 1 : -1) * (1 + r.below(64)));
            }
            for (int i = 0; i < 4; ++i) b[at + i] = (uint8_t)((uint32_t)v >> (big_endian ? 24 - 8 * i : 8 * i));
            break;
        }
        case 3: {  // a special 16-bit value
            at &= ~(size_t)1;
            if (at + 2 > b.size()) break;
            const int32_t v = kSpecial[r.below(sizeof kSpecial / sizeof *kSpecial)];
            b[at + (big_endian ? 1 : 0)] = (uint8_t)v;
            b[at + (big_endian ? 0 : 1)] = (uint8_t)(v >> 8);
            break;
        }
        default: {  // a short run of one byte
            const size_t run = 1 + r.below(16);
            const uint8_t v = r.below(2) ? 0 : 0xff;
            for (size_t i = at; i < b.size() && i < at + run; ++i) b[i] = v;
        }
        }
    }
}

// damage to the SMALL tables of a model (a byte-level mutation of the stream mostly lands in bit vectors and samples):
// block headers, header bytes, mapping values, superblock fields, counts, RRR shapes, the character map
template <class T>
void poke(T &v, Rng &r) {
    switch (r.below(4)) {
    case 0: v = (T)kSpecial[r.below(sizeof kSpecial / sizeof *kSpecial)]; break;
    case 1: v = (T)(v + (T)(1 + r.below(8))); break;
    case 2: v = (T)(v - (T)(1 + r.below(8))); break;
    default: v = (T)r.next();
    }
}
void mutate_rrr(fmx::RrrModel &x, Rng &r) {
    switch (r.below(8)) {
    case 0: poke(x.sample_size, r); break;
    case 1: poke(x.length, r); break;
    case 2: poke(x.total_ones, r); break;
    case 3: poke(x.bits_per_offset_pos, r); break;
    case 4: if (!x.classes.words.empty()) x.classes.words[r.below((uint32_t)x.classes.words.size())] = r.next(); break;
    case 5: if (!x.offsets.empty()) x.offsets[r.below((uint32_t)x.offsets.size())] = r.next(); break;
    case 6: if (!x.sampled_offsets.words.empty()) x.sampled_offsets.words[r.below((uint32_t)x.sampled_offsets.words.size())] = r.next(); break;
    default: if (!x.prefix_sums.words.empty()) x.prefix_sums.words[r.below((uint32_t)x.prefix_sums.words.size())] = r.next();
    }
}
void mutate_model(fmx::FmModel &m, Rng &r) {
    const int n_mut = 1 + (int)r.below(3);
    for (int k = 0; k < n_mut; ++k) {
        fmx::WfbbModel &w = m.wt;
        fmx::SuperBlockModel *sb = w.sb.empty() ? nullptr : &w.sb[r.below((uint32_t)w.sb.size())];
        switch (r.below(16)) {
        case 0: if (sb && !sb->block_headers.empty()) {
            auto &bh = sb->block_headers[r.below((uint32_t)sb->block_headers.size())];
            switch (r.below(5)) {
            case 0: poke(bh.bv_rank, r); break;
            case 1: poke(bh.bv_offset, r); break;
            case 2: poke(bh.var_off, r); break;
            case 3: poke(bh.sigma, r); break;
            default: poke(bh.tree_height, r);
            }
        } break;
        case 1: case 2: case 3: if (sb && !sb->var.empty()) {  // a byte / a 16-bit field of a block's variable-size header
            const size_t at = r.below((uint32_t)sb->var.size());
            if (r.below(2)) sb->var[at] = (uint8_t)r.next();
            else { sb->var[at] = (uint8_t)kSpecial[r.below(14)]; if (at + 1 < sb->var.size()) sb->var[at + 1] = (uint8_t)(r.below(2) ? 0 : 0xff); }
        } break;
        case 4: case 5: if (sb && !sb->mapping.empty()) poke(sb->mapping[r.below((uint32_t)sb->mapping.size())], r); break;
        case 6: if (sb) { if (r.below(2)) poke(sb->sigma, r); else poke(sb->block_size_log, r); } break;
        case 7: if (sb) mutate_rrr(sb->rank_support, r); break;
        case 8: if (!w.global_mapping.empty()) poke(w.global_mapping[r.below((uint32_t)w.global_mapping.size())], r); break;
        case 9: if (!w.super_rank.empty()) poke(w.super_rank[r.below((uint32_t)w.super_rank.size())], r); break;
        case 10: if (!w.count.empty()) poke(w.count[r.below((uint32_t)w.count.size())], r); break;
        case 11: if (!m.C.empty()) poke(m.C[r.below((uint32_t)m.C.size())], r); break;
        case 12: if (!m.map_vals.empty()) poke(m.map_vals[r.below((uint32_t)m.map_vals.size())], r); break;
        case 13: mutate_rrr(m.sampled, r); break;
        case 14: switch (r.below(6)) {
            case 0: poke(m.sample_rate, r); break;
            case 1: poke(m.length, r); break;
            case 2: poke(w.alphabet_size, r); break;
            case 3: poke(w.size, r); break;
            case 4: poke(m.bw_suffixes, r); break;
            default: poke(m.bw_positions, r);
            } break;
        default: {
            fmx::PackedVec &v = r.below(2) ? m.suffixes : m.positions;
            if (!v.words.empty()) v.words[r.below((uint32_t)v.words.size())] = r.next();
        }
        }
    }
}

// every query kind over an accepted image, copied into an exact-size heap block (ASan sees the first byte past it).
// directory: 0 = the tree walks alone; 4 / 6 / -1 = with the window directory grown over the image first, as fmx_to_device does by
// default (win_build_cell / win_build_other, entries of four or six bytes; -1: the flat form, win_build_flat), so that locate / extract / extractUntilBoundary take
// their steps from it: a directory made from a damaged tree holds arbitrary rows, and the walks over it must stay inside it.
void query_everything(const std::vector<uint8_t> &blob, const std::vector<uint16_t> &text, Rng &r, int directory = 0) {
    uint8_t *img = new uint8_t[blob.size()];
    memcpy(img, blob.data(), blob.size());
    if (directory) {
        g_where = "growing the window directory";
        sim_set_entry_bytes(directory);
        (void)sim_win_attach(img, nullptr);
        sim_set_entry_bytes(0);
    }
    BlobHeader h;
    memcpy(&h, img, sizeof h);
    const int n = 24;
    std::vector<uint16_t> pat;
    std::vector<int32_t> off(1, 0);
    for (int q = 0; q < n; ++q) {
        const int m = 1 + (int)r.below(12);
        const size_t from = r.below((uint32_t)(text.size() - 16));
        for (int i = 0; i < m; ++i) pat.push_back(r.below(16) == 0 ? (uint16_t)r.next() : text[from + i]);
        off.push_back((int32_t)pat.size());
    }
    std::vector<int32_t> counts(n), lf(n), st(n), range(2 * n);
    g_where = "count";
    sim_count(img, pat.data(), off.data(), n, counts.data(), lf.data(), st.data(), range.data());
    {  // the same batch from a suffix table grown over this image (fm_suffix_extend, fm_suffix_key, fm_suffix_lookup)
        std::vector<int32_t> c2(n), lf2(n), st2(n);
        int64_t answered = 0;
        g_where = "count with a suffix table";
        (void)sim_count_table(img, h.wt_sigma < 100 ? 3 : 2, pat.data(), off.data(), n, c2.data(), lf2.data(), st2.data(), &answered);
    }
    // ranges as the kernels may see them: what count left, plus a few arbitrary rows inside [0, length]
    const int32_t rows = h.length + 1;
    for (int q = 0; q < n; q += 3) {
        range[2 * q] = (int32_t)r.below((uint32_t)rows);
        range[2 * q + 1] = range[2 * q] + (int32_t)r.below(8);
        if (range[2 * q + 1] > rows) range[2 * q + 1] = rows;
    }
    for (int q = 0; q < n; ++q) {  // (the product clamps ranges to the index before the walk: keep them inside)
        if (range[2 * q] < 0) range[2 * q] = 0;
        if (range[2 * q + 1] > rows) range[2 * q + 1] = rows;
    }
    const int cap = 4;
    std::vector<int32_t> locs((size_t)n * cap), found(n);
    std::fill(lf.begin(), lf.end(), 0);
    std::fill(st.begin(), st.end(), 0);
    g_where = "locate";
    sim_locate_walk(img, range.data(), n, cap, locs.data(), cap, found.data(), lf.data(), st.data());
    if (h.enable_extract) {
        const int dst_len = 40;
        std::vector<int32_t> a(n), b(n), out_len(n), aux(n);
        std::vector<uint16_t> dst((size_t)n * dst_len);
        for (int q = 0; q < n; ++q) {
            a[q] = (int32_t)r.below((uint32_t)std::max(1, h.length)) - (r.below(16) == 0 ? 3 : 0);
            b[q] = a[q] + (int32_t)r.below(48) - (r.below(16) == 0 ? 5 : 0);
        }
        g_where = "extract";
        sim_extract(img, a.data(), b.data(), n, dst.data(), dst_len, (int32_t)r.below(4), out_len.data(), lf.data(), st.data());
        for (int mode = 0; mode < 3; ++mode)
            for (int acc = 0; acc < 4; ++acc) {
                // (the windowed forms hold 2 x sampleRate characters per lane; the library takes the literal form when
                // that would not fit — fmx_api.cpp: boundary_impl)
                if (acc && h.sample_rate > (1 << 20)) continue;
                g_where = "extractUntilBoundary";
                sim_extract_boundary(img, a.data(), n, (uint16_t)'\n', mode, dst.data(), dst_len, 0, out_len.data(), lf.data(),
                                     st.data(), aux.data(), acc);
            }
    }
    if (directory) sim_win_detach(img);
    delete[] img;
}

}  // namespace

int main(int argc, char **argv) {
    const long iterations = argc > 1 ? atol(argv[1]) : 2000;
    Rng r{argc > 2 ? strtoull(argv[2], nullptr, 10) * 0x9e3779b97f4a7c15ull + 1 : 88172645463325252ull};
    signal(SIGALRM, on_alarm);
#if FMX_COMPACT
    fmx::set_image_compact(1);  // built with -DFMX_COMPACT=1: compact images (RRR records) through the decoding form of the device header
#endif
    struct Base {
        std::vector<uint16_t> text;
        std::vector<uint8_t> ser, blob;
        fmx::FmModel model;
    };
    std::vector<Base> bases;
    const int kinds[][3] = {{24000, 0, 8}, {9000, 0, 1}, {30000, 0, 32}, {20000, 700, 4}, {6000, 1, 2}};  // n, alphabet kind, sampleRate
    for (const auto &k : kinds) {
        Base b;
        b.text.resize(k[0]);
        if (k[1] == 0)
            fmx_synth_log(7 + k[2], k[0], b.text.data());
        else if (k[1] == 1)
            for (auto &c : b.text) c = (uint16_t)('a' + r.below(3));  // tiny alphabet: run blocks, shallow trees
        else
            fmx_synth_log_multichar(11, k[0], k[1], b.text.data());
        fmx::FmModel m;
        std::string err;
        if (fmx::build_model(b.text.data(), k[0], k[2], true, m, err)) return printf("build failed: %s\n", err.c_str()), 2;
        fmx::emit_model(m, r.below(2) == 0, b.ser);
        if (fmx::flatten_model(m, b.blob, err)) return printf("flatten failed: %s\n", err.c_str()), 2;
        if (fmx::validate_blob(b.blob.data(), b.blob.size(), err)) return printf("a fresh image fails validation: %s\n", err.c_str()), 2;
        query_everything(b.blob, b.text, r);  // the undamaged image first
        query_everything(b.blob, b.text, r, 4);
        query_everything(b.blob, b.text, r, 6);
        query_everything(b.blob, b.text, r, -1);
        b.model = std::move(m);
        bases.push_back(std::move(b));
    }
    long parsed = 0, flattened = 0, images_accepted = 0, a_runs = 0, b_runs = 0, flat_invalid = 0;
    for (g_iter = 0; g_iter < iterations; ++g_iter) {
        const Base &base = bases[r.below((uint32_t)bases.size())];
        std::string err;
        alarm(20);
        if (g_iter % 3) {  // door A: a damaged stream — bytes of the stream, or fields of the model it is written from
            ++a_runs;
            std::vector<uint8_t> ser;
            if (g_iter % 3 == 1) {
                ser = base.ser;
                mutate(ser, r, true);
                if (r.below(16) == 0) ser.resize(r.below((uint32_t)ser.size()));
            } else {
                fmx::FmModel damaged = base.model;
                mutate_model(damaged, r);
                g_where = "emit_model";
                // (a writer cannot serialize what its own containers contradict: such a model never becomes a stream)
                if (damaged.wt.sb.size() != base.model.wt.sb.size()) continue;
                fmx::emit_model(damaged, false, ser);
            }
            fmx::FmModel m;
            g_where = "parse_model";
            if (fmx::parse_model(ser.data(), ser.size(), m, err)) continue;
            g_where = "validate_model";
            if (fmx::validate_model(m, err)) continue;  // (fmx_load: parse_model + validate_model)
            ++parsed;
            std::vector<uint8_t> blob;
            g_where = "flatten_model";
            if (fmx::flatten_model(m, blob, err)) continue;
            ++flattened;
            g_where = "validate_blob(flattened)";
            if (fmx::validate_blob(blob.data(), blob.size(), err)) {
                ++flat_invalid;  // (fmx_to_device refuses it the same way: an image made from a caller's bytes is validated)
                if (getenv("FUZZ_VERBOSE")) fprintf(stderr, "flattened image fails validate_blob: %s\n", err.c_str());
                continue;
            }
            query_everything(blob, base.text, r);
            query_everything(blob, base.text, r, g_iter % 3 == 0 ? 4 : (g_iter % 3 == 1 ? 6 : -1));
        } else {  // door B: a damaged image with a matching checksum
            ++b_runs;
            std::vector<uint8_t> blob = base.blob;
            mutate(blob, r, false);
            BlobHeader h;
            memcpy(&h, blob.data(), sizeof h);
            h.checksum = 0;
            memcpy(blob.data(), &h, sizeof h);
            h.checksum = fmx::image_checksum(blob.data(), blob.size());
            memcpy(blob.data(), &h, sizeof h);
            g_where = "validate_blob";
            if (fmx::validate_blob(blob.data(), blob.size(), err)) continue;
            ++images_accepted;
            query_everything(blob, base.text, r);
            query_everything(blob, base.text, r, g_iter % 3 == 0 ? 4 : (g_iter % 3 == 1 ? 6 : -1));
        }
    }
    alarm(0);
    printf("fuzz ok: %ld streams (%ld accepted by parse_model + validate_model, %ld flattened, %ld of those refused by "
           "validate_blob), %ld images (%ld accepted)\n", a_runs, parsed, flattened, flat_invalid, b_runs, images_accepted);
    return 0;
}
