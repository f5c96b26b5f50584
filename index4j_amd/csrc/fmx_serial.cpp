// fmx_serial.cpp — reader and writer for index4j's serialized layout (the hand-over format between
// a JVM running index4j and this library).
//
// Grammar (big-endian java.io.DataOutput primitives), SURVEY.md §8 A16:
//   FmIndex   FM:948-975    IntVector IV:196-203    VariableWidthIntVector VIV:175-181
//   RrrVector RRR:430-440   Wavelet   WFBB:1544-1570, SuperBlock WFBB:1651-1667, Block WFBB:1607-1613
//   optional ObjectOutputStream framing added by Serialization.writeToByteArray SER:67-79:
//   magic AC ED 00 05, then block-data records 0x77 <len8> / 0x7A <len32> of at most 1024 bytes.
// The framing and the HashMap key order of FM:956-960 are JDK behaviour stated from knowledge; no
// JVM exists in the build image to confirm them (the reader is order-agnostic and accepts both
// framed and raw streams).
#include "fmx_model.hpp"

#include <algorithm>
#include <cstring>

namespace fmx {
namespace {

struct Reader {
    const uint8_t *p;
    size_t n, pos = 0;
    int err = 0;  // 1 truncated, 2 bad version, 3 malformed
    int bad_version = 0;
    size_t left() const { return n - pos; }
    bool need(size_t k) {
        if (err) return false;
        if (left() < k) {
            err = 1;
            return false;
        }
        return true;
    }
    int u8() { return need(1) ? p[pos++] : 0; }
    int16_t i16() {
        if (!need(2)) return 0;
        int16_t v = (int16_t)((p[pos] << 8) | p[pos + 1]);
        pos += 2;
        return v;
    }
    int32_t i32() {
        if (!need(4)) return 0;
        uint32_t v = ((uint32_t)p[pos] << 24) | ((uint32_t)p[pos + 1] << 16) | ((uint32_t)p[pos + 2] << 8) | p[pos + 3];
        pos += 4;
        return (int32_t)v;
    }
    int64_t i64() {
        if (!need(8)) return 0;
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v = (v << 8) | p[pos + i];
        pos += 8;
        return (int64_t)v;
    }
    void version() {  // SER:46-56
        const int v = u8();
        if (v != 0 && !err) {
            err = 2;
            bad_version = v;
        }
    }
    // a length prefix that must be satisfiable by the remaining bytes (elem = bytes per element)
    int32_t count(size_t elem) {
        int32_t c = i32();
        if (!err && (c < 0 || (uint64_t)c * elem > left())) err = 3;
        return err ? 0 : c;
    }
};

void read_packed(Reader &r, PackedVec &v) {  // IV:211-227
    r.version();
    int32_t length = r.i32(), width = r.i32();
    if (r.err) return;
    if (length < 0 || width < 0 || width > 64) {
        r.err = 3;
        return;
    }
    int64_t words = words_for_bits((int64_t)length * width);
    if ((uint64_t)words * 8 > r.left()) {
        r.err = 1;
        return;
    }
    v.length = length;
    v.width = width;
    v.words.resize((size_t)words);
    for (int64_t i = 0; i < words; ++i) v.words[(size_t)i] = (uint64_t)r.i64();
}

void read_var(Reader &r, std::vector<uint64_t> &w) {  // VIV:189-198
    r.version();
    int32_t n = r.count(8);
    w.resize((size_t)n);
    for (int32_t i = 0; i < n; ++i) w[(size_t)i] = (uint64_t)r.i64();
}

void read_rrr(Reader &r, RrrModel &m) {  // RRR:448-469
    r.version();
    m.sample_size = r.i32();
    m.length = r.i32();
    m.total_ones = r.i32();
    m.bits_per_offset_pos = r.i32();
    read_packed(r, m.classes);
    read_var(r, m.offsets);
    read_packed(r, m.sampled_offsets);
    read_packed(r, m.prefix_sums);
    if (!r.err && (m.sample_size <= 0 || m.length < 0 || m.classes.width != 4)) r.err = 3;
}

void read_wavelet(Reader &r, WfbbModel &w) {  // WFBB:286-322
    r.version();
    w.size = r.i64();
    w.alphabet_size = r.i32();
    w.sampling_rate = r.i32();
    int32_t n = r.count(8);
    w.count.resize((size_t)n);
    for (auto &v : w.count) v = r.i64();
    n = r.count(8);
    w.hyper_rank.resize((size_t)n);
    for (auto &v : w.hyper_rank) v = r.i64();
    n = r.count(4);
    w.super_rank.resize((size_t)n);
    for (auto &v : w.super_rank) v = r.i32();
    n = r.count(2);
    w.global_mapping.resize((size_t)n);
    for (auto &v : w.global_mapping) v = r.i16();
    n = r.count(4);
    w.sb.resize((size_t)n);
    for (auto &sb : w.sb) {  // WFBB:1630-1649
        if (r.err) return;
        sb.sigma = r.i16();
        sb.block_size_log = r.i16();
        read_rrr(r, sb.rank_support);
        int32_t nb = r.count(16);
        sb.block_headers.resize((size_t)nb);
        for (auto &bh : sb.block_headers) {  // WFBB:1597-1605
            bh.bv_rank = r.i32();
            bh.bv_offset = r.i32();
            bh.var_off = r.i32();
            bh.sigma = r.i16();
            bh.tree_height = r.i16();
        }
        int32_t nv = r.count(1);
        sb.var.resize((size_t)nv);
        if (nv && r.need((size_t)nv)) {
            memcpy(sb.var.data(), r.p + r.pos, (size_t)nv);
            r.pos += (size_t)nv;
        }
        int32_t nm = r.count(2);
        sb.mapping.resize((size_t)nm);
        for (auto &v : sb.mapping) v = r.i16();
    }
}

struct Writer {
    std::vector<uint8_t> &o;
    void u8(int v) { o.push_back((uint8_t)v); }
    void i16(int v) {
        o.push_back((uint8_t)((v >> 8) & 0xff));
        o.push_back((uint8_t)(v & 0xff));
    }
    void i32(int32_t v) {
        uint32_t u = (uint32_t)v;
        for (int s = 24; s >= 0; s -= 8) o.push_back((uint8_t)((u >> s) & 0xff));
    }
    void i64(int64_t v) {
        uint64_t u = (uint64_t)v;
        for (int s = 56; s >= 0; s -= 8) o.push_back((uint8_t)((u >> s) & 0xff));
    }
};

void write_packed(Writer &w, const PackedVec &v) {  // IV:196-203 (no word count: derived from length*width)
    w.u8(0);
    w.i32(v.length);
    w.i32(v.width);
    for (uint64_t x : v.words) w.i64((int64_t)x);
}
void write_var(Writer &w, const std::vector<uint64_t> &v) {  // VIV:175-181
    w.u8(0);
    w.i32((int32_t)v.size());
    for (uint64_t x : v) w.i64((int64_t)x);
}
void write_rrr(Writer &w, const RrrModel &m) {  // RRR:430-440
    w.u8(0);
    w.i32(m.sample_size);
    w.i32(m.length);
    w.i32(m.total_ones);
    w.i32(m.bits_per_offset_pos);
    write_packed(w, m.classes);
    write_var(w, m.offsets);
    write_packed(w, m.sampled_offsets);
    write_packed(w, m.prefix_sums);
}
void write_wavelet(Writer &w, const WfbbModel &m) {  // WFBB:1544-1570
    w.u8(0);
    w.i64(m.size);
    w.i32(m.alphabet_size);
    w.i32(m.sampling_rate);
    w.i32((int32_t)m.count.size());
    for (auto v : m.count) w.i64(v);
    w.i32((int32_t)m.hyper_rank.size());
    for (auto v : m.hyper_rank) w.i64(v);
    w.i32((int32_t)m.super_rank.size());
    for (auto v : m.super_rank) w.i32(v);
    w.i32((int32_t)m.global_mapping.size());
    for (auto v : m.global_mapping) w.i16(v);
    w.i32((int32_t)m.sb.size());
    for (const auto &sb : m.sb) {  // WFBB:1651-1667
        w.i16(sb.sigma);
        w.i16(sb.block_size_log);
        write_rrr(w, sb.rank_support);
        w.i32((int32_t)sb.block_headers.size());
        for (const auto &bh : sb.block_headers) {  // WFBB:1607-1613
            w.i32(bh.bv_rank);
            w.i32(bh.bv_offset);
            w.i32(bh.var_off);
            w.i16(bh.sigma);
            w.i16(bh.tree_height);
        }
        w.i32((int32_t)sb.var.size());
        w.o.insert(w.o.end(), sb.var.begin(), sb.var.end());
        w.i32((int32_t)sb.mapping.size());
        for (auto v : sb.mapping) w.i16(v);
    }
}

// java.util.HashMap<Integer,Short>.keySet() order for FM:956-960, by replaying the puts (JDK 8+ HashMap.putVal / resize,
// stated from knowledge: no JVM here): the table starts at 16 slots and doubles when the size passes 0.75 x capacity — and
// also when a put makes a bucket 9 nodes long while the table has fewer than 64 slots (treeifyBin resizes instead of
// treeifying below MIN_TREEIFY_CAPACITY); slot = (h ^ (h >>> 16)) & (capacity - 1) with h = the int key; a resize splits every
// bucket in two keeping the nodes' relative order, so a bucket holds its keys in insertion order; keySet() walks the slots
// upwards.  A bucket that reaches 9 nodes at 64 slots or more becomes a red-black TREE bin, whose iteration order (root moved
// to the front, later nodes linked behind their tree parents) this replay does NOT model: `treeified` reports it, the order
// returned is the plain-bucket one, and fmx_save_key_order_modelled() tells the caller that the bytes — a stream FmIndex.read
// accepts, its reader is order-agnostic (FM:992-998) — may differ from a JVM's inside that bucket.
// map_keys is in insertion order: '\0' first, then first appearance (FM:396-420), or the order of the stream it was read from.
std::vector<int> hashmap_order(const FmModel &m, bool *treeified) {
    const int n = (int)m.map_keys.size();
    uint32_t cap = 16;
    bool tree = false;
    std::vector<std::vector<int>> bins(cap);
    auto slot_of = [&](int i, uint32_t c) {
        const uint32_t h = (uint32_t)m.map_keys[(size_t)i];
        return (h ^ (h >> 16)) & (c - 1);
    };
    auto resize = [&]() {
        std::vector<std::vector<int>> next((size_t)cap * 2);
        for (uint32_t b = 0; b < cap; ++b)
            for (int i : bins[b]) next[slot_of(i, cap * 2)].push_back(i);  // (b or b + cap: relative order kept)
        bins.swap(next);
        cap *= 2;
    };
    for (int i = 0; i < n; ++i) {
        std::vector<int> &bin = bins[slot_of(i, cap)];
        const size_t before = bin.size();
        bin.push_back(i);
        if (before >= 8) {  // putVal: binCount >= TREEIFY_THRESHOLD - 1 -> treeifyBin
            if (cap < 64)
                resize();
            else
                tree = true;
        }
        if ((uint32_t)(i + 1) > cap / 4 * 3) resize();  // ++size > threshold
    }
    std::vector<int> order;
    order.reserve((size_t)n);
    for (uint32_t b = 0; b < cap; ++b)
        for (int i : bins[b]) order.push_back(i);
    if (treeified) *treeified = tree;
    return order;
}

}  // namespace

bool key_order_is_modelled(const FmModel &m) {
    bool tree = false;
    (void)hashmap_order(m, &tree);
    return !tree;
}

// FM:983-1025.  Returns 0, or 1 truncated / 2 version / 3 malformed.
int parse_model(const uint8_t *buf, size_t len, FmModel &m, std::string &err) {
    std::vector<uint8_t> plain;
    bool corrupt_tail = false;
    if (len >= 4 && buf[0] == 0xAC && buf[1] == 0xED && buf[2] == 0x00 && buf[3] == 0x05) {  // SER:89-100
        plain.reserve(len);
        // ObjectInputStream's block-data reader (java.io.ObjectInputStream.BlockDataInputStream.readBlockHeader / refill), which
        // works LAZILY — a header is only looked at when FmIndex.read asks for bytes the earlier records did not hold:
        // payloads of TC_BLOCKDATA 0x77 <u8 len> and TC_BLOCKDATALONG 0x7A <i32 len> records are one byte sequence (a
        // primitive may straddle records; empty records are legal); TC_RESET 0x79 may stand between records; any other tag
        // ends the block data (EOFException for a reader that wants more); a negative long length is a
        // StreamCorruptedException for a reader that gets that far.  So: gather what is well-formed, parse, and let the
        // parser's "truncated" become "malformed" when the payload ended at a corrupt header.
        size_t pos = 4;
        while (pos < len) {
            size_t bl;
            if (buf[pos] == 0x79) {
                ++pos;
                continue;
            }
            if (buf[pos] == 0x77) {
                if (pos + 2 > len) break;
                bl = buf[pos + 1];
                pos += 2;
            } else if (buf[pos] == 0x7A) {
                if (pos + 5 > len) break;
                if (buf[pos + 1] & 0x80) {
                    corrupt_tail = true;
                    break;
                }
                bl = ((size_t)buf[pos + 1] << 24) | ((size_t)buf[pos + 2] << 16) | ((size_t)buf[pos + 3] << 8) | buf[pos + 4];
                pos += 5;
            } else {
                corrupt_tail = buf[pos] < 0x70 || buf[pos] > 0x7E;  // not a type code at all (TC_BASE .. TC_MAX)
                break;
            }
            if (bl > len - pos) bl = len - pos;  // a record cut short by the end of the buffer holds what it holds
            plain.insert(plain.end(), buf + pos, buf + pos + bl);
            pos += bl;
        }
        buf = plain.data();
        len = plain.size();
    }
    Reader r{buf, len};
    m = FmModel();
    r.version();
    m.sample_rate = r.i32();
    m.enable_extract = r.u8() != 0;
    m.bw_suffixes = r.i32();
    m.bw_positions = r.i32();
    m.length = r.i32();
    int32_t nk = r.count(6);
    m.map_keys.resize((size_t)nk);
    m.map_vals.resize((size_t)nk);
    for (int32_t i = 0; i < nk; ++i) {
        m.map_keys[(size_t)i] = r.i32();
        m.map_vals[(size_t)i] = r.i16();
    }
    int32_t nc = r.count(4);
    m.C.resize((size_t)nc);
    for (auto &v : m.C) v = r.i32();
    int32_t nl = r.count(4);
    m.look_up.resize((size_t)nl);
    for (auto &v : m.look_up) v = r.i32();
    read_packed(r, m.suffixes);
    if (m.enable_extract) read_packed(r, m.positions);
    read_rrr(r, m.sampled);
    read_wavelet(r, m.wt);
    if (!r.err && (m.sample_rate <= 0 || m.length <= 0)) r.err = 3;
    if (r.err == 1 && corrupt_tail) r.err = 3;
    if (r.err) {
        // SER:35-36: "Incompatible serial versions! Expected version %d but was %d."
        err = r.err == 2 ? "Incompatible serial versions! Expected version 0 but was " + std::to_string(r.bad_version) + "."
                         : (r.err == 1 ? "truncated stream" : "malformed stream");
        return r.err;
    }
    return 0;
}

// FM:948-975 (+ SER:67-79 framing)
void emit_model(const FmModel &m, bool framed, std::vector<uint8_t> &out) {
    std::vector<uint8_t> raw;
    Writer w{raw};
    w.u8(0);
    w.i32(m.sample_rate);
    w.u8(m.enable_extract ? 1 : 0);
    w.i32(m.bw_suffixes);
    w.i32(m.bw_positions);
    w.i32(m.length);
    w.i32((int32_t)m.map_keys.size());
    for (int i : hashmap_order(m, nullptr)) {
        w.i32(m.map_keys[(size_t)i]);
        w.i16(m.map_vals[(size_t)i]);
    }
    w.i32((int32_t)m.C.size());
    for (auto v : m.C) w.i32(v);
    w.i32((int32_t)m.look_up.size());
    for (auto v : m.look_up) w.i32(v);
    write_packed(w, m.suffixes);
    if (m.enable_extract) write_packed(w, m.positions);
    write_rrr(w, m.sampled);
    write_wavelet(w, m.wt);
    if (!framed) {
        out.swap(raw);
        return;
    }
    out.clear();
    out.reserve(raw.size() + raw.size() / 200 + 16);
    const uint8_t magic[4] = {0xAC, 0xED, 0x00, 0x05};
    out.insert(out.end(), magic, magic + 4);
    for (size_t pos = 0; pos < raw.size();) {
        size_t bl = std::min<size_t>(1024, raw.size() - pos);
        if (bl <= 255) {
            out.push_back(0x77);
            out.push_back((uint8_t)bl);
        } else {
            out.push_back(0x7A);
            out.push_back((uint8_t)(bl >> 24));
            out.push_back((uint8_t)(bl >> 16));
            out.push_back((uint8_t)(bl >> 8));
            out.push_back((uint8_t)bl);
        }
        out.insert(out.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + bl));
        pos += bl;
    }
}

// structural checks on a parsed stream before it is flattened and any kernel may walk the image (a kernel fault can take
// the GPU down); 0 ok, -3 (FMX_E_FORMAT) malformed.  What the image itself must satisfy: validate_blob.
int validate_model(const FmModel &m, std::string &err) {
    auto bad = [&](const char *what) {
        err = std::string("index fails validation: ") + what;
        return -3;
    };
    auto check_rrr = [&](const RrrModel &r) {
        if (r.sample_size <= 0 || r.length < 0 || r.classes.width != 4) return false;
        const int64_t nb = r.length / 15 + (r.length % 15 > 0);
        if (r.classes.length < nb) return false;
        if ((int64_t)r.classes.words.size() < words_for_bits((int64_t)r.classes.length * 4)) return false;
        const int64_t n_rec = r.classes.length / r.sample_size + 1;
        if (r.sampled_offsets.length < n_rec || r.prefix_sums.length < n_rec) return false;
        if (r.bits_per_offset_pos < 1 || r.bits_per_offset_pos > 32 || r.prefix_sums.width < 1 || r.prefix_sums.width > 32)
            return false;
        if (r.sampled_offsets.width < r.bits_per_offset_pos) return false;
        const uint64_t total_bits = (uint64_t)r.offsets.size() * 64;
        for (int64_t k = 0; k < n_rec; ++k)
            if (r.sampled_offsets.get_bits(k * r.sampled_offsets.width, r.bits_per_offset_pos) > total_bits) return false;
        return true;
    };
    if (m.sample_rate <= 0 || m.length <= 0) return bad("sampleRate / length");
    if (m.bw_suffixes < 1 || m.bw_suffixes > 32) return bad("bitWidthSuffixes");
    if (m.enable_extract && (m.bw_positions < 1 || m.bw_positions > 32)) return bad("bitWidthPositions");
    if (m.look_up.empty() || m.C.size() < m.look_up.size()) return bad("cumulativeCounts / monotonicLookUp sizes");
    if (m.map_keys.size() != m.map_vals.size()) return bad("monotonicMap keys / values");
    for (size_t i = 0; i < m.map_vals.size(); ++i) {
        if (m.map_vals[i] < 0 || (size_t)m.map_vals[i] + 1 >= m.C.size() || (size_t)m.map_vals[i] >= m.look_up.size())
            return bad("monotonicMap value outside cumulativeCounts");
        // every code of the map occurs in the BWT, so it is below the wavelet tree's alphabet size (WFBB:133); the
        // plan stage sizes its sort keys and histogram bins by that alphabet
        if (m.map_vals[i] >= m.wt.alphabet_size) return bad("monotonicMap value outside the wavelet tree's alphabet");
        if (m.map_keys[i] < 0 || m.map_keys[i] > 65535) return bad("monotonicMap key is not a char");
    }
    if (m.suffixes.width != m.bw_suffixes || m.suffixes.length < m.length / m.sample_rate + 1) return bad("suffixes");
    if (m.enable_extract && (m.positions.width != m.bw_positions || m.positions.length < m.length / m.sample_rate + 2))
        return bad("positions");
    if (m.sampled.length != m.length || !check_rrr(m.sampled)) return bad("sampledSuffixes");
    const WfbbModel &w = m.wt;
    if (w.size != m.length || w.alphabet_size <= 0 || w.alphabet_size > 32768) return bad("wavelet size / alphabet");
    if ((size_t)w.alphabet_size > m.look_up.size() + 1) return bad("wavelet alphabet larger than monotonicLookUp");
    for (const auto &sb : w.sb) {
        if (sb.block_size_log < 0 || sb.block_size_log > 20 || sb.sigma < -1) return bad("superblock header");
        if (!check_rrr(sb.rank_support)) return bad("superblock RRR");
        for (const auto &bh : sb.block_headers) {
            if (bh.tree_height < 0 || bh.tree_height > 30 || bh.sigma < 0) return bad("block tree height / sigma");
            const int64_t sigma = (int64_t)bh.sigma + 1;
            const int64_t need = (bh.tree_height > 1 ? (bh.tree_height - 1) * 4 : 0) + sigma * 5 + (sigma - 1) * 2;
            if (bh.var_off < 0 || bh.var_off + need > (int64_t)sb.var.size()) return bad("block header offset");
            if (bh.bv_offset < 0 || bh.bv_offset > sb.rank_support.length || bh.bv_rank < 0) return bad("block bitvector offset");
        }
    }
    return 0;
}

}  // namespace fmx
