// sanitizer driver for the host-side product code (builder, serializer, flattener) and the oracle
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "fmx_model.hpp"
extern "C" {
#include "index4j_oracle.h"
}
extern "C" int fmx_synth_log(uint64_t seed, int32_t n, uint16_t *out);
int main() {
    std::vector<uint16_t> text(300000);
    fmx_synth_log(42, (int32_t)text.size(), text.data());
    for (int sr : {1, 5, 32, 64}) {
        fmx::FmModel m;
        std::string err;
        if (fmx::build_model(text.data(), (int32_t)text.size(), sr, true, m, err)) { printf("build failed %s\n", err.c_str()); return 1; }
        std::vector<uint8_t> ser, ser2, blob;
        fmx::emit_model(m, true, ser);
        fmx::FmModel m2;
        if (fmx::parse_model(ser.data(), ser.size(), m2, err)) { printf("parse failed %s\n", err.c_str()); return 1; }
        fmx::emit_model(m2, true, ser2);
        if (ser != ser2) { printf("round trip differs\n"); return 1; }
        if (fmx::flatten_model(m2, blob, err)) { printf("flatten failed %s\n", err.c_str()); return 1; }
        int st = 0;
        OrcFmIndex *o = orc_fm_build(text.data(), (int32_t)text.size(), sr, 1, &st);
        uint8_t *ob; size_t ol;
        orc_fm_write(o, 1, &ob, &ol);
        if (ol != ser.size() || memcmp(ob, ser.data(), ol)) { printf("oracle bytes differ\n"); return 1; }
        // truncated / corrupted streams must be rejected cleanly
        for (size_t cut : {size_t(0), size_t(3), size_t(100), ser.size() / 2, ser.size() - 1}) {
            fmx::FmModel m3;
            (void)fmx::parse_model(ser.data(), cut, m3, err);
            OrcFmIndex *r = orc_fm_read(ser.data(), cut, &st);
            if (r) orc_fm_free(r);
        }
        uint16_t pat[8]; memcpy(pat, text.data() + 1000, 16);
        int c = orc_fm_count(o, pat, 0, 8, &st);
        int32_t locs[16]; int k = orc_fm_locate(o, pat, 0, 8, locs, 16, 16, &st);
        uint16_t dest[256]; int aux;
        int e = orc_fm_extract_until_boundary(o, 0, 5000, dest, 256, 0, '\n', &st, &aux);
        printf("sr %d ok: bytes %zu blob %zu count %d located %d line %d\n", sr, ser.size(), blob.size(), c, k, e);
        orc_free_buffer(ob);
        orc_fm_free(o);
    }
    return 0;
}
