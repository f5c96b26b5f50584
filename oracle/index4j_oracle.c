/*
 * index4j_oracle.c — TEST INFRASTRUCTURE ONLY (see index4j_oracle.h for the rules and for how the
 * oracle is pinned).  Plain C99, single-threaded except orc_fm_count_batch's optional OpenMP loop.
 *
 * Abbreviations used in the citations (all under
 * /root/reference/indices/src/main/java/com/dynatrace/):
 *   FM   fm/FmIndex.java                         WFBB wavelet/WaveletFixedBlockBoosting.java
 *   RRR  bitsequence/RrrVector.java              IV   intsequence/IntVector.java
 *   VIV  intsequence/VariableWidthIntVector.java CMN  intsequence/Common.java
 *   SER  serialization/Serialization.java
 *
 * Java semantics kept on purpose: `long` = int64_t, `int` = int32_t with wrap-around casts,
 * `byte` reads masked to unsigned (WFBB:240 etc.), logical shifts on unsigned copies.
 */
#include "index4j_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* counting mode                                                                              */
/* ------------------------------------------------------------------------------------------ */
static __thread OrcCounters g_cnt;
static OrcCounters g_cnt_total; /* merged by orc_fm_count_batch */

void orc_counters_reset(void) {
    memset(&g_cnt, 0, sizeof g_cnt);
    memset(&g_cnt_total, 0, sizeof g_cnt_total);
}
void orc_counters_get(OrcCounters *out) {
    *out = g_cnt;
    out->lf_steps += g_cnt_total.lf_steps;
    out->alg_bytes += g_cnt_total.alg_bytes;
    out->wt_levels += g_cnt_total.wt_levels;
    out->quirk_runblock_right += g_cnt_total.quirk_runblock_right;
    out->quirk_clamped_right += g_cnt_total.quirk_clamped_right;
    out->rank_calls += g_cnt_total.rank_calls;
    out->absent_superblock += g_cnt_total.absent_superblock;
    out->absent_block += g_cnt_total.absent_block;
    out->absent_scan_steps += g_cnt_total.absent_scan_steps;
    out->run_block += g_cnt_total.run_block;
}
#define CNT_BYTES(n) (g_cnt.alg_bytes += (uint64_t)(n))

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) {
        fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n);
        abort();
    }
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) {
        fprintf(stderr, "oracle: out of memory (%zu x %zu bytes)\n", n, s);
        abort();
    }
    return p;
}

/* ------------------------------------------------------------------------------------------ */
/* intsequence/Common.java                                                                    */
/* ------------------------------------------------------------------------------------------ */
/* CMN:26-93 LOW_BITS_SET[n] */
static inline uint64_t low_bits_set(int n) { return n >= 64 ? ~0ULL : ((1ULL << n) - 1ULL); }
/* CMN:96-161 HIGH_BITS_SET[n] = ~LOW_BITS_SET[n] */
static inline uint64_t high_bits_set(int n) { return ~low_bits_set(n); }
/* CMN:169-175 (=1 for 0, else floor(log2 v)+1; the de-Bruijn lookup at CMN:196-202 is a floor-log2) */
static int minimum_number_of_bits(int64_t value) {
    uint64_t v = (uint64_t)value;
    if (v == 0) return 1;
    int b = 0;
    while (v) {
        ++b;
        v >>= 1;
    }
    return b;
}

/* ------------------------------------------------------------------------------------------ */
/* intsequence/IntVector.java                                                                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t *data;
    int nwords;
    int length;
    int width;
} IntVec;

/* IV:46-55 */
static IntVec *iv_new(int length, int width) {
    IntVec *v = (IntVec *)xmalloc(sizeof *v);
    int64_t bits = (int64_t)length * (int64_t)width;
    v->nwords = (int)((bits % 64 == 0) ? bits / 64 : bits / 64 + 1);
    v->data = (uint64_t *)xcalloc((size_t)v->nwords + 1, 8); /* +1 guard word, never serialized */
    v->length = length;
    v->width = width;
    return v;
}
static void iv_free(IntVec *v) {
    if (v) {
        free(v->data);
        free(v);
    }
}
/* IV:91-119 */
static void iv_set(IntVec *v, int position, int64_t value_) {
    uint64_t value = (uint64_t)value_;
    int64_t bit_position = (int64_t)position * v->width;
    int w = (int)((uint64_t)bit_position >> 6);
    int offset = (int)(bit_position & 63);
    int ew = v->width;
    value &= low_bits_set(ew);
    if (offset + ew < 64) {
        v->data[w] &= ((~0ULL << (offset + ew)) | low_bits_set(offset));
        v->data[w] |= (value << offset);
    } else {
        v->data[w] &= low_bits_set(offset);
        v->data[w] |= (value << offset);
        if (((offset + ew) & 63) > 0) {
            offset = (offset + ew) & 63;
            v->data[w + 1] &= high_bits_set(offset);
            v->data[w + 1] |= (value >> (ew - offset));
        }
    }
}
/* IV:129-143 — position is scaled by the vector's width, the mask uses the caller's `length` */
static inline int64_t iv_get(const IntVec *v, int position, int length) {
    int64_t bit_position = (int64_t)position * v->width;
    int w = (int)((uint64_t)bit_position >> 6);
    int offset = (int)(bit_position & 63);
    uint64_t left = v->data[w] >> offset;
    if (offset + length > 64) {
        uint64_t right = (v->data[w + 1] & low_bits_set((offset + length) & 63)) << (64 - offset);
        return (int64_t)(left | right);
    }
    return (int64_t)(left & low_bits_set(length));
}
static int iv_size_in_bytes(const IntVec *v) { return v->nwords * 8; } /* IV:176-178 */

/* ------------------------------------------------------------------------------------------ */
/* intsequence/VariableWidthIntVector.java                                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t *data;
    int nwords;
} VarVec;

/* VIV:41-47 */
static VarVec *vv_new(int64_t bits_size) {
    VarVec *v = (VarVec *)xmalloc(sizeof *v);
    v->nwords = (int)((bits_size % 64 == 0) ? bits_size / 64 : bits_size / 64 + 1);
    v->data = (uint64_t *)xcalloc((size_t)v->nwords + 1, 8);
    return v;
}
static void vv_free(VarVec *v) {
    if (v) {
        free(v->data);
        free(v);
    }
}
/* VIV:94-118 */
static void vv_set(VarVec *v, int64_t position, int64_t value_, int bits) {
    uint64_t value = (uint64_t)value_;
    int w = (int)((uint64_t)position >> 6);
    int offset = (int)(position & 63);
    value &= low_bits_set(bits);
    if (offset + bits < 64) {
        v->data[w] &= ((~0ULL << (offset + bits)) | low_bits_set(offset));
        v->data[w] |= (value << offset);
    } else {
        v->data[w] &= low_bits_set(offset);
        v->data[w] |= (value << offset);
        if (((offset + bits) & 63) > 0) {
            offset = (offset + bits) & 63;
            v->data[w + 1] &= high_bits_set(offset);
            v->data[w + 1] |= (value >> (bits - offset));
        }
    }
}
/* VIV:127-140 */
static inline int64_t vv_get(const VarVec *v, int64_t position, int length) {
    int w = (int)((uint64_t)position >> 6);
    int offset = (int)(position & 63);
    uint64_t left = v->data[w] >> offset;
    if (offset + length > 64) {
        uint64_t right = (v->data[w + 1] & low_bits_set((offset + length) & 63)) << (64 - offset);
        return (int64_t)(left | right);
    }
    return (int64_t)(left & low_bits_set(length));
}
static int vv_size_in_bytes(const VarVec *v) { return v->nwords * 8; } /* VIV:164-166 */

/* ------------------------------------------------------------------------------------------ */
/* LSB-first bit buffer: stands in for it.unimi.dsi LongArrayBitVector (build side only)      */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t *w;
    int64_t n;
} BitBuf;
static BitBuf bb_new(int64_t n) {
    BitBuf b;
    b.n = n;
    b.w = (uint64_t *)xcalloc((size_t)(n / 64 + 2), 8);
    return b;
}
static inline void bb_set(BitBuf *b, int64_t i, int v) {
    if (v)
        b->w[i >> 6] |= (1ULL << (i & 63));
    else
        b->w[i >> 6] &= ~(1ULL << (i & 63));
}
static inline int bb_get(const BitBuf *b, int64_t i) { return (int)((b->w[i >> 6] >> (i & 63)) & 1ULL); }
/* getLong(from,to): bits [from,to) with bit `from` as the LSB (to-from <= 64) */
static inline uint64_t bb_get_long(const BitBuf *b, int64_t from, int64_t to) {
    int len = (int)(to - from);
    if (len <= 0) return 0;
    int off = (int)(from & 63);
    uint64_t v = b->w[from >> 6] >> off;
    if (off + len > 64) v |= b->w[(from >> 6) + 1] << (64 - off);
    return v & low_bits_set(len);
}

/* ------------------------------------------------------------------------------------------ */
/* bitsequence/RrrVector.java                                                                 */
/* ------------------------------------------------------------------------------------------ */
#define RRR_BLOCK 15 /* RRR:92 */
struct OrcRrr {
    int sample_size;            /* RRR:93 — counted in 15-bit BLOCKS (RRR:278, 326, 368) */
    int length;                 /* RRR:94 */
    int total_ones;             /* RRR:95 */
    IntVec *classes;            /* RRR:96 */
    VarVec *offsets;            /* RRR:97-98 */
    IntVec *sampled_offsets;    /* RRR:99-100 lengthOfSampledOffsets */
    IntVec *prefix_sums;        /* RRR:101 */
    int bits_per_offset_pos;    /* RRR:102-103 */
};

static uint16_t g_offset_of_value[32768];  /* RRR:104, literals RRR:488-8682 — generated */
static uint16_t g_value_of_offset[32768];  /* RRR:106, literals RRR:8705-16900 — generated */
static uint16_t g_card_offsets[16];        /* RRR:105, literals RRR:8692-8697 — generated */
static int g_bits_needed[16];              /* RRR:109-129 */
static int g_tables_ready = 0;
static int g_binom[16][16];

/* RRR:104-129 + the literal tables.  Rule (checked against the literals by
 * tools/check_rrr_tables.py, digests in tests/golden/rrr_tables.json): within class k the
 * offset of a 15-bit value is the rank of its set-bit-position tuple in the lexicographic
 * enumeration of the k-subsets of {0..14}; CARDINALITY_OFFSETS[k] = sum_{j<k} C(15,j). */
static void rrr_tables_init(void) {
    if (g_tables_ready) return;
    for (int n = 0; n < 16; n++) {
        g_binom[n][0] = 1;
        for (int k = 1; k < 16; k++) g_binom[n][k] = (n == 0) ? 0 : g_binom[n - 1][k - 1] + g_binom[n - 1][k];
    }
    int base = 0;
    for (int k = 0; k < 16; k++) {
        g_card_offsets[k] = (uint16_t)base;
        base += g_binom[15][k];
        g_bits_needed[k] = minimum_number_of_bits(g_binom[15][k]); /* RRR:125-127 */
    }
    for (int v = 0; v < 32768; v++) {
        int k = __builtin_popcount((unsigned)v);
        /* lexicographic rank of the sorted position tuple: count tuples that are smaller */
        int rank = 0, remaining = k, prev = -1;
        for (int p = 0; p < 15 && remaining > 0; p++) {
            if (v & (1 << p)) {
                /* all tuples that put a smaller position q in (prev, p) at this slot */
                for (int q = prev + 1; q < p; q++) rank += g_binom[14 - q][remaining - 1];
                prev = p;
                --remaining;
            }
        }
        g_offset_of_value[v] = (uint16_t)rank;
        g_value_of_offset[g_card_offsets[k] + rank] = (uint16_t)v;
    }
    g_tables_ready = 1;
}
const uint16_t *orc_rrr_table_offset_of_value(void) {
    rrr_tables_init();
    return g_offset_of_value;
}
const uint16_t *orc_rrr_table_value_of_offset(void) {
    rrr_tables_init();
    return g_value_of_offset;
}
const uint16_t *orc_rrr_table_cardinality_offsets(void) {
    rrr_tables_init();
    return g_card_offsets;
}
const int *orc_rrr_table_bits_needed(void) {
    rrr_tables_init();
    return g_bits_needed;
}

/* shared body of both public constructors (RRR:154-210 == RRR:225-286 except for numBlocks) */
static OrcRrr *rrr_build(const BitBuf *bv, int sample_size, int num_blocks) {
    rrr_tables_init();
    OrcRrr *r = (OrcRrr *)xmalloc(sizeof *r);
    r->sample_size = sample_size;
    r->length = (int)bv->n;
    int min_bits_for_offset = g_bits_needed[RRR_BLOCK / 2]; /* RRR:229 */
    r->classes = iv_new(num_blocks, minimum_number_of_bits(RRR_BLOCK)); /* RRR:234, MIN_BITS_FOR_CLASS=4 */
    IntVec *temporary = iv_new(num_blocks, min_bits_for_offset);      /* RRR:235 */
    int current_block = 0;
    int64_t total_bits_for_offsets = 0;
    r->total_ones = 0;
    for (int64_t i = 0; i < bv->n; i += RRR_BLOCK) { /* RRR:241-258 */
        int cardinality = 0;
        for (int j = 0; j < RRR_BLOCK && i + j < bv->n; j++) cardinality += bb_get(bv, i + j);
        iv_set(r->classes, current_block, cardinality);
        int64_t hi = i + RRR_BLOCK < bv->n ? i + RRR_BLOCK : bv->n;
        uint64_t block_value = bb_get_long(bv, i, hi);
        int64_t offset = g_offset_of_value[block_value]; /* RRR:250-253: 4 x u16 per long, LSB first */
        iv_set(temporary, current_block, offset);
        total_bits_for_offsets += g_bits_needed[cardinality];
        r->total_ones += cardinality;
        ++current_block;
    }
    r->offsets = vv_new(total_bits_for_offsets);                       /* RRR:261 */
    r->bits_per_offset_pos = minimum_number_of_bits(total_bits_for_offsets); /* RRR:262 */
    r->sampled_offsets = iv_new(num_blocks / sample_size + 1, r->bits_per_offset_pos); /* RRR:263 */
    r->prefix_sums = iv_new(num_blocks / sample_size + 2, minimum_number_of_bits(r->total_ones)); /* RRR:264 */
    int64_t current_bits_for_offset = 0;
    int current_sampled = 0;
    int64_t current_prefix_sum = 0;
    for (int64_t i = 0; i < bv->n; i += RRR_BLOCK) { /* RRR:268-284 */
        int cardinality = 0;
        for (int j = 0; j < RRR_BLOCK && i + j < bv->n; j++) cardinality += bb_get(bv, i + j);
        int64_t too_many = iv_get(temporary, (int)(i / RRR_BLOCK), min_bits_for_offset);
        int how_many = g_bits_needed[cardinality];
        vv_set(r->offsets, current_bits_for_offset, too_many, how_many);
        if ((i / RRR_BLOCK) % sample_size == 0) {
            iv_set(r->sampled_offsets, current_sampled, current_bits_for_offset);
            iv_set(r->prefix_sums, current_sampled++, current_prefix_sum);
        }
        current_bits_for_offset += how_many;
        current_prefix_sum += cardinality;
    }
    iv_set(r->prefix_sums, current_sampled, current_prefix_sum); /* RRR:285 */
    iv_free(temporary);
    return r;
}

/* RRR:225-286 */
static OrcRrr *rrr_from_bitbuf(const BitBuf *bv, int sample_size) {
    int num_blocks = (int)(bv->n / RRR_BLOCK + ((bv->n % RRR_BLOCK > 0) ? 1 : 0)); /* RRR:232 */
    return rrr_build(bv, sample_size, num_blocks);
}
OrcRrr *orc_rrr_from_bits(const uint8_t *bits, int64_t n, int sample) {
    BitBuf b = bb_new(n);
    for (int64_t i = 0; i < n; i++) bb_set(&b, i, bits[i] != 0);
    OrcRrr *r = rrr_from_bitbuf(&b, sample);
    free(b.w);
    return r;
}
/* RRR:143-211 — numBlocks = len/15 + len%15 (over-allocation quirk, RRR:157) */
OrcRrr *orc_rrr_from_ints(const int32_t *ints, int n_ints, int sample) {
    int64_t n = (int64_t)n_ints * 32;
    BitBuf b = bb_new(n);
    for (int64_t i = 0; i < n; i++) bb_set(&b, i, (int)(((uint32_t)ints[i / 32] >> (i % 32)) & 1u)); /* RRR:146-151 */
    int num_blocks = (int)(n / RRR_BLOCK + n % RRR_BLOCK);
    OrcRrr *r = rrr_build(&b, sample, num_blocks);
    free(b.w);
    return r;
}
void orc_rrr_free(OrcRrr *r) {
    if (!r) return;
    iv_free(r->classes);
    vv_free(r->offsets);
    iv_free(r->sampled_offsets);
    iv_free(r->prefix_sums);
    free(r);
}

/* RRR:314-349.  *status = ORC_E_JAVA_AIOOBE stands for the IllegalArgumentException at RRR:316-323. */
int orc_rrr_access(const OrcRrr *r, int position, int *status) {
    if (position < 0 || position >= r->length) {
        if (status) *status = ORC_E_JAVA_AIOOBE;
        return 0;
    }
    int block_id = position / RRR_BLOCK;
    int sampled = block_id / r->sample_size;
    int64_t cur = iv_get(r->sampled_offsets, sampled, r->bits_per_offset_pos);
    CNT_BYTES((r->bits_per_offset_pos + 7) / 8);
    int i;
    for (i = sampled * r->sample_size; i < position / RRR_BLOCK; i++) {
        int c = (int)iv_get(r->classes, i, 4);
        cur += g_bits_needed[c];
    }
    int cardinality = (int)iv_get(r->classes, i, 4);
    CNT_BYTES((i - sampled * r->sample_size + 1 + 1) / 2);
    int nbits = g_bits_needed[cardinality];
    int64_t offset = vv_get(r->offsets, cur, nbits);
    CNT_BYTES((nbits + 7) / 8);
    int64_t reverse = (int64_t)g_card_offsets[cardinality] + offset; /* RRR:342 */
    uint64_t block_value = g_value_of_offset[reverse];               /* RRR:343-345 */
    int bit_id = position % RRR_BLOCK;
    return (int)((block_value >> bit_id) & 1ULL);
}

/* RRR:358-396 */
int orc_rrr_rank_ones(const OrcRrr *r, int position) {
    if (position < 0) return 0;
    if (position >= r->length) return r->total_ones;
    int block_id = position / RRR_BLOCK;
    int sampled = block_id / r->sample_size;
    int prefix = (int)iv_get(r->prefix_sums, sampled, r->prefix_sums->width);
    int cur = (int)iv_get(r->sampled_offsets, sampled, r->bits_per_offset_pos);
    CNT_BYTES((r->prefix_sums->width + 7) / 8 + (r->bits_per_offset_pos + 7) / 8);
    int i;
    for (i = sampled * r->sample_size; i < position / RRR_BLOCK; i++) {
        int c = (int)iv_get(r->classes, i, 4);
        prefix += c;
        cur += g_bits_needed[c];
    }
    int cardinality = (int)iv_get(r->classes, i, 4);
    CNT_BYTES((i - sampled * r->sample_size + 1 + 1) / 2);
    int nbits = g_bits_needed[cardinality];
    int64_t offset = vv_get(r->offsets, cur, nbits);
    CNT_BYTES((nbits + 7) / 8);
    int64_t reverse = (int64_t)g_card_offsets[cardinality] + offset;
    uint64_t block_value = g_value_of_offset[reverse];
    int current_bit_position = i * RRR_BLOCK;
    int num_bits_to_use = position - current_bit_position;
    return prefix + __builtin_popcountll(block_value & low_bits_set(num_bits_to_use));
}
/* rankOnes over an array of positions (the checker / CPU figure of RrrVectorThroughputBenchmark.java:43-51's shape);
 * counting mode: algorithmic bytes of every call, merged over the threads */
void orc_rrr_rank_ones_batch(const OrcRrr *r, const int32_t *positions, int32_t n, int32_t *out, int threads) {
    (void)threads;
#ifdef _OPENMP
    if (threads > 1) {
        uint64_t ab = 0;
#pragma omp parallel num_threads(threads) reduction(+ : ab)
        {
            memset(&g_cnt, 0, sizeof g_cnt);
#pragma omp for schedule(static)
            for (int32_t i = 0; i < n; i++) out[i] = orc_rrr_rank_ones(r, positions[i]);
            ab += g_cnt.alg_bytes;
            memset(&g_cnt, 0, sizeof g_cnt);
        }
        g_cnt_total.alg_bytes += ab;
        return;
    }
#endif
    for (int32_t i = 0; i < n; i++) out[i] = orc_rrr_rank_ones(r, positions[i]);
}
/* RRR:405-410 */
int orc_rrr_rank_zeroes(const OrcRrr *r, int position) {
    if (position < 0) return 0;
    return position - orc_rrr_rank_ones(r, position);
}
/* RRR:418-423 */
int orc_rrr_estimated_memory(const OrcRrr *r) {
    return iv_size_in_bytes(r->classes) + vv_size_in_bytes(r->offsets) + iv_size_in_bytes(r->sampled_offsets) +
           iv_size_in_bytes(r->prefix_sums);
}

/* ------------------------------------------------------------------------------------------ */
/* wavelet/WaveletFixedBlockBoosting.java                                                     */
/* ------------------------------------------------------------------------------------------ */
#define SBS_LOG 20                      /* WFBB:93,97 */
#define SBS (1LL << SBS_LOG)            /* WFBB:98 */
#define HBS (1LL << 32)                 /* WFBB:99 */
#define BLOCK_HEADER_ITEM_SIZE 14       /* WFBB:95 (estimator constant; the record holds 16 bytes) */

typedef struct { /* WFBB:1589-1595 */
    int32_t bv_rank, bv_offset, var_off;
    int16_t sigma, tree_height;
} BlockHdr;

typedef struct { /* WFBB:1621-1629 */
    int16_t sigma, block_size_log;
    OrcRrr *rank_support;
    BlockHdr *block_headers;
    int n_blocks;
    uint8_t *var;
    int var_len;
    int16_t *mapping;
    int mapping_len;
} SuperBlock;

struct OrcWfbb { /* WFBB:105-112 */
    int64_t size;
    int alphabet_size;
    int64_t *count;
    int n_count;
    int64_t *hyper_rank;
    int n_hyper;
    int32_t *super_rank;
    int n_super_rank;
    int16_t *global_mapping;
    int n_global_mapping;
    SuperBlock *sb;
    int n_sb;
    int sampling_rate;
};

static inline int rd16(const uint8_t *p) { return ((p[1] << 8) & 0xff00) | (p[0] & 0xff); }               /* WFBB:240 */
static inline int rd24(const uint8_t *p) { return ((p[2] << 16) & 0xff0000) | ((p[1] << 8) & 0xff00) | (p[0] & 0xff); } /* WFBB:1107 */

/* WFBB:232-248 */
static int64_t compute_symbol_from_block_header(const uint8_t *hdr, int ptr, int64_t code, int64_t code_length) {
    int64_t block_c = 0, temp_code = 0;
    for (int64_t i = 1; i < code_length; ++i) {
        int64_t level_leaf_count = rd16(hdr + ptr);
        CNT_BYTES(2);
        ptr += 4;
        temp_code += level_leaf_count;
        block_c += level_leaf_count;
        temp_code <<= 1;
    }
    block_c += code - temp_code;
    return block_c;
}
/* WFBB:250-278 — returns (code << 32) | codeLength */
static int64_t restore_code_from_block_header(int64_t block_c, const uint8_t *hdr, int ptr, int64_t tree_height) {
    int32_t code = 0;
    int32_t code_length = 1;
    int64_t leaf_count = 0;
    while (code_length < tree_height) {
        code = (int32_t)((uint32_t)code << 1);
        int64_t level_leaf_count = rd16(hdr + ptr);
        CNT_BYTES(2);
        if (leaf_count + level_leaf_count > block_c) {
            code += (int32_t)(block_c - leaf_count);
            break;
        } else {
            code += (int32_t)level_leaf_count;
            ++code_length;
            leaf_count += level_leaf_count;
            ptr += 4;
        }
    }
    if (code_length == tree_height) {
        code = (int32_t)((uint32_t)code << 1);
        code += (int32_t)(block_c - leaf_count);
    }
    return (int64_t)(((uint64_t)(int64_t)code) << 32) | (int64_t)code_length;
}

/* ---- construction ---- */

/* WFBB:324-332 */
static void compute_symbol_freq(const int16_t *text, int64_t from, int64_t len, int64_t *freq, int sigma) {
    for (int i = 0; i < sigma; i++) freq[i] = 0;
    for (int64_t i = 0; i < len; ++i) freq[text[from + i]] += 1;
}

/* WFBB:334-360 with the comparator of WFBB:1684-1707: queue entries are (frequency, symbol list),
 * ordered by frequency, ties by element-wise list comparison.  Restated with an explicit
 * linear-scan minimum (the lists are disjoint, so the order is total and the heap shape is
 * irrelevant). */
typedef struct {
    int64_t key;
    int16_t *list;
    int n;
} HuffItem;
static int huff_cmp(const HuffItem *a, const HuffItem *b) {
    if (a->key < b->key) return -1;
    if (a->key > b->key) return 1;
    int m = a->n < b->n ? a->n : b->n;
    for (int i = 0; i < m; i++) {
        if (a->list[i] < b->list[i]) return -1;
        if (a->list[i] > b->list[i]) return 1;
    }
    return 0;
}
static void compute_huffman_code_lengths(const int64_t *freq, int64_t *code_length, int sigma) {
    for (int i = 0; i < sigma; i++) code_length[i] = 0;
    int cap = 0;
    for (int i = 0; i < sigma; i++)
        if (freq[i] > 0) ++cap;
    if (cap == 0) return;
    HuffItem *pq = (HuffItem *)xmalloc(sizeof(HuffItem) * (size_t)cap);
    int n = 0;
    for (int i = 0; i < sigma; i++) {
        if (freq[i] > 0) {
            pq[n].key = freq[i];
            pq[n].list = (int16_t *)xmalloc(sizeof(int16_t) * (size_t)cap);
            pq[n].list[0] = (int16_t)i;
            pq[n].n = 1;
            ++n;
        }
    }
    while (n > 1) {
        int xi = 0;
        for (int i = 1; i < n; i++)
            if (huff_cmp(&pq[i], &pq[xi]) < 0) xi = i;
        HuffItem x = pq[xi];
        pq[xi] = pq[--n];
        int yi = 0;
        for (int i = 1; i < n; i++)
            if (huff_cmp(&pq[i], &pq[yi]) < 0) yi = i;
        HuffItem y = pq[yi];
        pq[yi] = pq[--n];
        memcpy(x.list + x.n, y.list, sizeof(int16_t) * (size_t)y.n); /* v = x.value; v.addAll(y.value) */
        x.n += y.n;
        free(y.list);
        for (int i = 0; i < x.n; i++) ++code_length[x.list[i]];
        x.key += y.key;
        pq[n++] = x;
    }
    free(pq[0].list);
    free(pq);
}

typedef struct {
    int64_t key;
    int16_t value;
} LSTuple;
/* WFBB:1709-1718 */
static int ls_cmp(const void *a_, const void *b_) {
    const LSTuple *a = (const LSTuple *)a_, *b = (const LSTuple *)b_;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    if (a->value != b->value) return a->value < b->value ? -1 : 1;
    return 0;
}
/* sorted (codeLength, symbol) list of the symbols present — WFBB:423-431, 541-547, 763-769 */
static int sorted_symbols(const int64_t *freq, const int64_t *code_length, int sigma, LSTuple *sym) {
    int n = 0;
    for (int i = 0; i < sigma; i++)
        if (freq[i] > 0) {
            sym[n].key = code_length[i];
            sym[n].value = (int16_t)i;
            ++n;
        }
    qsort(sym, (size_t)n, sizeof(LSTuple), ls_cmp);
    return n;
}
/* WFBB:537-555 */
static void assign_canonical_huffman_codes(const int64_t *freq, const int64_t *code_length, int64_t *code, int sigma) {
    for (int i = 0; i < sigma; i++) code[i] = 0;
    LSTuple *sym = (LSTuple *)xmalloc(sizeof(LSTuple) * (size_t)sigma);
    int n = sorted_symbols(freq, code_length, sigma, sym);
    int64_t c = 0;
    for (int i = 0; i < n; ++i) {
        if (i != 0) c = (c + 1) << (sym[i].key - sym[i - 1].key);
        code[sym[i].value] = c;
    }
    free(sym);
}
static int list_count_equal(const int64_t *list, int from, int to, int64_t value) { /* WFBB:993-1001 */
    int c = 0;
    for (int i = from; i < to; i++)
        if (list[i] == value) ++c;
    return c;
}
static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* WFBB:570-810 */
static void encode_block(const OrcWfbb *w, const int16_t *text, int64_t block_ptr, const int64_t *block_rank,
                         int64_t block_size, BitBuf *sb_bv, int *ones_count, int64_t sb_bv_offset,
                         uint8_t *hdr_data, int64_t hdr_ptr) {
    int sigma = w->alphabet_size;
    int64_t *freq = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)sigma);
    int64_t *code_length = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)sigma);
    int64_t *code = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)sigma);
    compute_symbol_freq(text, block_ptr, block_size, freq, sigma);
    compute_huffman_code_lengths(freq, code_length, sigma);
    assign_canonical_huffman_codes(freq, code_length, code, sigma);
    int64_t max_code_length = -1;
    for (int i = 0; i < sigma; i++)
        if (code_length[i] > max_code_length) max_code_length = code_length[i];

    *ones_count = 0;
    int64_t *ones_in_bv = (int64_t *)xcalloc((size_t)sigma, sizeof(int64_t)); /* WFBB:597-600 */

    if (list_count_equal(freq, 0, sigma, 0) < sigma - 1) { /* WFBB:604: more than one symbol */
        /* internal node ids, WFBB:614-626 */
        size_t cap = 0;
        for (int i = 0; i < sigma; i++)
            if (freq[i] > 0) cap += (size_t)code_length[i];
        int64_t *ids = (int64_t *)xmalloc(sizeof(int64_t) * (cap ? cap : 1));
        size_t n_ids = 0;
        for (int i = 0; i < sigma; i++)
            if (freq[i] > 0)
                for (int64_t depth = 0; depth < code_length[i]; ++depth)
                    ids[n_ids++] = ((1LL << code_length[i]) | code[i]) >> (code_length[i] - depth);
        qsort(ids, n_ids, sizeof(int64_t), cmp_i64);
        size_t u = 0;
        for (size_t i = 0; i < n_ids; i++) /* removeConsecutives, WFBB:557-568 */
            if (i == 0 || ids[i] != ids[u - 1]) ids[u++] = ids[i];
        n_ids = u;
        /* node id -> bitvector index, WFBB:631-637 */
        size_t n_nodes = (size_t)1 << max_code_length;
        int64_t *bv_id = (int64_t *)xcalloc(n_nodes, sizeof(int64_t));
        for (size_t i = 0; i < n_ids; ++i) bv_id[ids[i]] = (int64_t)i;
        /* sizes, WFBB:640-656 */
        int64_t *bv_size = (int64_t *)xcalloc(n_ids, sizeof(int64_t));
        for (int i = 0; i < sigma; i++)
            if (freq[i] > 0)
                for (int64_t depth = 0; depth < code_length[i]; ++depth) {
                    int64_t id = ((1LL << code_length[i]) | code[i]) >> (code_length[i] - depth);
                    bv_size[bv_id[id]] += freq[i];
                }
        /* allocate, WFBB:659-666 */
        uint8_t **bv = (uint8_t **)xmalloc(sizeof(uint8_t *) * n_ids);
        for (size_t i = 0; i < n_ids; i++) bv[i] = (uint8_t *)xcalloc((size_t)bv_size[i], 1);
        /* fill, WFBB:669-700 */
        int64_t *visit = (int64_t *)xcalloc((size_t)1 << (max_code_length + 1), sizeof(int64_t));
        for (int64_t i = 0; i < block_size; ++i) {
            int16_t sym = text[block_ptr + i];
            int64_t pos = i;
            for (int64_t depth = 0; depth < code_length[sym]; ++depth) {
                int64_t id = (int64_t)(((uint64_t)((1LL << code_length[sym]) | code[sym])) >> (code_length[sym] - depth));
                if (depth > 0) {
                    pos -= visit[id ^ 1];
                    visit[id] += 1;
                }
                if ((code[sym] & (1LL << (code_length[sym] - depth - 1))) != 0) {
                    bv[bv_id[id]][pos] = 1;
                    ones_in_bv[bv_id[id]] += 1;
                    *ones_count += 1;
                }
            }
            visit[(1LL << code_length[sym]) | code[sym]] += 1;
        }
        /* append, WFBB:703-708 */
        for (size_t i = 0; i < n_ids; ++i)
            for (int64_t j = 0; j < bv_size[i]; ++j) {
                bb_set(sb_bv, sb_bv_offset, bv[i][j]);
                ++sb_bv_offset;
            }
        for (size_t i = 0; i < n_ids; i++) free(bv[i]);
        free(bv);
        free(visit);
        free(bv_size);
        free(bv_id);
        free(ids);
    }

    /* variable-size block header, WFBB:713-760 */
    int64_t mcl = max_code_length > 0 ? max_code_length : 0;
    int64_t *clf = (int64_t *)xcalloc((size_t)mcl + 1, sizeof(int64_t));
    for (int i = 0; i < sigma; i++)
        if (freq[i] > 0 && code_length[i] < max_code_length) clf[code_length[i]] += 1;
    int64_t *ltf = (int64_t *)xcalloc((size_t)mcl + 1, sizeof(int64_t));
    for (int i = 0; i < sigma; i++)
        if (freq[i] > 0)
            for (int64_t depth = 1; depth < code_length[i]; ++depth) ltf[depth] += freq[i];
    int64_t bp = hdr_ptr;
    for (int64_t depth = 1; depth < max_code_length; ++depth) {
        int16_t what = (int16_t)(int32_t)clf[depth];
        int16_t value = (int16_t)(ltf[depth] - 1);
        hdr_data[bp++] = (uint8_t)(what & 0xff);
        hdr_data[bp++] = (uint8_t)(((uint32_t)(int32_t)what >> 8) & 0xff);
        hdr_data[bp++] = (uint8_t)(value & 0xff);
        hdr_data[bp++] = (uint8_t)(((uint32_t)(int32_t)value >> 8) & 0xff);
    }
    /* leaves, WFBB:763-788 */
    LSTuple *sym = (LSTuple *)xmalloc(sizeof(LSTuple) * (size_t)sigma);
    int n_sym = sorted_symbols(freq, code_length, sigma, sym);
    for (int i = 0; i < n_sym; ++i) {
        int16_t symbol = sym[i].value;
        int64_t rank_value = block_rank[symbol];
        hdr_data[bp++] = (uint8_t)(symbol & 0xff);
        hdr_data[bp++] = (uint8_t)(((uint32_t)(int32_t)symbol >> 8) & 0xff);
        hdr_data[bp++] = (uint8_t)(rank_value & 0xff);
        hdr_data[bp++] = (uint8_t)(((uint64_t)rank_value >> 8) & 0xff);
        hdr_data[bp++] = (uint8_t)(((uint64_t)rank_value >> 16) & 0xff);
    }
    /* cumulative one-counts per level, WFBB:793-809 */
    int64_t n_internal = 1;
    int64_t ptr = 0;
    for (int64_t depth = 0; depth < max_code_length; ++depth) {
        int64_t one_bits = 0;
        for (int64_t j = 0; j < n_internal; ++j) {
            one_bits += ones_in_bv[ptr++];
            int16_t value = (int16_t)one_bits; /* u16 wrap, WFBB:798 */
            hdr_data[bp++] = (uint8_t)(value & 0xff);
            hdr_data[bp++] = (uint8_t)(((uint32_t)(int32_t)value >> 8) & 0xff);
        }
        if (depth + 1 != max_code_length) {
            int64_t next_level_leaf_count = clf[depth + 1];
            n_internal <<= 1;
            n_internal -= next_level_leaf_count;
        }
    }
    free(sym);
    free(ltf);
    free(clf);
    free(ones_in_bv);
    free(code);
    free(code_length);
    free(freq);
}

/* WFBB:362-535 */
static void encode_blocks_in_superblock(OrcWfbb *w, const int16_t *text, int64_t sb_ptr, int64_t sb_id,
                                        int64_t block_size_log) {
    SuperBlock *sb = &w->sb[sb_id];
    int sigma_g = w->alphabet_size;
    int64_t block_size = 1LL << block_size_log;
    int64_t sb_beg = sb_id * SBS;
    int64_t sb_end = sb_beg + SBS < w->size ? sb_beg + SBS : w->size;
    int64_t sb_size = sb_end - sb_beg;
    int64_t sb_sigma = (int64_t)sb->sigma + 1;
    sb->block_size_log = (int16_t)block_size_log;

    sb->mapping_len = (int)(sb_sigma * (SBS / block_size)); /* WFBB:377-387 */
    sb->mapping = (int16_t *)xmalloc(sizeof(int16_t) * (size_t)(sb->mapping_len ? sb->mapping_len : 1));
    for (int i = 0; i < sb->mapping_len; i++) sb->mapping[i] = (int16_t)(sigma_g - 1);

    int64_t sb_bv_size = 0, var_size = 0;
    int64_t n_blocks = (sb_size + block_size - 1) / block_size;
    sb->n_blocks = (int)n_blocks;
    sb->block_headers = (BlockHdr *)xcalloc((size_t)n_blocks, sizeof(BlockHdr));

    int64_t *freq = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)sigma_g);
    int64_t *code_length = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)sigma_g);
    int16_t *g2b = (int16_t *)xmalloc(sizeof(int16_t) * (size_t)sigma_g);
    LSTuple *sym = (LSTuple *)xmalloc(sizeof(LSTuple) * (size_t)sigma_g);

    for (int64_t block_id = 0; block_id < n_blocks; ++block_id) { /* WFBB:400-484 */
        BlockHdr *bh = &sb->block_headers[block_id];
        int64_t block_beg = block_id * block_size;
        int64_t block_end = block_beg + block_size < sb_size ? block_beg + block_size : sb_size;
        int64_t this_block_size = block_end - block_beg;
        int64_t block_ptr = sb_ptr + block_beg;
        int64_t bv_size = 0;
        for (int i = 0; i < sigma_g; i++) g2b[i] = (int16_t)sigma_g;
        compute_symbol_freq(text, block_ptr, this_block_size, freq, sigma_g);
        compute_huffman_code_lengths(freq, code_length, sigma_g);
        int n_sym = sorted_symbols(freq, code_length, sigma_g, sym);
        int64_t sigma = n_sym;
        int64_t tree_height = -1;
        for (int i = 0; i < sigma_g; i++)
            if (code_length[i] > tree_height) tree_height = code_length[i];
        for (int i = 0; i < n_sym; ++i) {
            g2b[sym[i].value] = (int16_t)i;
            if (n_sym > 1) bv_size += freq[sym[i].value] * code_length[sym[i].value];
        }
        bh->bv_offset = (int32_t)sb_bv_size;
        bh->var_off = (int32_t)var_size;
        bh->tree_height = (int16_t)tree_height;
        bh->sigma = (int16_t)(sigma - 1);
        for (int i = 0; i < sigma_g; ++i) { /* WFBB:457-473 */
            if (g2b[i] != sigma_g) {
                int16_t sb_char = w->global_mapping[sb_id * sigma_g + i];
                int64_t address = (int64_t)sb_char * (SBS / block_size) + block_id;
                int16_t a = (int16_t)(sigma_g - 2), b = g2b[i];
                sb->mapping[address] = (int16_t)(a < b ? a : b); /* Math.min on shorts promoted to int */
            }
        }
        sb_bv_size += bv_size;
        if (tree_height > 1) var_size += (tree_height - 1) * 4;
        var_size += sigma * 5;
        var_size += (sigma - 1) * 2;
    }

    sb->var_len = (int)var_size;
    sb->var = (uint8_t *)xcalloc((size_t)var_size + 16, 1);

    int64_t bv_rank = 0;
    BitBuf sb_bv = bb_new(sb_bv_size);
    int64_t *block_rank = (int64_t *)xcalloc((size_t)sigma_g, sizeof(int64_t));
    for (int64_t block_id = 0; block_id < n_blocks; ++block_id) { /* WFBB:499-531 */
        BlockHdr *bh = &sb->block_headers[block_id];
        int64_t block_beg = block_id * block_size;
        int64_t block_end = block_beg + block_size < sb_size ? block_beg + block_size : sb_size;
        int64_t this_block_size = block_end - block_beg;
        int64_t block_ptr = sb_ptr + block_beg;
        int ones_count = 0;
        encode_block(w, text, block_ptr, block_rank, this_block_size, &sb_bv, &ones_count, bh->bv_offset, sb->var,
                     bh->var_off);
        bh->bv_rank = (int32_t)bv_rank;
        bv_rank += ones_count;
        for (int64_t i = 0; i < this_block_size; ++i) block_rank[text[block_ptr + i]] += 1;
    }
    sb->rank_support = rrr_from_bitbuf(&sb_bv, w->sampling_rate); /* WFBB:534 */
    free(sb_bv.w);
    free(block_rank);
    free(sym);
    free(g2b);
    free(code_length);
    free(freq);
}

/* RRR(all-zero bitvector of length n).getEstimatedMemoryUsage(), WFBB:961-965 -> RRR:225-286, 418-423 */
static int64_t rrr_all_zero_estimated_memory(int64_t n, int sample) {
    rrr_tables_init();
    int64_t num_blocks = n / RRR_BLOCK + ((n % RRR_BLOCK > 0) ? 1 : 0);
    int64_t total_bits = num_blocks * g_bits_needed[0];
    int64_t words = 0, bits;
    bits = num_blocks * 4;
    words += (bits % 64 == 0) ? bits / 64 : bits / 64 + 1;
    words += (total_bits % 64 == 0) ? total_bits / 64 : total_bits / 64 + 1;
    bits = (num_blocks / sample + 1) * (int64_t)minimum_number_of_bits(total_bits);
    words += (bits % 64 == 0) ? bits / 64 : bits / 64 + 1;
    bits = (num_blocks / sample + 2) * (int64_t)minimum_number_of_bits(0);
    words += (bits % 64 == 0) ? bits / 64 : bits / 64 + 1;
    return (int64_t)(int32_t)(words * 8);
}

/* WFBB:812-991 */
static void encode_super_block(OrcWfbb *w, const int16_t *text, int64_t sb_ptr, int64_t sb_id) {
    int sigma = w->alphabet_size;
    int64_t sb_beg = sb_id * SBS;
    int64_t sb_end = sb_beg + SBS < w->size ? sb_beg + SBS : w->size;
    int64_t sb_size = sb_end - sb_beg;
    int64_t hb_id = (sb_id * SBS) / HBS;
    if (sb_id * SBS % HBS == 0)
        for (int i = 0; i < sigma; ++i) w->hyper_rank[hb_id * sigma + i] = w->count[i];
    for (int i = 0; i < sigma; ++i)
        w->super_rank[sb_id * sigma + i] = (int32_t)(w->count[i] - w->hyper_rank[hb_id * sigma + i]);
    for (int64_t i = 0; i < sb_size; ++i) w->count[text[sb_ptr + i]] += 1;
    int64_t sb_sigma = 0;
    for (int i = 0; i < sigma; ++i)
        if (w->super_rank[sb_id * sigma + i] + w->hyper_rank[hb_id * sigma + i] != w->count[i])
            w->global_mapping[sb_id * sigma + i] = (int16_t)sb_sigma++;
    w->sb[sb_id].sigma = (int16_t)(sb_sigma - 1);

    /* block-size search, WFBB:853-987 */
    int64_t best_log = 0, best_size = 0;
    int64_t smallest_log = (SBS_LOG < 16 ? SBS_LOG : 16) - 7;
    if (smallest_log < 0) smallest_log = 0;
    int64_t smallest = 1LL << smallest_log;
    int64_t max_blocks = SBS / smallest;
    int64_t **freq = (int64_t **)xmalloc(sizeof(int64_t *) * (size_t)max_blocks);
    for (int64_t i = 0; i < max_blocks; i++) freq[i] = (int64_t *)xcalloc((size_t)sigma, sizeof(int64_t));
    int64_t *code_length = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)sigma);
    int64_t compressed = 0, prev_uncompressed = 0;
    int64_t top_log = SBS_LOG < 16 ? SBS_LOG : 16;
    for (int64_t bsl = smallest_log; bsl <= top_log; ++bsl) {
        int64_t block_size = 1LL << bsl;
        int64_t n_blocks = (sb_size + block_size - 1) / block_size;
        int64_t enc = BLOCK_HEADER_ITEM_SIZE * n_blocks + sb_sigma * (SBS / block_size);
        if (bsl == smallest_log) {
            for (int64_t b = 0; b < n_blocks; ++b) {
                int64_t beg = b * block_size;
                int64_t end = beg + block_size < sb_size ? beg + block_size : sb_size;
                compute_symbol_freq(text, sb_ptr + beg, end - beg, freq[b], sigma);
            }
        } else {
            int64_t prev_blocks = (sb_size + (block_size / 2) - 1) / (block_size / 2);
            for (int64_t b = 0; b < prev_blocks; b += 2)
                for (int c = 0; c < sigma; ++c) {
                    int64_t prev = freq[b][c];
                    int64_t sum = (b + 1 < prev_blocks) ? freq[b + 1][c] : 0;
                    freq[b >> 1][c] = prev + sum;
                }
        }
        for (int64_t b = 0; b < n_blocks; ++b) {
            int64_t zeros = list_count_equal(freq[b], 0, sigma, 0);
            int64_t block_sigma = sigma - zeros;
            enc += block_sigma * 4;
            enc += (block_sigma - 1) * 2;
        }
        int64_t uncompressed = 0;
        for (int64_t b = 0; b < n_blocks; ++b) {
            compute_huffman_code_lengths(freq[b], code_length, sigma);
            int64_t mcl = -1;
            for (int i = 0; i < sigma; i++)
                if (code_length[i] > mcl) mcl = code_length[i];
            if (mcl > 1) enc += (mcl - 1) * 3;
            for (int c = 0; c < sigma; ++c) uncompressed += freq[b][c] * code_length[c];
        }
        if (uncompressed > 0) {
            if (bsl == smallest_log) {
                compressed = rrr_all_zero_estimated_memory(uncompressed, w->sampling_rate);
            } else {
                /* WFBB:967-971: (long)((double)c * ((double)u / (double)prev)); Java's (long) of NaN is 0,
                 * of +-Infinity saturates. */
                double scaling = (double)uncompressed / (double)prev_uncompressed;
                double prod = (double)compressed * scaling;
                if (prod != prod)
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
    free(code_length);
    for (int64_t i = 0; i < max_blocks; i++) free(freq[i]);
    free(freq);
    encode_blocks_in_superblock(w, text, sb_ptr, sb_id, best_log);
}

/* WFBB:130-154 */
OrcWfbb *orc_wfbb_build(const int16_t *text, int64_t n, int sampling_rate) {
    OrcWfbb *w = (OrcWfbb *)xcalloc(1, sizeof *w);
    w->size = n;
    w->sampling_rate = sampling_rate;
    int mx = INT32_MIN;
    for (int64_t i = 0; i < n; i++)
        if (mx < text[i]) mx = text[i];
    w->alphabet_size = mx + 1;
    int sigma = w->alphabet_size;
    w->n_count = sigma;
    w->count = (int64_t *)xcalloc((size_t)sigma, sizeof(int64_t));
    int64_t n_sb = (n + SBS - 1) / SBS;
    int64_t n_hb = (n + HBS - 1) / HBS;
    w->n_hyper = (int)(n_hb * sigma);
    w->hyper_rank = (int64_t *)xcalloc((size_t)w->n_hyper, sizeof(int64_t));
    w->n_super_rank = (int)(n_sb * sigma);
    w->super_rank = (int32_t *)xcalloc((size_t)w->n_super_rank, sizeof(int32_t));
    w->n_global_mapping = (int)(n_sb * sigma);
    w->global_mapping = (int16_t *)xmalloc(sizeof(int16_t) * (size_t)(w->n_global_mapping ? w->n_global_mapping : 1));
    for (int i = 0; i < w->n_global_mapping; i++) w->global_mapping[i] = (int16_t)(sigma - 1);
    w->n_sb = (int)n_sb;
    w->sb = (SuperBlock *)xcalloc((size_t)n_sb, sizeof(SuperBlock));
    for (int64_t sb_id = 0; sb_id < n_sb; ++sb_id) encode_super_block(w, text, sb_id * SBS, sb_id);
    return w;
}
void orc_wfbb_free(OrcWfbb *w) {
    if (!w) return;
    for (int i = 0; i < w->n_sb; i++) {
        orc_rrr_free(w->sb[i].rank_support);
        free(w->sb[i].block_headers);
        free(w->sb[i].var);
        free(w->sb[i].mapping);
    }
    free(w->sb);
    free(w->count);
    free(w->hyper_rank);
    free(w->super_rank);
    free(w->global_mapping);
    free(w);
}
int orc_wfbb_block_size_log(const OrcWfbb *w, int superblock) { return w->sb[superblock].block_size_log; }

/* ---- queries ---- */

/* WFBB:1010-1285.  *status receives ORC_E_JAVA_AIOOBE where the JVM would raise
 * ArrayIndexOutOfBoundsException (quirk Q3: position == size with size % 2^20 == 0). */
int64_t orc_wfbb_rank(const OrcWfbb *w, int64_t position, int16_t symbol, int *status) {
    if (position == 0) return 0;
    if (position > w->size) position = w->size;
    if (symbol >= w->alphabet_size) return 0;
    int sigma_g = w->alphabet_size;
    int64_t hb_id = position / HBS;
    int64_t sb_id = position / SBS;
    if (sb_id >= w->n_sb || symbol < 0) { /* globalMapping[...] / superBlockHeaderItems[...] out of range */
        if (status) *status = ORC_E_JAVA_AIOOBE;
        return 0;
    }
    int16_t sb_c = w->global_mapping[sb_id * sigma_g + symbol];
    int64_t sb_index = position % SBS;
    const SuperBlock *sb = &w->sb[sb_id];
    int64_t sb_sigma = (int64_t)sb->sigma + 1;
    int64_t bsl = sb->block_size_log;
    int64_t block_size = 1LL << bsl;
    int64_t blocks_in_sb_log = SBS_LOG - bsl;
    int64_t block_index = position & (block_size - 1);
    int64_t cur_block_size = block_size < w->size - (position - block_index) ? block_size : w->size - (position - block_index);
    int64_t block_id = (int64_t)((uint64_t)sb_index >> bsl);
    int64_t r_sb = w->super_rank[sb_id * sigma_g + symbol];
    int64_t r_hb = w->hyper_rank[hb_id * sigma_g + symbol];
    CNT_BYTES(2 + 4 + 4 + 8);
    g_cnt.rank_calls++;

    if (sb_c >= sb_sigma) { /* WFBB:1040-1042 */
        g_cnt.absent_superblock++;
        return r_hb + r_sb;
    }

    int16_t block_c = sb->mapping[((int64_t)sb_c << blocks_in_sb_log) + block_id];
    CNT_BYTES(2);
    if (block_c == sigma_g - 1) { /* WFBB:1048-1110 */
        g_cnt.absent_block++;
        ++block_id;
        int64_t blocks_in_sb = 1LL << blocks_in_sb_log;
        while (block_id < blocks_in_sb) {
            CNT_BYTES(2);
            g_cnt.absent_scan_steps++;
            if (sb->mapping[((int64_t)sb_c << blocks_in_sb_log) + block_id] != sigma_g - 1) break;
            ++block_id;
        }
        if (block_id == blocks_in_sb) {
            if ((sb_id + 1) * SBS >= w->size) {
                CNT_BYTES(8);
                return w->count[symbol];
            } else {
                CNT_BYTES(4);
                return r_hb + w->super_rank[(sb_id + 1) * sigma_g + symbol];
            }
        } else {
            block_c = sb->mapping[((int64_t)sb_c << blocks_in_sb_log) + block_id];
            const BlockHdr *bh = &sb->block_headers[block_id];
            int64_t tree_height = bh->tree_height;
            int32_t p = (int32_t)bh->var_off;
            p += (int32_t)((tree_height - 1) * 4); /* WFBB:1081 — no treeHeight>0 guard: -4 for run blocks */
            CNT_BYTES(16 + 3);
            if (tree_height == 0) g_cnt.quirk_runblock_right++;
            if (block_c == sigma_g - 2) g_cnt.quirk_clamped_right++;
            int64_t idx = (int64_t)p + (int64_t)block_c * 5 + 2;
            if (idx < 0 || idx + 2 >= sb->var_len) {
                if (status) *status = ORC_E_JAVA_AIOOBE;
                return 0;
            }
            int64_t r_b = ((sb->var[idx + 2] << 16) & 0xff0000) | ((sb->var[idx + 1] << 8) & 0xff00) | (sb->var[idx] & 0xff);
            return r_hb + r_sb + r_b;
        }
    }

    /* WFBB:1112-1138 */
    const BlockHdr *bh = &sb->block_headers[block_id];
    int64_t var_off = bh->var_off;
    int64_t tree_height = bh->tree_height;
    int32_t vptr = (int32_t)var_off;
    int32_t vtmp = vptr;
    if (tree_height > 0) vtmp += (int32_t)((tree_height - 1) * 4);
    int value = rd16(sb->var + vtmp + 5 * block_c);
    if (value != symbol) ++block_c; /* WFBB:1128-1130 clamped-mapping fix-up */
    int64_t r_b = rd24(sb->var + vtmp + block_c * 5 + 2);
    CNT_BYTES(16 + 5);

    if (tree_height == 0) { /* WFBB:1141-1146 */
        g_cnt.run_block++;
        return r_hb + r_sb + r_b + block_index;
    }

    int64_t code_result = restore_code_from_block_header(block_c, sb->var, (int)var_off, tree_height);
    int32_t code = (int32_t)((uint64_t)code_result >> 32);
    int32_t code_length = (int32_t)code_result;

    int64_t bv_rank = bh->bv_rank;
    int64_t bv_offset = bh->bv_offset;
    int64_t internal_nodes = 1;
    int64_t left_siblings = 0;
    int64_t left_total_bv = 0;
    int64_t node_bv_size = cur_block_size;
    int64_t depth_total_bv = node_bv_size;
    int64_t node_rank = block_index;
    int64_t block_sigma = (int64_t)bh->sigma + 1;
    vtmp = vptr;
    vtmp += (int32_t)((tree_height - 1) * 4);
    vtmp += (int32_t)(block_sigma * 5);
    int32_t second = vtmp;

    for (int64_t depth = 0; depth < code_length; ++depth) { /* WFBB:1185-1279 */
        g_cnt.wt_levels++;
        int64_t rank1 = orc_rrr_rank_ones(sb->rank_support, (int32_t)(bv_offset + left_total_bv + node_rank));
        int64_t left_ones = 0;
        if (left_siblings > 0) {
            left_ones = rd16(sb->var + (int32_t)(second + 2 * (left_siblings - 1)));
            CNT_BYTES(2);
        }
        rank1 -= bv_rank + left_ones;
        int64_t node_ones = rd16(sb->var + (int32_t)(second + 2 * left_siblings)) - left_ones;
        int64_t node_zeros = node_bv_size - node_ones;
        int64_t rank0 = node_rank - rank1;
        bv_rank += rd16(sb->var + (int32_t)(second + 2 * (internal_nodes - 1)));
        CNT_BYTES(4);
        second += (int32_t)(2 * internal_nodes);
        left_siblings <<= 1;
        int64_t next_bit = (code & (1LL << (code_length - depth - 1)));
        if (next_bit != 0) {
            node_rank = rank1;
            node_bv_size = node_ones;
            ++left_siblings;
            left_total_bv += node_zeros;
        } else {
            node_rank = rank0;
            node_bv_size = node_zeros;
        }
        if (depth + 1 != code_length) {
            int64_t next_leaf_count = rd16(sb->var + vptr);
            vptr += 2;
            int64_t next_total_bv = rd16(sb->var + vptr) + 1;
            vptr += 2;
            CNT_BYTES(4);
            left_total_bv -= (depth_total_bv - next_total_bv);
            bv_offset += depth_total_bv;
            depth_total_bv = next_total_bv;
            internal_nodes <<= 1;
            internal_nodes -= next_leaf_count;
            left_siblings -= next_leaf_count;
        }
    }
    return r_hb + r_sb + r_b + node_rank;
}

/* WFBB:1305-1537 */
int64_t orc_wfbb_inverse_select(const OrcWfbb *w, int64_t position) {
    int sigma_g = w->alphabet_size;
    int64_t hb_id = position / HBS;
    int64_t sb_id = position / SBS;
    int64_t sb_index = position % SBS;
    const SuperBlock *sb = &w->sb[sb_id];
    int64_t bsl = sb->block_size_log;
    int64_t block_size = 1LL << bsl;
    int64_t block_index = position & (block_size - 1);
    int64_t cur_block_size = block_size < w->size - (position - block_index) ? block_size : w->size - (position - block_index);
    int64_t block_id = sb_index >> bsl;
    const BlockHdr *bh = &sb->block_headers[block_id];
    int64_t var_off = bh->var_off;
    int64_t tree_height = bh->tree_height;
    int32_t p8 = (int32_t)var_off;
    int32_t copy_p8 = p8;
    int32_t tmp8 = p8;
    if (tree_height > 0) tmp8 += (int32_t)((tree_height - 1) * 4);
    int32_t p32 = tmp8;
    CNT_BYTES(4 + 16);

    if (tree_height == 0) { /* WFBB:1329-1355 */
        int c = (((((int16_t)(int8_t)sb->var[p32 + 1]) << 8) & 0x00ff00) | ((int16_t)(int8_t)sb->var[p32])) & 0x00ff; /* WFBB:1332: 8-bit mask (quirk Q1) */
        CNT_BYTES(2);
        if (position == 0) return c;
        int64_t r_b = rd24(sb->var + p32 + 2);
        int64_t r_sb = w->super_rank[sb_id * sigma_g + c];
        int64_t r_hb = w->hyper_rank[hb_id * sigma_g + c];
        CNT_BYTES(3 + 4 + 8);
        int64_t result = r_hb + r_sb + r_b + block_index;
        return (int64_t)((uint64_t)result << 32) | c;
    }

    int64_t code = 0, code_length = 0;
    int64_t bv_rank = bh->bv_rank;
    int64_t bv_offset = bh->bv_offset;
    int64_t internal_nodes = 1;
    int64_t left_siblings = 0;
    int64_t left_total_bv = 0;
    int64_t node_bv_size = cur_block_size;
    int64_t depth_total_bv = node_bv_size;
    int64_t node_rank = block_index;
    int64_t block_sigma = (int64_t)bh->sigma + 1;
    tmp8 = p8;
    tmp8 += (int32_t)((tree_height - 1) * 4);
    tmp8 += (int32_t)(block_sigma * 5);
    int32_t second = tmp8;

    for (int64_t depth = 0;; ++depth) { /* WFBB:1386-1493 */
        g_cnt.wt_levels++;
        int32_t rank_position = (int32_t)(bv_offset + left_total_bv + node_rank);
        int64_t rank1 = orc_rrr_rank_ones(sb->rank_support, rank_position);
        uint64_t save = g_cnt.alg_bytes; /* the access at the same bit re-reads the same fields: +0 bytes */
        int next_bit = orc_rrr_access(sb->rank_support, rank_position, NULL);
        g_cnt.alg_bytes = save;
        int64_t left_ones = 0;
        if (left_siblings > 0) {
            left_ones = rd16(sb->var + (int32_t)(second + 2 * (left_siblings - 1)));
            CNT_BYTES(2);
        }
        rank1 -= bv_rank + left_ones;
        int64_t node_ones = rd16(sb->var + (int32_t)(second + 2 * left_siblings)) - left_ones;
        int64_t node_zeros = node_bv_size - node_ones;
        int64_t rank0 = node_rank - rank1;
        bv_rank += rd16(sb->var + (int32_t)(second + 2 * (internal_nodes - 1)));
        CNT_BYTES(4);
        second += (int32_t)(internal_nodes * 2);
        left_siblings <<= 1;
        code <<= 1;
        ++code_length;
        if (next_bit) {
            code |= 1;
            node_rank = rank1;
            node_bv_size = node_ones;
            ++left_siblings;
            left_total_bv += node_zeros;
        } else {
            node_rank = rank0;
            node_bv_size = node_zeros;
        }
        if (depth + 1 < tree_height) {
            int64_t next_leaf_count = rd16(sb->var + p8);
            p8 += 2;
            int64_t next_total_bv = rd16(sb->var + p8) + 1;
            p8 += 2;
            CNT_BYTES(4);
            left_total_bv -= (depth_total_bv - next_total_bv);
            bv_offset += depth_total_bv;
            depth_total_bv = next_total_bv;
            internal_nodes <<= 1;
            internal_nodes -= next_leaf_count;
            if (left_siblings >= next_leaf_count)
                left_siblings -= next_leaf_count;
            else
                break;
        } else {
            break;
        }
    }

    int64_t block_c = compute_symbol_from_block_header(sb->var, copy_p8, code, code_length);
    int c = rd16(sb->var + (int32_t)(p32 + 5 * block_c));
    CNT_BYTES(2);
    if (position == 0) return c;
    int64_t r_b = rd24(sb->var + (int32_t)(p32 + block_c * 5 + 2));
    int64_t r_sb = w->super_rank[sb_id * sigma_g + c];
    int64_t r_hb = w->hyper_rank[hb_id * sigma_g + c];
    CNT_BYTES(3 + 4 + 8);
    int64_t result = r_hb + r_sb + r_b + node_rank;
    return (int64_t)((uint64_t)result << 32) | c;
}

/* ------------------------------------------------------------------------------------------ */
/* fm/FmIndex.java                                                                            */
/* ------------------------------------------------------------------------------------------ */
struct OrcFmIndex {
    int sample_rate;       /* FM:93 */
    int enable_extract;    /* FM:101 */
    int n_keys;            /* monotonicMap, FM:97, in insertion order */
    int32_t *map_keys;
    int16_t *map_vals;
    int16_t *char2code;    /* 65536 entries; getOrDefault(.., 0) */
    int32_t *C;            /* cumulativeCounts, FM:103 */
    int n_c;
    int32_t *look_up;      /* monotonicLookUp, FM:105 */
    int n_look;
    IntVec *suffixes;      /* FM:108 */
    IntVec *positions;     /* FM:112 */
    int bw_suffixes, bw_positions; /* FM:117,119 */
    OrcRrr *sampled;       /* FM:123 */
    OrcWfbb *wt;           /* FM:129 */
    int length;            /* FM:131 */
};

/* suffix array of a sequence ending in a unique smallest symbol: plain prefix doubling
 * (sort by first symbol, then repeatedly sort every group of equal rank by the rank k ahead).
 * Stands in for jsuffixarrays DivSufSort (FM:332-341); the SA of such a string is unique. */
static const int32_t *g_sa_rank;
static int32_t g_sa_k, g_sa_n;
static int sa_cmp_ahead(const void *a_, const void *b_) {
    int32_t a = *(const int32_t *)a_, b = *(const int32_t *)b_;
    int32_t ra = a + g_sa_k < g_sa_n ? g_sa_rank[a + g_sa_k] : -1;
    int32_t rb = b + g_sa_k < g_sa_n ? g_sa_rank[b + g_sa_k] : -1;
    return ra < rb ? -1 : (ra > rb ? 1 : 0);
}
static int sa_cmp_first(const void *a_, const void *b_) {
    int32_t a = *(const int32_t *)a_, b = *(const int32_t *)b_;
    return g_sa_rank[a] < g_sa_rank[b] ? -1 : (g_sa_rank[a] > g_sa_rank[b] ? 1 : 0);
}
static int32_t *build_suffix_array(const int16_t *s, int32_t n) {
    int32_t *sa = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)n);
    int32_t *rank = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)n);
    int32_t *tmp = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)n);
    for (int32_t i = 0; i < n; i++) {
        sa[i] = i;
        rank[i] = s[i];
    }
    g_sa_rank = rank;
    g_sa_n = n;
    qsort(sa, (size_t)n, sizeof(int32_t), sa_cmp_first);
    /* rank := index of the first suffix of the group (groups = equal first symbol) */
    for (int32_t t = 0; t < n; t++) tmp[sa[t]] = (t > 0 && s[sa[t]] == s[sa[t - 1]]) ? tmp[sa[t - 1]] : t;
    memcpy(rank, tmp, sizeof(int32_t) * (size_t)n);
    for (int32_t k = 1; k < n; k <<= 1) {
        g_sa_k = k;
        int all_distinct = 1;
        int32_t i = 0;
        while (i < n) {
            int32_t j = i + 1;
            while (j < n && rank[sa[j]] == rank[sa[i]]) ++j;
            if (j - i > 1) {
                all_distinct = 0;
                qsort(sa + i, (size_t)(j - i), sizeof(int32_t), sa_cmp_ahead);
                /* split the group: new rank = index of the first suffix of each sub-group */
                int32_t head = i;
                tmp[sa[i]] = i;
                for (int32_t t = i + 1; t < j; t++) {
                    if (sa_cmp_ahead(&sa[t - 1], &sa[t]) != 0) head = t;
                    tmp[sa[t]] = head;
                }
            } else {
                tmp[sa[i]] = i;
            }
            i = j;
        }
        if (all_distinct) break;
        memcpy(rank, tmp, sizeof(int32_t) * (size_t)n);
    }
    free(rank);
    free(tmp);
    return sa;
}

/* FM:155-174 */
OrcFmIndex *orc_fm_build(const uint16_t *input, int32_t n_in, int sample_rate, int enable_extract, int *status) {
    if (status) *status = 0;
    OrcFmIndex *f = (OrcFmIndex *)xcalloc(1, sizeof *f);
    f->sample_rate = sample_rate;
    f->enable_extract = enable_extract;
    /* FM:300-305 */
    int32_t n = n_in + 1;
    uint16_t *text = (uint16_t *)xmalloc(sizeof(uint16_t) * (size_t)n);
    memcpy(text, input, sizeof(uint16_t) * (size_t)n_in);
    text[n_in] = 0;
    f->length = n;

    /* FM:396-435 mapToMonotonicSequence */
    uint8_t *seen = (uint8_t *)xcalloc(65536, 1);
    int distinct = 0, other_terminating = 0;
    for (int32_t i = 0; i < n; i++) {
        if (!seen[text[i]]) {
            seen[text[i]] = 1;
            ++distinct;
        }
        if (text[i] == 0) ++other_terminating;
    }
    int mapped_value = (other_terminating != 1) ? 1 : 0;
    f->n_look = distinct + 1;
    f->look_up = (int32_t *)xcalloc((size_t)f->n_look, sizeof(int32_t));
    f->map_keys = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(distinct + 1));
    f->map_vals = (int16_t *)xmalloc(sizeof(int16_t) * (size_t)(distinct + 1));
    f->char2code = (int16_t *)xcalloc(65536, sizeof(int16_t));
    int32_t *present = (int32_t *)xmalloc(sizeof(int32_t) * 65536);
    for (int i = 0; i < 65536; i++) present[i] = -1;
    f->n_keys = 0;
    present[0] = f->n_keys;
    f->map_keys[f->n_keys] = 0;
    f->map_vals[f->n_keys] = (int16_t)mapped_value;
    f->n_keys++;
    f->look_up[mapped_value] = 0;
    mapped_value++;
    for (int32_t i = 0; i < n; i++) {
        uint16_t symbol = text[i];
        if (present[symbol] < 0) { /* putIfAbsent returned null */
            present[symbol] = f->n_keys;
            f->map_keys[f->n_keys] = symbol;
            f->map_vals[f->n_keys] = (int16_t)mapped_value;
            f->n_keys++;
            if (mapped_value < f->n_look) f->look_up[mapped_value] = symbol;
            mapped_value++;
        }
    }
    free(seen);
    if (f->n_keys > 32767) { /* FM:423-426 */
        free(present);
        free(text);
        if (status) *status = -1;
        orc_fm_free(f);
        return NULL;
    }
    for (int i = 0; i < f->n_keys; i++) f->char2code[f->map_keys[i]] = f->map_vals[i];
    int16_t *mapped = (int16_t *)xmalloc(sizeof(int16_t) * (size_t)n);
    for (int32_t i = 0; i < n - 1; i++) mapped[i] = f->map_vals[present[text[i]]];
    mapped[n - 1] = 0;
    free(present);
    free(text);

    /* FM:307-327 fillCumulativeCounts */
    int32_t *cc = (int32_t *)xcalloc(65536, sizeof(int32_t));
    for (int32_t i = 0; i < n; i++) cc[mapped[i]]++;
    int32_t off = cc[0];
    cc[0] = 0;
    for (int i = 1; i < f->n_look; i++) {
        int32_t prev = cc[i];
        cc[i] = cc[i - 1] + off;
        off = prev;
    }
    f->n_c = f->n_look + 1;
    f->C = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)f->n_c);
    memcpy(f->C, cc, sizeof(int32_t) * (size_t)f->n_look);
    f->C[f->n_look] = f->length;
    free(cc);

    /* FM:329-372 buildSuffixArrayAndSample */
    int32_t *sa = build_suffix_array(mapped, n);
    f->bw_suffixes = minimum_number_of_bits(n);
    f->suffixes = iv_new(n / sample_rate + 1, f->bw_suffixes);
    BitBuf which = bb_new(n);
    int sampling_index = 0;
    for (int32_t i = 0; i < n; i++) {
        if (sa[i] % sample_rate == 0) {
            iv_set(f->suffixes, sampling_index, sa[i]);
            bb_set(&which, i, 1);
            sampling_index++;
        }
    }
    f->sampled = rrr_from_bitbuf(&which, sample_rate);
    free(which.w);
    if (enable_extract) {
        f->bw_positions = f->bw_suffixes;
        f->positions = iv_new(n / sample_rate + 2, f->bw_positions);
        for (int32_t i = 0; i < n; i++)
            if (sa[i] % sample_rate == 0) iv_set(f->positions, sa[i] / sample_rate, i);
        iv_set(f->positions, (n - 1) / sample_rate + 1, iv_get(f->positions, 0, f->bw_positions));
    }
    /* FM:374-394 burrowsWheelerTransform */
    int16_t *bwt = (int16_t *)xmalloc(sizeof(int16_t) * (size_t)n);
    for (int32_t i = 0; i < n; i++) bwt[i] = (sa[i] == 0) ? mapped[n - 1] : mapped[sa[i] - 1];
    free(sa);
    free(mapped);
    f->wt = orc_wfbb_build(bwt, n, sample_rate); /* FM:173 */
    free(bwt);
    return f;
}

void orc_fm_free(OrcFmIndex *f) {
    if (!f) return;
    free(f->map_keys);
    free(f->map_vals);
    free(f->char2code);
    free(f->C);
    free(f->look_up);
    iv_free(f->suffixes);
    iv_free(f->positions);
    orc_rrr_free(f->sampled);
    orc_wfbb_free(f->wt);
    free(f);
}
int orc_fm_input_length(const OrcFmIndex *f) { return f->length; }
int orc_fm_alphabet_length(const OrcFmIndex *f) { return f->n_keys; }
int orc_fm_sample_rate(const OrcFmIndex *f) { return f->sample_rate; }
const OrcWfbb *orc_fm_wavelet(const OrcFmIndex *f) { return f->wt; }

static inline int16_t fm_map(const OrcFmIndex *f, uint16_t ch) { return f->char2code[ch]; } /* getOrDefault(c, 0) */

/* one LF-step: C[c] + rank_c(BWT, i) */
static inline int32_t fm_lf(const OrcFmIndex *f, int32_t i, int16_t c, int *status) {
    g_cnt.lf_steps++;
    return (int32_t)(f->C[c] + orc_wfbb_rank(f->wt, i, c, status));
}

/* FM:455-474 */
int orc_fm_count(const OrcFmIndex *f, const uint16_t *pattern, int offset, int length, int *status) {
    if (status) *status = 0;
    int i = (offset + length) - 1;
    if (i < 0) { /* pattern[-1] */
        if (status) *status = ORC_E_JAVA_AIOOBE;
        return 0;
    }
    int16_t c = fm_map(f, pattern[i]);
    if (c == 0) return 0;
    int32_t start = f->C[c];
    int32_t end = f->C[c + 1];
    while (start < end && i >= offset + 1) {
        c = fm_map(f, pattern[--i]);
        if (c == 0) return 0;
        start = fm_lf(f, start, c, status);
        end = fm_lf(f, end, c, status);
    }
    int32_t d = end - start;
    return d > 0 ? d : 0;
}

/* FM:504-552 */
int orc_fm_locate(const OrcFmIndex *f, const uint16_t *pattern, int offset, int length, int32_t *locations,
                  int locations_len, int max_matches, int *status) {
    if (status) *status = 0;
    int i = (offset + length) - 1;
    if (i < 0) {
        if (status) *status = ORC_E_JAVA_AIOOBE;
        return 0;
    }
    int16_t c = fm_map(f, pattern[i]);
    if (c == 0) return 0;
    int32_t start = f->C[c];
    int32_t end = f->C[c + 1];
    int matches = 0;
    while (start < end && i >= (offset + 1)) {
        c = fm_map(f, pattern[--i]);
        if (c == 0) return 0;
        start = fm_lf(f, start, c, status);
        end = fm_lf(f, end, c, status);
    }
    if (start < end) {
        i = start + 1;
        while (i <= end) {
            int32_t j = i;
            int32_t distance = 0;
            while (!orc_rrr_access(f->sampled, j - 1, status)) {
                int64_t tuple = orc_wfbb_inverse_select(f->wt, j - 1);
                c = (int16_t)tuple;
                j = fm_lf(f, j, c, status);
                ++distance;
            }
            if (matches >= locations_len) { /* locations[matchesPosition] out of bounds */
                if (status) *status = ORC_E_JAVA_AIOOBE;
                return matches;
            }
            locations[matches] =
                (int32_t)(iv_get(f->suffixes, orc_rrr_rank_ones(f->sampled, j) - 1, f->bw_suffixes) + distance);
            CNT_BYTES((f->bw_suffixes + 7) / 8);
            ++matches;
            if (matches == max_matches) break;
            ++i;
        }
    }
    return matches;
}

/* FM:564-608 */
int orc_fm_extract(const OrcFmIndex *f, int start, int stop, uint16_t *dest, int dest_len, int offset, int *status) {
    *status = 0;
    if (!f->enable_extract) {
        *status = ORC_E_NOT_ENABLED;
        return 0;
    }
    if (start < 0) {
        *status = ORC_E_POS_NEGATIVE;
        return 0;
    }
    if (stop >= f->length) {
        *status = ORC_E_STOP_TOO_LONG;
        return 0;
    }
    int s = f->sample_rate;
    int32_t sample_position = (int32_t)(iv_get(f->positions, (stop / s) + 1, f->bw_positions) + 1);
    int skip = s - stop % s;
    if ((stop / s) == f->positions->length - 2) skip = f->length - stop;
    int range = stop - start;
    if (dest_len - offset < range) {
        *status = ORC_E_DEST_TOO_SMALL;
        return 0;
    }
    int remaining = range;
    int distance = 0;
    while (remaining > 0) {
        int16_t c = (int16_t)orc_wfbb_inverse_select(f->wt, sample_position - 1);
        sample_position = fm_lf(f, sample_position, c, status);
        if (distance >= skip) {
            int idx = remaining - 1 + offset;
            if (idx < 0 || idx >= dest_len) {
                *status = ORC_E_JAVA_AIOOBE;
                return 0;
            }
            dest[idx] = (uint16_t)f->look_up[c];
            remaining--;
        }
        distance++;
    }
    return range;
}

/* FM:610-626 */
static int check_bounds_for_extraction(const OrcFmIndex *f, int from, int dest_len) {
    if (!f->enable_extract) return ORC_E_NOT_ENABLED;
    if (from < 0) return ORC_E_POS_NEGATIVE;
    if (from >= f->length) return ORC_E_POS_TOO_LONG;
    if (dest_len == 0) return ORC_E_DEST_SIZE_ZERO;
    return 0;
}

#define DEST_AT(idx_expr, val)                   \
    do {                                         \
        int idx__ = (idx_expr);                  \
        if (idx__ < 0 || idx__ >= dest_len) {    \
            *status = ORC_E_JAVA_AIOOBE;         \
            return 0;                            \
        }                                        \
        dest[idx__] = (uint16_t)(val);           \
    } while (0)

/* FM:640-759 (mode 0), FM:772-831 (mode 1), FM:844-922 (mode 2) */
int orc_fm_extract_until_boundary(const OrcFmIndex *f, int mode, int from, uint16_t *dest, int dest_len, int offset,
                                  uint16_t boundary, int *status, int *aux) {
    *status = 0;
    if (aux) *aux = 0;
    int s = f->sample_rate;
    int32_t sample_position;
    int skip;
    int down_len = 0;

    if (mode == 1) ++from; /* FM:774 */
    int e = check_bounds_for_extraction(f, from, dest_len);
    if (e) {
        *status = e;
        return 0;
    }
    int16_t mapped_boundary;

    if (mode == 0 || mode == 1) {
        sample_position = (int32_t)(iv_get(f->positions, (from / s) + 1, f->bw_positions) + 1);
        skip = s - from % s;
        if ((from / s) == f->positions->length - 2) skip = f->length - from;
        int down_pos = dest_len - 1;
        mapped_boundary = fm_map(f, boundary);
        if (mapped_boundary == 0) {
            *status = ORC_E_NO_BOUNDARY;
            return 0;
        }
        int remaining = dest_len;
        int distance = 0;
        if (mode == 0) {
            while (remaining > 0) { /* FM:665-686 */
                int16_t c = (int16_t)orc_wfbb_inverse_select(f->wt, sample_position - 1);
                sample_position = fm_lf(f, sample_position, c, status);
                if (distance >= skip) {
                    if (c == mapped_boundary) break;
                    if (c == 0) break;
                    DEST_AT(down_pos, f->look_up[c]);
                    down_pos--;
                    remaining--;
                }
                distance++;
            }
        } else {
            while (1) { /* FM:797-824 */
                int16_t c = (int16_t)orc_wfbb_inverse_select(f->wt, sample_position - 1);
                sample_position = fm_lf(f, sample_position, c, status);
                if (distance >= skip) {
                    if (c == mapped_boundary) break;
                    if (c == 0) break;
                    DEST_AT(down_pos, f->look_up[c]);
                    down_pos--;
                    if (down_pos == offset) {
                        *status = ORC_E_DOES_NOT_FIT;
                        if (aux) *aux = dest_len - offset;
                        return 0;
                    }
                }
                distance++;
            }
        }
        down_len = dest_len - (down_pos + 1);
        /* System.arraycopy(destination, downStreamPos + 1, destination, offset, downStreamLength) */
        if (down_len > 0) {
            if (offset < 0 || offset + down_len > dest_len) {
                *status = ORC_E_JAVA_AIOOBE;
                return 0;
            }
            memmove(dest + offset, dest + down_pos + 1, sizeof(uint16_t) * (size_t)down_len);
        } else if (offset < 0 || offset > dest_len) {
            /* arraycopy with length 0 still range-checks dstPos in [0, length] */
            *status = ORC_E_JAVA_AIOOBE;
            return 0;
        }
        if (mode == 1) return down_len;
    } else {
        mapped_boundary = fm_map(f, boundary);
        if (mapped_boundary == 0) {
            *status = ORC_E_NO_BOUNDARY;
            return 0;
        }
    }

    /* incremental (+4) searches: FM:692-758 (mode 0) / FM:853-921 (mode 2) */
    int step = 4;
    int up_pos;
    int final_pos = -1;
    int times_up = 1;
    while (final_pos == -1) {
        int prev_from = from;
        from += step;
        if (from > f->length - 1) from = f->length - 1;
        int remaining = from - prev_from;
        up_pos = (times_up - 1) * step + remaining - 1;
        sample_position = (int32_t)(iv_get(f->positions, (from / s) + 1, f->bw_positions) + 1);
        skip = s - from % s;
        if ((from / s) == f->positions->length - 2) skip = f->length - from;
        int distance = 0;
        while (remaining > 0) {
            int16_t c = (int16_t)orc_wfbb_inverse_select(f->wt, sample_position - 1);
            sample_position = fm_lf(f, sample_position, c, status);
            if (distance >= skip) {
                if (c == mapped_boundary) {
                    if (up_pos == 0) return 0; /* first char was a boundary */
                    final_pos = up_pos;
                }
                if (mode == 0) {
                    if (offset + down_len + up_pos >= dest_len) {
                        *status = ORC_E_DOES_NOT_FIT;
                        if (aux) *aux = offset + down_len + up_pos;
                        return 0;
                    }
                    DEST_AT(offset + down_len + up_pos, f->look_up[c]);
                    up_pos--;
                } else {
                    if (offset + up_pos >= dest_len) {
                        *status = ORC_E_DOES_NOT_FIT;
                        if (aux) *aux = offset + up_pos;
                        return 0;
                    }
                    if (up_pos > 0) { /* range is (from, boundary] */
                        DEST_AT(offset + up_pos - 1, f->look_up[c]);
                        up_pos--;
                    }
                }
                remaining--;
            }
            distance++;
        }
        if (from == f->length - 1) {
            if (mode == 0)
                final_pos = (up_pos < 0) ? 1 : up_pos + from - prev_from;
            else
                final_pos = up_pos + from - prev_from;
            break;
        }
        ++times_up;
    }
    return mode == 0 ? down_len + final_pos : final_pos - 1;
}

void orc_fm_count_batch(const OrcFmIndex *f, const uint16_t *pat, const int32_t *pat_off, int32_t n, int32_t *counts,
                        int32_t *status, int threads) {
    (void)threads;
#ifdef _OPENMP
    if (threads > 1) {
        uint64_t lf = 0, ab = 0, lv = 0, q1 = 0, q2 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
#pragma omp parallel num_threads(threads) reduction(+ : lf, ab, lv, q1, q2, c1, c2, c3, c4, c5)
        {
            memset(&g_cnt, 0, sizeof g_cnt);
#pragma omp for schedule(static)
            for (int32_t i = 0; i < n; i++) {
                int st = 0;
                counts[i] = orc_fm_count(f, pat + pat_off[i], 0, pat_off[i + 1] - pat_off[i], &st);
                if (status) status[i] = st;
            }
            lf += g_cnt.lf_steps;
            ab += g_cnt.alg_bytes;
            lv += g_cnt.wt_levels;
            q1 += g_cnt.quirk_runblock_right;
            q2 += g_cnt.quirk_clamped_right;
            c1 += g_cnt.rank_calls;
            c2 += g_cnt.absent_superblock;
            c3 += g_cnt.absent_block;
            c4 += g_cnt.absent_scan_steps;
            c5 += g_cnt.run_block;
            memset(&g_cnt, 0, sizeof g_cnt);
        }
        g_cnt_total.lf_steps += lf;
        g_cnt_total.alg_bytes += ab;
        g_cnt_total.wt_levels += lv;
        g_cnt_total.quirk_runblock_right += q1;
        g_cnt_total.quirk_clamped_right += q2;
        g_cnt_total.rank_calls += c1;
        g_cnt_total.absent_superblock += c2;
        g_cnt_total.absent_block += c3;
        g_cnt_total.absent_scan_steps += c4;
        g_cnt_total.run_block += c5;
        return;
    }
#endif
    for (int32_t i = 0; i < n; i++) {
        int st = 0;
        counts[i] = orc_fm_count(f, pat + pat_off[i], 0, pat_off[i + 1] - pat_off[i], &st);
        if (status) status[i] = st;
    }
}

/* fold this thread's counters into the batch total (threads of the batch helpers below) */
static void cnt_fold_thread(void) {
#pragma omp critical(orc_cnt_fold)
    {
        g_cnt_total.lf_steps += g_cnt.lf_steps;
        g_cnt_total.alg_bytes += g_cnt.alg_bytes;
        g_cnt_total.wt_levels += g_cnt.wt_levels;
        g_cnt_total.quirk_runblock_right += g_cnt.quirk_runblock_right;
        g_cnt_total.quirk_clamped_right += g_cnt.quirk_clamped_right;
        g_cnt_total.rank_calls += g_cnt.rank_calls;
        g_cnt_total.absent_superblock += g_cnt.absent_superblock;
        g_cnt_total.absent_block += g_cnt.absent_block;
        g_cnt_total.absent_scan_steps += g_cnt.absent_scan_steps;
        g_cnt_total.run_block += g_cnt.run_block;
    }
    memset(&g_cnt, 0, sizeof g_cnt);
}

/* FM:504-552 looped: row i of locs (loc_cap ints) is query i's `locations` array */
void orc_fm_locate_batch(const OrcFmIndex *f, const uint16_t *pat, const int32_t *pat_off, int32_t n, int max_matches,
                         int32_t *locs, int loc_cap, int32_t *found, int32_t *status, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(dynamic, 256)
        for (int32_t i = 0; i < n; i++) {
            int st = 0;
            found[i] = orc_fm_locate(f, pat + pat_off[i], 0, pat_off[i + 1] - pat_off[i], locs + (size_t)i * (size_t)loc_cap,
                                     loc_cap, max_matches, &st);
            if (status) status[i] = st;
        }
        cnt_fold_thread();
    }
}

/* FM:640-922 looped: row i of dst (dst_len chars) is query i's `destination` array (in/out) */
void orc_fm_extract_until_boundary_batch(const OrcFmIndex *f, int mode, const int32_t *from, int32_t n, uint16_t boundary,
                                         uint16_t *dst, int dst_len, int offset, int32_t *out_len, int32_t *status,
                                         int32_t *aux, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(dynamic, 256)
        for (int32_t i = 0; i < n; i++) {
            int st = 0, ax = 0;
            const int r = orc_fm_extract_until_boundary(f, mode, from[i], dst + (size_t)i * (size_t)dst_len, dst_len, offset,
                                                        boundary, &st, &ax);
            out_len[i] = st ? 0 : r;
            if (status) status[i] = st;
            if (aux) aux[i] = ax;
        }
        cnt_fold_thread();
    }
}

/* FM:564-608 looped */
void orc_fm_extract_batch(const OrcFmIndex *f, const int32_t *start, const int32_t *stop, int32_t n, uint16_t *dst,
                          int dst_len, int offset, int32_t *out_len, int32_t *status, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
#pragma omp for schedule(dynamic, 256)
        for (int32_t i = 0; i < n; i++) {
            int st = 0;
            const int r = orc_fm_extract(f, start[i], stop[i], dst + (size_t)i * (size_t)dst_len, dst_len, offset, &st);
            out_len[i] = st ? 0 : r;
            if (status) status[i] = st;
        }
        cnt_fold_thread();
    }
}

/* FM:239-298 */
int orc_convert_byte_pattern(const uint8_t *pattern, int offset, int length, uint16_t *dest, int *bad_value) {
    int pos = offset, i = 0;
    while (pos < length + offset) {
        int8_t first = (int8_t)pattern[pos];
        uint16_t next;
        if (first < 0) {
            if ((((uint32_t)(first & 0xF0)) >> 3) == 30) { /* 4-byte form */
                int8_t b2 = (int8_t)pattern[pos + 1], b3 = (int8_t)pattern[pos + 2], b4 = (int8_t)pattern[pos + 3];
                pos += 4;
                int before = (((first & 0x07) << 18) | ((b2 & 0x3F) << 12) | ((b3 & 0x3F) << 6) | (b4 & 0x3F)) & 0x1FFFFF;
                if (before > 32767) {
                    if (bad_value) *bad_value = before;
                    return -1;
                }
                next = (uint16_t)before;
            } else if ((((uint32_t)(first & 0xE0)) >> 4) == 14) { /* 3-byte form */
                int8_t b2 = (int8_t)pattern[pos + 1], b3 = (int8_t)pattern[pos + 2];
                pos += 3;
                next = (uint16_t)((((first & 0x0F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F)) & 0xFFFF);
            } else { /* 2-byte form */
                int8_t b2 = (int8_t)pattern[pos + 1];
                pos += 2;
                next = (uint16_t)((((first & 0x1F) << 6) | (b2 & 0x3F)) & 0x7FF);
            }
        } else {
            ++pos;
            next = (uint16_t)first;
        }
        dest[i++] = next;
    }
    return i;
}

/* ------------------------------------------------------------------------------------------ */
/* serialization                                                                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint8_t *p;
    size_t n, cap;
} OBuf;
static void ob_reserve(OBuf *b, size_t extra) {
    if (b->n + extra > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 4096;
        while (nc < b->n + extra) nc *= 2;
        b->p = (uint8_t *)realloc(b->p, nc);
        if (!b->p) abort();
        b->cap = nc;
    }
}
static void ob_u8(OBuf *b, int v) {
    ob_reserve(b, 1);
    b->p[b->n++] = (uint8_t)v;
}
static void ob_i16(OBuf *b, int v) { /* DataOutput.writeShort: big-endian */
    ob_u8(b, (v >> 8) & 0xff);
    ob_u8(b, v & 0xff);
}
static void ob_i32(OBuf *b, int32_t v) {
    uint32_t u = (uint32_t)v;
    ob_u8(b, (u >> 24) & 0xff);
    ob_u8(b, (u >> 16) & 0xff);
    ob_u8(b, (u >> 8) & 0xff);
    ob_u8(b, u & 0xff);
}
static void ob_i64(OBuf *b, int64_t v) {
    uint64_t u = (uint64_t)v;
    for (int s = 56; s >= 0; s -= 8) ob_u8(b, (int)((u >> s) & 0xff));
}
static void iv_write(const IntVec *v, OBuf *b) { /* IV:196-203 */
    ob_u8(b, 0);
    ob_i32(b, v->length);
    ob_i32(b, v->width);
    for (int i = 0; i < v->nwords; i++) ob_i64(b, (int64_t)v->data[i]);
}
static void vv_write(const VarVec *v, OBuf *b) { /* VIV:175-181 */
    ob_u8(b, 0);
    ob_i32(b, v->nwords);
    for (int i = 0; i < v->nwords; i++) ob_i64(b, (int64_t)v->data[i]);
}
static void rrr_write(const OrcRrr *r, OBuf *b) { /* RRR:430-440 */
    ob_u8(b, 0);
    ob_i32(b, r->sample_size);
    ob_i32(b, r->length);
    ob_i32(b, r->total_ones);
    ob_i32(b, r->bits_per_offset_pos);
    iv_write(r->classes, b);
    vv_write(r->offsets, b);
    iv_write(r->sampled_offsets, b);
    iv_write(r->prefix_sums, b);
}
static void wfbb_write(const OrcWfbb *w, OBuf *b) { /* WFBB:1544-1570, 1651-1667, 1607-1613 */
    ob_u8(b, 0);
    ob_i64(b, w->size);
    ob_i32(b, w->alphabet_size);
    ob_i32(b, w->sampling_rate);
    ob_i32(b, w->n_count);
    for (int i = 0; i < w->n_count; i++) ob_i64(b, w->count[i]);
    ob_i32(b, w->n_hyper);
    for (int i = 0; i < w->n_hyper; i++) ob_i64(b, w->hyper_rank[i]);
    ob_i32(b, w->n_super_rank);
    for (int i = 0; i < w->n_super_rank; i++) ob_i32(b, w->super_rank[i]);
    ob_i32(b, w->n_global_mapping);
    for (int i = 0; i < w->n_global_mapping; i++) ob_i16(b, w->global_mapping[i]);
    ob_i32(b, w->n_sb);
    for (int s = 0; s < w->n_sb; s++) {
        const SuperBlock *sb = &w->sb[s];
        ob_i16(b, sb->sigma);
        ob_i16(b, sb->block_size_log);
        rrr_write(sb->rank_support, b);
        ob_i32(b, sb->n_blocks);
        for (int i = 0; i < sb->n_blocks; i++) {
            ob_i32(b, sb->block_headers[i].bv_rank);
            ob_i32(b, sb->block_headers[i].bv_offset);
            ob_i32(b, sb->block_headers[i].var_off);
            ob_i16(b, sb->block_headers[i].sigma);
            ob_i16(b, sb->block_headers[i].tree_height);
        }
        ob_i32(b, sb->var_len);
        for (int i = 0; i < sb->var_len; i++) ob_u8(b, sb->var[i]);
        ob_i32(b, sb->mapping_len);
        for (int i = 0; i < sb->mapping_len; i++) ob_i16(b, sb->mapping[i]);
    }
}

/* java.util.HashMap<Integer,Short>.keySet() iteration order for FM:956-960 (JDK 8+ behaviour, stated from knowledge —
 * unverifiable here), by replaying the puts in insertion order: 16 slots at first; doubling when the size passes 0.75 x
 * capacity, and when a put makes a bucket 9 nodes long below 64 slots (treeifyBin resizes instead); slot = (h ^ (h >>> 16)) &
 * (capacity - 1); a resize keeps the relative order inside each half of a split bucket.  Tree bins (a 9-node bucket at >= 64
 * slots) are not modelled: the plain-bucket order is written. */
static void hashmap_order(const OrcFmIndex *f, int *order) {
    int n = f->n_keys;
    uint32_t cap = 16;
    /* slot of every key under the current capacity, recomputed per resize; bucket sizes for the 9-node rule */
    int *size_of = (int *)xcalloc(16, sizeof(int));
    for (int i = 0; i < n; i++) {
        uint32_t h = (uint32_t)f->map_keys[i];
        uint32_t slot = (h ^ (h >> 16)) & (cap - 1);
        int before = size_of[slot]++;
        int grow = 0;
        if (before >= 8 && cap < 64) grow = 1;
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) grow = (uint32_t)(i + 1) > cap / 4 * 3;
            if (!grow) continue;
            cap *= 2;
            free(size_of);
            size_of = (int *)xcalloc(cap, sizeof(int));
            for (int j = 0; j <= i; j++) {
                uint32_t hj = (uint32_t)f->map_keys[j];
                size_of[(hj ^ (hj >> 16)) & (cap - 1)]++;
            }
        }
    }
    free(size_of);
    /* stable bucket sort under the final capacity (= insertion order inside a bucket) */
    int *cnt = (int *)xcalloc((size_t)cap + 1, sizeof(int));
    for (int i = 0; i < n; i++) {
        uint32_t h = (uint32_t)f->map_keys[i];
        cnt[((h ^ (h >> 16)) & (cap - 1)) + 1]++;
    }
    for (uint32_t i = 0; i < cap; i++) cnt[i + 1] += cnt[i];
    for (int i = 0; i < n; i++) {
        uint32_t h = (uint32_t)f->map_keys[i];
        order[cnt[(h ^ (h >> 16)) & (cap - 1)]++] = i;
    }
    free(cnt);
}

/* SER:67-79: ObjectOutputStream = magic AC ED 00 05 + block-data records of <= 1024 bytes
 * (0x77 len8 for len <= 255, 0x7A len32 otherwise). */
static void frame_stream(const OBuf *raw, OBuf *out) {
    ob_u8(out, 0xAC);
    ob_u8(out, 0xED);
    ob_u8(out, 0x00);
    ob_u8(out, 0x05);
    size_t pos = 0;
    while (pos < raw->n) {
        size_t len = raw->n - pos;
        if (len > 1024) len = 1024;
        if (len <= 255) {
            ob_u8(out, 0x77);
            ob_u8(out, (int)len);
        } else {
            ob_u8(out, 0x7A);
            ob_i32(out, (int32_t)len);
        }
        ob_reserve(out, len);
        memcpy(out->p + out->n, raw->p + pos, len);
        out->n += len;
        pos += len;
    }
}

/* FM:948-975 */
int orc_fm_write(const OrcFmIndex *f, int framed, uint8_t **buf, size_t *len) {
    OBuf b = {0, 0, 0};
    ob_u8(&b, 0);
    ob_i32(&b, f->sample_rate);
    ob_u8(&b, f->enable_extract ? 1 : 0);
    ob_i32(&b, f->bw_suffixes);
    ob_i32(&b, f->bw_positions);
    ob_i32(&b, f->length);
    ob_i32(&b, f->n_keys);
    int *order = (int *)xmalloc(sizeof(int) * (size_t)f->n_keys);
    hashmap_order(f, order);
    for (int i = 0; i < f->n_keys; i++) {
        ob_i32(&b, f->map_keys[order[i]]);
        ob_i16(&b, f->map_vals[order[i]]);
    }
    free(order);
    ob_i32(&b, f->n_c);
    for (int i = 0; i < f->n_c; i++) ob_i32(&b, f->C[i]);
    ob_i32(&b, f->n_look);
    for (int i = 0; i < f->n_look; i++) ob_i32(&b, f->look_up[i]);
    iv_write(f->suffixes, &b);
    if (f->enable_extract) iv_write(f->positions, &b);
    rrr_write(f->sampled, &b);
    wfbb_write(f->wt, &b);
    if (framed) {
        OBuf o = {0, 0, 0};
        frame_stream(&b, &o);
        free(b.p);
        b = o;
    }
    *buf = b.p;
    *len = b.n;
    return 0;
}
void orc_free_buffer(uint8_t *buf) { free(buf); }

typedef struct {
    const uint8_t *p;
    size_t n, pos;
    int err;
} IBuf;
static int ib_u8(IBuf *b) {
    if (b->pos + 1 > b->n) {
        b->err = 1;
        return 0;
    }
    return b->p[b->pos++];
}
static int ib_i16(IBuf *b) {
    int hi = ib_u8(b), lo = ib_u8(b);
    return (int16_t)((hi << 8) | lo);
}
static int32_t ib_i32(IBuf *b) {
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) v = (v << 8) | (uint32_t)ib_u8(b);
    return (int32_t)v;
}
static int64_t ib_i64(IBuf *b) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | (uint64_t)ib_u8(b);
    return (int64_t)v;
}
static void check_version(IBuf *b) { /* SER:46-56 */
    if (ib_u8(b) != 0) b->err = 2;
}
static IntVec *iv_read(IBuf *b) { /* IV:211-227 */
    check_version(b);
    int length = ib_i32(b), width = ib_i32(b);
    if (b->err || length < 0 || width < 0 || width > 64) {
        b->err = b->err ? b->err : 3;
        return NULL;
    }
    IntVec *v = iv_new(length, width);
    if ((size_t)v->nwords * 8 > b->n - b->pos) {
        b->err = 1;
        return v;
    }
    for (int i = 0; i < v->nwords; i++) v->data[i] = (uint64_t)ib_i64(b);
    return v;
}
static VarVec *vv_read(IBuf *b) { /* VIV:189-198 */
    check_version(b);
    int nwords = ib_i32(b);
    if (b->err || nwords < 0 || (size_t)nwords * 8 > b->n - b->pos) {
        b->err = b->err ? b->err : 1;
        return NULL;
    }
    VarVec *v = vv_new((int64_t)nwords * 64);
    for (int i = 0; i < nwords; i++) v->data[i] = (uint64_t)ib_i64(b);
    return v;
}
static OrcRrr *rrr_read(IBuf *b) { /* RRR:448-469 */
    rrr_tables_init();
    check_version(b);
    OrcRrr *r = (OrcRrr *)xcalloc(1, sizeof *r);
    r->sample_size = ib_i32(b);
    r->length = ib_i32(b);
    r->total_ones = ib_i32(b);
    r->bits_per_offset_pos = ib_i32(b);
    r->classes = iv_read(b);
    r->offsets = vv_read(b);
    r->sampled_offsets = iv_read(b);
    r->prefix_sums = iv_read(b);
    return r;
}
static OrcWfbb *wfbb_read(IBuf *b) { /* WFBB:286-322, 1630-1649, 1597-1605 */
    check_version(b);
    OrcWfbb *w = (OrcWfbb *)xcalloc(1, sizeof *w);
    w->size = ib_i64(b);
    w->alphabet_size = ib_i32(b);
    w->sampling_rate = ib_i32(b);
#define RD_ARRAY(field, cnt, type, rd)                                              \
    do {                                                                            \
        w->cnt = ib_i32(b);                                                         \
        if (b->err || w->cnt < 0 || (size_t)w->cnt > b->n - b->pos) {               \
            b->err = b->err ? b->err : 1;                                           \
            return w;                                                               \
        }                                                                           \
        w->field = (type *)xmalloc(sizeof(type) * (size_t)(w->cnt ? w->cnt : 1));   \
        for (int i_ = 0; i_ < w->cnt; i_++) w->field[i_] = (type)rd(b);             \
    } while (0)
    RD_ARRAY(count, n_count, int64_t, ib_i64);
    RD_ARRAY(hyper_rank, n_hyper, int64_t, ib_i64);
    RD_ARRAY(super_rank, n_super_rank, int32_t, ib_i32);
    RD_ARRAY(global_mapping, n_global_mapping, int16_t, ib_i16);
#undef RD_ARRAY
    w->n_sb = ib_i32(b);
    if (b->err || w->n_sb < 0 || (size_t)w->n_sb > b->n - b->pos) {
        b->err = b->err ? b->err : 1;
        w->n_sb = 0;
        return w;
    }
    w->sb = (SuperBlock *)xcalloc((size_t)w->n_sb, sizeof(SuperBlock));
    for (int s = 0; s < w->n_sb && !b->err; s++) {
        SuperBlock *sb = &w->sb[s];
        sb->sigma = (int16_t)ib_i16(b);
        sb->block_size_log = (int16_t)ib_i16(b);
        sb->rank_support = rrr_read(b);
        sb->n_blocks = ib_i32(b);
        if (b->err || sb->n_blocks < 0 || (size_t)sb->n_blocks * 16 > b->n - b->pos) {
            b->err = b->err ? b->err : 1;
            sb->n_blocks = 0;
            return w;
        }
        sb->block_headers = (BlockHdr *)xcalloc((size_t)sb->n_blocks, sizeof(BlockHdr));
        for (int i = 0; i < sb->n_blocks; i++) {
            sb->block_headers[i].bv_rank = ib_i32(b);
            sb->block_headers[i].bv_offset = ib_i32(b);
            sb->block_headers[i].var_off = ib_i32(b);
            sb->block_headers[i].sigma = (int16_t)ib_i16(b);
            sb->block_headers[i].tree_height = (int16_t)ib_i16(b);
        }
        sb->var_len = ib_i32(b);
        if (b->err || sb->var_len < 0 || (size_t)sb->var_len > b->n - b->pos) {
            b->err = b->err ? b->err : 1;
            sb->var_len = 0;
            return w;
        }
        sb->var = (uint8_t *)xcalloc((size_t)sb->var_len + 16, 1);
        for (int i = 0; i < sb->var_len; i++) sb->var[i] = (uint8_t)ib_u8(b);
        sb->mapping_len = ib_i32(b);
        if (b->err || sb->mapping_len < 0 || (size_t)sb->mapping_len * 2 > b->n - b->pos) {
            b->err = b->err ? b->err : 1;
            sb->mapping_len = 0;
            return w;
        }
        sb->mapping = (int16_t *)xmalloc(sizeof(int16_t) * (size_t)(sb->mapping_len ? sb->mapping_len : 1));
        for (int i = 0; i < sb->mapping_len; i++) sb->mapping[i] = (int16_t)ib_i16(b);
    }
    return w;
}

/* FM:983-1025 (+ SER:89-100 de-framing when the buffer starts with AC ED 00 05).
 * *status: 0 ok, 1 truncated, 2 "Incompatible serial versions!", 3 malformed. */
OrcFmIndex *orc_fm_read(const uint8_t *buf, size_t len, int *status) {
    uint8_t *plain = NULL;
    int corrupt_tail = 0;
    if (len >= 4 && buf[0] == 0xAC && buf[1] == 0xED && buf[2] == 0x00 && buf[3] == 0x05) {
        plain = (uint8_t *)xmalloc(len);
        /* java.io.ObjectInputStream.BlockDataInputStream.readBlockHeader, read lazily as FmIndex.read pulls bytes:
         * TC_BLOCKDATA 0x77 <u8>, TC_BLOCKDATALONG 0x7A <i32 >= 0>, TC_RESET 0x79 skipped between records; anything else
         * ends the block data.  What is well-formed is gathered; a reader that needs more gets "truncated" (1), or
         * "malformed" (3) when the payload ended at a corrupt header. */
        size_t pos = 4, out = 0;
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
                    corrupt_tail = 1;
                    break;
                }
                bl = ((size_t)buf[pos + 1] << 24) | ((size_t)buf[pos + 2] << 16) | ((size_t)buf[pos + 3] << 8) | buf[pos + 4];
                pos += 5;
            } else {
                corrupt_tail = buf[pos] < 0x70 || buf[pos] > 0x7E;
                break;
            }
            if (bl > len - pos) bl = len - pos;
            memcpy(plain + out, buf + pos, bl);
            out += bl;
            pos += bl;
        }
        buf = plain;
        len = out;
    }
    IBuf b = {buf, len, 0, 0};
    OrcFmIndex *f = (OrcFmIndex *)xcalloc(1, sizeof *f);
    check_version(&b);
    f->sample_rate = ib_i32(&b);
    f->enable_extract = ib_u8(&b) != 0;
    f->bw_suffixes = ib_i32(&b);
    f->bw_positions = ib_i32(&b);
    f->length = ib_i32(&b);
    f->n_keys = ib_i32(&b);
    if (b.err || f->n_keys < 0 || f->n_keys > 65536) goto bad;
    f->map_keys = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(f->n_keys + 1));
    f->map_vals = (int16_t *)xmalloc(sizeof(int16_t) * (size_t)(f->n_keys + 1));
    f->char2code = (int16_t *)xcalloc(65536, sizeof(int16_t));
    for (int i = 0; i < f->n_keys; i++) {
        f->map_keys[i] = ib_i32(&b);
        f->map_vals[i] = (int16_t)ib_i16(&b);
        if (f->map_keys[i] >= 0 && f->map_keys[i] < 65536) f->char2code[f->map_keys[i]] = f->map_vals[i];
    }
    f->n_c = ib_i32(&b);
    if (b.err || f->n_c < 0 || (size_t)f->n_c * 4 > b.n - b.pos) goto bad;
    f->C = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(f->n_c + 1));
    for (int i = 0; i < f->n_c; i++) f->C[i] = ib_i32(&b);
    f->n_look = ib_i32(&b);
    if (b.err || f->n_look < 0 || (size_t)f->n_look * 4 > b.n - b.pos) goto bad;
    f->look_up = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)(f->n_look + 1));
    for (int i = 0; i < f->n_look; i++) f->look_up[i] = ib_i32(&b);
    f->suffixes = iv_read(&b);
    if (f->enable_extract) f->positions = iv_read(&b);
    f->sampled = rrr_read(&b);
    f->wt = wfbb_read(&b);
    if (b.err) goto bad;
    free(plain);
    if (status) *status = 0;
    return f;
bad:
    if (status) *status = b.err ? ((b.err == 1 && corrupt_tail) ? 3 : b.err) : 3;
    free(plain);
    orc_fm_free(f);
    return NULL;
}
