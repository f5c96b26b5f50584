/*
 * index4j_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the index4j backward-search path (FmIndex count / locate / extract /
 * extractUntilBoundary{,Left,Right} -> WaveletFixedBlockBoosting rank / inverseSelect ->
 * RrrVector rankOnes / access), of the constructors that produce those structures, and of the
 * serialized layout.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference/indices/src/main/java/com/dynatrace/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The shipped product (index4j_amd/) never links, imports or calls it.
 *
 * Parity pinning: the reference is pure Java and cannot be built or run in this image (no JDK).
 * The oracle is pinned by (1) the reference's own known-answer tests restated as literals in
 * tests/test_oracle_kat.py (RrrVectorTest, WaveletFixedBlockBoostingTest, FmIndexTest), (2) the
 * reference's definitional test oracles (overlapping regex count, sorted locations, substring,
 * boundary scanners — util/Util.java) re-implemented independently in Python over the reference's
 * own fixture HDFS_2k_multichar.log, (3) SHA-256 digests of the reference's three literal RRR
 * tables (tests/golden/rrr_tables.json).  Serialized BYTES are "parity unpinned": the reference
 * holds no golden serialized file and no JVM exists here to mint one.
 */
#ifndef INDEX4J_ORACLE_H
#define INDEX4J_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcFmIndex OrcFmIndex;
typedef struct OrcWfbb OrcWfbb;
typedef struct OrcRrr OrcRrr;

/* status codes: one per exception the reference can throw on this path (FM:566-576, 591-593,
 * 610-626, 659-661, 732-737, 816-821, 893-898) plus Java's implicit ArrayIndexOutOfBounds. */
enum {
    ORC_OK = 0,
    ORC_E_NOT_ENABLED = 1,     /* RuntimeException("Text recovery not enabled at build time") */
    ORC_E_POS_NEGATIVE = 2,    /* RuntimeException("Requested position less than 0") */
    ORC_E_STOP_TOO_LONG = 3,   /* RuntimeException("Stop position longer than index string") */
    ORC_E_DEST_TOO_SMALL = 4,  /* RuntimeException("Supplied destination is not large enough") */
    ORC_E_POS_TOO_LONG = 5,    /* RuntimeException("Requested position longer than index string") */
    ORC_E_DEST_SIZE_ZERO = 6,  /* IllegalArgumentException("Supplied destination for extraction has size zero") */
    ORC_E_NO_BOUNDARY = 7,     /* IllegalArgumentException("Boundary does not exist") */
    ORC_E_DOES_NOT_FIT = 8,    /* RuntimeException("Extraction does not fit ... Currently extracted: N") ; N in *aux */
    ORC_E_JAVA_AIOOBE = 9      /* ArrayIndexOutOfBoundsException raised implicitly by the JVM */
};

/* counters filled by the query functions (counting mode) */
typedef struct {
    uint64_t lf_steps;   /* evaluations of C[c] + rank_c(BWT, i)  (FM:469,470,535,599,669,718,801,879) */
    uint64_t alg_bytes;  /* logical field bytes the reference algorithm reads for those steps */
    uint64_t wt_levels;  /* wavelet-tree levels traversed (RRR rankOnes calls inside WFBB) */
    uint64_t quirk_runblock_right;   /* WFBB:1081 executed with treeHeight==0 (offset -4 read) */
    uint64_t quirk_clamped_right;    /* WFBB:1071-1104 with a clamped mapping entry (no fix-up) */
    uint64_t rank_calls;             /* WFBB.rank calls that reach the tables */
    uint64_t absent_superblock;      /* WFBB:1040-1042 exits */
    uint64_t absent_block;           /* WFBB:1048 taken */
    uint64_t absent_scan_steps;      /* mapping entries read by the scan of WFBB:1053-1059 */
    uint64_t run_block;              /* WFBB:1141-1146 exits */
} OrcCounters;

void orc_counters_reset(void);
void orc_counters_get(OrcCounters *out);

/* ---- RRR (bitsequence/RrrVector.java) ---- */
OrcRrr *orc_rrr_from_bits(const uint8_t *bits /* one byte per bit */, int64_t n, int sample);  /* RRR:225-286 */
OrcRrr *orc_rrr_from_ints(const int32_t *ints, int n_ints, int sample);                         /* RRR:143-211 */
int orc_rrr_access(const OrcRrr *r, int position, int *status);   /* RRR:314-349 */
int orc_rrr_rank_ones(const OrcRrr *r, int position);             /* RRR:358-396 */
int orc_rrr_rank_zeroes(const OrcRrr *r, int position);           /* RRR:405-410 */
void orc_rrr_rank_ones_batch(const OrcRrr *r, const int32_t *positions, int32_t n, int32_t *out, int threads);
int orc_rrr_estimated_memory(const OrcRrr *r);                    /* RRR:418-423 */
void orc_rrr_free(OrcRrr *r);
/* the three static tables (generated, RRR:104-129, 488-16900) as flat u16 arrays */
const uint16_t *orc_rrr_table_offset_of_value(void);   /* 32768 entries */
const uint16_t *orc_rrr_table_value_of_offset(void);   /* 32768 entries */
const uint16_t *orc_rrr_table_cardinality_offsets(void); /* 16 entries */
const int *orc_rrr_table_bits_needed(void);            /* 16 entries */

/* ---- WaveletFixedBlockBoosting (wavelet/WaveletFixedBlockBoosting.java) ---- */
OrcWfbb *orc_wfbb_build(const int16_t *text, int64_t n, int sampling_rate);   /* WFBB:130-154 */
int64_t orc_wfbb_rank(const OrcWfbb *w, int64_t position, int16_t symbol, int *status);  /* WFBB:1010-1285 */
int64_t orc_wfbb_inverse_select(const OrcWfbb *w, int64_t position);          /* WFBB:1305-1537 */
int orc_wfbb_block_size_log(const OrcWfbb *w, int superblock);
void orc_wfbb_free(OrcWfbb *w);

/* ---- FmIndex (fm/FmIndex.java) ---- */
/* FM:155-174. Returns NULL and sets *status=-1 for "Input has more than 32767 different symbols" (FM:423-426). */
OrcFmIndex *orc_fm_build(const uint16_t *text, int32_t n, int sample_rate, int enable_extract, int *status);
void orc_fm_free(OrcFmIndex *f);
int orc_fm_input_length(const OrcFmIndex *f);      /* FM:929 */
int orc_fm_alphabet_length(const OrcFmIndex *f);   /* FM:939 */
int orc_fm_sample_rate(const OrcFmIndex *f);
const OrcWfbb *orc_fm_wavelet(const OrcFmIndex *f);

int orc_fm_count(const OrcFmIndex *f, const uint16_t *pattern, int offset, int length, int *status);  /* FM:455-474 */
int orc_fm_locate(const OrcFmIndex *f, const uint16_t *pattern, int offset, int length,
                  int32_t *locations, int locations_len, int max_matches, int *status);               /* FM:504-552 */
int orc_fm_extract(const OrcFmIndex *f, int start, int stop, uint16_t *dest, int dest_len, int offset,
                   int *status);                                                                      /* FM:564-608 */
/* mode 0 = extractUntilBoundary (FM:640-759), 1 = ...Left (FM:772-831), 2 = ...Right (FM:844-922) */
int orc_fm_extract_until_boundary(const OrcFmIndex *f, int mode, int from, uint16_t *dest, int dest_len,
                                  int offset, uint16_t boundary, int *status, int *aux);

/* batch helpers used by the cpu_baseline leg and by parity tests (same semantics, looped) */
void orc_fm_count_batch(const OrcFmIndex *f, const uint16_t *pat, const int32_t *pat_off, int32_t n,
                        int32_t *counts, int32_t *status, int threads);

void orc_fm_locate_batch(const OrcFmIndex *f, const uint16_t *pat, const int32_t *pat_off, int32_t n, int max_matches,
                         int32_t *locs, int loc_cap, int32_t *found, int32_t *status, int threads);
void orc_fm_extract_batch(const OrcFmIndex *f, const int32_t *start, const int32_t *stop, int32_t n, uint16_t *dst,
                          int dst_len, int offset, int32_t *out_len, int32_t *status, int threads);
void orc_fm_extract_until_boundary_batch(const OrcFmIndex *f, int mode, const int32_t *from, int32_t n,
                                         uint16_t boundary, uint16_t *dst, int dst_len, int offset, int32_t *out_len,
                                         int32_t *status, int32_t *aux, int threads);

/* serialization (FM:948-1025, IV:196-227, VIV:175-198, RRR:430-469, WFBB:1544-1570, 1597-1667, SER:67-79) */
int orc_fm_write(const OrcFmIndex *f, int framed, uint8_t **buf, size_t *len);
OrcFmIndex *orc_fm_read(const uint8_t *buf, size_t len, int *status);
void orc_free_buffer(uint8_t *buf);

/* FM:239-298; returns number of chars, or -1 with *bad_value set for the ">32767" exception */
int orc_convert_byte_pattern(const uint8_t *pattern, int offset, int length, uint16_t *dest, int *bad_value);

#ifdef __cplusplus
}
#endif
#endif
