/*
 * fmx.h — C ABI of libfmx.so, the MI355X-native engine for index4j's backward-search path.
 *
 * This is the drop-in boundary.  index4j (pure Java) has no FFI seam of its own (FmIndex is a
 * final class), so the boundary is: the public FmIndex / FmIndexBuilder method
 * signatures on the host side, the serialized byte layout (FmIndex.write) as the hand-over format,
 * and the functions below as what a JNI / Panama binding of those methods calls.  Each function
 * cites the reference interface it replaces; paths are relative to
 * /root/reference/indices/src/main/java/com/dynatrace/ (FM = fm/FmIndex.java, FMB =
 * fm/FmIndexBuilder.java, SER = serialization/Serialization.java).
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  Every function returns FMX_OK or a
 * negative library error (never throws).  Per-query failures (the reference's exceptions) are
 * reported in status[] with the FMX_ST_* codes so the host binding can re-throw the identical
 * exception type and message (fmx_status_message).  All batch calls run on the GPU; there is no CPU
 * query path in this library: without a HIP device they return FMX_E_NO_DEVICE.
 */
#ifndef FMX_H
#define FMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fmx_index fmx_index;

/* library-level return codes */
#define FMX_OK 0
#define FMX_E_ARG (-1)        /* bad argument */
#define FMX_E_ALPHABET (-2)   /* IllegalArgumentException("Input has more than 32767 different symbols") FM:423-426 */
#define FMX_E_FORMAT (-3)     /* malformed / truncated serialized stream */
#define FMX_E_VERSION (-4)    /* IOException("Incompatible serial versions! ...") SER:46-56 */
#define FMX_E_NO_DEVICE (-5)  /* no HIP device / index not resident on a device */
#define FMX_E_HIP (-6)        /* a HIP runtime call failed (see fmx_last_error) */
#define FMX_E_NOMEM (-7)
#define FMX_E_UNSUPPORTED (-8) /* e.g. text longer than one 2^32 hyperblock */

/* per-query status codes (exceptions of the reference; 0 = no exception) */
#define FMX_ST_OK 0
#define FMX_ST_NOT_ENABLED 1     /* RuntimeException "Text recovery not enabled at build time"   FM:566-568, 611-613 */
#define FMX_ST_POS_NEGATIVE 2    /* RuntimeException "Requested position less than 0"            FM:570-572, 615-617 */
#define FMX_ST_STOP_TOO_LONG 3   /* RuntimeException "Stop position longer than index string"    FM:574-576 */
#define FMX_ST_DEST_TOO_SMALL 4  /* RuntimeException "Supplied destination is not large enough"  FM:591-593 */
#define FMX_ST_POS_TOO_LONG 5    /* RuntimeException "Requested position longer than index string" FM:619-621 */
#define FMX_ST_DEST_SIZE_ZERO 6  /* IllegalArgumentException "Supplied destination for extraction has size zero" FM:623-625 */
#define FMX_ST_NO_BOUNDARY 7     /* IllegalArgumentException "Boundary does not exist"           FM:659-661, 792-794, 849-851 */
#define FMX_ST_DOES_NOT_FIT 8    /* RuntimeException "Extraction does not fit in the supplied destination. Currently extracted: N" (N in aux[]) FM:732-737, 816-821, 893-898 */
#define FMX_ST_JAVA_AIOOBE 9     /* the JVM would raise ArrayIndexOutOfBoundsException (e.g. empty pattern FM:456-457, locations[] too small FM:538) */

/* ---- construction, persistence, lifetime ------------------------------------------------- */

/* new FmIndexBuilder().setSampleRate(s).setEnableExtraction(b).build(char[])  FMB:34-62 -> FM:155-174.
 * Host-side construction (suffix array, BWT, wavelet/RRR encoding); text = UTF-16 code units. */
int fmx_build(const uint16_t *text, int32_t n, int32_t sample_rate, int enable_extract, fmx_index **out);
/* The same index with the middle of the constructor computed on GPU `device`: FM:329-394 (suffix array by prefix
 * doubling, sampled rows, inverse samples, BWT) and FM:173 — the wavelet tree over the BWT (WFBB:130-154, 362-535,
 * 570-991) and its RRR vectors (RRR:225-286), encoded in HBM where the BWT lies (alphabets of up to 1,024 codes;
 * larger ones, and option "wavelet_on_device" = 0, encode the tree on the host).
 * The suffix array of a terminated text is unique and the encoders make the same choices, so the result is
 * byte-identical to fmx_build's (fmx_save gives the same bytes).  Optional statistics: doubling rounds after the initial 4-character sort, and rows
 * that went through a device sort, wall seconds of the device stage incl. transfers.  The handle still needs
 * fmx_to_device before queries. */
int fmx_build_on_device(const uint16_t *text, int32_t n, int32_t sample_rate, int enable_extract, int device,
                        fmx_index **out, int32_t *rounds, int64_t *rows_sorted, double *stage_seconds);
/* seconds fmx_build_on_device spent encoding the wavelet tree in HBM; 0 = it was encoded on the host */
double fmx_build_wavelet_seconds(const fmx_index *idx);

/* The suffix table of a resident index: fmx_to_device / fmx_attach_device_blob grow, level by level and with the very rank code the
 * queries run, the strings of 2, 3, ... codes that OCCUR in the text, each with its SA interval (the state of FM:455-474 after
 * a pattern's last characters), and hash ALL levels into one open-addressing table of 16-byte slots {key, start, end} (groups
 * of 64 slots, in-group slot = low bits of the first character's code XOR hash bits of the rest; at most 0.5 full, 0.7 where
 * only that keeps it inside its limit).  Depth *chars: as deep as a 64-bit key holds (8 codes of 8 bits, 4 of 16; option
 * "suffix_table_chars" caps it) while the table stays below the smaller of the budget (option "suffix_table_mb", default 256; 0
 * = no table) and 1 / "suffix_table_image_fraction" of the image (default an eighth): 5 characters in 8.4 MB on the 256 MiB
 * synthetic log.  count / locate batches start from it: one slot instead of 2 * (k - 1) rank evaluations for a pattern's last k
 * <= *chars characters; a string that is not in it (it does not occur, holds an unknown character, or its search raised a
 * status) is searched by the loop.  Results, statuses and LF-step counts are unchanged (option "suffix_table" = 0 makes
 * launches ignore it, for A/B).  *chars = 0: no table; *bytes = the table's size. */
int fmx_suffix_table_info(const fmx_index *idx, int32_t *chars, int64_t *bytes);

/* The window directory of a resident index (grown by fmx_to_device / fmx_attach_device_blob beside the image, like the suffix
 * table; option "window_cells": 0 = none, 1 = always, in cells, 2 = the default rule: where it fits a quarter of the device's free
 * memory — and in its FLAT form where that costs at most 1/128 of the device's memory (option "window_flat_fraction"; 0 = never by
 * itself) and the alphabet has at most 2,048 symbols (the symbol search then runs in LDS), 3 = the flat form by name.  Flat = one 32-bit word per BWT position instead of cells and entries: every step of a walk is
 * ONE sector, at 4 bytes per text byte; locate 20-25 % faster, extract / extractUntilBoundary 10 %; texts below 2^30 characters.
 * Cells: one
 * 64-byte cell per 112 consecutive BWT positions holding, for the window's three most frequent symbols ("classes"), their folded
 * rank at the window start and — in two bit planes — the positions they stand at, plus every position's bit of sampledSuffixes
 * (FM:123) — and one entry per position that holds none of the three: everything the LF-step of that row hands back, whatever
 * route of the tree (and whichever of the reference's quirks) it takes.  An entry is 4 bytes for alphabets of up to 2,048 symbols
 * (the row the step arrives at; its symbol is the largest c with cumulativeCounts[c] < row, found by a search the kernels run in
 * LDS; the few answers that are more than a row — a status, a quirk — sit in 8-byte slots behind the entries) and 6 bytes beyond
 * {next row, symbol, status, suspect}; option "window_entry_bytes" = 4 / 6 forces a form (0, the default: by the alphabet).  An LF-step of
 * locate / extract / extractUntilBoundary — inverseSelect (WFBB:1305-1537) of a position, the poll of FM:531, the rank of FM:534 —
 * then costs ONE 64-byte sector, or two, and no walk through the wavelet tree, for EVERY row of the index (the tree's loop over
 * the levels of a code is what a 64-lane wave runs to the deepest code among its positions); count() does not use it.
 * 0.57 + 0.18 x 4 = 1.27 bytes per text byte on log text (1.65 with 6-byte entries).  Every number in it is the step the index's own rank() / inverseSelect()
 * took when the directory was grown: results, statuses and LF-step counts do not depend on having one.  Budget: option
 * "window_cells_mb" (default 65,536) caps it in absolute terms beside the quarter-of-free-memory rule of "window_cells" = 2;
 * fmx_resident_bytes says what an index took.  *bytes = its size (0: none). */
int fmx_window_cells_info(const fmx_index *idx, int64_t *bytes);
/* What a resident handle holds in its device's memory, in bytes (each pointer nullable): the image, the suffix table (with its
 * order-1 statistics), the window directory.  All 0 for a handle that is not resident.  (index4j's own figure for comparison is
 * the serialized size, FmIndexSerializedSizeBenchmark.java:57: 0.44-0.47 bytes per text byte.) */
int fmx_resident_bytes(const fmx_index *idx, int64_t *image, int64_t *suffix_table, int64_t *window_directory);

/* FmIndex.read(ObjectInput) FM:983-1025; also accepts the ObjectOutputStream-framed form produced by
 * Serialization.writeToByteArray SER:67-79 (magic AC ED 00 05 + block-data records). */
int fmx_load(const uint8_t *ser, size_t len, fmx_index **out);

/* FmIndex.write(ObjectOutput) FM:948-975; framed != 0 adds the SER:67-79 ObjectOutputStream framing.
 * *buf is owned by the library until fmx_free_buffer. */
int fmx_save(const fmx_index *idx, int framed, uint8_t **buf, size_t *len);
/* FM:956-960 writes the character map in java.util.HashMap's keySet() order, which fmx_save reproduces by replaying the map's
 * puts (capacity doubling, the resize a 9-node bucket forces below 64 slots, insertion order inside a bucket).  1 = that replay
 * covers this index; 0 = a JVM would have turned one of the buckets into a TREE bin (9 keys in one slot at 64 slots or more:
 * thousands of symbols whose codes collide modulo the table size), whose iteration order is not modelled — fmx_save's stream is
 * still one FmIndex.read accepts (its reader does not depend on the order, FM:992-998) but may differ from a JVM's bytes inside
 * that bucket; < 0 = error.  The Java shim reports it as GpuFmIndex.isSerializedFormVerified(). */
int fmx_save_key_order_modelled(const fmx_index *idx);
void fmx_free_buffer(uint8_t *buf);
void fmx_free(fmx_index *idx);

/* getInputLength FM:929 (includes the sentinel), getAlphabetLength FM:939, builder knobs FMB:21-22 */
int32_t fmx_input_length(const fmx_index *idx);
int32_t fmx_alphabet_length(const fmx_index *idx);
int32_t fmx_sample_rate(const fmx_index *idx);
int32_t fmx_extract_enabled(const fmx_index *idx);

/* ---- the flat HBM image ("blob") --------------------------------------------------------- */

/* Relocatable, pointer-free image of the whole index (layout: index4j_amd/csrc/fmx_blob.hpp).
 * It is what lives in HBM, and what is broadcast to the other GPUs over RCCL. */
int fmx_blob(const fmx_index *idx, const uint8_t **blob, size_t *len);
/* copy the blob into HBM of `device` (hipMalloc + hipMemcpy) and make idx queryable there */
int fmx_to_device(fmx_index *idx, int device);
/* adopt a blob that already sits in device memory (e.g. the receive buffer of an RCCL broadcast).
 * The memory stays owned by the caller and must outlive the index. */
int fmx_attach_device_blob(void *device_blob, size_t len, int device, fmx_index **out);
void *fmx_device_blob(const fmx_index *idx, size_t *len);

/* ---- batched queries: host buffers (H2D copy, kernels, D2H copy, synchronous) ------------- */

/* Optional: pin a long-lived host buffer of the caller (a direct ByteBuffer of the Java shim, a reused array) so that the
 * host-buffer entry points move it by DMA without staging copies; buffers that are not registered work all the same.
 * fmx_count_batch with ALL of its arrays registered copies nothing: one launch reads the patterns from the mapped arrays and
 * stores counts / LF-steps / statuses into them (option "host_mapped" = 0: the chunk pipeline instead; "host_direct_stores" = 0:
 * that pipeline with result copies).  Register WHOLE arrays: a kernel is only handed an array that lies inside ONE range
 * registered through THIS function (the library keeps the table; an array pinned some other way, or spanning two
 * registrations, takes the staged copies), and the HIP runtime refuses to copy a range registered only in part (FMX_E_HIP).
 * A registered array must stay registered, and must not be written by the caller, while a call that was given it is in
 * flight: the kernels read and write its pages directly, and the offsets are validated on the host before the launch.
 * (hipHostRegister / hipHostUnregister; pages stay locked until unregistered.) */
int fmx_host_register(void *p, size_t bytes);
int fmx_host_unregister(void *p);

/* int count(char[] pattern, int offset, int length) FM:455-474, batched: pattern i is
 * pat[pat_off[i] .. pat_off[i+1]).  lf_steps[i] (nullable) = number of C[c]+rank evaluations spent
 * (FM:469-470). status[i] (nullable) is FMX_ST_JAVA_AIOOBE for an empty pattern.
 * Batches of >= 131,072 patterns (option "host_pipeline_min") travel in chunks of 262,144 patterns, a chunk's transfer
 * overlapping the kernels of the one before; pat_off must start at >= 0 and never decrease (FMX_E_ARG otherwise).
 * Small calls — up to 2,048 patterns (option "host_small_max"; the same holds for fmx_locate_batch, fmx_extract_batch,
 * fmx_extract_boundary_batch, fmx_locate_extract_batch and fmx_locate_lines_batch): everything the call moves goes through one pinned block the kernels read and write where it lies
 * (no copy calls): a batch of ONE — a Java caller's count(char[]) — costs ~25 us (locate ~50, extract of 64 characters ~70). */
int fmx_count_batch(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n,
                    int32_t *counts, int32_t *lf_steps, int32_t *status);

/* int locate(char[] pattern, int offset, int length, int[] locations, int maxMatches) FM:504-552.
 * locs is n rows of loc_cap ints (the caller's `locations` arrays); found[i] = return value =
 * number located (<= max_matches; max_matches = -1: unlimited, FM:488).  Hits are SA rows
 * start+1.. in order (FM:527-547), exactly the ones the reference returns. */
int fmx_locate_batch(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n,
                     int32_t max_matches, int32_t *locs, int32_t loc_cap, int32_t *found, int32_t *lf_steps,
                     int32_t *status);

/* int extract(int start, int stop, char[] destination, int offset) FM:564-608.  dst is n rows of
 * dst_len chars (row i = the `destination` array of query i, in/out); out_len[i] = return value. */
int fmx_extract_batch(const fmx_index *idx, const int32_t *start, const int32_t *stop, int32_t n, uint16_t *dst,
                      int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf_steps, int32_t *status);

/* extractUntilBoundary FM:640-759 (mode 0), extractUntilBoundaryLeft FM:772-831 (mode 1),
 * extractUntilBoundaryRight FM:844-922 (mode 2).  aux[i] = N of "Currently extracted: N". */
int fmx_extract_boundary_batch(const fmx_index *idx, const int32_t *from, int32_t n, uint16_t boundary, int mode,
                               uint16_t *dst, int32_t dst_len, int32_t offset, int32_t *out_len,
                               int32_t *lf_steps, int32_t *status, int32_t *aux);

/* ---- batched queries: device-resident buffers, asynchronous on `stream` (a hipStream_t) ----
 * Same semantics; every pointer is device memory on the index's device.  Nothing is synchronised:
 * the caller orders work through the stream (this is what bench.py times with HIP events).
 * Threading: the index is immutable once resident (the reference's FmIndex is @ThreadSafe, FM:82).  The host-buffer
 * entry points above may be called from any number of threads on one index at once (per-call device scratch).
 * The device-pointer entry points keep grow-only scratch per (index, stream): use one stream per thread. */
int fmx_count_batch_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                        int32_t *d_counts, int32_t *d_lf_steps, int32_t *d_status, void *stream);
/* The two stages of fmx_count_batch_dev, callable separately (bench.py times the second one alone):
 * plan = processing order of the batch (device bucket pass on the patterns' trailing characters, so that
 * neighbouring lanes walk the same SA intervals) + the mapped codes of each pattern's last characters; *d_plan is
 * an opaque handle into per-stream scratch owned by the index, or NULL for small batches.  It stays valid until
 * ANYTHING else plans on that stream (every count / locate / segment / pipeline call does) and only for the same
 * d_pat / d_pat_off contents.  ordered = the k_count kernel over that order; a handle that is no longer the
 * stream's live plan — or that was made for other d_pat / d_pat_off BUFFERS or another n — is ignored (the batch is
 * then processed in the caller's order: same results).  The library compares buffer addresses, not contents: a
 * caller that refills d_pat or d_pat_off in place must plan again.  Results are written at the ORIGINAL pattern index. */
int fmx_count_plan_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                       const void **d_plan, void *stream);
int fmx_count_ordered_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, const void *d_plan,
                          int32_t n, int32_t *d_counts, int32_t *d_lf_steps, int32_t *d_status, void *stream);
/* 1 if fmx_count_batch_dev / fmx_locate_batch_dev would run the plan stage for a batch of n patterns on this resident index, 0 if
 * they count in the caller's order — results are the same either way.  On an index with a suffix table the plan orders the
 * batch by the (estimated) first SA row of each pattern's tabulated suffix (option "plan_sa_key": 2 = estimated from the table's
 * two-character strings, the default; 1 = the table's own answer; 0 = by the trailing characters' codes, as on an index
 * without a table) and pays from "plan_sa_min" patterns on (option, default 786,432); with "plan_sa_key" 0 a batch is planned
 * if it holds at least "plan_min_per_string" (default 16) patterns per string of the table's deepest level.  An index without a
 * table plans every batch of "sort_min" patterns or more.  (fmx_count_plan_dev always plans: the caller asked.)
 * locate has an order of its own on top: batches of "walk_order_min" patterns or more (default 32,768; 0 = never) walk their
 * hits by the first row of the patterns' SA ranges ("walk_fine" = 0 drops that order's fine pass); extractUntilBoundary batches
 * of "boundary_order_min" queries or more (default 32,768) are taken by text position.  No order changes a result. */
int fmx_count_batch_is_planned(const fmx_index *idx, int32_t n);
/* The same question for every batch policy: kind 0 = count() planned (as above), 1 = locate() walks its hits by the first row
 * of the patterns' SA ranges ("walk_order_min"), 2 = extractUntilBoundary takes its queries by text position
 * ("boundary_order_min").  1 / 0; results never depend on the answer. */
int fmx_batch_policy(const fmx_index *idx, int kind, int64_t n);
int fmx_locate_batch_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                         int32_t max_matches, int32_t *d_locs, int32_t loc_cap, int32_t *d_found,
                         int32_t *d_lf_steps, int32_t *d_status, int32_t *d_range_ws /* 2*n ints */, void *stream);
int fmx_extract_batch_dev(const fmx_index *idx, const int32_t *d_start, const int32_t *d_stop, int32_t n,
                          uint16_t *d_dst, int32_t dst_len, int32_t offset, int32_t *d_out_len,
                          int32_t *d_lf_steps, int32_t *d_status, void *stream);
int fmx_extract_boundary_batch_dev(const fmx_index *idx, const int32_t *d_from, int32_t n, uint16_t boundary,
                                   int mode, uint16_t *d_dst, int32_t dst_len, int32_t offset, int32_t *d_out_len,
                                   int32_t *d_lf_steps, int32_t *d_status, int32_t *d_aux, void *stream);

/* ---- locate -> extract pipelines ---------------------------------------------------------------
 * The composite the reference times in locateAndExtractBenchmark (indices/src/jmh/java/com/dynatrace/fm/
 * FmIndexThroughputBenchmark.java:231-249): matches = locate(pattern, 0, length, locations, maxMatches), then
 * for each i < matches extract(locations[i], min(getInputLength(), locations[i] + maxExtractionLength),
 * destination, 0).  Here both stages run on the device and the hit positions never leave HBM.
 * Hit k of pattern i uses slot i*max_matches + k of locs / out_len / hit_status / hit_aux and destination
 * row (i*max_matches + k) of `row length` chars, written from offset 0.  Only slots k < found[i] are
 * written; all others keep the caller's values.  max_matches must be >= 1 (it is the row count per pattern,
 * and the locate limit FM:504-552); n*max_matches must fit an int32.
 *   status[i]     = status of locate(pattern i)              (AIOOBE cannot occur: loc_cap == max_matches)
 *   hit_status[s] = status of the extract call of slot s (e.g. FMX_ST_STOP_TOO_LONG when the hit lies within
 *                   extract_len of the end of the text: the benchmark's stop == getInputLength(), FM:572-574)
 * fmx_locate_lines_* replaces extract by extractUntilBoundary (mode 0) / ...Left (1) / ...Right (2),
 * FM:640-922 — "the lines that contain the pattern"; hit_aux as in fmx_extract_boundary_batch. */
int fmx_locate_extract_batch(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n,
                             int32_t max_matches, int32_t extract_len, int32_t *locs, int32_t *found, uint16_t *dst,
                             int32_t *out_len, int32_t *lf_steps, int32_t *status, int32_t *hit_status);
int fmx_locate_lines_batch(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n,
                           int32_t max_matches, uint16_t boundary, int mode, int32_t dst_len, int32_t *locs,
                           int32_t *found, uint16_t *dst, int32_t *out_len, int32_t *lf_steps, int32_t *status,
                           int32_t *hit_status, int32_t *hit_aux);
/* device-pointer forms: enqueue on `stream`; d_range_ws = 2*n ints; d_hit_status / d_hit_aux may be NULL */
int fmx_locate_extract_batch_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                                 int32_t max_matches, int32_t extract_len, int32_t *d_locs, int32_t *d_found,
                                 uint16_t *d_dst, int32_t *d_out_len, int32_t *d_lf_steps, int32_t *d_status,
                                 int32_t *d_hit_status, int32_t *d_range_ws, void *stream);
int fmx_locate_lines_batch_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                               int32_t max_matches, uint16_t boundary, int mode, int32_t dst_len, int32_t *d_locs,
                               int32_t *d_found, uint16_t *d_dst, int32_t *d_out_len, int32_t *d_lf_steps,
                               int32_t *d_status, int32_t *d_hit_status, int32_t *d_hit_aux, int32_t *d_range_ws,
                               void *stream);

/* ---- segment sets: texts beyond one FmIndex ----------------------------------------------------
 * FmIndex addresses its text with Java ints (`length` FM:131, RrrVector positions RRR:358), so a text of
 * >= 2^31 chars (BASELINE configs[4]: 2 GiB) is K independent FmIndex objects over consecutive pieces of the
 * text, cut at record boundaries; a caller sums count() over them and adds each piece's start to its
 * locate() results.  These entry points do that on the device for K handles resident on the same GPU:
 *   counts[i] = sum over segments of count(pattern i)            (int64: the sum can pass 2^31)
 *   locs      = n rows of max_matches int64 text positions: segment 0's hits (SA order, FM:526-548),
 *               then segment 1's, ... each moved by seg_base[s] (host array), truncated at max_matches;
 *               found[i] = number written (max_matches >= 1)
 *   status[i] = first non-zero per-segment status (an empty pattern fails the same way in every segment)
 * Occurrences that span a cut are not occurrences in any segment, exactly as with K Java objects.
 * Slots of a row at and beyond found[i] keep the caller's values — except in the row of a pattern whose status is non-zero, which is
 * unspecified from found[i] on (a segment's hits are stored straight into the set's rows before its status is known: option
 * "segments_direct").
 * Device forms: d_tmp = 3*n ints (count) / 4*n + n*max_matches ints (locate); d_lf_steps / d_status may be NULL. */
int fmx_count_segments(const fmx_index *const *segs, int32_t n_segs, const uint16_t *pat, const int32_t *pat_off,
                       int32_t n, int64_t *counts, int64_t *lf_steps, int32_t *status);
int fmx_locate_segments(const fmx_index *const *segs, int32_t n_segs, const int64_t *seg_base, const uint16_t *pat,
                        const int32_t *pat_off, int32_t n, int32_t max_matches, int64_t *locs, int32_t *found,
                        int32_t *status);
int fmx_count_segments_dev(const fmx_index *const *segs, int32_t n_segs, const uint16_t *d_pat, const int32_t *d_pat_off,
                           int32_t n, int64_t *d_counts, int64_t *d_lf_steps, int32_t *d_status, int32_t *d_tmp,
                           void *stream);
int fmx_locate_segments_dev(const fmx_index *const *segs, int32_t n_segs, const int64_t *seg_base, const uint16_t *d_pat,
                            const int32_t *d_pat_off, int32_t n, int32_t max_matches, int64_t *d_locs, int32_t *d_found,
                            int32_t *d_status, int32_t *d_tmp, void *stream);
/* count() AND locate() of one batch over a segment set in one pass: the range search of a segment (FM:455-474 = FM:506-523)
 * yields both the count and the SA range its hits are located from, so this costs one search per segment where
 * fmx_count_segments_dev + fmx_locate_segments_dev cost two.  Same outputs as the two calls (d_counts / d_lf_steps: int64 sums
 * over the segments, d_lf_steps may be NULL: the LF-steps of the searches, not of the walks); d_tmp = 4*n + n*max_matches ints. */
int fmx_count_locate_segments_dev(const fmx_index *const *segs, int32_t n_segs, const int64_t *seg_base, const uint16_t *d_pat,
                                  const int32_t *d_pat_off, int32_t n, int32_t max_matches, int64_t *d_counts, int64_t *d_lf_steps,
                                  int64_t *d_locs, int32_t *d_found, int32_t *d_status, int32_t *d_tmp, void *stream);

/* count() AND locate() over a segment set with host buffers (the host form of fmx_count_locate_segments_dev): counts / lf_steps
 * int64 sums over the segments (lf_steps nullable), locs n rows of max_matches int64 (in / out), found, status (nullable). */
int fmx_count_locate_segments(const fmx_index *const *segs, int32_t n_segs, const int64_t *seg_base, const uint16_t *pat,
                              const int32_t *pat_off, int32_t n, int32_t max_matches, int64_t *counts, int64_t *lf_steps,
                              int64_t *locs, int32_t *found, int32_t *status);

/* ---- replicas: one immutable index on several GPUs, one host process ---------------------------------
 * FmIndex is immutable and @ThreadSafe (FM:82; the reference's throughput benchmark gives every thread an index of its own,
 * indices/src/jmh/java/com/dynatrace/fm/FmIndexThroughputState.java:30), and every query of a batch is an independent read: so
 * the image is REPLICATED on the GPUs of a node and a batch is cut into contiguous shards, one per replica — no exchange on the
 * query path, the "gather" is that every shard stores into its own slice of the caller's arrays (SURVEY 8e scheme (i)).
 *
 * fmx_replicate: out[i] = a new handle, resident on devices[i] (i < n_devices; a device may be named more than once — two
 * replicas then share it).  The image goes from where `idx` has it — HBM of its device: one peer copy per destination over
 * xGMI, all destinations at once (root egress over all links, no ring); the host otherwise — and every replica grows its own
 * suffix table and window directory on its device, in parallel.  `idx` itself is unchanged (it may but need not be resident);
 * a replica answers every query and accessor, keeps no host model (fmx_save / fmx_blob: FMX_E_ARG) and is freed with fmx_free.
 * On failure nothing is left behind.  fmx_device_of: the device ordinal a handle is resident on, -1 = not resident. */
int fmx_replicate(const fmx_index *idx, const int32_t *devices, int32_t n_devices, fmx_index **out);
int fmx_device_of(const fmx_index *idx);
/* The shard arithmetic of every *_multi call: items [*lo, *hi) of n belong to part `part` of `parts` (contiguous; sizes differ
 * by at most one, the first n % parts parts hold the extra item). */
void fmx_shard_range(int64_t n, int32_t parts, int32_t part, int64_t *lo, int64_t *hi);

/* The host-buffer batch calls over a replica set: arguments as in the single-index forms; shard r — fmx_shard_range(n,
 * n_replicas, r) — runs on replicas[r] from a host thread of its own (the library keeps one worker per replica slot of a device;
 * the calling thread takes shard 0), all shards at once, each storing into rows / entries [lo, hi) of the caller's arrays.
 * Results are those of the single-index call on the whole batch, entry by entry.  Returns the first failing shard's error
 * (fmx_last_error: its message); the other shards still complete.  Replicas of DIFFERENT indexes are the caller's mistake. */
int fmx_count_batch_multi(const fmx_index *const *replicas, int32_t n_replicas, const uint16_t *pat, const int32_t *pat_off,
                          int32_t n, int32_t *counts, int32_t *lf_steps, int32_t *status);
int fmx_locate_batch_multi(const fmx_index *const *replicas, int32_t n_replicas, const uint16_t *pat, const int32_t *pat_off,
                           int32_t n, int32_t max_matches, int32_t *locs, int32_t loc_cap, int32_t *found, int32_t *lf_steps,
                           int32_t *status);
int fmx_extract_batch_multi(const fmx_index *const *replicas, int32_t n_replicas, const int32_t *start, const int32_t *stop,
                            int32_t n, uint16_t *dst, int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf_steps,
                            int32_t *status);
int fmx_extract_boundary_batch_multi(const fmx_index *const *replicas, int32_t n_replicas, const int32_t *from, int32_t n,
                                     uint16_t boundary, int mode, uint16_t *dst, int32_t dst_len, int32_t offset,
                                     int32_t *out_len, int32_t *lf_steps, int32_t *status, int32_t *aux);
/* BASELINE configs[4]: segs = n_replicas x n_segs handles, replica-major (segs[r * n_segs + s] = segment s on replica r's
 * device: fmx_replicate of every segment index onto the same device list); fmx_count_locate_segments per shard. */
int fmx_count_locate_segments_multi(const fmx_index *const *segs, int32_t n_replicas, int32_t n_segs, const int64_t *seg_base,
                                    const uint16_t *pat, const int32_t *pat_off, int32_t n, int32_t max_matches, int64_t *counts,
                                    int64_t *lf_steps, int64_t *locs, int32_t *found, int32_t *status);
/* Device-resident shards, asynchronous: entry r of every array is replica r's operand of fmx_count_batch_dev /
 * fmx_count_locate_segments_dev, resident on that replica's device (n[r] patterns; streams[r] a hipStream_t of that device or
 * NULL).  The launches of all replicas are issued at once, each from its device's worker thread; the call returns when they are
 * ENQUEUED.  fmx_multi_synchronize waits for streams[r] on every replica's device.  (What bench.py --single-process times.) */
int fmx_count_batch_multi_dev(const fmx_index *const *replicas, int32_t n_replicas, const uint16_t *const *d_pat,
                              const int32_t *const *d_pat_off, const int32_t *n, int32_t *const *d_counts,
                              int32_t *const *d_lf_steps, int32_t *const *d_status, void *const *streams);
int fmx_count_locate_segments_multi_dev(const fmx_index *const *segs, int32_t n_replicas, int32_t n_segs, const int64_t *seg_base,
                                        const uint16_t *const *d_pat, const int32_t *const *d_pat_off, const int32_t *n,
                                        int32_t max_matches, int64_t *const *d_counts, int64_t *const *d_lf_steps,
                                        int64_t *const *d_locs, int32_t *const *d_found, int32_t *const *d_status,
                                        int32_t *const *d_tmp, void *const *streams);
int fmx_multi_synchronize(const fmx_index *const *replicas, int32_t n_replicas, void *const *streams);

/* ---- WaveletFixedBlockBoosting as a stand-alone structure (the reference's public class, WFBB:130-154) ----
 * `sequence` = symbols already mapped to small non-negative integers (short[] text of WFBB:130).  The handle
 * answers only the two calls below (after fmx_to_device); free it with fmx_free. */
int fmx_wavelet_build(const int16_t *sequence, int64_t n, int32_t sampling_rate, fmx_index **out);
/* long rank(long position, short symbol) WFBB:1010-1285, batched */
int fmx_wavelet_rank_batch(const fmx_index *idx, const int64_t *positions, const int32_t *symbols, int32_t n,
                           int64_t *ranks, int32_t *status);
/* long inverseSelect(long position) WFBB:1305-1537, batched: packed[i] = (rank << 32) | symbol, and the bare
 * symbol for position 0, exactly as the reference returns it */
int fmx_wavelet_inverse_select_batch(const fmx_index *idx, const int64_t *positions, int32_t n, int64_t *packed,
                                     int32_t *status);

/* ---- RrrVector as a stand-alone structure (the reference's public class, RRR:225-286) -------------------
 * new RrrVector(BitVector, sampleSize): `bits` = one byte per bit.  The handle keeps the vector in its compressed
 * form (15-bit blocks, class + offset) and answers only the two calls below, after fmx_to_device; free it
 * with fmx_free.  (Inside an FM-index image the bit vectors are expanded instead, see fmx_blob.hpp.) */
int fmx_rrr_build(const uint8_t *bits, int64_t n, int32_t sample_size, fmx_index **out);
/* int rankOnes(int position) RRR:358-396, batched (0 below 0, totalOnes from length on) */
int fmx_rrr_rank_ones_batch(const fmx_index *idx, const int32_t *positions, int32_t n, int32_t *ranks);
/* boolean access(int position) RRR:314-349, batched; status = FMX_ST_JAVA_AIOOBE where the reference throws */
int fmx_rrr_access_batch(const fmx_index *idx, const int32_t *positions, int32_t n, uint8_t *bits, int32_t *status);
/* the same two calls with operands resident in HBM, asynchronous on `stream` (RrrVectorThroughputBenchmark.java:43-51 is
 * measured through these: positions in, ranks out, nothing crosses PCIe inside the timed region) */
int fmx_rrr_rank_ones_batch_dev(const fmx_index *idx, const int32_t *d_positions, int32_t n, int32_t *d_ranks, void *stream);
int fmx_rrr_access_batch_dev(const fmx_index *idx, const int32_t *d_positions, int32_t n, uint8_t *d_bits, int32_t *d_status,
                             void *stream);

/* ---- helpers ------------------------------------------------------------------------------ */

/* FmIndex.convertBytePatternToCharPattern FM:239-298.  Returns the number of chars, or -1 with
 * *bad_value set when the reference throws "Found a character that exceeds (32767): it was N". */
int fmx_convert_byte_pattern(const uint8_t *pattern, int32_t offset, int32_t length, uint16_t *dest,
                             int32_t *bad_value);
/* the reference's exception message for a status code ("%d" left in place for FMX_ST_DOES_NOT_FIT) */
const char *fmx_status_message(int status);
/* 1 if the exception type is IllegalArgumentException, 0 for RuntimeException, 2 for AIOOBE */
int fmx_status_kind(int status);
const char *fmx_last_error(void);
/* The host-buffer entry points recycle their device scratch between calls (up to 2 GiB per process); this
 * returns it to the driver. */
void fmx_release_scratch(void);
int fmx_device_count(void);
/* launch tunables: "block" = threads per workgroup (512 | 1024), "groups_per_cu" = grid cap per CU,
 * "sort_min" = smallest batch that is processed in suffix-sorted order (0 = never), "sort_bits" = sort key width,
 * "boundary_accel" = 0 forces the literal +4-chunk right walk of extractUntilBoundary, "boundary_group" = lanes
 * per extractUntilBoundary query (0 | 2 | 4 | 8 | 16), "coarse_bits" / "plan_fine" = bins and fine pass of the plan
 * stage, "plan_sa_key" / "plan_sa_min" / "plan_min_per_string" / "walk_order_min" / "walk_fine" / "boundary_order_min" = which batches are planned
 * and by what (fmx_count_batch_is_planned), "suffix_table" = 0: launches ignore the index's suffix table, "lf_steps_executed_only" = 1: the LF-step
 * output of count() leaves out the rank evaluations the suffix table answered (bench.py's executed-work figure; the
 * default reports the reference's count), "boundary_first_fill" = 0: a lane of extractUntilBoundary walks its two
 * sample intervals one after the other instead of interleaved (A/B; 2 = default).
 * Applied when an index is flattened or becomes resident afterwards: "suffix_table_mb" / "suffix_table_chars" (budget
 * and depth of the suffix table, 0 = none), "sb_cache_limit" (superblocks whose headers are staged in LDS), "map_by_symbol" / "map_fast" / "inv_fast"
 * (layout of the image: tests force the reference's own routes with them), "window_cells" (the window directory of
 * fmx_window_cells_info: 0 none, 1 always, 2 where it fits a quarter of the device's free memory).  Applied by
 * fmx_build_on_device: "wavelet_on_device" = 0 encodes the wavelet tree on the host.
 * Results are identical for every setting. */
/* Image form (fmx_set_option("image_compact", 0 | 1), applies to images flattened afterwards: fmx_to_device / fmx_blob of an index
 * that has none yet).  0 (default): the bit vectors of the wavelet tree and the sampled-row bitmap are EXPANDED into 16-byte cells
 * {ones before, 96 bits} — a rank is one load + three popcounts; 0.64 bytes per text byte resident on the 256 MiB log.
 * 1 (compact): they stay in the reference's own compression (15-bit blocks as class + offset, RRR:225-286) as 16-block records
 * + offsets stream, decoded by the kernels through the value-of-offset table in LDS — smaller (bytes per text byte and timings:
 * DESIGN.md 3), a rank costs a second dependent load.  Results are identical; an image says which form it is, and travels as
 * before (fmx_blob / fmx_attach_device_blob). */
int fmx_set_option(const char *name, int value);

/* deterministic synthetic workload (bench / tests): see index4j_amd/csrc/fmx_synth.cpp */
int fmx_synth_log(uint64_t seed, int32_t n, uint16_t *out);
/* the same log lines with runs of multi-byte characters dropped in at word boundaries (~6 % of the characters), `symbols`
 * distinct characters in all: the shape of the reference's fixture HDFS_2k_multichar.log and of the data set its
 * published numbers are quoted on (> 1,000 distinct symbols, README.md:291-292) */
int fmx_synth_log_multichar(uint64_t seed, int32_t n, int32_t symbols, uint16_t *out);
int fmx_synth_patterns(uint64_t seed, const uint16_t *text, int32_t n, int32_t m, int32_t count, uint16_t *pat,
                       int32_t *pat_off, int32_t *positions);

#ifdef __cplusplus
}
#endif
#endif /* FMX_H */
