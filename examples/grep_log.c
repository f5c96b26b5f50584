/* examples/grep_log.c — the C ABI of include/fmx.h end to end, in plain C (what a JNI / Panama / cgo binding calls):
 * index a text file on the GPU, count patterns, then fetch the lines that contain them.
 *
 *   cc -std=c99 -Iinclude examples/grep_log.c -Lindex4j_amd -lfmx -Wl,-rpath,$PWD/index4j_amd -Wl,-rpath,/opt/rocm/lib -o grep_log
 *   ./grep_log tests/golden/HDFS_2k_multichar.log WARN "blk_-1608999687919862906"
 *
 * The text is treated as ISO-8859-1 (one char per byte) to keep the example short; index4j itself indexes UTF-16
 * code units, and FmIndex.convertBytePatternToCharPattern (fmx_convert_byte_pattern) maps UTF-8 patterns. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fmx.h"

static void die(const char *what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, fmx_last_error());
    exit(1);
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <text file> <pattern>...\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 1;
    }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *bytes = malloc((size_t)size + 1);
    if (fread(bytes, 1, (size_t)size, f) != (size_t)size) return 1;
    fclose(f);
    uint16_t *text = malloc(((size_t)size + 1) * sizeof *text);
    for (long i = 0; i < size; ++i) text[i] = bytes[i];

    /* new FmIndexBuilder().setSampleRate(32).setEnableExtraction(true).build(text), suffix array on GPU 0 */
    fmx_index *idx = NULL;
    int32_t rounds = 0;
    double seconds = 0;
    int rc = fmx_build_on_device(text, (int32_t)size, 32, 1, 0, &idx, &rounds, NULL, &seconds);
    if (rc) die("fmx_build_on_device", rc);
    if ((rc = fmx_to_device(idx, 0))) die("fmx_to_device", rc);
    printf("indexed %d chars (alphabet %d), %d doubling rounds, device stage %.3f s\n", fmx_input_length(idx) - 1,
           fmx_alphabet_length(idx), rounds, seconds);

    /* the batch: all patterns of the command line in one call */
    const int32_t n = argc - 2, max_lines = 3, row = 400;
    int32_t *off = calloc((size_t)n + 1, sizeof *off);
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += strlen(argv[i + 2]);
    uint16_t *pat = malloc((total + 1) * sizeof *pat);
    for (int i = 0, at = 0; i < n; ++i) {
        for (const char *p = argv[i + 2]; *p; ++p) pat[at++] = (unsigned char)*p;
        off[i + 1] = at;
    }
    int32_t *counts = calloc((size_t)n, sizeof *counts), *status = calloc((size_t)n, sizeof *status);
    if ((rc = fmx_count_batch(idx, pat, off, n, counts, NULL, status))) die("fmx_count_batch", rc); /* FmIndex.count */

    /* locate + extractUntilBoundary('\n') per hit, fused on the device */
    const size_t slots = (size_t)n * max_lines;
    int32_t *locs = calloc(slots, sizeof *locs), *found = calloc((size_t)n, sizeof *found);
    int32_t *len = calloc(slots, sizeof *len), *hit_status = calloc(slots, sizeof *hit_status), *aux = calloc(slots, sizeof *aux);
    uint16_t *rows = calloc(slots * row, sizeof *rows);
    if ((rc = fmx_locate_lines_batch(idx, pat, off, n, max_lines, '\n', 0, row, locs, found, rows, len, NULL, status, hit_status, aux)))
        die("fmx_locate_lines_batch", rc);
    for (int i = 0; i < n; ++i) {
        if (status[i]) {
            printf("'%s': %s\n", argv[i + 2], fmx_status_message(status[i]));
            continue;
        }
        printf("'%s': %d occurrence(s)\n", argv[i + 2], counts[i]);
        for (int k = 0; k < found[i]; ++k) {
            const size_t q = (size_t)i * max_lines + (size_t)k;
            printf("  @%d: ", locs[q]);
            if (hit_status[q]) {
                printf("(%s)\n", fmx_status_message(hit_status[q]));
                continue;
            }
            for (int c = 0; c < len[q]; ++c) putchar(rows[q * row + (size_t)c] < 128 ? (int)rows[q * row + (size_t)c] : '?');
            putchar('\n');
        }
    }
    fmx_free(idx);
    return 0;
}
