// fmx_api.cpp — the C ABI of include/fmx.h: index lifetime, blob management, and the batch entry
// points that copy operands, launch the HIP kernels (fmx_kernels.hip) and copy results back.
// There is no CPU query path: without a HIP device every batch call fails with FMX_E_NO_DEVICE.
#include "../../include/fmx.h"
#include "fmx_device.hpp"
#include "fmx_build_stage.hpp"
#include "fmx_model.hpp"
#include "fmx_plan.hpp"

#include <hip/hip_runtime.h>

#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

// fmx_kernels.hip is compiled twice: namespace fmx serves expanded images, fmxc compact ones (BlobHeader.compact)
#define FMX_KERNEL_API                                                                                                  \
    int launch_suffix_level1(const fmx::DevIndex &, fmx::SuffixSlot *, uint32_t *, uint32_t, hipStream_t);              \
    int launch_suffix_expand(const fmx::DevIndex &, int, const fmx::SuffixSlot *, uint32_t, int, int, fmx::SuffixSlot *, uint32_t *, uint32_t, hipStream_t); \
    int launch_suffix_insert(const fmx::DevIndex &, const fmx::SuffixSlot *, uint32_t, int, fmx::SuffixSlot *, hipStream_t); \
    int launch_suffix_order1(const fmx::DevIndex &, float *, hipStream_t);                                              \
    int launch_win_build(const fmx::DevIndex &, int, uint32_t, fmx::Quad *, uint32_t *, hipStream_t);                   \
    int launch_win_other(const fmx::DevIndex &, int, uint32_t, fmx::Quad *, const uint32_t *, uint16_t *, uint32_t *, int, uint64_t *, uint32_t, hipStream_t); \
    int launch_win_flat(const fmx::DevIndex &, int, uint32_t, uint32_t *, uint32_t *, uint64_t *, uint32_t, hipStream_t);           \
    int launch_count_plan(const fmx::DevIndex &, int, const uint16_t *, const int32_t *, int32_t, void *, size_t, bool, fmx::CountPlan *, \
                          hipStream_t);                                                                                 \
    int launch_count(const fmx::DevIndex &, int, const uint16_t *, const int32_t *, const fmx::CountPlan *, bool, int32_t, int32_t *, \
                     int32_t *, int32_t *, int32_t *, hipStream_t);                                                     \
    size_t count_workspace_bytes(const fmx::DevIndex &, int32_t n);                                                     \
    int launch_locate_walk(const fmx::DevIndex &, int, const int32_t *, int32_t, int32_t, int32_t *, int32_t, int32_t *, \
                           int32_t *, int32_t *, const int32_t *, void *, size_t, bool, hipStream_t, int64_t *, int64_t); \
    int launch_segment_commit(int32_t *, int32_t *, const int32_t *, const int32_t *, int32_t, int32_t, int, hipStream_t); \
    size_t walk_workspace_bytes(const fmx::DevIndex &, int32_t n);                                                      \
    int launch_extract(const fmx::DevIndex &, int, const int32_t *, const int32_t *, int64_t, uint16_t *, int32_t, int32_t, \
                       int32_t *, int32_t *, int32_t *, const int32_t *, int32_t, int32_t, void *, size_t, bool, hipStream_t); \
    int launch_extract_boundary(const fmx::DevIndex &, int, const int32_t *, int64_t, uint16_t, int, uint16_t *, int32_t, \
                                int32_t, int32_t *, int32_t *, int32_t *, int32_t *, void *, size_t, const int32_t *, int32_t, \
                                void *, size_t, bool, hipStream_t);                                                     \
    size_t boundary_workspace_bytes(const fmx::DevIndex &, int64_t n, int n_cu);                                        \
    size_t boundary_order_bytes(const fmx::DevIndex &, int64_t n);                                                      \
    int launch_rrr_rank_ones(const fmx::DevIndex &, int, const int32_t *, int32_t, int32_t *, hipStream_t);             \
    int launch_rrr_access(const fmx::DevIndex &, int, const int32_t *, int32_t, uint8_t *, int32_t *, hipStream_t);     \
    int launch_segment_add_counts(int64_t *, int64_t *, int32_t *, const int32_t *, const int32_t *, const int32_t *, int32_t, \
                                  int, hipStream_t);                                                                    \
    int launch_segment_append_hits(int64_t *, int32_t *, int32_t *, const int32_t *, const int32_t *, const int32_t *, int32_t, \
                                   int32_t, int64_t, int, hipStream_t);                                                 \
    int launch_fill_offsets(int32_t *, int32_t, int32_t, int32_t, hipStream_t);                                         \
    int launch_wt_rank(const fmx::DevIndex &, int, const int64_t *, const int32_t *, int32_t, int64_t *, int32_t *, hipStream_t); \
    int launch_wt_inverse_select(const fmx::DevIndex &, int, const int64_t *, int32_t, int64_t *, int32_t *, hipStream_t); \
    int set_option(const char *, int);                                                                                  \

namespace fmx {
FMX_KERNEL_API
}  // namespace fmx
namespace fmxc {
FMX_KERNEL_API
}  // namespace fmxc
#undef FMX_KERNEL_API
struct fmx_index;
static bool image_is_compact(const fmx_index *idx);
// k_<launcher>(idx, args...): the launcher of the namespace that serves this index's image
#define FMX_DISPATCH_FN(NAME)                                                                             \
    template <class... A>                                                                                 \
    static auto k_##NAME(const fmx_index *idx, A &&...a) {                                                \
        return image_is_compact(idx) ? fmxc::NAME(std::forward<A>(a)...) : fmx::NAME(std::forward<A>(a)...); \
    }
FMX_DISPATCH_FN(launch_suffix_level1)
FMX_DISPATCH_FN(launch_suffix_expand)
FMX_DISPATCH_FN(launch_suffix_insert)
FMX_DISPATCH_FN(launch_suffix_order1)
FMX_DISPATCH_FN(launch_win_build)
FMX_DISPATCH_FN(launch_win_other)
FMX_DISPATCH_FN(launch_win_flat)
FMX_DISPATCH_FN(launch_count_plan)
FMX_DISPATCH_FN(launch_count)
FMX_DISPATCH_FN(count_workspace_bytes)
FMX_DISPATCH_FN(launch_locate_walk)
FMX_DISPATCH_FN(walk_workspace_bytes)
FMX_DISPATCH_FN(launch_extract)
FMX_DISPATCH_FN(launch_extract_boundary)
FMX_DISPATCH_FN(boundary_workspace_bytes)
FMX_DISPATCH_FN(boundary_order_bytes)
FMX_DISPATCH_FN(launch_wt_rank)
FMX_DISPATCH_FN(launch_wt_inverse_select)
#undef FMX_DISPATCH_FN

struct fmx_index {
    fmx::FmModel model;
    bool has_model = false;
    bool from_stream = false;    // the model was parsed from a caller's bytes (fmx_load), not built here: its image is validated
    std::vector<uint8_t> blob;   // host image (empty for attached device blobs)
    fmx::BlobHeader hdr;         // host copy of the header
    void *d_blob = nullptr;
    size_t d_len = 0;
    int device = -1;
    int n_cu = 256;
    bool owns_device = false;
    bool wavelet_only = false;  // built by fmx_wavelet_build: only the wavelet entry points apply
    bool rrr_only = false;      // built by fmx_rrr_build: only the RrrVector entry points apply
    double wavelet_device_seconds = 0;  // fmx_build_on_device: seconds of the wavelet encode in HBM (0: host encoder)
    void *d_suffix_table = nullptr;     // DevIndex.suffix_table (owned, whoever owns the image)
    void *d_suffix_order1 = nullptr;    // DevIndex.suffix_order1 (owned likewise)
    void *d_self = nullptr;             // DevIndex.self: the resident copy of `dev` the kernels' cold routes read (owned likewise)
    void *d_win = nullptr;              // DevIndex.win: the window directory's cells (owned likewise)
    void *d_win_other = nullptr;        // DevIndex.win_other: ... and the entries of the positions no class holds
    size_t win_bytes = 0;               // both together
    uint32_t win_unclean = 0;           // entries that carry a status or `suspect` (statistics)
    size_t suffix_table_bytes = 0;
    uint32_t suffix_table_strings = 0;  // strings (of 2 .. suffix_chars codes) the table holds
    uint32_t suffix_table_deepest = 0;  // ... of which strings of suffix_chars codes: what a batch's patterns spread over after the lookup
    fmx::DevIndex dev;
    // per-(stream, kind) scratch of the device-pointer entry points (grow-only; freed with the index):
    // kind 0 = plan stage (order + code words), kind 1 = extractUntilBoundary windows
    mutable std::mutex ws_mutex;
    mutable std::map<std::pair<void *, int>, std::pair<void *, size_t>> ws;
    struct Plan {  // the last fmx_count_plan_dev result per stream: the batch's records in processing order
        fmx::CountPlan plan;
        const uint16_t *pat = nullptr;
        const int32_t *pat_off = nullptr;  // the records carry lengths and code words cut with THESE offsets
    };
    // erased whenever anything else plans on the stream or its plan scratch moves: a stale handle then simply
    // means "the caller's order" (k_count maps the characters itself) instead of reading another batch's records
    mutable std::map<void *, Plan> plans;
    // a side stream (+ events) per caller's stream for the segment-set entry points: segment s + 1's range search runs beside
    // segment s's walk (locate_segments_impl); created on first use, destroyed with the index
    struct SideLane {
        hipStream_t s = nullptr;
        std::vector<hipEvent_t> ev;
    };
    mutable std::map<void *, SideLane> side;
};

static bool image_is_compact(const fmx_index *idx) { return idx->hdr.compact != 0; }

namespace {

thread_local std::string g_err;
std::atomic<int> g_wavelet_on_device{1};  // option "wavelet_on_device": 0 = fmx_build_on_device encodes the wavelet tree on the host
std::atomic<int> g_suffix_table_mb{256};  // option "suffix_table_mb": budget of the suffix table of indexes made resident afterwards (0 = none)
std::atomic<int> g_suffix_table_chars{8};  // option "suffix_table_chars": its depth (characters; the size limit and the key width may cut it)
// The plan stage (suffix order of a batch) pays while MANY patterns share the table string they start from — their first
// steps then read the same lines.  Option "plan_min_per_string": a batch is planned only if it holds at least this many
// patterns per string of the table's deepest level (0 = every batch of sort_min patterns or more is planned).  Measured
// (tools/depth_sort_probe.py, tools/nosort_probe.py): 1 M patterns over 26 K strings (depth 4): planned 0.168 ms, caller's
// order 0.179; over 227 K strings (depth 5): 0.150 / 0.141; 65,536 patterns over 227 K: 0.045 / 0.023.
std::atomic<int> g_plan_min_per_string{16};
std::atomic<int> g_plan_sa_min{786432};   // option "plan_sa_min" (plan_pays)
std::atomic<int> g_plan_sa_key_api{2};     // mirror of the kernels' option "plan_sa_key"
std::atomic<int> g_code_bits_12_api{1};   // mirror of the kernels' option "code_bits_12": the key width of suffix tables grown from now on
std::atomic<int> g_suffix_table_in_use{1};  // mirror of the kernels' A/B option "suffix_table": launches told to ignore the table plan as if there were none
std::atomic<int> g_suffix_table_image_fraction{8};  // option "suffix_table_image_fraction": the table stays below image / this (0 = only the budget counts)
// option "window_cells": indexes made resident afterwards grow a window directory (fmx_device.hpp "window directory": 64 bytes per
// 112 text characters + 8 per position no class holds, beside the image) — 0 = none, 1 = always, 2 (default) = where it fits a
// quarter of the device's free memory AND the absolute budget "window_cells_mb" (per index; default 65,536 MiB: a process that
// holds many indexes lowers it, or the quarter rule shrinks what is free geometrically)
std::atomic<int> g_window_cells{2};
std::atomic<int> g_window_cells_mb{65536};
// option "window_entry_bytes": the directory's entries — 0 = four bytes (the row; the symbol by a search over cumulativeCounts) where
// those fit LDS (fmx::kWinSymbolSearchMax), six bytes otherwise; 4 / 6 = that form whatever the alphabet (tests, A/B)
std::atomic<int> g_window_entry_bytes{0};
// option "window_flat_fraction": under window_cells = 2 the directory takes its FLAT form (4 bytes per text byte, every step of a walk
// one sector) where that costs at most 1 / this of the device's memory (default 128); 0 = never by itself (window_cells = 3 asks by name)
std::atomic<int> g_window_flat_fraction{128};
// host-buffer count(): batches of at least this many patterns go through the pipeline (smaller ones: one copy in, kernels, one copy out)
std::atomic<int> g_host_small_max{2048};  // option "host_small_max": host-array calls of at most this many patterns / queries go through one mapped pinned block (0: off)
std::atomic<int> g_host_pipeline_min{131072};
std::atomic<int> g_host_mapped{1};  // option "host_mapped": every array of a host-buffer count registered -> one launch over the mapped arrays, no copies
std::atomic<int> g_host_direct_stores{1};  // option "host_direct_stores": the pipeline's kernels store results straight into registered arrays
std::atomic<int> g_host_pipeline_chunk{262144};  // patterns per stage of that pipeline
// what fmx_host_register pinned: start -> bytes (mapped_range trusts nothing else)
std::mutex g_registered_mutex;
std::map<uintptr_t, size_t> g_registered;
std::atomic<int> g_sb_cache_limit{320};  // option "sb_cache_limit": applies to indexes made resident afterwards (tests: 0 = no LDS cache)
int fail(int code, const std::string &msg) {
    g_err = msg;
    // a failed runtime call also stays behind as the thread's "last error": taken off with the report, or the NEXT call's
    // launchers — which ask hipGetLastError() after their launches — would report this call's failure as their own
    if (code == FMX_E_HIP) (void)hipGetLastError();
    return code;
}
// (a failed runtime call also leaves its error as the thread's "last error": taken off here, or the next call's launchers —
// which ask hipGetLastError() after their launches — would report this call's failure as their own)
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            (void)hipGetLastError();                                                          \
            return fail(FMX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));       \
        }                                                                                     \
    } while (0)

int ensure_blob(fmx_index *idx) {
    if (!idx->blob.empty()) return FMX_OK;
    if (!idx->has_model) return fail(FMX_E_ARG, "index has neither a model nor a host blob");
    std::string err;
    int rc;
    try {
        rc = idx->rrr_only ? fmx::flatten_rrr_only(idx->model.sampled, idx->blob, err)
                           : fmx::flatten_model(idx->model, idx->blob, err);
    } catch (const std::exception &e) {  // (an image this host cannot hold: the ABI does not throw)
        idx->blob.clear();
        return fail(FMX_E_UNSUPPORTED, std::string("flattening the index: ") + e.what());
    }
    if (rc) return fail(rc == -3 ? FMX_E_FORMAT : FMX_E_UNSUPPORTED, err);
    // validate_model is about the stream's shapes; the image's own invariants (every mapping entry, skip pointer, path and
    // node record, sample and count inside its table) are what the kernels rely on: an image made from a caller's bytes has
    // to pass them too before any kernel may walk it (tests/cpp/fuzz_load.cpp)
    if (idx->from_stream && !idx->rrr_only && fmx::validate_blob(idx->blob.data(), idx->blob.size(), err)) {
        idx->blob.clear();
        return fail(FMX_E_FORMAT, err);
    }
    memcpy(&idx->hdr, idx->blob.data(), sizeof(fmx::BlobHeader));
    return FMX_OK;
}

void make_dev_index(fmx_index *idx) {
    const fmx::BlobHeader &h = idx->hdr;
    const uint8_t *b = static_cast<const uint8_t *>(idx->d_blob);
    fmx::DevIndex &d = idx->dev;
    auto at = [&](uint32_t off) { return b + ((uint64_t)off << 3); };
    d.base = b;
    d.C = reinterpret_cast<const int32_t *>(at(h.off_c));
    d.look_up = reinterpret_cast<const int32_t *>(at(h.off_lookup));
    d.char2code = reinterpret_cast<const int16_t *>(at(h.off_char2code));
    d.suffix_words = reinterpret_cast<const uint32_t *>(at(h.off_suffixes));
    d.pos_words = reinterpret_cast<const uint32_t *>(at(h.off_positions));
    d.sbc = reinterpret_cast<const fmx::SbcEntry *>(at(h.off_sbc));
    d.sbd = reinterpret_cast<const fmx::SbDesc *>(at(h.off_sbdesc));
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
    d.suffix_table = nullptr;
    d.suffix_chars = 0;
    d.suffix_key_bits = fmx::fmx_code_bits_for(h.wt_sigma, g_code_bits_12_api.load() != 0);
    d.suffix_shift = 0;
    d.suffix_mask = 0;
    d.suffix_order1 = nullptr;
    d.win = nullptr;
    d.win_other = nullptr;
    d.win_full = nullptr;
    d.win_entry4 = 0;
    d.win_flat = 0;
    d.c_lds = nullptr;
    d.c_lut = nullptr;
    d.c_lut_shift = 0;
    d.sb_cache = nullptr;
    d.sb_cache_limit = g_sb_cache_limit;
    d.wt_size = (uint32_t)h.wt_size;
    d.self = nullptr;
}

// DevIndex.self: a copy of the launch-independent DevIndex in HBM (no LDS cache, no suffix table: a cold route needs neither),
// pointing at itself.  Called after make_dev_index, on the index's device.
int publish_dev_index(fmx_index *idx) {
    if (!idx->d_self) HIP_TRY(hipMalloc(&idx->d_self, sizeof(fmx::DevIndex)));
    fmx::DevIndex copy = idx->dev;
    copy.self = static_cast<const fmx::DevIndex *>(idx->d_self);
    copy.sb_cache = nullptr;
    copy.suffix_table = nullptr;
    copy.suffix_order1 = nullptr;
    copy.win = nullptr;  // (a cold route is the tree walk itself; the directory is grown from its answers)
    copy.win_other = nullptr;
    copy.win_full = nullptr;
    copy.win_entry4 = 0;
    copy.win_flat = 0;
    copy.c_lds = nullptr;
    copy.c_lut = nullptr;
    HIP_TRY(hipMemcpy(idx->d_self, &copy, sizeof(copy), hipMemcpyHostToDevice));
    idx->dev.self = copy.self;
    return FMX_OK;
}

int require_device(const fmx_index *idx, bool rrr_handle = false) {
    if (!idx) return fail(FMX_E_ARG, "null index");
    if (idx->rrr_only != rrr_handle)
        return fail(FMX_E_ARG, rrr_handle ? "not an RrrVector handle" : "an RrrVector handle answers only fmx_rrr_* calls");
    if (!idx->d_blob) return fail(FMX_E_NO_DEVICE, "index is not resident on a HIP device (call fmx_to_device)");
    return FMX_OK;
}

constexpr int kWsPlan = 0, kWsBoundary = 1, kWsWalk = 2, kWsSegRange = 3, kWsSegCounts = 4;  // (kWsWalk: the walk order of locate, a plan-like head; kWsSegRange: a segment set's second {found, status, range} buffers)
std::atomic<int> g_segments_direct{1};   // option "segments_direct": 0 = every segment's hits staged and appended (A/B)
std::atomic<int> g_segments_overlap{1};  // option "segments_overlap": 0 = a segment set's kernels all on the caller's stream (A/B)
std::atomic<int> g_segments_overlap_min{262144};  // option "segments_overlap_min": ... and only for batches at least this large

// the side stream of `stream` with at least n_events events (nullptr: could not be made — the caller stays on one stream)
fmx_index::SideLane *side_lane(const fmx_index *idx, void *stream, size_t n_events) {
    std::lock_guard<std::mutex> lock(idx->ws_mutex);
    fmx_index::SideLane &l = idx->side[stream];
    if (!l.s && hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking) != hipSuccess) {
        l.s = nullptr;
        (void)hipGetLastError();
        return nullptr;
    }
    while (l.ev.size() < n_events) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        l.ev.push_back(e);
    }
    return &l;
}

// scratch of at least `bytes` for work enqueued on `stream`; reused across calls on the same stream
int get_workspace(const fmx_index *idx, void *stream, int kind, size_t bytes, void **out) {
    *out = nullptr;
    if (bytes == 0) return FMX_OK;
    std::lock_guard<std::mutex> lock(idx->ws_mutex);
    auto &slot = idx->ws[{stream, kind}];
    if (slot.second < bytes) {
        const bool has_head = kind == kWsPlan || kind == kWsWalk;
        if (has_head && bytes < fmx::kPlanHeadBytes) bytes = fmx::kPlanHeadBytes;
        if (slot.first) {
            HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
            (void)hipFree(slot.first);
            slot = {nullptr, 0};
            if (kind == kWsPlan) idx->plans.erase(stream);  // its perm / code words lived in the freed block
        }
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, bytes));
        // the plan kernels expect the head of their workspace zeroed and leave it zeroed (fmx_plan.hpp)
        if (has_head) HIP_TRY(hipMemsetAsync(p, 0, fmx::kPlanHeadBytes, static_cast<hipStream_t>(stream)));
        slot = {p, bytes};
    }
    *out = slot.first;
    return FMX_OK;
}

// Host-buffer entry points size their device copy of the characters from pat_off[n] alone: offsets that start below 0 or
// run backwards would make the kernels read outside it (k_plan_codes fetches 16-byte pattern tails).  One O(n) pass.
int check_offsets(const int32_t *pat_off, int32_t n) {
    if (!pat_off || n < 0) return fail(FMX_E_ARG, "bad arguments");
    if (pat_off[0] < 0) return fail(FMX_E_ARG, "pattern offsets start below 0");
    for (int32_t i = 0; i < n; ++i)
        if (pat_off[i + 1] < pat_off[i]) return fail(FMX_E_ARG, "pattern offsets decrease");
    return FMX_OK;
}

#define H2D(dst, src, bytes) HIP_TRY(hipMemcpy((dst), (src), (bytes), hipMemcpyHostToDevice))
// A batch refers to the characters [pat_off[0], pat_off[n]) of `pat`: only those travel, to the same element offsets of the device
// copy — a shard of a larger batch (fmx_*_multi hands pat_off + lo to the single-index call) ships its own characters, not
// everything before them.  (check_offsets has seen pat_off[0] >= 0.)
static inline size_t first_char(const int32_t *pat_off) { return (size_t)(pat_off[0] > 0 ? pat_off[0] : 0); }
#define D2H(dst, src, bytes) HIP_TRY(hipMemcpy((dst), (src), (bytes), hipMemcpyDeviceToHost))

// Device scratch of the host-buffer entry points.  hipMalloc / hipFree cost more than a small batch's kernels, so
// blocks are recycled: power-of-two size classes per device, at most kScratchCacheLimit bytes kept.
constexpr size_t kScratchCacheLimit = (size_t)2 << 30;
struct ScratchCache {
    std::mutex mutex;
    std::multimap<std::pair<int, size_t>, void *> free_blocks;
    size_t cached_bytes = 0;
    void *take(int device, size_t bytes) {
        std::lock_guard<std::mutex> lock(mutex);
        auto it = free_blocks.find({device, bytes});
        if (it == free_blocks.end()) return nullptr;
        void *p = it->second;
        free_blocks.erase(it);
        cached_bytes -= bytes;
        return p;
    }
    bool give(int device, size_t bytes, void *p) {
        std::lock_guard<std::mutex> lock(mutex);
        if (cached_bytes + bytes > kScratchCacheLimit) return false;
        free_blocks.insert({{device, bytes}, p});
        cached_bytes += bytes;
        return true;
    }
    void release_all() {
        std::lock_guard<std::mutex> lock(mutex);
        int current = 0;
        (void)hipGetDevice(&current);
        for (auto &kv : free_blocks) {
            (void)hipSetDevice(kv.first.first);
            (void)hipFree(kv.second);
        }
        (void)hipSetDevice(current);
        free_blocks.clear();
        cached_bytes = 0;
    }
};
ScratchCache g_scratch;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int device = 0;
    ~DevBuf() {
        if (p && !g_scratch.give(device, bytes, p)) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) {
        bytes = 256;
        while (bytes < n) bytes <<= 1;
        (void)hipGetDevice(&device);
        p = g_scratch.take(device, bytes);
        return p ? hipSuccess : hipMalloc(&p, bytes);
    }
    template <typename T>
    T *as() {
        return static_cast<T *>(p);
    }
};

// Pinned host staging of the pipelined host-buffer entry points, recycled like the device blocks (hipHostMalloc costs
// more than a batch).
struct PinCache {
    std::mutex mutex;
    std::multimap<size_t, void *> free_blocks;
    size_t cached_bytes = 0;
    void *take(size_t bytes) {
        std::lock_guard<std::mutex> lock(mutex);
        auto it = free_blocks.find(bytes);
        if (it == free_blocks.end()) return nullptr;
        void *p = it->second;
        free_blocks.erase(it);
        cached_bytes -= bytes;
        return p;
    }
    bool give(size_t bytes, void *p) {
        std::lock_guard<std::mutex> lock(mutex);
        if (cached_bytes + bytes > ((size_t)512 << 20)) return false;
        free_blocks.insert({bytes, p});
        cached_bytes += bytes;
        return true;
    }
    void release_all() {
        std::lock_guard<std::mutex> lock(mutex);
        for (auto &kv : free_blocks) (void)hipHostFree(kv.second);
        free_blocks.clear();
        cached_bytes = 0;
    }
};
PinCache g_pinned;
struct PinBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~PinBuf() {
        if (p && !g_pinned.give(bytes, p)) (void)hipHostFree(p);
    }
    hipError_t alloc(size_t n) {
        bytes = 4096;
        while (bytes < n) bytes <<= 1;
        p = g_pinned.take(bytes);
        return p ? hipSuccess : hipHostMalloc(&p, bytes, hipHostMallocDefault);
    }
    template <typename T>
    T *as() {
        return static_cast<T *>(p);
    }
};

constexpr size_t kHostSmallBytes = 1 << 20;  // ... and at most this much in one block (rows of extract / locate included)
// the one pinned block of a small call: pieces handed out 16-byte aligned, each with the address the device sees it at
struct SmallBlock {
    PinBuf buf;
    uint8_t *h = nullptr, *d = nullptr;
    size_t used = 0;
    static size_t up(size_t v) { return (v + 15) & ~(size_t)15; }
    // FMX_OK; -1 = pinned memory is not mapped here (the caller takes the copying path); < -1 = a HIP failure
    int init(size_t total) {
        if (buf.alloc(total + 16) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        h = buf.as<uint8_t>();
        void *dv = nullptr;
        if (hipHostGetDevicePointer(&dv, h, 0) != hipSuccess || !dv) {
            (void)hipGetLastError();
            return -1;
        }
        d = static_cast<uint8_t *>(dv);
        return FMX_OK;
    }
    template <typename T>
    T *take(size_t count, T **device) {
        T *host = reinterpret_cast<T *>(h + used);
        *device = reinterpret_cast<T *>(d + used);
        used += up(count * sizeof(T));
        return host;
    }
};

// Staging of PAGEABLE caller arrays (what a JVM heap array behind GetPrimitiveArrayCritical is): a plain hipMemcpy from pageable
// memory is staged by the runtime on the calling thread at a fraction of the link's rate (1 M x 8 chars: 1.0 ms per call against a
// PCIe floor of 0.34).  Here the bytes go through pinned staging of the library's own, copied by a few host threads side by side —
// a chunk's copy beside the DMA of the chunk before it — and travel by asynchronous DMA at the link's rate.
// A small pool of copy threads, started on first use and never stopped (like the per-device workers of fmx_multi.cpp).
struct CopyPool {
    struct Task {
        char *dst;
        const char *src;
        size_t bytes;
        std::atomic<int> *open;
    };
    std::mutex m;
    std::condition_variable cv;
    std::deque<Task> q;
    int started = 0;
    void worker() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return !q.empty(); });
                t = q.front();
                q.pop_front();
            }
            memcpy(t.dst, t.src, t.bytes);
            t.open->fetch_sub(1, std::memory_order_release);
        }
    }
    // dst[0 .. bytes) = src[0 .. bytes) by `threads` threads (the caller is one of them); returns when all of it is there
    void copy(void *dst, const void *src, size_t bytes, int threads) {
        constexpr size_t kMinPart = (size_t)256 << 10;
        int parts = threads;
        if ((size_t)parts > bytes / kMinPart) parts = (int)(bytes / kMinPart);
        if (parts <= 1) {
            memcpy(dst, src, bytes);
            return;
        }
        const size_t per = ((bytes / (size_t)parts) + 63) & ~(size_t)63;
        std::atomic<int> open{parts - 1};
        {
            std::lock_guard<std::mutex> lock(m);
            while (started < threads - 1) {  // (thread creation may throw: the entry points' guard reports it)
                std::thread(&CopyPool::worker, this).detach();
                ++started;
            }
            for (int i = 1; i < parts; ++i) {
                const size_t lo = (size_t)i * per, hi = i + 1 == parts ? bytes : std::min(bytes, lo + per);
                q.push_back(Task{static_cast<char *>(dst) + lo, static_cast<const char *>(src) + lo, hi > lo ? hi - lo : 0, &open});
            }
        }
        cv.notify_all();
        memcpy(dst, src, std::min(per, bytes));
        while (open.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }
};
CopyPool &copy_pool() {
    static CopyPool *pool = new CopyPool();  // leaked on purpose: its threads outlive every static destructor
    return *pool;
}
// option "host_stage_threads": host threads that stage a pageable array into pinned memory (0 / 1 = the runtime's own staging, the
// default BY MEASUREMENT: with the result copies gone — below — a 1 M-pattern call takes 0.58 ms with the runtime's staging and
// 0.70-0.73 with 3-6 threads of this pool: the runtime's copy is not what the call was waiting for; profiles/r06_experiments.txt 4)
std::atomic<int> g_host_stage_threads{0};

// Three streams per (host thread, device) for the pipelined host-buffer entry points.  Created on first use and never
// destroyed (a thread_local destructor would run while the HIP runtime may already be shutting down).
constexpr int kPipeStreams = 3;  // copies in, kernels, copies out
constexpr int kPipeEvents = 16;
struct PipeStreams {
    int device = -1;
    hipStream_t s[kPipeStreams] = {nullptr, nullptr, nullptr};
    hipEvent_t in[kPipeEvents] = {}, counted[kPipeEvents] = {}, done[kPipeEvents] = {};
};
// Sets whose thread has ended wait here for the next thread that needs one (a JVM's worker threads come and go: without this
// every new thread made 3 streams and 48 events that nobody ever used again).  Never destroyed: a thread_local destructor may
// run while the HIP runtime is shutting down, so it only hands its sets back.
struct PipePool {
    std::mutex mutex;
    std::vector<PipeStreams *> idle;
};
PipePool &pipe_pool() {
    static PipePool *pool = new PipePool();  // (leaked on purpose: outlives every thread_local destructor)
    return *pool;
}
struct ThreadPipeSets {
    std::vector<PipeStreams *> sets;
    ~ThreadPipeSets() {
        PipePool &pool = pipe_pool();
        std::lock_guard<std::mutex> lock(pool.mutex);
        for (PipeStreams *p : sets) pool.idle.push_back(p);
    }
};
int pipe_streams(int device, PipeStreams **out) {
    // one set per device this thread has used (a thread that alternates between devices keeps both; nothing is re-created)
    thread_local ThreadPipeSets mine;
    std::vector<PipeStreams *> &sets = mine.sets;
    for (PipeStreams *p : sets)
        if (p->device == device) {
            *out = p;
            return FMX_OK;
        }
    {
        PipePool &pool = pipe_pool();
        std::lock_guard<std::mutex> lock(pool.mutex);
        for (size_t i = 0; i < pool.idle.size(); ++i)
            if (pool.idle[i]->device == device) {  // (its thread drained its streams before it returned from its last call)
                sets.push_back(pool.idle[i]);
                pool.idle.erase(pool.idle.begin() + (ptrdiff_t)i);
                *out = sets.back();
                return FMX_OK;
            }
    }
    std::unique_ptr<PipeStreams> ps(new PipeStreams());
    auto undo = [&]() {  // a set that could not be completed is taken apart again
        for (int i = 0; i < kPipeStreams; ++i)
            if (ps->s[i]) (void)hipStreamDestroy(ps->s[i]);
        for (auto *set : {ps->in, ps->counted, ps->done})
            for (int i = 0; i < kPipeEvents; ++i)
                if (set[i]) (void)hipEventDestroy(set[i]);
    };
    for (int i = 0; i < kPipeStreams; ++i)
        if (hipError_t e = hipStreamCreateWithFlags(&ps->s[i], hipStreamNonBlocking); e != hipSuccess) {
            undo();
            HIP_TRY(e);
        }
    for (auto *set : {ps->in, ps->counted, ps->done})
        for (int i = 0; i < kPipeEvents; ++i)
            if (hipError_t e = hipEventCreateWithFlags(&set[i], hipEventDisableTiming); e != hipSuccess) {
                undo();
                HIP_TRY(e);
            }
    ps->device = device;
    sets.push_back(ps.release());
    *out = sets.back();
    return FMX_OK;
}

// A host-buffer entry point runs its kernels on a stream of the calling thread's own — not the default stream, whose work the
// plain copies around the kernels would wait behind (fmx_locate_lines_batch: 9.7 -> 6.9 ms per call) — and waits for that stream
// whichever way it leaves: declare it AFTER the call's device buffers and scratch, so that they return to their caches later.
struct HostCallStream {
    hipStream_t s = nullptr;
    int init(int device) {
        PipeStreams *ps = nullptr;
        const int rc = pipe_streams(device, &ps);
        if (rc == FMX_OK) s = ps->s[1];
        return rc;
    }
    ~HostCallStream() {
        if (s) (void)hipStreamSynchronize(s);
    }
};


// One pass over pat_off[lo .. hi]: offsets never decrease and end at or below `limit`; *uniform = every pattern of the
// run has the same length (then the offsets need not travel: k_fill_offsets).  Branch-free (AVX2 where the host has it); one
// core reads 1 M offsets in 200-300 us whatever the code — the pass is memory-bound; running it on a thread of its own, ahead
// of the issuing thread, was measured slower (the thread's start costs more than the pass hides: profiles/r04_experiments.txt).
// Order is judged by comparing neighbours — never by the sign of a 32-bit difference,
// which wraps ({8, 2000000000, -2000000000, 16} has no negative int32 difference and both ends in range) — and with
// monotonic offsets and checked ends every offset lies inside; equality of lengths may use the wrapping difference.
static bool scan_offsets_plain(const int32_t *o, int32_t lo, int32_t hi, int64_t limit, bool *uniform) {
    const uint32_t m0 = (uint32_t)o[lo + 1] - (uint32_t)o[lo];
    uint32_t bad = 0, diff = 0;
    for (int32_t i = lo; i < hi; ++i) {
        const int32_t a = o[i], b = o[i + 1];
        bad |= (uint32_t)(b < a);
        diff |= ((uint32_t)b - (uint32_t)a) ^ m0;
    }
    *uniform = diff == 0;
    return bad == 0 && o[lo] >= 0 && (int64_t)o[hi] <= limit;
}
__attribute__((target("avx2"))) static bool scan_offsets_avx2(const int32_t *o, int32_t lo, int32_t hi, int64_t limit, bool *uniform) {
    const uint32_t m0 = (uint32_t)o[lo + 1] - (uint32_t)o[lo];
    __m256i vbad = _mm256_setzero_si256(), vdiff = _mm256_setzero_si256();
    const __m256i vm0 = _mm256_set1_epi32((int)m0);
    int32_t i = lo;
    for (; i + 8 <= hi; i += 8) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(o + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(o + i + 1));
        vbad = _mm256_or_si256(vbad, _mm256_cmpgt_epi32(a, b));
        vdiff = _mm256_or_si256(vdiff, _mm256_xor_si256(_mm256_sub_epi32(b, a), vm0));
    }
    uint32_t bad = !_mm256_testz_si256(vbad, vbad), diff = !_mm256_testz_si256(vdiff, vdiff);
    for (; i < hi; ++i) {
        const int32_t a = o[i], b = o[i + 1];
        bad |= (uint32_t)(b < a);
        diff |= ((uint32_t)b - (uint32_t)a) ^ m0;
    }
    *uniform = diff == 0;
    return bad == 0 && o[lo] >= 0 && (int64_t)o[hi] <= limit;
}
bool scan_offsets(const int32_t *pat_off, int32_t lo, int32_t hi, int64_t limit, bool *uniform) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? scan_offsets_avx2(pat_off, lo, hi, limit, uniform) : scan_offsets_plain(pat_off, lo, hi, limit, uniform);
}

// Where a call's device scratch comes from.  Device-pointer entry points: the index's per-(stream, kind) buffers —
// asynchronous, the caller orders work through the stream and uses one stream per thread.  Host-buffer entry
// points: per-call blocks from the recycling cache, so that any number of host threads may query one index at
// once, as with the reference's immutable @ThreadSafe FmIndex (FM:82); the blocks live until the call has
// synchronised.
struct Scratch {
    const fmx_index *idx;
    void *stream;
    bool per_call;
    std::vector<std::unique_ptr<DevBuf>> owned;
    Scratch(const fmx_index *i, void *st, bool call) : idx(i), stream(st), per_call(call) {}
    int get(int kind, size_t bytes, void **out) {
        if (!per_call) return get_workspace(idx, stream, kind, bytes, out);
        *out = nullptr;
        if (bytes == 0) return FMX_OK;
        owned.emplace_back(new DevBuf());
        HIP_TRY(owned.back()->alloc(bytes));
        *out = owned.back()->p;
        return FMX_OK;
    }
};

}  // namespace

// The ABI never throws (an exception crossing into a JVM through JNI aborts it): every entry point that returns a status runs
// inside this guard.  What can throw in here: allocations sized by the caller's data, thread creation.
template <class F>
static int guarded(F &&body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(FMX_E_UNSUPPORTED, "out of host memory");
    } catch (const std::exception &e) {
        return fail(FMX_E_UNSUPPORTED, std::string("internal error: ") + e.what());
    }
}

// fmx_multi.cpp: a shard's failure, reported by a worker thread, becomes the CALLING thread's fmx_last_error
namespace fmx {
int api_fail(int code, const std::string &msg) { return fail(code, msg); }
}  // namespace fmx

extern "C" {

const char *fmx_last_error(void) { return g_err.c_str(); }

void fmx_release_scratch(void) {
    g_scratch.release_all();
    g_pinned.release_all();
}

int fmx_set_option(const char *name, int value) {
    return guarded([&]() -> int {
    if (name && !strcmp(name, "sb_cache_limit")) {
        if (value < 0 || value > 320) return fail(FMX_E_ARG, "bad value");
        g_sb_cache_limit = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "map_by_symbol")) {  // layout of the mapping tables of images flattened from now on
        if (value < -1 || value > 1) return fail(FMX_E_ARG, "bad value");
        fmx::set_map_by_symbol(value);
        return FMX_OK;
    }
    if (name && !strcmp(name, "map_fast")) {  // 0: images flattened from now on keep every mapping entry on the reference's route
        fmx::set_map_fast(value != 0);
        return FMX_OK;
    }
    if (name && !strcmp(name, "suffix_table_mb")) {  // budget for the suffix table of indexes made resident from now on
        if (value < 0 || value > (1 << 16)) return fail(FMX_E_ARG, "bad value");
        g_suffix_table_mb = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "host_mapped")) {
        g_host_mapped = value != 0;
        return FMX_OK;
    }
    if (name && !strcmp(name, "host_direct_stores")) {
        g_host_direct_stores = value != 0;
        return FMX_OK;
    }
    if (name && !strcmp(name, "host_pipeline_min")) {  // host-buffer count(): batches at least this large are pipelined (0 = never)
        if (value < 0) return fail(FMX_E_ARG, "bad value");
        g_host_pipeline_min = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "host_small_max")) {  // batches of at most this many patterns go through one mapped pinned block (0: off)
        if (value < 0) return fail(FMX_E_ARG, "bad value");
        g_host_small_max = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "host_stage_threads")) {  // host threads staging a pageable array into pinned memory (0 / 1: the runtime's own staging)
        if (value < 0 || value > 64) return fail(FMX_E_ARG, "bad value");
        g_host_stage_threads = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "host_pipeline_chunk")) {
        if (value < 65536) return fail(FMX_E_ARG, "bad value");
        g_host_pipeline_chunk = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "suffix_table_chars")) {  // depth of the suffix table of indexes made resident from now on
        if (value < 0 || value > 8) return fail(FMX_E_ARG, "bad value");
        g_suffix_table_chars = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "plan_sa_min")) {
        if (value < 0) return fail(FMX_E_ARG, "bad value");
        g_plan_sa_min = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "plan_min_per_string")) {
        if (value < 0) return fail(FMX_E_ARG, "bad value");
        g_plan_min_per_string = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "suffix_table_image_fraction")) {  // the table stays below image bytes / value (0: only the budget counts)
        if (value < 0) return fail(FMX_E_ARG, "bad value");
        g_suffix_table_image_fraction = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "wavelet_on_device")) {
        g_wavelet_on_device = value != 0;
        return FMX_OK;
    }
    if (name && !strcmp(name, "window_cells")) {  // window directory of indexes made resident from now on: 0 none, 1 always, 2 if it fits, 3 the flat form
        if (value < 0 || value > 3) return fail(FMX_E_ARG, "bad value");
        g_window_cells = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "window_flat_fraction")) {  // the flat form by itself up to 1 / value of the device's memory (0: never)
        if (value < 0) return fail(FMX_E_ARG, "bad value");
        g_window_flat_fraction = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "window_entry_bytes")) {  // entries of the window directories grown from now on: 0 by the alphabet, 4, 6
        if (value != 0 && value != 4 && value != 6) return fail(FMX_E_ARG, "bad value");
        g_window_entry_bytes = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "window_cells_mb")) {  // absolute budget of one index's window directory under "window_cells" = 2
        if (value < 0) return fail(FMX_E_ARG, "bad value");
        g_window_cells_mb = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "segments_overlap_min")) {  // smallest batch whose segment searches run beside the walks (side stream)
        if (value < 0) return fail(FMX_E_ARG, "bad value");
        g_segments_overlap_min = value;
        return FMX_OK;
    }
    if (name && !strcmp(name, "cells_split_blocks")) {  // tests: chunked decoding of short vectors too (same image)
        fmx::set_split_blocks(value);
        return FMX_OK;
    }
    if (name && !strcmp(name, "inv_fast")) {  // 0: images flattened from now on walk inverseSelect the reference's way
        fmx::set_inv_fast(value != 0);
        return FMX_OK;
    }
    if (name && !strcmp(name, "image_compact")) {  // images flattened from now on keep their bit vectors compressed (fmx.h)
        fmx::set_image_compact(value);
        return FMX_OK;
    }
    if (!name) return fail(FMX_E_ARG, "unknown option or bad value");
    if (!strcmp(name, "suffix_table")) g_suffix_table_in_use = value != 0;
    if (!strcmp(name, "plan_sa_key") && value >= 0 && value <= 2) g_plan_sa_key_api = value;
    if (!strcmp(name, "code_bits_12")) g_code_bits_12_api = value != 0;
    if (!strcmp(name, "segments_direct")) {
        g_segments_direct = value != 0;
        return FMX_OK;
    }
    if (!strcmp(name, "segments_overlap")) {
        g_segments_overlap = value != 0;
        return FMX_OK;
    }
    {  // launch options go to both kernel sets
        const int a = fmx::set_option(name, value), b = fmxc::set_option(name, value);
        if (a || b) return fail(FMX_E_ARG, "unknown option or bad value");
    }
    return FMX_OK;
    });
}

int fmx_device_count(void) {
    return guarded([&]() -> int {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
    });
}

int fmx_build(const uint16_t *text, int32_t n, int32_t sample_rate, int enable_extract, fmx_index **out) {
    return guarded([&]() -> int {
    if (!out || (!text && n > 0)) return fail(FMX_E_ARG, "null argument");
    std::unique_ptr<fmx_index> idx(new fmx_index());
    std::string err;
    int rc = fmx::build_model(text, n, sample_rate, enable_extract != 0, idx->model, err);
    if (rc == -2) return fail(FMX_E_ALPHABET, err);
    if (rc) return fail(FMX_E_ARG, err);
    idx->has_model = true;
    *out = idx.release();
    return FMX_OK;
    });
}

int fmx_build_on_device(const uint16_t *text, int32_t n, int32_t sample_rate, int enable_extract, int device,
                        fmx_index **out, int32_t *rounds, int64_t *rows_sorted, double *stage_seconds) {
    return guarded([&]() -> int {
    if (!out || (!text && n > 0) || device < 0) return fail(FMX_E_ARG, "bad arguments");
    std::unique_ptr<fmx_index> idx(new fmx_index());
    std::string err;
    fmx::SaStageStats stats;
    int rc = fmx::build_model(text, n, sample_rate, enable_extract != 0, idx->model, err, device, &stats,
                              g_wavelet_on_device != 0);
    if (rc == -2) return fail(FMX_E_ALPHABET, err);
    if (rc == -5) return fail(FMX_E_NO_DEVICE, err);
    if (rc == -6) return fail(FMX_E_HIP, err);
    if (rc) return fail(FMX_E_ARG, err);
    if (rounds) *rounds = stats.rounds;
    if (rows_sorted) *rows_sorted = (int64_t)stats.rows_sorted;
    if (stage_seconds) *stage_seconds = stats.seconds;
    idx->wavelet_device_seconds = stats.wavelet_seconds;
    idx->has_model = true;
    *out = idx.release();
    return FMX_OK;
    });
}

double fmx_build_wavelet_seconds(const fmx_index *idx) { return idx ? idx->wavelet_device_seconds : 0.0; }

int fmx_load(const uint8_t *ser, size_t len, fmx_index **out) {
    return guarded([&]() -> int {
    if (!out || !ser) return fail(FMX_E_ARG, "null argument");
    std::unique_ptr<fmx_index> idx(new fmx_index());
    std::string err;
    int rc;
    try {
        rc = fmx::parse_model(ser, len, idx->model, err);
    } catch (const std::exception &e) {
        return fail(FMX_E_FORMAT, std::string("reading the stream: ") + e.what());
    }
    if (rc == 2) return fail(FMX_E_VERSION, err);
    if (rc) return fail(FMX_E_FORMAT, err);
    rc = fmx::validate_model(idx->model, err);
    if (rc) return fail(rc, err);
    idx->has_model = true;
    idx->from_stream = true;
    *out = idx.release();
    return FMX_OK;
    });
}

int fmx_save(const fmx_index *idx, int framed, uint8_t **buf, size_t *len) {
    return guarded([&]() -> int {
    if (!idx || !buf || !len) return fail(FMX_E_ARG, "null argument");
    if (!idx->has_model || idx->wavelet_only || idx->rrr_only)
        return fail(FMX_E_ARG, "nothing to serialize (device-attached or wavelet-only handle)");
    std::vector<uint8_t> out;
    fmx::emit_model(idx->model, framed != 0, out);
    uint8_t *p = static_cast<uint8_t *>(malloc(out.size() ? out.size() : 1));
    if (!p) return fail(FMX_E_NOMEM, "out of memory");
    memcpy(p, out.data(), out.size());
    *buf = p;
    *len = out.size();
    return FMX_OK;
    });
}

int fmx_save_key_order_modelled(const fmx_index *idx) {
    return guarded([&]() -> int {
    if (!idx) return fail(FMX_E_ARG, "null argument");
    if (!idx->has_model || idx->wavelet_only || idx->rrr_only)
        return fail(FMX_E_ARG, "nothing to serialize (device-attached or wavelet-only handle)");
    return fmx::key_order_is_modelled(idx->model) ? 1 : 0;
    });
}

void fmx_free_buffer(uint8_t *buf) { free(buf); }

// Everything a handle owns on its device: freed on THAT device (fmx_free; fmx_to_device when the index moves to another GPU —
// every one of these pointers, DevIndex.self included, is an address in the old device's HBM that a kernel on the new device
// must never be handed).
static void release_device_state(fmx_index *idx) {
    if (idx->device >= 0) (void)hipSetDevice(idx->device);
    for (auto &kv : idx->ws)
        if (kv.second.first) (void)hipFree(kv.second.first);
    idx->ws.clear();
    idx->plans.clear();
    if (idx->owns_device && idx->d_blob) (void)hipFree(idx->d_blob);
    if (idx->d_suffix_table) (void)hipFree(idx->d_suffix_table);
    if (idx->d_suffix_order1) (void)hipFree(idx->d_suffix_order1);
    if (idx->d_self) (void)hipFree(idx->d_self);
    if (idx->d_win) (void)hipFree(idx->d_win);
    if (idx->d_win_other) (void)hipFree(idx->d_win_other);
    idx->d_blob = idx->d_suffix_table = idx->d_suffix_order1 = idx->d_self = idx->d_win = idx->d_win_other = nullptr;
    idx->d_len = 0;
    idx->win_bytes = idx->suffix_table_bytes = 0;
    idx->owns_device = false;
    idx->dev = fmx::DevIndex();
    for (auto &kv : idx->side) {
        for (hipEvent_t e : kv.second.ev) (void)hipEventDestroy(e);
        if (kv.second.s) {
            (void)hipStreamSynchronize(kv.second.s);
            (void)hipStreamDestroy(kv.second.s);
        }
    }
    idx->side.clear();
    idx->device = -1;
}

void fmx_free(fmx_index *idx) {
    if (!idx) return;
    release_device_state(idx);
    delete idx;
}

int32_t fmx_input_length(const fmx_index *idx) { return idx->has_model ? idx->model.length : idx->hdr.length; }
int32_t fmx_alphabet_length(const fmx_index *idx) {
    return idx->has_model ? (int32_t)idx->model.map_keys.size() : idx->hdr.n_keys;
}
int32_t fmx_sample_rate(const fmx_index *idx) { return idx->has_model ? idx->model.sample_rate : idx->hdr.sample_rate; }
int32_t fmx_extract_enabled(const fmx_index *idx) {
    return idx->has_model ? (idx->model.enable_extract ? 1 : 0) : idx->hdr.enable_extract;
}

// The window directory of a resident FM-index (fmx_device.hpp "window directory"): one 64-byte cell per 112 BWT positions and an
// 8-byte entry per position none of its window's three classes holds, made on the device from the index's own rank() /
// inverseSelect() (win_build_cell / win_build_other: every number checked against them).  Grown before the suffix table (whose
// growth then already runs over it).  Not having one is never an error.
static void build_window_cells(fmx_index *idx) {
    if (idx->d_win) (void)hipFree(idx->d_win);
    if (idx->d_win_other) (void)hipFree(idx->d_win_other);
    idx->d_win = idx->d_win_other = nullptr;
    idx->win_bytes = 0;
    idx->dev.win = nullptr;
    idx->dev.win_other = nullptr;
    idx->dev.win_full = nullptr;
    idx->dev.win_entry4 = 0;
    idx->dev.win_flat = 0;
    const int mode = g_window_cells.load();
    if (mode == 0 || idx->rrr_only || idx->wavelet_only || idx->hdr.kind != 0 || idx->hdr.wt_size <= 0 || !idx->dev.self) return;
    // THE FLAT FORM (fmx_device.hpp win_step): one 32-bit word per position instead of cells and entries — 4 bytes per text
    // byte, every step of a walk ONE sector (locate 18-24 % faster, extract 10 %).  Asked for by name (window_cells = 3), or
    // taken by the default rule (window_cells = 2) where it is SMALL against the device: at most 1 / window_flat_fraction of the
    // device's memory (default 128: 2.25 GB of 288 — texts up to 512 Mi characters), and like the cells' form inside a quarter of
    // what is free and the per-index budget.  Texts of 2^30 characters and more keep the cells.
    bool flat = mode == 3;
    // (by itself only for alphabets whose cumulativeCounts fit LDS: the flat form's words are rows, and a kernel that wants the symbol
    // searches for it — over 8 KB of LDS, not over a table of up to 32,768 entries in memory)
    if (mode == 2 && g_window_flat_fraction.load() > 0 && (uint64_t)idx->hdr.wt_size <= 0x3fffffffull &&
        idx->hdr.n_c <= fmx::kWinSymbolSearchMax) {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)idx->hdr.wt_size * 4 + ((size_t)4096 + (size_t)idx->hdr.wt_size / 512) * 8 + 64;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            flat = need <= total_b / (size_t)g_window_flat_fraction.load() && need <= free_b / 4 &&
                   need <= ((size_t)g_window_cells_mb.load() << 20);
        else
            (void)hipGetLastError();
    }
    if (flat) {
        const uint64_t n_pos = (uint64_t)idx->hdr.wt_size;
        if (n_pos <= 0x3fffffffull) {
            const size_t flat_bytes = ((size_t)n_pos * 4 + 7) & ~(size_t)7;
            const uint32_t full_cap = (uint32_t)std::min<uint64_t>(4096 + n_pos / 512, 0x3fffffffu);
            void *d_flat = nullptr;
            uint32_t tail[4] = {0, 0, 0, 0};
            if (hipMalloc(&d_flat, flat_bytes + 16 + (size_t)full_cap * 8) == hipSuccess) {
                uint8_t *f8 = static_cast<uint8_t *>(d_flat);
                if (hipMemset(f8 + flat_bytes, 0, 16) == hipSuccess &&
                    k_launch_win_flat(idx, idx->dev, idx->n_cu, (uint32_t)n_pos, static_cast<uint32_t *>(d_flat),
                                      reinterpret_cast<uint32_t *>(f8 + flat_bytes), reinterpret_cast<uint64_t *>(f8 + flat_bytes + 16),
                                      full_cap, nullptr) == 0 &&
                    hipMemcpy(tail, f8 + flat_bytes, sizeof tail, hipMemcpyDeviceToHost) == hipSuccess && tail[1] == 0) {
                    idx->win_unclean = tail[0];
                    idx->d_win = d_flat;
                    idx->win_bytes = (size_t)n_pos * 4 + (size_t)tail[2] * 8;
                    idx->dev.win = static_cast<const fmx::Quad *>(d_flat);
                    idx->dev.win_other = static_cast<const uint16_t *>(d_flat);  // (never read in this form; non-null like win)
                    idx->dev.win_full = reinterpret_cast<const uint64_t *>(f8 + flat_bytes + 16);
                    idx->dev.win_entry4 = 1;  // (the kernels that want symbols stage cumulativeCounts for the search, as for four-byte entries)
                    idx->dev.win_flat = 1;
                    return;
                }
                (void)hipFree(d_flat);
            }
            (void)hipGetLastError();
        }
        // (does not fit, or an answer that fits nowhere: the cells' form decides)
    }
    const size_t cells = fmx::win_cells_for((uint32_t)idx->hdr.wt_size);
    const size_t bytes = cells * 64;
    if (mode == 2) {  // cells + (at worst) an entry per position must fit a quarter of what is free, and the absolute budget
        size_t free_b = 0, total_b = 0;
        const size_t worst = bytes + (size_t)idx->hdr.wt_size * fmx::kWinEntryWords * sizeof(uint16_t);
        if (worst > ((size_t)g_window_cells_mb.load() << 20) || hipMemGetInfo(&free_b, &total_b) != hipSuccess || worst > free_b / 4) {
            (void)hipGetLastError();
            return;
        }
    }
    void *d_cells = nullptr, *d_counts = nullptr, *d_entries = nullptr;
    auto give_up = [&]() {
        (void)hipGetLastError();
        if (d_cells) (void)hipFree(d_cells);
        if (d_counts) (void)hipFree(d_counts);
        if (d_entries) (void)hipFree(d_entries);
    };
    if (hipMalloc(&d_cells, bytes) != hipSuccess || hipMalloc(&d_counts, cells * sizeof(uint32_t)) != hipSuccess) return give_up();
    if (k_launch_win_build(idx, idx->dev, idx->n_cu, (uint32_t)cells, static_cast<fmx::Quad *>(d_cells), static_cast<uint32_t *>(d_counts),
                           nullptr) != 0)
        return give_up();
    // every window's first entry = the class-3 positions of the windows before it (a prefix sum on the host: 4 bytes per window)
    std::vector<uint32_t> first;
    try {
        first.resize(cells);
    } catch (const std::exception &) {
        return give_up();
    }
    if (hipMemcpy(first.data(), d_counts, cells * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return give_up();
    uint64_t total = 0;
    for (size_t w = 0; w < cells; ++w) {
        const uint32_t c = first[w];
        first[w] = (uint32_t)total;
        total += c;
    }
    if (total > 0xffffffffull) return give_up();
    // The entries — four bytes each where cumulativeCounts fit LDS (the row alone: the symbol is found by a search), six otherwise —
    // and behind them, 8-byte aligned, four words: how many carry a status or `suspect` (statistics), whether some step's answer
    // did not fit (no directory in that form then), how many of the four-byte form's slots are taken; then those slots (eight
    // bytes each: the few answers that are more than a row).  A four-byte directory that runs out of slots is made again with
    // six-byte entries.
    const int want = g_window_entry_bytes.load();
    bool entry4 = want == 4 || (want == 0 && idx->hdr.n_c <= fmx::kWinSymbolSearchMax);
    for (;;) {
        const size_t per_entry = entry4 ? 4 : fmx::kWinEntryWords * sizeof(uint16_t);
        const size_t entry_bytes = ((size_t)total * per_entry + 7) & ~(size_t)7;
        const uint32_t full_cap = entry4 ? (uint32_t)std::min<uint64_t>(4096 + total / 512, 0x7fffffffu) : 0u;
        uint32_t tail[4] = {0, 0, 0, 0};
        uint8_t *e8 = nullptr;
        if (hipMalloc(&d_entries, entry_bytes + 16 + (size_t)full_cap * 8) != hipSuccess) return give_up();
        e8 = static_cast<uint8_t *>(d_entries);
        if (hipMemset(e8 + entry_bytes, 0, 16) != hipSuccess ||
            hipMemcpy(d_counts, first.data(), cells * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
            k_launch_win_other(idx, idx->dev, idx->n_cu, (uint32_t)cells, static_cast<fmx::Quad *>(d_cells), static_cast<const uint32_t *>(d_counts),
                               static_cast<uint16_t *>(d_entries), reinterpret_cast<uint32_t *>(e8 + entry_bytes), entry4 ? 1 : 0,
                               reinterpret_cast<uint64_t *>(e8 + entry_bytes + 16), full_cap, nullptr) != 0 ||
            hipMemcpy(tail, e8 + entry_bytes, sizeof tail, hipMemcpyDeviceToHost) != hipSuccess)
            return give_up();
        if (tail[1] != 0) {
            if (!entry4) return give_up();
            (void)hipFree(d_entries);  // (out of slots, or an answer that fits neither form: the six-byte form decides)
            d_entries = nullptr;
            entry4 = false;
            continue;
        }
        (void)hipFree(d_counts);
        idx->win_unclean = tail[0];
        idx->d_win = d_cells;
        idx->d_win_other = d_entries;
        idx->win_bytes = bytes + (size_t)total * per_entry + (entry4 ? (size_t)tail[2] * 8 : 0);
        idx->dev.win = static_cast<const fmx::Quad *>(d_cells);
        idx->dev.win_other = static_cast<const uint16_t *>(d_entries);
        idx->dev.win_entry4 = entry4 ? 1 : 0;
        idx->dev.win_full = entry4 ? reinterpret_cast<const uint64_t *>(e8 + entry_bytes + 16) : nullptr;
        return;
    }
}

int fmx_window_cells_info(const fmx_index *idx, int64_t *bytes) {
    return guarded([&]() -> int {
    if (!idx) return fail(FMX_E_ARG, "null index");
    if (bytes) *bytes = (int64_t)idx->win_bytes;
    return FMX_OK;
    });
}

int fmx_resident_bytes(const fmx_index *idx, int64_t *image, int64_t *suffix_table, int64_t *window_directory) {
    return guarded([&]() -> int {
    if (!idx) return fail(FMX_E_ARG, "null index");
    const bool resident = idx->d_blob != nullptr;
    if (image) *image = resident ? (int64_t)idx->d_len : 0;
    if (suffix_table)
        *suffix_table = resident ? (int64_t)idx->suffix_table_bytes +
                                       (idx->d_suffix_order1 ? (int64_t)idx->hdr.wt_sigma * idx->hdr.wt_sigma * 2 * (int64_t)sizeof(float) : 0)
                                 : 0;
    if (window_directory) *window_directory = resident ? (int64_t)idx->win_bytes : 0;
    return FMX_OK;
    });
}

int fmx_device_of(const fmx_index *idx) { return (idx && idx->d_blob) ? idx->device : -1; }

// The suffix table of a resident FM-index (fmx_device.hpp): grown level by level on the device — the strings of 2, 3, ... codes
// that occur in the text, each with its SA interval — and ALL levels hashed into one table of 16-byte slots (a pattern shorter
// than the deepest level still finds its whole search there).  How deep: option `suffix_table_chars` (default 8; at most what
// a 64-bit key holds: 8 codes of 8 bits, 4 of 16), cut where the table would pass its size limit = the smaller of the budget
// `suffix_table_mb` and 1 / `suffix_table_image_fraction` of the image (default an eighth).  Not having one is never an error.
static void build_suffix_table(fmx_index *idx) {
    if (idx->d_suffix_table) {
        (void)hipFree(idx->d_suffix_table);
        idx->d_suffix_table = nullptr;
        idx->suffix_table_bytes = 0;
    }
    if (idx->d_suffix_order1) {
        (void)hipFree(idx->d_suffix_order1);
        idx->d_suffix_order1 = nullptr;
    }
    idx->suffix_table_strings = idx->suffix_table_deepest = 0;
    idx->dev.suffix_table = nullptr;
    idx->dev.suffix_order1 = nullptr;
    idx->dev.suffix_chars = 0;
    idx->dev.suffix_key_bits = fmx::fmx_code_bits_for(idx->hdr.wt_sigma, g_code_bits_12_api.load() != 0);
    idx->dev.suffix_shift = 0;
    idx->dev.suffix_mask = 0;
    if (idx->rrr_only || idx->wavelet_only || idx->hdr.kind != 0) return;
    const uint64_t budget = (uint64_t)g_suffix_table_mb.load() << 20;
    const int key_bits = idx->dev.suffix_key_bits;
    int max_chars = g_suffix_table_chars.load();
    if (max_chars > 64 / key_bits) max_chars = 64 / key_bits;
    if (budget == 0 || max_chars < 2 || idx->hdr.wt_sigma < 2) return;
    uint64_t limit = budget;
    if (const int frac = g_suffix_table_image_fraction.load(); frac > 0)
        limit = std::min<uint64_t>(limit, std::max<uint64_t>(idx->d_len / (uint64_t)frac, 64 << 10));
    // slots for a set of strings: a power of two, half full at most — or, where only that keeps the table inside its limit, 0.7
    auto slots_for = [&](uint64_t strings) {
        uint64_t slots = 1024;
        while (slots < 2 * strings) slots <<= 1;
        if (slots * sizeof(fmx::SuffixSlot) > limit && slots / 2 >= 1024 && 10 * strings <= 7 * (slots / 2)) slots >>= 1;
        return slots;
    };
    // strings of all levels together: never more than the limit carries at 0.7 load, nor 2^30
    uint64_t cap64 = limit / sizeof(fmx::SuffixSlot) * 7 / 10;
    if (cap64 < 1024) cap64 = 1024;
    if (cap64 > 0x3fffffffu) cap64 = 0x3fffffffu;
    const uint32_t cap = (uint32_t)cap64;
    const uint32_t cap1 = (uint32_t)idx->hdr.wt_sigma;  // level 1: the characters (never inserted: cumulativeCounts has them)
    fmx::SuffixSlot *level1 = nullptr, *all = nullptr;
    uint32_t *d_count = nullptr;
    void *d_slots = nullptr;
    auto cleanup = [&]() {
        (void)hipGetLastError();
        if (level1) (void)hipFree(level1);
        if (all) (void)hipFree(all);
        if (d_count) (void)hipFree(d_count);
    };
    if (hipMalloc(reinterpret_cast<void **>(&level1), (size_t)cap1 * sizeof(fmx::SuffixSlot)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&all), (size_t)cap * sizeof(fmx::SuffixSlot)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&d_count), 64) != hipSuccess) {
        cleanup();
        return;
    }
    auto counted = [&](uint32_t *out) {  // the level's size, once its kernel has finished
        return hipMemcpy(out, d_count, 4, hipMemcpyDeviceToHost) == hipSuccess;
    };
    uint32_t n_prev = 0;
    if (hipMemset(d_count, 0, 64) != hipSuccess || k_launch_suffix_level1(idx, idx->dev, level1, d_count, cap1, nullptr) != 0 ||
        !counted(&n_prev) || n_prev == 0 || n_prev > cap1) {
        cleanup();
        return;
    }
    // work of one level = strings x (sigma - 1) rank pairs, most of them of absent characters: bounded, so that a large
    // alphabet (sigma up to 32,767) does not spend tens of seconds on a level that is thrown away in the end
    constexpr uint64_t kLevelWorkMax = 1ull << 35;
    uint32_t begin[10] = {0};  // strings of `len` codes: all[begin[len] .. begin[len + 1])
    const fmx::SuffixSlot *prev = level1;
    uint32_t total = 0;
    int chars = 1;
    while (chars < max_chars) {
        if ((uint64_t)n_prev * (uint64_t)(idx->hdr.wt_sigma - 1) > kLevelWorkMax) break;
        const uint32_t room = cap - total;
        uint32_t n_next = 0;
        if (room == 0) break;
        if (hipMemset(d_count, 0, 64) != hipSuccess ||
            k_launch_suffix_expand(idx, idx->dev, idx->n_cu, prev, n_prev, chars, key_bits, all + total, d_count, room, nullptr) != 0 ||
            !counted(&n_next)) {
            cleanup();
            return;
        }
        // (the next level does not fit, or its table would pass the limit: the levels so far are the table)
        if (n_next == 0 || n_next > room || slots_for((uint64_t)total + n_next) * sizeof(fmx::SuffixSlot) > limit) break;
        ++chars;
        begin[chars] = total;
        prev = all + total;
        n_prev = n_next;
        total += n_next;
        begin[chars + 1] = total;
    }
    if (chars < 2) {
        cleanup();
        return;
    }
    const uint64_t slots64 = slots_for(total);
    if (slots64 * sizeof(fmx::SuffixSlot) > std::max<uint64_t>(limit, 16 << 10) || slots64 > 0x40000000u) {
        cleanup();
        return;
    }
    const uint32_t slots = (uint32_t)slots64;
    int log2_slots = 0;
    while ((1u << log2_slots) < slots) ++log2_slots;
    fmx::DevIndex geometry = idx->dev;  // what fm_suffix_home needs
    geometry.suffix_chars = chars;
    geometry.suffix_shift = (uint32_t)(64 - (log2_slots - fmx::kSuffixGroupLog2));  // whole groups
    geometry.suffix_mask = slots - 1;
    bool ok = hipMalloc(&d_slots, (size_t)slots * sizeof(fmx::SuffixSlot)) == hipSuccess &&
              hipMemset(d_slots, 0xff, (size_t)slots * sizeof(fmx::SuffixSlot)) == hipSuccess;
    for (int len = 2; ok && len <= chars; ++len)
        ok = k_launch_suffix_insert(idx, geometry, all + begin[len], begin[len + 1] - begin[len], len,
                                       static_cast<fmx::SuffixSlot *>(d_slots), nullptr) == 0;
    if (!ok || hipStreamSynchronize(nullptr) != hipSuccess) {
        if (d_slots) (void)hipFree(d_slots);
        cleanup();
        return;
    }
    cleanup();
    idx->d_suffix_table = d_slots;
    idx->suffix_table_bytes = (size_t)slots * sizeof(fmx::SuffixSlot);
    idx->suffix_table_strings = total;
    idx->suffix_table_deepest = begin[chars + 1] - begin[chars];
    idx->dev.suffix_table = static_cast<const fmx::SuffixSlot *>(d_slots);
    idx->dev.suffix_chars = chars;
    idx->dev.suffix_shift = geometry.suffix_shift;
    idx->dev.suffix_mask = slots - 1;
    // order-1 statistics of the two-character strings for the plan's sort key (small alphabets; not having them is no error)
    if (key_bits == 8 && idx->hdr.wt_sigma <= fmx::kOrder1MaxSigma &&
        fmx::order1_lds_bytes(fmx::kPlanDefaultBins, idx->hdr.wt_sigma) <= fmx::kPlanCodesLdsMax) {
        void *d_o1 = nullptr;
        const size_t bytes = (size_t)idx->hdr.wt_sigma * idx->hdr.wt_sigma * 2 * sizeof(float);
        if (hipMalloc(&d_o1, bytes) == hipSuccess) {
            if (k_launch_suffix_order1(idx, idx->dev, static_cast<float *>(d_o1), nullptr) == 0 &&
                hipStreamSynchronize(nullptr) == hipSuccess) {
                idx->d_suffix_order1 = d_o1;
                idx->dev.suffix_order1 = static_cast<const float *>(d_o1);
            } else {
                (void)hipFree(d_o1);
            }
        }
    }
}

int fmx_suffix_table_info(const fmx_index *idx, int32_t *chars, int64_t *bytes) {
    return guarded([&]() -> int {
    if (!idx) return fail(FMX_E_ARG, "null index");
    if (chars) *chars = idx->dev.suffix_table ? idx->dev.suffix_chars : 0;
    if (bytes) *bytes = (int64_t)idx->suffix_table_bytes;
    return FMX_OK;
    });
}

int fmx_blob(const fmx_index *idx_c, const uint8_t **blob, size_t *len) {
    return guarded([&]() -> int {
    fmx_index *idx = const_cast<fmx_index *>(idx_c);
    if (!idx || !blob || !len) return fail(FMX_E_ARG, "null argument");
    int rc = ensure_blob(idx);
    if (rc) return rc;
    *blob = idx->blob.data();
    *len = idx->blob.size();
    return FMX_OK;
    });
}

int fmx_to_device(fmx_index *idx, int device) {
    return guarded([&]() -> int {
    if (!idx) return fail(FMX_E_ARG, "null index");
    int rc = ensure_blob(idx);
    if (rc) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(FMX_E_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(FMX_E_ARG, "device ordinal out of range");
    // a handle that is already resident — here or on another GPU — first gives back what it holds THERE: the image, the suffix
    // table, the window directory, DevIndex.self (the cold routes' resident copy) and the per-stream scratch are addresses in that
    // device's memory.  (An attached image stays the caller's; the handle merely lets go of it.)
    if (idx->device >= 0 || idx->d_blob) {
        (void)hipDeviceSynchronize();
        release_device_state(idx);
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(&idx->d_blob, idx->blob.size()));
    idx->owns_device = true;
    idx->d_len = idx->blob.size();
    idx->device = device;
    HIP_TRY(hipMemcpy(idx->d_blob, idx->blob.data(), idx->blob.size(), hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    idx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    make_dev_index(idx);
    if (int rc2 = publish_dev_index(idx)) return rc2;
    build_window_cells(idx);
    build_suffix_table(idx);
    return FMX_OK;
    });
}

int fmx_attach_device_blob(void *device_blob, size_t len, int device, fmx_index **out) {
    return guarded([&]() -> int {
    if (!device_blob || !out || len < sizeof(fmx::BlobHeader)) return fail(FMX_E_ARG, "bad blob");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(FMX_E_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<fmx_index> idx(new fmx_index());
    HIP_TRY(hipMemcpy(&idx->hdr, device_blob, sizeof(fmx::BlobHeader), hipMemcpyDeviceToHost));
    if (idx->hdr.magic != fmx::kBlobMagic || idx->hdr.version != fmx::kBlobVersion || idx->hdr.total_bytes != len)
        return fail(FMX_E_FORMAT, "device blob header mismatch");
    {
        // the image arrived from somewhere else (an RCCL broadcast): validate it on a host copy before any kernel
        // may walk it — sections, per-superblock tables, block headers, mapping entries, and the body checksum
        // (an uninitialised buffer: a zero-filled vector would touch every page of a GB-sized image once more per rank)
        std::unique_ptr<uint8_t[]> copy(new (std::nothrow) uint8_t[len]);
        if (!copy) return fail(FMX_E_NOMEM, "out of memory");
        HIP_TRY(hipMemcpy(copy.get(), device_blob, len, hipMemcpyDeviceToHost));
        std::string err;
        if (fmx::validate_blob(copy.get(), len, err)) return fail(FMX_E_FORMAT, err);
    }
    idx->rrr_only = idx->hdr.kind == 1;
    idx->d_blob = device_blob;
    idx->d_len = len;
    idx->device = device;
    idx->owns_device = false;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    idx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    make_dev_index(idx.get());
    if (int rc2 = publish_dev_index(idx.get())) return rc2;
    build_window_cells(idx.get());
    build_suffix_table(idx.get());
    *out = idx.release();
    return FMX_OK;
    });
}

// One immutable index on several GPUs of the node (fmx.h "replicas").  Every destination is served by a host thread of its own:
// allocation, the copy of the image — a peer copy out of the source's HBM where `src` is resident (the destinations pull at once:
// the source's egress goes over all its xGMI links, no ring), from the host image otherwise — and the growth of the replica's
// own window directory and suffix table.  The image needs no validation here: it is this process's own (an image from a caller's
// bytes was validated when it was flattened or attached).
int fmx_replicate(const fmx_index *src_c, const int32_t *devices, int32_t n_devices, fmx_index **out) {
    return guarded([&]() -> int {
    fmx_index *src = const_cast<fmx_index *>(src_c);
    if (!src || !devices || !out || n_devices < 1) return fail(FMX_E_ARG, "bad arguments");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(FMX_E_NO_DEVICE, "no HIP device visible");
    for (int32_t i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= n_dev) return fail(FMX_E_ARG, "device ordinal out of range");
    const bool from_device = src->d_blob != nullptr;
    if (!from_device) {
        int rc = ensure_blob(src);
        if (rc) return rc;
    }
    const size_t len = from_device ? src->d_len : src->blob.size();
    int caller_device = 0;
    (void)hipGetDevice(&caller_device);
    if (from_device) {  // whatever is still being written into the source image's neighbourhood is none of ours; its copy-in is done
        HIP_TRY(hipSetDevice(src->device));
        HIP_TRY(hipDeviceSynchronize());
    }
    std::vector<std::unique_ptr<fmx_index>> made((size_t)n_devices);
    std::vector<int> rcs((size_t)n_devices, FMX_OK);
    std::vector<std::string> errs((size_t)n_devices);
    auto make_one = [&](int32_t i) {
        auto failed = [&](int rc) {
            rcs[(size_t)i] = rc;
            errs[(size_t)i] = g_err;  // (this thread's message: handed to the caller's thread below)
        };
        std::unique_ptr<fmx_index> idx(new (std::nothrow) fmx_index());
        if (!idx) return failed(fail(FMX_E_NOMEM, "out of memory"));
        const int device = devices[i];
        auto hip_ok = [&](hipError_t e, const char *what) {
            if (e == hipSuccess) return true;
            failed(fail(FMX_E_HIP, std::string(what) + ": " + hipGetErrorString(e)));
            return false;
        };
        if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return;
        idx->hdr = src->hdr;
        idx->rrr_only = src->rrr_only;
        idx->wavelet_only = src->wavelet_only;
        idx->device = device;
        if (!hip_ok(hipMalloc(&idx->d_blob, len), "hipMalloc (replica image)")) {
            idx->device = -1;
            return;
        }
        idx->owns_device = true;
        idx->d_len = len;
        hipError_t e;
        if (!from_device)
            e = hipMemcpy(idx->d_blob, src->blob.data(), len, hipMemcpyHostToDevice);
        else if (device == src->device)
            e = hipMemcpy(idx->d_blob, src->d_blob, len, hipMemcpyDeviceToDevice);
        else
            e = hipMemcpyPeer(idx->d_blob, device, src->d_blob, src->device, len);
        if (!hip_ok(e, "copy of the image to the replica's device")) {
            release_device_state(idx.get());
            return;
        }
        hipDeviceProp_t prop;
        if (!hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) {
            release_device_state(idx.get());
            return;
        }
        idx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        make_dev_index(idx.get());
        if (int rc = publish_dev_index(idx.get())) {
            failed(rc);
            release_device_state(idx.get());
            return;
        }
        build_window_cells(idx.get());
        build_suffix_table(idx.get());
        made[(size_t)i] = std::move(idx);
    };
    {
        std::vector<std::thread> threads;
        for (int32_t i = 1; i < n_devices; ++i) threads.emplace_back(make_one, i);
        make_one(0);
        for (std::thread &t : threads) t.join();
    }
    (void)hipSetDevice(caller_device);
    for (int32_t i = 0; i < n_devices; ++i)
        if (rcs[(size_t)i] != FMX_OK || !made[(size_t)i]) {
            for (auto &m : made)
                if (m) release_device_state(m.get());
            (void)hipSetDevice(caller_device);
            return fail(rcs[(size_t)i] ? rcs[(size_t)i] : FMX_E_HIP,
                        "replica on device " + std::to_string(devices[i]) + ": " + errs[(size_t)i]);
        }
    for (int32_t i = 0; i < n_devices; ++i) out[i] = made[(size_t)i].release();
    return FMX_OK;
    });
}

// Pins a long-lived host buffer of the caller (a direct ByteBuffer of the Java shim, a numpy array) so that the
// host-buffer entry points move it by DMA without staging copies.
int fmx_host_register(void *p, size_t bytes) {
    return guarded([&]() -> int {
    if (!p || bytes == 0) return fail(FMX_E_ARG, "bad arguments");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    std::lock_guard<std::mutex> lock(g_registered_mutex);
    g_registered[reinterpret_cast<uintptr_t>(p)] = bytes;
    return FMX_OK;
    });
}
int fmx_host_unregister(void *p) {
    return guarded([&]() -> int {
    if (!p) return fail(FMX_E_ARG, "bad arguments");
    {
        std::lock_guard<std::mutex> lock(g_registered_mutex);
        g_registered.erase(reinterpret_cast<uintptr_t>(p));
    }
    HIP_TRY(hipHostUnregister(p));
    return FMX_OK;
    });
}

void *fmx_device_blob(const fmx_index *idx, size_t *len) {
    if (!idx) return nullptr;
    if (len) *len = idx->d_len;
    return idx->d_blob;
}

// ---- device-pointer entry points ---------------------------------------------------------------
// Every entry point is a thin wrapper over an *_impl that takes its scratch from a Scratch provider: the
// index's per-stream buffers here, per-call blocks for the host-buffer forms further down.

// stage 1 of count/locate: processing order of the batch (suffix-key radix sort) — nullptr when the
// batch is too small to be worth sorting
// would a batch of n patterns be planned when the library decides (count / locate / segment entry points)?
//  * an index with a suffix table, plan ordered by SA row (kernels' option plan_sa_key, the default): the bucket pass costs
//    ~45 us per 1 M patterns and k_count then runs 0.141 -> 0.089 ms — it pays from about 0.8 M patterns on (option
//    "plan_sa_min", default 786,432; below: 524,288 patterns 0.087 planned vs 0.079 in the caller's order);
//  * plan ordered by the trailing characters' codes (no table, or plan_sa_key 0): it pays while MANY patterns share the table
//    string they start from — at least "plan_min_per_string" (16) patterns per string of the table's deepest level; always
//    without a table.
static bool plan_pays(const fmx_index *idx, int32_t n) {
    const bool table = idx->dev.suffix_table && idx->suffix_table_deepest != 0 && g_suffix_table_in_use.load();
    if (!table) return true;
    if (g_plan_sa_key_api.load() != 0) return n >= g_plan_sa_min.load();
    const int per_string = g_plan_min_per_string.load();
    if (per_string <= 0) return true;
    return (uint64_t)n >= (uint64_t)per_string * idx->suffix_table_deepest;
}

// explicit: the caller asked for a plan (fmx_count_plan_dev) — it gets one whatever plan_pays says
static int plan_order(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n, Scratch &scratch,
                      fmx::CountPlan *plan, bool explicit_request = false) {
    *plan = fmx::CountPlan();
    if (!scratch.per_call) {  // whatever fmx_count_plan_dev left for this stream is about to be overwritten
        std::lock_guard<std::mutex> lock(idx->ws_mutex);
        idx->plans.erase(scratch.stream);
    }
    if (!explicit_request && !plan_pays(idx, n)) return FMX_OK;  // the caller's order: k_count maps the characters itself
    void *ws = nullptr;
    const size_t ws_bytes = k_count_workspace_bytes(idx, idx->dev, n);
    int rc = scratch.get(kWsPlan, ws_bytes, &ws);
    if (rc) return rc;
    if (!ws) return FMX_OK;
    // a per-stream workspace keeps its head zeroed between plans; a per-call block comes from the cache: clear it
    int e = k_launch_count_plan(idx, idx->dev, idx->n_cu, d_pat, d_pat_off, n, ws, ws_bytes, !scratch.per_call, plan,
                                   static_cast<hipStream_t>(scratch.stream));
    if (e) {
        // A plan that stopped half way (k_plan_codes ran, k_plan_scatter did not) leaves its histogram in the workspace's
        // head, which a per-stream workspace is trusted to hold zeroed between plans: clear it, and forget the stream's plan
        if (!scratch.per_call) {
            (void)hipMemsetAsync(ws, 0, fmx::kPlanHeadBytes, static_cast<hipStream_t>(scratch.stream));
            std::lock_guard<std::mutex> lock(idx->ws_mutex);
            idx->plans.erase(scratch.stream);
        }
        *plan = fmx::CountPlan();
        return fail(FMX_E_HIP, std::string("plan stage: ") + hipGetErrorString((hipError_t)e));
    }
    return FMX_OK;
}

int fmx_count_plan_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                       const void **d_plan, void *stream) {
    return guarded([&]() -> int {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || !d_plan || (n > 0 && !d_pat_off)) return fail(FMX_E_ARG, "bad arguments");
    fmx_index::Plan plan;
    Scratch scratch(idx, stream, false);
    rc = plan_order(idx, d_pat, d_pat_off, n, scratch, &plan.plan, true);
    if (rc) return rc;
    plan.pat = d_pat;
    plan.pat_off = d_pat_off;
    {
        std::lock_guard<std::mutex> lock(idx->ws_mutex);
        idx->plans[stream] = plan;
    }
    *d_plan = plan.plan.recs;
    return FMX_OK;
    });
}

// what the library decides for a batch of n queries (tests/test_gpu_policy.py pins the thresholds):
// kind 0 = count(): is the batch planned (fmx_count_batch_is_planned); 1 = locate(): are the hits walked by the first row of
// their ranges (walk_order_min); 2 = extractUntilBoundary: are the queries taken by text position (boundary_order_min)
int fmx_batch_policy(const fmx_index *idx, int kind, int64_t n) {
    if (!idx || n <= 0 || n > INT32_MAX) return 0;
    if (kind == 0) return fmx_count_batch_is_planned(idx, (int32_t)n);
    if (kind == 1) return k_walk_workspace_bytes(idx, idx->dev, (int32_t)n) != 0 ? 1 : 0;
    if (kind == 2) return k_boundary_order_bytes(idx, idx->dev, n) != 0 ? 1 : 0;
    return 0;
}

int fmx_count_batch_is_planned(const fmx_index *idx, int32_t n) {
    if (!idx || n <= 0) return 0;
    return (k_count_workspace_bytes(idx, idx->dev, n) != 0 && plan_pays(idx, n)) ? 1 : 0;
}

int fmx_count_ordered_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, const void *d_plan,
                          int32_t n, int32_t *d_counts, int32_t *d_lf_steps, int32_t *d_status, void *stream) {
    return guarded([&]() -> int {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!d_pat_off || !d_counts))) return fail(FMX_E_ARG, "bad arguments");
    fmx::CountPlan plan;  // only the stream's live plan for these very patterns is honoured; anything else
    if (d_plan) {         // (stale, foreign) means the caller's order
        std::lock_guard<std::mutex> lock(idx->ws_mutex);
        auto it = idx->plans.find(stream);
        if (it != idx->plans.end() && it->second.plan.recs == d_plan && it->second.plan.n == n && it->second.pat == d_pat &&
            it->second.pat_off == d_pat_off)
            plan = it->second.plan;
    }
    int e = k_launch_count(idx, idx->dev, idx->n_cu, d_pat, d_pat_off, &plan, false, n, d_counts, d_lf_steps, d_status,
                              nullptr, static_cast<hipStream_t>(stream));
    if (e) return fail(FMX_E_HIP, std::string("k_count launch: ") + hipGetErrorString((hipError_t)e));
    return FMX_OK;
    });
}

// allow_plan = false: the batch is counted in the caller's order whatever the policy says (the mapped host-buffer path: a plan
// would read the characters over PCIe in one kernel and leave the counting to the next — nothing would overlap)
static int count_impl(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n, int32_t *d_counts,
                      int32_t *d_lf_steps, int32_t *d_status, Scratch &scratch, bool allow_plan = true) {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!d_pat_off || !d_counts))) return fail(FMX_E_ARG, "bad arguments");
    fmx::CountPlan plan;
    if (allow_plan) {
        rc = plan_order(idx, d_pat, d_pat_off, n, scratch, &plan);
        if (rc) return rc;
    } else if (!scratch.per_call) {
        std::lock_guard<std::mutex> lock(idx->ws_mutex);
        idx->plans.erase(scratch.stream);
    }
    int e = k_launch_count(idx, idx->dev, idx->n_cu, d_pat, d_pat_off, &plan, false, n, d_counts, d_lf_steps, d_status,
                              nullptr, static_cast<hipStream_t>(scratch.stream));
    if (e) return fail(FMX_E_HIP, std::string("k_count launch: ") + hipGetErrorString((hipError_t)e));
    return FMX_OK;
}

int fmx_count_batch_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                        int32_t *d_counts, int32_t *d_lf_steps, int32_t *d_status, void *stream) {
    return guarded([&]() -> int {
    Scratch scratch(idx, stream, false);
    return count_impl(idx, d_pat, d_pat_off, n, d_counts, d_lf_steps, d_status, scratch);
    });
}

// k_locate_walk over the ranges of a batch, the patterns taken by the first row of their ranges where the batch is large
// enough for that to pay (walk_workspace_bytes)
static int walk_hits(const fmx_index *idx, const int32_t *d_range, int32_t n, int32_t max_matches, int32_t *d_locs,
                     int32_t loc_cap, int32_t *d_found, int32_t *d_lf_steps, int32_t *d_status, const int32_t *d_taken,
                     Scratch &scratch) {
    hipStream_t st = static_cast<hipStream_t>(scratch.stream);
    void *ws = nullptr;
    const size_t ws_bytes = k_walk_workspace_bytes(idx, idx->dev, n);
    int rc = scratch.get(kWsWalk, ws_bytes, &ws);
    if (rc) return rc;
    int e = k_launch_locate_walk(idx, idx->dev, idx->n_cu, d_range, n, max_matches, d_locs, loc_cap, d_found, d_lf_steps,
                                 d_status, d_taken, ws, ws_bytes, !scratch.per_call, st, nullptr, 0);
    if (e) {
        // (an order that stopped half way leaves its histogram in the head of a per-stream workspace: clear it)
        if (ws && !scratch.per_call) (void)hipMemsetAsync(ws, 0, fmx::kPlanHeadBytes, st);
        return fail(FMX_E_HIP, std::string("k_locate_walk launch: ") + hipGetErrorString((hipError_t)e));
    }
    return FMX_OK;
}

static int locate_impl(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                       int32_t max_matches, int32_t *d_locs, int32_t loc_cap, int32_t *d_found, int32_t *d_lf_steps,
                       int32_t *d_status, int32_t *d_range_ws, Scratch &scratch) {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || loc_cap < 0 || (n > 0 && (!d_pat_off || !d_found || !d_range_ws || (!d_locs && loc_cap > 0))))
        return fail(FMX_E_ARG, "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(scratch.stream);
    // found[] doubles as the scratch `counts` output of the range pass; the walk pass overwrites it
    fmx::CountPlan plan;
    rc = plan_order(idx, d_pat, d_pat_off, n, scratch, &plan);
    if (rc) return rc;
    int e = k_launch_count(idx, idx->dev, idx->n_cu, d_pat, d_pat_off, &plan, false, n, d_found, d_lf_steps, d_status,
                              d_range_ws, st);
    if (e) return fail(FMX_E_HIP, std::string("k_count launch: ") + hipGetErrorString((hipError_t)e));
    return walk_hits(idx, d_range_ws, n, max_matches, d_locs, loc_cap, d_found, d_lf_steps, d_status, nullptr, scratch);
}

int fmx_locate_batch_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                         int32_t max_matches, int32_t *d_locs, int32_t loc_cap, int32_t *d_found,
                         int32_t *d_lf_steps, int32_t *d_status, int32_t *d_range_ws, void *stream) {
    return guarded([&]() -> int {
    Scratch scratch(idx, stream, false);
    return locate_impl(idx, d_pat, d_pat_off, n, max_matches, d_locs, loc_cap, d_found, d_lf_steps, d_status, d_range_ws,
                       scratch);
    });
}

int fmx_extract_batch_dev(const fmx_index *idx, const int32_t *d_start, const int32_t *d_stop, int32_t n,
                          uint16_t *d_dst, int32_t dst_len, int32_t offset, int32_t *d_out_len, int32_t *d_lf_steps,
                          int32_t *d_status, void *stream) {
    return guarded([&]() -> int {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || dst_len < 0 || (n > 0 && (!d_start || !d_stop || !d_out_len || (!d_dst && dst_len > 0))))
        return fail(FMX_E_ARG, "bad arguments");
    int e = k_launch_extract(idx, idx->dev, idx->n_cu, d_start, d_stop, n, d_dst, dst_len, offset, d_out_len, d_lf_steps,
                                d_status, nullptr, 0, 0, nullptr, 0, true, static_cast<hipStream_t>(stream));
    if (e) return fail(FMX_E_HIP, std::string("k_extract launch: ") + hipGetErrorString((hipError_t)e));
    return FMX_OK;
    });
}

static int boundary_impl(const fmx_index *idx, const int32_t *d_from, int64_t n, uint16_t boundary, int mode,
                         uint16_t *d_dst, int32_t dst_len, int32_t offset, int32_t *d_out_len, int32_t *d_lf_steps,
                         int32_t *d_status, int32_t *d_aux, const int32_t *slot_found, int32_t slots, Scratch &scratch) {
    void *ws = nullptr;
    // two windows of sampleRate characters per lane; an index sampled so sparsely that they would take more than 8 GiB is
    // served by the literal form (no windows) instead of failing on the allocation
    size_t ws_bytes = k_boundary_workspace_bytes(idx, idx->dev, n, idx->n_cu);
    if (ws_bytes > ((size_t)8 << 30)) ws_bytes = 0;
    int rc = scratch.get(kWsBoundary, ws_bytes, &ws);
    if (rc) return rc;
    // large batches take their queries by text position (the walk order's workspace: the locate stage of a pipeline is done with it)
    void *order_ws = nullptr;
    const size_t order_bytes = k_boundary_order_bytes(idx, idx->dev, n);
    rc = scratch.get(kWsWalk, order_bytes, &order_ws);
    if (rc) return rc;
    int e = k_launch_extract_boundary(idx, idx->dev, idx->n_cu, d_from, n, boundary, mode, d_dst, dst_len, offset,
                                         d_out_len, d_lf_steps, d_status, d_aux, ws, ws_bytes, slot_found, slots, order_ws,
                                         order_bytes, !scratch.per_call, static_cast<hipStream_t>(scratch.stream));
    if (e) {
        if (order_ws && !scratch.per_call)
            (void)hipMemsetAsync(order_ws, 0, fmx::kPlanHeadBytes, static_cast<hipStream_t>(scratch.stream));
        return fail(FMX_E_HIP, std::string("k_extract_boundary launch: ") + hipGetErrorString((hipError_t)e));
    }
    return FMX_OK;
}

int fmx_extract_boundary_batch_dev(const fmx_index *idx, const int32_t *d_from, int32_t n, uint16_t boundary, int mode,
                                   uint16_t *d_dst, int32_t dst_len, int32_t offset, int32_t *d_out_len,
                                   int32_t *d_lf_steps, int32_t *d_status, int32_t *d_aux, void *stream) {
    return guarded([&]() -> int {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || dst_len < 0 || mode < 0 || mode > 2 || (n > 0 && (!d_from || !d_out_len || (!d_dst && dst_len > 0))))
        return fail(FMX_E_ARG, "bad arguments");
    Scratch scratch(idx, stream, false);
    return boundary_impl(idx, d_from, n, boundary, mode, d_dst, dst_len, offset, d_out_len, d_lf_steps, d_status, d_aux,
                         nullptr, 0, scratch);
    });
}

// ---- locate -> extract pipelines: the hit positions stay in HBM between the two stages ----

static int pipeline_args_ok(int32_t n, int32_t max_matches, int32_t row_len, const void *a, const void *b, const void *c,
                            const void *d, const void *e) {
    return n >= 0 && max_matches >= 1 && row_len >= 0 && (int64_t)n * max_matches <= INT32_MAX &&
           (n == 0 || (a && b && c && d && e));
}

static int locate_extract_impl(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                               int32_t max_matches, int32_t extract_len, int32_t *d_locs, int32_t *d_found,
                               uint16_t *d_dst, int32_t *d_out_len, int32_t *d_lf_steps, int32_t *d_status,
                               int32_t *d_hit_status, int32_t *d_range_ws, Scratch &scratch) {
    int rc = require_device(idx);
    if (rc) return rc;
    if (!pipeline_args_ok(n, max_matches, extract_len, d_pat_off, d_locs, d_found, d_out_len, d_range_ws) ||
        (n > 0 && extract_len > 0 && !d_dst))
        return fail(FMX_E_ARG, "bad arguments");
    rc = locate_impl(idx, d_pat, d_pat_off, n, max_matches, d_locs, max_matches, d_found, d_lf_steps, d_status, d_range_ws,
                     scratch);
    if (rc) return rc;
    // (the hits by text position: hits of equal patterns are equal positions; the walk order's workspace is free again)
    void *order_ws = nullptr;
    const size_t order_bytes = k_boundary_order_bytes(idx, idx->dev, (int64_t)n * max_matches);
    rc = scratch.get(kWsWalk, order_bytes, &order_ws);
    if (rc) return rc;
    int e = k_launch_extract(idx, idx->dev, idx->n_cu, d_locs, nullptr, (int64_t)n * max_matches, d_dst, extract_len, 0,
                                d_out_len, nullptr, d_hit_status, d_found, max_matches, extract_len, order_ws, order_bytes,
                                !scratch.per_call, static_cast<hipStream_t>(scratch.stream));
    if (e) {
        if (order_ws && !scratch.per_call)
            (void)hipMemsetAsync(order_ws, 0, fmx::kPlanHeadBytes, static_cast<hipStream_t>(scratch.stream));
        return fail(FMX_E_HIP, std::string("k_extract launch: ") + hipGetErrorString((hipError_t)e));
    }
    return FMX_OK;
}

int fmx_locate_extract_batch_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                                 int32_t max_matches, int32_t extract_len, int32_t *d_locs, int32_t *d_found,
                                 uint16_t *d_dst, int32_t *d_out_len, int32_t *d_lf_steps, int32_t *d_status,
                                 int32_t *d_hit_status, int32_t *d_range_ws, void *stream) {
    return guarded([&]() -> int {
    Scratch scratch(idx, stream, false);
    return locate_extract_impl(idx, d_pat, d_pat_off, n, max_matches, extract_len, d_locs, d_found, d_dst, d_out_len,
                               d_lf_steps, d_status, d_hit_status, d_range_ws, scratch);
    });
}

static int locate_lines_impl(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                             int32_t max_matches, uint16_t boundary, int mode, int32_t dst_len, int32_t *d_locs,
                             int32_t *d_found, uint16_t *d_dst, int32_t *d_out_len, int32_t *d_lf_steps,
                             int32_t *d_status, int32_t *d_hit_status, int32_t *d_hit_aux, int32_t *d_range_ws,
                             Scratch &scratch) {
    int rc = require_device(idx);
    if (rc) return rc;
    if (!pipeline_args_ok(n, max_matches, dst_len, d_pat_off, d_locs, d_found, d_out_len, d_range_ws) || mode < 0 ||
        mode > 2 || (n > 0 && dst_len > 0 && !d_dst))
        return fail(FMX_E_ARG, "bad arguments");
    rc = locate_impl(idx, d_pat, d_pat_off, n, max_matches, d_locs, max_matches, d_found, d_lf_steps, d_status, d_range_ws,
                     scratch);
    if (rc) return rc;
    return boundary_impl(idx, d_locs, (int64_t)n * max_matches, boundary, mode, d_dst, dst_len, 0, d_out_len, nullptr,
                         d_hit_status, d_hit_aux, d_found, max_matches, scratch);
}

int fmx_locate_lines_batch_dev(const fmx_index *idx, const uint16_t *d_pat, const int32_t *d_pat_off, int32_t n,
                               int32_t max_matches, uint16_t boundary, int mode, int32_t dst_len, int32_t *d_locs,
                               int32_t *d_found, uint16_t *d_dst, int32_t *d_out_len, int32_t *d_lf_steps,
                               int32_t *d_status, int32_t *d_hit_status, int32_t *d_hit_aux, int32_t *d_range_ws,
                               void *stream) {
    return guarded([&]() -> int {
    Scratch scratch(idx, stream, false);
    return locate_lines_impl(idx, d_pat, d_pat_off, n, max_matches, boundary, mode, dst_len, d_locs, d_found, d_dst,
                             d_out_len, d_lf_steps, d_status, d_hit_status, d_hit_aux, d_range_ws, scratch);
    });
}

// ---- segment sets ---------------------------------------------------------------------------------

static int segments_ok(const fmx_index *const *segs, int32_t n_segs) {
    if (!segs || n_segs < 1) return fail(FMX_E_ARG, "no segments");
    for (int32_t s = 0; s < n_segs; ++s) {
        int rc = require_device(segs[s]);
        if (rc) return rc;
        if (segs[s]->device != segs[0]->device) return fail(FMX_E_ARG, "segments live on different devices");
    }
    return FMX_OK;
}

static int count_segments_impl(const fmx_index *const *segs, int32_t n_segs, const uint16_t *d_pat, const int32_t *d_pat_off,
                               int32_t n, int64_t *d_counts, int64_t *d_lf_steps, int32_t *d_status, int32_t *d_tmp,
                               Scratch &scratch) {
    if (n < 0 || (n > 0 && (!d_pat_off || !d_counts || !d_tmp))) return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    hipStream_t st = static_cast<hipStream_t>(scratch.stream);
    // one processing order for all segments: any grouping is valid, and equal characters get equal codes in
    // every segment's alphabet, so the order derived from the first segment groups the batch for all of them.
    // The plan's code words are in the first segment's alphabet; the other segments translate them in LDS
    // (code -> character -> their own code), or map the characters themselves when codes are 16 bits wide.
    fmx::CountPlan plan;
    int rc = plan_order(segs[0], d_pat, d_pat_off, n, scratch, &plan);
    if (rc) return rc;
    int32_t *cnt = d_tmp, *lf = d_tmp + n, *sts = d_tmp + 2 * (size_t)n;
    for (int32_t s = 0; s < n_segs; ++s) {
        int e = k_launch_count(segs[s], segs[s]->dev, segs[s]->n_cu, d_pat, d_pat_off, &plan, s != 0, n, cnt, lf, sts, nullptr, st);
        if (e) return fail(FMX_E_HIP, std::string("k_count launch: ") + hipGetErrorString((hipError_t)e));
        e = fmx::launch_segment_add_counts(d_counts, d_lf_steps, d_status, cnt, lf, sts, n, s == 0, st);
        if (e) return fail(FMX_E_HIP, std::string("k_segment_add_counts launch: ") + hipGetErrorString((hipError_t)e));
    }
    return FMX_OK;
}

int fmx_count_segments_dev(const fmx_index *const *segs, int32_t n_segs, const uint16_t *d_pat, const int32_t *d_pat_off,
                           int32_t n, int64_t *d_counts, int64_t *d_lf_steps, int32_t *d_status, int32_t *d_tmp,
                           void *stream) {
    return guarded([&]() -> int {
    int rc = segments_ok(segs, n_segs);
    if (rc) return rc;
    Scratch scratch(segs[0], stream, false);
    return count_segments_impl(segs, n_segs, d_pat, d_pat_off, n, d_counts, d_lf_steps, d_status, d_tmp, scratch);
    });
}

// d_counts (nullable; with it d_lf_steps, nullable): ALSO the count() of every pattern summed over the segments — the range
// search of a segment yields both the count and the SA range the hits are located from, so a caller that wants count() and
// locate() of one batch (BASELINE configs[4]) pays ONE k_count per segment instead of two (fmx_count_locate_segments_dev)
static int locate_segments_impl(const fmx_index *const *segs, int32_t n_segs, const int64_t *seg_base, const uint16_t *d_pat,
                                const int32_t *d_pat_off, int32_t n, int32_t max_matches, int64_t *d_locs, int32_t *d_found,
                                int32_t *d_status, int32_t *d_tmp, Scratch &scratch, int64_t *d_counts = nullptr,
                                int64_t *d_lf_steps = nullptr) {
    if (n < 0 || max_matches < 1 || !seg_base || (int64_t)n * max_matches > INT32_MAX ||
        (n > 0 && (!d_pat_off || !d_locs || !d_found || !d_tmp)))
        return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    hipStream_t st = static_cast<hipStream_t>(scratch.stream);
    int32_t *seg_found = d_tmp, *seg_status = d_tmp + n, *range = d_tmp + 2 * (size_t)n, *seg_locs = d_tmp + 4 * (size_t)n;
    fmx::CountPlan plan;
    int rc = plan_order(segs[0], d_pat, d_pat_off, n, scratch, &plan);
    if (rc) return rc;
    // (the walk order's scratch belongs to the set's first index, like the plan's: ONE block for all segments — the size only
    // depends on n)
    void *ws = nullptr;
    const size_t ws_bytes = k_walk_workspace_bytes(segs[0], segs[0]->dev, n);
    rc = scratch.get(kWsWalk, ws_bytes, &ws);
    if (rc) return rc;
    // Segment s + 1's range search depends on the batch and its plan alone: it runs on a side stream BESIDE segment s's walk,
    // into a second set of {found, status, range} buffers (the sets alternate; the side stream waits for the append that last
    // read the set it is about to overwrite).  Two event edges per segment (~10 us) against a range search of 0.1 ms per million
    // patterns: only for batches that large, and only for the per-stream form (a host call's own stream lives for one call).
    int32_t *set_found[2] = {seg_found, nullptr}, *set_status[2] = {seg_status, nullptr}, *set_range[2] = {range, nullptr};
    fmx_index::SideLane *lane = nullptr;
    if (g_segments_overlap && !scratch.per_call && n_segs > 1 && n >= g_segments_overlap_min.load()) {
        void *second = nullptr;
        rc = scratch.get(kWsSegRange, (size_t)n * 4 * sizeof(int32_t), &second);
        if (rc) return rc;
        lane = side_lane(segs[0], scratch.stream, 2 * (size_t)n_segs + 2);  // (+ 1: LaneJoin's own)
        if (lane) {
            set_found[1] = static_cast<int32_t *>(second);
            set_status[1] = set_found[1] + n;
            set_range[1] = set_found[1] + 2 * (size_t)n;
        }
    }
    // (the counts of a segment, and its LF-steps, in buffers of their own — two sets beside the side stream — when the caller
    // wants count() as well: the walk overwrites `found`)
    int32_t *set_cnt[2] = {nullptr, nullptr}, *set_lf[2] = {nullptr, nullptr};
    if (d_counts) {
        void *extra = nullptr;
        rc = scratch.get(kWsSegCounts, (size_t)n * 4 * sizeof(int32_t), &extra);
        if (rc) return rc;
        for (int b = 0; b < 2; ++b) {
            set_cnt[b] = static_cast<int32_t *>(extra) + (size_t)b * 2 * (size_t)n;
            set_lf[b] = set_cnt[b] + n;
        }
    }
    // Whatever leaves this function early (a failed launch, event or wait) must not leave the side stream's range search running
    // on the set buffers and the plan records behind the caller's back: the caller's stream is made to wait for the side stream
    // (the last event of the lane serves; failing that, the host waits), so that the next call on `st` — or
    // fmx_release_scratch — finds them idle.
    struct LaneJoin {
        fmx_index::SideLane *lane;
        hipStream_t st;
        bool done = false;
        ~LaneJoin() {
            if (!lane || done) return;
            hipEvent_t last = lane->ev.back();
            if (hipEventRecord(last, lane->s) != hipSuccess || hipStreamWaitEvent(st, last, 0) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipStreamSynchronize(lane->s);
            }
        }
    } lane_join{lane, st};
    auto count_into = [&](int32_t s, hipStream_t on) {
        const int b = lane ? s & 1 : 0;
        return k_launch_count(segs[s], segs[s]->dev, segs[s]->n_cu, d_pat, d_pat_off, &plan, s != 0, n,
                              d_counts ? set_cnt[b] : set_found[b], (d_counts && d_lf_steps) ? set_lf[b] : nullptr, set_status[b],
                              set_range[b], on);
    };
    // (events of the lane: [0] = the plan is made, [1 + s] = segment s's ranges are there, [1 + n_segs + s] = segment s is appended)
    auto ev = [&](size_t i) { return lane->ev[i]; };
    int e = count_into(0, st);
    if (e) return fail(FMX_E_HIP, std::string("k_count launch: ") + hipGetErrorString((hipError_t)e));
    if (lane) {
        HIP_TRY(hipEventRecord(ev(0), st));  // (behind the plan and segment 0's search: the side stream starts from here)
        HIP_TRY(hipStreamWaitEvent(lane->s, ev(0), 0));
    }
    for (int32_t s = 0; s < n_segs; ++s) {
        const int b = lane ? s & 1 : 0;
        if (lane && s + 1 < n_segs) {
            if (s >= 1) HIP_TRY(hipStreamWaitEvent(lane->s, ev(1 + (size_t)n_segs + (size_t)(s - 1)), 0));  // set (s + 1) & 1 was segment s - 1's
            e = count_into(s + 1, lane->s);
            if (e) return fail(FMX_E_HIP, std::string("k_count launch: ") + hipGetErrorString((hipError_t)e));
            HIP_TRY(hipEventRecord(ev(1 + (size_t)(s + 1)), lane->s));
        } else if (!lane && s >= 1) {
            e = count_into(s, st);
            if (e) return fail(FMX_E_HIP, std::string("k_count launch: ") + hipGetErrorString((hipError_t)e));
        }
        if (lane && s >= 1) HIP_TRY(hipStreamWaitEvent(st, ev(1 + (size_t)s), 0));
        if (d_counts) {
            e = fmx::launch_segment_add_counts(d_counts, d_lf_steps, nullptr, set_cnt[b], set_lf[b], nullptr, n, s == 0, st);
            if (e) return fail(FMX_E_HIP, std::string("k_segment_add_counts launch: ") + hipGetErrorString((hipError_t)e));
        }
        // like the caller's loop `n += seg.locate(p, 0, len, locations, maxMatches - n)`: hits already
        // taken from earlier segments shrink this segment's limit
        // (the hits go straight into the set's rows behind those already taken, and a commit of a few bytes per pattern replaces
        // the append of every hit; option "segments_direct" = 0 keeps the staged form)
        const bool direct = g_segments_direct.load() != 0;
        e = k_launch_locate_walk(segs[s], segs[s]->dev, segs[s]->n_cu, set_range[b], n, max_matches, seg_locs, max_matches,
                                 set_found[b], nullptr, set_status[b], s ? d_found : nullptr, ws, ws_bytes, !scratch.per_call, st,
                                 direct ? d_locs : nullptr, direct ? seg_base[s] : 0);
        if (e) {
            if (ws && !scratch.per_call) (void)hipMemsetAsync(ws, 0, fmx::kPlanHeadBytes, st);
            return fail(FMX_E_HIP, std::string("k_locate_walk launch: ") + hipGetErrorString((hipError_t)e));
        }
        e = direct ? fmx::launch_segment_commit(d_found, d_status, set_found[b], set_status[b], n, max_matches, s == 0, st)
                   : fmx::launch_segment_append_hits(d_locs, d_found, d_status, seg_locs, set_found[b], set_status[b], n, max_matches,
                                                     seg_base[s], s == 0, st);
        if (e) return fail(FMX_E_HIP, std::string("k_segment_append_hits / k_segment_commit launch: ") + hipGetErrorString((hipError_t)e));
        if (lane && s + 2 < n_segs) HIP_TRY(hipEventRecord(ev(1 + (size_t)n_segs + (size_t)s), st));
    }
    lane_join.done = true;  // (every search of the side stream has been waited for by the walk that read its ranges)
    return FMX_OK;
}

int fmx_locate_segments_dev(const fmx_index *const *segs, int32_t n_segs, const int64_t *seg_base, const uint16_t *d_pat,
                            const int32_t *d_pat_off, int32_t n, int32_t max_matches, int64_t *d_locs, int32_t *d_found,
                            int32_t *d_status, int32_t *d_tmp, void *stream) {
    return guarded([&]() -> int {
    int rc = segments_ok(segs, n_segs);
    if (rc) return rc;
    Scratch scratch(segs[0], stream, false);
    return locate_segments_impl(segs, n_segs, seg_base, d_pat, d_pat_off, n, max_matches, d_locs, d_found, d_status, d_tmp,
                                scratch);
    });
}

int fmx_count_locate_segments_dev(const fmx_index *const *segs, int32_t n_segs, const int64_t *seg_base, const uint16_t *d_pat,
                                  const int32_t *d_pat_off, int32_t n, int32_t max_matches, int64_t *d_counts, int64_t *d_lf_steps,
                                  int64_t *d_locs, int32_t *d_found, int32_t *d_status, int32_t *d_tmp, void *stream) {
    return guarded([&]() -> int {
    int rc = segments_ok(segs, n_segs);
    if (rc) return rc;
    if (n > 0 && !d_counts) return fail(FMX_E_ARG, "bad arguments");
    Scratch scratch(segs[0], stream, false);
    return locate_segments_impl(segs, n_segs, seg_base, d_pat, d_pat_off, n, max_matches, d_locs, d_found, d_status, d_tmp,
                                scratch, d_counts, d_lf_steps);
    });
}

int fmx_count_segments(const fmx_index *const *segs, int32_t n_segs, const uint16_t *pat, const int32_t *pat_off,
                       int32_t n, int64_t *counts, int64_t *lf_steps, int32_t *status) {
    return guarded([&]() -> int {
    int rc = segments_ok(segs, n_segs);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!pat_off || !counts))) return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(segs[0]->device));
    rc = check_offsets(pat_off, n);
    if (rc) return rc;
    const size_t chars = (size_t)(pat_off[n] > 0 ? pat_off[n] : 0);
    DevBuf d_pat, d_off, d_cnt, d_lf, d_st, d_tmp;
    HostCallStream hs;
    rc = hs.init(segs[0]->device);
    if (rc) return rc;
    Scratch scratch(segs[0], hs.s, true);
    HostCallStream wait_first;  // (destroyed before the scratch above: the stream is drained, then the blocks go back)
    wait_first.s = hs.s;
    HIP_TRY(d_pat.alloc(chars * 2 + 8));
    HIP_TRY(d_off.alloc((size_t)(n + 1) * 4));
    HIP_TRY(d_cnt.alloc((size_t)n * 8));
    HIP_TRY(d_lf.alloc((size_t)n * 8));
    HIP_TRY(d_st.alloc((size_t)n * 4));
    HIP_TRY(d_tmp.alloc((size_t)n * 12));
    if (chars > first_char(pat_off)) H2D(d_pat.as<uint16_t>() + first_char(pat_off), pat + first_char(pat_off), (chars - first_char(pat_off)) * 2);
    H2D(d_off.p, pat_off, (size_t)(n + 1) * 4);
    rc = count_segments_impl(segs, n_segs, d_pat.as<uint16_t>(), d_off.as<int32_t>(), n, d_cnt.as<int64_t>(),
                             d_lf.as<int64_t>(), d_st.as<int32_t>(), d_tmp.as<int32_t>(), scratch);
    HIP_TRY(hipStreamSynchronize(hs.s));  // also on failure: the per-call blocks go back to the cache below
    if (rc) return rc;
    D2H(counts, d_cnt.p, (size_t)n * 8);
    if (lf_steps) D2H(lf_steps, d_lf.p, (size_t)n * 8);
    if (status) D2H(status, d_st.p, (size_t)n * 4);
    return FMX_OK;
    });
}

int fmx_locate_segments(const fmx_index *const *segs, int32_t n_segs, const int64_t *seg_base, const uint16_t *pat,
                        const int32_t *pat_off, int32_t n, int32_t max_matches, int64_t *locs, int32_t *found,
                        int32_t *status) {
    return guarded([&]() -> int {
    int rc = segments_ok(segs, n_segs);
    if (rc) return rc;
    if (n < 0 || max_matches < 1 || !seg_base || (int64_t)n * max_matches > INT32_MAX ||
        (n > 0 && (!pat_off || !locs || !found)))
        return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(segs[0]->device));
    rc = check_offsets(pat_off, n);
    if (rc) return rc;
    const size_t chars = (size_t)(pat_off[n] > 0 ? pat_off[n] : 0);
    const size_t slots = (size_t)n * (size_t)max_matches;
    DevBuf d_pat, d_off, d_locs, d_found, d_st, d_tmp;
    HostCallStream hs;
    rc = hs.init(segs[0]->device);
    if (rc) return rc;
    Scratch scratch(segs[0], hs.s, true);
    HostCallStream wait_first;  // (destroyed before the scratch above: the stream is drained, then the blocks go back)
    wait_first.s = hs.s;
    HIP_TRY(d_pat.alloc(chars * 2 + 8));
    HIP_TRY(d_off.alloc((size_t)(n + 1) * 4));
    HIP_TRY(d_locs.alloc(slots * 8));
    HIP_TRY(d_found.alloc((size_t)n * 4));
    HIP_TRY(d_st.alloc((size_t)n * 4));
    HIP_TRY(d_tmp.alloc(((size_t)n * 4 + slots) * 4));
    if (chars > first_char(pat_off)) H2D(d_pat.as<uint16_t>() + first_char(pat_off), pat + first_char(pat_off), (chars - first_char(pat_off)) * 2);
    H2D(d_off.p, pat_off, (size_t)(n + 1) * 4);
    H2D(d_locs.p, locs, slots * 8);  // in/out: slots without a hit keep the caller's values
    rc = locate_segments_impl(segs, n_segs, seg_base, d_pat.as<uint16_t>(), d_off.as<int32_t>(), n, max_matches,
                              d_locs.as<int64_t>(), d_found.as<int32_t>(), d_st.as<int32_t>(), d_tmp.as<int32_t>(),
                              scratch);
    HIP_TRY(hipStreamSynchronize(hs.s));
    if (rc) return rc;
    D2H(locs, d_locs.p, slots * 8);
    D2H(found, d_found.p, (size_t)n * 4);
    if (status) D2H(status, d_st.p, (size_t)n * 4);
    return FMX_OK;
    });
}

int fmx_count_locate_segments(const fmx_index *const *segs, int32_t n_segs, const int64_t *seg_base, const uint16_t *pat,
                              const int32_t *pat_off, int32_t n, int32_t max_matches, int64_t *counts, int64_t *lf_steps,
                              int64_t *locs, int32_t *found, int32_t *status) {
    return guarded([&]() -> int {
    int rc = segments_ok(segs, n_segs);
    if (rc) return rc;
    if (n < 0 || max_matches < 1 || !seg_base || (int64_t)n * max_matches > INT32_MAX ||
        (n > 0 && (!pat_off || !locs || !found || !counts)))
        return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(segs[0]->device));
    rc = check_offsets(pat_off, n);
    if (rc) return rc;
    const size_t chars = (size_t)(pat_off[n] > 0 ? pat_off[n] : 0);
    const size_t slots = (size_t)n * (size_t)max_matches;
    DevBuf d_pat, d_off, d_cnt, d_lf, d_locs, d_found, d_st, d_tmp;
    HostCallStream hs;
    rc = hs.init(segs[0]->device);
    if (rc) return rc;
    Scratch scratch(segs[0], hs.s, true);
    HostCallStream wait_first;  // (destroyed before the scratch above: the stream is drained, then the blocks go back)
    wait_first.s = hs.s;
    HIP_TRY(d_pat.alloc(chars * 2 + 8));
    HIP_TRY(d_off.alloc((size_t)(n + 1) * 4));
    HIP_TRY(d_cnt.alloc((size_t)n * 8));
    HIP_TRY(d_lf.alloc((size_t)n * 8));
    HIP_TRY(d_locs.alloc(slots * 8));
    HIP_TRY(d_found.alloc((size_t)n * 4));
    HIP_TRY(d_st.alloc((size_t)n * 4));
    HIP_TRY(d_tmp.alloc(((size_t)n * 4 + slots) * 4));
    if (chars > first_char(pat_off)) H2D(d_pat.as<uint16_t>() + first_char(pat_off), pat + first_char(pat_off), (chars - first_char(pat_off)) * 2);
    H2D(d_off.p, pat_off, (size_t)(n + 1) * 4);
    H2D(d_locs.p, locs, slots * 8);  // in/out: slots without a hit keep the caller's values
    rc = locate_segments_impl(segs, n_segs, seg_base, d_pat.as<uint16_t>(), d_off.as<int32_t>(), n, max_matches,
                              d_locs.as<int64_t>(), d_found.as<int32_t>(), d_st.as<int32_t>(), d_tmp.as<int32_t>(), scratch,
                              d_cnt.as<int64_t>(), lf_steps ? d_lf.as<int64_t>() : nullptr);
    HIP_TRY(hipStreamSynchronize(hs.s));
    if (rc) return rc;
    D2H(counts, d_cnt.p, (size_t)n * 8);
    if (lf_steps) D2H(lf_steps, d_lf.p, (size_t)n * 8);
    D2H(locs, d_locs.p, slots * 8);
    D2H(found, d_found.p, (size_t)n * 4);
    if (status) D2H(status, d_st.p, (size_t)n * 4);
    return FMX_OK;
    });
}

// ---- host-buffer entry points --------------------------------------------------------------------


constexpr int32_t kPipeChunkMin = 65536;  // smallest stage of the pipeline (patterns)

// is `p` host memory the GPU can DMA from / to directly (hipHostMalloc, hipHostRegister / fmx_host_register)?
static bool is_pinned(const void *p);
// ... from its first byte to its last (a caller that registered only the head of an array gets the staged path)
static bool is_pinned_range(const void *p, size_t bytes) {
    return bytes == 0 ? is_pinned(p) : is_pinned(p) && is_pinned(static_cast<const char *>(p) + bytes - 1);
}
static bool is_pinned(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'ed memory: "invalid value", not an error of ours
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// device address of a registered (mapped) host array, or nullptr.  [p, p + bytes) must lie inside ONE range registered through
// fmx_host_register (g_registered): probing the two ends with hipHostGetDevicePointer, as round 4 did, proves nothing on ROCm —
// a registered array's device pointer IS its host address, so two separate registrations with an unmapped gap between them
// looked like one range, and a kernel reading or writing the gap faults (with XNACK off that aborts the process: a JVM under
// JNI).  Arrays the caller pinned some other way simply take the staged path.
static void *mapped_range(const void *p, size_t bytes) {
    if (!p || bytes == 0) return nullptr;
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p);
    {
        std::lock_guard<std::mutex> lock(g_registered_mutex);
        auto it = g_registered.upper_bound(lo);  // the last registration that starts at or below p
        if (it == g_registered.begin()) return nullptr;
        --it;
        if (lo - it->first > it->second || bytes > it->second - (lo - it->first)) return nullptr;
    }
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, const_cast<void *>(p), 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return dev;
}

// fmx_count_batch with EVERY array registered (option "host_mapped", default 1): no copies at all — k_count reads the characters
// (and the offsets, unless the patterns are of one length: those offsets are made on the device) from the caller's mapped arrays
// and stores its results into them; the launch streams the batch over PCIe while it counts.  Returns -1 if the arrays are not
// all mapped (the caller then takes the pipeline).
static int count_batch_mapped(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n, int32_t *counts,
                              int32_t *lf_steps, int32_t *status) {
    // (the ends of the batch first: the size of the characters' range comes from the last offset)
    if (pat_off[0] < 0 || pat_off[n] < pat_off[0]) return fail(FMX_E_ARG, "pattern offsets start below 0 or decrease");
    const int64_t total_chars = pat_off[n];
    void *m_off = mapped_range(pat_off, ((size_t)n + 1) * 4);
    void *m_pat = total_chars ? mapped_range(pat, (size_t)total_chars * 2) : m_off;  // (no characters: never read)
    void *m_cnt = mapped_range(counts, (size_t)n * 4);
    void *m_lf = lf_steps ? mapped_range(lf_steps, (size_t)n * 4) : nullptr;
    void *m_st = status ? mapped_range(status, (size_t)n * 4) : nullptr;
    if (!m_off || !m_pat || !m_cnt || (lf_steps && !m_lf) || (status && !m_st)) return -1;
    bool uniform = false;
    if (pat_off[0] < 0 || total_chars < pat_off[0] || !scan_offsets(pat_off, 0, n, total_chars, &uniform))
        return fail(FMX_E_ARG, "pattern offsets decrease or leave the batch");
    PipeStreams *ps = nullptr;
    int rc = pipe_streams(idx->device, &ps);
    if (rc) return rc;
    hipStream_t st = ps->s[1];
    // ONE launch for the whole batch (in chunks — the host's pass over the offsets of chunk c + 1 beside the kernel of chunk c —
    // it was slower: 0.525 ms in four launches, 0.482 in two, 0.455 in one; a kernel that is fed over the link wants every
    // wave of the chip asking at once)
    DevBuf d_off;
    const int32_t *offsets = static_cast<const int32_t *>(m_off);
    if (uniform) {
        HIP_TRY(d_off.alloc(((size_t)n + 1) * 4));
        if (int e = fmx::launch_fill_offsets(d_off.as<int32_t>(), pat_off[0], pat_off[1] - pat_off[0], n + 1, st))
            return fail(FMX_E_HIP, std::string("offsets: ") + hipGetErrorString((hipError_t)e));
        offsets = d_off.as<int32_t>();
    }
    Scratch scratch(idx, st, true);
    rc = count_impl(idx, static_cast<const uint16_t *>(m_pat), offsets, n, static_cast<int32_t *>(m_cnt), static_cast<int32_t *>(m_lf),
                    static_cast<int32_t *>(m_st), scratch, false);
    const hipError_t e = hipStreamSynchronize(st);  // (also on failure: the per-call blocks go back to the cache below)
    if (rc) return rc;
    if (e != hipSuccess) return fail(FMX_E_HIP, std::string("mapped count: ") + hipGetErrorString(e));
    return FMX_OK;
}

// fmx_count_batch for large batches — what a JNI binding's count(char[][]) costs is PCIe, not the kernels: the batch goes
// to the GPU in chunks, chunk c's transfer overlapping the kernels of chunk c - 1 and the return of chunk c - 2 (three streams); offsets of equal-length
// runs are made on the device instead of being shipped; the offsets are validated chunk by chunk on the way (the same
// pass finds the equal-length runs).  Results of a chunk return as soon as its stage is done: straight into the
// caller's arrays when those are pinned (fmx_host_register), else through pinned staging the kernels store into, copied
// out in parts by whichever of the call's threads is idle (helper, feeder, this one).
static int count_batch_pipelined(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n, int32_t *counts,
                                 int32_t *lf_steps, int32_t *status) {
    const double t_enter = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (pat_off[0] < 0 || pat_off[n] < pat_off[0]) return fail(FMX_E_ARG, "pattern offsets start below 0 or decrease");
    const int64_t total_chars = pat_off[n];
    PipeStreams *ps = nullptr;
    int rc = pipe_streams(idx->device, &ps);
    if (rc) return rc;
    // equal chunks (a schedule of halving sizes — a small last chunk, so that little is left that overlaps nothing — was
    // slower: the first chunk's kernels then start after half of the transfer; profiles/r03_experiments.txt)
    const int32_t chunk = std::max<int32_t>(kPipeChunkMin, g_host_pipeline_chunk.load());
    std::vector<int32_t> bounds(1, 0);
    while (bounds.back() < n) bounds.push_back((int32_t)std::min<int64_t>(n, (int64_t)bounds.back() + chunk));
    const int32_t n_chunks = (int32_t)bounds.size() - 1;
    const bool direct_out = is_pinned_range(counts, (size_t)n * 4) && (!lf_steps || is_pinned_range(lf_steps, (size_t)n * 4)) &&
                            (!status || is_pinned_range(status, (size_t)n * 4));
    DevBuf d_pat, d_off, d_cnt, d_lf, d_st;
    PinBuf h_cnt, h_lf, h_st;
    PinBuf h_pat;  // pinned staging of the characters of a pageable source (option host_stage_threads >= 2); lives until the streams are drained
    HIP_TRY(d_pat.alloc((size_t)total_chars * 2 + 8));
    HIP_TRY(d_off.alloc(((size_t)n + 1 + (size_t)n_chunks) * 4));  // every chunk its own run of offsets (n_c + 1 entries)
    int32_t *o_cnt = counts, *o_lf = lf_steps, *o_st = status;  // where the D2H copies land
    // Registered result arrays are mapped into the device's address space: k_count then STORES counts, statuses and LF-steps
    // straight into them (a chunk of the pipeline is counted in the caller's order, so a wave's stores are consecutive words:
    // whole PCIe writes) — no result copies at all, which also leaves the link's other direction to the characters coming in.
    if (!direct_out) {
        HIP_TRY(h_cnt.alloc((size_t)n * 4));
        o_cnt = h_cnt.as<int32_t>();
        if (lf_steps) {
            HIP_TRY(h_lf.alloc((size_t)n * 4));
            o_lf = h_lf.as<int32_t>();
        }
        if (status) {
            HIP_TRY(h_st.alloc((size_t)n * 4));
            o_st = h_st.as<int32_t>();
        }
    }
    // (plain result arrays: the same stores go into the pinned STAGING the helper thread copies out of — o_cnt / o_lf / o_st
    // are registered arrays of the caller or that staging, mapped either way)
    int32_t *m_cnt = nullptr, *m_lf = nullptr, *m_st = nullptr;
    bool stores_out = false;
    if (g_host_direct_stores.load()) {
        // (a caller's array must lie inside one range registered through fmx_host_register: mapped_range; the library's own pinned
        // staging is mapped by construction — round 5's stricter mapped_range had silently sent it back to result COPIES, eight
        // device-to-host copies of 1 MB at 148 us each per 1 M-pattern call: 1.2 of the call's 1.75 ms, round 6)
        auto mapped_out = [&](int32_t *host) -> int32_t * {
            if (direct_out) return static_cast<int32_t *>(mapped_range(host, (size_t)n * 4));
            void *dev = nullptr;
            if (hipHostGetDevicePointer(&dev, host, 0) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
            return static_cast<int32_t *>(dev);
        };
        m_cnt = mapped_out(o_cnt);
        m_lf = lf_steps ? mapped_out(o_lf) : nullptr;
        m_st = status ? mapped_out(o_st) : nullptr;
        stores_out = m_cnt && (!lf_steps || m_lf) && (!status || m_st);
    }
    if (!stores_out) {
        HIP_TRY(d_cnt.alloc((size_t)n * 4));
        if (lf_steps) HIP_TRY(d_lf.alloc((size_t)n * 4));
        if (status) HIP_TRY(d_st.alloc((size_t)n * 4));
    }
    std::vector<std::unique_ptr<Scratch>> scratches;
    int failed = FMX_OK;
    std::atomic<int32_t> issued{0};
    std::atomic<bool> stop{false};
    std::atomic<int> out_error{0};
    auto drain = [&]() {  // false: a stream ended with an error (a failed kernel or copy of one of the last chunks)
        bool ok = true;
        for (int i = 0; i < kPipeStreams; ++i) ok = (hipStreamSynchronize(ps->s[i]) == hipSuccess) && ok;
        return ok;
    };
    struct DrainOnExit {  // declared after the buffers, before the threads: threads are joined, then the streams drained, then
        PipeStreams *ps;  // the blocks go back to their caches
        ~DrainOnExit() {
            for (int i = 0; i < kPipeStreams; ++i) (void)hipStreamSynchronize(ps->s[i]);
        }
    } drain_on_exit{ps};
    // registered result arrays: chunk b's results are in them once its stage is done
    auto copy_out = [&](int32_t b) {
        if (hipEventSynchronize(ps->done[b % kPipeEvents]) != hipSuccess) out_error = 1;
    };
    std::atomic<int32_t> copied{0};
    // Plain result arrays: a chunk's results leave the pinned staging in PARTS of kOutPart patterns, taken by whichever of this
    // call's three threads has nothing else to do — the helper, the feeder once the last characters are in HBM, this thread while
    // it waits and after the last launch.  (One thread copying a 262,144-pattern chunk out takes as long as the link takes to
    // bring the next one in, and the LAST chunk's copy overlaps nothing: 230 us of a 670 us call, round 6.)
    constexpr int32_t kOutPart = 32768;
    std::vector<int64_t> item_begin((size_t)n_chunks + 1, 0);  // parts of chunk b: items [item_begin[b], item_begin[b + 1])
    for (int32_t b = 0; b < n_chunks; ++b)
        item_begin[(size_t)b + 1] = item_begin[(size_t)b] + (bounds[(size_t)b + 1] - bounds[(size_t)b] + kOutPart - 1) / kOutPart;
    std::unique_ptr<std::atomic<int32_t>[]> parts_left(new std::atomic<int32_t>[(size_t)n_chunks]);
    for (int32_t b = 0; b < n_chunks; ++b) parts_left[(size_t)b].store((int32_t)(item_begin[(size_t)b + 1] - item_begin[(size_t)b]));
    std::atomic<int64_t> next_item{0};
    // one part, if one can be had: of a chunk that has been issued and (wait: once) its stage is done.  false: nothing to take now.
    auto take_item = [&](bool wait) -> bool {
        for (;;) {
            int64_t it = next_item.load(std::memory_order_acquire);
            const int32_t have = issued.load(std::memory_order_acquire);
            if (it >= item_begin[(size_t)have]) return false;
            int32_t b = copied.load(std::memory_order_acquire);
            while (it >= item_begin[(size_t)b + 1]) ++b;
            hipEvent_t ev = ps->done[b % kPipeEvents];
            if (wait) {  // (polling the event instead: no faster)
                if (hipEventSynchronize(ev) != hipSuccess) out_error = 1;  // (the part is still taken: the call fails as a whole)
            } else if (hipEventQuery(ev) != hipSuccess) {
                (void)hipGetLastError();  // not ready
                return false;
            }
            if (!next_item.compare_exchange_strong(it, it + 1, std::memory_order_acq_rel)) continue;
            const int32_t plo = bounds[(size_t)b] + (int32_t)(it - item_begin[(size_t)b]) * kOutPart;
            const int32_t phi = std::min(bounds[(size_t)b + 1], plo + kOutPart);
            memcpy(counts + plo, o_cnt + plo, (size_t)(phi - plo) * 4);
            if (lf_steps) memcpy(lf_steps + plo, o_lf + plo, (size_t)(phi - plo) * 4);
            if (status) memcpy(status + plo, o_st + plo, (size_t)(phi - plo) * 4);
            if (parts_left[(size_t)b].fetch_sub(1, std::memory_order_acq_rel) == 1) {
                for (;;) {  // `copied` = the chunks before it are out, whole
                    int32_t c = copied.load(std::memory_order_acquire);
                    if (c >= n_chunks || parts_left[(size_t)c].load(std::memory_order_acquire) != 0) break;
                    copied.compare_exchange_strong(c, c + 1, std::memory_order_acq_rel);
                }
            }
            return true;
        }
    };
    // (whatever leaves this function — an exception included — first stops and joins its threads: they work on this frame)
    struct JoinOnExit {
        std::thread &t;
        std::atomic<bool> &flag;
        ~JoinOnExit() {
            flag = true;
            if (t.joinable()) t.join();
        }
    };
    std::thread helper;
    JoinOnExit join_helper{helper, stop};
    if (!direct_out) {
        const int device = idx->device;
        helper = std::thread([&, device]() {
            (void)hipSetDevice(device);
            for (;;) {
                if (take_item(true)) continue;
                if (stop.load()) {
                    if (next_item.load(std::memory_order_acquire) >= item_begin[(size_t)issued.load(std::memory_order_acquire)]) return;
                } else {
                    std::this_thread::yield();
                }
            }
        });
    }
    static const bool timing = getenv("FMX_PIPE_TIMING") != nullptr;  // stderr: where a call's host time goes
    auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_scan = 0, t_in = 0, t_launch = 0, t_out = 0;
    const double t_begin = now();
    hipStream_t s_in = ps->s[0], s_k = ps->s[1], s_out = ps->s[2];
    const bool in_pinned = is_pinned_range(pat, (size_t)total_chars * 2) && is_pinned_range(pat_off, ((size_t)n + 1) * 4);
    // A pageable source is staged by whichever thread calls the copy: a FEEDER thread pushes the chunks' characters one
    // after the other with the plain copy (twice the rate of the asynchronous one from pageable memory; it returns when
    // the bytes are in HBM, so the kernels need no event), so that the link stays busy while this thread checks the
    // offsets and launches kernels.  (The two ends of every chunk are checked before anything moves.)
    for (int32_t c = 0; c < n_chunks; ++c) {
        const int64_t c0 = pat_off[bounds[(size_t)c]], c1 = pat_off[bounds[(size_t)c + 1]];
        if (c0 < 0 || c1 < c0 || c1 > total_chars) {
            stop = true;
            if (helper.joinable()) helper.join();
            return fail(FMX_E_ARG, "pattern offsets decrease or leave the batch");
        }
    }
    std::unique_ptr<std::atomic<uint8_t>[]> fed(new std::atomic<uint8_t>[(size_t)n_chunks]);  // chunk c's characters are in HBM (staged: on their way)
    for (int32_t c = 0; c < n_chunks; ++c) fed[(size_t)c].store(0);
    std::atomic<int> feed_error{0};
    std::atomic<bool> feed_stop{false};
    std::thread feeder;
    JoinOnExit join_feeder{feeder, feed_stop};
    const int stage_threads = g_host_stage_threads.load();
    const bool staged_in = !in_pinned && stage_threads >= 2 && total_chars > pat_off[0] &&
                           h_pat.alloc((size_t)(total_chars - pat_off[0]) * 2) == hipSuccess;
    if (!staged_in) (void)hipGetLastError();
    // (ONE feeder: two threads with every other chunk each are no faster — the runtime serialises plain copies — and the tail grows)
    if (!in_pinned) {
        const int device = idx->device;
        feeder = std::thread([&, device]() {
            (void)hipSetDevice(device);
            const int64_t first = pat_off[0];
            for (int32_t c = 0; c < n_chunks && !feed_stop.load(); ++c) {
                const int64_t c0 = pat_off[bounds[(size_t)c]], c1 = pat_off[bounds[(size_t)c + 1]];
                if (c1 > c0) {
                    if (staged_in) {
                        // this chunk's characters into the pinned staging by several threads, then DMA behind the chunk before it
                        uint16_t *stage = h_pat.as<uint16_t>() + (c0 - first);
                        copy_pool().copy(stage, pat + c0, (size_t)(c1 - c0) * 2, stage_threads);
                        if (hipMemcpyAsync(d_pat.as<uint16_t>() + c0, stage, (size_t)(c1 - c0) * 2, hipMemcpyHostToDevice, s_in) != hipSuccess)
                            feed_error = 1;
                    } else if (hipMemcpy(d_pat.as<uint16_t>() + c0, pat + c0, (size_t)(c1 - c0) * 2, hipMemcpyHostToDevice) != hipSuccess) {
                        feed_error = 1;
                    }
                }
                if (staged_in && hipEventRecord(ps->in[c % kPipeEvents], s_in) != hipSuccess) feed_error = 1;
                fed[(size_t)c].store(1, std::memory_order_release);
            }
            while (!direct_out && !feed_stop.load())  // the characters are in: results out, beside the helper
                if (!take_item(false)) std::this_thread::yield();
        });
    }
    for (int32_t c = 0; c < n_chunks && !failed; ++c) {
        const int32_t lo = bounds[(size_t)c], hi = bounds[(size_t)c + 1], n_c = hi - lo;
        const int slot = c % kPipeEvents;
        double t0 = now();
        // an event slot is reused every kPipeEvents chunks: its earlier chunk must have been copied out
        while (c >= kPipeEvents && copied.load(std::memory_order_acquire) <= c - kPipeEvents && !out_error) {
            if (direct_out) {
                copy_out(copied.load());
                copied.fetch_add(1);
            } else if (!take_item(false)) {
                std::this_thread::yield();
            }
        }
        // the chunk's characters start travelling before its offsets are looked at
        const int64_t c0 = pat_off[lo], c1 = pat_off[hi];
        hipError_t e = hipSuccess;
        if (c1 > c0 && in_pinned)
            e = hipMemcpyAsync(d_pat.as<uint16_t>() + c0, pat + c0, (size_t)(c1 - c0) * 2, hipMemcpyHostToDevice, s_in);
        t_in += now() - t0;
        t0 = now();
        bool uniform = false;
        if (!scan_offsets(pat_off, lo, hi, total_chars, &uniform)) {
            failed = fail(FMX_E_ARG, "pattern offsets decrease or leave the batch");
            break;
        }
        t_scan += now() - t0;
        t0 = now();
        int32_t *d_off_c = d_off.as<int32_t>() + lo + c;
        if (e == hipSuccess) {
            if (uniform)
                e = (hipError_t)fmx::launch_fill_offsets(d_off_c, (int32_t)c0, pat_off[lo + 1] - pat_off[lo], n_c + 1, s_k);
            else if (in_pinned)
                e = hipMemcpyAsync(d_off_c, pat_off + lo, (size_t)(n_c + 1) * 4, hipMemcpyHostToDevice, s_in);
            else
                e = hipMemcpyAsync(d_off_c, pat_off + lo, (size_t)(n_c + 1) * 4, hipMemcpyHostToDevice, s_k);
        }
        if (!in_pinned) {  // the feeder has this chunk's characters in HBM (or, staged: on their way, behind the chunk's event)?
            while (!fed[(size_t)c].load(std::memory_order_acquire))
                if (direct_out || !take_item(false)) std::this_thread::yield();
            if (feed_error) e = hipErrorUnknown;
            if (e == hipSuccess && staged_in) e = hipStreamWaitEvent(s_k, ps->in[slot], 0);
        }
        if (e == hipSuccess && in_pinned) {
            e = hipEventRecord(ps->in[slot], s_in);
            if (e == hipSuccess) e = hipStreamWaitEvent(s_k, ps->in[slot], 0);
        }
        if (e != hipSuccess) {
            failed = fail(FMX_E_HIP, std::string("host-buffer pipeline, copy in: ") + hipGetErrorString(e));
            break;
        }
        t_in += now() - t0;
        t0 = now();
        scratches.emplace_back(new Scratch(idx, s_k, true));
        if (stores_out)
            rc = count_impl(idx, d_pat.as<uint16_t>(), d_off_c, n_c, m_cnt + lo, lf_steps ? m_lf + lo : nullptr, status ? m_st + lo : nullptr,
                            *scratches.back());
        else
            rc = count_impl(idx, d_pat.as<uint16_t>(), d_off_c, n_c, d_cnt.as<int32_t>() + lo, lf_steps ? d_lf.as<int32_t>() + lo : nullptr,
                            status ? d_st.as<int32_t>() + lo : nullptr, *scratches.back());
        if (rc) {
            failed = rc;
            break;
        }
        t_launch += now() - t0;
        t0 = now();
        if (stores_out) {  // the chunk's results are in the caller's arrays when its kernel has ended
            e = hipEventRecord(ps->done[slot], s_k);
            if (e != hipSuccess) {
                failed = fail(FMX_E_HIP, std::string("host-buffer pipeline, results: ") + hipGetErrorString(e));
                break;
            }
            issued.store(c + 1, std::memory_order_release);
            t_out += now() - t0;
            continue;
        }
        e = hipEventRecord(ps->counted[slot], s_k);
        if (e == hipSuccess) e = hipStreamWaitEvent(s_out, ps->counted[slot], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(o_cnt + lo, d_cnt.as<int32_t>() + lo, (size_t)n_c * 4, hipMemcpyDeviceToHost, s_out);
        if (e == hipSuccess && lf_steps) e = hipMemcpyAsync(o_lf + lo, d_lf.as<int32_t>() + lo, (size_t)n_c * 4, hipMemcpyDeviceToHost, s_out);
        if (e == hipSuccess && status) e = hipMemcpyAsync(o_st + lo, d_st.as<int32_t>() + lo, (size_t)n_c * 4, hipMemcpyDeviceToHost, s_out);
        if (e == hipSuccess) e = hipEventRecord(ps->done[slot], s_out);
        if (e != hipSuccess) {
            failed = fail(FMX_E_HIP, std::string("host-buffer pipeline, copy out: ") + hipGetErrorString(e));
            break;
        }
        issued.store(c + 1, std::memory_order_release);
        t_out += now() - t0;
    }
    const double t_issued = now();
    if (!direct_out)
        while (take_item(true)) {
        }
    feed_stop = true;
    if (feeder.joinable()) feeder.join();
    stop = true;
    if (helper.joinable()) helper.join();
    const double t_joined = now();
    // registered result arrays: nobody has waited for the chunks whose event slot was never reused (all of them, for
    // fewer than kPipeEvents chunks) — wait for each stage here, so that a failed one is reported instead of FMX_OK
    if (direct_out && !failed)
        for (int32_t b = copied.load(); b < issued.load(); ++b) copy_out(b);
    // (also on failure: the per-call blocks go back to the cache when this call ends, nothing may still use them)
    if (!drain()) out_error = 1;
    if (timing)
        fprintf(stderr, "[fmx pipe] %d chunks: setup %.0f us | scan %.0f | copy-in calls %.0f | kernel launches %.0f | copy-out calls %.0f | "
                        "issue loop %.0f | helper join +%.0f | drain +%.0f (direct_out %d, staged_in %d)\n",
                n_chunks, t_begin - t_enter, t_scan, t_in, t_launch, t_out, t_issued - t_begin, t_joined - t_issued, now() - t_joined, (int)direct_out,
                (int)staged_in);
    if (failed) return failed;
    if (out_error) return fail(FMX_E_HIP, "host-buffer pipeline: a stage failed");
    return FMX_OK;
}

// fmx_count_batch for SMALL batches — a Java caller's count(char[]) is a batch of one, and what it costs is not the search but the
// calls around it: five blocking copies and a launch (91 us per call for one 8-character pattern; index4j's own count() takes 14 us
// on a host core).  Here everything a small call moves goes through ONE pinned block the kernels read and write where it lies (it
// is mapped into the device's address space: a few hundred bytes over the link, no copy call at all): memcpy in, one launch
// sequence, one wait, memcpy out.  option "host_small_max" (patterns; 0 = off).
constexpr size_t kHostSmallChars = 65536;
static int count_batch_small(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n, int32_t *counts,
                             int32_t *lf_steps, int32_t *status) {
    const int64_t first = first_char(pat_off);
    const size_t chars = (size_t)(pat_off[n] > first ? pat_off[n] - first : 0);
    // [characters | offsets (rebased to the block's characters) | counts | LF-steps | statuses], each 16-byte aligned
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_off = up(chars * 2 + 8), o_cnt = o_off + up(((size_t)n + 1) * 4), o_lf = o_cnt + up((size_t)n * 4),
                 o_st = o_lf + up((size_t)n * 4), total = o_st + up((size_t)n * 4);
    PinBuf block;
    HIP_TRY(block.alloc(total));
    uint8_t *h = block.as<uint8_t>();
    void *dv = nullptr;
    if (hipHostGetDevicePointer(&dv, h, 0) != hipSuccess || !dv) {
        (void)hipGetLastError();
        return -1;  // (not mapped on this platform: the caller takes the copying path)
    }
    uint8_t *d = static_cast<uint8_t *>(dv);
    if (chars) memcpy(h, pat + first, chars * 2);
    int32_t *h_off = reinterpret_cast<int32_t *>(h + o_off);
    for (int32_t i = 0; i <= n; ++i) h_off[i] = (int32_t)(pat_off[i] - first);
    HostCallStream hs;
    int rc = hs.init(idx->device);
    if (rc) return rc;
    Scratch scratch(idx, hs.s, true);
    HostCallStream wait_first;  // (destroyed before the scratch above and the block: the stream is drained, then they go back)
    wait_first.s = hs.s;
    rc = count_impl(idx, reinterpret_cast<const uint16_t *>(d), reinterpret_cast<const int32_t *>(d + o_off), n,
                    reinterpret_cast<int32_t *>(d + o_cnt), reinterpret_cast<int32_t *>(d + o_lf), reinterpret_cast<int32_t *>(d + o_st), scratch);
    HIP_TRY(hipStreamSynchronize(hs.s));
    if (rc) return rc;
    memcpy(counts, h + o_cnt, (size_t)n * 4);
    if (lf_steps) memcpy(lf_steps, h + o_lf, (size_t)n * 4);
    if (status) memcpy(status, h + o_st, (size_t)n * 4);
    return FMX_OK;
}

int fmx_count_batch(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n, int32_t *counts,
                    int32_t *lf_steps, int32_t *status) {
    static const bool timing = getenv("FMX_PIPE_TIMING") != nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    struct Report {
        bool on;
        std::chrono::steady_clock::time_point t0;
        ~Report() {
            if (on) fprintf(stderr, "[fmx pipe] whole call %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
    } report{timing, t_call};
    return guarded([&]() -> int {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!pat_off || !counts))) return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(idx->device));
    {
        const int pipe_min = g_host_pipeline_min;
        if (pipe_min > 0 && n >= pipe_min) {
            if (g_host_mapped.load()) {
                const int r = count_batch_mapped(idx, pat, pat_off, n, counts, lf_steps, status);
                if (r != -1) return r;
            }
            return count_batch_pipelined(idx, pat, pat_off, n, counts, lf_steps, status);
        }
    }
    rc = check_offsets(pat_off, n);
    if (rc) return rc;
    const size_t chars = (size_t)(pat_off[n] > 0 ? pat_off[n] : 0);
    if (n <= g_host_small_max.load() && chars - (size_t)first_char(pat_off) <= kHostSmallChars) {
        const int r = count_batch_small(idx, pat, pat_off, n, counts, lf_steps, status);
        if (r != -1) return r;
    }
    DevBuf d_pat, d_off, d_cnt, d_lf, d_st;
    HIP_TRY(d_pat.alloc(chars * 2 + 8));
    HIP_TRY(d_off.alloc((size_t)(n + 1) * 4));
    HIP_TRY(d_cnt.alloc((size_t)n * 4));
    HIP_TRY(d_lf.alloc((size_t)n * 4));
    HIP_TRY(d_st.alloc((size_t)n * 4));
    if (chars > first_char(pat_off)) H2D(d_pat.as<uint16_t>() + first_char(pat_off), pat + first_char(pat_off), (chars - first_char(pat_off)) * 2);
    H2D(d_off.p, pat_off, (size_t)(n + 1) * 4);
    HostCallStream hs;
    rc = hs.init(idx->device);
    if (rc) return rc;
    Scratch scratch(idx, hs.s, true);
    HostCallStream wait_first;  // (destroyed before the scratch above: the stream is drained, then the blocks go back)
    wait_first.s = hs.s;
    rc = count_impl(idx, d_pat.as<uint16_t>(), d_off.as<int32_t>(), n, d_cnt.as<int32_t>(), d_lf.as<int32_t>(),
                    d_st.as<int32_t>(), scratch);
    HIP_TRY(hipStreamSynchronize(hs.s));  // also on failure: the per-call blocks return to the cache when this call ends
    if (rc) return rc;
    D2H(counts, d_cnt.p, (size_t)n * 4);
    if (lf_steps) D2H(lf_steps, d_lf.p, (size_t)n * 4);
    if (status) D2H(status, d_st.p, (size_t)n * 4);
    return FMX_OK;
    });
}

int fmx_locate_batch(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n, int32_t max_matches,
                     int32_t *locs, int32_t loc_cap, int32_t *found, int32_t *lf_steps, int32_t *status) {
    return guarded([&]() -> int {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || loc_cap < 0 || (n > 0 && (!pat_off || !found || (!locs && loc_cap > 0))))
        return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(idx->device));
    rc = check_offsets(pat_off, n);
    if (rc) return rc;
    const size_t chars = (size_t)(pat_off[n] > 0 ? pat_off[n] : 0);
    const size_t loc_bytes = (size_t)n * (size_t)loc_cap * 4;
    // a small call (a Java caller's locate(char[], ...) is a batch of one): characters, offsets and the in / out `locations` rows in
    // ONE mapped pinned block the kernels read and write where it lies; found / LF-steps / statuses — updated with atomics — stay
    // in HBM and come down by three asynchronous copies into the block; one wait (count_batch_small has the story)
    if (n <= g_host_small_max.load() && chars - first_char(pat_off) <= kHostSmallChars && loc_bytes <= kHostSmallBytes) {
        const size_t first = first_char(pat_off), own = chars - first;
        SmallBlock blk;
        if (blk.init(SmallBlock::up(own * 2 + 8) + SmallBlock::up(((size_t)n + 1) * 4) + SmallBlock::up(loc_bytes) + 3 * SmallBlock::up((size_t)n * 4)) ==
            FMX_OK) {
            uint16_t *dp;
            int32_t *doff, *dl, *dfound, *dlf, *dst_;
            uint16_t *hp = blk.take<uint16_t>(own + 4, &dp);
            int32_t *hoff = blk.take<int32_t>((size_t)n + 1, &doff);
            int32_t *hl = blk.take<int32_t>((size_t)n * (size_t)loc_cap, &dl);
            int32_t *hfound = blk.take<int32_t>((size_t)n, &dfound), *hlf = blk.take<int32_t>((size_t)n, &dlf), *hst = blk.take<int32_t>((size_t)n, &dst_);
            (void)dfound;
            (void)dlf;
            (void)dst_;
            if (own) memcpy(hp, pat + first, own * 2);
            for (int32_t i = 0; i <= n; ++i) hoff[i] = (int32_t)((size_t)pat_off[i] - first);
            if (loc_bytes) memcpy(hl, locs, loc_bytes);
            DevBuf k_found, k_lf, k_st, k_ws;
            HIP_TRY(k_found.alloc((size_t)n * 4));
            HIP_TRY(k_lf.alloc((size_t)n * 4));
            HIP_TRY(k_st.alloc((size_t)n * 4));
            HIP_TRY(k_ws.alloc((size_t)n * 8));
            PipeStreams *ps = nullptr;
            rc = pipe_streams(idx->device, &ps);
            if (rc) return rc;
            hipStream_t st = ps->s[1];
            Scratch scratch(idx, st, true);
            struct SyncOnExit {
                hipStream_t s;
                ~SyncOnExit() { (void)hipStreamSynchronize(s); }
            } sync_on_exit{st};
            rc = locate_impl(idx, dp, doff, n, max_matches, dl, loc_cap, k_found.as<int32_t>(), lf_steps ? k_lf.as<int32_t>() : nullptr,
                             status ? k_st.as<int32_t>() : nullptr, k_ws.as<int32_t>(), scratch);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(hfound, k_found.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            if (lf_steps) HIP_TRY(hipMemcpyAsync(hlf, k_lf.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            if (status) HIP_TRY(hipMemcpyAsync(hst, k_st.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (loc_bytes) memcpy(locs, hl, loc_bytes);
            memcpy(found, hfound, (size_t)n * 4);
            if (lf_steps) memcpy(lf_steps, hlf, (size_t)n * 4);
            if (status) memcpy(status, hst, (size_t)n * 4);
            return FMX_OK;
        }
    }
    DevBuf d_pat, d_off, d_locs, d_found, d_lf, d_st, d_ws;
    // Registered `locations` / pattern arrays (fmx_host_register) are mapped into the device's address space: k_locate_walk then
    // stores the hits straight into the caller's rows — only the slots it fills travel, and they travel once (the array is in /
    // out, FM:504: unmapped it goes up and comes down whole) — and k_count reads the characters where they are.
    // found / LF-steps / statuses stay in HBM: the kernels update them with atomics.
    const bool mapped_ok = g_host_mapped.load() != 0;
    void *m_locs = mapped_ok ? mapped_range(locs, loc_bytes) : nullptr;
    void *m_pat = mapped_ok ? mapped_range(pat, chars * 2) : nullptr;
    const bool locs_mapped = m_locs != nullptr, pat_mapped = m_pat != nullptr;
    if (!pat_mapped) HIP_TRY(d_pat.alloc(chars * 2 + 8));
    HIP_TRY(d_off.alloc((size_t)(n + 1) * 4));
    if (!locs_mapped) HIP_TRY(d_locs.alloc(loc_bytes));
    HIP_TRY(d_found.alloc((size_t)n * 4));
    if (lf_steps) HIP_TRY(d_lf.alloc((size_t)n * 4));
    if (status) HIP_TRY(d_st.alloc((size_t)n * 4));
    HIP_TRY(d_ws.alloc((size_t)n * 8));
    // one stream, one wait: the small arrays go by asynchronous copies (DMA when the caller registered them, staged otherwise)
    PipeStreams *ps = nullptr;
    rc = pipe_streams(idx->device, &ps);
    if (rc) return rc;
    hipStream_t st = ps->s[1];
    Scratch scratch(idx, st, true);
    struct SyncOnExit {  // whatever leaves this function first waits for the stream, THEN the per-call blocks (declared above)
        hipStream_t s;   // return to their cache
        ~SyncOnExit() { (void)hipStreamSynchronize(s); }
    } sync_on_exit{st};
    if (chars > first_char(pat_off) && !pat_mapped)
        HIP_TRY(hipMemcpyAsync(d_pat.as<uint16_t>() + first_char(pat_off), pat + first_char(pat_off), (chars - first_char(pat_off)) * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_off.p, pat_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
    // `locations` is in/out (FM:504: the caller's array; slots beyond the hits keep the caller's values): unmapped it travels up
    // and down whole.  (Bringing the rows back through pinned staging and copying found[i] slots per row instead was measured
    // slower — 1.23 vs 1.05 ms on configs[2]: profiles/r03_experiments.txt §12.)
    if (loc_bytes && !locs_mapped) HIP_TRY(hipMemcpyAsync(d_locs.p, locs, loc_bytes, hipMemcpyHostToDevice, st));
    rc = locate_impl(idx, pat_mapped ? static_cast<const uint16_t *>(m_pat) : d_pat.as<uint16_t>(), d_off.as<int32_t>(), n, max_matches,
                     locs_mapped ? static_cast<int32_t *>(m_locs) : d_locs.as<int32_t>(), loc_cap, d_found.as<int32_t>(),
                     lf_steps ? d_lf.as<int32_t>() : nullptr, status ? d_st.as<int32_t>() : nullptr, d_ws.as<int32_t>(), scratch);
    if (rc) return rc;
    if (loc_bytes && !locs_mapped) HIP_TRY(hipMemcpyAsync(locs, d_locs.p, loc_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(found, d_found.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (lf_steps) HIP_TRY(hipMemcpyAsync(lf_steps, d_lf.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (status) HIP_TRY(hipMemcpyAsync(status, d_st.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return FMX_OK;
    });
}

// shared host-buffer driver of the two pipelines (mode < 0: fixed-length extract)
static int locate_pipeline_host(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n,
                                int32_t max_matches, int32_t row_len, uint16_t boundary, int mode, int32_t *locs,
                                int32_t *found, uint16_t *dst, int32_t *out_len, int32_t *lf_steps, int32_t *status,
                                int32_t *hit_status, int32_t *hit_aux) {
    int rc = require_device(idx);
    if (rc) return rc;
    if (!pipeline_args_ok(n, max_matches, row_len, pat_off, locs, found, out_len, pat_off) || mode > 2 ||
        (n > 0 && row_len > 0 && !dst))
        return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(idx->device));
    rc = check_offsets(pat_off, n);
    if (rc) return rc;
    const size_t chars = (size_t)(pat_off[n] > 0 ? pat_off[n] : 0);
    const size_t slots = (size_t)n * (size_t)max_matches;
    const size_t dst_bytes = slots * (size_t)row_len * 2;
    // a small call: characters, offsets, the in / out rows and per-hit arrays in ONE mapped pinned block (count_batch_small has the
    // story); found / LF-steps / statuses — updated with atomics — stay in HBM and come down by asynchronous copies into the block
    if (n <= g_host_small_max.load() && chars - first_char(pat_off) <= kHostSmallChars && dst_bytes + slots * 16 <= kHostSmallBytes) {
        const size_t first = first_char(pat_off), own = chars - first;
        SmallBlock blk;
        if (blk.init(SmallBlock::up(own * 2 + 8) + SmallBlock::up(((size_t)n + 1) * 4) + 4 * SmallBlock::up(slots * 4) + SmallBlock::up(dst_bytes) +
                     3 * SmallBlock::up((size_t)n * 4)) == FMX_OK) {
            uint16_t *dp, *ddst;
            int32_t *doff, *dl, *dlen, *dhst, *daux, *unused;
            uint16_t *hp = blk.take<uint16_t>(own + 4, &dp);
            int32_t *hoff = blk.take<int32_t>((size_t)n + 1, &doff);
            int32_t *hl = blk.take<int32_t>(slots, &dl);
            uint16_t *hdst = blk.take<uint16_t>(slots * (size_t)row_len, &ddst);
            int32_t *hlen = blk.take<int32_t>(slots, &dlen), *hhst = blk.take<int32_t>(slots, &dhst), *haux = blk.take<int32_t>(slots, &daux);
            int32_t *hfound = blk.take<int32_t>((size_t)n, &unused), *hlf = blk.take<int32_t>((size_t)n, &unused),
                    *hst = blk.take<int32_t>((size_t)n, &unused);
            if (own) memcpy(hp, pat + first, own * 2);
            for (int32_t i = 0; i <= n; ++i) hoff[i] = (int32_t)((size_t)pat_off[i] - first);
            memcpy(hl, locs, slots * 4);  // (rows and per-hit arrays are in / out: slots without a hit keep the caller's values)
            if (dst_bytes) memcpy(hdst, dst, dst_bytes);
            memcpy(hlen, out_len, slots * 4);
            if (hit_status) memcpy(hhst, hit_status, slots * 4);
            if (hit_aux) memcpy(haux, hit_aux, slots * 4);
            DevBuf k_found, k_lf, k_st, k_ws;
            HIP_TRY(k_found.alloc((size_t)n * 4));
            HIP_TRY(k_lf.alloc((size_t)n * 4));
            HIP_TRY(k_st.alloc((size_t)n * 4));
            HIP_TRY(k_ws.alloc((size_t)n * 8));
            PipeStreams *ps = nullptr;
            rc = pipe_streams(idx->device, &ps);
            if (rc) return rc;
            hipStream_t st = ps->s[1];
            Scratch scratch(idx, st, true);
            struct SyncOnExit {
                hipStream_t s;
                ~SyncOnExit() { (void)hipStreamSynchronize(s); }
            } sync_on_exit{st};
            if (mode < 0)
                rc = locate_extract_impl(idx, dp, doff, n, max_matches, row_len, dl, k_found.as<int32_t>(), ddst, dlen, k_lf.as<int32_t>(),
                                         k_st.as<int32_t>(), dhst, k_ws.as<int32_t>(), scratch);
            else
                rc = locate_lines_impl(idx, dp, doff, n, max_matches, boundary, mode, row_len, dl, k_found.as<int32_t>(), ddst, dlen,
                                       k_lf.as<int32_t>(), k_st.as<int32_t>(), dhst, daux, k_ws.as<int32_t>(), scratch);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(hfound, k_found.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(hlf, k_lf.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(hst, k_st.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            memcpy(locs, hl, slots * 4);
            memcpy(found, hfound, (size_t)n * 4);
            if (dst_bytes) memcpy(dst, hdst, dst_bytes);
            memcpy(out_len, hlen, slots * 4);
            if (lf_steps) memcpy(lf_steps, hlf, (size_t)n * 4);
            if (status) memcpy(status, hst, (size_t)n * 4);
            if (hit_status) memcpy(hit_status, hhst, slots * 4);
            if (hit_aux) memcpy(hit_aux, haux, slots * 4);
            return FMX_OK;
        }
    }
    DevBuf d_pat, d_off, d_locs, d_found, d_dst, d_len, d_lf, d_st, d_hst, d_aux, d_ws;
    HIP_TRY(d_pat.alloc(chars * 2 + 8));
    HIP_TRY(d_off.alloc((size_t)(n + 1) * 4));
    HIP_TRY(d_locs.alloc(slots * 4));
    HIP_TRY(d_found.alloc((size_t)n * 4));
    HIP_TRY(d_dst.alloc(dst_bytes));
    HIP_TRY(d_len.alloc(slots * 4));
    HIP_TRY(d_lf.alloc((size_t)n * 4));
    HIP_TRY(d_st.alloc((size_t)n * 4));
    HIP_TRY(d_hst.alloc(slots * 4));
    HIP_TRY(d_aux.alloc(slots * 4));
    HIP_TRY(d_ws.alloc((size_t)n * 8));
    // the kernels run on a stream of this thread's own (not the default stream: the plain copies around them then wait for
    // nothing but themselves), one wait before the results come down
    PipeStreams *ps = nullptr;
    rc = pipe_streams(idx->device, &ps);
    if (rc) return rc;
    hipStream_t st = ps->s[1];
    Scratch scratch(idx, st, true);
    struct SyncOnExit {  // (declared after the buffers and the scratch: whatever leaves first waits for the kernels)
        hipStream_t s;
        ~SyncOnExit() { (void)hipStreamSynchronize(s); }
    } sync_on_exit{st};
    if (chars > first_char(pat_off)) H2D(d_pat.as<uint16_t>() + first_char(pat_off), pat + first_char(pat_off), (chars - first_char(pat_off)) * 2);
    H2D(d_off.p, pat_off, (size_t)(n + 1) * 4);
    // rows and per-hit arrays are in/out like the reference's caller-owned arrays: slots without a hit keep
    // the caller's values
    H2D(d_locs.p, locs, slots * 4);
    if (dst_bytes) H2D(d_dst.p, dst, dst_bytes);
    H2D(d_len.p, out_len, slots * 4);
    if (hit_status) H2D(d_hst.p, hit_status, slots * 4);
    if (hit_aux) H2D(d_aux.p, hit_aux, slots * 4);
    if (mode < 0)
        rc = locate_extract_impl(idx, d_pat.as<uint16_t>(), d_off.as<int32_t>(), n, max_matches, row_len,
                                 d_locs.as<int32_t>(), d_found.as<int32_t>(), d_dst.as<uint16_t>(), d_len.as<int32_t>(),
                                 d_lf.as<int32_t>(), d_st.as<int32_t>(), d_hst.as<int32_t>(), d_ws.as<int32_t>(), scratch);
    else
        rc = locate_lines_impl(idx, d_pat.as<uint16_t>(), d_off.as<int32_t>(), n, max_matches, boundary, mode, row_len,
                               d_locs.as<int32_t>(), d_found.as<int32_t>(), d_dst.as<uint16_t>(), d_len.as<int32_t>(),
                               d_lf.as<int32_t>(), d_st.as<int32_t>(), d_hst.as<int32_t>(), d_aux.as<int32_t>(),
                               d_ws.as<int32_t>(), scratch);
    HIP_TRY(hipStreamSynchronize(st));
    if (rc) return rc;
    D2H(locs, d_locs.p, slots * 4);
    D2H(found, d_found.p, (size_t)n * 4);
    if (dst_bytes) D2H(dst, d_dst.p, dst_bytes);
    D2H(out_len, d_len.p, slots * 4);
    if (lf_steps) D2H(lf_steps, d_lf.p, (size_t)n * 4);
    if (status) D2H(status, d_st.p, (size_t)n * 4);
    if (hit_status) D2H(hit_status, d_hst.p, slots * 4);
    if (hit_aux) D2H(hit_aux, d_aux.p, slots * 4);
    return FMX_OK;
}

int fmx_locate_extract_batch(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n,
                             int32_t max_matches, int32_t extract_len, int32_t *locs, int32_t *found, uint16_t *dst,
                             int32_t *out_len, int32_t *lf_steps, int32_t *status, int32_t *hit_status) {
    return guarded([&]() -> int {
    return locate_pipeline_host(idx, pat, pat_off, n, max_matches, extract_len, 0, -1, locs, found, dst, out_len,
                                lf_steps, status, hit_status, nullptr);
    });
}

int fmx_locate_lines_batch(const fmx_index *idx, const uint16_t *pat, const int32_t *pat_off, int32_t n,
                           int32_t max_matches, uint16_t boundary, int mode, int32_t dst_len, int32_t *locs,
                           int32_t *found, uint16_t *dst, int32_t *out_len, int32_t *lf_steps, int32_t *status,
                           int32_t *hit_status, int32_t *hit_aux) {
    return guarded([&]() -> int {
    if (mode < 0) return fail(FMX_E_ARG, "bad arguments");
    return locate_pipeline_host(idx, pat, pat_off, n, max_matches, dst_len, boundary, mode, locs, found, dst, out_len,
                                lf_steps, status, hit_status, hit_aux);
    });
}

// Host-buffer extract / extractUntilBoundary: destination rows are in/out (entries the walk does not write keep the caller's
// values — FM:564, FM:640: the caller's array), so they travel up AND down whole, and they are what such a call costs (100,000
// rows of 1,024 chars: 195 MB each way).  The queries go in a few chunks over three streams — rows of chunk c + 1 on their way
// up, the kernels of chunk c, rows of chunk c - 1 on their way down — with asynchronous copies (DMA at the link's rate for
// arrays the caller registered, staged otherwise) and ONE wait at the end.  launch(lo, n_c, stream): the chunk's kernels.
extern "C++" {
struct HostColumn {  // one int32 per query
    const void *in;  // caller's array to ship up (nullptr: none)
    void *out;       // caller's array to bring down (nullptr: none)
    DevBuf *dev;
};
template <class Launch>
static int rows_pipeline_host(const fmx_index *idx, int32_t n, uint16_t *dst, int32_t dst_len, DevBuf &d_dst,
                              std::initializer_list<HostColumn> columns, Launch &&launch) {
    PipeStreams *ps = nullptr;
    int rc = pipe_streams(idx->device, &ps);
    if (rc) return rc;
    hipStream_t s_in = ps->s[0], s_k = ps->s[1], s_out = ps->s[2];
    struct DrainOnExit {  // (declared after the caller's buffers: whatever leaves first waits for the streams)
        PipeStreams *ps;
        ~DrainOnExit() {
            for (int i = 0; i < kPipeStreams; ++i) (void)hipStreamSynchronize(ps->s[i]);
        }
    } drain_on_exit{ps};
    const size_t row_bytes = (size_t)dst_len * 2;
    if (row_bytes && !is_pinned_range(dst, (size_t)n * row_bytes)) {
        // pageable rows: the plain copies (from pageable memory they run at twice the rate of the asynchronous ones), one launch
        for (const HostColumn &col : columns)
            if (col.in) HIP_TRY(hipMemcpy(col.dev->p, col.in, (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_dst.p, dst, (size_t)n * row_bytes, hipMemcpyHostToDevice));
        rc = launch(0, n, s_k);
        HIP_TRY(hipStreamSynchronize(s_k));
        if (rc) return rc;
        HIP_TRY(hipMemcpy(dst, d_dst.p, (size_t)n * row_bytes, hipMemcpyDeviceToHost));
        for (const HostColumn &col : columns)
            if (col.out) HIP_TRY(hipMemcpy(col.out, col.dev->p, (size_t)n * 4, hipMemcpyDeviceToHost));
        return FMX_OK;
    }
    // (registered rows brought DOWN by a copy kernel into the mapped array, so that the two directions would not share the DMA
    // engine: 10.8 ms against 10.2 with DMA both ways for 2 x 195 MB — what limits is the rate of DMA to and from REGISTERED
    // pageable memory, scattered 4 KB pages; profiles/r04_experiments.txt 16)
    // chunks of at least 4 MB of rows and 8,192 queries, at most 8 of them
    int64_t chunk = ((int64_t)n + 7) / 8;
    const int64_t min_rows = row_bytes ? (int64_t)(((size_t)4 << 20) / row_bytes) + 1 : n;
    chunk = std::max<int64_t>(chunk, std::max<int64_t>(8192, min_rows));
    int c = 0;
    for (int64_t lo = 0; lo < n; lo += chunk, ++c) {
        const int32_t n_c = (int32_t)std::min<int64_t>(chunk, (int64_t)n - lo);
        for (const HostColumn &col : columns)
            if (col.in)
                HIP_TRY(hipMemcpyAsync(col.dev->as<int32_t>() + lo, static_cast<const int32_t *>(col.in) + lo, (size_t)n_c * 4,
                                       hipMemcpyHostToDevice, s_in));
        if (row_bytes)
            HIP_TRY(hipMemcpyAsync(d_dst.as<uint16_t>() + lo * dst_len, dst + lo * dst_len, (size_t)n_c * row_bytes, hipMemcpyHostToDevice,
                                   s_in));
        HIP_TRY(hipEventRecord(ps->in[c], s_in));
        HIP_TRY(hipStreamWaitEvent(s_k, ps->in[c], 0));
        rc = launch((int32_t)lo, n_c, s_k);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ps->counted[c], s_k));
        HIP_TRY(hipStreamWaitEvent(s_out, ps->counted[c], 0));
        if (row_bytes)
            HIP_TRY(hipMemcpyAsync(dst + lo * dst_len, d_dst.as<uint16_t>() + lo * dst_len, (size_t)n_c * row_bytes, hipMemcpyDeviceToHost,
                                   s_out));
        for (const HostColumn &col : columns)
            if (col.out)
                HIP_TRY(hipMemcpyAsync(static_cast<int32_t *>(col.out) + lo, col.dev->as<int32_t>() + lo, (size_t)n_c * 4,
                                       hipMemcpyDeviceToHost, s_out));
    }
    for (int i = 0; i < kPipeStreams; ++i) HIP_TRY(hipStreamSynchronize(ps->s[i]));
    return FMX_OK;
}
}  // extern "C++"

int fmx_extract_batch(const fmx_index *idx, const int32_t *start, const int32_t *stop, int32_t n, uint16_t *dst,
                      int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf_steps, int32_t *status) {
    return guarded([&]() -> int {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || dst_len < 0 || (n > 0 && (!start || !stop || !out_len || (!dst && dst_len > 0))))
        return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(idx->device));
    const size_t dst_bytes = (size_t)n * (size_t)dst_len * 2;
    if (n <= g_host_small_max.load() && dst_bytes <= kHostSmallBytes) {  // a small call: one mapped pinned block (count_batch_small)
        SmallBlock blk;
        if (blk.init(5 * SmallBlock::up((size_t)n * 4) + SmallBlock::up(dst_bytes)) == FMX_OK) {
            int32_t *da, *db, *dlen, *dlf, *dst_;
            uint16_t *ddst;
            int32_t *ha = blk.take<int32_t>((size_t)n, &da), *hb = blk.take<int32_t>((size_t)n, &db);
            uint16_t *hdst = blk.take<uint16_t>((size_t)n * (size_t)dst_len, &ddst);
            int32_t *hlen = blk.take<int32_t>((size_t)n, &dlen), *hlf = blk.take<int32_t>((size_t)n, &dlf), *hst = blk.take<int32_t>((size_t)n, &dst_);
            memcpy(ha, start, (size_t)n * 4);
            memcpy(hb, stop, (size_t)n * 4);
            if (dst_bytes) memcpy(hdst, dst, dst_bytes);  // (rows are in / out: what a query does not write keeps the caller's values)
            HostCallStream hs;
            rc = hs.init(idx->device);
            if (rc) return rc;
            rc = fmx_extract_batch_dev(idx, da, db, n, ddst, dst_len, offset, dlen, dlf, dst_, hs.s);
            HIP_TRY(hipStreamSynchronize(hs.s));
            if (rc) return rc;
            if (dst_bytes) memcpy(dst, hdst, dst_bytes);
            memcpy(out_len, hlen, (size_t)n * 4);
            if (lf_steps) memcpy(lf_steps, hlf, (size_t)n * 4);
            if (status) memcpy(status, hst, (size_t)n * 4);
            return FMX_OK;
        }
    }
    DevBuf d_a, d_b, d_dst, d_len, d_lf, d_st;
    HIP_TRY(d_a.alloc((size_t)n * 4));
    HIP_TRY(d_b.alloc((size_t)n * 4));
    HIP_TRY(d_dst.alloc(dst_bytes));
    HIP_TRY(d_len.alloc((size_t)n * 4));
    HIP_TRY(d_lf.alloc((size_t)n * 4));
    HIP_TRY(d_st.alloc((size_t)n * 4));
    return rows_pipeline_host(idx, n, dst, dst_len, d_dst,
                              {{start, nullptr, &d_a}, {stop, nullptr, &d_b}, {nullptr, out_len, &d_len}, {nullptr, lf_steps, &d_lf},
                               {nullptr, status, &d_st}},
                              [&](int32_t lo, int32_t n_c, hipStream_t st) {
                                  return fmx_extract_batch_dev(idx, d_a.as<int32_t>() + lo, d_b.as<int32_t>() + lo, n_c,
                                                               d_dst.as<uint16_t>() + (int64_t)lo * dst_len, dst_len, offset,
                                                               d_len.as<int32_t>() + lo, d_lf.as<int32_t>() + lo, d_st.as<int32_t>() + lo, st);
                              });
    });
}

int fmx_extract_boundary_batch(const fmx_index *idx, const int32_t *from, int32_t n, uint16_t boundary, int mode,
                               uint16_t *dst, int32_t dst_len, int32_t offset, int32_t *out_len, int32_t *lf_steps,
                               int32_t *status, int32_t *aux) {
    return guarded([&]() -> int {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || dst_len < 0 || mode < 0 || mode > 2 || (n > 0 && (!from || !out_len || (!dst && dst_len > 0))))
        return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(idx->device));
    const size_t dst_bytes = (size_t)n * (size_t)dst_len * 2;
    if (n <= g_host_small_max.load() && dst_bytes <= kHostSmallBytes) {  // a small call: one mapped pinned block (count_batch_small)
        SmallBlock blk;
        if (blk.init(5 * SmallBlock::up((size_t)n * 4) + SmallBlock::up(dst_bytes)) == FMX_OK) {
            int32_t *da, *dlen, *dlf, *dst_, *daux;
            uint16_t *ddst;
            int32_t *ha = blk.take<int32_t>((size_t)n, &da);
            uint16_t *hdst = blk.take<uint16_t>((size_t)n * (size_t)dst_len, &ddst);
            int32_t *hlen = blk.take<int32_t>((size_t)n, &dlen), *hlf = blk.take<int32_t>((size_t)n, &dlf), *hst = blk.take<int32_t>((size_t)n, &dst_),
                    *haux = blk.take<int32_t>((size_t)n, &daux);
            memcpy(ha, from, (size_t)n * 4);
            if (dst_bytes) memcpy(hdst, dst, dst_bytes);
            if (aux) memcpy(haux, aux, (size_t)n * 4);  // (in / out like the rows: only a query that does not fit writes it)
            HostCallStream hs;
            rc = hs.init(idx->device);
            if (rc) return rc;
            Scratch scratch(idx, hs.s, true);
            HostCallStream wait_first;  // (drained before the scratch and the block go back)
            wait_first.s = hs.s;
            rc = boundary_impl(idx, da, n, boundary, mode, ddst, dst_len, offset, dlen, dlf, dst_, daux, nullptr, 0, scratch);
            HIP_TRY(hipStreamSynchronize(hs.s));
            if (rc) return rc;
            if (dst_bytes) memcpy(dst, hdst, dst_bytes);
            memcpy(out_len, hlen, (size_t)n * 4);
            if (lf_steps) memcpy(lf_steps, hlf, (size_t)n * 4);
            if (status) memcpy(status, hst, (size_t)n * 4);
            if (aux) memcpy(aux, haux, (size_t)n * 4);
            return FMX_OK;
        }
    }
    DevBuf d_a, d_dst, d_len, d_lf, d_st, d_aux;
    HIP_TRY(d_a.alloc((size_t)n * 4));
    HIP_TRY(d_dst.alloc(dst_bytes));
    HIP_TRY(d_len.alloc((size_t)n * 4));
    HIP_TRY(d_lf.alloc((size_t)n * 4));
    HIP_TRY(d_st.alloc((size_t)n * 4));
    HIP_TRY(d_aux.alloc((size_t)n * 4));
    std::vector<std::unique_ptr<Scratch>> scratches;  // (a chunk's window scratch lives until the call has drained)
    return rows_pipeline_host(idx, n, dst, dst_len, d_dst,
                              {{from, nullptr, &d_a}, {nullptr, out_len, &d_len}, {nullptr, lf_steps, &d_lf}, {nullptr, status, &d_st},
                               {nullptr, aux, &d_aux}},
                              [&](int32_t lo, int32_t n_c, hipStream_t st) {
                                  scratches.emplace_back(new Scratch(idx, st, true));
                                  return boundary_impl(idx, d_a.as<int32_t>() + lo, n_c, boundary, mode,
                                                       d_dst.as<uint16_t>() + (int64_t)lo * dst_len, dst_len, offset, d_len.as<int32_t>() + lo,
                                                       d_lf.as<int32_t>() + lo, d_st.as<int32_t>() + lo, d_aux.as<int32_t>() + lo, nullptr, 0,
                                                       *scratches.back());
                              });
    });
}

// ---- WaveletFixedBlockBoosting as a stand-alone structure ------------------------------------------------

int fmx_wavelet_build(const int16_t *sequence, int64_t n, int32_t sampling_rate, fmx_index **out) {
    return guarded([&]() -> int {
    if (!out || !sequence || n <= 0 || n >= ((int64_t)1 << 31) || sampling_rate <= 0)
        return fail(FMX_E_ARG, n == 0 ? "Input length must be > 0" : "bad arguments");  // WFBB:178-180
    for (int64_t i = 0; i < n; ++i)
        if (sequence[i] < 0) return fail(FMX_E_ARG, "negative symbol");
    std::unique_ptr<fmx_index> idx(new fmx_index());
    fmx::FmModel &m = idx->model;
    m.sample_rate = sampling_rate;
    m.enable_extract = false;
    m.length = (int32_t)n;
    m.bw_suffixes = 1;
    m.C.assign(1, 0);
    m.look_up.assign(1, 0);
    m.suffixes.init(0, 1);
    fmx::build_rrr(nullptr, 0, sampling_rate, m.sampled);
    fmx::build_wavelet(sequence, n, sampling_rate, m.wt);
    idx->has_model = true;
    idx->wavelet_only = true;
    *out = idx.release();
    return FMX_OK;
    });
}

// ---- RrrVector as a stand-alone structure (the reference's public class sdsl/RrrVector.java) ----
int fmx_rrr_build(const uint8_t *bits, int64_t n, int32_t sample_size, fmx_index **out) {
    return guarded([&]() -> int {
    if (!out || (!bits && n > 0) || n < 0 || n >= ((int64_t)1 << 31) || sample_size <= 0) return fail(FMX_E_ARG, "bad arguments");
    std::vector<uint64_t> words((size_t)(n / 64 + 2), 0);
    for (int64_t i = 0; i < n; ++i)
        if (bits[i]) words[(size_t)(i >> 6)] |= 1ull << (i & 63);
    std::unique_ptr<fmx_index> idx(new fmx_index());
    fmx::build_rrr(words.data(), n, sample_size, idx->model.sampled);  // RRR:225-286
    idx->model.length = (int32_t)n;
    idx->has_model = true;
    idx->rrr_only = true;
    *out = idx.release();
    return FMX_OK;
    });
}

int fmx_rrr_rank_ones_batch(const fmx_index *idx, const int32_t *positions, int32_t n, int32_t *ranks) {
    return guarded([&]() -> int {
    int rc = require_device(idx, true);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!positions || !ranks))) return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(idx->device));
    DevBuf d_pos, d_out;
    HIP_TRY(d_pos.alloc((size_t)n * 4));
    HIP_TRY(d_out.alloc((size_t)n * 4));
    H2D(d_pos.p, positions, (size_t)n * 4);
    int e = fmx::launch_rrr_rank_ones(idx->dev, idx->n_cu, d_pos.as<int32_t>(), n, d_out.as<int32_t>(), nullptr);
    if (e) return fail(FMX_E_HIP, std::string("k_rrr_rank_ones launch: ") + hipGetErrorString((hipError_t)e));
    HIP_TRY(hipDeviceSynchronize());
    D2H(ranks, d_out.p, (size_t)n * 4);
    return FMX_OK;
    });
}

int fmx_rrr_rank_ones_batch_dev(const fmx_index *idx, const int32_t *d_positions, int32_t n, int32_t *d_ranks, void *stream) {
    return guarded([&]() -> int {
    int rc = require_device(idx, true);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!d_positions || !d_ranks))) return fail(FMX_E_ARG, "bad arguments");
    int e = fmx::launch_rrr_rank_ones(idx->dev, idx->n_cu, d_positions, n, d_ranks, static_cast<hipStream_t>(stream));
    if (e) return fail(FMX_E_HIP, std::string("k_rrr_rank_ones launch: ") + hipGetErrorString((hipError_t)e));
    return FMX_OK;
    });
}

int fmx_rrr_access_batch_dev(const fmx_index *idx, const int32_t *d_positions, int32_t n, uint8_t *d_bits, int32_t *d_status,
                             void *stream) {
    return guarded([&]() -> int {
    int rc = require_device(idx, true);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!d_positions || !d_bits))) return fail(FMX_E_ARG, "bad arguments");
    int e = fmx::launch_rrr_access(idx->dev, idx->n_cu, d_positions, n, d_bits, d_status, static_cast<hipStream_t>(stream));
    if (e) return fail(FMX_E_HIP, std::string("k_rrr_access launch: ") + hipGetErrorString((hipError_t)e));
    return FMX_OK;
    });
}

int fmx_rrr_access_batch(const fmx_index *idx, const int32_t *positions, int32_t n, uint8_t *bits, int32_t *status) {
    return guarded([&]() -> int {
    int rc = require_device(idx, true);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!positions || !bits))) return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(idx->device));
    DevBuf d_pos, d_out, d_st;
    HIP_TRY(d_pos.alloc((size_t)n * 4));
    HIP_TRY(d_out.alloc((size_t)n));
    HIP_TRY(d_st.alloc((size_t)n * 4));
    H2D(d_pos.p, positions, (size_t)n * 4);
    int e = fmx::launch_rrr_access(idx->dev, idx->n_cu, d_pos.as<int32_t>(), n, d_out.as<uint8_t>(), d_st.as<int32_t>(),
                                   nullptr);
    if (e) return fail(FMX_E_HIP, std::string("k_rrr_access launch: ") + hipGetErrorString((hipError_t)e));
    HIP_TRY(hipDeviceSynchronize());
    D2H(bits, d_out.p, (size_t)n);
    if (status) D2H(status, d_st.p, (size_t)n * 4);
    return FMX_OK;
    });
}

static int wavelet_batch(const fmx_index *idx, const int64_t *positions, const int32_t *symbols, int32_t n, int64_t *out,
                         int32_t *status) {
    int rc = require_device(idx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!positions || !out))) return fail(FMX_E_ARG, "bad arguments");
    if (n == 0) return FMX_OK;
    HIP_TRY(hipSetDevice(idx->device));
    DevBuf d_pos, d_sym, d_out, d_st;
    HIP_TRY(d_pos.alloc((size_t)n * 8));
    HIP_TRY(d_sym.alloc((size_t)n * 4));
    HIP_TRY(d_out.alloc((size_t)n * 8));
    HIP_TRY(d_st.alloc((size_t)n * 4));
    H2D(d_pos.p, positions, (size_t)n * 8);
    int e;
    if (symbols) {
        H2D(d_sym.p, symbols, (size_t)n * 4);
        e = k_launch_wt_rank(idx, idx->dev, idx->n_cu, d_pos.as<int64_t>(), d_sym.as<int32_t>(), n, d_out.as<int64_t>(),
                                d_st.as<int32_t>(), nullptr);
    } else {
        e = k_launch_wt_inverse_select(idx, idx->dev, idx->n_cu, d_pos.as<int64_t>(), n, d_out.as<int64_t>(),
                                          d_st.as<int32_t>(), nullptr);
    }
    if (e) return fail(FMX_E_HIP, std::string("wavelet kernel launch: ") + hipGetErrorString((hipError_t)e));
    HIP_TRY(hipDeviceSynchronize());
    D2H(out, d_out.p, (size_t)n * 8);
    if (status) D2H(status, d_st.p, (size_t)n * 4);
    return FMX_OK;
}

int fmx_wavelet_rank_batch(const fmx_index *idx, const int64_t *positions, const int32_t *symbols, int32_t n,
                           int64_t *ranks, int32_t *status) {
    return guarded([&]() -> int {
    if (n > 0 && !symbols) return fail(FMX_E_ARG, "bad arguments");
    return wavelet_batch(idx, positions, symbols, n, ranks, status);
    });
}

int fmx_wavelet_inverse_select_batch(const fmx_index *idx, const int64_t *positions, int32_t n, int64_t *packed,
                                     int32_t *status) {
    return guarded([&]() -> int {
    return wavelet_batch(idx, positions, nullptr, n, packed, status);
    });
}

// ---- helpers ---------------------------------------------------------------------------------------

// FM:239-298 (Java `byte` is signed: the masks below restate the reference's expressions)
int fmx_convert_byte_pattern(const uint8_t *pattern, int32_t offset, int32_t length, uint16_t *dest,
                             int32_t *bad_value) {
    return guarded([&]() -> int {
    int pos = offset, i = 0;
    while (pos < length + offset) {
        const int first = (int8_t)pattern[pos];
        uint16_t next;
        if (first < 0) {
            if ((((uint32_t)(first & 0xF0)) >> 3) == 30) {  // 4-byte form, FM:248-267
                const int b2 = (int8_t)pattern[pos + 1], b3 = (int8_t)pattern[pos + 2], b4 = (int8_t)pattern[pos + 3];
                pos += 4;
                const int before = (((first & 0x07) << 18) | ((b2 & 0x3F) << 12) | ((b3 & 0x3F) << 6) | (b4 & 0x3F)) & 0x1FFFFF;
                if (before > 32767) {
                    if (bad_value) *bad_value = before;
                    return -1;
                }
                next = (uint16_t)before;
            } else if ((((uint32_t)(first & 0xE0)) >> 4) == 14) {  // 3-byte form, FM:269-279
                const int b2 = (int8_t)pattern[pos + 1], b3 = (int8_t)pattern[pos + 2];
                pos += 3;
                next = (uint16_t)((((first & 0x0F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F)) & 0xFFFF);
            } else {  // 2-byte form, FM:280-288
                const int b2 = (int8_t)pattern[pos + 1];
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
    });
}

const char *fmx_status_message(int status) {
    switch (status) {
        case FMX_ST_OK: return "";
        case FMX_ST_NOT_ENABLED: return "Text recovery not enabled at build time";
        case FMX_ST_POS_NEGATIVE: return "Requested position less than 0";
        case FMX_ST_STOP_TOO_LONG: return "Stop position longer than index string";
        case FMX_ST_DEST_TOO_SMALL: return "Supplied destination is not large enough";
        case FMX_ST_POS_TOO_LONG: return "Requested position longer than index string";
        case FMX_ST_DEST_SIZE_ZERO: return "Supplied destination for extraction has size zero";
        case FMX_ST_NO_BOUNDARY: return "Boundary does not exist";
        case FMX_ST_DOES_NOT_FIT: return "Extraction does not fit in the supplied destination. Currently extracted: %d";
        case FMX_ST_JAVA_AIOOBE: return "ArrayIndexOutOfBoundsException";
        default: return "unknown status";
    }
}

int fmx_status_kind(int status) {
    return guarded([&]() -> int {
    if (status == FMX_ST_DEST_SIZE_ZERO || status == FMX_ST_NO_BOUNDARY) return 1;
    if (status == FMX_ST_JAVA_AIOOBE) return 2;
    return 0;
    });
}

}  // extern "C"
