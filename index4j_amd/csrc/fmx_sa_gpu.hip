// Index construction, device stage: suffix array of the mapped text by prefix doubling, then the arrays
// FmIndex derives from it (FM:329-394) — BWT, the sampled-row bitmap, the sampled suffix values in row order
// and the inverse samples — all computed in HBM; only those results travel back to the host, where the
// wavelet tree is encoded (fmx_build.cpp).
//
// The suffix array of a text with a unique smallest terminator is unique, so this stage and the host's SA-IS
// (fmx_sais.hpp) produce the same arrays and the resulting index is byte-identical (tests/test_gpu_parity.py).
//
// Prefix doubling (Manber-Myers / Larsson-Sadakane, in the sort-based form that suits a GPU):
//   round 0 : sort suffixes by their first c codes (one key of <= 64 bits: c = 9 for a log's ~70 codes), rank = index
//             of the group's first row
//   round k : key = (rank[i], rank[i+h]) packed in 2*ceil(log2(L+1)) bits, h = c, 2c, 4c, ...; only rows of
//             groups that are still tied take part (the rest of SA is final and stays in place)
//   until every group is a single row.
// Device-wide sort / scan / select come from rocPRIM; the kernels here build keys, mark group heads and
// scatter ranks — streaming passes over 4- and 8-byte arrays, HBM-bound.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "fmx_build_stage.hpp"
#include "fmx_model.hpp"

namespace fmx {
namespace {

constexpr int kThreads = 256;
inline unsigned grid_of(int64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

#define SA_TRY(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            return -6;                                                                       \
        }                                                                                    \
    } while (0)

struct DevMem {
    std::vector<void *> ptrs;
    ~DevMem() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T>
    hipError_t alloc(T **out, size_t count) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T *>(p);
        return e;
    }
    void release(void *p) {  // ahead of the destructor
        for (void *&q : ptrs)
            if (q == p && p) {
                (void)hipFree(p);
                q = nullptr;
            }
    }
};

// round 0: the first `chars` codes of every suffix as one key, `bits` bits each (4 x 16 when the alphabet size is
// not known, else as many codes as fit 64 bits: 9 for a log's ~70 codes — the first round then settles nine
// characters and fewer rows stay tied).  Past the end = 0, which only the terminator carries, so keys that reach
// the end are unique anyway.
__global__ void k_sa_first_keys(const int16_t *__restrict__ seq, uint32_t L, int chars, int bits,
                                uint64_t *__restrict__ keys, uint32_t *__restrict__ sa) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= L) return;
    uint64_t k = 0;
    for (int j = 0; j < chars; ++j) k = (k << bits) | ((uint64_t)i + j < L ? (uint64_t)(uint16_t)seq[i + j] : 0ull);
    keys[i] = k;
    sa[i] = i;
}

// head[j] = j if row j starts a new group of equal keys, else 0 (an inclusive max-scan turns this into the
// index of the group's first row); `row_of` maps a slot of the compacted active list to its SA row (nullptr:
// identity).  A group boundary also lies wherever two neighbouring slots are not neighbouring rows.
__global__ void k_sa_mark_heads(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ row_of, uint32_t n,
                                uint32_t *__restrict__ head) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= n) return;
    const uint32_t row = row_of ? row_of[j] : j;
    bool first = j == 0 || keys[j] != keys[j - 1];
    if (!first && row_of) first = row_of[j - 1] + 1 != row;
    head[j] = first ? row : 0u;
}

// after the scan: rank of suffix sa[j] = first row of its group; rows whose group has a single member are final
__global__ void k_sa_assign_ranks(const uint32_t *__restrict__ sa_sorted, const uint32_t *__restrict__ row_of,
                                  const uint32_t *__restrict__ head, uint32_t n, uint32_t *__restrict__ rank,
                                  uint32_t *__restrict__ sa_full) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= n) return;
    const uint32_t row = row_of ? row_of[j] : j;
    const uint32_t suffix = sa_sorted[j];
    rank[suffix] = head[j];
    if (row_of) sa_full[row] = suffix;  // the sorted active rows go back into their places
}

// a row is still tied if its group (rows with the same rank) has another member: look at both neighbours
__global__ void k_sa_mark_active(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ rank, uint32_t L,
                                 uint8_t *__restrict__ active) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= L) return;
    const uint32_t r = rank[sa[j]];
    const bool tied = (j > 0 && rank[sa[j - 1]] == r) || (j + 1 < L && rank[sa[j + 1]] == r);
    active[j] = tied ? 1 : 0;
}

// the same test over the rows that were tied in the previous round only (a row that was settled stays settled)
__global__ void k_sa_mark_active_rows(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ rank, uint32_t L,
                                      const uint32_t *__restrict__ rows, uint32_t n, uint8_t *__restrict__ active) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = rows[i];
    const uint32_t r = rank[sa[j]];
    const bool tied = (j > 0 && rank[sa[j - 1]] == r) || (j + 1 < L && rank[sa[j + 1]] == r);
    active[i] = tied ? 1 : 0;
}

// key of an active row for the next round: (rank[i], rank[i + h] + 1 or 0 past the end), packed
__global__ void k_sa_next_keys(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ row_of,
                               const uint32_t *__restrict__ rank, uint32_t n, uint32_t L, uint32_t h, int low_bits,
                               uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= n) return;
    const uint32_t suffix = sa[row_of[j]];
    const uint64_t second = (uint64_t)suffix + h < L ? (uint64_t)rank[suffix + h] + 1ull : 0ull;
    keys[j] = ((uint64_t)rank[suffix] << low_bits) | second;
    vals[j] = suffix;
}

// FM:341-372 + FM:385-392 from the finished suffix array.  `which` gets one 64-bit word per wave through a
// ballot (bit j&63 of word j>>6 = row j is sampled), `flag` feeds the scan that numbers the sampled rows.
__global__ void k_sa_outputs(const uint32_t *__restrict__ sa, const int16_t *__restrict__ seq, uint32_t L,
                             uint32_t sample_rate, int16_t *__restrict__ bwt, uint64_t *__restrict__ which,
                             uint32_t *__restrict__ flag, uint32_t *__restrict__ position_vals) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    const bool in = j < L;
    const uint32_t s = in ? sa[j] : 1u;
    const bool sampled = in && (s % sample_rate == 0);
    const uint64_t word = __ballot(sampled);
    if (in) {
        bwt[j] = s == 0 ? seq[L - 1] : seq[s - 1];
        flag[j] = sampled ? 1u : 0u;
        if ((j & 63u) == 0) which[j >> 6] = word;
        if (sampled && position_vals) position_vals[s / sample_rate] = j;
    }
}
__global__ void k_sa_compact_samples(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ flag,
                                     const uint32_t *__restrict__ slot, uint32_t L, uint32_t *__restrict__ suffix_vals) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j < L && flag[j]) suffix_vals[slot[j] - 1] = sa[j];
}

// FM:396-435, first half: first position and count of every character value.  Characters below kLowChars (all of a
// typical log) go through LDS tables flushed once per workgroup; higher ones (CJK text ...) through a small hashed LDS
// table with tags — a character whose slot is taken by another goes straight to the global tables.
constexpr int kLowChars = 4096;
constexpr int kHighSlots = 2048;
__global__ void k_text_stats(const uint16_t *__restrict__ text, uint32_t n, uint32_t per_group,
                             uint32_t *__restrict__ first, uint32_t *__restrict__ count) {
    __shared__ uint32_t s_first[kLowChars + kHighSlots], s_count[kLowChars + kHighSlots], s_tag[kHighSlots];
    for (int i = threadIdx.x; i < kLowChars + kHighSlots; i += blockDim.x) {
        s_first[i] = 0xffffffffu;
        s_count[i] = 0;
    }
    for (int i = threadIdx.x; i < kHighSlots; i += blockDim.x) s_tag[i] = 0xffffffffu;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * per_group;
    const uint32_t hi = (uint32_t)(lo + per_group < n ? lo + per_group : n);
    for (uint32_t i = (uint32_t)lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t ch = text[i];
        int slot = -1;
        if (ch < (uint32_t)kLowChars) {
            slot = (int)ch;
        } else {
            const uint32_t h = (ch * 2654435761u) >> 21;  // 11 bits
            const uint32_t old = atomicCAS(&s_tag[h], 0xffffffffu, ch);
            if (old == 0xffffffffu || old == ch) slot = kLowChars + (int)h;
        }
        if (slot >= 0) {
            atomicMin(&s_first[slot], i);
            atomicAdd(&s_count[slot], 1u);
        } else {
            atomicMin(&first[ch], i);
            atomicAdd(&count[ch], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kLowChars + kHighSlots; i += blockDim.x)
        if (s_count[i]) {
            const uint32_t ch = i < kLowChars ? (uint32_t)i : s_tag[i - kLowChars];
            atomicMin(&first[ch], s_first[i]);
            atomicAdd(&count[ch], s_count[i]);
        }
}
// FM:427-433: characters -> codes, in place; the appended terminator gets code 0
__global__ void k_text_map(uint16_t *__restrict__ text, uint32_t L, const int16_t *__restrict__ code_of) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= L) return;
    text[i] = i + 1 == L ? (uint16_t)0 : (uint16_t)code_of[text[i]];
}

int ceil_log2(uint64_t v) {
    int b = 0;
    while ((1ull << b) < v) ++b;
    return b;
}

}  // namespace

void device_release(void *d_ptr) {
    if (d_ptr) (void)hipFree(d_ptr);
}

int device_alphabet_stage(const uint16_t *input, int32_t n_in, int device, std::vector<int32_t> &first,
                          std::vector<int64_t> &count, void **d_text, std::string &err) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        err = "no HIP device visible";
        return -5;
    }
    if (device < 0 || device >= n_dev || n_in < 0) {
        err = "bad arguments";
        return -1;
    }
    SA_TRY(hipSetDevice(device));
    uint16_t *d_raw = nullptr;
    uint32_t *d_tab = nullptr;
    SA_TRY(hipMalloc((void **)&d_raw, ((size_t)n_in + 1) * 2));
    struct Guard {
        void *p;
        ~Guard() {
            if (p) (void)hipFree(p);
        }
    } g_raw{d_raw}, g_tab{nullptr};
    SA_TRY(hipMalloc((void **)&d_tab, 2 * 65536 * 4));
    g_tab.p = d_tab;
    if (n_in) SA_TRY(hipMemcpy(d_raw, input, (size_t)n_in * 2, hipMemcpyHostToDevice));
    SA_TRY(hipMemset(d_tab, 0xff, 65536 * 4));
    SA_TRY(hipMemset(d_tab + 65536, 0, 65536 * 4));
    if (n_in) {
        const uint32_t per_group = 1u << 16;
        hipLaunchKernelGGL(k_text_stats, dim3((unsigned)(((int64_t)n_in + per_group - 1) / per_group)), dim3(1024), 0, 0, d_raw,
                           (uint32_t)n_in, per_group, d_tab, d_tab + 65536);
        SA_TRY(hipGetLastError());
    }
    std::vector<uint32_t> tab(2 * 65536);
    SA_TRY(hipMemcpy(tab.data(), d_tab, tab.size() * 4, hipMemcpyDeviceToHost));
    first.assign(65536, -1);
    count.assign(65536, 0);
    for (int ch = 0; ch < 65536; ++ch)
        if (tab[65536 + (size_t)ch]) {
            first[(size_t)ch] = (int32_t)tab[(size_t)ch];
            count[(size_t)ch] = tab[65536 + (size_t)ch];
        }
    *d_text = d_raw;
    g_raw.p = nullptr;  // handed to device_sa_stage
    return 0;
}

int device_sa_stage(const int16_t *seq, int32_t n, int sample_rate, bool extract, int device, SaStage &out,
                    SaStageStats *stats, std::string &err, WfbbModel *wt, int alphabet, void *d_text,
                    const int16_t *code_of) {
    struct TextGuard {  // the device copy of the text is this stage's to free, whatever happens
        void *p;
        ~TextGuard() {
            if (p) (void)hipFree(p);
        }
    } text_guard{d_text};
    if (n <= 0 || sample_rate <= 0) {
        err = "bad arguments";
        return -1;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        err = "no HIP device visible";
        return -5;
    }
    if (device < 0 || device >= n_dev) {
        err = "device ordinal out of range";
        return -1;
    }
    SA_TRY(hipSetDevice(device));
    const auto t_begin = std::chrono::steady_clock::now();
    const uint32_t L = (uint32_t)n;
    DevMem mem;
    int16_t *d_seq = nullptr;
    uint64_t *d_keys = nullptr, *d_keys_alt = nullptr;
    uint32_t *d_sa = nullptr, *d_vals = nullptr, *d_vals_alt = nullptr, *d_rank = nullptr, *d_head = nullptr,
             *d_rows = nullptr, *d_rows_alt = nullptr, *d_count = nullptr;
    uint8_t *d_active = nullptr;
    int16_t *d_codes = nullptr;
    if (d_text) {
        d_seq = static_cast<int16_t *>(d_text);  // mapped in place below
        SA_TRY(mem.alloc(&d_codes, 65536));
        SA_TRY(hipMemcpy(d_codes, code_of, 65536 * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_text_map, dim3(grid_of(L)), dim3(kThreads), 0, 0, static_cast<uint16_t *>(d_text), L, d_codes);
    } else {
        SA_TRY(mem.alloc(&d_seq, (size_t)L));
    }
    SA_TRY(mem.alloc(&d_keys, (size_t)L));
    SA_TRY(mem.alloc(&d_keys_alt, (size_t)L));
    SA_TRY(mem.alloc(&d_sa, (size_t)L));
    SA_TRY(mem.alloc(&d_vals, (size_t)L));
    SA_TRY(mem.alloc(&d_vals_alt, (size_t)L));
    SA_TRY(mem.alloc(&d_rank, (size_t)L));
    SA_TRY(mem.alloc(&d_head, (size_t)L));
    SA_TRY(mem.alloc(&d_rows, (size_t)L));
    SA_TRY(mem.alloc(&d_rows_alt, (size_t)L));
    SA_TRY(mem.alloc(&d_active, (size_t)L));
    SA_TRY(mem.alloc(&d_count, 1));
    if (!d_text) SA_TRY(hipMemcpy(d_seq, seq, (size_t)L * 2, hipMemcpyHostToDevice));

    // one temporary buffer for all rocPRIM calls (sizes queried for the full length)
    const int low_bits = ceil_log2((uint64_t)L + 1), high_bits = ceil_log2((uint64_t)L);
    size_t tmp_sort = 0, tmp_scan = 0, tmp_select = 0;
    SA_TRY(rocprim::radix_sort_pairs(nullptr, tmp_sort, d_keys, d_keys_alt, d_vals, d_vals_alt, (size_t)L, 0u, 64u));
    SA_TRY(rocprim::inclusive_scan(nullptr, tmp_scan, d_head, d_head, (size_t)L, rocprim::maximum<uint32_t>()));
    SA_TRY(rocprim::select(nullptr, tmp_select, rocprim::counting_iterator<uint32_t>(0), d_active, d_rows, d_count,
                           (size_t)L));
    size_t tmp_select_rows = 0;
    SA_TRY(rocprim::select(nullptr, tmp_select_rows, d_rows, d_active, d_rows_alt, d_count, (size_t)L));
    size_t tmp_bytes = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
    if (tmp_select > tmp_bytes) tmp_bytes = tmp_select;
    if (tmp_select_rows > tmp_bytes) tmp_bytes = tmp_select_rows;
    uint8_t *d_tmp = nullptr;
    SA_TRY(mem.alloc(&d_tmp, tmp_bytes));

    // round 0
    const int code_bits = alphabet > 0 ? (ceil_log2((uint64_t)alphabet) > 0 ? ceil_log2((uint64_t)alphabet) : 1) : 16;
    const int first_chars = 64 / code_bits > 16 ? 16 : 64 / code_bits;
    hipLaunchKernelGGL(k_sa_first_keys, dim3(grid_of(L)), dim3(kThreads), 0, 0, d_seq, L, first_chars, code_bits, d_keys, d_vals);
    size_t bytes = tmp_bytes;
    SA_TRY(rocprim::radix_sort_pairs(d_tmp, bytes, d_keys, d_keys_alt, d_vals, d_sa, (size_t)L, 0u,
                                     (unsigned)(first_chars * code_bits)));
    hipLaunchKernelGGL(k_sa_mark_heads, dim3(grid_of(L)), dim3(kThreads), 0, 0, d_keys_alt, (const uint32_t *)nullptr, L,
                       d_head);
    bytes = tmp_bytes;
    SA_TRY(rocprim::inclusive_scan(d_tmp, bytes, d_head, d_head, (size_t)L, rocprim::maximum<uint32_t>()));
    hipLaunchKernelGGL(k_sa_assign_ranks, dim3(grid_of(L)), dim3(kThreads), 0, 0, d_sa, (const uint32_t *)nullptr, d_head,
                       L, d_rank, d_sa);

    int rounds = 0;
    uint64_t sorted_rows = L;
    uint32_t n_prev = 0xffffffffu;  // tied rows of the previous round (none yet)
    for (uint64_t h = (uint64_t)first_chars; h < (uint64_t)L * 2; h <<= 1) {
        // rows still tied: among all rows after round 0, among the previous round's tied rows afterwards
        if (n_prev == 0xffffffffu) {
            hipLaunchKernelGGL(k_sa_mark_active, dim3(grid_of(L)), dim3(kThreads), 0, 0, d_sa, d_rank, L, d_active);
            bytes = tmp_bytes;
            SA_TRY(rocprim::select(d_tmp, bytes, rocprim::counting_iterator<uint32_t>(0), d_active, d_rows, d_count,
                                   (size_t)L));
        } else {
            hipLaunchKernelGGL(k_sa_mark_active_rows, dim3(grid_of(n_prev)), dim3(kThreads), 0, 0, d_sa, d_rank, L, d_rows,
                               n_prev, d_active);
            bytes = tmp_bytes;
            SA_TRY(rocprim::select(d_tmp, bytes, d_rows, d_active, d_rows_alt, d_count, (size_t)n_prev));
            uint32_t *t = d_rows;
            d_rows = d_rows_alt;
            d_rows_alt = t;
        }
        uint32_t n_active = 0;
        SA_TRY(hipMemcpy(&n_active, d_count, 4, hipMemcpyDeviceToHost));
        n_prev = n_active;
        if (n_active == 0) break;
        ++rounds;
        sorted_rows += n_active;
        const uint32_t hh = (uint32_t)(h < L ? h : L);
        hipLaunchKernelGGL(k_sa_next_keys, dim3(grid_of(n_active)), dim3(kThreads), 0, 0, d_sa, d_rows, d_rank, n_active, L,
                           hh, low_bits, d_keys, d_vals);
        bytes = tmp_bytes;
        SA_TRY(rocprim::radix_sort_pairs(d_tmp, bytes, d_keys, d_keys_alt, d_vals, d_vals_alt, (size_t)n_active, 0u,
                                         (unsigned)(low_bits + high_bits)));
        hipLaunchKernelGGL(k_sa_mark_heads, dim3(grid_of(n_active)), dim3(kThreads), 0, 0, d_keys_alt, d_rows, n_active,
                           d_head);
        bytes = tmp_bytes;
        SA_TRY(rocprim::inclusive_scan(d_tmp, bytes, d_head, d_head, (size_t)n_active, rocprim::maximum<uint32_t>()));
        hipLaunchKernelGGL(k_sa_assign_ranks, dim3(grid_of(n_active)), dim3(kThreads), 0, 0, d_vals_alt, d_rows, d_head,
                           n_active, d_rank, d_sa);
    }
    SA_TRY(hipGetLastError());

    // FM:329-394 from the suffix array
    const size_t n_words = (size_t)L / 64 + 2;
    const size_t n_samples = (size_t)L / (size_t)sample_rate + 2;
    int16_t *d_bwt = nullptr;
    uint64_t *d_which = nullptr;
    uint32_t *d_suffix_vals = nullptr, *d_position_vals = nullptr;
    SA_TRY(mem.alloc(&d_bwt, (size_t)L));
    SA_TRY(mem.alloc(&d_which, n_words));
    SA_TRY(mem.alloc(&d_suffix_vals, n_samples));
    SA_TRY(hipMemset(d_which, 0, n_words * 8));
    if (extract) {
        SA_TRY(mem.alloc(&d_position_vals, n_samples));
        SA_TRY(hipMemset(d_position_vals, 0, n_samples * 4));
    }
    uint32_t *d_flag = d_head, *d_slot = d_rows;  // reuse
    hipLaunchKernelGGL(k_sa_outputs, dim3(grid_of(((int64_t)L + 63) / 64 * 64)), dim3(kThreads), 0, 0, d_sa, d_seq, L,
                       (uint32_t)sample_rate, d_bwt, d_which, d_flag, d_position_vals);
    bytes = tmp_bytes;
    SA_TRY(rocprim::inclusive_scan(d_tmp, bytes, d_flag, d_slot, (size_t)L, rocprim::plus<uint32_t>()));
    hipLaunchKernelGGL(k_sa_compact_samples, dim3(grid_of(L)), dim3(kThreads), 0, 0, d_sa, d_flag, d_slot, L,
                       d_suffix_vals);
    SA_TRY(hipGetLastError());
    uint32_t n_sampled = 0;
    SA_TRY(hipMemcpy(&n_sampled, d_slot + (L - 1), 4, hipMemcpyDeviceToHost));

    // FM:173: the wavelet tree over the BWT, encoded where the BWT lies (the suffix-array buffers are released first)
    out.wavelet_done = false;
    double wt_seconds = 0;
    if (wt && device_wavelet_stage) {
        const auto t_wt = std::chrono::steady_clock::now();
        for (void *p : {(void *)d_keys, (void *)d_keys_alt, (void *)d_vals, (void *)d_vals_alt, (void *)d_rank}) mem.release(p);
        const int rc = device_wavelet_stage(d_bwt, (int64_t)L, sample_rate, alphabet, *wt, err);
        if (rc < 0) return rc;
        out.wavelet_done = rc == 0;
        wt_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wt).count();
    }
    out.bwt.clear();
    if (!out.wavelet_done) {
        out.bwt.resize((size_t)L);
        SA_TRY(hipMemcpy(out.bwt.data(), d_bwt, (size_t)L * 2, hipMemcpyDeviceToHost));
    }
    // FM:343-370: the sample vectors packed and the bitmap RRR-encoded where they lie
    out.vectors_done = false;
    if (wt && device_pack_values && device_rrr_of_bits) {
        const int width = 64 - __builtin_clzll((unsigned long long)L);  // CMN:169-175 minimumNumberOfBits(n), n >= 1
        int rc = device_pack_values(d_suffix_vals, (int64_t)n_sampled, (int64_t)L / sample_rate + 1, width, -1, out.suffixes, err);
        if (rc) return rc;
        if (extract) {
            const int64_t n_pos = (int64_t)(L - 1) / sample_rate + 1;  // slots 0 .. (n-1)/s hold samples
            rc = device_pack_values(d_position_vals, n_pos, (int64_t)L / sample_rate + 2, width, (int64_t)(L - 1) / sample_rate + 1,
                                    out.positions, err);
            if (rc) return rc;
        }
        rc = device_rrr_of_bits(d_which, (int64_t)L, sample_rate, out.sampled, err);
        if (rc) return rc;
        out.vectors_done = true;
        out.which.clear();
        out.suffix_vals.clear();
        out.position_vals.clear();
        if (stats) {
            stats->rounds = rounds;
            stats->rows_sorted = sorted_rows;
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
            stats->wavelet_seconds = out.wavelet_done ? wt_seconds : 0;
        }
        return 0;
    }
    out.which.assign(n_words, 0);
    out.suffix_vals.resize(n_sampled);
    SA_TRY(hipMemcpy(out.which.data(), d_which, n_words * 8, hipMemcpyDeviceToHost));
    if (n_sampled) SA_TRY(hipMemcpy(out.suffix_vals.data(), d_suffix_vals, (size_t)n_sampled * 4, hipMemcpyDeviceToHost));
    out.position_vals.clear();
    if (extract) {
        out.position_vals.resize(n_samples);
        SA_TRY(hipMemcpy(out.position_vals.data(), d_position_vals, n_samples * 4, hipMemcpyDeviceToHost));
    }
    if (stats) {
        stats->rounds = rounds;
        stats->rows_sorted = sorted_rows;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        stats->wavelet_seconds = out.wavelet_done ? wt_seconds : 0;
    }
    return 0;
}

}  // namespace fmx
