// The middle stage of index construction (FM:329-394): everything FmIndex derives from the suffix array of
// the mapped text.  Two interchangeable producers: the host's SA-IS (fmx_build.cpp) and the device's prefix
// doubling (fmx_sa_gpu.hip).  The suffix array is unique, so both give the same arrays.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "fmx_model.hpp"

namespace fmx {

struct SaStage {
    std::vector<int16_t> bwt;             // FM:385-392
    std::vector<uint64_t> which;          // bit i = row i is sampled (SA[i] % sampleRate == 0), n/64 + 2 words
    std::vector<uint32_t> suffix_vals;    // SA[i] of the sampled rows, in row order (FM:341-352)
    std::vector<uint32_t> position_vals;  // [SA[i] / sampleRate] = i for sampled rows (FM:356-366); n/s + 2 slots
    bool wavelet_done = false;            // the device stage also encoded the wavelet tree (then `bwt` stays empty)
    // the device stage also packed the samples and RRR-encoded the bitmap (then which / *_vals stay empty):
    bool vectors_done = false;
    PackedVec suffixes, positions;        // FM:343-344, 359-370 (incl. the wrap entry)
    RrrModel sampled;                     // FM:345-357 -> RRR:225-286
};

struct SaStageStats {
    int rounds = 0;            // doubling rounds after the initial 4-character sort
    uint64_t rows_sorted = 0;  // rows that went through a device sort, summed over rounds
    double seconds = 0;        // wall time of the stage incl. the transfers to and from HBM
    double wavelet_seconds = 0;  // of which: the wavelet-tree encode in HBM (0: encoded on the host)
};

// seq: mapped text incl. the terminator (code 0 at n-1 only); returns 0 or a negative fmx error code
int host_sa_stage(const int16_t *seq, int32_t n, int alphabet, int sample_rate, bool extract, SaStage &out);
// (weak: fmx_build.cpp also links without the HIP translation unit, e.g. in the sanitizer build of the host code)
// wt != nullptr: the wavelet tree (FM:173, WFBB:130-154) is encoded in HBM as well, from the BWT where it lies
// (fmx_wt_gpu.hip; superblocks of up to kWtMaxSigma distinct symbols — a text with a fuller one leaves wavelet_done false and the BWT in `out`)
// d_text != nullptr: the text already lies in HBM as raw characters (device_alphabet_stage) and is mapped there
// through code_of (65,536 entries) instead of being uploaded as `seq`; the stage frees it.
int device_sa_stage(const int16_t *seq, int32_t n, int sample_rate, bool extract, int device, SaStage &out,
                    SaStageStats *stats, std::string &err, WfbbModel *wt = nullptr, int alphabet = 0,
                    void *d_text = nullptr, const int16_t *code_of = nullptr) __attribute__((weak));
// FM:396-435, first half: uploads the text (n_in characters) and returns, per character value, its first position
// (-1: absent) and its count; *d_text = the device copy (n_in + 1 elements) for device_sa_stage
int device_alphabet_stage(const uint16_t *input, int32_t n_in, int device, std::vector<int32_t> &first,
                          std::vector<int64_t> &count, void **d_text, std::string &err) __attribute__((weak));
void device_release(void *d_ptr) __attribute__((weak));  // hipFree of a buffer a device stage handed out
int device_pack_values(const uint32_t *d_vals, int64_t n_vals, int64_t length, int width, int64_t wrap_index,
                       PackedVec &out, std::string &err) __attribute__((weak));
int device_rrr_of_bits(const uint64_t *d_bits, int64_t nbits, int sample, RrrModel &m, std::string &err)
    __attribute__((weak));
constexpr int kWtMaxSigma = 1536;  // distinct symbols of ONE superblock (a wave's LDS scratch: 38 bytes per symbol, 64 KiB per workgroup)
// 0 = encoded, 1 = not handled here (a superblock with more symbols than that, a code longer than 31 bits): encode on the
// host, < 0 = error
int device_wavelet_stage(const int16_t *d_bwt, int64_t n, int sampling_rate, int alphabet, WfbbModel &w,
                         std::string &err) __attribute__((weak));

}  // namespace fmx
