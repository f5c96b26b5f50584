// fmx_plan.hpp — what the plan stage of a batch (fmx_kernels.hip: k_plan_codes, k_plan_scatter) hands to k_count.
// Shared by the launchers and the C-ABI layer.
#pragma once

#include <cstddef>
#include <cstdint>

namespace fmx {

struct SortShape {
    int bits;         // bits per alphabet code
    int chars;        // trailing characters in the full key
    int total_bits;   // chars * bits (<= 32); with sa_key: bits of an SA row number of this index
    int coarse_bits;  // top bits used by the bucket pass
    int sa_key;       // 1: a pattern's key is the first SA row of its tabulated suffix (the suffix table's answer) — neighbours
                      // in that order read neighbouring BWT positions in their first step, and LF-mapping keeps rows that are
                      // preceded by the same character in order: the locality survives the steps; 0: the trailing characters'
                      // codes, the last character most significant (indexes without a suffix table)
};

struct CountPlan {
    const void *recs = nullptr;  // PlanRec[n] in processing order (device memory); nullptr = the caller's order
    int32_t n = 0;
    int code_bits = 8;           // width of one code in a record's code word (8, or 16 when sigma > 256)
    SortShape shape = {1, 1, 1, 1, 0};
    // the alphabet the code words are written in (the index the plan was made with): code -> char, alphabet size
    const int32_t *look_up = nullptr;
    int32_t sigma = 0;
    // *mixed == epoch: the batch holds patterns of different lengths (k_plan_codes stores the plan's epoch there when it sees
    // two lengths; never reset — the next plan has another epoch).  k_count regroups its workgroups by length only then.
    const uint32_t *mixed = nullptr;
    uint32_t epoch = 0;
    // k_count_lean's redo list (patterns that met a route it does not carry: room for n indices) and its counters {entries,
    // workgroups of the list pass that are done} — both zero between launches; nullptr = no room (the general k_count runs)
    int32_t *redo_list = nullptr;
    uint32_t *redo_count = nullptr;
};

// head of the plan workspace: histogram, cursors, ticket — all zero between plans (k_plan_scatter restores that)
constexpr size_t kPlanHeadBytes = 2 * ((size_t)4 << 14) + 256;

}  // namespace fmx
