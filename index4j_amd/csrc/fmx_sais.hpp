// fmx_sais.hpp — suffix array construction by induced sorting (SA-IS, Nong/Zhang/Chan 2009).
//
// Replaces the reference's third-party jsuffixarrays DivSufSort call (FM:332-341).  The indexed
// sequence always ends in a unique smallest symbol (FM:300-305, FM:433), so its suffix array is
// unique and any correct construction reproduces the reference's array.
//
// Precondition: s[n-1] == 0 and 0 occurs nowhere else; symbols in [0, K).
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

namespace fmx {
namespace sais_detail {

struct TypeBits {  // 1 = S-type, 0 = L-type
    std::vector<uint64_t> w;
    explicit TypeBits(int64_t n) : w((size_t)(n / 64 + 1), 0) {}
    inline bool get(int64_t i) const { return (w[(size_t)(i >> 6)] >> (i & 63)) & 1ULL; }
    inline void set(int64_t i, bool v) {
        if (v)
            w[(size_t)(i >> 6)] |= (1ULL << (i & 63));
        else
            w[(size_t)(i >> 6)] &= ~(1ULL << (i & 63));
    }
    inline bool lms(int64_t i) const { return i > 0 && get(i) && !get(i - 1); }
};

template <typename T>
static void get_buckets(const T *s, int32_t *bkt, int32_t n, int32_t K, bool end) {
    for (int32_t i = 0; i < K; ++i) bkt[i] = 0;
    for (int32_t i = 0; i < n; ++i) ++bkt[s[i]];
    int32_t sum = 0;
    for (int32_t i = 0; i < K; ++i) {
        sum += bkt[i];
        bkt[i] = end ? sum : sum - bkt[i];
    }
}

template <typename T>
static void induce_l(const TypeBits &t, int32_t *SA, const T *s, int32_t *bkt, int32_t n, int32_t K) {
    get_buckets(s, bkt, n, K, false);
    for (int32_t i = 0; i < n; ++i) {
        int32_t j = SA[i] - 1;
        if (j >= 0 && !t.get(j)) SA[bkt[s[j]]++] = j;
    }
}

template <typename T>
static void induce_s(const TypeBits &t, int32_t *SA, const T *s, int32_t *bkt, int32_t n, int32_t K) {
    get_buckets(s, bkt, n, K, true);
    for (int32_t i = n - 1; i >= 0; --i) {
        int32_t j = SA[i] - 1;
        if (j >= 0 && t.get(j)) SA[--bkt[s[j]]] = j;
    }
}

template <typename T>
static void sais(const T *s, int32_t *SA, int32_t n, int32_t K) {
    if (n == 1) {
        SA[0] = 0;
        return;
    }
    TypeBits t(n);
    t.set(n - 1, true);
    t.set(n - 2, false);
    for (int32_t i = n - 3; i >= 0; --i)
        t.set(i, (s[i] < s[i + 1] || (s[i] == s[i + 1] && t.get(i + 1))));

    std::vector<int32_t> bkt_v((size_t)K);
    int32_t *bkt = bkt_v.data();

    // stage 1: sort the LMS substrings
    get_buckets(s, bkt, n, K, true);
    for (int32_t i = 0; i < n; ++i) SA[i] = -1;
    for (int32_t i = 1; i < n; ++i)
        if (t.lms(i)) SA[--bkt[s[i]]] = i;
    induce_l(t, SA, s, bkt, n, K);
    induce_s(t, SA, s, bkt, n, K);

    // compact the sorted LMS substrings into SA[0, n1)
    int32_t n1 = 0;
    for (int32_t i = 0; i < n; ++i)
        if (t.lms(SA[i])) SA[n1++] = SA[i];

    // name them
    for (int32_t i = n1; i < n; ++i) SA[i] = -1;
    int32_t name = 0, prev = -1;
    for (int32_t i = 0; i < n1; ++i) {
        int32_t pos = SA[i];
        bool diff = false;
        for (int32_t d = 0; d < n; ++d) {
            if (prev == -1 || s[pos + d] != s[prev + d] || t.get(pos + d) != t.get(prev + d)) {
                diff = true;
                break;
            } else if (d > 0 && (t.lms(pos + d) || t.lms(prev + d))) {
                break;
            }
        }
        if (diff) {
            ++name;
            prev = pos;
        }
        SA[n1 + pos / 2] = name - 1;
    }
    for (int32_t i = n - 1, j = n - 1; i >= n1; --i)
        if (SA[i] >= 0) SA[j--] = SA[i];

    // stage 2: solve the reduced problem
    int32_t *SA1 = SA, *s1 = SA + n - n1;
    if (name < n1)
        sais<int32_t>(s1, SA1, n1, name);
    else
        for (int32_t i = 0; i < n1; ++i) SA1[s1[i]] = i;

    // stage 3: induce the final order
    get_buckets(s, bkt, n, K, true);
    for (int32_t i = 1, j = 0; i < n; ++i)
        if (t.lms(i)) s1[j++] = i;
    for (int32_t i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
    for (int32_t i = n1; i < n; ++i) SA[i] = -1;
    for (int32_t i = n1 - 1; i >= 0; --i) {
        int32_t j = SA[i];
        SA[i] = -1;
        SA[--bkt[s[j]]] = j;
    }
    induce_l(t, SA, s, bkt, n, K);
    induce_s(t, SA, s, bkt, n, K);
}

}  // namespace sais_detail

// s: n symbols in [0, K), s[n-1] == 0 unique smallest.  SA: n ints.
inline void suffix_array(const int16_t *s, int32_t n, int32_t K, int32_t *SA) {
    sais_detail::sais<int16_t>(s, SA, n, K);
}

}  // namespace fmx
