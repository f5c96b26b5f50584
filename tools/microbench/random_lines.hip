// tools/microbench/random_lines.hip — the ceiling the LF kernels run against: how many RANDOM 128-byte lines per second
// the memory system of one MI355X delivers to 16-byte loads (the access pattern of rank / inverseSelect: a chain of
// dependent 16-byte loads at unrelated addresses), for tables that fit the 256 MiB Infinity Cache and tables that do not.
//   independent = 4: four loads in flight per lane (a throughput ceiling); chained: each address depends on the loaded value
//   (the kernels' pattern).  8 waves per SIMD as in k_count.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Q { uint32_t x, y, z, w; };

template <int kChained>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lines(const Q *__restrict__ table, uint32_t line_mask,
                                                                                             int rounds, uint32_t *out) {
    uint32_t s0 = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (int r = 0; r < rounds; ++r) {
        if (kChained) {
            const Q v = table[(size_t)(s0 & line_mask) * 8 + (s0 >> 29)];
            acc += v.y;
            s0 = s0 * 1664525u + 1013904223u + v.x;  // (v.x is 0: the dependence is real, the sequence is known)
        } else {
            uint32_t a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s0 = s0 * 1664525u + 1013904223u;
                a[k] = s0;
            }
            Q v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = table[(size_t)(a[k] & line_mask) * 8 + (a[k] >> 29)];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += v[k].x ^ v[k].w;
        }
    }
    if (acc == 0x9e3779b9u) *out = acc;
}

int main() {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    uint32_t *out;
    CK(hipMalloc(&out, 64));
    for (size_t mib : {32, 128, 192, 512, 2048}) {
        Q *table;
        CK(hipMalloc(&table, mib << 20));
        CK(hipMemset(table, 0, mib << 20));
        const uint32_t lines = (uint32_t)((mib << 20) / 128), mask = lines - 1;  // (sizes are powers of two, or 192: masked to 128)
        const uint32_t use_mask = (lines & (lines - 1)) ? (1u << 31 >> __builtin_clz(lines)) - 1 : mask;
        for (int chained : {0, 1}) {
            const int rounds = chained ? 64 : 16;
            const int grid = 256 * 16;
            for (int rep = 0; rep < 3; ++rep) {
                if (rep == 2) CK(hipEventRecord(e0));
                if (chained)
                    hipLaunchKernelGGL(k_lines<1>, dim3(grid), dim3(512), 0, 0, table, use_mask, rounds, out);
                else
                    hipLaunchKernelGGL(k_lines<0>, dim3(grid), dim3(512), 0, 0, table, use_mask, rounds, out);
            }
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double loads = (double)grid * 512 * rounds * (chained ? 1 : 4);
            printf("table %5zu MiB (%u lines used), %s: %.1f G random 16-byte loads/s = %.2f TB/s of 128-byte lines\n", mib, use_mask + 1,
                   chained ? "chained (1 in flight per lane)   " : "independent (4 in flight per lane)", loads / (ms * 1e-3) / 1e9,
                   loads * 128 / (ms * 1e-3) / 1e12);
        }
        CK(hipFree(table));
    }
    return 0;
}
