// tools/microbench/sector_requests.hip — what a lane pays for reading MORE than 16 bytes of one random 64-byte sector
// (round 5: the window directory's cells are 64 bytes; a lane fetches one with four dwordx4 loads).  Chained, 8 waves per SIMD,
// 192 MiB table (inside the Infinity Cache) and 2 GiB (HBM), cells per second:
//   A  one 16-byte load per cell                       (tools/microbench/random_lines.hip's figure)
//   B2 / B4  two / four 16-byte loads of the SAME sector by the SAME lane, back to back
//   C4 four loads per wave-step, TRANSPOSED: instruction k, lane l reads quad (l & 3) of the cell of lane (l >> 2) + 16 k — every
//      instruction touches 16 whole sectors — and the lanes get their own cell's quads back through LDS
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Q { uint32_t x, y, z, w; };

template <int kMode>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_cells(const Q *__restrict__ table, uint32_t cell_mask,
                                                                                             int rounds, uint32_t *out) {
    __shared__ Q s_x[512 * 4];
    uint32_t s0 = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    const int lane = threadIdx.x & 63, wave_base = (threadIdx.x & ~63) * 4;
    for (int r = 0; r < rounds; ++r) {
        const size_t cell = (size_t)(s0 & cell_mask) * 4;  // 4 quads per 64-byte cell
        if (kMode == 1) {
            const Q v = table[cell + (s0 >> 30)];
            acc += v.y;
            s0 = s0 * 1664525u + 1013904223u + v.x;
        } else if (kMode == 2) {
            const Q v0 = table[cell], v1 = table[cell + 1];
            acc += v0.y ^ v1.z;
            s0 = s0 * 1664525u + 1013904223u + v0.x + v1.x;
        } else if (kMode == 4) {
            const Q v0 = table[cell], v1 = table[cell + 1], v2 = table[cell + 2], v3 = table[cell + 3];
            acc += v0.y ^ v1.z ^ v2.w ^ v3.y;
            s0 = s0 * 1664525u + 1013904223u + v0.x + v1.x + v2.x + v3.x;
        } else {  // transposed
            Q v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t other = __shfl(s0, (lane >> 2) + 16 * k);
                v[k] = table[(size_t)(other & cell_mask) * 4 + (lane & 3)];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) s_x[wave_base + (((lane >> 2) + 16 * k) * 4 + (lane & 3))] = v[k];
            // (a wave's own slots: no barrier needed beyond the wave's lockstep + LDS ordering)
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            const Q v0 = s_x[wave_base + lane * 4], v1 = s_x[wave_base + lane * 4 + 1], v2 = s_x[wave_base + lane * 4 + 2],
                    v3 = s_x[wave_base + lane * 4 + 3];
            __builtin_amdgcn_wave_barrier();
            acc += v0.y ^ v1.z ^ v2.w ^ v3.y;
            s0 = s0 * 1664525u + 1013904223u + v0.x + v1.x + v2.x + v3.x;
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
    for (size_t mib : {128, 2048}) {
        Q *table;
        CK(hipMalloc(&table, mib << 20));
        CK(hipMemset(table, 0, mib << 20));
        const uint32_t cells = (uint32_t)((mib << 20) / 64), mask = cells - 1;
        for (int mode : {1, 2, 4, 5}) {
            const int rounds = 64, grid = 256 * 16;
            for (int rep = 0; rep < 3; ++rep) {
                if (rep == 2) CK(hipEventRecord(e0));
                if (mode == 1) hipLaunchKernelGGL(k_cells<1>, dim3(grid), dim3(512), 0, 0, table, mask, rounds, out);
                if (mode == 2) hipLaunchKernelGGL(k_cells<2>, dim3(grid), dim3(512), 0, 0, table, mask, rounds, out);
                if (mode == 4) hipLaunchKernelGGL(k_cells<4>, dim3(grid), dim3(512), 0, 0, table, mask, rounds, out);
                if (mode == 5) hipLaunchKernelGGL(k_cells<5>, dim3(grid), dim3(512), 0, 0, table, mask, rounds, out);
            }
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double n = (double)grid * 512 * rounds;
            printf("table %5zu MiB, %s: %.1f G cells/s\n", mib,
                   mode == 1 ? "A  one 16-byte load per cell          " : mode == 2 ? "B2 two loads of the same sector       "
                   : mode == 4 ? "B4 four loads of the same sector      " : "C4 four transposed loads + LDS exchange", n / (ms * 1e-3) / 1e9);
        }
        CK(hipFree(table));
    }
    return 0;
}
