// tools/microbench/fetch_calibration.hip — what rocprofv3's FETCH_SIZE reports for SCATTERED 16-byte loads on gfx950.
// The guide calibrates the counter for wide streaming reads only (it reports 1/2 of the bytes there); k_count's traffic
// is scattered 16-byte loads, so bench.py's traffic_frac needs its own factor.  Every lane loads 16 bytes from lines
// nobody else touches: `lines` DISTINCT 128-byte-aligned lines of a table far larger than the 256 MiB Infinity Cache and
// the 32 MiB of L2, visited in a multiplicative-permutation order.  Known unique bytes: lines x 64 (or x 128).
//   fetch_calibration <table MiB> <loads> [stream]   prints one JSON line (HIP-event time of the measured launch)
// stream = 1: the same number of bytes as one coalesced streaming read (16 B per lane, consecutive), the guide's case.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

struct Q { uint32_t x, y, z, w; };

__global__ __launch_bounds__(512) void k_calib_scatter(const Q *__restrict__ table, uint64_t lines, uint64_t loads, uint64_t mult,
                                                       uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 512 + threadIdx.x; i < loads; i += (uint64_t)gridDim.x * 512) {
        const uint64_t line = (i * mult) % lines;        // mult coprime with lines: a permutation, every line at most once
        const Q v = table[line * 8 + (i & 7)];           // 16 bytes somewhere inside the 128-byte line
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
// both 64-byte halves of every line it visits (two 16-byte loads, 64 bytes apart): one fabric request per LINE means a miss
// fills the whole 128-byte line; two mean it fills 64-byte sectors
__global__ __launch_bounds__(512) void k_calib_halves(const Q *__restrict__ table, uint64_t lines, uint64_t loads, uint64_t mult,
                                                      uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 512 + threadIdx.x; i < loads; i += (uint64_t)gridDim.x * 512) {
        const uint64_t line = (i * mult) % lines;
        const Q v = table[line * 8 + (i & 3)];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        const Q w = table[line * 8 + 4 + ((i + v.x) & 3)];  // (address depends on the first load: never merged into one access)
        acc += w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
__global__ __launch_bounds__(512) void k_calib_stream(const Q *__restrict__ table, uint64_t loads, uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 512 + threadIdx.x; i < loads; i += (uint64_t)gridDim.x * 512) {
        const Q v = table[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}

int main(int argc, char **argv) {
    const uint64_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2048;
    uint64_t loads = argc > 2 ? strtoull(argv[2], nullptr, 10) : (1ull << 24);
    const int stream = argc > 3 ? atoi(argv[3]) : 0;
    const uint64_t bytes = mib << 20, lines = bytes / 128;
    if (loads > lines) loads = lines;
    Q *table;
    uint32_t *out;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(table, 1, bytes);
    (void)hipDeviceSynchronize();
    uint64_t mult = 2654435761ull;
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { uint64_t t = a % b; a = b; b = t; } return a; };
    while (gcd(mult, lines) != 1) mult += 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // ONE measured launch (a second launch over the same lines would find part of them in the Infinity Cache)
    (void)hipEventRecord(e0);
    if (stream == 2)
        hipLaunchKernelGGL(k_calib_halves, dim3(4096), dim3(512), 0, 0, table, lines, loads, mult, out);
    else if (stream)
        hipLaunchKernelGGL(k_calib_stream, dim3(4096), dim3(512), 0, 0, table, loads, out);
    else
        hipLaunchKernelGGL(k_calib_scatter, dim3(4096), dim3(512), 0, 0, table, lines, loads, mult, out);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("{\"kernel\": \"%s\", \"table_MiB\": %llu, \"loads\": %llu, \"bytes_loaded\": %llu, \"unique_lines_128B\": %llu, \"ms\": %.4f, "
           "\"loads_per_s\": %.4g}\n",
           stream == 2 ? "k_calib_halves" : stream ? "k_calib_stream" : "k_calib_scatter", (unsigned long long)mib, (unsigned long long)loads,
           (unsigned long long)(loads * (stream == 2 ? 32 : 16)), (unsigned long long)(stream == 1 ? loads / 8 : loads), ms, loads / (ms * 1e-3));
    return 0;
}
