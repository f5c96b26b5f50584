// tools/microbench/l2_across_launches.hip — does a line an XCD's L2 holds survive the end of a kernel?
// One workgroup per XCD (8 workgroups of 64 lanes); every lane chases its own chain of dependent 16-byte loads through
// a 2 MiB table (fits one 4 MiB L2) — the same chain twice per launch, each pass timed with the 100 MHz wall clock.
// Pass 2 of a launch runs out of L2.  If pass 1 of the NEXT launch is as fast as that, L2 contents survive the kernel
// boundary; if it is as slow as pass 1 of the first launch, every launch starts with cold L2s.
// build: hipcc -O3 --offload-arch=gfx950 l2_across_launches.hip -o l2_across_launches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

struct alignas(16) Node { unsigned next, pad[3]; };

__global__ __launch_bounds__(64) void k_chase(const Node *table, int steps, unsigned long long *out) {
    unsigned start = (blockIdx.x * 64u + threadIdx.x) * 1021u % (2u << 20 >> 4);
    unsigned long long t[3];
    unsigned p = start, sink = 0;
    t[0] = wall_clock64();
    for (int i = 0; i < steps; ++i) p = table[p].next;
    sink += p;
    t[1] = wall_clock64();
    p = start;
    for (int i = 0; i < steps; ++i) p = table[p].next;
    sink += p;
    t[2] = wall_clock64();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3 + 0] = t[1] - t[0];
        out[blockIdx.x * 3 + 1] = t[2] - t[1];
        out[blockIdx.x * 3 + 2] = sink;
    }
}
__global__ void k_flush(const uint4 *big, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += big[i].x;
    if (acc == 0x12345678u) *out = acc;
}

int main() {
    const unsigned n = 2u << 20 >> 4;  // 131,072 nodes of 16 bytes = 2 MiB
    std::vector<unsigned> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<Node> h(n);
    for (unsigned i = 0; i < n; ++i) h[perm[i]].next = perm[(i + 1) % n];  // one cycle through all nodes
    Node *d;
    unsigned long long *d_out;
    uint4 *big;
    unsigned *d_sink;
    const size_t big_n = (size_t)1 << 26;  // 1 GiB of uint4: evicts L2s and the Infinity Cache
    hipMalloc(&d, n * sizeof(Node));
    hipMalloc(&d_out, 8 * 3 * 8);
    hipMalloc(&big, big_n * 16);
    hipMalloc(&d_sink, 4);
    hipMemset(big, 1, big_n * 16);
    hipMemcpy(d, h.data(), n * sizeof(Node), hipMemcpyHostToDevice);
    const int steps = 2048;
    auto run = [&](const char *what) {
        hipLaunchKernelGGL(k_chase, dim3(8), dim3(64), 0, 0, d, steps, d_out);
        unsigned long long o[24];
        hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost);
        double p1 = 0, p2 = 0;
        for (int b = 0; b < 8; ++b) { p1 += o[b * 3] / 8.0; p2 += o[b * 3 + 1] / 8.0; }
        printf("%-58s pass 1: %6.1f ns per load, pass 2: %6.1f ns per load\n", what, p1 * 10.0 / steps, p2 * 10.0 / steps);
    };
    auto flush = [&]() {
        hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, big, big_n, d_sink);
        hipDeviceSynchronize();
    };
    flush();
    run("after streaming 1 GiB (cold L2, cold Infinity Cache):");
    run("the next launch (hipMemcpy of 192 bytes in between):");
    run("and the next:");
    // two launches back to back without a host round trip in between
    hipLaunchKernelGGL(k_chase, dim3(8), dim3(64), 0, 0, d, steps, d_out);
    run("second of two launches queued back to back:");
    flush();
    run("after streaming 1 GiB again:");
    return 0;
}
