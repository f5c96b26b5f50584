// tools/microbench/chase_latency.hip — latency of ONE dependent 16-byte load by table size (one wave per XCD, every lane its
// own chain; second pass over a table that fits L2 = L2 hit latency; big tables: what a random LF-walk pays per round trip,
// address translation included).
// build: hipcc -O3 --offload-arch=gfx950 chase_latency.hip -o chase_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

struct alignas(16) Node { unsigned next, pad[3]; };

__global__ __launch_bounds__(64) void k_chase(const Node *table, unsigned n, int steps, unsigned long long *out) {
    unsigned p = (unsigned)((blockIdx.x * 64ull + threadIdx.x) * 2654435761ull % n);
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < steps; ++i) p = table[p].next;
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2 + 0] = t1 - t0;
        out[blockIdx.x * 2 + 1] = p;
    }
}

int main() {
    const int steps = 4096;
    unsigned long long *d_out;
    (void)hipMalloc(&d_out, 2048 * 16);
    for (size_t mib : {2, 32, 256, 1024, 4096}) {
        const unsigned n = (unsigned)(mib << 20 >> 4);
        std::vector<unsigned> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(7);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<Node> h(n);
        for (unsigned i = 0; i < n; ++i) h[perm[i]].next = perm[(i + 1) % n];
        Node *d;
        (void)hipMalloc(&d, (size_t)n * sizeof(Node));
        (void)hipMemcpy(d, h.data(), (size_t)n * sizeof(Node), hipMemcpyHostToDevice);
        for (int waves : {8, 1024}) {  // one wave per XCD; one wave per SIMD of the whole chip
            double ns = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k_chase, dim3(waves), dim3(64), 0, 0, d, n, steps, d_out);
                std::vector<unsigned long long> o(waves * 2);
                (void)hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
                double t = 0;
                for (int b = 0; b < waves; ++b) t += (double)o[b * 2] / waves;
                ns = t * 10.0 / steps;
            }
            printf("table %5zu MiB, %4d waves of 64 chains: %7.1f ns per dependent load\n", mib, waves, ns);
        }
        (void)hipFree(d);
    }
    return 0;
}
