// tools/microbench/plan_primitives.hip — what the building blocks of a one-launch plan stage cost on gfx950:
//   * back-to-back launches of an empty kernel (normal and cooperative)
//   * a grid barrier made of one atomic counter (256 workgroups of 1,024 threads, all resident)
//   * returning global atomics: 1 M over 64 Ki addresses (uniform), over 4 Ki, and all on ONE address
//   * 1 M scattered 16-byte writes
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/plan_primitives.hip -o gpurun_out/plan_primitives
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}

__global__ __launch_bounds__(1024) void k_barrier(uint32_t *counter, int rounds, uint32_t *out) {
    uint32_t target = 0;
    for (int r = 0; r < rounds; ++r) {
        target += gridDim.x;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = target;
}

__global__ __launch_bounds__(1024) void k_atomics(uint32_t *bins, uint32_t mask, int per_thread, uint32_t *out) {
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    uint32_t r[8];
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        r[i & 7] = atomicAdd(&bins[(x >> 8) & mask], 1u);
    }
    for (int i = 0; i < 8 && i < per_thread; ++i) acc += r[i];
    if (acc == 0xffffffffu) *out = acc;
}

struct Q { uint32_t x, y, z, w; };
__global__ __launch_bounds__(1024) void k_scatter(Q *dst, uint32_t n, int per_thread) {
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        dst[(x >> 4) % n] = Q{x, x, x, x};
    }
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    uint32_t *d;
    CK(hipMalloc(&d, 64 << 20));
    CK(hipMemset(d, 0, 64 << 20));
    uint32_t *out = d + (1 << 22);
    // empty launches
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel, back to back:            %.2f us per launch\n", ms * 1e3 / 200);
    {
        void *args[] = {};
        for (int i = 0; i < 5; ++i) CK(hipLaunchCooperativeKernel((void *)k_empty, dim3(256), dim3(1024), args, 0, st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 50; ++i) CK(hipLaunchCooperativeKernel((void *)k_empty, dim3(256), dim3(1024), args, 0, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty COOPERATIVE launch (256 x 1024): %.2f us per launch\n", ms * 1e3 / 50);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, st);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty normal launch (256 x 1024):      %.2f us per launch\n", ms * 1e3 / 50);
    }
    // grid barrier
    for (int rounds : {1, 11, 101}) {
        CK(hipMemsetAsync(d, 0, 64, st));
        hipLaunchKernelGGL(k_barrier, dim3(256), dim3(1024), 0, st, d, rounds, out);
        CK(hipMemsetAsync(d, 0, 64, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_barrier, dim3(256), dim3(1024), 0, st, d, rounds, out);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("kernel with %3d grid barriers (256 x 1024): %.2f us\n", rounds, ms * 1e3);
    }
    // atomics
    for (uint32_t bins : {1u << 16, 1u << 12, 1u}) {
        CK(hipMemsetAsync(d, 0, 1 << 20, st));
        hipLaunchKernelGGL(k_atomics, dim3(256), dim3(1024), 0, st, d, bins - 1, 4, out);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_atomics, dim3(256), dim3(1024), 0, st, d, bins - 1, 4, out);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("1 M returning atomics over %6u addresses:  %.2f us\n", bins, ms * 1e3);
    }
    {
        Q *q = reinterpret_cast<Q *>(d);
        hipLaunchKernelGGL(k_scatter, dim3(256), dim3(1024), 0, st, q, 1u << 20, 4);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_scatter, dim3(256), dim3(1024), 0, st, q, 1u << 20, 4);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("1 M scattered 16-byte writes (16 MB region):    %.2f us\n", ms * 1e3);
    }
    return 0;
}
