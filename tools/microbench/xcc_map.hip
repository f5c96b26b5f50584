// tools/microbench/xcc_map.hip — which XCD a workgroup lands on (HW_REG_XCC_ID), by workgroup index and grid size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_xcc(unsigned *out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    for (int grid : {64, 4096}) {
        unsigned *d;
        hipMalloc(&d, grid * 4);
        hipLaunchKernelGGL(k_xcc, dim3(grid), dim3(512), 0, 0, d);
        std::vector<unsigned> h(grid);
        hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
        printf("grid %d, first 32 workgroups -> XCC_ID & 0xf:", grid);
        for (int i = 0; i < 32; ++i) printf(" %u", h[i] & 0xf);
        int match = 0;
        for (int i = 0; i < grid; ++i) match += ((h[i] & 0xf) == (unsigned)(i % 8));
        printf("\n  workgroups with XCC == index mod 8: %d of %d\n", match, grid);
        hipFree(d);
    }
    return 0;
}
