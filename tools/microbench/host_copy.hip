// tools/microbench/host_copy.hip — what it costs to get a caller's PAGEABLE buffers to the GPU and back (the host-buffer
// entry points of fmx.h, what a JNI binding calls): plain hipMemcpy, a pinned staging copy, in-place registration.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    for (size_t mb : {1, 4, 16, 64}) {
        const size_t n = mb << 20;
        char *pageable = (char *)malloc(n), *pinned, *dev;
        memset(pageable, 1, n);
        CK(hipHostMalloc((void **)&pinned, n, hipHostMallocDefault));
        memset(pinned, 2, n);
        CK(hipMalloc((void **)&dev, n));
        double best[6] = {1e9, 1e9, 1e9, 1e9, 1e9, 1e9};
        for (int rep = 0; rep < 5; ++rep) {
            double t = now();
            CK(hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice));
            best[0] = std::min(best[0], now() - t);
            t = now();
            CK(hipMemcpy(dev, pinned, n, hipMemcpyHostToDevice));
            best[1] = std::min(best[1], now() - t);
            t = now();
            memcpy(pinned, pageable, n);
            best[2] = std::min(best[2], now() - t);
            t = now();
            CK(hipHostRegister(pageable, n, hipHostRegisterDefault));
            double t1 = now();
            CK(hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice));
            double t2 = now();
            CK(hipHostUnregister(pageable));
            best[3] = std::min(best[3], now() - t);
            (void)t1; (void)t2;
            t = now();
            CK(hipMemcpy(pageable, dev, n, hipMemcpyDeviceToHost));
            best[4] = std::min(best[4], now() - t);
            t = now();
            {   // 4 threads copy quarters into the pinned buffer
                std::vector<std::thread> th;
                for (int k = 0; k < 4; ++k) th.emplace_back([&, k] { memcpy(pinned + k * (n / 4), pageable + k * (n / 4), n / 4); });
                for (auto &x : th) x.join();
            }
            best[5] = std::min(best[5], now() - t);
        }
        printf("%3zu MiB: hipMemcpy H2D pageable %.3f ms (%.1f GB/s) | pinned %.3f ms (%.1f GB/s) | memcpy->pinned 1 thread %.3f ms (%.1f GB/s), 4 threads %.3f ms | "
               "register+copy+unregister %.3f ms | D2H pageable %.3f ms (%.1f GB/s)\n",
               mb, best[0] * 1e3, n / best[0] / 1e9, best[1] * 1e3, n / best[1] / 1e9, best[2] * 1e3, n / best[2] / 1e9, best[5] * 1e3,
               best[3] * 1e3, best[4] * 1e3, n / best[4] / 1e9);
        free(pageable);
        CK(hipHostFree(pinned));
        CK(hipFree(dev));
    }
    return 0;
}
