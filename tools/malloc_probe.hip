// hipMalloc latency by size (first-call cost of the library's workspaces): tools/bin/malloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main()
{
    hipFree(0);
    const double gib[] = {1, 4, 8, 12, 14, 15, 16, 17, 18, 20, 24, 28, 32, 48, 64};
    for (double g : gib) {
        void *p = nullptr;
        const size_t n = (size_t)(g * 1073741824.0);
        auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, n);
        auto t1 = std::chrono::steady_clock::now();
        hipFree(p);
        auto t2 = std::chrono::steady_clock::now();
        printf("%5.1f GiB: hipMalloc %9.2f ms (%s), hipFree %9.2f ms\n", g, std::chrono::duration<double, std::milli>(t1 - t0).count(),
               hipGetErrorString(e), std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    // second round: does a freed large block come back fast?
    for (double g : {24.0, 64.0}) {
        void *p = nullptr;
        auto t0 = std::chrono::steady_clock::now();
        hipMalloc(&p, (size_t)(g * 1073741824.0));
        auto t1 = std::chrono::steady_clock::now();
        hipFree(p);
        printf("again %5.1f GiB: hipMalloc %9.2f ms\n", g, std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
    return 0;
}
