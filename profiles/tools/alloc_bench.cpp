#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
int main() {
    hipSetDevice(0);
    void* w; hipMalloc(&w, 1 << 20); hipFree(w);
    for (size_t mb : {16, 64, 256, 1024, 4096, 8192}) {
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = clk::now();
            void* p = nullptr;
            hipError_t e = hipMalloc(&p, mb << 20);
            auto t1 = clk::now();
            hipMemsetAsync(p, 0, 64, 0); hipDeviceSynchronize();
            auto t2 = clk::now();
            hipFree(p);
            auto t3 = clk::now();
            printf("%5zu MB: malloc %.3f ms (%d), free %.3f ms\n", mb, ms(t0, t1), (int)e, ms(t2, t3));
        }
    }
    // 45 buffers of 64 MB
    auto t0 = clk::now();
    std::vector<void*> v(45);
    for (auto& p : v) hipMalloc(&p, 64 << 20);
    auto t1 = clk::now();
    for (auto& p : v) hipFree(p);
    auto t2 = clk::now();
    printf("45 x 64 MB: malloc %.3f ms, free %.3f ms\n", ms(t0, t1), ms(t1, t2));
    // pinned host
    for (size_t mb : {64, 1024}) {
        auto a = clk::now(); void* h; hipHostMalloc(&h, mb << 20, hipHostMallocDefault); auto b = clk::now(); hipHostFree(h); auto c = clk::now();
        printf("hipHostMalloc %zu MB: %.3f ms, free %.3f ms\n", mb, ms(a, b), ms(b, c));
    }
    return 0;
}
