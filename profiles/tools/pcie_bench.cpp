// host -> device copy rate of this box from pinned memory, whole and in 16 chunks, and the rate at which N host threads
// rewrite a (int32, float32) entry stream into 4-byte and 2-byte packed forms:  hipcc -O2 -pthread pcie_bench.cpp -o pcie_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t nnz = 93500000;
    void *pin = nullptr, *dev = nullptr;
    hipHostMalloc(&pin, nnz * 4, hipHostMallocPortable);
    hipMalloc(&dev, nnz * 4);
    memset(pin, 1, nnz * 4);
    hipStream_t s; hipStreamCreate(&s);
    for (size_t bytes : {nnz * 4, nnz * 2}) {
        for (int chunks : {1, 16}) {
            for (int rep = 0; rep < 3; ++rep) {
                const double t0 = now();
                for (int k = 0; k < chunks; ++k)
                    hipMemcpyAsync((char*)dev + bytes / chunks * k, (char*)pin + bytes / chunks * k, bytes / chunks, hipMemcpyHostToDevice, s);
                hipStreamSynchronize(s);
                const double t = now() - t0;
                if (rep == 2) printf("H2D %zu MB in %d chunk(s): %.2f ms = %.1f GB/s\n", bytes >> 20, chunks, t * 1e3, bytes / t / 1e9);
            }
        }
    }
    std::vector<int32_t> idx(nnz); std::vector<float> val(nnz);
    for (size_t i = 0; i < nnz; ++i) { idx[i] = (int32_t)(i * 31 % 30000); val[i] = (float)(1 + i % 5); }
    for (int T : {16, 24, 32, 48, 64, 96, 128}) {
        for (int form : {4, 2}) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                const double t0 = now();
                std::vector<std::thread> th;
                for (int w = 0; w < T; ++w) th.emplace_back([&, w] {
                    const size_t a = nnz * w / T, b = nnz * (w + 1) / T;
                    if (form == 4) { uint32_t* o = (uint32_t*)pin; for (size_t i = a; i < b; ++i) o[i] = (uint32_t)idx[i] | ((uint32_t)(int32_t)val[i] << 16); }
                    else { uint16_t* o = (uint16_t*)pin; int32_t prev = -1; for (size_t i = a; i < b; ++i) { const int32_t d = idx[i] - prev; prev = idx[i]; o[i] = (uint16_t)((d & 255) | ((uint32_t)(int32_t)val[i] << 8)); } }
                });
                for (auto& t : th) t.join();
                best = std::min(best, now() - t0);
            }
            printf("pack %d-byte form, %3d threads: %.2f ms\n", form, T, best * 1e3);
        }
    }
    return 0;
}
