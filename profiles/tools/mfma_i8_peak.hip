// Issue-rate ceiling of the int8 matrix instructions on gfx950 (round 5): register operands only, independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 mfma_i8_peak.hip -o mfma_i8_peak.bin && ./mfma_i8_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int NACC, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k16(int iters, int* out, int seed) {
    v4i acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4i{0, 0, 0, 0};
    v4i a = v4i{seed + (int)threadIdx.x, seed, 1, 2}, b = v4i{seed, 3, (int)threadIdx.x, 5};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k32(int iters, int* out, int seed) {
    v16i acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    v4i a = v4i{seed + (int)threadIdx.x, seed, 1, 2}, b = v4i{seed, 3, (int)threadIdx.x, 5};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    int* out; hipMalloc(&out, 4 * 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int nacc, int waves, double ops_per, const char* name) {
        const int iters = 2000;
        kern<<<256, 64 * waves>>>(10, out, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<256, 64 * waves>>>(iters, out, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = 256.0 * waves * iters * nacc;
        printf("%s: %.3f ms, %.1f cycles per MFMA per SIMD at 2.4 GHz, %.0f TOPS\n", name, ms, ms * 1e-3 * 2.4e9 / (n / 1024), n * ops_per / (ms * 1e-3) / 1e12);
    };
    run(k16<16, 4>, 16, 4, 32768.0, "16x16x64 16 acc, 1 wave/SIMD");
    run(k16<16, 8>, 16, 8, 32768.0, "16x16x64 16 acc, 2 waves/SIMD");
    run(k16<40, 8>, 40, 8, 32768.0, "16x16x64 40 acc, 2 waves/SIMD");
    run(k16<16, 16>, 16, 16, 32768.0, "16x16x64 16 acc, 4 waves/SIMD");
    run(k32<4, 4>, 4, 4, 65536.0, "32x32x32 4 acc, 1 wave/SIMD");
    run(k32<10, 8>, 10, 8, 65536.0, "32x32x32 10 acc, 2 waves/SIMD");
    run(k32<4, 16>, 4, 16, 65536.0, "32x32x32 4 acc, 4 waves/SIMD");
    return 0;
}
