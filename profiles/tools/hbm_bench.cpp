// Achievable HBM rates of plain streaming kernels on this GPU (context for the roofline fractions in DESIGN.md):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_bench profiles/tools/hbm_bench.cpp && /tmp/hbm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_read(const f4* a, size_t n, float* out) {
    f4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}
__global__ void k_copy(const f4* a, f4* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_r2w1(const f4* a, const f4* b, f4* c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c[i] = a[i] + b[i];
}
__global__ void k_write(f4* c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c[i] = f4{1, 2, 3, 4};
}
template <typename F> static float timed(F f, int reps = 10) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
    for (size_t mb : {512, 2048}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        f4 *a, *b, *c; float* o;
        hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&o, 4);
        hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
        for (int grid : {2048, 8192}) {
            float t;
            t = timed([&] { k_read<<<grid, 256>>>(a, n, o); });            printf("%4zu MB grid %5d  read      %7.1f GB/s\n", mb, grid, bytes / t / 1e6);
            t = timed([&] { k_write<<<grid, 256>>>(c, n); });              printf("%4zu MB grid %5d  write     %7.1f GB/s\n", mb, grid, bytes / t / 1e6);
            t = timed([&] { k_copy<<<grid, 256>>>(a, b, n); });            printf("%4zu MB grid %5d  copy      %7.1f GB/s (read + written bytes)\n", mb, grid, 2 * bytes / t / 1e6);
            t = timed([&] { k_r2w1<<<grid, 256>>>(a, b, c, n); });         printf("%4zu MB grid %5d  2r + 1w   %7.1f GB/s\n", mb, grid, 3 * bytes / t / 1e6);
        }
        hipFree(a); hipFree(b); hipFree(c); hipFree(o);
    }
    return 0;
}
