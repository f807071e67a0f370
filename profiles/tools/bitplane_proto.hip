// Prototype of the bit-plane operator product (round 4): S = B . Qd where B is a 0/1 matrix kept as a bitmap (one bit per
// (row, column): "this entry of the count matrix is a 1") and Qd the operand cut into signed 8-bit digits, on
// v_mfma_i32_16x16x64_i8 -- exact integer arithmetic.  Stand-alone: random bitmap of the headline shape and density,
// checked against a CPU recomputation on sampled rows, timed.   hipcc --offload-arch=gfx950 -O3 bitplane_proto.hip -o bp && ./bp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__host__ __device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__host__ __device__ inline bool bit_at(int64_t row, int64_t col) { return (mix((uint64_t)row * 1000003ull + (uint64_t)col) & 1023) < 84; }   // 8.2 %
__host__ __device__ inline int8_t digit_at(int64_t k, int col, int d) { return (int8_t)(mix(0x9e3779b97f4a7c15ull + (uint64_t)k * 131 + col * 7 + d) & 0xff); }

constexpr int NCB = 3, ND = 4;     // 48 output columns, 4 digits
__global__ void k_fill_bitmap(uint64_t* bm, int64_t ntile, int KB) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ntile * KB * 16) return;
    const int r = (int)(w & 15);
    const int64_t tk = w >> 4;
    const int64_t tile = tk / KB, kb = tk % KB;
    uint64_t v = 0;
    for (int b = 0; b < 64; ++b) v |= (uint64_t)bit_at(tile * 16 + r, kb * 64 + b) << b;
    bm[w] = v;
}
__global__ void k_fill_digits(v4i* qd, int KB) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)KB * NCB * ND * 64) return;
    const int lane = (int)(t & 63);
    int64_t u = t >> 6;
    const int d = (int)(u % ND); u /= ND;
    const int cb = (int)(u % NCB);
    const int64_t kb = u / NCB;
    const int n = lane & 15, g = lane >> 4;
    int out[4];
    for (int wd = 0; wd < 4; ++wd) {
        unsigned x = 0;
        for (int e = 0; e < 4; ++e) x |= (unsigned)(uint8_t)digit_at(kb * 64 + g * 16 + wd * 4 + e, cb * 16 + n, d) << (8 * e);
        out[wd] = (int)x;
    }
    qd[t] = v4i{out[0], out[1], out[2], out[3]};
}

// 16 bits -> 16 bytes of 0 / 1
__device__ __forceinline__ v4i expand16(unsigned bits) {
    v4i r;
#pragma unroll
    for (int w = 0; w < 4; ++w) r[w] = (int)((((bits >> (4 * w)) & 0xfu) * 0x00204081u) & 0x01010101u);
    return r;
}

template <int RT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_bitplane(const uint64_t* __restrict__ bm, const v4i* __restrict__ qd, int64_t ntile, int KB, int* __restrict__ out) {
    __shared__ v4i lds[2][NCB * ND * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tile0 = ((int64_t)blockIdx.x * WAVES + wave) * RT;
    const int r = lane & 15, g = lane >> 4;
    v4i acc[RT][NCB][ND];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int d = 0; d < ND; ++d) acc[t][c][d] = v4i{0, 0, 0, 0};
    auto stage = [&](int kb, int buf) {
        for (int v = tid; v < NCB * ND * 64; v += 64 * WAVES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qd + (int64_t)kb * NCB * ND * 64 + v),
                                             (__attribute__((address_space(3))) void*)(lds[buf] + (v - lane)), 16, 0, 0);
    };
    uint64_t wcur[RT], wnext[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t tile = tile0 + t < ntile ? tile0 + t : ntile - 1;
        wcur[t] = bm[(tile * KB) * 16 + r];
    }
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = 0; kb < KB; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < KB) {
            stage(kb + 1, buf ^ 1);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const int64_t tile = tile0 + t < ntile ? tile0 + t : ntile - 1;
                wnext[t] = bm[(tile * KB + kb + 1) * 16 + r];
            }
        }
        v4i a[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) a[t] = expand16((unsigned)(wcur[t] >> (16 * g)) & 0xffffu);
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const v4i b = lds[buf][(c * ND + d) * 64 + lane];
#pragma unroll
                for (int t = 0; t < RT; ++t) acc[t][c][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t], b, acc[t][c][d], 0, 0, 0);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RT; ++t) wcur[t] = wnext[t];
    }
    // out[row][cb*16+n][d]: C/D layout col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        if (tile0 + t >= ntile) continue;
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t row = (tile0 + t) * 16 + g * 4 + q;
                    out[(row * (NCB * 16) + c * 16 + r) * ND + d] = acc[t][c][d][q];
                }
    }
}

int main() {
    const int64_t N = 100000, H = 10000;
    const int64_t ntile = (N + 15) / 16;
    const int KB = (int)((H + 63) / 64);
    uint64_t* bm; v4i* qd; int* out;
    CK(hipMalloc(&bm, sizeof(uint64_t) * ntile * KB * 16));
    CK(hipMalloc(&qd, sizeof(v4i) * (size_t)KB * NCB * ND * 64));
    CK(hipMalloc(&out, sizeof(int) * ntile * 16 * NCB * 16 * ND));
    k_fill_bitmap<<<(unsigned)((ntile * KB * 16 + 255) / 256), 256>>>(bm, ntile, KB);
    k_fill_digits<<<(unsigned)(((int64_t)KB * NCB * ND * 64 + 255) / 256), 256>>>(qd, KB);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, int rt, int waves, const char* name) {
        const unsigned grid = (unsigned)((ntile + rt * waves - 1) / (rt * waves));
        for (int i = 0; i < 3; ++i) kern<<<grid, 64 * waves>>>(bm, qd, ntile, KB, out);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) kern<<<grid, 64 * waves>>>(bm, qd, ntile, KB, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms per launch (grid %u)\n", name, ms / 10, grid);
    };
    run(k_bitplane<2, 4>, 2, 4, "RT=2 waves=4");
    run(k_bitplane<4, 2>, 4, 2, "RT=4 waves=2");
    run(k_bitplane<4, 4>, 4, 4, "RT=4 waves=4");
    run(k_bitplane<2, 8>, 2, 8, "RT=2 waves=8");
    run(k_bitplane<1, 8>, 1, 8, "RT=1 waves=8");
    // check sampled rows (last config's output)
    std::vector<int> h((size_t)NCB * 16 * ND);
    int bad = 0;
    for (int64_t row : {int64_t(0), int64_t(17), int64_t(4242), int64_t(99999)}) {
        CK(hipMemcpy(h.data(), out + row * NCB * 16 * ND, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
        for (int col = 0; col < 48; col += 7)
            for (int d = 0; d < ND; ++d) {
                long ref = 0;
                for (int64_t k = 0; k < (int64_t)KB * 64; ++k) if (bit_at(row, k)) ref += digit_at(k, col, d);
                if (ref != h[col * ND + d]) { if (bad < 5) printf("mismatch row %lld col %d digit %d: %d vs %ld\n", (long long)row, col, d, h[col * ND + d], ref); ++bad; }
            }
    }
    printf("check: %d mismatches\n", bad);
    return 0;
}
