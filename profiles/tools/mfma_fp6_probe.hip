// Round 6: could the bit-plane products run on the MX matrix instructions of gfx950?  v_mfma_f32_32x32x64_f8f6f4 with the bitmap as FP4
// (E2M1: 0 -> 0, 1 -> code 2 = 1.0) and the operand digits as FP6 (E2M3: an integer v in [-15, 15] is the code |v| | sign << 5 = v / 8, exact)
// accumulates exact integers / 8 in float32 below 2^24 / 8 -- at twice the int8 rate on paper.  This probe answers two questions on the device:
//   1. layout: which (register, bit field) of the A / B operands pairs with which, and are the sums exact (against a host product);
//   2. rate under load: random digits through the FP4 x FP6 form against the int8 form (the int8 product kernel is POWER-bound at ~1.6 GHz
//      with random digits, profiles/r05_bitplane_notes.txt section 2 -- a paper rate says nothing).
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/mfma_fp6_probe.hip -o /tmp/mfma_fp6_probe.bin && /tmp/mfma_fp6_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

// ---- 1. layout ---------------------------------------------------------------------------------------------------------------
// one wave: D = A (FP4, 32 x 64) . B (FP6 E2M3, 64 x 32), operands as the host packed them, scale bytes from the arguments
__global__ void k_one(const v8i* __restrict__ A, const v8i* __restrict__ B, float* __restrict__ D, int sa, int sb) {
    const int lane = threadIdx.x;
    v16f acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[lane], B[lane], acc, 4, 2, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) D[lane * 16 + i] = acc[i];
}
// the same with the scales as literal zeros (the compiler selects the unscaled instruction)
__global__ void k_one_noscale(const v8i* __restrict__ A, const v8i* __restrict__ B, float* __restrict__ D) {
    const int lane = threadIdx.x;
    v16f acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[lane], B[lane], acc, 4, 2, 0, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[lane * 16 + i] = acc[i];
}

static void set_bits(uint32_t* regs, int bit0, int nbits, uint32_t code) {
    for (int b = 0; b < nbits; ++b)
        if ((code >> b) & 1u) regs[(bit0 + b) >> 5] |= 1u << ((bit0 + b) & 31);
}

// ---- 2. rate -----------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(512) k_rate_i8(const v4i* __restrict__ A, const v4i* __restrict__ B, int iters, int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    v16i acc[2][NT];
    for (int t = 0; t < 2; ++t) for (int c = 0; c < NT; ++c) for (int i = 0; i < 16; ++i) acc[t][c][i] = 0;
    v4i a[2][2], b[2][NT];
    for (int s = 0; s < 2; ++s) {
        for (int t = 0; t < 2; ++t) a[s][t] = A[(s * 2 + t) * 64 + lane];
        for (int c = 0; c < NT; ++c) b[s][c] = B[(s * NT + c) * 64 + lane];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s][t], b[s][c], acc[t][c], 0, 0, 0);
    }
    int s = 0;
    for (int t = 0; t < 2; ++t) for (int c = 0; c < NT; ++c) for (int i = 0; i < 16; ++i) s += acc[t][c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int RT, int NT>
__global__ void __launch_bounds__(512) k_rate_f6(const v8i* __restrict__ A, const v8i* __restrict__ B, int iters, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    v16f acc[RT][NT];
    for (int t = 0; t < RT; ++t) for (int c = 0; c < NT; ++c) for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
    v8i a[2][RT], b[2][NT];
    for (int s = 0; s < 2; ++s) {
        for (int t = 0; t < RT; ++t) a[s][t] = A[(s * 2 + t) * 64 + lane];
        for (int c = 0; c < NT; ++c) b[s][c] = B[(s * 8 + c) * 64 + lane];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int t = 0; t < RT; ++t) acc[t][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s][t], b[s][c], acc[t][c], 4, 2, 0, 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < RT; ++t) for (int c = 0; c < NT; ++c) for (int i = 0; i < 16; ++i) s += acc[t][c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    srand(12345);
    // ---- layout: element e of lane (h = lane >> 5, r = lane & 31) is k = 32 h + e; FP4 nibble e at bit 4 e, FP6 field e at bit 6 e
    std::vector<int> Am(32 * 64), Bm(64 * 32);
    for (auto& v : Am) v = (rand() % 100) < 30 ? 1 : 0;
    for (auto& v : Bm) v = rand() % 31 - 15;
    std::vector<uint32_t> Ah(64 * 8, 0u), Bh(64 * 8, 0u);
    for (int lane = 0; lane < 64; ++lane) {
        const int h = lane >> 5, r = lane & 31;
        for (int e = 0; e < 32; ++e) {
            const int k = 32 * h + e;
            if (Am[r * 64 + k]) set_bits(&Ah[lane * 8], 4 * e, 4, 2u);
            const int v = Bm[k * 32 + r];
            set_bits(&Bh[lane * 8], 6 * e, 6, (uint32_t)(v < 0 ? 32 - v : v));           // sign << 5 | |v|
        }
    }
    v8i *dA, *dB; float* dD;
    hipMalloc(&dA, 64 * 32); hipMalloc(&dB, 64 * 32); hipMalloc(&dD, 64 * 16 * 4);
    hipMemcpy(dA, Ah.data(), 64 * 32, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bh.data(), 64 * 32, hipMemcpyHostToDevice);
    std::vector<float> D(64 * 16);
    auto check = [&](const char* what) {
        hipDeviceSynchronize();
        hipMemcpy(D.data(), dD, 64 * 16 * 4, hipMemcpyDeviceToHost);
        int bad = 0; double first_got = 0, first_want = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), col = lane & 31;
                int s = 0;
                for (int k = 0; k < 64; ++k) s += Am[row * 64 + k] * Bm[k * 32 + col];
                if (D[lane * 16 + i] != (float)s / 8.f) { if (!bad) { first_got = D[lane * 16 + i]; first_want = s / 8.0; } ++bad; }
            }
        printf("layout [%s]: %d of 1024 outputs differ from the host product (first: got %g want %g)\n", what, bad, first_got, first_want);
        return bad;
    };
    k_one_noscale<<<1, 64>>>(dA, dB, dD);
    const int bad0 = check("scale arguments literal 0: unscaled instruction");
    k_one<<<1, 64>>>(dA, dB, dD, 127, 127);
    check("scale bytes 127 (2^0) at run time");
    k_one<<<1, 64>>>(dA, dB, dD, 0, 0);
    check("scale bytes 0 at run time");
    if (bad0) {
        // which field pairs with which: A row 0 = one nibble p, B column 0 = one field q (code 8 = 1.0) -> D[0][0]
        printf("pairing of FP4 nibble p (rows) with FP6 field q (columns), lane half 0; '1' where D[0][0] == 1:\n");
        for (int p = 0; p < 32; ++p) {
            for (int q = 0; q < 32; ++q) {
                std::fill(Ah.begin(), Ah.end(), 0u); std::fill(Bh.begin(), Bh.end(), 0u);
                set_bits(&Ah[0], 4 * p, 4, 2u);
                set_bits(&Bh[0], 6 * q, 6, 8u);
                hipMemcpy(dA, Ah.data(), 64 * 32, hipMemcpyHostToDevice);
                hipMemcpy(dB, Bh.data(), 64 * 32, hipMemcpyHostToDevice);
                k_one_noscale<<<1, 64>>>(dA, dB, dD);
                hipDeviceSynchronize();
                hipMemcpy(D.data(), dD, 4, hipMemcpyDeviceToHost);
                putchar(D[0] == 1.f ? '1' : (D[0] == 0.f ? '.' : '?'));
            }
            putchar('\n');
        }
    }
    // exactness of long sums: 2048 accumulating instructions of all-ones x 15 / 8 = 64 x 15 / 8 each -> 245 760 exactly
    // ---- rate
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<uint32_t> Ar8(4 * 64 * 4), Br8(2 * 6 * 64 * 4), Ar6(4 * 64 * 8, 0u), Br6(2 * 8 * 64 * 8, 0u);
    for (auto& w : Ar8) { w = 0; for (int b = 0; b < 4; ++b) if (rand() % 100 < 9) w |= 1u << (8 * b); }
    for (auto& w : Br8) w = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
    for (size_t l = 0; l < Ar6.size() / 8; ++l)
        for (int e = 0; e < 32; ++e) if (rand() % 100 < 9) set_bits(&Ar6[l * 8], 4 * e, 4, 2u);
    for (size_t l = 0; l < Br6.size() / 8; ++l)
        for (int e = 0; e < 32; ++e) { const int v = rand() % 31 - 15; set_bits(&Br6[l * 8], 6 * e, 6, (uint32_t)(v < 0 ? 32 - v : v)); }
    v4i *dA8, *dB8; v8i *dA6, *dB6; int* o8; float* o6;
    hipMalloc(&dA8, Ar8.size() * 4); hipMalloc(&dB8, Br8.size() * 4); hipMalloc(&dA6, Ar6.size() * 4); hipMalloc(&dB6, Br6.size() * 4);
    hipMalloc(&o8, 256 * 512 * 4); hipMalloc(&o6, 256 * 512 * 4);
    hipMemcpy(dA8, Ar8.data(), Ar8.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB8, Br8.data(), Br8.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dA6, Ar6.data(), Ar6.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB6, Br6.data(), Br6.size() * 4, hipMemcpyHostToDevice);
    auto timeit = [&](auto launch, double macs_per_iter_wave, const char* name) {
        launch(200);
        hipDeviceSynchronize();
        const int iters = 40000;                           // tens of milliseconds: the power management has settled
        hipEventRecord(e0);
        launch(iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double macs = 256.0 * 8 * iters * macs_per_iter_wave;
        printf("%-58s %8.2f ms  %7.0f T MAC/s  (%.0f TOP/s)\n", name, ms, macs / (ms * 1e-3) / 1e12, 2 * macs / (ms * 1e-3) / 1e12);
    };
    // does the digits' bit pattern matter to a power-bound kernel?  balanced digits (random sign: the upper bits toggle) against
    // non-negative 7-bit digits and against 4-bit ones
    std::vector<uint32_t> Br7(Br8.size()), Br4(Br8.size());
    for (auto& w : Br7) w = ((uint32_t)rand() ^ ((uint32_t)rand() << 16)) & 0x7f7f7f7fu;
    for (auto& w : Br4) w = ((uint32_t)rand() ^ ((uint32_t)rand() << 16)) & 0x0f0f0f0fu;
    v4i *dB7, *dB4;
    hipMalloc(&dB7, Br7.size() * 4); hipMalloc(&dB4, Br4.size() * 4);
    hipMemcpy(dB7, Br7.data(), Br7.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB4, Br4.data(), Br4.size() * 4, hipMemcpyHostToDevice);
    // the int8 rate against the number of column tiles per wave (accumulator registers: 32 per tile pair)
    timeit([&](int it) { k_rate_i8<2><<<256, 512>>>(dA8, dB8, it, o8); }, 2.0 * 2 * 2 * 32768.0, "int8 32x32x32, 2 x 2 tiles, random digits");
    timeit([&](int it) { k_rate_i8<3><<<256, 512>>>(dA8, dB8, it, o8); }, 2.0 * 3 * 2 * 32768.0, "int8 32x32x32, 2 x 3 tiles, random digits");
    timeit([&](int it) { k_rate_i8<4><<<256, 512>>>(dA8, dB8, it, o8); }, 2.0 * 4 * 2 * 32768.0, "int8 32x32x32, 2 x 4 tiles, random digits");
    timeit([&](int it) { k_rate_i8<5><<<256, 512>>>(dA8, dB8, it, o8); }, 2.0 * 5 * 2 * 32768.0, "int8 32x32x32, 2 x 5 tiles, random digits");
    timeit([&](int it) { k_rate_i8<6><<<256, 512>>>(dA8, dB8, it, o8); }, 2.0 * 6 * 2 * 32768.0, "int8 32x32x32, 2 x 6 tiles, random digits");
    timeit([&](int it) { k_rate_i8<5><<<256, 256>>>(dA8, dB8, it, o8); }, 1.0 * 5 * 2 * 32768.0, "int8 32x32x32, 2 x 5 tiles, ONE wave per SIMD");
    timeit([&](int it) { k_rate_i8<4><<<256, 256>>>(dA8, dB8, it, o8); }, 1.0 * 4 * 2 * 32768.0, "int8 32x32x32, 2 x 4 tiles, ONE wave per SIMD");
    for (int rep = 0; rep < 2; ++rep) {
        timeit([&](int it) { k_rate_i8<5><<<256, 512>>>(dA8, dB7, it, o8); }, 2.0 * 5 * 2 * 32768.0, "int8 32x32x32, 2 x 5 tiles, non-negative 7-bit digits");
        timeit([&](int it) { k_rate_i8<5><<<256, 512>>>(dA8, dB4, it, o8); }, 2.0 * 5 * 2 * 32768.0, "int8 32x32x32, 2 x 5 tiles, non-negative 4-bit digits");
        timeit([&](int it) { k_rate_i8<5><<<256, 512>>>(dA8, dB8, it, o8); }, 2.0 * 5 * 2 * 32768.0, "int8 32x32x32, 2 x 5 tiles, random digits (4 digits)");
        timeit([&](int it) { k_rate_i8<4><<<256, 512>>>(dA8, dB8, it, o8); }, 2.0 * 4 * 2 * 32768.0, "int8 32x32x32, 2 x 4 tiles, random digits (3 digits)");
        timeit([&](int it) { k_rate_f6<2, 4><<<256, 512>>>(dA6, dB6, it, o6); }, 2.0 * 4 * 2 * 65536.0, "FP4 x FP6 32x32x64, 2 x 4 tiles, random digits");
        timeit([&](int it) { k_rate_f6<1, 8><<<256, 512>>>(dA6, dB6, it, o6); }, 2.0 * 8 * 1 * 65536.0, "FP4 x FP6 32x32x64, 1 x 8 tiles, random digits");
        timeit([&](int it) { k_rate_f6<1, 7><<<256, 512>>>(dA6, dB6, it, o6); }, 2.0 * 7 * 1 * 65536.0, "FP4 x FP6 32x32x64, 1 x 7 tiles, random digits");
    }
    return 0;
}
