// Prototype (round 5) of the second-generation bit-plane product: S = B . Qd on v_mfma_i32_16x16x64_i8 with
//   * the (sketch column, digit) pairs flattened into ONE N index: 40 columns x 4 digits = 160 = exactly ten 16-wide tiles
//     (the round-4 kernel padded 40 -> 48 columns and ran 12 tiles), 40 x 3 digits = 120 -> eight tiles;
//   * stages of 256 matrix columns (four MFMA k-steps) per barrier instead of 64, three LDS stages of digits;
//   * the bitmap words straight into registers (8 bytes per lane, tile and stage; a tile's 16 rows x 256 columns are one
//     contiguous 512-byte block), prefetched two stages ahead;
//   * the digit sums recombined in the epilogue (quad-lane adds of exact float64 integers), one float64 per (row, column).
// Stand-alone: random bitmap of the headline shape (125 000 x 10 000, 8.2 % ones), checked against a CPU recomputation on
// sampled rows, timed.   hipcc --offload-arch=gfx950 -O3 bp2_proto.hip -o bp2 && ./bp2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__host__ __device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__host__ __device__ inline bool bit_at(int64_t row, int64_t col, int64_t K) { return col < K && (mix((uint64_t)row * 1000003ull + (uint64_t)col) & 1023) < 84; }   // 8.2 %
__host__ __device__ inline int8_t digit_at(int64_t k, int col, int d) { return (int8_t)(mix(0x9e3779b97f4a7c15ull + (uint64_t)k * 131 + col * 7 + d) & 0xff); }

constexpr int kStageCols = 256;          // matrix columns per stage
constexpr int kSteps = kStageCols / 64;  // MFMA k-steps per stage

// bitmap: word (tile, sk, r, g) = columns sk*256 + g*64 .. +63 of row tile*16 + r;  layout [((tile * SK + sk) * 16 + r) * 4 + g]
__global__ void k_fill_bitmap(uint64_t* bm, int64_t ntile, int SK, int64_t K) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ntile * SK * 64) return;
    const int g = (int)(w & 3), r = (int)((w >> 2) & 15);
    const int64_t ts = w >> 6;
    const int64_t tile = ts / SK, sk = ts % SK;
    uint64_t v = 0;
    for (int b = 0; b < 64; ++b) v |= (uint64_t)bit_at(tile * 16 + r, sk * 256 + g * 64 + b, K) << b;
    bm[w] = v;
}
// digits: qd[((sk * kSteps + s) * NT + nt) * 64 + lane] = 16 bytes e = 0..15: flattened column f = nt*16 + (lane & 15) = col*ND + d,
// matrix column k = sk*256 + (lane >> 4)*64 + s*16 + e
template <int ND>
__global__ void k_fill_digits(v4i* qd, int SK, int NT, int L) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)SK * kSteps * NT * 64) return;
    const int lane = (int)(t & 63);
    int64_t u = t >> 6;
    const int nt = (int)(u % NT); u /= NT;
    const int s = (int)(u % kSteps);
    const int64_t sk = u / kSteps;
    const int n = lane & 15, g = lane >> 4;
    const int f = nt * 16 + n, col = f / ND, d = f % ND;
    int out[4];
    for (int wd = 0; wd < 4; ++wd) {
        unsigned x = 0;
        for (int e = 0; e < 4; ++e) x |= (unsigned)(uint8_t)(col < L ? digit_at(sk * 256 + g * 64 + s * 16 + wd * 4 + e, col, d) : 0) << (8 * e);
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

// RT tiles of 16 bitmap rows per wave, WAVES waves, NT flattened N tiles, ND digits per column.
// out[row][col] (float64, L columns) = sum_d 256^d * S_d[row][col]
template <int RT, int WAVES, int NT, int ND, int ABL = 0>
__global__ void __launch_bounds__(64 * WAVES) k_bp2(const uint64_t* __restrict__ bm, const v4i* __restrict__ qd, int64_t ntile, int SK, int L, int64_t nrows,
                                                    double* __restrict__ out) {
    constexpr int kStages = 3;
    constexpr int kVecs = kSteps * NT * 64;                 // 16-byte vectors of digits per stage
    constexpr int kPieces = kVecs / 64;                     // wave-wide copies (1 KB each) per stage
    constexpr int kLo = kPieces / WAVES, kExtra = kPieces % WAVES;
    extern __shared__ __align__(16) unsigned char smem[];
    v4i* lds = reinterpret_cast<v4i*>(smem);                // [kStages][kVecs]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tile0 = ((int64_t)blockIdx.x * WAVES + wave) * RT;
    const int r = lane & 15, g = lane >> 4;
    v4i acc[RT][NT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[t][c] = v4i{0, 0, 0, 0};
    const uint64_t* bmp[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t tile = tile0 + t < ntile ? tile0 + t : ntile - 1;
        bmp[t] = bm + (tile * SK * 16 + r) * 4 + g;
    }
    auto stage = [&](int sk, int st) {
        const int k = sk < SK ? sk : SK - 1;
#pragma unroll
        for (int u = 0; u <= kLo; ++u) {
            const int piece = u * WAVES + wave;
            if (u < kLo || wave < kExtra)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qd + (int64_t)k * kVecs + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(lds + st * kVecs + piece * 64), 16, 0, 0);
        }
    };
    auto bits = [&](int sk, uint64_t (&w)[RT]) {
        const int k = sk < SK ? sk : SK - 1;
#pragma unroll
        for (int t = 0; t < RT; ++t) w[t] = bmp[t][(int64_t)k * 64];
    };
    uint64_t w0[RT], w1[RT];
    bits(0, w0);
    stage(0, 0);
    stage(1, 1);
#pragma unroll 1
    for (int sk = 0; sk < SK; ++sk) {
        // the copies of stage sk have landed: everything but the newest stage's copies is complete (the bit loads were issued before them)
        if (ABL & 4) {}
        else if (wave < kExtra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLo + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLo) : "memory");
        if (!(ABL & 4)) __syncthreads();
        bits(sk + 1, w1);
        if (!(ABL & 4)) stage(sk + 2, (sk + 2) % kStages);
        const v4i* cur = lds + (sk % kStages) * kVecs;
#pragma unroll
        for (int s = 0; s < kSteps; ++s) {
            v4i a[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                if (ABL & 1) { const int x = (int)(w0[t] >> (16 * s)); a[t] = v4i{x, x, x, x}; }
                else a[t] = expand16((unsigned)(w0[t] >> (16 * s)) & 0xffffu);
            }
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const v4i b = (ABL & 2) ? cur[((c & 1)) * 64 + lane] : cur[(s * NT + c) * 64 + lane];
#pragma unroll
                for (int t = 0; t < RT; ++t) acc[t][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t], b, acc[t][c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) w0[t] = w1[t];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // C/D layout: column n = lane & 15, row = (lane >> 4) * 4 + reg.  f = nt*16 + n = col*ND + d: the ND digits of a column sit in
    // ND adjacent lanes; sum_d 256^d S_d as exact float64 integers, then lane (n % ND == j) keeps row reg j.
    const int n = lane & 15;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        if (tile0 + t >= ntile) continue;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const int f = c * 16 + n, col = f / ND, d = f % ND;
            const double sc = (double)(1u << (8 * d));
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (double)acc[t][c][q] * sc;
            if (ND == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] += __shfl_xor(v[q], 1, 64); v[q] += __shfl_xor(v[q], 2, 64); }
                const double mine = d == 0 ? v[0] : (d == 1 ? v[1] : (d == 2 ? v[2] : v[3]));
                const int64_t row = (tile0 + t) * 16 + g * 4 + d;
                if (row < nrows && col < L) out[row * L + col] = mine;
            } else {
                // three digits: lanes (d = 0, 1, 2) of a column may straddle nothing (16 % 3 != 0 -> a column can straddle two tiles): slow generic path
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t row = (tile0 + t) * 16 + g * 4 + q;
                    if (row < nrows && col < L) atomicAdd(out + row * L + col, v[q]);
                }
            }
        }
    }
}


// ---- 32 x 32 x 32 variant: tiles of 32 bitmap rows, flattened N tiles of 32, eight MFMA k-steps per 256-column stage ----
// bitmap: [((tile32 * SK + sk) * 32 + r) * 2 + h] 16 bytes = columns sk*256 + h*128 .. +127 of row tile32*32 + r
__global__ void k_fill_bitmap32(v4i* bm, int64_t ntile32, int SK, int64_t K) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ntile32 * SK * 64) return;
    const int h = (int)(w & 1), r = (int)((w >> 1) & 31);
    const int64_t ts = w >> 6;
    const int64_t tile = ts / SK, sk = ts % SK;
    unsigned v[4] = {0, 0, 0, 0};
    for (int b = 0; b < 128; ++b) v[b >> 5] |= (unsigned)bit_at(tile * 32 + r, sk * 256 + h * 128 + b, K) << (b & 31);
    bm[w] = v4i{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
}
// digits: qd[((sk * 8 + s) * NT + nt) * 64 + lane] = 16 bytes e: f = nt*32 + (lane & 31) = col*4 + d, k = sk*256 + (lane >> 5)*128 + s*16 + e
__global__ void k_fill_digits32(v4i* qd, int SK, int NT, int L) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)SK * 8 * NT * 64) return;
    const int lane = (int)(t & 63);
    int64_t u = t >> 6;
    const int nt = (int)(u % NT); u /= NT;
    const int s = (int)(u % 8);
    const int64_t sk = u / 8;
    const int n = lane & 31, h = lane >> 5;
    const int f = nt * 32 + n, col = f / 4, d = f % 4;
    int out[4];
    for (int wd = 0; wd < 4; ++wd) {
        unsigned x = 0;
        for (int e = 0; e < 4; ++e) x |= (unsigned)(uint8_t)(col < L ? digit_at(sk * 256 + h * 128 + s * 16 + wd * 4 + e, col, d) : 0) << (8 * e);
        out[wd] = (int)x;
    }
    qd[t] = v4i{out[0], out[1], out[2], out[3]};
}
typedef int v16i __attribute__((ext_vector_type(16)));
template <int RT, int WAVES, int NT, int ABL = 0>
__global__ void __launch_bounds__(64 * WAVES) k_bp3(const v4i* __restrict__ bm, const v4i* __restrict__ qd, int64_t ntile, int SK, int L, int64_t nrows,
                                                    double* __restrict__ out, long long* __restrict__ clk = nullptr) {
    const long long t_begin = __builtin_readcyclecounter();
    constexpr int kStages = 3;
    constexpr int kVecs = 8 * NT * 64;
    constexpr int kPieces = kVecs / 64;
    constexpr int kLo = kPieces / WAVES, kExtra = kPieces % WAVES;
    extern __shared__ __align__(16) unsigned char smem[];
    v4i* lds = reinterpret_cast<v4i*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tile0 = ((int64_t)blockIdx.x * WAVES + wave) * RT;
    v16i acc[RT][NT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][c][i] = 0;
    const v4i* bmp[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t tile = tile0 + t < ntile ? tile0 + t : ntile - 1;
        bmp[t] = bm + tile * SK * 64 + (lane & 31) * 2 + (lane >> 5);
    }
    auto stage = [&](int sk, int st) {
        const int k = sk < SK ? sk : SK - 1;
#pragma unroll
        for (int u = 0; u <= kLo; ++u) {
            const int piece = u * WAVES + wave;
            if (u < kLo || wave < kExtra)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qd + (int64_t)k * kVecs + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(lds + st * kVecs + piece * 64), 16, 0, 0);
        }
    };
    auto bits = [&](int sk, v4i (&w)[RT]) {
        const int k = sk < SK ? sk : SK - 1;
#pragma unroll
        for (int t = 0; t < RT; ++t) w[t] = bmp[t][(int64_t)k * 64];
    };
    v4i w0[RT], w1[RT];
    bits(0, w0);
    stage(0, 0);
    stage(1, 1);
#pragma unroll 1
    for (int sk = 0; sk < SK; ++sk) {
        if (ABL & 4) {}
        else if (wave < kExtra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLo + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLo) : "memory");
        if (!(ABL & 4)) __syncthreads();
        if (!(ABL & 8)) bits(sk + 1, w1);
        if (!(ABL & 4)) stage(sk + 2, (sk + 2) % kStages);
        const v4i* cur = lds + (sk % kStages) * kVecs + lane;
        // operand fragments are read PF groups ahead of the MFMAs that use them (the two waves of a SIMD leave the barrier together:
        // without the prefetch both sit in the same LDS round trip and the matrix pipe idles)
        constexpr int PF = 2, NG = 8 * NT;
        v4i bq[PF + 1];
#pragma unroll
        for (int p = 0; p < PF; ++p) bq[p] = cur[p * 64];
        v4i a[RT], an[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) a[t] = (ABL & 1) ? w0[t] : expand16((unsigned)w0[t][0] & 0xffffu);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int s = gi / NT, c = gi % NT;
            if (gi + PF < NG && (!(ABL & 2) || gi < 2)) bq[(gi + PF) % (PF + 1)] = cur[(gi + PF) * 64];
            if (c == 0 && s + 1 < 8) {
#pragma unroll
                for (int t = 0; t < RT; ++t) an[t] = (ABL & 1) ? w0[t] + s : expand16(((unsigned)w0[t][(s + 1) >> 1] >> (16 * ((s + 1) & 1))) & 0xffffu);
            }
            const v4i b = bq[gi % (PF + 1)];
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t], b, acc[t][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (c == NT - 1) {
#pragma unroll
                for (int t = 0; t < RT; ++t) a[t] = an[t];
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) w0[t] = w1[t];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (clk && threadIdx.x == 0 && blockIdx.x == 7) clk[0] = __builtin_readcyclecounter() - t_begin;
    // C/D layout (32 x 32): column n = lane & 31, register i holds row (i / 4) * 8 + (lane >> 5) * 4 + (i % 4)
    const int n = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        if (tile0 + t >= ntile) continue;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const int f = c * 32 + n, col = f >> 2, d = f & 3;
            const double sc = (double)(1u << (8 * d));
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] = (double)acc[t][c][i4 * 4 + q] * sc; v[q] += __shfl_xor(v[q], 1, 64); v[q] += __shfl_xor(v[q], 2, 64); }
                const double mine = d == 0 ? v[0] : (d == 1 ? v[1] : (d == 2 ? v[2] : v[3]));
                const int64_t row = (tile0 + t) * 32 + i4 * 8 + hh * 4 + d;
                if (row < nrows && col < L) out[row * L + col] = mine;
            }
        }
    }
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 125000, K = argc > 2 ? atoll(argv[2]) : 10000;
    const int L = 40;
    const int64_t ntile = (M + 15) / 16;
    const int SK = (int)((K + kStageCols - 1) / kStageCols);
    uint64_t* bm; v4i* qd; double* out;
    CK(hipMalloc(&bm, sizeof(uint64_t) * ntile * SK * 64));
    CK(hipMalloc(&qd, sizeof(v4i) * (size_t)SK * kSteps * 10 * 64));
    CK(hipMalloc(&out, sizeof(double) * ntile * 16 * L));
    k_fill_bitmap<<<(unsigned)((ntile * SK * 64 + 255) / 256), 256>>>(bm, ntile, SK, K);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int total_bad = 0;
    auto run = [&](auto kern, auto fill, int rt, int waves, int NT, int ND, const char* name) -> int {
        fill<<<(unsigned)(((int64_t)SK * kSteps * NT * 64 + 255) / 256), 256>>>(qd, SK, NT, L);
        const unsigned grid = (unsigned)((ntile + rt * waves - 1) / (rt * waves));
        const size_t lds = (size_t)3 * kSteps * NT * 64 * 16;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int i = 0; i < 3; ++i) { if (ND == 3) hipMemsetAsync(out, 0, sizeof(double) * ntile * 16 * L); kern<<<grid, 64 * waves, lds>>>(bm, qd, ntile, SK, L, M, out); }
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) kern<<<grid, 64 * waves, lds>>>(bm, qd, ntile, SK, L, M, out);
        hipEventRecord(e1);
        CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ops = 2.0 * ntile * 16 * (double)SK * 256 * NT * 16;
        printf("%s: %.3f ms per launch (grid %u, lds %zu): %.0f TOPS issued\n", name, ms / 10, grid, lds, ops / (ms / 10 * 1e-3) / 1e12);
        if (ND == 3) return 0;
        // check sampled rows
        std::vector<double> h(L);
        int bad = 0;
        for (int64_t row : {int64_t(0), int64_t(17), int64_t(4242), M - 1}) {
            CK(hipMemcpy(h.data(), out + row * L, sizeof(double) * L, hipMemcpyDeviceToHost));
            for (int col = 0; col < L; col += 3) {
                double ref = 0;
                for (int d = 0; d < ND; ++d) {
                    long s = 0;
                    for (int64_t k = 0; k < (int64_t)SK * 256; ++k) if (bit_at(row, k, K)) s += digit_at(k, col, d);
                    ref += (double)s * (double)(1u << (8 * d));
                }
                if (ref != h[col]) { if (bad < 5) printf("mismatch row %lld col %d: %.1f vs %.1f\n", (long long)row, col, h[col], ref); ++bad; }
            }
        }
        printf("   check: %d mismatches\n", bad);
        total_bad += bad;
        return 0;
    };
    {   // 32 x 32 x 32 variant
        const int64_t ntile32 = (M + 31) / 32;
        v4i* bm32; long long* clk; CK(hipMalloc(&clk, 64));
        CK(hipMalloc(&bm32, sizeof(v4i) * ntile32 * SK * 64));
        k_fill_bitmap32<<<(unsigned)((ntile32 * SK * 64 + 255) / 256), 256>>>(bm32, ntile32, SK, K);
        k_fill_digits32<<<(unsigned)(((int64_t)SK * 8 * 5 * 64 + 255) / 256), 256>>>(qd, SK, 5, L);
        CK(hipDeviceSynchronize());
        auto run32 = [&](auto kern, int rt, int waves, const char* name) -> int {
            const unsigned grid = (unsigned)((ntile32 + rt * waves - 1) / (rt * waves));
            const size_t lds = (size_t)3 * 8 * 5 * 64 * 16;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            for (int i = 0; i < 3; ++i) kern<<<grid, 64 * waves, lds>>>(bm32, qd, ntile32, SK, L, M, out, clk);
            CK(hipDeviceSynchronize());
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) kern<<<grid, 64 * waves, lds>>>(bm32, qd, ntile32, SK, L, M, out, clk);
            hipEventRecord(e1);
            CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long hclk = 0; CK(hipMemcpy(&hclk, clk, 8, hipMemcpyDeviceToHost));
            printf("   main loop of one wave: %lld s_memtime ticks (= %.3f ms at 100 MHz, %.3f ms at 2.4 GHz)\n", hclk, hclk / 1e5, hclk / 2.4e6);
            const double ops = 2.0 * ntile32 * 32 * (double)SK * 256 * 160;
            printf("%s: %.3f ms per launch (grid %u): %.0f TOPS issued\n", name, ms / 10, grid, ops / (ms / 10 * 1e-3) / 1e12);
            std::vector<double> h(L);
            int bad = 0;
            for (int64_t row : {int64_t(0), int64_t(17), int64_t(4242), M - 1}) {
                CK(hipMemcpy(h.data(), out + row * L, sizeof(double) * L, hipMemcpyDeviceToHost));
                for (int col = 0; col < L; col += 3) {
                    double ref = 0;
                    for (int d = 0; d < 4; ++d) {
                        long sm = 0;
                        for (int64_t k = 0; k < (int64_t)SK * 256; ++k) if (bit_at(row, k, K)) sm += digit_at(k, col, d);
                        ref += (double)sm * (double)(1u << (8 * d));
                    }
                    if (ref != h[col]) { if (bad < 5) printf("mismatch row %lld col %d: %.1f vs %.1f\n", (long long)row, col, h[col], ref); ++bad; }
                }
            }
            printf("   check: %d mismatches\n", bad);
            total_bad += bad;
            return 0;
        };
        run32(k_bp3<2, 8, 5>, 2, 8, "32x32x32 RT=2 waves=8");
        run32(k_bp3<2, 8, 5, 1>, 2, 8, "  abl 1: no expansion");
        run32(k_bp3<2, 8, 5, 2>, 2, 8, "  abl 2: four LDS reads per stage");
        run32(k_bp3<2, 8, 5, 4>, 2, 8, "  abl 4: no staging / barrier");
        run32(k_bp3<2, 8, 5, 8>, 2, 8, "  abl 8: no bit loads");
        run32(k_bp3<2, 8, 5, 5>, 2, 8, "  abl 1+4");
        run32(k_bp3<2, 8, 5, 6>, 2, 8, "  abl 2+4");
        run32(k_bp3<2, 8, 5, 15>, 2, 8, "  abl all");
        if (argc > 3) return 0;
    }
    run(k_bp2<4, 8, 10, 4>, k_fill_digits<4>, 4, 8, 10, 4, "RT=4 waves=8 NT=10 (4 digits)");
    if (argc > 3) return 0;
    run(k_bp2<4, 8, 10, 4, 1>, k_fill_digits<4>, 4, 8, 10, 4, "  ablation: no bit expansion");
    run(k_bp2<4, 8, 10, 4, 2>, k_fill_digits<4>, 4, 8, 10, 4, "  ablation: two LDS operand reads per step only");
    run(k_bp2<4, 8, 10, 4, 4>, k_fill_digits<4>, 4, 8, 10, 4, "  ablation: no staging, no barrier");
    run(k_bp2<4, 8, 10, 4, 7>, k_fill_digits<4>, 4, 8, 10, 4, "  ablation: all three");
    run(k_bp2<3, 8, 10, 4>, k_fill_digits<4>, 3, 8, 10, 4, "RT=3 waves=8 NT=10 (4 digits)");
    run(k_bp2<2, 8, 10, 4>, k_fill_digits<4>, 2, 8, 10, 4, "RT=2 waves=8 NT=10 (4 digits)");
    run(k_bp2<4, 4, 10, 4>, k_fill_digits<4>, 4, 4, 10, 4, "RT=4 waves=4 NT=10 (4 digits)");
    run(k_bp2<8, 4, 10, 4>, k_fill_digits<4>, 8, 4, 10, 4, "RT=8 waves=4 NT=10 (4 digits)");
    run(k_bp2<6, 4, 10, 4>, k_fill_digits<4>, 6, 4, 10, 4, "RT=6 waves=4 NT=10 (4 digits)");
    run(k_bp2<4, 8, 8, 3>, k_fill_digits<3>, 4, 8, 8, 3, "RT=4 waves=8 NT=8 (3 digits, timing only)");
    printf("total mismatches %d\n", total_bad);
    return 0;
}
