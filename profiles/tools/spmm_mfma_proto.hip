// Prototype (not part of libddx): Y = A Q for a sparse A (CSR, ~10 % dense, 125 000 x 10 000) and a 40-column float32 operand on the
// bfloat16 matrix pipe -- 16 x 32 windows of A densified in LDS, values and operand cut into three bf16 pieces, six products per
// tile ("plan for the next round").  Prints the error against a float64 gather and the launch time.
//   hipcc -O3 --offload-arch=gfx950 spmm_mfma_proto.hip -o spmm_mfma_proto && ./spmm_mfma_proto
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int NB = 3, NP = 3;          // 16-column blocks of the operand (48 >= 40), bf16 pieces
constexpr int KBS = 12;                // blocks of 32 operand rows per LDS slice (9 KB each)
constexpr int WAVES = 16, RB = 2;      // waves per workgroup, 16-row blocks per wave
constexpr int LDQ = 48;
#ifndef PROTO_ABL
#define PROTO_ABL 0      // ablations (wrong results): 1 no MFMAs, 2 no entry loads, 4 no image traffic
#endif

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_reference(const int32_t* indptr, const int32_t* cols, const float* vals, const float* Q, int M, double* Y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M || lane >= LDQ) return;
    double s = 0.0;
    for (int e = indptr[row]; e < indptr[row + 1]; ++e) s += (double)vals[e] * (double)Q[(size_t)cols[e] * LDQ + lane];
    Y[(size_t)row * LDQ + lane] = s;
}

// operand in MFMA B order: Qt[kb][nb][piece][lane][8], element (k, n) of the 32 x 16 block in lane (k / 8) * 16 + n, slot k % 8
__global__ void k_split_operand(const float* Q, int H, __bf16* Qt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (kb, nb, lane)
    const int nkb = (H + 31) / 32;
    if (t >= (int64_t)nkb * NB * 64) return;
    const int lane = (int)(t & 63), nb = (int)((t >> 6) % NB), kb = (int)(t / (64 * NB));
    const int n = nb * 16 + (lane & 15);
    for (int s = 0; s < 8; ++s) {
        const int k = kb * 32 + (lane >> 4) * 8 + s;
        const float v = k < H ? Q[(size_t)k * LDQ + n] : 0.f;
        const __bf16 a = (__bf16)v;
        const float r1 = v - (float)a;
        const __bf16 b = (__bf16)r1;
        const __bf16 c = (__bf16)(r1 - (float)b);
        const size_t base = ((((size_t)kb * NB + nb) * NP) * 64 + lane) * 8 + s;
        Qt[base] = a;
        Qt[base + 64 * 8] = b;
        Qt[base + 2 * 64 * 8] = c;
    }
}

typedef int32_t i4 __attribute__((ext_vector_type(4)));

// 16 buffered entries per row: lane (r, t) holds entries base + 4 t .. + 3 of row r (one 16-byte load each for columns and values);
// a second set holds the 16 entries behind them.  Entries past the end of the row carry the column INT_MAX.
struct RowBuf {
    i4 c;
    f4 v;
};
__device__ __forceinline__ RowBuf load16(const int32_t* __restrict__ cols, const float* __restrict__ vals, int e, int end) {
    RowBuf b;
    b.c = *reinterpret_cast<const i4*>(cols + e);           // (the arrays are padded by 64 entries)
    b.v = *reinterpret_cast<const f4*>(vals + e);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (e + j >= end) b.c[j] = 0x7fffffff;
    return b;
}

__global__ void __launch_bounds__(64 * WAVES) k_spmm_mfma(const int32_t* __restrict__ indptr, const int32_t* __restrict__ cols, const float* __restrict__ vals,
                                                       const __bf16* __restrict__ Qt, int M, int H, double* __restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u4* Bs = reinterpret_cast<u4*>(smem);                                   // [KBS][NB][NP][64] x 16 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u4* img = Bs + KBS * NB * NP * 64 + wave * NP * 64;                     // [NP][64] x 16 bytes, private to the wave
    uint16_t* img16 = reinterpret_cast<uint16_t*>(img);
    const int r = lane & 15, t = lane >> 4;
    const int nkb = (H + 31) / 32;
    int nxt_e[RB], end[RB];              // position of this lane's part of the NEXT 16 entries to load
    RowBuf cur[RB], nxt[RB];
    f4 acc[RB][NB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = ((blockIdx.x * WAVES + wave) * RB + i) * 16 + r;
        const int b0 = row < M ? indptr[row] : 0;
        end[i] = row < M ? indptr[row + 1] : 0;
        cur[i] = load16(cols, vals, b0 + 4 * t, end[i]);
        nxt[i] = load16(cols, vals, b0 + 16 + 4 * t, end[i]);
        nxt_e[i] = b0 + 32 + 4 * t;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[i][nb] = (f4)(0.f);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) img[p * 64 + lane] = (u4)(0u);          // the image is all zero between windows
    for (int s0 = 0; s0 < nkb; s0 += KBS) {
        __syncthreads();
        {
            const int nvec = (nkb - s0 < KBS ? nkb - s0 : KBS) * NB * NP * 64;
            const u4* src = reinterpret_cast<const u4*>(Qt) + (size_t)s0 * NB * NP * 64;
            for (int base = wave * 64; base < nvec; base += 64 * WAVES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + lane),
                                                 (__attribute__((address_space(3))) void*)(Bs + base), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const int kbn = nkb - s0 < KBS ? nkb - s0 : KBS;
        for (int kb = 0; kb < kbn; ++kb) {
            const int c0 = (s0 + kb) * 32;
            bf16x8 b[NB][NP];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int p = 0; p < NP; ++p) b[nb][p] = __builtin_bit_cast(bf16x8, Bs[((kb * NB + nb) * NP + p) * 64 + lane]);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                // densify the 16 x 32 window of row block i: every buffered entry whose column lies in the window
                auto put = [&](int col, float v, bool zero) {
                    const unsigned k = (unsigned)(col - c0);
                    if (k < 32u) {
                        const int slot = ((k >> 3) * 16 + r) * 8 + (k & 7);
                        if (zero) {
                            img16[slot] = 0; img16[slot + 64 * 8] = 0; img16[slot + 2 * 64 * 8] = 0;
                        } else {
                            const __bf16 a = (__bf16)v;
                            const float r1 = v - (float)a;
                            const __bf16 m = (__bf16)r1;
                            const __bf16 l = (__bf16)(r1 - (float)m);
                            img16[slot] = __builtin_bit_cast(uint16_t, a);
                            img16[slot + 64 * 8] = __builtin_bit_cast(uint16_t, m);
                            img16[slot + 2 * 64 * 8] = __builtin_bit_cast(uint16_t, l);
                        }
                    }
                };
                bool dirty = false;
                for (;;) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { put(cur[i].c[j], cur[i].v[j], false); put(nxt[i].c[j], nxt[i].v[j], false); }
                    // a row whose 32 buffered entries all lie before the end of this window may own more of it (rare: > 16 entries
                    // of one row in 32 columns): take the next 16 and look again
                    const int last_n = __shfl(nxt[i].c[3], 48 + r, 64);            // last buffered column of this lane's row
                    if (__ballot(last_n < c0 + 32) == 0ull) break;
                    dirty = true;                                                  // (entries of a dropped `cur` stay in the image: clear all of it below)
                    if (last_n < c0 + 32) {
                        cur[i] = nxt[i];
                        nxt[i] = load16(cols, vals, nxt_e[i], end[i]);
                        nxt_e[i] += 16;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                bf16x8 a[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) a[p] = __builtin_bit_cast(bf16x8, img[p * 64 + lane]);
                // back to zero: only what was written
                if (dirty) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) img[p * 64 + lane] = (u4)(0u);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { put(cur[i].c[j], 0.f, true); put(nxt[i].c[j], 0.f, true); }
                }
                // rows whose current 16 entries are all behind this window move on to the next 16 (the load has two windows to arrive)
                {
                    const int last_c = __shfl(cur[i].c[3], 48 + r, 64);
                    if (last_c < c0 + 32) {
                        cur[i] = nxt[i];
                        nxt[i] = load16(cols, vals, nxt_e[i], end[i]);
                        nxt_e[i] += 16;
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f4 c = acc[i][nb];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[nb][1], c, 0, 0, 0);      // small terms first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[nb][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[nb][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[nb][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[nb][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[nb][0], c, 0, 0, 0);
                    acc[i][nb] = c;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = ((blockIdx.x * WAVES + wave) * RB + i) * 16 + (lane >> 4) * 4 + c;
                if (row < M) Y[(size_t)row * LDQ + nb * 16 + (lane & 15)] = (double)acc[i][nb][c];
            }
}

int main() {
    const int M = 125000, H = 10000;
    std::vector<int32_t> indptr(M + 1, 0), cols;
    std::vector<float> vals;
    cols.reserve(130000000); vals.reserve(130000000);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int i = 0; i < M; ++i) {
        const double dens = (i < 100000 ? 0.0813 : 0.1626) * (0.9 + 0.2 * (double)(rnd() % 1000) / 1000.0);
        for (int c = 0; c < H; ++c)
            if ((double)(rnd() % 1000000) / 1e6 < dens) { cols.push_back(c); vals.push_back(0.05f + 3.0f * (float)(rnd() % 100000) / 1e5f); }
        indptr[i + 1] = (int32_t)cols.size();
    }
    const size_t nnz = cols.size();
    printf("M %d H %d nnz %zu (%.1f per row)\n", M, H, nnz, (double)nnz / M);
    std::vector<float> Q((size_t)H * LDQ, 0.f);
    for (int k = 0; k < H; ++k)
        for (int n = 0; n < 40; ++n) Q[(size_t)k * LDQ + n] = (float)((double)(rnd() % 2000001) / 1e6 - 1.0);
    int32_t *d_indptr, *d_cols; float *d_vals, *d_Q; __bf16* d_Qt; double *d_Y, *d_Yref;
    const int nkb = (H + 31) / 32;
    CHECK(hipMalloc(&d_indptr, 4 * (M + 1))); CHECK(hipMalloc(&d_cols, 4 * (nnz + 64))); CHECK(hipMalloc(&d_vals, 4 * (nnz + 64)));
    CHECK(hipMalloc(&d_Q, 4 * Q.size())); CHECK(hipMalloc(&d_Qt, (size_t)nkb * NB * NP * 64 * 16));
    CHECK(hipMalloc(&d_Y, 8 * (size_t)M * LDQ)); CHECK(hipMalloc(&d_Yref, 8 * (size_t)M * LDQ));
    CHECK(hipMemcpy(d_indptr, indptr.data(), 4 * (M + 1), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_cols, cols.data(), 4 * nnz, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_vals, vals.data(), 4 * nnz, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_Q, Q.data(), 4 * Q.size(), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_Y, 0, 8 * (size_t)M * LDQ));
    k_reference<<<(M + 3) / 4, 256>>>(d_indptr, d_cols, d_vals, d_Q, M, d_Yref);
    k_split_operand<<<(nkb * NB * 64 + 255) / 256, 256>>>(d_Q, H, d_Qt);
    const size_t lds = (size_t)(KBS * NB * NP * 64 + WAVES * NP * 64) * 16;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmm_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = (M / 16 + WAVES * RB - 1) / (WAVES * RB) + 1;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_spmm_mfma<<<grid, 64 * WAVES, lds>>>(d_indptr, d_cols, d_vals, d_Qt, M, H, d_Y);
    CHECK(hipDeviceSynchronize());
    CHECK(hipGetLastError());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        k_spmm_mfma<<<grid, 64 * WAVES, lds>>>(d_indptr, d_cols, d_vals, d_Qt, M, H, d_Y);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    std::vector<double> Y((size_t)M * LDQ), Yr((size_t)M * LDQ);
    CHECK(hipMemcpy(Y.data(), d_Y, 8 * Y.size(), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(Yr.data(), d_Yref, 8 * Yr.size(), hipMemcpyDeviceToHost));
    double maxabs = 0, maxref = 0, sumsq = 0, sumref = 0;
    for (size_t i = 0; i < Y.size(); ++i) {
        if (i % LDQ >= 40) continue;
        const double d = fabs(Y[i] - Yr[i]);
        maxabs = d > maxabs ? d : maxabs; maxref = fabs(Yr[i]) > maxref ? fabs(Yr[i]) : maxref;
        sumsq += d * d; sumref += Yr[i] * Yr[i];
    }
    printf("grid %d, LDS %zu bytes: %.3f ms per launch; max |err| %.3e (max |ref| %.3e), relative Frobenius error %.3e\n", grid, lds, best, maxabs, maxref,
           sqrt(sumsq / sumref));
    return 0;
}
