// Prototype of a register-broadcast inner loop for the operator products (round 4, late): the staged (offset, value) pairs of a
// round live in the registers of the 16 lanes of a DPP row (4 entries per lane) and reach the row's lanes by row_newbcast, the
// operand slice sits in LDS with a 256-byte pitch (bank = column) and is read with ds_read_b128 by 10 of the 16 lanes (4 sketch
// columns each; in-row lanes {0,1,2,3,12} take the column quads 0-4, {4,5,6,7,8} the quads 5-9, which makes the two rows that
// share an LDS service group read disjoint banks).  Measures LDS-pipe + VALU cycles per stored entry of that loop alone.
//   hipcc --offload-arch=gfx950 -O3 spmm_dpp_proto.hip -o dpp && ./dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N>
__device__ __forceinline__ uint32_t row_bcast(uint32_t v) {      // lane N of every row of 16 to the whole row
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + N, 0xf, 0xf, false);
}

template <int U>
__device__ __forceinline__ void trip8(const uint32_t (&offs)[4], const uint32_t (&vals)[4], uint32_t lane_base, double (&acc)[4], int T0) {}

constexpr int kRows = 624;
template <int PITCH_F>      // floats per operand row in LDS: 64 (256 B, conflict-free) or 40 (160 B)
__global__ void __launch_bounds__(1024) k_loop(const float* __restrict__ op, int rounds, double* __restrict__ out, int conflict_free_rows) {
    extern __shared__ __align__(16) float lds[];
    for (int i = threadIdx.x; i < kRows * PITCH_F; i += 1024) lds[i] = op[i % (kRows * 40)];
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 15;
    const bool active = j < 9 || j == 12;
    const int quad = j < 4 ? j : (j == 12 ? 4 : j + 1);           // {0,1,2,3,12} -> 0..4, {4..8} -> 5..9
    const uint32_t lane_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)lds + (active ? quad : 0) * 16;
    double acc[4] = {0, 0, 0, 0};
    uint32_t seed = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    for (int r = 0; r < rounds; ++r) {
        uint32_t offs[4], vals[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            seed = seed * 1664525u + 1013904223u;
            uint32_t row = (seed >> 8) % kRows;
            if (conflict_free_rows) row = (row & ~7u) | (uint32_t)(lane >> 4);
            offs[u] = row * (PITCH_F * 4);
            vals[u] = __float_as_uint(1.0f + (float)(seed & 255) * 0.001f);
        }
#define STEP(U, N)                                                                                                       \
    {                                                                                                                    \
        const uint32_t a = lane_base + row_bcast<N>(offs[U]);                                                            \
        const float v = __uint_as_float(row_bcast<N>(vals[U]));                                                          \
        const f4 q = *reinterpret_cast<const __attribute__((address_space(3))) f4*>((uintptr_t)a);                      \
        p = __builtin_elementwise_fma(q, (f4)(v), p);                                                                    \
    }
#define TRIP(U, N0)                                                                                                      \
    {                                                                                                                    \
        f4 p = (f4)(0.f);                                                                                                \
        STEP(U, N0) STEP(U, N0 + 1) STEP(U, N0 + 2) STEP(U, N0 + 3) STEP(U, N0 + 4) STEP(U, N0 + 5) STEP(U, N0 + 6) STEP(U, N0 + 7) \
        acc[0] += (double)p.x; acc[1] += (double)p.y; acc[2] += (double)p.z; acc[3] += (double)p.w;                      \
    }
        TRIP(0, 0) TRIP(0, 8) TRIP(1, 0) TRIP(1, 8) TRIP(2, 0) TRIP(2, 8) TRIP(3, 0) TRIP(3, 8)
    }
    if (active) out[(size_t)(blockIdx.x * 1024 + threadIdx.x)] = acc[0] + acc[1] + acc[2] + acc[3];
}


typedef float f3 __attribute__((ext_vector_type(3)));
template <int LSTRIDE, int PITCHB>
__global__ void __launch_bounds__(1024) k_loop96(const float* __restrict__ op, int rounds, double* __restrict__ out, int check) {
    extern __shared__ __align__(16) float lds[];
    for (int i = threadIdx.x; i < kRows * (PITCHB / 4) + 8; i += 1024) lds[i] = op[i % (kRows * 40)];
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 15;
    const bool active = j < 14;
    const uint32_t lane_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)lds + (active ? j : 0) * LSTRIDE;
    double acc[3] = {0, 0, 0};
    uint32_t seed = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    for (int r = 0; r < rounds; ++r) {
        uint32_t offs[4], vals[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            seed = seed * 1664525u + 1013904223u;
            uint32_t row = (seed >> 8) % kRows;
            offs[u] = row * PITCHB;
            vals[u] = __float_as_uint(1.0f + (float)(seed & 255) * 0.001f);
        }
#define RD96(U, N, Q)                                                                                                    \
    {                                                                                                                    \
        const uint32_t a = lane_base + row_bcast<N>(offs[U]);                                                            \
        asm volatile("ds_read_b96 %0, %1" : "=v"(Q) : "v"(a));                                                          \
    }
#define FM96(U, N, Q)                                                                                                    \
    {                                                                                                                    \
        const float v = __uint_as_float(row_bcast<N>(vals[U]));                                                          \
        p = __builtin_elementwise_fma(Q, (f3)(v), p);                                                                    \
    }
#define TRIP96(U, N0)                                                                                                    \
    {                                                                                                                    \
        f3 p = (f3)(0.f), q0, q1, q2, q3, q4, q5, q6, q7;                                                                \
        RD96(U, N0, q0) RD96(U, N0 + 1, q1) RD96(U, N0 + 2, q2) RD96(U, N0 + 3, q3) RD96(U, N0 + 4, q4) RD96(U, N0 + 5, q5) RD96(U, N0 + 6, q6) RD96(U, N0 + 7, q7) \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));   \
        FM96(U, N0, q0) FM96(U, N0 + 1, q1) FM96(U, N0 + 2, q2) FM96(U, N0 + 3, q3) FM96(U, N0 + 4, q4) FM96(U, N0 + 5, q5) FM96(U, N0 + 6, q6) FM96(U, N0 + 7, q7) \
        acc[0] += (double)p.x; acc[1] += (double)p.y; acc[2] += (double)p.z;                                             \
    }
        TRIP96(0, 0) TRIP96(0, 8) TRIP96(1, 0) TRIP96(1, 8) TRIP96(2, 0) TRIP96(2, 8) TRIP96(3, 0) TRIP96(3, 8)
    }
    if (check) {
        // one read against plain loads: does the hardware serve a 4-byte aligned ds_read_b96?
        f3 q;
        const uint32_t a = lane_base + 7 * PITCHB;
        asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a));
        const float* ref = lds + 7 * (PITCHB / 4) + (active ? j : 0) * (LSTRIDE / 4);
        out[(size_t)(blockIdx.x * 1024 + threadIdx.x)] = (q.x == ref[0] && q.y == ref[1] && q.z == ref[2]) ? 1.0 : 0.0;
        return;
    }
    if (active) out[(size_t)(blockIdx.x * 1024 + threadIdx.x)] = acc[0] + acc[1] + acc[2];
}

// Split of the 40 sketch columns into 32 + 8.  Pass 1: a half row (8 lanes x 16 bytes) reads the first 32 columns of an entry's operand row,
// so a DPP row serves two entries per step and a wave eight (the two halves get their (offset, value) by two bank-masked row_newbcast
// moves each).  Pass 2: the last 8 columns, two lanes per entry, four entries per DPP row and step (one per bank of four lanes).
template <int PITCHB>
__global__ void __launch_bounds__(1024) k_split(const float* __restrict__ op, int rounds, double* __restrict__ out, int pass) {
    extern __shared__ __align__(16) float lds[];
    for (int i = threadIdx.x; i < kRows * (PITCHB / 4); i += 1024) lds[i] = op[i % (kRows * 40)];
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 15;
    const uint32_t base3 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)lds;
    const uint32_t lane_base1 = base3 + (j & 7) * 16;                       // pass 1: columns 4 (j & 7) .. + 3
    const uint32_t lane_base2 = base3 + 128 + (j & 1) * 16;                 // pass 2: columns 32 + 4 (j & 1) .. + 3 (lanes 0-1 of each bank of 4)
    double acc[4] = {0, 0, 0, 0};
    uint32_t seed = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    for (int r = 0; r < rounds; ++r) {
        uint32_t offs[4], vals[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            seed = seed * 1664525u + 1013904223u;
            offs[u] = ((seed >> 8) % kRows) * PITCHB;
            vals[u] = __float_as_uint(1.0f + (float)(seed & 255) * 0.001f);
        }
        if (pass == 1) {
            // step s of a register U: entries 2 s (lanes 0-7 of the row) and 2 s + 1 (lanes 8-15): broadcast lanes 2 s and 2 s + 1
#define BC2(REG, N) ((uint32_t)__builtin_amdgcn_update_dpp((int)__builtin_amdgcn_update_dpp(0, (int)(REG), 0x150 + 2 * (N), 0xf, 0x3, false), (int)(REG), 0x150 + 2 * (N) + 1, 0xf, 0xc, false))
#define STEP1(U, N)                                                                                                      \
    {                                                                                                                    \
        const uint32_t a = lane_base1 + BC2(offs[U], N);                                                                 \
        const float v = __uint_as_float(BC2(vals[U], N));                                                                \
        const f4 q = *reinterpret_cast<const __attribute__((address_space(3))) f4*>((uintptr_t)a);                      \
        p = __builtin_elementwise_fma(q, (f4)(v), p);                                                                    \
    }
#define TRIP1(U)                                                                                                         \
    {                                                                                                                    \
        f4 p = (f4)(0.f);                                                                                                \
        STEP1(U, 0) STEP1(U, 1) STEP1(U, 2) STEP1(U, 3) STEP1(U, 4) STEP1(U, 5) STEP1(U, 6) STEP1(U, 7)                  \
        acc[0] += (double)p.x; acc[1] += (double)p.y; acc[2] += (double)p.z; acc[3] += (double)p.w;                      \
    }
            TRIP1(0) TRIP1(1) TRIP1(2) TRIP1(3)
        } else {
            // step s of a register U: entries 4 s + b for the four banks b: broadcast lanes 4 s + b into bank b
#define BC4(REG, N) ((uint32_t)__builtin_amdgcn_update_dpp((int)__builtin_amdgcn_update_dpp((int)__builtin_amdgcn_update_dpp((int)__builtin_amdgcn_update_dpp(0, (int)(REG), 0x150 + 4 * (N), 0xf, 0x1, false), (int)(REG), 0x150 + 4 * (N) + 1, 0xf, 0x2, false), (int)(REG), 0x150 + 4 * (N) + 2, 0xf, 0x4, false), (int)(REG), 0x150 + 4 * (N) + 3, 0xf, 0x8, false))
#define STEP2(U, N)                                                                                                      \
    {                                                                                                                    \
        const uint32_t a = lane_base2 + BC4(offs[U], N);                                                                 \
        const float v = __uint_as_float(BC4(vals[U], N));                                                                \
        const f4 q = *reinterpret_cast<const __attribute__((address_space(3))) f4*>((uintptr_t)a);                      \
        p = __builtin_elementwise_fma(q, (f4)(v), p);                                                                    \
    }
#define TRIP2(U0, U1)                                                                                                    \
    {                                                                                                                    \
        f4 p = (f4)(0.f);                                                                                                \
        STEP2(U0, 0) STEP2(U0, 1) STEP2(U0, 2) STEP2(U0, 3) STEP2(U1, 0) STEP2(U1, 1) STEP2(U1, 2) STEP2(U1, 3)          \
        acc[0] += (double)p.x; acc[1] += (double)p.y; acc[2] += (double)p.z; acc[3] += (double)p.w;                      \
    }
            TRIP2(0, 1) TRIP2(2, 3)
        }
    }
    out[(size_t)(blockIdx.x * 1024 + threadIdx.x)] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    const int rounds = 2000, blocks = 256;
    float* op; double* out;
    CK(hipMalloc(&op, sizeof(float) * kRows * 64));
    CK(hipMalloc(&out, sizeof(double) * blocks * 1024));
    std::vector<float> h(kRows * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 97) * 0.01f;
    CK(hipMemcpy(op, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; ++variant) {
        const int pitch = variant == 1 ? 40 : 64;
        const size_t lds_bytes = sizeof(float) * kRows * pitch;
        auto kern = variant == 1 ? k_loop<40> : k_loop<64>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        const int cf = variant == 2;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds_bytes, 0, op, 10, out, cf);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds_bytes, 0, op, rounds, out, cf);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_steps = (double)rounds * 64 * 16;                  // per CU (one block of 16 waves per CU)
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("%s: %.3f ms, %.2f CU-cycles (at 2.4 GHz) per wave-step = %.2f per stored entry (4 per step)\n",
               variant == 0 ? "pitch 256 B, random rows" : (variant == 1 ? "pitch 160 B, random rows" : "pitch 256 B, rows of different classes (control)"),
               ms, cyc / wave_steps, cyc / wave_steps / 4);
    }
    for (int v96 = 0; v96 < 2; ++v96) {
        const int pitchb = v96 ? 256 : 160;
        const size_t lds_bytes = sizeof(float) * (kRows * (pitchb / 4) + 8);
        auto k96 = v96 ? k_loop96<16, 256> : k_loop96<12, 160>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k96), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(k96, dim3(blocks), dim3(1024), lds_bytes, 0, op, 1, out, 1);
        CK(hipDeviceSynchronize());
        std::vector<double> ho(1024);
        CK(hipMemcpy(ho.data(), out, sizeof(double) * 1024, hipMemcpyDeviceToHost));
        int okc = 0;
        for (int i = 0; i < 1024; ++i) okc += ho[i] == 1.0;
        printf("ds_read_b96, lane stride %d B: %d of 1024 lanes read the right values\n", v96 ? 16 : 12, okc);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k96, dim3(blocks), dim3(1024), lds_bytes, 0, op, rounds, out, 0);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_steps = (double)rounds * 64 * 16, cyc = ms * 1e-3 * 2.4e9;
        printf("ds_read_b96, 14 lanes x 3 columns, lane stride %d B, pitch %d B (eight reads in flight, no overlap across trips): %.3f ms, %.2f CU-cycles per wave-step = %.2f per stored entry\n",
               v96 ? 16 : 12, pitchb, ms, cyc / wave_steps, cyc / wave_steps / 4);
    }
    for (int pass = 1; pass <= 2; ++pass) {
        const size_t lds_bytes = sizeof(float) * kRows * 40;
        auto ks = k_split<160>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(ks, dim3(blocks), dim3(1024), lds_bytes, 0, op, 10, out, pass);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(ks, dim3(blocks), dim3(1024), lds_bytes, 0, op, rounds, out, pass);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double entries = (double)rounds * 256 * 16, cyc = ms * 1e-3 * 2.4e9;      // 256 entries per wave and round, 16 waves per CU
        printf("split 32 + 8, pass %d (%s), pitch 160 B, random rows: %.3f ms = %.2f CU-cycles per stored entry\n", pass,
               pass == 1 ? "32 columns, 8 entries per wave-step" : "8 columns, 16 entries per wave-step", ms, cyc / entries);
    }
    return 0;
}
