// Exact brute-force kNN over the low-dimensional embedding and the graphs handed to community
// detection (phenograph.cluster / sc.pp.neighbors call sites, dd.py:317-336).
//
// kNN: exact, defined in float64 (subtract, multiply, add component by component, no fused multiply-add; ties by
// index), computed in three passes over the points sorted by their first principal component -- an MFMA screen
// that bounds every query's k-th distance from the tiles around it, an MFMA screen of the pruned tile range that
// lists every pair that could be among the k nearest, and an exact float64 evaluation + sort of those few
// candidates (stage_knn below).  The screens only have to be conservative; the result is
// bit-identical to an IEEE float64 brute force.
#include "ddx_prims.h"

#include <algorithm>
#include <cstdlib>

#include "ddx_internal.h"

namespace ddx {

constexpr int kMaxDim = 128;

typedef float f4 __attribute__((ext_vector_type(4)));

// slack of the bfloat16 distance screen, relative to |q|^2 + |c|^2 (derivation at the emit kernel)
constexpr float kScreenSlackBf = 4.2e-5f;

// ---- layouts -----------------------------------------------------------------------------------
// E   : row-major float32 [Mp][CP]            (exact re-evaluation gathers whole rows)
// Et  : MFMA operand layout, 16-point tiles:   Et[tile][s][k][j] = E[16*tile + j][4*s + k]
//       so that lane l of a wave reads operand element (point l&15, component 4s + (l>>4)) at
//       Et[tile*16*CP + s*64 + l] -- one fully coalesced 256-byte load per MFMA k-step, and the same
//       formula serves the A operand (queries) and the B operand (candidates).
// nrm : float32 squared norms; padding points carry +inf so they can never pass the screen.
// Eb  : bfloat16 split operands for v_mfma_f32_16x16x32_bf16: every coordinate a is stored as
//       hi = bf16(a) and lo = bf16(a - hi); for tile t, 32-component block kb and part p (0 = hi, 1 = lo)
//       lane l = ((d % 32) / 8) * 16 + j holds components d = 32*kb + 8*(l>>4) + 0..7 of point 16*t + j as
//       8 consecutive bf16 at Eb[(((t*KB + kb)*2 + p)*64 + l)*8], i.e. one coalesced 1 KB load per operand.
// Points are laid out in the order `perm` (ascending first principal component, see stage_knn): row r of every
// layout is embedding row perm[r];  p1[r] = its first component (+inf for padding rows).
__global__ void k_knn_prepare(const float* __restrict__ in, const int32_t* __restrict__ perm, int64_t M, int64_t Mp, int C, int CP,
                              float* __restrict__ E, float* __restrict__ Et, __bf16* __restrict__ Eb,
                              float* __restrict__ nrm, float* __restrict__ p1, f4* __restrict__ start4) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Mp) return;
    const int64_t src = r < M ? perm[r] : 0;
    p1[r] = r < M ? in[src * C] : __builtin_huge_valf();
    float n = 0.f;
    const int64_t tile = r >> 4;
    const int j = (int)(r & 15);
    const int KB = CP / 32;
    for (int d = 0; d < CP; ++d) {
        const float v = (r < M && d < C) ? in[src * C + d] : 0.f;
        E[r * CP + d] = v;
        Et[tile * 16 * CP + (d >> 2) * 64 + (d & 3) * 16 + j] = v;
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)hi);
        const int kb = d >> 5, lane = (((d & 31) >> 3) << 4) | j, e = d & 7;
        Eb[(((tile * KB + kb) * 2 + 0) * 64 + lane) * 8 + e] = hi;
        Eb[(((tile * KB + kb) * 2 + 1) * 64 + lane) * 8 + e] = lo;
        n = fmaf(v, v, n);
    }
    nrm[r] = (r < M) ? n : __builtin_huge_valf();
    // accumulator start value of the bfloat16 emit pass, as one MFMA C quad: -0.5*(1-slack)*|c|^2 (-inf for padding)
    const float h = (r < M) ? -0.5f * (1.0f - kScreenSlackBf) * n : -__builtin_huge_valf();
    start4[r] = f4{h, h, h, h};
}

// Threshold folding for the emit pass (embeddings of at most 30 components: two of the 32 padded components are free).
// The screen asks  acc > hr_q  with acc = q.c - 0.5 (1-s) |c|^2 from the matrix pipe and hr_q = 0.5 ((1-s) |q|^2 - T_q) a
// constant of the query.  Give every CANDIDATE the value 1 in components 30 and 31 and every QUERY the value -hr_q there,
// cut into four bfloat16 pieces (hi/lo of the two components: 32 mantissa bits, i.e. all of the float32): the matrix pipe
// then delivers acc - hr_q and the screen is "any of the wave's accumulators positive" -- a v_max3 tree and ONE compare per
// tile instead of eight compares and seven scalar ORs.  The four extra terms add four float32 roundings to the
// accumulation (2.4e-7 relative to |q|^2 + |c|^2 + T/2), at most 4.8e-7 (|q|^2 + |c|^2) where it matters -- near the decision d2 = T, hence T <= 2 (|q|^2 + |c|^2) -- covered by raising the slack
// from 4.0e-5 to 4.2e-5 (margin 1e-6 (|q|^2 + |c|^2)).
// Queries without a bound pass everything (+3e38, finite: inf * 0 would poison the hi*lo product), padding queries
// nothing (-3e38).  Eb itself (candidate role) is patched in place -- the bound pass, which needs the zeros, is over.
__global__ void k_knn_fold(const float* __restrict__ nrm, const float* __restrict__ thr, int64_t Mp, __bf16* __restrict__ Eb,
                           __bf16* __restrict__ Ebq) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per 16-byte vector of the operand image
    if (t >= Mp * 8) return;                                               // CP = 32: tile x 2 parts x 64 lanes
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 v = reinterpret_cast<const u4*>(Eb)[t];
    const int lane = (int)(t & 63), part = (int)((t >> 6) & 1);
    const int64_t tile = t >> 7;
    if ((lane >> 4) == 3) {                                                // components 24..31 of point 16*tile + (lane & 15)
        const int64_t r = tile * 16 + (lane & 15);
        const float n = nrm[r], T = thr[r];
        float x = -0.5f * ((1.0f - kScreenSlackBf) * n - T);               // -hr
        if (!(n < __builtin_huge_valf())) x = -3.0e38f;
        else if (!(T < __builtin_huge_valf())) x = 3.0e38f;
        const __bf16 p1 = (__bf16)x;
        const float r1 = x - (float)p1;
        const __bf16 p2 = (__bf16)r1;
        const float r2 = r1 - (float)p2;
        const __bf16 p3 = (__bf16)r2;
        const __bf16 p4 = (__bf16)(r2 - (float)p3);
        const __bf16 a = part == 0 ? p1 : p2, b = part == 0 ? p3 : p4;     // (component 30, component 31) of this part
        u4 q = v;
        q.w = (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
        reinterpret_cast<u4*>(Ebq)[t] = q;
        const __bf16 one = (__bf16)1.0f;
        const uint32_t ones = (uint32_t)__builtin_bit_cast(uint16_t, one) * 0x10001u;
        v.w = part == 0 ? ones : 0u;                                       // candidates: hi = 1, lo = 0
        reinterpret_cast<u4*>(Eb)[t] = v;
    } else {
        reinterpret_cast<u4*>(Ebq)[t] = v;
    }
}

// Screen slack: |fl32(|q|^2 + |c|^2 - 2 q.c) - d2| <= (CP + 8) * 2^-24 * (|q|^2 + |c|^2) for the float32
// MFMA dot product (a k-ordered fmaf chain) and float32 norms (4.3e-6 at CP = 64).  The screen is evaluated
// in the rearranged form  q.c > 0.5*(1-slack)*|c|^2 + 0.5*((1-slack)*|q|^2 - thr), whose extra float32
// roundings (<= 2e-7 relative to the norms) are covered by the margin in 8e-6.
constexpr float kScreenSlack = 8.0e-6f;
// Smallest sample distances kept per lane and row in the bound pass.  Whatever the lanes keep, the k-th smallest of the
// kept values is the k-th smallest of a subset, i.e. a valid upper bound; keeping 4 (instead of 6) loosens it for the
// queries where one lane meets more than 4 of the k nearest sample points, and lets four waves share a SIMD.
constexpr int kBoundKeepSmall = 4;   // k <= 48
constexpr int kBoundKeepLarge = 6;   // k <= 80
constexpr int kCandCap = 768;      // candidate slots per query between the emit and select passes

// ================================================================================================
// exact kNN in three passes (16x16 query x candidate tiles on v_mfma_f32_16x16x32_bf16 with a bfloat16 hi/lo split,
// or on v_mfma_f32_16x16x4_f32 with DDX_KNN_SCREEN=f32); points are in first-principal-component order:
//   1. bound : over the tiles nearest to the query in that order, every lane keeps the kBoundKeep smallest
//              *upper bounds* of the squared distance per screened row; the k-th smallest of the
//              16*kBoundKeep kept values of a row is an upper bound T_q of the query's true k-th
//              neighbour distance (k-th smallest of a subset >= k-th smallest of the whole set).
//   2. emit  : over the contiguous tile range with |c_1 - q_1| <= sqrt(T_q) (k_knn_window), a pair survives when a
//              *lower bound* of its squared distance is below T_q (so no true neighbour can be lost); survivors
//              (about a hundred per query) are appended to the query's candidate list.
//   3. select: one wave per query evaluates its candidates exactly (float64, the reference's
//              arithmetic, one candidate per lane) and sorts them by (distance, index) in LDS; the
//              first k are the result.  A query whose list overflowed is re-scanned over all points
//              by the same pass, so the result is exact in every case and independent of the
//              order in which survivors were appended.
// ================================================================================================

// Workgroup numbering of the MFMA passes.  Hardware workgroup b runs on XCD b % 8; consecutive query blocks screen nearly
// the same candidate tiles.  With chunk > 0 the workgroups resident on one XCD at a time are `chunk` CONSECUTIVE query
// blocks (so a candidate tile fetched into that XCD's L2 serves them all), and the XCDs take adjacent chunks of the query
// range (so all of them meet the same mix of narrow and wide windows).  Returns -1 for the padding of the last chunk.
__device__ __forceinline__ int64_t knn_block(int64_t nblocks, int chunk) {
    const int64_t b = blockIdx.x;
    if (chunk <= 0) return b < nblocks ? b : -1;
    const int64_t xcd = b & 7, s = b >> 3;
    const int64_t lb = ((s / chunk) * 8 + xcd) * chunk + (s % chunk);
    return lb < nblocks ? lb : -1;
}

// ---- candidate tiles are staged through LDS once per block (4 waves share them) ---------------------
// tiles per staged chunk: 16 KB of coordinates per buffer whatever the padded dimension
#ifndef DDX_CHUNK_TILES32
#define DDX_CHUNK_TILES32 8
#endif
__host__ __device__ constexpr int chunk_tiles(int CP) { return CP <= 32 ? DDX_CHUNK_TILES32 : DDX_CHUNK_TILES32 / 2; }

template <int CP>
struct TileStage {
    static constexpr int kChunkTiles = chunk_tiles(CP);
    static constexpr int kFloats = kChunkTiles * 16 * CP;          // coordinates of one chunk
    static constexpr int kPerThread = kFloats / 256;               // floats per thread (256 threads)
    static_assert(kPerThread % 4 == 0, "chunk must split into float4 per thread");
    f4 regs[kPerThread / 4];
    float nreg;                                                    // threads 0..127 carry one norm each
    // global -> registers (issue early), registers -> LDS (after the barrier that retires the old buffer)
    __device__ __forceinline__ void fetch(const float* __restrict__ Et, const float* __restrict__ nrm, int64_t chunk,
                                          int64_t n_chunk_tiles_total, int64_t tile_stride, int tid) {
#pragma unroll
        for (int u = 0; u < kPerThread / 4; ++u) {
            const int f = (u * 256 + tid) * 4;                     // float offset inside the chunk
            const int t = f / (16 * CP);                           // tile slot
            int64_t tile = (chunk * kChunkTiles + t);
            if (tile >= n_chunk_tiles_total) tile = n_chunk_tiles_total - 1;
            regs[u] = *reinterpret_cast<const f4*>(Et + tile * tile_stride * 16 * CP + (f - t * 16 * CP));
        }
        if (tid < kChunkTiles * 16) {
            int64_t tile = chunk * kChunkTiles + (tid >> 4);
            if (tile >= n_chunk_tiles_total) tile = n_chunk_tiles_total - 1;
            nreg = nrm[tile * tile_stride * 16 + (tid & 15)];
        }
    }
    __device__ __forceinline__ void commit(float* lds_coords, float* lds_norms, int tid) const {
#pragma unroll
        for (int u = 0; u < kPerThread / 4; ++u) *reinterpret_cast<f4*>(lds_coords + (u * 256 + tid) * 4) = regs[u];
        if (tid < kChunkTiles * 16) lds_norms[tid] = nreg;
    }
};

template <int CP, int RT>
struct QueryTiles {
    static constexpr int KS = CP / 4;
    float a[RT][KS];
    __device__ __forceinline__ void load(const float* __restrict__ Et, int64_t q0, int lane) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int s = 0; s < KS; ++s) a[rt][s] = Et[((q0 >> 4) + rt) * 16 * CP + s * 64 + lane];
    }
    __device__ __forceinline__ void dots(const float* b, f4 (&acc)[RT]) const {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);     // keep the matrix pipe fed while co-resident waves do their compares
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][s], b[s], acc[rt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
};

constexpr int kBoundRT = 2;   // query tiles per wave in the bound pass (register budget: 16 rows x kBoundKeep)
#ifndef DDX_EMIT_RT
#define DDX_EMIT_RT 2
#endif
constexpr int kEmitRT = DDX_EMIT_RT;    // query tiles per wave in the emit pass
#define DDX_EMIT_WAVES (DDX_EMIT_RT <= 2 ? 4 : 3)
// Waves per workgroup of the bfloat16 emit pass (they share the staged candidate chunks): 4, or 8 from kEmitWideFrom
// points on -- measured per launch: 2 waves 3.50 / 60.3 ms (125 k / 625 k points), 4 waves 2.52 / 43.0 ms, 8 waves
// 2.56 / 40.7 ms (at 125 k points 8 waves leave fewer than two workgroups per CU)
constexpr int64_t kEmitWideFrom = 400000;

template <int CP, int kBoundKeep>
__global__ void __launch_bounds__(256) k_knn_bound(const float* __restrict__ Et, const float* __restrict__ nrm,
                                                   int64_t Mp, int K, int include_self, int64_t nsamp_tiles,
                                                   int64_t tile_stride, int64_t tile_phase, int combine, float* __restrict__ thr_out) {
    constexpr int KS = CP / 4;
    constexpr int RT = kBoundRT, NV = 4 * RT;
    constexpr int kChunkTiles = chunk_tiles(CP);
    __shared__ __attribute__((aligned(16))) float lds_c[2][kChunkTiles * 16 * CP];
    __shared__ float lds_n[2][kChunkTiles * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * RT);     // Mp is a multiple of 256: no partial blocks
    QueryTiles<CP, RT> qt;
    qt.load(Et, q0, lane);
    const int rbase = 4 * (lane >> 4), jcol = lane & 15;
    float nq[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) nq[v] = nrm[q0 + (v >> 2) * 16 + rbase + (v & 3)];
    float best[NV][kBoundKeep];   // ascending
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int t = 0; t < kBoundKeep; ++t) best[v][t] = __builtin_huge_valf();
    const int64_t own_tile = q0 >> 4;
    const int64_t nchunks = (nsamp_tiles + kChunkTiles - 1) / kChunkTiles;
        // sample = the nsamp_tiles tiles around this block in first-component order (contiguous, see stage_knn)
    // (k > 16 * (kBoundKeep - 1): the window is dealt out over `tile_stride` launches, launch `tile_phase` taking every
    // tile_stride-th tile and the ceil(k / tile_stride)-th smallest of ITS points; the largest of those bounds holds at
    // least k points of the whole window below it -- `combine` keeps the maximum over the launches)
    const int64_t span = nsamp_tiles * tile_stride;
    int64_t tile0 = (((int64_t)blockIdx.x * 4) * RT) + 2 * RT - span / 2;
    if (tile0 > (Mp >> 4) - span) tile0 = (Mp >> 4) - span;
    if (tile0 < 0) tile0 = 0;
    tile0 += tile_phase;
    const float* Etw = Et + tile0 * 16 * CP;
    const float* nrmw = nrm + tile0 * 16;
TileStage<CP> st;
    st.fetch(Etw, nrmw, 0, nsamp_tiles, tile_stride, tid);
    st.commit(lds_c[0], lds_n[0], tid);
    __syncthreads();
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int buf = (int)(ch & 1);
        if (ch + 1 < nchunks) st.fetch(Etw, nrmw, ch + 1, nsamp_tiles, tile_stride, tid);
        const int ntile = (int)((nsamp_tiles - ch * kChunkTiles) < kChunkTiles ? (nsamp_tiles - ch * kChunkTiles) : kChunkTiles);
        for (int t = 0; t < ntile; ++t) {
            const int64_t tile = tile0 + (ch * kChunkTiles + t) * tile_stride;
            float b[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) b[s] = lds_c[buf][t * 16 * CP + s * 64 + lane];
            const float nc = lds_n[buf][t * 16 + jcol];
            f4 acc[RT];
            qt.dots(b, acc);
            const bool own = !include_self && (tile >= own_tile && tile < own_tile + RT);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float dot = acc[v >> 2][v & 3];
                const float n = nq[v] + nc;
                float ub = fmaf(-2.f, dot, n) + kScreenSlack * n;        // upper bound of the exact squared distance
                if (own && (tile * 16 + jcol) == (q0 + (v >> 2) * 16 + rbase + (v & 3))) ub = __builtin_huge_valf();
                if (!(ub < best[v][kBoundKeep - 1])) continue;           // also rejects NaN (padding rows/candidates)
                best[v][kBoundKeep - 1] = ub;
#pragma unroll
                for (int u = kBoundKeep - 1; u > 0; --u) {
                    const float lo = fminf(best[v][u - 1], best[v][u]), hi = fmaxf(best[v][u - 1], best[v][u]);
                    best[v][u - 1] = lo;
                    best[v][u] = hi;
                }
            }
        }
        if (ch + 1 < nchunks) st.commit(lds_c[buf ^ 1], lds_n[buf ^ 1], tid);
        __syncthreads();
    }
    // k-th smallest of the 16*kBoundKeep values of each row (held by the 16 lanes that share lane>>4)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        float kth = __builtin_huge_valf();
        int rank[kBoundKeep];
#pragma unroll
        for (int t = 0; t < kBoundKeep; ++t) rank[t] = 0;
        for (int o = 0; o < 16; ++o) {
            const int src = (lane & 48) | ((jcol + o) & 15);
#pragma unroll
            for (int u = 0; u < kBoundKeep; ++u) {
                const float other = __shfl(best[v][u], src, 64);
                const int okey = ((jcol + o) & 15) * kBoundKeep + u;      // tie-break key of the other value
#pragma unroll
                for (int t = 0; t < kBoundKeep; ++t) {
                    const int mkey = jcol * kBoundKeep + t;
                    rank[t] += (other < best[v][t] || (other == best[v][t] && okey < mkey)) ? 1 : 0;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kBoundKeep; ++t)
            if (rank[t] == K - 1) kth = best[v][t];
        // exactly one lane of the 16 holds the value of rank K-1: share it
        for (int o = 1; o < 16; o <<= 1) kth = fminf(kth, __shfl_xor(kth, o, 64));
        if (jcol == 0) {                                                      // +inf when fewer than K sample points
            float* dst = thr_out + q0 + (v >> 2) * 16 + rbase + (v & 3);
            *dst = combine ? fmaxf(*dst, kth) : kth;
        }
    }
}

template <int CP>
__global__ void __launch_bounds__(256) k_knn_emit(const float* __restrict__ Et, const float* __restrict__ nrm,
                                                  const float* __restrict__ thr, int64_t Mp, int include_self,
                                                  int32_t* __restrict__ ccount, int32_t* __restrict__ cbuf, const int32_t* __restrict__ win, int cap) {
    constexpr int KS = CP / 4;
    constexpr int RT = kEmitRT, NV = 4 * RT;
    constexpr int kChunkTiles = chunk_tiles(CP);
    __shared__ __attribute__((aligned(16))) float lds_c[2][kChunkTiles * 16 * CP];
    __shared__ float lds_n[2][kChunkTiles * 16];
    __shared__ int32_t lcnt[4][16 * kEmitRT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (lane < 16 * kEmitRT) lcnt[wave][lane] = 0;
    const int64_t q0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * RT);     // Mp is a multiple of 256: no partial blocks
    QueryTiles<CP, RT> qt;
    qt.load(Et, q0, lane);
    const int rbase = 4 * (lane >> 4), jcol = lane & 15;
    float hr[NV];   // 0.5*((1-slack)*|q|^2 - T_q): q.c must exceed 0.5*(1-slack)*|c|^2 + hr
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int64_t q = q0 + (v >> 2) * 16 + rbase + (v & 3);
        const float n = nrm[q], t = thr[q];
        hr[v] = 0.5f * ((1.0f - kScreenSlack) * n - t);
        if (!(n < __builtin_huge_valf())) hr[v] = __builtin_huge_valf();       // padding query: nothing passes
        else if (!(t < __builtin_huge_valf())) hr[v] = -__builtin_huge_valf(); // no bound: everything passes
    }
    // candidate tiles whose first component can be within reach of any query of this block (k_knn_window)
    const int64_t t_lo = win[2 * blockIdx.x], ntiles = win[2 * blockIdx.x + 1] - t_lo;
    if (ntiles <= 0) {                          // block-uniform: nothing can be within reach
        if (lane < 16 * kEmitRT) ccount[q0 + lane] = 0;
        return;
    }
    const int64_t own_tile = q0 >> 4;
    const int64_t nchunks = (ntiles + kChunkTiles - 1) / kChunkTiles;
    TileStage<CP> st;
    st.fetch(Et + t_lo * 16 * CP, nrm + t_lo * 16, 0, ntiles, 1, tid);
    st.commit(lds_c[0], lds_n[0], tid);
    __syncthreads();
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int buf = (int)(ch & 1);
        if (ch + 1 < nchunks) st.fetch(Et + t_lo * 16 * CP, nrm + t_lo * 16, ch + 1, ntiles, 1, tid);
        const int ntile = (int)((ntiles - ch * kChunkTiles) < kChunkTiles ? (ntiles - ch * kChunkTiles) : kChunkTiles);
        for (int t = 0; t < ntile; ++t) {
            const int64_t tile = t_lo + ch * kChunkTiles + t;
            float b[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) b[s] = lds_c[buf][t * 16 * CP + s * 64 + lane];
            const float hc = 0.5f * (1.0f - kScreenSlack) * lds_n[buf][t * 16 + jcol];   // +inf for padding candidates
            f4 acc[RT];
            qt.dots(b, acc);
            unsigned hits = 0;
#pragma unroll
            for (int v = 0; v < NV; ++v) hits |= (acc[v >> 2][v & 3] > hc + hr[v]) ? (1u << v) : 0u;
            if (__ballot(hits != 0)) {
                const int32_t cand = (int32_t)(tile * 16 + jcol);
                const bool own = !include_self && (tile >= own_tile && tile < own_tile + RT);
                while (hits) {
                    const int v = __ffs(hits) - 1;
                    hits &= hits - 1;
                    const int lq = (v >> 2) * 16 + rbase + (v & 3);
                    const int64_t q = q0 + lq;
                    if (own && q == cand) continue;
                    // this wave is the only writer of its queries' lists: the slot counter lives in LDS
                    const int slot = atomicAdd(&lcnt[wave][lq], 1);
                    if (slot < cap) cbuf[q * cap + slot] = cand;
                }
            }
        }
        if (ch + 1 < nchunks) st.commit(lds_c[buf ^ 1], lds_n[buf ^ 1], tid);
        __syncthreads();
    }
    if (lane < 16 * kEmitRT) ccount[q0 + lane] = lcnt[wave][lane];
}

// ---- bfloat16-split variant of the two MFMA passes ---------------------------------------------------------
// q.c ~= qh.ch + qh.cl + ql.ch with three v_mfma_f32_16x16x32_bf16 (16x the f32 MFMA rate each).  Dropped
// terms (ql.cl and the residuals of the two-term split) are <= 3*2^-18 |q_i||c_i| per component, the float32
// accumulation of 3*32 exact products adds <= ~6e-6 sum|q_i c_i|: |error(q.c)| <= 1.8e-5 |q||c| <= 0.9e-5 (|q|^2+|c|^2);
// doubled in the distance and with the float32 norms that is 2.5e-5 (|q|^2+|c|^2).  The slack below leaves 1.6x margin.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CP>
struct TileStageBf {           // one chunk = chunk_tiles(CP) tiles x (hi|lo) x KB kilobytes = 16 KB, as for float32
    static constexpr int kChunkTiles = chunk_tiles(CP);
    static constexpr int kBytes = kChunkTiles * 16 * CP * 4;       // 2 parts x 2 bytes = 4 bytes per coordinate
    static constexpr int kVec = kBytes / 16 / 256;                 // 16-byte vectors per thread
    f4 regs[kVec];
    float nreg;
    __device__ __forceinline__ void fetch(const __bf16* __restrict__ Eb, const float* __restrict__ nrm, int64_t chunk,
                                          int64_t ntiles_total, int64_t tile_stride, int tid) {
        constexpr int tile_vecs = 16 * CP * 4 / 16;                // 16-byte vectors per tile
#pragma unroll
        for (int u = 0; u < kVec; ++u) {
            const int vix = u * 256 + tid;
            const int t = vix / tile_vecs;
            int64_t tile = chunk * kChunkTiles + t;
            if (tile >= ntiles_total) tile = ntiles_total - 1;
            regs[u] = reinterpret_cast<const f4*>(Eb)[tile * tile_stride * tile_vecs + (vix - t * tile_vecs)];
        }
        if (tid < kChunkTiles * 16) {
            int64_t tile = chunk * kChunkTiles + (tid >> 4);
            if (tile >= ntiles_total) tile = ntiles_total - 1;
            nreg = nrm[tile * tile_stride * 16 + (tid & 15)];
        }
    }
    __device__ __forceinline__ void commit(f4* lds_c, float* lds_n, int tid) const {
#pragma unroll
        for (int u = 0; u < kVec; ++u) lds_c[u * 256 + tid] = regs[u];
        if (tid < kChunkTiles * 16) lds_n[tid] = nreg;
    }
    // emit pass: instead of |c|^2 store the accumulator start value -0.5*(1-slack)*|c|^2, four times (one MFMA C quad)
    __device__ __forceinline__ void commit_start(f4* lds_c, f4* lds_h, int tid) const {
#pragma unroll
        for (int u = 0; u < kVec; ++u) lds_c[u * 256 + tid] = regs[u];
        if (tid < kChunkTiles * 16) {
            const float h = -0.5f * (1.0f - kScreenSlackBf) * nreg;
            lds_h[tid] = f4{h, h, h, h};
        }
    }
};

template <int CP, int RT>
struct QueryTilesBf {
    static constexpr int KB = CP / 32;
    bf16x8 ah[RT][KB], al[RT][KB];
    __device__ __forceinline__ void load(const __bf16* __restrict__ Eb, int64_t q0, int lane) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const bf16x8* p = reinterpret_cast<const bf16x8*>(Eb) + ((((q0 >> 4) + rt) * KB + kb) * 2) * 64 + lane;
                ah[rt][kb] = p[0];
                al[rt][kb] = p[64];
            }
    }
    // tile image in LDS: [kb][part][lane] vectors of 16 bytes
    __device__ __forceinline__ void dots(const f4* tile, int lane, f4 (&acc)[RT]) const { dots_from(tile, lane, f4{0.f, 0.f, 0.f, 0.f}, acc); }
    // one 32-component block whose candidate operands are already in registers (CP = 32)
    __device__ __forceinline__ void dots_regs(const f4 rh, const f4 rl, const f4 start, f4 (&acc)[RT]) const {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, rh);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, rl);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][0], bh, start, 0, 0, 0);   // small terms first
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][0], bl, acc[rt], 0, 0, 0);
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][0], bh, acc[rt], 0, 0, 0);
        }
    }
    __device__ __forceinline__ void dots_from(const f4* tile, int lane, const f4 start, f4 (&acc)[RT]) const {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = start;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const f4 rh = tile[(kb * 2 + 0) * 64 + lane];
            const f4 rl = tile[(kb * 2 + 1) * 64 + lane];
            const bf16x8 bh = __builtin_bit_cast(bf16x8, rh);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, rl);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][kb], bh, acc[rt], 0, 0, 0);   // small terms first
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][kb], bl, acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][kb], bh, acc[rt], 0, 0, 0);
            }
        }
    }
};

template <int CP, int kBoundKeep>
__global__ void __launch_bounds__(256) k_knn_bound_bf(const __bf16* __restrict__ Eb, const float* __restrict__ nrm,
                                                      int64_t Mp, int K, int include_self, int64_t nsamp_tiles,
                                                      int64_t tile_stride, int64_t tile_phase, int combine, float* __restrict__ thr_out,
                                                      int xcd_chunk) {
    constexpr int RT = kBoundRT, NV = 4 * RT;
    constexpr int kChunkTiles = chunk_tiles(CP);
    constexpr int tile_vecs = 16 * CP * 4 / 16;
    __shared__ f4 lds_c[2][kChunkTiles * tile_vecs];
    __shared__ float lds_n[2][kChunkTiles * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t blk = knn_block(Mp / (4 * 16 * RT), xcd_chunk);
    if (blk < 0) return;
    const int64_t q0 = (blk * 4 + wave) * (16 * RT);
    QueryTilesBf<CP, RT> qt;
    qt.load(Eb, q0, lane);
    const int rbase = 4 * (lane >> 4), jcol = lane & 15;
    float nq[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) nq[v] = nrm[q0 + (v >> 2) * 16 + rbase + (v & 3)];
    float best[NV][kBoundKeep];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int t = 0; t < kBoundKeep; ++t) best[v][t] = __builtin_huge_valf();
    const int64_t own_tile = q0 >> 4;
    const int64_t nchunks = (nsamp_tiles + kChunkTiles - 1) / kChunkTiles;
        // sample = the nsamp_tiles tiles around this block in first-component order (contiguous, see stage_knn)
    // (k > 16 * (kBoundKeep - 1): the window is dealt out over `tile_stride` launches, launch `tile_phase` taking every
    // tile_stride-th tile and the ceil(k / tile_stride)-th smallest of ITS points; the largest of those bounds holds at
    // least k points of the whole window below it -- `combine` keeps the maximum over the launches)
    const int64_t span = nsamp_tiles * tile_stride;
    int64_t tile0 = ((blk * 4) * RT) + 2 * RT - span / 2;
    if (tile0 > (Mp >> 4) - span) tile0 = (Mp >> 4) - span;
    if (tile0 < 0) tile0 = 0;
    tile0 += tile_phase;
    const __bf16* Ebw = Eb + tile0 * 16 * CP * 2;
    const float* nrmw = nrm + tile0 * 16;
TileStageBf<CP> st;
    st.fetch(Ebw, nrmw, 0, nsamp_tiles, tile_stride, tid);
    st.commit(lds_c[0], lds_n[0], tid);
    __syncthreads();
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int buf = (int)(ch & 1);
        if (ch + 1 < nchunks) st.fetch(Ebw, nrmw, ch + 1, nsamp_tiles, tile_stride, tid);
        const int ntile = (int)((nsamp_tiles - ch * kChunkTiles) < kChunkTiles ? (nsamp_tiles - ch * kChunkTiles) : kChunkTiles);
        for (int t = 0; t < ntile; ++t) {
            const int64_t tile = tile0 + (ch * kChunkTiles + t) * tile_stride;
            const float nc = lds_n[buf][t * 16 + jcol];
            f4 acc[RT];
            qt.dots(lds_c[buf] + t * tile_vecs, lane, acc);
            const bool own = !include_self && (tile >= own_tile && tile < own_tile + RT);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float dot = acc[v >> 2][v & 3];
                const float n = nq[v] + nc;
                float ub = fmaf(-2.f, dot, n) + kScreenSlackBf * n;      // upper bound of the exact squared distance
                if (own && (tile * 16 + jcol) == (q0 + (v >> 2) * 16 + rbase + (v & 3))) ub = __builtin_huge_valf();
                if (!(ub < best[v][kBoundKeep - 1])) continue;
                best[v][kBoundKeep - 1] = ub;
#pragma unroll
                for (int u = kBoundKeep - 1; u > 0; --u) {
                    const float lo = fminf(best[v][u - 1], best[v][u]), hi = fmaxf(best[v][u - 1], best[v][u]);
                    best[v][u - 1] = lo;
                    best[v][u] = hi;
                }
            }
        }
        if (ch + 1 < nchunks) st.commit(lds_c[buf ^ 1], lds_n[buf ^ 1], tid);
        __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        float kth = __builtin_huge_valf();
        int rank[kBoundKeep];
#pragma unroll
        for (int t = 0; t < kBoundKeep; ++t) rank[t] = 0;
        for (int o = 0; o < 16; ++o) {
            const int src = (lane & 48) | ((jcol + o) & 15);
#pragma unroll
            for (int u = 0; u < kBoundKeep; ++u) {
                const float other = __shfl(best[v][u], src, 64);
                const int okey = ((jcol + o) & 15) * kBoundKeep + u;
#pragma unroll
                for (int t = 0; t < kBoundKeep; ++t) {
                    const int mkey = jcol * kBoundKeep + t;
                    rank[t] += (other < best[v][t] || (other == best[v][t] && okey < mkey)) ? 1 : 0;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kBoundKeep; ++t)
            if (rank[t] == K - 1) kth = best[v][t];
        for (int o = 1; o < 16; o <<= 1) kth = fminf(kth, __shfl_xor(kth, o, 64));
        if (jcol == 0) {                                                      // +inf when fewer than K sample points
            float* dst = thr_out + q0 + (v >> 2) * 16 + rbase + (v & 3);
            *dst = combine ? fmaxf(*dst, kth) : kth;
        }
    }
}

template <int CP, bool FOLD, int kEmitBW>
__global__ void __launch_bounds__(64 * kEmitBW) __attribute__((amdgpu_waves_per_eu(CP <= 64 ? DDX_EMIT_WAVES : 2, CP <= 64 ? DDX_EMIT_WAVES : 2))) k_knn_emit_bf(const __bf16* __restrict__ Eb, const __bf16* __restrict__ Ebq, const float* __restrict__ nrm, const f4* __restrict__ start4,
                                                     const float* __restrict__ thr, int64_t Mp, int include_self,
                                                     int32_t* __restrict__ ccount, int32_t* __restrict__ cbuf, const int32_t* __restrict__ win, int dbg, int cap,
                                                     int xcd_chunk) {
    constexpr int RT = kEmitRT, NV = 4 * RT;
    constexpr int kChunkTiles = chunk_tiles(CP);
    constexpr int tile_vecs = 16 * CP * 4 / 16;
    __shared__ f4 lds_c[2][kChunkTiles * tile_vecs];
    __shared__ f4 lds_h[2][kChunkTiles * 16];     // accumulator start values -0.5*(1-slack)*|c|^2 (see commit_start)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t blk = knn_block(Mp / (kEmitBW * 16 * RT), xcd_chunk);
    if (blk < 0) return;
    const int64_t q0 = (blk * kEmitBW + wave) * (16 * RT);
    QueryTilesBf<CP, RT> qt;
    qt.load(FOLD ? Ebq : Eb, q0, lane);          // FOLD: the query operands carry -hr in components 30 / 31 (k_knn_fold)
    const int rbase = 4 * (lane >> 4), jcol = lane & 15;
    // A wave is the only writer of its 16*RT queries' candidate lists, and the 16 lanes of a lane group see the same
    // NV queries: every lane keeps the NV slot counters of its group in registers (identical in the 16 lanes, updated
    // by all of them from the ballot masks), so appending needs no atomics and no cross-lane traffic.
    int cnt[NV];
    float hr[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        cnt[v] = 0;
        const int64_t q = q0 + (v >> 2) * 16 + rbase + (v & 3);
        const float n = nrm[q], t = thr[q];
        hr[v] = 0.5f * ((1.0f - kScreenSlackBf) * n - t);
        if (!(n < __builtin_huge_valf())) hr[v] = __builtin_huge_valf();
        else if (!(t < __builtin_huge_valf())) hr[v] = -__builtin_huge_valf();
        if (dbg & 1) hr[v] = __builtin_huge_valf();          // experiment: nothing passes the screen
        if (FOLD) hr[v] = 0.f;                               // the threshold travels inside the dot product
    }
    // candidate tiles whose first component can be within reach of any query of this block (k_knn_window)
    const int64_t t_lo = win[2 * blk], ntiles = win[2 * blk + 1] - t_lo;
    if (ntiles <= 0) {                          // block-uniform: nothing can be within reach
        if (lane < 16 * kEmitRT) ccount[q0 + lane] = 0;
        return;
    }
    const int32_t own_tile = (int32_t)(q0 >> 4);
    const int64_t nchunks = (ntiles + kChunkTiles - 1) / kChunkTiles;
    // Chunk staging by asynchronous global -> LDS copies (16 bytes per lane, LDS destination = wave-uniform base +
    // lane*16, no registers held): a chunk of kChunkTiles tiles is contiguous in Eb, as are its start quads.
    const f4* srcE = reinterpret_cast<const f4*>(Eb) + t_lo * tile_vecs;
    const f4* srcH = start4 + t_lo * 16;
    const int64_t last_vec = ntiles * tile_vecs - 1, last_h = ntiles * 16 - 1;
    auto stage = [&](int64_t ch, int buf) {
#pragma unroll
        for (int u = 0; u < kChunkTiles * tile_vecs / (64 * kEmitBW); ++u) {
            int64_t g = ch * (kChunkTiles * tile_vecs) + u * (64 * kEmitBW) + tid;
            if (g > last_vec) g = last_vec;                       // the ragged last chunk re-reads the last tile
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcE + g),
                                             (__attribute__((address_space(3))) void*)(lds_c[buf] + u * (64 * kEmitBW) + wave * 64), 16, 0, 0);
        }
        if (tid < kChunkTiles * 16) {                             // whole waves (kChunkTiles*16 is a multiple of 64)
            int64_t g = ch * (kChunkTiles * 16) + tid;
            if (g > last_h) g = last_h;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcH + g),
                                             (__attribute__((address_space(3))) void*)(lds_h[buf] + wave * 64), 16, 0, 0);
        }
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int buf = (int)(ch & 1);
        if (ch + 1 < nchunks) stage(ch + 1, buf ^ 1);
        const int ntile = (int)((ntiles - ch * kChunkTiles) < kChunkTiles ? (ntiles - ch * kChunkTiles) : kChunkTiles);
        const f4* tb = lds_c[buf];
        static_assert(CP % 32 == 0, "");
        // one compare per pair; the wave-wide masks live in scalar registers
        auto judge = [&](const f4 (&acc)[RT], const int32_t tile) {
            unsigned long long any = 0, hm[NV];
            if (FOLD) {
                // one compare per tile: the largest of the wave's accumulators against zero (v_max3 tree)
                float m = acc[0].x;
#pragma unroll
                for (int v = 1; v < NV; ++v) m = fmaxf(m, acc[v >> 2][v & 3]);
                any = __ballot(m > 0.f);
                if (any) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) hm[v] = __ballot(acc[v >> 2][v & 3] > 0.f);
                }
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    hm[v] = __ballot(acc[v >> 2][v & 3] > hr[v]);
                    any |= hm[v];
                }
            }
            if (any) {
                const int32_t cand = tile * 16 + jcol;
                const bool own = !include_self && (tile >= own_tile && tile < own_tile + RT);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    unsigned long long m = hm[v];
                    if (m == 0ull) continue;                     // wave-uniform: usually at most one of the masks is set
                    const int64_t q = q0 + (v >> 2) * 16 + rbase + (v & 3);
                    if (own) m &= ~__ballot(q == cand);          // a point is not its own neighbour
                    const unsigned m16 = (unsigned)(m >> (lane & 48)) & 0xffffu;     // the 16 candidates of this lane group's query
                    if (m16) {
                        if ((m16 >> jcol) & 1u) {
                            const int slot = cnt[v] + __popc(m16 & ((1u << jcol) - 1u));
                            if (slot < cap) cbuf[q * cap + slot] = cand;
                        }
                        cnt[v] += __popc(m16);
                    }
                }
            }
        };
        const int32_t tile0 = (int32_t)t_lo + (int32_t)ch * kChunkTiles;
        if (CP == 32) {
            // operands of the next tile are read from LDS while the current one is on the matrix pipe; two register
            // sets (A, B) alternate, two tiles per trip, so that nothing is copied between them
            f4 ah_ = tb[lane], al_ = tb[64 + lane], as_ = lds_h[buf][jcol];
            int t = 0;
            for (; t + 1 < ntile; t += 2) {
                const f4 bh_ = tb[(t + 1) * tile_vecs + lane], bl_ = tb[(t + 1) * tile_vecs + 64 + lane], bs_ = lds_h[buf][(t + 1) * 16 + jcol];
                f4 acc[RT];
                qt.dots_regs(ah_, al_, as_, acc);
                judge(acc, tile0 + t);
                const int t2 = t + 2 < ntile ? t + 2 : t + 1;
                ah_ = tb[t2 * tile_vecs + lane];
                al_ = tb[t2 * tile_vecs + 64 + lane];
                as_ = lds_h[buf][t2 * 16 + jcol];
                f4 acc2[RT];
                qt.dots_regs(bh_, bl_, bs_, acc2);
                judge(acc2, tile0 + t + 1);
            }
            if (t < ntile) {
                f4 acc[RT];
                qt.dots_regs(ah_, al_, as_, acc);
                judge(acc, tile0 + t);
            }
        } else {
            for (int t = 0; t < ntile; ++t) {
                f4 acc[RT];
                qt.dots_from(tb + t * tile_vecs, lane, lds_h[buf][t * 16 + jcol], acc);
                judge(acc, tile0 + t);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next chunk has landed before anyone crosses the barrier
        __syncthreads();
    }
    if (jcol == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) ccount[q0 + (v >> 2) * 16 + rbase + (v & 3)] = cnt[v];
    }
}

// Exact squared distance in the reference's arithmetic: float64, (a-b)*(a-b) rounded, then added,
// components in order -- bit-identical to the float64 brute-force definition (oracle knn_bruteforce_f64).
template <int CP>
__device__ __forceinline__ double exact_d2(const float* q /* LDS, CP floats */, const float* __restrict__ c) {
#pragma clang fp contract(off)
    double d2 = 0.0;
#pragma unroll
    for (int t = 0; t < CP; t += 4) {
        const f4 y = *reinterpret_cast<const f4*>(c + t);
        const f4 x = *reinterpret_cast<const f4*>(q + t);
        const double d0 = (double)x.x - (double)y.x; const double s0 = d0 * d0; d2 = d2 + s0;
        const double d1 = (double)x.y - (double)y.y; const double s1 = d1 * d1; d2 = d2 + s1;
        const double d2_ = (double)x.z - (double)y.z; const double s2 = d2_ * d2_; d2 = d2 + s2;
        const double d3 = (double)x.w - (double)y.w; const double s3 = d3 * d3; d2 = d2 + s3;
    }
    return d2;
}

constexpr int kSelMax = 1024;   // sort window of the select pass (power of two, >= kCandCap + 64)
constexpr int kSelSmall = 256;  // queries with at most this many candidates (nearly all) sort in a small window: 4x the waves per CU
constexpr int kSelHuge = 4096;  // k > 80: lists of up to kCandCapLarge entries (one wave per workgroup: 48 KB of LDS)
constexpr int kCandCapLarge = 3072;

// bitonic sort of d[0..P), ix[0..P) by (distance, index), ascending; one wave, P a power of two >= 64
__device__ __forceinline__ void wave_sort(double* d, int32_t* ix, int P, int lane) {
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (P >> 1); t += 64) {
                const int lo = ((t / stride) * stride * 2) + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double dl = d[lo], dh = d[hi];
                const int32_t il = ix[lo], ih = ix[hi];
                const bool gt = dl > dh || (dl == dh && il > ih);
                if (gt == up) { d[lo] = dh; d[hi] = dl; ix[lo] = ih; ix[hi] = il; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The same sort with the window in registers (P = 64 R entries, entry e = 64 r + lane): partners at least 64 apart sit in
// one lane, the others are exchanged by lane permutes -- no LDS traffic, hence none of the bank conflicts the strided
// pair accesses of wave_sort cost (9 in 10 of its LDS cycles).  Same network, same order.
template <int R>
__device__ __forceinline__ void wave_sort_regs(double (&d)[R], int32_t (&ix)[R], int lane) {
#pragma unroll
    for (int size = 2; size <= 64 * R; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {
                const int rs = stride >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r & rs) continue;
                    const int r2 = r | rs;
                    const bool up = (((r << 6) & size) == 0);
                    const bool gt = d[r] > d[r2] || (d[r] == d[r2] && ix[r] > ix[r2]);
                    if (gt == up) {
                        const double td = d[r]; d[r] = d[r2]; d[r2] = td;
                        const int32_t ti = ix[r]; ix[r] = ix[r2]; ix[r2] = ti;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = (r << 6) | lane;
                    const bool up = ((e & size) == 0);
                    const bool lower = ((lane & stride) == 0);
                    const double od = __shfl_xor(d[r], stride, 64);
                    const int32_t oi = __shfl_xor(ix[r], stride, 64);
                    const bool other_less = od < d[r] || (od == d[r] && oi < ix[r]);
                    const bool want_min = (lower == up);
                    if (want_min == other_less && !(od == d[r] && oi == ix[r])) { d[r] = od; ix[r] = oi; }
                }
            }
        }
    }
}

// one wave per query (WAVES per block, no block-level synchronisation).  The instances share the work by list length: an
// instance takes the queries with lo < min(length, cap) <= hi (SELMAX = kSelSmall: nearly all; kSelMax: the few longer or
// overflowed ones; kSelHuge: lists beyond 1024 entries, which only occur with the large cap of k > 80).
template <int CP, int SELMAX, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_knn_select(const float* __restrict__ E, const int32_t* __restrict__ perm, int64_t M, int K,
                                                    int include_self, const int32_t* __restrict__ ccount, const int32_t* __restrict__ cbuf,
                                                    int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                    int32_t* __restrict__ n_overflow, int cap, int lo, int hi) {
    __shared__ __attribute__((aligned(16))) double sd[WAVES][SELMAX];
    __shared__ int32_t si[WAVES][SELMAX];
    __shared__ __attribute__((aligned(16))) float sq[WAVES][CP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * WAVES + wave;
    if (q >= M) return;
    double* d = sd[wave];
    int32_t* ix = si[wave];
    float* qrow = sq[wave];
    for (int t = lane; t < CP; t += 64) qrow[t] = E[q * CP + t];
    const int cnt_all = ccount[q];
    const bool overflow = cnt_all > cap;
    const int cnt = overflow ? cap : cnt_all;
    if (!(cnt > lo && cnt <= hi) && !(cnt == 0 && lo == 0)) return;      // another instance's query
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // exact distances of the listed candidates, one per lane and step, then sort
    int P = 64;
    while (P < cnt) P <<= 1;
    auto candidate = [&](int t, double& dv, int32_t& iv) {
        dv = __builtin_huge_val();
        iv = 0x7fffffff;
        if (t < cnt) {
            const int64_t c = cbuf[q * (int64_t)cap + t];
            dv = exact_d2<CP>(qrow, E + c * CP);
            iv = perm[c];                            // ties are broken by the caller's point ids
        }
    };
    if (SELMAX == kSelSmall) {
        // the common instance keeps the window in registers; it only lands in LDS for the write-out / the overflow path
        auto sort_in_registers = [&](auto rtag) {
            constexpr int R = decltype(rtag)::value;
            double dr[R];
            int32_t ir[R];
#pragma unroll
            for (int r = 0; r < R; ++r) candidate(r * 64 + lane, dr[r], ir[r]);
            wave_sort_regs<R>(dr, ir, lane);
#pragma unroll
            for (int r = 0; r < R; ++r) { d[r * 64 + lane] = dr[r]; ix[r * 64 + lane] = ir[r]; }
        };
        if (P == 64) sort_in_registers(std::integral_constant<int, 1>());
        else if (P == 128) sort_in_registers(std::integral_constant<int, 2>());
        else sort_in_registers(std::integral_constant<int, 4>());
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        for (int t = lane; t < P; t += 64) {
            double dv;
            int32_t iv;
            candidate(t, dv, iv);
            d[t] = dv;
            ix[t] = iv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wave_sort(d, ix, P, lane);
    }
    int kept = cnt < K ? cnt : K;
    if (overflow) {
        // The list was cut at kCandCap entries.  Its k-th exact distance is still an upper bound of the true
        // k-th distance: rescan every point exactly, keep those not beyond it, and sort again.
        if (lane == 0) atomicAdd(n_overflow, 1);
        const double bound = d[K - 1];
        int fill = 0;                               // entries appended behind nothing: the window restarts empty
        for (int64_t c0 = 0; c0 < M; c0 += 64) {
            const int64_t c = c0 + lane;
            double dv = __builtin_huge_val();
            bool keep = false;
            if (c < M && (include_self || c != q)) {
                dv = exact_d2<CP>(qrow, E + c * CP);
                keep = dv <= bound;
            }
            const unsigned long long m = __ballot(keep);
            const int n_new = __popcll(m);
            if (fill + n_new > SELMAX) {           // pathological ties: compact to the best K and go on
                for (int t = fill + lane; t < SELMAX; t += 64) { d[t] = __builtin_huge_val(); ix[t] = 0x7fffffff; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                wave_sort(d, ix, SELMAX, lane);
                fill = K;
            }
            if (keep) {
                const int pos = fill + __popcll(m & ((1ull << lane) - 1ull));
                d[pos] = dv;
                ix[pos] = perm[c];
            }
            fill += n_new;
        }
        P = 64;
        while (P < fill) P <<= 1;
        for (int t = fill + lane; t < P; t += 64) { d[t] = __builtin_huge_val(); ix[t] = 0x7fffffff; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wave_sort(d, ix, P, lane);
        kept = fill < K ? fill : K;
    }
    const int64_t qo = perm[q];                     // row of the caller's table
    for (int s = lane; s < K; s += 64) {
        const bool ok = s < kept && ix[s] != 0x7fffffff;
        idx_out[qo * K + s] = ok ? ix[s] : -1;
        dist_out[qo * K + s] = ok ? d[s] : __builtin_huge_val();
    }
}

// ------------------------------------------------------------------------------------------------
// Other metrics (phenograph.cluster(primary_metric=...) behind dd.py:320-322: "manhattan" -> sklearn minkowski p = 1,
// "cosine" / "correlation" -> sklearn brute force): an exact scan in float64, one wave per query.  Not a fast path -- every
// query meets every point (M^2 C flop on the float64 VALU) -- but an exact one: candidates no farther than the current
// bound are collected in an LDS window; a full window is sorted, cut to the k best, and its k-th distance becomes the
// bound (so only the first windows see many candidates).  Ordering by (distance, index).  The distances are
//   manhattan   sum_c |a_c - b_c|
//   cosine      1 - a.b / (|a| |b|)     (rows are normalised first, a zero row stays zero: distance 1, as sklearn's normalize)
//   correlation cosine of the rows after subtracting their means
// dist2_out receives the squared distance (the convention of ddx_get_knn).
// ------------------------------------------------------------------------------------------------
// rows prepared for the scan: cosine -> x / |x|, correlation -> (x - mean) / |x - mean|, manhattan -> x     (float64)
__global__ void k_knn_generic_prepare(const float* __restrict__ emb, int64_t M, int C, int metric, double* __restrict__ out) {
#pragma clang fp contract(off)
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    double mean = 0.0;
    if (metric == 3) {
        for (int c = 0; c < C; ++c) mean += (double)emb[r * C + c];
        mean /= (double)C;
    }
    double n2 = 0.0;
    for (int c = 0; c < C; ++c) {
        const double v = (double)emb[r * C + c] - mean;
        n2 += v * v;
    }
    const double scale = (metric == 1) ? 1.0 : (n2 > 0.0 ? 1.0 / sqrt(n2) : 1.0);
    for (int c = 0; c < C; ++c) out[r * C + c] = ((double)emb[r * C + c] - mean) * scale;
}

template <int METRIC>   // 1 manhattan, 2 / 3 cosine on the prepared rows
__global__ void __launch_bounds__(256) k_knn_generic(const double* __restrict__ X, int64_t M, int C, int K, int include_self,
                                                     int32_t* __restrict__ idx_out, double* __restrict__ dist_out) {
#pragma clang fp contract(off)
    constexpr int WIN = 1024;
    __shared__ __attribute__((aligned(16))) double sd[4][WIN];
    __shared__ int32_t si[4][WIN];
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_smem[];
    double* sq = reinterpret_cast<double*>(gen_smem);              // [4][C] query rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= M) return;
    double* d = sd[wave];
    int32_t* ix = si[wave];
    double* qrow = sq + (size_t)wave * C;
    for (int t = lane; t < C; t += 64) qrow[t] = X[q * C + t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double bound = __builtin_huge_val();
    int fill = 0;
    for (int64_t c0 = 0; c0 < M; c0 += 64) {
        const int64_t c = c0 + lane;
        double dv = __builtin_huge_val();
        bool keep = false;
        if (c < M && (include_self || c != q)) {
            const double* row = X + c * C;
            double acc = 0.0;
            if (METRIC == 1) {
                for (int t = 0; t < C; ++t) acc = acc + fabs(qrow[t] - row[t]);
                dv = acc;
            } else {
                for (int t = 0; t < C; ++t) acc = acc + qrow[t] * row[t];
                dv = 1.0 - acc;
                if (dv < 0.0) dv = 0.0;                                 // (rounding: a row against itself)
            }
            keep = dv <= bound;
        }
        const unsigned long long m = __ballot(keep);
        const int n_new = __popcll(m);
        if (fill + n_new > WIN) {                                   // sort, keep the k best, tighten the bound
            for (int t = fill + lane; t < WIN; t += 64) { d[t] = __builtin_huge_val(); ix[t] = 0x7fffffff; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            wave_sort(d, ix, WIN, lane);
            fill = K;
            bound = d[K - 1];
            keep = keep && dv <= bound;
        }
        const unsigned long long m2 = __ballot(keep);
        if (keep) {
            const int pos = fill + __popcll(m2 & ((1ull << lane) - 1ull));
            d[pos] = dv;
            ix[pos] = (int32_t)c;
        }
        fill += __popcll(m2);
    }
    int P = 64;
    while (P < fill) P <<= 1;
    for (int t = fill + lane; t < P; t += 64) { d[t] = __builtin_huge_val(); ix[t] = 0x7fffffff; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    wave_sort(d, ix, P, lane);
    const int kept = fill < K ? fill : K;
    for (int s = lane; s < K; s += 64) {
        const bool ok = s < kept && ix[s] != 0x7fffffff;
        idx_out[q * K + s] = ok ? ix[s] : -1;
        dist_out[q * K + s] = ok ? d[s] * d[s] : __builtin_huge_val();
    }
}

int stage_knn_metric(ddx_ctx* ctx, int32_t k, int32_t include_self, int32_t metric) {
    const int64_t M = ctx->embM;
    const int C = ctx->C;
    if (k > 256) return set_err(ctx, DDX_E_UNSUPPORTED, "k=%d exceeds 256", k);
    DDX_TRY(ensure(ctx, ctx->pcaA, sizeof(double) * (size_t)M * C + 256));
    DDX_TRY(ensure(ctx, ctx->knn_idx, sizeof(int32_t) * (size_t)M * k));
    DDX_TRY(ensure(ctx, ctx->knn_dist, sizeof(double) * (size_t)M * k));
    double* X = ctx->pcaA.as<double>();
    ScopedTimer t(ctx, "knn_generic");
    k_knn_generic_prepare<<<(unsigned)ceil_div(M, 256), 256, 0, ctx->stream>>>(ctx->emb32.as<float>(), M, C, metric, X);
    const unsigned grid = (unsigned)ceil_div(M, 4);
    const size_t lds = sizeof(double) * 4 * (size_t)C;
    if (metric == 1)
        k_knn_generic<1><<<grid, 256, lds, ctx->stream>>>(X, M, C, k, include_self, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<double>());
    else
        k_knn_generic<2><<<grid, 256, lds, ctx->stream>>>(X, M, C, k, include_self, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<double>());
    DDX_HIP(ctx, hipGetLastError());
    ctx->knn_window_total = nullptr;
    ctx->K = k;
    ctx->knn_self = include_self != 0;
    ctx->have_knn = true;
    return DDX_OK;
}

// sort key of the point order: the first principal component
__global__ void k_knn_keys(const float* __restrict__ emb, int64_t M, int C, float* __restrict__ keys, int32_t* __restrict__ ids) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    keys[r] = emb[r * C];
    ids[r] = (int32_t)r;
}

// Candidate window of one emit block (its 4*16*kEmitRT consecutive queries in first-component order): a candidate c
// can only matter for query q if d2(q,c) <= T_q, and d2(q,c) >= (q_1 - c_1)^2, so c_1 must lie within sqrt(T_q) of
// q_1.  Points are sorted by that component, hence the admissible candidates of the whole block form one contiguous
// range of tiles [win[2b], win[2b+1]).  The radius carries a 1e-5 relative margin for the float32 square root.
template <int BW>      // waves (of 16 * kEmitRT queries) per emit block
__global__ void __launch_bounds__(64) k_knn_window(const float* __restrict__ p1, const float* __restrict__ thr, const float* __restrict__ nrm,
                                                   int64_t Mp, int32_t* __restrict__ win, unsigned long long* __restrict__ total_tiles) {
    constexpr int QB = BW * 16 * kEmitRT;
    const int lane = threadIdx.x;
    const int64_t qb = (int64_t)blockIdx.x * QB;
    float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
    for (int t = lane; t < QB; t += 64) {
        const int64_t q = qb + t;
        if (!(nrm[q] < __builtin_huge_valf())) continue;          // padding query
        const float T = thr[q];
        const float x = p1[q];
        const float r = T < __builtin_huge_valf() ? sqrtf(fmaxf(T, 0.f)) * 1.00001f + fabsf(x) * 2.4e-7f + 1e-30f : __builtin_huge_valf();
        lo = fminf(lo, x - r);
        hi = fmaxf(hi, x + r);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if (lane != 0) return;
    int64_t first = 0, last = 0;
    if (lo <= hi) {
        int64_t a = 0, b = Mp;                       // first position with p1 >= lo
        while (a < b) { const int64_t m = (a + b) >> 1; if (p1[m] < lo) a = m + 1; else b = m; }
        first = a;
        a = first; b = Mp;                           // first position with p1 > hi
        while (a < b) { const int64_t m = (a + b) >> 1; if (p1[m] <= hi) a = m + 1; else b = m; }
        last = a;
    }
    win[2 * blockIdx.x] = (int32_t)(first >> 4);
    win[2 * blockIdx.x + 1] = (int32_t)((last + 15) >> 4);
    atomicAdd(total_tiles, (unsigned long long)(((last + 15) >> 4) - (first >> 4)));     // statistics only (bench.py's flop count)
}

int stage_knn(ddx_ctx* ctx, int32_t k, int32_t include_self) {
    const int64_t M = ctx->embM;
    const int C = ctx->C;
    if (C > kMaxDim) return set_err(ctx, DDX_E_UNSUPPORTED, "embedding dimension %d exceeds %d", C, kMaxDim);
    constexpr int kMaxK = 256;
    if (k > kMaxK) return set_err(ctx, DDX_E_UNSUPPORTED, "k=%d exceeds %d", k, kMaxK);
    // k beyond what one bound launch ranks (16 lanes x (kBoundKeepLarge - 1) kept values per query): the sample window is
    // dealt out over `groups` launches, each bounding the ceil(k / groups)-th neighbour among its share (k_knn_bound)
    const int groups = (int)ceil_div(k, 16 * (kBoundKeepLarge - 1));
    const int k_bound = (int)ceil_div(k, groups);
    const bool keep_small = k_bound <= 16 * (kBoundKeepSmall - 1);
    const int cap = k <= 16 * (kBoundKeepLarge - 1) ? kCandCap : kCandCapLarge;      // candidate slots per query
    const int CP = (C <= 32) ? 32 : (C <= 64 ? 64 : 128);
    const int64_t Mp = ceil_div(M, 256) * 256;                 // whole blocks of queries in both MFMA passes
    // workspace (reuses the PCA row buffer): E [Mp*CP] | Et [Mp*CP] | Eb [Mp*CP as bf16 hi+lo] | nrm [Mp] | thr [Mp] | p1 [Mp] | keys [2*Mp]
    //            | ccount [Mp+64] | ids [2*Mp] | win [2*blocks] | cbuf [Mp*cap]
    const size_t f_words = (size_t)Mp * CP * 3 + 9 * (size_t)Mp + 16;
    const bool wide = ctx->opt.knn_bf16 && M >= kEmitWideFrom && !getenv("DDX_KNN_EMIT_NARROW");     // 8-wave emit blocks
    const int64_t emit_blocks = Mp / ((wide ? 8 : 4) * 16 * kEmitRT);
    const int xcd_chunk = ctx->opt.knn_xcd_chunk;                                  // DDX_KNN_XCD_CHUNK (0: workgroups in launch order)
    const bool fold = ctx->opt.knn_bf16 && ctx->opt.knn_fold && C <= 30;          // threshold folded into the operands (k_knn_fold)
    const size_t i_words = (size_t)Mp + 64 + 2 * (size_t)Mp + 2 * (size_t)emit_blocks + 64 + (size_t)Mp * cap + (fold ? (size_t)Mp * 32 + 16 : 0);
    DDX_TRY(ensure(ctx, ctx->pcaA, sizeof(float) * f_words + sizeof(int32_t) * i_words + 256));
    DDX_TRY(ensure(ctx, ctx->knn_idx, sizeof(int32_t) * (size_t)M * k));
    DDX_TRY(ensure(ctx, ctx->knn_dist, sizeof(double) * (size_t)M * k));
    float* E = ctx->pcaA.as<float>();
    float* Et = E + (size_t)Mp * CP;
    __bf16* Eb = reinterpret_cast<__bf16*>(Et + (size_t)Mp * CP);      // 2 parts x 2 bytes = CP floats per point
    float* nrm = Et + 2 * (size_t)Mp * CP;
    float* thr = nrm + Mp;
    float* p1 = thr + Mp;
    float* keys_in = p1 + Mp;
    float* keys_out = keys_in + Mp;
    f4* start4 = reinterpret_cast<f4*>(keys_out + Mp + ((4 - ((3 * (size_t)Mp * CP + 5 * (size_t)Mp) & 3)) & 3));   // 16-byte aligned
    const bool bf = ctx->opt.knn_bf16;                                  // DDX_KNN_SCREEN=f32 selects the float32 MFMA screen
    int32_t* ccount = reinterpret_cast<int32_t*>(reinterpret_cast<float*>(start4) + 4 * (size_t)Mp);   // [Mp] + overflow counter at [Mp]
    int32_t* ids_in = ccount + Mp + 64;
    int32_t* perm = ids_in + Mp;
    int32_t* win = perm + Mp;
    int32_t* cbuf = win + 2 * emit_blocks + 64;
    __bf16* Ebq = reinterpret_cast<__bf16*>((reinterpret_cast<uintptr_t>(cbuf + (size_t)Mp * cap) + 15) & ~(uintptr_t)15);   // query operands of the folded emit pass
    // Order the points by their first principal component (stable radix sort: ties by id).  Every pass below works
    // in that order: a query's neighbours are then confined to a window of positions around it (k_knn_window).
    {
        ScopedTimer t(ctx, "knn_prepare");
        k_knn_keys<<<(unsigned)ceil_div(M, 256), 256, 0, ctx->stream>>>(ctx->emb32.as<float>(), M, C, keys_in, ids_in);
        size_t tmp_bytes = 0;
        DDX_HIP(ctx, prim::sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, ids_in, perm, (int)M, 0, 32, ctx->stream));
        DDX_TRY(ensure(ctx, ctx->sort_tmp, tmp_bytes));
        DDX_HIP(ctx, prim::sort_pairs(ctx->sort_tmp.p, tmp_bytes, keys_in, keys_out, ids_in, perm, (int)M, 0, 32, ctx->stream));
        k_knn_prepare<<<(unsigned)ceil_div(Mp, 256), 256, 0, ctx->stream>>>(ctx->emb32.as<float>(), perm, M, Mp, C, CP, E, Et, Eb, nrm, p1, start4);
    }
    DDX_HIP(ctx, hipMemsetAsync(ccount, 0, sizeof(int32_t) * (Mp + 64), ctx->stream));
    // Bound pass: the K-th smallest distance inside the nsamp tiles nearest to the query in first-component order is
    // an upper bound of its true K-th distance (any subset gives one; this subset holds most of the true neighbours).
    const int64_t ntiles = Mp >> 4;
    // The subset grows with the point count (1/16 of the tiles, at least 512): a fixed-size subset would hold an ever
    // smaller share of the true neighbours, T_q would loosen and the candidate lists overflow.
    int64_t nsamp = std::max<int64_t>(512, ntiles / 16);
    if (ctx->opt.knn_sample_tiles > 0) nsamp = ctx->opt.knn_sample_tiles;
    if (nsamp < 2 * (int64_t)ceil_div(k, 16) + 8) nsamp = 2 * (int64_t)ceil_div(k, 16) + 8;
    if (nsamp > ntiles) nsamp = ntiles;
    const int64_t stride = groups;
    const int64_t nsamp_g = std::max<int64_t>(1, nsamp / groups);                    // tiles per launch
    {
        ScopedTimer t(ctx, "knn_bound");
        const int64_t nb_bound = Mp / (4 * 16 * kBoundRT);
        const unsigned grid = (unsigned)(bf && xcd_chunk > 0 ? ceil_div(nb_bound, 8 * (int64_t)xcd_chunk) * 8 * xcd_chunk : nb_bound);
#define DDX_BOUND_LAUNCH(KERNEL, OPERAND)                                                                                        \
    do {                                                                                                                       \
        for (int g = 0; g < groups; ++g) {                                                                                     \
            if (CP == 32 && keep_small) KERNEL<32, kBoundKeepSmall><<<grid, 256, 0, ctx->stream>>>(OPERAND, nrm, Mp, k_bound, include_self, nsamp_g, stride, g, g > 0, thr XCDARG); \
            else if (CP == 32) KERNEL<32, kBoundKeepLarge><<<grid, 256, 0, ctx->stream>>>(OPERAND, nrm, Mp, k_bound, include_self, nsamp_g, stride, g, g > 0, thr XCDARG);      \
            else if (CP == 64 && keep_small) KERNEL<64, kBoundKeepSmall><<<grid, 256, 0, ctx->stream>>>(OPERAND, nrm, Mp, k_bound, include_self, nsamp_g, stride, g, g > 0, thr XCDARG); \
            else if (CP == 64) KERNEL<64, kBoundKeepLarge><<<grid, 256, 0, ctx->stream>>>(OPERAND, nrm, Mp, k_bound, include_self, nsamp_g, stride, g, g > 0, thr XCDARG);      \
            else if (keep_small) KERNEL<128, kBoundKeepSmall><<<grid, 256, 0, ctx->stream>>>(OPERAND, nrm, Mp, k_bound, include_self, nsamp_g, stride, g, g > 0, thr XCDARG);    \
            else KERNEL<128, kBoundKeepLarge><<<grid, 256, 0, ctx->stream>>>(OPERAND, nrm, Mp, k_bound, include_self, nsamp_g, stride, g, g > 0, thr XCDARG);                    \
        }                                                                                                                      \
    } while (0)
#define XCDARG , xcd_chunk
        if (bf) DDX_BOUND_LAUNCH(k_knn_bound_bf, Eb);
#undef XCDARG
#define XCDARG
        else DDX_BOUND_LAUNCH(k_knn_bound, Et);
#undef XCDARG
#undef DDX_BOUND_LAUNCH
    }
    {
        ScopedTimer t(ctx, "knn_emit");
        const unsigned grid = (unsigned)emit_blocks;
        const unsigned grid_x = (unsigned)(xcd_chunk > 0 ? ceil_div(emit_blocks, 8 * (int64_t)xcd_chunk) * 8 * xcd_chunk : emit_blocks);
        const int dbg_mode = ctx->opt.knn_ablation;     // timing ablations (wrong results): non-zero only in -DDDX_ABLATION builds
        if (wide) k_knn_window<8><<<grid, 64, 0, ctx->stream>>>(p1, thr, nrm, Mp, win, reinterpret_cast<unsigned long long*>(ccount + Mp + 2));
        else k_knn_window<4><<<grid, 64, 0, ctx->stream>>>(p1, thr, nrm, Mp, win, reinterpret_cast<unsigned long long*>(ccount + Mp + 2));
#define DDX_EMIT_BF(CPV, FOLDV, QUERY)                                                                                                                      \
    do {                                                                                                                                                    \
        if (wide) k_knn_emit_bf<CPV, FOLDV, 8><<<grid_x, 512, 0, ctx->stream>>>(Eb, QUERY, nrm, start4, thr, Mp, include_self, ccount, cbuf, win, dbg_mode, cap, xcd_chunk); \
        else k_knn_emit_bf<CPV, FOLDV, 4><<<grid_x, 256, 0, ctx->stream>>>(Eb, QUERY, nrm, start4, thr, Mp, include_self, ccount, cbuf, win, dbg_mode, cap, xcd_chunk);      \
    } while (0)
        if (bf && fold) {
            k_knn_fold<<<(unsigned)ceil_div(Mp * 8, 256), 256, 0, ctx->stream>>>(nrm, thr, Mp, Eb, Ebq);
            DDX_EMIT_BF(32, true, Ebq);
        } else if (bf && CP == 32) DDX_EMIT_BF(32, false, Eb);
        else if (bf && CP == 64) DDX_EMIT_BF(64, false, Eb);
        else if (bf) DDX_EMIT_BF(128, false, Eb);
#undef DDX_EMIT_BF
        else if (CP == 32) k_knn_emit<32><<<grid, 256, 0, ctx->stream>>>(Et, nrm, thr, Mp, include_self, ccount, cbuf, win, cap);
        else if (CP == 64) k_knn_emit<64><<<grid, 256, 0, ctx->stream>>>(Et, nrm, thr, Mp, include_self, ccount, cbuf, win, cap);
        else k_knn_emit<128><<<grid, 256, 0, ctx->stream>>>(Et, nrm, thr, Mp, include_self, ccount, cbuf, win, cap);
    }
    {
        ScopedTimer t(ctx, "knn_select");
        const unsigned g2 = (unsigned)ceil_div(M, 4);
        int32_t* ki = ctx->knn_idx.as<int32_t>();
        double* kd = ctx->knn_dist.as<double>();
#define DDX_SELECT_LAUNCH(CPV)                                                                                                                          \
    do {                                                                                                                                                \
        k_knn_select<CPV, kSelSmall, 4><<<g2, 256, 0, ctx->stream>>>(E, perm, M, k, include_self, ccount, cbuf, ki, kd, ccount + Mp, cap, 0, kSelSmall);    \
        k_knn_select<CPV, kSelMax, 4><<<g2, 256, 0, ctx->stream>>>(E, perm, M, k, include_self, ccount, cbuf, ki, kd, ccount + Mp, cap, kSelSmall, kSelMax); \
        if (cap > kSelMax)                                                                                                                              \
            k_knn_select<CPV, kSelHuge, 1><<<(unsigned)M, 64, 0, ctx->stream>>>(E, perm, M, k, include_self, ccount, cbuf, ki, kd, ccount + Mp, cap, kSelMax, kSelHuge); \
    } while (0)
        if (CP == 32) DDX_SELECT_LAUNCH(32);
        else if (CP == 64) DDX_SELECT_LAUNCH(64);
        else DDX_SELECT_LAUNCH(128);
#undef DDX_SELECT_LAUNCH
    }
    DDX_HIP(ctx, hipGetLastError());
    if (ctx->opt.knn_debug) {
        std::vector<int32_t> h(Mp + 1);
        DDX_HIP(ctx, hipMemcpyAsync(h.data(), ccount, sizeof(int32_t) * (Mp + 1), hipMemcpyDeviceToHost, ctx->stream));
        DDX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double sum = 0; int mx = 0; int64_t over = 0;
        for (int64_t i = 0; i < M; ++i) { sum += h[i]; if (h[i] > mx) mx = h[i]; over += h[i] > cap; }
        std::vector<int32_t> hw(2 * emit_blocks);
        DDX_HIP(ctx, hipMemcpy(hw.data(), win, sizeof(int32_t) * 2 * emit_blocks, hipMemcpyDeviceToHost));
        double wsum = 0;
        for (int64_t b = 0; b < emit_blocks; ++b) wsum += hw[2 * b + 1] - hw[2 * b];
        fprintf(stderr, "[knn] k=%d sample tiles=%lld: candidates/query mean %.1f max %d, overflowed %lld (counter %d); emit window %.1f%% of the tiles\n",
                k, (long long)nsamp, sum / M, mx, (long long)over, h[Mp], 100.0 * wsum / ((double)emit_blocks * (double)ntiles));
    }
    ctx->knn_window_total = reinterpret_cast<const unsigned long long*>(ccount + Mp + 2);
    ctx->knn_window_pairs = (double)emit_blocks * (double)ntiles;
    ctx->K = k;
    ctx->knn_self = include_self != 0;
    ctx->have_knn = true;
    return DDX_OK;
}

// ------------------------------------------------------------------------------------------------
// graphs
// ------------------------------------------------------------------------------------------------
// per row: copy of the neighbour list sorted by index (insertion sort in LDS, K <= 64)
__global__ void __launch_bounds__(64) k_sort_neighbours(const int32_t* __restrict__ idx, int64_t M, int K,
                                                        int32_t* __restrict__ sorted) {
    extern __shared__ int32_t buf[];  // [K][64]
    const int tid = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * 64 + tid;
    if (r >= M) return;
    for (int s = 0; s < K; ++s) {
        const int32_t v = idx[r * K + s];
        int pos = s;
        while (pos > 0 && buf[(pos - 1) * 64 + tid] > v) {
            buf[pos * 64 + tid] = buf[(pos - 1) * 64 + tid];
            --pos;
        }
        buf[pos * 64 + tid] = v;
    }
    for (int s = 0; s < K; ++s) sorted[r * K + s] = buf[s * 64 + tid];
}

__device__ __forceinline__ bool contains_sorted(const int32_t* __restrict__ a, int n, int32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo < n && a[lo] == key;
}

// one thread per directed kNN relation (i -> j = idx[i][a]).
// modes 0/1: Jaccard J = |N(i) & N(j)| / (2K - |N(i) & N(j)|)   (phenograph jaccard_kernel)
//   mode 0 (prune): weight J*J when the relation is mutual, else 0
//   mode 1        : weight J when mutual, J/2 otherwise ((J + J^T)/2)
// mode 2: unit weight, self relation dropped.
// Non-mutual relations are flagged with a negative weight: the host adds the reverse entry for them.
__global__ void __launch_bounds__(256) k_edge_weights(const int32_t* __restrict__ idx, const int32_t* __restrict__ sorted,
                                                      int64_t M, int K, int mode, double* __restrict__ w_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * K) return;
    const int64_t i = t / K;
    const int32_t j = idx[t];
    if (j < 0 || j == i) { w_out[t] = 0.0; return; }
    const int32_t* Ni = sorted + i * K;
    const int32_t* Nj = sorted + (int64_t)j * K;
    const bool mutual = contains_sorted(Nj, K, (int32_t)i);
    double w;
    if (mode == 2) {
        w = 1.0;
    } else {
        int a = 0, b = 0, shared = 0;
        while (a < K && b < K) {
            const int32_t x = Ni[a], y = Nj[b];
            if (x == y) { ++shared; ++a; ++b; }
            else if (x < y) ++a;
            else ++b;
        }
        const double J = (double)shared / (2.0 * (double)K - (double)shared);
        if (mode == 0) w = mutual ? J * J : 0.0;
        else w = mutual ? (J + J) / 2.0 : J / 2.0;
    }
    w_out[t] = mutual ? w : -w;
}

// The same weights, one wave per node (K <= 64): N(i) sits sorted in LDS, lane l holds the l-th relation of i and,
// relation by relation, the wave loads N(j) (one coalesced row), every lane looks its element up in N(i) by
// binary search and a ballot counts the shared neighbours.  HALF (K <= 32, the shipped settings): the two halves of the
// wave take two relations at a time -- lanes 0-31 the even one, lanes 32-63 the odd one -- so all 64 lanes search.
// Four steps (four or eight rows) are in flight at a time.
template <bool HALF>
__global__ void __launch_bounds__(256) k_edge_weights_wave(const int32_t* __restrict__ idx, const int32_t* __restrict__ sorted,
                                                           int64_t M, int K, int mode, double* __restrict__ w_out) {
    __shared__ int32_t nS[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= M) return;
    int32_t* Ni = nS[wave];
    Ni[lane] = lane < K ? sorted[i * K + lane] : 0x7fffffff;
    const int32_t myj = lane < K ? idx[i * K + lane] : -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int PER = HALF ? 2 : 1;                // relations per step
    const int half = HALF ? (lane >> 5) : 0;
    const int sub = HALF ? (lane & 31) : lane;       // element of N(j) this lane looks up
    double myw = 0.0;
    auto weight = [&](int shared, bool mutual) -> double {
        double w;
        if (mode == 2) {
            w = 1.0;
        } else {
            const double J = (double)shared / (2.0 * (double)K - (double)shared);
            if (mode == 0) w = mutual ? J * J : 0.0;
            else w = mutual ? (J + J) / 2.0 : J / 2.0;
        }
        return mutual ? w : -w;
    };
    for (int t0 = 0; t0 < K; t0 += 4 * PER) {
        int32_t j[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rel = t0 + PER * u + half;
            j[u] = rel < K ? __shfl(myj, rel, 64) : -1;
            const bool ok = j[u] >= 0 && j[u] != (int32_t)i;
            y[u] = (ok && sub < K) ? sorted[(int64_t)j[u] * K + sub] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (t0 + PER * u >= K) break;
            const bool ok = j[u] >= 0 && j[u] != (int32_t)i;
            const bool valid = ok && sub < K;
            bool found = false;
            if (valid) {
                int lo = 0, hi = K;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (Ni[mid] < y[u]) lo = mid + 1; else hi = mid;
                }
                found = lo < K && Ni[lo] == y[u];
            }
            const unsigned long long fm = __ballot(found);
            const unsigned long long mm = __ballot(valid && y[u] == (int32_t)i);
            if (HALF) {
                // (the relation's validity is known to its own lane: an invalid relation keeps weight 0)
                const double w0 = weight(__popc((unsigned)fm), (unsigned)mm != 0u);
                const double w1 = weight(__popc((unsigned)(fm >> 32)), (unsigned)(mm >> 32) != 0u);
                const bool mine_ok = myj >= 0 && myj != (int32_t)i;
                if (lane == t0 + 2 * u) myw = mine_ok ? w0 : 0.0;
                if (lane == t0 + 2 * u + 1) myw = mine_ok ? w1 : 0.0;
            } else {
                const double w = ok ? weight(__popcll(fm), mm != 0ull) : 0.0;
                if (lane == t0 + u) myw = w;
            }
        }
    }
    if (lane < K) w_out[i * K + lane] = myw;
}

// ---- umap connectivities (mode 3: what sc.tl.leiden clusters on) --------------------------------------------
// oracle/dd_oracle.py:umap_connectivities states the computation (umap-learn's fuzzy_simplicial_set with
// local_connectivity = 1, set_op_mix_ratio = 1): distances rounded to float32, everything else float64.
// One thread per point: rho = smallest positive distance, sigma from 64 bisection steps, then the directed weights.
__global__ void __launch_bounds__(256) k_umap_directed(const int32_t* __restrict__ idx, const double* __restrict__ d2, int64_t M, int K,
                                                       double target, double* __restrict__ val) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double* row = d2 + i * K;
    double rho = __builtin_huge_val(), mean = 0.0;
    for (int j = 0; j < K; ++j) {
        const double d = (double)(float)sqrt(row[j]);
        if (d > 0.0 && d < rho) rho = d;
        mean = mean + d;
    }
    if (!(rho < __builtin_huge_val())) rho = 0.0;
    mean = mean / (double)K;
    double lo = 0.0, hi = __builtin_huge_val(), mid = 1.0;
    for (int it = 0; it < 64; ++it) {
        double psum = 0.0;
        for (int j = 1; j < K; ++j) {
            const double g = (double)(float)sqrt(row[j]) - rho;
            psum = psum + (g > 0.0 ? exp(-(g / mid)) : 1.0);
        }
        if (psum > target) {
            hi = mid;
            mid = (lo + hi) / 2.0;
        } else {
            lo = mid;
            mid = (hi < __builtin_huge_val()) ? (lo + hi) / 2.0 : mid * 2.0;
        }
    }
    const double floor_v = 1e-3 * mean;
    const double sigma = mid > floor_v ? mid : floor_v;
    for (int j = 0; j < K; ++j) {
        const int32_t c = idx[i * K + j];
        const double g = (double)(float)sqrt(row[j]) - rho;
        double v = (g <= 0.0 || sigma == 0.0) ? 1.0 : exp(-(g / sigma));
        if (c == (int32_t)i || c < 0) v = 0.0;
        val[i * K + j] = v;
    }
}

// fuzzy union a + b - a*b with the reverse relation (b = 0 when j does not list i); a one-directional relation is
// flagged by a negative weight, as for the other graph types (the assembly adds its reverse entry)
__global__ void __launch_bounds__(256) k_umap_union(const int32_t* __restrict__ idx, const double* __restrict__ val, int64_t M, int K,
                                                    double* __restrict__ w_out) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * K) return;
    const int64_t i = t / K;
    const int32_t j = idx[t];
    const double a = val[t];
    if (j < 0 || j == (int32_t)i || a == 0.0) { w_out[t] = 0.0; return; }
    double b = 0.0;
    bool mutual = false;
    for (int c = 0; c < K; ++c) {
        if (idx[(int64_t)j * K + c] == (int32_t)i) {
            const double bv = val[(int64_t)j * K + c];
            if (bv != 0.0) { b = bv; mutual = true; }
            break;
        }
    }
    w_out[t] = mutual ? (a + b) - a * b : -a;
}

// ---- symmetric CSR on the device ----------------------------------------------------------------------
// every relation with a non-zero weight contributes the pair (i,j); a one-directional relation (negative
// flag) also contributes (j,i).  Pairs are keyed (row << bits(M) | column) and radix-sorted, which yields rows
// in order and columns ascending inside a row (the adjacency order the community-detection spec visits).
__global__ void __launch_bounds__(256) k_pair_count(const double* __restrict__ w, int64_t n, int32_t* __restrict__ cnt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double v = w[t];
    cnt[t] = (v == 0.0) ? 0 : (v < 0.0 ? 2 : 1);
}

__global__ void __launch_bounds__(256) k_pair_emit(const int32_t* __restrict__ idx, const double* __restrict__ w, int64_t n,
                                                   int K, int shift, const int64_t* __restrict__ offs, uint64_t* __restrict__ keys,
                                                   double* __restrict__ vals) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double v = w[t];
    if (v == 0.0) return;
    const uint64_t i = (uint64_t)(t / K), j = (uint64_t)idx[t];
    const double av = v < 0.0 ? -v : v;
    int64_t o = offs[t];
    keys[o] = (i << shift) | j;            // shift = bits of the node count: the sort only passes over 2*shift bits
    vals[o] = av;
    if (v < 0.0) {
        keys[o + 1] = (j << shift) | i;
        vals[o + 1] = av;
    }
}

__global__ void k_rowptr_from_keys(const uint64_t* __restrict__ keys, int64_t n, int64_t M, int shift, int64_t* __restrict__ indptr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > M) return;
    int64_t lo = 0, hi = n;   // first key with row >= r
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(keys[mid] >> shift) < r) lo = mid + 1; else hi = mid;
    }
    indptr[r] = lo;
}

__global__ void k_cols_from_keys(const uint64_t* __restrict__ keys, int64_t n, int shift, int32_t* __restrict__ cols) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) cols[t] = (int32_t)(keys[t] & ((1ull << shift) - 1ull));
}

// device part: relation weights for the kNN table currently held by the context
static int graph_weights_device(ddx_ctx* ctx, int32_t mode);

int stage_graph_relations(ddx_ctx* ctx, int32_t mode, int32_t* idx_host, double* w_host) {
    const int64_t M = ctx->embM;
    const int K = ctx->K;
    DDX_TRY(graph_weights_device(ctx, mode));
    DDX_HIP(ctx, hipMemcpyAsync(idx_host, ctx->knn_idx.p, sizeof(int32_t) * (size_t)M * K, hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, hipMemcpyAsync(w_host, ctx->edge_w.p, sizeof(double) * (size_t)M * K, hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DDX_OK;
}

// host part (context-free, thread-safe): symmetric CSR from the relation table.  Every relation with a
// non-zero weight contributes (i,j); a non-mutual one (negative flag) also contributes (j,i).  Rows are
// sorted by neighbour index (the community-detection spec visits adjacency in this order).
void assemble_graph(int64_t M, int K, const int32_t* idx, const double* w, std::vector<int64_t>& ip,
                    std::vector<int32_t>& gi, std::vector<double>& gw) {
    ip.assign(M + 1, 0);
    for (int64_t i = 0; i < M; ++i)
        for (int a = 0; a < K; ++a) {
            const double v = w[i * K + a];
            if (v == 0.0) continue;
            ip[i + 1]++;
            if (v < 0.0) ip[idx[i * K + a] + 1]++;
        }
    for (int64_t i = 0; i < M; ++i) ip[i + 1] += ip[i];
    const int64_t E = ip[M];
    gi.assign(E, 0);
    gw.assign(E, 0.0);
    std::vector<int64_t> cur(ip.begin(), ip.end() - 1);
    for (int64_t i = 0; i < M; ++i)
        for (int a = 0; a < K; ++a) {
            const double v = w[i * K + a];
            if (v == 0.0) continue;
            const int32_t j = idx[i * K + a];
            const double av = v < 0.0 ? -v : v;
            gi[cur[i]] = j;
            gw[cur[i]++] = av;
            if (v < 0.0) {
                gi[cur[j]] = (int32_t)i;
                gw[cur[j]++] = av;
            }
        }
    std::vector<std::pair<int32_t, double>> tmp;
    for (int64_t i = 0; i < M; ++i) {
        const int64_t b = ip[i], e = ip[i + 1];
        tmp.resize(e - b);
        for (int64_t p = b; p < e; ++p) tmp[p - b] = {gi[p], gw[p]};
        std::sort(tmp.begin(), tmp.end(), [](const std::pair<int32_t, double>& x, const std::pair<int32_t, double>& y) { return x.first < y.first; });
        for (int64_t p = b; p < e; ++p) {
            gi[p] = tmp[p - b].first;
            gw[p] = tmp[p - b].second;
        }
    }
}

// relation weights on the device (no copies)
static int graph_weights_device(ddx_ctx* ctx, int32_t mode) {
    const int64_t M = ctx->embM;
    const int K = ctx->K;
    DDX_TRY(ensure(ctx, ctx->knn_sorted, sizeof(int32_t) * (size_t)M * K));
    DDX_TRY(ensure(ctx, ctx->edge_w, sizeof(double) * (size_t)M * K));
    ScopedTimer t(ctx, "graph_weights");
    if (mode == 3) {
        // directed weights go to knn_sorted's neighbour (reused as float64 scratch: M*K doubles live in pcaOp)
        DDX_TRY(ensure(ctx, ctx->pcaOp, sizeof(double) * (size_t)M * K));
        double* val = ctx->pcaOp.as<double>();
        k_umap_directed<<<(unsigned)ceil_div(M, 256), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<double>(), M, K, std::log2((double)K), val);
        k_umap_union<<<(unsigned)ceil_div(M * K, 256), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), val, M, K, ctx->edge_w.as<double>());
        return DDX_OK;
    }
    k_sort_neighbours<<<(unsigned)ceil_div(M, 64), 64, sizeof(int32_t) * K * 64, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), M, K, ctx->knn_sorted.as<int32_t>());
    if (K <= 32)
        k_edge_weights_wave<true><<<(unsigned)ceil_div(M, 4), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->knn_sorted.as<int32_t>(), M, K, mode,
                                                                                     ctx->edge_w.as<double>());
    else if (K <= 64)
        k_edge_weights_wave<false><<<(unsigned)ceil_div(M, 4), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->knn_sorted.as<int32_t>(), M, K, mode,
                                                                                      ctx->edge_w.as<double>());
    else
        k_edge_weights<<<(unsigned)ceil_div(M * K, 256), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->knn_sorted.as<int32_t>(), M, K, mode,
                                                                                ctx->edge_w.as<double>());
    return DDX_OK;
}

// whole graph stage on the device; the symmetric CSR lands in the context's host vectors
int stage_build_graph(ddx_ctx* ctx, int32_t mode) {
    const int64_t M = ctx->embM;
    const int K = ctx->K;
    const int64_t n = M * K;
    DDX_TRY(graph_weights_device(ctx, mode));
    // workspace of its own (not the PCA panel buffer: the graph of iteration i must outlive the PCA of iteration i + 1, whose
    // operator products run while the host finishes part B of iteration i -- part C then needs the graph again): offs i64[n+1] | keys u64[2n] x2 | vals f64[2n] x2 | indptr i64[M+1] | cols i32[2n] | cnt i32[n+1]
    // (sized by the same arithmetic that carves it: every piece is rounded up to 256 bytes)
    size_t bytes = 0;
    auto piece = [&](size_t sz) { const size_t o = bytes; bytes += (sz + 255) & ~(size_t)255; return o; };
    const size_t o_offs = piece(sizeof(int64_t) * (n + 1)), o_ka = piece(sizeof(uint64_t) * 2 * n), o_kb = piece(sizeof(uint64_t) * 2 * n);
    const size_t o_va = piece(sizeof(double) * 2 * n), o_vb = piece(sizeof(double) * 2 * n), o_ip = piece(sizeof(int64_t) * (M + 1));
    const size_t o_cols = piece(sizeof(int32_t) * 2 * n), o_cnt = piece(sizeof(int32_t) * (n + 1));
    ctx->g_nodes = -1;
    ctx->c_nodes = -1;
    DDX_TRY(ensure(ctx, ctx->graph_buf, bytes));
    unsigned char* base = ctx->graph_buf.as<unsigned char>();
    int64_t* offs = reinterpret_cast<int64_t*>(base + o_offs);
    uint64_t* keys_a = reinterpret_cast<uint64_t*>(base + o_ka);
    uint64_t* keys_b = reinterpret_cast<uint64_t*>(base + o_kb);
    double* vals_a = reinterpret_cast<double*>(base + o_va);
    double* vals_b = reinterpret_cast<double*>(base + o_vb);
    int64_t* d_indptr = reinterpret_cast<int64_t*>(base + o_ip);
    int32_t* d_cols = reinterpret_cast<int32_t*>(base + o_cols);
    int32_t* cnt = reinterpret_cast<int32_t*>(base + o_cnt);
    int64_t E = 0;
    {
        ScopedTimer t(ctx, "graph_assemble");
        k_pair_count<<<(unsigned)ceil_div(n, 256), 256, 0, ctx->stream>>>(ctx->edge_w.as<double>(), n, cnt);
        size_t tmp_bytes = 0;
        DDX_HIP(ctx, prim::exclusive_sum(nullptr, tmp_bytes, cnt, offs, (int)n + 1, ctx->stream));
        size_t tmp2 = 0;
        int shift = 1;                               // pairs are keyed row << shift | column with shift = bits(M)
        while (((int64_t)1 << shift) < M) ++shift;
        const int end_bit = 2 * shift;
        DDX_HIP(ctx, prim::sort_pairs(nullptr, tmp2, keys_a, keys_b, vals_a, vals_b, (int)(2 * n), 0, end_bit, ctx->stream));
        DDX_TRY(ensure(ctx, ctx->sort_tmp, std::max(tmp_bytes, tmp2)));
        // (k_pair_count fills cnt[0..n); the exclusive scan over n + 1 elements never adds cnt[n] to an output)
        DDX_HIP(ctx, prim::exclusive_sum(ctx->sort_tmp.p, tmp_bytes, cnt, offs, (int)n + 1, ctx->stream));
        DDX_HIP(ctx, hipMemcpyAsync(&E, offs + n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        k_pair_emit<<<(unsigned)ceil_div(n, 256), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->edge_w.as<double>(), n, K, shift, offs, keys_a, vals_a);
        DDX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (E > 0) {
            // the sort picks its algorithm (single block / merge / onesweep) by the element count, and each has its own
            // temporary-storage need: ask again for the actual count
            size_t tmp3 = 0;
            DDX_HIP(ctx, prim::sort_pairs(nullptr, tmp3, keys_a, keys_b, vals_a, vals_b, (int)E, 0, end_bit, ctx->stream));
            DDX_TRY(ensure(ctx, ctx->sort_tmp, tmp3));
            DDX_HIP(ctx, prim::sort_pairs(ctx->sort_tmp.p, tmp3, keys_a, keys_b, vals_a, vals_b, (int)E, 0, end_bit, ctx->stream));
            k_cols_from_keys<<<(unsigned)ceil_div(E, 256), 256, 0, ctx->stream>>>(keys_b, E, shift, d_cols);
        }
        k_rowptr_from_keys<<<(unsigned)ceil_div(M + 1, 256), 256, 0, ctx->stream>>>(keys_b, E, M, shift, d_indptr);
    }
    DDX_HIP(ctx, hipGetLastError());
    // the CSR stays on the device until ddx_get_graph copies it straight into the caller's buffers
    ctx->g_nodes = M;
    ctx->g_entries = E;
    ctx->g_d_indptr = d_indptr;
    ctx->g_d_cols = d_cols;
    ctx->g_d_vals = vals_b;
    return DDX_OK;
}

}  // namespace ddx
