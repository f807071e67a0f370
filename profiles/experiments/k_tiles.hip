// EXPERIMENT, not part of libddx (see profiles/r02o_mfma_products.txt for what was measured; this file is the last state
// tried -- A Q: k_tile_rows (v7) and the K-split k_tile_rows_part (v8, the one wired in), A^T Y: v5 -- and builds with
// profiles/experiments/mfma_products_integration.patch applied).
// libddx -- operator products of the randomized PCA on the matrix cores (DDX_SPMM=mfma; experimental).
//
// The LDS-staged products (k_pca.hip) read one 160-byte operand row per stored entry and are bound by the LDS pipe at
// 0.62 / 0.66 ms per launch.  Here the sparse matrix is turned, 64 rows x 32 columns at a time, into a DENSE half-
// precision tile in LDS and multiplied with v_mfma_f32_16x16x32_f16: a stored entry costs one 4-byte LDS write instead
// of a 160-byte read, and the ~90 % zeros of a tile cost matrix-core time that is otherwise idle.
//
// Precision.  Every value a is carried as hi = f16(s a), lo = f16(s a - hi) with a power-of-two scale s (22 bits);
// a product is hi*hi + hi*lo + lo*hi (three MFMAs, float32 accumulation inside the matrix core), and the float32
// accumulators are added into float64 registers every 128 columns.  Simulated on the CPU against an all-float64 run of
// scikit-learn's randomized PCA (profiles/tools/mfma_precision_sim.py, 30 000 x 3 000): 1.8e-6 per score column, against
// 7.9e-7 for the LDS products, 8.5e-6 for plain float32 GEMMs and 5e-5 for a bfloat16 split.
//
// dd.py:308-314 (sc.tl.pca -> sklearn randomized_svd: the products A Q and A^T Y of the power iterations).
#include <hip/hip_runtime.h>

#include "ddx_internal.h"

namespace ddx {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4t __attribute__((ext_vector_type(4)));
typedef int32_t i4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kTRows = 64;            // matrix rows of a tile (four MFMA row tiles), one row per lane while it is filled
constexpr int kTK = 32;               // columns of a tile = contraction depth of one MFMA
constexpr int kTNt = 3;               // MFMA column tiles: sketch widths up to 48
constexpr int kTStrideT = 144;         // bytes between the rows of the transposed tile of A^T Y (128 + 16)
constexpr int kTStride = 80;          // bytes between tile rows in LDS (64 + 16: the 16 rows of a fragment read hit distinct banks)
constexpr int kTFlush = 4;            // float32 accumulators are added into float64 every kTFlush tiles (128 columns)
constexpr float kTScaleA = 1024.0f;   // 2^10: |x - z| < 64 (log-normalised values, or scaled values clipped at 15)
constexpr int kTWaves = 4;            // waves per workgroup
constexpr int kTPre = 6;              // 64-record chunks of the next tile requested ahead

// ---- operand (the dense H x L or M x L factor) as MFMA B fragments ------------------------------------------------
// cmax[j] = max_r |X[r][j]| as float bits (non-negative floats order like unsigned integers)
__global__ void __launch_bounds__(256) k_tile_colmax(const double* __restrict__ X, int64_t R, int L, int64_t rows_per_block,
                                                     uint32_t* __restrict__ cmax) {
    __shared__ uint32_t red[256];
    const int tid = threadIdx.x, c = tid & 63, lane_r = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    float m = 0.f;
    if (c < L)
        for (int64_t r = r0 + lane_r; r < r1; r += 4) m = fmaxf(m, fabsf((float)X[r * L + c]));
    red[tid] = __float_as_uint(m);
    __syncthreads();
    if (tid < 64 && c < L) {
        uint32_t v = red[tid];
        for (int q = 1; q < 4; ++q) v = v > red[q * 64 + tid] ? v : red[q * 64 + tid];
        atomicMax(&cmax[c], v);
    }
}

// scale[j] = 2^e with max_r |X[r][j]| * 2^e in [2^12, 2^13]; inv[j] = 1 / (kTScaleA * scale[j])
__global__ void k_tile_scales(const uint32_t* __restrict__ cmax, int L, float* __restrict__ scale, double* __restrict__ inv) {
    const int j = threadIdx.x;
    if (j >= kTNt * 16) return;
    float s = 1.0f;
    if (j < L) {
        const float m = __uint_as_float(cmax[j]);
        if (m > 0.f && m < __builtin_huge_valf()) {
            int e;
            frexpf(m, &e);                        // m = f * 2^e, f in [0.5, 1)
            s = ldexpf(1.0f, 13 - e);
        }
    }
    scale[j] = s;
    inv[j] = 1.0 / ((double)kTScaleA * (double)s);
}

// frag[((slab * 3 + n) * 2 + part) * 64 + lane]: lane l holds rows 32*slab + 8*(l >> 4) + 0..7 of column 16*n + (l & 15)
__global__ void __launch_bounds__(256) k_tile_operand(const double* __restrict__ X, int64_t R, int L, const float* __restrict__ scale,
                                                      int64_t nslabs, h8* __restrict__ frag) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nslabs * kTNt * 64) return;
    const int lane = (int)(t & 63);
    const int64_t sn = t >> 6;
    const int n = (int)(sn % kTNt);
    const int64_t slab = sn / kTNt;
    const int j = 16 * n + (lane & 15);
    const float s = scale[j];
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t r = slab * kTK + 8 * (lane >> 4) + e;
        const float v = (r < R && j < L) ? (float)(X[r * L + j] * (double)s) : 0.f;
        const _Float16 h = (_Float16)v;
        hi[e] = h;
        lo[e] = (_Float16)(v - (float)h);
    }
    frag[(sn * 2 + 0) * 64 + lane] = hi;
    frag[(sn * 2 + 1) * 64 + lane] = lo;
}

// ---- the matrix as a stream of tiles ------------------------------------------------------------------------------
// Once per iteration the log-normalised CSR is re-ordered into tiles of 64 rows x 32 columns: the stored entries of a
// tile are consecutive 8-byte records {position inside the tile, value as (hi, lo) halves}, the tiles of a row block
// follow each other by column slab, and blk[b * (nslabs + 1) + s] is the first record of tile (b, s).  Both products
// stream these records with fully used lanes: A Q walks a row block's tiles, A^T Y walks the tiles of a column slab
// (2 KB runs, one per row block).  The order of the records inside a tile is irrelevant (every record has its own place
// in the dense tile), so the tile contents -- and with them the products -- do not depend on the scheduling.
constexpr int kPackSlabs = 16;                 // column slabs handled by one packing workgroup (512 columns)
constexpr int kPackCap = 7168;                 // records assembled in LDS per workgroup (56 KB); the rest go out directly

// rowseg2[row * (ng + 1) + g] = offset inside the row of its first entry with column >= g * 512  (k_row_segments, SR = 512)
__global__ void __launch_bounds__(256) k_tile_pack(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                   const float* __restrict__ x, const float* __restrict__ zcol,
                                                   const int32_t* __restrict__ rowseg2, int ng, int64_t M, int32_t H, int nslabs,
                                                   int32_t* __restrict__ blk, uint2* __restrict__ recs, int* __restrict__ flag) {
    __shared__ uint2 stage[kPackCap];
    __shared__ uint32_t hist[kPackSlabs], cnt[kPackSlabs], start[kPackSlabs + 1];
    __shared__ uint32_t before;                 // entries of the row block in front of this column group
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int64_t r0 = b * kTRows;
    const int32_t c0 = g * kPackSlabs * kTK;
    if (tid < kPackSlabs) { hist[tid] = 0; cnt[tid] = 0; }
    if (tid == 0) before = 0;
    __syncthreads();
    // 1. records per slab
    uint32_t mine_before = 0;
    for (int rr = wave; rr < kTRows; rr += 4) {
        const int64_t row = r0 + rr;
        if (row >= M) break;
        const int64_t base = indptr[row];
        const int32_t lo = rowseg2[row * (ng + 1) + g], hi = rowseg2[row * (ng + 1) + g + 1];
        if (lane == 0) mine_before += (uint32_t)lo;
        for (int32_t e = lo + lane; e < hi; e += 64) atomicAdd(&hist[(cols[base + e] - c0) >> 5], 1u);
    }
    if (lane == 0) atomicAdd(&before, mine_before);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int sl = 0; sl < kPackSlabs; ++sl) { start[sl] = run; run += hist[sl]; }
        start[kPackSlabs] = run;
    }
    __syncthreads();
    const int64_t r_last = (r0 + kTRows < M ? r0 + kTRows : M);
    const uint32_t gbase = (uint32_t)indptr[r0] + before;          // (positions fit 31 bits)
    for (int sl = tid; sl < kPackSlabs; sl += 256) {
        const int sgl = g * kPackSlabs + sl;
        if (sgl < nslabs) blk[b * (nslabs + 1) + sgl] = (int32_t)(gbase + start[sl]);
    }
    if (g == ng - 1 && tid == 0) blk[b * (nslabs + 1) + nslabs] = (int32_t)indptr[r_last];
    // 2. records into the staging range, slab by slab
    bool bad = false;
    for (int rr = wave; rr < kTRows; rr += 4) {
        const int64_t row = r0 + rr;
        if (row >= M) break;
        const int64_t base = indptr[row];
        const int32_t lo = rowseg2[row * (ng + 1) + g], hi = rowseg2[row * (ng + 1) + g + 1];
        for (int32_t e = lo + lane; e < hi; e += 64) {
            const int32_t c = cols[base + e];
            const float v = (x[base + e] - zcol[c]) * kTScaleA;
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            bad = bad || !(fabsf(v) < 60000.0f);
            const int sl = (c - c0) >> 5;
            const uint32_t slot = start[sl] + atomicAdd(&cnt[sl], 1u);
            uint2 rec;
            rec.x = ((uint32_t)rr * kTStride + (uint32_t)(c & 31) * 2u) | (((uint32_t)(c & 31) * kTStrideT + (uint32_t)rr * 2u) << 16);
            rec.y = (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
            if (slot < (uint32_t)kPackCap) stage[slot] = rec; else recs[gbase + slot] = rec;
        }
    }
    if (__any(bad) && lane == 0) atomicExch(flag, 1);
    __syncthreads();
    const uint32_t n = start[kPackSlabs] < (uint32_t)kPackCap ? start[kPackSlabs] : (uint32_t)kPackCap;
    for (uint32_t i = tid; i < n; i += 256) recs[gbase + i] = stage[i];
}

__device__ __forceinline__ void tile_clear(unsigned char* t, int bytes, int lane) {
    for (int i = lane * 16; i < bytes; i += 64 * 16) *reinterpret_cast<f4t*>(t + i) = f4t{0, 0, 0, 0};
}

// ---- A Q: one wave per 64 matrix rows -----------------------------------------------------------------------------
// The wave streams the tiles of its row block.  Two pairs of LDS tiles alternate: while the matrix cores work on tile s
// (A fragments read from one pair, operand fragments of slab s straight from global memory -- 6 KB per slab, shared by
// all waves through L1 / L2), the records of tile s + 1 are written into the other pair and those of tile s + 2 are on
// their way from memory; a pair is cleared as soon as its fragments have been read.  The slab loop is unrolled twice so
// that the two register sets (records, operand fragments) alternate without copies.
constexpr int kTileBytes = 2 * kTRows * kTStride;      // hi + lo tile of a wave (10 KB)
template <int NT>
struct RowsState {
    uint2 rec[kTPre];
    h8 bh[NT], bl[NT];
};
template <int NT>
__global__ void __launch_bounds__(64 * kTWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) k_tile_rows(const int32_t* __restrict__ blk, const uint2* __restrict__ recs, int64_t M, int nslabs, int L,
                                                            const h8* __restrict__ frag, const double* __restrict__ inv,
                                                            const double* __restrict__ tvec, double* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char tile_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* t0 = tile_lds + (size_t)wave * (2 * kTileBytes);
    const int64_t b = (int64_t)blockIdx.x * kTWaves + wave;
    if (b * kTRows >= M) return;
    tile_clear(t0, 2 * kTileBytes, lane);
    double acc[4][NT][4];
    f4t d[4][NT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            d[t][n] = f4t{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][n][i] = 0.0;
        }
    const int frag_row = (lane & 15) * kTStride + 16 * (lane >> 4);
    const int32_t* myblk = blk + b * (nslabs + 1);
    const uint2* myrec = recs + lane;
    const h8* myfrag = frag + lane;
    // (record reads run a little past a short tile: inside the records of the row block or the pad behind the buffer)
    auto fetch = [&](RowsState<NT>& st, int32_t p, int slab) {
#pragma unroll
        for (int u = 0; u < kTPre; ++u) st.rec[u] = myrec[p + u * 64];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            st.bh[n] = myfrag[((slab * kTNt + n) * 2 + 0) * 64];
            st.bl[n] = myfrag[((slab * kTNt + n) * 2 + 1) * 64];
        }
    };
    auto fill = [&](unsigned char* tile, const RowsState<NT>& st, int32_t p0, int32_t p1) {
        const int32_t n = p1 - p0 - lane;            // this lane's record u exists if u * 64 < n
#pragma unroll
        for (int u = 0; u < kTPre; ++u)
            if (u * 64 < n) {
                const uint32_t off = st.rec[u].x & 0xffffu;
                *reinterpret_cast<uint16_t*>(tile + off) = (uint16_t)(st.rec[u].y & 0xffffu);
                *reinterpret_cast<uint16_t*>(tile + kTRows * kTStride + off) = (uint16_t)(st.rec[u].y >> 16);
            }
        for (int32_t e = p0 + kTPre * 64 + lane; e < p1; e += 64) {      // (dense tiles)
            const uint2 r = recs[e];
            const uint32_t off = r.x & 0xffffu;
            *reinterpret_cast<uint16_t*>(tile + off) = (uint16_t)(r.y & 0xffffu);
            *reinterpret_cast<uint16_t*>(tile + kTRows * kTStride + off) = (uint16_t)(r.y >> 16);
        }
    };
    auto multiply = [&](const unsigned char* tile, const RowsState<NT>& st) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const h8 ah = *reinterpret_cast<const h8*>(tile + t * 16 * kTStride + frag_row);
            const h8 al = *reinterpret_cast<const h8*>(tile + kTRows * kTStride + t * 16 * kTStride + frag_row);
#pragma unroll
            for (int n = 0; n < NT; ++n) d[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, st.bh[n], d[t][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NT; ++n) d[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, st.bl[n], d[t][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NT; ++n) d[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, st.bh[n], d[t][n], 0, 0, 0);
        }
    };
    auto flush = [&]() {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][n][i] += (double)d[t][n][i];
                d[t][n] = f4t{0, 0, 0, 0};
            }
    };
    // one turn: tile s is in `cur` with its state in A; B holds the records / fragments of tile s + 1 (already fetched)
    auto turn = [&](int s, unsigned char* cur, unsigned char* nxt, RowsState<NT>& A, RowsState<NT>& B, int32_t& pa, int32_t& pb) {
        // tile s + 1 into the other pair, then its successor's records and fragments requested into A's record set...
        if (s + 1 < nslabs) fill(nxt, B, pa, pb);
        multiply(cur, A);                            // (A's fragments; A's records are dead since tile s was filled)
        const int s2 = s + 2 < nslabs ? s + 2 : nslabs - 1;
        pa = pb;
        pb = myblk[s2 + 1];
        fetch(A, pa, s2);                            // ...after the MFMAs have been issued with A's fragments
        tile_clear(cur, kTileBytes, lane);           // (LDS operations of a wave execute in order: the reads above come first)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((s % kTFlush) == kTFlush - 1 || s == nslabs - 1) flush();
    };
    RowsState<NT> A, B;
    int32_t pa = myblk[0], pb = myblk[1];
    fetch(A, pa, 0);
    fill(t0, A, pa, pb);                             // tile 0
    pa = pb;
    pb = myblk[nslabs > 1 ? 2 : 1];
    fetch(B, pa, nslabs > 1 ? 1 : 0);                // tile 1
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int s = 0; s < nslabs; s += 2) {
        turn(s, t0, t0 + kTileBytes, A, B, pa, pb);
        if (s + 1 < nslabs) turn(s + 1, t0 + kTileBytes, t0, B, A, pa, pb);
    }
    // D layout: lane l holds rows 4*(l >> 4) + 0..3 and column l & 15 of every 16 x 16 tile
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = 16 * n + (lane & 15);
            if (col >= L) continue;
            const double sc = inv[col], tv = tvec[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t r = b * kTRows + 16 * t + 4 * (lane >> 4) + i;
                if (r < M) out[r * L + col] = acc[t][n][i] * sc - tv;
            }
        }
}

// ---- A Q, many short waves: one wave per (64 rows, range of slabs); float32 partial sums, added in float64 afterwards ----
constexpr int kTSlabsPerPart = 40;     // 1 280 columns per partial sum
template <int NT>
#ifndef DDX_TT_DBG
#define DDX_TT_DBG 0
#endif
__global__ void __launch_bounds__(64 * kTWaves) __attribute__((amdgpu_waves_per_eu(4, 4))) k_tile_rows_part(const int32_t* __restrict__ blk, const uint2* __restrict__ recs, int64_t M, int nslabs,
                                                            const h8* __restrict__ frag, float* __restrict__ partial /* [parts][Mpad][16 * NT] */, int64_t Mpad) {
    extern __shared__ __align__(16) unsigned char tile_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* tile = tile_lds + (size_t)wave * kTileBytes;
    const int64_t b = (int64_t)blockIdx.x * kTWaves + wave;
    const int part = blockIdx.y;
    if (b * kTRows >= M) return;
    tile_clear(tile, kTileBytes, lane);
    f4t d[4][NT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) d[t][n] = f4t{0, 0, 0, 0};
    const int frag_row = (lane & 15) * kTStride + 16 * (lane >> 4);
    const int32_t* myblk = blk + b * (nslabs + 1);
    const uint2* myrec = recs + lane;
    const h8* myfrag = frag + lane;
    const int s0 = part * kTSlabsPerPart;
    const int s1 = s0 + kTSlabsPerPart < nslabs ? s0 + kTSlabsPerPart : nslabs;
    int32_t p0 = myblk[s0];
    for (int s = s0; s < s1; ++s) {
        const int32_t p1 = myblk[s + 1];
        h8 bh[NT], bl[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            bh[n] = myfrag[(((DDX_TT_DBG & 16) ? 0 : s * kTNt + n) * 2 + 0) * 64];
            bl[n] = myfrag[(((DDX_TT_DBG & 16) ? 0 : s * kTNt + n) * 2 + 1) * 64];
        }
        uint2 rec[kTPre];
#pragma unroll
        for (int u = 0; u < kTPre; ++u) rec[u] = myrec[p0 + u * 64];
        const int32_t n_mine = p1 - p0 - lane;
#pragma unroll
        for (int u = 0; u < kTPre; ++u)
            if (u * 64 < n_mine && !(DDX_TT_DBG & 2)) {
                const uint32_t off = rec[u].x & 0xffffu;
                *reinterpret_cast<uint16_t*>(tile + off) = (uint16_t)(rec[u].y & 0xffffu);
                *reinterpret_cast<uint16_t*>(tile + kTRows * kTStride + off) = (uint16_t)(rec[u].y >> 16);
            }
        for (int32_t e = p0 + kTPre * 64 + lane; e < p1; e += 64) {
            const uint2 r = recs[e];
            const uint32_t off = r.x & 0xffffu;
            *reinterpret_cast<uint16_t*>(tile + off) = (uint16_t)(r.y & 0xffffu);
            *reinterpret_cast<uint16_t*>(tile + kTRows * kTStride + off) = (uint16_t)(r.y >> 16);
        }
        p0 = p1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const h8 ah = (DDX_TT_DBG & 8) ? bh[0] : *reinterpret_cast<const h8*>(tile + t * 16 * kTStride + frag_row);
            const h8 al = (DDX_TT_DBG & 8) ? bl[0] : *reinterpret_cast<const h8*>(tile + kTRows * kTStride + t * 16 * kTStride + frag_row);
            if (DDX_TT_DBG & 4) { d[t][0][0] += (float)ah[0] + (float)al[0] + (float)bh[0][0] + (float)bl[1][0] + (float)bh[NT - 1][0] + (float)bl[NT - 1][0]; continue; }
#pragma unroll
            for (int n = 0; n < NT; ++n) d[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[n], d[t][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NT; ++n) d[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[n], d[t][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NT; ++n) d[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[n], d[t][n], 0, 0, 0);
        }
        if (!(DDX_TT_DBG & 1)) tile_clear(tile, kTileBytes, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    float* my = partial + ((int64_t)part * Mpad + b * kTRows) * (16 * NT);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) my[(16 * t + 4 * (lane >> 4) + i) * (16 * NT) + 16 * n + (lane & 15)] = d[t][n][i];
}

// out[r][c] = (sum over parts, in order, of partial[part][r][c]) * inv[c] - tvec[c]
__global__ void __launch_bounds__(256) k_tile_rows_sum(const float* __restrict__ partial, int parts, int64_t M, int64_t Mpad, int L, int ldp,
                                                       const double* __restrict__ inv, const double* __restrict__ tvec, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * ldp) return;
    const int64_t r = t / ldp;
    const int c = (int)(t - r * ldp);
    if (c >= L) return;
    double a = 0.0;
    for (int p = 0; p < parts; ++p) a += (double)partial[((int64_t)p * Mpad + r) * ldp + c];
    out[r * L + c] = a * inv[c] - tvec[c];
}

// ---- A^T Y: one wave per column slab (32 columns = 2 MFMA row tiles of A^T) and range of row blocks ------------------
// The same records, written transposed: tile row = column inside the slab, contraction index = row inside the block
// (64 = two MFMA steps).  Partial results per range of row blocks; k_sum_panels adds them in order.
constexpr int kTileBytesT = 2 * kTK * kTStrideT;       // hi + lo transposed tile of a wave (9 KB)
template <int NT>
__global__ void __launch_bounds__(64 * kTWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) k_tile_cols(const int32_t* __restrict__ blk, const uint2* __restrict__ recs, int64_t nblocks, int nslabs,
                                                            int32_t H, int L, int64_t blocks_per_part, const h8* __restrict__ frag,
                                                            const double* __restrict__ inv, double* __restrict__ partial) {
    extern __shared__ __align__(16) unsigned char tile_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* t0 = tile_lds + (size_t)wave * (2 * kTileBytesT);
    const int s = (int)(blockIdx.x * kTWaves + wave);
    const int64_t part = blockIdx.y;
    if (s >= nslabs) return;
    tile_clear(t0, 2 * kTileBytesT, lane);
    double acc[2][NT][4];
    f4t d[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            d[t][n] = f4t{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][n][i] = 0.0;
        }
    const int frag_row = (lane & 15) * kTStrideT + 16 * (lane >> 4);
    const int64_t b0 = part * blocks_per_part;
    const int64_t b1 = b0 + blocks_per_part < nblocks ? b0 + blocks_per_part : nblocks;
    if (b0 >= b1) return;
    auto put = [&](unsigned char* tile, const uint2 r) {
        const uint32_t off = r.x >> 16;
        *reinterpret_cast<uint16_t*>(tile + off) = (uint16_t)(r.y & 0xffffu);
        *reinterpret_cast<uint16_t*>(tile + kTK * kTStrideT + off) = (uint16_t)(r.y >> 16);
    };
    auto fill = [&](unsigned char* tile, const uint2 (&rec)[kTPre], int32_t p0, int32_t p1) {
#pragma unroll
        for (int u = 0; u < kTPre; ++u)
            if (p0 + u * 64 + lane < p1) put(tile, rec[u]);
        for (int32_t e = p0 + kTPre * 64; e < p1; e += 64)
            if (e + lane < p1) put(tile, recs[e + lane]);
    };
    auto bounds = [&](int64_t b, int32_t& p0, int32_t& p1) {
        const int64_t bb = b < b1 ? b : b1 - 1;
        p0 = blk[bb * (nslabs + 1) + s];
        p1 = blk[bb * (nslabs + 1) + s + 1];
    };
    uint2 rec[kTPre];
    int32_t pa, pb;
    bounds(b0, pa, pb);
#pragma unroll
    for (int u = 0; u < kTPre; ++u) rec[u] = recs[pa + u * 64 + lane];
    fill(t0, rec, pa, pb);
    bounds(b0 + 1, pa, pb);
#pragma unroll
    for (int u = 0; u < kTPre; ++u) rec[u] = recs[pa + u * 64 + lane];
    h8 nh[2][NT], nl[2][NT];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            nh[kk][n] = frag[(((2 * b0 + kk) * kTNt + n) * 2 + 0) * 64 + lane];
            nl[kk][n] = frag[(((2 * b0 + kk) * kTNt + n) * 2 + 1) * 64 + lane];
        }
    for (int64_t b = b0; b < b1; ++b) {
        unsigned char* cur = t0 + ((b - b0) & 1) * kTileBytesT;
        unsigned char* nxt = t0 + ((b - b0 + 1) & 1) * kTileBytesT;
        h8 bh[2][NT], bl[2][NT];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int n = 0; n < NT; ++n) { bh[kk][n] = nh[kk][n]; bl[kk][n] = nl[kk][n]; }
        if (b + 1 < b1) fill(nxt, rec, pa, pb);
        bounds(b + 2, pa, pb);
#pragma unroll
        for (int u = 0; u < kTPre; ++u) rec[u] = recs[pa + u * 64 + lane];
        const int64_t bn = b + 1 < b1 ? b + 1 : b;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                nh[kk][n] = frag[(((2 * bn + kk) * kTNt + n) * 2 + 0) * 64 + lane];
                nl[kk][n] = frag[(((2 * bn + kk) * kTNt + n) * 2 + 1) * 64 + lane];
            }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const h8 ah = *reinterpret_cast<const h8*>(cur + t * 16 * kTStrideT + 64 * kk + frag_row);
                const h8 al = *reinterpret_cast<const h8*>(cur + kTK * kTStrideT + t * 16 * kTStrideT + 64 * kk + frag_row);
#pragma unroll
                for (int n = 0; n < NT; ++n) d[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[kk][n], d[t][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) d[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[kk][n], d[t][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) d[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[kk][n], d[t][n], 0, 0, 0);
            }
        tile_clear(cur, kTileBytesT, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (((b - b0) & 1) == 1 || b == b1 - 1) {           // every 128 rows
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][n][i] += (double)d[t][n][i];
                    d[t][n] = f4t{0, 0, 0, 0};
                }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = 16 * n + (lane & 15);
            if (col >= L) continue;
            const double sc = inv[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int32_t j = s * kTK + 16 * t + 4 * (lane >> 4) + i;
                if (j < H) partial[((int64_t)part * H + j) * L + col] = acc[t][n][i] * sc;
            }
        }
}

// ---- host side ----------------------------------------------------------------------------------------------------
int tiles_prepare_operand(ddx_ctx* ctx, const double* X, int64_t R, int L) {
    if (L > kTNt * 16) return set_err(ctx, DDX_E_UNSUPPORTED, "sketch too wide for the matrix-core products");
    const int64_t nslabs = 2 * ceil_div(R, (int64_t)kTRows);      // whole 64-row blocks: A^T Y reads two slabs per block
    DDX_TRY(ensure(ctx, ctx->tile_frag, sizeof(h8) * (size_t)nslabs * kTNt * 2 * 64));
    DDX_TRY(ensure(ctx, ctx->tile_scale, 4096));
    unsigned char* sb = ctx->tile_scale.as<unsigned char>();
    uint32_t* cmax = reinterpret_cast<uint32_t*>(sb);               // [64]
    float* scale = reinterpret_cast<float*>(sb + 256);              // [64]
    double* inv = reinterpret_cast<double*>(sb + 512);              // [64]
    DDX_HIP(ctx, hipMemsetAsync(cmax, 0, 256, ctx->stream));
    int nb = (int)std::min<int64_t>(256, ceil_div(R, (int64_t)64));
    const int64_t rpb = ceil_div(R, (int64_t)nb);
    nb = (int)ceil_div(R, rpb);
    k_tile_colmax<<<nb, 256, 0, ctx->stream>>>(X, R, L, rpb, cmax);
    k_tile_scales<<<1, 64, 0, ctx->stream>>>(cmax, L, scale, inv);
    k_tile_operand<<<(unsigned)ceil_div(nslabs * kTNt * 64, (int64_t)256), 256, 0, ctx->stream>>>(X, R, L, scale, nslabs, ctx->tile_frag.as<h8>());
    return DDX_OK;
}

// seg[row * (ng + 1) + g] = offset inside the row of its first entry with column >= g * span (g = ng: the row length)
__global__ void k_tile_segments(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols, int64_t nrows, int ng, int span,
                                int32_t* __restrict__ seg) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrows * (ng + 1)) return;
    const int64_t row = t / (ng + 1);
    const int g = (int)(t - row * (ng + 1));
    const int64_t b = indptr[row];
    int32_t lo = 0, hi = (int32_t)(indptr[row + 1] - b);
    const int32_t want = g * span;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (cols[b + mid] < want) lo = mid + 1; else hi = mid;
    }
    seg[t] = lo;
}

// the tile stream of the current (log-normalised, maybe scaled) matrix; once per PCA
int tiles_pack(ddx_ctx* ctx) {
    const int64_t M = ctx->M;
    const int32_t H = ctx->H;
    const int nslabs = (int)ceil_div((int64_t)H, (int64_t)kTK);
    const int ng = (int)ceil_div((int64_t)nslabs, (int64_t)kPackSlabs);
    const int64_t nblocks = ceil_div(M, (int64_t)kTRows);
    int64_t nnz_aug = 0;
    DDX_HIP(ctx, hipMemcpyAsync(&nnz_aug, ctx->aug_indptr.as<int64_t>() + M, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DDX_TRY(ensure(ctx, ctx->tile_seg, sizeof(int32_t) * (size_t)M * (ng + 1)));
    DDX_TRY(ensure(ctx, ctx->tile_blk, sizeof(int32_t) * (size_t)nblocks * (nslabs + 1)));
    DDX_TRY(ensure(ctx, ctx->tile_recs, sizeof(uint2) * (size_t)(nnz_aug + 64)));
    DDX_TRY(ensure(ctx, ctx->tile_scale, 4096));
    unsigned char* sb = ctx->tile_scale.as<unsigned char>();
    if (!ctx->tile_flag_clean) { DDX_HIP(ctx, hipMemsetAsync(sb + 1024, 0, 64, ctx->stream)); ctx->tile_flag_clean = true; }
    ScopedTimer t(ctx, "tile_pack");
    k_tile_segments<<<(unsigned)ceil_div(M * (ng + 1), (int64_t)256), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_indices.as<int32_t>(), M, ng,
                                                                                        kPackSlabs * kTK, ctx->tile_seg.as<int32_t>());
    k_tile_pack<<<dim3((unsigned)ng, (unsigned)nblocks), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_indices.as<int32_t>(), ctx->aug_x.as<float>(),
                                                                              ctx->zcol.as<float>(), ctx->tile_seg.as<int32_t>(), ng, M, H, nslabs,
                                                                              ctx->tile_blk.as<int32_t>(), ctx->tile_recs.as<uint2>(), reinterpret_cast<int*>(sb + 1024));
    DDX_HIP(ctx, hipGetLastError());
    return DDX_OK;
}

int tiles_apply_rows(ddx_ctx* ctx, int L, const double* tvec, double* Yrow) {
    const int nslabs = (int)ceil_div((int64_t)ctx->H, (int64_t)kTK);
    const int64_t nblocks = ceil_div(ctx->M, (int64_t)kTRows);
    const double* inv = reinterpret_cast<const double*>(ctx->tile_scale.as<unsigned char>() + 512);
    const int parts = (int)ceil_div((int64_t)nslabs, (int64_t)kTSlabsPerPart);
    const int nt = L <= 32 ? 2 : 3;
    const int64_t Mpad = nblocks * kTRows;
    DDX_TRY(ensure(ctx, ctx->tile_part, sizeof(float) * (size_t)parts * Mpad * 16 * nt));
    const size_t lds = (size_t)kTWaves * kTileBytes;
    const dim3 grid((unsigned)ceil_div(nblocks, (int64_t)kTWaves), (unsigned)parts);
    if (nt == 2)
        k_tile_rows_part<2><<<grid, 64 * kTWaves, lds, ctx->stream>>>(ctx->tile_blk.as<int32_t>(), ctx->tile_recs.as<uint2>(), ctx->M, nslabs,
                                                                      ctx->tile_frag.as<h8>(), ctx->tile_part.as<float>(), Mpad);
    else
        k_tile_rows_part<3><<<grid, 64 * kTWaves, lds, ctx->stream>>>(ctx->tile_blk.as<int32_t>(), ctx->tile_recs.as<uint2>(), ctx->M, nslabs,
                                                                      ctx->tile_frag.as<h8>(), ctx->tile_part.as<float>(), Mpad);
    k_tile_rows_sum<<<(unsigned)ceil_div(ctx->M * 16 * nt, (int64_t)256), 256, 0, ctx->stream>>>(ctx->tile_part.as<float>(), parts, ctx->M, Mpad, L, 16 * nt, inv, tvec, Yrow);
    return DDX_OK;
}

// partial[parts x H x L]; returns the number of parts
int tiles_apply_cols(ddx_ctx* ctx, int L, double* partial, int max_parts, int* parts_out) {
    const size_t lds = (size_t)kTWaves * 2 * kTileBytesT;
    const int nslabs = (int)ceil_div((int64_t)ctx->H, (int64_t)kTK);
    const int64_t nblocks = ceil_div(ctx->M, (int64_t)kTRows);
    // enough (slab, part) tasks for two rounds of the 2 048 resident waves
    int parts = (int)ceil_div((int64_t)4096, (int64_t)nslabs);
    if (parts > max_parts) parts = max_parts;
    if (parts > nblocks) parts = (int)nblocks;
    int64_t bpp = ceil_div(nblocks, (int64_t)parts);
    bpp = (bpp + 1) & ~(int64_t)1;                              // whole pairs of row blocks between float64 additions
    parts = (int)ceil_div(nblocks, bpp);
    const double* inv = reinterpret_cast<const double*>(ctx->tile_scale.as<unsigned char>() + 512);
    const dim3 grid((unsigned)ceil_div((int64_t)nslabs, (int64_t)kTWaves), (unsigned)parts);
    DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_tile_cols<2>), (int)lds));
    DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_tile_cols<3>), (int)lds));
    if (L <= 32)
        k_tile_cols<2><<<grid, 64 * kTWaves, lds, ctx->stream>>>(ctx->tile_blk.as<int32_t>(), ctx->tile_recs.as<uint2>(), nblocks, nslabs, ctx->H, L, bpp,
                                                                 ctx->tile_frag.as<h8>(), inv, partial);
    else
        k_tile_cols<3><<<grid, 64 * kTWaves, lds, ctx->stream>>>(ctx->tile_blk.as<int32_t>(), ctx->tile_recs.as<uint2>(), nblocks, nslabs, ctx->H, L, bpp,
                                                                 ctx->tile_frag.as<h8>(), inv, partial);
    *parts_out = parts;
    return DDX_OK;
}

// a value outside the half-precision range was met by a product since the last check (the host falls back / reports)
int tiles_check(ddx_ctx* ctx) {
    if (!ctx->tile_scale.p || !ctx->tile_flag_clean) return DDX_OK;
    int flag = 0;
    DDX_HIP(ctx, hipMemcpyAsync(&flag, ctx->tile_scale.as<unsigned char>() + 1024, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tile_flag_clean = false;
    if (flag) return set_err(ctx, DDX_E_UNSUPPORTED, "matrix value outside the range of the matrix-core products (|x - z| >= 58); run with DDX_SPMM=lds");
    return DDX_OK;
}

}  // namespace ddx
