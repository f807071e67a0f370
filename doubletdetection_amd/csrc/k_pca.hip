// Truncated PCA of the augmented matrix = scikit-learn's seeded randomized SVD (what
// sc.tl.pca(svd_solver="auto") runs for the reference, dd.py:305-314), evaluated in float64 on the
// *implicit* matrix:  X = 1 z^T + L  with L stored sparse (CSR + column-major mirror), so that the
// centred operator is  A = L - 1 m^T  (m = column means of L).  The dense M x H matrix of dd.py:295
// is never formed: every product A Q / A^T Y is one pass over the stored entries (8 bytes each); the L-wide rows
// of the small operand are fetched from a float32 copy of it staged slice by slice in LDS (k_spmm_lds, the
// default), or gathered from L2 by the first-generation kernels (k_spmm_rows / k_spmm_cols: float64 mode,
// ddx_operator_apply, sketch widths > 42, DDX_SPMM=gather).  Sums are float64 throughout; products are float64 in the
// gather kernels and, in the LDS kernels, float32 within a trip of eight entries (DDX_SPMM_TRIP=f64: float64).
//
// Steps (sklearn/utils/extmath.py:287-372,531-607; sklearn/decomposition/_pca.py:731-766):
//   Q0 (host-drawn, seeded) -> n_iter x { Q <- orth(A Q) ; Q <- orth(A^T Q) } -> Q <- qr(A Q)
//   B = Q^T A ; SVD(B) via eigh(B B^T) ; U = Q Uhat ; sign fix on the component rows ; return U S.
// orth() is a Cholesky-QR instead of sklearn's LU, applied once per power iteration (A^T A Q) instead of after each
// half step: only the spanned subspace enters the next step, so the float64 result is the same to ~1e-12
// (oracle/dd_oracle.py:randomized_pca_f64 checks this); the final basis gets two passes (CholQR2).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>

#include "ddx_prims.h"

#include "ddx_internal.h"

namespace ddx {

constexpr int kMaxL = 64;   // sketch width limit (n_components + n_oversamples)

constexpr int kGatherDepth = 4;   // gathers in flight per lane and trip

typedef double d2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

// One gathered operand row is read as 16-byte pieces: 2 float64 or 4 float32 sketch columns per lane.
// The float32 variant reads a rounded copy of the operand (products and sums stay float64); it halves
// the gathered bytes, which is what bounds these kernels.  VW = sketch columns per lane.
template <typename T> struct Gather;
template <> struct Gather<double> {
    static constexpr int VW = 2;
    __device__ static __forceinline__ void load(const double* p, double (&v)[2]) {
        const d2v t = *reinterpret_cast<const d2v*>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <> struct Gather<float> {
    static constexpr int VW = 4;
    __device__ static __forceinline__ void load(const float* p, double (&v)[4]) {
        const f4v t = *reinterpret_cast<const f4v*>(p);
        v[0] = (double)t.x; v[1] = (double)t.y; v[2] = (double)t.z; v[3] = (double)t.w;
    }
};

// ------------------------------------------------------------------------------------------------
// SpMM over rows:  Y[i,:] = sum_j (x_ij - z_j) Q[j,:] - t[:]        (A Q, t = m^T Q)
// One wave per matrix row.  The 64 lanes are split into `slots` groups of `lpn` lanes; a group handles
// one stored entry at a time and each of its lanes VW adjacent sketch columns (one 16-byte load of Q),
// so a step consumes `slots` entries with fully coalesced row gathers.  Four steps are in flight per
// trip: all gathers are issued before the first multiply-add; the accumulation order (entry by entry)
// is fixed, so results are reproducible bit for bit.   Requires L % VW == 0.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_spmm_rows(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                   const float* __restrict__ x, const float* __restrict__ zcol,
                                                   const T* __restrict__ Q, int ld, int L, int lpn, int slots,
                                                   const double* __restrict__ tvec, int64_t M, double* __restrict__ Y, int accumulate = 0) {
    constexpr int VW = Gather<T>::VW;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int slot = lane / lpn, sub = lane - slot * lpn;
    const bool active = slot < slots;
    const int c0 = VW * sub;
    const int64_t b = indptr[row], e = indptr[row + 1];
    double acc[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) acc[c] = 0.0;
    for (int64_t base = b; base < e; base += 64) {
        const int64_t p = base + lane;
        int32_t j = 0;
        double d = 0.0;
        if (p < e) {
            j = cols[p];
            d = (double)x[p] - (double)zcol[j];
        }
        const int cnt = (int)((e - base) < 64 ? (e - base) : 64);
        for (int t0 = 0; t0 < cnt; t0 += kGatherDepth * slots) {
            int32_t jj[kGatherDepth];
            double dd[kGatherDepth];
#pragma unroll
            for (int u = 0; u < kGatherDepth; ++u) {
                const int tt = t0 + u * slots + slot;
                const int src = tt < 64 ? tt : 63;
                jj[u] = __shfl(j, src, 64);
                dd[u] = __shfl(d, src, 64);
                if (tt >= cnt) { dd[u] = 0.0; jj[u] = 0; }
            }
            if (active) {
                double qv[kGatherDepth][VW];
#pragma unroll
                for (int u = 0; u < kGatherDepth; ++u) Gather<T>::load(Q + (int64_t)jj[u] * ld + c0, qv[u]);
#pragma unroll
                for (int u = 0; u < kGatherDepth; ++u)
#pragma unroll
                    for (int c = 0; c < VW; ++c) acc[c] = fma(dd[u], qv[u][c], acc[c]);
            }
        }
    }
    // combine the slots in fixed order (slot 0 + slot 1 + ...)
#pragma unroll
    for (int c = 0; c < VW; ++c) {
        double sum = acc[c];
        for (int s = 1; s < slots; ++s) sum += __shfl(acc[c], lane + s * lpn, 64);
        if (slot == 0 && c0 + c < L) Y[row * L + c0 + c] = sum - tvec[c0 + c] + (accumulate ? Y[row * L + c0 + c] : 0.0);
    }
}

// ------------------------------------------------------------------------------------------------
// SpMM over columns:  W[j,:] = sum_i (x_ij - z_j) Y[i,:] - m_j u[:]     (A^T Y, u = 1^T Y)
// The mirror is ordered by (row panel, column).  One wave owns one (panel, column) segment and
// accumulates  Wp[panel][j][:] = sum_{i in panel} (x_ij - z_j) Y[i,:] ; a second tiny kernel adds the
// panels in order.  Only the rows of one panel (panel_rows x L x 4 or 8 B ~ 2.6 MB) are gathered while a
// panel is processed, and all blocks of a panel are placed on the same XCD (block b runs on XCD b % 8),
// so those gathers are served by that XCD's 4 MB L2 instead of the Infinity Cache.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_spmm_cols(const int64_t* __restrict__ cp_o, const int32_t* __restrict__ row_o,
                                                   const float* __restrict__ x_o, int P_o,
                                                   const int64_t* __restrict__ cp_s, const int32_t* __restrict__ row_s,
                                                   const float* __restrict__ x_s, int p_s0, int P_s, int P, int32_t H,
                                                   const float* __restrict__ zcol, const T* __restrict__ Yin, int ld,
                                                   int L, int lpn, int slots, double* __restrict__ Wp) {
    constexpr int VW = Gather<T>::VW;
    // XCD-aware decode: xcd = b % 8 handles panels xcd, xcd + 8, ... one after the other
    const int groups = (H + 3) >> 2;                 // 4 columns (one per wave) per block
    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t t = b >> 3;
    const int panel = xcd + 8 * (int)(t / groups);
    if (panel >= P) return;
    const int lane = threadIdx.x & 63;
    const int j = (int)(t % groups) * 4 + (threadIdx.x >> 6);
    if (j >= H) return;
    const int slot = lane / lpn, sub = lane - slot * lpn;
    const bool active = slot < slots;
    const int c0 = VW * sub;
    const double z = (double)zcol[j];
    double acc[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) acc[c] = 0.0;
    for (int seg = 0; seg < 2; ++seg) {
        int64_t lo, hi;
        const int32_t* rows;
        const float* x;
        if (seg == 0) {
            if (panel >= P_o) continue;
            lo = cp_o[(int64_t)panel * H + j]; hi = cp_o[(int64_t)panel * H + j + 1];
            rows = row_o; x = x_o;
        } else {
            const int ps = panel - p_s0;
            if (ps < 0 || ps >= P_s) continue;
            lo = cp_s[(int64_t)ps * H + j]; hi = cp_s[(int64_t)ps * H + j + 1];
            rows = row_s; x = x_s;
        }
        for (int64_t base = lo; base < hi; base += 64) {
            const int64_t p = base + lane;
            int32_t i = 0;
            double d = 0.0;
            if (p < hi) {
                i = rows[p];
                d = (double)x[p] - z;
            }
            const int cnt = (int)((hi - base) < 64 ? (hi - base) : 64);
            for (int t0 = 0; t0 < cnt; t0 += kGatherDepth * slots) {
                int32_t ii[kGatherDepth];
                double dd[kGatherDepth];
#pragma unroll
                for (int u = 0; u < kGatherDepth; ++u) {
                    const int tt = t0 + u * slots + slot;
                    const int src = tt < 64 ? tt : 63;
                    ii[u] = __shfl(i, src, 64);
                    dd[u] = __shfl(d, src, 64);
                    if (tt >= cnt) { dd[u] = 0.0; ii[u] = 0; }
                }
                if (active) {
                    double yv[kGatherDepth][VW];
#pragma unroll
                    for (int u = 0; u < kGatherDepth; ++u) Gather<T>::load(Yin + (int64_t)ii[u] * ld + c0, yv[u]);
#pragma unroll
                    for (int u = 0; u < kGatherDepth; ++u)
#pragma unroll
                        for (int c = 0; c < VW; ++c) acc[c] = fma(dd[u], yv[u][c], acc[c]);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < VW; ++c) {
        double sum = acc[c];
        for (int s = 1; s < slots; ++s) sum += __shfl(acc[c], lane + s * lpn, 64);
        if (slot == 0 && c0 + c < L) Wp[((int64_t)panel * H + j) * L + c0 + c] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged operator products (the default).  The gather kernels above fetch one 160-byte operand row per
// stored entry through L1/L2 (19.5 GB per launch at the headline workload, 12 TB/s out of L2) -- that, not
// HBM, bounds them.  Here a *slice* of the float32 operand (SR consecutive rows) is staged in LDS by one
// 1024-thread workgroup per CU and every per-entry fetch is an LDS read.
//
//   ROWS (A Q):   operand = Q (H rows), outputs = rows of the augmented matrix.  A workgroup owns a set of rows
//                 and walks all slices of Q; the entries of row i whose column falls in slice s are the
//                 contiguous sub-range rowseg[i][s..s+1] of the CSR row.
//   COLS (A^T Y): operand = Y (M rows), outputs = columns.  The column-major mirror is ordered by
//                 (row panel of SR rows, column), so the entries of column j inside slice s are contiguous.  A
//                 workgroup owns a set of columns and a strided subset of the slices (slice group g takes slices
//                 g, g + groups, ...), and writes one partial H x L block per group; k_sum_panels adds the groups.
//
// Ownership.  The 64 lanes of a wave form SLOTS groups of `lpn` lanes, a lane holding two adjacent sketch
// columns.  Each *group* owns kLdsOwnG outputs and keeps their float64 accumulators in registers across all
// slices (no replication across groups, no final reduction), so a wave advances SLOTS segments -- one per group
// -- in lock step, one stored entry per group and step.  Outputs are strided across workgroups
// (output = owner + owners * local), so every workgroup holds the same mix of original and (twice as long)
// synthetic rows.
//
// Pipeline.  The stored entries of the SLOTS current segments are fetched 64 per segment and round, one round
// ahead (global-load latency hides behind the previous round's arithmetic), converted once to
// (LDS byte address of the operand row, value x - z) in the wave's staging area -- padded with zeros to
// a full round so the step loop needs no bounds logic -- and then consumed a trip of eight steps at a time: the
// eight operand reads are issued together, the eight products of a lane's two columns are formed and summed in
// float32 with v_pk_fma_f32 (two chains of four), and the trip's sum is added to the float64 accumulator.
// All orders are fixed.
// ------------------------------------------------------------------------------------------------
#ifndef DDX_SPMM_OFF16
#define DDX_SPMM_OFF16 1    // packed-float32 trips stage 16-bit operand row indices (0: 32-bit byte offsets, the first version)
#endif
#ifndef DDX_SPMM_DBG
#define DDX_SPMM_DBG 0      // ablation builds only (profiles/tools/spmm_ablation.sh): 1 no operand reads, 2 no entry fetches, 4 no staged reads, 8 no slice staging, 16 no trips, 32 no entry staging, 64 conflict-free operand rows
#endif
#ifndef DDX_LDS_OWN
#define DDX_LDS_OWN 6
#endif
#ifndef DDX_SPMM_NT
#define DDX_SPMM_NT 0       // 1: stored entries are fetched with the non-temporal hint (they are read once; the operand slices should own the L2)
#endif
#ifndef DDX_SPMM_HALFTRIP
#define DDX_SPMM_HALFTRIP 1 // a round whose last trip would hold at most four steps ends with a half trip of four
#endif
#ifndef DDX_SPMM_ROUNDSUM
#define DDX_SPMM_ROUNDSUM 0 // 1: trip sums are added in float32 over a round (<= 64 entries) and enter the float64 accumulator once per round
#endif
constexpr int kLdsOwnG = DDX_LDS_OWN;        // outputs owned by one lane group
#ifndef DDX_LDS_WAVES
#define DDX_LDS_WAVES 16
#endif
constexpr int kLdsWaves = DDX_LDS_WAVES;      // waves per workgroup (16: one workgroup per CU; 8: two, each with half the LDS)
constexpr int kLdsWgPerCu = 16 / kLdsWaves;
constexpr int kLdsThreads = kLdsWaves * 64;
constexpr int kLdsChunk = 64;      // stored entries per segment and round
constexpr int kLdsBudget = 160 * 1024 / kLdsWgPerCu;
// staging per wave and lane group: 64 float64 values + 64 LDS offsets, the group strides padded by 16 bytes so
// that the groups' broadcast reads of entry t fall into different banks
constexpr int kLdsDStride = kLdsChunk + 2;    // doubles
constexpr int kLdsOStride = kLdsChunk + 4;    // uint32
// (float32 values of the packed trips take kLdsOStride * 4 bytes per group instead of kLdsDStride * 8)
constexpr int lds_value_bytes(bool pk) { return pk ? kLdsOStride * 4 : kLdsDStride * 8; }
constexpr int lds_stage_bytes(int slots, bool pk) { return kLdsWaves * slots * (lds_value_bytes(pk) + kLdsOStride * 4); }

// The block that finishes LAST adds the blocks' partials (round 6: the separate reduction launches are gone).  Every block writes its partial
// with device-scope stores (relaxed atomics: they go through to memory, past the writing XCD's L2, without the whole-L2 write-back a release
// fence costs on this chip -- a fence per block made the Gram kernel three times slower), waits for them to be acknowledged, takes a ticket from
// `counter`; the holder of the last ticket reads all partials with device-scope loads and reduces in the FIXED order of k_reduce_partials -- one
// wave per output, lanes over interleaved blocks, butterfly -- so the result does not depend on which block came last.  It puts the counter
// back to zero for the next launch on the stream.
__device__ __forceinline__ void partial_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double partial_load(const double* p) { return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// (s_flag: one int of the block's LDS -- a kernel whose dynamic LDS already fills the CU has no room for a static variable)
__device__ __forceinline__ bool last_block_ticket(int* counter, int nblocks, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's partial stores have reached memory
    __syncthreads();                                     // ... and so have everybody's in the block
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == nblocks - 1) ? 1 : 0;
        if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0;
}

__device__ __forceinline__ bool last_block_ticket(int* counter, int nblocks) {
    __shared__ int s_last;
    return last_block_ticket(counter, nblocks, &s_last);
}

__device__ __forceinline__ void reduce_partials_block(const double* __restrict__ partial, int nblocks, int width, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int c = wave; c < width; c += nwaves) {
        double s = 0.0;
        for (int b = lane; b < nblocks; b += 64) s += partial_load(&partial[(int64_t)b * width + c]);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) out[c] = s;
    }
}

struct LdsSpmmArgs {
    const float* op;     // operand, row-major [opRows x ld] float32
    int ld, L, lpn;
    int64_t opRows;
    int SR, nslices, groups;      // slice height, number of slices, slice groups (1 for ROWS)
    int64_t nOut;                 // outputs (M for ROWS, H for COLS)
    int owners;                   // workgroups along the outputs
    const int32_t* perm;          // outputs sorted by stored entries (descending): rank -> output
    // ROWS
    const int64_t* indptr; const int32_t* cols; const float* x; const float* zcol; const int32_t* rowseg; const double* tvec;
    int accumulate;               // ROWS, bit-plane mode: `out` already holds the bit-plane part of the product (k_bitplane.hip); the sparse part is added to it
    const double* srow;           // ... and the largest |srow[i] y[i][c]| per column goes to ymax[c] (atomic maxima of the bit patterns: exact in any
    unsigned long long* ymax;     //     order): what the A^T Y product cuts its digits by
    // COLS
    const int64_t* cp_o; const int32_t* row_o; const float* x_o; int P_o;
    const int64_t* cp_s; const int32_t* row_s; const float* x_s; int p_s0, P_s;
    int z_uniform;                // ROWS: every column has the same value for unstored entries (no scaling): zval, no table
    float zval;
    double* out;                  // ROWS: Y [M x L];  COLS: partials [groups x H x L]
    float* out32;                 // ROWS: padded float32 copy of Y [M x ld] for the A^T Y pass that follows (or null)
    // ROWS (k_spmm_packed): usum != nullptr: also the column sums of Y (the rank-one correction of the A^T Y product that follows, a launch of its
    // own otherwise): one partial per workgroup into upart, added in workgroup order by the one that finishes last (ticket ucount)
    double* usum; double* upart; int* ucount;
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int64_t read_lane64(int64_t v, int j) {
    return ((int64_t)__builtin_amdgcn_readlane((int)(v >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, j);
}

template <int SLOTS> struct LdsFetch {
    int32_t i[SLOTS];
    float x[SLOTS];
};

// entries [l[g] + r*64, ...) of the SLOTS current segments, one per lane and segment.  Lanes past the end of a
// segment re-read its first entry (always a readable address), so the loads are unconditional and nothing
// consumes them before the staging step of the *next* round -- that is what keeps them in flight.
template <int SLOTS>
__device__ __forceinline__ LdsFetch<SLOTS> lds_fetch(const int32_t* const (&idx)[SLOTS], const float* const (&x)[SLOTS],
                                                     const int32_t (&l)[SLOTS], const int32_t (&h)[SLOTS], int r, int lane) {
    LdsFetch<SLOTS> f;
#pragma unroll
    for (int g = 0; g < SLOTS; ++g) {
        int32_t p = l[g] + r * kLdsChunk + lane;         // positions fit 31 bits (stage_create_doublets enforces it)
        p = p < h[g] ? p : l[g];
        f.i[g] = idx[g][p];
        f.x[g] = x[g][p];
    }
    return f;
}

// one round: stage the fetched entries, then `nsteps` lock-step steps (rounded up to a multiple of 4 <= 64)
// PK: the eight products of a trip are formed and added in float32, two sketch columns per instruction
// (v_pk_fma_f32, two chains of four), and the trip's sum is added to the float64 accumulator.
// CPL: sketch columns per lane (2: ds_read_b64 per entry, 4: ds_read_b128).
template <bool ROWS, int SLOTS, bool PK, int CPL>
__device__ __forceinline__ void lds_round(const LdsFetch<SLOTS>& f, const int (&nvalid)[SLOTS], int nsteps, int32_t base, const double (&zc)[SLOTS],
                                          const unsigned char* opB, const float* zS, bool zuni, float zval, int ld, double* dS, uint32_t* offS, int lane,
                                          const double* myd, const uint32_t* myoff, double (&acc)[CPL]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef float fq __attribute__((ext_vector_type(CPL)));
    // value of the unstored entries of each entry's column.  ROWS: one constant when the matrix is not scaled (no
    // lookup at all); otherwise the SLOTS table reads are issued together, ahead of the first use (one LDS round trip
    // for the round instead of one per lane group)
    int idx[SLOTS];
    float zf[SLOTS];
#pragma unroll
    for (int g = 0; g < SLOTS; ++g) idx[g] = lane < nvalid[g] ? f.i[g] - base : 0;
    if (ROWS && !zuni) {
#pragma unroll
        for (int g = 0; g < SLOTS; ++g) zf[g] = zS[idx[g]];
    } else {
#pragma unroll
        for (int g = 0; g < SLOTS; ++g) zf[g] = ROWS ? zval : (float)zc[g];
    }
#pragma unroll
    for (int g = 0; g < SLOTS; ++g) {
        const bool ok = lane < nvalid[g];
        const int i = idx[g];
        if (DDX_SPMM_DBG & 32) continue;
        if (PK) {
            // x - z in float32 is the correctly rounded difference, i.e. what the float64 difference rounds to
            reinterpret_cast<float*>(dS)[g * kLdsOStride + lane] = ok ? f.x[g] - zf[g] : 0.0f;
        } else {
            const double z = ROWS ? (double)zf[g] : zc[g];
            dS[g * kLdsDStride + lane] = ok ? (double)f.x[g] - z : 0.0;
        }
        if (PK && DDX_SPMM_OFF16 && !(DDX_SPMM_DBG & (1 | 4 | 64)))       // operand row index inside the slice (< 2^16); the trip multiplies it out (v_mad_u32_u16)
            reinterpret_cast<uint16_t*>(offS + g * kLdsOStride)[lane] = (uint16_t)i;
        else
            offS[g * kLdsOStride + lane] = __umul24((uint32_t)i, (uint32_t)(ld * 4));      // i < 2^24, ld * 4 < 2^24
    }
    wave_lds_sync();
    if (PK) {
        const float* myf = reinterpret_cast<const float*>(myd);
        fq rsum = (fq)(0.0f);
        // full trips of eight steps, then -- when one to four steps are left -- a half trip of four (the staged round is
        // zero-padded to 64 entries; a zero entry adds an exact +0, so where the padding ends changes no bit of the sums)
        const bool half_tail = DDX_SPMM_HALFTRIP && DDX_SPMM_OFF16 && !DDX_SPMM_DBG && !DDX_SPMM_ROUNDSUM && ((nsteps - 1) & 7) < 4;
        const int nfull = half_tail ? (nsteps & ~7) : nsteps;
        for (int t0 = 0; t0 < ((DDX_SPMM_DBG & 16) ? 0 : nfull); t0 += 8) {
            f4v fv[2];
            u4 ov[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) fv[u] = *reinterpret_cast<const f4v*>(myf + t0 + 4 * u);
            fq q[8];
            if (DDX_SPMM_OFF16 && !(DDX_SPMM_DBG & (1 | 4 | 64))) {
                // eight 16-bit row indices in one 16-byte read (half the staged-offset traffic); address = index * row
                // bytes + this lane's column base in one v_mad_u32_u16 each (op_sel picks the high half of a pair)
                typedef __attribute__((address_space(3))) const fq lds_fq;
                const u4 pk16 = *reinterpret_cast<const u4*>(reinterpret_cast<const uint16_t*>(myoff) + t0);
                const uint32_t rowb = (uint32_t)(ld * 4);
                const uint32_t base3 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)opB;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t alo, ahi;
                    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(alo) : "v"(pk16[w]), "s"(rowb), "v"(base3));
                    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ahi) : "v"(pk16[w]), "s"(rowb), "v"(base3));
                    q[2 * w] = *reinterpret_cast<lds_fq*>((uintptr_t)alo);
                    q[2 * w + 1] = *reinterpret_cast<lds_fq*>((uintptr_t)ahi);
                }
            } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) ov[u] = *reinterpret_cast<const u4*>(myoff + t0 + 4 * u);
            if (DDX_SPMM_DBG & 4) {
#pragma unroll
                for (int u = 0; u < 2; ++u) { fv[u] = (f4v)((float)t0); ov[u] = (u4)((uint32_t)(t0 * 160 + u * 640)) + (u4){0u, 160u, 320u, 480u}; }
            }
            if (DDX_SPMM_DBG & 64) {
                // ablation: operand rows whose bank ranges never collide inside a half-wave (row classes 0, 1, 7 mod 8 for the
                // three lane groups): what conflict-free operand reads would be worth
                const uint32_t cls = (uint32_t)((myoff - offS) / kLdsOStride) == 0u ? 0u : ((uint32_t)((myoff - offS) / kLdsOStride) == 1u ? 1u : 7u);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[u][e] = ((((uint32_t)t0 + 4u * u + e) & 63u) * 8u + cls) * (uint32_t)(ld * 4);
            }
            if (DDX_SPMM_DBG & 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = (fq)(__builtin_bit_cast(float, ov[u >> 2][u & 3]));
            } else {
            q[0] = *reinterpret_cast<const fq*>(opB + ov[0].x); q[1] = *reinterpret_cast<const fq*>(opB + ov[0].y);
            q[2] = *reinterpret_cast<const fq*>(opB + ov[0].z); q[3] = *reinterpret_cast<const fq*>(opB + ov[0].w);
            q[4] = *reinterpret_cast<const fq*>(opB + ov[1].x); q[5] = *reinterpret_cast<const fq*>(opB + ov[1].y);
            q[6] = *reinterpret_cast<const fq*>(opB + ov[1].z); q[7] = *reinterpret_cast<const fq*>(opB + ov[1].w);
            }
            }
            fq p0 = q[0] * fv[0].x;
            fq p1 = q[1] * fv[0].y;
            p0 = __builtin_elementwise_fma(q[2], (fq)(fv[0].z), p0);
            p1 = __builtin_elementwise_fma(q[3], (fq)(fv[0].w), p1);
            p0 = __builtin_elementwise_fma(q[4], (fq)(fv[1].x), p0);
            p1 = __builtin_elementwise_fma(q[5], (fq)(fv[1].y), p1);
            p0 = __builtin_elementwise_fma(q[6], (fq)(fv[1].z), p0);
            p1 = __builtin_elementwise_fma(q[7], (fq)(fv[1].w), p1);
            p0 = p0 + p1;
            if (DDX_SPMM_ROUNDSUM) {
                rsum = rsum + p0;
            } else {
#pragma unroll
                for (int c = 0; c < CPL; ++c) acc[c] += (double)p0[c];
            }
        }
        if (DDX_SPMM_ROUNDSUM) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] += (double)rsum[c];
        }
        if (half_tail) {
            typedef __attribute__((address_space(3))) const fq lds_fq;
            typedef uint32_t u2 __attribute__((ext_vector_type(2)));
            const int t0 = nfull;
            const f4v fv = *reinterpret_cast<const f4v*>(myf + t0);
            const u2 pk16 = *reinterpret_cast<const u2*>(reinterpret_cast<const uint16_t*>(myoff) + t0);
            const uint32_t rowb = (uint32_t)(ld * 4);
            const uint32_t base3 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)opB;
            fq q[4];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                uint32_t alo, ahi;
                asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(alo) : "v"(pk16[w]), "s"(rowb), "v"(base3));
                asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ahi) : "v"(pk16[w]), "s"(rowb), "v"(base3));
                q[2 * w] = *reinterpret_cast<lds_fq*>((uintptr_t)alo);
                q[2 * w + 1] = *reinterpret_cast<lds_fq*>((uintptr_t)ahi);
            }
            fq p0 = q[0] * fv.x;
            fq p1 = q[1] * fv.y;
            p0 = __builtin_elementwise_fma(q[2], (fq)(fv.z), p0);
            p1 = __builtin_elementwise_fma(q[3], (fq)(fv.w), p1);
            p0 = p0 + p1;
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] += (double)p0[c];
        }
        wave_lds_sync();
        return;
    }
    typedef double d2 __attribute__((ext_vector_type(2)));
    for (int t0 = 0; t0 < nsteps; t0 += 8) {
        d2 dv[4];
        u4 ov[2];
#pragma unroll
        for (int u = 0; u < 4; ++u) dv[u] = *reinterpret_cast<const d2*>(myd + t0 + 2 * u);
#pragma unroll
        for (int u = 0; u < 2; ++u) ov[u] = *reinterpret_cast<const u4*>(myoff + t0 + 4 * u);
        fq q[8];
        q[0] = *reinterpret_cast<const fq*>(opB + ov[0].x); q[1] = *reinterpret_cast<const fq*>(opB + ov[0].y);
        q[2] = *reinterpret_cast<const fq*>(opB + ov[0].z); q[3] = *reinterpret_cast<const fq*>(opB + ov[0].w);
        q[4] = *reinterpret_cast<const fq*>(opB + ov[1].x); q[5] = *reinterpret_cast<const fq*>(opB + ov[1].y);
        q[6] = *reinterpret_cast<const fq*>(opB + ov[1].z); q[7] = *reinterpret_cast<const fq*>(opB + ov[1].w);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = fma(dv[u].x, (double)q[2 * u][c], acc[c]);
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = fma(dv[u].y, (double)q[2 * u + 1][c], acc[c]);
        }
    }
    wave_lds_sync();
}

// CPL = 4 ("quad" geometry): four groups of 16 lanes that coincide with the four lane groups in which the LDS
// services a ds_read_b128 -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32 -- so the lanes served in one
// LDS cycle all read the same operand row (no bank conflicts) or the same staged entry (broadcast).  A lane holds
// four adjacent sketch columns; with ld = 40 ten of the 16 lanes of a group work.
template <bool ROWS, int SLOTS, bool PK, int CPL, int OWN>
__global__ void __launch_bounds__(kLdsThreads) k_spmm_lds(const LdsSpmmArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* opS = reinterpret_cast<float*>(smem);
    float* zS = opS + (size_t)a.SR * a.ld;
    unsigned char* stg = reinterpret_cast<unsigned char*>(zS + (ROWS ? ((a.SR + 3) & ~3) : 0));   // 16-byte aligned
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* dS = reinterpret_cast<double*>(stg + wave * (SLOTS * (lds_value_bytes(PK) + kLdsOStride * 4)));
    uint32_t* offS = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(dS) + SLOTS * lds_value_bytes(PK));
    int slot, sub;
    if (CPL == 4) {
        const int h = lane & 31;
        const bool odd = (h >= 4 && h < 12) || (h >= 16 && h < 20) || h >= 28;
        slot = (lane >> 5) * 2 + (odd ? 1 : 0);
        sub = odd ? (h < 12 ? h - 4 : (h < 20 ? h - 8 : h - 16)) : (h < 4 ? h : (h < 16 ? h - 8 : h - 12));
    } else {
        slot = lane / a.lpn;
        sub = lane - slot * a.lpn;
    }
    const bool active = slot < SLOTS && CPL * sub < a.ld;
    if (slot >= SLOTS) slot = SLOTS - 1;            // idle lanes shadow the last group (reads only)
    if (CPL * sub >= a.ld) sub = 0;
    const double* myd = reinterpret_cast<const double*>(reinterpret_cast<const unsigned char*>(dS) + slot * lds_value_bytes(PK));
    const uint32_t* myoff = offS + slot * kLdsOStride;
    const unsigned char* opB = smem + 4 * CPL * sub;  // this lane's sketch columns of operand row 0
    const int owner = (int)(blockIdx.x % a.owners);
    const int group = (int)(blockIdx.x / a.owners);
    const bool zuni = ROWS && a.z_uniform != 0;
    constexpr int PERW = SLOTS * OWN;               // outputs per wave; local output m = k*SLOTS + g

    // Local output (k, g) of this wave -> rank in the outputs sorted by their number of stored entries -> output.
    // The SLOTS outputs of a unit are consecutive ranks (near-equal segment lengths, so the lock step wastes
    // little); the units of a wave, the waves of a workgroup and the workgroups are interleaved over the whole
    // ranking, so they all carry the same load.
    auto out_index = [&](int m) -> int64_t {
        const int k = m / SLOTS, g = m - k * SLOTS;
        const int64_t rank = (((int64_t)k * kLdsWaves + wave) * a.owners + owner) * SLOTS + g;
        return rank < a.nOut ? (int64_t)a.perm[rank] : a.nOut;
    };
    // lane m < PERW looks after the segment bounds of local output m
    const int64_t myout = lane < PERW ? out_index(lane) : a.nOut;
    const bool mine = myout < a.nOut;
    int64_t rowbase = 0;
    if (ROWS && mine) rowbase = a.indptr[myout];
    float zmine = 0.0f;                              // COLS: z of the owned column, handed out by readlane
    if (!ROWS && mine) zmine = a.zcol[myout];
    double acc[OWN][CPL];
#pragma unroll
    for (int k = 0; k < OWN; ++k)
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[k][c] = 0.0;

    for (int s = group; s < a.nslices; s += a.groups) {
        __syncthreads();
        const int64_t r0 = (int64_t)s * a.SR;
        const int nr = (int)((a.opRows - r0) < a.SR ? (a.opRows - r0) : a.SR);
        {
            const int nvec = (nr * a.ld + 3) >> 2;
            const f4v* src = reinterpret_cast<const f4v*>(a.op + r0 * a.ld);
            f4v* dst = reinterpret_cast<f4v*>(opS);
            // asynchronous global -> LDS copy (16 bytes per lane, LDS destination = wave-uniform base + lane*16):
            // all of a wave's pieces are in flight together and no register is tied up
            for (int base = wave * 64; base < nvec; base += kLdsThreads) {
                if (DDX_SPMM_DBG & 8) break;
                if (base + lane < nvec)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + lane),
                                                     (__attribute__((address_space(3))) void*)(dst + base), 16, 0, 0);
            }
            if (ROWS && !zuni)
                for (int i = threadIdx.x; i < nr; i += kLdsThreads) zS[i] = a.zcol[r0 + i];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        // COLS: the entries of a slice come from the mirror of the original rows, of the synthetic rows, or (the
        // one slice that straddles row N) from both, one after the other
        for (int st = 0; st < (ROWS ? 1 : 2); ++st) {
            const int32_t* sidx = a.cols;
            const float* sx = a.x;
            int32_t lo = 0, hi = 0;
            if (ROWS) {
                if (mine) {
                    const int32_t* rs = a.rowseg + myout * (a.nslices + 1) + s;
                    lo = (int32_t)(rowbase + rs[0]);
                    hi = (int32_t)(rowbase + rs[1]);
                }
            } else if (st == 0) {
                if (s >= a.P_o) continue;
                sidx = a.row_o; sx = a.x_o;
                if (mine) { lo = (int32_t)a.cp_o[(int64_t)s * a.nOut + myout]; hi = (int32_t)a.cp_o[(int64_t)s * a.nOut + myout + 1]; }
            } else {
                const int ps = s - a.p_s0;
                if (ps < 0 || ps >= a.P_s) continue;
                sidx = a.row_s; sx = a.x_s;
                if (mine) { lo = (int32_t)a.cp_s[(int64_t)ps * a.nOut + myout]; hi = (int32_t)a.cp_s[(int64_t)ps * a.nOut + myout + 1]; }
            }

            // unit k covers the SLOTS segments of local outputs k*SLOTS + g
            auto unit_bounds = [&](int k, int32_t (&l)[SLOTS], int32_t (&h)[SLOTS], const int32_t* (&ip)[SLOTS], const float* (&xp)[SLOTS], int& maxlen) {
                maxlen = 0;
#pragma unroll
                for (int g = 0; g < SLOTS; ++g) {
                    l[g] = __builtin_amdgcn_readlane(lo, k * SLOTS + g);
                    h[g] = __builtin_amdgcn_readlane(hi, k * SLOTS + g);
                    ip[g] = sidx;
                    xp[g] = sx;
                    const int len = h[g] - l[g];
                    maxlen = len > maxlen ? len : maxlen;
                }
            };
            int32_t l[SLOTS], h[SLOTS];
            const int32_t* ip[SLOTS];
            const float* xp[SLOTS];
            int maxlen;
            unit_bounds(0, l, h, ip, xp, maxlen);
            LdsFetch<SLOTS> cur = lds_fetch<SLOTS>(ip, xp, l, h, 0, lane);
#pragma unroll
            for (int k = 0; k < OWN; ++k) {
                int32_t l2[SLOTS], h2[SLOTS];
                const int32_t* ip2[SLOTS];
                const float* xp2[SLOTS];
#pragma unroll
                for (int g = 0; g < SLOTS; ++g) { l2[g] = 0; h2[g] = 0; ip2[g] = sidx; xp2[g] = sx; }
                int maxlen2 = 0;
                if (k + 1 < OWN) unit_bounds(k + 1, l2, h2, ip2, xp2, maxlen2);
                double zc[SLOTS];
#pragma unroll
                for (int g = 0; g < SLOTS; ++g)
                    zc[g] = ROWS ? 0.0 : (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zmine), k * SLOTS + g));
                for (int r = 0;; ++r) {
                    const bool more = (r + 1) * kLdsChunk < maxlen;
                    LdsFetch<SLOTS> nxt = cur;
                    if (more) nxt = lds_fetch<SLOTS>(ip, xp, l, h, r + 1, lane);
                    else if (k + 1 < OWN) nxt = lds_fetch<SLOTS>(ip2, xp2, l2, h2, 0, lane);
                    const int left = maxlen - r * kLdsChunk;
                    if (left > 0) {
                        const int nsteps = left < kLdsChunk ? left : kLdsChunk;
                        int nvalid[SLOTS];
#pragma unroll
                        for (int g = 0; g < SLOTS; ++g) nvalid[g] = (h[g] - l[g]) - r * kLdsChunk;
                        lds_round<ROWS, SLOTS, PK, CPL>(cur, nvalid, nsteps, (int32_t)r0, zc, opB, zS, zuni, a.zval, a.ld, dS, offS, lane, myd, myoff, acc[k]);
                    }
                    cur = nxt;
                    if (!more) break;
                }
                if (k + 1 < OWN) {
#pragma unroll
                    for (int g = 0; g < SLOTS; ++g) { l[g] = l2[g]; h[g] = h2[g]; ip[g] = ip2[g]; xp[g] = xp2[g]; }
                    maxlen = maxlen2;
                }
            }
        }
    }
    // every lane group writes its own outputs
    double cm[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) cm[c] = 0.0;
#pragma unroll
    for (int k = 0; k < OWN; ++k) {
        const int64_t o = __shfl(myout, k * SLOTS + slot, 64);     // the bounds lane of this group's k-th output knows it
        if (active && o < a.nOut) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int col = CPL * sub + c;
                if (col < a.L) {
                    if (ROWS) {
                        double y = acc[k][c] - a.tvec[col];
                        if (a.accumulate) y += a.out[o * a.L + col];
                        a.out[o * a.L + col] = y;
                        if (a.ymax) { const double v = fabs(a.srow[o] * y); cm[c] = v > cm[c] ? v : cm[c]; }
                        if (a.out32) a.out32[o * a.ld + col] = (float)y;
                    } else {
                        a.out[((int64_t)group * a.nOut + o) * a.L + col] = acc[k][c];
                    }
                }
            }
        }
    }
    if (ROWS && a.ymax) {                               // (uniform) column maxima: lanes -> LDS -> one global atomic per column and workgroup
        unsigned long long* red = reinterpret_cast<unsigned long long*>(smem);
        __syncthreads();                                // everybody is done with the operand slice
        if (threadIdx.x < 64) red[threadIdx.x] = 0ull;
        __syncthreads();
        if (active) {
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (CPL * sub + c < a.L && cm[c] > 0.0) atomicMax(&red[CPL * sub + c], (unsigned long long)__double_as_longlong(cm[c]));
        }
        __syncthreads();
        if ((int)threadIdx.x < a.L && red[threadIdx.x]) atomicMax(a.ymax + threadIdx.x, red[threadIdx.x]);
    }
}

// ------------------------------------------------------------------------------------------------
// Packed residual products (bit-plane mode, round 5).  Once the entries equal to 1 have gone to the matrix cores, a (row, slice)
// segment of the A Q pass holds ~7 entries and a round of k_spmm_lds above -- segment bounds by readlane, six 4-byte fetches, index
// arithmetic, six staging stores, two wave syncs, for ONE partial trip -- costs five times what its entries cost: 0.22 ms for a tenth
// of the entries against 0.65 ms for all of them.  But that staging work is the same in all sixteen products of an iteration (the
// matrix does not change between them), so it is done ONCE per iteration by k_pack_residual and the products read the result:
//
//   * same ownership as k_spmm_lds (wave = 3 lane groups x OWN outputs, ranked outputs interleaved over waves and workgroups),
//     same operand slices, same trips of 8 / 4 steps in the same order -- the sums are bit-identical to k_spmm_lds';
//   * per (wave, slice) the steps of its OWN units are concatenated into QUADS of 4 lock-step steps (3 slots x 4 entries:
//     float32 x - z and 16-bit operand row inside the slice, zero padded), twelve quads to a self-describing 1 KB BLOCK
//     (header: quads per unit in this block); a wave's blocks lie one behind the other in memory, slice after slice;
//   * the product kernel copies a wave's blocks global -> LDS asynchronously (one 1 KB instruction each) into a ring of two or three
//     buffers, ahead of the one it works through and WHATEVER slice they belong to: per block one header read and then nothing but
//     trips.
// ------------------------------------------------------------------------------------------------
constexpr int kPkQuadBytes = 80;        // val[3][4] float32 (48) | idx[3][4] uint16 (24) | pad (8): 16-byte aligned
constexpr int kPkQuads = 12;            // quads per block
constexpr int kPkHeader = 64;           // bytes before the first quad: quads per unit (uint8, at most 12 units)
constexpr int kPkBlockBytes = kPkHeader + kPkQuads * kPkQuadBytes;   // 1024
static_assert(kPkBlockBytes == 1024, "a block is one wave-wide 16-byte copy");

struct PackedArgs {
    LdsSpmmArgs a;                // geometry, sources, outputs as for k_spmm_lds (ROWS: indptr / cols / x / rowseg; COLS: the mirrors)
    int own;                      // outputs per lane group
    int nsl;                      // slices a workgroup walks (ROWS: all of them; COLS: every groups-th)
    const int32_t* blkptr;        // [waves x (nsl + 1)] block range of a wave's t-th slice
    const unsigned char* blocks;  // the packed blocks
};

// the segment of local output m = (k, g) of wave `wave` of workgroup (owner, group) in slice s: up to two source ranges
struct PkSeg { const int32_t* idx[2]; const float* x[2]; int32_t lo[2], hi[2]; };

template <bool ROWS>
__device__ __forceinline__ PkSeg pk_segment(const LdsSpmmArgs& a, int64_t out, int s) {
    PkSeg sg;
    sg.idx[0] = sg.idx[1] = a.cols; sg.x[0] = sg.x[1] = a.x;
    sg.lo[0] = sg.hi[0] = sg.lo[1] = sg.hi[1] = 0;
    if (out >= a.nOut) return sg;
    if (ROWS) {
        const int64_t base = a.indptr[out];
        const int32_t* rs = a.rowseg + out * (a.nslices + 1) + s;
        sg.lo[0] = (int32_t)(base + rs[0]); sg.hi[0] = (int32_t)(base + rs[1]);
    } else {
        if (s < a.P_o) {
            sg.idx[0] = a.row_o; sg.x[0] = a.x_o;
            sg.lo[0] = (int32_t)a.cp_o[(int64_t)s * a.nOut + out]; sg.hi[0] = (int32_t)a.cp_o[(int64_t)s * a.nOut + out + 1];
        }
        const int ps = s - a.p_s0;
        if (ps >= 0 && ps < a.P_s) {
            sg.idx[1] = a.row_s; sg.x[1] = a.x_s;
            sg.lo[1] = (int32_t)a.cp_s[(int64_t)ps * a.nOut + out]; sg.hi[1] = (int32_t)a.cp_s[(int64_t)ps * a.nOut + out + 1];
        }
    }
    return sg;
}

// One wave of this kernel = one (wave of the product kernel, slice) pair: wgl = product workgroup * 16 + wave, t = index in that
// workgroup's slice list.  Lane m < 3 own looks after segment (k, g) = (m / 3, m % 3); the entries of all segments are then spread
// over the 64 lanes (a wave scan of the lengths, a binary search per entry) so that the copy runs at full width.
// fill == 0: ptr[wgl * (nsl + 1) + t] = blocks the pair needs.   fill == 1: ptr holds the block offsets; writes the blocks (zeroed beforehand).
template <bool ROWS>
__global__ void __launch_bounds__(256) k_pack_residual(const LdsSpmmArgs a, int own, int nsl, int nwg, int32_t* __restrict__ ptr, unsigned char* __restrict__ blocks, int fill) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (int64_t)nwg * kLdsWaves * nsl) return;
    const int64_t wgl = pair / nsl;
    const int t = (int)(pair - wgl * nsl);
    const int bid = (int)(wgl / kLdsWaves), wave = (int)(wgl % kLdsWaves);
    const int owner = bid % a.owners, group = bid / a.owners;
    const int perw = 3 * own;
    const int k = lane / 3, g = lane - 3 * k;
    int64_t out = a.nOut;
    if (lane < perw) {
        const int64_t rank = (((int64_t)k * kLdsWaves + wave) * a.owners + owner) * 3 + g;
        out = rank < a.nOut ? (int64_t)a.perm[rank] : a.nOut;
    }
    const int s = ROWS ? t : group + t * a.groups;
    const PkSeg sg = pk_segment<ROWS>(a, s < a.nslices ? out : a.nOut, s < a.nslices ? s : 0);
    // (the two sources of a segment -- the original rows' and the synthetic rows' mirror in the one panel that straddles row N --
    // keep trips of their own, as in k_spmm_lds)
    const int n0 = sg.hi[0] - sg.lo[0], n1 = sg.hi[1] - sg.lo[1];
    // quads of the lane's unit = ceil(longest of its three segments / 4) per source; prefix over the units before it
    const int b3 = 3 * k < 62 ? 3 * k : 0;
    int mx0 = max(max(__shfl(n0, b3, 64), __shfl(n0, b3 + 1, 64)), __shfl(n0, b3 + 2, 64));
    int mx1 = max(max(__shfl(n1, b3, 64), __shfl(n1, b3 + 1, 64)), __shfl(n1, b3 + 2, 64));
    const int nq0 = lane < perw ? (mx0 + 3) >> 2 : 0;
    const int nq = lane < perw ? nq0 + ((mx1 + 3) >> 2) : 0;
    int pre = 0, total = 0;
    for (int kk = 0; kk < own; ++kk) {
        const int v = __shfl(nq, 3 * kk, 64);
        if (kk < k) pre += v;
        total += v;
    }
    const int64_t slot = wgl * (nsl + 1) + t;
    if (!fill) {
        if (lane == 0) ptr[slot] = (total + kPkQuads - 1) / kPkQuads;
        return;
    }
    if (total == 0) return;
    const int64_t blk0 = ptr[slot];
    if (lane < perw && g == 0 && nq > 0) {                // header bytes of the blocks this unit's quads fall into
        for (int bb = pre / kPkQuads; bb <= (pre + nq - 1) / kPkQuads; ++bb) {
            const int lo = max(pre, bb * kPkQuads), hi = min(pre + nq, (bb + 1) * kPkQuads);
            blocks[(blk0 + bb) * kPkBlockBytes + k] = (unsigned char)(hi - lo);
        }
    }
    // inclusive scan of the segment lengths over the lanes
    const int len = n0 + n1;
    int incl = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const int n_all = __shfl(incl, 63, 64);
    const float z = ROWS ? a.zval : (out < a.nOut ? a.zcol[out] : 0.0f);
    const int32_t base = s * a.SR;
    for (int e0 = 0; e0 < n_all; e0 += 64) {             // (all lanes stay in the loop: they are shuffle sources)
        const int e = e0 + lane;
        // the segment that holds entry e: the first lane whose inclusive count exceeds e
        int lo = 0, hi = 63;
        const int ee = e < n_all ? e : n_all - 1;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi) >> 1;
            const int v = __shfl(incl, mid, 64);
            if (v > ee) hi = mid; else lo = mid + 1;
        }
        const int m = lo;
        const int m_incl = __shfl(incl, m, 64), m_len = __shfl(len, m, 64), m_n0 = __shfl(n0, m, 64);
        const int m_lo0 = __shfl(sg.lo[0], m, 64), m_lo1 = __shfl(sg.lo[1], m, 64);
        const int m_pre = __shfl(pre, m, 64), m_nq0 = __shfl(nq0, m, 64);
        const float m_z = __shfl(z, m, 64);
        if (e < n_all) {
            const int el = ee - (m_incl - m_len);            // position inside segment m
            const int src = el < m_n0 ? 0 : 1;
            const int es = src == 0 ? el : el - m_n0;
            const int32_t p = (src == 0 ? m_lo0 : m_lo1) + es;
            const float* xs = ROWS ? a.x : (src == 0 ? a.x_o : a.x_s);
            const int32_t* is = ROWS ? a.cols : (src == 0 ? a.row_o : a.row_s);
            const int32_t ii = is[p];
            const float zz = (ROWS && !a.z_uniform) ? a.zcol[ii] : m_z;     // (A Q on a scaled matrix: the unstored value depends on the entry's column)
            const float v = xs[p] - zz;                      // x - z in float32: the correctly rounded difference (as k_spmm_lds stages it)
            const int32_t i = ii - base;
            const int mg = m % 3;
            const int qpos = m_pre + (src == 0 ? 0 : m_nq0) + (es >> 2), st = es & 3;
            unsigned char* q = blocks + (blk0 + qpos / kPkQuads) * kPkBlockBytes + kPkHeader + (qpos % kPkQuads) * kPkQuadBytes;
            reinterpret_cast<float*>(q)[mg * 4 + st] = v;
            reinterpret_cast<uint16_t*>(q + 48)[mg * 4 + st] = (uint16_t)i;
        }
    }
}

#ifndef DDX_PK_DBG
#define DDX_PK_DBG 0        // ablation builds only (timing; wrong results): 1 no operand-slice staging, 2 no trips, 4 no block copies
#endif
template <bool ROWS, int OWN, int RING>
__global__ void __launch_bounds__(kLdsThreads) k_spmm_packed(const PackedArgs pa) {
    constexpr int kPkRing = RING;       // block buffers per wave: the current block and the next RING - 1, already on their way
    typedef float fq __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const fq lds_fq;
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const LdsSpmmArgs& a = pa.a;
    extern __shared__ __align__(16) unsigned char smem[];
    float* opS = reinterpret_cast<float*>(smem);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned char* stg = smem + (((size_t)a.SR * a.ld * 4 + 15) & ~(size_t)15) + (size_t)wave * (kPkRing * kPkBlockBytes);     // the wave's ring of block buffers
    int slot = lane / a.lpn, sub = lane - slot * a.lpn;
    const bool active = slot < 3 && 2 * sub < a.ld;
    if (slot >= 3) slot = 2;                         // idle lanes shadow the last group (reads only)
    if (2 * sub >= a.ld) sub = 0;
    const int owner = (int)(blockIdx.x % a.owners);
    const int group = (int)(blockIdx.x / a.owners);
    constexpr int PERW = 3 * OWN;
    auto out_index = [&](int m) -> int64_t {
        const int k = m / 3, g = m - k * 3;
        const int64_t rank = (((int64_t)k * kLdsWaves + wave) * a.owners + owner) * 3 + g;
        return rank < a.nOut ? (int64_t)a.perm[rank] : a.nOut;
    };
    const int64_t myout = lane < PERW ? out_index(lane) : a.nOut;
    double acc[OWN][2];
#pragma unroll
    for (int k = 0; k < OWN; ++k) acc[k][0] = acc[k][1] = 0.0;
    const int32_t* bp = pa.blkptr + ((int64_t)blockIdx.x * kLdsWaves + wave) * (pa.nsl + 1);
    const uint32_t rowb = (uint32_t)(a.ld * 4);
    const uint32_t base3 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(smem + 8 * sub);
    // the wave's blocks: [bbeg, bend), block b into buffer (b - bbeg) % kPkRing
    const int32_t bbeg = __builtin_amdgcn_readfirstlane(bp[0]);
    const int32_t bend = __builtin_amdgcn_readfirstlane(bp[pa.nsl]);      // (slices past the last one hold no blocks)
    auto copy_block = [&](int32_t b) {
        if (DDX_PK_DBG & 4) return;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa.blocks + (int64_t)b * kPkBlockBytes + lane * 16),
                                         (__attribute__((address_space(3))) void*)(stg + ((b - bbeg) % kPkRing) * kPkBlockBytes), 16, 0, 0);
    };
    int32_t nf = bbeg;                                  // the next block to ask for
    for (; nf < bend && nf < bbeg + kPkRing; ++nf) copy_block(nf);
    int32_t b = bbeg;                                   // the next block to work through
    for (int t = 0; t < pa.nsl; ++t) {
        const int s = ROWS ? t : group + t * a.groups;
        if (s >= a.nslices) break;
        // everybody is done with the previous slice.  (A bare barrier: __syncthreads() would also wait for the block copies in flight.)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int32_t b1v = bp[t + 1];                  // (arrives with the slice)
        {
            const int64_t r0 = (int64_t)s * a.SR;
            const int nr = (int)((a.opRows - r0) < a.SR ? (a.opRows - r0) : a.SR);
            const int nvec = (nr * a.ld + 3) >> 2;
            const f4v* src = reinterpret_cast<const f4v*>(a.op + r0 * a.ld);
            f4v* dst = reinterpret_cast<f4v*>(opS);
            for (int base = wave * 64; base < nvec; base += kLdsThreads)
                if (base + lane < nvec && !(DDX_PK_DBG & 1))
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + lane),
                                                     (__attribute__((address_space(3))) void*)(dst + base), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the slice -- and with it every block asked for so far)
        }
        const int32_t b1 = __builtin_amdgcn_readfirstlane(b1v);
        asm volatile("s_barrier" ::: "memory");
        for (; b < b1; ++b) {
            // block b has landed once at most the copies asked for after it are in flight (a wave's copies complete in order)
            const int newer = nf - 1 - b;
            if (kPkRing > 3 && newer >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (kPkRing > 2 && newer == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (newer == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            static_assert(kPkRing >= 2 && kPkRing <= 4, "the wait counts above");
            const unsigned char* blk = stg + ((b - bbeg) % kPkRing) * kPkBlockBytes;
            // quads per unit of this block: three 32-bit words of four counts each, wave-uniform
            const uint32_t* hw = reinterpret_cast<const uint32_t*>(blk);
            uint32_t hdr[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) hdr[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)hw[i]);
            const unsigned char* qp = blk + kPkHeader + slot * 16;      // this group's values of quad 0; its indices at + 48 - slot * 8
            const unsigned char* ip = blk + kPkHeader + 48 + slot * 8;
#pragma unroll
            for (int k = 0; k < OWN; ++k) {
                const int n = (DDX_PK_DBG & 2) ? 0 : (int)((hdr[k >> 2] >> (8 * (k & 3))) & 0xffu);
                int q = 0;
                for (; q + 1 < n; q += 2) {              // a trip of eight steps = two quads
                    const f4v f0 = *reinterpret_cast<const f4v*>(qp), f1 = *reinterpret_cast<const f4v*>(qp + kPkQuadBytes);
                    const u2 i0 = *reinterpret_cast<const u2*>(ip), i1 = *reinterpret_cast<const u2*>(ip + kPkQuadBytes);
                    fq v[8];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t pk = w < 2 ? i0[w] : i1[w - 2];
                        uint32_t alo, ahi;
                        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(alo) : "v"(pk), "s"(rowb), "v"(base3));
                        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ahi) : "v"(pk), "s"(rowb), "v"(base3));
                        v[2 * w] = *reinterpret_cast<lds_fq*>((uintptr_t)alo);
                        v[2 * w + 1] = *reinterpret_cast<lds_fq*>((uintptr_t)ahi);
                    }
                    fq p0 = v[0] * f0.x;
                    fq p1 = v[1] * f0.y;
                    p0 = __builtin_elementwise_fma(v[2], (fq)(f0.z), p0);
                    p1 = __builtin_elementwise_fma(v[3], (fq)(f0.w), p1);
                    p0 = __builtin_elementwise_fma(v[4], (fq)(f1.x), p0);
                    p1 = __builtin_elementwise_fma(v[5], (fq)(f1.y), p1);
                    p0 = __builtin_elementwise_fma(v[6], (fq)(f1.z), p0);
                    p1 = __builtin_elementwise_fma(v[7], (fq)(f1.w), p1);
                    p0 = p0 + p1;
                    acc[k][0] += (double)p0[0];
                    acc[k][1] += (double)p0[1];
                    qp += 2 * kPkQuadBytes; ip += 2 * kPkQuadBytes;
                }
                if (q < n) {                             // a half trip of four
                    const f4v f0 = *reinterpret_cast<const f4v*>(qp);
                    const u2 i0 = *reinterpret_cast<const u2*>(ip);
                    fq v[4];
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        uint32_t alo, ahi;
                        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(alo) : "v"(i0[w]), "s"(rowb), "v"(base3));
                        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ahi) : "v"(i0[w]), "s"(rowb), "v"(base3));
                        v[2 * w] = *reinterpret_cast<lds_fq*>((uintptr_t)alo);
                        v[2 * w + 1] = *reinterpret_cast<lds_fq*>((uintptr_t)ahi);
                    }
                    fq p0 = v[0] * f0.x;
                    fq p1 = v[1] * f0.y;
                    p0 = __builtin_elementwise_fma(v[2], (fq)(f0.z), p0);
                    p1 = __builtin_elementwise_fma(v[3], (fq)(f0.w), p1);
                    p0 = p0 + p1;
                    acc[k][0] += (double)p0[0];
                    acc[k][1] += (double)p0[1];
                    qp += kPkQuadBytes; ip += kPkQuadBytes;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the buffer is handed to the copy of block b + kPkRing)
            __builtin_amdgcn_wave_barrier();
            if (nf < bend) { copy_block(nf); ++nf; }
        }
    }
    // every lane group writes its own outputs (as k_spmm_lds)
    double cm[2] = {0.0, 0.0}, cs[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < OWN; ++k) {
        const int64_t o = __shfl(myout, k * 3 + slot, 64);
        if (active && o < a.nOut) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int col = 2 * sub + c;
                if (col < a.L) {
                    if (ROWS) {
                        double y = acc[k][c] - a.tvec[col];
                        if (a.accumulate) y += a.out[o * a.L + col];
                        a.out[o * a.L + col] = y;
                        cs[c] += y;
                        if (a.ymax) { const double v = fabs(a.srow[o] * y); cm[c] = v > cm[c] ? v : cm[c]; }
                        if (a.out32) a.out32[o * a.ld + col] = (float)y;
                    } else {
                        a.out[((int64_t)group * a.nOut + o) * a.L + col] = acc[k][c];
                    }
                }
            }
        }
    }
    if (ROWS && a.ymax) {
        unsigned long long* red = reinterpret_cast<unsigned long long*>(smem);
        __syncthreads();
        if (threadIdx.x < 64) red[threadIdx.x] = 0ull;
        __syncthreads();
        if (active) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (2 * sub + c < a.L && cm[c] > 0.0) atomicMax(&red[2 * sub + c], (unsigned long long)__double_as_longlong(cm[c]));
        }
        __syncthreads();
        if ((int)threadIdx.x < a.L && red[threadIdx.x]) atomicMax(a.ymax + threadIdx.x, red[threadIdx.x]);
    }
    if (ROWS && a.usum) {
        // column sums of this workgroup's rows: (wave, lane group) partials through LDS, added in that order; then the last workgroup adds the workgroups'
        double* cs_s = reinterpret_cast<double*>(smem) + 64;          // (behind the 64 maxima)
        __syncthreads();
        if (active) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (2 * sub + c < a.L) cs_s[(wave * 3 + slot) * a.L + 2 * sub + c] = cs[c];
        }
        __syncthreads();
        if ((int)threadIdx.x < a.L) {
            double t = 0.0;
            for (int g = 0; g < kLdsWaves * 3; ++g) t += cs_s[g * a.L + threadIdx.x];
            partial_store(&a.upart[(int64_t)blockIdx.x * a.L + threadIdx.x], t);
        }
        int* s_flag = reinterpret_cast<int*>(cs_s + kLdsWaves * 3 * 64);      // (behind the partials, inside the operand slice's area)
        if (last_block_ticket(a.ucount, (int)gridDim.x, s_flag)) reduce_partials_block(a.upart, (int)gridDim.x, a.L, a.usum);
    }
}

// rowseg[row*(ns+1) + p] = offset inside row `row` of its first stored entry whose column is >= p*SR (p = ns: row length)
// (rows row0 .. nrows-1: the rows of the original cells keep their segments from iteration to iteration)
__global__ void k_row_segments(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols, int64_t row0, int64_t nrows, int ns,
                               int SR, int32_t* __restrict__ rowseg) {
    const int64_t t = row0 * (ns + 1) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrows * (ns + 1)) return;
    const int64_t row = t / (ns + 1);
    const int p = (int)(t - row * (ns + 1));
    const int64_t b = indptr[row], e = indptr[row + 1];
    const int64_t key = (int64_t)p * SR;
    int64_t lo = b, hi = e;
    if (p == ns) lo = e;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cols[mid] < key) lo = mid + 1; else hi = mid;
    }
    rowseg[t] = (int32_t)(lo - b);
}

// padded (and possibly float32-rounded) copy of an R x L operand: out[r*ld + c], zero in the padding columns
template <typename T>
__global__ void k_operand_copy(const double* __restrict__ in, int64_t R, int L, int ld, T* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * ld) return;
    const int64_t r = t / ld;
    const int c = (int)(t - r * ld);
    out[t] = (c < L) ? (T)in[r * L + c] : (T)0;
}

// W[j,c] = sum_p Wp[p][j][c] - m_j u_c   (panels added in order)
__global__ void k_sum_panels(const double* __restrict__ Wp, int P, int32_t H, int L, const double* __restrict__ colmean,
                             const double* __restrict__ uvec, double* __restrict__ W, const double* __restrict__ extra = nullptr, int nextra = 0,
                             const double* __restrict__ escale = nullptr) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)H * L) return;
    const int64_t j = t / L;
    const int c = (int)(t - j * L);
    double s = 0.0;
    for (int p = 0; p < P; ++p) s += Wp[(int64_t)p * H * L + t];
    double e = 0.0;
    for (int p = 0; p < nextra; ++p) e += extra[(int64_t)p * H * L + t];       // the bit-plane part of the product, one block per chunk of the rows
    if (escale) e *= escale[j];                                                // (scaled matrix: B^T diag(s) Y comes out of the matrix cores, 1 / sd_j is applied here)
    W[t] = (s + e) - colmean[j] * uvec[c];
}

// ------------------------------------------------------------------------------------------------
// tall-skinny helpers on R x L row-major float64 matrices (all reductions in fixed order)
// ------------------------------------------------------------------------------------------------
// partial[b][c] = sum over the block's rows of w_r * X[r,c]   (w == nullptr: ones); out[c] = sum of the partials (last block)
__global__ void __launch_bounds__(256) k_wcolsum_partial(const double* __restrict__ X, int64_t R, int L,
                                                         const double* __restrict__ wgt, int64_t rows_per_block,
                                                         double* __restrict__ partial, int* __restrict__ counter = nullptr, double* __restrict__ out = nullptr) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int groups = 256 / L;            // row lanes per block (L <= 128 -> >= 2)
    const int g = tid / L, c = tid - g * L;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
    double acc = 0.0;
    if (g < groups) {
        for (int64_t r = r0 + g; r < r1; r += 4 * groups) {      // four rows in flight; additions in the order of the plain loop
            double v[4], wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t ru = r + (int64_t)u * groups;
                v[u] = ru < r1 ? X[ru * L + c] : 0.0;
                wv[u] = (wgt && ru < r1) ? wgt[ru] : 1.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + (int64_t)u * groups < r1) acc += wgt ? wv[u] * v[u] : v[u];
        }
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < L) {
        double s = 0.0;
        for (int gg = 0; gg < groups; ++gg) s += red[gg * L + tid];
        if (counter) partial_store(&partial[(int64_t)blockIdx.x * L + tid], s);
        else partial[(int64_t)blockIdx.x * L + tid] = s;
    }
    if (counter && last_block_ticket(counter, (int)gridDim.x)) reduce_partials_block(partial, (int)gridDim.x, L, out);
}

// The same factorisation and inverse with the matrix in registers (sketch width LT known at compile time, one
// column per lane, everything unrolled): entries of other columns arrive through v_readlane instead of LDS round
// trips, which bound the kernel above (68 us at L = 40 against ~25 us here).
__device__ __forceinline__ double lane_value(double v, int j) {
    const int64_t b = __builtin_bit_cast(int64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)b, j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(b >> 32), j);
    return __builtin_bit_cast(double, ((int64_t)hi << 32) | (int64_t)lo);
}

template <int LT>
__device__ __forceinline__ void chol_inv_reg_wave(const double* __restrict__ G, double* __restrict__ Rinv, int* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int col = lane < LT ? lane : LT - 1;          // idle lanes shadow the last column
    double a[LT];
#pragma unroll
    for (int i = 0; i < LT; ++i) a[i] = partial_load(&G[i * LT + col]);      // (device-scope loads: G may have been written by other workgroups of this launch)
    double maxd = 0.0;
#pragma unroll
    for (int k = 0; k < LT; ++k) maxd = fmax(maxd, lane_value(a[k], k));
    const double floor_v = maxd * 1e-26 + 1e-300;
#pragma unroll
    for (int k = 0; k < LT; ++k) {
        double d = lane_value(a[k], k);
        if (!(d > floor_v)) {
            d = floor_v;
            if (lane == 0) atomicOr(flag, 1);
        }
        const double piv = sqrt(d);
        const double r = a[k] / piv;                     // R[k][lane] for lane > k
        a[k] = lane == k ? piv : r;
#pragma unroll
        for (int i = k + 1; i < LT; ++i) a[i] -= lane_value(r, i) * r;     // meaningful for lane >= i
    }
    // column `lane` of the inverse of the upper-triangular factor, bottom up
    double x[LT];
#pragma unroll
    for (int i = LT - 1; i >= 0; --i) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int p = i + 1; p < LT; ++p) {
            const double rip = lane_value(a[i], p);      // R[i][p]
            if ((p - i) & 1) s0 += rip * x[p]; else s1 += rip * x[p];      // x[p] = 0 beyond the diagonal
        }
        const double rii = lane_value(a[i], i);
        x[i] = i == lane ? 1.0 / rii : (i < lane ? -(s0 + s1) / rii : 0.0);
    }
    if (lane < LT) {
#pragma unroll
        for (int i = 0; i < LT; ++i) Rinv[i * LT + lane] = x[i];
    }
}

template <int LT>
__global__ void __launch_bounds__(64) k_chol_inv_reg(const double* __restrict__ G, double* __restrict__ Rinv, int* __restrict__ flag) {
    chol_inv_reg_wave<LT>(G, Rinv, flag);
}

// out[c] = sum_b partial[b][c]: one wave per output; lanes take interleaved blocks, then a fixed butterfly.
// Rinv != nullptr (a 40 x 40 Gram matrix, the default sketch): the block that finishes last goes on to the Cholesky factor of `out` and its
// inverse on its first wave (last_block_ticket) -- one launch less in every Cholesky-QR of the power iterations.
__global__ void __launch_bounds__(256) k_reduce_partials(const double* __restrict__ partial, int nblocks, int width, double* __restrict__ out,
                                                         int* __restrict__ counter = nullptr, double* __restrict__ Rinv = nullptr, int* __restrict__ flag = nullptr) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c < width) {
        double s = 0.0;
        for (int b = lane; b < nblocks; b += 64) s += partial[(int64_t)b * width + c];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) { if (counter) partial_store(&out[c], s); else out[c] = s; }
    }
    if (counter && last_block_ticket(counter, (int)gridDim.x) && threadIdx.x < 64) chol_inv_reg_wave<40>(out, Rinv, flag);
}

// partial Gram: G_b = X_b^T X_b for a block of rows, staged through LDS in 32-row tiles.  Only the pairs a <= b
// are accumulated (each thread owns a few of the L(L+1)/2), both triangles are written.
template <int MAXP>   // pairs per thread: ceil(L(L+1)/2 / 256)
__global__ void __launch_bounds__(256) k_gram_partial(const double* __restrict__ X, int64_t R, int L,
                                                      int64_t rows_per_block, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double tile[];  // [32][L]
    const int tid = threadIdx.x;
    const int npairs = L * (L + 1) / 2;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
    double acc[MAXP];
    int pa[MAXP], pb[MAXP];
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
        acc[q] = 0.0;
        int rest = tid + q * 256, a = 0;          // pair number -> (a, b) with a <= b, rows of the upper triangle in order
        while (a < L && rest >= L - a) { rest -= L - a; ++a; }
        pa[q] = a;
        pb[q] = a + rest;
    }
    for (int64_t rb = r0; rb < r1; rb += 32) {
        const int nr = (int)((r1 - rb) < 32 ? (r1 - rb) : 32);
        __syncthreads();
        for (int t0 = tid; t0 < nr * L; t0 += 4 * 256) {          // (all loads of a thread in flight before the LDS stores)
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = t0 + u * 256 < nr * L ? X[rb * L + t0 + u * 256] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + u * 256 < nr * L) tile[t0 + u * 256] = v[u];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            if (tid + q * 256 < npairs) {
                const int a = pa[q], bcol = pb[q];
                double s = acc[q];
                for (int r = 0; r < nr; ++r) s = fma(tile[r * L + a], tile[r * L + bcol], s);
                acc[q] = s;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
        if (tid + q * 256 < npairs) {
            double* out = partial + (int64_t)blockIdx.x * L * L;
            out[pa[q] * L + pb[q]] = acc[q];
            out[pb[q] * L + pa[q]] = acc[q];
        }
    }
}

// single wave: G (L x L, symmetric positive definite) -> Rinv with G = R^T R, R upper triangular.
// Lane j owns column j of the upper triangle (L <= 64); row k is normalised, then every lane updates its
// own column below row k -- no cross-lane writes, one wave barrier per step.
// flag[0] |= 1 when a pivot had to be floored (rank-deficient sketch).
__global__ void __launch_bounds__(64) k_chol_inv(const double* __restrict__ G, int L, double* __restrict__ Rinv,
                                                 int* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) double sm[];  // a[L*L] | inv[L*L]
    double* a = sm;
    double* inv = sm + L * L;   // built in LDS (a lane re-reads what it wrote: through global memory that costs a round trip per entry)
    const int lane = threadIdx.x;
    for (int t = lane; t < L * L; t += 64) { a[t] = G[t]; inv[t] = 0.0; }
    __syncthreads();
    double maxd = 0.0;
    for (int k = 0; k < L; ++k) maxd = fmax(maxd, a[k * L + k]);
    const double floor_v = maxd * 1e-26 + 1e-300;
    for (int k = 0; k < L; ++k) {
        double d = a[k * L + k];
        if (!(d > floor_v)) {
            d = floor_v;
            if (lane == 0) atomicOr(flag, 1);
        }
        const double piv = sqrt(d);
        __syncthreads();                       // everyone has read a[k][k] before it is overwritten
        if (lane == k) a[k * L + k] = piv;
        if (lane > k && lane < L) {
            const double r = a[k * L + lane] / piv;      // R[k][lane]
            a[k * L + lane] = r;
        }
        __syncthreads();
        if (lane > k && lane < L) {
            const double r = a[k * L + lane];
            int i = k + 1;
            for (; i + 3 <= lane; i += 4) {          // four independent updates in flight (LDS latency, not work, bounds this)
                const double r0 = a[k * L + i], r1 = a[k * L + i + 1], r2 = a[k * L + i + 2], r3 = a[k * L + i + 3];
                const double v0 = a[i * L + lane], v1 = a[(i + 1) * L + lane], v2 = a[(i + 2) * L + lane], v3 = a[(i + 3) * L + lane];
                a[i * L + lane] = v0 - r0 * r;
                a[(i + 1) * L + lane] = v1 - r1 * r;
                a[(i + 2) * L + lane] = v2 - r2 * r;
                a[(i + 3) * L + lane] = v3 - r3 * r;
            }
            for (; i <= lane; ++i) a[i * L + lane] -= a[k * L + i] * r;
        }
        __syncthreads();
    }
    // invert the upper-triangular factor, one column per lane
    for (int jx = lane; jx < L; jx += 64) {
        inv[jx * L + jx] = 1.0 / a[jx * L + jx];
        for (int i = jx - 1; i >= 0; --i) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int p = i + 1;
            for (; p + 3 <= jx; p += 4) {
                s0 += a[i * L + p] * inv[p * L + jx];
                s1 += a[i * L + p + 1] * inv[(p + 1) * L + jx];
                s2 += a[i * L + p + 2] * inv[(p + 2) * L + jx];
                s3 += a[i * L + p + 3] * inv[(p + 3) * L + jx];
            }
            for (; p <= jx; ++p) s0 += a[i * L + p] * inv[p * L + jx];
            const double s = (s0 + s1) + (s2 + s3);
            inv[i * L + jx] = -s / a[i * L + i];
        }
    }
    __syncthreads();
    for (int t = lane; t < L * L; t += 64) Rinv[t] = inv[t];
}

// out[R x L2] = X[R x L] * T[L x L2]   (64 rows per block, X tile and T staged in LDS)
// out32 != nullptr: also the padded float32 copy of the result that the next operator product stages (ld32 columns)
// cmax != nullptr: also the largest |cw_r out[r][c]| per column (cw == nullptr: ones), by integer atomicMax into zeros -- what the bit-plane
//   A Q product cuts its digits by (k_bp_colmax as a launch of its own otherwise); needs 64 * L2 more doubles of LDS
// tw != nullptr: also tvec[c] = sum_r tw_r out[r][c] (the rank-one correction m^T Q of the A Q product that follows: a k_wcolsum_partial launch
//   otherwise): one partial per block in row order, added by the block that finishes last in a fixed order; same extra LDS
__global__ void __launch_bounds__(256) k_right_mult(const double* __restrict__ X, int64_t R, int L,
                                                    const double* __restrict__ T, int L2, double* __restrict__ out,
                                                    float* __restrict__ out32 = nullptr, int ld32 = 0, const double* __restrict__ cw = nullptr,
                                                    unsigned long long* __restrict__ cmax = nullptr, const double* __restrict__ tw = nullptr,
                                                    double* __restrict__ tpart = nullptr, int* __restrict__ counter = nullptr, double* __restrict__ tvec = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double sm[];  // T[L*L2] then xt[64*L] (then y[64*L2])
    double* t_s = sm;
    double* x_s = sm + L * L2;
    double* y_s = x_s + 64 * L;
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int nr = (int)((R - r0) < 64 ? (R - r0) : 64);
    for (int t = tid; t < L * L2; t += 256) t_s[t] = T[t];
    for (int t0 = tid; t0 < nr * L; t0 += 4 * 256) {              // (all loads of a thread in flight before the LDS stores)
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = t0 + u * 256 < nr * L ? X[r0 * L + t0 + u * 256] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 + u * 256 < nr * L) x_s[t0 + u * 256] = v[u];
    }
    __syncthreads();
    for (int o = tid; o < nr * L2; o += 256) {
        const int r = o / L2, c = o - r * L2;
        double s = 0.0;
        for (int p = 0; p < L; ++p) s = fma(x_s[r * L + p], t_s[p * L2 + c], s);
        out[(r0 + r) * L2 + c] = s;
        if (out32) out32[(r0 + r) * ld32 + c] = (float)s;
        if (cmax || tw) y_s[o] = s;
    }
    if (cmax || tw) __syncthreads();
    if (tw) {
        if (tid < L2) {
            double a = 0.0;
            for (int r = 0; r < nr; ++r) a += tw[r0 + r] * y_s[r * L2 + tid];
            partial_store(&tpart[(int64_t)blockIdx.x * L2 + tid], a);
        }
    }
    if (cmax) {
        if (tid < L2) {
            double m = 0.0;
            for (int r = 0; r < nr; ++r) {
                const double v = fabs(cw ? cw[r0 + r] * y_s[r * L2 + tid] : y_s[r * L2 + tid]);
                m = v > m ? v : m;
            }
            if (m > 0.0) atomicMax(cmax + tid, (unsigned long long)__double_as_longlong(m));
        }
    }
    if (tw && last_block_ticket(counter, (int)gridDim.x)) reduce_partials_block(tpart, (int)gridDim.x, L2, tvec);
}

// per column c of X[R x C]: index of the largest |value| (first occurrence), then its sign
__global__ void __launch_bounds__(256) k_col_sign(const double* __restrict__ X, int64_t R, int C, double* __restrict__ sign_out) {
    __shared__ double bv[256];
    __shared__ int64_t bi[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double best = -1.0;
    int64_t besti = 0;
    for (int64_t r = tid; r < R; r += 256) {
        const double v = fabs(X[r * C + c]);
        if (v > best) { best = v; besti = r; }
    }
    bv[tid] = best;
    bi[tid] = besti;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            const double ov = bv[tid + off];
            const int64_t oi = bi[tid + off];
            if (ov > bv[tid] || (ov == bv[tid] && oi < bi[tid])) { bv[tid] = ov; bi[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double v = X[bi[0] * C + c];
        sign_out[c] = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
    }
}

__global__ void k_f64_to_f32(const double* __restrict__ in, int64_t n, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// ---- sketches wider than kMaxL columns -----------------------------------------------------------------------------
// The operator products run block by block over the sketch columns (kWideBlock = the width the LDS-staged kernels are
// built for); the tall-skinny helpers get tiled variants whose LDS need does not grow with the sketch width.
constexpr int kWideBlock = 40;

// out[r][c] = in[r][c0 + c] for c < Lb, 0 for Lb <= c < B      (a column block of an R x L matrix, zero padded)
__global__ void k_cols_pack(const double* __restrict__ in, int64_t R, int L, int c0, int Lb, int B, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * B) return;
    const int64_t r = t / B;
    const int c = (int)(t - r * B);
    out[t] = c < Lb ? in[r * L + c0 + c] : 0.0;
}

__global__ void k_cols_unpack(const double* __restrict__ in, int64_t R, int B, int c0, int Lb, int L, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * Lb) return;
    const int64_t r = t / Lb;
    const int c = (int)(t - r * Lb);
    out[r * L + c0 + c] = in[r * B + c];
}

// partial Gram of a block of rows, one 32 x 32 tile (ta <= tb) per workgroup: thread (i, j) owns the 2 x 2 entries
// (2i, 2i+1) x (2j, 2j+1) of the tile; rows are staged 32 at a time; both triangles are written.  Fixed summation order.
__global__ void __launch_bounds__(256) k_gram_tiled(const double* __restrict__ X, int64_t R, int L, int64_t rows_per_block, int ntiles,
                                                    double* __restrict__ partial) {
    __shared__ double xa[32][33], xb[32][33];
    int ta = 0, rest = blockIdx.y;                 // tile pair number -> (ta, tb), ta <= tb
    while (rest >= ntiles - ta) { rest -= ntiles - ta; ++ta; }
    const int tb = ta + rest;
    const int tid = threadIdx.x, i = tid >> 4, j = tid & 15;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    for (int64_t rb = r0; rb < r1; rb += 32) {
        const int nr = (int)((r1 - rb) < 32 ? (r1 - rb) : 32);
        __syncthreads();
        for (int t = tid; t < 32 * 32; t += 256) {
            const int rr = t >> 5, cc = t & 31;
            const int ca = ta * 32 + cc, cb = tb * 32 + cc;
            xa[rr][cc] = (rr < nr && ca < L) ? X[(rb + rr) * L + ca] : 0.0;
            xb[rr][cc] = (rr < nr && cb < L) ? X[(rb + rr) * L + cb] : 0.0;
        }
        __syncthreads();
        for (int rr = 0; rr < nr; ++rr) {
            const double a0 = xa[rr][2 * i], a1 = xa[rr][2 * i + 1], b0 = xb[rr][2 * j], b1 = xb[rr][2 * j + 1];
            acc[0][0] = fma(a0, b0, acc[0][0]); acc[0][1] = fma(a0, b1, acc[0][1]);
            acc[1][0] = fma(a1, b0, acc[1][0]); acc[1][1] = fma(a1, b1, acc[1][1]);
        }
    }
    double* out = partial + (int64_t)blockIdx.x * L * L;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int a = ta * 32 + 2 * i + u, b = tb * 32 + 2 * j + v;
            if (a < L && b < L && (ta < tb || a <= b)) {
                out[a * L + b] = acc[u][v];
                out[b * L + a] = acc[u][v];
            }
        }
}

// out[R x L2] = X[R x L] * T[L x L2], one 64 x 32 tile of the result per workgroup, the inner dimension in chunks of 32
__global__ void __launch_bounds__(256) k_right_mult_tiled(const double* __restrict__ X, int64_t R, int L, const double* __restrict__ T, int L2,
                                                          double* __restrict__ out) {
    __shared__ double xs[64][33], ts[32][33];
    const int tid = threadIdx.x, row = tid >> 2, cg = tid & 3;        // 8 result columns per thread
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0;
    for (int p0 = 0; p0 < L; p0 += 32) {
        __syncthreads();
        for (int t = tid; t < 64 * 32; t += 256) {
            const int rr = t >> 5, pp = t & 31;
            xs[rr][pp] = (r0 + rr < R && p0 + pp < L) ? X[(r0 + rr) * L + p0 + pp] : 0.0;
        }
        for (int t = tid; t < 32 * 32; t += 256) {
            const int pp = t >> 5, cc = t & 31;
            ts[pp][cc] = (p0 + pp < L && c0 + cc < L2) ? T[(int64_t)(p0 + pp) * L2 + c0 + cc] : 0.0;
        }
        __syncthreads();
        for (int pp = 0; pp < 32; ++pp) {
            const double x = xs[row][pp];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = fma(x, ts[pp][cg * 8 + u], acc[u]);
        }
    }
    if (r0 + row < R)
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + cg * 8 + u < L2) out[(r0 + row) * L2 + c0 + cg * 8 + u] = acc[u];
}

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
struct PcaWork {
    ddx_ctx* ctx;
    int L, lpn, slots;
    bool gather32;     // gather a float32 copy of the small operand (needs L % 4 == 0)
    float* op32;       // scratch for that copy: max(M, H) x L floats
    // LDS mode: two float32 operand copies live side by side (op32 = copy of an M-row sketch, opQ = copy of an H-row
    // one), each written by the kernel that produces the float64 matrix it mirrors (`*_of`); k_operand_copy only runs
    // when a product is asked for a matrix nobody mirrored.
    float* opQ = nullptr;
    const double* opY_of = nullptr;
    const double* opQ_of = nullptr;
    int64_t M;
    int32_t H;
    bool lds;          // LDS-staged products (needs gather32 and a slice that fits the LDS)
    bool bitplane = false;   // the entries equal to 1 go through the bit-plane products (k_bitplane.hip)
    bool fused_ymax = true;  // the sparse A Q kernel's column maxima may serve the A^T Y product that follows (off where row-side matrices are rewritten in place)
    const double* uvec_of = nullptr;   // the row-side matrix whose column sums the packed A Q kernel has just left in `small + 3 L^2 + L`
    const double* tvec_of = nullptr;   // the column-side matrix whose m^T Q the Cholesky-QR's right multiplication has just left in `small + 3 L^2`
    bool fused_qmax = false; // the Cholesky-QR's right multiplication leaves the column maxima of its result for the A Q product that follows (stage_pca's power
                             // iterations only: nothing rewrites the basis between the two)
    int rows_SR = 0, rows_ns = 0;   // A Q: slice height / slice count of the H-row operand
    double* partial;   // scratch for block partials
    double* small;     // [4*L*L + 4*L]: G, Rinv, T, vecs
    int* flag;
    // sketches wider than kMaxL: the products run on column blocks through `sub` (width kWideBlock)
    PcaWork* sub = nullptr;
    double* blkRow = nullptr;     // [M x kWideBlock]
    double* blkCol = nullptr;     // [H x kWideBlock]
};

constexpr int kCounterColsum = 8, kCounterGram = 9, kCounterUsum = 10;       // tickets of the last-block reductions (ints of PcaWork::flag's buffer, zero between launches)

static int wcolsum(PcaWork& w, const double* X, int64_t R, const double* wgt, double* out) {
    int nb = (int)std::min<int64_t>(512, ceil_div(R, 256));
    int64_t rpb = ceil_div(R, nb);
    nb = (int)ceil_div(R, rpb);
    k_wcolsum_partial<<<nb, 256, 0, w.ctx->stream>>>(X, R, w.L, wgt, rpb, w.partial, w.flag + kCounterColsum, out);
    return DDX_OK;
}

// G = X^T X; chol_to != nullptr (only honoured at the default sketch width 40): also Rinv = inverse of G's Cholesky factor, same launch
static int gram(PcaWork& w, const double* X, int64_t R, double* G, double* chol_to = nullptr) {
    if (w.L > kMaxL) {
        int nbw = (int)std::min<int64_t>(128, ceil_div(R, 256));
        const int64_t rpbw = ceil_div(R, nbw);
        nbw = (int)ceil_div(R, rpbw);
        const int nt = (w.L + 31) / 32;
        k_gram_tiled<<<dim3((unsigned)nbw, (unsigned)(nt * (nt + 1) / 2)), 256, 0, w.ctx->stream>>>(X, R, w.L, rpbw, nt, w.partial);
        k_reduce_partials<<<(unsigned)ceil_div(w.L * w.L, 4), 256, 0, w.ctx->stream>>>(w.partial, nbw, w.L * w.L, G);
        return DDX_OK;
    }
    int nb = (int)std::min<int64_t>(512, ceil_div(R, 128));      // pcaPartial holds 512 partial Gram matrices
    int64_t rpb = ceil_div(R, nb);
    nb = (int)ceil_div(R, rpb);
    const int LL = w.L * w.L;
    if (w.L * (w.L + 1) / 2 <= 4 * 256) k_gram_partial<4><<<nb, 256, sizeof(double) * 32 * w.L, w.ctx->stream>>>(X, R, w.L, rpb, w.partial);
    else k_gram_partial<(kMaxL * (kMaxL + 1) / 2 + 255) / 256><<<nb, 256, sizeof(double) * 32 * w.L, w.ctx->stream>>>(X, R, w.L, rpb, w.partial);
    if (chol_to && w.L == 40)
        k_reduce_partials<<<(unsigned)ceil_div(LL, 4), 256, 0, w.ctx->stream>>>(w.partial, nb, LL, G, w.flag + kCounterGram, chol_to, w.flag);
    else
        k_reduce_partials<<<(unsigned)ceil_div(LL, 4), 256, 0, w.ctx->stream>>>(w.partial, nb, LL, G);
    return DDX_OK;
}

static int right_mult(ddx_ctx* ctx, const double* X, int64_t R, int L, const double* T, int L2, double* out) {
    if ((size_t)(L * L2 + 64 * L) * sizeof(double) <= 64 * 1024)
        k_right_mult<<<(unsigned)ceil_div(R, 64), 256, sizeof(double) * (L * L2 + 64 * L), ctx->stream>>>(X, R, L, T, L2, out);
    else
        k_right_mult_tiled<<<dim3((unsigned)ceil_div(R, 64), (unsigned)ceil_div(L2, 32)), 256, 0, ctx->stream>>>(X, R, L, T, L2, out);
    return DDX_OK;
}

// Cholesky factor and its inverse on the host (sketches wider than one wave handles): same pivot floor as k_chol_inv
static void chol_inverse_host(int L, std::vector<double>& a, std::vector<double>& inv, int* flag) {
    double maxd = 0.0;
    for (int k = 0; k < L; ++k) maxd = std::max(maxd, a[(size_t)k * L + k]);
    const double floor_v = maxd * 1e-26 + 1e-300;
    for (int k = 0; k < L; ++k) {
        double d = a[(size_t)k * L + k];
        if (!(d > floor_v)) { d = floor_v; *flag |= 1; }
        const double piv = std::sqrt(d);
        a[(size_t)k * L + k] = piv;
        for (int j = k + 1; j < L; ++j) a[(size_t)k * L + j] /= piv;
        for (int i = k + 1; i < L; ++i) {
            const double rki = a[(size_t)k * L + i];
            for (int j = i; j < L; ++j) a[(size_t)i * L + j] -= rki * a[(size_t)k * L + j];
        }
    }
    inv.assign((size_t)L * L, 0.0);
    for (int j = 0; j < L; ++j) {
        inv[(size_t)j * L + j] = 1.0 / a[(size_t)j * L + j];
        for (int i = j - 1; i >= 0; --i) {
            double s = 0.0;
            for (int p = i + 1; p <= j; ++p) s += a[(size_t)i * L + p] * inv[(size_t)p * L + j];
            inv[(size_t)i * L + j] = -s / a[(size_t)i * L + i];
        }
    }
}

// X <- X R^-1 with X^T X = R^T R  (one Cholesky-QR pass); result lands in `out`
static int cholqr(PcaWork& w, const double* X, int64_t R, double* out) {
    double* G = w.small;
    double* Rinv = w.small + w.L * w.L;
    if (w.ctx->bp.ymax_of == out) w.ctx->bp.ymax_of = nullptr;      // (the matrix whose column maxima were taken is overwritten)
    ScopedTimer t(w.ctx, "pca_orth");
    DDX_TRY(gram(w, X, R, G, w.L == 40 ? Rinv : nullptr));
    if (w.L > kMaxL) {
        const int L = w.L;
        std::vector<double> hG((size_t)L * L), hInv;
        DDX_HIP(w.ctx, hipMemcpyAsync(hG.data(), G, sizeof(double) * L * L, hipMemcpyDeviceToHost, w.ctx->stream));
        DDX_HIP(w.ctx, wait_stream(w.ctx));
        int hflag = 0;
        chol_inverse_host(L, hG, hInv, &hflag);
        if (hflag) DDX_HIP(w.ctx, hipMemcpyAsync(w.flag, &hflag, sizeof(int), hipMemcpyHostToDevice, w.ctx->stream));
        DDX_HIP(w.ctx, hipMemcpyAsync(Rinv, hInv.data(), sizeof(double) * L * L, hipMemcpyHostToDevice, w.ctx->stream));
        DDX_TRY(right_mult(w.ctx, X, R, L, Rinv, L, out));
        DDX_HIP(w.ctx, wait_stream(w.ctx));       // hInv / hflag are stack-backed
        return DDX_OK;
    }
    // (sketch width 40, the default 30 + 10: factorised by the Gram kernel's last block, in registers)
    if (w.L != 40) k_chol_inv<<<1, 64, 2 * sizeof(double) * w.L * w.L, w.ctx->stream>>>(G, w.L, Rinv, w.flag);
    float* out32 = nullptr;
    const int ld = (w.L + 3) & ~3;
    if (w.lds && w.opQ) {
        if (w.opY_of == out) w.opY_of = nullptr;
        if (w.opQ_of == out) w.opQ_of = nullptr;
        if (R == w.H) { out32 = w.opQ; w.opQ_of = out; }
        else if (R == w.M) { out32 = w.op32; w.opY_of = out; }
    }
    // bit-plane route: the column maxima of the next A Q operand (diag(1 / sd) Q on a scaled matrix) fall out of this kernel -- into the
    // slots the last Y-side digit kernel zeroed (k_bp_digits: zero_me)
    BitPlanes& bp = w.ctx->bp;
    const bool fuse = w.fused_qmax && R == w.H && w.L <= 64;      // `out` is the operand of the next A Q product
    const bool qmax = fuse && w.bitplane && bp.cmax && bp.qmax_zeroed;
    if (bp.qmax_of == out) bp.qmax_of = nullptr;
    if (w.tvec_of == out) w.tvec_of = nullptr;
    if (w.uvec_of == out) w.uvec_of = nullptr;
    k_right_mult<<<(unsigned)ceil_div(R, 64), 256, sizeof(double) * (w.L * w.L + 64 * w.L + (fuse ? 64 * w.L : 0)), w.ctx->stream>>>(
        X, R, w.L, Rinv, w.L, out, out32, ld, qmax ? (bp.scaled ? bp.inv_sd : nullptr) : nullptr, qmax ? reinterpret_cast<unsigned long long*>(bp.cmax) : nullptr,
        fuse ? w.ctx->colmean.as<double>() : nullptr, w.partial, w.flag + kCounterColsum, w.small + 3 * w.L * w.L);
    if (fuse) w.tvec_of = out;
    if (qmax) { bp.qmax_of = out; bp.qmax_zeroed = false; }
    return DDX_OK;
}

// operand preparation: returns the pointer/leading dimension the gather kernels should use
template <typename T>
static const T* prepared_operand(PcaWork& w, const double* X, int64_t R, int ld) {
    T* out = reinterpret_cast<T*>(w.op32);
    k_operand_copy<T><<<(unsigned)ceil_div(R * ld, 256), 256, 0, w.ctx->stream>>>(X, R, w.L, ld, out);
    return out;
}

// lane-group geometry of the LDS kernels for a padded sketch width ld (0 slots: not applicable).
// "pair": two columns per lane, groups of ld/2 lanes (ld <= 42) or 16 lanes (ld <= 32); "quad": four columns per lane,
// four groups of 16 lanes, any ld <= 64.  At ld = 40 the two run at the same speed (0.67-0.71 ms; the quad geometry has
// no bank conflicts and half the staged-entry reads per entry, but neither the LDS nor the VALU is what bounds the
// kernel), so pair stays the default where it applies and quad serves the wider sketches.  DDX_SPMM_GEOM=pair|quad
// forces one of them.
// switches of the context whose stage is running on this thread (read from the environment once, at ddx_create)
static thread_local const Options* t_opt = nullptr;
static bool lds_packed();
static bool lds_quad(int ld) {
    const int mode = t_opt->spmm_geom;
    if (!lds_packed() || mode == 1) return false;       // (the float64-product trips keep the pair geometry: twice the staging)
    return mode == 2 || ld > 42;
}
constexpr int kLdsOwnQuad = 4;     // outputs owned by one lane group in the quad geometry (4 x 4 float64 accumulators)
static int lds_slots(int ld) { return lds_quad(ld) ? (ld <= 64 ? 4 : 0) : (ld <= 32 ? 4 : (ld <= 42 ? 3 : 0)); }
static int lds_lpn(int ld) { return (lds_quad(ld) || ld <= 32) ? 16 : ld / 2; }
// outputs per lane group of the A^T Y kernel in bit-plane mode: the mirror it walks holds a tenth of the entries, and what is left of
// its time is staging the 123 KB operand slices -- twice the columns per workgroup halve the slice loads (0.272 -> 0.230 ms per launch
// at the headline; the A Q kernel, whose workgroups each stage the whole operand anyway, does not gain: 0.226 -> 0.227)
constexpr int kLdsOwnSparseCols = 12;
static int lds_own(int ld, bool sparse_cols) { return lds_quad(ld) ? kLdsOwnQuad : ((sparse_cols && lds_packed() && lds_slots(ld) == 3) ? kLdsOwnSparseCols : kLdsOwnG); }
static int lds_owners(int64_t nOut, int slots, int ld, bool sparse_cols = false) {
    const int64_t need = ceil_div(nOut, (int64_t)kLdsWaves * slots * lds_own(ld, sparse_cols));
    constexpr int64_t per_round = 256 * kLdsWgPerCu;
    return (int)(need <= per_round ? need : per_round * ceil_div(need, per_round));     // whole rounds of resident workgroups
}

template <bool ROWS, int SLOTS, bool PK, int CPL, int OWN>
static int launch_lds_t(ddx_ctx* c, const LdsSpmmArgs& a, unsigned grid, size_t lds_bytes) {
    DDX_TRY(allow_dynamic_lds(c, reinterpret_cast<const void*>(&k_spmm_lds<ROWS, SLOTS, PK, CPL, OWN>), (int)kLdsBudget));
    k_spmm_lds<ROWS, SLOTS, PK, CPL, OWN><<<grid, kLdsThreads, lds_bytes, c->stream>>>(a);
    DDX_HIP(c, hipGetLastError());                // (a refused launch must not leave the iterate as it was)
    return DDX_OK;
}

// Default: the eight products of a trip in packed float32, trip sums added in float64 (profiles/tools/spmm_precision.py:
// 7e-7 per component against the all-float64 run at 50k x 20k, 4e-7 with float64 products; 0.69 instead of 0.84 ms).
// DDX_SPMM_TRIP=f64 selects float64 products.
static bool lds_packed() { return t_opt->trip_packed; }

template <bool ROWS>
static int launch_lds(ddx_ctx* c, const LdsSpmmArgs& a, int slots, unsigned grid, size_t lds_bytes, bool sparse_cols = false) {
    if (lds_quad(a.ld)) return launch_lds_t<ROWS, 4, true, 4, kLdsOwnQuad>(c, a, grid, lds_bytes);
    if (!ROWS && lds_own(a.ld, sparse_cols) == kLdsOwnSparseCols) return launch_lds_t<ROWS, 3, true, 2, kLdsOwnSparseCols>(c, a, grid, lds_bytes);
    if (lds_packed())
        return slots == 4 ? launch_lds_t<ROWS, 4, true, 2, kLdsOwnG>(c, a, grid, lds_bytes) : launch_lds_t<ROWS, 3, true, 2, kLdsOwnG>(c, a, grid, lds_bytes);
    return slots == 4 ? launch_lds_t<ROWS, 4, false, 2, kLdsOwnG>(c, a, grid, lds_bytes) : launch_lds_t<ROWS, 3, false, 2, kLdsOwnG>(c, a, grid, lds_bytes);
}

// packed residual products (bit-plane mode): the wave-ordered blocks of this iteration's reduced matrix, built at the first product that
// needs them (ctx->pk_valid is cleared by bp_refresh); see k_pack_residual / k_spmm_packed
static bool packed_applies(const ddx_ctx* c, const LdsSpmmArgs& a, int slots) {
    return c->opt.residual_packed && lds_packed() && !lds_quad(a.ld) && slots == 3 && a.ld == 40 && a.lpn == 20;
}

template <bool ROWS>
static int pack_residual(ddx_ctx* c, const LdsSpmmArgs& a, int own, int nsl, int nwg) {
    const int side = ROWS ? 0 : 1;
    ScopedTimer t(c, "residual_pack");
    const int64_t nwaves = (int64_t)nwg * kLdsWaves;
    const size_t n = (size_t)nwaves * (nsl + 1) + 1;
    DDX_TRY(ensure(c, c->pk_ptr[side], sizeof(int32_t) * 2 * n));
    int32_t* cnt = c->pk_ptr[side].as<int32_t>();
    int32_t* ptr = cnt + n;
    DDX_HIP(c, hipMemsetAsync(cnt, 0, sizeof(int32_t) * n, c->stream));
    const unsigned grid = (unsigned)ceil_div(nwaves * nsl, 4);          // one wave per (product wave, slice) pair
    k_pack_residual<ROWS><<<grid, 256, 0, c->stream>>>(a, own, nsl, nwg, cnt, nullptr, 0);
    size_t tmp = 0;
    DDX_HIP(c, prim::exclusive_sum(nullptr, tmp, cnt, ptr, n, c->stream));
    DDX_TRY(ensure(c, c->sort_tmp, tmp));
    DDX_HIP(c, prim::exclusive_sum(c->sort_tmp.p, tmp, cnt, ptr, n, c->stream));
    int32_t total = 0;
    DDX_HIP(c, hipMemcpyAsync(&total, ptr + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    DDX_HIP(c, wait_stream(c));
    DDX_TRY(ensure(c, c->pk_blocks[side], (size_t)(total + 1) * kPkBlockBytes));
    DDX_HIP(c, hipMemsetAsync(c->pk_blocks[side].p, 0, (size_t)(total + 1) * kPkBlockBytes, c->stream));
    k_pack_residual<ROWS><<<grid, 256, 0, c->stream>>>(a, own, nsl, nwg, ptr, c->pk_blocks[side].as<unsigned char>(), 1);
    DDX_HIP(c, hipGetLastError());
    c->pk_nblocks[side] = total;
    c->pk_valid[side] = true;
    return DDX_OK;
}

template <bool ROWS, int OWN, int RING>
static int launch_packed_ring(ddx_ctx* c, const PackedArgs& pa, unsigned grid, size_t lds_bytes) {
    DDX_TRY(allow_dynamic_lds(c, reinterpret_cast<const void*>(&k_spmm_packed<ROWS, OWN, RING>), (int)kLdsBudget));
    ScopedTimer t(c, ROWS ? "spmm_rows" : "spmm_cols");
    k_spmm_packed<ROWS, OWN, RING><<<grid, kLdsThreads, lds_bytes, c->stream>>>(pa);
    DDX_HIP(c, hipGetLastError());
    return DDX_OK;
}

template <bool ROWS, int OWN>
static int launch_packed(ddx_ctx* c, const LdsSpmmArgs& a, int nsl, unsigned grid) {
    const int side = ROWS ? 0 : 1;
    if (!c->pk_valid[side]) DDX_TRY((pack_residual<ROWS>(c, a, OWN, nsl, (int)grid)));
    PackedArgs pa{};
    pa.a = a; pa.own = OWN; pa.nsl = nsl;
    const size_t n = (size_t)grid * kLdsWaves * (nsl + 1) + 1;
    pa.blkptr = c->pk_ptr[side].as<int32_t>() + n;
    pa.blocks = c->pk_blocks[side].as<unsigned char>();
    // as many block buffers per wave as fit beside the operand slice: two beside the 772- / 784-row slices of the headline, three
    // beside slices of at most 716 rows
    const size_t slice_bytes = ((size_t)a.SR * a.ld * 4 + 15) & ~(size_t)15;
#ifndef DDX_PK_RING2
#define DDX_PK_RING2 0      // variant builds only: 1 = two buffers everywhere
#endif
    const bool three = !DDX_PK_RING2 && slice_bytes + (size_t)kLdsWaves * 3 * kPkBlockBytes <= (size_t)kLdsBudget;
    if (three) return launch_packed_ring<ROWS, OWN, 3>(c, pa, grid, slice_bytes + (size_t)kLdsWaves * 3 * kPkBlockBytes);
    return launch_packed_ring<ROWS, OWN, 2>(c, pa, grid, slice_bytes + (size_t)kLdsWaves * 2 * kPkBlockBytes);
}

static int apply_rows_wide(PcaWork& w, const double* Qcol, double* Yrow);
static int apply_cols_wide(PcaWork& w, const double* Yrow, double* Wcol);

static int apply_rows(PcaWork& w, const double* Qcol, double* Yrow) {  // A Q : [H x L] -> [M x L]
    if (!w.bitplane) DDX_TRY(ensure_full_rows(w.ctx));     // (the bit-plane route leaves the doublets' rows and the row-major values out)
    if (w.sub) return apply_rows_wide(w, Qcol, Yrow);
    ddx_ctx* c = w.ctx;
    double* tvec = w.small + 3 * w.L * w.L;
    if (w.uvec_of == Yrow) w.uvec_of = nullptr;
    if (w.tvec_of != Qcol) {                            // (else: left there by the kernel that produced Qcol, k_right_mult)
        ScopedTimer t(c, "pca_colsum");
        DDX_TRY(wcolsum(w, Qcol, w.H, c->colmean.as<double>(), tvec));
    }
    w.tvec_of = nullptr;
    // Y = diag(s) B Q on the matrix cores first (a timing scope of its own); the sparse kernel then sees only the entries other
    // than 1 and adds its part
    if (w.bitplane) DDX_TRY(bp_rows_product(c, Qcol, w.L, Yrow));
    const unsigned grid = (unsigned)ceil_div(w.M, 4);
    if (w.lds) {
        LdsSpmmArgs a{};
        a.ld = (w.L + 3) & ~3; a.L = w.L; a.lpn = lds_lpn(a.ld);
        const int slots = lds_slots(a.ld);
        if (w.opQ && w.opQ_of == Qcol) {
            a.op = w.opQ;                                   // mirrored by the kernel that produced Qcol
        } else if (w.opQ) {
            k_operand_copy<float><<<(unsigned)ceil_div((int64_t)w.H * a.ld, 256), 256, 0, c->stream>>>(Qcol, w.H, w.L, a.ld, w.opQ);
            w.opQ_of = Qcol;
            a.op = w.opQ;
        } else {
            a.op = prepared_operand<float>(w, Qcol, w.H, a.ld);
        }
        if (w.opQ) {                                        // this product mirrors its own result for the A^T Y pass
            if (w.opQ_of == Yrow) w.opQ_of = nullptr;
            a.out32 = w.op32;
            w.opY_of = Yrow;
        }
        a.opRows = w.H; a.SR = w.rows_SR; a.nslices = w.rows_ns; a.groups = 1;
        a.nOut = w.M; a.owners = lds_owners(w.M, slots, a.ld);
        a.indptr = c->aug_indptr.as<int64_t>(); a.cols = c->aug_indices.as<int32_t>(); a.x = c->aug_x.as<float>();
        a.zcol = c->zcol.as<float>(); a.rowseg = c->rowseg.as<int32_t>(); a.tvec = tvec; a.out = Yrow;
        a.z_uniform = c->scaled ? 0 : 1;              // unscaled: every column's unstored value is log(pseudocount)
        a.zval = c->zvalue;
        a.perm = c->rank_rows;
        a.accumulate = 0;
        if (w.bitplane) {
            a.indptr = c->bp.rest_indptr; a.cols = c->bp.rest_cols; a.x = c->bp.rest_x;
            a.accumulate = 1;
            a.srow = c->bp.srow;
            a.ymax = reinterpret_cast<unsigned long long*>(c->bp.cmax + 64);
            c->bp.ymax_of = w.fused_ymax ? Yrow : nullptr;
        }
        if (w.bitplane && packed_applies(c, a, slots)) {
            // (the column sums of Y for the A^T Y product that follows: from this kernel's epilogue)
            // (not where Yrow is a block buffer that is rewritten before the A^T Y product reads it: fused_ymax says so)
            if (w.fused_ymax) {
                a.usum = w.small + 3 * w.L * w.L + w.L; a.upart = w.partial; a.ucount = w.flag + kCounterUsum;
                w.uvec_of = Yrow;
            }
            if (c->opt.residual_rows_own == kLdsOwnSparseCols) {
                a.owners = lds_owners(w.M, slots, a.ld, true);
                return launch_packed<true, kLdsOwnSparseCols>(c, a, a.nslices, (unsigned)a.owners);
            }
            return launch_packed<true, kLdsOwnG>(c, a, a.nslices, (unsigned)a.owners);
        }
        const size_t lds_bytes = (size_t)a.SR * a.ld * 4 + (size_t)((a.SR + 3) & ~3) * 4 + lds_stage_bytes(slots, lds_packed());
        ScopedTimer t(c, "spmm_rows");
        DDX_TRY(launch_lds<true>(c, a, slots, (unsigned)a.owners, lds_bytes));
        return DDX_OK;
    }
    ScopedTimer t(c, "spmm_rows");
    if (w.gather32) {
        const int ld = (w.L + 3) & ~3, lpn = ld / 4;
        const float* op = prepared_operand<float>(w, Qcol, w.H, ld);
        if (w.bitplane)
            k_spmm_rows<float><<<grid, 256, 0, c->stream>>>(c->bp.rest_indptr, c->bp.rest_cols, c->bp.rest_x, c->zcol.as<float>(), op, ld, w.L, lpn, 64 / lpn, tvec, w.M,
                                                            Yrow, 1);
        else
        k_spmm_rows<float><<<grid, 256, 0, c->stream>>>(c->aug_indptr.as<int64_t>(), c->aug_indices.as<int32_t>(), c->aug_x.as<float>(),
                                                        c->zcol.as<float>(), op, ld, w.L, lpn, 64 / lpn, tvec, w.M, Yrow);
    } else {
        const int ld = (w.L + 1) & ~1, lpn = ld / 2;
        const double* op = (ld == w.L) ? Qcol : prepared_operand<double>(w, Qcol, w.H, ld);
        k_spmm_rows<double><<<grid, 256, 0, c->stream>>>(c->aug_indptr.as<int64_t>(), c->aug_indices.as<int32_t>(), c->aug_x.as<float>(),
                                                         c->zcol.as<float>(), op, ld, w.L, lpn, 64 / lpn, tvec, w.M, Yrow);
    }
    return DDX_OK;
}

static int apply_cols(PcaWork& w, const double* Yrow, double* Wcol) {  // A^T Y : [M x L] -> [H x L]
    if (w.sub) return apply_cols_wide(w, Yrow, Wcol);
    ddx_ctx* c = w.ctx;
    if (c->bp.qmax_of == Wcol) c->bp.qmax_of = nullptr;     // (the result overwrites a matrix whose by-products are held)
    if (w.tvec_of == Wcol) w.tvec_of = nullptr;
    double* uvec = w.small + 3 * w.L * w.L + w.L;
    if (w.uvec_of != Yrow) {                            // (else: left there by the kernel that produced Yrow, k_spmm_packed)
        ScopedTimer t(c, "pca_colsum");
        DDX_TRY(wcolsum(w, Yrow, w.M, nullptr, uvec));
    }
    if (!w.bitplane) DDX_TRY(ensure_full_mirror(c));     // (ddx_lognormalise leaves it out when the bit-plane route is expected)
    const int P = (int)ceil_div(w.M, c->panel_rows);
    const int groups = (w.H + 3) / 4;
    const int64_t grid = 8 * ceil_div(P, 8) * groups;
    const double* w1 = nullptr;                     // bit-plane part of the product: nw1 partial blocks
    int nw1 = 0;
    if (w.bitplane) DDX_TRY(bp_cols_product(c, Yrow, w.L, &w1, &nw1));
    if (w.lds) {
        LdsSpmmArgs a{};
        a.ld = (w.L + 3) & ~3; a.L = w.L; a.lpn = lds_lpn(a.ld);
        const int slots = lds_slots(a.ld);
        if (w.opQ && w.opY_of == Yrow) {
            a.op = w.op32;                                  // mirrored by the kernel that produced Yrow
        } else {
            a.op = prepared_operand<float>(w, Yrow, w.M, a.ld);
            if (w.opQ) w.opY_of = Yrow;
        }
        if (w.opQ && w.opQ_of == Wcol) w.opQ_of = nullptr;   // the result overwrites a mirrored matrix
        if (w.opQ && w.opY_of == Wcol) w.opY_of = nullptr;
        a.opRows = w.M; a.SR = c->panel_rows; a.nslices = P;
        a.nOut = w.H; a.owners = lds_owners(w.H, slots, a.ld, w.bitplane);
        a.groups = std::max(1, std::min(P, 512 * kLdsWgPerCu / a.owners));
        a.zcol = c->zcol.as<float>();
        a.cp_o = c->csc_o_colptr.as<int64_t>(); a.row_o = c->csc_o_row.as<int32_t>(); a.x_o = c->csc_o_x.as<float>(); a.P_o = c->P_o;
        a.cp_s = c->csc_s_colptr.as<int64_t>(); a.row_s = c->csc_s_row.as<int32_t>(); a.x_s = c->csc_s_x.as<float>();
        if (w.bitplane) {                               // the reduced mirrors: entries other than 1
            a.cp_o = c->bp.restm_colptr; a.row_o = c->bp.restm_row; a.x_o = c->bp.restm_x;
            if (c->P_s > 0) { a.cp_s = c->bp.restm_s_colptr; a.row_s = c->bp.restm_s_row; a.x_s = c->bp.restm_s_x; }
        }
        a.p_s0 = c->p_s0; a.P_s = c->P_s;
        a.out = c->pcaPanel.as<double>();
        a.perm = c->rank_cols;
        if (w.bitplane && packed_applies(c, a, slots) && lds_own(a.ld, true) == kLdsOwnSparseCols) {
            DDX_TRY((launch_packed<false, kLdsOwnSparseCols>(c, a, (int)ceil_div(P, a.groups), (unsigned)(a.owners * a.groups))));
        } else {
            const size_t lds_bytes = (size_t)a.SR * a.ld * 4 + lds_stage_bytes(slots, lds_packed());
            ScopedTimer t(c, "spmm_cols");
            DDX_TRY(launch_lds<false>(c, a, slots, (unsigned)(a.owners * a.groups), lds_bytes, w.bitplane));
        }
        ScopedTimer t(c, "spmm_sum");
        k_sum_panels<<<(unsigned)ceil_div((int64_t)w.H * w.L, 256), 256, 0, c->stream>>>(c->pcaPanel.as<double>(), a.groups, w.H, w.L, c->colmean.as<double>(),
                                                                                          uvec, Wcol, w1, nw1, (w.bitplane && c->bp.scaled) ? c->bp.inv_sd : nullptr);
        return DDX_OK;
    }
    ScopedTimer t(c, "spmm_cols");
    if (w.gather32) {
        const int ld = (w.L + 3) & ~3, lpn = ld / 4;
        const float* op = prepared_operand<float>(w, Yrow, w.M, ld);
        if (w.bitplane)
            k_spmm_cols<float><<<(unsigned)grid, 256, 0, c->stream>>>(c->bp.restm_colptr, c->bp.restm_row, c->bp.restm_x, c->P_o,
                                                                  c->P_s > 0 ? c->bp.restm_s_colptr : c->csc_s_colptr.as<int64_t>(), c->P_s > 0 ? c->bp.restm_s_row : c->csc_s_row.as<int32_t>(),
                                                                  c->P_s > 0 ? c->bp.restm_s_x : c->csc_s_x.as<float>(), c->p_s0,
                                                                  c->P_s, P, w.H, c->zcol.as<float>(), op, ld, w.L, lpn, 64 / lpn, c->pcaPanel.as<double>());
        else
        k_spmm_cols<float><<<(unsigned)grid, 256, 0, c->stream>>>(c->csc_o_colptr.as<int64_t>(), c->csc_o_row.as<int32_t>(), c->csc_o_x.as<float>(), c->P_o,
                                                              c->csc_s_colptr.as<int64_t>(), c->csc_s_row.as<int32_t>(), c->csc_s_x.as<float>(), c->p_s0,
                                                              c->P_s, P, w.H, c->zcol.as<float>(), op, ld, w.L, lpn, 64 / lpn, c->pcaPanel.as<double>());
    } else {
        const int ld = (w.L + 1) & ~1, lpn = ld / 2;
        const double* op = (ld == w.L) ? Yrow : prepared_operand<double>(w, Yrow, w.M, ld);
        k_spmm_cols<double><<<(unsigned)grid, 256, 0, c->stream>>>(c->csc_o_colptr.as<int64_t>(), c->csc_o_row.as<int32_t>(), c->csc_o_x.as<float>(), c->P_o,
                                                               c->csc_s_colptr.as<int64_t>(), c->csc_s_row.as<int32_t>(), c->csc_s_x.as<float>(), c->p_s0,
                                                               c->P_s, P, w.H, c->zcol.as<float>(), op, ld, w.L, lpn, 64 / lpn, c->pcaPanel.as<double>());
    }
    k_sum_panels<<<(unsigned)ceil_div((int64_t)w.H * w.L, 256), 256, 0, c->stream>>>(c->pcaPanel.as<double>(), P, w.H, w.L, c->colmean.as<double>(),
                                                                                      uvec, Wcol, w1, nw1, (w.bitplane && c->bp.scaled) ? c->bp.inv_sd : nullptr);
    return DDX_OK;
}

static int apply_rows_wide(PcaWork& w, const double* Qcol, double* Yrow) {
    hipStream_t st = w.ctx->stream;
    for (int c0 = 0; c0 < w.L; c0 += kWideBlock) {
        const int Lb = std::min(kWideBlock, w.L - c0);
        k_cols_pack<<<(unsigned)ceil_div((int64_t)w.H * kWideBlock, 256), 256, 0, st>>>(Qcol, w.H, w.L, c0, Lb, kWideBlock, w.blkCol);
        DDX_TRY(apply_rows(*w.sub, w.blkCol, w.blkRow));
        k_cols_unpack<<<(unsigned)ceil_div(w.M * Lb, 256), 256, 0, st>>>(w.blkRow, w.M, kWideBlock, c0, Lb, w.L, Yrow);
    }
    return DDX_OK;
}

static int apply_cols_wide(PcaWork& w, const double* Yrow, double* Wcol) {
    hipStream_t st = w.ctx->stream;
    for (int c0 = 0; c0 < w.L; c0 += kWideBlock) {
        const int Lb = std::min(kWideBlock, w.L - c0);
        k_cols_pack<<<(unsigned)ceil_div(w.M * kWideBlock, 256), 256, 0, st>>>(Yrow, w.M, w.L, c0, Lb, kWideBlock, w.blkRow);
        DDX_TRY(apply_cols(*w.sub, w.blkRow, w.blkCol));
        k_cols_unpack<<<(unsigned)ceil_div((int64_t)w.H * Lb, 256), 256, 0, st>>>(w.blkCol, w.H, kWideBlock, c0, Lb, w.L, Wcol);
    }
    return DDX_OK;
}

// decide whether the LDS-staged products apply and build the row-segment table of the A Q pass
// would the bit-plane products serve a sketch of L columns on this context's matrix (see bp_setup)
static bool bp_possible(const ddx_ctx* ctx, int L, bool gather32) {
    return ctx->opt.bitplane != 0 && gather32 && L <= 40 && ctx->have_lognorm && ctx->N >= 32 && ctx->S <= ctx->N / 2 &&
           (ctx->opt.bitplane == 2 || ctx->N >= 4096) && (!ctx->scaled || (ctx->bp.ready && ctx->bp.values && ctx->bp.scaled));
}

static int bp_setup(ddx_ctx* ctx, int L, PcaWork& w) {
    // bit planes: a sketch whose digits fit the kernel's tiles (40 columns), and -- unless forced -- a matrix large enough for the dense
    // passes to pay.  A scaled matrix takes the route when ddx_scale scaled the bit-plane structures (bp_scale: it does whenever they hold
    // the iteration's values); one scaled on the full arrays keeps the plain products.
    // Wider sketches reach this with L = kWideBlock: stage_pca runs their products block by block (apply_*_wide).
    w.bitplane = bp_possible(ctx, L, w.gather32);
    if (w.bitplane) {
        const bool fresh = !ctx->bp.ready;
        DDX_TRY(bp_build(ctx));
        DDX_TRY(bp_refresh(ctx));
        if (fresh) ctx->rowseg_rows = -1;              // the original rows' segments now refer to the reduced rows
    }
    return DDX_OK;
}

static int lds_setup(ddx_ctx* ctx, int L, PcaWork& w) {
    w.lds = false;
    w.bitplane = false;
    if (w.gather32 && !ctx->opt.spmm_lds) return bp_setup(ctx, L, w);
    if (!(w.gather32 && ctx->opt.spmm_lds)) return DDX_OK;
    const int64_t M = ctx->M;
    const int32_t H = ctx->H;
    const int ld = (L + 3) & ~3;
    const int slots = lds_slots(ld);
    if (!slots) return DDX_OK;
    const bool cols_fit = (size_t)ctx->panel_rows * ld * 4 + lds_stage_bytes(slots, lds_packed()) <= (size_t)kLdsBudget;
    int srmax = ((kLdsBudget - lds_stage_bytes(slots, lds_packed())) / (ld * 4 + 4)) & ~3;
    if (srmax > kLdsPanelRows) srmax = kLdsPanelRows & ~3;      // same slice height in both passes
    if (!cols_fit || srmax < 64) return DDX_OK;
    w.lds = true;
    w.rows_ns = (int)ceil_div(H, srmax);
    w.rows_SR = (int)((ceil_div(H, w.rows_ns) + 3) & ~3);
    DDX_TRY(bp_setup(ctx, L, w));
    const void* before = ctx->rowseg.p;
    DDX_TRY(ensure(ctx, ctx->rowseg, sizeof(int32_t) * (size_t)M * (w.rows_ns + 1)));
    ScopedTimer t(ctx, "row_segments");
    if (!w.bitplane) DDX_TRY(ensure_full_mirror(ctx));
    const int64_t* row_ip = w.bitplane ? ctx->bp.rest_indptr : ctx->aug_indptr.as<int64_t>();
    const int32_t* row_ix = w.bitplane ? ctx->bp.rest_cols : ctx->aug_indices.as<int32_t>();
    DDX_TRY(stage_rankings(ctx, row_ip, w.bitplane ? ctx->bp.restm_colptr : ctx->csc_o_colptr.as<int64_t>(),
                           (w.bitplane && ctx->P_s > 0) ? ctx->bp.restm_s_colptr : ctx->csc_s_colptr.as<int64_t>()));
    // the original cells' rows (columns fixed for the whole fit) are cut once; every iteration cuts its synthetic rows
    const bool kept = before && before == ctx->rowseg.p && ctx->rowseg_rows == ctx->N && ctx->rowseg_ns == w.rows_ns && ctx->rowseg_SR == w.rows_SR &&
                      ctx->bp_rowseg == w.bitplane;
    const int64_t N = ctx->N;
    if (!kept && N > 0)
        k_row_segments<<<(unsigned)ceil_div(N * (w.rows_ns + 1), 256), 256, 0, ctx->stream>>>(row_ip, row_ix, 0, N, w.rows_ns, w.rows_SR, ctx->rowseg.as<int32_t>());
    if (M > N)
        k_row_segments<<<(unsigned)ceil_div((M - N) * (w.rows_ns + 1), 256), 256, 0, ctx->stream>>>(row_ip, row_ix, N, M, w.rows_ns, w.rows_SR, ctx->rowseg.as<int32_t>());
    ctx->rowseg_rows = ctx->N; ctx->rowseg_ns = w.rows_ns; ctx->rowseg_SR = w.rows_SR; ctx->bp_rowseg = w.bitplane;
    return DDX_OK;
}

static int pca_work_init(ddx_ctx* ctx, int L, PcaWork& w) {
    const int64_t M = ctx->M;
    const int32_t H = ctx->H;
    DDX_TRY(ensure(ctx, ctx->pcaSmall, sizeof(double) * (4 * L * L + 4 * L) + 256));
    DDX_TRY(ensure(ctx, ctx->pcaPartial, sizeof(double) * 512 * (size_t)std::max(L * L, 128)));
    DDX_TRY(ensure(ctx, ctx->pcaVec, 256));
    DDX_TRY(ensure(ctx, ctx->pcaPanel, sizeof(double) * (size_t)ceil_div(M, ctx->panel_rows) * H * L));
    w.ctx = ctx;
    w.L = L;
    w.lpn = (L + 1) / 2;
    w.slots = 64 / w.lpn;
    w.M = M;
    w.H = H;
    w.partial = ctx->pcaPartial.as<double>();
    w.small = ctx->pcaSmall.as<double>();
    w.flag = ctx->pcaVec.as<int>();
    DDX_HIP(ctx, hipMemsetAsync(w.flag, 0, 16 * sizeof(int), ctx->stream));       // rank flag + the tickets of the last-block reductions
    w.gather32 = ctx->opt.gather_f32;
    w.lds = false;
    const int64_t maxR = M > H ? M : (int64_t)H;
    DDX_TRY(ensure(ctx, ctx->pcaOp, sizeof(double) * (size_t)maxR * (L + 4)));
    w.op32 = ctx->pcaOp.as<float>();
    return DDX_OK;
}

// mode 0: out[M x n] = A X        (X: H x n)        mode 2: out[H x n] = A^T (A X)   (X: H x n)
// mode 1: out[H x n] = A^T X      (X: M x n)        mode 3: out[M x n] = A (A^T X)   (X: M x n)
// for a caller-supplied block of n <= 64 vectors.  Always gathers float64: this entry point serves the
// exact-PCA regimes, where the Gram matrix is formed column block by column block.
int stage_operator_apply(ddx_ctx* ctx, int32_t mode, const double* X, int32_t n, double* out) {
    t_opt = &ctx->opt;
    const int64_t M = ctx->M;
    const int32_t H = ctx->H;
    PcaWork w;
    DDX_TRY(pca_work_init(ctx, n, w));
    w.gather32 = false;
    DDX_TRY(ensure(ctx, ctx->pcaA, sizeof(double) * 2 * (size_t)M * n));
    DDX_TRY(ensure(ctx, ctx->pcaB, sizeof(double) * 2 * (size_t)H * n));
    double* rowA = ctx->pcaA.as<double>();
    double* rowB = rowA + (size_t)M * n;
    double* colA = ctx->pcaB.as<double>();
    double* colB = colA + (size_t)H * n;
    const bool in_rows = (mode == 1 || mode == 3);        // input lives on the row side (M x n)
    const bool out_rows = (mode == 0 || mode == 3);
    double* in_d = in_rows ? rowA : colA;
    DDX_HIP(ctx, hipMemcpyAsync(in_d, X, sizeof(double) * (size_t)(in_rows ? M : (int64_t)H) * n, hipMemcpyHostToDevice, ctx->stream));
    double* res = nullptr;
    switch (mode) {
        case 0: DDX_TRY(apply_rows(w, colA, rowA)); res = rowA; break;
        case 1: DDX_TRY(apply_cols(w, rowA, colA)); res = colA; break;
        case 2: DDX_TRY(apply_rows(w, colA, rowA)); DDX_TRY(apply_cols(w, rowA, colB)); res = colB; break;
        case 3: DDX_TRY(apply_cols(w, rowA, colA)); DDX_TRY(apply_rows(w, colA, rowB)); res = rowB; break;
        default: return set_err(ctx, DDX_E_ARG, "operator mode must be 0..3");
    }
    DDX_HIP(ctx, hipMemcpyAsync(out, res, sizeof(double) * (size_t)(out_rows ? M : (int64_t)H) * n, hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));
    DDX_HIP(ctx, hipGetLastError());
    return DDX_OK;
}

int stage_pca(ddx_ctx* ctx, int32_t C, int32_t oversample, int32_t n_iter, const double* q0, int64_t q0_rows) {
    t_opt = &ctx->opt;
    const int L = C + oversample;
    const bool tall_tiled = L > kMaxL;   // the tall-skinny helpers run tiled, the Cholesky factor on the host (gram / cholqr test w.L themselves)
    // the products run on column blocks of kWideBlock: always beyond kMaxL columns; from 41 columns on when the bit planes can serve the
    // blocks (two 40-column block products on the matrix cores + the packed residue beat one plain product of the quad geometry)
    const bool wide = tall_tiled || (L > kWideBlock && bp_possible(ctx, kWideBlock, ctx->opt.gather_f32) && ctx->opt.spmm_lds);
    const int64_t M = ctx->M;
    const int32_t H = ctx->H;
    const bool transposed = M < H;
    const int64_t mn = transposed ? M : (int64_t)H;
    if (n_iter < 0) n_iter = ((double)C < 0.1 * (double)mn) ? 7 : 4;

    // row-side (M x L) and column-side (H x L) buffers, two of each (ping/pong)
    DDX_TRY(ensure(ctx, ctx->pcaA, sizeof(double) * 2 * (size_t)M * L));
    DDX_TRY(ensure(ctx, ctx->pcaB, sizeof(double) * 2 * (size_t)H * L));
    constexpr int kSubSmall = 4 * kWideBlock * kWideBlock + 4 * kWideBlock;      // the block work's own small area behind the main one
    DDX_TRY(ensure(ctx, ctx->pcaSmall, sizeof(double) * (4 * (size_t)L * L + 4 * L + kSubSmall) + 256));
    DDX_TRY(ensure(ctx, ctx->pcaPartial, sizeof(double) * (tall_tiled ? 128 : 512) * (size_t)std::max(L * L, 128 * 4)));
    DDX_TRY(ensure(ctx, ctx->pcaVec, 256));
    DDX_TRY(ensure(ctx, ctx->pcaPanel, sizeof(double) * (size_t)ceil_div(M, ctx->panel_rows) * H * (wide ? kWideBlock : L)));
    if (wide) DDX_TRY(ensure(ctx, ctx->pcaBlk, sizeof(double) * (size_t)(M + H) * kWideBlock));
    DDX_TRY(ensure(ctx, ctx->emb64, sizeof(double) * (size_t)M * C));
    DDX_TRY(ensure(ctx, ctx->emb32, sizeof(float) * (size_t)M * C));
    DDX_TRY(ensure(ctx, ctx->sing, sizeof(double) * L));
    double* rowA = ctx->pcaA.as<double>();
    double* rowB = rowA + (size_t)M * L;
    double* colA = ctx->pcaB.as<double>();
    double* colB = colA + (size_t)H * L;

    PcaWork w;
    w.ctx = ctx;
    w.L = L;
    w.lpn = (L + 1) / 2;
    w.slots = 64 / w.lpn;
    {
        // DDX_PCA_GATHER=f64 gathers the float64 iterates themselves; the default gathers a float32-rounded
        // copy (float64 products and sums), which moves half the bytes through L2
        w.gather32 = ctx->opt.gather_f32;
    }
    w.M = M;
    w.H = H;
    PcaWork wsub;
    if (wide) {
        w.lds = false;
        wsub = w;
        wsub.L = kWideBlock;
        wsub.lpn = (kWideBlock + 1) / 2;
        wsub.slots = 64 / wsub.lpn;
        wsub.fused_ymax = false;                              // (the block buffers are rewritten between the two products of a power iteration)
        DDX_TRY(lds_setup(ctx, kWideBlock, wsub));
    } else {
        DDX_TRY(lds_setup(ctx, L, w));
        w.fused_qmax = true;
    }
    const int64_t maxR = M > H ? M : (int64_t)H;
    DDX_TRY(ensure(ctx, ctx->pcaOp, sizeof(double) * (size_t)maxR * (L + 4)));
    w.op32 = ctx->pcaOp.as<float>();
    if (w.lds) {
        // second operand copy behind the first (pcaOp holds max(M, H) x (L + 4) doubles = twice that many floats)
        w.opQ = w.op32 + (size_t)maxR * (L + 4);
        w.opQ_of = w.opY_of = nullptr;
        if (((L + 3) & ~3) != L)             // padding columns of the mirrored copies stay zero
            DDX_HIP(ctx, hipMemsetAsync(w.op32, 0, sizeof(double) * (size_t)maxR * (L + 4), ctx->stream));
    }
    w.M = M;
    w.H = H;
    w.partial = ctx->pcaPartial.as<double>();
    w.small = ctx->pcaSmall.as<double>();
    w.flag = ctx->pcaVec.as<int>();
    DDX_HIP(ctx, hipMemsetAsync(w.flag, 0, 16 * sizeof(int), ctx->stream));       // rank flag + the tickets of the last-block reductions
    if (wide) {
        wsub.op32 = w.op32;
        wsub.opQ = nullptr;                                   // (block products copy their operand each time)
        wsub.partial = w.partial;
        wsub.small = w.small + 4 * (size_t)L * L + 4 * L;
        wsub.flag = w.flag;
        w.sub = &wsub;
        w.blkRow = ctx->pcaBlk.as<double>();
        w.blkCol = w.blkRow + (size_t)M * kWideBlock;
    }

    // the start matrix stays on the device: the boosting iterations of a fit all use the same seeded draw
    DDX_TRY(ensure(ctx, ctx->pcaQ0, sizeof(double) * (size_t)q0_rows * L));
    if (q0) {
        DDX_HIP(ctx, hipMemcpyAsync(ctx->pcaQ0.p, q0, sizeof(double) * (size_t)q0_rows * L, hipMemcpyHostToDevice, ctx->stream));
        ctx->q0_rows = q0_rows;
        ctx->q0_cols = L;
    }
    // Bit-plane products of the power iterations before the last one may carry fewer digits (option bp_digits_early, off by default): what
    // such an iteration loses is a perturbation of the subspace the following iterations start from -- the signal components forget it, the
    // unconverged trailing components carry it to the end (2.3e-6 instead of 3.7e-7 at configs[1] with three digits: inside the 1e-5 bar,
    // but 0.65 % of the community labels of the 8192-cell whole-fit test then differ from the float64 oracle's).  The last iteration and the
    // projection always run at full width.
    struct EarlyDigits {
        ddx_ctx* c;
        ~EarlyDigits() { c->bp.nd_now = 0; }
        int early() const { return c->opt.bp_digits_early; }
        void set(bool on) { c->bp.nd_now = (on && early() && early() < c->opt.bp_digits) ? early() : 0; }
    } early{ctx};
    double* Qfinal;    // orthonormal basis (M x L normal branch, H x L transposed branch)
    double* Bt;        // projection on the other side (H x L normal, M x L transposed)
    int64_t RQ, RB;
    if (!transposed) {
        DDX_HIP(ctx, hipMemcpyAsync(colA, ctx->pcaQ0.p, sizeof(double) * (size_t)H * L, hipMemcpyDeviceToDevice, ctx->stream));
        // one normalisation per power iteration: orth(A^T orth(A Q)) and orth(A^T A Q) span the same
        // subspace, and in float64 a single step of A^T A (condition (s1/s40)^2) loses nothing measurable
        // (scores agree with the LU-per-half-step evaluation to 1e-12)
        for (int it = 0; it < n_iter; ++it) {
            early.set(it + 1 < n_iter);
            DDX_TRY(apply_rows(w, colA, rowA));
            DDX_TRY(apply_cols(w, rowA, colB));
            DDX_TRY(cholqr(w, colB, H, colA));
        }
        early.set(false);
        DDX_TRY(apply_rows(w, colA, rowA));
        DDX_TRY(cholqr(w, rowA, M, rowB));
        DDX_TRY(cholqr(w, rowB, M, rowA));       // second pass: orthonormal to working precision
        DDX_TRY(apply_cols(w, rowA, colB));
        Qfinal = rowA; RQ = M;
        Bt = colB; RB = H;
    } else {
        DDX_HIP(ctx, hipMemcpyAsync(rowA, ctx->pcaQ0.p, sizeof(double) * (size_t)M * L, hipMemcpyDeviceToDevice, ctx->stream));
        for (int it = 0; it < n_iter; ++it) {
            early.set(it + 1 < n_iter);
            DDX_TRY(apply_cols(w, rowA, colA));
            DDX_TRY(apply_rows(w, colA, rowB));
            DDX_TRY(cholqr(w, rowB, M, rowA));
        }
        early.set(false);
        DDX_TRY(apply_cols(w, rowA, colA));
        DDX_TRY(cholqr(w, colA, H, colB));
        DDX_TRY(cholqr(w, colB, H, colA));
        DDX_TRY(apply_rows(w, colA, rowB));
        Qfinal = colA; RQ = H;
        Bt = rowB; RB = M;
    }
    // small eigenproblem: B B^T = Bt^T Bt = Uhat diag(s^2) Uhat^T
    double* G = w.small;
    DDX_TRY(gram(w, Bt, RB, G));
    std::vector<double> hG((size_t)L * L), evals(L), evecs((size_t)L * L);
    DDX_HIP(ctx, hipMemcpyAsync(hG.data(), G, sizeof(double) * L * L, hipMemcpyDeviceToHost, ctx->stream));
    int hflag = 0;
    DDX_HIP(ctx, hipMemcpyAsync(&hflag, w.flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));
    for (size_t t = 0; t < hG.size(); ++t)
        if (!std::isfinite(hG[t])) return set_err(ctx, DDX_E_NUMERIC, "non-finite sketch (degenerate input matrix?)");
    // symmetrise against rounding, then Jacobi
    for (int a = 0; a < L; ++a)
        for (int b = a + 1; b < L; ++b) hG[a * L + b] = hG[b * L + a] = 0.5 * (hG[a * L + b] + hG[b * L + a]);
    jacobi_eigh(L, hG.data(), evals.data(), evecs.data());  // ascending eigenvalues, columns = vectors
    // T1 = Uhat[:, :C] (descending singular values);  s = sqrt(eval)
    std::vector<double> T1((size_t)L * C), svals(C), T2((size_t)L * C);
    for (int c = 0; c < C; ++c) {
        const int src = L - 1 - c;
        svals[c] = std::sqrt(std::max(evals[src], 0.0));
        for (int a = 0; a < L; ++a) T1[(size_t)a * C + c] = evecs[(size_t)a * L + src];
    }
    double* dT = w.small + 2 * L * L;  // L x C
    double* dSign = w.small + 3 * L * L + 2 * L;
    DDX_HIP(ctx, hipMemcpyAsync(dT, T1.data(), sizeof(double) * L * C, hipMemcpyHostToDevice, ctx->stream));
    // sign decision on the component rows: components ~ (other side) * Uhat
    double* signSrc = transposed ? Qfinal : Bt;
    const int64_t signR = transposed ? RQ : RB;        // both are H
    double* scratchHC = transposed ? colB : colA;      // H x C scratch (colA/colB are H x L >= H x C)
    {
        ScopedTimer t(ctx, "pca_finish");
        DDX_TRY(right_mult(ctx, signSrc, signR, L, dT, C, scratchHC));
        k_col_sign<<<C, 256, 0, ctx->stream>>>(scratchHC, signR, C, dSign);
    }
    std::vector<double> hsign(C);
    DDX_HIP(ctx, hipMemcpyAsync(hsign.data(), dSign, sizeof(double) * C, hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));
    for (int c = 0; c < C; ++c) {
        const double f = hsign[c] * (transposed ? 1.0 : svals[c]);
        for (int a = 0; a < L; ++a) T2[(size_t)a * C + c] = T1[(size_t)a * C + c] * f;
    }
    DDX_HIP(ctx, hipMemcpyAsync(dT, T2.data(), sizeof(double) * L * C, hipMemcpyHostToDevice, ctx->stream));
    DDX_HIP(ctx, hipMemcpyAsync(ctx->sing.p, svals.data(), sizeof(double) * C, hipMemcpyHostToDevice, ctx->stream));
    {
        ScopedTimer t(ctx, "pca_finish");
        const double* scoreSrc = transposed ? Bt : Qfinal;  // both M x L
        DDX_TRY(right_mult(ctx, scoreSrc, M, L, dT, C, ctx->emb64.as<double>()));
        k_f64_to_f32<<<(unsigned)ceil_div(M * C, 256), 256, 0, ctx->stream>>>(ctx->emb64.as<double>(), M * C, ctx->emb32.as<float>());
    }
    DDX_HIP(ctx, wait_stream(ctx));  // T2/svals are stack-backed host buffers
    DDX_HIP(ctx, hipGetLastError());
    ctx->C = C;
    ctx->embM = M;
    ctx->have_emb = true;
    ctx->have_knn = false;
    if (hflag) {
        // not fatal: the sketch is wider than the numerical rank; the leading components are unaffected
        ctx->err = "rank-deficient sketch (a pivot was floored in the Cholesky-QR of the randomized PCA): the matrix has "
                   "fewer independent directions than n_components + n_oversamples; trailing components are arbitrary";
        return DDX_W_RANK;
    }
    return DDX_OK;
}

static void sym_eigh_ql(int n, double* v, double* d);

// ------------------------------------------------------------------------------------------------
// Exact truncated PCA of the sparse operator: block Lanczos (dd.py:296-297,308 -- pseudocount == 1 keeps the matrix sparse
// and sc.tl.pca switches to svd_solver="arpack", i.e. a truncated SVD converged to tolerance).
//
// Upstream's ARPACK is a single-vector Lanczos: 232 operator applications per PCA at configs[1], each one a pass over the
// matrix.  The block version builds the same Krylov space L vectors at a time -- every step is ONE pair of the 40-column
// LDS-staged products -- with full re-orthogonalisation against all earlier blocks (two passes of classical Gram-Schmidt
// by small cross-Gram kernels, Cholesky-QR twice), collects the projected matrix T = V^T (A^T A) V from the
// orthogonalisation coefficients, and stops when the Ritz pairs of the wanted components have residuals below tol * theta
// (residual of a pair = |T[j+1][j] z_j|, the part of the operator's image that left the space).  Everything tall stays on
// the device; the host sees the small coefficient blocks and solves the (j+1) L eigenproblem of T (Householder + QL).
// Works on the smaller side: A^T A (vectors of length H) when H <= M, A A^T otherwise.
// ------------------------------------------------------------------------------------------------
// partial[b] = X^T W over the block's rows (L x L, row-major), L <= 64.  Thread (ta, tb) of the 16 x 16 owns the pairs
// (ta + 16 p, tb + 16 q): per staged row four values of X and four of W feed sixteen multiply-adds.
__global__ void __launch_bounds__(256) k_cross_gram_partial(const double* __restrict__ X, const double* __restrict__ W, int64_t R, int L, int64_t rows_per_block,
                                                            double* __restrict__ partial) {
    extern __shared__ double cg_s[];                  // [2][32][L] row tiles of X and W
    double* xs = cg_s;
    double* ws = cg_s + 32 * L;
    X += (size_t)blockIdx.y * R * L;                   // batched over the stored blocks (grid.y): block i against the same W
    partial += (size_t)blockIdx.y * gridDim.x * L * L;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    const int ta = threadIdx.x >> 4, tb = threadIdx.x & 15;
    double acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = 0.0;
    for (int64_t rb = r0; rb < r1; rb += 32) {
        const int nr = (int)(r1 - rb < 32 ? r1 - rb : 32);
        for (int e = threadIdx.x; e < nr * L; e += 256) { xs[e] = X[rb * L + e]; ws[e] = W[rb * L + e]; }
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            double xv[4], wv[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                xv[p] = ta + 16 * p < L ? xs[r * L + ta + 16 * p] : 0.0;
                wv[p] = tb + 16 * p < L ? ws[r * L + tb + 16 * p] : 0.0;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[p][q] = fma(xv[p], wv[q], acc[p][q]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ta + 16 * p, j = tb + 16 * q;
            if (i < L && j < L) partial[(int64_t)blockIdx.x * L * L + i * L + j] = acc[p][q];
        }
}

// out[i][c] = sum_b partial[i][b][c] (the batched form of k_reduce_partials: grid.y = i)
__global__ void __launch_bounds__(256) k_reduce_partials_batched(const double* __restrict__ partial, int nblocks, int width, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= width) return;
    const double* p = partial + (size_t)blockIdx.y * nblocks * width;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += p[(int64_t)b * width + c];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[(size_t)blockIdx.y * width + c] = s;
}

// out = (SUBTRACT ? out : 0) -/+ sum_i V_i G_i over the nblk stored blocks (V_i: R x L at V + i R L, G_i: L x L at G + i L L).
// A workgroup takes 64 rows: per block i the coefficient block and the 64 x L tile of V_i are staged in LDS (rows padded by one
// value against bank conflicts); thread (row pair rl / rl + 32, column class cg) keeps 2 x 8 sums for columns cg + 8 q.
template <bool SUBTRACT>
__global__ void __launch_bounds__(256) k_block_apply_all(const double* __restrict__ V, const double* __restrict__ G, int64_t R, int L, int nblk,
                                                         double* __restrict__ out) {
    extern __shared__ double ba_s[];                  // G_i [L x L] | V tile [64 x (L + 1)]
    double* gs = ba_s;
    double* vs = ba_s + L * L;
    const int rl = threadIdx.x >> 3, cg = threadIdx.x & 7;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int nr = (int)(R - row0 < 64 ? R - row0 : 64);
    double acc[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[u][q] = 0.0;
    for (int i = 0; i < nblk; ++i) {
        __syncthreads();
        for (int e = threadIdx.x; e < L * L; e += 256) gs[e] = G[(size_t)i * L * L + e];
        const double* src = V + ((size_t)i * R + row0) * L;
        for (int e = threadIdx.x; e < nr * L; e += 256) { const int r = e / L; vs[r * (L + 1) + (e - r * L)] = src[e]; }
        __syncthreads();
        for (int k = 0; k < L; ++k) {
            const double v0 = vs[rl * (L + 1) + k], v1 = vs[(rl + 32) * (L + 1) + k];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int col = cg + 8 * q;
                const double g = col < L ? gs[k * L + col] : 0.0;
                acc[0][q] = fma(v0, g, acc[0][q]);
                acc[1][q] = fma(v1, g, acc[1][q]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = rl + 32 * u;
        if (r >= nr) continue;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int col = cg + 8 * q;
            if (col >= L) continue;
            double* o = out + (row0 + r) * L + col;
            *o = SUBTRACT ? *o - acc[u][q] : acc[u][q];
        }
    }
}

// W -= V G   (R x L, G: L x L row-major in device memory)
__global__ void __launch_bounds__(256) k_block_subtract(const double* __restrict__ V, const double* __restrict__ G, int64_t R, int L, double* __restrict__ W) {
    extern __shared__ double bs_g[];                  // G
    for (int e = threadIdx.x; e < L * L; e += 256) bs_g[e] = G[e];
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= R * L) return;
    const int64_t r = t / L;
    const int c = (int)(t - r * L);
    double a = 0.0;
    for (int k = 0; k < L; ++k) a = fma(V[r * L + k], bs_g[k * L + c], a);
    W[t] -= a;
}

// out (+)= V Z   (R x L times L x L2)
__global__ void __launch_bounds__(256) k_block_accumulate(const double* __restrict__ V, const double* __restrict__ Z, int64_t R, int L, int L2, int first,
                                                          double* __restrict__ out) {
    extern __shared__ double ba_z[];
    for (int e = threadIdx.x; e < L * L2; e += 256) ba_z[e] = Z[e];
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= R * L2) return;
    const int64_t r = t / L2;
    const int c = (int)(t - r * L2);
    double a = first ? 0.0 : out[t];
    for (int k = 0; k < L; ++k) a = fma(V[r * L + k], ba_z[k * L2 + c], a);
    out[t] = a;
}

__global__ void k_scale_cols(const double* __restrict__ in, int64_t R, int L, int C, const double* __restrict__ f, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * C) return;
    const int64_t r = t / C;
    const int c = (int)(t - r * C);
    out[t] = in[r * L + c] * f[c];
}

static int cross_gram(PcaWork& w, const double* X, const double* W, int64_t R, double* G_dev, double* G_host) {
    const int L = w.L;
    int nb = (int)std::min<int64_t>(256, ceil_div(R, 64));
    const int64_t rpb = ceil_div(ceil_div(R, nb), 32) * 32;
    nb = (int)ceil_div(R, rpb);
    k_cross_gram_partial<<<nb, 256, sizeof(double) * 64 * L, w.ctx->stream>>>(X, W, R, L, rpb, w.partial);
    k_reduce_partials<<<(unsigned)ceil_div(L * L, 4), 256, 0, w.ctx->stream>>>(w.partial, nb, L * L, G_dev);
    if (G_host) DDX_HIP(w.ctx, hipMemcpyAsync(G_host, G_dev, sizeof(double) * L * L, hipMemcpyDeviceToHost, w.ctx->stream));
    return DDX_OK;
}

constexpr int kCrossGramChunks = 64;      // row chunks per stored block in the batched cross-Gram
// G_i = V_i^T W for the blocks i < nblk at once (device: G_dev[i]; host copy when asked -- asynchronous, the caller synchronises)
static int cross_gram_all(PcaWork& w, const double* Vall, int nblk, const double* W, int64_t R, double* G_dev, double* G_host) {
    const int L = w.L;
    int nb = (int)std::min<int64_t>(kCrossGramChunks, ceil_div(R, 64));      // (x nblk workgroups: plenty, and the partial sums stay small)
    const int64_t rpb = ceil_div(ceil_div(R, nb), 32) * 32;
    nb = (int)ceil_div(R, rpb);
    k_cross_gram_partial<<<dim3((unsigned)nb, (unsigned)nblk), 256, sizeof(double) * 64 * L, w.ctx->stream>>>(Vall, W, R, L, rpb, w.partial);
    k_reduce_partials_batched<<<dim3((unsigned)ceil_div(L * L, 4), (unsigned)nblk), 256, 0, w.ctx->stream>>>(w.partial, nb, L * L, G_dev);
    if (G_host) DDX_HIP(w.ctx, hipMemcpyAsync(G_host, G_dev, sizeof(double) * (size_t)nblk * L * L, hipMemcpyDeviceToHost, w.ctx->stream));
    return DDX_OK;
}

int stage_pca_block_lanczos(ddx_ctx* ctx, int32_t C, int32_t oversample, double tol, int32_t max_steps, const double* q0, int32_t* steps_out,
                            ddx_eigh_fn eigh, void* eigh_user) {
    t_opt = &ctx->opt;
    const int L = C + oversample;
    if (L > kMaxL) return set_err(ctx, DDX_E_UNSUPPORTED, "block Lanczos: %d components + %d exceed %d columns", C, oversample, kMaxL);
    const int64_t M = ctx->M;
    const int32_t H = ctx->H;
    const bool cols_side = H <= M;                     // vectors of length H, operator A^T A
    const int64_t R = cols_side ? (int64_t)H : M, Ro = cols_side ? M : (int64_t)H;
    if (max_steps < 2) max_steps = 2;
    if ((int64_t)(max_steps + 1) * L > R) max_steps = (int)std::max<int64_t>(1, R / L - 1);
    DDX_TRY(ensure(ctx, ctx->pcaA, sizeof(double) * 2 * (size_t)M * L));
    DDX_TRY(ensure(ctx, ctx->pcaB, sizeof(double) * 2 * (size_t)H * L));
    DDX_TRY(ensure(ctx, ctx->pcaSmall, sizeof(double) * ((8 + (size_t)max_steps + 2) * L * L + 8 * L) + 256));
    DDX_TRY(ensure(ctx, ctx->pcaPartial, sizeof(double) * std::max<size_t>(512 * (size_t)std::max(L * L, 128 * 4), (size_t)kCrossGramChunks * ((size_t)max_steps + 2) * L * L)));
    DDX_TRY(ensure(ctx, ctx->pcaVec, 256));
    DDX_TRY(ensure(ctx, ctx->pcaPanel, sizeof(double) * (size_t)ceil_div(M, ctx->panel_rows) * H * L));
    DDX_TRY(ensure(ctx, ctx->pcaBlk, sizeof(double) * (size_t)(max_steps + 2) * R * L));
    DDX_TRY(ensure(ctx, ctx->emb64, sizeof(double) * (size_t)M * C));
    DDX_TRY(ensure(ctx, ctx->emb32, sizeof(float) * (size_t)M * C));
    DDX_TRY(ensure(ctx, ctx->sing, sizeof(double) * L));
    double* rowA = ctx->pcaA.as<double>();
    double* rowB = rowA + (size_t)M * L;
    double* colA = ctx->pcaB.as<double>();
    double* colB = colA + (size_t)H * L;
    double* Vall = ctx->pcaBlk.as<double>();
    auto Vblk = [&](int j) { return Vall + (size_t)j * R * L; };
    PcaWork w;
    w.ctx = ctx; w.L = L; w.lpn = (L + 1) / 2; w.slots = 64 / w.lpn; w.gather32 = ctx->opt.gather_f32; w.M = M; w.H = H;
    DDX_TRY(lds_setup(ctx, L, w));
    w.fused_ymax = cols_side;              // (row-side blocks are re-orthogonalised in place between the products)
    const int64_t maxR = M > H ? M : (int64_t)H;
    DDX_TRY(ensure(ctx, ctx->pcaOp, sizeof(double) * (size_t)maxR * (L + 4)));
    w.op32 = ctx->pcaOp.as<float>();
    if (w.lds) {
        w.opQ = w.op32 + (size_t)maxR * (L + 4);
        w.opQ_of = w.opY_of = nullptr;
        if (((L + 3) & ~3) != L) DDX_HIP(ctx, hipMemsetAsync(w.op32, 0, sizeof(double) * (size_t)maxR * (L + 4), ctx->stream));
    }
    w.partial = ctx->pcaPartial.as<double>();
    w.small = ctx->pcaSmall.as<double>();
    w.flag = ctx->pcaVec.as<int>();
    DDX_HIP(ctx, hipMemsetAsync(w.flag, 0, 16 * sizeof(int), ctx->stream));       // rank flag + the tickets of the last-block reductions
    double* dG = w.small + 4 * L * L + 4 * L;          // cross-Gram block (the first 4 L^2 + 4 L belong to cholqr and the products)
    double* dZ = dG + L * L;                           // L x L block of Ritz vectors
    double* dGall = dZ + 2 * L * L;                    // cross-Gram blocks against every stored block: (max_steps + 1) L^2
    // small side scratch: W (the image of a block), T1 (a second block)
    double* Wb = cols_side ? colA : rowA;
    double* T1 = cols_side ? colB : rowB;
    double* Yo = cols_side ? rowA : colA;              // the other side's image
    auto op = [&](const double* X, double* Wout) -> int {          // Wout = (A^T A or A A^T) X
        if (cols_side) { DDX_TRY(apply_rows(w, X, Yo)); DDX_TRY(apply_cols(w, Yo, Wout)); }
        else { DDX_TRY(apply_cols(w, X, Yo)); DDX_TRY(apply_rows(w, Yo, Wout)); }
        return DDX_OK;
    };
    // V_0 = orth(start)
    DDX_HIP(ctx, hipMemcpyAsync(Wb, q0, sizeof(double) * (size_t)R * L, hipMemcpyHostToDevice, ctx->stream));
    DDX_TRY(cholqr(w, Wb, R, T1));
    DDX_TRY(cholqr(w, T1, R, Vblk(0)));
    const int nmax = (max_steps + 1) * L;
    std::vector<double> T((size_t)nmax * nmax, 0.0), Gh((size_t)L * L), Gall, work, theta, Tsub((size_t)L * L);
    auto add_block = [&](int bi, int bj, const std::vector<double>& G, bool accumulate) {     // T[bi][bj] (+)= G, mirrored
        for (int a = 0; a < L; ++a)
            for (int b = 0; b < L; ++b) {
                double& t = T[(size_t)(bi * L + a) * nmax + bj * L + b];
                t = accumulate ? t + G[(size_t)a * L + b] : G[(size_t)a * L + b];
                T[(size_t)(bj * L + b) * nmax + bi * L + a] = t;
            }
    };
    const unsigned gtile = (unsigned)ceil_div(R, 64);
    const size_t tile_lds = sizeof(double) * ((size_t)L * L + 64 * (size_t)(L + 1));      // 65 KB at L = 64
    DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_block_apply_all<true>), (int)tile_lds));
    DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_block_apply_all<false>), (int)tile_lds));
    int steps = 0;
    std::vector<double> Z;                              // Ritz vectors of the last solve: n x n row-major, columns = vectors
    int n = 0;
    double t_prod = 0.0, t_orth = 0.0, t_ritz = 0.0;
    int next_check = 12, prev_step = 0;
    double prev_res = -1.0, last_res = -1.0;
    bool converged = false;
    auto now = [&]() { (void)wait_stream(ctx); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool dbg = ctx->opt.pca_debug;
    for (int j = 0; j < max_steps; ++j) {
        double t0 = dbg ? now() : 0.0;
        DDX_TRY(op(Vblk(j), Wb));
        if (dbg) { const double t1 = now(); t_prod += t1 - t0; t0 = t1; }
        // two passes of block Gram-Schmidt against ALL stored blocks at once: one batched cross-Gram, one subtraction and one
        // copy of the coefficients per pass (the first version went block by block: 2 (j + 1) round trips per step)
        for (int pass = 0; pass < 2; ++pass) {
            Gall.resize((size_t)(j + 1) * L * L);
            DDX_TRY(cross_gram_all(w, Vall, j + 1, Wb, R, dGall, Gall.data()));
            k_block_apply_all<true><<<gtile, 256, tile_lds, ctx->stream>>>(Vall, dGall, R, L, j + 1, Wb);
            DDX_HIP(ctx, wait_stream(ctx));
            for (int i = 0; i <= j; ++i) {
                std::copy(Gall.begin() + (size_t)i * L * L, Gall.begin() + (size_t)(i + 1) * L * L, Gh.begin());
                add_block(i, j, Gh, pass > 0);
            }
        }
        // next block: orth(remainder), made orthogonal to the earlier blocks once more (a remainder that has lost rank -- converged
        // directions -- leaves arbitrary vectors behind the Cholesky floor)
        DDX_TRY(cholqr(w, Wb, R, T1));
        DDX_TRY(cross_gram_all(w, Vall, j + 1, T1, R, dGall, nullptr));
        k_block_apply_all<true><<<gtile, 256, tile_lds, ctx->stream>>>(Vall, dGall, R, L, j + 1, T1);
        DDX_TRY(cholqr(w, T1, R, Vblk(j + 1)));
        DDX_TRY(cross_gram(w, Vblk(j + 1), Wb, R, dG, Tsub.data()));              // T[j+1][j] = V_{j+1}^T (remainder)
        DDX_HIP(ctx, wait_stream(ctx));
        if (dbg) { const double t1 = now(); t_orth += t1 - t0; t0 = t1; }
        steps = j + 1;
        // Rayleigh-Ritz on the blocks 0..j -- the host's solve, as costly as several steps once the space is large: at steps 8
        // and 12, then where the residual's decay so far says the tolerance is met (at most 8 steps ahead), and at the last step
        const bool last = j + 1 >= max_steps || (int64_t)(j + 3) * L > R;
        if (!last && j + 1 < next_check) continue;
        n = (j + 1) * L;
        Z.assign((size_t)n * n, 0.0);
        theta.assign(n, 0.0);
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) Z[(size_t)a * n + b] = 0.5 * (T[(size_t)a * nmax + b] + T[(size_t)b * nmax + a]);
        for (size_t t = 0; t < Z.size(); ++t)
            if (!std::isfinite(Z[t])) return set_err(ctx, DDX_E_NUMERIC, "block Lanczos: non-finite projected matrix (degenerate input matrix?)");
        // eigenvectors in the columns of Z, eigenvalues ascending: the caller's solver (LAPACK behind numpy.linalg.eigh for
        // the Python host) or the built-in Householder + QL
        if (eigh) {
            if (eigh(n, std::min(n, L), Z.data(), theta.data(), eigh_user) != 0) return set_err(ctx, DDX_E_NUMERIC, "block Lanczos: the caller's eigen-solver failed");
        } else {
            sym_eigh_ql(n, Z.data(), theta.data());
        }
        double worst = 0.0;
        for (int c = 0; c < C && c < n; ++c) {
            const int col = n - 1 - c;
            double r2 = 0.0;
            for (int a = 0; a < L; ++a) {
                double sacc = 0.0;
                for (int b = 0; b < L; ++b) sacc += Tsub[(size_t)a * L + b] * Z[(size_t)(j * L + b) * n + col];
                r2 += sacc * sacc;
            }
            const double th = std::fabs(theta[col]);
            worst = std::max(worst, th > 0.0 ? std::sqrt(r2) / th : 0.0);
        }
        if (dbg) {
            t_ritz += now() - t0;
            fprintf(stderr, "[lanczos] step %d: n = %d, worst residual / eigenvalue %.2e; so far products %.1f ms, orthogonalisation %.1f ms, Ritz %.1f ms\n", j + 1, n, worst,
                    t_prod, t_orth, t_ritz);
        }
        last_res = worst;
        if (worst <= tol) converged = true;
        if (worst <= tol || last) break;
        // the next solve where the decay says the tolerance is met.  A solve costs as much as ~20 steps by now (n = 880: 31 ms on the
        // host against 1.4 ms per step on the device, profiles/r05_block_lanczos.txt), and the decay accelerates (0.25 decades per step
        // early, 0.5 late): rather a step too many than a solve too many -- 0.8 of the steps the observed rate asks for, a rate of 0.5
        // per step assumed at the first check
        if (worst > 0.0) {
            const double rate = (prev_res > 0.0 && worst < prev_res) ? std::pow(worst / prev_res, 1.0 / (double)(j + 1 - prev_step)) : 0.5;      // per step, < 1
            const double need = std::log(tol / worst) / std::log(std::min(rate, 0.9));
            next_check = j + 1 + (int)std::min(12.0, std::max(2.0, std::ceil(0.8 * need)));
        } else {
            next_check = j + 1 + 4;
        }
        prev_res = worst;
        prev_step = j + 1;
    }
    if (steps_out) *steps_out = steps;
    // X = V Z[:, top L] (small side x L; only the first C columns are results, the rest fills the product's width)
    const int nb = n / L;
    std::vector<double> Zb((size_t)std::max(nb, 1) * L * L), svals(L, 0.0);
    for (int c = 0; c < L; ++c) svals[c] = std::sqrt(std::max(c < n ? theta[n - 1 - c] : 0.0, 0.0));
    for (int i = 0; i < nb; ++i)
        for (int a = 0; a < L; ++a)
            for (int c = 0; c < L; ++c) Zb[((size_t)i * L + a) * L + c] = c < n ? Z[(size_t)(i * L + a) * n + (n - 1 - c)] : 0.0;
    DDX_HIP(ctx, hipMemcpyAsync(dGall, Zb.data(), sizeof(double) * (size_t)nb * L * L, hipMemcpyHostToDevice, ctx->stream));
    k_block_apply_all<false><<<gtile, 256, tile_lds, ctx->stream>>>(Vall, dGall, R, L, nb, Wb);
    DDX_HIP(ctx, wait_stream(ctx));          // Zb leaves scope
    // components (H x L) for the sign decision, scores (M x C) = U S
    w.opQ_of = w.opY_of = nullptr;                      // (the accumulated block has no float32 mirror yet)
    double* dSign = w.small + 3 * L * L + 2 * L;
    std::vector<double> hsign(L, 1.0), f(L, 1.0);
    const double* comps;
    const double* left = nullptr;
    if (cols_side) {
        comps = Wb;                                      // right singular vectors
    } else {
        DDX_TRY(apply_cols(w, Wb, colA));                // A^T U = V S
        comps = colA;
        left = Wb;
    }
    k_col_sign<<<L, 256, 0, ctx->stream>>>(comps, H, L, dSign);
    DDX_HIP(ctx, hipMemcpyAsync(hsign.data(), dSign, sizeof(double) * L, hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));
    double* dF = w.small + 3 * L * L + 3 * L;
    const double* scoreSrc;
    if (cols_side) {
        DDX_TRY(apply_rows(w, Wb, rowB));                // A V = U S
        scoreSrc = rowB;
        for (int c = 0; c < L; ++c) f[c] = hsign[c] != 0.0 ? hsign[c] : 1.0;
    } else {
        scoreSrc = left;
        for (int c = 0; c < L; ++c) f[c] = (hsign[c] != 0.0 ? hsign[c] : 1.0) * svals[c];
    }
    DDX_HIP(ctx, hipMemcpyAsync(dF, f.data(), sizeof(double) * L, hipMemcpyHostToDevice, ctx->stream));
    k_scale_cols<<<(unsigned)ceil_div(M * C, 256), 256, 0, ctx->stream>>>(scoreSrc, M, L, C, dF, ctx->emb64.as<double>());
    k_f64_to_f32<<<(unsigned)ceil_div(M * C, 256), 256, 0, ctx->stream>>>(ctx->emb64.as<double>(), M * C, ctx->emb32.as<float>());
    DDX_HIP(ctx, hipMemcpyAsync(ctx->sing.p, svals.data(), sizeof(double) * C, hipMemcpyHostToDevice, ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));
    DDX_HIP(ctx, hipGetLastError());
    (void)Ro;
    ctx->C = C;
    ctx->embM = M;
    ctx->have_emb = true;
    ctx->have_knn = false;
    if (!converged) {
        // not fatal here: the embedding holds the best Ritz pairs of the space that was built.  Upstream's ARPACK is converged to
        // working precision at every size, so the caller falls back to an exact decomposition or tells the user.
        ctx->err = "block Lanczos stopped after " + std::to_string(steps) + " steps (" + std::to_string((long long)(steps + 1) * L) + " of " +
                   std::to_string((long long)R) + " dimensions) with a relative residual of " + std::to_string(last_res) + " > tolerance " + std::to_string(tol);
        return DDX_W_UNCONVERGED;
    }
    return DDX_OK;
}

// Symmetric eigen-decomposition: Householder reduction to tridiagonal form followed by the implicit QL iteration
// (the classical tred2 / tql2 pair of Bowdler, Martin, Reinsch and Wilkinson).  v: n x n row-major, on entry the
// symmetric matrix, on exit the eigenvectors in its columns; d: eigenvalues, ascending.
static void sym_eigh_ql(int n, double* v, double* d) {
    std::vector<double> e(n, 0.0);
    for (int j = 0; j < n; ++j) d[j] = v[(n - 1) * n + j];
    // Householder reduction
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) {
                d[j] = v[(i - 1) * n + j];
                v[i * n + j] = 0.0;
                v[j * n + i] = 0.0;
            }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                v[j * n + i] = f;
                g = e[j] + v[j * n + j] * f;
                for (int k = j + 1; k <= i - 1; ++k) {
                    g += v[k * n + j] * d[k];
                    e[k] += v[k * n + j] * f;
                }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j];
                g = e[j];
                for (int k = j; k <= i - 1; ++k) v[k * n + j] -= (f * e[k] + g * d[k]);
                d[j] = v[(i - 1) * n + j];
                v[i * n + j] = 0.0;
            }
        }
        d[i] = h;
    }
    // accumulate the transformations
    for (int i = 0; i < n - 1; ++i) {
        v[(n - 1) * n + i] = v[i * n + i];
        v[i * n + i] = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = v[k * n + i + 1] / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += v[k * n + i + 1] * v[k * n + j];
                for (int k = 0; k <= i; ++k) v[k * n + j] -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) v[k * n + i + 1] = 0.0;
    }
    for (int j = 0; j < n; ++j) {
        d[j] = v[(n - 1) * n + j];
        v[(n - 1) * n + j] = 0.0;
    }
    v[(n - 1) * n + n - 1] = 1.0;
    e[0] = 0.0;
    // implicit QL
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = std::pow(2.0, -52.0);
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) {
            if (std::fabs(e[m]) <= eps * tst1) break;
            ++m;
        }
        if (m > l) {
            int iter = 0;
            do {
                ++iter;
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c;
                const double el1 = e[l + 1];
                double s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2;
                    c2 = c;
                    s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = std::hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; ++k) {
                        h = v[k * n + i + 1];
                        v[k * n + i + 1] = s * v[k * n + i] + c * h;
                        v[k * n + i] = c * v[k * n + i] - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
        }
        d[l] += f;
        e[l] = 0.0;
    }
    // ascending order
    for (int i = 0; i < n - 1; ++i) {
        int k = i;
        double p = d[i];
        for (int j = i + 1; j < n; ++j)
            if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) {
            d[k] = d[i];
            d[i] = p;
            for (int j = 0; j < n; ++j) std::swap(v[j * n + i], v[j * n + k]);
        }
    }
}


// eigen-decomposition of the small symmetric matrix B B^T on the host (float64): ascending eigenvalues, eigenvectors
// in the columns of evecs.  (A cyclic Jacobi iteration did this first: 0.53 ms for 40 x 40 against 0.16 ms.)
void jacobi_eigh(int n, double* a, double* evals, double* evecs) {
    for (int t = 0; t < n * n; ++t) evecs[t] = a[t];
    sym_eigh_ql(n, evecs, evals);
}

}  // namespace ddx
