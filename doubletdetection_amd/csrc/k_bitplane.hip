// Bit-plane operator products (sc.tl.pca call site, dd.py:305-314; the two sparse products of every power iteration).
//
// Nine in ten stored entries of the count matrix are a 1, and the value the PCA sees for such an entry depends on its ROW
// only (x = log(1 / lib_i * median + pseudocount), dd.py:286-297).  So the operator splits:
//
//      L  =  diag(s) B  +  R ,      s_i = x_i(1) - z ,   B = [count == 1] (zeros and ones),   R = the entries with other counts
//
// R keeps going through the LDS-staged sparse kernels (k_pca.hip) -- they simply see a matrix with a tenth of the entries.
// B is kept as a BITMAP -- one bit per (row, column), 64 times less memory traffic than the 8 bytes per entry of the sparse
// form -- and multiplied on the matrix cores in exact integer arithmetic: the float64 operand is cut column by column into
// a fixed-point number of ND signed 8-bit digits (ND = 3: 22 bits below the column's largest element, ND = 4: 30), and
// B . digit_d runs on v_mfma_i32_32x32x32_i8 (zeros and ones against 8-bit digits, 32-bit sums: no rounding anywhere, any
// order gives the same bits).  The digit sums are recombined, scaled back and by s_i in float64.
//
// Round 5 (round 4's first version: original rows only, 16 x 16 x 64 tiles over 48 padded columns x 4 digits, one barrier
// per 64 matrix columns, digit sums through global memory: 0.22 - 0.26 ms per product and 9 ms of structure building per
// context and fit -- not faster than the sparse kernel it relieved):
//   * ALL rows take the route: the synthetic doublets' bitmaps and reduced structures are rebuilt every iteration;
//   * the (sketch column, digit) pairs are flattened into one N index, f = column * ND + digit: 40 columns x 4 digits are
//     exactly five 32-wide tiles, 40 x 3 digits four (128 for 120);
//   * stages of 256 matrix columns (eight MFMA k-steps) per barrier, three stages of digits in LDS by asynchronous copies,
//     the operand fragments read two MFMA groups ahead (the two waves of a SIMD leave a barrier together: without the
//     prefetch both sit in the same LDS round trip), the bitmap words straight into registers one stage ahead;
//   * the epilogue recombines the digits through LDS and writes the float64 result (A Q) or one partial block per chunk of
//     the k dimension (A^T Y) -- no digit sums in global memory, no combine kernels.
// Measured stand-alone (profiles/tools/bp2_proto.hip, profiles/r05_bitplane_notes.txt): the matrix pipe is busy 82 % of the
// kernel's cycles; the clock drops to ~1.6 GHz under it (power), which is what the wall-clock sees.
//
// Scaled matrices (standard_scaling, dd.py:302-303): (x - mean_j) / sd_j makes the value of an unclipped entry equal to 1
// s_i / sd_j, a diagonal factor -- the bitmaps stay what they are, 1 / sd_j goes into the operand rows before the digits (A Q)
// and into the epilogue of A^T Y (k_sum_panels), the statistics come from one narrow product (columns s, x(1)^2, 1).  Columns in
// which such an entry could reach the +-max_value clip are demoted for the fit: all their entries sit in the reduced structures
// with their own clipped values (bp_scale in k_sparse.hip; BitPlanes::demote).
#include "ddx_prims.h"

#include <type_traits>

#include "ddx_internal.h"

namespace ddx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kBpStageCols = 256;       // matrix columns (k) per LDS stage = eight MFMA k-steps of 32
constexpr int kBpSteps = 8;
constexpr int kBpWaves = 8;             // waves per workgroup of the product kernel
constexpr int kBpStages = 3;

// padded row index ("k space" of the A^T Y product, tile space of the A Q product): the original rows are padded to a multiple
// of 256, the synthetic rows follow.  pr -> row of the augmented matrix, or -1
__device__ __forceinline__ int64_t bp_row_of(int64_t pr, int64_t Npad, int64_t N, int64_t M) {
    if (pr < Npad) return pr < N ? pr : -1;
    const int64_t r = N + (pr - Npad);
    return r < M ? r : -1;
}

// ---- bitmaps ---------------------------------------------------------------------------------------------------------
// Row bitmap: 16-byte word (tile, sk, r, h) holds columns sk*256 + h*128 .. +127 of padded row tile*32 + r;
// layout [(tile * SK + sk) * 64 + r * 2 + h]: the 32 rows x 256 columns of a (tile, stage) pair are one contiguous KB.
// One workgroup per tile: the tile's image is assembled in LDS (integer atomics: bits set in any order), `sk_chunk` stages at
// a time, and written out in full lines.
__global__ void __launch_bounds__(256) k_bp_rows_bitmap(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols, const float* __restrict__ raw,
                                                        int64_t tile0, int64_t Npad, int64_t N, int64_t M, int SK, int sk_chunk, const uint8_t* __restrict__ demoted,
                                                        v4i* __restrict__ bm) {
    extern __shared__ __align__(16) uint32_t bp_img[];              // [32 rows][sk_chunk * 8 words]
    const int64_t tile = tile0 + blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpr = sk_chunk * 8;                                    // words per row of the image
    for (int sk0 = 0; sk0 < SK; sk0 += sk_chunk) {
        const int nsk = SK - sk0 < sk_chunk ? SK - sk0 : sk_chunk;
        for (int i = threadIdx.x; i < 32 * wpr; i += 256) bp_img[i] = 0u;
        __syncthreads();
        const int32_t c_lo = sk0 * kBpStageCols, c_hi = (sk0 + nsk) * kBpStageCols;
        for (int r = wave; r < 32; r += 4) {
            const int64_t row = bp_row_of(tile * 32 + r, Npad, N, M);
            if (row < 0) continue;
            const int64_t b = indptr[row], e = indptr[row + 1];
            for (int64_t p = b + lane; p < e; p += 64) {
                const int32_t c = cols[p];
                if (raw[p] == 1.0f && c >= c_lo && c < c_hi && !demoted[c]) atomicOr(&bp_img[r * wpr + ((c - c_lo) >> 5)], 1u << (c & 31));
            }
        }
        __syncthreads();
        // 16-byte pieces (sk, r, h) -> out[(tile * SK + sk0 + sk) * 64 + r * 2 + h]
        for (int i = threadIdx.x; i < nsk * 64; i += 256) {
            const int sk = i >> 6, rh = i & 63, r = rh >> 1, h = rh & 1;
            const uint32_t* src = bp_img + r * wpr + sk * 8 + h * 4;
            bm[(tile * SK + sk0 + sk) * 64 + rh] = v4i{(int)src[0], (int)src[1], (int)src[2], (int)src[3]};
        }
        __syncthreads();
    }
}

// The same bits by columns: word (ctile, skr, c, h) holds padded rows skr*256 + h*128 .. +127 of column ctile*32 + c.
// One wave per 64 (padded rows) x 64 (columns) block: lane = row reads its 64 column bits, 64 ballots transpose them.
// prow0: first padded row of the range to transpose (a multiple of 64), nblk_r: 64-row blocks of the range.
__global__ void __launch_bounds__(256) k_bp_transpose(const v4i* __restrict__ bm, int64_t prow0, int64_t nblk_r, int SKc, int64_t SKr, int32_t Hpad32,
                                                      v4i* __restrict__ bmT) {
    const int lane = threadIdx.x & 63;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nblk_c = SKc * 4;                                      // 64-column blocks
    if (blk >= nblk_r * nblk_c) return;
    const int64_t rb = blk / nblk_c;
    const int cb = (int)(blk - rb * nblk_c);
    const int64_t pr = prow0 + rb * 64 + lane;                       // this lane's padded row
    // columns cb*64 .. +63 of row pr: stage sk = cb / 4, half h = (cb & 3) >> 1, 64-bit half (cb & 1) of the 16-byte word
    const uint64_t* src = reinterpret_cast<const uint64_t*>(bm + ((pr >> 5) * SKc + (cb >> 2)) * 64 + (pr & 31) * 2 + ((cb & 3) >> 1));
    const uint64_t w = src[cb & 1];
    uint64_t mine = 0;
    for (int c = 0; c < 64; ++c) {
        const uint64_t t = __ballot((w >> c) & 1ull);
        if (lane == c) mine = t;
    }
    const int32_t col = cb * 64 + lane;
    if (col >= Hpad32) return;
    const int64_t pr0 = prow0 + rb * 64;
    uint64_t* dst = reinterpret_cast<uint64_t*>(bmT + ((int64_t)(col >> 5) * SKr + (pr0 >> 8)) * 64 + (col & 31) * 2 + ((pr0 & 255) >> 7));
    dst[(pr0 & 127) >> 6] = mine;
}

// ---- reduced structures: the entries other than 1 ----------------------------------------------------------------------
// (an entry stays out of the bitmaps -- goes to the reduced structures -- when its count is not 1 or its column is demoted)
__global__ void k_bp_flags(const float* __restrict__ raw, const int32_t* __restrict__ cols, const uint8_t* __restrict__ demoted, int64_t n, int32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    flag[i] = (i < n && (raw[i] != 1.0f || demoted[cols[i]])) ? 1 : 0;            // (element n: 0, so that the scan's last element is the total)
}

// kept entries -> idx_out / pos_out (pos: position in the full arrays, for the per-iteration value refresh; may be null) / x_out (may be null)
__global__ void k_bp_compact(const int32_t* __restrict__ flag, const int32_t* __restrict__ scan, const int32_t* __restrict__ idx, const float* __restrict__ x,
                             int64_t n, int32_t* __restrict__ idx_out, int32_t* __restrict__ pos_out, float* __restrict__ x_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const int32_t o = scan[i];
    idx_out[o] = idx[i];
    if (pos_out) pos_out[o] = (int32_t)i;
    if (x_out) x_out[o] = x[i];
}

// out[i] = base + scan[ptr[i] - e0]
__global__ void k_bp_pointers(const int64_t* __restrict__ ptr, int64_t nptr, int64_t e0, const int32_t* __restrict__ scan, int64_t base, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nptr) return;
    out[i] = base + scan[ptr[i] - e0];
}

// row of every entry of a reduced CSR (once per fit: the original rows' reduced values are evaluated entry by entry)
__global__ void __launch_bounds__(256) k_bp_rows_of(const int64_t* __restrict__ indptr, int64_t nrows, int32_t* __restrict__ row_out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int64_t e = indptr[r + 1];
    for (int64_t p = indptr[r] + lane; p < e; p += 64) row_out[p] = (int32_t)r;
}

// this iteration's value of an entry from its count and row: the (row, count) table, the full evaluation for the rare others
__device__ __forceinline__ float bp_value(float c, int64_t row, const float* __restrict__ tab, const double* __restrict__ lib64, float med, float pc, int use_log1p) {
    const int k = small_count(c, kLognormTab);
    return k ? tab[row * kLognormTab + (k - 1)] : lognorm_value(c, lib64[row], med, pc, use_log1p != 0);
}

__global__ void k_bp_values(const float* __restrict__ raw, const int32_t* __restrict__ rows, int64_t n, const float* __restrict__ tab, const double* __restrict__ lib64,
                            const float* __restrict__ med, float pc, int use_log1p, float* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = bp_value(raw[i], rows[i], tab, lib64, med[0], pc, use_log1p);
}

// ---- synthetic rows straight from their parents' structures ---------------------------------------------------------------
// A doublet's counts are the sums of its parents' (dd.py:406-410; scipy drops exact zeros), and the parents are rows whose
// structures exist for the whole fit: the bitmap B (count == 1) and the reduced row R (other stored counts; counts here are
// non-negative integers -- ddx_ctx::counts_exact --, so an entry of R is an explicit zero or >= 2; in a DEMOTED column B has no
// bits and R holds the counts equal to 1 as well: the formulas below then give bitmap 0 and reduced R0 | R1 there, which is what
// a demoted column's doublet entries must be).  With NZ = B | [R != 0]:
//     bitmap of the doublet     (B0 & ~NZ1) | (B1 & ~NZ0)                      one parent has a 1, the other nothing
//     its reduced columns       (NZ0 & NZ1) | ([R0 != 0] & ~NZ1) | ([R1 != 0] & ~NZ0)
// so the merged row (k_doublet_fill, on ~900 entries per parent) never has to exist: one wave per doublet reads 2 x 1.25 KB of
// bitmap and 2 x ~90 reduced entries.  Pass 1 (FILL = false) writes the bitmap row and counts the reduced entries; after the
// scan of the counts pass 2 writes them -- column and this iteration's VALUE, slots in column order by popcount ranks, the same
// order the merge gives.  LDS per wave: five arrays of W = 8 * SK words (the two bitmaps, the two [R != 0] masks, the ranks).
__device__ __forceinline__ int64_t bp_word_of(int64_t pr, int d, int SK) {          // 32-bit word d of padded row pr in the row bitmap
    return (((pr >> 5) * SK + (d >> 3)) * 64 + (pr & 31) * 2 + ((d & 7) >> 2)) * 4 + (d & 3);
}

__device__ __forceinline__ float bp_lookup(const int32_t* __restrict__ cols, const float* __restrict__ raw, int n, int32_t j) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cols[mid] < j) lo = mid + 1; else hi = mid;
    }
    return (lo < n && cols[lo] == j) ? raw[lo] : 0.f;
}

template <bool FILL>
__global__ void __launch_bounds__(256) k_bp_synth(uint32_t* bm, int SK, int64_t Npad, int64_t N, int64_t S, int64_t nrows_pad, const int64_t* __restrict__ parents,
                                                  const int64_t* __restrict__ rip, const int32_t* rcols, const float* __restrict__ rraw, int32_t* __restrict__ cnt,
                                                  const float* __restrict__ tab, const double* __restrict__ lib64, const float* __restrict__ med, float pc, int use_log1p,
                                                  int32_t* cols_out, float* __restrict__ x_out) {
    extern __shared__ __align__(16) uint32_t bp_syn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = SK * 8;
    const int64_t s = (int64_t)blockIdx.x * 4 + wave;
    if (s >= nrows_pad) return;
    const int64_t pr_out = Npad + s;
    if (s >= S) {                                                     // padding rows of the last tiles: no bits
        if (!FILL) for (int d = lane; d < W; d += 64) bm[bp_word_of(pr_out, d, SK)] = 0u;
        return;
    }
    uint32_t* b0 = bp_syn + (size_t)wave * 5 * W;
    uint32_t* b1 = b0 + W;
    uint32_t* r0 = b1 + W;
    uint32_t* r1 = r0 + W;
    uint32_t* pre = r1 + W;
    const int64_t p0 = parents[2 * s], p1 = parents[2 * s + 1];
    for (int d = lane; d < W; d += 64) {
        b0[d] = bm[bp_word_of(p0, d, SK)];
        b1[d] = bm[bp_word_of(p1, d, SK)];
        r0[d] = 0u;
        r1[d] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t ra0 = rip[p0], ra1 = rip[p1];
    const int rn0 = (int)(rip[p0 + 1] - ra0), rn1 = (int)(rip[p1 + 1] - ra1);
    for (int e = lane; e < rn0; e += 64)
        if (rraw[ra0 + e] != 0.f) { const int32_t c = rcols[ra0 + e]; atomicOr(&r0[c >> 5], 1u << (c & 31)); }
    for (int e = lane; e < rn1; e += 64)
        if (rraw[ra1 + e] != 0.f) { const int32_t c = rcols[ra1 + e]; atomicOr(&r1[c >> 5], 1u << (c & 31)); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto reduced_word = [&](int d) {
        const uint32_t nz0 = b0[d] | r0[d], nz1 = b1[d] | r1[d];
        return (nz0 & nz1) | (r0[d] & ~nz1) | (r1[d] & ~nz0);
    };
    int carry = 0;
    for (int k0 = 0; k0 < W; k0 += 64) {
        const int d = k0 + lane;
        uint32_t res = 0u;
        if (d < W) {
            res = reduced_word(d);
            if (!FILL) {
                const uint32_t nz0 = b0[d] | r0[d], nz1 = b1[d] | r1[d];
                bm[bp_word_of(pr_out, d, SK)] = (b0[d] & ~nz1) | (b1[d] & ~nz0);
            }
        }
        const int c = __popc(res);
        int x = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (FILL && d < W) pre[d] = (uint32_t)(carry + x - c);
        carry += __shfl(x, 63, 64);
    }
    if (!FILL) {
        if (lane == 0) cnt[s] = carry;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t row = N + s;
    const int64_t base = rip[row];
    const float m = med[0];
    auto emit = [&](int32_t j, float c) {
        const int d = j >> 5;
        const int slot = (int)pre[d] + __popc(reduced_word(d) & ((1u << (j & 31)) - 1u));
        cols_out[base + slot] = j;
        x_out[base + slot] = bp_value(c, row, tab, lib64, m, pc, use_log1p);
    };
    for (int e = lane; e < rn0; e += 64) {                            // parent 0's counts >= 2 (+ whatever parent 1 holds there)
        const float c = rraw[ra0 + e];
        if (c == 0.f) continue;
        const int32_t j = rcols[ra0 + e];
        const uint32_t bit = 1u << (j & 31);
        const float c1 = (b1[j >> 5] & bit) ? 1.f : ((r1[j >> 5] & bit) ? bp_lookup(rcols + ra1, rraw + ra1, rn1, j) : 0.f);
        emit(j, __fadd_rn(c, c1));
    }
    for (int e = lane; e < rn1; e += 64) {                            // parent 1's counts >= 2 where parent 0 holds 1 or nothing
        const float c = rraw[ra1 + e];
        if (c == 0.f) continue;
        const int32_t j = rcols[ra1 + e];
        const uint32_t bit = 1u << (j & 31);
        if (r0[j >> 5] & bit) continue;
        emit(j, __fadd_rn((b0[j >> 5] & bit) ? 1.f : 0.f, c));
    }
    for (int d = lane; d < W; d += 64) {                              // 1 + 1
        uint32_t ov = b0[d] & b1[d];
        while (ov) {
            const int b = __ffs(ov) - 1;
            ov &= ov - 1u;
            emit(d * 32 + b, 2.f);
        }
    }
}

// library sizes of the doublets from their parents' (exact: integers below 2^23, see k_counts_exact)
__global__ void k_bp_synth_libs(const int64_t* __restrict__ parents, int64_t N, int64_t S, float* __restrict__ lib32, double* __restrict__ lib64) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const float v = __fadd_rn(lib32[parents[2 * s]], lib32[parents[2 * s + 1]]);
    lib32[N + s] = v;
    lib64[N + s] = (double)v;
}

// s_i = x_i(1) - z, the float32 difference the sparse path forms for such an entry (exactly what the float64 difference rounds to)
__global__ void k_bp_row_scale(const float* __restrict__ tab, int tab_stride, float z, int64_t M, double* __restrict__ s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) s[i] = (double)(tab[i * tab_stride] - z);
}

// ---- per product: the operand as digits -------------------------------------------------------------------------------
// largest |w_r X[r][c]| per column (w == nullptr: ones).  A workgroup reduces its share of the rows in LDS; non-negative
// float64 values order like their bit patterns, so the workgroups combine by an integer atomicMax (exact in any order).
// cmax must be zeroed before.  L <= 64.
__global__ void __launch_bounds__(256) k_bp_colmax(const double* __restrict__ X, const double* __restrict__ wgt, int64_t R, int L, double* __restrict__ cmax) {
    __shared__ double red[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    double m = 0.0;
    if (c < L)
        for (int64_t r = (int64_t)blockIdx.x * 4 + q; r < R; r += (int64_t)gridDim.x * 4) {
            const double v = fabs(wgt ? wgt[r] * X[r * L + c] : X[r * L + c]);
            m = v > m ? v : m;                               // (NaN never wins: a NaN operand ends in the rank warning upstream)
        }
    red[q][c] = m;
    __syncthreads();
    if (q == 0 && c < L) {
        for (int o = 1; o < 4; ++o) m = red[o][c] > m ? red[o][c] : m;
        if (m > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(cmax) + c, (unsigned long long)__double_as_longlong(m));
    }
}

// shift of a column: |X 2^sh| <= 2^(8 ND - 2), so that the last balanced digit stays inside int8
__device__ __forceinline__ int bp_shift(double cmax, int ND) {
    if (!(cmax > 0.0) || !(cmax < 1e300)) return 0;
    int e;
    (void)frexp(cmax, &e);                                   // cmax = m 2^e, 0.5 <= m < 1
    return 8 * ND - 2 - e;
}

// digits in the B-operand layout of v_mfma_i32_32x32x32_i8:
//   qd[((sk * 8 + s) * NT + nt) * 64 + h * 32 + n] = 16 bytes e = 0..15: digit d of column c, f = nt * 32 + n = c * ND + d,
//   operand row k = sk * 256 + h * 128 + s * 16 + e  (ROWMAP: k is a padded row index of the augmented matrix)
// One thread per (sk, s, h, column slot): 16 operand values -> ND vectors.  Slots past L write the zero padding of the last tile.
template <int ND, bool ROWMAP>
__global__ void __launch_bounds__(256) k_bp_digits(const double* __restrict__ X, const double* __restrict__ wgt, int64_t R, int L, int NT, int nslot,
                                                   const double* __restrict__ cmax, int64_t SK, int64_t Npad, int64_t N, v4i* __restrict__ qd,
                                                   double* __restrict__ zero_me) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the column maxima of the OTHER operand side are collected next (by atomic maxima, into zeros): nobody reads the old ones any more
    if (zero_me && t < 64) zero_me[t] = 0.0;
    if (t >= SK * kBpSteps * 2 * nslot) return;
    const int slot = (int)(t % nslot);
    int64_t u = t / nslot;
    const int h = (int)(u & 1); u >>= 1;
    const int s = (int)(u & 7);
    const int64_t sk = u >> 3;
    unsigned out[ND][4] = {};
    if (slot < L) {
        const double scale = ldexp(1.0, bp_shift(cmax[slot], ND));
        const double lim = ldexp(1.0, 8 * ND - 2);
        for (int e = 0; e < 16; ++e) {
            const int64_t k = sk * kBpStageCols + h * 128 + s * 16 + e;
            const int64_t row = ROWMAP ? bp_row_of(k, Npad, N, R) : (k < R ? k : -1);
            int v = 0;
            if (row >= 0) {
                const double x = wgt ? wgt[row] * X[row * L + slot] : X[row * L + slot];
                const double q = rint(x * scale);
                v = (q >= -lim && q <= lim) ? (int)q : 0;          // (non-finite operands: zero; caught by the rank check upstream)
            }
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const int dg = d + 1 < ND ? ((v + 128) & 255) - 128 : v;    // balanced digits; the last one is what is left (|.| <= 65)
                v = (v - dg) >> 8;
                out[d][e >> 2] |= (unsigned)(uint8_t)(int8_t)dg << (8 * (e & 3));
            }
        }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int f = slot * ND + d;
        if (f < NT * 32) qd[((sk * kBpSteps + s) * NT + (f >> 5)) * 64 + h * 32 + (f & 31)] = v4i{(int)out[d][0], (int)out[d][1], (int)out[d][2], (int)out[d][3]};
    }
}

// 16 bits -> 16 bytes of 0 / 1 (the A operand of the MFMA)
__device__ __forceinline__ v4i bp_expand16(unsigned bits) {
    v4i r;
#pragma unroll
    for (int w = 0; w < 4; ++w) r[w] = (int)((((bits >> (4 * w)) & 0xfu) * 0x00204081u) & 0x01010101u);
    return r;
}

struct BpProductArgs {
    const v4i* bm;            // bitmap [tile][SK][64]
    const v4i* qd;            // digits [SK][8][NT][64]
    int64_t ntile;            // 32-row tiles of the bitmap
    int SK;                   // stages of the k dimension in use
    int64_t SKstride;         // stages per tile in the bitmap's layout (>= SK)
    int sk_per_chunk;         // stages per chunk (grid.y chunks)
    int L;                    // sketch columns
    const double* cmax;       // [64] largest |operand| per column: the value of a digit sum is V 2^-shift(cmax)
    // ROWS (A Q): output row = bp_row_of(tile * 32 + r); out[row][c] = srow[row] * value
    const double* srow; int64_t Npad, N, M;
    // COLS (A^T Y): output row = tile * 32 + r < nOut; out[(chunk * nOut + row)][c] = value
    int64_t nOut;
    double* out;
    int dbg;                  // timing experiments (option bp_dbg_mode; wrong results): 1 no digit copies inside the loop, 2 no bitmap copies, 4 no matrix instructions
};

// S = B . digits for RT tiles of 32 bitmap rows per wave, over the stages of this workgroup's chunk; see the file header.
template <int RT, int NT, int ND, bool ROWS>
__global__ void __launch_bounds__(64 * kBpWaves) k_bp_product(const BpProductArgs a) {
    constexpr int kVecs = kBpSteps * NT * 64;                    // 16-byte vectors of digits per stage
    constexpr int kPieces = kVecs / 64;                          // wave-wide copies (1 KB each) per stage
    constexpr int kLo = kPieces / kBpWaves, kExtra = kPieces % kBpWaves;   // waves < kExtra carry one more piece
    extern __shared__ __align__(16) unsigned char bp_smem[];
    v4i* lds = reinterpret_cast<v4i*>(bp_smem);                  // [kBpStages][kVecs]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tile0 = ((int64_t)blockIdx.x * kBpWaves + wave) * RT;
    const int chunk = blockIdx.y;
    const int sk0 = chunk * a.sk_per_chunk;
    const int sk1 = sk0 + a.sk_per_chunk < a.SK ? sk0 + a.sk_per_chunk : a.SK;
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
        const int64_t tile = tile0 + t < a.ntile ? tile0 + t : a.ntile - 1;
        bmp[t] = a.bm + tile * a.SKstride * 64 + (lane & 31) * 2 + (lane >> 5);       // lane (h, r) reads word r*2 + h
    }
    auto stage = [&](int sk, int st) {                           // (a stage past the end re-reads the last one: harmless, same count)
        const int k = sk < sk1 ? sk : sk1 - 1;
#pragma unroll
        for (int u = 0; u <= kLo; ++u) {
            const int piece = u * kBpWaves + wave;
            if (u < kLo || wave < kExtra)                         // (wave-uniform)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.qd + (int64_t)k * kVecs + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(lds + st * kVecs + piece * 64), 16, 0, 0);
        }
    };
    auto bits = [&](int sk, v4i (&w)[RT]) {
        const int k = sk < sk1 ? sk : sk1 - 1;
#pragma unroll
        for (int t = 0; t < RT; ++t) w[t] = bmp[t][(int64_t)k * 64];
    };
    v4i w0[RT], w1[RT];
    bits(sk0, w0);
    stage(sk0, 0);
    stage(sk0 + 1, 1);
#pragma unroll 1
    for (int sk = sk0; sk < sk1; ++sk) {
        const int it = sk - sk0;
        // the copies of stage sk have landed: all but the newest stage's copies are complete (a wave's memory operations complete in
        // issue order; the bit loads were issued before those copies)
        if (wave < kExtra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLo + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLo) : "memory");
        __syncthreads();                                          // ... for every wave; and everybody is done with stage sk - 1's buffer
        bits(sk + 1, w1);
        stage(sk + 2, (it + 2) % kBpStages);
        const v4i* cur = lds + (it % kBpStages) * kVecs + lane;
        // operand fragments are read PF groups ahead of the MFMAs that use them
        constexpr int PF = 2, NG = kBpSteps * NT;
        v4i bq[PF + 1];
#pragma unroll
        for (int p = 0; p < PF; ++p) bq[p] = cur[p * 64];
        v4i av[RT], an[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) { av[t] = bp_expand16((unsigned)w0[t][0] & 0xffffu); an[t] = v4i{0, 0, 0, 0}; }
        // the next k-step's bit expansion (4 RT words of four bytes) is spread over this k-step's NT groups of matrix instructions, order
        // pinned, instead of one burst in front of the step's first group.  Worth 1 % here (0.159 -> 0.158 ms, profiles/r06j_expand_ab_kernels.txt):
        // this kernel is power-bound, not issue-bound -- the same change was worth 7 % in the MX kernel, whose two waves per SIMD added
        // their vector and matrix time (profiles/r06_mx_notes.txt)
        constexpr int kWordsPerGroup = (4 * RT + NT - 1) / NT;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int s = gi / NT, c = gi % NT;
            if (gi + PF < NG) bq[(gi + PF) % (PF + 1)] = cur[(gi + PF) * 64];
            const v4i b = bq[gi % (PF + 1)];
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[t], b, acc[t][c], 0, 0, 0);
            if (s + 1 < kBpSteps) {
#pragma unroll
                for (int wi = c * kWordsPerGroup; wi < (c + 1) * kWordsPerGroup && wi < 4 * RT; ++wi) {
                    const int t = wi >> 2, w = wi & 3;
                    const unsigned bits16 = ((unsigned)w0[t][(s + 1) >> 1] >> (16 * ((s + 1) & 1))) & 0xffffu;
                    an[t][w] = (int)((((bits16 >> (4 * w)) & 0xfu) * 0x00204081u) & 0x01010101u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (c == NT - 1) {
#pragma unroll
                for (int t = 0; t < RT; ++t) av[t] = an[t];
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) w0[t] = w1[t];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the copies issued past the end)
    __syncthreads();                                              // the stage buffers become the epilogue's scratch
    // Epilogue.  C/D layout of the 32 x 32 tile: column n = lane & 31, register i holds row (i / 4) * 8 + (lane >> 5) * 4 + (i % 4).
    // A band of 8 rows (registers 4b .. 4b+3 of both lane halves) goes through this wave's LDS scratch as int32 [8][F + 4]; every lane
    // then recombines the ND digits of its (row, column) outputs: V = sum_d 256^d S_d, an exact float64 integer.
    constexpr int F = NT * 32, FS = F + 4;
    int32_t* scr = reinterpret_cast<int32_t*>(bp_smem) + wave * (8 * FS);
    const int n = lane & 31, hh = lane >> 5;
    // output o = lane + 64 i of a band is (row o / L, column o % L): this lane's columns and their scales 2^-shift
    constexpr int kOutPerLane = (8 * 40 + 63) / 64;
    double myscale[kOutPerLane];
#pragma unroll
    for (int i = 0; i < kOutPerLane; ++i) {
        const int o = lane + 64 * i;
        myscale[i] = o < 8 * a.L ? ldexp(1.0, -bp_shift(a.cmax[o % a.L], ND)) : 0.0;
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t tile = tile0 + t;
        if (tile >= a.ntile) continue;                            // (wave-uniform)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) scr[(hh * 4 + q) * FS + c * 32 + n] = acc[t][c][b * 4 + q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < kOutPerLane; ++i) {
                const int o = lane + 64 * i;
                if (o >= 8 * a.L) break;
                const int rr = o / a.L, col = o - rr * a.L;
                const int32_t* dp = scr + rr * FS + col * ND;
                double V = (double)dp[ND - 1];
#pragma unroll
                for (int d = ND - 2; d >= 0; --d) V = V * 256.0 + (double)dp[d];      // exact: integers below 2^53
                const double val = V * myscale[i];                                   // a power of two: exact
                const int64_t pr = tile * 32 + b * 8 + rr;
                if (ROWS) {
                    const int64_t row = bp_row_of(pr, a.Npad, a.N, a.M);
                    if (row >= 0) a.out[row * a.L + col] = a.srow[row] * val;
                } else if (pr < a.nOut) {
                    a.out[((int64_t)chunk * a.nOut + pr) * a.L + col] = val;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- the same products on the MX matrix instruction of gfx950 (round 6) -----------------------------------------------------------
// v_mfma_f32_32x32x64_f8f6f4 takes 4-bit and 6-bit floating-point operands at twice the int8 rate on paper and, measured under random digits,
// at 2.6 x the rate the power-bound int8 kernel reaches (profiles/tools/mfma_fp6_probe.hip: 3.6 against 1.36 P MAC/s).  Its small floats
// hold small integers EXACTLY:
//   * the bitmap is the A operand in FP4 (E2M1): bit 0 -> code 0 = 0, bit 1 -> code 1 = 0.5;
//   * the operand is cut into SIX balanced digits of base 31, d in [-15, 15], the B operand in FP6 (E2M3): the code |d| | sign << 5 is
//     the value d / 8 (subnormals k / 8 for k < 8, then 1 + m / 8) -- 31^6 / 2 = 2^28.7 levels below the column's largest element
//     (four int8 digits: 2^30, three: 2^22);
//   * products 0.5 x d / 8 and their float32 sums are multiples of 1 / 16 below 2^24 / 16 (|sum| <= 15 K / 16, K <= 2^20 per chunk):
//     exact in any order; the epilogue reads them back as integers and recombines V = sum_d 31^d S_d in float64 (< 2^53), as the int8 form does.
// Geometry: 40 columns x 6 digits are split into four quarters of 10 columns = 60 of 64 flattened columns = two 32-wide tiles; a wave owns
// 4 row tiles x the 2 tiles of one quarter (128 accumulator registers, two waves per SIMD) -- every operand fragment read from LDS feeds four
// matrix instructions (two would leave the LDS as busy as the matrix pipe); a workgroup of 8 waves = 2 row groups x 4 quarters = 256 rows.
// A stage of 256 k values is four k-steps of 64; its operand fragments (24 bytes per lane and tile: 16 + 8, so that every LDS read is
// aligned) are 48 KB, three stages 144 KB; the workgroup's 8 KB of bitmap per stage go through LDS as well (a ring of two: 160 KB in all)
// so that the loop holds no ordinary global load -- the compiler answers one with s_waitcnt vmcnt(0), which would drain the copies in flight.
// Inside a k-step's 32 values per lane half the order is permuted (kF6Perm): the A operand's nibble pairs are then picked by selectors
// (w >> 2 j) & 0x03030303 -- three instructions per eight bits.
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kF6Digits = 6;
constexpr int kF6Quarter = 10;                                 // sketch columns per quarter
constexpr int kF6Steps = 4;                                    // k-steps of 64 per stage
constexpr int kF6StepBytes = 8 * 1024 + 8 * 512;               // per k-step: eight tiles x (64 lanes x 16 bytes) | eight tiles x (64 lanes x 8 bytes)
constexpr int kF6StageBytes = kF6Steps * kF6StepBytes;         // 48 KB
constexpr int kF6StagePieces = kF6StageBytes / 1024;           // wave-wide 1 KB copies per stage
constexpr int kF6BitsBytes = kBpWaves * 1024;                  // the workgroup's eight (tile, stage) KB of bitmap
constexpr int kF6Lds = kBpStages * kF6StageBytes + 2 * kF6BitsBytes;
constexpr double kF6Lim = 443751840.0;                         // (31^6 - 1) / 2: the largest magnitude six balanced digits hold
constexpr double kF6Scale = kF6Lim * (1.0 - 1.0 / 1048576.0);  // what the column's largest element is scaled to
// element e of a lane's 32 (nibble e of the A operand, field e of the B operand) is bit kF6Perm(e) of the bitmap word
__host__ __device__ constexpr int kF6Perm(int e) { return 8 * ((e >> 1) & 3) + 2 * (e >> 3) + (e & 1); }

// operand -> digits in the B-operand layout: lane (h, n) of tile nt holds, for k-step s of stage sk, the 32 six-bit codes of
// k = sk * 256 + h * 128 + s * 32 + kF6Perm(e) and flattened column nt * 32 + n = quarter * 64 + (column % 10) * 6 + digit.
// One workgroup per stage: thread (s, h, column) cuts 32 operand values into 6 fragments of 24 bytes inside an LDS image of the stage,
// which then leaves in full lines.
template <bool ROWMAP>
__global__ void __launch_bounds__(320) k_bp_digits6(const double* __restrict__ X, const double* __restrict__ wgt, int64_t R, int L, const double* __restrict__ cmax,
                                                    int64_t Npad, int64_t N, unsigned char* __restrict__ qd, double* __restrict__ zero_me) {
    extern __shared__ __align__(16) unsigned char d6_img[];            // [kF6StageBytes]
    const int64_t sk = blockIdx.x;
    const int tid = threadIdx.x;
    if (zero_me && sk == 0 && tid < 64) zero_me[tid] = 0.0;
    for (int i = tid; i < kF6StageBytes / 16; i += 320) reinterpret_cast<v4i*>(d6_img)[i] = v4i{0, 0, 0, 0};      // (padding columns, columns past L)
    __syncthreads();
    const int col = tid % 40, sh = tid / 40, h = sh & 1, s = sh >> 1;
    if (col < L) {
        unsigned out[kF6Digits][6] = {};
        const double cm = cmax[col];
        const double scale = (cm > 0.0 && cm < 1e300) ? kF6Scale / cm : 0.0;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int64_t k = sk * kBpStageCols + h * 128 + s * 32 + kF6Perm(e);
            const int64_t row = ROWMAP ? bp_row_of(k, Npad, N, R) : (k < R ? k : -1);
            unsigned u = (unsigned)kF6Lim;                               // (the digits of zero)
            if (row >= 0) {
                const double x = wgt ? wgt[row] * X[row * L + col] : X[row * L + col];
                const double q = rint(x * scale);
                // (a column maximum collected by another kernel may differ from |x| here in the last bits: the scale keeps 2^-20 of
                //  headroom and the end of the range clamps; non-finite operands: zero, caught by the rank check upstream)
                if (fabs(q) < 1e300) u = (unsigned)((int)fmin(fmax(q, -kF6Lim), kF6Lim) + (int)kF6Lim);
            }
            // v + lim = sum_d (digit_d + 15) 31^d: the balanced digits are the base-31 digits of the shifted value, less 15 each
#pragma unroll
            for (int d = 0; d < kF6Digits; ++d) {
                const unsigned qv = u / 31u;
                const int dg = (int)(u - 31u * qv) - 15;
                u = qv;
                const unsigned code = dg < 0 ? (32u | (unsigned)(-dg)) : (unsigned)dg;
                const int bit0 = 6 * e;
                out[d][bit0 >> 5] |= code << (bit0 & 31);
                if ((bit0 & 31) > 26) out[d][(bit0 >> 5) + 1] |= code >> (32 - (bit0 & 31));
            }
        }
        unsigned char* base = d6_img + s * kF6StepBytes;
        const int quarter = col / kF6Quarter, cl = col - quarter * kF6Quarter;
#pragma unroll
        for (int d = 0; d < kF6Digits; ++d) {
            const int f = cl * kF6Digits + d;
            const int nt = quarter * 2 + (f >> 5), lane = h * 32 + (f & 31);
            *reinterpret_cast<v4i*>(base + nt * 1024 + lane * 16) = v4i{(int)out[d][0], (int)out[d][1], (int)out[d][2], (int)out[d][3]};
            *reinterpret_cast<v2i*>(base + 8 * 1024 + nt * 512 + lane * 8) = v2i{(int)out[d][4], (int)out[d][5]};
        }
    }
    __syncthreads();
    v4i* dst = reinterpret_cast<v4i*>(qd + sk * (int64_t)kF6StageBytes);
    for (int i = tid; i < kF6StageBytes / 16; i += 320) dst[i] = reinterpret_cast<const v4i*>(d6_img)[i];
}

// 32 bits (in the permuted order) -> 32 FP4 codes, 0 or 1 = 0.5: byte m of the selector (w >> 2 j) & 0x03030303 holds the bits of
// nibbles 2 m and 2 m + 1 of word j and picks the byte 0x00 / 0x01 / 0x10 / 0x11
__device__ __forceinline__ v8i bp_expand_fp4(unsigned w) {
    v8i r = {};
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (int)__builtin_amdgcn_perm(0x11100100u, 0x11100100u, (w >> (2 * j)) & 0x03030303u);
    return r;
}

template <bool ROWS, int DBG = 0>
__global__ void __launch_bounds__(64 * kBpWaves) k_bp_product6(const BpProductArgs a) {
    constexpr int RT = 4, NTW = 2;
    constexpr int kPer = kF6StagePieces / kBpWaves;               // 1 KB copies of digits per wave and stage (+ one of bitmap)
    static_assert(kF6StagePieces % kBpWaves == 0, "");
    extern __shared__ __align__(16) unsigned char bp_smem[];
    unsigned char* const bits_ring = bp_smem + kBpStages * kF6StageBytes;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, quarter = wave & 3;
    const int64_t wg_tile0 = (int64_t)blockIdx.x * kBpWaves;      // the workgroup's eight row tiles: two groups of four
    const int64_t tile0 = wg_tile0 + grp * RT;
    const int chunk = blockIdx.y;
    const int sk0 = chunk * a.sk_per_chunk;
    const int sk1 = sk0 + a.sk_per_chunk < a.SK ? sk0 + a.sk_per_chunk : a.SK;
    v16f acc[RT][NTW];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NTW; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
    const unsigned char* qd = reinterpret_cast<const unsigned char*>(a.qd);
    // this wave copies the bitmap KB of the workgroup's tile number `wave` (a tile past the end re-reads the last one: its rows are not stored)
    const v4i* my_bm = a.bm + (wg_tile0 + wave < a.ntile ? wg_tile0 + wave : a.ntile - 1) * a.SKstride * 64 + lane;
    auto stage = [&](int sk, int st) {                           // (a stage past the end re-reads the last one: harmless, same count)
        const int k = sk < sk1 ? sk : sk1 - 1;
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int piece = u * kBpWaves + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qd + (int64_t)k * kF6StageBytes + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(bp_smem + st * kF6StageBytes + piece * 1024), 16, 0, 0);
        }
    };
    auto bits = [&](int sk, int st) {
        const int k = sk < sk1 ? sk : sk1 - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(my_bm + (int64_t)k * 64),
                                         (__attribute__((address_space(3))) void*)(bits_ring + st * kF6BitsBytes + wave * 1024), 16, 0, 0);
    };
    stage(sk0, 0);
    bits(sk0, 0);
    stage(sk0 + 1, 1);
    const int word_off = ((lane & 31) * 2 + (lane >> 5)) * 16;   // lane (h, r) reads word r * 2 + h of a tile's KB
#pragma unroll 1
    for (int sk = sk0; sk < sk1; ++sk) {
        const int it = sk - sk0;
        // stage sk's digits and bitmap have landed once only the newest stage's copies are in flight (a wave's copies complete in issue
        // order); everybody is done with the buffers about to be refilled.  A bare barrier: __syncthreads() would wait for every copy.
        if constexpr (DBG != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kPer) : "memory");
        if constexpr (!(DBG & 2)) bits(sk + 1, (it + 1) & 1);
        // (the stage's digit copies are issued one at a time between the matrix instructions below: as a block here they and their address
        //  arithmetic ran while the matrix pipe idled -- 0.25 us of a stage's 1.65)
        const int k_next = sk + 2 < sk1 ? sk + 2 : sk1 - 1;
        const unsigned char* const src_next = qd + (int64_t)k_next * kF6StageBytes + wave * 1024 + lane * 16;
        unsigned char* const dst_next = bp_smem + ((it + 2) % kBpStages) * kF6StageBytes + wave * 1024;
        v4i w0[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) w0[t] = *reinterpret_cast<const v4i*>(bits_ring + (it & 1) * kF6BitsBytes + (grp * RT + t) * 1024 + word_off);
        const unsigned char* cur = bp_smem + (it % kBpStages) * kF6StageBytes + (quarter * NTW) * 1024 + lane * 16;
        const unsigned char* cur8 = bp_smem + (it % kBpStages) * kF6StageBytes + 8 * 1024 + (quarter * NTW) * 512 + lane * 8;
        constexpr int PF = 2, NG = kF6Steps * NTW;
        v4i bq[PF + 1];
        v2i bq8[PF + 1];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            bq[p] = *reinterpret_cast<const v4i*>(cur + (p / NTW) * kF6StepBytes + (p % NTW) * 1024);
            bq8[p] = *reinterpret_cast<const v2i*>(cur8 + (p / NTW) * kF6StepBytes + (p % NTW) * 512);
        }
        v8i av[RT], an[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) { av[t] = bp_expand_fp4((unsigned)w0[t][0]); an[t] = v8i{}; }
        // Every matrix instruction is followed by its share of the NEXT k-step's bit expansion (half a tile: six vector instructions) and the
        // order is pinned: a wave then keeps the matrix pipe busy by itself -- the 32 cycles an instruction occupies it cover the 24 of the
        // vector work behind it.  Left to the scheduler the expansions of a step came in one burst of 44 in front of its eight matrix
        // instructions, the two waves of a SIMD (in step after every barrier) burst together, and vector and matrix time added up
        // (profiles/tools/mx_ablation.py: 1.8 us per stage for 0.6 + 1.0).
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int s = gi / NTW, c = gi % NTW;
            if (gi + PF < NG) {
                const int g2 = gi + PF;
                bq[g2 % (PF + 1)] = *reinterpret_cast<const v4i*>(cur + (g2 / NTW) * kF6StepBytes + (g2 % NTW) * 1024);
                bq8[g2 % (PF + 1)] = *reinterpret_cast<const v2i*>(cur8 + (g2 / NTW) * kF6StepBytes + (g2 % NTW) * 512);
            }
            const v4i b4 = bq[gi % (PF + 1)];
            const v2i b2 = bq8[gi % (PF + 1)];
            const v8i b = v8i{b4[0], b4[1], b4[2], b4[3], b2[0], b2[1], 0, 0};
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                if constexpr (!(DBG & 4)) acc[t][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av[t], b, acc[t][c], 4, 2, 0, 0, 0, 0);
                else acc[t][c][0] += (float)(av[t][0] ^ av[t][3] ^ b[0] ^ b[5]);
                if constexpr (!(DBG & 1)) {
                    constexpr int kEvery = (NG * RT) / kPer;                      // matrix instructions per copy
                    const int m = gi * RT + t;
                    if (m % kEvery == 1 && m / kEvery < kPer) {
                        const int u = m / kEvery;                                  // piece u * kBpWaves + wave
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_next + u * kBpWaves * 1024),
                                                         (__attribute__((address_space(3))) void*)(dst_next + u * kBpWaves * 1024), 16, 0, 0);
                    }
                }
                if (s + 1 < kF6Steps) {
                    // half of tile (c * 2 + t / 2)'s next fragment: words 2 (t & 1) and 2 (t & 1) + 1
                    const int tile = c * (RT / NTW) + (t >> 1), j0 = 2 * (t & 1);
                    const unsigned w = (unsigned)w0[tile][s + 1];
                    an[tile][j0] = (int)__builtin_amdgcn_perm(0x11100100u, 0x11100100u, (w >> (2 * j0)) & 0x03030303u);
                    an[tile][j0 + 1] = (int)__builtin_amdgcn_perm(0x11100100u, 0x11100100u, (w >> (2 * j0 + 2)) & 0x03030303u);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c == NTW - 1) {
#pragma unroll
                for (int t = 0; t < RT; ++t) av[t] = an[t];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (the copies issued past the end); the stage buffers become the epilogue's scratch
    // Epilogue: as k_bp_product's (same C/D layout), per quarter: 8 rows x 10 columns per band, six digits of base 31 each
    constexpr int F = NTW * 32, FS = F + 4;
    int32_t* scr = reinterpret_cast<int32_t*>(bp_smem) + wave * (8 * FS);
    const int n = lane & 31, hh = lane >> 5;
    constexpr int kOutPerLane = (8 * kF6Quarter + 63) / 64;
    double myscale[kOutPerLane];
#pragma unroll
    for (int i = 0; i < kOutPerLane; ++i) {
        const int o = lane + 64 * i;
        const int col = quarter * kF6Quarter + o % kF6Quarter;
        const double cm = (o < 8 * kF6Quarter && col < a.L) ? a.cmax[col] : 0.0;
        myscale[i] = (cm > 0.0 && cm < 1e300) ? cm / kF6Scale : 0.0;
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t tile = tile0 + t;
        if (tile >= a.ntile) continue;                            // (wave-uniform)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int c = 0; c < NTW; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) scr[(hh * 4 + q) * FS + c * 32 + n] = (int32_t)(acc[t][c][b * 4 + q] * 16.f);     // exact: a multiple of 1 / 16
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < kOutPerLane; ++i) {
                const int o = lane + 64 * i;
                if (o >= 8 * kF6Quarter) break;
                const int rr = o / kF6Quarter, cl = o - rr * kF6Quarter;
                const int col = quarter * kF6Quarter + cl;
                if (col >= a.L) continue;
                const int32_t* dp = scr + rr * FS + cl * kF6Digits;
                double V = (double)dp[kF6Digits - 1];
#pragma unroll
                for (int d = kF6Digits - 2; d >= 0; --d) V = V * 31.0 + (double)dp[d];          // exact: integers below 2^53
                const double val = V * myscale[i];
                const int64_t pr = tile * 32 + b * 8 + rr;
                if (ROWS) {
                    const int64_t row = bp_row_of(pr, a.Npad, a.N, a.M);
                    if (row >= 0) a.out[row * a.L + col] = a.srow[row] * val;
                } else if (pr < a.nOut) {
                    a.out[((int64_t)chunk * a.nOut + pr) * a.L + col] = val;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static size_t bp_align(size_t v) { return (v + 255) & ~(size_t)255; }
static int bp_workspace(ddx_ctx* ctx);
static int bp_col_chunks(const BitPlanes& bp, int64_t SK, int* per_out, int tiles_per_wg = kBpWaves * 2);
struct BpProductArgs;
template <int RT, int NT, int ND, bool ROWS> static int bp_launch_t(ddx_ctx* ctx, const BpProductArgs& a, int chunks);

static int bp_launch_bitmaps(ddx_ctx* ctx, int64_t tile0, int64_t ntiles, bool rows_too = true) {
    BitPlanes& bp = ctx->bp;
    if (ntiles <= 0) return DDX_OK;
    const int sk_chunk = std::min(bp.SKc, 48);                    // 32 rows x 48 stages x 32 bytes = 48 KB of LDS
    if (rows_too)
    k_bp_rows_bitmap<<<(unsigned)ntiles, 256, (size_t)32 * sk_chunk * 32, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_indices.as<int32_t>(),
                                                                                        ctx->aug_raw.as<float>(), tile0, bp.Npad, ctx->N, ctx->M, bp.SKc, sk_chunk, ctx->bp_demote.as<uint8_t>(),
                                                                                        reinterpret_cast<v4i*>(bp.bm_rows));
    const int64_t nblk_r = ntiles / 2 + (ntiles & 1);             // 64-row blocks (tile0 is even: Npad is a multiple of 256)
    const int64_t nblk = nblk_r * bp.SKc * 4;
    k_bp_transpose<<<(unsigned)ceil_div(nblk, 4), 256, 0, ctx->stream>>>(reinterpret_cast<const v4i*>(bp.bm_rows), tile0 * 32, nblk_r, bp.SKc, bp.SKr, (int32_t)(bp.ntile_c * 32),
                                                                         reinterpret_cast<v4i*>(bp.bm_cols));
    return DDX_OK;
}

// flags -> scan -> compaction of n entries (raw values `raw`, indices `idx`); returns the kept count in *kept (synchronises)
static int bp_reduce(ddx_ctx* ctx, const float* raw, const int32_t* idx, const float* x, int64_t n, int32_t* idx_out, int32_t* pos_out, float* x_out,
                     int64_t cap_out, int32_t* kept, bool count_only, int32_t** scan_out, const int32_t* cols) {
    DDX_TRY(ensure(ctx, ctx->sort_keys_in, sizeof(int32_t) * (size_t)(n + 1)));
    DDX_TRY(ensure(ctx, ctx->sort_keys_out, sizeof(int32_t) * (size_t)(n + 1)));
    int32_t* flag = ctx->sort_keys_in.as<int32_t>();
    int32_t* scan = ctx->sort_keys_out.as<int32_t>();
    size_t tmp = 0;
    DDX_HIP(ctx, prim::exclusive_sum(nullptr, tmp, flag, scan, (size_t)(n + 1), ctx->stream));
    DDX_TRY(ensure(ctx, ctx->sort_tmp, tmp));
    const unsigned ge = (unsigned)ceil_div(n + 1, 256);
    k_bp_flags<<<ge, 256, 0, ctx->stream>>>(raw, cols, ctx->bp_demote.as<uint8_t>(), n, flag);
    DDX_HIP(ctx, prim::exclusive_sum(ctx->sort_tmp.p, tmp, flag, scan, (size_t)(n + 1), ctx->stream));
    if (scan_out) *scan_out = scan;
    if (count_only) {
        DDX_HIP(ctx, hipMemcpyAsync(kept, scan + n, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        DDX_HIP(ctx, wait_stream(ctx));
        return DDX_OK;
    }
    (void)cap_out;
    k_bp_compact<<<ge, 256, 0, ctx->stream>>>(flag, scan, idx, x, n, idx_out, pos_out, x_out);
    return DDX_OK;
}

// Builds what lasts for a fit: the geometry, the original cells' bitmaps (by rows and by columns) and the reduced structures of
// their other entries (with positions in the full arrays, for the per-iteration value refresh).  Once per fit and context.
int bp_build(ddx_ctx* ctx) {
    BitPlanes& bp = ctx->bp;
    if (bp.ready) return DDX_OK;
    const int64_t N = ctx->N;
    const int32_t H = ctx->H;
    const int64_t nnz = ctx->nnz;
    ScopedTimer t(ctx, "bitplane_build");
    bp.SKc = (int)ceil_div(H, kBpStageCols);
    bp.Npad = ceil_div(N, kBpStageCols) * kBpStageCols;
    bp.ntile_o = bp.Npad / 32;
    const int64_t Scap = N / 2 + 2;                               // room for the synthetic rows (grows on demand in bp_refresh)
    bp.cap_rows = bp.Npad + ceil_div(Scap, kBpStageCols) * kBpStageCols;
    bp.ntile_c = ceil_div(H, 32);
    const int64_t nseg = (int64_t)ctx->P_o * H;                   // (panel, column) segments of the originals' mirror
    // pass 1: how many entries other than 1
    int32_t total_r = 0;
    int32_t* scan = nullptr;
    // the demoted columns of this fit (scaled matrices; none before the first scaling has chosen them)
    DDX_TRY(ensure(ctx, ctx->bp_demote, (size_t)H + 256));
    if ((int64_t)bp.demote.size() == H && bp.n_demoted > 0)
        DDX_HIP(ctx, hipMemcpyAsync(ctx->bp_demote.p, bp.demote.data(), (size_t)H, hipMemcpyHostToDevice, ctx->stream));
    else
        DDX_HIP(ctx, hipMemsetAsync(ctx->bp_demote.p, 0, (size_t)H, ctx->stream));
    DDX_TRY(bp_reduce(ctx, ctx->aug_raw.as<float>(), nullptr, nullptr, nnz, nullptr, nullptr, nullptr, 0, &total_r, true, &scan, ctx->aug_indices.as<int32_t>()));
    bp.nrest_o = total_r;
    bp.cap_rest_s = std::max<int64_t>(std::max<int64_t>(ctx->cap_synth / 3, bp.want_rest_s), 1024);   // (a synthetic row keeps ~15 % of its entries)
    // one buffer for everything that lasts for the fit
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += bp_align(bytes); return o; };
    const int64_t ntile_all = bp.cap_rows / 32, SKr_cap = bp.cap_rows / kBpStageCols;
    bp.SKr_cap = SKr_cap;
    const size_t o_bmr = carve(sizeof(v4i) * (size_t)ntile_all * bp.SKc * 64), o_bmc = carve(sizeof(v4i) * (size_t)bp.ntile_c * SKr_cap * 64);
    const int64_t cap_rest = total_r + bp.cap_rest_s;
    const size_t o_rip = carve(sizeof(int64_t) * (size_t)(N + Scap + 2)), o_rc = carve(sizeof(int32_t) * (size_t)cap_rest + 256), o_rx = carve(sizeof(float) * (size_t)cap_rest + 256);
    const size_t o_rp = carve(sizeof(int32_t) * (size_t)total_r + 256), o_rr = carve(sizeof(float) * (size_t)total_r + 256);
    const size_t o_mcp = carve(sizeof(int64_t) * (size_t)(nseg + 1)), o_mr = carve(sizeof(int32_t) * (size_t)total_r + 256), o_mp = carve(sizeof(float) * (size_t)total_r + 256);
    const size_t o_mx = carve(sizeof(float) * (size_t)total_r + 256), o_s = carve(sizeof(double) * (size_t)(N + Scap + 2));
    DDX_TRY(ensure(ctx, ctx->bp_buf, off));
    bp.buf_bytes = off;
    char* b = ctx->bp_buf.as<char>();
    bp.bm_rows = b + o_bmr;
    bp.bm_cols = b + o_bmc;
    bp.rest_indptr = reinterpret_cast<int64_t*>(b + o_rip);
    bp.rest_cols = reinterpret_cast<int32_t*>(b + o_rc);
    bp.rest_x = reinterpret_cast<float*>(b + o_rx);
    bp.rest_row = reinterpret_cast<int32_t*>(b + o_rp);
    bp.rest_raw = reinterpret_cast<float*>(b + o_rr);
    bp.restm_colptr = reinterpret_cast<int64_t*>(b + o_mcp);
    bp.restm_row = reinterpret_cast<int32_t*>(b + o_mr);
    bp.restm_raw = reinterpret_cast<float*>(b + o_mp);
    bp.restm_x = reinterpret_cast<float*>(b + o_mx);
    bp.srow = reinterpret_cast<double*>(b + o_s);
    bp.cap_srow = N + Scap + 2;
    bp.cap_rest = cap_rest;
    // the column bitmap's padding (rows between N and Npad, columns past H) must read as zeros; the synthetic stages are rewritten every iteration
    DDX_HIP(ctx, hipMemsetAsync(bp.bm_cols, 0, sizeof(v4i) * (size_t)bp.ntile_c * SKr_cap * 64, ctx->stream));
    DDX_HIP(ctx, hipMemsetAsync(bp.bm_rows, 0, sizeof(v4i) * (size_t)ntile_all * bp.SKc * 64, ctx->stream));
    const unsigned ge = (unsigned)ceil_div(nnz + 1, 256);
    int32_t* flag = ctx->sort_keys_in.as<int32_t>();
    // row-major (the scan of pass 1 is still in place)
    k_bp_compact<<<ge, 256, 0, ctx->stream>>>(flag, scan, ctx->aug_indices.as<int32_t>(), ctx->aug_raw.as<float>(), nnz, bp.rest_cols, nullptr, bp.rest_raw);
    k_bp_pointers<<<(unsigned)ceil_div(N + 1, 256), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), N + 1, 0, scan, 0, bp.rest_indptr);
    k_bp_rows_of<<<(unsigned)ceil_div(N, 4), 256, 0, ctx->stream>>>(bp.rest_indptr, N, bp.rest_row);
    // column-major mirror of the same entries ((panel, column, row) order), built from the reduced rows (its own timing scope:
    // scopes do not nest)
    timing_end(ctx);
    const int rc_m = bp_originals_mirror(ctx);
    timing_begin(ctx, "bitplane_build");
    DDX_TRY(rc_m);
    // bitmaps of the original rows.  SKr is provisional (no synthetic rows yet): the layout of the column bitmap uses the CAPACITY
    bp.SKr = bp.SKr_cap;
    const int64_t M_keep = ctx->M;
    ctx->M = ctx->N;                                              // (bp_row_of: no synthetic rows in this pass)
    DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_bp_rows_bitmap), 64 * 1024));
    const int rc = bp_launch_bitmaps(ctx, 0, bp.ntile_o);
    ctx->M = M_keep;
    DDX_TRY(rc);
    DDX_HIP(ctx, wait_stream(ctx));
    DDX_HIP(ctx, hipGetLastError());
    bp.ready = true;
    bp.values = false;
    return DDX_OK;
}

int bp_synth_libs(ddx_ctx* ctx) {
    ScopedTimer t(ctx, "row_sums");
    k_bp_synth_libs<<<(unsigned)ceil_div(ctx->S, 256), 256, 0, ctx->stream>>>(ctx->parents.as<int64_t>(), ctx->N, ctx->S, ctx->lib32.as<float>(), ctx->lib64.as<double>());
    return DDX_OK;
}

// The per-fit structures of another context of the same GPU, device to device (a follower of the fit: building them again
// would cost every context the passes over all stored entries).
int bp_clone(ddx_ctx* ctx, const CloneView& src) {
    ctx->bp = BitPlanes();
    if (!src.bp.ready) return DDX_OK;
    DDX_TRY(ensure(ctx, ctx->bp_buf, src.bp.buf_bytes));
    // (the parts of the buffer the source rewrites every iteration -- reduced values, row scales, the synthetic rows' bitmaps and reduced
    // entries -- may arrive torn: this context's own bp_refresh rewrites every one of them before anything reads them)
    DDX_HIP(ctx, hipMemcpyAsync(ctx->bp_buf.p, src.bp_buf, src.bp.buf_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    BitPlanes bp = src.bp;
    const ptrdiff_t delta = ctx->bp_buf.as<char>() - reinterpret_cast<const char*>(src.bp_buf);
    auto move = [&](auto*& p) { if (p) p = reinterpret_cast<std::remove_reference_t<decltype(p)>>(reinterpret_cast<char*>(p) + delta); };
    move(bp.bm_rows); move(bp.bm_cols); move(bp.rest_indptr); move(bp.rest_cols); move(bp.rest_row); move(bp.rest_raw); move(bp.rest_x);
    move(bp.restm_colptr); move(bp.restm_row); move(bp.restm_raw); move(bp.restm_x); move(bp.srow);
    bp.restm_s_colptr = nullptr; bp.restm_s_row = nullptr; bp.restm_s_x = nullptr;          // (views into the source's per-iteration buffers)
    bp.qd = nullptr; bp.cmax = nullptr; bp.part = nullptr; bp.ymax_of = nullptr;
    bp.values = false;
    bp.scaled = false; bp.inv_sd = nullptr; bp.rowop = nullptr;
    bp.ntile_s = 0; bp.nrest_s = 0;
    DDX_TRY(ensure(ctx, ctx->bp_demote, (size_t)src.H + 256));
    if ((int64_t)bp.demote.size() == src.H && bp.n_demoted > 0)
        DDX_HIP(ctx, hipMemcpyAsync(ctx->bp_demote.p, bp.demote.data(), (size_t)src.H, hipMemcpyHostToDevice, ctx->stream));
    else
        DDX_HIP(ctx, hipMemsetAsync(ctx->bp_demote.p, 0, (size_t)src.H, ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));          // (the host vector is a local copy)
    ctx->bp = bp;
    return DDX_OK;
}

// whether the counts just made resident should get their bit planes right away (the leader of a fit builds, its followers copy)
bool bp_wanted_at_upload(const ddx_ctx* ctx) {
    const Options& o = ctx->opt;
    return o.bitplane != 0 && o.gather_f32 && ctx->N >= 32 && (o.bitplane == 2 || ctx->N >= 4096);
}

// What changes with the iteration, after the row-major normalisation (called by ddx_lognormalise when the route is expected, else by
// the first PCA that takes it): the reduced values of the original rows (row-major: gathered from the full array; mirror: evaluated
// from the reduced mirror's own counts), the row scales, the synthetic rows' reduced CSR, their reduced mirror (built straight from
// it), their bitmaps -- and the column means, whose bit-plane part is one narrow product on the matrix cores.  The full column-major
// mirror of the iteration is not needed for any of it.
constexpr int kBpRetry = 1000;          // (internal) the per-fit structures were too small for this iteration: build them again and repeat
static int bp_refresh_once(ddx_ctx* ctx) {
    BitPlanes& bp = ctx->bp;
    if (bp.values) return DDX_OK;
    const int64_t N = ctx->N, M = ctx->M, S = ctx->S;
    if (bp.Npad + ceil_div(S, kBpStageCols) * kBpStageCols > bp.cap_rows || M + 2 > bp.cap_srow)
        return set_err(ctx, DDX_E_UNSUPPORTED, "bit planes: %lld synthetic rows exceed the planned capacity", (long long)S);
    bp.ntile_s = ceil_div(S, 32);
    bp.ntile_s += bp.ntile_s & 1;                                 // whole 64-row blocks for the transpose
    bp.SKr = bp.SKr_cap;
    bp.SKr_used = bp.Npad / kBpStageCols + ceil_div(S, kBpStageCols);
    DDX_TRY(bp_workspace(ctx));
    bp.nrest_s = 0;
    bp.scaled = false;
    bp.inv_sd = nullptr;
    ctx->pk_valid[0] = ctx->pk_valid[1] = false;             // the packed blocks of the sparse products describe the last iteration's matrix
    {
        ScopedTimer t(ctx, "bitplane_values");
        const int use_log1p = (ctx->pseudocount == 1.0f);
        if (bp.nrest_o > 0)
            k_bp_values<<<(unsigned)ceil_div(bp.nrest_o, 256), 256, 0, ctx->stream>>>(bp.rest_raw, bp.rest_row, bp.nrest_o, ctx->lognorm_tab.as<float>(), ctx->lib64.as<double>(),
                                                                                     ctx->median.as<float>(), ctx->pseudocount, use_log1p, bp.rest_x);
        k_bp_row_scale<<<(unsigned)ceil_div(M, 256), 256, 0, ctx->stream>>>(ctx->lognorm_tab.as<float>(), 16, ctx->zvalue, M, bp.srow);
        if (S > 0 && !ctx->synth_rows) {
            // the doublets' bitmap rows and reduced CSR straight from their parents' (k_bp_synth): count, scan, fill
            const int64_t nrows_pad = bp.ntile_s * 32;
            const size_t lds = sizeof(uint32_t) * 4 * 5 * (size_t)bp.SKc * 8;
            DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_bp_synth<false>), (int)lds));
            DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_bp_synth<true>), (int)lds));
            const unsigned g = (unsigned)ceil_div(nrows_pad, 4);
            int32_t* cnt = ctx->synth_counts.as<int32_t>();
            k_bp_synth<false><<<g, 256, lds, ctx->stream>>>(reinterpret_cast<uint32_t*>(bp.bm_rows), bp.SKc, bp.Npad, N, S, nrows_pad, ctx->parents.as<int64_t>(), bp.rest_indptr,
                                                            bp.rest_cols, bp.rest_raw, cnt, nullptr, nullptr, nullptr, 0.f, 0, nullptr, nullptr);
            DDX_TRY(scan_counts(ctx, cnt, S, bp.nrest_o, bp.rest_indptr + N));
            int64_t last = 0;
            DDX_HIP(ctx, hipMemcpyAsync(&last, bp.rest_indptr + M, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
            DDX_HIP(ctx, wait_stream(ctx));
            const int64_t kept = last - bp.nrest_o;
            if (kept > bp.cap_rest_s) {
                bp.want_rest_s = kept + kept / 2;
                bp.ready = false;
                return kBpRetry;
            }
            bp.nrest_s = kept;
            k_bp_synth<true><<<g, 256, lds, ctx->stream>>>(reinterpret_cast<uint32_t*>(bp.bm_rows), bp.SKc, bp.Npad, N, S, nrows_pad, ctx->parents.as<int64_t>(), bp.rest_indptr,
                                                           bp.rest_cols, bp.rest_raw, nullptr, ctx->lognorm_tab.as<float>(), ctx->lib64.as<double>(), ctx->median.as<float>(),
                                                           ctx->pseudocount, use_log1p, bp.rest_cols, bp.rest_x);
            DDX_TRY(bp_launch_bitmaps(ctx, bp.ntile_o, bp.ntile_s, false));       // (the transpose only)
        } else if (S > 0) {
            // synthetic rows: reduced CSR behind the original rows' (one array, one row pointer over all M rows)
            const int64_t e0 = ctx->nnz;
            const int64_t n_s = ctx->nnz_aug - e0;                     // (read back by ddx_lognormalise)
            int32_t kept = 0;
            int32_t* scan = nullptr;
            DDX_TRY(bp_reduce(ctx, ctx->aug_raw.as<float>() + e0, nullptr, nullptr, n_s, nullptr, nullptr, nullptr, 0, &kept, true, &scan, ctx->aug_indices.as<int32_t>() + e0));
            if (kept > bp.cap_rest_s) {
                // more entries other than 1 among the synthetic rows than planned for: build the per-fit structures again with room for them
                bp.want_rest_s = (int64_t)kept + kept / 2;
                bp.ready = false;
                return kBpRetry;
            }
            bp.nrest_s = kept;
            const unsigned ge = (unsigned)ceil_div(n_s + 1, 256);
            int32_t* flag = ctx->sort_keys_in.as<int32_t>();
            k_bp_compact<<<ge, 256, 0, ctx->stream>>>(flag, scan, ctx->aug_indices.as<int32_t>() + e0, ctx->aug_x.as<float>() + e0, n_s, bp.rest_cols + bp.nrest_o, nullptr,
                                                      bp.rest_x + bp.nrest_o);
            k_bp_pointers<<<(unsigned)ceil_div(S + 1, 256), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>() + N, S + 1, e0, scan, bp.nrest_o, bp.rest_indptr + N);
            // bitmaps of the synthetic rows (tiles behind the padded originals)
            DDX_TRY(bp_launch_bitmaps(ctx, bp.ntile_o, bp.ntile_s));
        }
    }
    // reduced mirrors: the synthetic rows' built from their reduced CSR (scope mirror_build), the originals' values (lognorm_cols)
    DDX_TRY(bp_reduced_mirrors(ctx));
    // column means: sums of the entries equal to 1 = B^T s on the matrix cores (one column of four digits), + the reduced mirrors' sums
    {
        const int64_t SK = bp.SKr_used;
        int per = 1;
        const int chunks = bp_col_chunks(bp, SK, &per);
        v4i* qd = reinterpret_cast<v4i*>(bp.qd);
        double* cmaxS = bp.cmax + 128;
        {
            ScopedTimer t(ctx, "bitplane_prep");
            k_bp_colmax<<<(unsigned)std::min<int64_t>(1024, ceil_div(M, 64)), 256, 0, ctx->stream>>>(bp.srow, nullptr, M, 1, cmaxS);     // (into zeros: bp_workspace)
            const int nslot = 8;
            const int64_t nthreads = std::max<int64_t>(64, SK * kBpSteps * 2 * nslot);
            k_bp_digits<4, true><<<(unsigned)ceil_div(nthreads, 256), 256, 0, ctx->stream>>>(bp.srow, nullptr, M, 1, 1, nslot, cmaxS, SK, bp.Npad, N, qd, nullptr);
        }
        BpProductArgs a{};
        a.bm = reinterpret_cast<const v4i*>(bp.bm_cols); a.qd = qd; a.ntile = bp.ntile_c; a.SK = (int)SK; a.SKstride = bp.SKr; a.sk_per_chunk = per; a.L = 1;
        a.cmax = cmaxS; a.nOut = ctx->H; a.out = bp.part;
        {
            ScopedTimer t(ctx, "bitplane_cols");
            DDX_TRY((bp_launch_t<2, 1, 4, false>(ctx, a, chunks)));
        }
        DDX_TRY(bp_colmean(ctx, bp.part, chunks));
    }
    DDX_HIP(ctx, hipGetLastError());
    bp.values = true;
    return DDX_OK;
}

int bp_refresh(ddx_ctx* ctx) {
    int rc = bp_refresh_once(ctx);
    if (rc == kBpRetry) {
        // (the old structures are abandoned, not released: a follower context may be copying them at this moment; the arena takes the block
        // back when the context is reset for its next fit)
        ctx->bp_buf = DevBuf();
        DDX_TRY(bp_build(ctx));
        ctx->rowseg_rows = -1;
        rc = bp_refresh_once(ctx);
        if (rc == kBpRetry) return set_err(ctx, DDX_E_NOMEM, "bit planes: the synthetic rows' reduced entries do not fit the rebuilt structures");
    }
    return rc;
}

// chunks of the k dimension of the A^T Y product (the padded rows): enough workgroups for a whole round of the GPU
static int bp_col_chunks(const BitPlanes& bp, int64_t SK, int* per_out, int tiles_per_wg) {
    const int64_t wgcols = ceil_div(bp.ntile_c, tiles_per_wg);
    int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(SK, 256 / std::max<int64_t>(1, wgcols)));
    const int per = (int)ceil_div(SK, chunks);
    if (per_out) *per_out = per;
    return (int)ceil_div(SK, per);
}

// work space of the products, sized once per iteration (bp_refresh) for both of them: digits | column maxima (Q side, Y side) | partial blocks
static int bp_workspace(ddx_ctx* ctx) {
    BitPlanes& bp = ctx->bp;
    const int64_t SKmax = std::max<int64_t>(bp.SKc, bp.SKr_cap);
    const size_t dig = bp_align(std::max(sizeof(v4i) * (size_t)SKmax * kBpSteps * 5 * 64, (size_t)SKmax * kF6StageBytes));
    // (the chunk count of a product follows from the stages IN USE, ceil(SK / ceil(SK / c)) <= c with c the bound below -- which the count
    // for the capacity does not bound: 8192 cells x 6000 genes cut 40 used stages into 20 chunks, 49 stages of capacity into 17)
    const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(bp.SKr_cap, 256 / std::max<int64_t>(1, ceil_div(bp.ntile_c, kBpWaves * 2))));
    const size_t prt = bp_align(sizeof(double) * (size_t)chunks * ctx->H * 40);
    const size_t rop = bp_align(sizeof(double) * 3 * (size_t)bp.cap_srow);
    const size_t need = dig + bp_align(sizeof(double) * 192) + prt + rop;
    DDX_TRY(ensure(ctx, ctx->bp_work, need));
    char* b = ctx->bp_work.as<char>();
    bp.qd = b;
    bp.cmax = reinterpret_cast<double*>(b + dig);                 // [0..63]: Q side, [64..127]: Y side, [128..]: the row scales (column means, scale statistics)
    bp.part = reinterpret_cast<double*>(b + dig + bp_align(sizeof(double) * 192));
    bp.rowop = reinterpret_cast<double*>(b + dig + bp_align(sizeof(double) * 192) + prt);
    DDX_HIP(ctx, hipMemsetAsync(bp.cmax, 0, sizeof(double) * 192, ctx->stream));
    bp.ymax_of = nullptr;
    bp.qmax_of = nullptr;
    bp.qmax_zeroed = true;
    return DDX_OK;
}

// option bp_dbg_sk (timing only): every chunk stops after that many stages
static BpProductArgs bp_dbg_args(const ddx_ctx* ctx, const BpProductArgs& a, int chunks) {
    BpProductArgs b = a;
    if (ctx->opt.bp_dbg_sk > 0 && ctx->opt.bp_dbg_sk < a.sk_per_chunk) { b.sk_per_chunk = ctx->opt.bp_dbg_sk; b.SK = std::min(a.SK, chunks * b.sk_per_chunk); }
    b.dbg = ctx->opt.bp_dbg_mode;
    return b;
}

template <int RT, int NT, int ND, bool ROWS>
static int bp_launch_t(ddx_ctx* ctx, const BpProductArgs& a, int chunks) {
    const size_t lds = (size_t)kBpStages * kBpSteps * NT * 64 * 16;
    DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_bp_product<RT, NT, ND, ROWS>), (int)lds));
    const dim3 grid((unsigned)ceil_div(a.ntile, kBpWaves * RT), (unsigned)chunks);
    k_bp_product<RT, NT, ND, ROWS><<<grid, 64 * kBpWaves, lds, ctx->stream>>>(a);
    return DDX_OK;
}

template <bool ROWS>
static int bp_launch(ddx_ctx* ctx, const BpProductArgs& a0, int chunks, int ND, int RT) {
    const BpProductArgs a = bp_dbg_args(ctx, a0, chunks);
    if (ND == 2) return RT == 1 ? bp_launch_t<1, 3, 2, ROWS>(ctx, a, chunks) : bp_launch_t<2, 3, 2, ROWS>(ctx, a, chunks);
    if (ND == 3) return RT == 1 ? bp_launch_t<1, 4, 3, ROWS>(ctx, a, chunks) : bp_launch_t<2, 4, 3, ROWS>(ctx, a, chunks);
    return RT == 1 ? bp_launch_t<1, 5, 4, ROWS>(ctx, a, chunks) : bp_launch_t<2, 5, 4, ROWS>(ctx, a, chunks);
}

template <bool ROWS>
static int bp_launch6(ddx_ctx* ctx, const BpProductArgs& a0, int chunks) {
    const BpProductArgs a = bp_dbg_args(ctx, a0, chunks);
    const size_t lds = (size_t)kF6Lds;
    const dim3 grid((unsigned)ceil_div(a.ntile, kBpWaves), (unsigned)chunks);      // (2 row groups x 4 tiles per workgroup)
    auto go = [&](auto tag) -> int {
        constexpr int DBG = decltype(tag)::value;
        DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_bp_product6<ROWS, DBG>), (int)lds));
        k_bp_product6<ROWS, DBG><<<grid, 64 * kBpWaves, lds, ctx->stream>>>(a);
        return DDX_OK;
    };
#ifdef DDX_ABLATION
    switch (a.dbg) {                                              // (timing experiments, wrong results: profiles/tools/mx_ablation.py)
        case 1: return go(std::integral_constant<int, 1>());
        case 2: return go(std::integral_constant<int, 2>());
        case 3: return go(std::integral_constant<int, 3>());
        case 4: return go(std::integral_constant<int, 4>());
        case 7: return go(std::integral_constant<int, 7>());
        case 8: return go(std::integral_constant<int, 8>());
        default: break;
    }
#endif
    return go(std::integral_constant<int, 0>());
}

// the products of sketches up to 40 columns wide run on the MX instruction unless option bp_format says int8
static bool bp_use_mx(const ddx_ctx* ctx, int L) { return ctx->opt.bp_mx && L <= 4 * kF6Quarter; }

template <bool ROWMAP>
static int bp_launch_digits6(ddx_ctx* ctx, const double* X, const double* wgt, int64_t R, int L, const double* cmax, int64_t SK, int64_t Npad, int64_t N, void* qd, double* zero_me) {
    DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_bp_digits6<ROWMAP>), kF6StageBytes));
    k_bp_digits6<ROWMAP><<<(unsigned)SK, 320, kF6StageBytes, ctx->stream>>>(X, wgt, R, L, cmax, Npad, N, reinterpret_cast<unsigned char*>(qd), zero_me);
    return DDX_OK;
}

// tiles per wave: the geometry that fills the GPU's 256 compute units better over whole rounds of workgroups
static int bp_pick_rt(int64_t ntile, int chunks) {
    double best = -1.0;
    int pick = 2;
    for (int rt = 2; rt >= 1; --rt) {
        const int64_t wgs = ceil_div(ntile, kBpWaves * rt) * chunks;
        const double rounds = (double)ceil_div(wgs, 256);
        const double eff = (double)ntile * chunks / (rounds * 256.0 * kBpWaves * rt) * (rt == 2 ? 1.0 : 0.93);     // (one tile per wave re-reads every operand fragment: slower per tile)
        if (eff > best) { best = eff; pick = rt; }
    }
    return pick;
}

// digits of the product being issued: the option, or what stage_pca asked for its early power iterations (bp.nd_now)
static int bp_digits_now(const ddx_ctx* ctx) {
    const int nd = ctx->bp.nd_now ? ctx->bp.nd_now : ctx->opt.bp_digits;
    return nd <= 2 ? 2 : nd == 3 ? 3 : 4;
}

template <bool ROWMAP, typename... Args>
static void bp_launch_digits(int ND, unsigned grid, hipStream_t st, Args... args) {
    if (ND == 2) k_bp_digits<2, ROWMAP><<<grid, 256, 0, st>>>(args...);
    else if (ND == 3) k_bp_digits<3, ROWMAP><<<grid, 256, 0, st>>>(args...);
    else k_bp_digits<4, ROWMAP><<<grid, 256, 0, st>>>(args...);
}

// Y[i][:] = s_i (B Q)[i][:] for every row i of the augmented matrix (plain stores: the sparse product that follows adds its part
// and, through *ymax, collects the column maxima of diag(s) Y that the A^T Y product will cut its digits by)
int bp_rows_product(ddx_ctx* ctx, const double* Q, int L, double* Y) {
    BitPlanes& bp = ctx->bp;
    const bool mx = bp_use_mx(ctx, L);
    const int ND = bp_digits_now(ctx);
    const int NT = (40 * ND + 31) / 32;                           // 32-wide tiles of the flattened (column, digit) index: 3 / 4 / 5
    v4i* qd = reinterpret_cast<v4i*>(bp.qd);
    double* cmaxQ = bp.cmax;
    double* cmaxY = bp.cmax + 64;
    {
        ScopedTimer t(ctx, "bitplane_prep");
        // scaled matrix: the operand is diag(1 / sd) Q -- the digits are cut from the weighted rows, the bitmaps and the row scales stay
        const double* wq = bp.scaled ? bp.inv_sd : nullptr;
        if (bp.qmax_of != Q) {                                    // (else: left there by the kernel that produced Q, k_right_mult)
            if (!bp.qmax_zeroed) DDX_HIP(ctx, hipMemsetAsync(cmaxQ, 0, sizeof(double) * 64, ctx->stream));
            k_bp_colmax<<<(unsigned)std::min<int64_t>(256, ceil_div(ctx->H, 64)), 256, 0, ctx->stream>>>(Q, wq, ctx->H, L, cmaxQ);      // (into zeros)
        }
        bp.qmax_of = nullptr;
        bp.qmax_zeroed = false;
        const int nslot = (NT * 32 + ND - 1) / ND;
        const int64_t nthreads = std::max<int64_t>(64, (int64_t)bp.SKc * kBpSteps * 2 * nslot);
        if (mx) {
            DDX_TRY(bp_launch_digits6<false>(ctx, Q, wq, ctx->H, L, cmaxQ, bp.SKc, 0, 0, bp.qd, cmaxY));
        } else
        bp_launch_digits<false>(ND, (unsigned)ceil_div(nthreads, 256), ctx->stream, Q, wq, (int64_t)ctx->H, L, NT, nslot, (const double*)cmaxQ, (int64_t)bp.SKc, (int64_t)0, (int64_t)0, qd, cmaxY);
    }
    bp.ymax_of = nullptr;
    BpProductArgs a{};
    a.bm = reinterpret_cast<const v4i*>(bp.bm_rows); a.qd = qd; a.ntile = bp.ntile_o + bp.ntile_s; a.SK = bp.SKc; a.SKstride = bp.SKc; a.sk_per_chunk = bp.SKc; a.L = L;
    a.cmax = cmaxQ;
    a.srow = bp.srow; a.Npad = bp.Npad; a.N = ctx->N; a.M = ctx->M; a.nOut = ctx->M; a.out = Y;
    ScopedTimer t(ctx, "bitplane_rows");
    if (mx) return bp_launch6<true>(ctx, a, 1);
    return bp_launch<true>(ctx, a, 1, ND, bp_pick_rt(a.ntile, 1));
}

// partial blocks of W1[j][:] = sum over the rows i of B[i][j] s_i Y[i][:]: *chunks blocks [H x L] float64 at *part, to be added by k_sum_panels
int bp_cols_product(ddx_ctx* ctx, const double* Y, int L, const double** part, int* chunks_out) {
    BitPlanes& bp = ctx->bp;
    const bool mx = bp_use_mx(ctx, L);
    const int ND = bp_digits_now(ctx);
    const int NT = (40 * ND + 31) / 32;                           // 32-wide tiles of the flattened (column, digit) index: 3 / 4 / 5
    const int64_t SK = bp.SKr_used;
    int per = 1;
    const int chunks = bp_col_chunks(bp, SK, &per, mx ? kBpWaves : kBpWaves * 2);
    v4i* qd = reinterpret_cast<v4i*>(bp.qd);
    double* cmaxQ = bp.cmax;
    double* cmaxY = bp.cmax + 64;
    {
        ScopedTimer t(ctx, "bitplane_prep");
        if (bp.ymax_of != Y) {                                    // Y was not written by the sparse A Q kernel of this context: its maxima by a pass of their own
            DDX_HIP(ctx, hipMemsetAsync(cmaxY, 0, sizeof(double) * 64, ctx->stream));
            k_bp_colmax<<<(unsigned)std::min<int64_t>(1024, ceil_div(ctx->M, 64)), 256, 0, ctx->stream>>>(Y, bp.srow, ctx->M, L, cmaxY);
        }
        const int nslot = (NT * 32 + ND - 1) / ND;
        const int64_t nthreads = std::max<int64_t>(64, SK * kBpSteps * 2 * nslot);
        if (mx) {
            DDX_TRY(bp_launch_digits6<true>(ctx, Y, bp.srow, ctx->M, L, cmaxY, SK, bp.Npad, ctx->N, bp.qd, cmaxQ));
        } else
        bp_launch_digits<true>(ND, (unsigned)ceil_div(nthreads, 256), ctx->stream, Y, (const double*)bp.srow, (int64_t)ctx->M, L, NT, nslot, (const double*)cmaxY, (int64_t)SK, (int64_t)bp.Npad, (int64_t)ctx->N, qd, cmaxQ);
    }
    bp.ymax_of = nullptr;
    bp.qmax_zeroed = true;                                        // (zero_me of the digit kernel above)
    bp.qmax_of = nullptr;
    BpProductArgs a{};
    a.bm = reinterpret_cast<const v4i*>(bp.bm_cols); a.qd = qd; a.ntile = bp.ntile_c; a.SK = (int)SK; a.SKstride = bp.SKr; a.sk_per_chunk = per; a.L = L;
    a.cmax = cmaxY; a.nOut = ctx->H; a.out = bp.part;
    // (the bitmap's stage stride is the capacity SKr; the stages in use are [0, SKr_used): the chunks cover only those)
    ScopedTimer t(ctx, "bitplane_cols");
    if (mx) DDX_TRY(bp_launch6<false>(ctx, a, chunks));
    else DDX_TRY(bp_launch<false>(ctx, a, chunks, ND, 2));
    *part = bp.part;
    *chunks_out = chunks;
    return DDX_OK;
}

// ---- standard scaling on this route ----------------------------------------------------------------------------------------
// per row the three operands of the scale statistics: s_i = x_i(1) - z (what an entry equal to 1 adds to sum(x - z)), the float32
// square of x_i(1) (what it adds to the sum of squares, oracle: mean of float32 squares) and 1 (the count)
__global__ void k_bp_row_stats(const float* __restrict__ tab, int tab_stride, float z, int64_t M, double* __restrict__ op) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float x1 = tab[i * tab_stride];
    const float sq = x1 * x1;
    op[3 * i] = (double)(x1 - z);
    op[3 * i + 1] = (double)sq;
    op[3 * i + 2] = 1.0;
}

// B^T [s | x(1)^2 | 1]: *chunks partial blocks [H x 3] float64 at *parts (one narrow product on the matrix cores: three columns of four
// digits; the count column is exact, the other two carry 30 bits below the column's largest element)
int bp_scale_sums(ddx_ctx* ctx, const double** parts, int* chunks_out) {
    BitPlanes& bp = ctx->bp;
    const int64_t M = ctx->M, SK = bp.SKr_used;
    int per = 1;
    const int chunks = bp_col_chunks(bp, SK, &per);
    v4i* qd = reinterpret_cast<v4i*>(bp.qd);
    double* cmaxS = bp.cmax + 128;
    {
        ScopedTimer t(ctx, "bitplane_prep");
        DDX_HIP(ctx, hipMemsetAsync(cmaxS, 0, sizeof(double) * 64, ctx->stream));
        k_bp_row_stats<<<(unsigned)ceil_div(M, 256), 256, 0, ctx->stream>>>(ctx->lognorm_tab.as<float>(), kLognormTab, ctx->zvalue, M, bp.rowop);
        k_bp_colmax<<<(unsigned)std::min<int64_t>(1024, ceil_div(M, 64)), 256, 0, ctx->stream>>>(bp.rowop, nullptr, M, 3, cmaxS);
        const int nslot = 8;
        const int64_t nthreads = std::max<int64_t>(64, SK * kBpSteps * 2 * nslot);
        k_bp_digits<4, true><<<(unsigned)ceil_div(nthreads, 256), 256, 0, ctx->stream>>>(bp.rowop, nullptr, M, 3, 1, nslot, cmaxS, SK, bp.Npad, ctx->N, qd, nullptr);
    }
    bp.ymax_of = nullptr;
    BpProductArgs a{};
    a.bm = reinterpret_cast<const v4i*>(bp.bm_cols); a.qd = qd; a.ntile = bp.ntile_c; a.SK = (int)SK; a.SKstride = bp.SKr; a.sk_per_chunk = per; a.L = 3;
    a.cmax = cmaxS; a.nOut = ctx->H; a.out = bp.part;
    {
        ScopedTimer t(ctx, "bitplane_cols");
        DDX_TRY((bp_launch_t<2, 1, 4, false>(ctx, a, chunks)));
    }
    *parts = bp.part;
    *chunks_out = chunks;
    return DDX_OK;
}

// the per-fit structures again with the columns `want` demoted, and this iteration's (unscaled) values on them; followers that clone from
// now on get the new structures (the old buffer is abandoned, not released: a follower may be copying it)
int bp_rebuild_demoted(ddx_ctx* ctx, const std::vector<uint8_t>& want) {
    BitPlanes& bp = ctx->bp;
    int64_t n = 0;
    for (uint8_t f : want) n += f ? 1 : 0;
    std::vector<uint8_t> keep = want;
    const int64_t want_rest_s = bp.want_rest_s;
    bp.ready = false;
    bp.values = false;
    ctx->bp_buf = DevBuf();
    bp.demote = std::move(keep);
    bp.n_demoted = n;
    bp.demote_decided = true;
    bp.want_rest_s = want_rest_s;
    DDX_TRY(bp_build(ctx));
    ctx->rowseg_rows = -1;
    DDX_TRY(bp_refresh(ctx));
    publish_clone_view(ctx);
    return DDX_OK;
}

}  // namespace ddx
