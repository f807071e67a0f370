// Sparse-matrix stages of the boosting iteration on gfx950:
//   - resident counts + memoised library sizes           (dd.py:178-184)
//   - synthetic doublets: CSR two-row gather + sorted merge (dd.py:385-402)
//   - log-normalisation of the augmented matrix, kept sparse (dd.py:286-298)
//   - optional standard scaling                            (dd.py:302-303)
// All kernels are HBM/L2-bound streaming or gather kernels: one wavefront (64 lanes) owns one matrix
// row so that loads of a row are contiguous 256-byte segments; nothing here is GEMM-shaped.
#include "ddx_prims.h"

#include <algorithm>
#include <cmath>

#include "ddx_internal.h"

namespace ddx {

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lower_bound_i32(const int32_t* __restrict__ a, int n, int32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int upper_bound_i32(const int32_t* __restrict__ a, int n, int32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] <= key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// row id of CSR position `pos`: largest r with indptr[r] <= pos (rows may be empty)
__device__ __forceinline__ int64_t row_of_pos(const int64_t* __restrict__ indptr, int64_t nrows, int64_t pos) {
    int64_t lo = 0, hi = nrows;  // invariant: indptr[lo] <= pos < indptr[hi]
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (indptr[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

// Sequential (scipy csr_matvec order; exact and hence order-free for count data -- a scipy that reduces pairwise
// can differ by an ulp or two on fractional input) float32 row sum and sklearn's double L1 norm.  One wave per
// row: the wave loads 64 entries at a time (coalesced), then every lane replays the same left-to-right
// additions through readlane broadcasts, so the result is the scalar loop's, bit for bit.
__global__ void __launch_bounds__(256) k_row_sums(const int64_t* __restrict__ indptr, const float* __restrict__ val,
                                                  int64_t row0, int64_t nrows, float* __restrict__ lib32,
                                                  double* __restrict__ lib64) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int64_t row = row0 + r;
    const int64_t b = indptr[row], e = indptr[row + 1];
    float s32 = 0.f;
    double s64 = 0.0;
    for (int64_t base = b; base < e; base += 64) {
        const int64_t p = base + lane;
        const float v = (p < e) ? val[p] : 0.f;
        const int cnt = (int)((e - base) < 64 ? (e - base) : 64);
        for (int t = 0; t < cnt; ++t) {
            const float vt = __shfl(v, t, 64);
            s32 = __fadd_rn(s32, vt);
            s64 = __dadd_rn(s64, fabs((double)vt));
        }
    }
    if (lane == 0) {
        lib32[row] = s32;
        lib64[row] = s64;
    }
}

// Counts are normally non-negative integers with library sizes far below 2^22: then every partial sum of a row -- of an
// original cell or of a doublet (the sum of two of them) -- is an integer below 2^24, exact in float32 whatever the
// order, and the sequential replay above can be replaced by a lane-strided sum.  flag[0] stays 1 iff that holds.
// The library sizes of the originals are computed by the lane-strided kernel first and checked here; only when the
// check fails does the sequential replay run (1.0 ms per fit at the headline workload against 0.15 ms).
__global__ void k_counts_exact(const float* __restrict__ val, int64_t nnz, const float* __restrict__ lib32, int64_t nrows, int* __restrict__ flag) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = true;
    if (t < nnz) { const float v = val[t]; ok = v >= 0.f && v == truncf(v); }
    if (t < nrows) ok = ok && lib32[t] < 4194304.f;      // (2^22: a lane-strided sum of non-negative integers that comes out below it was exact)
    if (!ok) flag[0] = 0;
}

__global__ void __launch_bounds__(256) k_row_sums_exact(const int64_t* __restrict__ indptr, const float* __restrict__ val,
                                                        int64_t row0, int64_t nrows, float* __restrict__ lib32,
                                                        double* __restrict__ lib64) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int64_t row = row0 + r;
    float s = 0.f;
    const int64_t e = indptr[row + 1];
    for (int64_t p = indptr[row] + lane; p < e; p += 256) {         // four loads in flight (any order is exact here)
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p + 64 * u < e ? val[p + 64 * u] : 0.f;
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        lib32[row] = s;
        lib64[row] = (double)s;
    }
}

// ------------------------------------------------------------------------------------------------
// doublets
// ------------------------------------------------------------------------------------------------
// Classification used by the fill kernel.  A = row p0, B = row p1, both sorted.
//   A element i : merged position i + #{b < a_i};   matched b adds its value;
//   B element j : merged position j + #{a <= b_j};  dropped when matched (already emitted by A).
// An entry is kept iff its float32 sum is != 0 (scipy csr_plus_csr drops exact zeros).

// single-block exclusive scan of int32 counts into int64 row pointers: out[i] = base + sum_{t<i} in[t],
// for i in [0, n]; n <= a few hundred thousand rows, one launch of 1024 threads.
__global__ void __launch_bounds__(1024) k_scan_counts(const int32_t* __restrict__ in, int64_t n, int64_t base,
                                                      int64_t* __restrict__ out) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry = base;
    __syncthreads();
    for (int64_t start = 0; start < n; start += 1024) {
        const int64_t i = start + tid;
        int64_t v = (i < n) ? (int64_t)in[i] : 0;
        int64_t x = v;  // inclusive scan inside the wave
        for (int off = 1; off < 64; off <<= 1) {
            int64_t y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int64_t woff = 0;
        for (int t = 0; t < w; ++t) woff += wsum[t];
        const int64_t c = carry;
        if (i < n) out[i] = c + woff + x - v;
        __syncthreads();
        if (tid == 1023) carry = c + woff + x;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}

int scan_counts(ddx_ctx* ctx, const int32_t* in, int64_t n, int64_t base, int64_t* out) {
    k_scan_counts<<<1, 1024, 0, ctx->stream>>>(in, n, base, out);
    return DDX_OK;
}

constexpr int kMergeTile = 2048;
constexpr int kMergeStage = 3840;     // column entries of both parents staged in LDS (15 KB: four workgroups per CU)

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_doublet_fill(const int64_t* __restrict__ indptr,
                                                      const int32_t* __restrict__ indices,
                                                      const float* __restrict__ val,
                                                      const int64_t* __restrict__ parents,
                                                      const int64_t* __restrict__ out_off /* [S]: start of row s in the outputs */,
                                                      int32_t* __restrict__ out_indices, float* __restrict__ out_val,
                                                      int32_t* __restrict__ counts /* [S]: entries written, or null */) {
    __shared__ int32_t wsum[4];
    __shared__ int32_t run_base;
    __shared__ int32_t s_idx[kMergeStage];          // both parents' column lists when they fit: the searches then run in LDS
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t s = blockIdx.x;
    const int64_t p0 = parents[2 * s], p1 = parents[2 * s + 1];
    const int64_t a0 = indptr[p0], b0 = indptr[p1];
    const int la = (int)(indptr[p0 + 1] - a0), lb = (int)(indptr[p1 + 1] - b0);
    const int32_t* A = indices + a0;
    const int32_t* B = indices + b0;
    const int L = la + lb;
    if (L <= kMergeStage) {                         // block-uniform
        // (all loads of a thread in flight before the first LDS store: parents hold ~900 entries, 4 per thread)
        for (int i0 = 0; i0 < L; i0 += 4 * 256) {
            int32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 + tid;
                v[u] = i < la ? A[i] : (i < L ? B[i - la] : 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 + tid;
                if (i < L) s_idx[i] = v[u];
            }
        }
        A = s_idx;
        B = s_idx + la;
        __syncthreads();
    }
    const int64_t o0 = out_off[s];
    if (tid == 0) run_base = 0;
    // Merge path (round 5; rounds 1-4 let every element find its merged position by a binary search in the other row: ~12 LDS reads
    // per element).  The merged sequence -- elements of A and B by (column, A before B) -- is cut into runs of PER positions, one per
    // thread: the thread finds where its run starts on the merge path (ONE binary search over the diagonal i + j = d) and then
    // merges PER steps sequentially.  An element of B that follows its equal in A is the "matched" one: A's element carries the sum
    // (float32, __fadd_rn: the reference's scipy addition), B's slot is dropped; exact zeros are dropped as before.
    constexpr int PER = kMergeTile / 256;
    for (int t0 = 0; t0 < L; t0 += kMergeTile) {
        const int d0 = t0 + tid * PER;                  // first merged position of this thread
        int i = 0, j = 0;
        if (d0 < L) {
            // smallest i in [max(0, d0 - lb), min(d0, la)] with A[i] > B[d0 - i - 1] (i.e. B[d0 - i - 1] already precedes A[i]);
            // A goes first among equals, so "A[i] > B[..]" is strict
            int lo = d0 - lb > 0 ? d0 - lb : 0, hi = d0 < la ? d0 : la;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (A[mid] <= B[d0 - mid - 1]) lo = mid + 1; else hi = mid;
            }
            i = lo; j = d0 - lo;
        }
        int32_t ocol[PER];
        float oval[PER];
        int keep[PER];
        int tot = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            ocol[q] = 0; oval[q] = 0.f; keep[q] = 0;
            if (d0 + q < L) {
                const bool takeA = i < la && (j >= lb || A[i] <= B[j]);
                if (takeA) {
                    const int32_t col = A[i];
                    float v = val[a0 + i];
                    if (j < lb && B[j] == col) v = __fadd_rn(v, val[b0 + j]);      // its equal in B follows in the next slot
                    ocol[q] = col; oval[q] = v; keep[q] = (v != 0.f);
                    ++i;
                } else {
                    const int32_t col = B[j];
                    const bool matched = i > 0 && A[i - 1] == col;
                    const float v = val[b0 + j];
                    ocol[q] = col; oval[q] = v; keep[q] = (!matched && v != 0.f);
                    ++j;
                }
            }
            tot += keep[q];
        }
        int x = tot;
        for (int off = 1; off < 64; off <<= 1) {
            int y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int woff = 0;
        for (int t = 0; t < w; ++t) woff += wsum[t];
        int base = run_base + woff + x - tot;
        const int first = base;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (keep[q]) {
                out_indices[o0 + base] = ocol[q];
                out_val[o0 + base] = oval[q];
                ++base;
            }
        }
        __syncthreads();
        if (tid == 255) run_base = first + tot;
        __syncthreads();
    }
    if (counts && tid == 0) counts[s] = run_base;
}

// rows written at padded offsets (|parent 0| + |parent 1| slots each) moved to their final CSR positions
__global__ void __launch_bounds__(256) k_doublet_compact(const int64_t* __restrict__ pad_off, const int32_t* __restrict__ pad_idx,
                                                         const float* __restrict__ pad_val, const int64_t* __restrict__ indptr_s /* aug_indptr + N */,
                                                         int64_t S, int32_t* __restrict__ out_indices, float* __restrict__ out_val) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int64_t src = pad_off[s], dst = indptr_s[s];
    const int n = (int)(indptr_s[s + 1] - dst);
    for (int i0 = lane; i0 < n; i0 += 256) {              // four elements of both arrays in flight per lane
        int32_t ci[4];
        float cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u;
            ci[u] = i < n ? pad_idx[src + i] : 0;
            cv[u] = i < n ? pad_val[src + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u;
            if (i < n) { out_indices[dst + i] = ci[u]; out_val[dst + i] = cv[u]; }
        }
    }
}

// sort keys for the output rankings: stored entries per row / per column (over all panels of both mirrors)
__global__ void k_row_lengths(const int64_t* __restrict__ indptr, int64_t M, uint32_t* __restrict__ keys, int32_t* __restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    keys[i] = (uint32_t)(indptr[i + 1] - indptr[i]);
    ids[i] = (int32_t)i;
}
__global__ void k_col_lengths(const int64_t* __restrict__ cp_o, int P_o, const int64_t* __restrict__ cp_s, int P_s, int32_t H,
                              uint32_t* __restrict__ keys, int32_t* __restrict__ ids) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= H) return;
    int64_t n = 0;
    for (int p = 0; p < P_o; ++p) n += cp_o[(int64_t)p * H + j + 1] - cp_o[(int64_t)p * H + j];
    for (int p = 0; p < P_s; ++p) n += cp_s[(int64_t)p * H + j + 1] - cp_s[(int64_t)p * H + j];
    keys[j] = (uint32_t)n;
    ids[j] = j;
}

// Rankings used by the LDS-staged operator products: rows and columns sorted by their number of stored entries
// (descending, ties by index -- the radix sort is stable).  Rebuilt per iteration: the synthetic rows change.
int stage_rankings(ddx_ctx* ctx, const int64_t* indptr, const int64_t* cp_o, const int64_t* cp_s) {
    const int64_t M = ctx->M;
    const int32_t H = ctx->H;
    const int64_t n = M + H;
    DDX_TRY(ensure(ctx, ctx->rank_buf, (sizeof(uint32_t) * 2 + sizeof(int32_t) * 2) * (size_t)n));
    uint32_t* keys_in = ctx->rank_buf.as<uint32_t>();
    uint32_t* keys_out = keys_in + n;
    int32_t* ids_in = reinterpret_cast<int32_t*>(keys_out + n);
    int32_t* ids_out = ids_in + n;
    k_row_lengths<<<(unsigned)ceil_div(M, 256), 256, 0, ctx->stream>>>(indptr, M, keys_in, ids_in);
    k_col_lengths<<<(unsigned)ceil_div(H, 256), 256, 0, ctx->stream>>>(cp_o, ctx->P_o, cp_s, ctx->P_s, H, keys_in + M, ids_in + M);
    size_t tmp_r = 0, tmp_c = 0;
    int bits_r = 1, bits_c = 1;                 // a row holds at most H entries, a column at most M: sort only those bits
    while (((int64_t)1 << bits_r) <= H) ++bits_r;
    while (((int64_t)1 << bits_c) <= M) ++bits_c;
    DDX_HIP(ctx, prim::sort_pairs_desc(nullptr, tmp_r, keys_in, keys_out, ids_in, ids_out, (int)M, 0, bits_r, ctx->stream));
    DDX_HIP(ctx, prim::sort_pairs_desc(nullptr, tmp_c, keys_in + M, keys_out + M, ids_in + M, ids_out + M, (int)H, 0, bits_c, ctx->stream));
    DDX_TRY(ensure(ctx, ctx->sort_tmp, std::max(tmp_r, tmp_c)));
    DDX_HIP(ctx, prim::sort_pairs_desc(ctx->sort_tmp.p, tmp_r, keys_in, keys_out, ids_in, ids_out, (int)M, 0, bits_r, ctx->stream));
    DDX_HIP(ctx, prim::sort_pairs_desc(ctx->sort_tmp.p, tmp_c, keys_in + M, keys_out + M, ids_in + M, ids_out + M, (int)H, 0, bits_c, ctx->stream));
    ctx->rank_rows = ids_out;
    ctx->rank_cols = ids_out + M;
    return DDX_OK;
}

// ------------------------------------------------------------------------------------------------
// column-major mirror
// ------------------------------------------------------------------------------------------------
__global__ void k_iota_u32(uint32_t* out, int64_t n, uint32_t base) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = base + (uint32_t)i;
}

// sort key of every stored entry of rows [row_lo, row_lo+nrows): (panel(row) - panel0) * H + column
__global__ void __launch_bounds__(256) k_panel_keys(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                    int64_t row_lo, int64_t nrows, int32_t H, int32_t panel0, int32_t panel_rows,
                                                    int64_t e0, int32_t* __restrict__ keys, int32_t* __restrict__ rowid) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int64_t row = row_lo + r;
    const int32_t base = ((int32_t)(row / panel_rows) - panel0) * H;
    for (int64_t p = indptr[row] + lane; p < indptr[row + 1]; p += 64) {
        keys[p - e0] = base + cols[p];
        rowid[p - e0] = (int32_t)row;                // looked up again after the sort (cheaper than a search in indptr)
    }
}

// colptr[j] = first sorted position whose key >= j, j in [0, nkeys]
__global__ void k_colptr_from_sorted(const int32_t* __restrict__ keys, int64_t n, int32_t nkeys, int64_t* __restrict__ colptr) {
    int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nkeys) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < j) lo = mid + 1; else hi = mid;
    }
    colptr[j] = lo;
}

// gather row id and raw value of every mirror entry; pos = CSR position, rowid indexed by pos - e0
__global__ void k_csc_gather(const uint32_t* __restrict__ pos, int64_t n, int64_t e0, const int32_t* __restrict__ rowid,
                             const float* __restrict__ raw, int32_t* __restrict__ row_out, float* __restrict__ raw_out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t p = pos[t];
    row_out[t] = rowid[p - e0];
    raw_out[t] = raw[p];
}


// ------------------------------------------------------------------------------------------------
// column-major mirror by counting sort (the default): the (panel, column) order with rows ascending inside a segment is
// what a stable sort by key = panel * H + column produces, but the keys have structure a radix sort cannot use -- the
// rows of a panel are contiguous in the CSR and a row holds every column at most once.  A panel is cut into 16
// sub-blocks of consecutive rows (49 rows for the 784-row panels); one workgroup per sub-block
//   1. counts the sub-block's entries per column in an LDS histogram (integer atomics: exact in any order),
//   2. (after per-panel prefixes over the 16 sub-blocks and one device-wide scan of the (panel, column) totals)
//   3. scatters its rows ONE AFTER THE OTHER: the entries of a row are written by all threads at once -- no two of them
//      share a column -- each to the running position of its column, which then advances; a barrier separates rows.
// So inside a (panel, column) segment the entries appear by sub-block and, inside a sub-block, by row: ascending rows,
// bit-identical to the stable sort (tests/test_gpu_parity.py compares both paths), at a third of its cost: the
// entries are read twice and written once instead of three radix passes over (key, position) pairs plus a gather.
// ------------------------------------------------------------------------------------------------
constexpr int kMirrorSub = 16;          // sub-blocks per panel
constexpr int kMirrorThreads = 1024;
constexpr int kMirrorMaxH = 36 * 1024;  // LDS histogram of 32-bit counters: 144 KB of the 160 KB

__global__ void __launch_bounds__(kMirrorThreads) k_mirror_count(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                                 int64_t row_lo, int64_t row_hi, int32_t sub_rows, int64_t sb0, int32_t H,
                                                                 uint16_t* __restrict__ cnt, int32_t W, int32_t wpp, uint32_t* __restrict__ bnd) {
    extern __shared__ uint32_t hist[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int j = tid; j < H; j += kMirrorThreads) hist[j] = 0;
    __syncthreads();
    const int64_t sb = sb0 + blockIdx.x;
    int64_t r0 = sb * sub_rows, r1 = r0 + sub_rows;
    if (r0 < row_lo) r0 = row_lo;
    if (r1 > row_hi) r1 = row_hi;
    for (int64_t r = r0 + wave; r < r1; r += kMirrorThreads / 64) {
        const int64_t b = indptr[r], e = indptr[r + 1];
        if (!bnd) {
            for (int64_t p = b + lane; p < e; p += 64) atomicAdd(&hist[cols[p]], 1u);
            continue;
        }
        // with the tile placement: bnd[row][w] = position of the row's first entry whose column is >= w * W (w = 0 .. wpp)
        uint32_t* mine = bnd + (size_t)(r - row_lo) * (size_t)(wpp + 1);
        int carry = -1;                                  // window of the previous entry
        for (int64_t p0 = b; p0 < e; p0 += 64) {         // (positions fit 31 bits: checked at upload)
            const int64_t p = p0 + lane;
            int wnd = wpp;                               // lanes behind the row end close the remaining windows
            if (p < e) {
                const int32_t c = cols[p];
                atomicAdd(&hist[c], 1u);
                wnd = c / W;
            }
            int prev = __shfl_up(wnd, 1, 64);
            if (lane == 0) prev = carry;
            if (p <= e)                                  // (p == e: the first lane behind the row)
                for (int w = prev + 1; w <= wnd; ++w) mine[w] = (uint32_t)p;
            carry = __shfl(wnd, 63, 64);
        }
        if (carry < wpp && lane == 0)                    // empty row, or a row that ends with a full chunk
            for (int w = carry + 1; w <= wpp; ++w) mine[w] = (uint32_t)e;
    }
    __syncthreads();
    uint16_t* out = cnt + (size_t)blockIdx.x * H;
    for (int j = tid; j < H; j += kMirrorThreads) out[j] = (uint16_t)hist[j];
}

// per (panel, column): total over the panel's sub-blocks -> tot; the counts become exclusive prefixes inside the panel
__global__ void __launch_bounds__(256) k_mirror_prefix(uint16_t* __restrict__ cnt, int64_t sb0, int64_t n_sb, int32_t panel0, int32_t npanels,
                                                       int32_t H, uint32_t* __restrict__ tot) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)npanels * H) return;
    const int32_t p = (int32_t)(t / H);
    const int32_t j = (int32_t)(t - (int64_t)p * H);
    uint32_t run = 0;
    for (int b = 0; b < kMirrorSub; ++b) {
        const int64_t sb = ((int64_t)panel0 + p) * kMirrorSub + b - sb0;
        if (sb < 0 || sb >= n_sb) continue;
        uint16_t* c = cnt + (size_t)sb * H + j;
        const uint32_t v = *c;
        *c = (uint16_t)run;
        run += v;
    }
    tot[t] = run;
}

__global__ void __launch_bounds__(kMirrorThreads) k_mirror_scatter(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                                   const float* __restrict__ raw, int64_t row_lo, int64_t row_hi,
                                                                   int32_t sub_rows, int64_t sb0, int32_t panel0, int32_t H,
                                                                   const uint16_t* __restrict__ pre, const int64_t* __restrict__ colptr,
                                                                   int32_t* __restrict__ row_out, float* __restrict__ raw_out) {
    extern __shared__ uint32_t off[];
    const int tid = threadIdx.x;
    const int64_t sb = sb0 + blockIdx.x;
    const int32_t p = (int32_t)(sb / kMirrorSub) - panel0;
    const uint16_t* mine = pre + (size_t)blockIdx.x * H;
    const int64_t* cp = colptr + (int64_t)p * H;
    for (int j = tid; j < H; j += kMirrorThreads) off[j] = (uint32_t)cp[j] + mine[j];       // positions fit 31 bits (checked at upload)
    int64_t r0 = sb * sub_rows, r1 = r0 + sub_rows;
    if (r0 < row_lo) r0 = row_lo;
    if (r1 > row_hi) r1 = row_hi;
    __syncthreads();
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t b = indptr[r], e = indptr[r + 1];
        for (int64_t q = b + tid; q < e; q += kMirrorThreads) {
            const int32_t j = cols[q];
            const uint32_t pos = off[j];          // a row holds column j at most once: no other thread touches off[j] now
            off[j] = pos + 1;
            row_out[pos] = (int32_t)r;
            raw_out[pos] = raw[q];
        }
        __syncthreads();
    }
}

__global__ void k_colptr_from_totals(const int64_t* __restrict__ scan, int64_t nkeys, int64_t total, int64_t* __restrict__ colptr) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nkeys) colptr[t] = scan[t];
    else if (t == nkeys) colptr[t] = total;
}

// ------------------------------------------------------------------------------------------------
// Placement by tiles (the default for the LDS panels).  k_mirror_scatter above writes every entry with two scattered
// 4-byte stores; the 16 sub-blocks of a panel each own ~6 entries of a (panel, column) segment, so practically every
// store ends up as a partial-line write in HBM -- 2 x 3e7 of them per build of the synthetic rows, 1.15 ms, which no
// amount of latency hiding improves (measured: one of the two stores removed = half the time).  Here a workgroup owns
// a tile = (panel of 784 rows) x (window of W ~ 45 consecutive columns), whose entries form ONE contiguous range of
// the mirror (~5 000 entries): it assembles that range in LDS and writes it out with full, coalesced lines.
//   * k_mirror_count (which reads every column index anyway) also records, per row and window, where the row's entries
//     of that window start: bnd[row][w] = position of the row's first entry with column >= w * W.  (A search per tile
//     and row -- 16-ary, two dependent rounds of probes -- was tried first: 32 scattered line requests per row and tile
//     kept each CU's L1 busy for 25 of the tile's 50 us.)
//   * 8 lanes look after a row: the row's entries inside the window are [bnd[row][w], bnd[row][w + 1]), fetched with
//     one 8-wide load (rows with more take further turns); the loads of all rows of a wave are in flight together.
//   * position inside the (panel, column) segment = number of earlier rows of the panel that hold the column: every
//     entry sets bit (column, row) of a column-major bitmap in LDS; after a prefix over the 25 words of each column
//     the rank of an entry is a table value plus one popcount.  Exact in any order: the mirror is bit-identical to
//     the stable sort and to the counting scatter.
//   * a 128-byte line of the CSR holds the entries of ~5 neighbouring windows: the tiles are numbered so that the
//     workgroups running on one XCD at a time (block b runs on XCD b % 8) work on consecutive windows of one panel,
//     and the line is fetched from HBM once and then found in that XCD's L2 (L2 hit rate 35 % -> 95 %, fetched bytes
//     3.2 GB -> 0.74 GB per build: profiles/r02n_mirror_notes.txt).
// Tiles larger than the LDS range (rare: windows are sized for 70 % of it) spill their tail entries directly.
// ------------------------------------------------------------------------------------------------
#ifndef DDX_TILE_CAP
#define DDX_TILE_CAP 7168
#define DDX_TILE_MAXW 96
#endif
constexpr int kTileRows = kLdsPanelRows;                  // rows of a panel
constexpr int kTileWpc = (kTileRows + 31) / 32;           // bitmap words per column
constexpr int kTileRpw = (kTileRows + 15) / 16;           // rows per wave
#ifndef DDX_TILE_LPR
#define DDX_TILE_LPR 8
#endif
#ifndef DDX_TILE_FILL
#define DDX_TILE_FILL 0.7
#endif
constexpr int kTileLpr = DDX_TILE_LPR;                    // lanes per row
constexpr int kTileRps = 64 / kTileLpr;                   // rows per wave and step
constexpr int kTileSteps = (kTileRpw + kTileRps - 1) / kTileRps;
constexpr int kTileCap = DDX_TILE_CAP;                    // entries assembled in LDS
constexpr int kTileMaxW = DDX_TILE_MAXW;                  // widest window
constexpr size_t kTileLds = (size_t)kTileCap * 8 + (size_t)kTileMaxW * kTileWpc * 6 + ((size_t)kTileMaxW + 4) * 4 + 64;
static_assert(kTileLds <= 80 * 1024, "two tile workgroups per CU");
static_assert(kTileRows <= 65535, "segment ranks are kept in 16 bits");

#ifdef DDX_TILE_PROF
__device__ unsigned long long g_tile_prof[8];
#define TILE_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); if (i) atomicAdd(&g_tile_prof[i], now_ - stamp_); stamp_ = now_; } } while (0)
#else
#define TILE_STAMP(i) do { } while (0)
#endif
__global__ void __launch_bounds__(1024, 8) k_mirror_tiles(const int32_t* __restrict__ cols,
                                                          const float* __restrict__ raw, int64_t row_lo, int64_t row_hi, int32_t panel0,
                                                          int32_t H, int32_t W, int32_t wpp, int32_t ntiles, const uint32_t* __restrict__ bnd, const int64_t* __restrict__ colptr,
                                                          int32_t* __restrict__ row_out, float* __restrict__ raw_out) {
#ifdef DDX_TILE_PROF
    unsigned long long stamp_ = 0;
#endif
    extern __shared__ __align__(16) unsigned char tile_smem[];
    int32_t* stR = reinterpret_cast<int32_t*>(tile_smem);
    float* stV = reinterpret_cast<float*>(stR + kTileCap);
    uint32_t* bm = reinterpret_cast<uint32_t*>(stV + kTileCap);
    uint32_t* baseS = bm + kTileMaxW * kTileWpc;              // [w + 1] offsets of the window's columns inside the tile
    uint16_t* pre = reinterpret_cast<uint16_t*>(baseS + kTileMaxW + 4);   // [w x words] entries of the column in earlier words
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, quad = lane / kTileLpr, l16 = lane % kTileLpr;
    TILE_STAMP(0);
    // tile number: the blocks of one XCD (b % 8) walk a contiguous eighth of the tiles
    const int32_t per = (ntiles + 7) >> 3;
    const int32_t t = (int32_t)(blockIdx.x & 7) * per + (int32_t)(blockIdx.x >> 3);
    if (t >= ntiles) return;
    const int32_t p = t / wpp;
    const int32_t j0 = (t - p * wpp) * W;
    const int32_t j1 = j0 + W < H ? j0 + W : H;
    const int32_t w = j1 - j0;
    const int64_t prow0 = ((int64_t)panel0 + p) * kTileRows;
    const int64_t* cp = colptr + (int64_t)p * H;
    const int64_t gbase = cp[j0];

    for (int i = tid; i < w * kTileWpc; i += 1024) bm[i] = 0u;
    for (int i = tid; i <= w; i += 1024) baseS[i] = (uint32_t)(cp[j0 + i] - gbase);

    // 1. per row: its entries inside the window are [lo, hi) of the boundary table; the first 16 stay in registers
    const int rb = wave * kTileRpw + quad;
    const int32_t wnd = t - p * wpp;
    uint32_t lo[kTileSteps], hi[kTileSteps];
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) {
        const int rl = rb + kTileRps * s;
        const int64_t row = prow0 + rl;
        lo[s] = hi[s] = 0;
        if (kTileRps * s + quad < kTileRpw && rl < kTileRows && row >= row_lo && row < row_hi) {
            const uint32_t* bp = bnd + (size_t)(row - row_lo) * (size_t)(wpp + 1) + wnd;
            lo[s] = bp[0];
            hi[s] = bp[1];
        }
    }
    int32_t c1[kTileSteps];
    float v1[kTileSteps];
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) {
        const uint32_t q = lo[s] + l16;
        c1[s] = -1;
        v1[s] = 0.f;
        if (q < hi[s]) { c1[s] = cols[q]; v1[s] = raw[q]; }
    }
    __syncthreads();
    TILE_STAMP(1);
    TILE_STAMP(2);
    uint32_t more = 0;
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) {
        const int rl = rb + kTileRps * s;
        if (c1[s] >= 0) atomicOr(&bm[(c1[s] - j0) * kTileWpc + (rl >> 5)], 1u << (rl & 31));
        if (hi[s] - lo[s] > (uint32_t)kTileLpr) more |= 1u << s;
    }
    if (more) {                                        // (rare) rows with more than 16 entries inside the window
#pragma unroll 1
        for (int s = 0; s < kTileSteps; ++s) {
            if (!((more >> s) & 1u)) continue;
            const int rl = rb + kTileRps * s;
            const uint32_t* bp = bnd + (size_t)(prow0 + rl - row_lo) * (size_t)(wpp + 1) + wnd;
            for (uint32_t q = bp[0] + kTileLpr + l16; q < bp[1]; q += kTileLpr)
                atomicOr(&bm[(cols[q] - j0) * kTileWpc + (rl >> 5)], 1u << (rl & 31));
        }
    }
    __syncthreads();
    TILE_STAMP(3);
    // 2. per column: entries in the earlier words of the bitmap
    if (tid < w) {
        uint32_t run = 0;
        for (int k = 0; k < kTileWpc; ++k) {
            pre[tid * kTileWpc + k] = (uint16_t)run;
            run += __popc(bm[tid * kTileWpc + k]);
        }
    }
    __syncthreads();
    TILE_STAMP(4);
    // 3. place
    auto place = [&](int32_t c, float v, int rl) {
        const int jl = c - j0;
        const uint32_t word = bm[jl * kTileWpc + (rl >> 5)];
        const uint32_t pos = baseS[jl] + pre[jl * kTileWpc + (rl >> 5)] + __popc(word & ((1u << (rl & 31)) - 1u));
        if (pos < (uint32_t)kTileCap) { stR[pos] = (int32_t)(prow0 + rl); stV[pos] = v; }
        else { row_out[gbase + pos] = (int32_t)(prow0 + rl); raw_out[gbase + pos] = v; }
    };
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s)
        if (c1[s] >= 0) place(c1[s], v1[s], rb + kTileRps * s);
    if (more) {
#pragma unroll 1
        for (int s = 0; s < kTileSteps; ++s) {
            if (!((more >> s) & 1u)) continue;
            const int rl = rb + kTileRps * s;
            const uint32_t* bp = bnd + (size_t)(prow0 + rl - row_lo) * (size_t)(wpp + 1) + wnd;
            for (uint32_t q = bp[0] + kTileLpr + l16; q < bp[1]; q += kTileLpr) place(cols[q], raw[q], rl);
        }
    }
    __syncthreads();
    TILE_STAMP(5);
    // 4. the tile's range of the mirror, full lines
    const uint32_t ntile = baseS[w] < (uint32_t)kTileCap ? baseS[w] : (uint32_t)kTileCap;
    for (uint32_t i = tid; i < ntile; i += 1024) {
        row_out[gbase + i] = stR[i];
        raw_out[gbase + i] = stV[i];
    }
    TILE_STAMP(6);
}

// the CSR a mirror is built from: row pointer over all rows, columns, and the per-entry payload that travels with the row index
// (the raw counts for the full mirrors; the log-normalised values for the reduced mirror of the synthetic rows, k_bitplane.hip)
struct MirrorSrc { const int64_t* indptr; const int32_t* cols; const float* payload; };

static int build_csc_sorted(ddx_ctx* ctx, const MirrorSrc& src, int64_t e0, int64_t n, int64_t row_lo, int64_t row_hi, int32_t panel0,
                            int32_t npanels, DevBuf& colptr, DevBuf& rows, DevBuf& raws);

// Build the (panel, column)-ordered mirror of CSR entries [e0, e0+n) (rows [row_lo,row_hi)).
static int build_csc(ddx_ctx* ctx, const MirrorSrc& src, int64_t e0, int64_t n, int64_t row_lo, int64_t row_hi, int32_t panel0,
                     int32_t npanels, DevBuf& colptr, DevBuf& rows, DevBuf& raws) {
    const int32_t H = ctx->H;
    if (H > kMirrorMaxH || ctx->panel_rows % kMirrorSub != 0 || ctx->panel_rows / kMirrorSub >= 65536 || ctx->opt.mirror_mode == 0 || n == 0)
        return build_csc_sorted(ctx, src, e0, n, row_lo, row_hi, panel0, npanels, colptr, rows, raws);
    const int64_t nkeys = (int64_t)npanels * H;
    if (nkeys >= ((int64_t)1 << 30)) return set_err(ctx, DDX_E_UNSUPPORTED, "panel x column key space too large");
    const int32_t sub_rows = ctx->panel_rows / kMirrorSub;
    const int64_t sb0 = row_lo / sub_rows, sb1 = (row_hi - 1) / sub_rows + 1, n_sb = sb1 - sb0;
    DDX_TRY(ensure(ctx, colptr, sizeof(int64_t) * (nkeys + 1)));
    // scratch: cnt u16[n_sb x H] | tot u32[nkeys] | scan i64[nkeys]   (in the sort key buffer of the fallback path)
    size_t bytes = 0;
    auto piece = [&](size_t sz) { const size_t o = bytes; bytes += (sz + 255) & ~(size_t)255; return o; };
    const size_t o_cnt = piece(sizeof(uint16_t) * (size_t)n_sb * H), o_tot = piece(sizeof(uint32_t) * nkeys), o_scan = piece(sizeof(int64_t) * nkeys);
    DDX_TRY(ensure(ctx, ctx->sort_keys_in, bytes));
    unsigned char* base = ctx->sort_keys_in.as<unsigned char>();
    uint16_t* cnt = reinterpret_cast<uint16_t*>(base + o_cnt);
    uint32_t* tot = reinterpret_cast<uint32_t*>(base + o_tot);
    int64_t* scan = reinterpret_cast<int64_t*>(base + o_scan);
    DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_mirror_count), kMirrorMaxH * 4));
    DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_mirror_scatter), kMirrorMaxH * 4));
    const size_t lds = sizeof(uint32_t) * (size_t)H;
    // tile placement: window width = 70 % of the LDS range at the average segment length
    const bool tiles = ctx->opt.mirror_mode == 2 && ctx->panel_rows == kTileRows;
    int32_t W = 0, wpp = 0;
    uint32_t* bnd = nullptr;
    if (tiles) {
        const double avg = (double)n / (double)nkeys;
        W = (int32_t)(DDX_TILE_FILL * kTileCap / (avg > 1.0 ? avg : 1.0));
        W = W < 8 ? 8 : (W > kTileMaxW ? kTileMaxW : W);
        wpp = (int32_t)ceil_div((int64_t)H, (int64_t)W);                     // windows per panel
        if ((int64_t)wpp * npanels >= ((int64_t)1 << 30)) return set_err(ctx, DDX_E_UNSUPPORTED, "too many mirror tiles");
        DDX_TRY(ensure(ctx, ctx->sort_keys_out, sizeof(uint32_t) * (size_t)(row_hi - row_lo) * (size_t)(wpp + 1)));
        bnd = ctx->sort_keys_out.as<uint32_t>();
    }
    ScopedTimer t(ctx, "mirror_build");
    k_mirror_count<<<(unsigned)n_sb, kMirrorThreads, lds, ctx->stream>>>(src.indptr, src.cols, row_lo, row_hi, sub_rows, sb0, H, cnt, W, wpp, bnd);
    k_mirror_prefix<<<(unsigned)ceil_div(nkeys, 256), 256, 0, ctx->stream>>>(cnt, sb0, n_sb, panel0, npanels, H, tot);
    size_t tmp_bytes = 0;
    DDX_HIP(ctx, prim::exclusive_sum(nullptr, tmp_bytes, tot, scan, (size_t)nkeys, ctx->stream));
    DDX_TRY(ensure(ctx, ctx->sort_tmp, tmp_bytes));
    DDX_HIP(ctx, prim::exclusive_sum(ctx->sort_tmp.p, tmp_bytes, tot, scan, (size_t)nkeys, ctx->stream));
    k_colptr_from_totals<<<(unsigned)ceil_div(nkeys + 1, 256), 256, 0, ctx->stream>>>(scan, nkeys, n, colptr.as<int64_t>());
    if (tiles) {
        const int64_t ntiles = (int64_t)wpp * npanels;
        DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_mirror_tiles), (int)kTileLds));
        k_mirror_tiles<<<(unsigned)(ceil_div(ntiles, (int64_t)8) * 8), 1024, kTileLds, ctx->stream>>>(
            src.cols, src.payload, row_lo, row_hi, panel0, H, W, wpp, (int32_t)ntiles, bnd, colptr.as<int64_t>(), rows.as<int32_t>(), raws.as<float>());
#ifdef DDX_TILE_PROF
        {
            unsigned long long h[8];
            wait_stream(ctx);
            hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile_prof), sizeof(h));
            fprintf(stderr, "tile phases (100 MHz ticks per tile, %lld tiles):", (long long)ntiles);
            for (int i = 1; i < 7; ++i) fprintf(stderr, " %.1f", (double)h[i] / (double)ntiles);
            fprintf(stderr, "\n");
            unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            hipMemcpyToSymbol(HIP_SYMBOL(g_tile_prof), z, sizeof(z));
        }
#endif
    } else {
        k_mirror_scatter<<<(unsigned)n_sb, kMirrorThreads, lds, ctx->stream>>>(src.indptr, src.cols,
                                                                               src.payload, row_lo, row_hi, sub_rows, sb0, panel0, H, cnt,
                                                                               colptr.as<int64_t>(), rows.as<int32_t>(), raws.as<float>());
    }
    DDX_HIP(ctx, hipGetLastError());
    (void)e0;
    return DDX_OK;
}

// Fallback (very wide matrices, DDX_MIRROR=sort): the same mirror by a stable radix sort.
// Build the (panel, column)-ordered mirror of CSR entries [e0, e0+n) (rows [row_lo,row_hi)): stable radix
// sort of (key, position) pairs -> inside a (panel, column) segment entries are in increasing row order.
static int build_csc_sorted(ddx_ctx* ctx, const MirrorSrc& src, int64_t e0, int64_t n, int64_t row_lo, int64_t row_hi, int32_t panel0,
                            int32_t npanels, DevBuf& colptr, DevBuf& rows, DevBuf& raws) {
    const int32_t H = ctx->H;
    const int64_t nkeys64 = (int64_t)npanels * H;
    if (nkeys64 >= ((int64_t)1 << 30)) return set_err(ctx, DDX_E_UNSUPPORTED, "panel x column key space too large");
    const int32_t nkeys = (int32_t)nkeys64;
    DDX_TRY(ensure(ctx, colptr, sizeof(int64_t) * (nkeys + 1)));
    if (n == 0) {
        DDX_HIP(ctx, hipMemsetAsync(colptr.p, 0, sizeof(int64_t) * (nkeys + 1), ctx->stream));
        return DDX_OK;
    }
    DDX_TRY(ensure(ctx, ctx->sort_keys_in, sizeof(int32_t) * n));
    DDX_TRY(ensure(ctx, ctx->sort_keys_out, sizeof(int32_t) * n));
    DDX_TRY(ensure(ctx, ctx->sort_vals_in, sizeof(uint32_t) * n));
    DDX_TRY(ensure(ctx, ctx->sort_vals_out, sizeof(uint32_t) * n));
    DDX_TRY(ensure(ctx, ctx->sort_rowid, sizeof(int32_t) * n));
    k_iota_u32<<<(unsigned)ceil_div(n, 256), 256, 0, ctx->stream>>>(ctx->sort_vals_in.as<uint32_t>(), n, (uint32_t)e0);
    k_panel_keys<<<(unsigned)ceil_div(row_hi - row_lo, 4), 256, 0, ctx->stream>>>(src.indptr, src.cols,
                                                                                 row_lo, row_hi - row_lo, H, panel0, ctx->panel_rows, e0,
                                                                                 ctx->sort_keys_in.as<int32_t>(), ctx->sort_rowid.as<int32_t>());
    int end_bit = 1;
    while (((int64_t)1 << end_bit) < nkeys64) ++end_bit;
    size_t tmp_bytes = 0;
    DDX_HIP(ctx, prim::sort_pairs(nullptr, tmp_bytes, ctx->sort_keys_in.as<int32_t>(), ctx->sort_keys_out.as<int32_t>(),
                                                    ctx->sort_vals_in.as<uint32_t>(), ctx->sort_vals_out.as<uint32_t>(),
                                                    (int)n, 0, end_bit, ctx->stream));
    DDX_TRY(ensure(ctx, ctx->sort_tmp, tmp_bytes));
    {
        ScopedTimer t(ctx, "csc_radix_sort");
        DDX_HIP(ctx, prim::sort_pairs(ctx->sort_tmp.p, tmp_bytes, ctx->sort_keys_in.as<int32_t>(),
                                                        ctx->sort_keys_out.as<int32_t>(), ctx->sort_vals_in.as<uint32_t>(),
                                                        ctx->sort_vals_out.as<uint32_t>(), (int)n, 0, end_bit, ctx->stream));
    }
    {
        ScopedTimer t(ctx, "csc_gather");
        k_colptr_from_sorted<<<(unsigned)ceil_div(nkeys + 1, 256), 256, 0, ctx->stream>>>(ctx->sort_keys_out.as<int32_t>(), n, nkeys,
                                                                                          colptr.as<int64_t>());
        k_csc_gather<<<(unsigned)ceil_div(n, 256), 256, 0, ctx->stream>>>(ctx->sort_vals_out.as<uint32_t>(), n, e0, ctx->sort_rowid.as<int32_t>(),
                                                                          src.payload, rows.as<int32_t>(), raws.as<float>());
    }
    DDX_HIP(ctx, hipGetLastError());
    return DDX_OK;
}

// grow a buffer while keeping its first keep_bytes.  The old block is abandoned to the arena (handed out again when the context is
// reset for its next fit), not released: a follower context may be copying the original cells' part of it at this moment (CloneView)
static int ensure_keep(ddx_ctx* ctx, DevBuf& b, size_t bytes, size_t keep_bytes) {
    if (bytes <= b.cap && b.p) return DDX_OK;
    DevBuf nb;
    DDX_TRY(ensure(ctx, nb, bytes));
    if (b.p && keep_bytes) {
        hipError_t e = hipMemcpyAsync(nb.p, b.p, keep_bytes, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = wait_stream(ctx);
        if (e != hipSuccess) {
            release(ctx, nb);
            return set_err(ctx, DDX_E_HIP, "device copy during growth failed: %s", hipGetErrorString(e));
        }
    }
    b = nb;
    return DDX_OK;
}

// ------------------------------------------------------------------------------------------------
// stage: upload counts
// ------------------------------------------------------------------------------------------------
// the (panel, column)-ordered mirror of the original rows (counts as the payload; the values follow per iteration)
static int originals_mirror(ddx_ctx* ctx) {
    if (ctx->mirror_o) return DDX_OK;
    const int64_t nnz = ctx->nnz;
    DDX_TRY(ensure(ctx, ctx->csc_o_row, sizeof(int32_t) * (size_t)(nnz + 1)));
    DDX_TRY(ensure(ctx, ctx->csc_o_raw, sizeof(float) * (size_t)(nnz + 1)));
    DDX_TRY(ensure(ctx, ctx->csc_o_x, sizeof(float) * (size_t)(nnz + 1)));
    const MirrorSrc full{ctx->aug_indptr.as<int64_t>(), ctx->aug_indices.as<int32_t>(), ctx->aug_raw.as<float>()};
    DDX_TRY(build_csc(ctx, full, 0, nnz, 0, ctx->N, 0, ctx->P_o, ctx->csc_o_colptr, ctx->csc_o_row, ctx->csc_o_raw));
    ctx->mirror_o = true;
    return DDX_OK;
}

// bit-plane mode, once per fit (bp_build): the reduced mirror of the original rows from their reduced CSR, into the per-fit buffer
int bp_originals_mirror(ddx_ctx* ctx) {
    BitPlanes& bp = ctx->bp;
    const size_t nseg = (size_t)ctx->P_o * ctx->H;
    DevBuf cp, rw, pl;                                             // (views into ctx->bp_buf: nothing to grow, nothing to release)
    cp.p = bp.restm_colptr; cp.cap = sizeof(int64_t) * (nseg + 1);
    rw.p = bp.restm_row; rw.cap = sizeof(int32_t) * (size_t)bp.nrest_o + 256;
    pl.p = bp.restm_raw; pl.cap = sizeof(float) * (size_t)bp.nrest_o + 256;
    const MirrorSrc red{bp.rest_indptr, bp.rest_cols, bp.rest_raw};
    return build_csc(ctx, red, 0, bp.nrest_o, 0, ctx->N, 0, ctx->P_o, cp, rw, pl);
}

int stage_upload_counts(ddx_ctx* ctx, int64_t N, int32_t H, const int64_t* indptr, const int32_t* indices,
                        const float* data, bool from_device) {
    const int64_t nnz = from_device ? ctx->nnz : indptr[N];
    if (nnz >= (int64_t)1 << 31) return set_err(ctx, DDX_E_UNSUPPORTED, "more than 2^31-1 stored entries");
    if (!from_device) {
        arena_hint(ctx, (size_t)nnz * 90 + (size_t)N * 6000 + ((size_t)1 << 30));
        // worst-case-ish room for the synthetic part (default boost_rate 0.25 needs ~0.5*nnz); grows on demand
        const int64_t cap_s = nnz / 2 + nnz / 8 + 1024;
        DDX_TRY(ensure(ctx, ctx->aug_indptr, sizeof(int64_t) * (N + N / 2 + 2)));
        DDX_TRY(ensure(ctx, ctx->aug_indices, sizeof(int32_t) * (size_t)(nnz + cap_s)));
        DDX_TRY(ensure(ctx, ctx->aug_raw, sizeof(float) * (size_t)(nnz + cap_s)));
        DDX_TRY(ensure(ctx, ctx->aug_x, sizeof(float) * (size_t)(nnz + cap_s)));
        ctx->cap_synth = cap_s;
        DDX_HIP(ctx, hipMemcpyAsync(ctx->aug_indptr.p, indptr, sizeof(int64_t) * (N + 1), hipMemcpyHostToDevice, ctx->stream));
        if (nnz) {
            DDX_HIP(ctx, hipMemcpyAsync(ctx->aug_indices.p, indices, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, ctx->stream));
            DDX_HIP(ctx, hipMemcpyAsync(ctx->aug_raw.p, data, sizeof(float) * nnz, hipMemcpyHostToDevice, ctx->stream));
        }
        ctx->h_indptr.assign(indptr, indptr + N + 1);
        ctx->have_counts = false;
        DDX_TRY(validate_csr(ctx, ctx->aug_indptr.as<int64_t>(), ctx->aug_indices.as<int32_t>(), ctx->aug_raw.as<float>(), N, H));
    }
    ctx->N = N;
    ctx->H = H;
    ctx->nnz = nnz;
    ctx->S = 0;
    ctx->M = N;
    ctx->have_synth = ctx->have_lognorm = ctx->scaled = ctx->have_emb = ctx->have_knn = false;
    // whatever was derived from the previous resident rows is stale: the slice segments of the A Q pass (a second
    // ddx_select_columns of the same width would otherwise keep the old rows' segments) and the bit planes
    ctx->rowseg_rows = -1;
    ctx->bp.ready = ctx->bp.values = ctx->bp.scaled = false;
    ctx->bp.demote.clear();
    ctx->bp.n_demoted = 0;
    ctx->bp.demote_decided = false;
    DDX_TRY(ensure(ctx, ctx->lib32, sizeof(float) * (N + N / 2 + 2)));
    DDX_TRY(ensure(ctx, ctx->lib64, sizeof(double) * (N + N / 2 + 2)));
    {
        ScopedTimer t(ctx, "row_sums");
        DDX_TRY(ensure(ctx, ctx->median, 256));
        int one = 1, exact = 0;
        int* flag = ctx->median.as<int>() + 8;
        DDX_HIP(ctx, hipMemcpyAsync(flag, &one, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        k_row_sums_exact<<<(unsigned)ceil_div(N, 4), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_raw.as<float>(), 0, N,
                                                                            ctx->lib32.as<float>(), ctx->lib64.as<double>());
        const int64_t span = nnz > N ? nnz : N;
        k_counts_exact<<<(unsigned)ceil_div(span, 256), 256, 0, ctx->stream>>>(ctx->aug_raw.as<float>(), nnz, ctx->lib32.as<float>(), N, flag);
        DDX_HIP(ctx, hipMemcpyAsync(&exact, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        DDX_HIP(ctx, wait_stream(ctx));
        ctx->counts_exact = exact != 0 && !ctx->opt.row_sums_sequential;
        if (!ctx->counts_exact)           // fractional / negative / huge counts: scipy's sequential order, replayed
            k_row_sums<<<(unsigned)ceil_div(N, 4), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_raw.as<float>(), 0, N,
                                                                          ctx->lib32.as<float>(), ctx->lib64.as<double>());
    }
    ctx->panel_rows = (ctx->opt.spmm_lds && ctx->opt.gather_f32) ? kLdsPanelRows : kGatherPanelRows;
    ctx->P_o = (int32_t)ceil_div(N, ctx->panel_rows);
    ctx->mirror_o = false;
    if (bp_wanted_at_upload(ctx)) {
        // bitmaps + reduced structures of the original rows, once per fit (the followers copy them).  The bit-plane route never reads
        // the full column-major mirror of the original rows -- its reduced mirror is built from the reduced rows --, so that one
        // (a pass over all stored entries, 8 bytes each to keep and to copy to every follower) waits for somebody to ask: ensure_full_mirror
        DDX_TRY(bp_build(ctx));
    } else {
        DDX_TRY(originals_mirror(ctx));
    }
    DDX_HIP(ctx, wait_stream(ctx));
    ctx->have_counts = true;
    publish_clone_view(ctx);
    return DDX_OK;
}

// the resident counts are complete (the stream has been drained): what followers may copy from now on, see CloneView
void publish_clone_view(ddx_ctx* ctx) {
    std::lock_guard<std::mutex> lock(ctx->view_mu);
    CloneView& v = ctx->view;
    v.N = ctx->N; v.nnz = ctx->nnz; v.H = ctx->H; v.panel_rows = ctx->panel_rows; v.P_o = ctx->P_o;
    v.counts_exact = ctx->counts_exact; v.mirror_o = ctx->mirror_o;
    v.aug_indptr = ctx->aug_indptr.p; v.aug_indices = ctx->aug_indices.p; v.aug_raw = ctx->aug_raw.p;
    v.lib32 = ctx->lib32.p; v.lib64 = ctx->lib64.p;
    v.csc_o_colptr = ctx->csc_o_colptr.p; v.csc_o_row = ctx->csc_o_row.p; v.csc_o_raw = ctx->csc_o_raw.p;
    v.h_indptr = ctx->h_indptr;
    v.bp = ctx->bp;
    v.bp_buf = ctx->bp_buf.p;
    v.valid = true;
}

// ------------------------------------------------------------------------------------------------
// stage: clone the resident counts of another context on the same GPU (device-to-device)
// ------------------------------------------------------------------------------------------------
int stage_clone_counts(ddx_ctx* ctx, ddx_ctx* src) {
    // everything is read from the view the source published when its counts became resident -- never from the source's live state, which
    // its own iterations may be changing on another thread at this moment
    CloneView v;
    {
        std::lock_guard<std::mutex> lock(src->view_mu);
        v = src->view;
    }
    if (!v.valid) return set_err(ctx, DDX_E_ARG, "source context holds no counts");
    const int64_t N = v.N, nnz = v.nnz;
    const int32_t H = v.H;
    const int64_t cap_s = nnz / 2 + nnz / 8 + 1024;
    ctx->have_counts = false;
    arena_hint(ctx, (size_t)nnz * 90 + (size_t)N * 6000 + ((size_t)1 << 30));
    DDX_TRY(ensure(ctx, ctx->aug_indptr, sizeof(int64_t) * (N + N / 2 + 2)));
    DDX_TRY(ensure(ctx, ctx->aug_indices, sizeof(int32_t) * (size_t)(nnz + cap_s)));
    DDX_TRY(ensure(ctx, ctx->aug_raw, sizeof(float) * (size_t)(nnz + cap_s)));
    DDX_TRY(ensure(ctx, ctx->aug_x, sizeof(float) * (size_t)(nnz + cap_s)));
    DDX_TRY(ensure(ctx, ctx->lib32, sizeof(float) * (N + N / 2 + 2)));
    DDX_TRY(ensure(ctx, ctx->lib64, sizeof(double) * (N + N / 2 + 2)));
    const size_t cp_bytes = sizeof(int64_t) * ((size_t)v.P_o * H + 1);
    if (v.mirror_o) {
        DDX_TRY(ensure(ctx, ctx->csc_o_row, sizeof(int32_t) * (size_t)(nnz + 1)));
        DDX_TRY(ensure(ctx, ctx->csc_o_raw, sizeof(float) * (size_t)(nnz + 1)));
        DDX_TRY(ensure(ctx, ctx->csc_o_x, sizeof(float) * (size_t)(nnz + 1)));
        DDX_TRY(ensure(ctx, ctx->csc_o_colptr, cp_bytes));
    }
    DDX_TRY(ensure(ctx, ctx->median, 256));
    auto copy = [&](DevBuf& d, const void* s, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpyAsync(d.p, s, bytes, hipMemcpyDeviceToDevice, ctx->stream) : hipSuccess;
    };
    DDX_HIP(ctx, copy(ctx->aug_indptr, v.aug_indptr, sizeof(int64_t) * (N + 1)));
    DDX_HIP(ctx, copy(ctx->aug_indices, v.aug_indices, sizeof(int32_t) * nnz));
    DDX_HIP(ctx, copy(ctx->aug_raw, v.aug_raw, sizeof(float) * nnz));
    DDX_HIP(ctx, copy(ctx->lib32, v.lib32, sizeof(float) * N));
    DDX_HIP(ctx, copy(ctx->lib64, v.lib64, sizeof(double) * N));
    if (v.mirror_o) {
        DDX_HIP(ctx, copy(ctx->csc_o_colptr, v.csc_o_colptr, cp_bytes));
        DDX_HIP(ctx, copy(ctx->csc_o_row, v.csc_o_row, sizeof(int32_t) * nnz));
        DDX_HIP(ctx, copy(ctx->csc_o_raw, v.csc_o_raw, sizeof(float) * nnz));
    }
    ctx->mirror_o = v.mirror_o;
    ctx->cap_synth = cap_s;
    ctx->h_indptr = std::move(v.h_indptr);
    ctx->N = N;
    ctx->H = H;
    ctx->nnz = nnz;
    ctx->S = 0;
    ctx->M = N;
    ctx->panel_rows = v.panel_rows;
    ctx->P_o = v.P_o;
    ctx->counts_exact = v.counts_exact;
    ctx->have_synth = ctx->have_lognorm = ctx->scaled = ctx->have_emb = ctx->have_knn = false;
    ctx->rowseg_rows = -1;
    DDX_TRY(bp_clone(ctx, v));
    DDX_HIP(ctx, wait_stream(ctx));
    ctx->have_counts = true;
    publish_clone_view(ctx);
    return DDX_OK;
}

// ------------------------------------------------------------------------------------------------
// stage: create doublets
// ------------------------------------------------------------------------------------------------
static int lognorm_rows(ddx_ctx* ctx);
static int scale_full_rows(ddx_ctx* ctx);
static int scale_full_mirror(ddx_ctx* ctx);

// one merge per doublet: rows are written at padded offsets into the (not yet rebuilt) mirror buffers, counted on the way,
// scanned into the row pointer and moved to their final positions
static int materialise_synthetic(ddx_ctx* ctx) {
    const int64_t N = ctx->N, S = ctx->S;
    if (S) {
        DDX_TRY(ensure(ctx, ctx->pad_off, sizeof(int64_t) * (S + 1)));
        DDX_HIP(ctx, hipMemcpyAsync(ctx->pad_off.p, ctx->h_pad_off.data(), sizeof(int64_t) * (S + 1), hipMemcpyHostToDevice, ctx->stream));
        {
            ScopedTimer t(ctx, "doublet_fill");
            k_doublet_fill<<<(unsigned)S, 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_indices.as<int32_t>(),
                                                                 ctx->aug_raw.as<float>(), ctx->parents.as<int64_t>(), ctx->pad_off.as<int64_t>(),
                                                                 ctx->csc_s_row.as<int32_t>(), ctx->csc_s_raw.as<float>(), ctx->synth_counts.as<int32_t>());
        }
        k_scan_counts<<<1, 1024, 0, ctx->stream>>>(ctx->synth_counts.as<int32_t>(), S, ctx->nnz, ctx->aug_indptr.as<int64_t>() + N);
        {
            ScopedTimer t(ctx, "doublet_compact");
            k_doublet_compact<<<(unsigned)ceil_div(S, 4), 256, 0, ctx->stream>>>(ctx->pad_off.as<int64_t>(), ctx->csc_s_row.as<int32_t>(), ctx->csc_s_raw.as<float>(),
                                                                                 ctx->aug_indptr.as<int64_t>() + N, S, ctx->aug_indices.as<int32_t>(),
                                                                                 ctx->aug_raw.as<float>());
        }
        DDX_HIP(ctx, hipGetLastError());
        ctx->mirror_full = false;             // (the fill used the synthetic mirror's buffers as its scratch)
    }
    ctx->synth_rows = true;
    return DDX_OK;
}

int stage_create_doublets(ddx_ctx* ctx, int64_t S, const int64_t* parents) {
    const int64_t N = ctx->N;
    // capacity from the host copy of the row pointer: |row p0| + |row p1| bounds each synthetic row; the running sum
    // is where the fill kernel writes row s before the rows are compacted to their true lengths
    int64_t cap = 0;
    ctx->h_pad_off.resize((size_t)S + 1);
    for (int64_t s = 0; s < S; ++s) {
        const int64_t a = parents[2 * s], b = parents[2 * s + 1];
        ctx->h_pad_off[s] = cap;
        cap += (ctx->h_indptr[a + 1] - ctx->h_indptr[a]) + (ctx->h_indptr[b + 1] - ctx->h_indptr[b]);
    }
    ctx->h_pad_off[S] = cap;
    if (ctx->nnz + cap >= (int64_t)1 << 31) return set_err(ctx, DDX_E_UNSUPPORTED, "augmented matrix exceeds 2^31-1 entries");
    if (cap > ctx->cap_synth || !ctx->aug_x.p) {
        const size_t tot = (size_t)(ctx->nnz + cap + 1024);
        DDX_TRY(ensure_keep(ctx, ctx->aug_indices, sizeof(int32_t) * tot, sizeof(int32_t) * ctx->nnz));
        DDX_TRY(ensure_keep(ctx, ctx->aug_raw, sizeof(float) * tot, sizeof(float) * ctx->nnz));
        DDX_TRY(ensure_keep(ctx, ctx->aug_x, sizeof(float) * tot, 0));
        ctx->cap_synth = cap + 1024;
    }
    DDX_TRY(ensure_keep(ctx, ctx->aug_indptr, sizeof(int64_t) * (N + S + 2), sizeof(int64_t) * (N + 1)));
    DDX_TRY(ensure_keep(ctx, ctx->lib32, sizeof(float) * (N + S + 2), sizeof(float) * N));
    DDX_TRY(ensure_keep(ctx, ctx->lib64, sizeof(double) * (N + S + 2), sizeof(double) * N));
    DDX_TRY(ensure(ctx, ctx->parents, sizeof(int64_t) * 2 * (S + 1)));
    DDX_TRY(ensure(ctx, ctx->synth_counts, sizeof(int32_t) * (S + 1)));
    DDX_TRY(ensure(ctx, ctx->csc_s_row, sizeof(int32_t) * (size_t)(ctx->cap_synth + 1)));
    DDX_TRY(ensure(ctx, ctx->csc_s_raw, sizeof(float) * (size_t)(ctx->cap_synth + 1)));
    DDX_TRY(ensure(ctx, ctx->csc_s_x, sizeof(float) * (size_t)(ctx->cap_synth + 1)));
    ctx->S = S;
    ctx->M = N + S;
    ctx->have_lognorm = ctx->scaled = ctx->have_emb = ctx->have_knn = false;
    ctx->mirror_full = false;
    ctx->bp.values = false;
    ctx->synth_rows = ctx->rows_x = false;
    if (S) DDX_HIP(ctx, hipMemcpyAsync(ctx->parents.p, parents, sizeof(int64_t) * 2 * S, hipMemcpyHostToDevice, ctx->stream));
    ctx->have_synth = true;
    // The bit-plane route derives everything it needs of a doublet from its parents' structures (k_bp_synth): the merged rows
    // are then only written when somebody asks for them (ensure_full_rows)
    if (bp_lean(ctx)) return DDX_OK;
    return materialise_synthetic(ctx);
}

bool bp_lean(const ddx_ctx* ctx) {
    return ctx->opt.synthetic_derived && bp_wanted_at_upload(ctx) && ctx->bp.ready && ctx->counts_exact && ctx->S > 0 && ctx->S <= ctx->N / 2 &&
           ctx->bp.SKc * 8 * 20 * sizeof(uint32_t) <= 128 * 1024;
}

// rows N..M of the row-major arrays for the current doublets, and this iteration's values of all rows
int ensure_full_rows(ddx_ctx* ctx) {
    if (!ctx->have_synth) return DDX_OK;
    if (!ctx->synth_rows) {
        DDX_TRY(materialise_synthetic(ctx));
        DDX_HIP(ctx, hipMemcpyAsync(&ctx->nnz_aug, ctx->aug_indptr.as<int64_t>() + ctx->M, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        DDX_HIP(ctx, wait_stream(ctx));
    }
    if (!ctx->rows_x && ctx->have_lognorm) DDX_TRY(lognorm_rows(ctx));
    // (a matrix scaled on the bit-plane structures: the row-major values follow when somebody asks for them)
    if (ctx->scaled && ctx->have_lognorm && !ctx->rows_scaled) DDX_TRY(scale_full_rows(ctx));
    return DDX_OK;
}

// ------------------------------------------------------------------------------------------------
// normalisation
// ------------------------------------------------------------------------------------------------
// np.median of M float32 values from the ascending sort: middle element, or the float32 mean of the
// two middle elements.
__global__ void k_median_from_sorted(const float* __restrict__ sorted, int64_t M, float* __restrict__ med) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (M & 1) med[0] = sorted[M / 2];
        else med[0] = __fdiv_rn(__fadd_rn(sorted[M / 2 - 1], sorted[M / 2]), 2.0f);
    }
}

// Most stored entries are small integer counts, and lognorm_value is a function of (count, row): in the row-major pass
// the values of the counts 1..16 are evaluated once per row and every entry with such a count takes its value from
// there -- the same function of the same arguments, hence the same bits.  Anything else (larger or fractional counts, explicit zeros) is queued per wave and evaluated 64
// at a time, so that the ~150 float64 instructions of a division and a logarithm are spent on full waves only.

__global__ void k_lognorm_table(const double* __restrict__ lib64, const float* __restrict__ med, float pc, int use_log1p, int64_t M,
                                float* __restrict__ tab) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * kLognormTab) return;
    const int64_t row = t / kLognormTab;
    const int c = (int)(t - row * kLognormTab) + 1;
    tab[t] = lognorm_value((float)c, lib64[row], med[0], pc, use_log1p != 0);
}

// per-wave queue of entries that need the full evaluation
struct LognormQueue {
    int64_t pos[128];
    float v[128];
    int32_t row[128];
};

__device__ __forceinline__ void lognorm_enqueue(LognormQueue& q, int& qn, bool slow, int64_t pos, float v, int32_t row, int lane,
                                                const double* __restrict__ lib64, float m, float pc, bool use_log1p, float* __restrict__ x) {
    const unsigned long long mask = __ballot(slow);
    if (mask == 0ull) return;
    if (slow) {
        const int at = qn + __popcll(mask & ((1ull << lane) - 1ull));
        q.pos[at] = pos; q.v[at] = v; q.row[at] = row;
    }
    qn += __popcll(mask);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (qn >= 64) {
        const int64_t p0 = q.pos[lane];
        x[p0] = lognorm_value(q.v[lane], lib64[q.row[lane]], m, pc, use_log1p);
        const int rest = qn - 64;
        int64_t p1 = 0; float v1 = 0.f; int32_t r1 = 0;
        if (lane < rest) { p1 = q.pos[64 + lane]; v1 = q.v[64 + lane]; r1 = q.row[64 + lane]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < rest) { q.pos[lane] = p1; q.v[lane] = v1; q.row[lane] = r1; }
        qn = rest;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ void lognorm_flush(LognormQueue& q, int qn, int lane, const double* __restrict__ lib64, float m, float pc,
                                              bool use_log1p, float* __restrict__ x) {
    if (lane < qn) x[q.pos[lane]] = lognorm_value(q.v[lane], lib64[q.row[lane]], m, pc, use_log1p);
}

// row-major pass: a workgroup of four waves takes consecutive blocks of rows; a wave walks its rows 64 entries at a time
__global__ void __launch_bounds__(256) k_lognorm_rows(const int64_t* __restrict__ indptr, const float* __restrict__ raw,
                                                      const double* __restrict__ lib64, const float* __restrict__ med,
                                                      const float* __restrict__ tab, float pc, int use_log1p, int64_t M, int rows_per_wave,
                                                      float* __restrict__ x) {
    __shared__ LognormQueue queue[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LognormQueue& q = queue[wave];
    int qn = 0;
    const float m = med[0];
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
    for (int64_t row = r0; row < r0 + rows_per_wave && row < M; ++row) {
        const int64_t b = indptr[row], e = indptr[row + 1];
        const float tv = tab[row * kLognormTab + (lane & (kLognormTab - 1))];     // lane c-1 holds the value of count c
        for (int64_t base = b; base < e; base += 256) {      // all lanes stay in the loop: they are shuffle sources
            float v4[4];                                      // four loads in flight per lane
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] = base + 64 * u + lane < e ? raw[base + 64 * u + lane] : 1.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t p = base + 64 * u + lane;
                const bool valid = p < e;
                const float v = v4[u];
                const int c = small_count(v, kLognormTab);
                const float r = __shfl(tv, c ? c - 1 : 0, 64);
                if (valid && c) x[p] = r;
                lognorm_enqueue(q, qn, valid && !c, p, v, (int32_t)row, lane, lib64, m, pc, use_log1p != 0, x);
            }
        }
    }
    lognorm_flush(q, qn, lane, lib64, m, pc, use_log1p != 0, x);
}

// column-major pass: the (row, count) table again.  The mirror is ordered by (panel, column, row): a workgroup takes a
// sixteenth of one panel's entries and first copies the panel's slice of the table (784 rows x 16 counts, 50 KB) into
// LDS, so the per-entry lookup is an LDS read; entries outside the table are evaluated in place by the lanes that hold
// them.  Measured per iteration, both mirrors: every entry evaluated in place 0.58 ms (VALU-bound: ~150 float64
// instructions per entry); the table gathered from global memory 0.84 ms (8 MB, one line request per entry); plain copy
// of the three streams 0.29 ms.
constexpr int kLognormParts = 16;       // workgroups per panel
__global__ void __launch_bounds__(512) k_lognorm_csc(const int32_t* __restrict__ rows, const float* __restrict__ raw,
                                                     const int64_t* __restrict__ colptr, int32_t H, int32_t panel0, int32_t panel_rows,
                                                     int64_t M, const double* __restrict__ lib64, const float* __restrict__ med,
                                                     const float* __restrict__ tab, float pc, int use_log1p, float* __restrict__ x) {
    extern __shared__ __align__(16) float lognorm_tabS[];
    float* tabS = lognorm_tabS;
    const int32_t p = (int32_t)(blockIdx.x / kLognormParts), part = (int32_t)(blockIdx.x % kLognormParts);
    const int64_t prow0 = ((int64_t)panel0 + p) * panel_rows;
    const int64_t nrow = M - prow0 < panel_rows ? M - prow0 : panel_rows;
    for (int64_t i = threadIdx.x; i < nrow * kLognormTab; i += 512) tabS[i] = tab[prow0 * kLognormTab + i];
    const int64_t b = colptr[(int64_t)p * H], e = colptr[(int64_t)(p + 1) * H];
    const int64_t lo = b + (e - b) * part / kLognormParts, hi = b + (e - b) * (part + 1) / kLognormParts;
    const float m = med[0];
    __syncthreads();
    auto value = [&](int32_t r, float v) -> float {
        const int c = small_count(v, kLognormTab);
        return c ? tabS[(int64_t)(r - prow0) * kLognormTab + (c - 1)] : lognorm_value(v, lib64[r], m, pc, use_log1p != 0);
    };
    // entries in front of / behind the 16-byte aligned body, then the body four entries per lane and load
    const int64_t up = (lo + 3) & ~(int64_t)3;
    const int64_t lo4 = up < hi ? up : hi;
    const int64_t hi4 = lo4 + ((hi - lo4) & ~(int64_t)3);
    {
        const int64_t nh = lo4 - lo, nt = hi - hi4;
        if (threadIdx.x < nh + nt) {
            const int64_t t = threadIdx.x < nh ? lo + threadIdx.x : hi4 + (threadIdx.x - nh);
            x[t] = value(rows[t], raw[t]);
        }
    }
    typedef int32_t i4v __attribute__((ext_vector_type(4)));
    typedef float f4v_ __attribute__((ext_vector_type(4)));
    for (int64_t t = lo4 + 4 * threadIdx.x; t < hi4; t += 2048) {
        const i4v r = *reinterpret_cast<const i4v*>(rows + t);
        const f4v_ v = *reinterpret_cast<const f4v_*>(raw + t);
        f4v_ out;
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = value(r[i], v[i]);
        *reinterpret_cast<f4v_*>(x + t) = out;
    }
}

// (panels too tall for the table in LDS -- the gather geometry, DDX_SPMM=gather: every entry evaluated in place)
__global__ void __launch_bounds__(256) k_lognorm_csc_direct(const int32_t* __restrict__ rows, const float* __restrict__ raw,
                                                            const int64_t* __restrict__ colptr, int32_t nkeys,
                                                            const double* __restrict__ lib64, const float* __restrict__ med,
                                                            float pc, int use_log1p, float* __restrict__ x) {
    const int64_t n = colptr[nkeys];
    const float m = med[0];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        x[t] = lognorm_value(raw[t], lib64[rows[t]], m, pc, use_log1p != 0);
}

__global__ void k_fill_f32(float* out, int64_t n, float v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

// Column sums over the stored entries of both column-major mirrors, in two deterministic stages.  A (panel, column)
// segment is contiguous, and so are the segments of consecutive columns of one panel: stage 1 gives every segment to
// 16 lanes (lane-strided float64 partial sums, fixed butterfly), stage 2 adds the panels of a column in order.
// mode 0: colmean[j] = sum(x - z_j) / M.
// mode 1 (scale statistics): stat[2j] = sum(x - z_j), stat[2j+1] = sum(float32(x*x)).
__global__ void __launch_bounds__(256) k_col_partials(const int64_t* __restrict__ cp, const float* __restrict__ x, int64_t nseg, int32_t H,
                                                      const float* __restrict__ zcol, int mode, double* __restrict__ part) {
    // 16 lanes per segment (a segment holds a few dozen entries): 16 segments per workgroup
    const int sub = threadIdx.x & 15;
    const int64_t seg = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = seg < nseg;
    double a = 0.0, b = 0.0;
    if (live) {
        const double z = (double)zcol[(int)(seg % H)];
        const int64_t lo = cp[seg], hi = cp[seg + 1];
        // four loads in flight per lane; the additions stay in the order of the plain loop
        for (int64_t t = lo + sub; t < hi; t += 64) {
            float xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = t + 16 * u < hi ? x[t + 16 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t + 16 * u < hi) {
                    a += (double)xv[u] - z;
                    if (mode) { const float sq = xv[u] * xv[u]; b += (double)sq; }
                }
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        if (mode) b += __shfl_xor(b, o, 64);
    }
    if (live && sub == 0) {
        part[seg] = a;
        if (mode) part[nseg + seg] = b;
    }
}

__global__ void k_col_reduce(const double* __restrict__ part_o, int P_o, int64_t nseg_o, const double* __restrict__ part_s, int P_s,
                             int64_t nseg_s, int32_t H, int64_t M, int mode, double* __restrict__ out, const double* __restrict__ extra = nullptr,
                             int nextra = 0, const double* __restrict__ escale = nullptr) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= H) return;
    double a = 0.0, b = 0.0;
    for (int p = 0; p < P_o; ++p) {
        a += part_o[(int64_t)p * H + j];
        if (mode) b += part_o[nseg_o + (int64_t)p * H + j];
    }
    for (int p = 0; p < P_s; ++p) {
        a += part_s[(int64_t)p * H + j];
        if (mode) b += part_s[nseg_s + (int64_t)p * H + j];
    }
    for (int p = 0; p < nextra; ++p) a += (escale ? escale[j] : 1.0) * extra[(int64_t)p * H + j];     // (bit-plane mode: the entries equal to 1, one block per chunk of the rows)
    if (mode == 0) out[j] = a / (double)M;
    else { out[2 * j] = a; out[2 * j + 1] = b; }
}

// column sums over the given pair of (panel, column)-ordered mirrors (+ `nextra` blocks of H ready-made sums)
static int col_sums_of(ddx_ctx* ctx, const int64_t* cp_o, const float* x_o, const int64_t* cp_s, const float* x_s, int mode, double* out,
                       const double* extra = nullptr, int nextra = 0, const double* escale = nullptr) {
    const int32_t H = ctx->H;
    const int64_t nseg_o = (int64_t)ctx->P_o * H, nseg_s = (int64_t)ctx->P_s * H;
    DDX_TRY(ensure(ctx, ctx->col_part, sizeof(double) * 2 * (size_t)(nseg_o + nseg_s + 2)));
    double* part_o = ctx->col_part.as<double>();
    double* part_s = part_o + 2 * nseg_o;
    if (nseg_o) k_col_partials<<<(unsigned)ceil_div(nseg_o, 16), 256, 0, ctx->stream>>>(cp_o, x_o, nseg_o, H, ctx->zcol.as<float>(), mode, part_o);
    if (nseg_s) k_col_partials<<<(unsigned)ceil_div(nseg_s, 16), 256, 0, ctx->stream>>>(cp_s, x_s, nseg_s, H, ctx->zcol.as<float>(), mode, part_s);
    k_col_reduce<<<(unsigned)ceil_div(H, 256), 256, 0, ctx->stream>>>(part_o, ctx->P_o, nseg_o, part_s, ctx->P_s, nseg_s, H, ctx->M, mode, out, extra, nextra, escale);
    return DDX_OK;
}

static int col_sums(ddx_ctx* ctx, int mode, double* out) {
    return col_sums_of(ctx, ctx->csc_o_colptr.as<int64_t>(), ctx->csc_o_x.as<float>(), ctx->csc_s_colptr.as<int64_t>(), ctx->csc_s_x.as<float>(), mode, out);
}

// values of a (panel, column)-ordered mirror (rows, raw counts) from this iteration's normalisation: x[t] = lognorm(raw[t], row)
static int lognorm_mirror(ddx_ctx* ctx, const int32_t* rows, const float* raw, const int64_t* colptr, int32_t npanels, int32_t panel0, float* x) {
    if (npanels <= 0) return DDX_OK;
    const int use_log1p = (ctx->pseudocount == 1.0f);
    const size_t lds = sizeof(float) * (size_t)ctx->panel_rows * kLognormTab;
    if (lds <= 64 * 1024) {
        DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_lognorm_csc), (int)lds));
        k_lognorm_csc<<<(unsigned)(npanels * kLognormParts), 512, lds, ctx->stream>>>(rows, raw, colptr, ctx->H, panel0, ctx->panel_rows, ctx->M, ctx->lib64.as<double>(),
                                                                                     ctx->median.as<float>(), ctx->lognorm_tab.as<float>(), ctx->pseudocount, use_log1p, x);
    } else {
        k_lognorm_csc_direct<<<2048, 256, 0, ctx->stream>>>(rows, raw, colptr, npanels * ctx->H, ctx->lib64.as<double>(), ctx->median.as<float>(), ctx->pseudocount,
                                                            use_log1p, x);
    }
    return DDX_OK;
}

// The full column-major mirror of this iteration's matrix -- the synthetic rows' part and the values of both parts.  The bit-plane
// route (k_bitplane.hip) does without it (its reduced mirrors hold a tenth of the entries), so ddx_lognormalise leaves it out when
// that route is expected; whoever needs it after all (ddx_scale, the plain sparse products, ddx_operator_apply) asks here.
int ensure_full_mirror(ddx_ctx* ctx) {
    if (ctx->mirror_full) return DDX_OK;
    if (!ctx->have_lognorm) return set_err(ctx, DDX_E_ARG, "the column-major mirror needs ddx_lognormalise first");
    DDX_TRY(ensure_full_rows(ctx));
    DDX_TRY(originals_mirror(ctx));
    const int64_t N = ctx->N, M = ctx->M;
    const MirrorSrc full{ctx->aug_indptr.as<int64_t>(), ctx->aug_indices.as<int32_t>(), ctx->aug_raw.as<float>()};
    DDX_TRY(build_csc(ctx, full, ctx->nnz, ctx->nnz_aug - ctx->nnz, N, M, ctx->p_s0, ctx->P_s > 0 ? ctx->P_s : 1, ctx->csc_s_colptr, ctx->csc_s_row, ctx->csc_s_raw));
    {
        ScopedTimer t(ctx, "lognorm_cols");
        DDX_TRY(lognorm_mirror(ctx, ctx->csc_o_row.as<int32_t>(), ctx->csc_o_raw.as<float>(), ctx->csc_o_colptr.as<int64_t>(), ctx->P_o, 0, ctx->csc_o_x.as<float>()));
        if (ctx->P_s > 0)
            DDX_TRY(lognorm_mirror(ctx, ctx->csc_s_row.as<int32_t>(), ctx->csc_s_raw.as<float>(), ctx->csc_s_colptr.as<int64_t>(), ctx->P_s, ctx->p_s0, ctx->csc_s_x.as<float>()));
        else if (ctx->panel_rows * kLognormTab * sizeof(float) > 64 * 1024)        // (the direct kernel walks colptr[nkeys]: one empty panel)
            DDX_TRY(lognorm_mirror(ctx, ctx->csc_s_row.as<int32_t>(), ctx->csc_s_raw.as<float>(), ctx->csc_s_colptr.as<int64_t>(), 1, ctx->p_s0, ctx->csc_s_x.as<float>()));
    }
    if (ctx->scaled) DDX_TRY(scale_full_mirror(ctx));       // (scaled on the bit-plane structures before anybody needed the full mirror)
    DDX_HIP(ctx, hipGetLastError());
    ctx->mirror_full = true;
    return DDX_OK;
}

// bit-plane mode, once per iteration (called by bp_refresh): the reduced mirror of the synthetic rows, built straight from their
// reduced CSR with the log-normalised values as the payload, and the values of the original rows' reduced mirror
int bp_reduced_mirrors(ddx_ctx* ctx) {
    BitPlanes& bp = ctx->bp;
    const int32_t H = ctx->H;
    const int32_t np = ctx->P_s > 0 ? ctx->P_s : 1;
    DDX_TRY(ensure(ctx, ctx->bp_ms_row, sizeof(int32_t) * (size_t)(bp.nrest_s + 64)));
    DDX_TRY(ensure(ctx, ctx->bp_ms_x, sizeof(float) * (size_t)(bp.nrest_s + 64)));
    const MirrorSrc red{bp.rest_indptr, bp.rest_cols, bp.rest_x};
    DDX_TRY(build_csc(ctx, red, bp.nrest_o, bp.nrest_s, ctx->N, ctx->M, ctx->p_s0, np, ctx->bp_ms_colptr, ctx->bp_ms_row, ctx->bp_ms_x));
    bp.restm_s_colptr = ctx->bp_ms_colptr.as<int64_t>();
    bp.restm_s_row = ctx->bp_ms_row.as<int32_t>();
    bp.restm_s_x = ctx->bp_ms_x.as<float>();
    (void)H;
    return lognorm_mirror(ctx, bp.restm_row, bp.restm_raw, bp.restm_colptr, ctx->P_o, 0, bp.restm_x);
}

// bit-plane mode: colmean = (sums of the entries equal to 1, `nparts` blocks from the matrix cores, + sums over the reduced mirrors) / M
int bp_colmean(ddx_ctx* ctx, const double* parts, int nparts) {
    BitPlanes& bp = ctx->bp;
    ScopedTimer t(ctx, "col_sums");
    return col_sums_of(ctx, bp.restm_colptr, bp.restm_x, bp.restm_s_colptr, bp.restm_s_x, 0, ctx->colmean.as<double>(), parts, nparts);
}

// values of every stored entry of the row-major arrays (needs the iteration's table, median and library sizes)
static int lognorm_rows(ddx_ctx* ctx) {
    const int rows_per_wave = 8;          // the queue of rare entries fills over several rows
    k_lognorm_rows<<<(unsigned)ceil_div(ctx->M, 4 * rows_per_wave), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_raw.as<float>(), ctx->lib64.as<double>(),
                                                                                          ctx->median.as<float>(), ctx->lognorm_tab.as<float>(), ctx->pseudocount,
                                                                                          ctx->pseudocount == 1.0f, ctx->M, rows_per_wave, ctx->aug_x.as<float>());
    ctx->rows_x = true;
    ctx->rows_scaled = false;
    return DDX_OK;
}

int stage_lognormalise(ddx_ctx* ctx, float pseudocount) {
    const int64_t N = ctx->N, S = ctx->S, M = ctx->M;
    const int32_t H = ctx->H;
    const bool lean = bp_wanted_at_upload(ctx) && ctx->bp.ready && S <= N / 2;       // the bit-plane route is expected to serve this matrix
    const bool derived = lean && !ctx->synth_rows && bp_lean(ctx);                   // ... and the doublets exist as parents only
    if (derived) {
        DDX_TRY(bp_synth_libs(ctx));          // lib[p0] + lib[p1]: exact, the counts being small integers
    } else {
        if (!ctx->synth_rows) DDX_TRY(materialise_synthetic(ctx));
        // exact number of synthetic entries (one 8-byte read-back per iteration; sizes the sorts)
        int64_t nnz_aug = 0;
        DDX_HIP(ctx, hipMemcpyAsync(&nnz_aug, ctx->aug_indptr.as<int64_t>() + M, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        DDX_HIP(ctx, wait_stream(ctx));
        ctx->nnz_aug = nnz_aug;
        if (S) {
            ScopedTimer t(ctx, "row_sums");
            if (ctx->counts_exact)
                k_row_sums_exact<<<(unsigned)ceil_div(S, 4), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_raw.as<float>(), N, S,
                                                                                    ctx->lib32.as<float>(), ctx->lib64.as<double>());
            else
                k_row_sums<<<(unsigned)ceil_div(S, 4), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_raw.as<float>(), N, S,
                                                                              ctx->lib32.as<float>(), ctx->lib64.as<double>());
        }
    }
    // median of the augmented library sizes
    DDX_TRY(ensure(ctx, ctx->lib_sorted, sizeof(float) * M));
    DDX_TRY(ensure(ctx, ctx->median, 256));
    {
        size_t tmp_bytes = 0;
        DDX_HIP(ctx, prim::sort_keys(nullptr, tmp_bytes, ctx->lib32.as<float>(), ctx->lib_sorted.as<float>(), (int)M, 0, 32, ctx->stream));
        DDX_TRY(ensure(ctx, ctx->sort_tmp, tmp_bytes));
        ScopedTimer t(ctx, "median");
        DDX_HIP(ctx, prim::sort_keys(ctx->sort_tmp.p, tmp_bytes, ctx->lib32.as<float>(), ctx->lib_sorted.as<float>(), (int)M, 0, 32, ctx->stream));
        k_median_from_sorted<<<1, 64, 0, ctx->stream>>>(ctx->lib_sorted.as<float>(), M, ctx->median.as<float>());
    }
    ctx->p_s0 = (int32_t)(N / ctx->panel_rows);
    ctx->P_s = S ? (int32_t)((M - 1) / ctx->panel_rows) - ctx->p_s0 + 1 : 0;
    const int use_log1p = (pseudocount == 1.0f);
    DDX_TRY(ensure(ctx, ctx->lognorm_tab, sizeof(float) * (size_t)M * kLognormTab));
    ctx->pseudocount = pseudocount;
    {
        ScopedTimer t(ctx, "lognorm_rows");
        k_lognorm_table<<<(unsigned)ceil_div(M * kLognormTab, 256), 256, 0, ctx->stream>>>(ctx->lib64.as<double>(), ctx->median.as<float>(), pseudocount,
                                                                                          use_log1p, M, ctx->lognorm_tab.as<float>());
        ctx->rows_x = false;
        if (!derived) DDX_TRY(lognorm_rows(ctx));       // (derived: the bit-plane structures take their values from the table; aug_x on demand)
    }
    DDX_TRY(ensure(ctx, ctx->zcol, sizeof(float) * H));
    DDX_TRY(ensure(ctx, ctx->colmean, sizeof(double) * H));
    const float z = use_log1p ? 0.f : (float)std::log((double)pseudocount);
    k_fill_f32<<<(unsigned)ceil_div(H, 256), 256, 0, ctx->stream>>>(ctx->zcol.as<float>(), H, z);
    ctx->zvalue = z;
    ctx->pseudocount = pseudocount;
    ctx->bp.values = false;              // the bit-plane structures hold the last iteration's values
    ctx->mirror_full = false;
    ctx->have_lognorm = true;
    ctx->scaled = false;
    ctx->have_emb = ctx->have_knn = false;
    if (lean) {
        // the bit-plane route is expected to serve this matrix: its structures (a tenth of the entries in sparse form) are all the
        // products need, and they also give the column means -- the full mirror is only built if somebody asks (ensure_full_mirror)
        DDX_TRY(bp_refresh(ctx));
    } else {
        DDX_TRY(ensure_full_mirror(ctx));
        ScopedTimer t(ctx, "col_sums");
        DDX_TRY(col_sums(ctx, 0, ctx->colmean.as<double>()));
    }
    DDX_HIP(ctx, hipGetLastError());
    return DDX_OK;
}

// ------------------------------------------------------------------------------------------------
// standard scaling (restated scanpy pp.scale; see oracle/dd_oracle.py:scale_like_scanpy)
// ------------------------------------------------------------------------------------------------
// per column: mean (f64), unbiased variance from the mean of float32-rounded squares, std==0 -> 1
__global__ void k_scale_stats(const double* __restrict__ stat, const int64_t* __restrict__ cp_o, int P_o,
                              const int64_t* __restrict__ cp_s, int P_s, const float* __restrict__ zcol, int64_t M, int32_t H,
                              double* __restrict__ mean_out, double* __restrict__ std_out) {
#pragma clang fp contract(off)
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= H) return;
    const double z = (double)zcol[j];
    int64_t stored = 0;
    for (int p = 0; p < P_o; ++p) stored += cp_o[(int64_t)p * H + j + 1] - cp_o[(int64_t)p * H + j];
    for (int p = 0; p < P_s; ++p) stored += cp_s[(int64_t)p * H + j + 1] - cp_s[(int64_t)p * H + j];
    const double cnt = (double)stored;
    const float zsq = zcol[j] * zcol[j];
    const double zz = (double)zsq;
    const double mean = z + stat[2 * j] / (double)M;
    const double mean_sq = (stat[2 * j + 1] + ((double)M - cnt) * zz) / (double)M;
    double var = mean_sq - mean * mean;
    if (M != 1) var *= (double)M / (double)(M - 1);
    double sd = (var > 0.0) ? sqrt(var) : 0.0;  // (var < 0 only by rounding of a constant column)
    if (sd == 0.0) sd = 1.0;
    mean_out[j] = mean;
    std_out[j] = sd;
}

__device__ __forceinline__ float scale_value(float x, double mean, double sd, float maxv) {
#pragma clang fp contract(off)
    const float r1 = (float)((double)x - mean);
    float r2 = (float)((double)r1 / sd);
    if (maxv > 0.f) r2 = fminf(fmaxf(r2, -maxv), maxv);
    return r2;
}

__global__ void k_scale_rows(const int32_t* __restrict__ cols, int64_t n_ptr_index, const int64_t* __restrict__ indptr,
                             const double* __restrict__ mean, const double* __restrict__ sd, float maxv,
                             float* __restrict__ x) {
    const int64_t n = indptr[n_ptr_index];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = cols[t];
        x[t] = scale_value(x[t], mean[j], sd[j], maxv);
    }
}

__global__ void __launch_bounds__(256) k_scale_cols(const int64_t* __restrict__ cp, int P, int32_t H,
                                                    const double* __restrict__ mean, const double* __restrict__ sd,
                                                    float maxv, float* __restrict__ x) {
    const int j = blockIdx.x;
    const double m = mean[j], s = sd[j];
    for (int p = 0; p < P; ++p)
        for (int64_t t = cp[(int64_t)p * H + j] + threadIdx.x; t < cp[(int64_t)p * H + j + 1]; t += 256)
            x[t] = scale_value(x[t], m, s, maxv);
}

__global__ void k_scale_zcol(const double* __restrict__ mean, const double* __restrict__ sd, float maxv, int32_t H,
                             float* __restrict__ zcol) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < H) zcol[j] = scale_value(zcol[j], mean[j], sd[j], maxv);
}

// the scaling in force applied to the full arrays that were (re)built after it (ensure_full_rows / ensure_full_mirror)
static int scale_full_rows(ddx_ctx* ctx) {
    const int32_t H = ctx->H;
    const double* mean = ctx->colstat.as<double>() + 2 * H;
    const double* sd = ctx->colstat.as<double>() + 3 * H;
    ScopedTimer t(ctx, "scale");
    k_scale_rows<<<2048, 256, 0, ctx->stream>>>(ctx->aug_indices.as<int32_t>(), ctx->M, ctx->aug_indptr.as<int64_t>(), mean, sd, ctx->scale_max, ctx->aug_x.as<float>());
    ctx->rows_scaled = true;
    return DDX_OK;
}

static int scale_full_mirror(ddx_ctx* ctx) {
    const int32_t H = ctx->H;
    const double* mean = ctx->colstat.as<double>() + 2 * H;
    const double* sd = ctx->colstat.as<double>() + 3 * H;
    ScopedTimer t(ctx, "scale");
    k_scale_cols<<<(unsigned)H, 256, 0, ctx->stream>>>(ctx->csc_o_colptr.as<int64_t>(), ctx->P_o, H, mean, sd, ctx->scale_max, ctx->csc_o_x.as<float>());
    k_scale_cols<<<(unsigned)H, 256, 0, ctx->stream>>>(ctx->csc_s_colptr.as<int64_t>(), ctx->P_s, H, mean, sd, ctx->scale_max, ctx->csc_s_x.as<float>());
    return DDX_OK;
}

int stage_scale(ddx_ctx* ctx, float max_value) {
    const int32_t H = ctx->H;
    const int64_t M = ctx->M;
    DDX_TRY(ensure(ctx, ctx->colstat, sizeof(double) * 8 * H + 256));
    ctx->scale_max = max_value;
    // the bit-plane structures hold this iteration's matrix (ddx_lognormalise took the lean route): scale THEM -- a tenth of the entries in
    // sparse form, the rest through 1 / sd in the matrix-core products -- and leave the full arrays to whoever asks (ensure_full_*)
    if (ctx->bp.ready && ctx->bp.values && !ctx->mirror_full) return bp_scale(ctx, max_value);
    DDX_TRY(ensure_full_mirror(ctx));
    double* stat = ctx->colstat.as<double>();
    double* mean = stat + 2 * H;
    double* sd = stat + 3 * H;
    ScopedTimer t(ctx, "scale");
    DDX_TRY(col_sums(ctx, 1, stat));
    k_scale_stats<<<(unsigned)ceil_div(H, 256), 256, 0, ctx->stream>>>(stat, ctx->csc_o_colptr.as<int64_t>(), ctx->P_o, ctx->csc_s_colptr.as<int64_t>(),
                                                                       ctx->P_s, ctx->zcol.as<float>(), M, H, mean, sd);
    k_scale_rows<<<2048, 256, 0, ctx->stream>>>(ctx->aug_indices.as<int32_t>(), M, ctx->aug_indptr.as<int64_t>(), mean, sd, max_value,
                                                ctx->aug_x.as<float>());
    ctx->rows_scaled = true;
    k_scale_cols<<<(unsigned)H, 256, 0, ctx->stream>>>(ctx->csc_o_colptr.as<int64_t>(), ctx->P_o, H, mean, sd, max_value, ctx->csc_o_x.as<float>());
    k_scale_cols<<<(unsigned)H, 256, 0, ctx->stream>>>(ctx->csc_s_colptr.as<int64_t>(), ctx->P_s, H, mean, sd, max_value, ctx->csc_s_x.as<float>());
    k_scale_zcol<<<(unsigned)ceil_div(H, 256), 256, 0, ctx->stream>>>(mean, sd, max_value, H, ctx->zcol.as<float>());
    DDX_TRY(col_sums(ctx, 0, ctx->colmean.as<double>()));
    DDX_HIP(ctx, hipGetLastError());
    ctx->scaled = true;
    ctx->bp.values = false;                          // (the bit-plane structures, if any, describe the unscaled matrix)
    ctx->have_emb = ctx->have_knn = false;
    return DDX_OK;
}

// ---- sc.pp.scale on the bit-plane structures (dd.py:302-303) -----------------------------------------------------------------
// Statistics of column j from three sources: the reduced mirrors (sums of x - z and of float32 squares, counts from the segment
// pointers) and the bitmap's entries (B^T [s | x(1)^2 | 1] from the matrix cores).  Same formulas as k_scale_stats.  Besides mean and sd:
// inv_sd, bsum = (B^T s)_j (the bitmap's share of the new column mean is bsum / sd), and the clip test of the column's entries equal to 1:
// unclipped they become s_i / sd_j + zr_j with zr_j = (z - mean_j) / sd_j, s_i > 0; none of them reaches +-maxv when zr_j >= -maxv
// and smax / sd_j + zr_j <= maxv (smax: the largest s of any row -- conservative).  flags[0] counts the columns still in the bitmaps that
// fail the test (the structures must be rebuilt with them demoted), flags[1] those that fail it with the safety margin `tight` < 1
// (chosen at the first scaling of a fit); want[j] = demoted already, or failing with the margin.
__global__ void k_bp_scale_stats(const double* __restrict__ stat, const int64_t* __restrict__ cp_o, int P_o, const int64_t* __restrict__ cp_s, int P_s,
                                 const double* __restrict__ parts, int nparts, const float* __restrict__ zcol, int64_t M, int32_t H, float maxv, double tight,
                                 const double* __restrict__ smax_p, const uint8_t* __restrict__ demoted, double* __restrict__ mean_out, double* __restrict__ sd_out,
                                 double* __restrict__ inv_out, double* __restrict__ bsum_out, uint8_t* __restrict__ want, int* __restrict__ flags) {
#pragma clang fp contract(off)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= H) return;
    const double z = (double)zcol[j];
    int64_t stored = 0;
    for (int p = 0; p < P_o; ++p) stored += cp_o[(int64_t)p * H + j + 1] - cp_o[(int64_t)p * H + j];
    for (int p = 0; p < P_s; ++p) stored += cp_s[(int64_t)p * H + j + 1] - cp_s[(int64_t)p * H + j];
    double bs = 0.0, bq = 0.0, bc = 0.0;
    for (int p = 0; p < nparts; ++p) {
        const double* b = parts + ((int64_t)p * H + j) * 3;
        bs += b[0]; bq += b[1]; bc += b[2];
    }
    const double cnt = (double)stored + rint(bc);
    const float zsq = zcol[j] * zcol[j];
    const double zz = (double)zsq;
    const double mean = z + (stat[2 * j] + bs) / (double)M;
    const double mean_sq = ((stat[2 * j + 1] + bq) + ((double)M - cnt) * zz) / (double)M;
    double var = mean_sq - mean * mean;
    if (M != 1) var *= (double)M / (double)(M - 1);
    double sd = (var > 0.0) ? sqrt(var) : 0.0;
    if (sd == 0.0) sd = 1.0;
    mean_out[j] = mean;
    sd_out[j] = sd;
    bsum_out[j] = bs;
    uint8_t w = demoted[j];
    // The factor the matrix-core products apply to column j's entries equal to 1.  What the dense matrix holds for such an entry is the
    // float32 x'_ij against the float32 background z'_j, i.e. L_ij = s_i / sd_j + c_j with c_j = (z - mean_j) / sd_j - z'_j, the rounding
    // of the background -- the same for every entry of the column, and a perturbation that is coherent along a column moves the
    // principal components far more than the entries' individual roundings do.  It is folded into the factor through the column's
    // mean s (c_j s_i / sbar_j: exact in the column's sum, what is left has zero mean over the column).  A demoted column has no bits:
    // factor 0, so that its (large) 1 / sd does not take digits away from the others.
    const double zr = (z - mean) / sd;
    double inv = 0.0;
    if (!w) {
        inv = 1.0 / sd;
        if (bs > 0.0 && bc >= 0.5) {
            const float zs = scale_value(zcol[j], mean, sd, maxv);
            const double c = zr - (double)zs;
            if (fabs(c) <= 1e-5 * (fabs(zr) + 1.0)) inv += c * rint(bc) / bs;      // (a clipped background is no rounding: that column is about to be demoted)
        }
    }
    inv_out[j] = inv;
    if (maxv > 0.f && !w) {
        const double top = smax_p[0] / sd + zr, lim = (double)maxv;
        if (zr < -lim || top > lim) atomicAdd(&flags[0], 1);
        if (zr < -lim * tight || top > lim * tight) { atomicAdd(&flags[1], 1); w = 1; }
    }
    want[j] = w;
}

__global__ void k_bp_scale_rest(const int32_t* __restrict__ cols, int64_t n, const double* __restrict__ mean, const double* __restrict__ sd, float maxv,
                                float* __restrict__ x) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = cols[t];
        x[t] = scale_value(x[t], mean[j], sd[j], maxv);
    }
}

constexpr double kBpClipMargin = 0.9;      // a column is demoted at the first scaling of a fit when an entry equal to 1 could come within 10 % of the clip

int bp_scale(ddx_ctx* ctx, float max_value) {
    BitPlanes& bp = ctx->bp;
    const int32_t H = ctx->H;
    const int64_t M = ctx->M;
    double* stat = ctx->colstat.as<double>();
    double* mean = stat + 2 * H;
    double* sd = stat + 3 * H;
    double* inv = stat + 4 * H;
    double* bsum = stat + 5 * H;
    uint8_t* want = reinterpret_cast<uint8_t*>(stat + 6 * H);
    int* flags = reinterpret_cast<int*>(stat + 7 * H);
    for (int attempt = 0;; ++attempt) {
        const double* parts = nullptr;
        int nparts = 0;
        DDX_TRY(bp_scale_sums(ctx, &parts, &nparts));
        int hflags[2] = {0, 0};
        {
            ScopedTimer t(ctx, "scale");
            DDX_TRY(col_sums_of(ctx, bp.restm_colptr, bp.restm_x, bp.restm_s_colptr, bp.restm_s_x, 1, stat));
            DDX_HIP(ctx, hipMemsetAsync(flags, 0, 2 * sizeof(int), ctx->stream));
            k_bp_scale_stats<<<(unsigned)ceil_div(H, 256), 256, 0, ctx->stream>>>(stat, bp.restm_colptr, ctx->P_o, bp.restm_s_colptr, ctx->P_s, parts, nparts, ctx->zcol.as<float>(), M, H,
                                                                                  max_value, kBpClipMargin, bp.cmax + 128, ctx->bp_demote.as<uint8_t>(), mean, sd, inv, bsum, want, flags);
            DDX_HIP(ctx, hipMemcpyAsync(hflags, flags, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            DDX_HIP(ctx, wait_stream(ctx));
        }
        const bool choose = !bp.demote_decided && hflags[1] > 0;
        if (hflags[0] == 0 && !choose) break;
        if (attempt) return set_err(ctx, DDX_E_NUMERIC, "standard scaling: %d columns still reach the clip after their demotion", hflags[0]);
        // columns whose entries equal to 1 reach the clip (or, at the first scaling of the fit, come close): out of the bitmaps for the
        // rest of the fit, then the statistics again on the rebuilt structures (the sums themselves do not depend on the split)
        std::vector<uint8_t> hwant((size_t)H);
        DDX_HIP(ctx, hipMemcpyAsync(hwant.data(), want, (size_t)H, hipMemcpyDeviceToHost, ctx->stream));
        DDX_HIP(ctx, wait_stream(ctx));
        DDX_TRY(bp_rebuild_demoted(ctx, hwant));
    }
    bp.demote_decided = true;
    {
        ScopedTimer t(ctx, "scale");
        const int64_t n_rest = bp.nrest_o + bp.nrest_s;
        if (n_rest > 0) k_bp_scale_rest<<<(unsigned)std::min<int64_t>(4096, ceil_div(n_rest, 256)), 256, 0, ctx->stream>>>(bp.rest_cols, n_rest, mean, sd, max_value, bp.rest_x);
        k_scale_cols<<<(unsigned)H, 256, 0, ctx->stream>>>(bp.restm_colptr, ctx->P_o, H, mean, sd, max_value, bp.restm_x);
        if (ctx->P_s > 0) k_scale_cols<<<(unsigned)H, 256, 0, ctx->stream>>>(bp.restm_s_colptr, ctx->P_s, H, mean, sd, max_value, bp.restm_s_x);
        k_scale_zcol<<<(unsigned)ceil_div(H, 256), 256, 0, ctx->stream>>>(mean, sd, max_value, H, ctx->zcol.as<float>());
        // new column means: (bitmap share (B^T s)_j / sd_j + sums of x' - z'_j over the reduced mirrors) / M
        DDX_TRY(col_sums_of(ctx, bp.restm_colptr, bp.restm_x, bp.restm_s_colptr, bp.restm_s_x, 0, ctx->colmean.as<double>(), bsum, 1, inv));
    }
    DDX_HIP(ctx, hipGetLastError());
    bp.scaled = true;
    bp.inv_sd = inv;
    ctx->pk_valid[0] = ctx->pk_valid[1] = false;
    ctx->scaled = true;
    ctx->rows_scaled = false;
    ctx->have_emb = ctx->have_knn = false;
    return DDX_OK;
}

// ------------------------------------------------------------------------------------------------
// test helper: densify rows of the matrix handed to PCA
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dense_rows(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                    const float* __restrict__ x, const float* __restrict__ zcol, int32_t H,
                                                    int64_t row0, float* __restrict__ out) {
    const int64_t r = blockIdx.x;
    float* o = out + r * (int64_t)H;
    for (int j = threadIdx.x; j < H; j += 256) o[j] = zcol[j];
    __syncthreads();
    const int64_t b = indptr[row0 + r], e = indptr[row0 + r + 1];
    for (int64_t p = b + threadIdx.x; p < e; p += 256) o[cols[p]] = x[p];
}

int stage_dense_rows(ddx_ctx* ctx, int64_t row0, int64_t nrows, float* out_host) {
    DevBuf tmp;
    DDX_TRY(ensure(ctx, tmp, sizeof(float) * nrows * ctx->H));
    k_dense_rows<<<(unsigned)nrows, 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_indices.as<int32_t>(), ctx->aug_x.as<float>(),
                                                           ctx->zcol.as<float>(), ctx->H, row0, tmp.as<float>());
    hipError_t e = hipMemcpyAsync(out_host, tmp.p, sizeof(float) * nrows * ctx->H, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = wait_stream(ctx);
    release(ctx, tmp);
    if (e != hipSuccess) return set_err(ctx, DDX_E_HIP, "dense rows copy failed: %s", hipGetErrorString(e));
    return DDX_OK;
}

}  // namespace ddx
