// Bit-plane operator products (sc.tl.pca call site, dd.py:305-314; the two sparse products of every power iteration).
//
// Nine in ten stored entries of the count matrix are a 1, and the value the PCA sees for such an entry depends on its ROW
// only (x = log(1 / lib_i * median + pseudocount), dd.py:286-297).  So the operator splits:
//
//      L  =  diag(s) B  +  R ,      s_i = x_i(1) - z ,   B = [count == 1] (zeros and ones),   R = the entries with other counts
//
// R keeps going through the LDS-staged sparse kernels (k_pca.hip).  B is kept as a BITMAP -- one bit per (row, column), 16
// times less memory traffic than the 8 bytes per entry of the sparse form -- and multiplied on the matrix cores in exact
// integer arithmetic: the float64 operand is cut column by column into a 31-bit fixed-point number, that number into four
// signed 8-bit digits, and  B . digit_d  runs on v_mfma_i32_16x16x64_i8 (zeros and ones against 8-bit digits, 32-bit sums:
// no rounding anywhere, any order gives the same bits).  The four digit sums are recombined in 64-bit integers, scaled
// back and by s_i in float64.  Against the float32 operand copy of the sparse path (24 bits relative to each element) the
// fixed point carries 31 bits relative to the column's largest element.
//
// Only the original cells' rows take this route (their pattern is fixed for a fit: the bitmaps, the reduced sparse
// structures and their positions in the full arrays are built once per fit and context); the synthetic doublets, new in
// every iteration, stay sparse.  Scaled matrices (standard_scaling: the value then depends on the column too) and
// sketches wider than 64 columns keep the plain sparse products.
#include "ddx_prims.h"

#include "ddx_internal.h"

namespace ddx {

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kBpDigits = 4;

// ---- once per fit: bitmaps and reduced structures ---------------------------------------------------------------
// bitmap of the rows: word (tile, kb, r) holds columns kb*64 .. +63 of row tile*16 + r; layout [(tile * KB + kb) * 16 + r]
__global__ void k_bp_rows_bitmap(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols, const float* __restrict__ raw, int64_t N, int KB,
                                 uint64_t* __restrict__ bm) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ntile = (N + 15) >> 4;
    if (w >= ntile * KB * 16) return;
    const int r = (int)(w & 15);
    const int64_t tk = w >> 4;
    const int64_t tile = tk / KB;
    const int kb = (int)(tk - tile * KB);
    const int64_t row = tile * 16 + r;
    uint64_t v = 0;
    if (row < N) {
        const int64_t b = indptr[row], e = indptr[row + 1];
        const int32_t c0 = kb * 64;
        int64_t lo = b, hi = e;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (cols[mid] < c0) lo = mid + 1; else hi = mid; }
        for (int64_t p = lo; p < e && cols[p] < c0 + 64; ++p)
            if (raw[p] == 1.0f) v |= 1ull << (cols[p] - c0);
    }
    bm[w] = v;
}

// the same bits by columns: word (ctile, kbr, c) holds rows kbr*64 .. +63 of column ctile*16 + c.  One wave per 64 x 64 block.
__global__ void __launch_bounds__(256) k_bp_transpose(const uint64_t* __restrict__ bm, int64_t N, int32_t H, int KB, int64_t KBr,
                                                      uint64_t* __restrict__ bmT) {
    const int lane = threadIdx.x & 63;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= KBr * KB) return;
    const int64_t rb = blk / KB;
    const int kb = (int)(blk - rb * KB);
    const int64_t row = rb * 64 + lane;
    const uint64_t w = row < N ? bm[((row >> 4) * KB + kb) * 16 + (row & 15)] : 0ull;
    uint64_t mine = 0;
    for (int c = 0; c < 64; ++c) {
        const uint64_t t = __ballot((w >> c) & 1ull);
        if (lane == c) mine = t;
    }
    const int64_t col = (int64_t)kb * 64 + lane;
    if (col < (((int64_t)H + 15) & ~(int64_t)15)) bmT[((col >> 4) * KBr + rb) * 16 + (col & 15)] = mine;
}

__global__ void k_bp_flags(const float* __restrict__ raw, int64_t n, int32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    flag[i] = (i < n && raw[i] != 1.0f) ? 1 : 0;            // (element n: 0, so that the scan's last element is the total)
}

__global__ void k_bp_compact(const int32_t* __restrict__ flag, const int32_t* __restrict__ scan, const int32_t* __restrict__ idx, int64_t n,
                             int32_t* __restrict__ idx_out, int32_t* __restrict__ pos_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    idx_out[scan[i]] = idx[i];
    pos_out[scan[i]] = (int32_t)i;
}

__global__ void k_bp_pointers(const int64_t* __restrict__ ptr, int64_t nptr, const int32_t* __restrict__ scan, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nptr) return;
    out[i] = scan[ptr[i]];
}

// ---- once per iteration: values of the reduced structures, row scales ---------------------------------------------
__global__ void k_bp_gather(const float* __restrict__ x, const int32_t* __restrict__ pos, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[pos[i]];
}

// s_i = x_i(1) - z, the float32 difference the sparse path forms for such an entry (exactly what the float64 difference rounds to)
__global__ void k_bp_row_scale(const float* __restrict__ tab, int tab_stride, float z, int64_t N, double* __restrict__ s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) s[i] = (double)(tab[i * tab_stride] - z);
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

// shift of column c: |X 2^sh| <= 2^30
__device__ __forceinline__ int bp_shift(double cmax) {
    if (!(cmax > 0.0) || !(cmax < 1e300)) return 0;
    int e;
    (void)frexp(cmax, &e);                                   // cmax = m 2^e, 0.5 <= m < 1
    return 30 - e;
}

// digits in the B-operand layout of v_mfma_i32_16x16x64_i8: qd[((kb * NCB + cb) * 4 + d) * 64 + lane] = 16 bytes =
// digit d of rows k = kb*64 + (lane >> 4)*16 + 0..15, column cb*16 + (lane & 15)
__global__ void __launch_bounds__(256) k_bp_digits(const double* __restrict__ X, const double* __restrict__ wgt, int64_t R, int L, int NCB,
                                                   const double* __restrict__ cmax, int64_t KB, v4i* __restrict__ qd) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= KB * NCB * 64) return;
    const int lane = (int)(t & 63);
    const int64_t u = t >> 6;
    const int cb = (int)(u % NCB);
    const int64_t kb = u / NCB;
    const int col = cb * 16 + (lane & 15);
    const int g = lane >> 4;
    unsigned out[kBpDigits][4] = {};
    if (col < L) {
        const double scale = ldexp(1.0, bp_shift(cmax[col]));
        for (int e = 0; e < 16; ++e) {
            const int64_t k = kb * 64 + g * 16 + e;
            int v = 0;
            if (k < R) {
                const double x = wgt ? wgt[k] * X[k * L + col] : X[k * L + col];
                const double q = rint(x * scale);
                v = (q >= -1073741824.0 && q <= 1073741824.0) ? (int)q : 0;       // (non-finite operands: zero; caught by the rank check upstream)
            }
#pragma unroll
            for (int d = 0; d < kBpDigits; ++d) {
                const int dg = d + 1 < kBpDigits ? ((v + 128) & 255) - 128 : v;    // balanced digits; the last one is what is left (|.| <= 65)
                v = (v - dg) >> 8;
                out[d][e >> 2] |= (unsigned)(uint8_t)(int8_t)dg << (8 * (e & 3));
            }
        }
    }
#pragma unroll
    for (int d = 0; d < kBpDigits; ++d) qd[((kb * NCB + cb) * kBpDigits + d) * 64 + lane] = v4i{(int)out[d][0], (int)out[d][1], (int)out[d][2], (int)out[d][3]};
}

// 16 bits -> 16 bytes of 0 / 1 (the A operand of the MFMA)
__device__ __forceinline__ v4i bp_expand16(unsigned bits) {
    v4i r;
#pragma unroll
    for (int w = 0; w < 4; ++w) r[w] = (int)((((bits >> (4 * w)) & 0xfu) * 0x00204081u) & 0x01010101u);
    return r;
}

// Digit sums S_d = B . digit_d over k-blocks [kb0, kb1) for the tiles of this workgroup (WAVES waves x RT tiles of 16
// bitmap rows), written to part[chunk][row][col][d] (int32) for the combine kernels.
// Both operands of a k-block -- the digit blocks (NCB x 4 KB) and the workgroup's bitmap words (RT * WAVES x 128 bytes) --
// arrive in LDS by asynchronous copies into a ring of stages, stages - 1 k-blocks ahead.  What a CU can take in is about
// 10 bytes per clock, whatever the source (MI355X_MICROARCH.md), so the digit blocks are shared by as many rows as the
// register file allows: 8 waves x 64 rows (252 registers per lane: the 16 x 48 x 4-digit accumulators of four tiles).
// A wave issues a fixed number of copies per k-block (waves of the first half carry the odd piece), so "at most
// (stages - 2) * that many outstanding" means the next k-block has landed (memory operations of a wave complete in
// issue order; the loop issues nothing else).
template <int RT, int WAVES, int NCB>
__global__ void __launch_bounds__(64 * WAVES) k_bp_product(const uint64_t* __restrict__ bm, const v4i* __restrict__ qd, int64_t ntile, int64_t KB, int kb_per_chunk,
                                                           int64_t nrows, int32_t* __restrict__ part) {
    constexpr int kStages = NCB <= 3 ? 4 : 3;                     // (64 KB of static LDS at most)
    constexpr int kDigVecs = NCB * kBpDigits * 64;               // 16-byte vectors of digits per k-block
    constexpr int kBmWords = RT * WAVES * 16;                    // 8-byte bitmap words per k-block
    constexpr int kDigPieces = kDigVecs / 64;                    // one piece = one wave-wide copy (1 KB)
    constexpr int kBmPieces = kBmWords * 2 / 64;                 // (4-byte copies: 256 bytes per wave-wide copy)
    static_assert(kBmPieces % WAVES == 0, "bitmap pieces divide evenly");
    constexpr int kDigLo = kDigPieces / WAVES, kDigExtra = kDigPieces % WAVES;   // waves < kDigExtra carry one more piece
    __shared__ v4i lds_d[kStages][kDigVecs];
    __shared__ uint64_t lds_b[kStages][kBmWords];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tileb = (int64_t)blockIdx.x * (WAVES * RT);    // first tile of the workgroup
    const int chunk = blockIdx.y;
    const int64_t kb0 = (int64_t)chunk * kb_per_chunk, kb1 = kb0 + kb_per_chunk < KB ? kb0 + kb_per_chunk : KB;
    const int r = lane & 15, g = lane >> 4;
    v4i acc[RT][NCB][kBpDigits];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int d = 0; d < kBpDigits; ++d) acc[t][c][d] = v4i{0, 0, 0, 0};
    const uint32_t* bm32 = reinterpret_cast<const uint32_t*>(bm);
    auto stage = [&](int64_t kb, int st) {                       // (a k-block past the end re-reads the last one: harmless, same count)
        const int64_t k = kb < kb1 ? kb : kb1 - 1;
#pragma unroll
        for (int u = 0; u <= kDigLo; ++u) {
            const int piece = u * WAVES + wave;
            if (u < kDigLo || wave < kDigExtra)                   // (wave-uniform)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qd + k * kDigVecs + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(lds_d[st] + piece * 64), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kBmPieces / WAVES; ++u) {
            const int w0 = (u * WAVES + wave) * 64;               // 4-byte items of the stage's bitmap image; word = item / 2
            const int item = w0 + lane;
            int64_t tile = tileb + (item >> 5);                   // 32 items (16 words) per tile
            if (tile >= ntile) tile = ntile - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bm32 + ((tile * KB + k) * 16) * 2 + (item & 31)),
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<uint32_t*>(lds_b[st]) + w0), 4, 0, 0);
        }
    };
    auto wait_next = [&]() {                                     // all but the copies of the newest kStages - 2 k-blocks have landed
        if (wave < kDigExtra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kStages - 2) * (kDigLo + 1 + kBmPieces / WAVES)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kStages - 2) * (kDigLo + kBmPieces / WAVES)) : "memory");
    };
#pragma unroll
    for (int st = 0; st < kStages - 1; ++st) stage(kb0 + st, st);
    wait_next();
    __syncthreads();
    for (int64_t kb = kb0; kb < kb1; ++kb) {
        const int st = (int)((kb - kb0) % kStages);
        stage(kb + kStages - 1, (int)((kb - kb0 + kStages - 1) % kStages));
        v4i a[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const uint64_t w = lds_b[st][(wave * RT + t) * 16 + r];
            a[t] = bp_expand16((unsigned)(w >> (16 * g)) & 0xffffu);
        }
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int d = 0; d < kBpDigits; ++d) {
                const v4i b = lds_d[st][(c * kBpDigits + d) * 64 + lane];
#pragma unroll
                for (int t = 0; t < RT; ++t) acc[t][c][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t], b, acc[t][c][d], 0, 0, 0);
            }
        wait_next();
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the copies issued past the end)
    // C/D layout: column = lane & 15, row = (lane >> 4) * 4 + reg
    const int64_t tile0 = tileb + (int64_t)wave * RT;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        if (tile0 + t >= ntile) continue;
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t row = (tile0 + t) * 16 + g * 4 + q;
                if (row >= nrows) continue;
                v4i o;
#pragma unroll
                for (int d = 0; d < kBpDigits; ++d) o[d] = acc[t][c][d][q];
                *reinterpret_cast<v4i*>(part + (((int64_t)chunk * nrows + row) * (NCB * 16) + c * 16 + r) * kBpDigits) = o;
            }
    }
}

__device__ __forceinline__ double bp_value(const int32_t* __restrict__ part, int chunks, int64_t nrows, int64_t row, int ncols, int col, double cmax) {
    long long V = 0;
    for (int ch = 0; ch < chunks; ++ch) {
        const v4i p = *reinterpret_cast<const v4i*>(part + (((int64_t)ch * nrows + row) * ncols + col) * kBpDigits);
        V += (long long)p[0] + ((long long)p[1] << 8) + ((long long)p[2] << 16) + ((long long)p[3] << 24);     // integers: exact in any order
    }
    return ldexp((double)V, -bp_shift(cmax));
}

// A Q:   Y[i][c] += s_i S[i][c] for the original rows (the sparse kernel left the other entries' sum minus the centring term there);
//        Y32 = the float32 copy [. x ld] the A^T Y pass gathers (zero in the padding columns)
__global__ void k_bp_combine_rows(const int32_t* __restrict__ part, int chunks, int64_t N, int NCB, int L, int ld, const double* __restrict__ cmax,
                                  const double* __restrict__ srow, double* __restrict__ Y, float* __restrict__ Y32) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * ld) return;
    const int64_t i = t / ld;
    const int c = (int)(t - i * ld);
    double y = 0.0;
    if (c < L) {
        y = Y[i * L + c] + srow[i] * bp_value(part, chunks, N, i, NCB * 16, c, cmax[c]);
        Y[i * L + c] = y;
    }
    if (Y32) Y32[t] = (float)y;
}

// A^T Y:  W1[j][c] = S[j][c]   ([H x L] float64; k_sum_panels adds it to the sparse kernels' panel sums)
__global__ void k_bp_combine_cols(const int32_t* __restrict__ part, int chunks, int64_t nrows_pad, int32_t H, int NCB, int L, const double* __restrict__ cmax,
                                  double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)H * L) return;
    const int64_t j = t / L;
    const int c = (int)(t - j * L);
    out[t] = bp_value(part, chunks, nrows_pad, j, NCB * 16, c, cmax[c]);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static size_t bp_align(size_t v) { return (v + 255) & ~(size_t)255; }

// Builds the bitmaps and the reduced structures of the original cells' rows from the resident counts and their
// column-major mirror.  Once per fit and context (ctx->bp.ready).
int bp_build(ddx_ctx* ctx) {
    BitPlanes& bp = ctx->bp;
    if (bp.ready) return DDX_OK;
    const int64_t N = ctx->N;
    const int32_t H = ctx->H;
    const int64_t nnz = ctx->nnz;
    ScopedTimer t(ctx, "bitplane_build");
    bp.KBc = (int)ceil_div(H, 64);
    bp.KBr = ceil_div(N, 64);
    bp.ntile_r = ceil_div(N, 16);
    bp.ntile_c = ceil_div(H, 16);
    const int64_t nseg = (int64_t)ctx->P_o * H;                   // (panel, column) segments of the originals' mirror
    // pass 1: flags + scans (scratch in sort_keys_in / sort_keys_out), totals to the host
    DDX_TRY(ensure(ctx, ctx->sort_keys_in, sizeof(int32_t) * (size_t)(nnz + 1)));
    DDX_TRY(ensure(ctx, ctx->sort_keys_out, sizeof(int32_t) * (size_t)(nnz + 1)));
    int32_t* flag = ctx->sort_keys_in.as<int32_t>();
    int32_t* scan = ctx->sort_keys_out.as<int32_t>();
    size_t tmp = 0;
    DDX_HIP(ctx, prim::exclusive_sum(nullptr, tmp, flag, scan, (size_t)(nnz + 1), ctx->stream));
    DDX_TRY(ensure(ctx, ctx->sort_tmp, tmp));
    const unsigned ge = (unsigned)ceil_div(nnz + 1, 256);
    int32_t total_r = 0, total_m = 0;
    // row-major
    k_bp_flags<<<ge, 256, 0, ctx->stream>>>(ctx->aug_raw.as<float>(), nnz, flag);
    DDX_HIP(ctx, prim::exclusive_sum(ctx->sort_tmp.p, tmp, flag, scan, (size_t)(nnz + 1), ctx->stream));
    DDX_HIP(ctx, hipMemcpyAsync(&total_r, scan + nnz, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    bp.nrest = total_r;
    // one buffer for everything that lives as long as the fit
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += bp_align(bytes); return o; };
    const size_t o_bmr = carve(sizeof(uint64_t) * (size_t)bp.ntile_r * bp.KBc * 16), o_bmc = carve(sizeof(uint64_t) * (size_t)bp.ntile_c * bp.KBr * 16);
    const size_t o_rip = carve(sizeof(int64_t) * (size_t)(N + 1)), o_rc = carve(sizeof(int32_t) * (size_t)total_r), o_rp = carve(sizeof(int32_t) * (size_t)total_r);
    const size_t o_rx = carve(sizeof(float) * (size_t)total_r + 256);
    const size_t o_mcp = carve(sizeof(int64_t) * (size_t)(nseg + 1)), o_mr = carve(sizeof(int32_t) * (size_t)total_r), o_mp = carve(sizeof(int32_t) * (size_t)total_r);
    const size_t o_mx = carve(sizeof(float) * (size_t)total_r + 256), o_s = carve(sizeof(double) * (size_t)N);
    DDX_TRY(ensure(ctx, ctx->bp_buf, off));
    char* b = ctx->bp_buf.as<char>();
    bp.bm_rows = reinterpret_cast<uint64_t*>(b + o_bmr);
    bp.bm_cols = reinterpret_cast<uint64_t*>(b + o_bmc);
    bp.rest_indptr = reinterpret_cast<int64_t*>(b + o_rip);
    bp.rest_cols = reinterpret_cast<int32_t*>(b + o_rc);
    bp.rest_pos = reinterpret_cast<int32_t*>(b + o_rp);
    bp.rest_x = reinterpret_cast<float*>(b + o_rx);
    bp.restm_colptr = reinterpret_cast<int64_t*>(b + o_mcp);
    bp.restm_row = reinterpret_cast<int32_t*>(b + o_mr);
    bp.restm_pos = reinterpret_cast<int32_t*>(b + o_mp);
    bp.restm_x = reinterpret_cast<float*>(b + o_mx);
    bp.srow = reinterpret_cast<double*>(b + o_s);
    k_bp_compact<<<ge, 256, 0, ctx->stream>>>(flag, scan, ctx->aug_indices.as<int32_t>(), nnz, bp.rest_cols, bp.rest_pos);
    k_bp_pointers<<<(unsigned)ceil_div(N + 1, 256), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), N + 1, scan, bp.rest_indptr);
    // column-major mirror (same entries in (panel, column, row) order)
    k_bp_flags<<<ge, 256, 0, ctx->stream>>>(ctx->csc_o_raw.as<float>(), nnz, flag);
    DDX_HIP(ctx, prim::exclusive_sum(ctx->sort_tmp.p, tmp, flag, scan, (size_t)(nnz + 1), ctx->stream));
    DDX_HIP(ctx, hipMemcpyAsync(&total_m, scan + nnz, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    k_bp_compact<<<ge, 256, 0, ctx->stream>>>(flag, scan, ctx->csc_o_row.as<int32_t>(), nnz, bp.restm_row, bp.restm_pos);
    k_bp_pointers<<<(unsigned)ceil_div(nseg + 1, 256), 256, 0, ctx->stream>>>(ctx->csc_o_colptr.as<int64_t>(), nseg + 1, scan, bp.restm_colptr);
    // bitmaps
    k_bp_rows_bitmap<<<(unsigned)ceil_div(bp.ntile_r * bp.KBc * 16, 256), 256, 0, ctx->stream>>>(ctx->aug_indptr.as<int64_t>(), ctx->aug_indices.as<int32_t>(),
                                                                                                ctx->aug_raw.as<float>(), N, bp.KBc, bp.bm_rows);
    k_bp_transpose<<<(unsigned)ceil_div(bp.KBr * bp.KBc, 4), 256, 0, ctx->stream>>>(bp.bm_rows, N, H, bp.KBc, bp.KBr, bp.bm_cols);
    DDX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DDX_HIP(ctx, hipGetLastError());
    if (total_m != total_r) return set_err(ctx, DDX_E_NUMERIC, "bit planes: the mirror holds %d entries other than 1, the rows %d", total_m, total_r);
    bp.ready = true;
    bp.values = false;
    return DDX_OK;
}

// values of this iteration's matrix for the reduced structures (after ddx_lognormalise)
int bp_refresh(ddx_ctx* ctx) {
    BitPlanes& bp = ctx->bp;
    if (bp.values) return DDX_OK;
    ScopedTimer t(ctx, "bitplane_values");
    if (bp.nrest > 0) {
        const unsigned g = (unsigned)ceil_div(bp.nrest, 256);
        k_bp_gather<<<g, 256, 0, ctx->stream>>>(ctx->aug_x.as<float>(), bp.rest_pos, bp.nrest, bp.rest_x);
        k_bp_gather<<<g, 256, 0, ctx->stream>>>(ctx->csc_o_x.as<float>(), bp.restm_pos, bp.nrest, bp.restm_x);
    }
    k_bp_row_scale<<<(unsigned)ceil_div(ctx->N, 256), 256, 0, ctx->stream>>>(ctx->lognorm_tab.as<float>(), 16, ctx->zvalue, ctx->N, bp.srow);
    bp.values = true;
    return DDX_OK;
}

constexpr int kBpRT = 4, kBpWaves = 8;   // 64 bitmap rows per wave, 512 per workgroup
constexpr int kBpColChunks = 12;         // the A^T Y product splits the rows (its k dimension) over this many workgroups per column block

static int bp_workspace(ddx_ctx* ctx, int NCB) {
    BitPlanes& bp = ctx->bp;
    const int64_t KBmax = std::max<int64_t>(bp.KBc, bp.KBr);
    const size_t dig = bp_align(sizeof(v4i) * (size_t)KBmax * NCB * kBpDigits * 64);
    const size_t prt = bp_align(sizeof(int32_t) * (size_t)std::max<int64_t>(ctx->N, (int64_t)kBpColChunks * bp.ntile_c * 16) * NCB * 16 * kBpDigits);
    const size_t need = dig + bp_align(sizeof(double) * 64) + prt + bp_align(sizeof(double) * (size_t)ctx->H * 64);
    DDX_TRY(ensure(ctx, ctx->bp_work, need));
    char* b = ctx->bp_work.as<char>();
    bp.qd = b;
    bp.cmax = reinterpret_cast<double*>(b + dig);
    bp.part = reinterpret_cast<int32_t*>(b + dig + bp_align(sizeof(double) * 64));
    bp.w1 = reinterpret_cast<double*>(b + dig + bp_align(sizeof(double) * 64) + prt);
    return DDX_OK;
}

template <int NCB>
static void bp_launch(ddx_ctx* ctx, const uint64_t* bm, const v4i* qd, int64_t ntile, int64_t KB, int chunks, int per, int64_t nrows, int32_t* part) {
    const dim3 grid((unsigned)ceil_div(ntile, kBpRT * kBpWaves), (unsigned)chunks);
    k_bp_product<kBpRT, kBpWaves, NCB><<<grid, 64 * kBpWaves, 0, ctx->stream>>>(bm, qd, ntile, KB, per, nrows, part);
}

static void bp_product(ddx_ctx* ctx, int NCB, const uint64_t* bm, const v4i* qd, int64_t ntile, int64_t KB, int chunks, int per, int64_t nrows, int32_t* part) {
    if (NCB == 1) bp_launch<1>(ctx, bm, qd, ntile, KB, chunks, per, nrows, part);
    else if (NCB == 2) bp_launch<2>(ctx, bm, qd, ntile, KB, chunks, per, nrows, part);
    else if (NCB == 3) bp_launch<3>(ctx, bm, qd, ntile, KB, chunks, per, nrows, part);
    else bp_launch<4>(ctx, bm, qd, ntile, KB, chunks, per, nrows, part);
}

// Y[i][:] += s_i (B Q)[i][:] for the original cells' rows i < N; Y32 (may be null): the float32 copy [M x ld] refreshed for those rows
int bp_rows_product(ddx_ctx* ctx, const double* Q, int L, int ld, double* Y, float* Y32) {
    BitPlanes& bp = ctx->bp;
    const int NCB = (ld + 15) / 16;
    DDX_TRY(bp_workspace(ctx, NCB));
    v4i* qd = reinterpret_cast<v4i*>(bp.qd);
    DDX_HIP(ctx, hipMemsetAsync(bp.cmax, 0, sizeof(double) * 64, ctx->stream));
    k_bp_colmax<<<(unsigned)std::min<int64_t>(256, ceil_div(ctx->H, 64)), 256, 0, ctx->stream>>>(Q, nullptr, ctx->H, L, bp.cmax);
    k_bp_digits<<<(unsigned)ceil_div((int64_t)bp.KBc * NCB * 64, 256), 256, 0, ctx->stream>>>(Q, nullptr, ctx->H, L, NCB, bp.cmax, bp.KBc, qd);
    bp_product(ctx, NCB, bp.bm_rows, qd, bp.ntile_r, bp.KBc, 1, bp.KBc, ctx->N, bp.part);
    k_bp_combine_rows<<<(unsigned)ceil_div(ctx->N * ld, 256), 256, 0, ctx->stream>>>(bp.part, 1, ctx->N, NCB, L, ld, bp.cmax, bp.srow, Y, Y32);
    return DDX_OK;
}

// W1[j][:] = sum over the original cells i < N of B[i][j] s_i Y[i][:]   ([H x L] float64, returned in *w1)
int bp_cols_product(ddx_ctx* ctx, const double* Y, int L, int ld, const double** w1) {
    BitPlanes& bp = ctx->bp;
    const int NCB = (ld + 15) / 16;
    DDX_TRY(bp_workspace(ctx, NCB));
    v4i* qd = reinterpret_cast<v4i*>(bp.qd);
    DDX_HIP(ctx, hipMemsetAsync(bp.cmax, 0, sizeof(double) * 64, ctx->stream));
    k_bp_colmax<<<(unsigned)std::min<int64_t>(1024, ceil_div(ctx->N, 64)), 256, 0, ctx->stream>>>(Y, bp.srow, ctx->N, L, bp.cmax);
    k_bp_digits<<<(unsigned)ceil_div(bp.KBr * NCB * 64, 256), 256, 0, ctx->stream>>>(Y, bp.srow, ctx->N, L, NCB, bp.cmax, bp.KBr, qd);
    const int chunks = (int)std::min<int64_t>(kBpColChunks, bp.KBr);
    const int per = (int)ceil_div(bp.KBr, chunks);
    const int used = (int)ceil_div(bp.KBr, per);
    const int64_t rows_pad = bp.ntile_c * 16;
    bp_product(ctx, NCB, bp.bm_cols, qd, bp.ntile_c, bp.KBr, used, per, rows_pad, bp.part);
    k_bp_combine_cols<<<(unsigned)ceil_div((int64_t)ctx->H * L, 256), 256, 0, ctx->stream>>>(bp.part, used, rows_pad, ctx->H, NCB, L, bp.cmax, bp.w1);
    *w1 = bp.w1;
    return DDX_OK;
}

}  // namespace ddx
