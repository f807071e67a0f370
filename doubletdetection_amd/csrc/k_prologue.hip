// fit() prologue on the device (dd.py:165-176).
//
// gene variances (dd.py:167-170): scipy evaluates  X.power(2).mean(axis=0) - X.mean(axis=0)**2  for a
// float32 CSR as   sum_i fl(fl(x_ig^2) * fl(1/N))  -  ( sum_i fl(x_ig * fl(1/N)) )^2   where both sums
// are *sequential float32 accumulations in row order* (ones(1,N) @ X -> csc_matvec on the transpose).
// The order decides which genes sit at the rank-H boundary of argsort, so it is reproduced exactly:
// a stable radix sort by column brings each gene's values together in row order, then one wave per
// gene replays the scalar loop (coalesced 64-entry loads, readlane-broadcast sequential adds).
//
// column restriction (dd.py:174-176, tocsc()[:, top].tocsr()): keep the selected genes, renumber them
// to their position in `top` (ascending-variance order), sort every row by the new column id.
#include "ddx_prims.h"

#include "ddx_internal.h"

namespace ddx {

// Canonical-form check of an uploaded CSR (one wave per row): column ids inside [0, G), strictly increasing inside a
// row (sorted, no duplicates -- the doublet merge and its binary searches rely on it), finite values.
// flags: bit 0 non-finite value, bit 1 column out of range, bit 2 unsorted or duplicate column.
__global__ void __launch_bounds__(256) k_validate_csr(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                      const float* __restrict__ vals, int64_t N, int32_t G, int* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;
    const int64_t b = indptr[row], e = indptr[row + 1];
    int bad = 0;
    for (int64_t p0 = b + lane; p0 < e; p0 += 256) {                  // four steps of 64 entries requested together
        int32_t c[4], cp[4];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = p0 + 64 * u;
            c[u] = p < e ? cols[p] : 0;
            cp[u] = (p < e && p > b) ? cols[p - 1] : -1;
            v[u] = p < e ? vals[p] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + 64 * u >= e) break;
            if (!(fabsf(v[u]) <= 3.4028234663852886e38f)) bad |= 1;   // NaN or infinity
            if (c[u] < 0 || c[u] >= G) bad |= 2;
            if (cp[u] >= c[u]) bad |= 4;
        }
    }
    if (bad) atomicOr(flags, bad);
}

int validate_csr(ddx_ctx* ctx, const int64_t* indptr, const int32_t* cols, const float* vals, int64_t N, int32_t G) {
    DDX_TRY(ensure(ctx, ctx->median, 256));
    int* flag = ctx->median.as<int>() + 12;
    int h = 0;
    DDX_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    k_validate_csr<<<(unsigned)ceil_div(N, 4), 256, 0, ctx->stream>>>(indptr, cols, vals, N, G, flag);
    DDX_HIP(ctx, hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));
    if (h & 1) return set_err(ctx, DDX_E_ARG, "counts contain a non-finite value (NaN or infinity)");
    if (h & 2) return set_err(ctx, DDX_E_ARG, "CSR column index outside [0, %d)", G);
    if (h & 4) return set_err(ctx, DDX_E_ARG, "CSR rows must hold strictly increasing column indices (sorted, no duplicates)");
    return DDX_OK;
}

__global__ void k_colptr_i32(const int32_t* __restrict__ keys, int64_t n, int32_t G, int64_t* __restrict__ colptr) {
    int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > G) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < j) lo = mid + 1; else hi = mid;
    }
    colptr[j] = lo;
}

// state (or null): the two running sums per gene, read at the start and written back at the end -- the sequential additions
// of a gene's entries then continue across calls (the rows of a matrix folded in a few at a time, in row order).
// var_out (or null): the variance from the sums as they stand after this call.
__global__ void __launch_bounds__(256) k_gene_var(const int64_t* __restrict__ colptr, const float* __restrict__ vals,
                                                  int32_t G, float rinv, float* __restrict__ var_out, float* __restrict__ state) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= G) return;
    const int64_t b = colptr[g], e = colptr[g + 1];
    float s1 = 0.f, s2 = 0.f;
    if (state) { s1 = state[2 * g]; s2 = state[2 * g + 1]; }
    for (int64_t base = b; base < e; base += 64) {
        const int64_t p = base + lane;
        const float v = (p < e) ? vals[p] : 0.f;
        const float sq = v * v;            // X.power(2): float32 square
        const int a1 = __builtin_bit_cast(int, v * rinv);     // (X * (1/N)) with the scalar rounded to float32
        const int a2 = __builtin_bit_cast(int, sq * rinv);
        const int cnt = (int)((e - base) < 64 ? (e - base) : 64);
        if (cnt == 64) {
            // full chunk: the 64 sequential additions with compile-time lane numbers (v_readlane_b32 into an SGPR, one
            // dependent v_add per sum) -- the two chains are independent and interleave
#pragma unroll
            for (int t = 0; t < 64; ++t) {
                s1 = s1 + __builtin_bit_cast(float, __builtin_amdgcn_readlane(a1, t));
                s2 = s2 + __builtin_bit_cast(float, __builtin_amdgcn_readlane(a2, t));
            }
        } else {
            for (int t = 0; t < cnt; ++t) {
                s1 = s1 + __builtin_bit_cast(float, __builtin_amdgcn_readlane(a1, t));
                s2 = s2 + __builtin_bit_cast(float, __builtin_amdgcn_readlane(a2, t));
            }
        }
    }
    if (lane == 0) {
        if (state) { state[2 * g] = s1; state[2 * g + 1] = s2; }
        if (var_out) {
            const float m2 = s1 * s1;
            var_out[g] = s2 - m2;
        }
    }
}

__global__ void k_variance_from_sums(const float* __restrict__ state, int32_t G, float* __restrict__ var_out) {
#pragma clang fp contract(off)
    const int32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const float s1 = state[2 * g], s2 = state[2 * g + 1];
    const float m2 = s1 * s1;
    var_out[g] = s2 - m2;
}

int gene_sums_fold(ddx_ctx* ctx, int32_t G, int64_t n_rows, int64_t row0, int64_t row1, int64_t e0, int64_t e1, int64_t max_entries) {
    if (row0 == 0) {
        ctx->hvg_rows = -1;
        ctx->hvg_G = G;
        DDX_TRY(ensure(ctx, ctx->hvg_state, sizeof(float) * 2 * (size_t)G));
        DDX_TRY(ensure(ctx, ctx->hvg_keys, sizeof(int32_t) * (size_t)(max_entries + 1)));
        DDX_TRY(ensure(ctx, ctx->hvg_vals, sizeof(float) * (size_t)(max_entries + 1)));
        DDX_TRY(ensure(ctx, ctx->hvg_colptr, sizeof(int64_t) * ((size_t)G + 1)));
        DDX_HIP(ctx, hipMemsetAsync(ctx->hvg_state.p, 0, sizeof(float) * 2 * (size_t)G, ctx->stream));
        ctx->hvg_rows = 0;
    }
    if (ctx->hvg_rows != row0 || ctx->hvg_G != G || e1 - e0 > max_entries) { ctx->hvg_rows = -1; return DDX_OK; }     // out of step: the caller falls back to the whole-matrix pass
    const int64_t n = e1 - e0;
    if (n > 0) {
        int end_bit = 1;
        while ((1 << end_bit) < G) ++end_bit;
        size_t tmp_bytes = 0;
        DDX_HIP(ctx, prim::sort_pairs(nullptr, tmp_bytes, ctx->raw_indices.as<int32_t>() + e0, ctx->hvg_keys.as<int32_t>(), ctx->raw_data.as<float>() + e0,
                                      ctx->hvg_vals.as<float>(), (int)max_entries, 0, end_bit, ctx->stream));
        DDX_TRY(ensure(ctx, ctx->sort_tmp, tmp_bytes));
        DDX_HIP(ctx, prim::sort_pairs(ctx->sort_tmp.p, tmp_bytes, ctx->raw_indices.as<int32_t>() + e0, ctx->hvg_keys.as<int32_t>(), ctx->raw_data.as<float>() + e0,
                                      ctx->hvg_vals.as<float>(), (int)n, 0, end_bit, ctx->stream));
        k_colptr_i32<<<(unsigned)ceil_div(G + 1, 256), 256, 0, ctx->stream>>>(ctx->hvg_keys.as<int32_t>(), n, G, ctx->hvg_colptr.as<int64_t>());
        const float rinv = (float)(1.0 / (double)n_rows);
        k_gene_var<<<(unsigned)ceil_div(G, 4), 256, 0, ctx->stream>>>(ctx->hvg_colptr.as<int64_t>(), ctx->hvg_vals.as<float>(), G, rinv, nullptr, ctx->hvg_state.as<float>());
    }
    ctx->hvg_rows = row1;
    return DDX_OK;
}

int stage_gene_variances(ddx_ctx* ctx, float* var_out) {
    const int64_t n = ctx->raw_nnz;
    const int32_t G = ctx->rawG;
    if (ctx->hvg_rows == ctx->rawN && ctx->hvg_G == G && ctx->hvg_state.p && ctx->opt.hvg_fold) {
        // the sums were folded in while the matrix arrived (ddx_upload_raw): only the last step is left
        DevBuf var;
        DDX_TRY(ensure(ctx, var, sizeof(float) * G));
        ScopedTimer t(ctx, "hvg_variance");
        k_variance_from_sums<<<(unsigned)ceil_div(G, 256), 256, 0, ctx->stream>>>(ctx->hvg_state.as<float>(), G, var.as<float>());
        hipError_t e = hipMemcpyAsync(var_out, var.p, sizeof(float) * G, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = wait_stream(ctx);
        release(ctx, var);
        // (the work space of the folding goes back to the context; a second call takes the whole-matrix pass)
        release(ctx, ctx->hvg_keys); release(ctx, ctx->hvg_vals); release(ctx, ctx->hvg_colptr); release(ctx, ctx->hvg_state);
        ctx->hvg_rows = -1;
        if (e != hipSuccess) return set_err(ctx, DDX_E_HIP, "gene variance stage failed: %s", hipGetErrorString(e));
        return DDX_OK;
    }
    DevBuf keys_out, vals_out, colptr, var;
    int rc = DDX_OK;
    auto cleanup = [&]() { release(ctx, keys_out); release(ctx, vals_out); release(ctx, colptr); release(ctx, var); };
    if ((rc = ensure(ctx, keys_out, sizeof(int32_t) * (n + 1))) || (rc = ensure(ctx, vals_out, sizeof(float) * (n + 1))) ||
        (rc = ensure(ctx, colptr, sizeof(int64_t) * (G + 1))) || (rc = ensure(ctx, var, sizeof(float) * G))) {
        cleanup();
        return rc;
    }
    int end_bit = 1;
    while ((1 << end_bit) < G) ++end_bit;
    hipError_t e = hipSuccess;
    if (n > 0) {
        size_t tmp_bytes = 0;
        e = prim::sort_pairs(nullptr, tmp_bytes, ctx->raw_indices.as<int32_t>(), keys_out.as<int32_t>(),
                                               ctx->raw_data.as<float>(), vals_out.as<float>(), (int)n, 0, end_bit, ctx->stream);
        if (e == hipSuccess && (rc = ensure(ctx, ctx->sort_tmp, tmp_bytes)) == DDX_OK) {
            ScopedTimer t(ctx, "hvg_sort");
            e = prim::sort_pairs(ctx->sort_tmp.p, tmp_bytes, ctx->raw_indices.as<int32_t>(), keys_out.as<int32_t>(),
                                                   ctx->raw_data.as<float>(), vals_out.as<float>(), (int)n, 0, end_bit, ctx->stream);
        }
    }
    if (e == hipSuccess && rc == DDX_OK) {
        ScopedTimer t(ctx, "hvg_variance");
        k_colptr_i32<<<(unsigned)ceil_div(G + 1, 256), 256, 0, ctx->stream>>>(keys_out.as<int32_t>(), n, G, colptr.as<int64_t>());
        const float rinv = (float)(1.0 / (double)ctx->rawN);
        k_gene_var<<<(unsigned)ceil_div(G, 4), 256, 0, ctx->stream>>>(colptr.as<int64_t>(), vals_out.as<float>(), G, rinv, var.as<float>(), nullptr);
    }
    if (e == hipSuccess && rc == DDX_OK) e = hipMemcpyAsync(var_out, var.p, sizeof(float) * G, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == DDX_OK) e = wait_stream(ctx);
    cleanup();
    if (rc != DDX_OK) return rc;
    if (e != hipSuccess) return set_err(ctx, DDX_E_HIP, "gene variance stage failed: %s", hipGetErrorString(e));
    return DDX_OK;
}

// ------------------------------------------------------------------------------------------------
// column restriction
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_count_kept(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                    const int32_t* __restrict__ newid, int64_t N, int32_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;
    int c = 0;
    const int64_t e = indptr[row + 1];
    for (int64_t p = indptr[row] + lane; p < e; p += 256) {          // four dependent load pairs in flight per lane
        int32_t j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = p + 64 * u < e ? cols[p + 64 * u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) c += (j[u] >= 0 && newid[j[u]] >= 0);
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (lane == 0) counts[row] = c;
}

// order-preserving compaction of the kept entries of each row (one wave per row, ballot prefix)
__global__ void __launch_bounds__(256) k_compact_kept(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                      const float* __restrict__ vals, const int32_t* __restrict__ newid,
                                                      int64_t N, const int64_t* __restrict__ out_ptr,
                                                      int32_t* __restrict__ out_cols, float* __restrict__ out_vals) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;
    int64_t o = out_ptr[row];
    const int64_t b = indptr[row], e = indptr[row + 1];
    for (int64_t base = b; base < e; base += 256) {                  // four 64-entry steps requested together, written in order
        int32_t nid[4];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = base + 64 * u + lane;
            const int32_t j = p < e ? cols[p] : -1;
            v[u] = p < e ? vals[p] : 0.f;
            nid[u] = j;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) nid[u] = nid[u] >= 0 ? newid[nid[u]] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long m = __ballot(nid[u] >= 0);
            if (nid[u] >= 0) {
                const int before = __popcll(m & ((1ull << lane) - 1ull));
                out_cols[o + before] = nid[u];
                out_vals[o + before] = v[u];
            }
            o += __popcll(m);
        }
    }
}

__global__ void k_fill_i32(int32_t* out, int64_t n, int32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

__global__ void k_scatter_newid(const int64_t* __restrict__ sel, int32_t H, int32_t* __restrict__ newid) {
    int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < H) newid[sel[t]] = t;
}

// exclusive scan of per-row counts (same single-block scheme as the doublet stage)
__global__ void __launch_bounds__(1024) k_scan_rows(const int32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t start = 0; start < n; start += 1024) {
        const int64_t i = start + tid;
        const int64_t v = (i < n) ? (int64_t)in[i] : 0;
        int64_t x = v;
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

// The same compaction with the kept entries written in the order of their NEW column numbers (the selected genes are numbered by
// variance rank, dd.py:283, so a row's kept entries change order): a wave marks the row's new columns in a bitmap of H bits in
// LDS, prefix-sums the words' popcounts, and an entry's slot is the number of marked columns below its own -- no sort of the
// compacted rows afterwards (rounds 1-4: a segmented radix sort of all kept entries, 1.1 ms at the headline).  W = ceil(H / 32).
__global__ void __launch_bounds__(256) k_compact_ranked(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                        const float* __restrict__ vals, const int32_t* __restrict__ newid,
                                                        int64_t N, int W, const int64_t* __restrict__ out_ptr,
                                                        int32_t* __restrict__ out_cols, float* __restrict__ out_vals) {
    extern __shared__ uint32_t ck_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= N) return;
    uint32_t* bm = ck_lds + (size_t)wave * 2 * W;
    uint32_t* pre = bm + W;
    for (int d = lane; d < W; d += 64) bm[d] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t b = indptr[row], e = indptr[row + 1];
    for (int64_t p = b + lane; p < e; p += 256) {
        int32_t j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = p + 64 * u < e ? cols[p + 64 * u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = j[u] >= 0 ? newid[j[u]] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j[u] >= 0) atomicOr(&bm[j[u] >> 5], 1u << (j[u] & 31));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int carry = 0;
    for (int k0 = 0; k0 < W; k0 += 64) {
        const int d = k0 + lane;
        const int c = d < W ? __popc(bm[d]) : 0;
        int x = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (d < W) pre[d] = (uint32_t)(carry + x - c);
        carry += __shfl(x, 63, 64);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t o = out_ptr[row];
    for (int64_t p = b + lane; p < e; p += 256) {
        int32_t j[4];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            j[u] = p + 64 * u < e ? cols[p + 64 * u] : -1;
            v[u] = p + 64 * u < e ? vals[p + 64 * u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = j[u] >= 0 ? newid[j[u]] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j[u] >= 0) {
                const int slot = (int)pre[j[u] >> 5] + __popc(bm[j[u] >> 5] & ((1u << (j[u] & 31)) - 1u));
                out_cols[o + slot] = j[u];
                out_vals[o + slot] = v[u];
            }
    }
}

int stage_select_columns(ddx_ctx* ctx, const int64_t* cols, int32_t H) {
    const int64_t N = ctx->rawN;
    const int32_t G = ctx->rawG;
    for (int32_t t = 0; t < H; ++t)
        if (cols[t] < 0 || cols[t] >= G) return set_err(ctx, DDX_E_ARG, "column %lld out of range", (long long)cols[t]);
    DevBuf sel, newid, counts, tk, tv;
    int rc = DDX_OK;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() { release(ctx, sel); release(ctx, newid); release(ctx, counts); release(ctx, tk); release(ctx, tv); };
#define PR_TRY(x) if ((rc = (x)) != DDX_OK) { cleanup(); return rc; }
#define PR_HIP(x) if ((e = (x)) != hipSuccess) { cleanup(); return set_err(ctx, DDX_E_HIP, "%s: %s", #x, hipGetErrorString(e)); }
    PR_TRY(ensure(ctx, sel, sizeof(int64_t) * H));
    PR_TRY(ensure(ctx, newid, sizeof(int32_t) * G));
    PR_TRY(ensure(ctx, counts, sizeof(int32_t) * (N + 1)));
    PR_TRY(ensure(ctx, ctx->aug_indptr, sizeof(int64_t) * (N + N / 2 + 2)));
    PR_HIP(hipMemcpyAsync(sel.p, cols, sizeof(int64_t) * H, hipMemcpyHostToDevice, ctx->stream));
    int64_t kept = 0;
    {
        ScopedTimer t(ctx, "hvg_select");
        k_fill_i32<<<(unsigned)ceil_div(G, 256), 256, 0, ctx->stream>>>(newid.as<int32_t>(), G, -1);
        k_scatter_newid<<<(unsigned)ceil_div(H, 256), 256, 0, ctx->stream>>>(sel.as<int64_t>(), H, newid.as<int32_t>());
        k_count_kept<<<(unsigned)ceil_div(N, 4), 256, 0, ctx->stream>>>(ctx->raw_indptr.as<int64_t>(), ctx->raw_indices.as<int32_t>(),
                                                                        newid.as<int32_t>(), N, counts.as<int32_t>());
        k_scan_rows<<<1, 1024, 0, ctx->stream>>>(counts.as<int32_t>(), N, ctx->aug_indptr.as<int64_t>());
    }
    ctx->h_indptr.resize(N + 1);
    PR_HIP(hipMemcpyAsync(ctx->h_indptr.data(), ctx->aug_indptr.p, sizeof(int64_t) * (N + 1), hipMemcpyDeviceToHost, ctx->stream));
    PR_HIP(wait_stream(ctx));
    kept = ctx->h_indptr[N];
    if (kept >= (int64_t)1 << 31) { cleanup(); return set_err(ctx, DDX_E_UNSUPPORTED, "more than 2^31-1 stored entries"); }
    const int64_t cap_s = kept / 2 + kept / 8 + 1024;
    PR_TRY(ensure(ctx, ctx->aug_indices, sizeof(int32_t) * (size_t)(kept + cap_s)));
    PR_TRY(ensure(ctx, ctx->aug_raw, sizeof(float) * (size_t)(kept + cap_s)));
    PR_TRY(ensure(ctx, ctx->aug_x, sizeof(float) * (size_t)(kept + cap_s)));
    ctx->cap_synth = cap_s;
    const int W = (H + 31) / 32;
    const size_t ranked_lds = sizeof(uint32_t) * 4 * 2 * (size_t)W;
    if (kept > 0 && ranked_lds <= 64 * 1024) {
        // compaction straight into new-column order
        PR_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_compact_ranked), (int)ranked_lds));
        ScopedTimer t(ctx, "hvg_select");
        k_compact_ranked<<<(unsigned)ceil_div(N, 4), 256, ranked_lds, ctx->stream>>>(ctx->raw_indptr.as<int64_t>(), ctx->raw_indices.as<int32_t>(),
                                                                                    ctx->raw_data.as<float>(), newid.as<int32_t>(), N, W,
                                                                                    ctx->aug_indptr.as<int64_t>(), ctx->aug_indices.as<int32_t>(), ctx->aug_raw.as<float>());
    } else if (kept > 0) {
        // (more selected columns than a wave's bitmap holds: order-preserving compaction, then a segmented sort of the rows)
        PR_TRY(ensure(ctx, tk, sizeof(int32_t) * (size_t)(kept + 1)));
        PR_TRY(ensure(ctx, tv, sizeof(float) * (size_t)(kept + 1)));
        int end_bit = 1;
        while ((1 << end_bit) < H) ++end_bit;
        size_t tmp_bytes = 0;
        const int64_t* offs = ctx->aug_indptr.as<int64_t>();
        PR_HIP(prim::segmented_sort_pairs(nullptr, tmp_bytes, tk.as<int32_t>(), ctx->aug_indices.as<int32_t>(),
                                                           tv.as<float>(), ctx->aug_raw.as<float>(), (int)kept, (int)N, offs, offs + 1,
                                                           0, end_bit, ctx->stream));
        PR_TRY(ensure(ctx, ctx->sort_tmp, tmp_bytes));
        ScopedTimer t(ctx, "hvg_select");
        k_compact_kept<<<(unsigned)ceil_div(N, 4), 256, 0, ctx->stream>>>(ctx->raw_indptr.as<int64_t>(), ctx->raw_indices.as<int32_t>(),
                                                                          ctx->raw_data.as<float>(), newid.as<int32_t>(), N,
                                                                          ctx->aug_indptr.as<int64_t>(), tk.as<int32_t>(), tv.as<float>());
        PR_HIP(prim::segmented_sort_pairs(ctx->sort_tmp.p, tmp_bytes, tk.as<int32_t>(), ctx->aug_indices.as<int32_t>(),
                                                           tv.as<float>(), ctx->aug_raw.as<float>(), (int)kept, (int)N, offs, offs + 1,
                                                           0, end_bit, ctx->stream));
    }
    PR_HIP(wait_stream(ctx));
    cleanup();
#undef PR_TRY
#undef PR_HIP
    ctx->nnz = kept;
    return stage_upload_counts(ctx, N, H, nullptr, nullptr, nullptr, true);
}

}  // namespace ddx
