// Exact brute-force kNN over the low-dimensional embedding and the graphs handed to community
// detection (phenograph.cluster / sc.pp.neighbors call sites, dd.py:317-336).
//
// kNN: one thread owns one query (its coordinates live in VGPRs as float64); every thread of the
// wave walks the same candidate, whose coordinates are wave-uniform and therefore come through the
// scalar cache (no LDS staging, no per-lane loads).  Squared distances are accumulated exactly as
// the float64 reference does (subtract, multiply, add -- no fused multiply-add) so that the ordering
// by (distance, index) is bit-identical to an IEEE float64 brute force.  The running top-k of each
// thread lives in LDS, slot-major ([slot][thread]) so lanes hit distinct banks; insertions are rare
// after the first few hundred candidates (expected k*ln(M/k) per query).
#include <algorithm>

#include "ddx_internal.h"

namespace ddx {

constexpr int kKnnThreads = 128;
constexpr int kMaxDim = 64;

__global__ void k_f32_to_f64_pad(const float* __restrict__ in, int64_t rows, int C, int CP, double* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * CP) return;
    const int64_t r = i / CP;
    const int c = (int)(i - r * CP);
    out[i] = (c < C) ? (double)in[r * C + c] : 0.0;
}

template <int CP>
__global__ void __launch_bounds__(kKnnThreads) k_knn_brute(const double* __restrict__ E, int64_t M, int C, int K,
                                                           int include_self, int32_t* __restrict__ idx_out,
                                                           double* __restrict__ dist_out) {
#pragma clang fp contract(off)  // (a-b)*(a-b) must round before the add: no fused multiply-add
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ld = reinterpret_cast<double*>(smem);                       // [K][kKnnThreads]
    int32_t* li = reinterpret_cast<int32_t*>(smem + sizeof(double) * K * kKnnThreads);
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * kKnnThreads + tid;
    const bool valid = q < M;
    double qv[CP];
#pragma unroll
    for (int t = 0; t < CP; ++t) qv[t] = (valid && t < C) ? E[q * CP + t] : 0.0;
    int count = 0;
    double worst = __builtin_huge_val();
    for (int64_t c = 0; c < M; ++c) {
        const double* __restrict__ ec = E + c * CP;   // wave-uniform address
        double d2 = 0.0;
#pragma unroll
        for (int t = 0; t < CP; ++t) {  // padded coordinates are 0: they add exactly +0.0
            const double diff = qv[t] - ec[t];
            const double sq = diff * diff;      // rounded product (contract(off) above), then rounded add
            d2 = d2 + sq;
        }
        if (valid && d2 < worst && (include_self || c != q)) {
            int pos = (count < K) ? count : K - 1;
            while (pos > 0 && ld[(pos - 1) * kKnnThreads + tid] > d2) {
                ld[pos * kKnnThreads + tid] = ld[(pos - 1) * kKnnThreads + tid];
                li[pos * kKnnThreads + tid] = li[(pos - 1) * kKnnThreads + tid];
                --pos;
            }
            ld[pos * kKnnThreads + tid] = d2;
            li[pos * kKnnThreads + tid] = (int32_t)c;
            if (count < K) ++count;
            if (count == K) worst = ld[(K - 1) * kKnnThreads + tid];
        }
    }
    if (valid) {
        for (int s = 0; s < K; ++s) {
            idx_out[q * K + s] = (s < count) ? li[s * kKnnThreads + tid] : -1;
            dist_out[q * K + s] = (s < count) ? ld[s * kKnnThreads + tid] : __builtin_huge_val();
        }
    }
}

int stage_knn(ddx_ctx* ctx, int32_t k, int32_t include_self) {
    const int64_t M = ctx->embM;
    const int C = ctx->C;
    if (C > kMaxDim) return set_err(ctx, DDX_E_UNSUPPORTED, "embedding dimension %d exceeds %d", C, kMaxDim);
    const int CP = (C <= 32) ? 32 : 64;
    DDX_TRY(ensure(ctx, ctx->pcaA, sizeof(double) * (size_t)M * CP));  // reuse PCA workspace for the f64 copy
    DDX_TRY(ensure(ctx, ctx->knn_idx, sizeof(int32_t) * (size_t)M * k));
    DDX_TRY(ensure(ctx, ctx->knn_dist, sizeof(double) * (size_t)M * k));
    double* E = ctx->pcaA.as<double>();
    k_f32_to_f64_pad<<<(unsigned)ceil_div(M * CP, 256), 256, 0, ctx->stream>>>(ctx->emb32.as<float>(), M, C, CP, E);
    const size_t lds = (sizeof(double) + sizeof(int32_t)) * (size_t)k * kKnnThreads;
    {
        ScopedTimer t(ctx, "knn_brute");
        const unsigned grid = (unsigned)ceil_div(M, kKnnThreads);
        if (lds > 48 * 1024) {
            DDX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_knn_brute<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            DDX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_knn_brute<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (CP == 32)
            k_knn_brute<32><<<grid, kKnnThreads, lds, ctx->stream>>>(E, M, C, k, include_self, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<double>());
        else
            k_knn_brute<64><<<grid, kKnnThreads, lds, ctx->stream>>>(E, M, C, k, include_self, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<double>());
    }
    DDX_HIP(ctx, hipGetLastError());
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

int stage_build_graph(ddx_ctx* ctx, int32_t mode) {
    const int64_t M = ctx->embM;
    const int K = ctx->K;
    DDX_TRY(ensure(ctx, ctx->knn_sorted, sizeof(int32_t) * (size_t)M * K));
    DDX_TRY(ensure(ctx, ctx->edge_w, sizeof(double) * (size_t)M * K));
    {
        ScopedTimer t(ctx, "graph_weights");
        k_sort_neighbours<<<(unsigned)ceil_div(M, 64), 64, sizeof(int32_t) * K * 64, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), M, K, ctx->knn_sorted.as<int32_t>());
        k_edge_weights<<<(unsigned)ceil_div(M * K, 256), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->knn_sorted.as<int32_t>(), M, K, mode,
                                                                                ctx->edge_w.as<double>());
    }
    std::vector<int32_t> idx((size_t)M * K);
    std::vector<double> w((size_t)M * K);
    DDX_HIP(ctx, hipMemcpyAsync(idx.data(), ctx->knn_idx.p, sizeof(int32_t) * idx.size(), hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, hipMemcpyAsync(w.data(), ctx->edge_w.p, sizeof(double) * w.size(), hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // symmetric CSR on the host: every relation with a non-zero weight contributes (i,j); a
    // non-mutual one (negative flag) also contributes (j,i)
    std::vector<int64_t>& ip = ctx->g_indptr;
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
    ctx->g_indices.assign(E, 0);
    ctx->g_weights.assign(E, 0.0);
    std::vector<int64_t> cur(ip.begin(), ip.end() - 1);
    for (int64_t i = 0; i < M; ++i)
        for (int a = 0; a < K; ++a) {
            const double v = w[i * K + a];
            if (v == 0.0) continue;
            const int32_t j = idx[i * K + a];
            const double av = v < 0.0 ? -v : v;
            ctx->g_indices[cur[i]] = j;
            ctx->g_weights[cur[i]++] = av;
            if (v < 0.0) {
                ctx->g_indices[cur[j]] = (int32_t)i;
                ctx->g_weights[cur[j]++] = av;
            }
        }
    // rows sorted by neighbour index (the community-detection spec visits adjacency in this order)
    std::vector<std::pair<int32_t, double>> tmp;
    for (int64_t i = 0; i < M; ++i) {
        const int64_t b = ip[i], e = ip[i + 1];
        tmp.resize(e - b);
        for (int64_t p = b; p < e; ++p) tmp[p - b] = {ctx->g_indices[p], ctx->g_weights[p]};
        std::sort(tmp.begin(), tmp.end(), [](const std::pair<int32_t, double>& x, const std::pair<int32_t, double>& y) { return x.first < y.first; });
        for (int64_t p = b; p < e; ++p) {
            ctx->g_indices[p] = tmp[p - b].first;
            ctx->g_weights[p] = tmp[p - b].second;
        }
    }
    return DDX_OK;
}

}  // namespace ddx
