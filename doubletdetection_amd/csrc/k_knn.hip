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
#include <cstdlib>

#include "ddx_internal.h"

namespace ddx {

constexpr int kMaxDim = 64;

typedef float f4 __attribute__((ext_vector_type(4)));

// ---- layouts -----------------------------------------------------------------------------------
// E   : row-major float32 [Mp][CP]            (exact re-evaluation gathers whole rows)
// Et  : MFMA operand layout, 16-point tiles:   Et[tile][s][k][j] = E[16*tile + j][4*s + k]
//       so that lane l of a wave reads operand element (point l&15, component 4s + (l>>4)) at
//       Et[tile*16*CP + s*64 + l] -- one fully coalesced 256-byte load per MFMA k-step, and the same
//       formula serves the A operand (queries) and the B operand (candidates).
// nrm : float32 squared norms; padding points carry +inf so they can never pass the screen.
__global__ void k_knn_prepare(const float* __restrict__ in, int64_t M, int64_t Mp, int C, int CP,
                              float* __restrict__ E, float* __restrict__ Et, float* __restrict__ nrm) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Mp) return;
    float n = 0.f;
    const int64_t tile = r >> 4;
    const int j = (int)(r & 15);
    for (int d = 0; d < CP; ++d) {
        const float v = (r < M && d < C) ? in[r * C + d] : 0.f;
        E[r * CP + d] = v;
        Et[tile * 16 * CP + (d >> 2) * 64 + (d & 3) * 16 + j] = v;
        n = fmaf(v, v, n);
    }
    nrm[r] = (r < M) ? n : __builtin_huge_valf();
}

// Exact squared distance in the reference's arithmetic: float64, (a-b)*(a-b) rounded, then added,
// components in order -- bit-identical to the float64 brute-force definition (oracle knn_bruteforce_f64).
template <int CP>
__device__ __forceinline__ double exact_d2(const float* __restrict__ a, const float* __restrict__ b) {
#pragma clang fp contract(off)
    double d2 = 0.0;
#pragma unroll
    for (int t = 0; t < CP; t += 4) {
        const f4 x = *reinterpret_cast<const f4*>(a + t);
        const f4 y = *reinterpret_cast<const f4*>(b + t);
        const double d0 = (double)x.x - (double)y.x; const double s0 = d0 * d0; d2 = d2 + s0;
        const double d1 = (double)x.y - (double)y.y; const double s1 = d1 * d1; d2 = d2 + s1;
        const double d2_ = (double)x.z - (double)y.z; const double s2 = d2_ * d2_; d2 = d2 + s2;
        const double d3 = (double)x.w - (double)y.w; const double s3 = d3 * d3; d2 = d2 + s3;
    }
    return d2;
}

// Screen slack: |fl32(|q|^2 + |c|^2 - 2 q.c) - d2| <= (CP + 8) * 2^-24 * (|q|^2 + |c|^2) for the float32
// MFMA dot product (a k-ordered fmaf chain) and float32 norms; 7e-6 covers CP <= 64 with margin.
constexpr float kScreenSlack = 7.0e-6f;

constexpr int kQPerWave = 32;                 // two 16-row MFMA tiles of queries per wave
constexpr int kQPerBlock = 4 * kQPerWave;     // 4 waves, no cross-wave sharing

// kNN = float32 MFMA screen + exact float64 confirmation.
//  screen : for a 16x16 (query x candidate) tile the squared distances are |q|^2 + |c|^2 - 2 q.c with
//           q.c from CP/4 v_mfma_f32_16x16x4_f32; a pair survives when the value minus a rigorous
//           rounding bound is below the query's current k-th best distance (kept as a float32 upper
//           bound).  After the first few hundred candidates almost nothing survives.
//  confirm: every surviving lane re-evaluates its pair exactly in float64 (all survivors of a tile in
//           parallel) and inserts it into the query's sorted top-k list in LDS, ordered by
//           (distance, index) so the result does not depend on the visiting order.
template <int CP>
__global__ void __launch_bounds__(256) k_knn_mfma(const float* __restrict__ E, const float* __restrict__ Et,
                                                  const float* __restrict__ nrm, int64_t M, int64_t Mp, int K,
                                                  int include_self, int32_t* __restrict__ idx_out,
                                                  double* __restrict__ dist_out, int debug_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ld = reinterpret_cast<double*>(smem);                                   // [K][kQPerBlock]
    int32_t* li = reinterpret_cast<int32_t*>(ld + (size_t)K * kQPerBlock);          // [K][kQPerBlock]
    int32_t* cnt = li + (size_t)K * kQPerBlock;                                     // [kQPerBlock]
    float* thr = reinterpret_cast<float*>(cnt + kQPerBlock);                        // [kQPerBlock] float32 upper bound of the k-th best
    int32_t* owner = reinterpret_cast<int32_t*>(thr + kQPerBlock);                  // [kQPerBlock]
    constexpr int KS = CP / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q0 = ((int64_t)blockIdx.x * 4 + wave) * kQPerWave;                // first query of this wave
    if (q0 >= Mp) return;                                                           // (whole wave; no block barriers are used)
    const int lq0 = wave * kQPerWave;                                               // first local query slot
    for (int t = lane; t < kQPerWave; t += 64) { cnt[lq0 + t] = 0; thr[lq0 + t] = __builtin_huge_valf(); owner[lq0 + t] = -1; }

    // A operands: 2 query tiles x KS k-steps
    float a[2][KS];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int s = 0; s < KS; ++s) a[rt][s] = Et[((q0 >> 4) + rt) * 16 * CP + s * 64 + lane];
    // rows held by this lane in the MFMA result: row(reg) = 4*(lane>>4) + reg, column = lane & 15
    const int rbase = 4 * (lane >> 4);
    float nq[2][4];
    int64_t qid[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            qid[rt][r] = q0 + rt * 16 + rbase + r;
            nq[rt][r] = nrm[qid[rt][r]];      // +inf for padding queries: they never pass the screen
        }
    float th[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) th[rt][r] = __builtin_huge_valf();

    const int64_t ntiles = Mp >> 4;
    float b[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) b[s] = Et[s * 64 + lane];
    float ncand = nrm[lane & 15];
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        // prefetch the next candidate tile while this one is in the matrix pipe
        float bn[KS];
        const int64_t tn = (tile + 1 < ntiles) ? tile + 1 : tile;
#pragma unroll
        for (int s = 0; s < KS; ++s) bn[s] = Et[tn * 16 * CP + s * 64 + lane];
        const float ncand_n = nrm[tn * 16 + (lane & 15)];

        f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][s], b[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][s], b[s], acc1, 0, 0, 0);
        }
        const int64_t cand = tile * 16 + (lane & 15);
        unsigned hits = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float n0 = nq[0][r] + ncand, n1 = nq[1][r] + ncand;
            const float s0 = fmaf(-2.f, acc0[r], n0) - kScreenSlack * n0;
            const float s1 = fmaf(-2.f, acc1[r], n1) - kScreenSlack * n1;
            if (s0 < th[0][r] && (include_self || cand != qid[0][r])) hits |= 1u << r;
            if (s1 < th[1][r] && (include_self || cand != qid[1][r])) hits |= 16u << r;
        }
        if (debug_mode == 1) hits = 0;   // profiling aid: screen only
        if (__ballot(hits != 0)) {
            // confirm: each lane walks its own surviving pairs; lanes work in parallel
            while (__ballot(hits != 0)) {
                int lq = -1;
                double d2 = 0.0;
                if (hits) {
                    const int bit = __ffs(hits) - 1;
                    hits &= hits - 1;
                    lq = lq0 + (bit >> 2) * 16 + rbase + (bit & 3);
                    const int64_t qg = q0 + (bit >> 2) * 16 + rbase + (bit & 3);
                    d2 = exact_d2<CP>(E + qg * CP, E + cand * CP);
                    // cheap exact pre-check against the list's current k-th entry
                    const int c0 = cnt[lq];
                    if (c0 == K) {
                        const double wd = ld[(K - 1) * kQPerBlock + lq];
                        const int32_t wi = li[(K - 1) * kQPerBlock + lq];
                        if (!(d2 < wd || (d2 == wd && (int32_t)cand < wi))) lq = -1;
                    }
                }
                // one inserter per query at a time (lanes of this wave only ever touch this wave's queries)
                bool pending = lq >= 0;
                while (__ballot(pending)) {
                    if (pending) owner[lq] = lane;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (pending && owner[lq] == lane) {
                        const int c0 = cnt[lq];
                        bool take = true;
                        if (c0 == K) {
                            const double wd = ld[(K - 1) * kQPerBlock + lq];
                            const int32_t wi = li[(K - 1) * kQPerBlock + lq];
                            take = d2 < wd || (d2 == wd && (int32_t)cand < wi);
                        }
                        if (take) {
                            int pos = (c0 < K) ? c0 : K - 1;
                            while (pos > 0) {
                                const double pd = ld[(pos - 1) * kQPerBlock + lq];
                                const int32_t pi = li[(pos - 1) * kQPerBlock + lq];
                                if (!(pd > d2 || (pd == d2 && pi > (int32_t)cand))) break;
                                ld[pos * kQPerBlock + lq] = pd;
                                li[pos * kQPerBlock + lq] = pi;
                                --pos;
                            }
                            ld[pos * kQPerBlock + lq] = d2;
                            li[pos * kQPerBlock + lq] = (int32_t)cand;
                            const int c1 = (c0 < K) ? c0 + 1 : K;
                            cnt[lq] = c1;
                            if (c1 == K) thr[lq] = __double2float_ru(ld[(K - 1) * kQPerBlock + lq]);
                        }
                        pending = false;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // refresh the float32 thresholds of the rows this lane screens
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) th[rt][r] = thr[lq0 + rt * 16 + rbase + r];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) b[s] = bn[s];
        ncand = ncand_n;
    }
    // write back: lane t < 32 owns local query t
    for (int t = lane; t < kQPerWave; t += 64) {
        const int64_t qg = q0 + t;
        if (qg < M) {
            const int c = cnt[lq0 + t];
            for (int s = 0; s < K; ++s) {
                idx_out[qg * K + s] = (s < c) ? li[s * kQPerBlock + lq0 + t] : -1;
                dist_out[qg * K + s] = (s < c) ? ld[s * kQPerBlock + lq0 + t] : __builtin_huge_val();
            }
        }
    }
}

int stage_knn(ddx_ctx* ctx, int32_t k, int32_t include_self) {
    const int64_t M = ctx->embM;
    const int C = ctx->C;
    if (C > kMaxDim) return set_err(ctx, DDX_E_UNSUPPORTED, "embedding dimension %d exceeds %d", C, kMaxDim);
    const int CP = (C <= 32) ? 32 : 64;
    const int64_t Mp = ceil_div(M, kQPerWave) * kQPerWave;     // whole query tiles per wave
    // workspace (reuses the PCA row buffer): E [Mp*CP] | Et [Mp*CP] | nrm [Mp]
    DDX_TRY(ensure(ctx, ctx->pcaA, sizeof(float) * ((size_t)Mp * CP * 2 + Mp + 64)));
    DDX_TRY(ensure(ctx, ctx->knn_idx, sizeof(int32_t) * (size_t)M * k));
    DDX_TRY(ensure(ctx, ctx->knn_dist, sizeof(double) * (size_t)M * k));
    float* E = ctx->pcaA.as<float>();
    float* Et = E + (size_t)Mp * CP;
    float* nrm = Et + (size_t)Mp * CP;
    k_knn_prepare<<<(unsigned)ceil_div(Mp, 256), 256, 0, ctx->stream>>>(ctx->emb32.as<float>(), M, Mp, C, CP, E, Et, nrm);
    const size_t lds = (sizeof(double) + sizeof(int32_t)) * (size_t)k * kQPerBlock + 3 * sizeof(int32_t) * kQPerBlock;
    {
        ScopedTimer t(ctx, "knn_brute");
        const unsigned grid = (unsigned)ceil_div(Mp, kQPerBlock);
        const char* dbg_env = getenv("DDX_KNN_DEBUG");
        const int dbg = dbg_env ? atoi(dbg_env) : 0;
        if (lds > 48 * 1024) {
            DDX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_knn_mfma<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            DDX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_knn_mfma<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (CP == 32)
            k_knn_mfma<32><<<grid, 256, lds, ctx->stream>>>(E, Et, nrm, M, Mp, k, include_self, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<double>(), dbg);
        else
            k_knn_mfma<64><<<grid, 256, lds, ctx->stream>>>(E, Et, nrm, M, Mp, k, include_self, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<double>(), dbg);
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

// device part: relation weights for the kNN table currently held by the context
int stage_graph_relations(ddx_ctx* ctx, int32_t mode, int32_t* idx_host, double* w_host) {
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

int stage_build_graph(ddx_ctx* ctx, int32_t mode) {
    const int64_t M = ctx->embM;
    const int K = ctx->K;
    std::vector<int32_t> idx((size_t)M * K);
    std::vector<double> w((size_t)M * K);
    DDX_TRY(stage_graph_relations(ctx, mode, idx.data(), w.data()));
    assemble_graph(M, K, idx.data(), w.data(), ctx->g_indptr, ctx->g_indices, ctx->g_weights);
    return DDX_OK;
}

}  // namespace ddx
