// Parts A and C of the community-detection specification (oracle/louvain_ref.py:presweep, refine) on the GPU:
// synchronous sub-round sweeps of the local-moving step on integer-quantised weights, followed (part A) by an exact
// aggregation of the communities or started (part C) from the partition the sequential levels found.
// Part A is applied to the symmetric CSR that ddx_build_graph left on the device; the host only sees the aggregated
// graph (a few thousand super-nodes), runs the sequential multi-level optimisation on it (ddx_louvain_sequential) and
// hands the labels of the super-nodes back for part C (ddx_refine_communities), which returns the final labels.
// This replaces the Louvain stage inside phenograph.cluster / sc.tl.louvain (dd.py:320-322, 337-342) together with
// louvain.cpp.
//
// Why it is exact and order-free: edge weights are rounded once to multiples of 2^-20 and every sum (node strength,
// community total, node-to-community weight, aggregated edge weight) is an int64 sum -- atomics and sorts may
// reorder them freely.  The only floating-point work is the score
//     score(v,c) = (double)W(v,c) * (double)2m  -  (gamma * (double)tot'_c) * (double)k_v
// three multiplications and a subtraction, evaluated without FMA contraction exactly as the host and the Python
// specification do; ties go to the smaller community id, so the lane order inside a wave does not matter either.
#include "ddx_prims.h"

#include <algorithm>

#include "ddx_internal.h"

namespace ddx {

#pragma clang fp contract(off)

constexpr double kWeightScale = 1048576.0;   // 2^20, as in louvain.cpp / louvain_ref.py
constexpr int kLvCap = 4096;                 // neighbours of one node handled by the LDS path (one wave per workgroup there)

__global__ void k_lv_quantise(const double* __restrict__ w, int64_t E, int64_t* __restrict__ wq) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) wq[e] = (int64_t)rint(w[e] * kWeightScale);
}

// strength of every node (self loops included), identity communities, 2m, largest degree
__global__ void k_lv_strength(const int64_t* __restrict__ indptr, const int64_t* __restrict__ wq, int64_t n, int64_t* __restrict__ K,
                              int32_t* __restrict__ comm /* null: leave the partition alone */, unsigned long long* __restrict__ m2,
                              int32_t* __restrict__ maxdeg, int32_t* __restrict__ nbig, int32_t* __restrict__ big_list) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    int64_t s = 0;
    const int64_t b = indptr[v], e = indptr[v + 1];
    for (int64_t p = b; p < e; ++p) s += wq[p];
    K[v] = s;
    if (comm) comm[v] = (int32_t)v;
    atomicAdd(m2, (unsigned long long)s);
    atomicMax(maxdeg, (int32_t)(e - b));
    if (e - b > 64) big_list[atomicAdd(nbig, 1)] = (int32_t)v;     // handled by the LDS variant of the sweep (any order)
}

__global__ void k_lv_totals(const int32_t* __restrict__ comm, const int64_t* __restrict__ K, int64_t n,
                            unsigned long long* __restrict__ tot, int32_t* __restrict__ size) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    atomicAdd(tot + comm[v], (unsigned long long)K[v]);
    atomicAdd(size + comm[v], 1);
}

__device__ __forceinline__ int64_t shfl64(int64_t v, int src) {
    const int lo = __shfl((int)v, src, 64), hi = __shfl((int)(v >> 32), src, 64);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}

// wave-wide argmax of (score, smaller community wins ties); c < 0 marks "no candidate"
__device__ __forceinline__ void wave_best(double& s, int32_t& c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double s2 = __shfl_xor(s, off, 64);
        const int32_t c2 = __shfl_xor(c, off, 64);
        const bool take = c2 >= 0 && (c < 0 || s2 > s || (s2 == s && c2 < c));
        if (take) { s = s2; c = c2; }
    }
}

// One synchronous sweep: one wave per node decides from (comm, tot, size) and writes next[v].  Nodes with at most 64
// neighbours (nearly all) are handled in registers by the BIG = false instance, which needs no LDS and so keeps the
// CU full of waves (the work is a chain of dependent loads); the few others are listed in big_list and go through LDS.
template <bool BIG>
__global__ void __launch_bounds__(256) k_lv_sweep(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                  const int64_t* __restrict__ wq, const int64_t* __restrict__ K,
                                                  const int32_t* __restrict__ comm, const unsigned long long* __restrict__ tot,
                                                  const int32_t* __restrict__ size, int64_t n, double gamma, double m2d,
                                                  const int32_t* __restrict__ big_list, int32_t* __restrict__ next,
                                                  unsigned long long* __restrict__ tot_clear, int32_t* __restrict__ size_clear,
                                                  int cls_shift /* sweep number */, int cls_mod /* sub-rounds */, int cls_now /* this sub-round */) {
    __shared__ int32_t cS[1][BIG ? kLvCap : 1];
    __shared__ int64_t wS[1][BIG ? kLvCap : 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t v = BIG ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wave;        // BIG: 64-thread workgroups
    if (v >= n) return;
    // the totals of the NEXT sweep are accumulated into the other pair of arrays: clear entry v of it here (this
    // instance visits every node id once), which replaces two memset launches per sweep
    if (!BIG && lane == 0 && tot_clear) { tot_clear[v] = 0ull; size_clear[v] = 0; }
    if (BIG) v = big_list[v];
    const int64_t b = indptr[v];
    const int deg = (int)(indptr[v + 1] - b);
    if (!BIG && deg > 64) return;
    const int32_t own = comm[v];
    // sub-rounds: only the nodes of this sub-round's class decide, everybody else stays where it is
    if ((int)(((((uint32_t)v * 2654435761u) >> 16) + (uint32_t)cls_shift) % (uint32_t)cls_mod) != cls_now) {
        if (lane == 0) next[v] = own;
        return;
    }
    const int64_t kvi = K[v];
    const double kv = (double)kvi;
    double best_s = 0.0;
    int32_t best_c = -1;
    int64_t w_own = 0;
    if (!BIG) {
        int32_t c = -1;
        int64_t w = 0;
        if (lane < deg) {
            const int32_t u = cols[b + lane];
            w = wq[b + lane];
            c = (u == (int32_t)v) ? -1 : comm[u];
        }
        int64_t W = 0;
        bool leader = c >= 0;
        for (int j = 0; j < deg; ++j) {
            const int32_t cj = __shfl(c, j, 64);
            const int64_t wj = shfl64(w, j);
            if (cj == c) {
                W += wj;
                if (j < lane) leader = false;
            }
            if (cj == own) w_own += wj;
        }
        if (leader && c != own) {
            best_s = (double)W * m2d - (gamma * (double)(int64_t)tot[c]) * kv;
            best_c = c;
        }
    } else {
        // LDS path: sort the (community, weight) pairs of the node by community (wave-wide bitonic sort), then every
        // run of equal communities is summed by the lane that finds its first element.  O(d log^2 d / 64) per lane.
        const int d = deg < kLvCap ? deg : kLvCap;       // deg > kLvCap is rejected on the host before the launch
        int P = 64;
        while (P < d) P <<= 1;
        for (int i = lane; i < P; i += 64) {
            int32_t c = 0x7fffffff;                      // padding sorts last
            int64_t wv = 0;
            if (i < d) {
                const int32_t u = cols[b + i];
                c = (u == (int32_t)v) ? 0x7ffffffe : comm[u];     // self loops: a class of their own, ignored below
                wv = wq[b + i];
            }
            cS[0][i] = c;
            wS[0][i] = wv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = lane; t < (P >> 1); t += 64) {
                    const int lo = ((t / stride) * stride * 2) + (t % stride);
                    const int hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const int32_t cl = cS[0][lo], ch = cS[0][hi];
                    if ((cl > ch) == up) {
                        const int64_t wl = wS[0][lo], wh = wS[0][hi];
                        cS[0][lo] = ch; cS[0][hi] = cl;
                        wS[0][lo] = wh; wS[0][hi] = wl;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        for (int i = lane; i < d; i += 64) {
            const int32_t c = cS[0][i];
            if (c >= 0x7ffffffe) continue;               // self loops / padding
            if (i > 0 && cS[0][i - 1] == c) continue;    // not the first element of its run
            int64_t W = 0;
            for (int j = i; j < d && cS[0][j] == c; ++j) W += wS[0][j];
            if (c == own) {
                w_own = W;
            } else {
                const double s = (double)W * m2d - (gamma * (double)(int64_t)tot[c]) * kv;
                if (best_c < 0 || s > best_s || (s == best_s && c < best_c)) { best_s = s; best_c = c; }
            }
        }
        // the lane that led the own community holds w_own; everybody else 0
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) w_own += shfl64(w_own, lane ^ off);
    }
    wave_best(best_s, best_c);
    if (lane == 0) {
        const double own_score = (double)w_own * m2d - (gamma * (double)((int64_t)tot[own] - kvi)) * kv;
        int32_t target = own;
        if (best_c >= 0 && best_s > own_score && !(size[own] == 1 && size[best_c] == 1 && best_c > own)) target = best_c;
        next[v] = target;
    }
}

__global__ void k_lv_used(const int32_t* __restrict__ comm, int64_t n, int32_t* __restrict__ used) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) used[comm[v]] = 1;
}

__global__ void k_lv_member(const int32_t* __restrict__ comm, const int32_t* __restrict__ renum, int64_t n, int32_t* __restrict__ member) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) member[v] = renum[comm[v]];
}

__global__ void __launch_bounds__(256) k_lv_edge_keys(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                      const int32_t* __restrict__ member, int64_t n, int shift, uint64_t* __restrict__ keys) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    const uint64_t hi = (uint64_t)member[v] << shift;
    for (int64_t p = indptr[v] + lane; p < indptr[v + 1]; p += 64) keys[p] = hi | (uint32_t)member[cols[p]];
}

__global__ void k_lv_rowptr(const uint64_t* __restrict__ keys, int64_t n, int64_t rows, int shift, int64_t* __restrict__ indptr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(keys[mid] >> shift) < r) lo = mid + 1; else hi = mid;
    }
    indptr[r] = lo;
}

__global__ void k_lv_unpack(const uint64_t* __restrict__ keys, const int64_t* __restrict__ sums, int64_t n, int shift,
                            int32_t* __restrict__ cols, double* __restrict__ w) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    cols[t] = (int32_t)(keys[t] & ((1ull << shift) - 1ull));
    w[t] = (double)sums[t] / kWeightScale;
}

__global__ void k_lv_compose(const int32_t* __restrict__ first, const int32_t* __restrict__ second, int64_t n, int32_t* __restrict__ out) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) out[v] = second[first[v]];
}

struct LvGraph {            // a CSR on the device
    int64_t n = 0, E = 0;
    const int64_t* indptr = nullptr;
    const int32_t* cols = nullptr;
    const double* w = nullptr;
};

struct LvScratch {          // sized for the finest level, reused by the coarser ones
    int64_t *wq, *vals_b, *sums, *K;
    uint64_t *keys_a, *keys_b;
    unsigned long long *tot, *tot2, *scal;
    int32_t *comm, *next, *size, *size2, *used, *renum, *big_list;
};

constexpr int kSubrounds = DDX_SUBROUNDS;

// `sweeps` sweeps of `subrounds` synchronous sub-rounds from the partition in sc.comm; `cur` receives the array
// (sc.comm or sc.next) that holds the result.  Every sub-round recomputes the community totals from scratch: they
// ping-pong between two pairs of arrays, sub-round t reads pair t & 1 and clears the other one for sub-round t + 1
// (tot | tot2 and size | size2 are adjacent pieces of the scratch buffer: one memset each clears both before the first).
static int lv_sweeps(ddx_ctx* ctx, const LvGraph& in, double gamma, int32_t sweeps, int subrounds, const LvScratch& sc, int64_t m2,
                     int32_t nbig, int32_t*& cur) {
    const int64_t n = in.n;
    hipStream_t st = ctx->stream;
    cur = sc.comm;
    int32_t* nxt = sc.next;
    if (sweeps <= 0 || m2 <= 0) return DDX_OK;
    DDX_HIP(ctx, hipMemsetAsync(sc.tot, 0, (size_t)(reinterpret_cast<unsigned char*>(sc.tot2) - reinterpret_cast<unsigned char*>(sc.tot)) + sizeof(int64_t) * n, st));
    DDX_HIP(ctx, hipMemsetAsync(sc.size, 0, (size_t)(reinterpret_cast<unsigned char*>(sc.size2) - reinterpret_cast<unsigned char*>(sc.size)) + sizeof(int32_t) * n, st));
    int t = 0;
    for (int s = 0; s < sweeps; ++s) {
        for (int r = 0; r < subrounds; ++r, ++t) {
            unsigned long long* tot = (t & 1) ? sc.tot2 : sc.tot;
            unsigned long long* tot_other = (t & 1) ? sc.tot : sc.tot2;
            int32_t* size = (t & 1) ? sc.size2 : sc.size;
            int32_t* size_other = (t & 1) ? sc.size : sc.size2;
            k_lv_totals<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(cur, sc.K, n, tot, size);
            k_lv_sweep<false><<<(unsigned)ceil_div(n, 4), 256, 0, st>>>(in.indptr, in.cols, sc.wq, sc.K, cur, tot, size, n, gamma, (double)m2, sc.big_list, nxt,
                                                                        tot_other, size_other, s, subrounds, r);
            if (nbig > 0)
                k_lv_sweep<true><<<(unsigned)nbig, 64, 0, st>>>(in.indptr, in.cols, sc.wq, sc.K, cur, tot, size, nbig, gamma, (double)m2, sc.big_list, nxt,
                                                                nullptr, nullptr, s, subrounds, r);
            std::swap(cur, nxt);      // a sweep that moves nothing reproduces its input, so running all of them equals stopping early
        }
    }
    return DDX_OK;
}

// one level: `sweeps` synchronous sweeps on `in`, exact aggregation into (member, out)
static int coarsen_level(ddx_ctx* ctx, const LvGraph& in, double gamma, int32_t sweeps, const LvScratch& sc, int32_t* member,
                         int64_t* c_indptr, int32_t* c_cols, double* c_w, LvGraph& out) {
    const int64_t n = in.n, E = in.E;
    hipStream_t st = ctx->stream;
    DDX_HIP(ctx, hipMemsetAsync(sc.scal, 0, 256, st));
    if (E > 0) k_lv_quantise<<<(unsigned)ceil_div(E, 256), 256, 0, st>>>(in.w, E, sc.wq);
    k_lv_strength<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(in.indptr, sc.wq, n, sc.K, sc.comm, sc.scal, reinterpret_cast<int32_t*>(sc.scal + 1),
                                                              reinterpret_cast<int32_t*>(sc.scal + 1) + 1, sc.big_list);
    unsigned long long h_scal[2] = {0, 0};
    DDX_HIP(ctx, hipMemcpyAsync(h_scal, sc.scal, sizeof(h_scal), hipMemcpyDeviceToHost, st));
    DDX_HIP(ctx, hipStreamSynchronize(st));
    const int64_t m2 = (int64_t)h_scal[0];
    const int32_t maxdeg = (int32_t)(h_scal[1] & 0xffffffffull);
    const int32_t nbig = (int32_t)(h_scal[1] >> 32);
    if (maxdeg > kLvCap) return set_err(ctx, DDX_E_UNSUPPORTED, "a node with %d neighbours exceeds the device sweep's capacity (%d)", maxdeg, kLvCap);
    int32_t* cur = sc.comm;
    DDX_TRY(lv_sweeps(ctx, in, gamma, sweeps, kSubrounds, sc, m2, nbig, cur));
    // renumber the surviving communities by ascending id
    DDX_HIP(ctx, hipMemsetAsync(sc.used, 0, sizeof(int32_t) * (n + 1), st));
    k_lv_used<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(cur, n, sc.used);
    size_t tmp_scan = 0, tmp_sort = 0, tmp_red = 0;
    DDX_HIP(ctx, prim::exclusive_sum(nullptr, tmp_scan, sc.used, sc.renum, (int)n + 1, st));
    int shift = 1;                                   // keys: coarse row << shift | coarse column, shift = bits(n)
    while (((int64_t)1 << shift) < n) ++shift;
    const int end_bit = 2 * shift;
    int64_t* runs_d = reinterpret_cast<int64_t*>(sc.scal + 2);
    if (E > 0) {
        DDX_HIP(ctx, prim::sort_pairs(nullptr, tmp_sort, sc.keys_a, sc.keys_b, sc.wq, sc.vals_b, (int)E, 0, end_bit, st));
        DDX_HIP(ctx, prim::reduce_by_key_sum(nullptr, tmp_red, sc.keys_b, sc.keys_a, sc.vals_b, sc.sums, runs_d, (size_t)E, st));
    }
    DDX_TRY(ensure(ctx, ctx->sort_tmp, std::max(tmp_scan, std::max(tmp_sort, tmp_red))));
    DDX_HIP(ctx, prim::exclusive_sum(ctx->sort_tmp.p, tmp_scan, sc.used, sc.renum, (int)n + 1, st));
    k_lv_member<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(cur, sc.renum, n, member);
    int64_t runs = 0;
    if (E > 0) {
        k_lv_edge_keys<<<(unsigned)ceil_div(n, 4), 256, 0, st>>>(in.indptr, in.cols, member, n, shift, sc.keys_a);
        DDX_HIP(ctx, prim::sort_pairs(ctx->sort_tmp.p, tmp_sort, sc.keys_a, sc.keys_b, sc.wq, sc.vals_b, (int)E, 0, end_bit, st));
        DDX_HIP(ctx, prim::reduce_by_key_sum(ctx->sort_tmp.p, tmp_red, sc.keys_b, sc.keys_a, sc.vals_b, sc.sums, runs_d, (size_t)E, st));
    }
    int32_t nc = 0;
    DDX_HIP(ctx, hipMemcpyAsync(&nc, sc.renum + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DDX_HIP(ctx, hipMemcpyAsync(&runs, runs_d, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    DDX_HIP(ctx, hipStreamSynchronize(st));
    if (E == 0) runs = 0;
    if (runs > 0) k_lv_unpack<<<(unsigned)ceil_div(runs, 256), 256, 0, st>>>(sc.keys_a, sc.sums, runs, shift, c_cols, c_w);
    k_lv_rowptr<<<(unsigned)ceil_div((int64_t)nc + 1, 256), 256, 0, st>>>(sc.keys_a, runs, nc, shift, c_indptr);
    DDX_HIP(ctx, hipGetLastError());
    out.n = nc;
    out.E = runs;
    out.indptr = c_indptr;
    out.cols = c_cols;
    out.w = c_w;
    return DDX_OK;
}

// result of the last level packed for one device-to-host copy: w f64[E] | indptr i64[nc+1] | member i32[n] | cols i32[E]
__global__ void k_lv_pack(const double* __restrict__ w, const int64_t* __restrict__ indptr, const int32_t* __restrict__ member,
                          const int32_t* __restrict__ cols, int64_t E, int64_t nc, int64_t n, unsigned char* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double* ow = reinterpret_cast<double*>(out);
    int64_t* oi = reinterpret_cast<int64_t*>(ow + E);
    int32_t* om = reinterpret_cast<int32_t*>(oi + nc + 1);
    int32_t* oc = om + n;
    if (t < E) { ow[t] = w[t]; oc[t] = cols[t]; }
    if (t <= nc) oi[t] = indptr[t];
    if (t < n) om[t] = member[t];
}

struct LvSets {             // two output sets the levels write alternately + the composed member tables
    int32_t* member[2];
    int64_t* indptr[2];
    int32_t* cols[2];
    double* w[2];
    int32_t *total_a, *total_b;
};

// Layout of the work buffer for a graph of n nodes / E entries: scratch (wq, keys x2, vals, sums: E each; K, tot x2: n;
// comm, next, size x2, used, renum, big_list: n) + the output sets.  Every piece is rounded up to 256 bytes; the buffer
// is sized from the very arithmetic that carves it.  Returns the bytes needed; binds the pointers when base != null.
static size_t lv_bind(unsigned char* base, int64_t n, int64_t E, LvScratch& sc, LvSets& o) {
    size_t bytes = 0;
    auto piece = [&](size_t sz) { const size_t at = bytes; bytes += (sz + 255) & ~(size_t)255; return base ? base + at : nullptr; };
    sc.wq = reinterpret_cast<int64_t*>(piece(sizeof(int64_t) * E));
    sc.keys_a = reinterpret_cast<uint64_t*>(piece(sizeof(uint64_t) * E));
    sc.keys_b = reinterpret_cast<uint64_t*>(piece(sizeof(uint64_t) * E));
    sc.vals_b = reinterpret_cast<int64_t*>(piece(sizeof(int64_t) * E));
    sc.sums = reinterpret_cast<int64_t*>(piece(sizeof(int64_t) * E));
    sc.K = reinterpret_cast<int64_t*>(piece(sizeof(int64_t) * n));
    sc.tot = reinterpret_cast<unsigned long long*>(piece(sizeof(int64_t) * n));
    sc.tot2 = reinterpret_cast<unsigned long long*>(piece(sizeof(int64_t) * n));
    sc.comm = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    sc.next = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    sc.size = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    sc.size2 = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    sc.used = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * (n + 1)));
    sc.renum = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * (n + 1)));
    sc.big_list = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    sc.scal = reinterpret_cast<unsigned long long*>(piece(256));        // [0] = 2m, [1] = max degree | #big nodes, [2] = runs
    for (int i = 0; i < 2; ++i) {
        o.member[i] = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
        o.indptr[i] = reinterpret_cast<int64_t*>(piece(sizeof(int64_t) * (n + 1)));
        o.cols[i] = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * E));
        o.w[i] = reinterpret_cast<double*>(piece(sizeof(double) * E));
    }
    o.total_a = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    o.total_b = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    return bytes;
}

int stage_coarsen_graph(ddx_ctx* ctx, double gamma, int32_t sweeps, int32_t levels) {
    const int64_t n = ctx->g_nodes;
    const int64_t E = ctx->g_entries;
    ctx->c_nodes = -1;
    ctx->lv_host_valid = false;
    LvScratch sc;
    LvSets sets;
    DDX_TRY(ensure(ctx, ctx->lv_buf, lv_bind(nullptr, n, E, sc, sets)));
    lv_bind(ctx->lv_buf.as<unsigned char>(), n, E, sc, sets);
    int32_t** member_set = sets.member;
    int64_t** indptr_set = sets.indptr;
    int32_t** cols_set = sets.cols;
    double** w_set = sets.w;
    int32_t* total_a = sets.total_a;
    int32_t* total_b = sets.total_b;
    ScopedTimer t(ctx, "graph_coarsen");
    LvGraph cur;
    cur.n = n; cur.E = E; cur.indptr = ctx->g_d_indptr; cur.cols = ctx->g_d_cols; cur.w = ctx->g_d_vals;
    const int32_t* total = nullptr;          // member of every original node in the current coarse graph
    ctx->lv_levels = levels;
    for (int lvl = 0; lvl < levels; ++lvl) {
        const int o = lvl & 1;
        LvGraph nextg;
        if (lvl < ddx_ctx::kLvKeep) {           // (two output sets: the graphs of levels 0 and 1 survive two levels of part A)
            ctx->lv_n[lvl] = cur.n; ctx->lv_E[lvl] = cur.E;
            ctx->lv_indptr[lvl] = cur.indptr; ctx->lv_cols[lvl] = cur.cols; ctx->lv_w[lvl] = cur.w;
            ctx->lv_member[lvl] = member_set[o];
        }
        DDX_TRY(coarsen_level(ctx, cur, gamma, sweeps, sc, member_set[o], indptr_set[o], cols_set[o], w_set[o], nextg));
        if (!total) {
            total = member_set[o];
        } else {
            int32_t* dst = (total == total_a) ? total_b : total_a;
            k_lv_compose<<<(unsigned)ceil_div(n, 256), 256, 0, ctx->stream>>>(total, member_set[o], n, dst);
            total = dst;
        }
        cur = nextg;
    }
    DDX_HIP(ctx, hipGetLastError());
    if (!total) return set_err(ctx, DDX_E_ARG, "levels must be >= 1");
    ctx->c_nodes = cur.n;
    ctx->c_entries = cur.E;
    ctx->c_d_member = total;
    ctx->c_d_indptr = cur.indptr;
    ctx->c_d_cols = cur.cols;
    ctx->c_d_vals = cur.w;
    // one packed copy to pinned host memory instead of four small ones (each costs a round trip)
    const size_t packed = 8 * (size_t)cur.E + 8 * (size_t)(cur.n + 1) + 4 * (size_t)n + 4 * (size_t)cur.E;
    DDX_TRY(ensure(ctx, ctx->lv_pack, packed + 64));
    if (packed > ctx->lv_host_cap) {
        if (ctx->lv_host) (void)hipHostFree(ctx->lv_host);
        ctx->lv_host = nullptr;
        ctx->lv_host_cap = 0;
        DDX_HIP(ctx, hipHostMalloc(&ctx->lv_host, packed * 2 + 4096, hipHostMallocDefault));
        ctx->lv_host_cap = packed * 2 + 4096;
    }
    const int64_t span = std::max<int64_t>(std::max<int64_t>(cur.E, cur.n + 1), n);
    k_lv_pack<<<(unsigned)ceil_div(span, 256), 256, 0, ctx->stream>>>(cur.w, cur.indptr, total, cur.cols, cur.E, cur.n, n, ctx->lv_pack.as<unsigned char>());
    DDX_HIP(ctx, hipMemcpyAsync(ctx->lv_host, ctx->lv_pack.p, packed, hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, hipGetLastError());
    ctx->lv_host_valid = true;
    return DDX_OK;
}

// ---- part C on the device ----------------------------------------------------------------------------------------------
// lab[v] = coarse_labels[member[v]]; first[l] = smallest v carrying label l
__global__ void k_lv_project(const int32_t* __restrict__ member, const int32_t* __restrict__ coarse_labels, int64_t n, int32_t* __restrict__ lab,
                             int32_t* __restrict__ first) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int32_t l = coarse_labels[member[v]];
    lab[v] = l;
    atomicMin(first + l, (int32_t)v);
}

__global__ void k_lv_name(const int32_t* __restrict__ lab, const int32_t* __restrict__ first, int64_t n, int32_t* __restrict__ comm) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) comm[v] = first[lab[v]];
}

int stage_refine_communities(ddx_ctx* ctx, const int32_t* coarse_labels, double gamma, int32_t sweeps, int32_t* labels_out) {
    const int64_t n = ctx->g_nodes, E = ctx->g_entries, nc = ctx->c_nodes;
    hipStream_t st = ctx->stream;
    LvScratch sc;
    LvSets sets;
    if (lv_bind(nullptr, n, E, sc, sets) > ctx->lv_buf.cap) return set_err(ctx, DDX_E_ARG, "the coarsening work space is gone");
    if (ctx->lv_levels < 1 || ctx->lv_levels > ddx_ctx::kLvKeep)
        return set_err(ctx, DDX_E_UNSUPPORTED, "part C on the device follows at most %d levels of part A (%d were run)", ddx_ctx::kLvKeep, ctx->lv_levels);
    lv_bind(ctx->lv_buf.as<unsigned char>(), n, E, sc, sets);
    for (int64_t c = 0; c < nc; ++c)
        if (coarse_labels[c] < 0 || coarse_labels[c] >= nc) return set_err(ctx, DDX_E_ARG, "coarse label %d out of range at %lld", coarse_labels[c], (long long)c);
    ScopedTimer t(ctx, "graph_refine");
    // the scratch of part A is free again: labels of the level above -> renum, smallest member per label -> used
    DDX_HIP(ctx, hipMemcpyAsync(sc.renum, coarse_labels, sizeof(int32_t) * nc, hipMemcpyHostToDevice, st));
    for (int level = ctx->lv_levels - 1; level >= 0; --level) {
        LvGraph g;
        g.n = ctx->lv_n[level]; g.E = ctx->lv_E[level]; g.indptr = ctx->lv_indptr[level]; g.cols = ctx->lv_cols[level]; g.w = ctx->lv_w[level];
        DDX_HIP(ctx, hipMemsetAsync(sc.used, 0x7f, sizeof(int32_t) * (g.n + 1), st));
        DDX_HIP(ctx, hipMemsetAsync(sc.scal, 0, 256, st));
        k_lv_project<<<(unsigned)ceil_div(g.n, 256), 256, 0, st>>>(ctx->lv_member[level], sc.renum, g.n, sc.next, sc.used);
        k_lv_name<<<(unsigned)ceil_div(g.n, 256), 256, 0, st>>>(sc.next, sc.used, g.n, sc.comm);
        if (g.E > 0) k_lv_quantise<<<(unsigned)ceil_div(g.E, 256), 256, 0, st>>>(g.w, g.E, sc.wq);
        k_lv_strength<<<(unsigned)ceil_div(g.n, 256), 256, 0, st>>>(g.indptr, sc.wq, g.n, sc.K, nullptr, sc.scal, reinterpret_cast<int32_t*>(sc.scal + 1),
                                                                    reinterpret_cast<int32_t*>(sc.scal + 1) + 1, sc.big_list);
        unsigned long long h_scal[2] = {0, 0};
        DDX_HIP(ctx, hipMemcpyAsync(h_scal, sc.scal, sizeof(h_scal), hipMemcpyDeviceToHost, st));
        DDX_HIP(ctx, hipStreamSynchronize(st));
        const int64_t m2 = (int64_t)h_scal[0];
        const int32_t nbig = (int32_t)(h_scal[1] >> 32);
        int32_t* cur = sc.comm;
        DDX_TRY(lv_sweeps(ctx, g, gamma, sweeps, kSubrounds, sc, m2, nbig, cur));
        if (level > 0) DDX_HIP(ctx, hipMemcpyAsync(sc.renum, cur, sizeof(int32_t) * g.n, hipMemcpyDeviceToDevice, st));      // names of this level's communities are its node ids
        else DDX_HIP(ctx, hipMemcpyAsync(labels_out, cur, sizeof(int32_t) * g.n, hipMemcpyDeviceToHost, st));
    }
    DDX_HIP(ctx, hipStreamSynchronize(st));
    DDX_HIP(ctx, hipGetLastError());
    // canonical numbering: by ascending smallest member = order of first appearance
    std::vector<int32_t> rank((size_t)n, -1);
    int32_t k = 0;
    for (int64_t v = 0; v < n; ++v) {
        int32_t& r = rank[labels_out[v]];
        if (r < 0) r = k++;
        labels_out[v] = r;
    }
    return DDX_OK;
}

}  // namespace ddx
