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
#include <type_traits>

#include "ddx_internal.h"

namespace ddx {

#pragma clang fp contract(off)

constexpr double kWeightScale = 1048576.0;   // 2^20, as in louvain.cpp / louvain_ref.py
constexpr int kLvCap = 4096;                 // neighbours of one node handled by the LDS path (one wave per workgroup there)

__global__ void k_lv_quantise(const double* __restrict__ w, int64_t E, int64_t* __restrict__ wq) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) wq[e] = (int64_t)rint(w[e] * kWeightScale);
}

// strength of every node (self loops included); every node its own community (comm = id, total = strength, one member);
// 2m, the largest degree and the list of the nodes with more than 64 neighbours (those take the hash-table sweep).
// scal: [0] = 2m, [1] = max degree (low 32 bits) | number of big nodes (high 32 bits), [3] = largest edge weight
__global__ void __launch_bounds__(256) k_lv_strength(const int64_t* __restrict__ indptr, const int64_t* __restrict__ wq, int64_t n, int64_t* __restrict__ K,
                                                     int32_t* __restrict__ comm, unsigned long long* __restrict__ tot, int32_t* __restrict__ size,
                                                     unsigned long long* __restrict__ scal, int32_t* __restrict__ big_list) {
    // 16 lanes per node (consecutive entries of a row on consecutive lanes); a workgroup takes 256 nodes in 16 passes and
    // ends with one atomic per wave on each shared scalar (a few thousand per launch: they all hit the same three words)
    const int sub = threadIdx.x & 15;
    auto xor64 = [](int64_t x, int off) { return ((int64_t)__shfl_xor((int)(x >> 32), off, 64) << 32) | (uint32_t)__shfl_xor((int)x, off, 64); };
    int64_t ws = 0, wm = 0;
    int wd = 0;
    for (int pass = 0; pass < 16; ++pass) {
        const int64_t v = (int64_t)blockIdx.x * 256 + pass * 16 + (threadIdx.x >> 4);
        int64_t s = 0, wmax = 0;
        int deg = 0;
        if (v < n) {
            const int64_t b = indptr[v], e = indptr[v + 1];
            for (int64_t p = b + sub; p < e; p += 16) { const int64_t w = wq[p]; s += w; wmax = w > wmax ? w : wmax; }
            deg = (int)(e - b);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += xor64(s, off);      // the node's 16 lanes
        if (v < n && sub == 0) {
            K[v] = s;
            comm[v] = (int32_t)v;
            tot[v] = (unsigned long long)s;
            size[v] = 1;
            if (deg > 64) big_list[atomicAdd(reinterpret_cast<int32_t*>(scal + 1) + 1, 1)] = (int32_t)v;     // (any order)
            ws += s;
        }
        wd = deg > wd ? deg : wd;
        wm = wmax > wm ? wmax : wm;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ws += xor64(ws, off);
        const int od = __shfl_xor(wd, off, 64);
        wd = od > wd ? od : wd;
        const int64_t om = xor64(wm, off);
        wm = om > wm ? om : wm;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(scal, (unsigned long long)ws);
        atomicMax(reinterpret_cast<int32_t*>(scal + 1), wd);
        atomicMax(scal + 3, (unsigned long long)wm);
    }
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

// One synchronous sub-round for the nodes of one class, v = first + step * i (class = v mod step): one wave per node
// decides from (comm, tot, size) and writes next[v]; k_lv_apply then carries the moves out.  Nodes with at most 64
// neighbours (nearly all on the original graph) are handled here in registers, which needs no LDS and so keeps the CU full
// of waves (the work is a chain of dependent loads); the others are listed in big_list and go through k_lv_sweep_big.
// Lane l holds neighbour l; W(v, c) is gathered neighbour by neighbour (readlane + one masked add), so every lane of a
// community ends with the same W and the same score -- duplicates change nothing in the arg-max, and the weight to the
// own community is read from the first lane that holds it.  W32: every quantised edge weight of the level is below 2^25
// (always on the original graph, whose weights are at most 1), so 64 of them sum in 32 bits.
// A wave decides kLvPerWave nodes, the loads of all of them issued before any is used (the kernel is a chain of four
// dependent fetches: row bounds -> neighbours -> their communities -> those communities' totals).  Measured at the headline
// size (62 500 nodes per launch): 1 node 32.2 us, 2 nodes 33.9 us -- the gathers, not the latency of the chain, set the pace.
constexpr int kLvPerWave = 1;
template <bool W32>
__global__ void __launch_bounds__(256) k_lv_sweep(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                  const int64_t* __restrict__ wq, const int64_t* __restrict__ K,
                                                  const int32_t* __restrict__ comm, const unsigned long long* __restrict__ tot,
                                                  const int32_t* __restrict__ size, int64_t n, double gamma, double m2d,
                                                  int first, int step, int32_t* __restrict__ next) {
    typedef typename std::conditional<W32, uint32_t, int64_t>::type acc_t;
    constexpr int NV = kLvPerWave;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t slot0 = ((int64_t)blockIdx.x * 4 + wave) * NV;
    int64_t v[NV], b[NV], kvi[NV];
    int deg[NV];
    int32_t own[NV], c[NV], size_own[NV], size_c[NV];
    unsigned long long tot_own[NV], tot_c[NV];
    acc_t w[NV];
    bool live[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = first + (int64_t)step * (slot0 + i);
        live[i] = v[i] < n;
        const int64_t vv = live[i] ? v[i] : 0;
        b[i] = indptr[vv];
        deg[i] = (int)(indptr[vv + 1] - b[i]);
        own[i] = comm[vv];
        kvi[i] = K[vv];
        live[i] = live[i] && deg[i] <= 64;          // (the others are k_lv_sweep_big's)
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        tot_own[i] = tot[own[i]];
        size_own[i] = size[own[i]];
        c[i] = -1;
        w[i] = 0;
        if (live[i] && lane < deg[i]) {
            const int32_t u = cols[b[i] + lane];
            w[i] = (acc_t)wq[b[i] + lane];
            c[i] = (u == (int32_t)v[i]) ? -1 : comm[u];
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        tot_c[i] = c[i] >= 0 ? tot[c[i]] : 0ull;
        size_c[i] = c[i] >= 0 ? size[c[i]] : 0;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (!live[i]) continue;                      // (wave-uniform)
        const double kv = (double)kvi[i];
        acc_t W = 0;
        // four neighbours per trip (the loop's own three scalar instructions were a third of a trip of one); lanes past the degree hold
        // c = -1 and w = 0, so running up to three lanes over changes no sum that is read
        for (int j0 = 0; j0 < deg[i]; j0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = (j0 + u) & 63;
                const int32_t cj = __builtin_amdgcn_readlane(c[i], j);
                acc_t wj;
                if (W32) wj = (acc_t)__builtin_amdgcn_readlane((int)w[i], j);
                else wj = (acc_t)(((int64_t)__builtin_amdgcn_readlane((int)((int64_t)w[i] >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(int64_t)w[i], j));
                W += (cj == c[i]) ? wj : (acc_t)0;
            }
        }
        double best_s = 0.0;
        int32_t best_c = -1;
        if (c[i] >= 0 && c[i] != own[i]) {
            best_s = (double)(int64_t)W * m2d - (gamma * (double)(int64_t)tot_c[i]) * kv;
            best_c = c[i];
        }
        const unsigned long long own_lanes = __ballot(c[i] == own[i]);      // (own >= 0; the self loop carries c = -1)
        int64_t w_own = 0;
        if (own_lanes) {
            const int src = __ffsll((long long)own_lanes) - 1;
            if (W32) w_own = (int64_t)(uint32_t)__shfl((int)W, src, 64);
            else w_own = shfl64((int64_t)W, src);
        }
        wave_best(best_s, best_c);
        int32_t size_best = 0;
        if (best_c >= 0) size_best = __shfl(size_c[i], __ffsll((long long)__ballot(c[i] == best_c)) - 1, 64);
        if (lane == 0) {
            const double own_score = (double)w_own * m2d - (gamma * (double)((int64_t)tot_own[i] - kvi[i])) * kv;
            int32_t target = own[i];
            if (best_c >= 0 && best_s > own_score && !(size_own[i] == 1 && size_best == 1 && best_c > own[i])) target = best_c;
            next[v[i]] = target;
        }
    }
}

// The same decision for a node with more than 64 neighbours (coarse levels, hubs): one wave per listed node of the class
// groups the neighbours' communities in an LDS hash table (integer sums: the insertion order does not matter), then the
// lanes scan the table.  slots = power of two >= 2 * (largest degree of the level), 12 bytes each.
__global__ void __launch_bounds__(64) k_lv_sweep_big(const int64_t* __restrict__ indptr, const int32_t* __restrict__ cols,
                                                     const int64_t* __restrict__ wq, const int64_t* __restrict__ K,
                                                     const int32_t* __restrict__ comm, const unsigned long long* __restrict__ tot,
                                                     const int32_t* __restrict__ size, double gamma, double m2d, int first, int step,
                                                     const int32_t* __restrict__ big_list, int slots, int32_t* __restrict__ next) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lv_smem[];
    unsigned long long* wS = reinterpret_cast<unsigned long long*>(lv_smem);
    int32_t* cS = reinterpret_cast<int32_t*>(wS + slots);
    const int lane = threadIdx.x;
    const int64_t v = big_list[blockIdx.x];
    if ((int)(v % step) != first) return;
    for (int i = lane; i < slots; i += 64) { cS[i] = -1; wS[i] = 0ull; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t b = indptr[v];
    const int deg = (int)(indptr[v + 1] - b);
    const int32_t own = comm[v];
    const unsigned mask = (unsigned)slots - 1u;
    for (int i = lane; i < deg; i += 64) {
        const int32_t u = cols[b + i];
        if (u == (int32_t)v) continue;
        const int32_t c = comm[u];
        unsigned h = ((unsigned)c * 2654435761u) & mask;
        for (;;) {
            const int32_t old = atomicCAS(&cS[h], -1, c);
            if (old == -1 || old == c) { atomicAdd(&wS[h], (unsigned long long)wq[b + i]); break; }
            h = (h + 1u) & mask;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t kvi = K[v];
    const double kv = (double)kvi;
    double best_s = 0.0;
    int32_t best_c = -1;
    int64_t w_own = 0;
    for (int i = lane; i < slots; i += 64) {
        const int32_t c = cS[i];
        if (c < 0) continue;
        const int64_t W = (int64_t)wS[i];
        if (c == own) { w_own = W; continue; }
        const double sc = (double)W * m2d - (gamma * (double)(int64_t)tot[c]) * kv;
        if (best_c < 0 || sc > best_s || (sc == best_s && c < best_c)) { best_s = sc; best_c = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w_own += shfl64(w_own, lane ^ off);      // one lane met the own community (or none)
    wave_best(best_s, best_c);
    if (lane == 0) {
        const double own_score = (double)w_own * m2d - (gamma * (double)((int64_t)tot[own] - kvi)) * kv;
        int32_t target = own;
        if (best_c >= 0 && best_s > own_score && !(size[own] == 1 && size[best_c] == 1 && best_c > own)) target = best_c;
        next[v] = target;
    }
}

// carry out the moves of a sub-round: the totals follow the movers (integer atomics, exact in any order), nobody else
__global__ void k_lv_apply(const int32_t* __restrict__ next, const int64_t* __restrict__ K, int64_t n, int first, int step,
                           int32_t* __restrict__ comm, unsigned long long* __restrict__ tot, int32_t* __restrict__ size) {
    const int64_t v = first + (int64_t)step * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= n) return;
    const int32_t t = next[v], o = comm[v];
    if (t == o) return;
    const unsigned long long k = (unsigned long long)K[v];
    atomicAdd(tot + t, k);
    atomicAdd(tot + o, 0ull - k);
    atomicAdd(size + t, 1);
    atomicAdd(size + o, -1);
    comm[v] = t;
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

constexpr int kLvKeep = ddx_ctx::kLvKeep;     // levels of part A whose graphs, strengths and member counts part C finds again

struct LvScratch {          // sized for the finest level, reused by the coarser ones
    int64_t *vals_b, *sums;
    uint64_t *keys_a, *keys_b;
    unsigned long long *tot, *scal;
    int32_t *comm, *next, *size, *used, *renum, *lab;
    // kept per level for part C (level l = the graph the (l+1)-th application of part A started from)
    int64_t* wq[kLvKeep];       // quantised weights of the level's graph
    int64_t* K[kLvKeep];        // node strengths
    int32_t* csize[kLvKeep];    // members of every node of the level above
    int32_t* big[kLvKeep];      // nodes with more than 64 neighbours
    int64_t* Ktop;              // strengths of the nodes of the last aggregated graph
};

struct LvSets {             // two output sets the levels write alternately + the composed member tables
    int32_t* member[2];
    int64_t* indptr[2];
    int32_t* cols[2];
    double* w[2];
    int32_t *total_a, *total_b;
};

// Layout of the work buffer for a graph of n nodes / E entries.  Every piece is rounded up to 256 bytes; the buffer is
// sized from the very arithmetic that carves it.  Returns the bytes needed; binds the pointers when base != null.
static size_t lv_bind(unsigned char* base, int64_t n, int64_t E, LvScratch& sc, LvSets& o) {
    size_t bytes = 0;
    auto piece = [&](size_t sz) { const size_t at = bytes; bytes += (sz + 255) & ~(size_t)255; return base ? base + at : nullptr; };
    sc.keys_a = reinterpret_cast<uint64_t*>(piece(sizeof(uint64_t) * E));
    sc.keys_b = reinterpret_cast<uint64_t*>(piece(sizeof(uint64_t) * E));
    sc.vals_b = reinterpret_cast<int64_t*>(piece(sizeof(int64_t) * E));
    sc.sums = reinterpret_cast<int64_t*>(piece(sizeof(int64_t) * E));
    sc.tot = reinterpret_cast<unsigned long long*>(piece(sizeof(int64_t) * n));
    sc.comm = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    sc.next = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    sc.size = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    sc.used = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * (n + 1)));
    sc.renum = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * (n + 1)));
    sc.lab = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    sc.scal = reinterpret_cast<unsigned long long*>(piece(256));        // [0] = 2m, [1] = max degree | #big nodes, [2] = runs
    for (int l = 0; l < kLvKeep; ++l) {
        sc.wq[l] = reinterpret_cast<int64_t*>(piece(sizeof(int64_t) * E));
        sc.K[l] = reinterpret_cast<int64_t*>(piece(sizeof(int64_t) * n));
        sc.csize[l] = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
        sc.big[l] = reinterpret_cast<int32_t*>(piece(sizeof(int32_t) * n));
    }
    sc.Ktop = reinterpret_cast<int64_t*>(piece(sizeof(int64_t) * n));
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

constexpr int kSubrounds = DDX_SUBROUNDS;

// `sweeps` sweeps of `subrounds` synchronous sub-rounds on `in` from the partition in sc.comm with its totals in sc.tot /
// sc.size (oracle/louvain_ref.py:_sync_sweeps): in sub-round r of sweep s the nodes with (v + s) mod subrounds == r decide
// at once -- they are v = first, first + subrounds, ... , so a launch covers exactly them -- and k_lv_apply then moves
// them and their share of the totals.  (A sweep that moves nothing changes nothing, so running all of them equals the
// specification's early stop.)
static int lv_sweeps(ddx_ctx* ctx, const LvGraph& in, const int64_t* wq, const int64_t* K, double gamma, int32_t sweeps, int subrounds,
                     const LvScratch& sc, int64_t m2, const int32_t* big_list, int32_t nbig, int32_t maxdeg, bool narrow) {
    const int64_t n = in.n;
    hipStream_t st = ctx->stream;
    if (sweeps <= 0 || m2 <= 0) return DDX_OK;
    int slots = 128;
    while (slots < 2 * maxdeg) slots <<= 1;
    const size_t big_lds = (size_t)slots * 12;
    if (nbig > 0 && big_lds > 64 * 1024) DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_lv_sweep_big), 12 * 2 * kLvCap));
    for (int s = 0; s < sweeps; ++s) {
        for (int r = 0; r < subrounds; ++r) {
            const int first = ((r - s) % subrounds + subrounds) % subrounds;
            const int64_t cnt = first < n ? (n - first + subrounds - 1) / subrounds : 0;      // nodes of this class
            if (cnt <= 0) continue;
            if (narrow)
                k_lv_sweep<true><<<(unsigned)ceil_div(cnt, 4 * kLvPerWave), 256, 0, st>>>(in.indptr, in.cols, wq, K, sc.comm, sc.tot, sc.size, n, gamma, (double)m2, first, subrounds, sc.next);
            else
                k_lv_sweep<false><<<(unsigned)ceil_div(cnt, 4 * kLvPerWave), 256, 0, st>>>(in.indptr, in.cols, wq, K, sc.comm, sc.tot, sc.size, n, gamma, (double)m2, first, subrounds, sc.next);
            if (nbig > 0)
                k_lv_sweep_big<<<(unsigned)nbig, 64, big_lds, st>>>(in.indptr, in.cols, wq, K, sc.comm, sc.tot, sc.size, gamma, (double)m2, first, subrounds,
                                                                    big_list, slots, sc.next);
            k_lv_apply<<<(unsigned)ceil_div(cnt, 256), 256, 0, st>>>(sc.next, K, n, first, subrounds, sc.comm, sc.tot, sc.size);
        }
    }
    return DDX_OK;
}

// members and strength of every node of the aggregated graph: csize[renum[c]] = size[c], Knext[renum[c]] = tot[c]
__global__ void k_lv_carry(const int32_t* __restrict__ used, const int32_t* __restrict__ renum, const int32_t* __restrict__ size,
                           const unsigned long long* __restrict__ tot, int64_t n, int32_t* __restrict__ csize, int64_t* __restrict__ Knext) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n || !used[c]) return;
    const int32_t r = renum[c];
    if (csize) csize[r] = size[c];
    Knext[r] = (int64_t)tot[c];
}

// one level: `sweeps` synchronous sweeps on `in`, exact aggregation into (member, out)
static int coarsen_level(ddx_ctx* ctx, const LvGraph& in, double gamma, int32_t sweeps, const LvScratch& sc, int64_t* wq, int64_t* K, int32_t* csize,
                         int32_t* big_list, int64_t& m2_out, int32_t& maxdeg_out, int32_t& nbig_out, bool& narrow_out, int32_t* member, int64_t* c_indptr, int32_t* c_cols, double* c_w, LvGraph& out) {
    const int64_t n = in.n, E = in.E;
    hipStream_t st = ctx->stream;
    DDX_HIP(ctx, hipMemsetAsync(sc.scal, 0, 256, st));
    if (E > 0) k_lv_quantise<<<(unsigned)ceil_div(E, 256), 256, 0, st>>>(in.w, E, wq);
    k_lv_strength<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(in.indptr, wq, n, K, sc.comm, sc.tot, sc.size, sc.scal, big_list);
    unsigned long long h_scal[4] = {0, 0, 0, 0};
    DDX_HIP(ctx, hipMemcpyAsync(h_scal, sc.scal, sizeof(h_scal), hipMemcpyDeviceToHost, st));
    DDX_HIP(ctx, wait_stream(ctx));
    const int64_t m2 = (int64_t)h_scal[0];
    const int32_t maxdeg = (int32_t)(h_scal[1] & 0xffffffffull);
    const int32_t nbig = (int32_t)(h_scal[1] >> 32);
    if (maxdeg > kLvCap) return set_err(ctx, DDX_E_UNSUPPORTED, "a node with %d neighbours exceeds the device sweep's capacity (%d)", maxdeg, kLvCap);
    const bool narrow = h_scal[3] < (1ull << 25);      // 64 such weights sum below 2^31: the register sweep adds them in 32 bits
    m2_out = m2; maxdeg_out = maxdeg; nbig_out = nbig; narrow_out = narrow;
    DDX_TRY(lv_sweeps(ctx, in, wq, K, gamma, sweeps, kSubrounds, sc, m2, big_list, nbig, maxdeg, narrow));
    const int32_t* cur = sc.comm;
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
        DDX_HIP(ctx, prim::sort_pairs(nullptr, tmp_sort, sc.keys_a, sc.keys_b, wq, sc.vals_b, (int)E, 0, end_bit, st));
        DDX_HIP(ctx, prim::reduce_by_key_sum(nullptr, tmp_red, sc.keys_b, sc.keys_a, sc.vals_b, sc.sums, runs_d, (size_t)E, st));
    }
    DDX_TRY(ensure(ctx, ctx->sort_tmp, std::max(tmp_scan, std::max(tmp_sort, tmp_red))));
    DDX_HIP(ctx, prim::exclusive_sum(ctx->sort_tmp.p, tmp_scan, sc.used, sc.renum, (int)n + 1, st));
    k_lv_member<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(cur, sc.renum, n, member);
    k_lv_carry<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(sc.used, sc.renum, sc.size, sc.tot, n, csize, sc.Ktop);
    int64_t runs = 0;
    if (E > 0) {
        k_lv_edge_keys<<<(unsigned)ceil_div(n, 4), 256, 0, st>>>(in.indptr, in.cols, member, n, shift, sc.keys_a);
        DDX_HIP(ctx, prim::sort_pairs(ctx->sort_tmp.p, tmp_sort, sc.keys_a, sc.keys_b, wq, sc.vals_b, (int)E, 0, end_bit, st));
        DDX_HIP(ctx, prim::reduce_by_key_sum(ctx->sort_tmp.p, tmp_red, sc.keys_b, sc.keys_a, sc.vals_b, sc.sums, runs_d, (size_t)E, st));
    }
    int32_t nc = 0;
    DDX_HIP(ctx, hipMemcpyAsync(&nc, sc.renum + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DDX_HIP(ctx, hipMemcpyAsync(&runs, runs_d, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    DDX_HIP(ctx, wait_stream(ctx));
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

int stage_coarsen_graph(ddx_ctx* ctx, double gamma, int32_t sweeps, int32_t levels) {
    const int64_t n = ctx->g_nodes;
    const int64_t E = ctx->g_entries;
    ctx->c_nodes = -1;
    ctx->lv_host_valid = false;
    LvScratch sc;
    LvSets sets;
    DDX_TRY(ensure(ctx, ctx->lv_buf, lv_bind(nullptr, n, E, sc, sets)));
    lv_bind(ctx->lv_buf.as<unsigned char>(), n, E, sc, sets);
    ScopedTimer t(ctx, "graph_coarsen");
    LvGraph cur;
    cur.n = n; cur.E = E; cur.indptr = ctx->g_d_indptr; cur.cols = ctx->g_d_cols; cur.w = ctx->g_d_vals;
    const int32_t* total = nullptr;          // member of every original node in the current coarse graph
    ctx->lv_levels = levels;
    for (int lvl = 0; lvl < levels; ++lvl) {
        const int o = lvl & 1;
        const int keep = lvl < kLvKeep ? lvl : kLvKeep - 1;       // (deeper levels reuse the last kept slot: part C refuses them)
        LvGraph nextg;
        if (lvl < kLvKeep) {           // (two output sets: the graphs of levels 0 and 1 survive two levels of part A)
            ctx->lv_n[lvl] = cur.n; ctx->lv_E[lvl] = cur.E;
            ctx->lv_indptr[lvl] = cur.indptr; ctx->lv_cols[lvl] = cur.cols; ctx->lv_w[lvl] = cur.w;
            ctx->lv_member[lvl] = sets.member[o];
        }
        int64_t m2 = 0;
        int32_t maxdeg = 0, nbig = 0;
        bool narrow = false;
        DDX_TRY(coarsen_level(ctx, cur, gamma, sweeps, sc, sc.wq[keep], sc.K[keep], sc.csize[keep], sc.big[keep], m2, maxdeg, nbig, narrow, sets.member[o],
                              sets.indptr[o], sets.cols[o], sets.w[o], nextg));
        if (lvl == 0) ctx->lv_m2 = m2;
        if (lvl < kLvKeep) { ctx->lv_maxdeg[lvl] = maxdeg; ctx->lv_nbig[lvl] = nbig; ctx->lv_narrow[lvl] = narrow; }
        if (!total) {
            total = sets.member[o];
        } else {
            int32_t* dst = (total == sets.total_a) ? sets.total_b : sets.total_a;
            k_lv_compose<<<(unsigned)ceil_div(n, 256), 256, 0, ctx->stream>>>(total, sets.member[o], n, dst);
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
// Communities keep the ids part B gave them (the label space of the coarsest graph) on the way down, so their totals are
// carried from level to level unchanged (the aggregation preserves strengths); only the member counts are per level.
// tot[lab[x]] += Ktop[x] over the nodes of the coarsest graph
__global__ void k_lv_top_totals(const int32_t* __restrict__ lab, const int64_t* __restrict__ Ktop, int64_t nc, unsigned long long* __restrict__ tot) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < nc) atomicAdd(tot + lab[x], (unsigned long long)Ktop[x]);
}

// size[lab[x]] += csize[x] over the nodes x of the level above (csize = how many nodes of this level x stands for)
__global__ void k_lv_level_sizes(const int32_t* __restrict__ lab, const int32_t* __restrict__ csize, int64_t n_up, int32_t* __restrict__ size) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n_up) atomicAdd(size + lab[x], csize[x]);
}

// comm[v] = lab[member[v]]
__global__ void k_lv_project(const int32_t* __restrict__ member, const int32_t* __restrict__ lab, int64_t n, int32_t* __restrict__ comm) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) comm[v] = lab[member[v]];
}

int stage_refine_communities(ddx_ctx* ctx, const int32_t* coarse_labels, double gamma, int32_t sweeps, int32_t* labels_out) {
    const int64_t n = ctx->g_nodes, E = ctx->g_entries, nc = ctx->c_nodes;
    hipStream_t st = ctx->stream;
    LvScratch sc;
    LvSets sets;
    if (lv_bind(nullptr, n, E, sc, sets) > ctx->lv_buf.cap) return set_err(ctx, DDX_E_ARG, "the coarsening work space is gone");
    if (ctx->lv_levels < 1 || ctx->lv_levels > kLvKeep)
        return set_err(ctx, DDX_E_UNSUPPORTED, "part C on the device follows at most %d levels of part A (%d were run)", kLvKeep, ctx->lv_levels);
    lv_bind(ctx->lv_buf.as<unsigned char>(), n, E, sc, sets);
    for (int64_t c = 0; c < nc; ++c)
        if (coarse_labels[c] < 0 || coarse_labels[c] >= nc) return set_err(ctx, DDX_E_ARG, "coarse label %d out of range at %lld", coarse_labels[c], (long long)c);
    ScopedTimer t(ctx, "graph_refine");
    // 2m of the original graph = sum of all strengths (the same on every level)
    DDX_HIP(ctx, hipMemcpyAsync(sc.lab, coarse_labels, sizeof(int32_t) * nc, hipMemcpyHostToDevice, st));
    DDX_HIP(ctx, hipMemsetAsync(sc.tot, 0, sizeof(int64_t) * nc, st));
    k_lv_top_totals<<<(unsigned)ceil_div(nc, 256), 256, 0, st>>>(sc.lab, sc.Ktop, nc, sc.tot);
    int64_t n_up = nc;
    for (int level = ctx->lv_levels - 1; level >= 0; --level) {
        LvGraph g;
        g.n = ctx->lv_n[level]; g.E = ctx->lv_E[level]; g.indptr = ctx->lv_indptr[level]; g.cols = ctx->lv_cols[level]; g.w = ctx->lv_w[level];
        DDX_HIP(ctx, hipMemsetAsync(sc.size, 0, sizeof(int32_t) * nc, st));
        k_lv_level_sizes<<<(unsigned)ceil_div(n_up, 256), 256, 0, st>>>(sc.lab, sc.csize[level], n_up, sc.size);
        k_lv_project<<<(unsigned)ceil_div(g.n, 256), 256, 0, st>>>(ctx->lv_member[level], sc.lab, g.n, sc.comm);
        DDX_TRY(lv_sweeps(ctx, g, sc.wq[level], sc.K[level], gamma, sweeps, kSubrounds, sc, ctx->lv_m2, sc.big[level], ctx->lv_nbig[level], ctx->lv_maxdeg[level], ctx->lv_narrow[level]));
        if (level > 0) DDX_HIP(ctx, hipMemcpyAsync(sc.lab, sc.comm, sizeof(int32_t) * g.n, hipMemcpyDeviceToDevice, st));
        else DDX_HIP(ctx, hipMemcpyAsync(labels_out, sc.comm, sizeof(int32_t) * g.n, hipMemcpyDeviceToHost, st));
        n_up = g.n;
    }
    DDX_HIP(ctx, wait_stream(ctx));
    DDX_HIP(ctx, hipGetLastError());
    // canonical numbering: by ascending smallest member = order of first appearance
    std::vector<int32_t> rank((size_t)std::max<int64_t>(n, nc), -1);
    int32_t k = 0;
    for (int64_t v = 0; v < n; ++v) {
        int32_t& r = rank[labels_out[v]];
        if (r < 0) r = k++;
        labels_out[v] = r;
    }
    return DDX_OK;
}

}  // namespace ddx
