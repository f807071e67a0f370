// Exact brute-force kNN over the low-dimensional embedding and the graphs handed to community
// detection (phenograph.cluster / sc.pp.neighbors call sites, dd.py:317-336).
//
// kNN: exact, defined in float64 (subtract, multiply, add component by component, no fused multiply-add; ties by
// index), computed in three passes over the points sorted by their first principal component -- an MFMA screen
// that bounds every query's k-th distance from the tiles around it, an MFMA screen of the pruned tile range that
// lists every pair that could be among the k nearest, and an exact float64 evaluation + sort of those few
// candidates (stage_knn below).  The screens only have to be conservative; the result is
// bit-identical to an IEEE float64 brute force.
#include "ddx_prims.h"

#include <algorithm>
#include <cstdlib>

#include "ddx_internal.h"

namespace ddx {

constexpr int kMaxDim = 128;

typedef float f4 __attribute__((ext_vector_type(4)));

// slack of the bfloat16 distance screen, relative to |q|^2 + |c|^2 (derivation at the emit kernel)
constexpr float kScreenSlackBf = 4.2e-5f;

// ---- layouts -----------------------------------------------------------------------------------
// E   : row-major float32 [Mp][CP]            (exact re-evaluation gathers whole rows)
// nrm : float32 squared norms; padding points carry +inf so they can never pass the screen.
// Eb  : bfloat16 split operands for v_mfma_f32_16x16x32_bf16: every coordinate a is stored as
//       hi = bf16(a) and lo = bf16(a - hi); for tile t, 32-component block kb and part p (0 = hi, 1 = lo)
//       lane l = ((d % 32) / 8) * 16 + j holds components d = 32*kb + 8*(l>>4) + 0..7 of point 16*t + j as
//       8 consecutive bf16 at Eb[(((t*KB + kb)*2 + p)*64 + l)*8], i.e. one coalesced 1 KB load per operand.
// Points are laid out in the order `perm` (cell by cell, ascending first principal component inside a cell, see
// stage_knn): row r of every layout is embedding row perm[r];  p1[r] = its first component (+inf for padding rows).
__global__ void k_knn_prepare(const float* __restrict__ in, const int32_t* __restrict__ perm, int64_t M, int64_t Mp, int C, int CP,
                              float* __restrict__ E, __bf16* __restrict__ Eb,
                              float* __restrict__ nrm, float* __restrict__ p1, f4* __restrict__ start4) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Mp) return;
    const int64_t src = r < M ? perm[r] : 0;
    p1[r] = r < M ? in[src * C] : __builtin_huge_valf();
    float n = 0.f;
    const int64_t tile = r >> 4;
    const int j = (int)(r & 15);
    const int KB = CP / 32;
    for (int d = 0; d < CP; ++d) {
        const float v = (r < M && d < C) ? in[src * C + d] : 0.f;
        E[r * CP + d] = v;
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)hi);
        const int kb = d >> 5, lane = (((d & 31) >> 3) << 4) | j, e = d & 7;
        Eb[(((tile * KB + kb) * 2 + 0) * 64 + lane) * 8 + e] = hi;
        Eb[(((tile * KB + kb) * 2 + 1) * 64 + lane) * 8 + e] = lo;
        n = fmaf(v, v, n);
    }
    nrm[r] = (r < M) ? n : __builtin_huge_valf();
    // accumulator start value of the bfloat16 emit pass, as one MFMA C quad: -0.5*(1-slack)*|c|^2 (-inf for padding)
    const float h = (r < M) ? -0.5f * (1.0f - kScreenSlackBf) * n : -__builtin_huge_valf();
    start4[r] = f4{h, h, h, h};
}

// Threshold folding for the emit pass (embeddings of at most 30 components: two of the 32 padded components are free).
// The screen asks  acc > hr_q  with acc = q.c - 0.5 (1-s) |c|^2 from the matrix pipe and hr_q = 0.5 ((1-s) |q|^2 - T_q) a
// constant of the query.  Give every CANDIDATE the value 1 in components 30 and 31 and every QUERY the value -hr_q there,
// cut into four bfloat16 pieces (hi/lo of the two components: 32 mantissa bits, i.e. all of the float32): the matrix pipe
// then delivers acc - hr_q and the screen is "any of the wave's accumulators positive" -- a v_max3 tree and ONE compare per
// tile instead of eight compares and seven scalar ORs.  The four extra terms add four float32 roundings to the
// accumulation (2.4e-7 relative to |q|^2 + |c|^2 + T/2), at most 4.8e-7 (|q|^2 + |c|^2) where it matters -- near the decision d2 = T, hence T <= 2 (|q|^2 + |c|^2) -- covered by raising the slack
// from 4.0e-5 to 4.2e-5 (margin 1e-6 (|q|^2 + |c|^2)).
// Queries without a bound pass everything (+3e38, finite: inf * 0 would poison the hi*lo product), padding queries
// nothing (-3e38).  Eb itself (candidate role) is patched in place -- the bound pass, which needs the zeros, is over.
__global__ void k_knn_fold(const float* __restrict__ nrm, const float* __restrict__ thr, int64_t Mp, __bf16* __restrict__ Eb,
                           __bf16* __restrict__ Ebq) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per 16-byte vector of the operand image
    if (t >= Mp * 8) return;                                               // CP = 32: tile x 2 parts x 64 lanes
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 v = reinterpret_cast<const u4*>(Eb)[t];
    const int lane = (int)(t & 63), part = (int)((t >> 6) & 1);
    const int64_t tile = t >> 7;
    if ((lane >> 4) == 3) {                                                // components 24..31 of point 16*tile + (lane & 15)
        const int64_t r = tile * 16 + (lane & 15);
        const float n = nrm[r], T = thr[r];
        float x = -0.5f * ((1.0f - kScreenSlackBf) * n - T);               // -hr
        if (!(n < __builtin_huge_valf())) x = -3.0e38f;
        else if (!(T < __builtin_huge_valf())) x = 3.0e38f;
        const __bf16 p1 = (__bf16)x;
        const float r1 = x - (float)p1;
        const __bf16 p2 = (__bf16)r1;
        const float r2 = r1 - (float)p2;
        const __bf16 p3 = (__bf16)r2;
        const __bf16 p4 = (__bf16)(r2 - (float)p3);
        const __bf16 a = part == 0 ? p1 : p2, b = part == 0 ? p3 : p4;     // (component 30, component 31) of this part
        u4 q = v;
        q.w = (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
        reinterpret_cast<u4*>(Ebq)[t] = q;
        const __bf16 one = (__bf16)1.0f;
        const uint32_t ones = (uint32_t)__builtin_bit_cast(uint16_t, one) * 0x10001u;
        v.w = part == 0 ? ones : 0u;                                       // candidates: hi = 1, lo = 0
        reinterpret_cast<u4*>(Eb)[t] = v;
    } else {
        reinterpret_cast<u4*>(Ebq)[t] = v;
    }
}

// Smallest sample distances kept per lane and row in the bound pass.  Whatever the lanes keep, the k-th smallest of the
// kept values is the k-th smallest of a subset, i.e. a valid upper bound; keeping 4 (instead of 6) loosens it for the
// queries where one lane meets more than 4 of the k nearest sample points, and lets four waves share a SIMD.
constexpr int kBoundKeepSmall = 4;   // k <= 48
constexpr int kBoundKeepLarge = 6;   // k <= 80
constexpr int kCandCap = 768;      // candidate slots per query between the emit and select passes

// ================================================================================================
// exact kNN in three passes (16x16 query x candidate tiles on v_mfma_f32_16x16x32_bf16 with a bfloat16 hi/lo split);
// the points are laid out cell by cell (section "Cells" below), first-principal-component order inside a cell:
//   1. bound : over the tiles of the query's own cell and of the cells nearest to it, every lane keeps the kBoundKeep
//              smallest *upper bounds* of the squared distance per screened row; the k-th smallest of the
//              16*kBoundKeep kept values of a row is an upper bound T_q of the query's true k-th
//              neighbour distance (k-th smallest of a subset >= k-th smallest of the whole set).
//   2. emit  : over the candidate tiles that the cell test cannot rule out for the wave's 32 queries (k_knn_tilelists),
//              a pair survives when a *lower bound* of its squared distance is below T_q (so no true neighbour can
//              be lost); survivors (well under a hundred per query) are appended to the query's candidate list.
//   3. select: one wave per query evaluates its candidates exactly (float64, the reference's
//              arithmetic, one candidate per lane) and sorts them by (distance, index); the
//              first k are the result.  A query whose list overflowed is re-scanned over all points
//              by the same pass, so the result is exact in every case and independent of the
//              order in which survivors were appended.
// Both MFMA passes walk a LIST of tiles (the emit pass's with a mask of the block's waves that screen the tile): the tiles
// of a step (8, or 4 for more than 32 components) are copied into LDS asynchronously from wherever they lie.
// ================================================================================================

// Workgroup numbering of the MFMA passes.  Hardware workgroup b runs on XCD b % 8; consecutive query blocks screen nearly
// the same candidate tiles.  With chunk > 0 the workgroups resident on one XCD at a time are `chunk` CONSECUTIVE query
// blocks (so a candidate tile fetched into that XCD's L2 serves them all), and the XCDs take adjacent chunks of the query
// range (so all of them meet the same mix of short and long lists).  Returns -1 for the padding of the last chunk.
__device__ __forceinline__ int64_t knn_block(int64_t nblocks, int chunk) {
    const int64_t b = blockIdx.x;
    if (chunk <= 0) return b < nblocks ? b : -1;
    const int64_t xcd = b & 7, s = b >> 3;
    const int64_t lb = ((s / chunk) * 8 + xcd) * chunk + (s % chunk);
    return lb < nblocks ? lb : -1;
}

// tiles per staged step: 16 KB of coordinates per buffer whatever the padded dimension
__host__ __device__ constexpr int chunk_tiles(int CP) { return CP <= 32 ? 8 : 4; }

constexpr int kBoundRT = 2;   // query tiles per wave in the bound pass (register budget: 16 rows x kBoundKeep)
constexpr int kBoundWaves = 8; // waves per workgroup of the bound pass: they share the staged sample, and staging is what bounds the pass
constexpr int kEmitRT = 2;    // query tiles per wave in the emit pass
constexpr int kEmitWaves = 4;  // waves per workgroup of the emit pass (they share the staged candidate tiles)
constexpr int kEmitSegSteps = 4;    // steps (of 8 or 4 tiles) of a block's list that one workgroup screens
constexpr int kEmitLog = 128;      // records a wave logs in LDS before it files them (a tile adds up to 64: filed when fewer are free)

// ---- bfloat16-split MFMA screen ----------------------------------------------------------------------------
// q.c ~= qh.ch + qh.cl + ql.ch with three v_mfma_f32_16x16x32_bf16.  Dropped
// terms (ql.cl and the residuals of the two-term split) are <= 3*2^-18 |q_i||c_i| per component, the float32
// accumulation of 3*32 exact products adds <= ~6e-6 sum|q_i c_i|: |error(q.c)| <= 1.8e-5 |q||c| <= 0.9e-5 (|q|^2+|c|^2);
// doubled in the distance and with the float32 norms that is 2.5e-5 (|q|^2+|c|^2).  The slack kScreenSlackBf leaves 1.6x margin.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CP, int RT>
struct QueryTilesBf {
    static constexpr int KB = CP / 32;
    bf16x8 ah[RT][KB], al[RT][KB];
    __device__ __forceinline__ void load(const __bf16* __restrict__ Eb, int64_t q0, int lane) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const bf16x8* p = reinterpret_cast<const bf16x8*>(Eb) + ((((q0 >> 4) + rt) * KB + kb) * 2) * 64 + lane;
                ah[rt][kb] = p[0];
                al[rt][kb] = p[64];
            }
    }
    // tile image in LDS: [kb][part][lane] vectors of 16 bytes
    __device__ __forceinline__ void dots(const f4* tile, int lane, f4 (&acc)[RT]) const { dots_from(tile, lane, f4{0.f, 0.f, 0.f, 0.f}, acc); }
    // one 32-component block whose candidate operands are already in registers (CP = 32)
    __device__ __forceinline__ void dots_regs(const f4 rh, const f4 rl, const f4 start, f4 (&acc)[RT]) const {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, rh);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, rl);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][0], bh, start, 0, 0, 0);   // small terms first
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][0], bl, acc[rt], 0, 0, 0);
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][0], bh, acc[rt], 0, 0, 0);
        }
    }
    __device__ __forceinline__ void dots_from(const f4* tile, int lane, const f4 start, f4 (&acc)[RT]) const {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = start;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const f4 rh = tile[(kb * 2 + 0) * 64 + lane];
            const f4 rl = tile[(kb * 2 + 1) * 64 + lane];
            const bf16x8 bh = __builtin_bit_cast(bf16x8, rh);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, rl);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][kb], bh, acc[rt], 0, 0, 0);   // small terms first
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][kb], bl, acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][kb], bh, acc[rt], 0, 0, 0);
            }
        }
    }
};

// One step of a tile list copied into LDS by asynchronous global -> LDS loads (16 bytes per lane, LDS destination =
// wave-uniform base + lane*16, no registers held).  `ent`: lane j (j < G) holds list entry j of the step; a tile is
// CP*4 vectors, so the tile a wave copies with one instruction is wave-uniform.  PER = floats of per-point side data
// copied along (4: the accumulator start quads of the emit pass, 1: the squared norms of the bound pass).
template <int CP, int BW, int PER>
__device__ __forceinline__ void stage_tiles(const f4* __restrict__ srcE, const float* __restrict__ side, f4* lds_c, float* lds_s, int ent, int tid,
                                            int wave, int lane) {
    constexpr int G = chunk_tiles(CP);
    constexpr int tile_vecs = CP * 4;
#pragma unroll
    for (int u = 0; u < G * tile_vecs / (64 * BW); ++u) {
        const int v0 = u * (64 * BW) + wave * 64;                   // first vector of this wave's instruction (wave-uniform)
        const int slot = __builtin_amdgcn_readfirstlane(v0 / tile_vecs);
        const int64_t tile = __builtin_amdgcn_readlane(ent, slot);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcE + tile * tile_vecs + (v0 % tile_vecs) + lane),
                                         (__attribute__((address_space(3))) void*)(lds_c + v0), 16, 0, 0);
    }
    // side data: G tiles x 16 points, one item per lane
    if (tid < G * 16) {                                           // whole waves (G*16 is 128 or 64)
        const int64_t tile = __shfl(ent, tid >> 4, 64);
        if (PER == 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(side + (tile * 16 + (tid & 15)) * 4),
                                             (__attribute__((address_space(3))) void*)(lds_s + wave * 256), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(side + tile * 16 + (tid & 15)),
                                             (__attribute__((address_space(3))) void*)(lds_s + wave * 64), 4, 0, 0);
    }
}

// Bound pass.  blist[blk][*] = the block's sample (k_knn_boundlists): entries of bcount[blk] tiles.  k beyond what one launch
// ranks (k > 16 * (kBoundKeep - 1)): the sample is dealt out over `stride` launches, launch `phase` taking every stride-th
// entry and the ceil(k / stride)-th smallest of ITS points; the largest of those bounds holds at least k points of the
// whole sample below it -- `combine` keeps the maximum over the launches.
template <int CP, int kBoundKeep>
__global__ void __launch_bounds__(64 * kBoundWaves) k_knn_bound_bf(const __bf16* __restrict__ Eb, const float* __restrict__ nrm,
                                                      int64_t Mp, int K, int include_self, const int32_t* __restrict__ blist, const int32_t* __restrict__ bcount,
                                                      int64_t bcap, int stride, int phase, int combine, float* __restrict__ thr_out,
                                                      int xcd_chunk) {
    constexpr int RT = kBoundRT, NV = 4 * RT;
    constexpr int G = chunk_tiles(CP);
    constexpr int tile_vecs = CP * 4;
    __shared__ f4 lds_c[2][G * tile_vecs];
    __shared__ float lds_n[2][G * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t blk = knn_block(Mp / (kBoundWaves * 16 * RT), xcd_chunk);
    if (blk < 0) return;
    const int64_t q0 = (blk * kBoundWaves + wave) * (16 * RT);
    QueryTilesBf<CP, RT> qt;
    qt.load(Eb, q0, lane);
    const int rbase = 4 * (lane >> 4), jcol = lane & 15;
    float nq[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) nq[v] = nrm[q0 + (v >> 2) * 16 + rbase + (v & 3)];
    float best[NV][kBoundKeep];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int t = 0; t < kBoundKeep; ++t) best[v][t] = __builtin_huge_valf();
    float tau[NV];                                  // see the shortcut in the loop: -inf lets everything through while a list is not full
#pragma unroll
    for (int v = 0; v < NV; ++v) tau[v] = nq[v] < __builtin_huge_valf() ? -__builtin_huge_valf() : __builtin_huge_valf();     // (padding queries: nothing ever passes)
    constexpr float kHalfOnePlusSlack = 0.5f * (1.f + kScreenSlackBf);
    const int64_t own_tile = q0 >> 4;
    const int32_t* lst = blist + blk * bcap;
    const int n_all = bcount[blk];
    const int n = n_all > phase ? (n_all - phase + stride - 1) / stride : 0;       // entries of this launch
    const int nsteps = (n + G - 1) / G;
    auto entry = [&](int step) {                                                     // lane j < G: entry j of the step (the ragged end repeats the last one)
        int i = step * G + (lane & (G - 1));
        if (i > n - 1) i = n - 1;
        return lst[phase + (int64_t)stride * i];
    };
    const f4* srcE = reinterpret_cast<const f4*>(Eb);
    if (nsteps > 0) {
        int e0 = entry(0), e1 = nsteps > 1 ? entry(1) : 0;
        stage_tiles<CP, kBoundWaves, 1>(srcE, nrm, lds_c[0], lds_n[0], e0, tid, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int step = 0; step < nsteps; ++step) {
            const int buf = step & 1;
            int e2 = 0;
            if (step + 1 < nsteps) stage_tiles<CP, kBoundWaves, 1>(srcE, nrm, lds_c[buf ^ 1], lds_n[buf ^ 1], e1, tid, wave, lane);
            if (step + 2 < nsteps) e2 = entry(step + 2);
            const int ntile = (n - step * G) < G ? (n - step * G) : G;
            for (int t = 0; t < ntile; ++t) {
                const int64_t tile = __builtin_amdgcn_readlane(e0, t);
                const float nc = lds_n[buf][t * 16 + jcol];
                f4 acc[RT];
                qt.dots(lds_c[buf] + t * tile_vecs, lane, acc);
                const bool own = !include_self && (tile >= own_tile && tile < own_tile + RT);
                // Most (tile, accumulator slot) pairs change nothing once the lists have filled: ub < best  <=>  dot - tau > c with
                // tau = ((1 + slack) |q|^2 - best) / 2 per query and c = (1 + slack) |c|^2 / 2 per candidate -- one subtraction and
                // one compare per pair; only the slots in which some lane of the wave sees a candidate run the exact test below
                // (a slot holds 64 of the wave's 512 four-best lists: after j tiles it still has a taker with probability ~256 / j).
                // The shortcut rounds differently from the test it stands for; letting a borderline candidate go or sending one
                // through in vain changes a bound by nothing that matters -- any k-th smallest of upper bounds of sample points
                // is a valid threshold.
                const float cthr = kHalfOnePlusSlack * nc * (1.f - 4e-7f);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float dot = acc[v >> 2][v & 3];
                    if (__ballot(dot - tau[v] > cthr) == 0ull) continue;
                    const float nn = nq[v] + nc;
                    float ub = fmaf(-2.f, dot, nn) + kScreenSlackBf * nn;     // upper bound of the exact squared distance
                    if (own && (tile * 16 + jcol) == (q0 + (v >> 2) * 16 + rbase + (v & 3))) ub = __builtin_huge_valf();
                    if (!(ub < best[v][kBoundKeep - 1])) continue;           // also rejects NaN (padding rows / candidates)
                    best[v][kBoundKeep - 1] = ub;
#pragma unroll
                    for (int u = kBoundKeep - 1; u > 0; --u) {
                        const float lo = fminf(best[v][u - 1], best[v][u]), hi = fmaxf(best[v][u - 1], best[v][u]);
                        best[v][u - 1] = lo;
                        best[v][u] = hi;
                    }
                    tau[v] = 0.5f * ((1.f + kScreenSlackBf) * nq[v] - best[v][kBoundKeep - 1]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next step has landed before anyone crosses the barrier
            __syncthreads();
            e0 = e1;
            e1 = e2;
        }
    }
    // k-th smallest of the 16*kBoundKeep values of each row (held by the 16 lanes that share lane>>4)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        float kth = __builtin_huge_valf();
        int rank[kBoundKeep];
#pragma unroll
        for (int t = 0; t < kBoundKeep; ++t) rank[t] = 0;
        for (int o = 0; o < 16; ++o) {
            const int src = (lane & 48) | ((jcol + o) & 15);
#pragma unroll
            for (int u = 0; u < kBoundKeep; ++u) {
                const float other = __shfl(best[v][u], src, 64);
                const int okey = ((jcol + o) & 15) * kBoundKeep + u;      // tie-break key of the other value
#pragma unroll
                for (int t = 0; t < kBoundKeep; ++t) {
                    const int mkey = jcol * kBoundKeep + t;
                    rank[t] += (other < best[v][t] || (other == best[v][t] && okey < mkey)) ? 1 : 0;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kBoundKeep; ++t)
            if (rank[t] == K - 1) kth = best[v][t];
        // exactly one lane of the 16 holds the value of rank K-1: share it
        for (int o = 1; o < 16; o <<= 1) kth = fminf(kth, __shfl_xor(kth, o, 64));
        if (jcol == 0) {                                                      // +inf when fewer than K sample points
            float* dst = thr_out + q0 + (v >> 2) * 16 + rbase + (v & 3);
            *dst = combine ? fmaxf(*dst, kth) : kth;
        }
    }
}

// Emit pass.  elist[blk][*]: tile | (waves of the block that screen it) << 24, ecount[blk] entries (a multiple of
// the step size, padded with mask-0 entries); candidates are appended to cbuf[q][*], ccount[q] counts them (beyond `cap`:
// overflow).  The waves of a block share the staged tiles: the pass is bound by that staging traffic (L2 / Infinity Cache ->
// LDS), not by the matrix pipe, so a block is as many waves as a workgroup holds (16: 512 queries per staged tile).
// RTQ: query tiles per wave.  2 (default): a wave screens 32 queries, the block's waves map one to one onto the waves the lists' masks were
// built for.  4 (option knn_emit_rt, CP = 32): a wave screens 64 queries = two of the lists' waves -- every candidate tile read from LDS and
// every ballot / branch / log slot serves twice the queries, at half the waves per block.
template <int CP, bool FOLD, int kEmitBW, int RTQ = kEmitRT>
__global__ void __launch_bounds__(64 * kEmitBW) __attribute__((amdgpu_waves_per_eu((CP <= 64 && RTQ == kEmitRT) ? 4 : 2, (CP <= 64 && RTQ == kEmitRT) ? 4 : 2)))
k_knn_emit_bf(const __bf16* __restrict__ Eb, const __bf16* __restrict__ Ebq, const float* __restrict__ nrm, const f4* __restrict__ start4,
              const float* __restrict__ thr, int64_t Mp, int include_self, int32_t* __restrict__ ccount, int32_t* __restrict__ cbuf,
              const int32_t* __restrict__ elist, const int32_t* __restrict__ ecount, int64_t ecap, int dbg,
              int cap, int nseg, int seg_steps, long long* __restrict__ tdbg) {
    constexpr int RT = RTQ, NV = 4 * RT;
    constexpr int WM = RTQ / kEmitRT;                    // waves of the lists' geometry per wave of this kernel
    static_assert(RTQ % kEmitRT == 0 && NV <= 16, "");
    constexpr int G = chunk_tiles(CP);
    constexpr int tile_vecs = CP * 4;
    __shared__ f4 lds_c[2][G * tile_vecs];
    __shared__ f4 lds_h[2][G * 16];     // accumulator start values -0.5*(1-slack)*|c|^2, one MFMA C quad per candidate (k_knn_prepare)
    __shared__ unsigned long long hlog[kEmitBW][kEmitLog];     // hits of this wave, one record per LANE with a hit: tile | lane << 24 | (its accumulator slots that passed) << 32
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // work item = (query block, segment of its list): workgroup id = blk * nseg + seg.  nseg is a multiple of 8, so the same
    // segment of consecutive query blocks -- nearly the same candidate tiles -- runs on one XCD at about the same time.
    const int64_t blk = blockIdx.x / nseg;
    const int seg = (int)(blockIdx.x % nseg);
    const int nent = ecount[blk];
    const int s_lo = seg * seg_steps;
    if (s_lo * G >= nent) return;                // (block-uniform) the list ends before this segment
    const int nsteps = min(nent / G - s_lo, seg_steps);
    if (tdbg && tid == 0) { tdbg[4 * (size_t)blockIdx.x] = wall_clock64(); tdbg[4 * (size_t)blockIdx.x + 2] = nsteps; }
    const int64_t q0 = (blk * kEmitBW + wave) * (16 * RT);
    QueryTilesBf<CP, RT> qt;
    qt.load(FOLD ? Ebq : Eb, q0, lane);          // FOLD: the query operands carry -hr in components 30 / 31 (k_knn_fold)
    const int rbase = 4 * (lane >> 4), jcol = lane & 15;
    // Most tiles leave a hit or two somewhere in the wave (~70 survivors per query), so what runs per tile WITH a hit is the hot
    // path: a lane with a hit writes ONE 8-byte record -- the tile, its lane number, the bits of its accumulator slots that passed --
    // at a slot from the ballot's prefix count; no per-slot ballots, no scalar branching per slot (rounds 3-4: ~60 instructions per
    // tile with a hit).  The log is decoded a lane per record when fewer than 64 records are free: query and candidate from the lane
    // number and bit positions, the query itself dropped there, one atomic per candidate for its slot in the query's list (several
    // work items append to one list: its order depends on the run, the select pass sorts it).
    int nlog = 0;                                   // wave-uniform
    auto flush_log = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < nlog; i += 64) {
            const unsigned long long rec = hlog[wave][i];
            const int32_t tile = (int32_t)(rec & 0xffffffull);
            const int ln = (int)((rec >> 24) & 63ull);
            unsigned mk = (unsigned)(rec >> 32);
            const int32_t cand = tile * 16 + (ln & 15);
            while (mk) {
                const int v = __builtin_ctz(mk);
                mk &= mk - 1u;
                const int64_t q = q0 + (v >> 2) * 16 + 4 * (ln >> 4) + (v & 3);
                if (!include_self && q == cand) continue;              // a point is not its own neighbour
                const int slot = atomicAdd(&ccount[q], 1);
                if (slot < cap) cbuf[q * cap + slot] = cand;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        nlog = 0;
    };
    float hr[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int64_t q = q0 + (v >> 2) * 16 + rbase + (v & 3);
        const float n = nrm[q], t = thr[q];
        hr[v] = 0.5f * ((1.0f - kScreenSlackBf) * n - t);
        if (!(n < __builtin_huge_valf())) hr[v] = __builtin_huge_valf();
        else if (!(t < __builtin_huge_valf())) hr[v] = -__builtin_huge_valf();
        if (dbg & 1) hr[v] = __builtin_huge_valf();          // experiment: nothing passes the screen
        if (FOLD) hr[v] = 0.f;                               // the threshold travels inside the dot product
    }
    const int32_t* lst = elist + blk * ecap + (int64_t)s_lo * G;
    const f4* srcE = reinterpret_cast<const f4*>(Eb);
    const float* srcH = reinterpret_cast<const float*>(start4);
    // (p: packed entries, lane j < G holds entry j of the step; e: their tile numbers)
    int p0 = lst[lane & (G - 1)], p1 = nsteps > 1 ? lst[G + (lane & (G - 1))] : 0;
    int e0 = p0 & 0xffffff;
    stage_tiles<CP, kEmitBW, 4>(srcE, srcH, lds_c[0], reinterpret_cast<float*>(lds_h[0]), e0, tid, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        int p2 = 0;
        if (step + 1 < nsteps) stage_tiles<CP, kEmitBW, 4>(srcE, srcH, lds_c[buf ^ 1], reinterpret_cast<float*>(lds_h[buf ^ 1]), p1 & 0xffffff, tid, wave, lane);
        if (step + 2 < nsteps) p2 = lst[(step + 2) * G + (lane & (G - 1))];
        // this wave's tiles of the step (wave-uniform mask)
        unsigned tmask = (unsigned)__ballot(lane < G && ((((unsigned)p0 >> 24) >> (wave * WM)) & ((1u << WM) - 1u))) & ((1u << G) - 1u);
        const f4* tb = lds_c[buf];
        const f4* th = lds_h[buf];
        static_assert(CP % 32 == 0, "");
        auto judge = [&](const f4 (&acc)[RT], const int32_t tile) {
            unsigned mk = 0;
            unsigned long long any;
            if (FOLD) {
                // one compare per tile: the largest of the wave's accumulators against zero (v_max3 tree)
                float m = acc[0].x;
#pragma unroll
                for (int v = 1; v < NV; ++v) m = fmaxf(m, acc[v >> 2][v & 3]);
                any = __ballot(m > 0.f);
                if (any == 0ull) return;
                // the slots that passed, from the sign bits (two instructions per slot; +0 counts as passed: the select pass evaluates
                // every candidate exactly, one more is harmless)
                unsigned neg = 0;
#pragma unroll
                for (int v = 0; v < NV; ++v) neg |= (__float_as_uint(acc[v >> 2][v & 3]) >> 31) << v;
                mk = m > 0.f ? (~neg & ((1u << NV) - 1u)) : 0u;
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) mk |= (acc[v >> 2][v & 3] > hr[v] ? 1u : 0u) << v;
                any = __ballot(mk != 0u);
                if (any == 0ull) return;
            }
            if (mk != 0u) {
                const int slot = nlog + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(any >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)any, 0u));
                hlog[wave][slot] = (unsigned long long)((uint32_t)tile | (uint32_t)lane << 24) | ((unsigned long long)mk << 32);
            }
            nlog += __popcll(any);
            if (nlog > kEmitLog - 64) flush_log();
        };
        auto tile_of = [&](int t) { return (int32_t)__builtin_amdgcn_readlane(e0, t); };
        if (CP == 32) {
            // operands of the wave's next tile are read from LDS while the current one is on the matrix pipe; two register
            // sets (A, B) alternate, two tiles per trip, so that nothing is copied between them
            if (tmask) {
                int t = __builtin_ctz(tmask);
                tmask &= tmask - 1u;
                f4 ah_ = tb[t * tile_vecs + lane], al_ = tb[t * tile_vecs + 64 + lane], as_ = th[t * 16 + jcol];
                while (true) {
                    int t1 = -1;
                    f4 bh_ = ah_, bl_ = al_, bs_ = as_;
                    if (tmask) {
                        t1 = __builtin_ctz(tmask);
                        tmask &= tmask - 1u;
                        bh_ = tb[t1 * tile_vecs + lane];
                        bl_ = tb[t1 * tile_vecs + 64 + lane];
                        bs_ = th[t1 * 16 + jcol];
                    }
                    f4 acc[RT];
                    qt.dots_regs(ah_, al_, as_, acc);
                    judge(acc, tile_of(t));
                    if (t1 < 0) break;
                    t = -1;
                    if (tmask) {
                        t = __builtin_ctz(tmask);
                        tmask &= tmask - 1u;
                        ah_ = tb[t * tile_vecs + lane];
                        al_ = tb[t * tile_vecs + 64 + lane];
                        as_ = th[t * 16 + jcol];
                    }
                    f4 acc2[RT];
                    qt.dots_regs(bh_, bl_, bs_, acc2);
                    judge(acc2, tile_of(t1));
                    if (t < 0) break;
                }
            }
        } else {
            while (tmask) {
                const int t = __builtin_ctz(tmask);
                tmask &= tmask - 1u;
                f4 acc[RT];
                qt.dots_from(tb + t * tile_vecs, lane, th[t * 16 + jcol], acc);
                judge(acc, tile_of(t));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next step has landed before anyone crosses the barrier
        __syncthreads();
        p0 = p1;
        p1 = p2;
        e0 = p0 & 0xffffff;
    }
    if (nlog) flush_log();
    if (tdbg) {
        __syncthreads();
        if (tid == 0) tdbg[4 * (size_t)blockIdx.x + 1] = wall_clock64();
    }
}

// Exact squared distance in the reference's arithmetic: float64, (a-b)*(a-b) rounded, then added,
// components in order -- bit-identical to the float64 brute-force definition (oracle knn_bruteforce_f64).
template <int CP>
__device__ __forceinline__ double exact_d2(const float* q /* LDS, CP floats */, const float* __restrict__ c) {
#pragma clang fp contract(off)
    double d2 = 0.0;
#pragma unroll
    for (int t = 0; t < CP; t += 4) {
        const f4 y = *reinterpret_cast<const f4*>(c + t);
        const f4 x = *reinterpret_cast<const f4*>(q + t);
        const double d0 = (double)x.x - (double)y.x; const double s0 = d0 * d0; d2 = d2 + s0;
        const double d1 = (double)x.y - (double)y.y; const double s1 = d1 * d1; d2 = d2 + s1;
        const double d2_ = (double)x.z - (double)y.z; const double s2 = d2_ * d2_; d2 = d2 + s2;
        const double d3 = (double)x.w - (double)y.w; const double s3 = d3 * d3; d2 = d2 + s3;
    }
    return d2;
}

constexpr int kSelMax = 1024;   // sort window of the select pass (power of two, >= kCandCap + 64)
constexpr int kSelSmall = 256;  // queries with at most this many candidates (nearly all) sort in a small window: 4x the waves per CU
constexpr int kSelHuge = 4096;  // k > 80: lists of up to kCandCapLarge entries (one wave per workgroup: 48 KB of LDS)
constexpr int kCandCapLarge = 3072;

// bitonic sort of d[0..P), ix[0..P) by (distance, index), ascending; one wave, P a power of two >= 64
__device__ __forceinline__ void wave_sort(double* d, int32_t* ix, int P, int lane) {
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (P >> 1); t += 64) {
                const int lo = ((t / stride) * stride * 2) + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double dl = d[lo], dh = d[hi];
                const int32_t il = ix[lo], ih = ix[hi];
                const bool gt = dl > dh || (dl == dh && il > ih);
                if (gt == up) { d[lo] = dh; d[hi] = dl; ix[lo] = ih; ix[hi] = il; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The same sort with the window in registers (P = 64 R entries, entry e = 64 r + lane): partners at least 64 apart sit in
// one lane, the others are exchanged by lane permutes -- no LDS traffic, hence none of the bank conflicts the strided
// pair accesses of wave_sort cost (9 in 10 of its LDS cycles).  Same network, same order.
template <int R>
__device__ __forceinline__ void wave_sort_regs(double (&d)[R], int32_t (&ix)[R], int lane) {
#pragma unroll
    for (int size = 2; size <= 64 * R; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {
                const int rs = stride >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r & rs) continue;
                    const int r2 = r | rs;
                    const bool up = (((r << 6) & size) == 0);
                    const bool gt = d[r] > d[r2] || (d[r] == d[r2] && ix[r] > ix[r2]);
                    if (gt == up) {
                        const double td = d[r]; d[r] = d[r2]; d[r2] = td;
                        const int32_t ti = ix[r]; ix[r] = ix[r2]; ix[r2] = ti;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = (r << 6) | lane;
                    const bool up = ((e & size) == 0);
                    const bool lower = ((lane & stride) == 0);
                    const double od = __shfl_xor(d[r], stride, 64);
                    const int32_t oi = __shfl_xor(ix[r], stride, 64);
                    const bool other_less = od < d[r] || (od == d[r] && oi < ix[r]);
                    const bool want_min = (lower == up);
                    if (want_min == other_less && !(od == d[r] && oi == ix[r])) { d[r] = od; ix[r] = oi; }
                }
            }
        }
    }
}

// bitonic sort of an LDS window of P <= 1024 entries through registers (wave_sort_regs: no strided LDS pair accesses); larger
// windows fall back to the in-LDS network
__device__ __forceinline__ void wave_sort_window(double* d, int32_t* ix, int P, int lane) {
    auto via_regs = [&](auto rtag) {
        constexpr int R = decltype(rtag)::value;
        double dr[R];
        int32_t ir[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { dr[r] = d[r * 64 + lane]; ir[r] = ix[r * 64 + lane]; }
        wave_sort_regs<R>(dr, ir, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) { d[r * 64 + lane] = dr[r]; ix[r * 64 + lane] = ir[r]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    if (P == 64) via_regs(std::integral_constant<int, 1>());
    else if (P == 128) via_regs(std::integral_constant<int, 2>());
    else if (P == 256) via_regs(std::integral_constant<int, 4>());
    else if (P == 512) via_regs(std::integral_constant<int, 8>());
    else if (P == 1024) via_regs(std::integral_constant<int, 16>());
    else wave_sort(d, ix, P, lane);
}

// one wave per query (WAVES per block, no block-level synchronisation).  The instances share the work by list length: an
// instance takes the queries with lo < min(length, cap) <= hi (SELMAX = kSelSmall: nearly all; kSelMax: the few longer or
// overflowed ones; kSelHuge: lists beyond 1024 entries, which only occur with the large cap of k > 80).
template <int CP, int SELMAX, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_knn_select(const float* __restrict__ E, const int32_t* __restrict__ perm, int64_t M, int K,
                                                    int include_self, const int32_t* __restrict__ ccount, const int32_t* __restrict__ cbuf,
                                                    int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                    int32_t* __restrict__ n_overflow, int32_t* __restrict__ ovf_q, double* __restrict__ ovf_bound,
                                                    int cap, int lo, int hi) {
    __shared__ __attribute__((aligned(16))) double sd[WAVES][SELMAX];
    __shared__ int32_t si[WAVES][SELMAX];
    __shared__ __attribute__((aligned(16))) float sq[WAVES][CP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * WAVES + wave;
    if (q >= M) return;
    double* d = sd[wave];
    int32_t* ix = si[wave];
    float* qrow = sq[wave];
    for (int t = lane; t < CP; t += 64) qrow[t] = E[q * CP + t];
    const int cnt_all = ccount[q];
    const bool overflow = cnt_all > cap;
    const int cnt = overflow ? cap : cnt_all;
    if (!(cnt > lo && cnt <= hi) && !(cnt == 0 && lo == 0)) return;      // another instance's query
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // exact distances of the listed candidates, one per lane and step, then sort
    int P = 64;
    while (P < cnt) P <<= 1;
    auto candidate = [&](int t, double& dv, int32_t& iv) {
        dv = __builtin_huge_val();
        iv = 0x7fffffff;
        if (t < cnt) {
            const int64_t c = cbuf[q * (int64_t)cap + t];
            dv = exact_d2<CP>(qrow, E + c * CP);
            iv = perm[c];                            // ties are broken by the caller's point ids
        }
    };
    if (SELMAX == kSelSmall) {
        // the common instance keeps the window in registers; it only lands in LDS for the write-out / the overflow path
        auto sort_in_registers = [&](auto rtag) {
            constexpr int R = decltype(rtag)::value;
            double dr[R];
            int32_t ir[R];
#pragma unroll
            for (int r = 0; r < R; ++r) candidate(r * 64 + lane, dr[r], ir[r]);
            wave_sort_regs<R>(dr, ir, lane);
#pragma unroll
            for (int r = 0; r < R; ++r) { d[r * 64 + lane] = dr[r]; ix[r * 64 + lane] = ir[r]; }
        };
        if (P == 64) sort_in_registers(std::integral_constant<int, 1>());
        else if (P == 128) sort_in_registers(std::integral_constant<int, 2>());
        else sort_in_registers(std::integral_constant<int, 4>());
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        for (int t = lane; t < P; t += 64) {
            double dv;
            int32_t iv;
            candidate(t, dv, iv);
            d[t] = dv;
            ix[t] = iv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wave_sort_window(d, ix, P, lane);           // through registers up to 1024 entries
    }
    const int kept = cnt < K ? cnt : K;
    if (overflow) {
        // The list was cut at `cap` entries.  Its k-th exact distance is still an upper bound of the true k-th
        // distance: k_knn_rescan goes over every point with it.
        if (lane == 0) {
            const int slot = atomicAdd(n_overflow, 1);
            ovf_q[slot] = (int32_t)q;
            ovf_bound[slot] = d[K - 1];
        }
        return;
    }
    const int64_t qo = perm[q];                     // row of the caller's table
    for (int s = lane; s < K; s += 64) {
        const bool ok = s < kept && ix[s] != 0x7fffffff;
        idx_out[qo * K + s] = ok ? ix[s] : -1;
        dist_out[qo * K + s] = ok ? d[s] : __builtin_huge_val();
    }
}

// Queries whose candidate list overflowed: exact scan with the bound the select pass derived from the cut list, one workgroup
// of 16 waves per query, over the tiles the emit pass listed for the query's wave (every point within the query's bound T_q
// lies in one of them -- that is what the lists are -- so nothing beyond them can be among the K nearest).  Every wave takes
// a share of the tiles and keeps the points within the bound in its own LDS window (a full window is sorted, cut to the K
// best and its K-th distance becomes the wave's bound); the K nearest are among the waves' K best, which wave 0 merges.
// Same arithmetic and tie rule as the select pass.
constexpr int kCellDim = 32;                 // leading components the cells live in (zero padded)
constexpr int kRescanWaves = 16, kRescanWin = 512, kRescanChunk = 1024;
template <int CP>
__global__ void __launch_bounds__(64 * kRescanWaves) k_knn_rescan(const float* __restrict__ E, const int32_t* __restrict__ perm, int64_t M, int K, int include_self,
                                                                  const int32_t* __restrict__ n_overflow, const int32_t* __restrict__ ovf_q,
                                                                  const double* __restrict__ ovf_bound, const int32_t* __restrict__ elist,
                                                                  const int32_t* __restrict__ ecount, int64_t ecap, int BW,
                                                                  const int32_t* __restrict__ tilecell, const float* __restrict__ cenR, const float* __restrict__ invD,
                                                                  int Kc, int64_t ntiles, const unsigned* __restrict__ r2max, const float* __restrict__ St_lo,
                                                                  const float* __restrict__ St_hi, const float* __restrict__ tr_lo, const float* __restrict__ tr_hi,
                                                                  int32_t* __restrict__ idx_out, double* __restrict__ dist_out, long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
    double* sd = reinterpret_cast<double*>(rs_smem);                                        // [waves][win] + merge [waves * 256]
    int32_t* si = reinterpret_cast<int32_t*>(sd + kRescanWaves * kRescanWin + kRescanWaves * 256);
    float* sq = reinterpret_cast<float*>(si + kRescanWaves * kRescanWin + kRescanWaves * 256);   // [CP] query row
    __shared__ int s_fill[kRescanWaves];
    __shared__ int32_t s_tiles[kRescanChunk];
    __shared__ int s_ntile, s_kept;
    __shared__ float s_dq[1024], s_pq[1024];                    // the query against every cell centre: distance, dot product
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = *n_overflow;
    for (int it = blockIdx.x; it < n; it += gridDim.x) {
        const int64_t q = ovf_q[it];
        if (dbg && it < 4096 && threadIdx.x == 0) dbg[it * 8 + 0] = wall_clock64();
        for (int t = threadIdx.x; t < CP; t += blockDim.x) sq[t] = E[q * CP + t];
        __syncthreads();
        double* d = sd + wave * kRescanWin;
        int32_t* ix = si + wave * kRescanWin;
        double bound = ovf_bound[it];
        int fill = 0;
        // The bound is tighter than the T_q the lists were made for, and the query is one point, not a tile of 16: the cell
        // tests are worth repeating.  Against a tile of cell B (centre mu_B) a point within r of the query lies (a) at a distance
        // from mu_B within r of the query's, (b) along (mu_B - mu_A) / |mu_B - mu_A| within r of the query's projection.  The
        // tile's intervals are tr_lo / tr_hi (k_knn_slabs) and St_lo / St_hi; the query's side is computed here with the
        // margins of k_knn_slabs.  (a) is the one that tells for the usual customer, a query far from every centre.
        for (int c = threadIdx.x; c < Kc; c += blockDim.x) {
            const float* mu = cenR + (size_t)c * kCellDim;
            float pq = 0.f, dq = 0.f;
#pragma unroll
            for (int t = 0; t < kCellDim; ++t) {
                pq = fmaf(mu[t], sq[t], pq);
                const float v = sq[t] - mu[t];
                dq = fmaf(v, v, dq);
            }
            s_pq[c] = pq;
            s_dq[c] = sqrtf(dq);
        }
        if (threadIdx.x == 0) s_kept = 0;
        __syncthreads();
        const int cellA = tilecell[q >> 4];
        const float pA = s_pq[cellA];
        const float R2 = __uint_as_float(*r2max) * 1.01f;
        const float reach = (float)(sqrt(bound) * 1.00002) + 1e-30f;
        const float* iDrow = invD + (size_t)cellA * Kc;
        const float* stl = St_lo + (size_t)cellA * ntiles;
        const float* sth = St_hi + (size_t)cellA * ntiles;
        const int64_t blk = q / (BW * 16 * kEmitRT);
        const int qwave = (int)((q / (16 * kEmitRT)) % BW);
        const int32_t* lst = elist + blk * ecap;
        const int nent = ecount[blk];
        // The list is taken in chunks: the whole workgroup reads a chunk's entries (one coalesced load), keeps the tiles of the
        // query's wave packed in LDS, and the waves then go over those -- eight tiles (two points per lane) per wave and step,
        // so that the only dependent global loads of a step are the points' rows, two of them in flight per lane.
        auto push = [&](bool keep, double dv, int64_t c) {
            unsigned long long m = __ballot(keep);
            if (fill + __popcll(m) > kRescanWin) {       // sort, keep the K best, tighten the bound
                for (int t = fill + lane; t < kRescanWin; t += 64) { d[t] = __builtin_huge_val(); ix[t] = 0x7fffffff; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                wave_sort_window(d, ix, kRescanWin, lane);
                fill = K;
                bound = d[K - 1];
                keep = keep && dv <= bound;
                m = __ballot(keep);
            }
            if (keep) {
                const int pos = fill + __popcll(m & ((1ull << lane) - 1ull));
                d[pos] = dv;
                ix[pos] = perm[c];
            }
            fill += __popcll(m);
        };
        for (int base = 0; base < nent; base += kRescanChunk) {
            if (threadIdx.x == 0) s_ntile = 0;
            __syncthreads();
            for (int j = threadIdx.x; j < kRescanChunk; j += 64 * kRescanWaves) {
                const int e = base + j;
                bool on = false;
                int32_t tile = 0;
                if (e < nent) {
                    const unsigned v = (unsigned)lst[e];
                    on = ((v >> 24) >> qwave) & 1u;
                    tile = (int32_t)(v & 0xffffffu);
                    if (on) {
                        const int B = tilecell[tile];
                        const float dq = s_dq[B];
                        const float ga = fmaxf(dq * (1.f - 1e-5f) - tr_hi[tile] * (1.f + 1e-5f), tr_lo[tile] * (1.f - 1e-5f) - dq * (1.f + 1e-5f));
                        const float iD = iDrow[B];
                        const float f = (s_pq[B] - pA) * iD;
                        const float mg = 6e-6f * R2 * iD + 3e-7f * fabsf(f);
                        const float gb = fmaxf(f - mg + stl[tile], -sth[tile] - f - mg);
                        on = !(fmaxf(ga, gb) > reach);
                    }
                }
                const unsigned long long m = __ballot(on);
                int wbase = 0;
                if (lane == 0 && m) { wbase = atomicAdd(&s_ntile, __popcll(m)); if (dbg) atomicAdd(&s_kept, __popcll(m)); }
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                if (on) s_tiles[wbase + __popcll(m & ((1ull << lane) - 1ull))] = tile;
            }
            __syncthreads();
            const int nt = s_ntile;
            for (int t0 = wave * 8; t0 < nt; t0 += 8 * kRescanWaves) {
                const int ta = t0 + (lane >> 4), tb = ta + 4;
                int64_t ca = -1, cb = -1;
                if (ta < nt) ca = (int64_t)s_tiles[ta] * 16 + (lane & 15);
                if (tb < nt) cb = (int64_t)s_tiles[tb] * 16 + (lane & 15);
                const bool oka = ca >= 0 && ca < M && (include_self || ca != q);
                const bool okb = cb >= 0 && cb < M && (include_self || cb != q);
                // float32 first: a sum of rounded squares is within 34 * 2^-24 = 2e-6 of the float64 one, relatively, so anything
                // beyond the bound by more than 1e-5 cannot pass the exact test -- which most points then never reach
                const float* ra = E + (oka ? ca : 0) * CP;
                const float* rb = E + (okb ? cb : 0) * CP;
                float fa = 0.f, fb = 0.f;
#pragma unroll
                for (int t = 0; t < CP; t += 4) {
                    const f4 x = *reinterpret_cast<const f4*>(sq + t);
                    const f4 ya = *reinterpret_cast<const f4*>(ra + t);
                    const f4 yb = *reinterpret_cast<const f4*>(rb + t);
                    const f4 da = x - ya, db = x - yb;
                    fa = fmaf(da.x, da.x, fa); fa = fmaf(da.y, da.y, fa); fa = fmaf(da.z, da.z, fa); fa = fmaf(da.w, da.w, fa);
                    fb = fmaf(db.x, db.x, fb); fb = fmaf(db.y, db.y, fb); fb = fmaf(db.z, db.z, fb); fb = fmaf(db.w, db.w, fb);
                }
                const double lim = bound * 1.00001 + 1e-30;
                double dva = __builtin_huge_val(), dvb = __builtin_huge_val();
                bool ka = false, kb = false;
                if (oka && (double)fa <= lim) { dva = exact_d2<CP>(sq, ra); ka = dva <= bound; }
                if (okb && (double)fb <= lim) { dvb = exact_d2<CP>(sq, rb); kb = dvb <= bound; }
                push(ka, dva, ca);
                kb = kb && dvb <= bound;
                push(kb, dvb, cb);
            }
            __syncthreads();
        }
        if (dbg && it < 4096 && threadIdx.x == 0) { dbg[it * 8 + 1] = wall_clock64(); dbg[it * 8 + 4] = nent; dbg[it * 8 + 5] = s_kept; }
        int P = 64;
        while (P < fill) P <<= 1;
        for (int t = fill + lane; t < P; t += 64) { d[t] = __builtin_huge_val(); ix[t] = 0x7fffffff; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wave_sort_window(d, ix, P, lane);
        if (lane == 0) s_fill[wave] = fill < K ? fill : K;
        __syncthreads();
        if (dbg && it < 4096 && threadIdx.x == 0) dbg[it * 8 + 2] = wall_clock64();
        if (wave == 0) {
            double* md = sd + kRescanWaves * kRescanWin;
            int32_t* mi = si + kRescanWaves * kRescanWin;
            int total = 0;
            for (int w = 0; w < kRescanWaves; ++w) {
                const int nw = s_fill[w];
                for (int t = lane; t < nw; t += 64) { md[total + t] = sd[w * kRescanWin + t]; mi[total + t] = si[w * kRescanWin + t]; }
                total += nw;
            }
            int PM = 64;
            while (PM < total) PM <<= 1;
            for (int t = total + lane; t < PM; t += 64) { md[t] = __builtin_huge_val(); mi[t] = 0x7fffffff; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            wave_sort_window(md, mi, PM, lane);
            const int kept = total < K ? total : K;
            const int64_t qo = perm[q];
            for (int t = lane; t < K; t += 64) {
                const bool ok = t < kept && mi[t] != 0x7fffffff;
                idx_out[qo * K + t] = ok ? mi[t] : -1;
                dist_out[qo * K + t] = ok ? md[t] : __builtin_huge_val();
            }
        }
        __syncthreads();
        if (dbg && it < 4096 && threadIdx.x == 0) { dbg[it * 8 + 3] = wall_clock64(); dbg[it * 8 + 6] = blockIdx.x; }
    }
}

// ------------------------------------------------------------------------------------------------
// Other metrics (phenograph.cluster(primary_metric=...) behind dd.py:320-322: "manhattan" -> sklearn minkowski p = 1,
// "cosine" / "correlation" -> sklearn brute force): an exact scan in float64, one wave per query.  Not a fast path -- every
// query meets every point (M^2 C flop on the float64 VALU) -- but an exact one: candidates no farther than the current
// bound are collected in an LDS window; a full window is sorted, cut to the k best, and its k-th distance becomes the
// bound (so only the first windows see many candidates).  Ordering by (distance, index).  The distances are
//   manhattan   sum_c |a_c - b_c|
//   cosine      1 - a.b / (|a| |b|)     (rows are normalised first, a zero row stays zero: distance 1, as sklearn's normalize)
//   correlation cosine of the rows after subtracting their means
// dist2_out receives the squared distance (the convention of ddx_get_knn).
// ------------------------------------------------------------------------------------------------
// rows prepared for the scan: cosine -> x / |x|, correlation -> (x - mean) / |x - mean|, manhattan -> x     (float64)
__global__ void k_knn_generic_prepare(const float* __restrict__ emb, int64_t M, int C, int metric, double* __restrict__ out) {
#pragma clang fp contract(off)
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    double mean = 0.0;
    if (metric == 3) {
        for (int c = 0; c < C; ++c) mean += (double)emb[r * C + c];
        mean /= (double)C;
    }
    double n2 = 0.0;
    for (int c = 0; c < C; ++c) {
        const double v = (double)emb[r * C + c] - mean;
        n2 += v * v;
    }
    const double scale = (metric == 1) ? 1.0 : (n2 > 0.0 ? 1.0 / sqrt(n2) : 1.0);
    for (int c = 0; c < C; ++c) out[r * C + c] = ((double)emb[r * C + c] - mean) * scale;
}

template <int METRIC>   // 1 manhattan, 2 / 3 cosine on the prepared rows
__global__ void __launch_bounds__(256) k_knn_generic(const double* __restrict__ X, int64_t M, int C, int K, int include_self,
                                                     int32_t* __restrict__ idx_out, double* __restrict__ dist_out) {
#pragma clang fp contract(off)
    constexpr int WIN = 1024;
    __shared__ __attribute__((aligned(16))) double sd[4][WIN];
    __shared__ int32_t si[4][WIN];
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_smem[];
    double* sq = reinterpret_cast<double*>(gen_smem);              // [4][C] query rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= M) return;
    double* d = sd[wave];
    int32_t* ix = si[wave];
    double* qrow = sq + (size_t)wave * C;
    for (int t = lane; t < C; t += 64) qrow[t] = X[q * C + t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double bound = __builtin_huge_val();
    int fill = 0;
    for (int64_t c0 = 0; c0 < M; c0 += 64) {
        const int64_t c = c0 + lane;
        double dv = __builtin_huge_val();
        bool keep = false;
        if (c < M && (include_self || c != q)) {
            const double* row = X + c * C;
            double acc = 0.0;
            if (METRIC == 1) {
                for (int t = 0; t < C; ++t) acc = acc + fabs(qrow[t] - row[t]);
                dv = acc;
            } else {
                for (int t = 0; t < C; ++t) acc = acc + qrow[t] * row[t];
                dv = 1.0 - acc;
                if (dv < 0.0) dv = 0.0;                                 // (rounding: a row against itself)
            }
            keep = dv <= bound;
        }
        const unsigned long long m = __ballot(keep);
        const int n_new = __popcll(m);
        if (fill + n_new > WIN) {                                   // sort, keep the k best, tighten the bound
            for (int t = fill + lane; t < WIN; t += 64) { d[t] = __builtin_huge_val(); ix[t] = 0x7fffffff; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            wave_sort(d, ix, WIN, lane);
            fill = K;
            bound = d[K - 1];
            keep = keep && dv <= bound;
        }
        const unsigned long long m2 = __ballot(keep);
        if (keep) {
            const int pos = fill + __popcll(m2 & ((1ull << lane) - 1ull));
            d[pos] = dv;
            ix[pos] = (int32_t)c;
        }
        fill += __popcll(m2);
    }
    int P = 64;
    while (P < fill) P <<= 1;
    for (int t = fill + lane; t < P; t += 64) { d[t] = __builtin_huge_val(); ix[t] = 0x7fffffff; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    wave_sort(d, ix, P, lane);
    const int kept = fill < K ? fill : K;
    for (int s = lane; s < K; s += 64) {
        const bool ok = s < kept && ix[s] != 0x7fffffff;
        idx_out[q * K + s] = ok ? ix[s] : -1;
        dist_out[q * K + s] = ok ? d[s] * d[s] : __builtin_huge_val();
    }
}

int stage_knn_metric(ddx_ctx* ctx, int32_t k, int32_t include_self, int32_t metric) {
    const int64_t M = ctx->embM;
    const int C = ctx->C;
    if (k > 256) return set_err(ctx, DDX_E_UNSUPPORTED, "k=%d exceeds 256", k);
    DDX_TRY(ensure(ctx, ctx->pcaA, sizeof(double) * (size_t)M * C + 256));
    DDX_TRY(ensure(ctx, ctx->knn_idx, sizeof(int32_t) * (size_t)M * k));
    DDX_TRY(ensure(ctx, ctx->knn_dist, sizeof(double) * (size_t)M * k));
    double* X = ctx->pcaA.as<double>();
    ScopedTimer t(ctx, "knn_generic");
    k_knn_generic_prepare<<<(unsigned)ceil_div(M, 256), 256, 0, ctx->stream>>>(ctx->emb32.as<float>(), M, C, metric, X);
    const unsigned grid = (unsigned)ceil_div(M, 4);
    const size_t lds = sizeof(double) * 4 * (size_t)C;
    if (metric == 1)
        k_knn_generic<1><<<grid, 256, lds, ctx->stream>>>(X, M, C, k, include_self, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<double>());
    else
        k_knn_generic<2><<<grid, 256, lds, ctx->stream>>>(X, M, C, k, include_self, ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<double>());
    DDX_HIP(ctx, hipGetLastError());
    ctx->knn_window_total = nullptr;
    ctx->knn_overflow = nullptr;
    ctx->knn_ccount = nullptr;
    ctx->knn_perm = nullptr;
    ctx->K = k;
    ctx->knn_self = include_self != 0;
    ctx->have_knn = true;
    return DDX_OK;
}

// ------------------------------------------------------------------------------------------------
// Cells: the pruning structure of the emit pass.
//
// d(q, c) >= |u.q - u.c| for ANY vector u with |u| <= 1.  The points are grouped into Kc cells (a few rounds of Lloyd's
// k-means on the 32 leading components, started from evenly spaced ranks of the first-component order) and laid out cell
// by cell (cells in the order of their centres' first component, first-component order inside a cell).  A 16-point tile
// has a nominal cell O (the cell of its first point).  For every tile t and every cell c the table S holds the interval
// of  f_{O,c}(x) = u_{O,c}.x ,  u_{O,c} = (mu_c - mu_O) / |mu_c - mu_O| ,  over the points x of t  (lo / hi, widened by the
// float32 rounding of the evaluation).  A query tile s (cell A) against a candidate tile t (cell B) along u_{A,B}:
// the queries lie in [S_lo[s][B], S_hi[s][B]], the candidates in [-S_hi[t][A], -S_lo[t][A]] (u_{B,A} = -u_{A,B});
// when the gap between the two intervals exceeds sqrt(max T_q of the query tile), no candidate of t can be within
// reach of a query of s and the pair of tiles is never screened.  The direction between the two cell centres carries
// their whole separation while a tile's extent along it is one coordinate's worth of spread: on the benchmark
// embedding (12 cell types, 18 noise components) 15-20 % of the tile pairs survive, against 68 % for windows on the
// first component alone -- bounding boxes or balls over the leading components prune next to nothing there
// (profiles/r04_knn_prune_study.txt).  The test along the first component itself is kept beside it (it separates tiles
// of one cell).  None of this can change the result: the screen stays conservative whatever the cells look like.
// ------------------------------------------------------------------------------------------------
constexpr float kCellFix = 1048576.0f;       // fixed-point grid of the centre sums (2^20): integer sums are exact in any order
constexpr int kCellRounds = 2;               // Lloyd rounds (on every kCellSub-th point of the first-component order) before the assignment
constexpr int kCellSub = 4;

__device__ __forceinline__ unsigned ordered_bits(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void k_cells_init(const float* __restrict__ emb, const int32_t* __restrict__ perm1, int64_t M, int C, int Kc,
                             float* __restrict__ cen, float* __restrict__ cn) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Kc) return;
    int64_t pos = ((int64_t)c * M) / Kc + M / (2 * (int64_t)Kc);
    if (pos >= M) pos = M - 1;
    const int64_t src = perm1[pos];
    float n = 0.f;
    for (int d = 0; d < kCellDim; ++d) {
        const float v = d < C ? emb[src * C + d] : 0.f;
        cen[c * kCellDim + d] = v;
        n = fmaf(v, v, n);
    }
    cn[c] = n;
}

// nearest centre of point ids[stride * i], i < n (ties: the smaller cell id): lane = point, the four waves of a block share
// the cells out.  label / dcen (squared distance to that centre, >= 0) are indexed by the point's id.
__global__ void __launch_bounds__(256) k_cells_assign(const float* __restrict__ emb, const int32_t* __restrict__ ids, int64_t n, int stride, int C,
                                                      const float* __restrict__ cen, const float* __restrict__ cn, int Kc,
                                                      int32_t* __restrict__ label, float* __restrict__ dcen) {
    __shared__ unsigned long long best[4][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const int64_t r = i < n ? ids[i * stride] : 0;
    float x[kCellDim];
    float xn = 0.f;
#pragma unroll
    for (int d = 0; d < kCellDim; ++d) {
        x[d] = (i < n && d < C) ? emb[r * C + d] : 0.f;
        xn = fmaf(x[d], x[d], xn);
    }
    float bs = __builtin_huge_valf();
    int bc = 0x7fffffff;
    for (int c = w; c < Kc; c += 4) {
        const float* mu = cen + (size_t)c * kCellDim;
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < kCellDim; ++d) dot = fmaf(mu[d], x[d], dot);
        const float s = fmaf(-2.f, dot, cn[c]);
        if (s < bs) { bs = s; bc = c; }
    }
    best[w][lane] = ((unsigned long long)ordered_bits(bs) << 32) | (unsigned)bc;
    __syncthreads();
    if (w == 0 && i < n) {
        unsigned long long b = best[0][lane];
#pragma unroll
        for (int o = 1; o < 4; ++o) b = best[o][lane] < b ? best[o][lane] : b;
        int c = (int)(unsigned)(b & 0xffffffffull);
        if (c >= Kc) c = 0;                                   // (NaN coordinates: validated away upstream)
        label[r] = c;
        if (dcen) {
            const unsigned ob = (unsigned)(b >> 32);
            const float s = __uint_as_float((ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob);
            dcen[r] = fmaxf(s + xn, 0.f);
        }
    }
}

// (thread = (point, component): the 32 lanes of a point add to 32 consecutive sums of its cell)
__global__ void k_cells_accumulate(const float* __restrict__ emb, const int32_t* __restrict__ ids, int64_t n, int stride, int C,
                                   const int32_t* __restrict__ label, unsigned long long* __restrict__ sums, int32_t* __restrict__ counts) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t / kCellDim;
    const int d = (int)(t % kCellDim);
    if (i >= n) return;
    const int64_t r = ids[i * stride];
    const int c = label[r];
    if (d < C) {
        const long long v = (long long)rintf(emb[r * C + d] * kCellFix);
        atomicAdd(&sums[(size_t)c * kCellDim + d], (unsigned long long)v);     // two's complement: exact in any order
    }
    if (d == 0) atomicAdd(&counts[c], 1);
}

// centre = mean of the members (an empty cell keeps its centre); clears the sums for the next round
__global__ void k_cells_mean(unsigned long long* __restrict__ sums, int32_t* __restrict__ counts, int Kc, float* __restrict__ cen,
                             float* __restrict__ cn) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Kc) return;
    const int n = counts[c];
    float nn = 0.f;
    for (int d = 0; d < kCellDim; ++d) {
        float v = cen[c * kCellDim + d];
        if (n > 0) v = (float)((double)(long long)sums[(size_t)c * kCellDim + d] / ((double)n * (double)kCellFix));
        cen[c * kCellDim + d] = v;
        nn = fmaf(v, v, nn);
        sums[(size_t)c * kCellDim + d] = 0ull;
    }
    cn[c] = nn;
    counts[c] = 0;
}

// cells ordered by the first component of their centres (ties by id): rank, centres re-indexed by rank
__global__ void __launch_bounds__(1024) k_cells_rank(const float* __restrict__ cen, int Kc, int32_t* __restrict__ rank, float* __restrict__ cenR) {
    __shared__ float key[1024];
    const int c = threadIdx.x;
    if (c < Kc) key[c] = cen[c * kCellDim];
    __syncthreads();
    if (c >= Kc) return;
    int r = 0;
    const float k = key[c];
    for (int o = 0; o < Kc; ++o) r += (key[o] < k || (key[o] == k && o < c)) ? 1 : 0;
    rank[c] = r;
    for (int d = 0; d < kCellDim; ++d) cenR[r * kCellDim + d] = cen[c * kCellDim + d];
}

__global__ void k_cells_keys(const int32_t* __restrict__ order, const int32_t* __restrict__ label, const int32_t* __restrict__ rank, int64_t M,
                             uint32_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    keys[i] = (uint32_t)rank[label[order[i]]];
}

// 1 / |mu_B - mu_A|, shrunk by 1e-5 so that the direction is no longer than 1 whatever the float32 roundings; 0 on the diagonal
__global__ void k_cells_invdist(const float* __restrict__ cenR, int Kc, float* __restrict__ invD) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)Kc * Kc) return;
    const int a = (int)(t / Kc), b = (int)(t % Kc);
    float d2 = 0.f;
    for (int d = 0; d < kCellDim; ++d) {
        const float v = cenR[b * kCellDim + d] - cenR[a * kCellDim + d];
        d2 = fmaf(v, v, d2);
    }
    invD[t] = (a != b && d2 > 0.f) ? (1.0f - 1e-5f) / sqrtf(d2) : 0.f;
}

// nominal cell and first-component interval of every tile; largest squared norm (float bits, non-negative: integer max)
__global__ void k_knn_tileinfo(const uint32_t* __restrict__ cellpos, const float* __restrict__ p1, const float* __restrict__ nrm, int64_t M,
                               int64_t ntiles, int32_t* __restrict__ tilecell, float* __restrict__ tp1lo, float* __restrict__ tp1hi,
                               unsigned* __restrict__ r2max) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf(), nmax = 0.f;
    for (int j = 0; j < 16; ++j) {
        const int64_t r = t * 16 + j;
        if (r >= M) break;
        lo = fminf(lo, p1[r]);
        hi = fmaxf(hi, p1[r]);
        nmax = fmaxf(nmax, nrm[r]);
    }
    tilecell[t] = t * 16 < M ? (int32_t)cellpos[t * 16] : 0;
    tp1lo[t] = lo;
    tp1hi[t] = hi;
    if (nmax > 0.f) atomicMax(r2max, __float_as_uint(nmax));
}

// S[t][c] (and its transpose St[c][t]): interval of u_{O(t),c}.x over the points of tile t, see above.
// Rounding: every dot product is a 32-term float32 fma chain, |error| <= 1.92e-6 |mu| |x| <= 1.92e-6 R2 (R2 = largest squared
// norm; the centres are means of points); two of them, times 1/D, plus the roundings of the subtraction and the
// product (<= 1.2e-7 |f|): the margin 6e-6 R2 / D + 3e-7 |f| covers it with room to spare.
// Block = 4 waves = 16 tiles; lane = point.  The intervals of 64 cells are collected in LDS and written out in rows.
__global__ void __launch_bounds__(256) k_knn_slabs(const float* __restrict__ E, int CP, const float* __restrict__ nrm, const int32_t* __restrict__ tilecell,
                                                   const float* __restrict__ cenR, const float* __restrict__ invD, int Kc, int64_t ntiles,
                                                   const unsigned* __restrict__ r2max, float* __restrict__ S_lo, float* __restrict__ S_hi,
                                                   float* __restrict__ St_lo, float* __restrict__ St_hi, float* __restrict__ tr_lo, float* __restrict__ tr_hi) {
    __shared__ float slo[16][65], shi[16][65];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t r = (int64_t)blockIdx.x * 256 + tid;
    const int64_t tile = r >> 4;
    const int tl = tid >> 4;                                   // tile of the block
    const bool pad = !(nrm[r] < __builtin_huge_valf());
    float x[kCellDim];
#pragma unroll
    for (int d = 0; d < kCellDim; ++d) x[d] = E[r * CP + d];
    const int O = tilecell[tile];
    float pO = 0.f;
    {
        const float* mu = cenR + (size_t)O * kCellDim;
        float dO = 0.f;
#pragma unroll
        for (int d = 0; d < kCellDim; ++d) {
            pO = fmaf(mu[d], x[d], pO);
            const float v = x[d] - mu[d];
            dO = fmaf(v, v, dO);
        }
        // distance from the tile's own centre (first kCellDim components: a projection, so never more than the full distance);
        // a sum of 32 rounded squares of rounded differences is within 5e-6 of the true one, relatively: the users widen by 1e-5
        dO = sqrtf(dO);
        float lo = pad ? __builtin_huge_valf() : dO, hi = pad ? -__builtin_huge_valf() : dO;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, 64));
            hi = fmaxf(hi, __shfl_xor(hi, o, 64));
        }
        if ((lane & 15) == 0 && tile < ntiles) { tr_lo[tile] = lo; tr_hi[tile] = hi; }
    }
    const float R2 = __uint_as_float(*r2max) * 1.01f;
    const float* iDrow = invD + (size_t)O * Kc;
    for (int c0 = 0; c0 < Kc; c0 += 64) {
        const int cend = (Kc - c0) < 64 ? (Kc - c0) : 64;
        for (int cc = 0; cc < cend; ++cc) {
            const int c = c0 + cc;
            const float* mu = cenR + (size_t)c * kCellDim;
            float p = 0.f;
#pragma unroll
            for (int d = 0; d < kCellDim; ++d) p = fmaf(mu[d], x[d], p);
            const float iD = iDrow[c];
            const float f = (p - pO) * iD;
            const float m = 6e-6f * R2 * iD + 3e-7f * fabsf(f);
            float lo = pad ? __builtin_huge_valf() : f - m;
            float hi = pad ? -__builtin_huge_valf() : f + m;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                lo = fminf(lo, __shfl_xor(lo, o, 64));
                hi = fmaxf(hi, __shfl_xor(hi, o, 64));
            }
            if ((lane & 15) == 0) { slo[tl][cc] = lo; shi[tl][cc] = hi; }
        }
        __syncthreads();
        // natural layout: 16 rows of `cend` consecutive cells
        for (int e = tid; e < 16 * 64; e += 256) {
            const int t = e >> 6, cc = e & 63;
            const int64_t tg = (int64_t)blockIdx.x * 16 + t;
            if (cc < cend && tg < ntiles) {
                S_lo[tg * Kc + c0 + cc] = slo[t][cc];
                S_hi[tg * Kc + c0 + cc] = shi[t][cc];
            }
        }
        // transposed layout: per cell 16 consecutive tiles
        for (int e = tid; e < 16 * 64; e += 256) {
            const int cc = e >> 4, t = e & 15;
            const int64_t tg = (int64_t)blockIdx.x * 16 + t;
            if (cc < cend && tg < ntiles) {
                St_lo[(size_t)(c0 + cc) * ntiles + tg] = slo[t][cc];
                St_hi[(size_t)(c0 + cc) * ntiles + tg] = shi[t][cc];
            }
        }
        __syncthreads();
    }
}

// Per emit block (BW waves of 2 query tiles): the list of candidate tiles that at least one of its waves has to screen,
// elist = tile | waves << 24 (tiles ascending), padded with mask-0 entries to a multiple of G (the emit pass's step).
// A lane tests one candidate tile against its wave's two query tiles; a round covers 64 * BW tiles.
template <int BW>
__global__ void __launch_bounds__(64 * BW) k_knn_tilelists(const float* __restrict__ S_lo, const float* __restrict__ S_hi, const float* __restrict__ St_lo,
                                                           const float* __restrict__ St_hi, const int32_t* __restrict__ tilecell,
                                                           const float* __restrict__ tp1lo, const float* __restrict__ tp1hi, const float* __restrict__ thr,
                                                           const float* __restrict__ nrm, int64_t ntiles, int Kc, int G, int32_t* __restrict__ elist,
                                                           int32_t* __restrict__ ecount, int64_t ecap,
                                                           unsigned long long* __restrict__ total) {
    __shared__ unsigned long long wm[BW][BW];
    __shared__ int wcnt[BW];
    __shared__ int s_last;
    if (threadIdx.x == 0) s_last = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t blk = blockIdx.x;
    const int64_t s0 = (blk * BW + wave) * kEmitRT;                  // first query tile of the wave
    static_assert(16 * kEmitRT <= 64, "a lane per query of the wave");
    // reach of the wave's query tiles: lanes 16 u .. 16 u + 15 hold tile s0 + u
    float rs[kEmitRT];
    {
        float r = -__builtin_huge_valf();
        if (lane < 16 * kEmitRT) {
            const int64_t q = s0 * 16 + lane;
            if (nrm[q] < __builtin_huge_valf()) {
                const float T = thr[q];
                r = T < __builtin_huge_valf() ? sqrtf(fmaxf(T, 0.f)) * 1.00001f + 1e-30f : __builtin_huge_valf();
            }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) r = fmaxf(r, __shfl_xor(r, o, 64));
#pragma unroll
        for (int u = 0; u < kEmitRT; ++u) rs[u] = __shfl(r, 16 * u, 64);
    }
    int A[kEmitRT];
    float q1lo[kEmitRT], q1hi[kEmitRT];
#pragma unroll
    for (int u = 0; u < kEmitRT; ++u) {
        A[u] = tilecell[s0 + u];
        q1lo[u] = tp1lo[s0 + u];
        q1hi[u] = tp1hi[s0 + u];
    }
    int32_t* lst = elist + blk * ecap;
    int count = 0;
    unsigned long long screened = 0;
    for (int64_t base = 0; base < ntiles; base += 64 * BW) {
#pragma unroll
        for (int j = 0; j < BW; ++j) {
            const int64_t t = base + 64 * j + lane;
            bool pass = false;
            if (t < ntiles) {
                const int B = tilecell[t];
                const float c1l = tp1lo[t], c1h = tp1hi[t];
#pragma unroll
                for (int u = 0; u < kEmitRT; ++u) {
                    const float a = St_lo[(size_t)A[u] * ntiles + t], b = St_hi[(size_t)A[u] * ntiles + t];
                    const float ql = S_lo[(s0 + u) * Kc + B], qh = S_hi[(s0 + u) * Kc + B];
                    const float gap = fmaxf(fmaxf(-b - qh, ql + a), fmaxf(c1l - q1hi[u], q1lo[u] - c1h));
                    pass = pass || !(gap > rs[u]);
                }
            }
            const unsigned long long m = __ballot(pass);
            if (lane == 0) wm[wave][j] = m;
        }
        __syncthreads();
        // thread (wave j, lane) owns tile base + 64 j + lane: collect the waves that screen it
        unsigned mask = 0;
#pragma unroll
        for (int w = 0; w < BW; ++w) mask |= (unsigned)((wm[w][wave] >> lane) & 1ull) << w;
        const unsigned long long anyb = __ballot(mask != 0u);
        if (lane == 0) wcnt[wave] = __popcll(anyb);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < BW; ++w) {
            const int c = wcnt[w];
            before += w < wave ? c : 0;
            all += c;
        }
        if (mask != 0u) {
            const int64_t t = base + 64 * wave + lane;
            const int slot = count + before + __popcll(anyb & ((1ull << lane) - 1ull));
            lst[slot] = (int32_t)((uint32_t)t | mask << 24);
            screened += (unsigned long long)__popc(mask);
        }
        if (mask != 0u) atomicMax(&s_last, (int)(base + 64 * wave + lane));      // the padding repeats the last listed tile
        count += all;
        __syncthreads();
    }
    const int lt_all = s_last;
    const int padded = (count + G - 1) / G * G;
    if (tid < padded - count) lst[count + tid] = lt_all;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) screened += __shfl_xor(screened, o, 64);
    if (lane == 0) atomicAdd(total, screened);                       // statistics only (bench.py's flop count)
    if (tid == 0) ecount[blk] = padded;
}

// Cells by distance from cell A (one block per cell, Kc <= 1024 threads' worth): nlist[A][*] = tiles of the other cells, nearest
// cell first, up to `budget` tiles; ncount[A] = how many.  ctile[c] = first tile whose nominal cell is c (ctile[Kc] = number of
// real tiles).
__global__ void __launch_bounds__(1024) k_cells_neighbours(const float* __restrict__ cenR, const int32_t* __restrict__ ctile, int Kc, int budget,
                                                           int32_t* __restrict__ nlist, int32_t* __restrict__ ncount) {
    __shared__ float dist[1024];
    __shared__ int order[1024];
    __shared__ int cum[1025];
    const int A = blockIdx.x, c = threadIdx.x;
    if (c < Kc) {
        float d2 = 0.f;
        for (int d = 0; d < kCellDim; ++d) {
            const float v = cenR[c * kCellDim + d] - cenR[A * kCellDim + d];
            d2 = fmaf(v, v, d2);
        }
        dist[c] = c == A ? -1.f : d2;
    }
    __syncthreads();
    if (c < Kc) {
        const float k = dist[c];
        int r = 0;
        for (int o = 0; o < Kc; ++o) r += (dist[o] < k || (dist[o] == k && o < c)) ? 1 : 0;
        order[r] = c;                                               // order[0] = A
    }
    __syncthreads();
    // inclusive prefix of the tile counts in that order (cell A itself counts nothing)
    int mine = 0;
    if (c < Kc && c > 0) { const int cell = order[c]; mine = ctile[cell + 1] - ctile[cell]; }
    cum[c + 1] = mine;
    if (c == 0) cum[0] = 0;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int add = c + 1 > o ? cum[c + 1 - o] : 0;
        __syncthreads();
        cum[c + 1] += add;
        __syncthreads();
    }
    if (c < Kc && c > 0) {
        const int cell = order[c], start = cum[c];                   // tiles of the cells before this one
        const int t0 = ctile[cell], nt = ctile[cell + 1] - t0;
        for (int j = 0; j < nt && start + j < budget; ++j) nlist[(size_t)A * budget + start + j] = t0 + j;
    }
    if (c == 0) ncount[A] = cum[Kc] < budget ? cum[Kc] : budget;
}

// Sample of every bound block (kBoundWaves x 32 queries, which may lie in two or three cells): the tiles of its own cell(s) around it
// (a cell is in first-component order), then the tiles of the cells nearest to its first and to its last cell, taken in turn,
// nsamp tiles in all where there are that many.  No tile may be listed twice (the k-th smallest of a multiset is not a
// bound): a bitmap over the tiles in LDS (dynamic, ceil(ntiles / 32) words) keeps track.
__global__ void __launch_bounds__(64) k_knn_boundlists(const int32_t* __restrict__ tilecell, const int32_t* __restrict__ ctile, const int32_t* __restrict__ nlist,
                                                       const int32_t* __restrict__ ncount, int Kc, int budget, int nsamp, int ntr, int every,
                                                       int32_t* __restrict__ blist, int32_t* __restrict__ bcount) {
    extern __shared__ unsigned seen[];
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const int t_blk = (int)(b * kBoundWaves * kBoundRT);
    const int t_end = min(t_blk + kBoundWaves * kBoundRT, ntr);              // real tiles of the block
    for (int i = lane; i < (ntr + 31) / 32; i += 64) seen[i] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int32_t* dst = blist + b * (int64_t)nsamp;
    if (t_blk >= ntr) {                                            // a block of padding queries
        if (lane == 0) bcount[b] = 0;
        return;
    }
    const int A0 = tilecell[t_blk], A1 = tilecell[t_end - 1];
    const int ts = ctile[A0], te = ctile[A1 + 1];
    const int others = Kc > 1 ? ncount[A0] + (A1 != A0 ? ncount[A1] : 0) : 0;
    int own = te - ts;
    if (own > nsamp) own = nsamp;
    if (others >= nsamp / 2 && own > nsamp / 2) own = nsamp / 2;     // big cells leave half of the sample to their neighbours
    int lo = (t_blk + t_end) / 2 - own / 2;
    if (lo > te - own) lo = te - own;
    if (lo < ts) lo = ts;
    for (int i = lane; i < own; i += 64) {
        dst[i] = lo + i;
        atomicOr(&seen[(lo + i) >> 5], 1u << ((lo + i) & 31));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int count = own;
    // every `every`-th tile of the whole set: whatever the cells look like, a uniform sample of a 1/every share of the points
    // bounds the k-th distance by the (k * every)-th one -- it is what keeps a query that lies far from its own cell from
    // collecting tens of thousands of candidates
    if (Kc > 1 && every > 0) {
        for (int base = (int)(b % every); base < ntr && count < nsamp; base += 64 * every) {
            const int t = base + lane * every;
            bool keep = false;
            if (t < ntr) {
                const unsigned bit = 1u << (t & 31);
                keep = (atomicOr(&seen[t >> 5], bit) & bit) == 0u;
            }
            const unsigned long long m = __ballot(keep);
            const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
            if (keep && pos < nsamp) dst[pos] = t;
            count += __popcll(m);
            if (count > nsamp) count = nsamp;
        }
    }
    if (Kc > 1) {
        const int n0 = ncount[A0], n1 = A1 != A0 ? ncount[A1] : 0;
        const int32_t* l0 = nlist + (size_t)A0 * budget;
        const int32_t* l1 = nlist + (size_t)A1 * budget;
        for (int base = 0; (base < n0 || base < n1) && count < nsamp; base += 64) {
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const int n = which ? n1 : n0;
                const int32_t* l = which ? l1 : l0;
                bool keep = false;
                int e = 0;
                if (base + lane < n) {
                    e = l[base + lane];
                    const int t = e;
                    const unsigned bit = 1u << (t & 31);
                    keep = (atomicOr(&seen[t >> 5], bit) & bit) == 0u;       // tiles of one list are distinct: no two lanes race for a bit
                }
                const unsigned long long m = __ballot(keep);
                const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                if (keep && pos < nsamp) dst[pos] = e;
                count += __popcll(m);
                if (count > nsamp) count = nsamp;
            }
        }
    }
    if (lane == 0) bcount[b] = count;
}

// sort key of the point order: the first principal component
__global__ void k_knn_keys(const float* __restrict__ emb, int64_t M, int C, float* __restrict__ keys, int32_t* __restrict__ ids) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    keys[r] = emb[r * C];
    ids[r] = (int32_t)r;
}

// number of cells for M points: cells of about 768 points (48 tiles), at most 512; small inputs keep the plain
// first-component order (one cell: the tile test reduces to the first-component windows).  Measured, all stages of a search
// in ms (profiles/r04_knn_notes.txt): 125 k points 64 / 128 / 192 / 256 / 384 cells 3.15 / 3.11 / 3.09 / 3.23 / 3.72; 625 k points
// 256 / 384 / 512 / 640 / 1024 cells 30.5 / 30.0 / 29.9 / 30.5 / 33.8 (two rounds of k-means place a thousand centres worse than
// five hundred, and the per-tile tables grow with the cell count)
static int default_cells(int64_t M) {
    if (M < 16384) return 1;
    const int64_t want = ceil_div(ceil_div(M, 768), 64) * 64;
    return (int)std::min<int64_t>(512, std::max<int64_t>(64, want));
}

// candidates the emit pass listed for every query, in the caller's point order (statistics for the parity tests)
__global__ void k_knn_counts_by_id(const int32_t* __restrict__ ccount, const int32_t* __restrict__ perm, int64_t M, int32_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < M) out[perm[r]] = ccount[r];
}

int stage_knn_candidate_counts(ddx_ctx* ctx, int32_t* host_out) {
    if (!ctx->knn_ccount || !ctx->knn_perm) return set_err(ctx, DDX_E_ARG, "no euclidean kNN result");
    const int64_t M = ctx->embM;
    DDX_TRY(ensure(ctx, ctx->sort_vals_out, sizeof(int32_t) * (size_t)M));
    k_knn_counts_by_id<<<(unsigned)ceil_div(M, 256), 256, 0, ctx->stream>>>(ctx->knn_ccount, ctx->knn_perm, M, ctx->sort_vals_out.as<int32_t>());
    DDX_HIP(ctx, hipMemcpyAsync(host_out, ctx->sort_vals_out.p, sizeof(int32_t) * (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));
    return DDX_OK;
}

// first tile whose nominal cell is >= c (the nominal cells ascend along the tiles): ctile[0..Kc], ctile[Kc] = real tiles
__global__ void k_cells_tilestart(const int32_t* __restrict__ tilecell, int ntr, int Kc, int32_t* __restrict__ ctile) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > Kc) return;
    int a = 0, b = ntr;
    while (a < b) { const int m = (a + b) >> 1; if (tilecell[m] < c) a = m + 1; else b = m; }
    ctile[c] = a;
}

int stage_knn(ddx_ctx* ctx, int32_t k, int32_t include_self) {
    const int64_t M = ctx->embM;
    const int C = ctx->C;
    if (C > kMaxDim) return set_err(ctx, DDX_E_UNSUPPORTED, "embedding dimension %d exceeds %d", C, kMaxDim);
    constexpr int kMaxK = 256;
    if (k > kMaxK) return set_err(ctx, DDX_E_UNSUPPORTED, "k=%d exceeds %d", k, kMaxK);
    if (M >= ((int64_t)1 << 24)) return set_err(ctx, DDX_E_UNSUPPORTED, "kNN of %lld points (the bound lists' bitmap holds 2^24)", (long long)M);
    // k beyond what one bound launch ranks (16 lanes x (kBoundKeepLarge - 1) kept values per query): the sample is
    // dealt out over `groups` launches, each bounding the ceil(k / groups)-th neighbour among its share (k_knn_bound_bf)
    const int groups = (int)ceil_div(k, 16 * (kBoundKeepLarge - 1));
    const int k_bound = (int)ceil_div(k, groups);
    const bool keep_small = k_bound <= 16 * (kBoundKeepSmall - 1);
    const int cap = k <= 16 * (kBoundKeepLarge - 1) ? kCandCap : kCandCapLarge;      // candidate slots per query
    const int CP = (C <= 32) ? 32 : (C <= 64 ? 64 : 128);
    const int G = chunk_tiles(CP);                               // tiles per staged step
    const int64_t Mp = ceil_div(M, 512) * 512;                 // whole blocks of queries in both MFMA passes
    const int64_t ntiles = Mp >> 4;
    const int ntr = (int)ceil_div(M, 16);                        // tiles that hold points
    int Kc = ctx->opt.knn_cells > 0 ? ctx->opt.knn_cells : default_cells(M);
    if (Kc > 1024) Kc = 1024;
    if ((int64_t)Kc > M) Kc = (int)std::max<int64_t>(1, M);
    int BW = ctx->opt.knn_emit_waves;                            // waves per emit block (they share the staged tiles)
    if (BW != 4 && BW != 8) BW = kEmitWaves;                      // (a list entry carries 8 wave bits)
    const int64_t emit_blocks = Mp / (BW * 16 * kEmitRT);
    const int64_t bound_blocks = Mp / (kBoundWaves * 16 * kBoundRT);
    const int xcd_chunk = ctx->opt.knn_xcd_chunk;                                  // 0: workgroups in launch order
    const bool fold = ctx->opt.knn_fold && C <= 30;                                // threshold folded into the operands (k_knn_fold)
    // sample of the bound pass: grows with the point count (1/24 of the tiles, at least 512) -- a fixed-size subset would hold
    // an ever smaller share of the true neighbours, T_q would loosen and the candidate lists overflow
    int64_t nsamp = std::max<int64_t>(512, ntiles / 24);          // (625 k points: 2442 / 1800 / 1536 / 1280 tiles -> 25.5 / 24.9 / 24.7 / 27.0 ms per search)
    if (ctx->opt.knn_sample_tiles > 0) nsamp = ctx->opt.knn_sample_tiles;
    if (nsamp < 2 * (int64_t)ceil_div(k, 16) + 8) nsamp = 2 * (int64_t)ceil_div(k, 16) + 8;
    if (nsamp > ntr) nsamp = ntr;
    // workspace (reuses the PCA row buffer): E [Mp*CP] | Eb [Mp*CP as bf16 hi+lo] | nrm [Mp] | thr [Mp] | p1 [Mp] | keys [2*Mp]
    //            | start4 [4*Mp] | ccount [Mp+64] | ids [2*Mp] | cbuf [Mp*cap] | Ebq
    const size_t f_words = (size_t)Mp * CP * 2 + 9 * (size_t)Mp + 16;
    const size_t i_words = (size_t)Mp + 64 + 2 * (size_t)Mp + 64 + (size_t)Mp * cap + (fold ? (size_t)Mp * 32 + 16 : 0);
    DDX_TRY(ensure(ctx, ctx->pcaA, sizeof(float) * f_words + sizeof(int32_t) * i_words + 256));
    DDX_TRY(ensure(ctx, ctx->knn_idx, sizeof(int32_t) * (size_t)M * k));
    DDX_TRY(ensure(ctx, ctx->knn_dist, sizeof(double) * (size_t)M * k));
    float* E = ctx->pcaA.as<float>();
    __bf16* Eb = reinterpret_cast<__bf16*>(E + (size_t)Mp * CP);       // 2 parts x 2 bytes = CP floats per point
    float* nrm = E + 2 * (size_t)Mp * CP;
    float* thr = nrm + Mp;
    float* p1 = thr + Mp;
    float* keys_in = p1 + Mp;
    float* keys_out = keys_in + Mp;
    f4* start4 = reinterpret_cast<f4*>(keys_out + Mp + ((4 - ((2 * (size_t)Mp * CP + 5 * (size_t)Mp) & 3)) & 3));   // 16-byte aligned
    int32_t* ccount = reinterpret_cast<int32_t*>(reinterpret_cast<float*>(start4) + 4 * (size_t)Mp);   // [Mp] + overflow counter at [Mp]
    int32_t* ids_in = ccount + Mp + 64;
    int32_t* perm = ids_in + Mp;
    int32_t* cbuf = perm + Mp + 64;
    __bf16* Ebq = reinterpret_cast<__bf16*>((reinterpret_cast<uintptr_t>(cbuf + (size_t)Mp * cap) + 15) & ~(uintptr_t)15);   // query operands of the folded emit pass
    // cell work space (a buffer of its own)
    const int64_t ecap = ntiles + 8;
    size_t cw = 0;
    auto carve = [&](size_t bytes) { const size_t o = cw; cw += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_label = carve(sizeof(int32_t) * (size_t)M), o_kin = carve(sizeof(uint32_t) * (size_t)M), o_cellpos = carve(sizeof(uint32_t) * (size_t)M);
    const size_t o_perm1 = carve(sizeof(int32_t) * (size_t)M);
    const size_t o_cen = carve(sizeof(float) * (size_t)Kc * kCellDim), o_cenR = carve(sizeof(float) * (size_t)Kc * kCellDim), o_cn = carve(sizeof(float) * (size_t)Kc);
    const size_t o_sums = carve(sizeof(unsigned long long) * (size_t)Kc * kCellDim), o_counts = carve(sizeof(int32_t) * (size_t)Kc + 16);
    const size_t o_rank = carve(sizeof(int32_t) * (size_t)Kc), o_invD = carve(sizeof(float) * (size_t)Kc * Kc);
    const size_t o_ctile = carve(sizeof(int32_t) * ((size_t)Kc + 1)), o_ncount = carve(sizeof(int32_t) * (size_t)Kc), o_nlist = carve(sizeof(int32_t) * (size_t)Kc * nsamp);
    const size_t o_tilecell = carve(sizeof(int32_t) * (size_t)ntiles), o_t1lo = carve(sizeof(float) * (size_t)ntiles), o_t1hi = carve(sizeof(float) * (size_t)ntiles);
    const size_t o_trlo = carve(sizeof(float) * (size_t)ntiles), o_trhi = carve(sizeof(float) * (size_t)ntiles);
    const size_t o_Slo = carve(sizeof(float) * (size_t)ntiles * Kc), o_Shi = carve(sizeof(float) * (size_t)ntiles * Kc);
    const size_t o_Stlo = carve(sizeof(float) * (size_t)ntiles * Kc), o_Sthi = carve(sizeof(float) * (size_t)ntiles * Kc);
    const size_t o_blist = carve(sizeof(int32_t) * (size_t)bound_blocks * nsamp), o_bcount = carve(sizeof(int32_t) * (size_t)bound_blocks);
    const size_t o_elist = carve(sizeof(int32_t) * (size_t)emit_blocks * ecap);
    const size_t o_ecount = carve(sizeof(int32_t) * (size_t)emit_blocks);
    const size_t o_ovq = carve(sizeof(int32_t) * (size_t)Mp), o_ovb = carve(sizeof(double) * (size_t)Mp);
    DDX_TRY(ensure(ctx, ctx->knn_cells, cw));
    char* cb = ctx->knn_cells.as<char>();
    int32_t* label = reinterpret_cast<int32_t*>(cb + o_label);
    uint32_t* kin = reinterpret_cast<uint32_t*>(cb + o_kin);
    uint32_t* cellpos = reinterpret_cast<uint32_t*>(cb + o_cellpos);
    int32_t* perm1 = reinterpret_cast<int32_t*>(cb + o_perm1);
    float* cen = reinterpret_cast<float*>(cb + o_cen);
    float* cenR = reinterpret_cast<float*>(cb + o_cenR);
    float* cn = reinterpret_cast<float*>(cb + o_cn);
    unsigned long long* sums = reinterpret_cast<unsigned long long*>(cb + o_sums);
    int32_t* counts = reinterpret_cast<int32_t*>(cb + o_counts);       // [Kc] + largest squared norm (float bits) at [Kc + 1]
    unsigned* r2max = reinterpret_cast<unsigned*>(counts + Kc + 1);
    int32_t* rank = reinterpret_cast<int32_t*>(cb + o_rank);
    float* invD = reinterpret_cast<float*>(cb + o_invD);
    int32_t* ctile = reinterpret_cast<int32_t*>(cb + o_ctile);
    int32_t* ncount = reinterpret_cast<int32_t*>(cb + o_ncount);
    int32_t* nlist = reinterpret_cast<int32_t*>(cb + o_nlist);
    int32_t* tilecell = reinterpret_cast<int32_t*>(cb + o_tilecell);
    float* t1lo = reinterpret_cast<float*>(cb + o_t1lo);
    float* t1hi = reinterpret_cast<float*>(cb + o_t1hi);
    float* tr_lo = reinterpret_cast<float*>(cb + o_trlo);
    float* tr_hi = reinterpret_cast<float*>(cb + o_trhi);
    float* S_lo = reinterpret_cast<float*>(cb + o_Slo);
    float* S_hi = reinterpret_cast<float*>(cb + o_Shi);
    float* St_lo = reinterpret_cast<float*>(cb + o_Stlo);
    float* St_hi = reinterpret_cast<float*>(cb + o_Sthi);
    int32_t* blist = reinterpret_cast<int32_t*>(cb + o_blist);
    int32_t* bcount = reinterpret_cast<int32_t*>(cb + o_bcount);
    int32_t* elist = reinterpret_cast<int32_t*>(cb + o_elist);
    int32_t* ecount = reinterpret_cast<int32_t*>(cb + o_ecount);
    int32_t* ovf_q = reinterpret_cast<int32_t*>(cb + o_ovq);
    double* ovf_bound = reinterpret_cast<double*>(cb + o_ovb);
    const float* emb = ctx->emb32.as<float>();
    // Order of the points: cell by cell, first principal component inside a cell (stable radix sorts: ties by id).  Every
    // pass below works in that order.
    {
        ScopedTimer t(ctx, "knn_prepare");
        k_knn_keys<<<(unsigned)ceil_div(M, 256), 256, 0, ctx->stream>>>(emb, M, C, keys_in, ids_in);
        size_t tmp_bytes = 0;
        DDX_HIP(ctx, prim::sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, ids_in, Kc > 1 ? perm1 : perm, (int)M, 0, 32, ctx->stream));
        DDX_TRY(ensure(ctx, ctx->sort_tmp, tmp_bytes));
        DDX_HIP(ctx, prim::sort_pairs(ctx->sort_tmp.p, tmp_bytes, keys_in, keys_out, ids_in, Kc > 1 ? perm1 : perm, (int)M, 0, 32, ctx->stream));
        DDX_HIP(ctx, hipMemsetAsync(sums, 0, sizeof(unsigned long long) * (size_t)Kc * kCellDim, ctx->stream));
        DDX_HIP(ctx, hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)Kc + 16, ctx->stream));
    }
    if (Kc > 1) {
        ScopedTimer t(ctx, "knn_cells");
        const unsigned gK = (unsigned)ceil_div(Kc, 256), gM = (unsigned)ceil_div(M, 256), gA = (unsigned)ceil_div(M, 64);
        k_cells_init<<<gK, 256, 0, ctx->stream>>>(emb, perm1, M, C, Kc, cen, cn);
        const int sub = M >= (int64_t)Kc * 64 * kCellSub ? kCellSub : 1;
        const int64_t ns = M / sub;
        for (int round = 0; round < kCellRounds; ++round) {
            k_cells_assign<<<(unsigned)ceil_div(ns, 64), 256, 0, ctx->stream>>>(emb, perm1, ns, sub, C, cen, cn, Kc, label, nullptr);
            k_cells_accumulate<<<(unsigned)ceil_div(ns * kCellDim, 256), 256, 0, ctx->stream>>>(emb, perm1, ns, sub, C, label, sums, counts);
            k_cells_mean<<<gK, 256, 0, ctx->stream>>>(sums, counts, Kc, cen, cn);
        }
        // order inside a cell: by distance from its centre (the few points far from every centre -- whose intervals along any
        // direction are wide -- end up together in the last tiles of their cell instead of widening many tiles)
        k_cells_assign<<<gA, 256, 0, ctx->stream>>>(emb, ids_in, M, 1, C, cen, cn, Kc, label, keys_in);
        size_t tb1 = 0;
        DDX_HIP(ctx, prim::sort_pairs(nullptr, tb1, keys_in, keys_out, ids_in, perm1, (int)M, 0, 32, ctx->stream));
        DDX_TRY(ensure(ctx, ctx->sort_tmp, tb1));
        DDX_HIP(ctx, prim::sort_pairs(ctx->sort_tmp.p, tb1, keys_in, keys_out, ids_in, perm1, (int)M, 0, 32, ctx->stream));
        k_cells_rank<<<1, 1024, 0, ctx->stream>>>(cen, Kc, rank, cenR);
        k_cells_keys<<<gM, 256, 0, ctx->stream>>>(perm1, label, rank, M, kin);
        int bits = 1;
        while ((1 << bits) < Kc) ++bits;
        size_t tb2 = 0;
        DDX_HIP(ctx, prim::sort_pairs(nullptr, tb2, kin, cellpos, perm1, perm, (int)M, 0, bits, ctx->stream));
        DDX_TRY(ensure(ctx, ctx->sort_tmp, tb2));
        DDX_HIP(ctx, prim::sort_pairs(ctx->sort_tmp.p, tb2, kin, cellpos, perm1, perm, (int)M, 0, bits, ctx->stream));
        k_cells_invdist<<<(unsigned)ceil_div((int64_t)Kc * Kc, 256), 256, 0, ctx->stream>>>(cenR, Kc, invD);
    } else {
        DDX_HIP(ctx, hipMemsetAsync(cellpos, 0, sizeof(uint32_t) * (size_t)M, ctx->stream));
        DDX_HIP(ctx, hipMemsetAsync(cenR, 0, sizeof(float) * kCellDim, ctx->stream));
        DDX_HIP(ctx, hipMemsetAsync(invD, 0, sizeof(float), ctx->stream));
    }
    {
        ScopedTimer t(ctx, "knn_tables");
        k_knn_prepare<<<(unsigned)ceil_div(Mp, 256), 256, 0, ctx->stream>>>(emb, perm, M, Mp, C, CP, E, Eb, nrm, p1, start4);
        k_knn_tileinfo<<<(unsigned)ceil_div(ntiles, 256), 256, 0, ctx->stream>>>(cellpos, p1, nrm, M, ntiles, tilecell, t1lo, t1hi, r2max);
        k_cells_tilestart<<<(unsigned)ceil_div(Kc + 1, 256), 256, 0, ctx->stream>>>(tilecell, ntr, Kc, ctile);
        if (Kc > 1) k_cells_neighbours<<<(unsigned)Kc, 1024, 0, ctx->stream>>>(cenR, ctile, Kc, (int)nsamp, nlist, ncount);
        const size_t bl_lds = sizeof(unsigned) * (size_t)((ntr + 31) / 32 + 1);        // one bit per tile: 128 KB at the 2^24-point limit
        if (bl_lds > 64 * 1024) DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_knn_boundlists), (int)bl_lds));
        k_knn_boundlists<<<(unsigned)bound_blocks, 64, bl_lds, ctx->stream>>>(tilecell, ctile, nlist, ncount, Kc, (int)nsamp, (int)nsamp, ntr, ctx->opt.knn_sample_every, blist, bcount);
        k_knn_slabs<<<(unsigned)(Mp / 256), 256, 0, ctx->stream>>>(E, CP, nrm, tilecell, cenR, invD, Kc, ntiles, r2max, S_lo, S_hi, St_lo, St_hi, tr_lo, tr_hi);
    }
    DDX_HIP(ctx, hipMemsetAsync(ccount, 0, sizeof(int32_t) * (Mp + 64), ctx->stream));
    {
        ScopedTimer t(ctx, "knn_bound");
        const unsigned grid = (unsigned)(xcd_chunk > 0 ? ceil_div(bound_blocks, 8 * (int64_t)xcd_chunk) * 8 * xcd_chunk : bound_blocks);
        for (int g = 0; g < groups; ++g) {
#define DDX_BOUND(CPV, KEEP) k_knn_bound_bf<CPV, KEEP><<<grid, 64 * kBoundWaves, 0, ctx->stream>>>(Eb, nrm, Mp, k_bound, include_self, blist, bcount, nsamp, groups, g, g > 0, thr, xcd_chunk)
            if (CP == 32 && keep_small) DDX_BOUND(32, kBoundKeepSmall);
            else if (CP == 32) DDX_BOUND(32, kBoundKeepLarge);
            else if (CP == 64 && keep_small) DDX_BOUND(64, kBoundKeepSmall);
            else if (CP == 64) DDX_BOUND(64, kBoundKeepLarge);
            else if (keep_small) DDX_BOUND(128, kBoundKeepSmall);
            else DDX_BOUND(128, kBoundKeepLarge);
#undef DDX_BOUND
        }
    }
    unsigned long long* wtotal = reinterpret_cast<unsigned long long*>(ccount + Mp + 2);
    {
        ScopedTimer t(ctx, "knn_lists");
        const unsigned grid = (unsigned)emit_blocks;
#define DDX_LISTS(BWV) k_knn_tilelists<BWV><<<grid, 64 * BWV, 0, ctx->stream>>>(S_lo, S_hi, St_lo, St_hi, tilecell, t1lo, t1hi, thr, nrm, ntiles, Kc, G, elist, ecount, ecap, wtotal)
        if (BW == 8) DDX_LISTS(8);
        else DDX_LISTS(4);
#undef DDX_LISTS
    }
    {
        ScopedTimer t(ctx, "knn_emit");
        const int seg_steps = ctx->opt.knn_seg_steps > 0 ? ctx->opt.knn_seg_steps : (M >= 400000 ? 2 * kEmitSegSteps : kEmitSegSteps);   // (measured: 1.33 ms at 125 k points with 4, 18.6 ms at 625 k with 8)
        const int nseg = (int)(ceil_div(ceil_div(ecap, G), (int64_t)seg_steps * 8) * 8);     // segments of the longest possible list, a multiple of 8
        const unsigned grid_x = (unsigned)(emit_blocks * nseg);
        const int dbg_mode = ctx->opt.knn_ablation;     // timing ablations (wrong results): non-zero only in -DDDX_ABLATION builds
        long long* tdbg = nullptr;                       // knn_debug: start / end / steps of every work item
        if (ctx->opt.knn_debug) { DDX_HIP(ctx, hipMalloc(&tdbg, sizeof(long long) * 4 * (size_t)grid_x)); DDX_HIP(ctx, hipMemsetAsync(tdbg, 0, sizeof(long long) * 4 * (size_t)grid_x, ctx->stream)); }
#define DDX_EMIT_ONE(CPV, FOLDV, QUERY, BWV) k_knn_emit_bf<CPV, FOLDV, BWV><<<grid_x, 64 * BWV, 0, ctx->stream>>>(Eb, QUERY, nrm, start4, thr, Mp, include_self, ccount, cbuf, elist, ecount, ecap, dbg_mode, cap, nseg, seg_steps, tdbg)
#define DDX_EMIT_BF(CPV, FOLDV, QUERY)                          \
    do {                                                        \
        if (BW == 8) DDX_EMIT_ONE(CPV, FOLDV, QUERY, 8);        \
        else DDX_EMIT_ONE(CPV, FOLDV, QUERY, 4);                \
    } while (0)
        // option knn_emit_rt=4: 64 queries per wave, two waves per block, on the lists built for four waves of 32
#define DDX_EMIT_WIDE(FOLDV, QUERY) k_knn_emit_bf<32, FOLDV, 2, 2 * kEmitRT><<<grid_x, 128, 0, ctx->stream>>>(Eb, QUERY, nrm, start4, thr, Mp, include_self, ccount, cbuf, elist, ecount, ecap, dbg_mode, cap, nseg, seg_steps, tdbg)
        const bool wide_waves = ctx->opt.knn_emit_rt == 4 && BW == 4 && CP == 32;
        if (fold) {
            k_knn_fold<<<(unsigned)ceil_div(Mp * 8, 256), 256, 0, ctx->stream>>>(nrm, thr, Mp, Eb, Ebq);
            if (wide_waves) DDX_EMIT_WIDE(true, Ebq);
            else DDX_EMIT_BF(32, true, Ebq);
        } else if (CP == 32 && wide_waves) DDX_EMIT_WIDE(false, Eb);
        else if (CP == 32) DDX_EMIT_BF(32, false, Eb);
        else if (CP == 64) DDX_EMIT_BF(64, false, Eb);
        else DDX_EMIT_BF(128, false, Eb);
#undef DDX_EMIT_WIDE
#undef DDX_EMIT_BF
#undef DDX_EMIT_ONE
        if (tdbg) {
            std::vector<long long> ht(4 * (size_t)grid_x);
            DDX_HIP(ctx, wait_stream(ctx));
            DDX_HIP(ctx, hipMemcpy(ht.data(), tdbg, sizeof(long long) * ht.size(), hipMemcpyDeviceToHost));
            DDX_HIP(ctx, hipFree(tdbg));
            long long t0 = 0, t1 = 0;
            std::vector<std::pair<long long, int>> ev;
            std::vector<double> dur;
            double steps = 0;
            for (size_t i = 0; i < grid_x; ++i) {
                if (!ht[4 * i]) continue;
                if (!t0 || ht[4 * i] < t0) t0 = ht[4 * i];
                t1 = std::max(t1, ht[4 * i + 1]);
                ev.emplace_back(ht[4 * i], 1);
                ev.emplace_back(ht[4 * i + 1], -1);
                dur.push_back((ht[4 * i + 1] - ht[4 * i]) * 0.01);
                steps += (double)ht[4 * i + 2];
            }
            std::sort(ev.begin(), ev.end());
            std::sort(dur.begin(), dur.end());
            // resident work items over time: average, and per tenth of the span
            const double span = (t1 - t0) * 0.01;
            std::vector<double> tenth(10, 0.0);
            int cur = 0;
            double area = 0;
            for (size_t i = 0; i + 1 < ev.size(); ++i) {
                cur += ev[i].second;
                const double a = (ev[i + 1].first - ev[i].first) * 0.01 * cur;
                area += a;
                tenth[std::min<size_t>(9, (size_t)((ev[i].first - t0) * 0.01 / span * 10))] += a;
            }
            if (!dur.empty()) {
                fprintf(stderr, "[knn emit] %zu work items with work (of %u launched), %.0f steps; span %.1f us; item duration us: median %.2f, 90%% %.2f, 99%% %.2f, max %.2f; resident items: average %.1f;"
                                " per tenth of the span:", dur.size(), grid_x, steps, span, dur[dur.size() / 2], dur[dur.size() * 9 / 10], dur[dur.size() * 99 / 100], dur.back(), area / span);
                for (int i = 0; i < 10; ++i) fprintf(stderr, " %.0f", tenth[i] / (span / 10));
                fprintf(stderr, "\n");
            }
        }
    }
    {
        ScopedTimer t(ctx, "knn_select");
        const unsigned g2 = (unsigned)ceil_div(M, 4);
        int32_t* ki = ctx->knn_idx.as<int32_t>();
        double* kd = ctx->knn_dist.as<double>();
#define DDX_SELECT_LAUNCH(CPV)                                                                                                                          \
    do {                                                                                                                                                \
        k_knn_select<CPV, kSelSmall, 4><<<g2, 256, 0, ctx->stream>>>(E, perm, M, k, include_self, ccount, cbuf, ki, kd, ccount + Mp, ovf_q, ovf_bound, cap, 0, kSelSmall);    \
        k_knn_select<CPV, kSelMax, 4><<<g2, 256, 0, ctx->stream>>>(E, perm, M, k, include_self, ccount, cbuf, ki, kd, ccount + Mp, ovf_q, ovf_bound, cap, kSelSmall, kSelMax); \
        if (cap > kSelMax)                                                                                                                              \
            k_knn_select<CPV, kSelHuge, 1><<<(unsigned)M, 64, 0, ctx->stream>>>(E, perm, M, k, include_self, ccount, cbuf, ki, kd, ccount + Mp, ovf_q, ovf_bound, cap, kSelMax, kSelHuge); \
    } while (0)
        const size_t rs_lds = (size_t)(kRescanWaves * kRescanWin + kRescanWaves * 256) * (sizeof(double) + sizeof(int32_t)) + sizeof(float) * CP;
#define DDX_RESCAN(CPV)                                                                                                                       \
    do {                                                                                                                                      \
        DDX_TRY(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&k_knn_rescan<CPV>), (int)rs_lds));                                      \
        k_knn_rescan<CPV><<<128, 64 * kRescanWaves, rs_lds, ctx->stream>>>(E, perm, M, k, include_self, ccount + Mp, ovf_q, ovf_bound, elist, ecount, ecap, BW, tilecell, cenR, invD, Kc, ntiles, r2max, St_lo, St_hi, tr_lo, tr_hi, ki, kd, rdbg); \
    } while (0)
        long long* rdbg = nullptr;
        if (ctx->opt.knn_debug) { DDX_HIP(ctx, hipMalloc(&rdbg, sizeof(long long) * 8 * 4096)); DDX_HIP(ctx, hipMemsetAsync(rdbg, 0, sizeof(long long) * 8 * 4096, ctx->stream)); }
        if (CP == 32) { DDX_SELECT_LAUNCH(32); DDX_RESCAN(32); }
        else if (CP == 64) { DDX_SELECT_LAUNCH(64); DDX_RESCAN(64); }
        else { DDX_SELECT_LAUNCH(128); DDX_RESCAN(128); }
#undef DDX_RESCAN
#undef DDX_SELECT_LAUNCH
        if (rdbg) {
            std::vector<long long> hd(8 * 4096);
            DDX_HIP(ctx, wait_stream(ctx));
            DDX_HIP(ctx, hipMemcpy(hd.data(), rdbg, sizeof(long long) * hd.size(), hipMemcpyDeviceToHost));
            DDX_HIP(ctx, hipFree(rdbg));
            long long t0 = 0;
            for (int i = 0; i < 4096; ++i) if (hd[i * 8] && (!t0 || hd[i * 8] < t0)) t0 = hd[i * 8];
            for (int i = 0; i < 4096 && hd[i * 8]; ++i)
                fprintf(stderr, "[knn rescan] query %d block %lld: start %.1f us, scan %.1f, sorts %.1f, merge %.1f; list %lld entries, %lld tiles scanned\n", i, hd[i * 8 + 6],
                        (hd[i * 8] - t0) * 0.01, (hd[i * 8 + 1] - hd[i * 8]) * 0.01, (hd[i * 8 + 2] - hd[i * 8 + 1]) * 0.01, (hd[i * 8 + 3] - hd[i * 8 + 2]) * 0.01, hd[i * 8 + 4], hd[i * 8 + 5]);
        }
    }
    DDX_HIP(ctx, hipGetLastError());
    if (ctx->opt.knn_debug) {
        std::vector<int32_t> h(Mp + 1);
        DDX_HIP(ctx, hipMemcpyAsync(h.data(), ccount, sizeof(int32_t) * (Mp + 1), hipMemcpyDeviceToHost, ctx->stream));
        DDX_HIP(ctx, wait_stream(ctx));
        double sum = 0; int mx = 0; int64_t over = 0, longer = 0;
        for (int64_t i = 0; i < M; ++i) { sum += h[i]; if (h[i] > mx) mx = h[i]; over += h[i] > cap; longer += h[i] > kSelSmall; }
        std::vector<int32_t> hl(emit_blocks);
        DDX_HIP(ctx, hipMemcpy(hl.data(), ecount, sizeof(int32_t) * emit_blocks, hipMemcpyDeviceToHost));
        unsigned long long scr = 0;
        DDX_HIP(ctx, hipMemcpy(&scr, wtotal, sizeof(scr), hipMemcpyDeviceToHost));
        double lsum = 0;
        for (int64_t b = 0; b < emit_blocks; ++b) lsum += hl[b];
        int smin = (int)M, smax = (int)M;
        if (Kc > 1) {
            std::vector<uint32_t> hc(M);
            DDX_HIP(ctx, hipMemcpy(hc.data(), cellpos, sizeof(uint32_t) * M, hipMemcpyDeviceToHost));
            std::vector<int> hs(Kc, 0);
            for (int64_t i = 0; i < M; ++i) ++hs[hc[i]];
            smin = smax = hs[0];
            for (int c = 1; c < Kc; ++c) { smin = std::min(smin, hs[c]); smax = std::max(smax, hs[c]); }
        }
        fprintf(stderr, "[knn] k=%d cells=%d (sizes %d..%d) sample tiles=%lld: candidates/query mean %.1f max %d, %lld beyond 256, overflowed %lld (counter %d); "
                        "emit block stages %.1f%% of the tiles, a wave screens %.1f%%\n",
                k, Kc, smin, smax, (long long)nsamp, sum / M, mx, (long long)longer, (long long)over, h[Mp], 100.0 * lsum / ((double)emit_blocks * (double)ntiles),
                100.0 * (double)scr / ((double)(Mp / 32) * (double)ntiles));
    }
    if (ctx->opt.knn_debug) {
        // how the work is spread: tiles screened per wave, and how far single queries' bounds stand out in their tile
        std::vector<int32_t> he(emit_blocks);
        std::vector<int32_t> hm((size_t)emit_blocks * ecap);
        DDX_HIP(ctx, hipMemcpy(he.data(), ecount, sizeof(int32_t) * emit_blocks, hipMemcpyDeviceToHost));
        DDX_HIP(ctx, hipMemcpy(hm.data(), elist, sizeof(int32_t) * hm.size(), hipMemcpyDeviceToHost));
        std::vector<int> per_wave;
        for (int64_t b = 0; b < emit_blocks; ++b) {
            std::vector<int> c(BW, 0);
            for (int i = 0; i < he[b]; ++i) for (int w = 0; w < BW; ++w) c[w] += (((unsigned)hm[b * ecap + i] >> 24) >> w) & 1u;
            for (int w = 0; w < BW; ++w) per_wave.push_back(c[w]);
        }
        std::sort(per_wave.begin(), per_wave.end());
        auto q = [&](double f) { return 100.0 * per_wave[(size_t)(f * (per_wave.size() - 1))] / (double)ntiles; };
        std::vector<float> ht(Mp);
        DDX_HIP(ctx, hipMemcpy(ht.data(), thr, sizeof(float) * Mp, hipMemcpyDeviceToHost));
        int64_t r15 = 0, r2 = 0, r3 = 0;
        for (int64_t t = 0; t < M / 16; ++t) {
            float v[16];
            for (int j = 0; j < 16; ++j) v[j] = ht[t * 16 + j];
            std::sort(v, v + 16);
            const float med = v[7];
            for (int j = 0; j < 16; ++j) { r15 += v[j] > 1.5f * med; r2 += v[j] > 2.f * med; r3 += v[j] > 3.f * med; }
        }
        fprintf(stderr, "[knn] tiles screened per wave (%% of all): median %.1f, 90%% %.1f, 99%% %.1f, 99.9%% %.1f, max %.1f; bounds above 1.5 / 2 / 3 x their tile's median: %lld / %lld / %lld\n",
                q(0.5), q(0.9), q(0.99), q(0.999), q(1.0), (long long)r15, (long long)r2, (long long)r3);
    }
    ctx->knn_window_total = wtotal;
    ctx->knn_window_pairs = (double)(Mp / (16 * kEmitRT)) * (double)ntiles;     // (wave of 32 queries, candidate tile) pairs
    ctx->knn_overflow = ccount + Mp;
    ctx->knn_ccount = ccount;
    ctx->knn_perm = perm;
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

// The same weights, one wave per node (K <= 64): N(i) sits sorted in LDS, lane l holds the l-th relation of i and,
// relation by relation, the wave loads N(j) (one coalesced row), every lane looks its element up in N(i) by
// binary search and a ballot counts the shared neighbours.  HALF (K <= 32, the shipped settings): the two halves of the
// wave take two relations at a time -- lanes 0-31 the even one, lanes 32-63 the odd one -- so all 64 lanes search.
// Four steps (four or eight rows) are in flight at a time.
template <bool HALF>
__global__ void __launch_bounds__(256) k_edge_weights_wave(const int32_t* __restrict__ idx, const int32_t* __restrict__ sorted,
                                                           int64_t M, int K, int mode, double* __restrict__ w_out) {
    __shared__ int32_t nS[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= M) return;
    int32_t* Ni = nS[wave];
    Ni[lane] = lane < K ? sorted[i * K + lane] : 0x7fffffff;
    const int32_t myj = lane < K ? idx[i * K + lane] : -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int PER = HALF ? 2 : 1;                // relations per step
    const int half = HALF ? (lane >> 5) : 0;
    const int sub = HALF ? (lane & 31) : lane;       // element of N(j) this lane looks up
    double myw = 0.0;
    auto weight = [&](int shared, bool mutual) -> double {
        double w;
        if (mode == 2) {
            w = 1.0;
        } else {
            const double J = (double)shared / (2.0 * (double)K - (double)shared);
            if (mode == 0) w = mutual ? J * J : 0.0;
            else w = mutual ? (J + J) / 2.0 : J / 2.0;
        }
        return mutual ? w : -w;
    };
    // HALF: the weight is a function of the shared count (<= K <= 32) and the mutual flag only: lane l evaluates both for l shared
    // neighbours once, a relation then reads its value from lane `shared` (the two float64 divisions per relation were a third of the
    // kernel's instructions; the values are the same expression's, bit for bit)
    const double tabM = HALF ? weight(lane, true) : 0.0, tabN = HALF ? weight(lane, false) : 0.0;
    auto table = [&](int shared, bool mutual) -> double {           // (both arguments wave-uniform)
        const double t = mutual ? tabM : tabN;
        const long long bits = __builtin_bit_cast(long long, t);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, shared);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), shared);
        return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
    };
    for (int t0 = 0; t0 < K; t0 += 4 * PER) {
        int32_t j[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rel = t0 + PER * u + half;
            j[u] = rel < K ? __shfl(myj, rel, 64) : -1;
            const bool ok = j[u] >= 0 && j[u] != (int32_t)i;
            y[u] = (ok && sub < K) ? sorted[(int64_t)j[u] * K + sub] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (t0 + PER * u >= K) break;
            const bool ok = j[u] >= 0 && j[u] != (int32_t)i;
            const bool valid = ok && sub < K;
            bool found = false;
            if (valid) {
                int lo = 0, hi = K;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (Ni[mid] < y[u]) lo = mid + 1; else hi = mid;
                }
                found = lo < K && Ni[lo] == y[u];
            }
            const unsigned long long fm = __ballot(found);
            const unsigned long long mm = __ballot(valid && y[u] == (int32_t)i);
            if (HALF) {
                // (the relation's validity is known to its own lane: an invalid relation keeps weight 0)
                const double w0 = table(__popc((unsigned)fm), (unsigned)mm != 0u);
                const double w1 = table(__popc((unsigned)(fm >> 32)), (unsigned)(mm >> 32) != 0u);
                const bool mine_ok = myj >= 0 && myj != (int32_t)i;
                if (lane == t0 + 2 * u) myw = mine_ok ? w0 : 0.0;
                if (lane == t0 + 2 * u + 1) myw = mine_ok ? w1 : 0.0;
            } else {
                const double w = ok ? weight(__popcll(fm), mm != 0ull) : 0.0;
                if (lane == t0 + u) myw = w;
            }
        }
    }
    if (lane < K) w_out[i * K + lane] = myw;
}

// ---- umap connectivities (mode 3: what sc.tl.leiden clusters on) --------------------------------------------
// oracle/dd_oracle.py:umap_connectivities states the computation (umap-learn's fuzzy_simplicial_set with
// local_connectivity = 1, set_op_mix_ratio = 1): distances rounded to float32, everything else float64.
// One thread per point: rho = smallest positive distance, sigma from 64 bisection steps, then the directed weights.
__global__ void __launch_bounds__(256) k_umap_directed(const int32_t* __restrict__ idx, const double* __restrict__ d2, int64_t M, int K,
                                                       double target, double* __restrict__ val) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double* row = d2 + i * K;
    double rho = __builtin_huge_val(), mean = 0.0;
    for (int j = 0; j < K; ++j) {
        const double d = (double)(float)sqrt(row[j]);
        if (d > 0.0 && d < rho) rho = d;
        mean = mean + d;
    }
    if (!(rho < __builtin_huge_val())) rho = 0.0;
    mean = mean / (double)K;
    double lo = 0.0, hi = __builtin_huge_val(), mid = 1.0;
    for (int it = 0; it < 64; ++it) {
        double psum = 0.0;
        for (int j = 1; j < K; ++j) {
            const double g = (double)(float)sqrt(row[j]) - rho;
            psum = psum + (g > 0.0 ? exp(-(g / mid)) : 1.0);
        }
        if (psum > target) {
            hi = mid;
            mid = (lo + hi) / 2.0;
        } else {
            lo = mid;
            mid = (hi < __builtin_huge_val()) ? (lo + hi) / 2.0 : mid * 2.0;
        }
    }
    const double floor_v = 1e-3 * mean;
    const double sigma = mid > floor_v ? mid : floor_v;
    for (int j = 0; j < K; ++j) {
        const int32_t c = idx[i * K + j];
        const double g = (double)(float)sqrt(row[j]) - rho;
        double v = (g <= 0.0 || sigma == 0.0) ? 1.0 : exp(-(g / sigma));
        if (c == (int32_t)i || c < 0) v = 0.0;
        val[i * K + j] = v;
    }
}

// fuzzy union a + b - a*b with the reverse relation (b = 0 when j does not list i); a one-directional relation is
// flagged by a negative weight, as for the other graph types (the assembly adds its reverse entry)
__global__ void __launch_bounds__(256) k_umap_union(const int32_t* __restrict__ idx, const double* __restrict__ val, int64_t M, int K,
                                                    double* __restrict__ w_out) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * K) return;
    const int64_t i = t / K;
    const int32_t j = idx[t];
    const double a = val[t];
    if (j < 0 || j == (int32_t)i || a == 0.0) { w_out[t] = 0.0; return; }
    double b = 0.0;
    bool mutual = false;
    for (int c = 0; c < K; ++c) {
        if (idx[(int64_t)j * K + c] == (int32_t)i) {
            const double bv = val[(int64_t)j * K + c];
            if (bv != 0.0) { b = bv; mutual = true; }
            break;
        }
    }
    w_out[t] = mutual ? (a + b) - a * b : -a;
}

// ---- symmetric CSR on the device ----------------------------------------------------------------------
// every relation with a non-zero weight contributes the pair (i,j); a one-directional relation (negative
// flag) also contributes (j,i).  Pairs are keyed (row << bits(M) | column) and radix-sorted, which yields rows
// in order and columns ascending inside a row (the adjacency order the community-detection spec visits).
__global__ void __launch_bounds__(256) k_pair_count(const double* __restrict__ w, int64_t n, int32_t* __restrict__ cnt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double v = w[t];
    cnt[t] = (v == 0.0) ? 0 : (v < 0.0 ? 2 : 1);
}

__global__ void __launch_bounds__(256) k_pair_emit(const int32_t* __restrict__ idx, const double* __restrict__ w, int64_t n,
                                                   int K, int shift, const int64_t* __restrict__ offs, uint64_t* __restrict__ keys,
                                                   double* __restrict__ vals) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double v = w[t];
    if (v == 0.0) return;
    const uint64_t i = (uint64_t)(t / K), j = (uint64_t)idx[t];
    const double av = v < 0.0 ? -v : v;
    int64_t o = offs[t];
    keys[o] = (i << shift) | j;            // shift = bits of the node count: the sort only passes over 2*shift bits
    vals[o] = av;
    if (v < 0.0) {
        keys[o + 1] = (j << shift) | i;
        vals[o + 1] = av;
    }
}

__global__ void k_rowptr_from_keys(const uint64_t* __restrict__ keys, int64_t n, int64_t M, int shift, int64_t* __restrict__ indptr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > M) return;
    int64_t lo = 0, hi = n;   // first key with row >= r
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(keys[mid] >> shift) < r) lo = mid + 1; else hi = mid;
    }
    indptr[r] = lo;
}

__global__ void k_cols_from_keys(const uint64_t* __restrict__ keys, int64_t n, int shift, int32_t* __restrict__ cols) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) cols[t] = (int32_t)(keys[t] & ((1ull << shift) - 1ull));
}

// device part: relation weights for the kNN table currently held by the context
static int graph_weights_device(ddx_ctx* ctx, int32_t mode);

int stage_graph_relations(ddx_ctx* ctx, int32_t mode, int32_t* idx_host, double* w_host) {
    const int64_t M = ctx->embM;
    const int K = ctx->K;
    DDX_TRY(graph_weights_device(ctx, mode));
    DDX_HIP(ctx, hipMemcpyAsync(idx_host, ctx->knn_idx.p, sizeof(int32_t) * (size_t)M * K, hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, hipMemcpyAsync(w_host, ctx->edge_w.p, sizeof(double) * (size_t)M * K, hipMemcpyDeviceToHost, ctx->stream));
    DDX_HIP(ctx, wait_stream(ctx));
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

// relation weights on the device (no copies)
static int graph_weights_device(ddx_ctx* ctx, int32_t mode) {
    const int64_t M = ctx->embM;
    const int K = ctx->K;
    DDX_TRY(ensure(ctx, ctx->knn_sorted, sizeof(int32_t) * (size_t)M * K));
    DDX_TRY(ensure(ctx, ctx->edge_w, sizeof(double) * (size_t)M * K));
    ScopedTimer t(ctx, "graph_weights");
    if (mode == 3) {
        // directed weights go to knn_sorted's neighbour (reused as float64 scratch: M*K doubles live in pcaOp)
        DDX_TRY(ensure(ctx, ctx->pcaOp, sizeof(double) * (size_t)M * K));
        double* val = ctx->pcaOp.as<double>();
        k_umap_directed<<<(unsigned)ceil_div(M, 256), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->knn_dist.as<double>(), M, K, std::log2((double)K), val);
        k_umap_union<<<(unsigned)ceil_div(M * K, 256), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), val, M, K, ctx->edge_w.as<double>());
        return DDX_OK;
    }
    k_sort_neighbours<<<(unsigned)ceil_div(M, 64), 64, sizeof(int32_t) * K * 64, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), M, K, ctx->knn_sorted.as<int32_t>());
    if (K <= 32)
        k_edge_weights_wave<true><<<(unsigned)ceil_div(M, 4), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->knn_sorted.as<int32_t>(), M, K, mode,
                                                                                     ctx->edge_w.as<double>());
    else if (K <= 64)
        k_edge_weights_wave<false><<<(unsigned)ceil_div(M, 4), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->knn_sorted.as<int32_t>(), M, K, mode,
                                                                                      ctx->edge_w.as<double>());
    else
        k_edge_weights<<<(unsigned)ceil_div(M * K, 256), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->knn_sorted.as<int32_t>(), M, K, mode,
                                                                                ctx->edge_w.as<double>());
    return DDX_OK;
}

// whole graph stage on the device; the symmetric CSR lands in the context's host vectors
int stage_build_graph(ddx_ctx* ctx, int32_t mode) {
    const int64_t M = ctx->embM;
    const int K = ctx->K;
    const int64_t n = M * K;
    DDX_TRY(graph_weights_device(ctx, mode));
    // workspace of its own (not the PCA panel buffer: the graph of iteration i must outlive the PCA of iteration i + 1, whose
    // operator products run while the host finishes part B of iteration i -- part C then needs the graph again): offs i64[n+1] | keys u64[2n] x2 | vals f64[2n] x2 | indptr i64[M+1] | cols i32[2n] | cnt i32[n+1]
    // (sized by the same arithmetic that carves it: every piece is rounded up to 256 bytes)
    size_t bytes = 0;
    auto piece = [&](size_t sz) { const size_t o = bytes; bytes += (sz + 255) & ~(size_t)255; return o; };
    const size_t o_offs = piece(sizeof(int64_t) * (n + 1)), o_ka = piece(sizeof(uint64_t) * 2 * n), o_kb = piece(sizeof(uint64_t) * 2 * n);
    const size_t o_va = piece(sizeof(double) * 2 * n), o_vb = piece(sizeof(double) * 2 * n), o_ip = piece(sizeof(int64_t) * (M + 1));
    const size_t o_cols = piece(sizeof(int32_t) * 2 * n), o_cnt = piece(sizeof(int32_t) * (n + 1));
    ctx->g_nodes = -1;
    ctx->c_nodes = -1;
    DDX_TRY(ensure(ctx, ctx->graph_buf, bytes));
    unsigned char* base = ctx->graph_buf.as<unsigned char>();
    int64_t* offs = reinterpret_cast<int64_t*>(base + o_offs);
    uint64_t* keys_a = reinterpret_cast<uint64_t*>(base + o_ka);
    uint64_t* keys_b = reinterpret_cast<uint64_t*>(base + o_kb);
    double* vals_a = reinterpret_cast<double*>(base + o_va);
    double* vals_b = reinterpret_cast<double*>(base + o_vb);
    int64_t* d_indptr = reinterpret_cast<int64_t*>(base + o_ip);
    int32_t* d_cols = reinterpret_cast<int32_t*>(base + o_cols);
    int32_t* cnt = reinterpret_cast<int32_t*>(base + o_cnt);
    int64_t E = 0;
    {
        ScopedTimer t(ctx, "graph_assemble");
        k_pair_count<<<(unsigned)ceil_div(n, 256), 256, 0, ctx->stream>>>(ctx->edge_w.as<double>(), n, cnt);
        size_t tmp_bytes = 0;
        DDX_HIP(ctx, prim::exclusive_sum(nullptr, tmp_bytes, cnt, offs, (int)n + 1, ctx->stream));
        size_t tmp2 = 0;
        int shift = 1;                               // pairs are keyed row << shift | column with shift = bits(M)
        while (((int64_t)1 << shift) < M) ++shift;
        const int end_bit = 2 * shift;
        DDX_HIP(ctx, prim::sort_pairs(nullptr, tmp2, keys_a, keys_b, vals_a, vals_b, (int)(2 * n), 0, end_bit, ctx->stream));
        DDX_TRY(ensure(ctx, ctx->sort_tmp, std::max(tmp_bytes, tmp2)));
        // (k_pair_count fills cnt[0..n); the exclusive scan over n + 1 elements never adds cnt[n] to an output)
        DDX_HIP(ctx, prim::exclusive_sum(ctx->sort_tmp.p, tmp_bytes, cnt, offs, (int)n + 1, ctx->stream));
        DDX_HIP(ctx, hipMemcpyAsync(&E, offs + n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        k_pair_emit<<<(unsigned)ceil_div(n, 256), 256, 0, ctx->stream>>>(ctx->knn_idx.as<int32_t>(), ctx->edge_w.as<double>(), n, K, shift, offs, keys_a, vals_a);
        DDX_HIP(ctx, wait_stream(ctx));
        if (E > 0) {
            // the sort picks its algorithm (single block / merge / onesweep) by the element count, and each has its own
            // temporary-storage need: ask again for the actual count
            size_t tmp3 = 0;
            DDX_HIP(ctx, prim::sort_pairs(nullptr, tmp3, keys_a, keys_b, vals_a, vals_b, (int)E, 0, end_bit, ctx->stream));
            DDX_TRY(ensure(ctx, ctx->sort_tmp, tmp3));
            DDX_HIP(ctx, prim::sort_pairs(ctx->sort_tmp.p, tmp3, keys_a, keys_b, vals_a, vals_b, (int)E, 0, end_bit, ctx->stream));
            k_cols_from_keys<<<(unsigned)ceil_div(E, 256), 256, 0, ctx->stream>>>(keys_b, E, shift, d_cols);
        }
        k_rowptr_from_keys<<<(unsigned)ceil_div(M + 1, 256), 256, 0, ctx->stream>>>(keys_b, E, M, shift, d_indptr);
    }
    DDX_HIP(ctx, hipGetLastError());
    // the CSR stays on the device until ddx_get_graph copies it straight into the caller's buffers
    ctx->g_nodes = M;
    ctx->g_entries = E;
    ctx->g_d_indptr = d_indptr;
    ctx->g_d_cols = d_cols;
    ctx->g_d_vals = vals_b;
    return DDX_OK;
}

}  // namespace ddx
