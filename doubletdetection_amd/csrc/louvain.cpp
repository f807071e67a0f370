// Host C++ community detection for libddx: deterministic multi-level modularity optimisation
// (Blondel et al. 2008) with a resolution parameter.  It stands in for the native Louvain code the
// reference reaches through phenograph.cluster (dd.py:320-322) and sc.tl.louvain (dd.py:337-342).
// The specification -- visiting order, tie breaking, float64 operation order -- is the pure-Python
// text in oracle/louvain_ref.py; this file must reproduce it bit for bit (tests/test_host_native.py).
// Compiled with -ffp-contract=off so that no multiply-add is fused.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/ddx.h"

namespace {

struct SplitMix64 {
    uint64_t state;
    explicit SplitMix64(uint64_t seed) : state(seed) {}
    uint64_t next() {
        state += 0x9E3779B97F4A7C15ULL;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
};

struct Graph {
    std::vector<int64_t> indptr;
    std::vector<int32_t> indices;
    std::vector<double> weights;
    int64_t n() const { return (int64_t)indptr.size() - 1; }
};

constexpr double kMinGain = 1e-6;

// one level of local moving; returns true when any node moved
bool one_level(const Graph& g, double gamma, SplitMix64& rng, std::vector<int32_t>& comm, double* q_out) {
    const int64_t n = g.n();
    std::vector<double> deg(n, 0.0), loops(n, 0.0);
    for (int64_t v = 0; v < n; ++v) {
        double s = 0.0;
        for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
            s += g.weights[e];
            if (g.indices[e] == v) loops[v] += g.weights[e];
        }
        deg[v] = s;
    }
    double m2 = 0.0;
    for (int64_t v = 0; v < n; ++v) m2 += deg[v];
    comm.resize(n);
    for (int64_t v = 0; v < n; ++v) comm[v] = (int32_t)v;
    if (m2 == 0.0) {
        if (q_out) *q_out = 0.0;
        return false;
    }
    std::vector<double> tot(deg), in_(loops);
    // Visiting order: index order starting at a seeded offset (sequential memory access; a random permutation
    // costs 1.7x more time in cache misses for the same modularity).
    const int64_t start = (int64_t)(rng.next() % (uint64_t)n);
    std::vector<double> neigh_w(n, -1.0);
    std::vector<int32_t> seen;
    seen.reserve(256);
    bool improved = false;
    // Q is the sum over communities with tot > 0 in ascending id order (the specification's order).  That set only
    // shrinks during a level, so it is kept as a compacted ascending list: same terms, same order, O(live) per pass.
    std::vector<int32_t> live;
    live.reserve(n);
    for (int64_t c = 0; c < n; ++c)
        if (tot[c] > 0.0) live.push_back((int32_t)c);
    auto quality_live = [&]() {
        double q = 0.0;
        size_t w = 0;
        for (size_t i = 0; i < live.size(); ++i) {
            const int32_t c = live[i];
            if (tot[c] > 0.0) {
                const double b = tot[c] / m2;
                q += in_[c] / m2 - gamma * b * b;
                live[w++] = c;
            }
        }
        live.resize(w);
        return q;
    };
    double new_q = quality_live();
    // Nodes to visit in the next pass: flagged when a neighbour moves away from them, then put in visiting order
    // (a flag scan when many are flagged, a sort of the flagged list otherwise: the same sequence either way).
    std::vector<char> flag(n, 0);
    std::vector<int32_t> cur_list, next_list;
    bool first = true;
    int64_t moves = 0;
    auto visit = [&](int32_t v) {
        const int32_t c_old = comm[v];
        const double kv = deg[v];
        seen.clear();
        seen.push_back(c_old);
        neigh_w[c_old] = 0.0;
        for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
            const int32_t u = g.indices[e];
            if (u == v) continue;
            const int32_t c = comm[u];
            if (neigh_w[c] == -1.0) {
                neigh_w[c] = 0.0;
                seen.push_back(c);
            }
            neigh_w[c] += g.weights[e];
        }
        tot[c_old] -= kv;
        in_[c_old] -= 2.0 * neigh_w[c_old] + loops[v];
        int32_t best = c_old;
        double best_gain = neigh_w[c_old] - gamma * tot[c_old] * kv / m2;
        for (size_t t = 1; t < seen.size(); ++t) {
            const int32_t c = seen[t];
            const double gn = neigh_w[c] - gamma * tot[c] * kv / m2;
            if (gn > best_gain) {
                best_gain = gn;
                best = c;
            }
        }
        tot[best] += kv;
        in_[best] += 2.0 * neigh_w[best] + loops[v];
        comm[v] = best;
        if (best != c_old) {
            ++moves;
            for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
                const int32_t u = g.indices[e];
                if (u != v && comm[u] != best && !flag[u]) {
                    flag[u] = 1;
                    next_list.push_back(u);
                }
            }
        }
        for (int32_t c : seen) neigh_w[c] = -1.0;
    };
#ifdef DDX_LOUVAIN_DEBUG
    const bool dbg = true;           // trace of the passes on stderr: debug builds only (-DDDX_LOUVAIN_DEBUG)
#else
    const bool dbg = false;
#endif
    while (true) {
        const double cur_q = new_q;
        moves = 0;
        next_list.clear();
        if (first) {
            for (int64_t oi = 0; oi < n; ++oi) visit((int32_t)((start + oi) % n));
            first = false;
        } else {
            for (int32_t v : cur_list) visit(v);
        }
        new_q = quality_live();
        if (dbg) std::fprintf(stderr, "[louvain]   pass: n=%lld moves=%lld dQ=%.3e\n", (long long)n, (long long)moves, new_q - cur_q);
        if (moves > 0) improved = true;
        if (!(moves > 0 && new_q - cur_q > kMinGain)) break;
        if ((int64_t)next_list.size() * 8 > n) {
            cur_list.clear();
            for (int64_t oi = 0; oi < n; ++oi) {
                const int32_t v = (int32_t)((start + oi) % n);
                if (flag[v]) cur_list.push_back(v);
            }
        } else {
            cur_list = next_list;
            std::sort(cur_list.begin(), cur_list.end(), [&](int32_t x, int32_t y) {
                const int64_t kx = x >= start ? x - start : x - start + n;
                const int64_t ky = y >= start ? y - start : y - start + n;
                return kx < ky;
            });
        }
        for (int32_t v : next_list) flag[v] = 0;
    }
    if (q_out) *q_out = new_q;
    return improved;
}

// super-node graph; communities renumbered by ascending id; renum[c_old] = new id or -1
void aggregate(const Graph& g, const std::vector<int32_t>& comm, Graph& out, std::vector<int32_t>& renum) {
    const int64_t n = g.n();
    renum.assign(n, -1);
    for (int64_t v = 0; v < n; ++v) renum[comm[v]] = 0;
    int32_t k = 0;
    for (int64_t c = 0; c < n; ++c)
        if (renum[c] == 0) renum[c] = k++;
    // members grouped by new id, ascending node order inside a group (counting sort)
    std::vector<int64_t> start(k + 1, 0);
    for (int64_t v = 0; v < n; ++v) start[renum[comm[v]] + 1]++;
    for (int32_t c = 0; c < k; ++c) start[c + 1] += start[c];
    std::vector<int32_t> members(n);
    {
        std::vector<int64_t> cur(start.begin(), start.end() - 1);
        for (int64_t v = 0; v < n; ++v) members[cur[renum[comm[v]]]++] = (int32_t)v;
    }
    out.indptr.assign(1, 0);
    out.indices.clear();
    out.weights.clear();
    std::vector<double> acc(k, 0.0);
    std::vector<char> touched(k, 0);
    std::vector<int32_t> tl;
    for (int32_t cn = 0; cn < k; ++cn) {
        tl.clear();
        for (int64_t mi = start[cn]; mi < start[cn + 1]; ++mi) {
            const int32_t v = members[mi];
            for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
                const int32_t t = renum[comm[g.indices[e]]];
                if (!touched[t]) {
                    touched[t] = 1;
                    tl.push_back(t);
                    acc[t] = g.weights[e];
                } else {
                    acc[t] += g.weights[e];
                }
            }
        }
        std::sort(tl.begin(), tl.end());
        for (int32_t t : tl) {
            out.indices.push_back(t);
            out.weights.push_back(acc[t]);
            touched[t] = 0;
        }
        out.indptr.push_back((int64_t)out.indices.size());
    }
}

// ------------------------------------------------------------------------------------------------
// Part A of the specification (oracle/louvain_ref.py:presweep): synchronous sweeps on integer-quantised
// weights, then exact aggregation.  This is the host statement of what k_louvain.hip runs on the GPU.
// ------------------------------------------------------------------------------------------------
constexpr double kWeightScale = 1048576.0;   // 2^20

void quantise(const Graph& g, std::vector<int64_t>& wq, std::vector<int64_t>& K, int64_t& m2) {
    const int64_t n = g.n();
    const int64_t nnz = (int64_t)g.indices.size();
    wq.resize(nnz);
    for (int64_t e = 0; e < nnz; ++e) wq[e] = (int64_t)std::nearbyint(g.weights[e] * kWeightScale);
    K.assign(n, 0);
    m2 = 0;
    for (int64_t v = 0; v < n; ++v) {
        int64_t s = 0;
        for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) s += wq[e];
        K[v] = s;
        m2 += s;
    }
}

// `sweeps` sweeps of `subrounds` synchronous sub-rounds from the partition in comm (oracle/louvain_ref.py:_sync_sweeps):
// in sub-round r of sweep s the nodes with (v + s) % subrounds == r decide at once, the others stay.
void sync_sweeps(const Graph& g, const std::vector<int64_t>& wq, const std::vector<int64_t>& K, int64_t m2, double gamma, int sweeps,
                 int subrounds, std::vector<int32_t>& comm) {
    const int64_t n = g.n();
    if (subrounds < 1) subrounds = 1;
    std::vector<int32_t> next(n);
    std::vector<int64_t> tot(n), acc(n, -1);
    std::vector<int32_t> size(n), seen;
    for (int sweep = 0; sweep < sweeps && m2 > 0; ++sweep) {
        int64_t moves = 0;
        for (int r = 0; r < subrounds; ++r) {
            std::fill(tot.begin(), tot.end(), 0);
            std::fill(size.begin(), size.end(), 0);
            for (int64_t v = 0; v < n; ++v) {
                tot[comm[v]] += K[v];
                size[comm[v]]++;
            }
            const double m2d = (double)m2;
            for (int64_t v = 0; v < n; ++v) {
                const int32_t own = comm[v];
                next[v] = own;
                if ((int)((v + sweep) % subrounds) != r) continue;
                seen.clear();
                for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
                    const int32_t u = g.indices[e];
                    if (u == v) continue;
                    const int32_t c = comm[u];
                    if (acc[c] < 0) {
                        acc[c] = 0;
                        seen.push_back(c);
                    }
                    acc[c] += wq[e];
                }
                const double kv = (double)K[v];
                const int64_t w_own = acc[own] < 0 ? 0 : acc[own];
                const double own_score = (double)w_own * m2d - (gamma * (double)(tot[own] - K[v])) * kv;
                int32_t best = -1;
                double best_score = 0.0;
                for (int32_t c : seen) {
                    if (c == own) continue;
                    const double sc = (double)acc[c] * m2d - (gamma * (double)tot[c]) * kv;
                    if (best < 0 || sc > best_score || (sc == best_score && c < best)) {
                        best = c;
                        best_score = sc;
                    }
                }
                for (int32_t c : seen) acc[c] = -1;
                if (best >= 0 && best_score > own_score && !(size[own] == 1 && size[best] == 1 && best > own)) {
                    next[v] = best;
                    ++moves;
                }
            }
            comm.swap(next);
        }
        if (moves == 0) break;
    }
}

// labels 0..K-1 by ascending smallest member (oracle/louvain_ref.py:canonical_labels); ids in `labels` are < bound
void canonical_labels(std::vector<int32_t>& labels, int64_t bound) {
    const int64_t n = (int64_t)labels.size();
    std::vector<int32_t> rank(bound, -1);
    int32_t k = 0;
    for (int64_t v = 0; v < n; ++v) {         // first occurrence in node order = smallest member
        int32_t& r = rank[labels[v]];
        if (r < 0) r = k++;
        labels[v] = r;
    }
}

// One level of part C: refinement sweeps on g from the partition `labels` (ids in [0, n) are the community ids; anything
// else is canonicalised first).  canonical: number the result by ascending smallest member, else leave the raw ids.
void refine(const Graph& g, double gamma, int sweeps, int subrounds, std::vector<int32_t>& labels, bool canonical) {
    const int64_t n = g.n();
    int32_t lo = 0, hi = -1;
    for (int64_t v = 0; v < n; ++v) { lo = std::min(lo, labels[v]); hi = std::max(hi, labels[v]); }
    if (lo < 0 || hi >= n) {
        // (negative ids are rejected by the callers; ids >= n: rank by first appearance)
        canonical_labels(labels, (int64_t)hi + 1);
    }
    std::vector<int64_t> wq, K;
    int64_t m2;
    quantise(g, wq, K, m2);
    sync_sweeps(g, wq, K, m2, gamma, sweeps, subrounds, labels);
    if (canonical) canonical_labels(labels, n);
}

void presweep(const Graph& g, double gamma, int sweeps, int subrounds, std::vector<int32_t>& member, Graph& coarse) {
    const int64_t n = g.n();
    std::vector<int64_t> wq, K;
    int64_t m2;
    quantise(g, wq, K, m2);
    std::vector<int32_t> comm(n);
    for (int64_t v = 0; v < n; ++v) comm[v] = (int32_t)v;
    sync_sweeps(g, wq, K, m2, gamma, sweeps, subrounds, comm);
    // exact aggregation; coarse nodes numbered by ascending community id
    std::vector<int32_t> renum(n, -1);
    for (int64_t v = 0; v < n; ++v) renum[comm[v]] = 0;
    int32_t k = 0;
    for (int64_t c = 0; c < n; ++c)
        if (renum[c] == 0) renum[c] = k++;
    member.resize(n);
    for (int64_t v = 0; v < n; ++v) member[v] = renum[comm[v]];
    std::vector<int64_t> start(k + 1, 0);
    for (int64_t v = 0; v < n; ++v) start[member[v] + 1]++;
    for (int32_t c = 0; c < k; ++c) start[c + 1] += start[c];
    std::vector<int32_t> members(n);
    {
        std::vector<int64_t> cur(start.begin(), start.end() - 1);
        for (int64_t v = 0; v < n; ++v) members[cur[member[v]]++] = (int32_t)v;
    }
    coarse.indptr.assign(1, 0);
    coarse.indices.clear();
    coarse.weights.clear();
    std::vector<int64_t> sum(k, 0);
    std::vector<char> touched(k, 0);
    std::vector<int32_t> tl;
    for (int32_t cn = 0; cn < k; ++cn) {
        tl.clear();
        for (int64_t mi = start[cn]; mi < start[cn + 1]; ++mi) {
            const int32_t v = members[mi];
            for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
                const int32_t t = member[g.indices[e]];
                if (!touched[t]) {
                    touched[t] = 1;
                    tl.push_back(t);
                    sum[t] = 0;
                }
                sum[t] += wq[e];
            }
        }
        std::sort(tl.begin(), tl.end());
        for (int32_t t : tl) {
            coarse.indices.push_back(t);
            coarse.weights.push_back((double)sum[t] / kWeightScale);
            touched[t] = 0;
        }
        coarse.indptr.push_back((int64_t)coarse.indices.size());
    }
}

// Part B: multi-level sequential optimisation; labels numbered by ascending representative id
void sequential_levels(Graph& g, double gamma, uint64_t seed, std::vector<int32_t>& membership, double* q_out) {
    const int64_t n_nodes = g.n();
    SplitMix64 rng(seed);
    membership.resize(n_nodes);
    for (int64_t v = 0; v < n_nodes; ++v) membership[v] = (int32_t)v;
    std::vector<int32_t> comm, renum;
    double q = 0.0;
#ifdef DDX_LOUVAIN_DEBUG
    const bool dbg = true;           // trace of the passes on stderr: debug builds only (-DDDX_LOUVAIN_DEBUG)
#else
    const bool dbg = false;
#endif
    while (true) {
        auto t0 = std::chrono::steady_clock::now();
        const bool improved = one_level(g, gamma, rng, comm, &q);
        auto t1 = std::chrono::steady_clock::now();
        Graph next;
        aggregate(g, comm, next, renum);
        if (dbg) {
            auto t2 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[louvain] level n=%lld nnz=%lld -> %lld nodes; move %.1f ms, aggregate %.1f ms, Q=%.6f\n",
                         (long long)g.n(), (long long)g.indices.size(), (long long)next.n(),
                         std::chrono::duration<double, std::milli>(t1 - t0).count(),
                         std::chrono::duration<double, std::milli>(t2 - t1).count(), q);
        }
        for (int64_t v = 0; v < n_nodes; ++v) membership[v] = renum[comm[membership[v]]];
        g.indptr.swap(next.indptr);
        g.indices.swap(next.indices);
        g.weights.swap(next.weights);
        if (!improved) break;
    }
    if (q_out) *q_out = q;
}

// ------------------------------------------------------------------------------------------------
// Part B' of the specification (oracle/louvain_ref.py:_leiden_sequential): Leiden (Traag, Waltman, van Eck 2019)
// with the refinement's randomised merge taken at theta -> 0.  Stands in for leidenalg, which the reference reaches
// through sc.tl.leiden (dd.py:329,337-342).
// ------------------------------------------------------------------------------------------------
constexpr int kLeidenMaxIterations = 16;

void degrees(const Graph& g, std::vector<double>& deg, std::vector<double>& loops, double& m2) {
    const int64_t n = g.n();
    deg.assign(n, 0.0);
    loops.assign(n, 0.0);
    for (int64_t v = 0; v < n; ++v) {
        double s = 0.0;
        for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
            s += g.weights[e];
            if (g.indices[e] == v) loops[v] += g.weights[e];
        }
        deg[v] = s;
    }
    m2 = 0.0;
    for (int64_t v = 0; v < n; ++v) m2 += deg[v];
}

// local moving from the partition in comm (ids < n), with the option of leaving for an empty community
bool leiden_move(const Graph& g, double gamma, SplitMix64& rng, std::vector<int32_t>& comm) {
    const int64_t n = g.n();
    std::vector<double> deg, loops;
    double m2;
    degrees(g, deg, loops, m2);
    if (m2 == 0.0) return false;
    std::vector<double> tot(n, 0.0), in_(n, 0.0);
    std::vector<int32_t> size(n, 0);
    for (int64_t v = 0; v < n; ++v) {
        tot[comm[v]] += deg[v];
        size[comm[v]]++;
    }
    for (int64_t v = 0; v < n; ++v)
        for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e)
            if (comm[g.indices[e]] == comm[v]) in_[comm[v]] += g.weights[e];
    std::vector<int32_t> empty;
    for (int64_t c = n - 1; c >= 0; --c)
        if (size[c] == 0) empty.push_back((int32_t)c);
    const int64_t start = (int64_t)(rng.next() % (uint64_t)n);
    std::vector<double> neigh_w(n, -1.0);
    std::vector<int32_t> seen;
    seen.reserve(256);
    auto quality = [&]() {
        double q = 0.0;
        for (int64_t c = 0; c < n; ++c)
            if (tot[c] > 0.0) {
                const double b = tot[c] / m2;
                q += in_[c] / m2 - gamma * b * b;
            }
        return q;
    };
    bool improved = false;
    double new_q = quality();
    std::vector<char> active(n, 1), next_active(n, 0);
    while (true) {
        const double cur_q = new_q;
        int64_t moves = 0;
        std::fill(next_active.begin(), next_active.end(), 0);
        for (int64_t oi = 0; oi < n; ++oi) {
            const int32_t v = (int32_t)((start + oi) % n);
            if (!active[v]) continue;
            const int32_t c_old = comm[v];
            const double kv = deg[v];
            seen.clear();
            seen.push_back(c_old);
            neigh_w[c_old] = 0.0;
            for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
                const int32_t u = g.indices[e];
                if (u == v) continue;
                const int32_t c = comm[u];
                if (neigh_w[c] == -1.0) {
                    neigh_w[c] = 0.0;
                    seen.push_back(c);
                }
                neigh_w[c] += g.weights[e];
            }
            tot[c_old] -= kv;
            in_[c_old] -= 2.0 * neigh_w[c_old] + loops[v];
            size[c_old]--;
            int32_t best = c_old;
            double best_gain = neigh_w[c_old] - gamma * tot[c_old] * kv / m2;
            for (size_t t = 1; t < seen.size(); ++t) {
                const int32_t c = seen[t];
                const double gn = neigh_w[c] - gamma * tot[c] * kv / m2;
                if (gn > best_gain) {
                    best_gain = gn;
                    best = c;
                }
            }
            double w_best = neigh_w[best];
            if (best_gain < 0.0 && size[c_old] > 0) {
                best = empty.back();
                empty.pop_back();
                w_best = 0.0;
            }
            tot[best] += kv;
            in_[best] += 2.0 * w_best + loops[v];
            size[best]++;
            comm[v] = best;
            if (best != c_old) {
                ++moves;
                if (size[c_old] == 0) empty.push_back(c_old);
                for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
                    const int32_t u = g.indices[e];
                    if (u != v && comm[u] != best) next_active[u] = 1;
                }
            }
            for (int32_t c : seen) neigh_w[c] = -1.0;
        }
        new_q = quality();
        if (moves > 0) improved = true;
        if (!(moves > 0 && new_q - cur_q > kMinGain)) break;
        active.swap(next_active);
    }
    return improved;
}

// refinement: merges of single nodes into well-connected groups inside their community
void leiden_refine(const Graph& g, double gamma, SplitMix64& rng, const std::vector<int32_t>& comm,
                   std::vector<int32_t>& ref) {
    const int64_t n = g.n();
    std::vector<double> deg, loops;
    double m2;
    degrees(g, deg, loops, m2);
    ref.resize(n);
    for (int64_t v = 0; v < n; ++v) ref[v] = (int32_t)v;
    if (m2 == 0.0) return;
    std::vector<double> ktot(n, 0.0), ext(n, 0.0);
    for (int64_t v = 0; v < n; ++v) ktot[comm[v]] += deg[v];
    for (int64_t v = 0; v < n; ++v) {
        double s = 0.0;
        for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
            const int32_t u = g.indices[e];
            if (u != v && comm[u] == comm[v]) s += g.weights[e];
        }
        ext[v] = s;
    }
    std::vector<double> rtot(deg), neigh_w(n, -1.0);
    std::vector<int32_t> rsize(n, 1), seen;
    const int64_t start = (int64_t)(rng.next() % (uint64_t)n);
    for (int64_t oi = 0; oi < n; ++oi) {
        const int32_t v = (int32_t)((start + oi) % n);
        if (rsize[ref[v]] != 1) continue;
        const double kc = ktot[comm[v]];
        const double kv = deg[v];
        if (!(ext[v] >= gamma * kv * (kc - kv) / m2)) continue;
        seen.clear();
        for (int64_t e = g.indptr[v]; e < g.indptr[v + 1]; ++e) {
            const int32_t u = g.indices[e];
            if (u == v || comm[u] != comm[v]) continue;
            const int32_t r = ref[u];
            if (neigh_w[r] == -1.0) {
                neigh_w[r] = 0.0;
                seen.push_back(r);
            }
            neigh_w[r] += g.weights[e];
        }
        int32_t best = -1;
        double best_gain = 0.0;
        for (int32_t r : seen) {
            if (!(ext[r] >= gamma * rtot[r] * (kc - rtot[r]) / m2)) continue;
            const double gn = neigh_w[r] - gamma * rtot[r] * kv / m2;
            if (gn > best_gain) {
                best_gain = gn;
                best = r;
            }
        }
        if (best >= 0) {
            ext[best] = ext[best] + ext[v] - 2.0 * neigh_w[best];
            rtot[best] += kv;
            rsize[best]++;
            rsize[v] = 0;
            ref[v] = best;
        }
        for (int32_t r : seen) neigh_w[r] = -1.0;
    }
}

// number of distinct ids in comm (ids < n); renum[c] = rank of c among the used ids, -1 when unused
int32_t rank_ids(const std::vector<int32_t>& comm, int64_t n, std::vector<int32_t>& renum) {
    renum.assign(n, -1);
    for (size_t v = 0; v < comm.size(); ++v) renum[comm[v]] = 0;
    int32_t k = 0;
    for (int64_t c = 0; c < n; ++c)
        if (renum[c] == 0) renum[c] = k++;
    return k;
}

void leiden_levels(const Graph& g0, double gamma, uint64_t seed, std::vector<int32_t>& partition) {
    const int64_t n = g0.n();
    SplitMix64 rng(seed);
    partition.resize(n);
    for (int64_t v = 0; v < n; ++v) partition[v] = (int32_t)v;
    std::vector<int32_t> node_of(n), comm, ref, renum, cren, init;
    for (int it = 0; it < kLeidenMaxIterations; ++it) {
        Graph g = g0;
        for (int64_t v = 0; v < n; ++v) node_of[v] = (int32_t)v;
        init = partition;
        bool any_move = false;
        while (true) {
            comm = init;
            const bool moved = leiden_move(g, gamma, rng, comm);
            any_move = any_move || moved;
            const int64_t ng = g.n();
            if (rank_ids(comm, ng, cren) == ng) {
                for (int64_t v = 0; v < n; ++v) partition[v] = comm[node_of[v]];
                break;
            }
            leiden_refine(g, gamma, rng, comm, ref);
            Graph next;
            aggregate(g, ref, next, renum);
            const int64_t nn = next.n();
            init.assign(nn, 0);
            for (int64_t v = 0; v < ng; ++v) init[renum[ref[v]]] = cren[comm[v]];
            for (int64_t v = 0; v < n; ++v) node_of[v] = renum[ref[node_of[v]]];
            g.indptr.swap(next.indptr);
            g.indices.swap(next.indices);
            g.weights.swap(next.weights);
            if (nn == ng) {
                for (int64_t v = 0; v < n; ++v) partition[v] = init[node_of[v]];
                break;
            }
        }
        rank_ids(partition, n, renum);
        for (int64_t v = 0; v < n; ++v) partition[v] = renum[partition[v]];
        if (!any_move) break;
    }
}

// part A applied DDX_PRESWEEP_LEVELS times: graphs[0] = the original, graphs[l + 1] = aggregate of graphs[l], members[l]: V_l -> V_{l+1}
void presweep_levels(const Graph& g0, double gamma, std::vector<Graph>& graphs, std::vector<std::vector<int32_t>>& members) {
    graphs.clear();
    members.clear();
    graphs.push_back(g0);
    for (int lvl = 0; lvl < DDX_PRESWEEP_LEVELS; ++lvl) {
        Graph coarse;
        std::vector<int32_t> member;
        presweep(graphs.back(), gamma, DDX_PRESWEEPS, DDX_SUBROUNDS, member, coarse);
        members.push_back(std::move(member));
        graphs.push_back(std::move(coarse));
    }
}

// part C on the way back down: lab labels the nodes of graphs.back(); on return it labels the nodes of graphs[0]
void refine_down(const std::vector<Graph>& graphs, const std::vector<std::vector<int32_t>>& members, double gamma, std::vector<int32_t>& lab,
                 bool refine_levels) {
    for (int level = (int)members.size() - 1; level >= 0; --level) {
        const std::vector<int32_t>& m = members[level];
        std::vector<int32_t> down(m.size());
        for (size_t v = 0; v < m.size(); ++v) down[v] = lab[m[v]];
        if (refine_levels) refine(graphs[level], gamma, DDX_REFINE_SWEEPS, DDX_SUBROUNDS, down, false);      // ids of part B throughout
        lab.swap(down);
    }
    if (refine_levels && !members.empty()) canonical_labels(lab, graphs[0].n());
}

int load_graph(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights, Graph& g) {
    if (n_nodes < 0 || !indptr) return DDX_E_ARG;
    const int64_t nnz = n_nodes ? indptr[n_nodes] : 0;
    if (nnz > 0 && (!indices || !weights)) return DDX_E_ARG;
    g.indptr.assign(indptr, indptr + n_nodes + 1);
    g.indices.assign(indices, indices + nnz);
    g.weights.assign(weights, weights + nnz);
    for (int64_t e = 0; e < nnz; ++e)
        if (g.indices[e] < 0 || g.indices[e] >= n_nodes) return DDX_E_ARG;
    return DDX_OK;
}

}  // namespace

extern "C" int ddx_louvain_sequential(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights,
                                      double gamma, uint64_t seed, int32_t* labels_out, double* quality_out) {
    if (!labels_out) return DDX_E_ARG;
    if (n_nodes == 0) return DDX_OK;
    Graph g;
    const int rc = load_graph(n_nodes, indptr, indices, weights, g);
    if (rc != DDX_OK) return rc;
    std::vector<int32_t> membership;
    sequential_levels(g, gamma, seed, membership, quality_out);
    for (int64_t v = 0; v < n_nodes; ++v) labels_out[v] = membership[v];
    return DDX_OK;
}

extern "C" int ddx_leiden_sequential(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights,
                                     double gamma, uint64_t seed, int32_t* labels_out) {
    if (!labels_out) return DDX_E_ARG;
    if (n_nodes == 0) return DDX_OK;
    Graph g;
    const int rc = load_graph(n_nodes, indptr, indices, weights, g);
    if (rc != DDX_OK) return rc;
    std::vector<int32_t> partition;
    leiden_levels(g, gamma, seed, partition);
    for (int64_t v = 0; v < n_nodes; ++v) labels_out[v] = partition[v];
    return DDX_OK;
}

extern "C" int ddx_leiden(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights, double gamma,
                          uint64_t seed, int32_t* labels_out) {
    if (!labels_out) return DDX_E_ARG;
    if (n_nodes == 0) return DDX_OK;
    Graph g0;
    const int rc = load_graph(n_nodes, indptr, indices, weights, g0);
    if (rc != DDX_OK) return rc;
    std::vector<Graph> graphs;
    std::vector<std::vector<int32_t>> members;
    presweep_levels(g0, gamma, graphs, members);
    std::vector<int32_t> lab;
    leiden_levels(graphs.back(), gamma, seed, lab);
    refine_down(graphs, members, gamma, lab, true);
    for (int64_t v = 0; v < n_nodes; ++v) labels_out[v] = lab[v];
    return DDX_OK;
}

extern "C" int ddx_presweep(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights, double gamma,
                            int32_t sweeps, int32_t subrounds, int32_t* member_out, int64_t* n_coarse_out, int64_t* c_indptr_out,
                            int32_t* c_indices_out, double* c_weights_out) {
    if (!member_out || !n_coarse_out || !c_indptr_out) return DDX_E_ARG;
    *n_coarse_out = 0;
    if (n_nodes == 0) { c_indptr_out[0] = 0; return DDX_OK; }
    Graph g, coarse;
    const int rc = load_graph(n_nodes, indptr, indices, weights, g);
    if (rc != DDX_OK) return rc;
    std::vector<int32_t> member;
    presweep(g, gamma, sweeps, subrounds, member, coarse);
    const int64_t nc = coarse.n();
    *n_coarse_out = nc;
    for (int64_t v = 0; v < n_nodes; ++v) member_out[v] = member[v];
    for (int64_t c = 0; c <= nc; ++c) c_indptr_out[c] = coarse.indptr[c];
    const int64_t cnnz = (int64_t)coarse.indices.size();
    if (cnnz > 0 && (!c_indices_out || !c_weights_out)) return DDX_E_ARG;
    for (int64_t e = 0; e < cnnz; ++e) {
        c_indices_out[e] = coarse.indices[e];
        c_weights_out[e] = coarse.weights[e];
    }
    return DDX_OK;
}

extern "C" int ddx_refine(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights, const int32_t* labels_in,
                          double gamma, int32_t sweeps, int32_t subrounds, int32_t canonical, int32_t* labels_out) {
    if (!labels_out || !labels_in || sweeps < 0) return DDX_E_ARG;
    if (n_nodes == 0) return DDX_OK;
    Graph g;
    const int rc = load_graph(n_nodes, indptr, indices, weights, g);
    if (rc != DDX_OK) return rc;
    std::vector<int32_t> lab(labels_in, labels_in + n_nodes);
    for (int64_t v = 0; v < n_nodes; ++v)
        if (lab[v] < 0) return DDX_E_ARG;
    refine(g, gamma, sweeps, subrounds, lab, canonical != 0);
    for (int64_t v = 0; v < n_nodes; ++v) labels_out[v] = lab[v];
    return DDX_OK;
}

extern "C" int ddx_louvain(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights,
                           double gamma, uint64_t seed, int32_t* labels_out, double* quality_out) {
    if (!labels_out) return DDX_E_ARG;
    if (n_nodes == 0) return DDX_OK;
    Graph g0;
    const int rc = load_graph(n_nodes, indptr, indices, weights, g0);
    if (rc != DDX_OK) return rc;
    std::vector<Graph> graphs;
    std::vector<std::vector<int32_t>> members;
    presweep_levels(g0, gamma, graphs, members);
    std::vector<int32_t> lab;
    Graph top = graphs.back();                           // sequential_levels consumes its graph
    sequential_levels(top, gamma, seed, lab, quality_out);
    refine_down(graphs, members, gamma, lab, true);
    for (int64_t v = 0; v < n_nodes; ++v) labels_out[v] = lab[v];
    return DDX_OK;
}

// Helper threads of the restart batches, budgeted per PROCESS: several boosting iterations finish their device part at about the same
// time and each may run a batch of 20 independent restarts -- 7 jobs x 20 threads on a host that allows 16 CPUs' worth of time (the pods
// these GPUs come in) throttled the lane threads that feed the GPU.  A job's calling thread always works; helpers beyond it are taken
// from a budget (ddx_set_helper_threads; negative: unlimited) and handed back when the batch is done, so a job that runs alone (the
// last iteration of a fit, on the critical path) gets them all and jobs that overlap share them.  The result does not depend on the
// number of threads (runs are independent and applied in run order).
static std::atomic<int> g_helper_limit{-1};
static std::atomic<int> g_helpers_out{0};

extern "C" int ddx_set_helper_threads(int32_t n) {
    g_helper_limit.store(n < 0 ? -1 : n);
    return DDX_OK;
}

static int take_helpers(int want) {
    if (want <= 0) return 0;
    const int limit = g_helper_limit.load();
    if (limit < 0) { g_helpers_out.fetch_add(want); return want; }
    int cur = g_helpers_out.load();
    for (;;) {
        const int take = std::min(want, limit - cur);
        if (take <= 0) return 0;
        if (g_helpers_out.compare_exchange_weak(cur, cur + take)) return take;
    }
}

static void give_helpers(int n) {
    if (n > 0) g_helpers_out.fetch_sub(n);
}

// PhenoGraph's restart rule on top of part B (upstream phenograph.core.runlouvain, reached from dd.py:320-322: the
// Louvain executable is run again and again, each time from another random node order; a run replaces the best result
// when its modularity exceeds the best by more than q_tol; the loop ends after `stall` consecutive runs without such a
// gain).  Run r uses seed + r, so the outcome is a function of (graph, gamma, seed, q_tol, stall) only.  Runs are
// independent of each other: a batch of `stall` of them is evaluated on host threads at once and the rule is then
// applied in run order; runs of the batch behind the stopping point are discarded, exactly as if never started.
extern "C" int ddx_louvain_best_of(int64_t n_nodes, const int64_t* indptr, const int32_t* indices, const double* weights,
                                   double gamma, uint64_t seed, double q_tol, int32_t stall, int32_t max_runs, int32_t threads,
                                   int32_t presweeps, int32_t* labels_out, double* quality_out, int32_t* runs_out) {
    if (!labels_out || stall < 1 || max_runs < 1) return DDX_E_ARG;
    if (runs_out) *runs_out = 0;
    if (quality_out) *quality_out = 0.0;
    if (n_nodes == 0) return DDX_OK;
    Graph g;
    const int rc = load_graph(n_nodes, indptr, indices, weights, g);
    if (rc != DDX_OK) return rc;
    std::vector<Graph> graphs;
    std::vector<std::vector<int32_t>> members;
    if (presweeps) {
        presweep_levels(g, gamma, graphs, members);
        g = graphs.back();
    }
    std::vector<int32_t> best;
    double best_q = 0.0;
    int32_t run = 0, updated = 0;
    bool have = false;
    const int nthreads = threads < 1 ? 1 : threads;
    while (run - updated < stall && run < max_runs) {
        const int batch = std::min<int32_t>(stall, max_runs - run);
        std::vector<std::vector<int32_t>> memb(batch);
        std::vector<double> q(batch, 0.0);
        const int helpers = take_helpers(std::min(nthreads, batch) - 1);
        std::atomic<int> next{0};
        auto work = [&]() {
            for (int b = next.fetch_add(1); b < batch; b = next.fetch_add(1)) {
                Graph copy = g;                      // sequential_levels consumes its graph
                sequential_levels(copy, gamma, seed + (uint64_t)(run + b), memb[b], &q[b]);
            }
        };
        if (helpers <= 0) {
            work();
        } else {
            std::vector<std::thread> pool;
            for (int t = 0; t < helpers; ++t) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
            give_helpers(helpers);
        }
        for (int b = 0; b < batch && run - updated < stall; ++b, ++run) {
            if (!have || q[b] - best_q > q_tol) {
                best.swap(memb[b]);
                best_q = q[b];
                updated = run;
                have = true;
            }
        }
    }
    if (presweeps) refine_down(graphs, members, gamma, best, true);      // part C on the kept run
    for (int64_t v = 0; v < n_nodes; ++v) labels_out[v] = best[v];
    if (quality_out) *quality_out = best_q;
    if (runs_out) *runs_out = run;
    return DDX_OK;
}

extern "C" int ddx_relabel_by_size(int64_t n, const int32_t* labels, int64_t min_cluster_size, int64_t* out) {
    if (n < 0 || (n > 0 && (!labels || !out))) return DDX_E_ARG;
    int32_t maxl = -1;
    for (int64_t i = 0; i < n; ++i) {
        if (labels[i] < 0) return DDX_E_ARG;
        maxl = std::max(maxl, labels[i]);
    }
    std::vector<int64_t> count((size_t)maxl + 1, 0);
    for (int64_t i = 0; i < n; ++i) count[labels[i]]++;
    std::vector<int32_t> order;
    for (int32_t c = 0; c <= maxl; ++c)
        if (count[c] > 0) order.push_back(c);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return count[a] > count[b]; });
    std::vector<int64_t> map((size_t)maxl + 1, -1);
    int64_t next = 0;
    for (int32_t c : order) {
        if (min_cluster_size >= 0 && count[c] <= min_cluster_size) map[c] = -1;
        else map[c] = next++;
    }
    for (int64_t i = 0; i < n; ++i) out[i] = map[labels[i]];
    return DDX_OK;
}
