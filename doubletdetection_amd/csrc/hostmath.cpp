// Host float64 pieces of the scoring stage (dd.py:344-383): hypergeometric log survival function
// (scipy.stats.hypergeom.logsf, scipy/stats/_discrete_distns.py:672-721) and the per-community
// bookkeeping.  O(#communities) work per iteration -- deliberately not a GPU kernel.
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/ddx.h"

namespace {

// lgamma of the integer arguments the hypergeometric terms use, memoised: one scoring call evaluates it ~10^6 times
// (18 per term, ~50 terms per community, ~2 000 communities), always at integers <= M + 2.  The table holds
// std::lgamma(i) itself, so results are bit-identical to calling it; it only grows, under a mutex, and readers keep a
// reference to the version they started with.
std::mutex g_lg_mutex;
std::shared_ptr<const std::vector<double>> g_lg_table;

std::shared_ptr<const std::vector<double>> lgamma_table(int64_t upto) {
    std::lock_guard<std::mutex> lock(g_lg_mutex);
    if (!g_lg_table || (int64_t)g_lg_table->size() <= upto) {
        auto t = std::make_shared<std::vector<double>>();
        const size_t old = g_lg_table ? g_lg_table->size() : 0;
        t->resize((size_t)upto + 1025);
        for (size_t i = 0; i < old; ++i) (*t)[i] = (*g_lg_table)[i];
        for (size_t i = old; i < t->size(); ++i) (*t)[i] = std::lgamma((double)i);
        g_lg_table = t;
    }
    return g_lg_table;
}

thread_local const std::vector<double>* t_lg = nullptr;

inline double lg(double x) {
    // every argument on this path is a non-negative integer; fall back to the library for anything else
    if (t_lg && x >= 0.0 && x < (double)t_lg->size()) {
        const size_t i = (size_t)x;
        if ((double)i == x) return (*t_lg)[i];
    }
    return std::lgamma(x);
}

inline double betaln(double a, double b) { return lg(a) + lg(b) - lg(a + b); }

// scipy hypergeom._logpmf(k, M=tot, n=good, N=draw)
double logpmf(double k, double tot, double good, double draw) {
    const double bad = tot - good;
    return betaln(good + 1, 1) + betaln(bad + 1, 1) + betaln(tot - draw + 1, draw + 1) - betaln(k + 1, good - k + 1) -
           betaln(draw - k + 1, bad - draw + k + 1) - betaln(tot + 1, 1);
}

double logsumexp_range(int64_t k_lo, int64_t k_hi, double tot, double good, double draw) {
    if (k_hi < k_lo) return -std::numeric_limits<double>::infinity();
    std::vector<double> v((size_t)(k_hi - k_lo + 1));
    double mx = -std::numeric_limits<double>::infinity();
    for (int64_t k = k_lo; k <= k_hi; ++k) {
        v[(size_t)(k - k_lo)] = logpmf((double)k, tot, good, draw);
        if (v[(size_t)(k - k_lo)] > mx) mx = v[(size_t)(k - k_lo)];
    }
    if (!std::isfinite(mx)) return mx;
    double s = 0.0;
    for (double x : v) s += std::exp(x - mx);
    return std::log(s) + mx;
}

double hypergeom_logsf(int64_t k, int64_t M, int64_t n, int64_t N) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    if (!(M > 0 && n >= 0 && N >= 0 && n <= M && N <= M)) return nan;   // scipy _argcheck
    const int64_t lo = std::max<int64_t>(N - (M - n), 0), hi = std::min(n, N);
    if (k < lo) return 0.0;
    if (k >= hi) return -std::numeric_limits<double>::infinity();
    const double tot = (double)M, good = (double)n, draw = (double)N, quant = (double)k;
    if ((quant + 0.5) * (tot + 0.5) < (good - 0.5) * (draw - 0.5)) {
        // fewer terms below: log(1 - cdf)
        const double logcdf = logsumexp_range(lo, k, tot, good, draw);
        return std::log1p(-std::exp(logcdf));
    }
    return logsumexp_range(k + 1, hi, tot, good, draw);
}

}  // namespace

extern "C" int ddx_hypergeom_logsf(int64_t k, int64_t M, int64_t n, int64_t N, double* out) {
    if (!out) return DDX_E_ARG;
    *out = hypergeom_logsf(k, M, n, N);
    return DDX_OK;
}

extern "C" int ddx_score_communities(const int64_t* full, int64_t n_aug, int64_t n_cells, double* scores,
                                     double* log_p) {
    if (!full || !scores || !log_p || n_cells < 0 || n_aug < n_cells) return DDX_E_ARG;
    const int64_t S = n_aug - n_cells;
    const double nan = std::numeric_limits<double>::quiet_NaN();
    int64_t min_id = std::numeric_limits<int64_t>::max(), max_id = std::numeric_limits<int64_t>::min();
    for (int64_t i = 0; i < n_aug; ++i) {
        if (full[i] < min_id) min_id = full[i];
        if (full[i] > max_id) max_id = full[i];
    }
    const auto table_ref = lgamma_table(n_aug + 2);       // memoised lgamma for this call (this thread)
    struct Scope { Scope(const std::vector<double>* p) { t_lg = p; } ~Scope() { t_lg = nullptr; } } scope(table_ref.get());
    if (n_aug > 0 && min_id >= -1 && max_id < n_aug) {
        // the usual case (labels -1 .. K-1 from ddx_relabel_by_size): direct tables instead of hash maps
        const size_t K = (size_t)(max_id + 2);                   // slot 0 holds label -1
        std::vector<int64_t> n_orig(K, 0), n_synth(K, 0);
        for (int64_t i = 0; i < n_cells; ++i) n_orig[(size_t)(full[i] + 1)]++;
        for (int64_t i = n_cells; i < n_aug; ++i) n_synth[(size_t)(full[i] + 1)]++;
        std::vector<double> score(K, 0.0), logp(K, 0.0);
        for (size_t c = 0; c < K; ++c) {
            if (n_orig[c] == 0) continue;
            const int64_t oc = n_orig[c], sc = n_synth[c];
            score[c] = (double)sc / (double)(sc + oc);
            logp[c] = hypergeom_logsf(sc, n_aug, S, sc + oc);
        }
        for (int64_t i = 0; i < n_cells; ++i) {
            const size_t c = (size_t)(full[i] + 1);
            scores[i] = score[c];
            log_p[i] = logp[c];
            if (full[i] == -1) scores[i] = log_p[i] = nan;
        }
        return DDX_OK;
    }
    std::unordered_map<int64_t, int64_t> n_orig, n_synth;
    for (int64_t i = 0; i < n_aug; ++i) {
        if (i < n_cells) n_orig[full[i]]++; else n_synth[full[i]]++;
    }
    std::unordered_map<int64_t, std::pair<double, double>> table;
    for (const auto& kv : n_orig) {
        const int64_t oc = kv.second;
        auto it = n_synth.find(kv.first);
        const int64_t sc = (it == n_synth.end()) ? 0 : it->second;
        const double score = (double)sc / (double)(sc + oc);
        table[kv.first] = {score, hypergeom_logsf(sc, n_aug, S, sc + oc)};
    }
    for (int64_t i = 0; i < n_cells; ++i) {
        const auto& pr = table[full[i]];
        scores[i] = pr.first;
        log_p[i] = pr.second;
        if (min_id < 0 && full[i] == -1) scores[i] = log_p[i] = nan;
    }
    return DDX_OK;
}
