// Host float64 pieces of the scoring stage (dd.py:344-383): hypergeometric log survival function
// (scipy.stats.hypergeom.logsf, scipy/stats/_discrete_distns.py:672-721) and the per-community
// bookkeeping.  O(#communities) work per iteration -- deliberately not a GPU kernel.
#include <cmath>
#include <cstdint>
#include <limits>
#include <unordered_map>
#include <vector>

#include "../../include/ddx.h"

namespace {

inline double betaln(double a, double b) { return std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b); }

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
    std::unordered_map<int64_t, int64_t> n_orig, n_synth;
    int64_t min_id = std::numeric_limits<int64_t>::max();
    for (int64_t i = 0; i < n_aug; ++i) {
        if (full[i] < min_id) min_id = full[i];
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
    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (int64_t i = 0; i < n_cells; ++i) {
        const auto& pr = table[full[i]];
        scores[i] = pr.first;
        log_p[i] = pr.second;
        if (min_id < 0 && full[i] == -1) scores[i] = log_p[i] = nan;
    }
    return DDX_OK;
}
