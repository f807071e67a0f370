"""Exact kNN on embeddings chosen to hurt the pruning structure (GPU only; cells are in use from 16 384 points).

The benchmark embedding is kind to the search: twelve cell types, smooth densities.  These are not: thousands of identical
points (every candidate list of theirs overflows and goes through the rescan; ties are decided by index), pure noise without
any structure (the cell tests prune little), points on a line at equal spacing (exact distance ties everywhere), a few far
outliers, 50 components (the 64-wide kernels) and 100 neighbours (the bound pass in several launches).  Every query is
compared with a float64 brute force that resolves ties by index (torch on the GPU: subtract, square, add, component by
component -- the oracle's arithmetic; the oracle's own statement, numpy on the host, checks a sample of it, and
`orc.knn_bruteforce_f64` proper a small case)."""
import numpy as np
import pytest

from oracle import dd_oracle as orc

pytestmark = pytest.mark.gpu


def _brute_force_ties(emb, k, include_self):
    """(indices, squared distances) of the k nearest by (distance, index), whatever the number of ties."""
    import torch

    M, C = emb.shape
    E = torch.from_numpy(emb).to("cuda:0", torch.float64)
    out_i = np.empty((M, k), np.int64)
    out_d = np.empty((M, k))
    step = max(64, min(2048, (1 << 27) // M))
    for s in range(0, M, step):
        q = torch.arange(s, min(M, s + step), device="cuda:0")
        d2 = torch.zeros((len(q), M), dtype=torch.float64, device="cuda:0")
        for c in range(C):
            diff = E[q, c][:, None] - E[None, :, c]
            d2 += diff * diff
        if not include_self:
            d2[torch.arange(len(q)), q] = float("inf")
        kth = torch.topk(d2, k, dim=1, largest=False).values[:, -1]
        rows, cols = torch.nonzero(d2 <= kth[:, None], as_tuple=True)          # everything tied with the k-th comes along
        vals = d2[rows, cols].cpu().numpy()
        rows, cols = rows.cpu().numpy(), cols.cpu().numpy()
        order = np.lexsort((cols, vals, rows))
        rows, cols, vals = rows[order], cols[order], vals[order]
        first = np.searchsorted(rows, np.arange(len(q)))
        take = first[:, None] + np.arange(k)[None, :]
        out_i[s:s + len(q)] = cols[take]
        out_d[s:s + len(q)] = vals[take]
    return out_i, out_d


def _oracle_rows_ties(emb, queries, k, include_self):
    """``orc.knn_bruteforce_f64``'s arithmetic and (distance, index) order for the given queries only (numpy on the host)."""
    e = np.asarray(emb, dtype=np.float64)
    out = np.empty((len(queries), k), dtype=np.int64)
    for r, q in enumerate(queries):
        d2 = np.zeros(e.shape[0])
        for c in range(e.shape[1]):
            diff = e[q, c] - e[:, c]
            d2 += diff * diff
        if not include_self:
            d2[q] = np.inf
        kth = np.partition(d2, k - 1)[k - 1]
        cand = np.flatnonzero(d2 <= kth)
        out[r] = cand[np.lexsort((cand, d2[cand]))][:k]
    return out


def _clusters(rng, n, c, centres=8, spread=1.0):
    mu = rng.normal(size=(centres, c)) * 6.0
    return (mu[rng.integers(0, centres, size=n)] + rng.normal(size=(n, c)) * spread).astype(np.float32)


def _embeddings():
    rng = np.random.default_rng(12)
    out = {}
    e = _clusters(rng, 24_000, 30)
    e[2_000:5_000] = e[7]                        # 3 001 identical points
    e[9_000:11_000] = e[8_999]                   # 2 001 more
    out["duplicates"] = (e, 30, False)
    out["noise"] = (rng.normal(size=(20_000, 30)).astype(np.float32), 30, False)
    line = np.zeros((20_000, 30), np.float32)
    line[:, 0] = np.arange(20_000, dtype=np.float32) * 0.25          # exact spacing: ties between left and right neighbours
    line[:, 1] = 3.0
    out["line"] = (line[rng.permutation(20_000)], 30, True)
    e = _clusters(rng, 20_000, 30)
    e[::997] *= 40.0                             # a handful of points far from every centre
    out["outliers"] = (e, 30, False)
    out["wide_many"] = (_clusters(rng, 18_000, 50), 100, False)
    return out


@pytest.mark.parametrize("name", ["duplicates", "noise", "line", "outliers", "wide_many", "duplicates-wide_waves", "outliers-wide_waves"])
def test_every_query_on_a_hostile_embedding(name, monkeypatch):
    from doubletdetection_amd import _lib

    if name.endswith("-wide_waves"):             # the emit pass with 64 queries per wave (option knn_emit_rt=4: built, no faster, kept exact)
        monkeypatch.setitem(_lib.OPTIONS, "knn_emit_rt", "4")
        name = name.split("-")[0]
    emb, k, include_self = _embeddings()[name]
    ctx = _lib.Context(0)
    try:
        ctx.set_embedding(emb)
        ctx.knn(k, include_self)
        idx, dist = ctx.get_knn()
        ref_i, ref_d2 = _brute_force_ties(emb, k, include_self)
        np.testing.assert_array_equal(idx, ref_i)
        np.testing.assert_array_equal(dist, np.sqrt(ref_d2))
        # the brute force itself against the oracle's on a sample
        sample = np.sort(np.random.default_rng(3).choice(len(emb), size=200, replace=False))
        np.testing.assert_array_equal(ref_i[sample], _oracle_rows_ties(emb, sample, k, include_self))
        if name == "duplicates":
            assert ctx.knn_overflow_count() >= 5_000          # the identical points all overflowed and were rescanned
    finally:
        ctx.close()


def test_the_tie_aware_brute_force_is_the_oracles():
    rng = np.random.default_rng(5)
    emb = rng.normal(size=(900, 30)).astype(np.float32)
    emb[100:160] = emb[3]
    for k, include_self in ((10, True), (30, False)):
        oi, od = orc.knn_bruteforce_f64(emb, k, include_self=include_self)
        bi, bd2 = _brute_force_ties(emb, k, include_self)
        np.testing.assert_array_equal(bi, oi)
        np.testing.assert_array_equal(np.sqrt(bd2), od)
