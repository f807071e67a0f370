"""Size-independent properties of the HIP path at BASELINE.json sizes (configs[1]: 50k x 20k, ~5 % nnz),
where the CPU oracle would take minutes per stage.  GPU only."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

N, G, DENS = 50_000, 20_000, 0.05


@pytest.fixture(scope="module")
def data():
    from doubletdetection_amd._synthetic import make_counts

    return make_counts(N, G, density=DENS, device="cuda:0", seed=11)


@pytest.fixture(scope="module")
def staged(data):
    from doubletdetection_amd import _lib

    ctx = _lib.Context(0)
    ctx.upload_raw(data)
    var = ctx.gene_variances()
    top = np.argsort(var)[-10000:]
    ctx.select_columns(top)
    rng = np.random.default_rng(0)
    S = N // 4
    parents = rng.choice(N, size=(S, 2), replace=False)
    ctx.create_doublets(parents)
    yield ctx, var, top, parents
    ctx.close()


def test_prologue_properties(data, staged):
    ctx, var, top, _ = staged
    # variance vector: float64 recomputation agrees to float32 accumulation error; selection is the top-H set
    d64 = data.astype(np.float64)
    m1 = np.asarray(d64.mean(axis=0)).ravel()
    m2 = np.asarray(d64.multiply(d64).mean(axis=0)).ravel()
    ref = m2 - m1 ** 2
    np.testing.assert_allclose(var, ref, rtol=2e-3, atol=1e-6)
    assert var[top].min() >= np.sort(var)[-10000]
    sub = ctx.get_counts()
    assert sub.shape == (N, 10000) and sub.has_sorted_indices
    assert np.all(np.diff(sub.indptr) >= 0)
    # column j of the restricted matrix is gene top[j]: column sums agree exactly (integer counts)
    np.testing.assert_array_equal(np.asarray(sub.sum(axis=0)).ravel(), np.asarray(data[:, top].sum(axis=0)).ravel())
    np.testing.assert_array_equal(ctx.lib_size(), np.asarray(sub.sum(axis=1)).ravel())


def test_doublet_linearity_and_canonical_form(staged):
    ctx, _, _, parents = staged
    sub = ctx.get_counts()
    synth = ctx.get_synth()
    assert synth.shape == (parents.shape[0], 10000)
    assert synth.has_canonical_format and np.all(synth.data != 0)
    lib = np.asarray(sub.sum(axis=1)).ravel()
    # linearity: row sums and column sums of parent0 + parent1
    np.testing.assert_array_equal(np.asarray(synth.sum(axis=1)).ravel(), lib[parents[:, 0]] + lib[parents[:, 1]])
    rows = np.random.default_rng(1).choice(parents.shape[0], size=300, replace=False)
    want = sub[parents[rows, 0]] + sub[parents[rows, 1]]
    got = synth[rows]
    assert (got != want).nnz == 0
    # entries of a doublet = union of its parents' supports
    nnz_par = np.diff(sub.indptr)
    assert np.all(np.diff(synth.indptr) <= nnz_par[parents[:, 0]] + nnz_par[parents[:, 1]])
    assert np.all(np.diff(synth.indptr) >= np.maximum(nnz_par[parents[:, 0]], nnz_par[parents[:, 1]]))


def test_pca_and_knn_properties(staged):
    ctx, _, _, parents = staged
    ctx.lognormalise(0.1)
    lib, med = ctx.aug_lib()
    assert med == np.median(lib)
    M, H, C = ctx.M, ctx.H, 30
    q0 = np.random.RandomState(0).normal(size=(H, C + 10)).astype(np.float32).astype(np.float64)
    ctx.pca(C, q0)
    emb, sing = ctx.embedding_f64()
    assert emb.shape == (M, C) and np.all(np.isfinite(emb))
    # U*S: columns are centred, mutually orthogonal, with norms equal to the singular values, sorted
    np.testing.assert_allclose(emb.mean(axis=0), 0.0, atol=1e-9 * sing[0])
    gram = emb.T @ emb
    np.testing.assert_allclose(gram, np.diag(sing ** 2), rtol=1e-9, atol=1e-9 * sing[0] ** 2)
    assert np.all(np.diff(sing) <= 0)
    # kNN: ordering, self exclusion, and exactness against a float64 brute force for sampled queries
    ctx.knn(30, False)
    idx, dist = ctx.get_knn()
    assert idx.min() >= 0 and idx.max() < M
    assert np.all(idx != np.arange(M)[:, None])
    assert np.all(np.diff(dist, axis=1) >= 0)
    e32 = ctx.embedding()
    e = e32.astype(np.float64)
    for qi in np.random.default_rng(3).choice(M, size=40, replace=False):
        d2 = np.zeros(M)
        for c in range(C):
            diff = e[qi, c] - e[:, c]
            d2 += diff * diff
        d2[qi] = np.inf
        order = np.lexsort((np.arange(M), d2))[:30]
        np.testing.assert_array_equal(idx[qi], order)
        np.testing.assert_array_equal(dist[qi], np.sqrt(d2[order]))
    # graph: symmetric, no self loops, weights in (0, 1]
    ip, ix, w = ctx.build_graph(0)
    Gm = sp.csr_matrix((w, ix, ip), shape=(M, M))
    assert abs(Gm - Gm.T).nnz == 0 and Gm.diagonal().sum() == 0
    assert w.min() > 0 and w.max() <= 1.0
    # community detection part A at this size: device == host statement, bit for bit
    from doubletdetection_amd import _lib
    m_dev, ip_dev, ix_dev, w_dev = ctx.coarsen_graph(1.0)
    total, gr = None, (ip, ix, w)
    for _ in range(_lib.PRESWEEP_LEVELS):
        mm, *gr = _lib.presweep(*gr, 1.0)
        total = mm if total is None else mm[total]
    np.testing.assert_array_equal(m_dev, total)
    np.testing.assert_array_equal(ip_dev, gr[0])
    np.testing.assert_array_equal(ix_dev, gr[1])
    np.testing.assert_array_equal(w_dev, gr[2])
    assert len(ip_dev) - 1 < M // 20                          # it coarsens by more than an order of magnitude
    coarse = _lib.louvain_sequential(ip_dev, ix_dev, w_dev, 1.0, 0)[0]
    np.testing.assert_array_equal(ctx.refine_communities(coarse, 1.0), _lib.louvain(ip, ix, w, 1.0, 0)[0])


def test_c2_pca_matches_f64_oracle(staged):
    """configs[1] (50 000 x 20 000 -> 62 500 x 10 000): device PCA against the float64 oracle and sklearn on the same matrix."""
    from conftest import pca_against_f64_oracle

    ctx = staged[0]
    ctx.lognormalise(0.1)
    pca_against_f64_oracle(ctx)


def test_c2_sparse_pca_scores_match_an_exact_decomposition(staged):
    """pseudocount = 1 at configs[1] (dd.py:296-297,308: sparse matrix, ARPACK upstream): the block Lanczos solver on the device
    (ddx_pca_exact_sparse, with the tolerance the classifier uses) against the exact truncated PCA of the matrix read back from
    the device -- eigen-decomposition of the 10 000 x 10 000 Gram matrix of the centred matrix in float64.  Bar: 1e-4 per
    component (north star), components 1..30."""
    import time

    ctx = staged[0]
    ctx.lognormalise(1.0)
    M, H, C = ctx.M, ctx.H, 30
    start = np.random.RandomState(0).normal(size=(H, C + 10))
    t0 = time.perf_counter()
    steps = ctx.pca_exact_sparse(C, start, tol=1e-6, max_steps=48)
    ctx.synchronize()
    t1 = time.perf_counter()
    assert ctx.lanczos_converged
    emb, sing = ctx.embedding_f64()
    A = np.empty((M, H), dtype=np.float64)
    for r in range(0, M, 4096):
        n = min(4096, M - r)
        A[r:r + n] = ctx.aug_dense_rows(r, n)
    A -= A.mean(axis=0)
    evals, evecs = np.linalg.eigh(A.T @ A)
    V = evecs[:, ::-1][:, :C]
    V = V * np.sign(V[np.argmax(np.abs(V), axis=0), np.arange(C)])
    want = A @ V
    rel = np.linalg.norm(emb - want, axis=0) / np.linalg.norm(want, axis=0)
    print(f"block Lanczos {M} x {H}: {steps} steps, {1e3 * (t1 - t0):.1f} ms; scores against the exact decomposition: max {rel.max():.2e} "
          f"(signal components 1-12: {rel[:12].max():.2e}); singular values max rel {np.abs(sing / np.sqrt(evals[::-1][:C]) - 1).max():.2e}")
    assert rel.max() <= 1e-4, rel
    np.testing.assert_allclose(sing, np.sqrt(evals[::-1][:C]), rtol=1e-7)
    ctx.lognormalise(0.1)                              # (the module's other tests expect the default matrix)


def test_operator_product_variants_agree_at_full_size(data, staged, monkeypatch):
    """The same randomized PCA through the implementations of the operator products: bit planes on the int8 matrix cores
    + LDS-staged sparse products for the entries other than 1 (the default from 4096 cells on; four or three 8-bit digits),
    the LDS-staged float32 operand with float32 products inside a trip for every entry (bitplane=0, the default of
    rounds 1-4), L2-gather float32 operand (option spmm=gather), L2-gather float64 operand (pca_gather=f64), LDS with
    float64 products, LDS in the quad geometry.
    The sparse variants compute the same products up to the float32 trip sums and the summation order; the
    float64 mode differs by the float32 rounding of the operand copies (< 1e-5 per component, as in the small
    oracle test); the bit planes replace nine in ten of those roundings by a 30-bit (22-bit) fixed point."""
    import os

    from doubletdetection_amd import _lib

    ctx, _, top, parents = staged
    ctx.lognormalise(0.1)
    M, H, C = ctx.M, ctx.H, 30
    q0 = np.random.RandomState(0).normal(size=(H, C + 10)).astype(np.float32).astype(np.float64)
    ctx.pca(C, q0)
    emb_bp, sing_bp = ctx.embedding_f64()           # the default: bit planes, four int8 digits

    def other(env):
        for k, v in env.items():
            monkeypatch.setitem(_lib.OPTIONS, k, v)
        c2 = _lib.Context(0)                        # the panel height of the mirror is fixed at upload
        try:
            c2.upload_raw(data)
            c2.gene_variances()
            c2.select_columns(top)
            c2.create_doublets(parents)
            c2.lognormalise(0.1)
            c2.pca(C, q0)
            return c2.embedding_f64()
        finally:
            c2.close()
            for k in env:
                monkeypatch.delitem(_lib.OPTIONS, k, raising=False)

    def rel_dev(a, b):
        return (np.linalg.norm(a - b, axis=0) / np.linalg.norm(b, axis=0)).max()

    emb_lds, sing_lds = other({"bitplane": "0"})
    # the column-major mirror built by counting sort (default) and by the stable radix sort it replaces hold the same
    # entries in the same order: bit-identical scores
    # (default: counting sort placed by LDS tiles; "scatter": counting sort with scattered stores; "sort": radix sort)
    for mode in ("sort", "scatter"):
        emb_m, sing_m = other({"mirror": mode, "bitplane": "0"})
        np.testing.assert_array_equal(emb_m, emb_lds)
        np.testing.assert_array_equal(sing_m, sing_lds)
        emb_m, sing_m = other({"mirror": mode})     # ... and the reduced mirrors of the bit-plane route are cut from it
        np.testing.assert_array_equal(emb_m, emb_bp)
        np.testing.assert_array_equal(sing_m, sing_bp)
    emb_g, sing_g = other({"spmm": "gather"})
    # (rounding noise of the two summation orders, amplified through seven power iterations of unconverged
    # trailing components: 4e-10 observed on the singular values)
    np.testing.assert_allclose(sing_g, sing_lds, rtol=1e-8)
    assert rel_dev(emb_g, emb_lds) < 1e-6
    emb_f, sing_f = other({"spmm": "gather", "pca_gather": "f64"})
    assert rel_dev(emb_lds, emb_f) < 1e-5
    # the other two LDS variants: float64 products inside a trip, and the quad geometry at width 40
    for env in ({"spmm_trip": "f64", "bitplane": "0"}, {"spmm_geom": "quad", "bitplane": "0"}):
        emb_v, sing_v = other(env)
        np.testing.assert_allclose(sing_v, sing_lds, rtol=1e-8)
        assert rel_dev(emb_v, emb_lds) < 2e-6, env
    # bit planes against the all-float64 run: closer than the float32 operand copies with four digits, inside the bar with three
    d4, d_lds = rel_dev(emb_bp, emb_f), rel_dev(emb_lds, emb_f)
    emb_3, sing_3 = other({"bp_digits": "3"})
    d3 = rel_dev(emb_3, emb_f)
    print(f"against the float64-gather run: bit planes 4 digits {d4:.2e}, 3 digits {d3:.2e}, float32 operand copies {d_lds:.2e}")
    np.testing.assert_allclose(sing_bp, sing_f, rtol=1e-8)
    assert d4 < 2e-6 and d3 < 1e-5, (d4, d3)
    # fewer digits in the power iterations before the last one only (option bp_digits_early, off by default): the last iteration and the
    # projection keep four.  What an early iteration loses perturbs the start of the next ones: the signal components forget it, the
    # unconverged trailing ones carry it to the end -- inside the bar here, but not everywhere (include/ddx.h): measured, left off
    # the same products on the MX matrix instruction (option bp_format=mx6: FP4 bitmap x six base-31 digits in FP6 = 28.7 bits; exact sums)
    emb_mx, sing_mx = other({"bp_format": "mx6"})
    d_mx = rel_dev(emb_mx, emb_f)
    print(f"against the float64-gather run: default (int8, four digits) {d4:.2e}, MX form {d_mx:.2e}; MX against int8 {rel_dev(emb_mx, emb_bp):.2e}")
    np.testing.assert_allclose(sing_mx, sing_f, rtol=1e-8)
    assert d_mx < 2e-6
    dev = {}
    for early in ("4", "3", "2"):
        emb_e, sing_e = other({"bp_digits_early": early})
        dev[early] = rel_dev(emb_e, emb_f)
        np.testing.assert_allclose(sing_e, sing_f, rtol=1e-6)
    print("early power iterations on fewer digits, against the float64-gather run: " + ", ".join(f"{k} digits {v:.2e}" for k, v in dev.items()))
    assert dev["3"] < 1e-5, dev
    # forced on / off does not depend on the automatic rule
    emb_2, sing_2 = other({"bitplane": "2"})
    np.testing.assert_array_equal(emb_2, emb_bp)
    # the sparse part of the bit-plane route through the packed blocks (default) and straight from the reduced CSR / mirror: the same
    # entries against the same operand copies in the same order; only where a unit's steps continue in the next 1 KB block does a
    # trip of eight become two of four (float32 sums inside a trip): rounding noise
    emb_p, sing_p = other({"residual": "plain"})
    np.testing.assert_allclose(sing_p, sing_bp, rtol=1e-9)
    assert rel_dev(emb_p, emb_bp) < 1e-6


@pytest.mark.parametrize("algorithm", ["louvain", "leiden"])
def test_fit_is_deterministic_and_finds_doublets(data, algorithm):
    from doubletdetection_amd import BoostClassifier

    res = []
    for _ in range(2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            clf = BoostClassifier(n_iters=3, clustering_algorithm=algorithm, random_state=7, n_jobs=-1).fit(data)
        res.append((clf.all_log_p_values_.copy(), clf.communities_.copy(), clf.doublet_score()))
    np.testing.assert_array_equal(res[0][0], res[1][0])          # tests/test_package.py:25-38 at full size
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_equal(np.ma.getdata(res[0][2]), np.ma.getdata(res[1][2]))
    assert clf.all_log_p_values_.shape == (3, N) and clf.synth_communities_.shape == (3, N // 4)
    # the generator plants 5 % true doublets (sums of two cells): their scores must rank above singlets'
    score = np.ma.filled(clf.doublet_score(), 0.0)
    lib = np.asarray(data.sum(axis=1)).ravel()
    top = np.argsort(score)[-N // 50:]
    assert lib[top].mean() > 1.3 * lib.mean()
