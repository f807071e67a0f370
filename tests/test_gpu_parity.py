"""Parity of the HIP path (through the C-ABI) against the reference's golden vectors and the CPU
oracle, stage by stage and end to end.  Needs a real MI355X: run with `pytest -m gpu` via gpurun."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import CASES, csr_from, golden_kwargs, load_golden
from oracle import dd_oracle as orc
from oracle import louvain_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from doubletdetection_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


def _same_csr(a, b):
    a = sp.csr_matrix(a); b = sp.csr_matrix(b)
    assert a.shape == b.shape
    np.testing.assert_array_equal(a.indptr, b.indptr)
    np.testing.assert_array_equal(a.indices, b.indices)
    np.testing.assert_array_equal(a.data, b.data)


def _ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


# ---- a4: memoised library sizes and normalised values (dd.py:178-184), bit exact -----------------
@pytest.mark.parametrize("case", CASES)
def test_counts_resident_and_memoised(ctx, case):
    g = load_golden(case)
    raw = csr_from(g, "raw_hvg")
    ctx.upload_counts(raw)
    _same_csr(ctx.get_counts(), raw)
    np.testing.assert_array_equal(ctx.lib_size(), g["lib_size"])
    np.testing.assert_array_equal(ctx.normed(), g["normed_data"])


# ---- a3: HVG prologue on the device (dd.py:165-176), bit exact ---------------------------------------
@pytest.mark.parametrize("case", ["case_a_hvg_pheno", "case_b_transposed_louvain", "case_e_pc1_sparse"])
def test_hvg_prologue(ctx, case):
    g = load_golden(case)
    kw = golden_kwargs(g)
    raw = orc.coerce_counts(csr_from(g, "counts"))
    ctx.upload_raw(raw)
    var = ctx.gene_variances()
    np.testing.assert_array_equal(var, orc.gene_variances(raw))       # float32, scipy's accumulation order
    top = np.argsort(var)[-kw["n_top_var_genes"]:]
    np.testing.assert_array_equal(top, g["top_var_genes"])
    ctx.select_columns(top)
    _same_csr(ctx.get_counts(), csr_from(g, "raw_hvg"))
    np.testing.assert_array_equal(ctx.lib_size(), g["lib_size"])
    np.testing.assert_array_equal(ctx.normed(), g["normed_data"])


def test_hvg_prologue_non_integer_values_and_empty_genes(ctx):
    # the accumulation order only shows with values whose partial sums round: use non-integers
    rng = np.random.default_rng(3)
    dense = (rng.random((700, 900)) < 0.2) * rng.gamma(2.0, 1.7, size=(700, 900))
    dense[:, 17] = 0; dense[:, 400:420] = 0; dense[5] = 0
    raw = sp.csr_matrix(dense.astype(np.float32))
    ctx.upload_raw(raw)
    var = ctx.gene_variances()
    np.testing.assert_array_equal(var, orc.gene_variances(raw))
    top = np.argsort(var)[-300:]
    ctx.select_columns(top)
    _same_csr(ctx.get_counts(), raw.tocsc()[:, top].tocsr())


# ---- a6: synthetic doublets (dd.py:385-402), bit exact ---------------------------------------------
@pytest.mark.parametrize("case", CASES)
def test_doublets_match_reference(ctx, case):
    g = load_golden(case)
    ctx.upload_counts(csr_from(g, "raw_hvg"))
    for it in range(g["parents"].shape[0]):
        ctx.create_doublets(g["parents"][it])
        _same_csr(ctx.get_synth(), csr_from(g, f"synth{it}"))


def test_doublets_edge_cases(ctx):
    rng = np.random.default_rng(0)
    # long rows (merged length > the 2048-entry LDS tile), empty rows, identical parents, explicit
    # zeros and cancelling values (scipy's csr_plus_csr drops exact zeros)
    dense = (rng.random((40, 6000)) < 0.55) * rng.integers(1, 9, size=(40, 6000))
    dense[3] = 0
    dense[4] = 0
    raw = sp.csr_matrix(dense.astype(np.float32))
    extra = sp.csr_matrix((np.array([0.0, 0.0, 5.0, -5.0], np.float32), (np.array([3, 5, 6, 7]), np.array([10, 11, 12, 12]))),
                          shape=raw.shape)
    raw = sp.csr_matrix(raw + extra)        # rows 6/7 carry +5/-5 at column 12
    raw.sort_indices()
    # put stored zeros back (scipy's add dropped them): build by hand
    coo = raw.tocoo()
    rows = np.r_[coo.row, [3, 5]]; cols = np.r_[coo.col, [10, 11]]; vals = np.r_[coo.data, [0.0, 0.0]].astype(np.float32)
    raw = sp.csr_matrix((vals, (rows, cols)), shape=raw.shape)
    raw.sort_indices()
    assert (raw.data == 0).sum() >= 1
    ctx.upload_counts(raw)
    parents = np.array([[0, 1], [2, 2], [3, 4], [3, 9], [6, 7], [5, 5], [10, 3], [39, 0], [6, 6], [7, 7]], dtype=np.int64)
    ctx.create_doublets(parents)
    got = ctx.get_synth()
    want = orc.create_doublets(raw, parents)
    want.sort_indices()
    _same_csr(got, want)
    assert got[2].nnz == 0                                    # two empty parents
    ctx.create_doublets(np.zeros((0, 2), dtype=np.int64))     # S = 0
    assert ctx.get_synth().shape == (0, raw.shape[1])


def test_library_sizes_integer_and_fractional_counts(ctx):
    """Small integer counts take an order-free row sum (exact in float32, so bit-identical to the reference whatever
    order its scipy sums in); fractional input falls back to a left-to-right float32 accumulation (csr_matvec order),
    which can differ from a scipy that reduces pairwise by an ulp or two."""
    rng = np.random.default_rng(5)
    dense = (rng.random((300, 900)) < 0.3) * rng.gamma(2.0, 37.123, size=(300, 900))
    raw = sp.csr_matrix(dense.astype(np.float32))
    raw.sort_indices()
    ctx.upload_counts(raw)
    parents = rng.integers(0, 300, size=(80, 2)).astype(np.int64)
    ctx.create_doublets(parents)
    ctx.lognormalise(0.1)
    synth = orc.create_doublets(raw, parents)
    want = np.concatenate([orc.library_sizes(raw), orc.library_sizes(synth)])
    lib, _ = ctx.aug_lib()
    np.testing.assert_allclose(lib, want, rtol=2e-6)
    seq = np.array([np.add.accumulate(r.data, dtype=np.float32)[-1] if r.nnz else np.float32(0) for r in sp.vstack((raw, synth)).tocsr()],
                   dtype=np.float32)
    np.testing.assert_array_equal(lib, seq)                   # the documented order, bit for bit
    # and the integer case through the fast path: same answer as any order
    ints = sp.csr_matrix(np.floor(dense / 20.0).astype(np.float32))
    ints.eliminate_zeros()
    ints.sort_indices()
    ctx.upload_counts(ints)
    ctx.create_doublets(parents)
    ctx.lognormalise(0.1)
    want = np.concatenate([orc.library_sizes(ints), orc.library_sizes(orc.create_doublets(ints, parents))])
    np.testing.assert_array_equal(ctx.aug_lib()[0], want)


def test_column_mirror_placement_modes(monkeypatch):
    """The column-major mirror (dd.py has no counterpart: it serves A^T Y of the PCA) built three ways -- LDS tiles
    (default), scattered stores, radix sort -- on a matrix that leaves the comfortable regime: rows holding every
    column (more than 16 entries of a row inside a tile window), a block of columns held by every row (tiles larger
    than the LDS range: spill path), empty rows and empty columns, a panel straddling original and synthetic rows;
    and counts outside the log-normalisation table (both passes: the row-major copy and the mirror must agree)."""
    from doubletdetection_amd import _lib

    rng = np.random.default_rng(11)
    N, H = 1700, 1300
    dense = (rng.random((N, H)) < 0.02) * rng.integers(1, 9, size=(N, H))
    dense[:, 40:150] = rng.integers(1, 5, size=(N, 110))          # columns stored in every row
    dense[100:130, :] = rng.integers(1, 4, size=(30, H))           # rows that hold every column
    dense[300:340, :] = 0                                          # empty rows
    dense[:, 700:760] = 0                                          # empty columns
    dense[5, 0] = 7
    dense = dense.astype(np.float64)
    # counts outside the (row, count) table of the log-normalisation: large and fractional values (evaluated in place)
    big = rng.random((N, H)) < 0.002
    dense[big] = rng.choice([17.0, 37.0, 250.0, 2.5, 0.25], size=int(big.sum()))
    dense[300:340, :] = 0
    dense[:, 700:760] = 0
    counts = sp.csr_matrix(dense.astype(np.float32))
    parents = rng.choice(N, size=(N // 4, 2), replace=False)
    Y = rng.normal(size=(N + N // 4, 5))

    def products(env):
        for k, v in env.items():
            monkeypatch.setitem(_lib.OPTIONS, k, v)
        c = _lib.Context(0)
        try:
            c.upload_counts(counts)
            c.create_doublets(parents)
            c.lognormalise(0.1)
            X = c.aug_dense_rows(0, c.M).astype(np.float64)
            return X, c.operator_apply(Y, 1)
        finally:
            c.close()
            for k in env:
                monkeypatch.delitem(_lib.OPTIONS, k, raising=False)

    X, tiles = products({})
    _, scatter = products({"mirror": "scatter"})
    _, sort = products({"mirror": "sort"})
    np.testing.assert_array_equal(tiles, sort)
    np.testing.assert_array_equal(scatter, sort)
    # against the row-major copy: A^T Y = (X - 1 mu^T)^T Y
    ref = (X - X.mean(axis=0)).T @ Y
    np.testing.assert_allclose(tiles, ref, rtol=0, atol=1e-9 * np.abs(ref).max())


# ---- a7: log-normalisation (dd.py:286-298) -----------------------------------------------------------
@pytest.mark.parametrize("case", ["case_a_hvg_pheno", "case_b_transposed_louvain", "case_d_replace_single"])
def test_lognormalised_matrix(ctx, case):
    g = load_golden(case)
    kw = golden_kwargs(g)
    raw = csr_from(g, "raw_hvg")
    ctx.upload_counts(raw)
    ctx.create_doublets(g["parents"][0])
    ctx.lognormalise(kw.get("pseudocount", 0.1))
    aug, aug_lib, med = orc.lognormalise(orc.l1_normalise_rows(raw), orc.library_sizes(raw), csr_from(g, "synth0"),
                                         kw.get("pseudocount", 0.1))
    lib, m = ctx.aug_lib()
    np.testing.assert_array_equal(lib, aug_lib)               # integer-valued float32: bit exact
    assert m == med
    dense = ctx.aug_dense_rows(0, ctx.M)
    # the reference's matrix rows stored in the golden file (numpy float32 log) vs correctly rounded log
    sel = g["pca_in0_rowsel"]
    # numpy's SIMD float32 log is not correctly rounded (documented max error ~4 ulp, ISA dependent);
    # the kernel returns the correctly rounded value, so agreement is "within numpy's own error"
    ulp = _ulp_diff(dense[sel], g["pca_in0_rows"])
    assert ulp.max() <= 4, ulp.max()
    assert (ulp == 0).mean() > 0.85, (ulp == 0).mean()
    ulp_all = _ulp_diff(dense, aug)
    assert ulp_all.max() <= 4
    # independent check of "correctly rounded": float64 log of the same float32 argument
    scaled = (sp.vstack((orc.l1_normalise_rows(raw), orc.l1_normalise_rows(csr_from(g, "synth0")))) * med).toarray()
    exact = np.log((scaled + np.float32(kw.get("pseudocount", 0.1))).astype(np.float64)).astype(np.float32)
    np.testing.assert_array_equal(dense, exact)
    vals, z = ctx.aug_values()
    assert np.all(z == np.float32(np.log(np.float32(kw.get("pseudocount", 0.1)))))
    assert vals.shape[0] == raw.nnz + csr_from(g, "synth0").nnz


# ---- a8: standard scaling (restated scanpy; parity unpinned upstream) --------------------------------
def test_scaled_matrix(ctx):
    g = load_golden("case_c_reftest_scaled")
    raw = csr_from(g, "raw_hvg")
    ctx.upload_counts(raw)
    ctx.create_doublets(g["parents"][0])
    ctx.lognormalise(0.1)
    ctx.scale(15.0)
    aug, _, _ = orc.lognormalise(orc.l1_normalise_rows(raw), orc.library_sizes(raw), csr_from(g, "synth0"), 0.1)
    want = orc.scale_like_scanpy(aug, 15)
    got = ctx.aug_dense_rows(0, ctx.M)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)
    sel = g["unpinned_pca_in0_rowsel"]
    np.testing.assert_allclose(got[sel], g["unpinned_pca_in0_rows"], rtol=2e-6, atol=2e-6)


# ---- a9: randomized PCA (dd.py:305-314 -> sklearn) ----------------------------------------------------
@pytest.mark.parametrize("gather", ["f32", "f64", "bitplane", "bitplane3", "bitplane_mx"])
@pytest.mark.parametrize("case", ["case_a_hvg_pheno", "case_b_transposed_louvain", "case_c_reftest_scaled",
                                  "case_d_replace_single"])
def test_pca_scores(case, gather, monkeypatch):
    # default: the operator products gather a float32-rounded copy of the 40-column iterate (float64
    # products and sums); option pca_gather=f64 gathers the float64 iterate itself.  A context takes the process-wide options
    # when it is created: make one for this setting.
    # "bitplane" / "bitplane3": the entries equal to 1 through the int8 matrix cores with four / three digits (k_bitplane.hip;
    # the automatic rule only takes that route from 4096 cells on: forced here); "bitplane_mx": the same products on the MX matrix
    # instruction (FP4 bitmap x six base-31 digits in FP6, option bp_format=mx6)
    from doubletdetection_amd import _lib

    if gather.startswith("bitplane"):
        monkeypatch.setitem(_lib.OPTIONS, "bitplane", "2")
        monkeypatch.setitem(_lib.OPTIONS, "bp_digits", "3" if gather.endswith("3") else "4")
        if gather.endswith("_mx"):
            monkeypatch.setitem(_lib.OPTIONS, "bp_format", "mx6")
    else:
        monkeypatch.setitem(_lib.OPTIONS, "pca_gather", gather)
    with _lib.Context(0) as ctx:
        _pca_scores_body(ctx, case, gather)


def _pca_scores_body(ctx, case, gather):
    # bar: 1e-4 (sklearn f32 vs f64 differs by up to 8e-4); three digits on these 500-cell matrices: 2e-5 (4e-6 at the BASELINE sizes)
    tol = {"f64": 1e-7, "bitplane3": 5e-5}.get(gather, 1e-5)
    g = load_golden(case)
    kw = golden_kwargs(g)
    raw = csr_from(g, "raw_hvg")
    seed = kw.get("random_state", 0)
    C = g["pca_f32"].shape[2]
    ctx.upload_counts(raw)
    ctx.create_doublets(g["parents"][0])
    ctx.lognormalise(kw.get("pseudocount", 0.1))
    if kw.get("standard_scaling"):
        ctx.scale(15.0)
    M, H = ctx.M, ctx.H
    q0 = orc.pca_start_matrix(seed, H if M >= H else M, C + 10, round_f32=True)
    ctx.pca(C, q0)
    emb64, sing = ctx.embedding_f64()
    emb32 = ctx.embedding()
    np.testing.assert_array_equal(emb32, emb64.astype(np.float32))
    # (1) against the float64 oracle evaluated on the matrix the GPU actually holds
    X = ctx.aug_dense_rows(0, M)
    want, s_want, _ = orc.randomized_pca_f64(X, C, seed, round_q0_f32=True)
    dev = orc.per_component_rel_dev(emb64, want)
    assert dev.max() < tol, dev
    np.testing.assert_allclose(sing, s_want, rtol=1e-9 if gather == "f64" else (1e-6 if gather == "bitplane3" else 1e-7))
    # (2) H3 acceptance band against scikit-learn itself: <= 1e-4 of the float64 run, and no further
    #     from the float32 run the reference performs than sklearn-f64 is
    ref64 = g["pca_it0_sklearn_f64"]
    ref32 = g["pca_f32"][0]
    d64 = orc.per_component_rel_dev(emb64, ref64)
    assert d64.max() < 1e-4, d64
    d32 = orc.per_component_rel_dev(emb64, ref32)
    dref = orc.per_component_rel_dev(ref32, ref64)
    assert d32.max() <= 2.0 * dref.max() + 1e-6, (d32.max(), dref.max())


# ---- a10/a11: exact kNN --------------------------------------------------------------------------------
@pytest.mark.parametrize("case", CASES)
def test_knn_exact(ctx, case):
    g = load_golden(case)
    emb = g["pca_f32"][0]
    ctx.set_embedding(emb)
    ctx.knn(10, True)
    idx, dist = ctx.get_knn()
    wi, wd = orc.knn_bruteforce_f64(emb, 10, include_self=True)
    np.testing.assert_array_equal(idx, wi)                    # bit-identical ordering incl. ties
    np.testing.assert_array_equal(dist, wd)
    assert (idx == g["knn10_brute_self"]).all(axis=1).mean() >= 0.999      # sklearn brute (scanpy path)
    if g["knn30_kdtree"].size:
        ctx.knn(30, False)
        idx, dist = ctx.get_knn()
        wi, wd = orc.knn_bruteforce_f64(emb, 30, include_self=False)
        np.testing.assert_array_equal(idx, wi)
        np.testing.assert_array_equal(dist, wd)
        assert (idx == g["knn30_kdtree"]).all(axis=1).mean() >= 0.99       # sklearn kd_tree (phenograph path)


def test_knn_duplicates_and_wide_embedding(ctx):
    rng = np.random.default_rng(1)
    emb = rng.normal(size=(700, 40)).astype(np.float32)
    emb[100:130] = emb[5]                                      # 31 identical points: ties broken by index
    ctx.set_embedding(emb)
    for k, self_ in ((10, True), (30, False), (64, False)):
        ctx.knn(k, self_)
        idx, dist = ctx.get_knn()
        wi, wd = orc.knn_bruteforce_f64(emb, k, include_self=self_)
        np.testing.assert_array_equal(idx, wi)
        np.testing.assert_array_equal(dist, wd)


# ---- graphs handed to community detection -----------------------------------------------------------
@pytest.mark.parametrize("case", ["case_a_hvg_pheno", "case_c_reftest_scaled"])
def test_graphs(ctx, case):
    g = load_golden(case)
    emb = g["pca_f32"][0]
    ctx.set_embedding(emb)
    ctx.knn(30, False)
    idx, _ = ctx.get_knn(with_dist=False)
    for mode, prune in ((0, True), (1, False)):
        ip, ix, w = ctx.build_graph(mode)
        G = orc.jaccard_graph(idx.astype(np.int64), prune=prune)
        np.testing.assert_array_equal(ip, G.indptr)
        np.testing.assert_array_equal(ix, G.indices)
        np.testing.assert_array_equal(w, G.data)
    ctx.knn(10, True)
    idx, _ = ctx.get_knn(with_dist=False)
    ip, ix, w = ctx.build_graph(2)
    G = orc.union_knn_graph(idx.astype(np.int64))
    np.testing.assert_array_equal(ip, G.indptr)
    np.testing.assert_array_equal(ix, G.indices)
    np.testing.assert_array_equal(w, G.data)
    # umap connectivities (what leiden clusters on): same topology, weights to the last bits of exp()
    idx, dist = ctx.get_knn()
    ip3, ix3, w3 = ctx.build_graph(3)
    U = orc.umap_connectivities(idx.astype(np.int64), dist)
    np.testing.assert_array_equal(ip3, U.indptr)
    np.testing.assert_array_equal(ix3, U.indices)
    np.testing.assert_allclose(w3, U.data, rtol=1e-12, atol=0)
    np.testing.assert_array_equal(ip3, ip)


@pytest.mark.parametrize("case", ["case_a_hvg_pheno", "case_b_transposed_louvain"])
def test_device_presweeps_match_host(ctx, case):
    """Part A of the community detection on the device (ddx_coarsen_graph) against the host statement and the
    Python specification: member table and aggregated graph bit for bit, for every graph type and both gammas."""
    from doubletdetection_amd import _lib

    g = load_golden(case)
    ctx.set_embedding(g["pca_f32"][0])
    for k, self_, mode, gamma in ((30, False, 0, 1.0), (30, False, 1, 1.0), (10, True, 2, 4.0), (10, True, 2, 1.0)):
        ctx.knn(k, self_)
        ip, ix, w = ctx.build_graph(mode)
        for sweeps in (1, _lib.PRESWEEPS):
            m_dev, ip_dev, ix_dev, w_dev = ctx.coarsen_graph(gamma, sweeps, levels=1)
            m_host, ip_host, ix_host, w_host = _lib.presweep(ip, ix, w, gamma, sweeps)
            np.testing.assert_array_equal(m_dev, m_host)
            np.testing.assert_array_equal(ip_dev, ip_host)
            np.testing.assert_array_equal(ix_dev, ix_host)
            np.testing.assert_array_equal(w_dev, w_host)
        m_ref, ip_ref, ix_ref, w_ref = louvain_ref.presweep(ip, ix, w, gamma)
        np.testing.assert_array_equal(m_dev, m_ref)
        np.testing.assert_array_equal(w_dev, w_ref)
        # several levels on the device = the host statement applied repeatedly
        for levels in (2, 3):
            m_dev, ip_dev, ix_dev, w_dev = ctx.coarsen_graph(gamma, levels=levels)
            total, gr = None, (ip, ix, w)
            for _ in range(levels):
                mm, *gr = _lib.presweep(*gr, gamma)
                total = mm if total is None else mm[total]
            np.testing.assert_array_equal(m_dev, total)
            np.testing.assert_array_equal(ip_dev, gr[0])
            np.testing.assert_array_equal(ix_dev, gr[1])
            np.testing.assert_array_equal(w_dev, gr[2])
        # device A + host B + device C = host A + B + C, for the Louvain levels, PhenoGraph's restart rule and Leiden
        m_dev, ip_dev, ix_dev, w_dev = ctx.coarsen_graph(gamma)
        coarse = _lib.louvain_sequential(ip_dev, ix_dev, w_dev, gamma, 3)[0]
        np.testing.assert_array_equal(ctx.refine_communities(coarse, gamma), _lib.louvain(ip, ix, w, gamma, 3)[0])
        coarse = _lib.louvain_best_of(ip_dev, ix_dev, w_dev, gamma, 3, 1e-3, threads=4, presweeps=False)[0]
        np.testing.assert_array_equal(ctx.refine_communities(coarse, gamma), _lib.louvain_best_of(ip, ix, w, gamma, 3, 1e-3, threads=4)[0])
        coarse = _lib.leiden_sequential(ip_dev, ix_dev, w_dev, gamma, 3)
        np.testing.assert_array_equal(ctx.refine_communities(coarse, gamma), _lib.leiden(ip, ix, w, gamma, 3))
        # part C alone: any labelling of the coarse nodes, any number of sweeps; one level of part A
        rng = np.random.default_rng(5)
        arbitrary = rng.integers(0, max(2, len(ip_dev) // 7), size=len(ip_dev) - 1).astype(np.int32)
        graphs, members = _lib.presweep_levels(ip, ix, w, gamma)
        for sweeps in (0, 1, _lib.REFINE_SWEEPS):
            np.testing.assert_array_equal(ctx.refine_communities(arbitrary, gamma, sweeps), _lib.refine_down(graphs, members, arbitrary, gamma, sweeps))
        m1, ip1, ix1, w1 = ctx.coarsen_graph(gamma, levels=1)
        coarse = _lib.louvain_sequential(ip1, ix1, w1, gamma, 9)[0]
        np.testing.assert_array_equal(ctx.refine_communities(coarse, gamma), _lib.refine(ip, ix, w, coarse[m1], gamma))


def test_device_presweeps_high_degree_nodes(ctx):
    """Nodes with more than 64 neighbours take the LDS path of the sweep kernel; a star-heavy graph exercises it."""
    from doubletdetection_amd import _lib

    rng = np.random.default_rng(11)
    n = 3000
    emb = rng.normal(size=(n, 8)).astype(np.float32)
    emb[:40] *= 0.02                                          # a tight core that ends up in everybody's neighbour list
    ctx.set_embedding(emb)
    ctx.knn(30, False)
    ip, ix, w = ctx.build_graph(1)
    assert np.diff(ip).max() > 64
    m_dev, ip_dev, ix_dev, w_dev = ctx.coarsen_graph(1.0, levels=1)
    m_host, ip_host, ix_host, w_host = _lib.presweep(ip, ix, w, 1.0)
    np.testing.assert_array_equal(m_dev, m_host)
    np.testing.assert_array_equal(ip_dev, ip_host)
    np.testing.assert_array_equal(ix_dev, ix_host)
    np.testing.assert_array_equal(w_dev, w_host)
    # ... and part C on the same graph (the hubs go through the LDS path there too)
    coarse = _lib.louvain_sequential(ip_dev, ix_dev, w_dev, 1.0, 0)[0]
    np.testing.assert_array_equal(ctx.refine_communities(coarse, 1.0), _lib.refine(ip, ix, w, coarse[m_dev], 1.0))


def test_device_presweeps_refuse_hubs_beyond_capacity_and_fit_falls_back(ctx):
    """A node with more neighbours than the sweep kernel's LDS table (4096) makes ddx_coarsen_graph return
    DDX_E_UNSUPPORTED; BoostClassifier then fetches the graph and runs the whole specification on the host."""
    from doubletdetection_amd import BoostClassifier, _lib

    rng = np.random.default_rng(12)
    n = 8000
    emb = rng.normal(size=(n, 32))
    emb = (10.0 * emb / np.linalg.norm(emb, axis=1, keepdims=True)).astype(np.float32)   # a sphere: neighbours far apart
    emb[:35] = (0.01 * rng.normal(size=(35, 32))).astype(np.float32)                     # ... except the core at its centre
    ctx.set_embedding(emb)
    ctx.knn(30, False)
    ip, ix, w = ctx.build_graph(1)
    assert np.diff(ip).max() > 4096
    with pytest.raises(_lib.DdxError) as err:
        ctx.coarsen_graph(1.0)
    assert err.value.code == _lib.E_UNSUPPORTED
    # the classifier's fallback: same labels as the host specification on the fetched graph
    from doubletdetection_amd.classifier import _HipEngine
    eng = _HipEngine.__new__(_HipEngine)
    eng.ctx = ctx
    graph = None
    ctx.build_graph(1, fetch=False)
    try:
        graph = ctx.coarsen_graph(1.0)
    except _lib.DdxError:
        graph = ctx.fetch_graph()
    full, _, _, _ = BoostClassifier._cluster_and_score(graph, 1.0, 0, 10, n - 100)
    lab = _lib.louvain(ip, ix, w, 1.0, 0)[0]
    np.testing.assert_array_equal(full, _lib.relabel_by_size(lab, 10))


# ---- whole fit -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("route", ["auto", "bitplane"])
@pytest.mark.parametrize("case", ["case_a_hvg_pheno", "case_b_transposed_louvain", "case_c_reftest_scaled",
                                  "case_d_replace_single"])
def test_fit_matches_oracle_and_reference_run(case, route, monkeypatch):
    from doubletdetection_amd import BoostClassifier, _lib

    if route == "bitplane":                       # the route the BASELINE sizes take, forced onto the golden matrices
        monkeypatch.setitem(_lib.OPTIONS, "bitplane", "2")
    g = load_golden(case)
    kw = golden_kwargs(g)
    counts = csr_from(g, "counts")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = BoostClassifier(**kw).fit(counts)
        orc_kw = {k: v for k, v in kw.items()}
        ocl = orc.OracleClassifier(pca="f64", **orc_kw).fit(counts)
    np.testing.assert_array_equal(np.asarray(clf.parents_, dtype=np.int64), g["parents"])
    # integer work: identical community assignments to the float64 oracle
    np.testing.assert_array_equal(clf.communities_, ocl.communities_)
    np.testing.assert_array_equal(clf.synth_communities_, ocl.synth_communities_)
    np.testing.assert_array_equal(clf.all_scores_, ocl.all_scores_)
    np.testing.assert_allclose(clf.all_log_p_values_, ocl.all_log_p_values_, rtol=1e-9, atol=1e-9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_array_equal(clf.predict(), ocl.predict())
    # and against the reference's own run (float32 sklearn PCA inside): labels agree for nearly all cells
    agree = np.mean(clf.communities_ == g["communities"])
    assert agree > 0.95, agree


def test_reference_test_suite_scenario():
    """tests/test_package.py of the reference, on the GPU path: runs for all three algorithm names,
    and two classifiers with the same seed give identical scores (test_package.py:25-38)."""
    from doubletdetection_amd import BoostClassifier

    counts = np.random.default_rng(42).poisson(1.0, size=(500, 100))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for algo in ("louvain", "phenograph"):
            clf = BoostClassifier(n_iters=2, clustering_algorithm=algo, standard_scaling=True)
            clf.fit(counts).predict(p_thresh=1e-16, voter_thresh=0.5)
            clf.doublet_score()
        s = []
        for _ in range(2):
            clf = BoostClassifier(n_iters=2, clustering_algorithm="leiden", standard_scaling=True, random_state=123)
            clf.fit(counts).predict(p_thresh=1e-16, voter_thresh=0.5)
            s.append(clf.doublet_score())
    np.testing.assert_equal(s[0], s[1])
    assert clf.all_log_p_values_.shape == (2, 500) and clf.communities_.shape == (2, 500)
    assert clf.synth_communities_.shape == (2, 125)


# ---- API corners on the GPU path ---------------------------------------------------------------------------
@pytest.mark.parametrize("n_components,algo", [(45, "louvain"), (12, "leiden"), (54, "phenograph"), (50, "louvain")])
def test_other_sketch_widths_and_leiden(n_components, algo):
    """Sketch widths other than the default 40: > 42 columns take the quad geometry of the LDS-staged products (four
    columns per lane, up to the limit of 64), <= 32 the 4-group pair geometry; and the leiden name
    end to end (umap connectivities, Leiden refinement)."""
    from doubletdetection_amd import BoostClassifier
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(900, 700, density=0.15, n_types=5, seed=21)
    kw = dict(n_iters=2, n_top_var_genes=600, n_components=n_components, clustering_algorithm=algo, random_state=2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = BoostClassifier(**kw).fit(counts)
        ref = orc.OracleClassifier(pca="f64", **kw).fit(counts)
    np.testing.assert_array_equal(clf.communities_, ref.communities_)
    np.testing.assert_array_equal(clf.all_scores_, ref.all_scores_)
    np.testing.assert_allclose(clf.all_log_p_values_, ref.all_log_p_values_, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("n_components,k,algo", [(100, 100, "phenograph"), (60, 200, "phenograph"), (128, 30, "phenograph"), (70, None, "louvain")])
def test_wide_sketches_many_dimensions_many_neighbours(n_components, k, algo):
    """What the reference accepts and round 2 refused (dd.py:108-112, 320-322): n_components beyond 54 (sketches wider than
    64 columns run block by block, the orthonormalisations tiled), embeddings of up to 128 dimensions and up to 256
    neighbours in the kNN kernels (the bound pass split over several launches, longer candidate lists) -- stage by stage
    against the float64 oracle, then the whole fit."""
    from doubletdetection_amd import BoostClassifier, _lib
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(2400, 900, density=0.2, n_types=6, seed=33)
    ckw = {} if k is None else {"k": k}
    kw = dict(n_iters=2, n_top_var_genes=800, n_components=n_components, clustering_algorithm=algo, clustering_kwargs=dict(ckw),
              random_state=4)
    with _lib.Context(0) as ctx:
        _, sub = orc.select_hvg(sp.csr_matrix(counts, dtype=np.float32), 800)
        ctx.upload_counts(sub)
        parents = np.random.default_rng(4).choice(2400, size=(600, 2), replace=False)
        ctx.create_doublets(parents)
        ctx.lognormalise(0.1)
        M, H = ctx.M, ctx.H
        assert orc.sklearn_solver_policy(M, H, n_components) == "randomized"
        ctx.pca(n_components, orc.pca_start_matrix(4, H, n_components + 10))
        emb64, sing = ctx.embedding_f64()
        dense = ctx.aug_dense_rows(0, M)
        want, s_want, _ = orc.randomized_pca_f64(dense, n_components, 4)
        rel = orc.per_component_rel_dev(emb64, want)
        # (the default sketch stays below 1e-5, test_pca_scores; 128 components of an 800-gene matrix reach deep into the
        # noise floor, where neighbouring singular values differ by 1e-3 and every rounding of the float32 operand copy is
        # amplified accordingly: the bar here is the north star's 1e-4)
        print(f"n_components={n_components}: max relative deviation per component {rel.max():.2e}")
        assert rel.max() <= (1e-5 if n_components <= 100 else 1e-4), rel
        np.testing.assert_allclose(sing, s_want, rtol=1e-6)
        kk = 30 if k is None else k
        ctx.knn(kk, False)
        idx, dist = ctx.get_knn()
        ref_idx, ref_dist = orc.knn_bruteforce_f64(ctx.embedding(), kk, False)
        np.testing.assert_array_equal(idx, ref_idx)
        np.testing.assert_array_equal(dist, ref_dist)
        G = orc.jaccard_graph(ref_idx, prune=True)
        ip, ix, w = ctx.build_graph(0)
        np.testing.assert_array_equal(ip, G.indptr)
        np.testing.assert_array_equal(ix, G.indices)
        np.testing.assert_allclose(w, G.data, rtol=1e-12)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = BoostClassifier(**kw).fit(counts)
        kw["clustering_kwargs"] = dict(ckw)
        ref = orc.OracleClassifier(pca="f64", **kw).fit(counts)
    np.testing.assert_array_equal(clf.communities_, ref.communities_)
    np.testing.assert_array_equal(clf.all_scores_, ref.all_scores_)
    np.testing.assert_allclose(clf.all_log_p_values_, ref.all_log_p_values_, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("metric", ["manhattan", "cosine", "correlation"])
def test_other_knn_metrics_against_sklearn(ctx, metric):
    """phenograph.cluster(primary_metric=...) (dd.py:320-322): the exact float64 scan on the device against scikit-learn's
    NearestNeighbors with the same metric (what upstream calls), on the reference-generated embedding of a golden case and on
    a larger random one; then a whole fit against the oracle, whose kNN is sklearn's."""
    from doubletdetection_amd import BoostClassifier
    from doubletdetection_amd._synthetic import make_counts

    g = load_golden("case_a_hvg_pheno")
    for emb, k in ((g["pca_f32"][0], 30), (np.random.default_rng(3).normal(size=(5000, 20)).astype(np.float32), 45)):
        ctx.set_embedding(emb)
        ctx.knn(k, False, metric)
        idx, dist = ctx.get_knn()
        ref_idx, ref_dist = orc.knn_metric(emb, k, False, metric)
        same = np.mean(np.all(idx == ref_idx, axis=1))
        assert same >= 0.99, same                                   # (ties / last-bit differences of the dot products)
        np.testing.assert_allclose(dist, ref_dist, rtol=1e-9, atol=1e-12)
        assert np.all(idx != np.arange(emb.shape[0])[:, None])
    counts = make_counts(900, 700, density=0.15, n_types=5, seed=21)
    kw = dict(n_iters=2, n_top_var_genes=600, clustering_kwargs={"primary_metric": metric}, random_state=2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = BoostClassifier(**kw).fit(counts)
        kw["clustering_kwargs"] = {"primary_metric": metric}
        ref = orc.OracleClassifier(pca="f64", **kw).fit(counts)
    agree = np.mean(clf.communities_ == ref.communities_)
    assert agree >= 0.98, agree


def test_api_corners_on_gpu(capsys):
    from doubletdetection_amd import BoostClassifier

    rng = np.random.default_rng(5)
    dense = rng.poisson(0.8, size=(700, 150)).astype(np.int64)        # dense ndarray input is sparsified (dd.py:157-160)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # single iteration: score-gap predict, bool labels, suggested cutoff (dd.py:243-252)
        one = BoostClassifier(n_iters=1, clustering_kwargs={"prune": False}, verbose=True).fit(dense)
        lab = one.predict()
        assert lab.dtype == bool and hasattr(one, "suggested_score_cutoff_")
        assert one.doublet_score().shape == (700,)
        out = capsys.readouterr().out
        assert "Sparsifying matrix." in out and "Iteration   1/1" in out and "Found clusters" in out
        # replace=True with boost_rate > 0.5, odd sketch width (n_components=25 -> 35 columns), louvain + scaling
        many = BoostClassifier(n_iters=2, replace=True, boost_rate=0.8, n_components=25, clustering_algorithm="louvain",
                               standard_scaling=True, random_state=3).fit(sp.csr_matrix(dense))
        ref = orc.OracleClassifier(n_iters=2, replace=True, boost_rate=0.8, n_components=25, clustering_algorithm="louvain",
                                   standard_scaling=True, random_state=3, pca="f64").fit(sp.csr_matrix(dense))
    assert many.synth_communities_.shape == (2, 560)
    np.testing.assert_array_equal(np.asarray(many.parents_), np.asarray(ref.parents_))
    np.testing.assert_array_equal(many.communities_, ref.communities_)
    np.testing.assert_array_equal(many.all_scores_, ref.all_scores_)
    # sparse input in another format / dtype is coerced like check_array does
    again = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        again = BoostClassifier(n_iters=2, replace=True, boost_rate=0.8, n_components=25, clustering_algorithm="louvain",
                                standard_scaling=True, random_state=3).fit(sp.csc_matrix(dense.astype(np.float64)))
    np.testing.assert_array_equal(again.all_log_p_values_, many.all_log_p_values_)


@pytest.mark.parametrize("shape,regime", [((5000, 60), "covariance_eigh"), ((300, 100), "full"), ((120, 400), "full")])
def test_exact_pca_regimes(shape, regime):
    """Small inputs for which PCA(svd_solver="auto") takes an exact solver: scores vs sklearn in float64,
    and the whole fit vs the oracle."""
    from doubletdetection_amd import BoostClassifier

    counts = np.random.default_rng(9).poisson(1.0, size=shape)
    n_comp = 30 if min(shape) > 40 else 20
    kw = dict(n_iters=2, clustering_algorithm="louvain", n_components=n_comp, random_state=4)
    S = int(0.25 * shape[0])
    assert orc.sklearn_solver_policy(shape[0] + S, shape[1], n_comp) == regime
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = BoostClassifier(**kw).fit(counts)
        ref = orc.OracleClassifier(pca="f64", **kw).fit(counts)
        skl = orc.OracleClassifier(pca="sklearn", **kw).fit(counts)     # what the reference would run (float32)
    np.testing.assert_array_equal(clf.communities_, ref.communities_)
    np.testing.assert_array_equal(clf.all_scores_, ref.all_scores_)
    # the oracle's float64 exact PCA against sklearn's own float32 run: singular values (= column norms of
    # U S); individual components of pure-noise data are near-degenerate and not comparable one by one
    sv_ref = np.linalg.norm(ref.embeddings_[0].astype(np.float64), axis=0)
    sv_skl = np.linalg.norm(skl.embeddings_[0].astype(np.float64), axis=0)
    np.testing.assert_allclose(sv_ref, sv_skl, rtol=1e-4)


def test_sparse_arpack_path_pseudocount_one(ctx):
    """pseudocount=1 keeps the matrix sparse upstream and PCA runs ARPACK (dd.py:296-297,308): golden case E."""
    from doubletdetection_amd import BoostClassifier

    g = load_golden("case_e_pc1_sparse")
    kw = golden_kwargs(g)
    raw = csr_from(g, "raw_hvg")
    ctx.upload_counts(raw)
    ctx.create_doublets(g["parents"][0])
    ctx.lognormalise(1.0)
    vals, z = ctx.aug_values()
    want = csr_from(g, "pca_in0")                      # the reference's own log1p(CSR) matrix
    assert np.all(z == 0)
    assert _ulp_diff(vals, want.data).max() <= 4       # numpy log1p vs correctly rounded
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = BoostClassifier(**kw).fit(csr_from(g, "counts"))
        ref = orc.OracleClassifier(pca="f64", **kw).fit(csr_from(g, "counts"))
    # ARPACK converges to the exact truncated SVD whatever drives it: embeddings agree to solver tolerance
    np.testing.assert_array_equal(np.asarray(clf.parents_), g["parents"])
    agree = np.mean(clf.communities_ == ref.communities_)
    assert agree > 0.98, agree
    agree_ref = np.mean(clf.communities_ == g["communities"])
    assert agree_ref > 0.95, agree_ref


def test_sparse_path_on_few_genes_is_exact():
    """pseudocount=1 with 100 - 200 highly variable genes: a Krylov space of 40-column blocks has no room to converge in so few
    dimensions (round 4 returned whatever one or two steps gave, silently); the host now takes the exact decomposition of the
    small Gram matrix there -- ARPACK upstream is exact at every size -- and the fit equals the float64 oracle."""
    from doubletdetection_amd import BoostClassifier
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(1200, 600, density=0.2, n_types=5, seed=21)
    for ntop in (100, 160, 320):
        kw = dict(n_top_var_genes=ntop, pseudocount=1.0, n_iters=2, random_state=3, clustering_algorithm="louvain")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            clf = BoostClassifier(**kw).fit(counts)
            ref = orc.OracleClassifier(pca="f64", **kw).fit(counts)
        np.testing.assert_array_equal(np.asarray(clf.parents_), np.asarray(ref.parents_))
        agree = np.mean(clf.communities_ == ref.communities_)
        assert agree > 0.98, (ntop, agree)
        np.testing.assert_allclose(clf.all_scores_[clf.communities_ == ref.communities_], ref.all_scores_[clf.communities_ == ref.communities_], rtol=0.05, atol=0.02)


def test_block_lanczos_reports_an_unconverged_solve(ctx):
    """ddx_pca_exact_sparse with too few steps for its tolerance returns DDX_W_UNCONVERGED (the best Ritz pairs in place): the caller
    is told instead of handed an unconverged embedding as if it were the answer."""
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(3000, 1500, density=0.1, n_types=5, seed=5)
    ctx.upload_counts(counts)
    ctx.create_doublets(np.random.default_rng(0).choice(3000, size=(750, 2), replace=False))
    ctx.lognormalise(1.0)
    start = np.random.RandomState(0).normal(size=(1500, 40))
    steps = ctx.pca_exact_sparse(30, start, tol=1e-12, max_steps=3)
    assert steps == 3 and ctx.lanczos_converged is False
    steps = ctx.pca_exact_sparse(30, start, tol=1e-6, max_steps=36)
    assert ctx.lanczos_converged is True and steps < 36
    emb, _ = ctx.embedding_f64()
    X = ctx.aug_dense_rows(0, ctx.M)
    want, _, _ = orc.exact_pca_f64(X, 30)
    dev = orc.per_component_rel_dev(emb, want)
    assert dev.max() < 1e-4, dev


def test_bitplane_route_corner_shapes(monkeypatch):
    """The bit-plane route (forced) where its bookkeeping has corners: no synthetic rows at all, a number of cells and genes that is no
    multiple of the 32-row tiles / 256-column stages, more synthetic rows than half the cells (the route steps aside), and a second
    iteration with a different number of synthetic rows on the same context.  PCA against the float64 oracle every time."""
    from doubletdetection_amd import _lib
    from doubletdetection_amd._synthetic import make_counts

    monkeypatch.setitem(_lib.OPTIONS, "bitplane", "2")
    counts = make_counts(1111, 777, density=0.15, n_types=4, seed=9)
    rng = np.random.default_rng(5)
    with _lib.Context(0) as c:
        c.upload_counts(counts)
        for S in (0, 277, 555, 700, 33):
            parents = rng.choice(1111, size=(S, 2), replace=True)
            c.create_doublets(parents)
            c.lognormalise(0.1)
            M, H = c.M, c.H
            q0 = orc.pca_start_matrix(0, H if M >= H else M, 40)
            c.pca(30, q0)
            assert c.bitplane_stats()["active"] == (S <= 1111 // 2)
            emb, sing = c.embedding_f64()
            want, s_want, _ = orc.randomized_pca_f64(c.aug_dense_rows(0, M), 30, 0)
            dev = orc.per_component_rel_dev(emb, want)
            assert dev.max() < 1e-5, (S, dev.max())
            np.testing.assert_allclose(sing, s_want, rtol=1e-7)
            # the full column-major mirror was left out; whoever asks gets it: A^T x through the plain kernels = through the dense matrix
            x = rng.normal(size=(M, 3))
            got = c.operator_apply(x, 1)
            A = c.aug_dense_rows(0, M).astype(np.float64)
            A -= A.mean(axis=0)
            np.testing.assert_allclose(got, A.T @ x, rtol=1e-6, atol=1e-6)


def test_doublets_derived_from_their_parents_structures():
    """Bit-plane route: a doublet's bitmap row, reduced entries and library size are derived from its parents' structures (k_bp_synth)
    instead of from the merged row (option synthetic=merged: k_doublet_fill + flags / scan / compaction).  Both constructions must give
    the SAME structures, hence bit-identical PCA scores -- on counts with explicit stored zeros, counts beyond the (row, count) table,
    overlaps of every kind (1 + 1, 1 + c, c + c') and a doublet of a cell with itself; and whoever asks for the merged rows, their values
    or dense rows afterwards gets exactly the oracle's (built on demand: ensure_full_rows)."""
    from doubletdetection_amd import _lib
    from doubletdetection_amd._synthetic import make_counts

    rng = np.random.default_rng(17)
    counts = make_counts(3000, 1100, density=0.12, n_types=5, seed=4).tocsr().astype(np.float32)
    counts.data[rng.random(counts.nnz) < 0.02] = 0.0                       # explicit zeros stay stored
    big = rng.random(counts.nnz) < 0.01
    counts.data[big] = rng.integers(17, 400, size=int(big.sum())).astype(np.float32)
    N, H = counts.shape
    S = 900
    parents = rng.choice(N, size=(S, 2), replace=True)
    parents[0] = (5, 5)
    parents[1] = (parents[2][1], parents[2][0])
    results = {}
    for how in ("derived", "merged"):
        with _lib.Context(0) as c:
            c.set_option("bitplane", "2")
            c.set_option("synthetic", how)
            c.timing_enable(True)
            c.upload_counts(counts)
            for it in range(2):                                              # (a second iteration on the same context: other parents)
                par = parents if it == 0 else parents[::-1].copy()
                c.create_doublets(par)
                c.lognormalise(0.1)
                c.pca(30, orc.pca_start_matrix(0, H, 40))
                stats = c.bitplane_stats()
                assert stats["active"]
                emb, sing = c.embedding_f64()
                results[(how, it)] = (emb.copy(), sing.copy(), stats["rest_synthetic"], c.aug_lib())
            # the derived construction never wrote a merged row; the first consumer of the rows -- here a plain float64 product A x
            # (ddx_operator_apply, the exact-PCA regimes' entry point) -- makes it do so, once
            assert c.timings().get("doublet_fill", (0, 0.0))[0] == (0 if how == "derived" else 2)
            xop = np.random.default_rng(23).normal(size=(H, 3))
            ax = c.operator_apply(xop, 0)
            assert c.timings()["doublet_fill"][0] == (1 if how == "derived" else 2)
            A = c.aug_dense_rows(0, N + S).astype(np.float64)
            np.testing.assert_allclose(ax, (A - A.mean(axis=0)) @ xop, rtol=1e-6, atol=1e-6)
            # read-backs after the lean iteration: the merged rows, this iteration's values, dense rows
            syn = c.get_synth()
            assert c.timings()["doublet_fill"][0] == (1 if how == "derived" else 2)
            want = orc.create_doublets(counts, parents[::-1])
            want.sort_indices()
            np.testing.assert_array_equal(syn.indptr, want.indptr)
            np.testing.assert_array_equal(syn.indices, want.indices)
            np.testing.assert_array_equal(syn.data, want.data)
            assert c.aug_nnz() == counts.nnz + want.nnz
            dense = c.aug_dense_rows(N, S)
            ref, ref_lib, ref_med = orc.lognormalise(orc.l1_normalise_rows(counts), orc.library_sizes(counts), want, 0.1)
            assert _ulp_diff(dense, np.asarray(ref)[N:]).max() <= 4                             # (numpy's float32 log: see test_lognormalised_matrix)
            lib, med = c.aug_lib()
            np.testing.assert_array_equal(lib, ref_lib.astype(np.float32))
            assert med == np.float32(ref_med)
            results[(how, "dense")] = dense
    np.testing.assert_array_equal(results[("derived", "dense")], results[("merged", "dense")])
    for it in range(2):
        d, m = results[("derived", it)], results[("merged", it)]
        assert d[2] == m[2] and d[2] > 0
        np.testing.assert_array_equal(d[3][0], m[3][0])
        assert d[3][1] == m[3][1]
        np.testing.assert_array_equal(d[1], m[1])
        np.testing.assert_array_equal(d[0], m[0])


def test_a_failing_stage_inside_the_pca_surfaces_as_an_error():
    """Error path (fault injection, option fault=1: every request for a larger dynamic-LDS limit is refused): the first operator
    product of ddx_pca cannot be launched -- the call must return the HIP error, not DDX_OK on an untouched iterate."""
    from doubletdetection_amd import _lib
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(2000, 900, density=0.1, n_types=4, seed=2)
    with _lib.Context(0) as c:
        c.upload_counts(counts)
        c.create_doublets(np.random.default_rng(1).choice(2000, size=(500, 2), replace=False))
        c.lognormalise(0.1)
        q0 = orc.pca_start_matrix(0, 900, 40)
        c.set_option("testing", "1")             # (fault injection is locked without it)
        c.set_option("fault", "1")
        with pytest.raises(_lib.DdxError) as err:
            c.pca(30, q0)
        assert err.value.code == _lib.E_HIP and "fault injection" in str(err.value)
        c.set_option("fault", "0")
        c.pca(30, q0)                                   # and the context is usable again
        emb, _ = c.embedding_f64()
        assert np.isfinite(emb).all()


def test_fit_with_empty_cells_and_empty_genes():
    """Ragged input: cells without any count (library size 0: sklearn's row normalisation leaves them untouched, their
    log-normalised row is the constant log(pseudocount)), genes nobody expresses, a cell with a single count.  The
    whole fit equals the float64 oracle; doublets of an empty parent equal the other parent."""
    from doubletdetection_amd import BoostClassifier
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(700, 500, density=0.15, n_types=4, seed=13).tolil()
    for r in (0, 17, 350, 699):
        counts[r, :] = 0
    counts[5, :] = 0
    counts[5, 123] = 2
    counts[:, [3, 250, 499]] = 0
    counts = sp.csr_matrix(counts.tocsr(), dtype=np.float32)
    counts.eliminate_zeros()
    for kw in (dict(n_top_var_genes=400, clustering_algorithm="louvain"), dict(n_top_var_genes=0, standard_scaling=True)):
        kw = dict(n_iters=2, random_state=4, **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            clf = BoostClassifier(**kw).fit(counts)
            ref = orc.OracleClassifier(pca="f64", **kw).fit(counts)
        np.testing.assert_array_equal(np.asarray(clf.parents_), np.asarray(ref.parents_))
        np.testing.assert_array_equal(clf.communities_, ref.communities_)
        np.testing.assert_array_equal(clf.synth_communities_, ref.synth_communities_)
        np.testing.assert_array_equal(clf.all_scores_, ref.all_scores_)
        np.testing.assert_allclose(clf.all_log_p_values_, ref.all_log_p_values_, rtol=1e-9, atol=1e-9)
        assert np.all(np.isfinite(clf.all_scores_[~np.isnan(clf.all_scores_)]))


def test_device_side_input_validation():
    """A float32 CSR is not read by the host at all (check_array would only run its finite check over it): the device
    validates what it receives.  NaN / inf still raise check_array's ValueError (dd.py:149-155); rows with unsorted or
    duplicate columns are canonicalised as scipy's indexing would (dd.py:174-176) and give the canonical matrix's result."""
    from doubletdetection_amd import BoostClassifier
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(800, 500, density=0.15, n_types=4, seed=3)
    kw = dict(n_iters=2, n_top_var_genes=400, clustering_algorithm="louvain", random_state=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = BoostClassifier(**kw).fit(counts)
        for bad_value in (np.nan, np.inf):
            bad = counts.copy()
            bad.data[1234] = bad_value
            with pytest.raises(ValueError, match="NaN|infinity"):
                BoostClassifier(**kw).fit(bad)
        # reverse the column order inside every row and split one entry into two duplicates
        ip = counts.indptr
        idx = np.concatenate([counts.indices[ip[i]:ip[i + 1]][::-1] for i in range(counts.shape[0])]).astype(np.int32)
        dat = np.concatenate([counts.data[ip[i]:ip[i + 1]][::-1] for i in range(counts.shape[0])]).astype(np.float32)
        messy = sp.csr_matrix((dat, idx, ip.copy()), shape=counts.shape)
        assert not messy.has_sorted_indices
        got = BoostClassifier(**kw).fit(messy)
        np.testing.assert_array_equal(got.all_log_p_values_, base.all_log_p_values_)
        np.testing.assert_array_equal(messy.indices, idx)                  # the caller's matrix is left alone
        off = counts.copy()
        off.indices[off.indptr[7] - 1] = 9999                               # last entry of a row: beyond the 500 genes
        with pytest.raises((ValueError, RuntimeError)):
            BoostClassifier(**kw).fit(off)


def test_unsupported_regimes_raise_clearly():
    from doubletdetection_amd import BoostClassifier

    rng = np.random.default_rng(6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(NotImplementedError, match="normalizer"):
            BoostClassifier(n_iters=2, normalizer=lambda x: x).fit(rng.poisson(1.0, size=(600, 100)))


def test_greedy_memory_chunk_falls_back_without_leaving_an_error_behind():
    """The context asks the driver for one large chunk sized from the input; a request the device cannot satisfy falls
    back to the exact size -- and the failed hipMalloc must not stay behind as HIP's sticky last error, which a later
    hipGetLastError() check (end of build_csc, of the PCA, ...) would report as its own (round-2 advice).  ddx_trim
    returns the chunks."""
    from doubletdetection_amd import _lib
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(3000, 900, density=0.1, n_types=4, seed=5)
    with _lib.Context(0) as ctx:
        ctx.reserve_hint(1 << 46)                                 # 64 TB: no device has that
        ctx.upload_raw(counts)
        ctx.select_columns(np.argsort(ctx.gene_variances())[-500:])
        ctx.create_doublets(np.random.default_rng(0).choice(3000, size=(750, 2), replace=False))
        ctx.lognormalise(0.1)
        ctx.pca(20, orc.pca_start_matrix(0, 500, 30))             # ends with DDX_HIP(hipGetLastError())
        ctx.knn(30, False)
        ctx.build_graph(0)
        held = ctx.device_bytes()
        assert 0 < held < (8 << 30)                               # the exact-size fallback, not the greedy guess
        ctx.trim(0)
        assert ctx.device_bytes() == 0
        ctx.reserve_hint(0)
        ctx.upload_raw(counts)                                    # and the context is usable again
        assert ctx.device_bytes() > 0


def test_packed_upload_equals_plain_upload(monkeypatch):
    """dd.py:149-160 (the matrix handed to fit()).  The raw matrix travels packed -- by default 2 bytes per entry (step from the
    row's previous column | count << 8; entries that do not fit that are listed whole beside the codes), with
    upload=packed32 4 bytes (column | count << 16; falls back to plain copies when a count does not fit) -- or plain;
    all three leave the same device matrix, whatever the values."""
    from doubletdetection_amd import _lib
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(60_000, 3000, density=0.02, seed=4)            # > 2^20 stored entries: the packed paths apply
    counts.data[::1001] = 40_000.0                                        # large but representable counts (listed / 4-byte form)
    assert counts.nnz > (1 << 20)

    def device_matrix(env, mat, columns=None):
        for k, v in env.items():
            monkeypatch.setitem(_lib.OPTIONS, k, v)
        c = _lib.Context(0)
        try:
            c.upload_raw(mat)
            var = c.gene_variances()
            # (the 2-byte route folds the gene sums in while the matrix arrives; a second call takes the one-pass route)
            np.testing.assert_array_equal(c.gene_variances(), var)
            c.select_columns(np.sort(np.argsort(var)[-500:]) if columns is None else columns)
            return var, sp.csr_matrix(c.get_counts())
        finally:
            c.close()
            for k in env:
                monkeypatch.delitem(_lib.OPTIONS, k, raising=False)

    # (upload=packed / packed32 wait for the pinned staging buffer; by default the first matrix of a process travels
    # plain while the buffer is being pinned in the background)
    var_q, sub_q = device_matrix({"upload": "plain"}, counts)
    for form in ("packed", "packed32"):
        var_p, sub_p = device_matrix({"upload": form}, counts)
        np.testing.assert_array_equal(var_p, var_q)
        _same_csr(sub_p, sub_q)
    # entries neither code can hold (fractional, >= 65 536, 256), long column steps, empty rows, a row whose only column
    # is the last one: the 2-byte form lists them, the 4-byte form falls back to the plain copies
    odd = counts.copy()
    odd.data[5] = 2.5
    odd.data[77] = 70_000.0
    odd.data[4321] = 255.0
    odd.data[4322] = 256.0
    odd.data[-3] = -0.0                                                   # a stored negative zero keeps its sign bit
    lil = odd[:2000].tolil()
    lil[7, :] = 0
    lil[8, :] = 0
    lil[8, 2999] = 3.0
    odd = sp.vstack([lil.tocsr(), odd[2000:]]).tocsr().astype(np.float32)
    odd.sort_indices()
    assert odd.indptr[8] == odd.indptr[7] and odd.indptr[9] == odd.indptr[8] + 1
    everything = np.arange(odd.shape[1])
    var_g, all_g = device_matrix({"upload": "plain"}, odd, everything)
    for form in ("packed", "packed32"):
        var_f, all_f = device_matrix({"upload": form}, odd, everything)
        np.testing.assert_array_equal(var_f, var_g)
        _same_csr(all_f, all_g)
    # the whole matrix came back: it is the caller's, sign bits included
    _same_csr(all_g, odd)
    assert np.array_equal(all_f.data.view(np.uint32), odd.data.view(np.uint32))
    # a matrix of fractions: every entry would have to be listed, so both packed routes hand over to the plain copies
    halves = counts.copy()
    halves.data *= np.float32(0.5)
    var_h, sub_h = device_matrix({"upload": "plain"}, halves)
    for form in ("packed", "packed32"):
        var_p, sub_p = device_matrix({"upload": form}, halves)
        np.testing.assert_array_equal(var_p, var_h)
        _same_csr(sub_p, sub_h)
    # a matrix the validation rejects is rejected the same way on every route (it arrives as it is)
    broken = counts.copy()
    row = int(np.argmax(np.diff(broken.indptr) > 3))
    p0 = broken.indptr[row]
    broken.indices[p0], broken.indices[p0 + 1] = broken.indices[p0 + 1], broken.indices[p0]
    for form in ("plain", "packed", "packed32"):
        monkeypatch.setitem(_lib.OPTIONS, "upload", form)
        c = _lib.Context(0)
        try:
            with pytest.raises(_lib.DdxError):
                c.upload_raw(broken)
        finally:
            c.close()
            monkeypatch.delitem(_lib.OPTIONS, "upload")


def test_second_column_selection_after_a_pca_matches_a_fresh_context():
    """dd.py:165-184 through the C-ABI: ddx_select_columns may be called again on the same raw matrix after a PCA has run
    (BoostClassifier never does -- every fit uploads -- but the boundary allows it).  The row segments the LDS-staged products
    cache for the original cells belong to the first selection; the second one must not reuse them: same width, other
    columns, embedding compared with a context that only ever saw the second selection."""
    from doubletdetection_amd import _lib
    from doubletdetection_amd._synthetic import make_counts

    N, G, H = 20_000, 4000, 1500                       # large enough for the LDS-staged products (and their row segments)
    counts = make_counts(N, G, density=0.06, seed=21)
    parents = np.random.default_rng(3).choice(N, size=(N // 4, 2), replace=False)
    q0 = np.random.RandomState(0).normal(size=(H, 40)).astype(np.float32).astype(np.float64)

    def embedding(ctx, cols):
        ctx.select_columns(cols)
        ctx.create_doublets(parents)
        ctx.lognormalise(0.1)
        ctx.pca(30, q0)
        return ctx.embedding().copy()

    a = _lib.Context(0)
    b = _lib.Context(0)
    try:
        a.upload_raw(counts)
        order = np.argsort(a.gene_variances())
        cols_a, cols_b = np.sort(order[-H:]), np.sort(order[-2 * H:-H])
        first = embedding(a, cols_a)
        second = embedding(a, cols_b)                  # same context, same width, other columns
        b.upload_raw(counts)
        fresh = embedding(b, cols_b)
        np.testing.assert_array_equal(second, fresh)
        assert not np.array_equal(first, second)
    finally:
        a.close()
        b.close()


def test_contexts_staging_the_same_matrix_share_one_packing(monkeypatch):
    """dd.py:149-160 with several GPUs driven by one process (classifier.py:_stage, one leader context per GPU uploading the same
    host matrix from its own thread): the 2-byte image is packed ONCE; the contexts that arrive while it is being packed attach
    to the job and copy its chunks from the same pinned buffer (ddx_get_upload_form = 2) instead of sending the plain arrays.
    Whatever route a context took, its device matrix is the caller's."""
    import threading

    from doubletdetection_amd import _lib
    from doubletdetection_amd._synthetic import make_counts

    counts = make_counts(120_000, 6000, density=0.03, seed=11)           # 21 M stored entries: a few milliseconds of packing
    counts.data[::977] = 300.0                                            # some entries outside the 2-byte code (listed beside it)
    monkeypatch.setitem(_lib.OPTIONS, "upload", "packed")                 # (waits for the pinned buffer instead of going plain once)
    n_ctx = 4
    ctxs = [_lib.Context(0) for _ in range(n_ctx)]
    try:
        ctxs[0].upload_raw(counts)                                        # pins the staging buffer, grows the arena
        assert ctxs[0].upload_form() == 1
        seen = set()
        for attempt in range(5):
            errors, forms = [], [None] * n_ctx
            gate = threading.Barrier(n_ctx)

            def stage(i):
                try:
                    gate.wait()
                    ctxs[i].upload_raw(counts)
                    forms[i] = ctxs[i].upload_form()
                except Exception as err:                                  # noqa: BLE001
                    errors.append(err)

            threads = [threading.Thread(target=stage, args=(i,)) for i in range(n_ctx)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            assert not errors, errors
            assert sorted(forms).count(1) >= 1, forms                     # somebody packed
            seen.update(forms)
            var0 = ctxs[0].gene_variances()
            cols = np.sort(np.argsort(var0)[-400:])
            ref = None
            for c in ctxs:
                np.testing.assert_array_equal(c.gene_variances(), var0)
                c.select_columns(cols)
                got = sp.csr_matrix(c.get_counts())
                if ref is None:
                    ref = got
                    _same_csr(got, counts[:, cols].tocsr())
                else:
                    _same_csr(got, ref)
            if 2 in seen:
                break
        if 2 not in seen:                                                 # (a host that never ran two of the threads side by side)
            pytest.skip(f"no context ever arrived while another was packing: {seen}")
    finally:
        for c in ctxs:
            c.close()
