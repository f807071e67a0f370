"""standard_scaling=True and sketches wider than 40 columns on the bit-plane route (GPU only; dd.py:302-303, 108-112, 305-314).

Round 6 put both on the int8 matrix-core products.  A scaled entry equal to 1 is s_i / sd_j unless the clip reaches it, so the bitmaps
stay and 1 / sd_j travels in the operand (A Q) / the epilogue (A^T Y); the columns in which such an entry could reach +-max_value are
demoted to the sparse residue with their own clipped values.  What is checked here, at sizes where the route is selected by itself:

* the route is taken (`bitplane_stats()`: active, scaled) and some -- not all -- columns are demoted on a matrix with rare genes;
* the scaled matrix read back from the route's context equals the one of a context that scales the full arrays (option bitplane=0) and
  the CPU restatement of sc.pp.scale;
* the PCA on it is within 1e-5 of the float64 oracle ON THE SCALED MATRIX READ BACK (what the reference would hand to sc.tl.pca);
* a wide sketch (n_components = 50: two 40-column block products) is within 1e-5 of the float64 oracle, on the route;
* whole fits with standard_scaling: seven device contexts == one, and the first iterations against the CPU oracle.
"""
import warnings

import numpy as np
import pytest

from oracle import dd_oracle as orc

pytestmark = pytest.mark.gpu


def _native_louvain(indptr, indices, weights, gamma, seed):
    from doubletdetection_amd import _lib

    return _lib.louvain(indptr, indices, weights, gamma, seed)[0].astype(np.int64)


@pytest.fixture(scope="module")
def counts_rare():
    """6 000 cells x 2 500 genes, all genes kept: the Gamma(0.3) base rates leave hundreds of genes stored in well under 1 % of the
    cells -- the ones whose entries reach the +15 clip."""
    from doubletdetection_amd._synthetic import make_counts

    return make_counts(6000, 2500, density=0.07, n_types=6, seed=808)


def _prepare(counts, options=None, max_value=15.0):
    from doubletdetection_amd import _lib

    ctx = _lib.Context(0)
    ctx.apply_options(options)
    N = counts.shape[0]
    parents = np.random.default_rng(11).choice(N, size=(N // 4, 2), replace=False)
    ctx.upload_counts(counts)
    ctx.create_doublets(parents)
    ctx.lognormalise(0.1)
    if max_value is not None:
        ctx.scale(max_value)
    return ctx


def _dense(ctx, chunk=2048):
    return np.vstack([ctx.aug_dense_rows(r, min(chunk, ctx.M - r)) for r in range(0, ctx.M, chunk)])


@pytest.mark.parametrize("max_value", [15.0, 6.0])
def test_scaled_matrix_on_the_route_equals_the_full_array_scaling(counts_rare, max_value):
    a = _prepare(counts_rare, max_value=max_value)
    b = _prepare(counts_rare, {"bitplane": "0"}, max_value=max_value)
    u = _prepare(counts_rare, {"bitplane": "0"}, max_value=None)
    try:
        st = a.bitplane_stats()
        print(f"max_value {max_value}: {st}")
        assert st["active"] and st["scaled"], st
        assert 0 < st["demoted_columns"] < a.H, st
        assert not b.bitplane_stats()["active"]
        got = _dense(a)                                # (the route scales its reduced structures; the rows follow on demand)
        ref = _dense(b)
        want = orc.scale_like_scanpy(_dense(u), max_value=max_value)
        np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)
        assert got.max() == max_value                  # the clip is reached (rare genes)
        assert np.array_equal(got == max_value, want == max_value)
        # PCA on the route against the float64 oracle on the matrix read back, and against the plain products
        q0 = orc.pca_start_matrix(0, a.H, 40)
        a.pca(30, q0)
        assert a.bitplane_stats()["active"]
        emb_a, sing_a = a.embedding_f64()
        b.pca(30, q0)
        emb_b, sing_b = b.embedding_f64()
        want_emb, s_want, _ = orc.randomized_pca_f64(got, 30, 0)
        dev_a = orc.per_component_rel_dev(emb_a, want_emb).max()
        dev_b = orc.per_component_rel_dev(emb_b, want_emb).max()
        print(f"  PCA vs float64 oracle on the scaled matrix: route {dev_a:.2e}, plain products {dev_b:.2e}")
        assert dev_a <= 1e-5 and dev_b <= 1e-5
        np.testing.assert_allclose(sing_a, s_want, rtol=1e-6)
    finally:
        a.close(); b.close(); u.close()


def test_wide_sketch_runs_in_blocks_on_the_route(counts_rare):
    """n_components = 50 (sketch of 60 columns): two 40-column block products per operator product, on the matrix cores."""
    for max_value in (None, 15.0):
        ctx = _prepare(counts_rare, max_value=max_value)
        try:
            q0 = orc.pca_start_matrix(0, ctx.H, 60)
            ctx.pca(50, q0)
            st = ctx.bitplane_stats()
            assert st["active"] and st["scaled"] == (max_value is not None), st
            emb, sing = ctx.embedding_f64()
            want, s_want, _ = orc.randomized_pca_f64(_dense(ctx), 50, 0)
            dev = orc.per_component_rel_dev(emb, want).max()
            print(f"sketch of 60 columns, scaling {max_value}: {dev:.2e}")
            assert dev <= 1e-5
            np.testing.assert_allclose(sing, s_want, rtol=1e-6)
        finally:
            ctx.close()


def test_scaled_fit_seven_contexts_equal_one_and_match_the_oracle():
    from doubletdetection_amd import BoostClassifier
    from doubletdetection_amd._synthetic import make_counts

    data = make_counts(8192, 9000, density=0.06, n_types=8, doublet_frac=0.08, seed=607)
    kw = dict(n_iters=7, clustering_algorithm="louvain", standard_scaling=True, random_state=0, n_top_var_genes=5000)
    fits = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for lanes in (7, 1):
            clf = BoostClassifier(streams_per_device=lanes, **kw).fit(data)
            assert clf._lanes_used == lanes
            assert clf._last_bitplane["active"] and clf._last_bitplane["scaled"], clf._last_bitplane
            fits[lanes] = clf
        ref = orc.OracleClassifier(pca="f64", louvain_fn=_native_louvain, **dict(kw, n_iters=2)).fit(data)
    print("scaled fit:", fits[7]._last_bitplane)
    assert fits[7]._last_bitplane["demoted_columns"] > 0      # (followers copied structures the leader had rebuilt at its first scaling)
    for name in ("all_log_p_values_", "all_scores_", "communities_", "synth_communities_"):
        np.testing.assert_array_equal(getattr(fits[7], name), getattr(fits[1], name), err_msg=name)
    np.testing.assert_array_equal(np.asarray(fits[7].parents_[:2]), np.asarray(ref.parents_))
    agree = float(np.mean(fits[7].communities_[:2] == ref.communities_))
    print(f"scaled fit, 8192 cells: communities identical to the float64 oracle for {agree:.4%} of (iteration, cell) pairs")
    from sklearn.metrics import adjusted_rand_score
    assert min(adjusted_rand_score(fits[7].communities_[i], ref.communities_[i]) for i in range(2)) > 0.98
