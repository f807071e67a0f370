"""The layout bench.py measures, tested as a whole fit (GPU only; dd.py:192-198, 305-314).

The headline fit is: automatic bit-plane route (from 4096 cells on) x doublets derived from their parents' structures
(k_bp_synth) x seven device contexts whose followers copy the leader's resident counts and bit-plane structures WHILE the
leader already iterates.  The golden cases force that route onto 500-cell matrices with two lanes; here it is taken the way
production takes it -- default options, sizes at which it is selected by itself:

* 8 192 x 6 400 restricted to its 6 000 most variable genes (the CPU oracle is still affordable): seven lanes == one lane, attribute by attribute; the seven-lane fit ==
  `oracle.OracleClassifier(pca="f64")` (communities and scores identical, log p 1e-9);
* BASELINE configs[1] (50 000 x 20 000, 5 %): seven lanes == one lane;
* followers that clone while the leader's create_doublets has to GROW the buffers they are copying from (boost_rate 0.5 at
  >= 4096 cells, the advisor's round-5 scenario): equal to the single-lane run.

Every test asserts that the bit-plane route was active and how many lanes ran, so that none of them can pass on a silent
fall-back to the plain route or to a single context.
"""
import warnings

import numpy as np
import pytest

from oracle import dd_oracle as orc

pytestmark = pytest.mark.gpu

ATTRS = ("all_log_p_values_", "all_scores_", "communities_", "synth_communities_")


def _native_louvain(indptr, indices, weights, gamma, seed):
    from doubletdetection_amd import _lib

    return _lib.louvain(indptr, indices, weights, gamma, seed)[0].astype(np.int64)


def _native_best_of(indptr, indices, weights, gamma, seed, q_tol):
    from doubletdetection_amd import _lib

    return _lib.louvain_best_of(indptr, indices, weights, gamma, seed, q_tol, threads=8)[0].astype(np.int64)


def _fit(data, lanes, bitplane=True, **kw):
    from doubletdetection_amd import BoostClassifier

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = BoostClassifier(streams_per_device=lanes, **kw).fit(data)
    assert clf._lanes_used == min(lanes, kw.get("n_iters", 10)), clf._lanes_used
    assert clf._last_bitplane is not None and clf._last_bitplane["active"] == bitplane, clf._last_bitplane
    return clf


def _same_fit(a, b):
    np.testing.assert_array_equal(np.asarray(a.parents_), np.asarray(b.parents_))
    for name in ATTRS:
        np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg=name)
    assert hasattr(a, "top_var_genes_") == hasattr(b, "top_var_genes_")      # (dd.py:165-176: only set when genes are dropped)
    if hasattr(a, "top_var_genes_"):
        np.testing.assert_array_equal(a.top_var_genes_, b.top_var_genes_)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.array_equal(a.predict(), b.predict(), equal_nan=True)


@pytest.fixture(scope="module")
def data_8k():
    from doubletdetection_amd._synthetic import make_counts

    return make_counts(8192, 6400, density=0.08, n_types=8, doublet_frac=0.08, seed=606)


@pytest.fixture(scope="module")
def fit_8k_seven(data_8k):
    return _fit(data_8k, 7, n_iters=7, random_state=0, n_top_var_genes=6000)


def test_seven_contexts_equal_one_context_on_the_automatic_bitplane_route(data_8k, fit_8k_seven):
    """(a) default options at 8 192 cells: BoostClassifier(n_iters=7, streams_per_device=7) == streams_per_device=1."""
    one = _fit(data_8k, 1, n_iters=7, random_state=0, n_top_var_genes=6000)
    _same_fit(fit_8k_seven, one)


def test_seven_context_fit_matches_the_float64_oracle(data_8k, fit_8k_seven):
    """(b) the same seven-lane fit against the CPU oracle with the float64 PCA: integer results identical, log p to 1e-9."""
    clf = fit_8k_seven
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = orc.OracleClassifier(pca="f64", louvain_fn=_native_louvain, best_of_fn=_native_best_of, n_iters=7, random_state=0,
                                   n_top_var_genes=6000).fit(data_8k)
    np.testing.assert_array_equal(clf.top_var_genes_, ref.top_var_genes_)
    np.testing.assert_array_equal(np.asarray(clf.parents_), np.asarray(ref.parents_))
    agree = float(np.mean(clf.communities_ == ref.communities_))
    print(f"8192 cells, 7 contexts, bit planes: community labels identical to the float64 oracle for {agree:.4%} of (iteration, cell) pairs")
    np.testing.assert_array_equal(clf.communities_, ref.communities_)
    np.testing.assert_array_equal(clf.synth_communities_, ref.synth_communities_)
    np.testing.assert_array_equal(clf.all_scores_, ref.all_scores_)
    np.testing.assert_allclose(clf.all_log_p_values_, ref.all_log_p_values_, rtol=1e-9, atol=1e-9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.array_equal(clf.predict(), ref.predict(), equal_nan=True)


def test_the_mx_form_of_the_products_gives_the_same_fit(data_8k, fit_8k_seven, monkeypatch):
    """The products on the MX matrix instruction (option bp_format=mx6: FP4 bitmap x six base-31 digits in FP6, 28.7 bits) against the
    default int8 form (30 bits) -- which the test above holds to the float64 oracle: the same seven-lane fit, compared attribute by
    attribute.  (The three-digit schedule bp_digits_early=3 fails exactly this comparison: 0.65 % of the labels.)"""
    from doubletdetection_amd import _lib

    monkeypatch.setitem(_lib.OPTIONS, "bp_format", "mx6")
    from doubletdetection_amd import classifier

    classifier.release_device_memory()               # (parked contexts keep the options they were opened with)
    try:
        mx = _fit(data_8k, 7, n_iters=7, random_state=0, n_top_var_genes=6000)
        assert mx._last_bitplane["format"] == "mx6"
        agree = float(np.mean(mx.communities_ == fit_8k_seven.communities_))
        print(f"8192 cells, 7 contexts: community labels of the MX form identical to the int8 form's for {agree:.4%} of (iteration, cell) pairs")
        _same_fit(mx, fit_8k_seven)
    finally:
        monkeypatch.delitem(_lib.OPTIONS, "bp_format", raising=False)
        classifier.release_device_memory()


def test_c2_seven_contexts_equal_one_context():
    """(c) BASELINE configs[1]: the shipped layout against the single-context run of the same fit."""
    from doubletdetection_amd._synthetic import make_counts

    data = make_counts(50_000, 20_000, density=0.05, device="cuda:0", seed=2)
    seven = _fit(data, 7, n_iters=7, random_state=0)
    one = _fit(data, 1, n_iters=7, random_state=0)
    _same_fit(seven, one)


@pytest.mark.parametrize("kw, bitplane", [(dict(boost_rate=0.5), True), (dict(boost_rate=0.6, replace=True), False)], ids=["rate0.5", "rate0.6-replace"])
def test_followers_clone_while_the_leader_grows_its_buffers(kw, bitplane):
    """boost_rate above ~0.31 makes the leader's first create_doublets outgrow the room reserved behind the original rows
    (and, with replacement above 0.5, the row pointer and library-size arrays): the buffers its followers are copying at that
    moment must stay where they are and what they hold (ddx.h: ddx_clone_counts, CloneView).  Repeated, because the overlap
    is a matter of timing.  (More doublets than half the cells leave the bit-plane route: that case runs the plain products.)"""
    from doubletdetection_amd._synthetic import make_counts

    data = make_counts(6144, 3000, density=0.1, n_types=6, seed=91)
    one = _fit(data, 1, bitplane, n_iters=4, random_state=5, **kw)
    for _ in range(3):
        many = _fit(data, 4, bitplane, n_iters=4, random_state=5, **kw)
        _same_fit(many, one)
