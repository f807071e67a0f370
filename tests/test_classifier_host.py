"""Host logic of the drop-in BoostClassifier (no GPU): API surface, validation, warnings, RNG stream,
predict / doublet_score against the reference's golden outputs, end-to-end equality with the reference
run when the device stages are supplied by the oracle engine."""
import inspect
import warnings

import numpy as np
import pytest

from conftest import csr_from, load_golden
from doubletdetection_amd import BoostClassifier
from oracle_engine import make_engine_factory


def test_constructor_signature_matches_reference():
    # doubletdetection.py:73-88
    sig = inspect.signature(BoostClassifier.__init__)
    want = [("boost_rate", 0.25), ("n_components", 30), ("n_top_var_genes", 10000), ("replace", False),
            ("clustering_algorithm", "phenograph"), ("clustering_kwargs", None), ("n_iters", 10),
            ("normalizer", None), ("pseudocount", 0.1), ("random_state", 0), ("verbose", False),
            ("standard_scaling", False), ("n_jobs", 1)]
    params = list(sig.parameters.values())[1:]
    positional = [p for p in params if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert [(p.name, p.default) for p in positional] == want
    assert all(p.kind == p.KEYWORD_ONLY for p in params[len(want):])
    assert list(inspect.signature(BoostClassifier.predict).parameters)[1:] == ["p_thresh", "voter_thresh"]
    assert inspect.signature(BoostClassifier.predict).parameters["p_thresh"].default == 1e-7
    assert inspect.signature(BoostClassifier.predict).parameters["voter_thresh"].default == 0.9


def test_validation_and_warnings():
    # tests/test_package.py:45-48
    with pytest.raises(ValueError):
        BoostClassifier(n_iters=2, clustering_algorithm="my_clusters", standard_scaling=True)
    with pytest.raises(ValueError):
        BoostClassifier(clustering_algorithm="louvain", clustering_kwargs={"key_added": "x"})
    with pytest.raises(ValueError):
        BoostClassifier(clustering_algorithm="leiden", clustering_kwargs={"random_state": 1})
    with pytest.raises(AssertionError):
        BoostClassifier(n_components=50, n_top_var_genes=40)
    with pytest.warns(UserWarning, match="experimental"):
        BoostClassifier(clustering_algorithm="leiden")
    with pytest.warns(UserWarning, match="trimmed to 0.5"):
        clf = BoostClassifier(boost_rate=0.7)
    assert clf.boost_rate == 0.5
    with pytest.warns(UserWarning, match="prune=False"):
        BoostClassifier(n_iters=1)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        clf = BoostClassifier(boost_rate=0.7, replace=True)
        assert clf.boost_rate == 0.7
        assert BoostClassifier(n_top_var_genes=20).n_components == 20          # silent cap
        assert BoostClassifier(n_top_var_genes=-5).n_top_var_genes == 0
        kw = BoostClassifier(clustering_algorithm="louvain").clustering_kwargs
        assert kw == {"directed": False, "resolution": 4}
        assert BoostClassifier().clustering_kwargs == {"prune": True}
    with pytest.raises(TypeError):
        BoostClassifier(clustering_kwargs={"no_such_option": 1})


def test_fit_rejects_bad_input():
    clf = BoostClassifier(clustering_algorithm="louvain")
    with pytest.raises(ValueError):
        clf.fit(np.array([1.0, 2.0, 3.0]))                       # not 2-D
    bad = np.ones((600, 120)); bad[3, 4] = np.nan
    with pytest.raises(ValueError):
        clf.fit(bad)
    with pytest.raises(NotImplementedError):
        BoostClassifier(normalizer=lambda x: x).fit(np.ones((600, 120)))


def test_fit_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from doubletdetection_amd._lib import DdxError

    counts = np.random.default_rng(0).poisson(1.0, size=(600, 120))
    with pytest.raises(DdxError):
        BoostClassifier(n_iters=2, clustering_algorithm="louvain").fit(counts)


def test_f9_predict_and_doublet_score_match_reference():
    g = load_golden("f9_predict")
    clf = BoostClassifier(n_iters=g["multi_logp"].shape[0], clustering_algorithm="louvain")
    clf.all_log_p_values_ = g["multi_logp"].copy()
    clf.all_scores_ = g["multi_scores"].copy()
    for tag in ("default", "loose", "mid"):
        pt, vt = g[f"multi_{tag}_params"]
        lab = clf.predict(p_thresh=pt, voter_thresh=vt)
        np.testing.assert_array_equal(lab, g[f"multi_{tag}_labels"])
        np.testing.assert_array_equal(clf.voting_average_, g[f"multi_{tag}_voting"])
    ds = clf.doublet_score()
    assert isinstance(ds, np.ma.MaskedArray)
    np.testing.assert_array_equal(np.ma.getmaskarray(ds), g["multi_dscore_mask"])
    ok = ~g["multi_dscore_mask"]
    np.testing.assert_array_equal(np.ma.getdata(ds)[ok], g["multi_dscore_data"][ok])
    for tag in ("gap", "flat", "allnan_but_one"):
        sc = g[f"single_{tag}_scores"][None, :]
        with pytest.warns(UserWarning):
            one = BoostClassifier(n_iters=1)
        one.all_scores_ = sc.copy()
        one.all_log_p_values_ = np.where(np.isnan(sc), np.nan, -sc * 20)
        lab = one.predict()
        assert lab.dtype == bool
        np.testing.assert_array_equal(lab.astype(np.float64), g[f"single_{tag}_labels"])
        np.testing.assert_array_equal(one.suggested_score_cutoff_, g[f"single_{tag}_cutoff"])
        np.testing.assert_array_equal(one.doublet_score(), g[f"single_{tag}_dscore"])


@pytest.mark.parametrize("case", ["case_a_hvg_pheno", "case_b_transposed_louvain", "case_c_reftest_scaled"])
def test_end_to_end_equals_reference_run_with_oracle_engine(case, monkeypatch):
    """Everything the host layer owns (coercion, HVG, parent stream, kwargs plan, native Louvain +
    scoring, attribute assembly, predict) reproduces the reference's own run bit for bit."""
    from conftest import golden_kwargs

    g = load_golden(case)
    kw = golden_kwargs(g)
    monkeypatch.setattr(BoostClassifier, "_engine_factory", staticmethod(make_engine_factory(kw.get("random_state", 0))))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = BoostClassifier(**kw)
        clf.fit(csr_from(g, "counts"))
    if "top_var_genes" in g.files:
        np.testing.assert_array_equal(clf.top_var_genes_, g["top_var_genes"])
    np.testing.assert_array_equal(np.asarray(clf.parents_, dtype=np.int64), g["parents"])
    assert isinstance(clf.parents_, list) and isinstance(clf.parents_[0][0], list)
    assert isinstance(clf.parents_[0][0][0], np.int64)
    np.testing.assert_array_equal(clf.communities_, g["communities"])
    np.testing.assert_array_equal(clf.synth_communities_, g["synth_communities"])
    assert clf.communities_.dtype == np.float64 and clf.synth_communities_.dtype == np.float64
    np.testing.assert_array_equal(clf.all_scores_, g["all_scores"])
    np.testing.assert_allclose(clf.all_log_p_values_, g["all_log_p_values"], rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(clf.predict(), g["labels_default"])
    np.testing.assert_array_equal(clf.voting_average_, g["voting_average_default"])


def test_second_fit_continues_the_rng_stream(monkeypatch):
    g = load_golden("case_c_reftest_scaled")
    monkeypatch.setattr(BoostClassifier, "_engine_factory", staticmethod(make_engine_factory(0)))
    clf = BoostClassifier(n_iters=1, clustering_algorithm="louvain", standard_scaling=True)
    counts = csr_from(g, "counts")
    clf.fit(counts)
    first = np.asarray(clf.parents_)
    clf.fit(counts)
    second = np.asarray(clf.parents_)
    assert not np.array_equal(first, second)
    np.testing.assert_array_equal(first[0], g["parents"][0])
    np.testing.assert_array_equal(second[0], g["parents"][1])


def test_a_fit_that_fails_while_staging_leaves_the_rng_stream_untouched(monkeypatch):
    """The parent draws start before the input has been validated (they overlap the upload); upstream draws nothing before
    check_array has passed (dd.py:149-155 precede dd.py:394): after a rejected input the next fit draws what a fresh stream draws."""
    g = load_golden("case_c_reftest_scaled")
    monkeypatch.setattr(BoostClassifier, "_engine_factory", staticmethod(make_engine_factory(0)))
    clf = BoostClassifier(n_iters=1, clustering_algorithm="louvain", standard_scaling=True)
    counts = csr_from(g, "counts")
    bad = counts.toarray().astype(np.float64)
    bad[3, 4] = np.nan
    with pytest.raises(ValueError):
        clf.fit(bad)
    clf.fit(counts)
    np.testing.assert_array_equal(np.asarray(clf.parents_)[0], g["parents"][0])
    clf.fit(counts)
    np.testing.assert_array_equal(np.asarray(clf.parents_)[0], g["parents"][1])


def test_reference_plot_functions_accept_the_classifier(monkeypatch):
    """doubletdetection.plot.convergence / threshold only read ``n_iters`` and
    ``all_log_p_values_`` (plot.py:65-66,123); run the reference's own plotting code on a fitted drop-in and
    compare the curves it draws with the same numbers computed from predict().  Needs the reference tree
    (this container) and matplotlib; skipped elsewhere."""
    import importlib.util
    import os

    plot_py = "/root/reference/doubletdetection/plot.py"
    if not os.path.exists(plot_py):
        pytest.skip("reference tree not mounted")
    mpl = pytest.importorskip("matplotlib")
    mpl.use("Agg")
    spec = importlib.util.spec_from_file_location("_ref_plot", plot_py)
    plot = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(plot)

    g = load_golden("case_a_hvg_pheno")
    from conftest import golden_kwargs
    kw = golden_kwargs(g)
    monkeypatch.setattr(BoostClassifier, "_engine_factory", staticmethod(make_engine_factory(kw.get("random_state", 0))))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = BoostClassifier(**kw).fit(csr_from(g, "counts"))
        fig = plot.convergence(clf, show=False, p_thresh=1e-3, voter_thresh=0.5)
        drawn = fig.axes[0].lines[0].get_ydata()
        assert len(drawn) == clf.n_iters
        # the last point of the convergence curve is the number of doublets predict() calls with the same thresholds
        assert drawn[-1] == np.nansum(clf.predict(p_thresh=1e-3, voter_thresh=0.5))
        fig2 = plot.threshold(clf, show=False, p_step=20)
        assert fig2.axes, "threshold() drew nothing"
    import matplotlib.pyplot as plt
    plt.close("all")


def test_clustering_kwargs_never_silently_ignored():
    """Every accepted clustering keyword either shapes the plan the way it shapes the upstream call, is a no-op
    upstream too, or raises -- never a silent difference from the reference (dd.py:116-119,320-322,337-342)."""
    def plan(algo, **kw):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return BoostClassifier(clustering_algorithm=algo, clustering_kwargs=kw, random_state=3)._cluster_plan()

    # phenograph defaults: k=30, no self, pruned Jaccard graph, Louvain binaries take no resolution and no seed
    assert plan("phenograph") == (30, False, 0, 1.0, 3, 10, False, 1e-3)
    assert plan("phenograph", resolution_parameter=2.5, seed=9) == (30, False, 0, 1.0, 3, 10, False, 1e-3)
    assert plan("phenograph", clustering_algo="leiden", resolution_parameter=2.5, seed=9, prune=False, k=15,
                min_cluster_size=4) == (15, False, 1, 2.5, 9, 4, True, None)
    assert plan("phenograph", nn_method="brute", n_jobs=4, q_tol=1e-4, louvain_time_limit=10) == (30, False, 0, 1.0, 3, 10, False, 1e-4)
    assert plan("phenograph", primary_metric="cosine") == (30, False, 0, 1.0, 3, 10, False, 1e-3)      # the metric only changes the kNN stage
    # directed=True: no symmetrisation upstream (prune is ignored), the Louvain converter adds both orientations:
    # twice the averaged graph, the same partition
    assert plan("phenograph", directed=True) == (30, False, 1, 1.0, 3, 10, False, 1e-3)
    assert plan("phenograph", directed=True, prune=True) == (30, False, 1, 1.0, 3, 10, False, 1e-3)
    for bad in ({"directed": True, "clustering_algo": "leiden"}, {"jaccard": False}, {"primary_metric": "chebyshev"}, {"nn_method": "faiss"},
                {"partition_type": object()}, {"clustering_algo": "leiden", "n_iterations": 3},
                {"clustering_algo": "leiden", "use_weights": False}):
        with pytest.raises(NotImplementedError):
            plan("phenograph", **bad)
    with pytest.raises(ValueError):
        plan("phenograph", clustering_algo="spectral")
    # scanpy: louvain ignores weights unless asked, leiden uses them unless asked not to
    assert plan("louvain") == (10, True, 2, 4.0, 3, None, False, None)
    assert plan("louvain", use_weights=True, resolution=1.5) == (10, True, 3, 1.5, 3, None, False, None)
    assert plan("leiden") == (10, True, 3, 4.0, 3, None, True, None)
    assert plan("leiden", use_weights=False) == (10, True, 2, 4.0, 3, None, True, None)
    # scanpy's directed=True doubles every edge of the symmetric connectivities: the same modularity term by term
    assert plan("louvain", directed=True) == plan("louvain", directed=False) == plan("louvain")
    assert plan("leiden", directed=True) == plan("leiden")
    for algo, bad in (("louvain", {"restrict_to": ("a", ["1"])}),
                      ("leiden", {"adjacency": 1}), ("louvain", {"obsp": "x"}), ("leiden", {"neighbors_key": "n"}),
                      ("louvain", {"partition_type": object()}), ("louvain", {"flavor": "igraph"}),
                      ("leiden", {"flavor": "igraph"}), ("leiden", {"n_iterations": 2})):
        with pytest.raises(NotImplementedError):
            plan(algo, **bad)
    with pytest.raises(TypeError):
        plan("louvain", n_iterations=2)            # sc.tl.louvain has no such keyword


def test_device_limits_raise_before_any_upload():
    """n_components / k beyond what the device kernels hold fail up front with a clear message (not mid-fit)."""
    class Untouchable:
        def __init__(self, device):
            raise AssertionError("an engine was created before the limits were checked")

    counts = np.random.default_rng(0).poisson(1.0, size=(900, 700))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kw in (dict(n_components=129, n_top_var_genes=0), dict(n_components=400, n_top_var_genes=0),
                   dict(clustering_kwargs={"k": 257})):
            clf = BoostClassifier(n_iters=2, **kw)
            clf._engine_factory = Untouchable
            with pytest.raises(NotImplementedError, match="device"):
                clf.fit(counts)


def test_non_canonical_csr_is_canonicalised(monkeypatch):
    """Duplicate / unsorted column entries are summed and sorted on the way in, as scipy's own indexing and
    addition (dd.py:174-176,397-399) would do; the caller's matrix is left untouched."""
    import scipy.sparse as sp

    seen = {}

    class Recorder:
        def __init__(self, device):
            pass

        def close(self):
            pass

        def upload(self, csr):
            # what libddx does with the arrays it receives (ddx_upload_counts validates them on the device)
            from doubletdetection_amd import _lib

            if not csr.has_canonical_format:
                raise _lib.DdxError(_lib.E_ARG, "CSR rows must hold strictly increasing column indices (sorted, no duplicates)")
            seen["csr"] = csr
            raise RuntimeError("stop here")

    indptr = np.array([0, 3, 5], dtype=np.int32)
    indices = np.array([2, 0, 2, 1, 1], dtype=np.int32)
    data = np.array([1, 2, 3, 4, 5], dtype=np.float32)
    x = sp.csr_matrix((data, indices, indptr), shape=(2, 3))
    assert not x.has_canonical_format
    clf = BoostClassifier(n_iters=2, n_top_var_genes=0, clustering_algorithm="louvain")
    clf._engine_factory = Recorder
    with pytest.raises(RuntimeError, match="stop here"):
        clf.fit(x)
    got = seen["csr"]
    assert got.has_canonical_format and got.dtype == np.float32
    np.testing.assert_array_equal(got.toarray(), [[2, 0, 4], [0, 9, 0]])
    np.testing.assert_array_equal(x.indices, indices)           # caller's arrays untouched


@pytest.mark.parametrize("devices,streams", [([0], 1), ([0], 3), ([0, 1], 1), ([2, 0, 1], 2)])
def test_lanes_do_not_change_results(monkeypatch, devices, streams):
    """Boosting iterations dealt out over GPUs x streams of one process (dd.py:192-198 are independent given the
    pre-drawn parents): every layout gives the arrays of the single-lane run, and followers clone their leader."""
    from oracle_engine import OracleEngine

    g = load_golden("case_c_reftest_scaled")
    counts = csr_from(g, "counts")
    made = []

    def factory(device):
        e = OracleEngine(device)
        e.seed = 0
        e.cloned_from = None
        made.append(e)
        return e

    plain_clone = OracleEngine.clone_from

    def clone_from(self, other):
        plain_clone(self, other)
        self.cloned_from = other

    monkeypatch.setattr(OracleEngine, "clone_from", clone_from, raising=False)
    kw = dict(n_iters=5, clustering_algorithm="louvain", standard_scaling=True, random_state=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = BoostClassifier(streams_per_device=1, **kw)
        base._engine_factory = make_engine_factory(0)
        base.fit(counts)
        clf = BoostClassifier(devices=devices, streams_per_device=streams, **kw)
        clf._engine_factory = factory
        clf.fit(counts)
    for name in ("all_log_p_values_", "all_scores_", "communities_", "synth_communities_"):
        np.testing.assert_array_equal(getattr(clf, name), getattr(base, name))
    np.testing.assert_array_equal(np.asarray(clf.parents_), np.asarray(base.parents_))
    want_lanes = min(5, len(devices) * streams)
    assert clf._lanes_used == want_lanes and len(made) == want_lanes
    leaders = [e for e in made if e.cloned_from is None]
    assert sorted(e.device for e in leaders) == sorted(devices)
    for e in made:
        if e.cloned_from is not None:
            assert e.cloned_from.device == e.device and e.cloned_from.cloned_from is None


def test_contexts_per_gpu_follow_the_iterations_and_the_memory(monkeypatch):
    """streams_per_device=None: one context per iteration of a GPU's share (dd.py:192-198), up to seven, and never more than
    the GPU's memory and the parking allowance hold beside the leader; explicit settings are kept."""
    monkeypatch.delenv("DDX_STREAMS", raising=False)
    clf = BoostClassifier()
    assert [clf._stream_count(n) for n in (1, 2, 4, 5, 6, 7, 10, 11, 25, 50)] == [1, 2, 4, 5, 6, 7, 7, 7, 7, 7]
    assert clf._stream_count(10, n_devices=2) == 5 and clf._stream_count(10, n_devices=4) == 3
    assert BoostClassifier(streams_per_device=2)._stream_count(10) == 2
    monkeypatch.setenv("DDX_STREAMS", "3")
    assert clf._stream_count(10) == 3
    monkeypatch.delenv("DDX_STREAMS")
    with pytest.raises(ValueError):
        BoostClassifier(streams_per_device=0)._stream_count(10)

    class Ctx:
        def __init__(self, held, free):
            self.held, self.free = held, free

        def device_bytes(self):
            return self.held

        def device_memory(self):
            return self.free, 288 << 30

    class Leader:
        device = 0

        def __init__(self, held, free):
            self.ctx = Ctx(held, free)

    monkeypatch.delenv("DDX_PARK_MAX_GB", raising=False)
    assert clf._stream_count(10, 1, Leader(12 << 30, 240 << 30)) == 6        # default allowance: a quarter of the GPU (72 GB)
    assert clf._stream_count(10, 1, Leader(30 << 30, 240 << 30)) == 2
    monkeypatch.setenv("DDX_PARK_MAX_GB", "256")
    assert clf._stream_count(10, 1, Leader(40 << 30, 240 << 30)) == 6        # 5 followers of 40 GB fit into 216 GB
    monkeypatch.setenv("DDX_PARK_MAX_GB", "128")
    assert clf._stream_count(10, 1, Leader(40 << 30, 240 << 30)) == 3        # ... but only three such contexts can stay parked
    assert clf._stream_count(10, 1, Leader(12 << 30, 240 << 30)) == 7
    monkeypatch.setenv("DDX_PARK_MAX_GB", "256")
    assert clf._stream_count(10, 1, Leader(60 << 30, 140 << 30)) == 3        # two more of 60 GB, not four
    assert clf._stream_count(10, 1, Leader(100 << 30, 50 << 30)) == 1
    # a follower holds the restricted counts only: a 38 GB leader and a 30 GB follower fit the default 72 GB (configs[3] on one GPU),
    # two 38 GB contexts would not
    monkeypatch.delenv("DDX_PARK_MAX_GB", raising=False)

    class SizedCtx(Ctx):
        def __init__(self, held, free, follower):
            super().__init__(held, free)
            self.follower = follower

        def follower_bytes(self):
            return self.follower

    lead = Leader(38 << 30, 240 << 30)
    assert clf._stream_count(10, 1, lead) == 1
    lead.ctx = SizedCtx(38 << 30, 240 << 30, 30 << 30)
    assert clf._stream_count(10, 1, lead) == 2
    lead.ctx = SizedCtx(38 << 30, 240 << 30, 0)                              # no counts yet: as large as the leader
    assert clf._stream_count(10, 1, lead) == 1
    lead.ctx = SizedCtx(11 << 30, 240 << 30, 9 << 30)
    assert clf._stream_count(10, 1, lead) == 7


def test_host_wait_follows_the_cpu_allowance_and_the_world(monkeypatch):
    """How lane threads wait for the GPU (ddx.h: host_wait): spinning while the host has a CPU for every waiting thread of one rank;
    sleeping polls when several ranks share the node or the allowance is smaller than the lanes; never overriding the caller's choice."""
    from doubletdetection_amd import _lib, classifier

    class Ctx:
        def __init__(self):
            self.opts = {}

        def set_option(self, k, v):
            self.opts[k] = v

    class Eng:
        def __init__(self):
            self.ctx = Ctx()

    monkeypatch.delenv("DDX_OPTIONS", raising=False)
    monkeypatch.setattr(classifier, "_cpu_allowance", lambda: 16.0)
    clf = BoostClassifier()
    assert clf._fit_switches(Eng(), world=1).ctx.opts == {"host_wait": "spin"}
    assert clf._fit_switches(Eng(), world=8).ctx.opts == {"host_wait": "block"}
    monkeypatch.setattr(classifier, "_cpu_allowance", lambda: 4.0)
    assert clf._fit_switches(Eng(), world=1).ctx.opts == {"host_wait": "block"}
    assert BoostClassifier(streams_per_device=2)._fit_switches(Eng(), world=1).ctx.opts == {"host_wait": "spin"}
    monkeypatch.setitem(_lib.OPTIONS, "host_wait", "spin")
    assert clf._fit_switches(Eng(), world=8).ctx.opts == {}
