"""TEST INFRASTRUCTURE (oracle side) -- runs the *reference's own* ``doubletdetection.py``.

This module is only usable inside the build container, where ``/root/reference`` is mounted.
It never copies reference source: it loads
``/root/reference/doubletdetection/doubletdetection.py`` by path after planting three stand-in
modules (``anndata``, ``scanpy``, ``phenograph``) into ``sys.modules`` -- those three packages are
imported by the reference (doubletdetection.py:9,12,13) but are not installed in this image and
cannot be installed (no network).  Everything in the reference that is *not* a call into one of
those packages therefore executes verbatim (doublet creation :385-402, normalisation :286-298,
community bookkeeping + hypergeometric scoring :344-383, predict :216-254, doublet_score :256-272).

The stand-ins route the third-party calls to hooks:

* ``sc.tl.pca``      -> ``sklearn.decomposition.PCA(n_components, svd_solver, random_state)``;
                        that is literally what scanpy calls for a dense array.
* ``sc.pp.scale``    -> ``oracle.dd_oracle.scale_like_scanpy`` (restated from scanpy's documentation;
                        *not* reference-generated -- scanpy itself is absent).
* ``sc.pp.neighbors`` + ``sc.tl.louvain/leiden`` and ``phenograph.cluster``
                     -> either an injected community vector (bit-exact tests of the reference's
                        bookkeeping lines) or the oracle's deterministic clustering restatement.

Only ``oracle/make_golden.py`` and tests guarded by ``ref_available()`` import this file.
Nothing here ships to, or is read on, the GPU box.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np

REF_DD = "/root/reference/doubletdetection/doubletdetection.py"


def ref_available() -> bool:
    return os.path.isfile(REF_DD)


class _Obs(dict):
    """Minimal stand-in for ``AnnData.obs`` (a column store)."""


class _AnnData:
    """Container-only stand-in for ``anndata.AnnData`` (reference call site :300-301)."""

    def __init__(self, X):
        self.X = X
        self.obs = _Obs()
        self.obsm = {}
        self.uns = {}
        _Recorder.last_anndata_input = X

    @property
    def shape(self):
        return self.X.shape


class _Recorder:
    """What the reference handed to its third-party callees during the last run."""

    last_anndata_input = None
    pca_inputs: list = []
    pca_outputs: list = []
    scale_outputs: list = []
    neighbor_calls: list = []
    cluster_calls: list = []

    @classmethod
    def reset(cls):
        cls.last_anndata_input = None
        cls.pca_inputs = []
        cls.pca_outputs = []
        cls.scale_outputs = []
        cls.neighbor_calls = []
        cls.cluster_calls = []


class Hooks:
    """Mutable hook table; tests/golden generation overwrite entries."""

    scale = None          # f(X_dense_f32, max_value) -> X
    pca = None            # f(X, n_comps, random_state, svd_solver) -> scores
    cluster_scanpy = None  # f(X_pca, algo, n_neighbors, random_state, kwargs) -> int labels [M]
    cluster_phenograph = None  # f(X_pca, n_jobs, kwargs) -> int labels [M]


def _default_pca(X, n_comps, random_state, svd_solver):
    from sklearn.decomposition import PCA

    return PCA(n_components=n_comps, svd_solver=svd_solver, random_state=random_state).fit_transform(X)


def _install_standins():
    anndata = types.ModuleType("anndata")
    anndata.AnnData = _AnnData

    sc = types.ModuleType("scanpy")
    sc.settings = types.SimpleNamespace(n_jobs=1)
    sc.pp = types.SimpleNamespace()
    sc.tl = types.SimpleNamespace()

    def scale(adata, max_value=None):
        adata.X = Hooks.scale(adata.X, max_value)
        _Recorder.scale_outputs.append(np.array(adata.X, copy=True))

    def pca(adata, n_comps=None, random_state=0, svd_solver=None):
        _Recorder.pca_inputs.append(adata.X.copy())
        fn = Hooks.pca or _default_pca
        scores = fn(adata.X, n_comps, random_state, svd_solver)
        # scanpy stores X_pca in its `dtype` argument's type (float32 default)
        adata.obsm["X_pca"] = np.asarray(scores).astype(np.float32)
        _Recorder.pca_outputs.append(adata.obsm["X_pca"].copy())

    def neighbors(adata, random_state=0, method="umap", n_neighbors=15):
        rec = dict(random_state=random_state, method=method, n_neighbors=n_neighbors)
        _Recorder.neighbor_calls.append(rec)
        adata.uns["_neighbors"] = rec

    def _clus(algo):
        def run(adata, key_added=None, random_state=0, **kw):
            _Recorder.cluster_calls.append(dict(algo=algo, key_added=key_added,
                                                random_state=random_state, kw=dict(kw)))
            nb = adata.uns["_neighbors"]
            labels = Hooks.cluster_scanpy(adata.obsm["X_pca"], algo, nb["n_neighbors"],
                                          random_state, dict(kw))
            # scanpy stores a string categorical; the reference parses it with dtype=int (:343)
            adata.obs[key_added] = np.asarray(labels).astype(int).astype("U")
        return run

    sc.pp.scale = scale
    sc.pp.neighbors = neighbors
    sc.tl.pca = pca
    sc.tl.louvain = _clus("louvain")
    sc.tl.leiden = _clus("leiden")

    phenograph = types.ModuleType("phenograph")

    def cluster(data, n_jobs=1, **kw):
        _Recorder.cluster_calls.append(dict(algo="phenograph", n_jobs=n_jobs, kw=dict(kw)))
        labels = Hooks.cluster_phenograph(data, n_jobs, dict(kw))
        return np.asarray(labels).astype(int), None, None

    phenograph.cluster = cluster
    return {"anndata": anndata, "scanpy": sc, "phenograph": phenograph}


_REF_MODULE = None


def load_reference():
    """Import the reference's doubletdetection.py by path (cached)."""
    global _REF_MODULE
    if _REF_MODULE is not None:
        return _REF_MODULE
    if not ref_available():
        raise RuntimeError("reference tree not mounted; golden fixtures must be used instead")
    saved = {k: sys.modules.get(k) for k in ("anndata", "scanpy", "phenograph")}
    sys.modules.update(_install_standins())
    try:
        spec = importlib.util.spec_from_file_location("_ref_doubletdetection", REF_DD)
        mod = importlib.util.module_from_spec(spec)
        sys.dont_write_bytecode, old = True, sys.dont_write_bytecode
        try:
            spec.loader.exec_module(mod)
        finally:
            sys.dont_write_bytecode = old
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _REF_MODULE = mod
    return mod


def recorder():
    return _Recorder


def make_probe_class():
    """Subclass of the reference classifier that snapshots ``_raw_synthetics`` per iteration."""
    ref = load_reference()

    class Probe(ref.BoostClassifier):
        def _createDoublets(self):  # noqa: N802  (reference's own name, :385)
            super()._createDoublets()
            if not hasattr(self, "probe_synthetics"):
                self.probe_synthetics = []
            self.probe_synthetics.append(self._raw_synthetics.copy())

    return Probe
