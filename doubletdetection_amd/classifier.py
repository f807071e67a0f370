"""BoostClassifier -- MI355X-native drop-in for DoubletDetection's classifier.

Mirrors the public surface of the reference class (constructor signature and defaults
``doubletdetection/doubletdetection.py:73-88``, ``fit`` ``:135``, ``predict`` ``:216``,
``doublet_score`` ``:256``, fitted attributes ``:54-71``), so code written against the reference runs
unchanged.  The boosting loop itself (``:192-198`` -> ``_one_fit`` ``:274-383`` ->
``_createDoublets`` ``:385-402``) does not run through numpy / scipy / scanpy / phenograph: every
iteration is executed by hand-written HIP kernels for gfx950 behind the C-ABI in ``include/ddx.h``.

Division of labour (host keeps only what SURVEY.md section 3.5 assigns to it):

* host (this file): argument validation, warnings, the numpy ``Generator`` parent draws and the
  legacy ``RandomState`` PCA start matrix (so both random streams are the reference's, bit for bit),
  the one-off HVG ``argsort``, orchestration, ``predict`` / ``doublet_score`` post-processing;
* device (libddx.so): resident counts, synthetic doublets, log-normalisation, optional scaling,
  randomized PCA, exact kNN, Jaccard / neighbour graphs;
* host C++ inside libddx.so: deterministic Louvain and the hypergeometric scoring, run on worker
  threads so they overlap with the next iteration's GPU work.

Multi-GPU: one process per GPU (``torch.distributed``; RCCL when the backend is ``nccl``).  Boosting
iterations are independent given the pre-drawn parent indices, so iteration ``i`` runs on rank
``i % world_size`` and a single all-gather of the per-iteration result rows assembles the fitted
attributes on every rank.
"""
from __future__ import annotations

import os
import warnings
from collections.abc import Callable
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import scipy.sparse as sp_sparse
from numpy.typing import NDArray

from . import _lib

__all__ = ["BoostClassifier"]

_ALGORITHMS = ("louvain", "phenograph", "leiden")

# keyword arguments of the upstream clustering entry points that are understood here; anything else
# raises TypeError exactly like passing an unknown keyword to the upstream function would.
_PHENOGRAPH_KW = {"k", "prune", "min_cluster_size", "resolution_parameter", "seed", "n_jobs", "q_tol",
                  "louvain_time_limit", "nn_method", "primary_metric", "directed", "jaccard",
                  "clustering_algo", "n_iterations", "use_weights", "partition_type"}
_SCANPY_KW = {"resolution", "directed", "use_weights", "partition_type", "restrict_to", "adjacency",
              "neighbors_key", "obsp", "copy", "flavor", "n_iterations"}


def _dist_info():
    """(rank, world_size, backend or None) of an initialised torch.distributed job, else (0, 1, None)."""
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover - torch always present in the target image
        return 0, 1, None
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), dist.get_backend()
    return 0, 1, None


# Device contexts (stream + the memory chunks their buffers are carved from) are parked per GPU when a fit ends and
# taken up again by the next fit of this process, like a caching allocator.  Measured on MI355X (profiles/r02_alloc_notes.txt):
# a fit that has to obtain its ~10 GB chunks from the driver anew stalls for 0.8-1.5 s every other fit once two contexts
# per GPU are in play, against 0.25 s for the whole fit.  The reference frees its (host) intermediates when fit() returns
# (dd.py:200-205); to get that behaviour set DDX_KEEP_CONTEXT=0 (read when a fit ends), or call
# release_device_memory() at any time -- it is also registered with atexit.
_CONTEXT_POOL: dict = {}          # device -> [parked _lib.Context, ...]


def _cpu_allowance() -> float:
    """CPUs' worth of time this process may use: the cgroup quota when there is one (the pods these GPUs come in show 256 CPUs and
    allow 16), else the CPUs it may run on."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1.0, float(quota) / float(period))
    except (OSError, ValueError):
        pass
    try:
        return float(len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return float(os.cpu_count() or 1)


_SHARE_TOKEN = {}


def _share_upload(rank, world, backend, one_leader=True):
    """One packing per node under torch.distributed (dd.py:149-160 x ranks): the ranks agree on a name for the node's shared image once per
    process group -- rank 0 draws it, everybody receives it -- and tell the library their place on the node.  DDX_UPLOAD_SHARE=0 turns it off."""
    if not hasattr(_lib, "set_upload_share"):
        return
    try:
        if world <= 1 or backend is None or not one_leader or os.environ.get("DDX_UPLOAD_SHARE", "1") == "0":
            _lib.set_upload_share("", 0, 1)
            return
        import torch.distributed as dist

        key = id(dist.group.WORLD)
        if key not in _SHARE_TOKEN:
            import time

            box = [f"{os.getpid()}_{int(time.time() * 1e6) % 10 ** 12}" if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            _SHARE_TOKEN.clear()
            _SHARE_TOKEN[key] = box[0]
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        local_rank = int(os.environ.get("LOCAL_RANK", rank)) % max(1, local_world)
        _lib.set_upload_share(_SHARE_TOKEN[key], local_rank, local_world)
    except Exception:                      # (sharing is an optimisation: any trouble and every rank packs for itself)
        try:
            _lib.set_upload_share("", 0, 1)
        except Exception:
            pass


def _keep_contexts() -> bool:
    return os.environ.get("DDX_KEEP_CONTEXT", "1") not in ("", "0")


def _park_limit_bytes(total_bytes: int | None = None) -> int:
    """HBM the parked contexts of one GPU may hold between fits: a quarter of the GPU's memory by default (72 of the 288 GB of
    an MI355X -- whatever else shares the GPU keeps the rest; DDX_PARK_MAX_GB raises or lowers it).  What goes beyond is
    returned to the driver when a context is parked, largest holder first."""
    env = os.environ.get("DDX_PARK_MAX_GB")
    if env:
        return int(float(env) * (1 << 30))
    return int(0.25 * (total_bytes if total_bytes else 288 << 30))


def _park(device, ctx) -> None:
    parked = _CONTEXT_POOL.setdefault(device, [])
    parked.append(ctx)
    try:
        total = ctx.device_memory()[1]
    except Exception:
        total = None
    limit = _park_limit_bytes(total)
    held = [c.device_bytes() for c in parked]
    while sum(held) > limit:
        big = max(range(len(parked)), key=lambda i: held[i])
        parked[big].trim(max(0, limit - (sum(held) - held[big])))      # what the others leave of the allowance
        new = parked[big].device_bytes()
        if new == held[big]:
            break
        held[big] = new


def release_device_memory() -> None:
    """Destroy the parked device contexts and free their HBM."""
    while _CONTEXT_POOL:
        _, parked = _CONTEXT_POOL.popitem()
        for ctx in parked:
            ctx.close()


import atexit  # noqa: E402

atexit.register(release_device_memory)


class _HipEngine:
    """Device side of one fit on one GPU: thin sequencing of the libddx stages."""

    def __init__(self, device: int):
        self.device = device
        parked = _CONTEXT_POOL.get(device)
        if parked:
            self.ctx = parked.pop()
            self.ctx.apply_options()                     # a parked context starts its next fit with the switches as they are NOW
        else:
            self.ctx = _lib.Context(device)
        timing = os.environ.get("DDX_TIMING") == "1"     # per-kernel HIP-event timing (bench.py / profiling)
        self.ctx.timing_enable(timing)
        self.timing = timing
        if timing:
            self.ctx.timing_reset()
            self.ctx.timing_reference()                  # clock origin of the scope intervals (followers adopt their leader's)

    def close(self, discard: bool = False):
        """Park the context for the next fit, or destroy it (``discard``: a fit that failed does not hand its context,
        possibly in an error state, to the next one)."""
        if self.ctx is not None:
            if discard:
                ctx, self.ctx = self.ctx, None
                ctx.close()
                return
            if self.ctx._guard:
                self.ctx.check_memory()                  # overflow detector (tests): raises, naming the buffer
            if _keep_contexts():
                _park(self.device, self.ctx)
            else:
                self.ctx.close()                         # frees every HBM buffer of the fit
            self.ctx = None

    def upload(self, csr):
        self.ctx.upload_counts(csr)

    def stage_raw(self, csr):
        self.ctx.upload_raw(csr)

    def gene_variances(self):
        return self.ctx.gene_variances()

    def select_columns(self, cols):
        self.ctx.select_columns(cols)

    def clone_from(self, other):
        """Take over the resident counts of another engine on the same GPU (device-to-device, ddx_clone_counts)."""
        self.ctx.clone_counts_from(other.ctx)
        if self.timing and other.timing:
            self.ctx.timing_reference(other.ctx)         # one time axis for the streams of a GPU

    def run_iteration(self, parents, pseudocount, standard_scaling, n_components, q0, knn_k, include_self,
                      graph_mode, gamma=None, pca_lock=None, verbose=False, metric="euclidean"):
        """One boosting iteration on the device.  Returns the symmetric graph (indptr, indices, weights), or --
        when ``gamma`` is given -- the result of the synchronous pre-sweeps run on the device:
        (member, coarse indptr, coarse indices, coarse weights)."""
        self.first_half(parents, pseudocount, standard_scaling, n_components, q0, pca_lock, verbose)
        return self.second_half(knn_k, include_self, graph_mode, gamma, verbose, metric)

    def first_half(self, parents, pseudocount, standard_scaling, n_components, q0, pca_lock=None, verbose=False, after_scale=None):
        """dd.py:275-314: synthetic doublets, normalisation, optional scaling, PCA.  Touches neither the graph nor the
        coarsening work space of the previous iteration, so that iteration's part C can still follow (``refine``).

        ``pca_lock`` (optional, DDX_PCA_LOCK=1): the contexts (streams) of one GPU take turns in the PCA stage.  What a
        second stream buys is that its latency-bound stages (graph construction, community pre-sweeps, sorts, the small
        factorisations) run in the shadow of the other stream's operator products."""
        c = self.ctx
        if verbose:
            print("\nCreating synthetic doublets...")        # the reference's stage messages (dd.py:275-276,280-281,305-306,315-316)
        c.create_doublets(parents)
        if verbose:
            print("Normalizing...")
        c.lognormalise(pseudocount)
        if standard_scaling:
            c.scale(15.0)
        if after_scale is not None:
            after_scale()                 # (the leader's first scaling has fixed the per-fit structures its followers copy)
        if verbose:
            print("Running PCA...")
        if pca_lock is not None:
            pca_lock.acquire()
        try:
            self._pca(n_components, q0)
        finally:
            if pca_lock is not None:
                pca_lock.release()

    def second_half(self, knn_k, include_self, graph_mode, gamma=None, verbose=False, metric="euclidean"):
        """dd.py:315-343 up to the sequential part: kNN, graph, and (``gamma`` given) part A of the community detection."""
        c = self.ctx
        if verbose:
            print("Clustering augmented data set...\n")
        c.knn(knn_k, include_self, metric)
        if gamma is None:
            return c.build_graph(graph_mode)  # symmetric CSR assembled on the device
        c.build_graph(graph_mode, fetch=False)
        try:
            return c.coarsen_graph(gamma)
        except _lib.DdxError as err:          # a hub with more neighbours than the device sweep handles
            if err.code != _lib.E_UNSUPPORTED:
                raise
            return c.fetch_graph()

    def refine(self, coarse_labels, gamma):
        """Part C of the community detection on the device: the labels part B gave to the coarse nodes of the LAST
        second_half -> final labels of the augmented cells.  Valid until the next second_half of this engine."""
        return self.ctx.refine_communities(coarse_labels, gamma)

    def _pca(self, n_components, q0):
        c = self.ctx
        if isinstance(q0, str) and q0 == "arpack":
            self._pca_arpack(n_components, self._arpack_seed)
        elif q0 is None:
            self._pca_exact(n_components)
        elif getattr(self, "_q0_resident", None) is q0:
            # every iteration of a fit starts from the same seeded matrix: it is uploaded once per array object
            c.pca(n_components, None, q0_rows=q0.shape[0])
        else:
            c.pca(n_components, q0)
            self._q0_resident = q0

    def _apply(self, X, mode, block=40):
        """ddx_operator_apply on any number of vectors (the entry point takes at most 64 at a time)."""
        X = np.asarray(X, dtype=np.float64)
        if X.shape[1] <= 64:
            return self.ctx.operator_apply(X, mode)
        return np.hstack([self.ctx.operator_apply(np.ascontiguousarray(X[:, j:j + block]), mode) for j in range(0, X.shape[1], block)])

    def _pca_arpack(self, n_components, seed):
        """pseudocount == 1 without scaling keeps the matrix sparse upstream and switches sc.tl.pca to
        svd_solver="arpack" (dd.py:296-297,308): an implicitly-centred truncated SVD converged to tolerance.  Upstream's
        ARPACK is a single-vector Lanczos on the smaller Gram operator (232 passes over the matrix per PCA at
        configs[1]); the device runs the block version of the same Krylov method behind one C-ABI call
        (``ddx_pca_exact_sparse``: one pair of 40-column products per step, full re-orthogonalisation, Rayleigh-Ritz of the
        small projected matrix), until the residual of every wanted pair is below 1e-5 of its eigenvalue -- the scores then sit
        within 1e-5 of ARPACK's (profiles/r04_block_lanczos.txt), the tolerance they are compared at being 1e-4.  The start block is drawn as the randomized PCA draws its own; the converged subspace does
        not depend on it.  Sign convention as in sklearn's PCA arpack branch (svd_flip(u_based_decision=False))."""
        c = self.ctx
        small = min(c.M, c.H)
        over = max(0, min(10, small - n_components))
        width = n_components + over
        self.lanczos_steps = 0
        if small <= 20 * width:
            # a Krylov space of 40-column blocks cannot grow far in so few dimensions (fewer than ~8 blocks leave the trailing
            # components unconverged): upstream's ARPACK is exact at every size, and so is the eigen-decomposition of the
            # (at most 800 x 800) Gram matrix
            self._pca_exact(n_components)
            return
        start = np.random.RandomState(seed).normal(size=(small, width))
        # residual 1e-6 of the eigenvalue (upstream: eigsh(tol=0), i.e. working precision; the scores are compared at 1e-4): convergence is
        # superlinear, a digit costs about two steps (profiles/r05_block_lanczos.txt)
        self.lanczos_steps = c.pca_exact_sparse(n_components, start, tol=1e-6, max_steps=48, n_oversamples=over)
        if not getattr(c, "lanczos_converged", True):
            # upstream's eigsh(tol=0) always converges.  Cheap and exact while the smaller side is small (the Gram matrix by ~small / 40
            # operator round trips + one dense eigh: under a second up to ~3 000); beyond that a second, longer Krylov run before giving up
            if small <= 3072:
                self._pca_exact(n_components)
                return
            self.lanczos_steps = c.pca_exact_sparse(n_components, start, tol=1e-6, max_steps=96, n_oversamples=over)
            if not getattr(c, "lanczos_converged", True):
                warnings.warn("truncated PCA of the sparse operator (pseudocount == 1): the block Lanczos solver stopped before its "
                              f"tolerance after {self.lanczos_steps} steps; trailing components may be off by more than 1e-5",
                              RuntimeWarning, stacklevel=2)

    def _pca_exact(self, n_components, block=40):
        """sklearn's exact regimes ("full" / "covariance_eigh"): eigen-decomposition of the smaller Gram
        matrix of the centred operator, built block-wise on the device; scores = A V (= U S), sign fixed on
        the component rows as svd_flip(u_based_decision=False) does."""
        c = self.ctx
        M, H = c.M, c.H
        small, mode = (H, 2) if H <= M else (M, 3)
        gram = np.empty((small, small))
        for j0 in range(0, small, block):
            n = min(block, small - j0)
            unit = np.zeros((small, n))
            unit[j0 + np.arange(n), np.arange(n)] = 1.0
            gram[:, j0:j0 + n] = c.operator_apply(unit, mode)
        gram = 0.5 * (gram + gram.T)
        evals, evecs = np.linalg.eigh(gram)
        top = np.argsort(evals)[::-1][:n_components]
        sing = np.sqrt(np.maximum(evals[top], 0.0))
        if H <= M:
            comps = evecs[:, top]                                   # H x C: right singular vectors
            scores = None
        else:
            left = evecs[:, top]                                    # M x C: left singular vectors
            comps = self._apply(left, 1) / np.where(sing > 0, sing, 1.0)
            scores = left * sing
        pick = np.argmax(np.abs(comps), axis=0)
        signs = np.sign(comps[pick, np.arange(comps.shape[1])])
        if scores is None:
            scores = self._apply(comps * signs, 0)
        else:
            scores = scores * signs
        c.set_embedding(scores.astype(np.float32))

    def timings(self):
        return self.ctx.timings()

    def timing_intervals(self):
        return self.ctx.timing_intervals() if self.timing else np.zeros((0, 2))

    def aug_nnz(self):
        return self.ctx.aug_nnz()

    def knn_window_fraction(self):
        return self.ctx.knn_window_fraction()

    def bitplane_stats(self):
        return self.ctx.bitplane_stats()


class BoostClassifier:
    """Classifier for doublets in single-cell RNA-seq data (GPU implementation).

    Parameters (identical meaning and defaults to the reference, ``doubletdetection.py:25-52``):
        boost_rate: Proportion of cell population size to produce as synthetic doublets.
        n_components: Number of principal components used for clustering.
        n_top_var_genes: Number of highest variance genes to use; all genes when zero.
        replace: If False, a cell is a synthetic doublet's parent at most once.
        clustering_algorithm: "louvain", "leiden" or "phenograph".
        clustering_kwargs: keyword arguments of the clustering algorithm (``prune`` defaults to True
            for phenograph; ``directed=False``, ``resolution=4`` for louvain / leiden).
        n_iters: Number of fit operations from which to collect p-values (default 10).
        normalizer: unsupported on the GPU path (see ``fit``); must stay ``None``.
        pseudocount: Pseudocount used in the default log-normalisation.
        random_state: Seeds PCA and the parent draws.
        verbose: Print progress messages.
        standard_scaling: Standard-scale the normalised matrix before PCA.
        n_jobs: host worker threads for community detection and scoring (-1: all cores; 1, the default, lets the library
            choose: min(32, cores)).

    Build-only keywords (after the reference's, so positional use is unaffected):
        device: GPU ordinal; default ``LOCAL_RANK`` under torch.distributed, else 0.
        devices: list of GPU ordinals driven by THIS process (one host thread and at least one device context per GPU;
            boosting iterations are dealt out over them).  Default: ``[device]``.
        streams_per_device: device contexts (HIP streams) per GPU, each running its own boosting iterations; the
            once-per-fit prologue is shared by device-to-device copies.  Default: up to 7 (one per iteration when the GPU has
            fewer to run), bounded by the GPU's memory (``DDX_STREAMS`` overrides).

    Attributes after ``fit`` / ``predict``: ``all_log_p_values_``, ``all_scores_``, ``communities_``,
    ``labels_``, ``parents_``, ``suggested_score_cutoff_``, ``synth_communities_``, ``top_var_genes_``,
    ``voting_average_`` -- shapes and dtypes as in the reference (``doubletdetection.py:54-71``).
    """

    _engine_factory = _HipEngine  # tests replace this to exercise host logic without a GPU

    def __init__(
        self,
        boost_rate: float = 0.25,
        n_components: int = 30,
        n_top_var_genes: int = 10000,
        replace: bool = False,
        clustering_algorithm: str = "phenograph",
        clustering_kwargs: dict | None = None,
        n_iters: int = 10,
        normalizer: Callable | None = None,
        pseudocount: float = 0.1,
        random_state: int = 0,
        verbose: bool = False,
        standard_scaling: bool = False,
        n_jobs: int = 1,
        *,
        device: int | None = None,
        devices: list | None = None,
        streams_per_device: int | None = None,
    ) -> None:
        if clustering_algorithm not in _ALGORITHMS:
            raise ValueError("Clustering algorithm needs to be one of ['louvain', 'phenograph', 'leiden']")
        self.clustering_algorithm = clustering_algorithm
        self.boost_rate = boost_rate
        self.replace = replace
        self.n_iters = n_iters
        self.normalizer = normalizer
        self.pseudocount = pseudocount
        self.random_state = random_state
        self.verbose = verbose
        self.standard_scaling = standard_scaling
        self.n_jobs = n_jobs
        self.device = device
        self.devices = None if devices is None else [int(d) for d in devices]
        self.streams_per_device = streams_per_device
        # one Generator for the lifetime of the object: a second fit() continues the stream, as upstream
        self.rng = np.random.default_rng(self.random_state)

        if self.clustering_algorithm == "leiden":
            warnings.warn("Leiden clustering is experimental and results have not been validated.")

        # an untouched n_components is silently capped by n_top_var_genes; negative n_top_var_genes -> 0
        untouched = n_components == 30 and n_top_var_genes > 0
        self.n_components = min(n_components, n_top_var_genes) if untouched else n_components
        self.n_top_var_genes = max(0, n_top_var_genes)

        self.clustering_kwargs = clustering_kwargs if isinstance(clustering_kwargs, dict) else {}
        self._set_clustering_kwargs()

        if not self.replace and self.boost_rate > 0.5:
            warnings.warn("boost_rate is trimmed to 0.5 when replace=False."
                          " Set replace=True to use greater boost rates.")
            self.boost_rate = 0.5

        assert self.n_top_var_genes == 0 or self.n_components <= self.n_top_var_genes, (
            "n_components={0} cannot be larger than n_top_var_genes={1}".format(n_components, n_top_var_genes))

    # ------------------------------------------------------------------------------------------
    def _set_clustering_kwargs(self) -> None:
        kw = self.clustering_kwargs
        if self.clustering_algorithm == "phenograph":
            kw.setdefault("prune", True)
            if self.n_iters == 1 and kw.get("prune") is True:
                warnings.warn("Using phenograph parameter prune=False is strongly recommended when "
                              "running only one iteration. Otherwise, expect many NaN labels.")
            unknown = set(kw) - _PHENOGRAPH_KW
        else:
            kw.setdefault("directed", False)
            kw.setdefault("resolution", 4)
            if "key_added" in kw:
                raise ValueError("'key_added' param cannot be overriden")
            if "random_state" in kw:
                raise ValueError("'random_state' param cannot be overriden. Please use classifier 'random_state'.")
            unknown = set(kw) - _SCANPY_KW
        if unknown:
            raise TypeError(f"unsupported clustering_kwargs for {self.clustering_algorithm}: {sorted(unknown)}")

    # Device limits of the hand-written kernels: the kNN kernels take at most 128 embedding
    # dimensions and 256 neighbours.  (The randomized sketch may have any width: beyond 64 columns the operator products
    # run block by block over the sketch.)
    _MAX_EMBED = 128
    _MAX_K = 256

    def _cluster_plan(self):
        """(k, include_self, graph_mode, gamma, seed, min_cluster_size, leiden, q_tol) for the chosen algorithm
        (q_tol: tolerance of PhenoGraph's best-of-restarts rule; None = one run).

        Every accepted keyword either changes the plan the way it changes the upstream call, is a documented no-op
        (worker counts, time limits, exact nearest-neighbour back-ends), or raises ``NotImplementedError`` -- a
        keyword is never swallowed while the result silently differs from the reference's."""
        kw = self.clustering_kwargs

        def unsupported(name, value, why):
            raise NotImplementedError(f"clustering_kwargs[{name!r}]={value!r} is not implemented on the GPU path: {why}")

        if self.clustering_algorithm == "phenograph":
            # phenograph.cluster(data, clustering_algo="louvain", k=30, directed=False, prune=False,
            #   min_cluster_size=10, jaccard=True, primary_metric="euclidean", n_jobs=-1, q_tol=1e-3,
            #   louvain_time_limit=2000, nn_method="kdtree", partition_type=None, resolution_parameter=1,
            #   n_iterations=-1, use_weights=True, seed=None)                        [dd.py:320-322]
            if not kw.get("jaccard", True):
                # upstream's alternative is a Gaussian kernel of the raw distances with sigma = 1: in a 30-dimensional
                # embedding the weights span tens of orders of magnitude, which the exact integer weight grid of the
                # community detection (2^-20) cannot hold
                unsupported("jaccard", False, "only the Jaccard graphs are built")
            if str(kw.get("primary_metric", "euclidean")).lower() not in _lib.Context.METRICS:
                unsupported("primary_metric", kw["primary_metric"], "implemented: euclidean, manhattan, cosine, correlation")
            if kw.get("nn_method", "kdtree") not in ("kdtree", "brute"):
                unsupported("nn_method", kw["nn_method"], "the device search is exact ('kdtree' and 'brute' give it)")
            algo = kw.get("clustering_algo", "louvain")
            if algo not in ("louvain", "leiden"):
                raise ValueError("clustering_algo needs to be one of ['louvain', 'leiden']")
            if kw.get("partition_type") is not None:
                unsupported("partition_type", kw["partition_type"], "only RBConfigurationVertexPartition (the default)")
            k = int(kw.get("k", 30))
            directed = bool(kw.get("directed", False))
            if directed and algo == "leiden":
                unsupported("directed", True, "leidenalg's directed modularity of the one-sided Jaccard graph is not restated")
            # directed=True skips upstream's symmetrisation (prune has no say then) and hands the edge i -> j of every
            # neighbour pair to the Louvain converter, which stores each edge in both directions and adds the weights
            # of duplicates: the undirected graph with J_ij once for one-sided and twice for mutual neighbours, i.e.
            # twice the averaged graph -- and modularity does not see a common factor.
            mode = 1 if (directed or not kw.get("prune")) else 0
            mcs = int(kw.get("min_cluster_size", 10))
            if algo == "leiden":
                # upstream: leidenalg on the Jaccard graph with resolution_parameter / seed / use_weights / n_iterations
                if kw.get("n_iterations", -1) != -1:
                    unsupported("n_iterations", kw["n_iterations"], "Leiden is iterated until stable (-1)")
                if not kw.get("use_weights", True):
                    unsupported("use_weights", False, "Leiden runs on the Jaccard weights")
                seed = kw.get("seed")
                seed = self.random_state if seed is None else seed
                return k, False, mode, float(kw.get("resolution_parameter", 1.0)), int(seed), mcs, True, None
            # clustering_algo="louvain": upstream's Louvain binaries take no resolution, no seed and always use the
            # weights -- resolution_parameter / seed / use_weights / n_iterations only reach leidenalg.  The
            # deterministic Louvain here is seeded from the classifier's random_state and, like upstream's runlouvain, re-run
            # from other node orders until 20 runs in a row fail to raise the modularity by more than q_tol.
            return k, False, mode, 1.0, int(self.random_state), mcs, False, float(kw.get("q_tol", 1e-3))

        # sc.tl.louvain(adata, resolution, random_state, restrict_to=None, key_added, adjacency=None,
        #   flavor="vtraag", directed=True, use_weights=False, partition_type=None, neighbors_key=None, obsp=None, copy)
        # sc.tl.leiden(adata, resolution, restrict_to=None, random_state, key_added, adjacency=None, directed,
        #   use_weights=True, n_iterations=-1, partition_type=None, neighbors_key=None, obsp=None, copy, flavor)
        # directed=True (scanpy's own default; the reference passes False, dd.py:415-416) turns the symmetric
        # connectivities into an igraph with both orientations of every edge: out- and in-strengths are equal and the
        # directed modularity is the undirected one term by term -- the same objective, so both values are accepted.
        for name in ("restrict_to", "adjacency", "neighbors_key", "obsp", "partition_type"):
            if kw.get(name) is not None:
                unsupported(name, kw[name], "the graph is the one sc.pp.neighbors builds on the PCA embedding")
        leiden = self.clustering_algorithm == "leiden"
        flavor = kw.get("flavor", "leidenalg" if leiden else "vtraag")
        if flavor != ("leidenalg" if leiden else "vtraag"):
            unsupported("flavor", flavor, "only the default flavor is restated")
        if leiden and kw.get("n_iterations", -1) != -1:
            unsupported("n_iterations", kw["n_iterations"], "Leiden is iterated until stable (-1)")
        if not leiden and "n_iterations" in kw:
            raise TypeError("louvain() got an unexpected keyword argument 'n_iterations'")
        # sc.tl.louvain ignores the edge weights unless use_weights=True; sc.tl.leiden uses the umap connectivities
        # unless use_weights=False
        weighted = bool(kw.get("use_weights", leiden))
        return 10, True, 3 if weighted else 2, float(kw["resolution"]), int(self.random_state), None, leiden, None

    def _knn_metric(self):
        """phenograph.cluster(primary_metric=...) (dd.py:320-322); scanpy's neighbours (dd.py:331-336) are euclidean."""
        if self.clustering_algorithm != "phenograph":
            return "euclidean"
        return str(self.clustering_kwargs.get("primary_metric", "euclidean")).lower()

    def _check_device_limits(self, num_cells, num_genes):
        """Fail before anything is uploaded when a request exceeds what the device kernels hold;
        the reference has no such limits, so say so instead of failing in the middle of a fit."""
        n_comp = self.n_components
        k = self._cluster_plan()[0]
        if k > self._MAX_K:
            raise NotImplementedError(f"k={k} nearest neighbours requested; the device kNN holds at most {self._MAX_K}")
        if n_comp > self._MAX_EMBED:
            raise NotImplementedError(f"n_components={n_comp}: the device kNN works on at most {self._MAX_EMBED} "
                                      "embedding dimensions")

    @staticmethod
    def _cluster_and_score(graph, gamma, seed, min_cluster_size, num_cells, leiden=False, q_tol=None, threads=1, sink=None):
        """Host C++ on a whole graph: Louvain (or Leiden), parts A + B + C -> size-sorted labels -> per-community
        hypergeometric test.  (The device route splits this: part A and C on the GPU around ``_part_b``.)"""
        import time

        t0 = time.perf_counter()
        graph = graph() if callable(graph) else graph
        t1 = time.perf_counter()
        if leiden:
            labels = _lib.leiden(*graph, gamma, seed)
        elif q_tol is not None:
            labels = _lib.louvain_best_of(*graph, gamma, seed, q_tol, threads=threads)[0]
        else:
            labels, _ = _lib.louvain(*graph, gamma, seed)
        t2 = time.perf_counter()
        full = _lib.relabel_by_size(labels, min_cluster_size)
        scores, logp = _lib.score_communities(full, num_cells)
        if sink is not None:
            sink(full, scores, logp)
        t3 = time.perf_counter()
        return full, scores, logp, (t1 - t0, t2 - t1, t3 - t2)

    @staticmethod
    def _part_b(coarse, gamma, seed, leiden=False, q_tol=None, threads=1):
        """Part B (or B') of the community detection on the coarse graph part A left: labels of the coarse nodes."""
        import time

        t0 = time.perf_counter()
        _, indptr, indices, weights = coarse
        if leiden:
            labels = _lib.leiden_sequential(indptr, indices, weights, gamma, seed)
        elif q_tol is not None:
            labels = _lib.louvain_best_of(indptr, indices, weights, gamma, seed, q_tol, threads=threads, presweeps=False)[0]
        else:
            labels = _lib.louvain_sequential(indptr, indices, weights, gamma, seed)[0]
        return labels, time.perf_counter() - t0

    @staticmethod
    def _score_labels(labels, min_cluster_size, num_cells, t_louvain, sink=None):
        import time

        t0 = time.perf_counter()
        full = _lib.relabel_by_size(labels, min_cluster_size)
        scores, logp = _lib.score_communities(full, num_cells)
        if sink is not None:
            sink(full, scores, logp)                 # the iteration's rows of the fitted attributes, written by this worker
        return full, scores, logp, (0.0, t_louvain, time.perf_counter() - t0)

    # ------------------------------------------------------------------------------------------
    def fit(self, raw_counts: NDArray | sp_sparse.csr_matrix) -> "BoostClassifier":
        """Fits the classifier on raw_counts (cells x genes).

        Sets ``all_scores_``, ``all_log_p_values_``, ``communities_``, ``top_var_genes_``,
        ``parents_``, ``synth_communities_`` and returns ``self``.
        """
        if self.normalizer is not None:
            # upstream's custom-normalizer branch cannot complete either (it references variables that
            # only the default branch defines, doubletdetection.py:301,372 -> UnboundLocalError)
            raise NotImplementedError("a user `normalizer` callable cannot run on the GPU path; leave it None")
        import time

        t_fit0 = time.perf_counter()
        rank, world, backend = _dist_info()
        if "DDX_UPLOAD_THREADS" not in os.environ:
            # every rank packs its own copy of the matrix for the upload: share the host cores (one node assumed).
            # Passed through the C-ABI (the pool is resized when the figure changes); the environment is not touched.
            # (a rank's share of the CPU time the node ALLOWS -- the pods show 256 CPUs and allow 16: eight ranks x 16 packing threads
            # would be throttled, not faster)
            _lib.set_upload_threads(max(2, min(16, int(_cpu_allowance()) // world)) if world > 1 else 0)
        self._upload_form_used = None
        _share_upload(rank, world, backend, one_leader=len(self._device_list(world)) == 1)
        staged = getattr(self, "_staged", None)
        drawer = ThreadPoolExecutor(max_workers=1)
        draws = None
        if staged is not None and staged[0] is raw_counts:
            _, csr, leaders, restrict = staged                        # counts already resident in HBM
            t_fit0 = time.perf_counter()
        else:
            self._drop_stage()
            # The host's random draws (ten parent draws without replacement: ~10 ms at the headline size) need nothing but the shape:
            # they start NOW, behind the validation and the upload (ctypes drops the GIL), instead of behind the device prologue
            # only -- the first iterations used to wait ~4 ms for them.  The reference draws nothing before check_array has passed
            # (dd.py:149-155 precede dd.py:394), so a fit that fails while staging puts the generator back where it was.
            shape = getattr(raw_counts, "shape", None)
            if shape is not None and len(shape) == 2 and shape[0] >= 1 and shape[1] >= 1:
                rng_state = self.rng.bit_generator.state
                g_early = self.n_top_var_genes if 0 < self.n_top_var_genes < shape[1] else shape[1]
                draws = drawer.submit(self._draw, int(shape[0]), int(g_early))
            try:
                csr, leaders, restrict = self._stage(raw_counts, rank, world)
            except BaseException:
                if draws is not None:
                    try:
                        draws.result()
                    except Exception:
                        pass
                    self.rng.bit_generator.state = rng_state
                drawer.shutdown(wait=True)
                raise
        self._staged = None
        t_staged = time.perf_counter()
        lead_ctx = getattr(next(iter(leaders.values())), "ctx", None)
        if lead_ctx is not None and hasattr(lead_ctx, "upload_form") and restrict:
            self._upload_form_used = lead_ctx.upload_form()     # 0 plain arrays, 1 packed here, 2 another context's / the node's packed image
        num_cells = csr.shape[0]
        num_genes = self.n_top_var_genes if restrict else csr.shape[1]
        if draws is not None and (int(shape[0]), int(g_early)) != (num_cells, num_genes):      # (cannot happen: the staged matrix has the input's shape)
            draws.result()
            self.rng.bit_generator.state = rng_state
            draws = None
        if draws is None:
            draws = drawer.submit(self._draw, num_cells, num_genes)     # overlaps the device prologue
        lanes = []
        ok = False
        try:
            devs = list(leaders)
            if restrict:
                # dd.py:165-176 -- float32 variances on the device in scipy's evaluation order (identical on every GPU, so
                # one computes them); the ordering itself stays numpy's argsort so ties fall exactly as they do upstream
                gene_variances = leaders[devs[0]].gene_variances()
                self.top_var_genes_ = np.argsort(gene_variances)[-self.n_top_var_genes:]
                self._on_each(leaders.values(), lambda e: e.select_columns(self.top_var_genes_))
            n_mine = len([i for i in range(self.n_iters) if i % world == rank])
            lanes = self._open_lanes(leaders, n_mine)
            t_prologue = time.perf_counter() - t_staged
            self._fit_resident(lanes, num_cells, num_genes, rank, world, backend, draws)
            self._host_timings["prologue"] = t_prologue
            self._host_timings["stage"] = t_staged - t_fit0
            ok = True
        finally:
            failed = not ok        # (sys.exc_info() would also be set when a caller's own `except` block runs this fit)
            t0 = time.perf_counter()
            try:
                draws.result()             # never leave the Generator in use by a worker
            except Exception:
                pass
            drawer.shutdown(wait=True)
            for e in {id(e): e for e in list(leaders.values()) + [ln[1] for ln in lanes]}.values():
                if failed:
                    self._discard(e)
                else:
                    e.close()
            if failed:
                # the workers fill these row by row while the fit runs: a fit that failed leaves none of them half-written
                for name in ("all_scores_", "all_log_p_values_", "communities_", "synth_communities_"):
                    self.__dict__.pop(name, None)
        self._host_timings["close"] = time.perf_counter() - t0
        self._host_timings["fit_total"] = time.perf_counter() - t_fit0
        return self

    @staticmethod
    def _is_plain_csr(x):
        """A scipy CSR matrix that check_array(accept_sparse="csr", dtype="float32") would hand back untouched."""
        return (sp_sparse.issparse(x) and x.format == "csr" and x.dtype == np.float32 and x.ndim == 2
                and x.indices.dtype == np.int32 and x.shape[0] >= 1 and x.shape[1] >= 1)

    def _coerce(self, raw_counts):
        """dd.py:149-160: float32 CSR from an ndarray or any sparse matrix (finite, 2-D).

        Returns (csr, validated).  A float32 CSR is what check_array would return unchanged after reading every value
        once on the host (finite check); that pass and the canonical-form check run on the device instead, on the
        uploaded arrays (ddx_upload_raw / ddx_upload_counts validate what they receive) -- ``validated`` is False then
        and ``_stage`` falls back to this host path if the device objects."""
        if self._is_plain_csr(raw_counts):
            return raw_counts, False
        return self._coerce_on_host(raw_counts), True

    def _coerce_on_host(self, raw_counts):
        from sklearn.utils import check_array

        raw_counts = check_array(raw_counts, accept_sparse="csr", ensure_all_finite=True, ensure_2d=True,
                                 dtype="float32")
        if not sp_sparse.issparse(raw_counts):
            if self.verbose:
                print("Sparsifying matrix.")
            raw_counts = sp_sparse.csr_matrix(raw_counts)
        if not raw_counts.has_canonical_format:
            # the device merges assume sorted, duplicate-free rows; scipy's own row indexing / addition
            # (dd.py:174-176,397-399) canonicalises on the way, so doing it here changes nothing downstream
            raw_counts = raw_counts.copy()
            raw_counts.sum_duplicates()
        return raw_counts

    # ---- devices, streams, lanes ---------------------------------------------------------------------------------
    def _device_list(self, world):
        if self.devices is not None:
            if not self.devices:
                raise ValueError("devices must name at least one GPU")
            return list(dict.fromkeys(self.devices))
        if self.device is not None:
            return [int(self.device)]
        return [int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0]

    def _host_threads(self, world: int = 1):
        """Host threads for community detection + scoring (they run while the GPU works on the next iteration).
        n_jobs > 1: that many; n_jobs <= 0: every core; n_jobs == 1 (the reference's default, where it only sizes
        PhenoGraph's process pool) leaves the choice to the library: min(32, cores).  DDX_HOST_THREADS overrides.
        With one process per GPU (``world`` ranks on this node, dd.py:192-198 sharded) the ranks share the host: every
        automatic figure is a rank's share, cores // world -- 8 ranks x 5 contexts x 20 restart threads would otherwise
        be 800 runnable threads on one node."""
        env = os.environ.get("DDX_HOST_THREADS")
        if env:
            return max(1, int(env))
        cores = max(1, int(min(os.cpu_count() or 1, _cpu_allowance() if world > 1 else (os.cpu_count() or 1))) // max(1, int(world)))
        if self.n_jobs is None or self.n_jobs <= 0:
            return cores
        if self.n_jobs == 1:
            return min(32, cores)
        return max(1, min(int(self.n_jobs), cores) if world > 1 else int(self.n_jobs))

    _AUTO_STREAMS = 7

    def _stream_count(self, n_mine=None, n_devices=1, leader=None):
        """Device contexts per GPU.  An explicit ``streams_per_device`` / ``DDX_STREAMS`` is taken as it is.  Otherwise: one per
        iteration the GPU has to run, up to seven, and no more than the GPU's memory and the parking allowance hold beside the
        leader.  Measured at the headline shape at the end of round 5 (ms per fit, same box, alternating): 10 iterations on
        5 / 6 / 7 / 8 contexts 120 / 122 / 114 / 118, 25 iterations on 5 / 7 / 8 contexts 282 / 263 / 269, 6 iterations on 3 / 6
        contexts 84 / 82 -- the iterations of a round run in step (all in their PCA, then all in their kNN); uneven shares
        (7 contexts: 3 x 2 + 4 x 1 iterations) pull them apart, and kernels that do not compete for the same unit overlap
        better.  (Rounds 3-4 evened the shares out -- 10 iterations: 5 x 2 -- when a product kernel filled every CU's LDS and
        the contexts took turns at the PCA anyway.)"""
        n = self.streams_per_device
        if n is None and "DDX_STREAMS" in os.environ:
            n = int(os.environ["DDX_STREAMS"])
        if n is not None:
            if n < 1:
                raise ValueError("streams_per_device must be at least 1")
            return int(n)
        if n_mine is None:
            return self._AUTO_STREAMS
        per_device = max(1, -(-int(n_mine) // max(1, n_devices)))          # iterations a GPU has to run
        n = min(self._AUTO_STREAMS, per_device)
        ctx = getattr(leader, "ctx", None)
        if ctx is not None and hasattr(ctx, "device_memory"):
            held = max(1, ctx.device_bytes())
            free, total = ctx.device_memory()
            # a follower ends up as large as its leader (the leader's first chunk is sized for a whole fit); parked
            # contexts of earlier fits are handed out again before anything new is allocated, so they count as room
            pool = [c.device_bytes() for c in _CONTEXT_POOL.get(getattr(leader, "device", None), ())]
            parked = sum(pool)
            # a follower holds the restricted counts only (no raw matrix, no HVG temporaries): what its first chunk will be,
            # or what the followers of the previous fit grew to (the parked contexts other than the largest)
            fol = held
            estimate = getattr(ctx, "follower_bytes", None)
            if estimate is not None:
                est = int(estimate())
                if est > 0:
                    fol = max(1, min(held, max([est] + sorted(pool)[:-1])))
            room = int(0.9 * (free + parked))
            n = max(1, min(n, 1 + room // fol))
            if _keep_contexts():
                # what a fit allocates should also fit the allowance of the parked contexts: a context trimmed at the end
                # of every fit obtains its memory from the driver again at the start of the next (seconds at this size)
                n = max(1, min(n, 1 + max(0, _park_limit_bytes(total) - held) // fol))
        return int(n)

    @staticmethod
    def _on_each(items, fn):
        """fn(item) for every item, on one host thread per item when there are several (ctypes calls drop the GIL, so
        the GPUs of a node upload / compute side by side); the first exception is re-raised."""
        items = list(items)
        if len(items) <= 1:
            return [fn(it) for it in items]
        with ThreadPoolExecutor(max_workers=len(items)) as pool:
            return [f.result() for f in [pool.submit(fn, it) for it in items]]

    def _stage(self, raw_counts, rank, world):
        """Validate the input and make it resident on every GPU of this process: one leader context per GPU."""
        csr, validated = self._coerce(raw_counts)
        restrict = 0 < self.n_top_var_genes < csr.shape[1]
        self._check_device_limits(csr.shape[0], self.n_top_var_genes if restrict else csr.shape[1])
        leaders = {}
        try:
            for dev in self._device_list(world):
                leaders[dev] = self._fit_switches(self._engine_factory(dev))
            put = (lambda e: e.stage_raw(csr)) if restrict else (lambda e: e.upload(csr))
            try:
                self._on_each(leaders.values(), put)
            except _lib.DdxError as err:
                if validated or err.code != _lib.E_ARG:
                    raise
                # the device found a non-finite value or a non-canonical row in a matrix the host had not read: let
                # check_array raise its ValueError (dd.py:149-155), or canonicalise and upload again
                csr = self._coerce_on_host(raw_counts)
                put = (lambda e: e.stage_raw(csr)) if restrict else (lambda e: e.upload(csr))
                self._on_each(leaders.values(), put)
        except Exception:
            for e in leaders.values():
                self._discard(e)
            raise
        return csr, leaders, restrict

    def _fit_switches(self, engine, world=None):
        """Switches of a device context that follow from this fit's circumstances.  (The bit planes of the operator products need
        none since round 6: scaled matrices take that route -- 1 / sd_j is a diagonal factor, ddx_scale -- and sketches wider than
        40 columns run their products in 40-column blocks on it, ddx_pca.)  How the lane threads wait for the GPU: spinning (the
        runtime's way, lowest latency) keeps one CPU busy per waiting thread; with several ranks on one node, or fewer CPUs allowed
        than lanes, they sleep between polls instead (option host_wait=block, ddx.h) -- unless the caller chose (DDX_OPTIONS / OPTIONS)."""
        ctx = getattr(engine, "ctx", None)
        if ctx is not None and hasattr(ctx, "set_option") and "host_wait" not in _lib.current_options():
            if world is None:
                world = _dist_info()[1]
            lanes = self.streams_per_device or self._AUTO_STREAMS
            ctx.set_option("host_wait", "block" if (world > 1 or _cpu_allowance() < lanes * world + 1) else "spin")
        return engine

    def _open_lanes(self, leaders, n_mine):
        """[(device, engine)]: the leader context of every GPU plus streams_per_device - 1 followers that copy its resident
        counts device-to-device.  Lanes are ordered stream-major so that a short job reaches every GPU first."""
        streams = self._stream_count(n_mine, len(leaders), next(iter(leaders.values())))
        needed = max(1, n_mine)                  # a lane without an iteration to run is not opened
        lanes = [(dev, eng) for dev, eng in leaders.items()]
        followers = []
        for k in range(1, streams):
            for dev, eng in leaders.items():
                if len(lanes) + len(followers) >= needed:
                    break
                followers.append((dev, eng))
        made = []
        try:
            for dev, leader in followers:
                f = self._fit_switches(self._engine_factory(dev))
                # The copy itself waits until the follower's own host thread starts (`_fit_resident`: the first thing a lane
                # does): what it reads of the leader -- the original cells' rows, their mirror, their bit planes -- is
                # constant for the fit, so the leader is already on its first iteration while its followers copy (round 4
                # copied here, with the leader idle: 4 x 3.5 GB device to device before any iteration began).  The library's side of
                # that contract is ddx.h's ddx_clone_counts: the view a leader publishes stays valid and constant for the fit.
                # With standard_scaling the leader's FIRST scaling may still move columns out of the bitmaps (the ones whose entries
                # could reach the clip): its followers copy after that (`_fit_resident`), or each would repeat the rebuild.
                f._clone_source = leader
                made.append((dev, f))
        except Exception:
            for _, f in made:
                self._discard(f)
            raise
        return lanes + made

    @staticmethod
    def _discard(engine):
        """Close an engine after a failure: the HIP engine destroys its context instead of parking it."""
        try:
            engine.close(discard=True)
        except TypeError:                 # test engines without the keyword
            engine.close()

    def _drop_stage(self):
        staged = getattr(self, "_staged", None)
        if staged is not None:
            for e in staged[2].values():
                e.close()
        self._staged = None

    def stage(self, raw_counts) -> "BoostClassifier":
        """Build-only extension: validate ``raw_counts`` and make them resident in HBM ahead of
        ``fit(raw_counts)`` (same object), so that a caller can keep the PCIe upload out of a timed
        region.  ``fit`` on any other object simply stages that object itself."""
        rank, world, _ = _dist_info()
        self._drop_stage()
        csr, leaders, restrict = self._stage(raw_counts, rank, world)
        self._staged = (raw_counts, csr, leaders, restrict)
        return self

    def _draw(self, num_cells, num_genes):
        """Host-side random draws of one fit: the parents of every iteration (dd.py:394; the Generator stream must be
        consumed in iteration order whatever rank runs the iteration) and, for the randomized PCA regime, sklearn's
        start matrix.  fit() runs this on a worker thread while the device executes the prologue."""
        num_synths = int(self.boost_rate * num_cells)
        all_parents = [self.rng.choice(num_cells, size=(num_synths, 2), replace=self.replace)
                       for _ in range(self.n_iters)]
        M = num_cells + num_synths
        q0 = None
        sparse_branch = self.pseudocount == 1 and not self.standard_scaling
        if not sparse_branch and self._pca_regime(M, num_genes, self.n_components) == "randomized":
            sketch = self.n_components + 10
            q0_rows = num_genes if M >= num_genes else M
            # sklearn draws the start matrix from the legacy RandomState and casts it to the data dtype
            q0 = np.random.RandomState(self.random_state).normal(size=(q0_rows, sketch))
            q0 = q0.astype(np.float32).astype(np.float64)
        return all_parents, q0

    def _fit_resident(self, lanes, num_cells, num_genes, rank, world, backend, draws=None):
        import threading
        import time

        self._num_cells, self._num_genes = num_cells, num_genes
        num_synths = int(self.boost_rate * num_cells)
        n_iters = self.n_iters

        t_setup0 = time.perf_counter()
        all_parents, q0_drawn = draws.result() if draws is not None else self._draw(num_cells, num_genes)

        M = num_cells + num_synths
        n_comp = self.n_components
        sparse_branch = self.pseudocount == 1 and not self.standard_scaling      # dd.py:296-297,308
        regime = "arpack" if sparse_branch else self._pca_regime(M, self._num_genes, n_comp)
        if regime == "arpack":
            if not 1 <= n_comp < min(M, self._num_genes):
                raise ValueError(f"n_components={n_comp} must be strictly less than min(n_samples, n_features)="
                                 f"{min(M, self._num_genes)} with svd_solver='arpack'")
            q0 = "arpack"
            for _, eng in lanes:
                eng._arpack_seed = self.random_state
        elif regime == "randomized":
            q0 = q0_drawn
        else:
            q0 = None      # exact regime: no random start (engine builds and diagonalises the Gram matrix)

        knn_k, include_self, graph_mode, gamma, seed, min_cluster_size, leiden, q_tol = self._cluster_plan()
        metric = self._knn_metric()

        # iteration i belongs to rank i % world (one process per GPU under torch.distributed); inside this process the
        # rank's iterations are dealt out over the lanes (GPUs x streams)
        mine = [i for i in range(n_iters) if i % world == rank]
        lanes = lanes[:max(1, len(mine))]
        share = [mine[k::len(lanes)] for k in range(len(lanes))]
        # A/B switch: DDX_PCA_LOCK=1 makes the PCA stages of one GPU take turns (their operator products fill every CU by
        # themselves).  Measured at the headline size: 228.5 ms per fit with the lock, 225 ms without (3 contexts: 228.7 /
        # 222.5) -- the hardware scheduler interleaves the streams at least as well, so the default is no lock.
        use_lock = os.environ.get("DDX_PCA_LOCK", "0") not in ("", "0")
        pca_locks = {dev: (threading.Lock() if use_lock and sum(1 for d, _ in lanes if d == dev) > 1 else None) for dev, _ in lanes}
        workers = self._host_threads(world)
        self._host_threads_used = workers
        # host threads one clustering job may use for its batch of restarts (the jobs of different iterations overlap)
        restart_threads = max(1, min(20, workers // max(1, min(workers, len(mine)))))
        self._helper_budget = None
        if hasattr(_lib, "set_helper_threads"):
            # Several ranks on one node: the restart batches of a rank's iterations share a budget of helper threads -- the rank's share of
            # the CPU time the node allows (cgroup quota: 16 CPUs on pods that show 256; its lane threads sleep while they wait, host_wait =
            # block) -- and a job may ask for a full batch of 20: it gets what is free, all of it when it runs alone (the last iteration of a
            # fit, on the critical path).  One rank: no budget.  Measured at the headline on such a pod (profiles/r06_host_budget.txt): a
            # budget of 8 helpers beside the seven spinning lanes made the fit 5 ms SLOWER than 7 x 20 oversubscribed threads -- part B of
            # iteration i has to be back before the lane reaches iteration i + 1's refinement, and throttling costs less than waiting for it.
            budget = -1
            if world > 1 and "DDX_HOST_THREADS" not in os.environ:
                budget = max(0, int(_cpu_allowance() // world) - 1)
                restart_threads = 20
                self._helper_budget = budget
            _lib.set_helper_threads(budget)
        # (the most threads one clustering job can have running: its own + the whole budget)
        self._restart_threads_used = restart_threads if self._helper_budget is None else min(restart_threads, self._helper_budget + 1)
        local = {}
        # Without a process group every iteration runs here: the worker that scores an iteration writes its rows of the
        # fitted attributes itself (behind the GPU's work on the other iterations) instead of leaving 30 MB of copies to the
        # end of fit().  With one, the rows travel through the gather first (_gather_rows).
        direct = backend is None
        if direct:
            self.all_scores_ = np.empty((n_iters, num_cells))
            self.all_log_p_values_ = np.empty((n_iters, num_cells))
            all_communities = np.empty((n_iters, num_cells))
            all_synth_communities = np.empty((n_iters, num_synths))

        def sink_for(i):
            if not direct:
                return None

            def sink(full, scores, logp):
                self.all_scores_[i] = scores
                self.all_log_p_values_[i] = logp
                all_communities[i] = full[:num_cells]
                all_synth_communities[i] = full[num_cells:]
            return sink

        # per-fit structures of a leader are final once its first scaling has run (no scaling: from the start)
        structures_final = {id(eng): threading.Event() for _, eng in lanes if getattr(eng, "_clone_source", None) is None}
        if not self.standard_scaling:
            for ev in structures_final.values():
                ev.set()
        host = {"draws": time.perf_counter() - t_setup0, "device_stages": 0.0, "wait_workers": 0.0, "graph_assembly": 0.0, "louvain": 0.0, "score": 0.0}
        t_dev0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=max(1, min(workers, max(1, len(mine))))) as pool:
            pending = {}

            def drive(k):
                """The iterations of lane k.  The sequential part B of iteration i runs on a host worker while the GPU
                already works on iteration i + 1 (doublets ... PCA); part C of iteration i (refinement on the device,
                needs B's labels) is slotted in behind that PCA, before the next graph overwrites the previous one."""
                dev, engine = lanes[k]
                src = getattr(engine, "_clone_source", None)
                final = structures_final.get(id(engine))
                if src is not None:                  # a follower: take over the leader's resident counts first (_open_lanes)
                    engine._clone_source = None
                    ev = structures_final.get(id(src))
                    if ev is not None:
                        ev.wait()
                    engine.clone_from(src)
                try:
                    iterate(k, dev, engine, final)
                finally:
                    if final is not None:
                        final.set()              # (a leader that failed, or ran on a test engine, must not leave its followers waiting)

            def iterate(k, dev, engine, final):
                kw = {"verbose": True} if self.verbose else {}
                kw2 = dict(kw)
                if metric != "euclidean":
                    kw2["metric"] = metric
                split = hasattr(engine, "first_half") and hasattr(engine, "refine")
                waiting = None                   # (iteration, future of its part B)

                def finish(item):
                    i, fut = item
                    coarse_labels, t_b = fut.result()
                    labels = engine.refine(coarse_labels, gamma)
                    pending[i] = pool.submit(self._score_labels, labels, min_cluster_size, num_cells, t_b, sink_for(i))

                for i in share[k]:
                    if self.verbose:
                        print("Iteration {:3}/{}".format(i + 1, n_iters))
                    if split:
                        kw1 = kw
                        if final is not None and not final.is_set() and isinstance(engine, _HipEngine):
                            kw1 = dict(kw, after_scale=final.set)
                        engine.first_half(all_parents[i], self.pseudocount, self.standard_scaling, n_comp, q0, pca_locks[dev], **kw1)
                        if waiting is not None:
                            finish(waiting)
                            waiting = None
                        graph = engine.second_half(knn_k, include_self, graph_mode, gamma, **kw2)
                    else:
                        graph = engine.run_iteration(all_parents[i], self.pseudocount, self.standard_scaling, n_comp,
                                                     q0, knn_k, include_self, graph_mode, gamma, pca_locks[dev], **kw2)
                    if split and len(graph) == 4:
                        waiting = (i, pool.submit(self._part_b, graph, gamma, seed, leiden, q_tol, restart_threads))
                    elif len(graph) == 4:
                        # engine contract: a 3-tuple is the whole symmetric graph (indptr, indices, weights); a 4-tuple (member,
                        # coarse indptr, indices, weights) is the result of the device pre-sweeps and needs `refine` for part C
                        raise TypeError("an engine that returns a pre-coarsened graph must also provide first_half / second_half / refine")
                    else:
                        pending[i] = pool.submit(self._cluster_and_score, graph, gamma, seed, min_cluster_size, num_cells, leiden,
                                                 q_tol, restart_threads, sink_for(i))
                if waiting is not None:
                    finish(waiting)

            self._on_each(range(len(lanes)), drive)
            host["device_stages"] = time.perf_counter() - t_dev0
            t0 = time.perf_counter()
            for i in mine:
                full, scores, logp, (tg, tl, ts) = pending[i].result()
                local[i] = (full, scores, logp)
                host["graph_assembly"] += tg
                host["louvain"] += tl
                host["score"] += ts
            host["wait_workers"] = time.perf_counter() - t0
        self._host_timings = host
        self._lanes_used = len(lanes)
        self._device_timings = {}
        for _, engine in lanes:
            for name, (launches, ms) in (engine.timings() if hasattr(engine, "timings") else {}).items():
                a = self._device_timings.get(name, (0, 0.0))
                self._device_timings[name] = (a[0] + launches, a[1] + ms)
        # share of the wall-clock during which at least one kernel scope of this process was running on a GPU (bench.py)
        self._device_busy_ms = None
        spans = [e.timing_intervals() for _, e in lanes if hasattr(e, "timing_intervals")]
        spans = np.concatenate(spans) if spans else np.zeros((0, 2))
        if len(spans) and len({d for d, _ in lanes}) == 1:
            spans = spans[np.argsort(spans[:, 0])]
            busy, cur_lo, cur_hi = 0.0, spans[0, 0], spans[0, 1]
            for lo, hi in spans[1:]:
                if lo > cur_hi:
                    busy += cur_hi - cur_lo
                    cur_lo, cur_hi = lo, hi
                else:
                    cur_hi = max(cur_hi, hi)
            self._device_busy_ms = busy + (cur_hi - cur_lo)
        lead = lanes[0][1]
        if mine and hasattr(lead, "aug_nnz") and getattr(lead, "timing", False):
            # stored entries of the last augmented matrix (a diagnostic for bench.py's byte models; asking makes the device write the
            # merged rows of the last iteration's doublets, which the bit-plane route otherwise never does -- so only when profiling)
            self._last_nnz_aug = lead.aug_nnz()
        if mine and hasattr(lead, "knn_window_fraction"):
            self._last_knn_window = lead.knn_window_fraction()   # share of the tile pairs the last kNN screened
            self._last_bitplane = lead.bitplane_stats() if hasattr(lead, "bitplane_stats") else None

        t_asm0 = time.perf_counter()
        if not direct:
            self.all_scores_ = np.zeros((n_iters, num_cells))
            self.all_log_p_values_ = np.zeros((n_iters, num_cells))
            all_communities = np.zeros((n_iters, num_cells))
            all_synth_communities = np.zeros((n_iters, num_synths))
        rows = self._gather_rows(local, mine, n_iters, num_cells, num_synths, rank, world, backend, lanes[0][0])
        for i in range(n_iters):
            full, scores, logp = rows[i]
            if not direct:
                self.all_scores_[i] = scores
                self.all_log_p_values_[i] = logp
                all_communities[i] = full[:num_cells]
                all_synth_communities[i] = full[num_cells:]
            if self.verbose:
                sizes = np.unique(full, return_counts=True)[1].tolist()
                print("Found clusters [{0}, ... {2}], with sizes: {1}\n".format(full.min(), sizes, full.max()))
        self.communities_ = all_communities
        self.synth_communities_ = all_synth_communities
        self._parent_arrays = all_parents
        self._parents_lists = None
        self._host_timings["assemble"] = time.perf_counter() - t_asm0
        return self

    @property
    def parents_(self):
        """Parent cells' indexes for each synthetic doublet, one entry per iteration, in the reference's
        format (list of lists of ``[np.int64, np.int64]``, dd.py:395).  The nested lists are built on
        first access (the int64 [S,2] arrays are what ``fit`` keeps)."""
        if getattr(self, "_parents_lists", None) is None:
            if getattr(self, "_parent_arrays", None) is None:
                raise AttributeError("parents_ is set by fit()")
            self._parents_lists = [[list(p) for p in choices] for choices in self._parent_arrays]
        return self._parents_lists

    @parents_.setter
    def parents_(self, value):
        self._parents_lists = value

    @staticmethod
    def _pca_regime(M, H, n_comp):
        """sklearn's svd_solver='auto' policy for a dense M x H array (sklearn/decomposition/_pca.py:524-536)."""
        if not 1 <= n_comp <= min(M, H):
            raise ValueError(f"n_components={n_comp} must be between 1 and min(n_samples, n_features)={min(M, H)}")
        if H <= 1000 and M >= 10 * H:
            regime = "covariance_eigh"
        elif max(M, H) <= 500:
            regime = "full"
        elif n_comp < 0.8 * min(M, H):
            regime = "randomized"
        else:
            regime = "full"
        # (the exact regimes build the smaller Gram matrix from operator products on the device and diagonalise it on the
        # host: any size works; beyond a few thousand rows / columns the host eigen-decomposition dominates)
        return regime

    def _gather_rows(self, local, mine, n_iters, num_cells, num_synths, rank, world, backend, device):
        """Single collective: every rank contributes the result rows of its iterations, packed as bytes --
        [communities int32 x M | scores float64 x N | log p float64 x N], i.e. 4 M + 16 N = 21 N bytes per iteration at
        boost_rate 0.25.  It runs whenever a torch.distributed process group is up (also a one-rank group: same code
        path), never otherwise."""
        if backend is None:
            return {i: local[i] for i in range(n_iters)}
        import torch
        import torch.distributed as dist

        M = num_cells + num_synths
        off_s = (4 * M + 7) & ~7                      # float64 fields start 8-byte aligned
        width = off_s + 16 * num_cells
        per_rank = (n_iters + world - 1) // world
        buf = np.zeros((per_rank, width), dtype=np.uint8)
        for slot, i in enumerate(mine):
            full, scores, logp = local[i]
            buf[slot, :4 * M] = np.ascontiguousarray(full, dtype=np.int32).view(np.uint8)
            buf[slot, off_s:off_s + 8 * num_cells] = np.ascontiguousarray(scores, dtype=np.float64).view(np.uint8)
            buf[slot, off_s + 8 * num_cells:] = np.ascontiguousarray(logp, dtype=np.float64).view(np.uint8)
        use_cuda = "nccl" in str(backend)        # RCCL needs device tensors ("nccl", or "cpu:gloo,cuda:nccl")
        t = torch.from_numpy(buf)
        if use_cuda:
            t = t.to(f"cuda:{device}")
        out = torch.empty((world * per_rank, width), dtype=torch.uint8, device=t.device)
        dist.all_gather_into_tensor(out, t)          # the one collective of the path (RCCL under nccl)
        out = out.cpu().numpy().reshape(world, per_rank, width)
        rows = {}
        for i in range(n_iters):
            row = np.ascontiguousarray(out[i % world, i // world])
            rows[i] = (row[:4 * M].view(np.int32).astype(np.int64),
                       row[off_s:off_s + 8 * num_cells].view(np.float64).copy(),
                       row[off_s + 8 * num_cells:].view(np.float64).copy())
        return rows

    # ------------------------------------------------------------------------------------------
    def predict(self, p_thresh: float = 1e-7, voter_thresh: float = 0.9) -> NDArray:
        """Doublet calls: 0 singlet, 1 doublet, NaN unscored.

        n_iters > 1: a cell is called when the fraction of iterations with log p <= log(p_thresh)
        reaches voter_thresh (non-finite log p-values are masked out of the vote); sets ``labels_`` and
        ``voting_average_``.  n_iters == 1: scores are cut at the largest gap between consecutive
        distinct scores; sets ``labels_`` (bool) and ``suggested_score_cutoff_``.
        """
        if self.n_iters > 1:
            cut = np.log(p_thresh)
            with np.errstate(invalid="ignore"):
                valid_logp = np.ma.masked_invalid(self.all_log_p_values_)
                vote = np.mean(valid_logp <= cut, axis=0)
                called = (vote >= voter_thresh).astype(float)
            self.labels_ = np.ma.filled(called, np.nan)
            self.voting_average_ = np.ma.filled(vote, np.nan)
            return self.labels_
        scored = ~np.isnan(self.all_scores_)
        candidates = np.unique(self.all_scores_[scored])
        where = 0
        if candidates.size > 1:
            where = int(np.argmax(np.diff(candidates))) + 1
        self.suggested_score_cutoff_ = candidates[where]
        with np.errstate(invalid="ignore"):
            self.labels_ = self.all_scores_[0, :] >= self.suggested_score_cutoff_
        self.labels_[~scored[0, :]] = np.nan
        return self.labels_

    def doublet_score(self) -> NDArray:
        """Average negative log p-value over iterations (higher = more doublet-like)."""
        if self.n_iters > 1:
            with np.errstate(invalid="ignore"):
                return -np.mean(np.ma.masked_invalid(self.all_log_p_values_), axis=0)
        return -self.all_log_p_values_[0]
