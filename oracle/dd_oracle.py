"""TEST INFRASTRUCTURE -- CPU oracle for the BoostClassifier.fit() hot path.

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product (``doubletdetection_amd``) never
imports anything under ``oracle/`` and fails loudly if its HIP library is missing.

It restates, stage by stage, what the reference (``/root/reference/doubletdetection/
doubletdetection.py``, abbreviated ``dd.py`` below) computes, using the same numpy / scipy /
scikit-learn calls the reference (or the absent scanpy / phenograph packages it delegates to) makes.
Each function cites the reference lines it follows.

Parity status
-------------
* Stages a2-a7, a12, a13 (coercion, HVG, memoisation, doublet creation, log-normalisation, community
  bookkeeping + hypergeometric test, predict, doublet_score) are **pinned**: ``tests/golden/*.npz``
  holds outputs of the reference's own lines, produced by ``oracle/make_golden.py`` through
  ``oracle/ref_harness.py`` (reference executed verbatim with stand-in third-party modules), and
  ``tests/test_oracle_golden.py`` checks this file against them.
* a9 (PCA) is pinned through scikit-learn, which is what scanpy calls: the golden vectors hold
  ``PCA(n_components, svd_solver="auto", random_state).fit_transform`` outputs in f32 and f64, and
  ``randomized_pca_f64`` below is checked against them.
* a10/a11 kNN are pinned through scikit-learn ``NearestNeighbors`` (what phenograph / scanpy call for
  exact search).
* a8 (``sc.pp.scale``), the Jaccard/prune graph, the umap connectivities and the community detection
  itself are **parity unpinned**: scanpy, phenograph, louvain, leidenalg are absent from this image and
  from /root/reference, PhenoGraph's Louvain is not even self-deterministic (dd.py:47-48), and the
  reference's tests assert nothing about them.  They are restated from the upstream packages'
  published behaviour; the deterministic community detection is specified in ``oracle/louvain_ref.py``.
"""
from __future__ import annotations

import collections

import numpy as np
import scipy.sparse as sp
from scipy.stats import hypergeom

from . import louvain_ref


# --------------------------------------------------------------------------------------------------
# a2: input coercion (dd.py:149-160)
# --------------------------------------------------------------------------------------------------
def coerce_counts(raw_counts) -> sp.csr_matrix:
    """float32 CSR from ndarray / any sparse matrix.  dd.py:149-160.

    ``check_array(accept_sparse="csr", dtype="float32", ensure_all_finite, ensure_2d)`` then
    ``csr_matrix(dense)`` for dense input (which drops zeros).
    """
    from sklearn.utils import check_array

    x = check_array(raw_counts, accept_sparse="csr", ensure_all_finite=True, ensure_2d=True,
                    dtype="float32")
    if not sp.issparse(x):
        x = sp.csr_matrix(x)
    return x


# --------------------------------------------------------------------------------------------------
# a3: highly-variable-gene selection (dd.py:165-176)
# --------------------------------------------------------------------------------------------------
def gene_variances(csr: sp.csr_matrix) -> np.ndarray:
    """Population variance per gene, E[x^2] - E[x]^2, in the matrix dtype (float32).  dd.py:167-170.

    scipy evaluates ``mean(axis=0)`` as ``(X * (1/N)).sum(axis=0)``, i.e. a per-gene sequential
    float32 accumulation in row order; keeping the very same expression keeps the rounding.
    """
    return (np.array(csr.power(2).mean(axis=0)) - (np.array(csr.mean(axis=0))) ** 2)[0]


def select_hvg(csr: sp.csr_matrix, n_top_var_genes: int):
    """Returns (top_var_genes or None, column-restricted CSR).  dd.py:165-176.

    Columns are re-ordered to ascending-variance order (``argsort(var)[-H:]``), CSR indices sorted.
    """
    if n_top_var_genes > 0 and n_top_var_genes < csr.shape[1]:
        var = gene_variances(csr)
        top = np.argsort(var)[-n_top_var_genes:]
        sub = csr.tocsc()[:, top].tocsr()
        return top, sub
    return None, csr


# --------------------------------------------------------------------------------------------------
# a4: memoised library sizes and L1-normalised rows (dd.py:178-184)
# --------------------------------------------------------------------------------------------------
def library_sizes(csr: sp.csr_matrix) -> np.ndarray:
    """Row sums in float32.  dd.py:182 / :288."""
    return np.asarray(np.sum(csr, axis=1)).ravel()


def l1_normalise_rows(csr: sp.csr_matrix) -> sp.csr_matrix:
    """value / (double) sum(|row|), rounded back to float32; empty rows untouched.

    dd.py:183-184 / :290-291 -> sklearn ``_inplace_csr_row_normalize_l1``
    (sklearn/utils/sparsefuncs_fast.pyx:510-539).
    """
    out = csr.copy()
    data64 = np.abs(out.data.astype(np.float64))
    n = out.shape[0]
    rowsum = np.zeros(n, dtype=np.float64)
    # exact for count data; np.add.reduceat would mis-handle empty rows
    row_of = np.repeat(np.arange(n), np.diff(out.indptr))
    np.add.at(rowsum, row_of, data64)
    safe = np.where(rowsum == 0.0, 1.0, rowsum)
    out.data = (out.data.astype(np.float64) / safe[row_of]).astype(np.float32)
    return out


# --------------------------------------------------------------------------------------------------
# a6: synthetic doublets (dd.py:385-402)
# --------------------------------------------------------------------------------------------------
def draw_parents(rng: np.random.Generator, num_cells: int, boost_rate: float, replace: bool) -> np.ndarray:
    """int64[S, 2] parent indices from the classifier's Generator stream.  dd.py:391-394."""
    num_synths = int(boost_rate * num_cells)
    return rng.choice(num_cells, size=(num_synths, 2), replace=replace)


def create_doublets(csr: sp.csr_matrix, parents: np.ndarray) -> sp.csr_matrix:
    """raw[p0] + raw[p1]: sorted indices, exact zeros dropped.  dd.py:397-399."""
    return csr[parents[:, 0], :] + csr[parents[:, 1], :]


# --------------------------------------------------------------------------------------------------
# a7: log-normalisation of the augmented matrix (dd.py:286-298)
# --------------------------------------------------------------------------------------------------
def lognormalise(normed_raw: sp.csr_matrix, lib_size: np.ndarray, synth: sp.csr_matrix,
                 pseudocount: float):
    """Returns (aug matrix, aug_lib_size, median).  dense float32 if pseudocount != 1 else CSR."""
    synth_lib = library_sizes(synth)
    aug_lib = np.concatenate([lib_size, synth_lib])
    normed_synth = l1_normalise_rows(synth)
    aug = sp.vstack((normed_raw, normed_synth))
    med = np.median(aug_lib)
    scaled = aug * med
    if pseudocount != 1:
        out = np.log(scaled.toarray() + pseudocount)
    else:
        out = np.log1p(scaled)
    return out, aug_lib, med


# --------------------------------------------------------------------------------------------------
# a8: sc.pp.scale(max_value=15) (dd.py:302-303) -- PARITY UNPINNED (scanpy absent)
# --------------------------------------------------------------------------------------------------
def scale_like_scanpy(X: np.ndarray, max_value=None) -> np.ndarray:
    """Restatement of scanpy>=1.10 ``pp.scale`` on a dense float32 array.

    mean in float64; mean of float32-rounded squares in float64; unbiased variance; std==0 -> 1;
    ``X -= mean`` and ``X /= std`` evaluated in float64 and stored back to float32 (numpy in-place
    same-kind casting); symmetric clip to [-max_value, max_value].
    """
    X = np.array(X, copy=True)
    n = X.shape[0]
    mean = X.mean(axis=0, dtype=np.float64)
    mean_sq = (X * X).mean(axis=0, dtype=np.float64)
    var = mean_sq - mean ** 2
    if n != 1:
        var *= n / (n - 1)
    std = np.sqrt(var)
    std[std == 0] = 1
    X -= mean
    X /= std
    if max_value is not None:
        np.clip(X, -max_value, max_value, out=X)
    return X


# --------------------------------------------------------------------------------------------------
# a9: truncated PCA (dd.py:305-314 -> scanpy -> sklearn PCA(svd_solver="auto"))
# --------------------------------------------------------------------------------------------------
def pca_sklearn(X: np.ndarray, n_comps: int, random_state: int, svd_solver: str = "auto") -> np.ndarray:
    """What the reference runs for a dense matrix: sklearn PCA.fit_transform (returns U*S)."""
    from sklearn.decomposition import PCA

    return PCA(n_components=n_comps, svd_solver=svd_solver, random_state=random_state).fit_transform(X)


def sklearn_solver_policy(n_samples: int, n_features: int, n_components: int) -> str:
    """sklearn/decomposition/_pca.py:524-536 for svd_solver='auto' on a dense array."""
    if n_features <= 1000 and n_samples >= 10 * n_features:
        return "covariance_eigh"
    if max(n_samples, n_features) <= 500:
        return "full"
    if 1 <= n_components < 0.8 * min(n_samples, n_features):
        return "randomized"
    return "full"


def pca_start_matrix(seed: int, rows: int, size: int, round_f32: bool = True) -> np.ndarray:
    """Q0 = RandomState(seed).normal(size=(rows, size)); sklearn casts it to A's dtype (float32).

    sklearn/utils/extmath.py:297-300 (check_random_state(int) -> legacy MT19937 RandomState).
    """
    q0 = np.random.RandomState(seed).normal(size=(rows, size))
    if round_f32:
        q0 = q0.astype(np.float32).astype(np.float64)
    return q0


def randomized_pca_f64(X: np.ndarray, n_comps: int, seed: int, round_q0_f32: bool = True,
                       n_oversamples: int = 10, normalizer: str = "LU"):
    """Float64 evaluation of sklearn's randomized PCA on the (float32-valued) matrix X.

    Follows sklearn/decomposition/_pca.py:731-766 and sklearn/utils/extmath.py:287-372,531-607,
    895-950: centre, Q0 normal, n_iter x {normalise(A Q), normalise(A^T Q)}, QR(A Q), B = Q^T A,
    SVD(B), U = Q Uhat, transpose handling, v-based sign flip, return U*S.
    ``normalizer`` only changes the basis of the iterated subspace, not the subspace ("LU" is what
    sklearn uses; "QR" is what the GPU path uses) -- the returned scores agree to ~1e-12.
    Returns (scores[M, C] float64, singular_values[C], components[C, H]).
    """
    import scipy.linalg as sla

    A = np.asarray(X, dtype=np.float64)
    A = A - A.mean(axis=0)
    n_samples, n_features = A.shape
    size = n_comps + n_oversamples
    n_iter = 7 if n_comps < 0.1 * min(A.shape) else 4
    transpose = n_samples < n_features
    if transpose:
        A = A.T
    Q = pca_start_matrix(seed, A.shape[1], size, round_f32=round_q0_f32)

    def norm(Y):
        if normalizer == "LU":
            return sla.lu(Y, permute_l=True, check_finite=False)[0]
        return sla.qr(Y, mode="economic", check_finite=False)[0]

    for _ in range(n_iter):
        Q = norm(A @ Q)
        Q = norm(A.T @ Q)
    Q = sla.qr(A @ Q, mode="economic", check_finite=False)[0]
    B = Q.T @ A
    Uhat, s, Vt = sla.svd(B, full_matrices=False, lapack_driver="gesdd")
    U = Q @ Uhat
    if transpose:
        U, s, Vt = Vt[:n_comps, :].T, s[:n_comps], U[:, :n_comps].T
    else:
        U, s, Vt = U[:, :n_comps], s[:n_comps], Vt[:n_comps, :]
    # svd_flip(u_based_decision=False): largest-|.| entry of each component row made positive
    idx = np.argmax(np.abs(Vt), axis=1)
    signs = np.sign(Vt[np.arange(Vt.shape[0]), idx])
    U = U * signs[None, :]
    Vt = Vt * signs[:, None]
    return U * s[None, :], s, Vt


def exact_pca_f64(X: np.ndarray, n_comps: int):
    """sklearn's exact regimes ("full": LAPACK SVD of the centred matrix; "covariance_eigh": eigh of the
    covariance) in float64: scores = U S with the v-based sign flip (sklearn/decomposition/_pca.py:
    _fit_full + svd_flip(u_based_decision=False))."""
    A = np.asarray(X, dtype=np.float64)
    A = A - A.mean(axis=0)
    U, s, Vt = np.linalg.svd(A, full_matrices=False)
    U, s, Vt = U[:, :n_comps], s[:n_comps], Vt[:n_comps]
    idx = np.argmax(np.abs(Vt), axis=1)
    signs = np.sign(Vt[np.arange(Vt.shape[0]), idx])
    return U * signs[None, :] * s[None, :], s, Vt * signs[:, None]


def pca_f64(X: np.ndarray, n_comps: int, seed: int):
    """Float64 evaluation of whatever PCA(svd_solver="auto") selects for X."""
    if sklearn_solver_policy(X.shape[0], X.shape[1], n_comps) == "randomized":
        return randomized_pca_f64(X, n_comps, seed)
    return exact_pca_f64(X, n_comps)


def per_component_rel_dev(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """||a[:, c] - b[:, c]|| / ||b[:, c]|| per PCA component (the H3 acceptance metric)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b, axis=0) / np.linalg.norm(b, axis=0)


# --------------------------------------------------------------------------------------------------
# a10/a11: exact kNN over the embedding
# --------------------------------------------------------------------------------------------------
def knn_exact(emb: np.ndarray, k: int, include_self: bool, algorithm: str = "brute"):
    """Exact Euclidean kNN.

    phenograph (dd.py:320-322): NearestNeighbors(k+1, kd_tree) with the self column dropped ->
    ``knn_exact(emb, 30, include_self=False, algorithm="kd_tree")``.
    scanpy (dd.py:331-336): n_neighbors=10 counts the cell itself -> ``knn_exact(emb, 10, True)``.
    Returns (indices int64 [M, k], distances float64 [M, k]) ordered by increasing distance.
    """
    from sklearn.neighbors import NearestNeighbors

    emb = np.asarray(emb)
    kk = k if include_self else k + 1
    nn = NearestNeighbors(n_neighbors=kk, algorithm=algorithm, metric="euclidean").fit(emb)
    dist, idx = nn.kneighbors(emb)
    if not include_self:
        idx, dist = idx[:, 1:], dist[:, 1:]
    return idx, dist


def knn_metric(emb: np.ndarray, k: int, include_self: bool, metric: str):
    """phenograph.cluster(primary_metric=...) (dd.py:320-322 [upstream phenograph.core.find_neighbors], restated): "manhattan"
    is sklearn's minkowski with p = 1, "cosine" / "correlation" go to sklearn's brute force; k + 1 neighbours, the point
    itself dropped.  The arithmetic is scikit-learn's own (float64 here, so that ties fall by index)."""
    from sklearn.neighbors import NearestNeighbors

    e = np.asarray(emb, dtype=np.float64)
    kk = k if include_self else k + 1
    if metric == "manhattan":
        nn = NearestNeighbors(n_neighbors=kk, algorithm="brute", metric="minkowski", p=1)
    else:
        nn = NearestNeighbors(n_neighbors=kk, algorithm="brute", metric=metric)
    dist, idx = nn.fit(e).kneighbors(e)
    if not include_self:
        # drop the point itself (it is at distance 0; with duplicate points sklearn may list the duplicate first)
        out_i = np.empty((e.shape[0], k), dtype=np.int64)
        out_d = np.empty((e.shape[0], k))
        for r in range(e.shape[0]):
            keep = np.flatnonzero(idx[r] != r)[:k]
            if len(keep) < k:
                keep = np.arange(1, k + 1)
            out_i[r], out_d[r] = idx[r, keep], dist[r, keep]
        return out_i, out_d
    return idx, dist


def knn_bruteforce_f64(emb: np.ndarray, k: int, include_self: bool):
    """Independent float64 definition used to adjudicate ties: squared distances by direct
    differences, ordering by (distance, index).  O(M^2) memory in blocks; small M only."""
    e = np.asarray(emb, dtype=np.float64)
    m = e.shape[0]
    kk = k if include_self else k + 1
    out_i = np.empty((m, kk), dtype=np.int64)
    out_d = np.empty((m, kk), dtype=np.float64)
    blk = 512
    for s in range(0, m, blk):
        q = e[s:s + blk]
        d2 = np.zeros((q.shape[0], m))
        for c in range(e.shape[1]):
            diff = q[:, c][:, None] - e[:, c][None, :]
            d2 += diff * diff
        if not include_self:
            d2[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = -1.0  # self always first
        order = np.lexsort((np.broadcast_to(np.arange(m), d2.shape), d2), axis=1)[:, :kk]
        out_i[s:s + blk] = order
        out_d[s:s + blk] = np.take_along_axis(d2, order, axis=1)
    if not include_self:
        out_i, out_d = out_i[:, 1:], out_d[:, 1:]
    return out_i, np.sqrt(np.maximum(out_d, 0.0))


# --------------------------------------------------------------------------------------------------
# graphs handed to community detection -- PARITY UNPINNED (phenograph / scanpy / umap absent)
# --------------------------------------------------------------------------------------------------
def jaccard_graph(idx: np.ndarray, prune: bool) -> sp.csr_matrix:
    """PhenoGraph graph from a kNN table without self (dd.py:320-322 [upstream phenograph.cluster]).

    J_ij = |N(i) & N(j)| / (2k - |N(i) & N(j)|) for j in N(i).  prune=True keeps mutual pairs with
    weight J_ij * J_ji; prune=False averages (J + J^T)/2.  Returned symmetric, no explicit zeros.
    """
    m, k = idx.shape
    rows = np.repeat(np.arange(m), k)
    cols = np.asarray(idx).ravel()
    # |N(i) & N(j)| for every pair: (B B^T)_ij with B the m x m kNN indicator; only the kNN positions are kept
    B = sp.csr_matrix((np.ones(m * k, dtype=np.float32), (rows, cols)), shape=(m, m))
    B.sum_duplicates()
    B.data[:] = 1.0
    shared_all = (B @ B.T).tocsr()
    shared = np.asarray(shared_all[rows, cols]).ravel().astype(np.float64)
    w = shared / (2.0 * k - shared)
    J = sp.coo_matrix((w, (rows, cols)), shape=(m, m)).tocsr()
    if prune:
        G = J.multiply(J.T)
    else:
        G = (J + J.T) / 2.0
    G = sp.csr_matrix(G)
    G.eliminate_zeros()
    G.sort_indices()
    return G


def union_knn_graph(idx_with_self: np.ndarray) -> sp.csr_matrix:
    """Topology of scanpy's umap connectivities: i~j iff j in kNN(i) or i in kNN(j), self excluded,
    unit weights (``sc.tl.louvain`` ignores weights, use_weights=False).  dd.py:331-342."""
    m, k = idx_with_self.shape
    rows = np.repeat(np.arange(m), k)
    cols = idx_with_self.ravel()
    keep = rows != cols
    A = sp.coo_matrix((np.ones(keep.sum()), (rows[keep], cols[keep])), shape=(m, m)).tocsr()
    A = ((A + A.T) > 0).astype(np.float64)
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A


def umap_connectivities(idx_with_self: np.ndarray, dist: np.ndarray) -> sp.csr_matrix:
    """Weights of scanpy's ``sc.pp.neighbors(method="umap")`` graph, which ``sc.tl.leiden`` uses (use_weights=True):
    umap-learn's ``fuzzy_simplicial_set`` (McInnes et al. 2018; ``smooth_knn_dist`` + ``compute_membership_strengths``
    + fuzzy union, local_connectivity = 1, set_op_mix_ratio = 1).  PARITY UNPINNED (umap-learn absent).  Restated:

    * distances enter as float32 (what scanpy hands over), everything else is float64;
    * rho_i = smallest positive distance of row i (0 if none);
    * sigma_i from 64 bisection steps on  sum_{j>=1} exp(-max(d_ij - rho_i, 0) / sigma) = log2(k)   (lo = 0,
      hi = inf, start 1, doubling while hi is infinite) -- upstream stops early once within 1e-5, the restatement
      always runs the 64 steps; then the floor sigma_i >= 1e-3 * mean(d_i.) (upstream uses the mean of *all* distances
      for a row without a positive distance; such a row has mean 0 here and keeps its sigma);
    * w_ij = 0 for j = i, 1 if d_ij <= rho_i or sigma_i = 0, else exp(-(d_ij - rho_i) / sigma_i);
    * symmetric weight  a + b - a*b  with a = w_ij, b = w_ji (0 when absent)."""
    idx = np.asarray(idx_with_self, dtype=np.int64)
    d = np.asarray(dist, dtype=np.float64).astype(np.float32).astype(np.float64)
    m, k = idx.shape
    target = float(np.log2(k))
    pos = np.where(d > 0.0, d, np.inf)
    rho = pos.min(axis=1)
    rho[~np.isfinite(rho)] = 0.0
    lo = np.zeros(m)
    hi = np.full(m, np.inf)
    mid = np.ones(m)
    gap = np.maximum(d[:, 1:] - rho[:, None], 0.0)
    for _ in range(64):
        psum = np.zeros(m)
        for j in range(k - 1):                       # left-to-right, as the device does
            psum = psum + np.where(gap[:, j] > 0.0, np.exp(-(gap[:, j] / mid)), 1.0)
        over = psum > target
        hi = np.where(over, mid, hi)
        lo = np.where(over, lo, mid)
        mid = np.where(over, (lo + hi) / 2.0, np.where(np.isinf(hi), mid * 2.0, (lo + hi) / 2.0))
    rowmean = np.zeros(m)
    for j in range(k):
        rowmean = rowmean + d[:, j]
    rowmean = rowmean / k
    sigma = np.maximum(mid, 1e-3 * rowmean)
    rows = np.repeat(np.arange(m), k)
    cols = idx.ravel()
    dd = d.ravel() - rho[rows]
    val = np.where((dd <= 0.0) | (sigma[rows] == 0.0), 1.0, np.exp(-(dd / sigma[rows])))
    val = np.where(cols == rows, 0.0, val)
    keep = (cols >= 0) & (val != 0.0)
    A = sp.coo_matrix((val[keep], (rows[keep], cols[keep])), shape=(m, m)).tocsr()
    A.sum_duplicates()
    T = A.T.tocsr()
    S = A + T - A.multiply(T)
    S = sp.csr_matrix(S)
    S.eliminate_zeros()
    S.sort_indices()
    return S


def relabel_by_size(labels: np.ndarray, min_cluster_size: int | None = None) -> np.ndarray:
    """Labels 0..K-1 by descending community size (ties: smaller original label first).

    phenograph ``sort_by_size`` additionally turns communities of size <= min_cluster_size into -1.
    """
    labels = np.asarray(labels)
    uniq, counts = np.unique(labels, return_counts=True)
    order = np.lexsort((uniq, -counts))
    out = np.empty(labels.shape, dtype=np.int64)
    nxt = 0
    for o in order:
        if min_cluster_size is not None and counts[o] <= min_cluster_size:
            out[labels == uniq[o]] = -1
        else:
            out[labels == uniq[o]] = nxt
            nxt += 1
    return out


def _knn_sklearn_all_cores(emb, k, include_self):
    """sklearn exact search with every host core (used only when timing the CPU baseline)."""
    from sklearn.neighbors import NearestNeighbors

    kk = k if include_self else k + 1
    nn = NearestNeighbors(n_neighbors=kk, algorithm="kd_tree" if not include_self else "brute", n_jobs=-1).fit(emb)
    dist, idx = nn.kneighbors(emb)
    if not include_self:
        idx, dist = idx[:, 1:], dist[:, 1:]
    return idx, dist


def cluster_embedding(emb: np.ndarray, algorithm: str, clustering_kwargs: dict, random_state: int,
                      louvain_fn=None, knn_fn=None, best_of_fn=None) -> np.ndarray:
    """kNN -> graph -> deterministic community detection -> size-sorted labels.

    ``louvain_fn(indptr, indices, weights, gamma, seed) -> labels``: defaults to the pure-Python
    specification in ``oracle/louvain_ref.py``; bench's cpu_baseline leg passes a compiled
    implementation of the same specification to keep the timing meaningful.
    ``best_of_fn(indptr, indices, weights, gamma, seed, q_tol) -> labels``: PhenoGraph's restart rule
    (``louvain_ref.louvain_best_of`` by default).
    """
    louvain_fn = louvain_fn or louvain_ref.louvain
    best_of_fn = best_of_fn or (lambda ip, ix, w, gamma, seed, q_tol: louvain_ref.louvain_best_of(ip, ix, w, gamma, seed, q_tol)[0])
    knn_fn = knn_fn or knn_bruteforce_f64
    kw = dict(clustering_kwargs or {})
    if algorithm == "phenograph":
        k = int(kw.get("k", 30))
        metric = str(kw.get("primary_metric", "euclidean")).lower()
        if metric in ("euclidean", "minkowski"):
            idx, _ = knn_fn(emb, k, include_self=False)
        else:
            idx, _ = knn_metric(emb, k, False, metric)
        G = jaccard_graph(idx, prune=bool(kw.get("prune", True)))
        if kw.get("clustering_algo", "louvain") == "leiden":
            # phenograph hands resolution_parameter / seed to leidenalg only
            seed = kw.get("seed", None)
            seed = random_state if seed is None else int(seed)
            lab = louvain_ref.leiden(G.indptr, G.indices, G.data, float(kw.get("resolution_parameter", 1.0)), seed)
        else:
            # the Louvain binaries take neither a resolution nor a seed (gamma = 1, deterministic seed = random_state) and
            # are re-run until 20 runs in a row gain less than q_tol (phenograph.core.runlouvain)
            lab = best_of_fn(G.indptr, G.indices, G.data, 1.0, int(random_state), float(kw.get("q_tol", 1e-3)))
        return relabel_by_size(lab, int(kw.get("min_cluster_size", 10)))
    idx, dist = knn_fn(emb, 10, include_self=True)
    gamma = float(kw.get("resolution", 4))
    leiden = algorithm == "leiden"
    # sc.tl.leiden runs on the umap connectivities (use_weights=True); sc.tl.louvain ignores the weights
    # (use_weights=False) unless told otherwise
    G = umap_connectivities(idx, dist) if kw.get("use_weights", leiden) else union_knn_graph(idx)
    if leiden:
        lab = louvain_ref.leiden(G.indptr, G.indices, G.data, gamma, int(random_state))
    else:
        lab = louvain_fn(G.indptr, G.indices, G.data, gamma, int(random_state))
    return relabel_by_size(lab, None)


# --------------------------------------------------------------------------------------------------
# a12: community bookkeeping + hypergeometric test (dd.py:344-383)
# --------------------------------------------------------------------------------------------------
def score_communities(full: np.ndarray, num_cells: int):
    """scores, log_p_values (float64[N]) from the community vector of the augmented set.

    score_c = synth_c / (synth_c + orig_c); logp_c = hypergeom.logsf(synth_c, M, S, synth_c+orig_c)
    for communities holding >= 1 original cell; community -1 -> NaN.
    """
    full = np.asarray(full)
    comm, synth_comm = full[:num_cells], full[num_cells:]
    M, S = full.shape[0], full.shape[0] - num_cells
    n_synth = collections.Counter(synth_comm.tolist())
    n_orig = collections.Counter(comm.tolist())
    score_of, logp_of = {}, {}
    for c, oc in n_orig.items():
        sc = n_synth.get(c, 0)
        score_of[c] = float(sc) / (sc + oc)
        logp_of[c] = float(hypergeom.logsf(sc, M, S, sc + oc))
    scores = np.array([score_of[c] for c in comm.tolist()], dtype=np.float64)
    logp = np.array([logp_of[c] for c in comm.tolist()], dtype=np.float64)
    if full.min() < 0:
        scores[comm == -1] = np.nan
        logp[comm == -1] = np.nan
    return scores, logp


# --------------------------------------------------------------------------------------------------
# a13: predict / doublet_score (dd.py:216-272)
# --------------------------------------------------------------------------------------------------
def predict(all_log_p, all_scores, p_thresh=1e-7, voter_thresh=0.9):
    """Returns dict(labels, voting_average | suggested_score_cutoff).  dd.py:231-254."""
    n_iters = all_log_p.shape[0]
    log_p_thresh = np.log(p_thresh)
    if n_iters > 1:
        with np.errstate(invalid="ignore"):
            va = np.mean(np.ma.masked_invalid(all_log_p) <= log_p_thresh, axis=0)
            labels = np.ma.filled((va >= voter_thresh).astype(float), np.nan)
            va = np.ma.filled(va, np.nan)
        return dict(labels=labels, voting_average=va)
    cuts = np.unique(all_scores[~np.isnan(all_scores)])
    drop = (np.argmax(cuts[1:] - cuts[:-1]) + 1) if len(cuts) > 1 else 0
    cutoff = cuts[drop]
    with np.errstate(invalid="ignore"):
        labels = all_scores[0, :] >= cutoff
    labels[np.isnan(all_scores)[0, :]] = np.nan
    return dict(labels=labels, suggested_score_cutoff=cutoff)


def doublet_score(all_log_p):
    """-mean over iterations of the masked log p-values (MaskedArray when n_iters>1).  dd.py:266-272."""
    if all_log_p.shape[0] > 1:
        with np.errstate(invalid="ignore"):
            avg = np.mean(np.ma.masked_invalid(all_log_p), axis=0)
    else:
        avg = all_log_p[0]
    return -avg


# --------------------------------------------------------------------------------------------------
# whole-fit CPU restatement (the "port" timed by bench.py's cpu_baseline leg)
# --------------------------------------------------------------------------------------------------
class OracleClassifier:
    """CPU restatement of BoostClassifier.fit()/predict()/doublet_score() built from the stages above.

    Same constructor semantics as the reference for the arguments it takes; ``pca`` selects
    "sklearn" (float32, what the reference runs) or "f64" (``randomized_pca_f64``).
    """

    def __init__(self, boost_rate=0.25, n_components=30, n_top_var_genes=10000, replace=False,
                 clustering_algorithm="phenograph", clustering_kwargs=None, n_iters=10,
                 pseudocount=0.1, random_state=0, standard_scaling=False, pca="sklearn",
                 louvain_fn=None, knn_fn=None, best_of_fn=None):
        if clustering_algorithm not in ("louvain", "phenograph", "leiden"):
            raise ValueError("Clustering algorithm needs to be one of ['louvain', 'phenograph', 'leiden']")
        self.boost_rate = 0.5 if (not replace and boost_rate > 0.5) else boost_rate
        self.n_components = min(n_components, n_top_var_genes) if (
            n_components == 30 and n_top_var_genes > 0) else n_components
        self.n_top_var_genes = max(0, n_top_var_genes)
        self.replace = replace
        self.clustering_algorithm = clustering_algorithm
        self.clustering_kwargs = dict(clustering_kwargs or {})
        if clustering_algorithm == "phenograph":
            self.clustering_kwargs.setdefault("prune", True)
        else:
            self.clustering_kwargs.setdefault("directed", False)
            self.clustering_kwargs.setdefault("resolution", 4)
        self.n_iters = n_iters
        self.pseudocount = pseudocount
        self.random_state = random_state
        self.standard_scaling = standard_scaling
        self.pca = pca
        self.louvain_fn = louvain_fn
        self.knn_fn = knn_fn
        self.best_of_fn = best_of_fn
        self.rng = np.random.default_rng(random_state)
        self.timings = collections.defaultdict(float)

    def fit(self, raw_counts):
        import time

        t0 = time.perf_counter()
        raw = coerce_counts(raw_counts)
        self.top_var_genes_, raw = select_hvg(raw, self.n_top_var_genes)
        N = raw.shape[0]
        lib = library_sizes(raw)
        normed = l1_normalise_rows(raw)
        S = int(self.boost_rate * N)
        self.all_scores_ = np.zeros((self.n_iters, N))
        self.all_log_p_values_ = np.zeros((self.n_iters, N))
        self.communities_ = np.zeros((self.n_iters, N))
        self.synth_communities_ = np.zeros((self.n_iters, S))
        self.parents_ = []
        self.embeddings_ = []
        self.timings["prologue"] += time.perf_counter() - t0
        for it in range(self.n_iters):
            t0 = time.perf_counter()
            parents = draw_parents(self.rng, N, self.boost_rate, self.replace)
            synth = create_doublets(raw, parents)
            t1 = time.perf_counter()
            aug, _, _ = lognormalise(normed, lib, synth, self.pseudocount)
            if self.standard_scaling:
                if sp.issparse(aug):
                    aug = aug.toarray()             # scanpy densifies to zero-centre; PCA then sees a dense array
                aug = scale_like_scanpy(aug, max_value=15)
            t2 = time.perf_counter()
            if sp.issparse(aug):
                # dd.py:308: svd_solver="arpack" on the CSR matrix (sklearn centres implicitly)
                emb = pca_sklearn(aug.astype(np.float32 if self.pca == "sklearn" else np.float64), self.n_components,
                                  self.random_state, svd_solver="arpack").astype(np.float32)
            elif self.pca == "sklearn":
                emb = pca_sklearn(aug, self.n_components, self.random_state).astype(np.float32)
            else:
                emb = pca_f64(aug, self.n_components, self.random_state)[0].astype(np.float32)
            t3 = time.perf_counter()
            full = cluster_embedding(emb, self.clustering_algorithm, self.clustering_kwargs,
                                     self.random_state, self.louvain_fn, self.knn_fn, self.best_of_fn)
            t4 = time.perf_counter()
            sc_, lp_ = score_communities(full, N)
            t5 = time.perf_counter()
            self.all_scores_[it], self.all_log_p_values_[it] = sc_, lp_
            self.communities_[it] = full[:N]
            self.synth_communities_[it] = full[N:]
            self.parents_.append([list(p) for p in parents])
            self.embeddings_.append(emb)
            for name, dt in (("doublets", t1 - t0), ("lognorm", t2 - t1), ("pca", t3 - t2),
                             ("cluster", t4 - t3), ("score", t5 - t4)):
                self.timings[name] += dt
        return self

    def predict(self, p_thresh=1e-7, voter_thresh=0.9):
        r = predict(self.all_log_p_values_, self.all_scores_, p_thresh, voter_thresh)
        self.labels_ = r["labels"]
        if "voting_average" in r:
            self.voting_average_ = r["voting_average"]
        else:
            self.suggested_score_cutoff_ = r["suggested_score_cutoff"]
        return self.labels_

    def doublet_score(self):
        return doublet_score(self.all_log_p_values_)
