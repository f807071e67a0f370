"""Shared by oracle/make_clustering_fixture.py (build container, needs networkx) and tests/test_clustering_independent.py:
the seeded embeddings and the graphs on which the community detection of the build is cross-checked against an
implementation the builder did not write (networkx.community.louvain_communities / modularity).

The graphs are what the reference hands to its (absent) native Louvain / Leiden codes: PhenoGraph's pruned Jaccard graph
on k = 30 neighbours (dd.py:320-322, resolution 1) and scanpy's neighbour graph on 10 neighbours (dd.py:331-342,
resolution 4; unit weights for sc.tl.louvain, umap connectivities for sc.tl.leiden)."""
from __future__ import annotations

import numpy as np

N_LARGE = 20000
SEED_LARGE = 1


def make_embedding(n: int, seed: int, dims: int = 30, types: int = 12) -> np.ndarray:
    """A PCA-like embedding: a dominant library-size axis, `types` cell types separated on the leading components,
    isotropic noise on all of them (the shape of the benchmark embedding)."""
    rng = np.random.default_rng(seed)
    centers = rng.normal(0.0, 3.3, size=(types, dims))
    centers[:, 12:] = 0.0
    lab = rng.integers(0, types, size=n)
    e = centers[lab] + rng.normal(0.0, 1.4, size=(n, dims))
    e[:, 0] += rng.normal(0.0, 6.6, size=n)
    return e.astype(np.float32)


# name -> (k, include_self, graph kind, resolution, weighted, flavour)
FLAVOURS = {
    "phenograph": (30, False, "jaccard_pruned", 1.0, True, "louvain"),
    "scanpy_louvain": (10, True, "union", 4.0, False, "louvain"),
    "scanpy_leiden": (10, True, "umap", 4.0, True, "leiden"),
}


def oracle_graph(emb, flavour):
    """Symmetric scipy CSR the oracle builds for `flavour` (exact kNN by scikit-learn on all cores)."""
    from oracle import dd_oracle as orc

    k, include_self, kind, _, _, _ = FLAVOURS[flavour]
    idx, _ = orc._knn_sklearn_all_cores(emb, k, include_self)
    # distances recomputed in float64 from the neighbour table (scikit-learn's float32 brute force returns 1e-3 instead of
    # 0 for a point's distance to itself, which the umap weights would take for the nearest neighbour's distance)
    e = np.asarray(emb, dtype=np.float64)
    dist = np.sqrt(((e[:, None, :] - e[idx]) ** 2).sum(axis=2))
    order = np.argsort(dist, axis=1, kind="stable")
    idx, dist = np.take_along_axis(idx, order, axis=1), np.take_along_axis(dist, order, axis=1)
    if kind == "jaccard_pruned":
        G = orc.jaccard_graph(idx, prune=True)
    elif kind == "union":
        G = orc.union_knn_graph(idx)
    else:
        G = orc.umap_connectivities(idx, dist)
    return without_self_loops(G)


def without_self_loops(G):
    """Duplicate points (replace=True draws, golden case d) put a point into its own neighbour list; networkx and a
    both-directions CSR count the weight of a self-loop differently in the degrees, so the cross-check leaves them out."""
    import scipy.sparse as sp

    G = sp.csr_matrix(G).copy()
    G.setdiag(0.0)
    G.eliminate_zeros()
    G.sort_indices()
    return G


def modularity(indptr, indices, weights, labels, gamma: float) -> float:
    """Q = sum_c [ in_c / 2m - gamma (tot_c / 2m)^2 ] of a symmetric CSR graph (both directions stored), written
    independently of the oracle and of libddx: the yardstick of the GPU tests (networkx.community.modularity gives
    the same number, test_modularity_yardstick_is_networkx's)."""
    indptr = np.asarray(indptr)
    indices = np.asarray(indices)
    w = np.asarray(weights, dtype=np.float64)
    _, lab = np.unique(np.asarray(labels), return_inverse=True)
    rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    m2 = w.sum()
    nc = lab.max() + 1
    tot = np.bincount(lab[rows], weights=w, minlength=nc)
    same = lab[rows] == lab[indices]
    inn = np.bincount(lab[rows][same], weights=w[same], minlength=nc)
    return float((inn / m2 - gamma * (tot / m2) ** 2).sum())


def adjusted_rand(a, b) -> float:
    from sklearn.metrics import adjusted_rand_score

    return float(adjusted_rand_score(np.asarray(a), np.asarray(b)))
