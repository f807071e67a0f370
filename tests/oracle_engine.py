"""Test-only engine: the oracle's stages behind the product's engine interface, so that the host
logic of BoostClassifier (validation, RNG stream order, sharding, gather, predict) can be exercised on
machines without a GPU.  Never used by the product."""
import numpy as np

from oracle import dd_oracle as orc


class OracleEngine:
    def __init__(self, device):
        self.device = device

    def close(self):
        pass

    def stage_raw(self, csr):
        self.staged = csr.copy()

    def gene_variances(self):
        return orc.gene_variances(self.staged)

    def select_columns(self, cols):
        self.upload(self.staged.tocsc()[:, cols].tocsr())

    def upload(self, csr):
        self.raw = csr.copy()
        self.lib = orc.library_sizes(self.raw)
        self.normed = orc.l1_normalise_rows(self.raw)

    def clone_from(self, other):
        self.raw, self.lib, self.normed = other.raw, other.lib, other.normed

    def run_iteration(self, parents, pseudocount, standard_scaling, n_components, q0, knn_k, include_self, graph_mode, gamma=None,
                      pca_lock=None, verbose=False, metric="euclidean"):
        synth = orc.create_doublets(self.raw, parents)
        aug, _, _ = orc.lognormalise(self.normed, self.lib, synth, pseudocount)
        import scipy.sparse as sp

        if standard_scaling:
            aug = orc.scale_like_scanpy(aug.toarray() if sp.issparse(aug) else aug, 15)
        if sp.issparse(aug):
            emb = orc.pca_sklearn(aug.astype(np.float64), n_components, self.seed, svd_solver="arpack").astype(np.float32)
        else:
            emb = None
        # the product passes the start matrix it drew; the oracle draws the same one from the seed
        if emb is None:
            emb = orc.pca_f64(aug, n_components, self.seed)[0].astype(np.float32)
        idx, dist = orc.knn_bruteforce_f64(emb, knn_k, include_self) if metric == "euclidean" else orc.knn_metric(emb, knn_k, include_self, metric)
        if graph_mode == 3:
            G = orc.umap_connectivities(idx, dist)
        elif graph_mode == 2:
            G = orc.union_knn_graph(idx)
        else:
            G = orc.jaccard_graph(idx, prune=(graph_mode == 0))
        return G.indptr.astype(np.int64), G.indices.astype(np.int32), G.data.astype(np.float64)

    def timings(self):
        return {}


def make_engine_factory(seed):
    def factory(device):
        e = OracleEngine(device)
        e.seed = seed
        return e
    return factory
