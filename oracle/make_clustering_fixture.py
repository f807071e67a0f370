"""TEST INFRASTRUCTURE -- writes tests/golden/clustering_networkx.npz (build container only: needs networkx).

Run:  python -m oracle.make_clustering_fixture

The reference's community detection (PhenoGraph's Louvain binaries behind dd.py:320-322, louvain-igraph / leidenalg
behind dd.py:337-342) is absent and cannot be pinned bit for bit.  This script pins the QUALITY of the build's
deterministic Louvain / Leiden against an implementation the builder did not write: networkx 3.4's
``louvain_communities`` (Blondel et al. 2008 with a resolution parameter) and ``modularity``.  For every graph it
stores, for five networkx seeds, the modularity it reaches, the number of communities, the partitions and the
adjusted Rand index between every two networkx runs (the spread a second correct Louvain shows against the first), so that the
``-m gpu`` tests can hold the device path to the same numbers on a box without networkx.

Graphs: the three flavours of tests/clustering_cases.py on a seeded 20 000-point embedding, and PhenoGraph's graph on
the reference-generated PCA embedding of every golden case (tests/golden/case_*.npz, iteration 0).
"""
from __future__ import annotations

import glob
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import clustering_cases as cc  # noqa: E402

NX_SEEDS = (0, 1, 2, 3, 4)


def nx_graph(G):
    import networkx as nx

    C = sp.triu(sp.csr_matrix(G)).tocoo()
    g = nx.Graph()
    g.add_nodes_from(range(G.shape[0]))
    g.add_weighted_edges_from(zip(C.row.tolist(), C.col.tolist(), C.data.tolist()))
    return g


def nx_louvain(G, gamma, seed):
    """(labels, modularity) of networkx's Louvain on the symmetric CSR graph G."""
    import networkx as nx

    g = nx_graph(G)
    comms = nx.community.louvain_communities(g, weight="weight", resolution=gamma, seed=seed)
    lab = np.empty(G.shape[0], dtype=np.int64)
    for i, c in enumerate(comms):
        lab[list(c)] = i
    return lab, float(nx.community.modularity(g, comms, weight="weight", resolution=gamma))


def record(out, name, G, gamma):
    labs, qs = [], []
    for s in NX_SEEDS:
        lab, q = nx_louvain(G, gamma, s)
        labs.append(lab)
        qs.append(q)
        assert abs(q - cc.modularity(G.indptr, G.indices, G.data, lab, gamma)) < 1e-9
    aris = [cc.adjusted_rand(labs[i], labs[j]) for i in range(len(labs)) for j in range(i + 1, len(labs))]
    out[name + "_nx_q"] = np.asarray(qs)
    out[name + "_nx_ncomm"] = np.asarray([len(np.unique(l)) for l in labs], dtype=np.int64)
    out[name + "_nx_labels"] = np.asarray(labs).astype(np.int16 if max(l.max() for l in labs) < 32768 else np.int32)
    out[name + "_nx_ari_between_runs"] = np.asarray(aris)
    out[name + "_gamma"] = np.float64(gamma)
    out[name + "_graph_entries"] = np.int64(G.nnz)
    out[name + "_graph_weight"] = np.float64(G.data.sum())
    print(f"{name}: nodes {G.shape[0]} entries {G.nnz} gamma {gamma}: nx Q {qs} communities {out[name + '_nx_ncomm'].tolist()} "
          f"ARI between nx runs {aris}")


def main():
    out = {"n_large": np.int64(cc.N_LARGE), "seed_large": np.int64(cc.SEED_LARGE), "nx_seeds": np.asarray(NX_SEEDS)}
    emb = cc.make_embedding(cc.N_LARGE, cc.SEED_LARGE)
    out["emb_large_checksum"] = np.float64(emb.astype(np.float64).sum())     # the tests regenerate it from the seed
    for flavour, (_, _, _, gamma, _, _) in cc.FLAVOURS.items():
        record(out, "large_" + flavour, cc.oracle_graph(emb, flavour), gamma)
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "case_*.npz"))):
        case = os.path.basename(path)[:6]
        z = np.load(path)
        G = cc.oracle_graph(np.asarray(z["pca_f32"][0]), "phenograph")
        record(out, case + "_phenograph", G, 1.0)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clustering_networkx.npz"), **out)


if __name__ == "__main__":
    main()
