"""Seeded synthetic count matrices shaped like scRNA-seq data (SURVEY.md section 8(d) recipe).

The reference ships no generator and no data; its only test draws ``np.random.poisson`` counts
(tests/test_package.py:8).  The benchmark configurations in BASELINE.json are all synthetic, so this
module provides one reproducible recipe for tests and for ``bench.py``:

* K cell types; per-gene base rate ~ Gamma(0.3, 1); per type a random 10 % of genes is multiplied by
  exp(N(0,1)); profiles normalised to sum 1;
* cell depth ~ LogNormal(log(mu_depth), 0.3), mu_depth calibrated by bisection on a pilot so that the
  realised density is the target density;
* counts ~ Poisson(depth * profile), float32 CSR with int32 indices;
* finally ``doublet_frac`` of the rows are replaced by the sum of two random other rows.

``device="cuda"`` draws the Poisson field with torch on the GPU (plumbing only: the product path does
not depend on torch); the two back-ends draw different streams, each is deterministic for its seed.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def _profiles(rng, n_genes, n_types):
    base = rng.gamma(0.3, 1.0, size=n_genes) + 1e-12
    prof = np.tile(base, (n_types, 1))
    for t in range(n_types):
        pick = rng.random(n_genes) < 0.10
        prof[t, pick] *= np.exp(rng.normal(0.0, 1.0, size=int(pick.sum())))
    prof /= prof.sum(axis=1, keepdims=True)
    return prof


def _density_for_depth(prof, depth, rng, n_pilot=400):
    types = rng.integers(0, prof.shape[0], size=n_pilot)
    d = depth * np.exp(rng.normal(0.0, 0.3, size=n_pilot))
    lam = d[:, None] * prof[types]
    return float(np.mean(1.0 - np.exp(-lam)))


def _calibrate_depth(prof, density, seed):
    lo, hi = 1.0, 50.0 * prof.shape[1]
    for _ in range(40):
        mid = np.sqrt(lo * hi)
        if _density_for_depth(prof, mid, np.random.default_rng(seed + 1)) < density:
            lo = mid
        else:
            hi = mid
    return np.sqrt(lo * hi)


def make_counts(n_cells: int, n_genes: int, density: float = 0.05, n_types: int = 12,
                doublet_frac: float = 0.05, seed: int = 20250227, device: str = "cpu",
                chunk: int = 4096) -> sp.csr_matrix:
    """float32 CSR (n_cells x n_genes) of synthetic counts; see module docstring."""
    rng = np.random.default_rng(seed)
    prof = _profiles(rng, n_genes, n_types)
    mu_depth = _calibrate_depth(prof, density, seed)
    types = rng.integers(0, n_types, size=n_cells)
    depth = mu_depth * np.exp(rng.normal(0.0, 0.3, size=n_cells))

    indptr = [np.zeros(1, dtype=np.int64)]
    idx_parts, val_parts = [], []
    if device == "cpu":
        for s in range(0, n_cells, chunk):
            lam = depth[s:s + chunk, None] * prof[types[s:s + chunk]]
            c = rng.poisson(lam)
            r, j = np.nonzero(c)
            idx_parts.append(j.astype(np.int32))
            val_parts.append(c[r, j].astype(np.float32))
            indptr.append(np.bincount(r, minlength=lam.shape[0]).astype(np.int64))
    else:
        import torch

        g = torch.Generator(device=device)
        g.manual_seed(seed)
        tprof = torch.as_tensor(prof, dtype=torch.float32, device=device)
        ttypes = torch.as_tensor(types, device=device)
        tdepth = torch.as_tensor(depth, dtype=torch.float32, device=device)
        for s in range(0, n_cells, chunk):
            lam = tdepth[s:s + chunk, None] * tprof[ttypes[s:s + chunk]]
            c = torch.poisson(lam, generator=g)
            nz = c > 0
            j = nz.nonzero()[:, 1]
            idx_parts.append(j.to(torch.int32).cpu().numpy())
            val_parts.append(c[nz].cpu().numpy().astype(np.float32))
            indptr.append(nz.sum(dim=1).cpu().numpy().astype(np.int64))
    counts_per_row = np.concatenate(indptr[1:]) if len(indptr) > 1 else np.zeros(0, np.int64)
    ip = np.concatenate([[0], np.cumsum(counts_per_row)]).astype(np.int64)
    x = sp.csr_matrix((np.concatenate(val_parts), np.concatenate(idx_parts), ip),
                      shape=(n_cells, n_genes))
    n_dbl = int(doublet_frac * n_cells)
    if n_dbl > 0:
        targets = rng.choice(n_cells, size=n_dbl, replace=False)
        a = rng.integers(0, n_cells, size=n_dbl)
        b = rng.integers(0, n_cells, size=n_dbl)
        dbl = (x[a] + x[b]).tocsr()
        keep = np.ones(n_cells, dtype=bool)
        keep[targets] = False
        # splice: rows in `targets` come from dbl, others from x
        order = np.empty(n_cells, dtype=np.int64)
        order[np.flatnonzero(keep)] = np.arange(keep.sum())
        order[targets] = keep.sum() + np.arange(n_dbl)
        x = sp.vstack([x[np.flatnonzero(keep)], dbl]).tocsr()[order]
    x = x.astype(np.float32)
    x.sort_indices()
    if x.indices.dtype != np.int32:
        x.indices = x.indices.astype(np.int32)
    return x
