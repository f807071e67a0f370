import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from doubletdetection_amd import _lib
from doubletdetection_amd._synthetic import make_counts
_lib.OPTIONS["pca_debug"] = "1"
X = make_counts(50_000, 20_000, density=0.05, device="cuda:0", seed=11)
c = _lib.Context(0)
c.upload_raw(X); c.select_columns(np.argsort(c.gene_variances())[-10000:])
c.create_doublets(np.random.default_rng(0).choice(50_000, size=(12_500, 2), replace=False))
c.lognormalise(1.0)
start = np.random.RandomState(0).normal(size=(c.H, 40))
c.pca_exact_sparse(30, start, tol=1e-6, max_steps=8)
c.pca_exact_sparse(30, start, tol=1e-6, max_steps=24)
