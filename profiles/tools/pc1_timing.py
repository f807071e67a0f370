"""pseudocount = 1 (dd.py:296-297,308: sparse matrix, ARPACK) at BASELINE configs[1]: operator products per PCA, time per
product (by block width), seconds per PCA."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import _lib
from doubletdetection_amd.classifier import _HipEngine
from doubletdetection_amd._synthetic import make_counts
X = make_counts(50_000, 20_000, density=0.05, device="cuda:0", seed=11)
eng = _HipEngine(0)
c = eng.ctx
c.upload_raw(X); c.select_columns(np.argsort(c.gene_variances())[-10000:])
c.create_doublets(np.random.default_rng(0).choice(50_000, size=(12_500, 2), replace=False))
c.lognormalise(1.0)
M, H = c.M, c.H
for n in (1, 8, 40):
    x = np.random.default_rng(1).normal(size=(H, n))
    c.operator_apply(x, 2)
    t0 = time.perf_counter()
    for _ in range(20):
        c.operator_apply(x, 2)
    print(f"A^T A x, {n} vector(s): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call")
t0 = time.perf_counter()
eng._pca_arpack(30, 0)
print(f"ARPACK PCA: {time.perf_counter() - t0:.3f} s, {eng.arpack_products} Gram-operator products")
