"""Matrix-core products (DDX_SPMM=mfma) against the LDS products: PCA scores and launch times at the headline workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import _lib
from doubletdetection_amd._synthetic import make_counts
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
X = make_counts(cells, 30_000, density=0.03, device="cuda:0", seed=20250227)
res = {}
for mode in ("lds", "mfma"):
    os.environ["DDX_SPMM"] = mode
    c = _lib.Context(0)
    c.upload_raw(X); var = c.gene_variances(); c.select_columns(np.argsort(var)[-10000:])
    c.create_doublets(np.random.default_rng(0).choice(cells, size=(cells // 4, 2), replace=False)); c.lognormalise(0.1)
    q0 = np.random.RandomState(0).normal(size=(10000, 40)).astype(np.float32).astype(np.float64)
    c.timing_enable(True)
    for rep in range(2):
        c.timing_reset()
        c.pca(30, q0)
        t = c.timings()
    print(mode, {k: round(v[1] / max(v[0], 1), 4) for k, v in t.items() if k.startswith(("spmm", "pca"))}, flush=True)
    res[mode] = c.embedding_f64()
    c.close()
(e0, s0), (e1, s1) = res["lds"], res["mfma"]
rel = np.linalg.norm(e1 - e0, axis=0) / np.linalg.norm(e0, axis=0)
print("max rel diff of a score column", rel.max(), "column", rel.argmax(), "first 12:", rel[:12].max())
print("max rel diff of a singular value", (np.abs(s1 - s0) / s0).max())
