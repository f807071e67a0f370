"""Operator-product time and PCA precision for the library named by DDX_LIB / the DDX_SPMM_* switches: average launch of
the two products at the headline size, and the largest relative deviation per component from the float64-gather run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import _lib
from doubletdetection_amd._synthetic import make_counts
from oracle import dd_oracle as orc
X = make_counts(100_000, 30_000, density=0.03, device="cuda:0", seed=20250227)
c = _lib.Context(0)
c.upload_raw(X); var = c.gene_variances(); c.select_columns(np.argsort(var)[-10000:])
c.create_doublets(np.random.default_rng(0).choice(100_000, size=(25_000, 2), replace=False)); c.lognormalise(0.1)
q0 = np.random.RandomState(0).normal(size=(10000, 40)).astype(np.float32).astype(np.float64)
c.timing_enable(True)
for rep in range(3):
    c.timing_reset()
    c.pca(30, q0)
    t = c.timings()
emb, _ = c.embedding_f64()
ref_path = "/tmp/spmm_ref.npy"
if os.environ.get("DDX_SPMM") == "gather" and os.environ.get("DDX_PCA_GATHER") == "f64":
    np.save(ref_path, emb)
dev = orc.per_component_rel_dev(emb, np.load(ref_path)).max() if os.path.exists(ref_path) else float("nan")
tag = os.environ.get("DDX_LIB", "default").split("/")[-1] + " " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("DDX_SPMM") or k.startswith("DDX_PCA"))
print(f"{tag:50s} rows {t['spmm_rows'][1] / t['spmm_rows'][0]:.4f} ms  cols {t['spmm_cols'][1] / t['spmm_cols'][0]:.4f} ms   max dev vs f64 {dev:.2e}")
