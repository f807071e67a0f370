"""Time of the column-major mirror build for the library named by DDX_LIB (experiment builds give wrong results)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import _lib
from doubletdetection_amd._synthetic import make_counts
X = make_counts(100_000, 30_000, density=0.03, device="cuda:0", seed=20250227)
c = _lib.Context(0)
c.timing_enable(True)
c.upload_raw(X); var = c.gene_variances(); c.select_columns(np.argsort(var)[-10000:])
t = c.timings()
print("orig", {k: round(v[1] / max(v[0], 1), 4) for k, v in t.items() if k.startswith("mirror")})
rng = np.random.default_rng(0)
c.timing_reset()
for rep in range(6):
    c.create_doublets(rng.choice(100_000, size=(25_000, 2), replace=False)); c.lognormalise(0.1)
t = c.timings()
print("synth", {k: round(v[1] / max(v[0], 1), 4) for k, v in t.items() if k.startswith(("mirror", "lognorm", "doublet"))})
if os.environ.get("MIRROR_CHECK"):
    q = np.random.RandomState(1).normal(size=(c.M, 8))
    y = c.operator_apply(q, 1)
    print("checksum", float(np.abs(y).sum()), float((y * np.arange(1, 9)).sum()))
