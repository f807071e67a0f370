"""Wall-clock of ddx_upload_raw (dd.py:149-160 on the way in) at the headline shape, packed and plain, against the link's own rate
(profiles/tools/pcie_bench.cpp), and of the prologue calls that follow it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd.classifier import _HipEngine
from doubletdetection_amd._synthetic import make_counts
X = make_counts(100_000, 30_000, density=0.03116, device="cuda:0", seed=11)
print("nnz", X.nnz)
for mode in ("packed", "packed32", "packed", "packed32", "plain"):
    os.environ["DDX_UPLOAD"] = mode
    os.environ["DDX_UPLOAD_DEBUG"] = os.environ.get("UPLOAD_DEBUG_LEVEL", "1")
    eng = _HipEngine(0)
    c = eng.ctx
    for rep in range(4):
        t0 = time.perf_counter(); c.upload_raw(X); t1 = time.perf_counter()
        v = c.gene_variances(); t2 = time.perf_counter()
        c.select_columns(np.argsort(v)[-10000:]); t3 = time.perf_counter()
        print(f"{mode}: upload_raw {1e3 * (t1 - t0):.2f} ms, gene_variances {1e3 * (t2 - t1):.2f} ms, argsort + select_columns {1e3 * (t3 - t2):.2f} ms")
    del eng, c
