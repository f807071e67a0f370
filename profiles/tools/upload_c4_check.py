import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from doubletdetection_amd import _lib
from doubletdetection_amd._synthetic import make_counts
_lib.OPTIONS["upload_debug"] = "1"
X = make_counts(500000, 33000, density=0.02, seed=3)
print(type(X), X.dtype, X.indices.dtype, X.indptr.dtype, X.nnz, flush=True)
c = _lib.Context(0)
for i in range(5):
    t0 = time.perf_counter(); c.upload_raw(X); c.synchronize(); t1 = time.perf_counter()
    v = c.gene_variances(); t2 = time.perf_counter()
    print(f"upload_raw {1e3*(t1-t0):.1f} ms, gene_variances {1e3*(t2-t1):.1f} ms", flush=True)
