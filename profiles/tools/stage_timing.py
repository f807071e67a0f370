import sys, time, warnings
sys.path.insert(0, '/root/repo')
import numpy as np
from doubletdetection_amd import BoostClassifier
from doubletdetection_amd._synthetic import make_counts
X = make_counts(100000, 30000, density=0.03, device="cuda:0", seed=20250227)
print(type(X), X.dtype, X.indices.dtype, X.indptr.dtype, X.nnz)
for rep in range(3):
    clf = BoostClassifier(n_iters=10, random_state=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter(); csr = clf._coerce(X); t1 = time.perf_counter()
        clf.stage(X); t2 = time.perf_counter()
        clf.fit(X); t3 = time.perf_counter()
        clf2 = BoostClassifier(n_iters=10, random_state=0)
        t4 = time.perf_counter(); clf2.fit(X); t5 = time.perf_counter()
    print(f"coerce {t1-t0:.3f}  stage(incl coerce) {t2-t1:.3f}  fit(staged) {t3-t2:.3f}  fit(unstaged) {t5-t4:.3f}")
