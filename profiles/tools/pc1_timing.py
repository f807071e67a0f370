"""pseudocount = 1 (dd.py:296-297,308: sparse matrix, ARPACK) at BASELINE configs[1]: seconds per fit and per iteration."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import BoostClassifier
from doubletdetection_amd._synthetic import make_counts
X = make_counts(50_000, 20_000, density=0.05, device="cuda:0", seed=11)
for pc in (1.0, 0.1):
    for rep in range(2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            clf = BoostClassifier(n_iters=2, pseudocount=pc, random_state=0, n_jobs=-1, streams_per_device=1).fit(X)
            dt = time.perf_counter() - t0
    print(f"pseudocount={pc}: 2 iterations in {dt:.3f} s ({X.shape[0] * 1 / dt:.0f} cells/s at n_iters=2)")
