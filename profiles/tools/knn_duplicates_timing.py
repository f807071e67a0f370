"""kNN time with many identical points (empty cells embed identically): python profiles/tools/knn_duplicates_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from doubletdetection_amd import _lib
rng = np.random.default_rng(0)
M = 125_000
mu = rng.normal(size=(12, 30)) * 6
base = (mu[rng.integers(0, 12, size=M)] + rng.normal(size=(M, 30))).astype(np.float32)
for ndup in (0, 2_000, 20_000, 60_000):
    e = base.copy()
    if ndup:
        e[10_000:10_000 + ndup] = e[3]
    c = _lib.Context(0)
    c.set_embedding(e)
    c.knn(30, False); c.synchronize()
    t0 = time.perf_counter(); c.knn(30, False); c.synchronize(); dt = time.perf_counter() - t0
    print(f"{ndup:6d} identical points of {M}: kNN {dt * 1e3:8.2f} ms, overflowed lists {c.knn_overflow_count()}", flush=True)
    c.close()
