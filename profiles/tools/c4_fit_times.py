"""Fit-by-fit times of the 500 000 x 33 000 configuration in one process (what the first fits pay for: pinning the staging buffer,
growing the contexts): python profiles/tools/c4_fit_times.py [n_fits]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from doubletdetection_amd import BoostClassifier, _lib
from doubletdetection_amd._synthetic import make_counts

X = make_counts(500000, 33000, density=0.02, seed=3, device="cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
for i in range(n):
    clf = BoostClassifier(n_iters=10, random_state=0)
    t0 = time.perf_counter(); clf.fit(X); dt = time.perf_counter() - t0
    ht = clf._host_timings
    from doubletdetection_amd import classifier as _cl
    sizes = [round(c.device_bytes() / 2**30, 1) for d in _cl._CONTEXT_POOL.values() for c in d]
    print(f"fit {i}: {dt * 1e3:7.1f} ms  contexts (GiB) {sizes}  stage {ht['stage'] * 1e3:6.1f}  prologue {ht['prologue'] * 1e3:5.1f}  device stages {ht['device_stages'] * 1e3:6.1f}  close {ht['close'] * 1e3:5.1f}", flush=True)
