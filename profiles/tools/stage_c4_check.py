"""Where the `stage` part of a 500 000 x 33 000 fit goes: python profiles/tools/stage_c4_check.py"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from doubletdetection_amd import BoostClassifier, _lib, classifier
from doubletdetection_amd._synthetic import make_counts

X = make_counts(500000, 33000, density=0.02, seed=3, device="cuda:0")
print(type(X), X.nnz, flush=True)
marks = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            marks.append((name, 1e3 * (time.perf_counter() - t0)))
    setattr(obj, name, g)
for i in range(4):
    clf = BoostClassifier(n_iters=2, random_state=0)
    marks.clear()
    wrap(clf, "_coerce"); wrap(clf, "_engine_factory"); wrap(clf, "_check_device_limits"); wrap(clf, "_open_lanes")
    orig = classifier._HipEngine.stage_raw
    def timed(self, csr, _o=orig):
        t0 = time.perf_counter(); _o(self, csr); marks.append(("stage_raw", 1e3 * (time.perf_counter() - t0)))
    classifier._HipEngine.stage_raw = timed
    t0 = time.perf_counter(); clf.fit(X); t1 = time.perf_counter()
    classifier._HipEngine.stage_raw = orig
    print(f"fit {1e3*(t1-t0):.1f} ms", {k: round(v, 1) for k, v in clf._host_timings.items()}, [(n, round(v, 1)) for n, v in marks], flush=True)
