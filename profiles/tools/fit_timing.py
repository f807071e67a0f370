"""Wall time of consecutive fits with per-phase host timings (diagnostics).  python profiles/tools/fit_timing.py [streams] [resident]"""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from doubletdetection_amd import BoostClassifier
from doubletdetection_amd._synthetic import make_counts

from doubletdetection_amd import classifier as _cl

def _wrap(name):
    fn = getattr(_cl._HipEngine, name)
    def timed(self, *a, **k):
        t = time.perf_counter()
        r = fn(self, *a, **k)
        if os.environ.get("TRACE_ENGINE"):
            print(f"    engine.{name}: {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
        return r
    setattr(_cl._HipEngine, name, timed)

for _n in ("__init__", "stage_raw", "gene_variances", "select_columns", "clone_from", "close"):
    _wrap(_n)

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
resident = len(sys.argv) > 2 and sys.argv[2] == "resident"
X = make_counts(100_000, 30_000, density=0.03, device="cuda:0", seed=20250227)
torch.cuda.empty_cache()
for step in range(7):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = BoostClassifier(n_iters=10, random_state=0, n_jobs=-1, streams_per_device=streams)
        if resident:
            clf.stage(X)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        clf.fit(X)
        dt = time.perf_counter() - t0
    print(f"step {step}: {dt*1e3:.1f} ms", {k: round(v * 1e3, 1) for k, v in clf._host_timings.items()}, flush=True)
