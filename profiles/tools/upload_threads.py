"""upload_raw wall-clock against the number of packing threads (ddx_set_upload_threads), headline shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import _lib
from doubletdetection_amd.classifier import _HipEngine
from doubletdetection_amd._synthetic import make_counts
X = make_counts(100_000, 30_000, density=0.03116, device="cuda:0", seed=11)
os.environ["DDX_UPLOAD"] = "packed"
eng = _HipEngine(0)
c = eng.ctx
c.upload_raw(X)
for T in (8, 16, 24, 32, 48, 64, 96, 128, 48, 16):
    _lib.set_upload_threads(T)
    c.upload_raw(X)
    ts = []
    for rep in range(5):
        t0 = time.perf_counter(); c.upload_raw(X); ts.append(time.perf_counter() - t0)
    print(f"{T:4d} threads: upload_raw min {1e3 * min(ts):.2f} ms, median {1e3 * sorted(ts)[2]:.2f} ms")
