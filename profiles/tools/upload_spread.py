"""Spread of ddx_upload_raw over 35 consecutive calls at the headline shape (packed form): python profiles/tools/upload_spread.py
(end of round 5: min 6.5 - 7.5, median 6.5 - 7.6, p90 6.6 - 12.4, max 29 - 35 ms over three runs on one box; drawing the pieces of a chunk
from one counter instead of giving every thread a fixed share was no better: median 6.8 - 8.1, p90 9.9 - 47 ms; neither were pieces the
waiting thread may take over from a pool thread that wakes late, nor chunk events kept between calls.  One call in ~11 spends 20 - 30 ms
in the packing phase whatever the variant: nothing in libddx is periodic at that rate.)"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from doubletdetection_amd import _lib
from doubletdetection_amd._synthetic import make_counts
X = make_counts(100_000, 30_000, density=0.03, device="cuda:0", seed=20250227)
_lib.OPTIONS["upload"] = "packed"
c = _lib.Context(0)
ts = []
for rep in range(40):
    t0 = time.perf_counter(); c.upload_raw(X); ts.append(1e3 * (time.perf_counter() - t0))
ts = np.array(ts[5:])
print(os.environ.get("DDX_LIB", "main")[-12:], "upload_raw ms: min %.2f median %.2f p90 %.2f max %.2f" % (ts.min(), np.median(ts), np.percentile(ts, 90), ts.max()))
