"""Cost of the live HIP-event timing (DDX_TIMING=1, what bench.py's roofline block reads) on a whole fit."""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401
from doubletdetection_amd import BoostClassifier  # noqa: E402
from doubletdetection_amd._synthetic import make_counts  # noqa: E402

X = make_counts(100000, 30000, density=0.03, device="cuda:0", seed=20250227)
for mode in ("0", "1", "0", "1"):
    if mode == "1":
        os.environ["DDX_TIMING"] = "1"
    else:
        os.environ.pop("DDX_TIMING", None)
    ts = []
    for rep in range(4):
        clf = BoostClassifier(n_iters=10, random_state=0, n_jobs=-1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            clf.stage(X)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            clf.fit(X)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    print("DDX_TIMING=" + mode, [round(t * 1e3, 1) for t in ts])
