import sys, warnings, numpy as np
sys.path.insert(0, "/root/repo")
import torch; torch.cuda.init()
from doubletdetection_amd import BoostClassifier, _lib
from doubletdetection_amd._synthetic import make_counts
_lib.OPTIONS["arena_guard"] = "1"
data = make_counts(8192, 6000, density=0.08, n_types=8, doublet_frac=0.08, seed=606)
for lanes in (1, 7):
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            clf = BoostClassifier(streams_per_device=lanes, n_iters=7, random_state=0).fit(data)
        print("lanes", lanes, "ok", clf._last_bitplane)
    except Exception as e:
        print("lanes", lanes, "FAILED:", e)
