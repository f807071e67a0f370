"""Device memory held per context after a fit at the headline shape, and the number of contexts the automatic rule opens:
    python profiles/tools/context_bytes.py [key=value ...]   (options for ddx_set_option)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doubletdetection_amd import BoostClassifier, classifier, _lib
from doubletdetection_amd._synthetic import make_counts

for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.OPTIONS[k] = v
X = make_counts(100000, 30000, density=0.03, device=torch.device("cuda:0"), seed=20250227)
for rep in range(3):
    t0 = time.perf_counter()
    clf = BoostClassifier(random_state=0).fit(X)
    t1 = time.perf_counter()
    pool = classifier._CONTEXT_POOL
    sizes = [round(c.device_bytes() / 2**30, 2) for d in pool.values() for c in d]
    print(f"fit {rep}: {t1 - t0:.3f} s; parked contexts (GiB): {sizes}", flush=True)
