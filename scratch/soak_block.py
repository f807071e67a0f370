import os, sys, time, faulthandler
sys.path.insert(0, os.getcwd())
faulthandler.enable()
import numpy as np, torch
from doubletdetection_amd import BoostClassifier, _lib
from doubletdetection_amd._synthetic import make_counts
_lib.OPTIONS["host_wait"] = sys.argv[1] if len(sys.argv) > 1 else "block"
nfits = int(sys.argv[2]) if len(sys.argv) > 2 else 120
X = make_counts(20000, 8000, density=0.05, device="cuda:0", seed=3)
import warnings; warnings.simplefilter("ignore")
t0 = time.time()
for i in range(nfits):
    clf = BoostClassifier(n_iters=6, random_state=i, clustering_algorithm=("phenograph", "louvain", "leiden")[i % 3]).fit(X)
    if i % 20 == 0:
        print(f"fit {i}: {time.time() - t0:.1f} s", flush=True)
print(f"loop done {time.time() - t0:.1f} s; cpu {time.process_time():.1f} s", flush=True)
faulthandler.dump_traceback_later(40, exit=False)
