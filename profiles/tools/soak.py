import os, sys, time, resource
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from doubletdetection_amd import BoostClassifier, _lib
from doubletdetection_amd.classifier import _CONTEXT_POOL
from doubletdetection_amd._synthetic import make_counts
X = make_counts(20000, 8000, density=0.05, device="cuda:0", seed=3)
import warnings; warnings.simplefilter("ignore")
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
t0 = time.time()
for i in range(120):
    clf = BoostClassifier(n_iters=6, random_state=i, clustering_algorithm=("phenograph", "louvain", "leiden")[i % 3]).fit(X)
    lab = clf.predict()
    if i % 20 == 0:
        held = sum(c.device_bytes() for cs in _CONTEXT_POOL.values() for c in cs)
        free, total = torch.cuda.mem_get_info()
        print(f"fit {i:3d}: {time.time() - t0:6.1f} s, parked contexts {sum(len(v) for v in _CONTEXT_POOL.values())} holding {held / 2**30:.2f} GB, device used {(total - free) / 2**30:.2f} GB, host max RSS {rss():.0f} MB, doublets {int(np.nansum(lab))}")
