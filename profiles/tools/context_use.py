"""What the contexts of a fit reserve (chunks sized from an estimate at upload) against what their buffers occupied at the peak
(ddx_arena_peak): python profiles/tools/context_use.py [cells genes density [streams]]   (default: configs[3]'s shape)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doubletdetection_amd import BoostClassifier, classifier
from doubletdetection_amd._synthetic import make_counts

N, G, dens = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (500_000, 33_000, 0.02)
streams = int(sys.argv[4]) if len(sys.argv) > 4 else None
X = make_counts(N, G, density=dens, device=torch.device("cuda:0"), seed=20250227)
for rep in range(3):
    t0 = time.perf_counter()
    clf = BoostClassifier(random_state=0, streams_per_device=streams).fit(X)
    t1 = time.perf_counter()
    pool = classifier._CONTEXT_POOL
    rows = [(round(c.device_bytes() / 2**30, 2), round(c.arena_peak() / 2**30, 2), round(c.follower_bytes() / 2**30, 2)) for d in pool.values() for c in d]
    print(f"fit {rep}: {t1 - t0:.3f} s; parked contexts (GiB reserved, GiB at the peak, follower estimate): {rows}", flush=True)
