"""Scope times of the community-detection stages on the device (coarsening sweeps, refinement) for the library in place, on the benchmark
embedding's graph (A/B pattern of profiles/tools/ab.sh).   python profiles/tools/coarsen_time.py [label]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from doubletdetection_amd import _lib
M = 125000
rng = np.random.default_rng(0)
centers = rng.normal(size=(12, 30)) * 6
emb = (centers[rng.integers(0, 12, M)] + rng.normal(size=(M, 30))).astype(np.float32)
ctx = _lib.Context(0)
ctx.timing_enable(True)
ctx.set_embedding(emb)
ctx.knn(30, False)
ctx.build_graph(0, fetch=False)
for rep in range(3):
    ctx.timing_reset()
    m_dev, ip, ix, w = ctx.coarsen_graph(1.0)
    ctx.synchronize()
t = ctx.timings()
print(sys.argv[1] if len(sys.argv) > 1 else "", {k: round(v[1], 3) for k, v in t.items() if k.startswith("graph")}, "ms; checksum", int(np.sum(m_dev.astype(np.int64) * (1 + np.arange(len(m_dev)) % 5))), len(ip) - 1)
ctx.close()
