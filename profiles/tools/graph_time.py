"""Scope times of the graph stage (Jaccard weights, assembly) on the benchmark embedding, for the library in place (A/B: profiles/tools/ab.sh
pattern).   python profiles/tools/graph_time.py [label]"""
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
for rep in range(3):
    ctx.timing_reset()
    ctx.build_graph(0, fetch=False)
    ctx.synchronize()
t = ctx.timings()
ip, ix, w = ctx.fetch_graph()
print(sys.argv[1] if len(sys.argv) > 1 else "", {k: round(v[1] / max(v[0], 1), 4) for k, v in t.items() if k.startswith("graph")}, "checksum", float(np.sum(w * (1 + (ix % 7)))), len(ix))
ctx.close()
