import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
from doubletdetection_amd import _lib
M = 125000
rng = np.random.default_rng(0)
centers = rng.normal(size=(12, 30)) * 6
emb = (centers[rng.integers(0, 12, M)] + rng.normal(size=(M, 30))).astype(np.float32)
ctx = _lib.Context(0)
ctx.timing_enable(True)
ctx.set_embedding(emb)
for rep in range(3):
    ctx.timing_reset()
    ctx.knn(30, False)
    ctx.synchronize()
    print({k: round(v[1], 3) for k, v in ctx.timings().items()})
