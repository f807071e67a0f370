"""kNN stage timings for the library named by DDX_LIB (and DDX_KNN_FOLD): bound / emit / select per launch at the headline
size and at a c4-like size, plus a checksum of the neighbour table (every variant must print the same one)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import _lib
import clustering_cases as cc
for M in (125_000, 625_000):
    emb = cc.make_embedding(M, 3)
    ctx = _lib.Context(0)
    ctx.timing_enable(True)
    ctx.set_embedding(emb)
    for rep in range(3):
        ctx.timing_reset()
        ctx.knn(30, False)
        ctx.synchronize()
        t = {k: round(v[1], 3) for k, v in ctx.timings().items()}
    idx, _ = ctx.get_knn(with_dist=False)
    print(os.environ.get("DDX_LIB", "default").split("/")[-1], "fold=" + os.environ.get("DDX_KNN_FOLD", "1"), "xcd=" + os.environ.get("DDX_KNN_XCD_CHUNK", "0"), M, t,
          "window", round(ctx.knn_window_fraction(), 3), "crc", zlib.crc32(idx.tobytes()))
    ctx.close()
