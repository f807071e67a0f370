"""kNN of the benchmark embedding (built by the device pipeline itself) for several cell counts: stage times, pruning
statistics and an all-query comparison with a float64 brute force in torch (the oracle's arithmetic: subtract, square, add
component by component).   python profiles/tools/knn_cells_check.py N G density [cells,cells,...] [check_queries]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

N, G, DENS = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
cells = [int(c) for c in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 0]
n_check = int(sys.argv[5]) if len(sys.argv) > 5 else -1
from doubletdetection_amd import _lib
from doubletdetection_amd._synthetic import make_counts

X = make_counts(N, G, density=DENS, device="cuda:0")
_lib.OPTIONS["knn_debug"] = os.environ.get("KNN_DEBUG", "1")


def embedding():
    ctx = _lib.Context(0)
    ctx.upload_raw(X)
    var = ctx.gene_variances()
    ctx.select_columns(np.argsort(var)[-10000:])
    S = N // 4
    ctx.create_doublets(np.random.default_rng(0).choice(N, size=(S, 2), replace=False))
    ctx.lognormalise(0.1)
    q0 = np.random.RandomState(0).normal(size=(ctx.H, 40)).astype(np.float32).astype(np.float64)
    ctx.pca(30, q0)
    e = ctx.embedding()
    ctx.close()
    return e


emb = embedding()
M, C = emb.shape
print("embedding", emb.shape, "std", np.round(emb.std(axis=0)[[0, 1, 11, 12, 29]], 2), flush=True)


def brute(queries, k=30):
    E = torch.from_numpy(emb).to("cuda:0", torch.float64)
    out_i = np.empty((len(queries), k), np.int64)
    out_d = np.empty((len(queries), k))
    B = 1024
    for s in range(0, len(queries), B):
        q = torch.from_numpy(queries[s:s + B]).to("cuda:0")
        d2 = torch.zeros((len(q), M), dtype=torch.float64, device="cuda:0")
        for c in range(C):
            diff = E[q, c][:, None] - E[None, :, c]
            d2 += diff * diff
        d2[torch.arange(len(q)), q] = float("inf")
        v, i = torch.topk(d2, k + 2, dim=1, largest=False)
        v, i = v.cpu().numpy(), i.cpu().numpy()
        for r in range(len(q)):
            o = np.lexsort((i[r], v[r]))[:k]
            out_i[s + r], out_d[s + r] = i[r][o], v[r][o]
    return out_i, out_d


ref = None
for kc in cells:
    _lib.OPTIONS["knn_cells"] = str(kc)
    ctx = _lib.Context(0)
    ctx.timing_enable(True)
    ctx.set_embedding(emb)
    for rep in range(3):
        ctx.timing_reset()
        t0 = time.perf_counter()
        ctx.knn(30, False)
        ctx.synchronize()
        wall = time.perf_counter() - t0
        if rep == 2:
            tm = {k: round(v[1], 3) for k, v in ctx.timings().items() if k.startswith("knn")}
            print("cells=%d wall %.2f ms  %s  screened %.3f  overflow %d" % (kc, wall * 1e3, tm, ctx.knn_window_fraction(), ctx.knn_overflow_count()), flush=True)
    idx, dist = ctx.get_knn()
    ctx.close()
    if n_check != 0:
        qs = np.arange(M) if n_check < 0 else np.sort(np.random.default_rng(5).choice(M, size=n_check, replace=False))
        if ref is None:
            t0 = time.time()
            ref = brute(qs)
            print("brute force of %d queries: %.1f s" % (len(qs), time.time() - t0), flush=True)
        bad_i = int((idx[qs] != ref[0]).any(axis=1).sum())
        bad_d = int((dist[qs] != np.sqrt(ref[1])).any(axis=1).sum())
        print("cells=%d: rows with different indices %d, different distances %d of %d" % (kc, bad_i, bad_d, len(qs)), flush=True)
