"""How much do boosting iterations gain from running on several contexts (streams) of ONE GPU at once?
Each lane = one ddx_ctx driven by its own host thread; iterations are independent (dd.py:192-198).
    python profiles/tools/lanes_experiment.py [cells genes density]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (runtime order, see _lib.load)
from doubletdetection_amd import _lib
from doubletdetection_amd._synthetic import make_counts

N, G, D = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (100_000, 30_000, 0.03)
X = make_counts(N, G, density=D, device="cuda:0", seed=20250227)
rng = np.random.default_rng(0)
ITERS = 12
parents = [rng.choice(N, size=(N // 4, 2), replace=False) for _ in range(ITERS)]
q0 = np.random.RandomState(0).normal(size=(10000, 40)).astype(np.float32).astype(np.float64)


def make_ctx():
    c = _lib.Context(0)
    c.upload_raw(X)
    var = c.gene_variances()
    c.select_columns(np.argsort(var)[-10000:])
    return c


PCA_LOCK = threading.Lock() if os.environ.get("PCA_LOCK") else None


def iteration(c, p, first):
    c.create_doublets(p)
    c.lognormalise(0.1)
    if PCA_LOCK:
        PCA_LOCK.acquire()
    try:
        if first:
            c.pca(30, q0)
        else:
            c.pca(30, None, q0_rows=10000)
    finally:
        if PCA_LOCK:
            PCA_LOCK.release()
    c.knn(30, False)
    c.build_graph(0, fetch=False)
    c.coarsen_graph(1.0)


for lanes in (1, 2, 3):
    ctxs = [make_ctx() for _ in range(lanes)]
    for c in ctxs:                       # warm-up: allocations, start matrix
        iteration(c, parents[0], True)
        c.synchronize()

    def work(k):
        c = ctxs[k]
        for i in range(k, ITERS, lanes):
            iteration(c, parents[i], False)
        c.synchronize()

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(lanes)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"lanes={lanes}: {ITERS} iterations in {dt*1e3:.1f} ms = {dt/ITERS*1e3:.2f} ms per iteration", flush=True)
    for c in ctxs:
        c.close()
