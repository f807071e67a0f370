"""Effect of PhenoGraph's best-of-restarts rule at the benchmark size: modularity and doublet calls vs a single run."""
import sys, os, warnings, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import BoostClassifier, _lib
from doubletdetection_amd._synthetic import make_counts

N, G, D = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (100_000, 30_000, 0.03)
X = make_counts(N, G, density=D, device="cuda:0", seed=20250227)
warnings.simplefilter("ignore")
# coarse graph of one iteration, then both rules on it
c = _lib.Context(0)
c.upload_raw(X); var = c.gene_variances(); c.select_columns(np.argsort(var)[-10000:])
c.create_doublets(np.random.default_rng(0).choice(N, size=(N // 4, 2), replace=False)); c.lognormalise(0.1)
q0 = np.random.RandomState(0).normal(size=(10000, 40)).astype(np.float32).astype(np.float64)
c.pca(30, q0); c.knn(30, False); c.build_graph(0, fetch=False)
member, ip, ix, w = c.coarsen_graph(1.0)
t = time.perf_counter(); lab1, q1 = _lib.louvain_sequential(ip, ix, w, 1.0, 0); t1 = time.perf_counter() - t
t = time.perf_counter(); labb, qb, runs = _lib.louvain_best_of(ip, ix, w, 1.0, 0, 1e-3, threads=20, presweeps=False); tb = time.perf_counter() - t
print(f"coarse graph: {len(ip) - 1} super-nodes, {len(ix)} entries")
print(f"single run   : Q = {q1:.6f}, {len(np.unique(lab1))} communities, {t1 * 1e3:.1f} ms")
print(f"best of {runs:3d}  : Q = {qb:.6f}, {len(np.unique(labb))} communities, {tb * 1e3:.1f} ms (20 host threads)")
c.close()
res = {}
for name, kw in (("best-of (default)", {}), ("single run (q_tol=inf)", {"clustering_kwargs": {"q_tol": float("inf")}})):
    t = time.perf_counter()
    clf = BoostClassifier(n_iters=10, random_state=0, n_jobs=-1, **kw).fit(X)
    dt = time.perf_counter() - t
    res[name] = (clf.predict(), np.ma.filled(clf.doublet_score(), np.nan), dt)
    print(f"{name}: fit {dt * 1e3:.0f} ms, {int(np.nansum(res[name][0]))} doublet calls")
a, b = res["best-of (default)"], res["single run (q_tol=inf)"]
both = ~np.isnan(a[0]) & ~np.isnan(b[0])
print("calls that differ:", int(np.sum(a[0][both] != b[0][both])), "of", int(both.sum()),
      "; score correlation", float(np.corrcoef(np.nan_to_num(a[1]), np.nan_to_num(b[1]))[0, 1]))
