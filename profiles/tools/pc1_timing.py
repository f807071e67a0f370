"""pseudocount = 1 (dd.py:296-297,308: sparse matrix, ARPACK) at BASELINE configs[1]: operator products per PCA, time per
product (by block width), seconds per PCA."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import _lib
from doubletdetection_amd.classifier import _HipEngine
from doubletdetection_amd._synthetic import make_counts
X = make_counts(50_000, 20_000, density=0.05, device="cuda:0", seed=11)
eng = _HipEngine(0)
c = eng.ctx
c.upload_raw(X); c.select_columns(np.argsort(c.gene_variances())[-10000:])
c.create_doublets(np.random.default_rng(0).choice(50_000, size=(12_500, 2), replace=False))
c.lognormalise(1.0)
M, H = c.M, c.H
for n in (1, 8, 40):
    x = np.random.default_rng(1).normal(size=(H, n))
    c.operator_apply(x, 2)
    t0 = time.perf_counter()
    for _ in range(20):
        c.operator_apply(x, 2)
    print(f"A^T A x, {n} vector(s): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call")
start = np.random.RandomState(0).normal(size=(H, 40))
results = {}
for ms in (8, 12, 16, 20, 24):
    c.pca_exact_sparse(30, start, tol=1e-9, max_steps=ms)
    t0 = time.perf_counter()
    st = c.pca_exact_sparse(30, start, tol=1e-9, max_steps=ms)
    results[ms] = (time.perf_counter() - t0, st, c.embedding().astype(np.float64))
# against scipy's ARPACK (what upstream runs) driven by the same device operator
from scipy.sparse.linalg import LinearOperator, eigsh
pad = np.zeros((H, 8))
def gram(x):
    pad[:, 0] = np.asarray(x, dtype=np.float64).ravel()
    return c.operator_apply(pad, 2)[:, 0]
t0 = time.perf_counter()
ev, V = eigsh(LinearOperator((H, H), dtype=np.float64, matvec=gram), k=30, tol=0.0, v0=np.random.RandomState(0).uniform(-1, 1, size=H), which="LM")
print(f"host ARPACK on the device operator: {time.perf_counter() - t0:.2f} s")
top = np.argsort(ev)[::-1]
V = V[:, top]
sg = np.sign(V[np.argmax(np.abs(V), axis=0), np.arange(30)])
ref = c.operator_apply(V * sg, 0)
for ms, (dt, st, emb) in results.items():
    rel = np.linalg.norm(emb - ref, axis=0) / np.linalg.norm(ref, axis=0)
    print(f"block Lanczos, {st:2d} steps: {dt * 1e3:7.1f} ms; scores against ARPACK's, relative per component: 1-12 max {rel[:12].max():.1e}, 13-30 max {rel[12:].max():.1e}")
# the default path's randomized PCA on the same matrix, for scale
q0 = np.random.RandomState(0).normal(size=(H, 40)).astype(np.float32).astype(np.float64)
c.pca(30, q0)
t0 = time.perf_counter()
c.pca(30, q0); c.synchronize()
print(f"randomized PCA (default path): {(time.perf_counter() - t0) * 1e3:.1f} ms")

for tol in (1e-5, 1e-6, 1e-7):
    c.pca_exact_sparse(30, start, tol=tol, max_steps=40)
    t0 = time.perf_counter()
    st = c.pca_exact_sparse(30, start, tol=tol, max_steps=40)
    dt = time.perf_counter() - t0
    emb = c.embedding().astype(np.float64)
    rel = np.linalg.norm(emb - ref, axis=0) / np.linalg.norm(ref, axis=0)
    print(f"tol {tol:.0e}: {st} steps, {dt * 1e3:.1f} ms, scores against ARPACK's max {rel.max():.1e}")
