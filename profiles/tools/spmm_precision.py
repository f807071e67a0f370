"""Precision of the randomized PCA through the variants of the operator products, at BASELINE configs[1] size
(50k x 20k, 5 % nnz).  Each variant runs in its own process (the switches are read once):
  f64      DDX_SPMM=gather DDX_PCA_GATHER=f64   float64 operand, float64 products (the yardstick)
  gather   DDX_SPMM=gather                      float32 operand copy, float64 products
  lds64    DDX_SPMM_TRIP=f64                    LDS-staged float32 operand, float64 products
  lds32    (default)                            LDS-staged float32 operand, float32 products in trips of 8
Usage: python profiles/tools/spmm_precision.py            (driver: runs the four and prints the table)"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

VARIANTS = {"f64": {"DDX_SPMM": "gather", "DDX_PCA_GATHER": "f64"}, "gather": {"DDX_SPMM": "gather"},
            "lds64": {"DDX_SPMM_TRIP": "f64"}, "lds32": {"DDX_SPMM_TRIP": "f32"}}
OUT = os.environ.get("DDX_PRECISION_DIR", "/tmp/ddx_precision")


def run_one(name):
    import torch  # noqa: F401  (HIP runtime first)
    from doubletdetection_amd import _lib
    from doubletdetection_amd._synthetic import make_counts

    N, G = 50_000, 20_000
    data = make_counts(N, G, density=0.05, device="cuda:0", seed=11)
    ctx = _lib.Context(0)
    ctx.upload_raw(data)
    top = np.argsort(ctx.gene_variances())[-10000:]
    ctx.select_columns(top)
    parents = np.random.default_rng(0).choice(N, size=(N // 4, 2), replace=False)
    ctx.create_doublets(parents)
    ctx.lognormalise(0.1)
    q0 = np.random.RandomState(0).normal(size=(ctx.H, 40)).astype(np.float32).astype(np.float64)
    ctx.pca(30, q0)
    emb, sing = ctx.embedding_f64()
    ctx.knn(30, False)
    idx = ctx.get_knn()[0]
    np.savez(os.path.join(OUT, name + ".npz"), emb=emb, sing=sing, idx=idx)
    ctx.close()


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:
        return run_one(sys.argv[1])
    for name, env in VARIANTS.items():
        e = dict(os.environ)
        for k in ("DDX_SPMM", "DDX_PCA_GATHER", "DDX_SPMM_TRIP"):
            e.pop(k, None)
        e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), name], check=True, env=e)
    ref = np.load(os.path.join(OUT, "f64.npz"))
    print("variant   max rel err of a component (2-norm)   max rel err of a singular value   kNN rows identical to f64")
    for name in ("gather", "lds64", "lds32"):
        v = np.load(os.path.join(OUT, name + ".npz"))
        sign = np.sign((v["emb"] * ref["emb"]).sum(axis=0))
        rel = np.linalg.norm(v["emb"] * sign - ref["emb"], axis=0) / np.linalg.norm(ref["emb"], axis=0)
        rs = np.abs(v["sing"] - ref["sing"]) / ref["sing"]
        same = np.mean((np.sort(v["idx"], axis=1) == np.sort(ref["idx"], axis=1)).all(axis=1))
        print(f"{name:8s}  {rel.max():.3e} (component {rel.argmax()}; first 12: {rel[:12].max():.3e})   {rs.max():.3e}   {same:.4%}")


if __name__ == "__main__":
    main()
