"""Average launch time of the two operator products inside one randomized PCA at the headline size
(100k x 30k, 3 % nnz); tolerant of wrong numerics (ablation builds)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401
from doubletdetection_amd import _lib  # noqa: E402
from doubletdetection_amd._synthetic import make_counts  # noqa: E402

N, G = 100_000, 30_000
data = make_counts(N, G, density=0.03, device="cuda:0", seed=20250227)
ctx = _lib.Context(0)
ctx.upload_raw(data)
top = np.argsort(ctx.gene_variances())[-10000:]
ctx.select_columns(top)
ctx.create_doublets(np.random.default_rng(0).choice(N, size=(N // 4, 2), replace=False))
ctx.lognormalise(0.1)
q0 = np.random.RandomState(0).normal(size=(ctx.H, 40)).astype(np.float32).astype(np.float64)
ctx.timing_enable(True)
for rep in range(2):
    ctx.timing_reset()
    try:
        ctx.pca(30, q0)
    except Exception as e:  # noqa: BLE001
        print("pca:", str(e)[:80])
t = ctx.timings()
out = {k: round(v[1] / max(v[0], 1), 4) for k, v in t.items() if k.startswith("spmm") or k.startswith("bitplane")}
try:
    ctx.timing_reset()
    ctx.knn(30, False)
    ctx.synchronize()
    out.update({k: round(v[1] / max(v[0], 1), 4) for k, v in ctx.timings().items() if k.startswith("knn")})
except Exception as e:  # noqa: BLE001
    print("knn:", str(e)[:80])
print(sys.argv[1] if len(sys.argv) > 1 else "", out)
ctx.close()
