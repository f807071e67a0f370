"""Round 6: launch time of the matrix-core product kernels against the number of stages a chunk runs (option bp_dbg_sk: timing only, wrong
results), for the int8 and the MX (FP4 x FP6) form: separates a launch's fixed cost from its per-stage cost.   Needs a library built with -DDDX_ABLATION
(bash profiles/tools/build_variant.sh ablation -DDDX_ABLATION; DDX_LIB=... python profiles/tools/mx_stage_sweep.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import warnings
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import _lib
from doubletdetection_amd._synthetic import make_counts
X = make_counts(100_000, 30_000, density=0.03, device="cuda:0", seed=20250227)
top = None
for fmt in ("int8", "mx6"):
    for sk in (0, 30, 20, 10, 5):
        _lib.OPTIONS["bp_format"] = fmt
        _lib.OPTIONS["bp_dbg_sk"] = str(sk)
        c = _lib.Context(0)
        c.upload_raw(X)
        if top is None:
            top = np.argsort(c.gene_variances())[-10000:]
        else:
            c.gene_variances()
        c.select_columns(top)
        c.create_doublets(np.random.default_rng(0).choice(100_000, size=(25_000, 2), replace=False)); c.lognormalise(0.1)
        q0 = np.random.RandomState(0).normal(size=(10000, 40)).astype(np.float32).astype(np.float64)
        c.timing_enable(True)
        for rep in range(2):
            c.timing_reset()
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    c.pca(30, q0)
            except Exception as e:
                pass
            t = c.timings()
        print(fmt, "stages per chunk", sk or "all", {k: round(1e3 * v[1] / max(v[0], 1), 1) for k, v in t.items() if k.startswith("bitplane")}, "us per launch", flush=True)
        c.close()
