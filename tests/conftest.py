import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# every device context made by the tests carries pattern-filled pads behind its buffers; BoostClassifier checks them when
# a fit ends, the module-level contexts of the stage tests when they are closed (tests/test_gpu_*.py)
# (process-wide option of the package's contexts -- doubletdetection_amd._lib.OPTIONS -- not an environment variable)
from doubletdetection_amd import _lib as _ddx_lib  # noqa: E402

_ddx_lib.OPTIONS.setdefault("arena_guard", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """PyTorch ships its own copy of the HIP runtime; libddx.so links the system one.  Both work in one process when
    torch initialises its runtime first (bench.py's order) -- the other way round torch later reports "no
    ROCm-capable device".  Some GPU tests use torch for data generation, so initialise it before any test touches
    libddx, whatever order the tests were selected in.  No-op on a machine without a GPU."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:      # torch absent or unusable: the tests that need it will say so
        pass
    yield


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def csr_from(g, prefix):
    shape = tuple(int(x) for x in g[prefix + "_shape"])
    return sp.csr_matrix((g[prefix + "_data"], g[prefix + "_indices"], g[prefix + "_indptr"]), shape=shape)


def golden_kwargs(g):
    """Constructor kwargs a golden case was generated with (stored as strings)."""
    import ast

    out = {}
    for k, v in zip(g["kw_keys"].tolist(), g["kw_vals"].tolist()):
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = v
    return out


CASES = ["case_a_hvg_pheno", "case_b_transposed_louvain", "case_c_reftest_scaled",
         "case_d_replace_single", "case_e_pc1_sparse"]


@pytest.fixture(params=CASES)
def golden_case(request):
    g = load_golden(request.param)
    return request.param, g, golden_kwargs(g)


def pca_against_f64_oracle(ctx, C=30, seed=0, chunk=4096):
    """The device's randomized PCA at a BASELINE size against the float64 oracle on the SAME matrix: the augmented matrix
    handed to PCA is read back densified (M x H float32, chunk by chunk), `orc.randomized_pca_f64` (sklearn's algorithm
    in float64, oracle/dd_oracle.py) and sklearn's own PCA in float64 run on the host cores.  Bars: <= 1e-5 relative per
    component against the float64 oracle, <= 1e-4 against sklearn (north star: 1e-4)."""
    import time

    from oracle import dd_oracle as orc

    M, H = ctx.M, ctx.H
    q0 = orc.pca_start_matrix(seed, H if M >= H else M, C + 10)
    ctx.pca(C, q0)
    emb, sing = ctx.embedding_f64()
    t0 = time.perf_counter()
    dense = np.empty((M, H), dtype=np.float32)
    for r in range(0, M, chunk):
        n = min(chunk, M - r)
        dense[r:r + n] = ctx.aug_dense_rows(r, n)
    t1 = time.perf_counter()
    want, s_want, _ = orc.randomized_pca_f64(dense, C, seed)
    t2 = time.perf_counter()
    rel = orc.per_component_rel_dev(emb, want)
    print(f"PCA {M} x {H}: device vs float64 oracle, relative deviation per component max {rel.max():.2e} (mean {rel.mean():.2e}); "
          f"singular values max rel {np.abs(sing / s_want - 1).max():.2e}; read-back {t1 - t0:.1f} s, oracle {t2 - t1:.1f} s")
    assert rel.max() <= 1e-5, rel
    np.testing.assert_allclose(sing, s_want, rtol=1e-6)
    del want
    skl64 = orc.pca_sklearn(dense.astype(np.float64), C, seed)
    rel_skl = orc.per_component_rel_dev(emb, skl64)
    print(f"  vs sklearn PCA(svd_solver='auto') in float64: max {rel_skl.max():.2e} ({time.perf_counter() - t2:.1f} s)")
    assert rel_skl.max() <= 1e-4, rel_skl
    return rel.max(), rel_skl.max()
