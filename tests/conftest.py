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
os.environ.setdefault("DDX_ARENA_GUARD", "1")


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
