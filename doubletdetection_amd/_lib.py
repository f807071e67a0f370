"""ctypes binding of libddx.so (include/ddx.h).  No torch, no numpy C-API: plain pointers and sizes.

The library is required: there is no CPU fallback.  ``load()`` raises if it is missing; creating a
``Context`` raises if no gfx950 GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DDX_LIB") or os.path.join(_HERE, "libddx.so")      # DDX_LIB: an experimental build (profiles/tools)
ABI_VERSION = 5

_lib = None

c_i64_p = C.POINTER(C.c_int64)
c_i32_p = C.POINTER(C.c_int32)
c_f32_p = C.POINTER(C.c_float)
c_f64_p = C.POINTER(C.c_double)


E_ARG = -1           # DDX_E_ARG of include/ddx.h
E_HIP = -2           # DDX_E_HIP
W_UNCONVERGED = 2    # DDX_W_UNCONVERGED
E_UNSUPPORTED = -5   # DDX_E_UNSUPPORTED


class DdxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libddx error {code}: {message}")
        self.code = code


_SIGNATURES = {
    "ddx_abi_version": (C.c_int, []),
    "ddx_last_error": (C.c_char_p, [C.c_void_p]),
    "ddx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ddx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ddx_destroy": (C.c_int, [C.c_void_p]),
    "ddx_synchronize": (C.c_int, [C.c_void_p]),
    "ddx_device_bytes": (C.c_int, [C.c_void_p, c_i64_p]),
    "ddx_device_memory": (C.c_int, [C.c_void_p, c_i64_p, c_i64_p]),
    "ddx_pack_rows16": (C.c_int, [C.c_int64, c_i64_p, c_i32_p, c_f32_p, C.POINTER(C.c_uint16), C.c_int64, c_i32_p, c_i32_p, c_f32_p, c_i64_p]),
    "ddx_check_memory": (C.c_int, [C.c_void_p]),
    "ddx_reserve_hint": (C.c_int, [C.c_void_p, C.c_int64]),
    "ddx_trim": (C.c_int, [C.c_void_p, C.c_int64]),
    "ddx_set_upload_threads": (C.c_int, [C.c_int32]),
    "ddx_set_helper_threads": (C.c_int, [C.c_int32]),
    "ddx_set_upload_share": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32]),
    "ddx_upload_raw": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, c_i64_p, c_i32_p, c_f32_p]),
    "ddx_gene_variances": (C.c_int, [C.c_void_p, c_f32_p]),
    "ddx_select_columns": (C.c_int, [C.c_void_p, c_i64_p, C.c_int32]),
    "ddx_upload_counts": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, c_i64_p, c_i32_p, c_f32_p]),
    "ddx_clone_counts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddx_get_counts_nnz": (C.c_int, [C.c_void_p, c_i64_p]),
    "ddx_get_counts": (C.c_int, [C.c_void_p, c_i64_p, c_i32_p, c_f32_p]),
    "ddx_get_lib_size": (C.c_int, [C.c_void_p, c_f32_p]),
    "ddx_get_normed": (C.c_int, [C.c_void_p, c_f32_p]),
    "ddx_create_doublets": (C.c_int, [C.c_void_p, C.c_int64, c_i64_p]),
    "ddx_get_synth_nnz": (C.c_int, [C.c_void_p, c_i64_p]),
    "ddx_get_synth": (C.c_int, [C.c_void_p, c_i64_p, c_i32_p, c_f32_p]),
    "ddx_lognormalise": (C.c_int, [C.c_void_p, C.c_float]),
    "ddx_get_aug_lib": (C.c_int, [C.c_void_p, c_f32_p, c_f32_p]),
    "ddx_get_aug_nnz": (C.c_int, [C.c_void_p, c_i64_p]),
    "ddx_get_aug_values": (C.c_int, [C.c_void_p, c_f32_p, c_f32_p]),
    "ddx_get_aug_dense_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, c_f32_p]),
    "ddx_scale": (C.c_int, [C.c_void_p, C.c_float]),
    "ddx_pca": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_f64_p, C.c_int64]),
    "ddx_operator_apply": (C.c_int, [C.c_void_p, C.c_int32, c_f64_p, C.c_int32, c_f64_p]),
    "ddx_get_embedding": (C.c_int, [C.c_void_p, c_f32_p]),
    "ddx_get_embedding_f64": (C.c_int, [C.c_void_p, c_f64_p, c_f64_p]),
    "ddx_set_embedding": (C.c_int, [C.c_void_p, c_f32_p, C.c_int64, C.c_int32]),
    "ddx_knn": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "ddx_knn_metric": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "ddx_get_knn": (C.c_int, [C.c_void_p, c_i32_p, c_f64_p]),
    "ddx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "ddx_pca_exact_sparse": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, c_f64_p, C.POINTER(C.c_int32), C.c_void_p,
                                       C.c_void_p]),
    "ddx_get_knn_window_fraction": (C.c_int, [C.c_void_p, c_f64_p]),
    "ddx_get_knn_overflow_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ddx_get_knn_candidate_counts": (C.c_int, [C.c_void_p, c_i32_p]),
    "ddx_get_bitplane_stats": (C.c_int, [C.c_void_p, c_i64_p]),
    "ddx_arena_peak": (C.c_int, [C.c_void_p, c_i64_p]),
    "ddx_get_upload_form": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "ddx_build_graph": (C.c_int, [C.c_void_p, C.c_int32]),
    "ddx_graph_relations": (C.c_int, [C.c_void_p, C.c_int32, c_i32_p, c_f64_p]),
    "ddx_assemble_graph": (C.c_int, [C.c_int64, C.c_int32, c_i32_p, c_f64_p, c_i64_p, c_i32_p, c_f64_p]),
    "ddx_get_graph_size": (C.c_int, [C.c_void_p, c_i64_p, c_i64_p]),
    "ddx_get_graph": (C.c_int, [C.c_void_p, c_i64_p, c_i32_p, c_f64_p]),
    "ddx_louvain": (C.c_int, [C.c_int64, c_i64_p, c_i32_p, c_f64_p, C.c_double, C.c_uint64, c_i32_p, c_f64_p]),
    "ddx_louvain_sequential": (C.c_int, [C.c_int64, c_i64_p, c_i32_p, c_f64_p, C.c_double, C.c_uint64, c_i32_p, c_f64_p]),
    "ddx_louvain_best_of": (C.c_int, [C.c_int64, c_i64_p, c_i32_p, c_f64_p, C.c_double, C.c_uint64, C.c_double, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, c_i32_p, c_f64_p, c_i32_p]),
    "ddx_leiden": (C.c_int, [C.c_int64, c_i64_p, c_i32_p, c_f64_p, C.c_double, C.c_uint64, c_i32_p]),
    "ddx_leiden_sequential": (C.c_int, [C.c_int64, c_i64_p, c_i32_p, c_f64_p, C.c_double, C.c_uint64, c_i32_p]),
    "ddx_presweep": (C.c_int, [C.c_int64, c_i64_p, c_i32_p, c_f64_p, C.c_double, C.c_int32, C.c_int32, c_i32_p, c_i64_p, c_i64_p, c_i32_p, c_f64_p]),
    "ddx_refine": (C.c_int, [C.c_int64, c_i64_p, c_i32_p, c_f64_p, c_i32_p, C.c_double, C.c_int32, C.c_int32, C.c_int32, c_i32_p]),
    "ddx_coarsen_graph": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_int32]),
    "ddx_refine_communities": (C.c_int, [C.c_void_p, c_i32_p, C.c_double, C.c_int32, c_i32_p]),
    "ddx_get_coarse_size": (C.c_int, [C.c_void_p, c_i64_p, c_i64_p]),
    "ddx_get_coarse_graph": (C.c_int, [C.c_void_p, c_i32_p, c_i64_p, c_i32_p, c_f64_p]),
    "ddx_relabel_by_size": (C.c_int, [C.c_int64, c_i32_p, C.c_int64, c_i64_p]),
    "ddx_hypergeom_logsf": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_f64_p]),
    "ddx_score_communities": (C.c_int, [c_i64_p, C.c_int64, C.c_int64, c_f64_p, c_f64_p]),
    "ddx_timing_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "ddx_timing_reset": (C.c_int, [C.c_void_p]),
    "ddx_timing_count": (C.c_int, [C.c_void_p, c_i32_p]),
    "ddx_timing_get": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, c_i64_p, c_f64_p]),
    "ddx_timing_reference": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddx_timing_intervals": (C.c_int, [C.c_void_p, C.c_int64, c_f64_p, c_i64_p]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))


def load():
    """dlopen libddx.so and declare every prototype of include/ddx.h.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m doubletdetection_amd._build` "
            "(there is no CPU fallback for the HIP path)")
    # PyTorch-ROCm bundles its own copy of the HIP runtime.  The two coexist in one process when torch's initialises
    # first; if torch is already imported, make sure it has done so before libddx touches the device.  (A process that
    # imports torch only *after* its first ddx context should call torch.cuda.init() before creating that context.)
    import sys

    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.ddx_abi_version() != ABI_VERSION:
        raise ImportError(f"libddx ABI {lib.ddx_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def _p(arr, ptype):
    return arr.ctypes.data_as(ptype)


def _check(rc, ctx=None):
    if rc == 0:
        return
    msg = load().ddx_last_error(ctx)
    msg = msg.decode() if msg else "unknown"
    if rc > 0:                      # DDX_W_*: the call succeeded, the library has something to say
        import warnings

        warnings.warn(f"libddx: {msg}", RuntimeWarning, stacklevel=3)
        return
    raise DdxError(rc, msg)


def device_count() -> int:
    n = C.c_int(0)
    rc = load().ddx_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def pack_rows16(indptr, indices, data, capacity=None):
    """Host side of the 2-byte transfer form (ddx_pack_rows16): (codes uint16[nnz], listed positions, columns, values)."""
    ip = np.ascontiguousarray(indptr, dtype=np.int64)
    ix = np.ascontiguousarray(indices, dtype=np.int32)
    d = np.ascontiguousarray(data, dtype=np.float32)
    nnz = int(ip[-1])
    cap = nnz if capacity is None else int(capacity)
    codes = np.zeros(max(nnz, 1), dtype=np.uint16)
    pos = np.zeros(max(cap, 1), dtype=np.int32)
    col = np.zeros(max(cap, 1), dtype=np.int32)
    val = np.zeros(max(cap, 1), dtype=np.float32)
    n = C.c_int64(0)
    _check(load().ddx_pack_rows16(len(ip) - 1, _p(ip, c_i64_p), _p(ix, c_i32_p), _p(d, c_f32_p), _p(codes, C.POINTER(C.c_uint16)), cap,
                                  _p(pos, c_i32_p), _p(col, c_i32_p), _p(val, c_f32_p), C.byref(n)))
    k = min(n.value, cap)
    return codes[:nnz], pos[:k], col[:k], val[:k], n.value


def set_upload_threads(n: int) -> None:
    """Host threads that pack the raw matrix for the upload (process-wide; 0 = the library's default)."""
    _check(load().ddx_set_upload_threads(int(n)))


def set_upload_share(name: str, local_rank: int = 0, local_world: int = 1) -> None:
    """One packing per node for one-process-per-GPU runs: the node's ranks share a POSIX shared-memory image of the packed matrix
    (include/ddx.h: ddx_set_upload_share).  Empty name / local_world <= 1: off."""
    _check(load().ddx_set_upload_share((name or "").encode(), int(local_rank), int(local_world)))


def set_helper_threads(n: int) -> None:
    """Helper threads the restart batches of louvain_best_of may have running in this process at any time (negative: no limit)."""
    _check(load().ddx_set_helper_threads(int(n)))


# ---- context-free host routines --------------------------------------------------------------
def louvain(indptr, indices, weights, gamma: float, seed: int):
    """Deterministic Louvain of libddx (host C++).  Returns (labels int32[n], quality)."""
    lib = load()
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    n = indptr.shape[0] - 1
    labels = np.empty(n, dtype=np.int32)
    q = C.c_double(0.0)
    _check(lib.ddx_louvain(n, _p(indptr, c_i64_p), _p(indices, c_i32_p), _p(weights, c_f64_p), float(gamma),
                           int(seed) & 0xFFFFFFFFFFFFFFFF, _p(labels, c_i32_p), C.byref(q)))
    return labels, q.value


PRESWEEPS = 6         # DDX_PRESWEEPS of include/ddx.h
PRESWEEP_LEVELS = 2   # DDX_PRESWEEP_LEVELS
SUBROUNDS = 2         # DDX_SUBROUNDS
REFINE_SWEEPS = 3     # DDX_REFINE_SWEEPS


def louvain_sequential(indptr, indices, weights, gamma: float, seed: int):
    """Part B of the specification only (sequential multi-level optimisation).  Returns (labels, quality)."""
    lib = load()
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    n = indptr.shape[0] - 1
    labels = np.empty(n, dtype=np.int32)
    q = C.c_double(0.0)
    _check(lib.ddx_louvain_sequential(n, _p(indptr, c_i64_p), _p(indices, c_i32_p), _p(weights, c_f64_p), float(gamma),
                                      int(seed) & 0xFFFFFFFFFFFFFFFF, _p(labels, c_i32_p), C.byref(q)))
    return labels, q.value


def louvain_best_of(indptr, indices, weights, gamma: float, seed: int, q_tol: float = 1e-3, stall: int = 20, max_runs: int = 1000,
                    threads: int = 1, presweeps: bool = True):
    """PhenoGraph's restart rule around the deterministic Louvain (ddx_louvain_best_of).  Returns (labels, quality, runs)."""
    lib = load()
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    n = indptr.shape[0] - 1
    labels = np.empty(n, dtype=np.int32)
    q = C.c_double(0.0)
    runs = C.c_int32(0)
    _check(lib.ddx_louvain_best_of(n, _p(indptr, c_i64_p), _p(indices, c_i32_p), _p(weights, c_f64_p), float(gamma),
                                   int(seed) & 0xFFFFFFFFFFFFFFFF, float(q_tol), int(stall), int(max_runs), int(threads),
                                   1 if presweeps else 0, _p(labels, c_i32_p), C.byref(q), C.byref(runs)))
    return labels, q.value, runs.value


def _leiden_call(name, indptr, indices, weights, gamma, seed):
    lib = load()
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    n = indptr.shape[0] - 1
    labels = np.empty(n, dtype=np.int32)
    _check(getattr(lib, name)(n, _p(indptr, c_i64_p), _p(indices, c_i32_p), _p(weights, c_f64_p), float(gamma),
                              int(seed) & 0xFFFFFFFFFFFFFFFF, _p(labels, c_i32_p)))
    return labels


def leiden(indptr, indices, weights, gamma: float, seed: int):
    """Pre-sweeps (part A) followed by sequential Leiden (part B') on the host.  Returns labels int32[n]."""
    return _leiden_call("ddx_leiden", indptr, indices, weights, gamma, seed)


def leiden_sequential(indptr, indices, weights, gamma: float, seed: int):
    """Part B' of the specification only (Leiden on the given graph).  Returns labels int32[n]."""
    return _leiden_call("ddx_leiden_sequential", indptr, indices, weights, gamma, seed)


def refine(indptr, indices, weights, labels, gamma: float, sweeps: int = REFINE_SWEEPS, subrounds: int = SUBROUNDS, canonical: bool = True):
    """One level of part C on the host: refinement sweeps on a graph from `labels` (ids < n are kept as community ids).
    Returns labels int32[n], numbered by ascending smallest member unless ``canonical=False`` (raw ids, to chain levels)."""
    lib = load()
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    n = indptr.shape[0] - 1
    out = np.empty(n, dtype=np.int32)
    _check(lib.ddx_refine(n, _p(indptr, c_i64_p), _p(indices, c_i32_p), _p(weights, c_f64_p), _p(labels, c_i32_p), float(gamma),
                          int(sweeps), int(subrounds), 1 if canonical else 0, _p(out, c_i32_p)))
    return out


def refine_down(graphs, members, labels, gamma: float, sweeps: int = REFINE_SWEEPS):
    """Part C over all levels on the host, as ddx_louvain runs it: `labels` label the nodes of graphs[-1]; graphs[l + 1] is
    the aggregate of graphs[l] with member table members[l].  Returns the canonical labels of the nodes of graphs[0]."""
    lab = np.asarray(labels, dtype=np.int32)
    for level in range(len(members) - 1, -1, -1):
        lab = refine(*graphs[level], lab[members[level]], gamma, sweeps, canonical=(level == 0))
    return lab


def presweep_levels(indptr, indices, weights, gamma: float, levels: int = None):
    """Part A applied `levels` times on the host: (graphs, members) as refine_down takes them."""
    graphs, members = [(indptr, indices, weights)], []
    for _ in range(PRESWEEP_LEVELS if levels is None else levels):
        m, *g = presweep(*graphs[-1], gamma)
        members.append(m)
        graphs.append(tuple(g))
    return graphs, members


def presweep(indptr, indices, weights, gamma: float, sweeps: int = PRESWEEPS, subrounds: int = SUBROUNDS):
    """Part A on the host.  Returns (member int32[n], c_indptr, c_indices, c_weights)."""
    lib = load()
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    n = indptr.shape[0] - 1
    nnz = int(indptr[-1]) if n > 0 else 0
    member = np.empty(n, dtype=np.int32)
    c_indptr = np.zeros(n + 1, dtype=np.int64)
    c_indices = np.empty(max(nnz, 1), dtype=np.int32)
    c_weights = np.empty(max(nnz, 1), dtype=np.float64)
    nc = C.c_int64(0)
    _check(lib.ddx_presweep(n, _p(indptr, c_i64_p), _p(indices, c_i32_p), _p(weights, c_f64_p), float(gamma), int(sweeps),
                            int(subrounds), _p(member, c_i32_p), C.byref(nc), _p(c_indptr, c_i64_p), _p(c_indices, c_i32_p), _p(c_weights, c_f64_p)))
    k = nc.value
    e = int(c_indptr[k])
    return member, c_indptr[:k + 1].copy(), c_indices[:e].copy(), c_weights[:e].copy()


def relabel_by_size(labels, min_cluster_size=None):
    lib = load()
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    out = np.empty(labels.shape[0], dtype=np.int64)
    mcs = -1 if min_cluster_size is None else int(min_cluster_size)
    _check(lib.ddx_relabel_by_size(labels.shape[0], _p(labels, c_i32_p), mcs, _p(out, c_i64_p)))
    return out


def assemble_graph(idx, w):
    """Symmetric CSR (indptr, indices, weights) from a relation table (host C++, thread-safe)."""
    lib = load()
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float64)
    n, k = idx.shape
    ip = np.empty(n + 1, dtype=np.int64)
    ix = np.empty(2 * n * k, dtype=np.int32)
    wt = np.empty(2 * n * k, dtype=np.float64)
    _check(lib.ddx_assemble_graph(n, k, _p(idx, c_i32_p), _p(w, c_f64_p), _p(ip, c_i64_p), _p(ix, c_i32_p), _p(wt, c_f64_p)))
    e = int(ip[-1])
    return ip, ix[:e], wt[:e]


def hypergeom_logsf(k, M, n, N) -> float:
    out = C.c_double(0.0)
    _check(load().ddx_hypergeom_logsf(int(k), int(M), int(n), int(N), C.byref(out)))
    return out.value


def score_communities(full, num_cells: int):
    lib = load()
    full = np.ascontiguousarray(full, dtype=np.int64)
    scores = np.empty(num_cells, dtype=np.float64)
    logp = np.empty(num_cells, dtype=np.float64)
    _check(lib.ddx_score_communities(_p(full, c_i64_p), full.shape[0], int(num_cells), _p(scores, c_f64_p),
                                     _p(logp, c_f64_p)))
    return scores, logp


# ---- device context ----------------------------------------------------------------------------
# Tuning / diagnostic switches applied to every context this process opens (ddx_set_option; include/ddx.h lists the keys).
# ``OPTIONS`` is the programmatic handle; the environment variable DDX_OPTIONS="key=value,key=value" is read here, in the
# Python host layer, for tools and tests -- the library itself never looks at the environment.
OPTIONS: dict = {}


def current_options() -> dict:
    opts = {}
    for item in os.environ.get("DDX_OPTIONS", "").split(","):
        if item.strip():
            k, _, v = item.partition("=")
            opts[k.strip()] = v.strip()
    opts.update({str(k): str(v) for k, v in OPTIONS.items()})
    return opts


import contextlib
import threading

_blas_lock = threading.Lock()
_blas_depth = 0
_blas_limit = None


@contextlib.contextmanager
def single_threaded_blas():
    """The host's BLAS / LAPACK on ONE thread while any caller is inside (small dense problems: a 256-thread pool costs more
    than it returns, and its workers keep spinning afterwards -- which slows the kernel launches of every lane of the
    process).  The limit is process-wide, so it is reference-counted: the first entry sets it, the last exit lifts it,
    whatever order the lanes (threads) come and go in."""
    global _blas_depth, _blas_limit
    with _blas_lock:
        if _blas_depth == 0:
            try:
                from threadpoolctl import threadpool_limits

                _blas_limit = threadpool_limits(limits=1)
            except Exception:          # threadpoolctl absent: correct, just slower
                _blas_limit = None
        _blas_depth += 1
    try:
        yield
    finally:
        with _blas_lock:
            _blas_depth -= 1
            if _blas_depth == 0 and _blas_limit is not None:
                _blas_limit.restore_original_limits()
                _blas_limit = None


_EIGH_FN = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


class Context:
    """One GPU, one stream, all device buffers of a fit (ddx_ctx)."""

    def __init__(self, device: int = 0):
        self._lib = load()
        self._h = C.c_void_p(None)
        _check(self._lib.ddx_create(int(device), C.byref(self._h)))
        self.device = device
        self.N = self.H = self.S = 0
        self._guard = False
        self.apply_options()

    def set_option(self, key: str, value) -> None:
        self._c(self._lib.ddx_set_option(self._h, str(key).encode(), str(value).encode()))
        if key == "arena_guard":
            self._guard = str(value) not in ("", "0")
        elif key == "defaults":
            self._guard = False

    def apply_options(self, options: dict | None = None) -> None:
        """Every switch back to its default, then the process-wide ones (``OPTIONS`` / DDX_OPTIONS) and ``options``."""
        self.set_option("defaults", "")
        for k, v in {**current_options(), **(options or {})}.items():
            self.set_option(k, v)

    def close(self):
        if self._h:
            try:
                if self._guard:
                    rc = self._lib.ddx_check_memory(self._h)
                    if rc == -4:                        # DDX_E_NUMERIC: a buffer was overrun
                        _check(rc, self._h)
            finally:
                self._lib.ddx_destroy(self._h)
                self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _c(self, rc):
        _check(rc, self._h)

    @property
    def M(self):
        return self.N + self.S

    def synchronize(self):
        self._c(self._lib.ddx_synchronize(self._h))

    def check_memory(self):
        """Raises if a kernel wrote past the end of a device buffer (needs DDX_ARENA_GUARD=1 when the context is made)."""
        self._c(self._lib.ddx_check_memory(self._h))

    def device_bytes(self) -> int:
        v = C.c_int64(0)
        self._c(self._lib.ddx_device_bytes(self._h, C.byref(v)))
        return v.value

    def device_memory(self):
        """(free, total) bytes of the context's GPU as the driver reports them."""
        f, t = C.c_int64(0), C.c_int64(0)
        self._c(self._lib.ddx_device_memory(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def arena_peak(self) -> int:
        """The most this context's buffers have occupied at any one time (bytes)."""
        out = np.zeros(1, dtype=np.int64)
        self._c(self._lib.ddx_arena_peak(self._h, _p(out, c_i64_p)))
        return int(out[0])

    def follower_bytes(self) -> int:
        """Device memory a context that clones this one's counts starts with (stage_clone_counts, csrc/k_sparse.hip: its first chunk is
        sized from the RESTRICTED counts -- no raw matrix, no HVG temporaries), so that the number of contexts a GPU is given
        need not assume every follower is as large as its leader.  0 when this context holds no counts."""
        nnz = C.c_int64(0)
        if self._lib.ddx_get_counts_nnz(self._h, C.byref(nnz)) != 0:
            return 0
        return int(nnz.value) * 90 + int(getattr(self, "N", 0) or 0) * 6000 + (1 << 30)

    def reserve_hint(self, nbytes: int):
        """Size of the next memory chunk the context requests from the driver (0: the library's own guess)."""
        self._c(self._lib.ddx_reserve_hint(self._h, int(nbytes)))

    def trim(self, keep_bytes: int = 0):
        """Forget the last fit and return device memory to the driver until at most keep_bytes remain."""
        self._c(self._lib.ddx_trim(self._h, int(keep_bytes)))

    # prologue
    def upload_raw(self, csr):
        ip = np.ascontiguousarray(csr.indptr, dtype=np.int64)
        ix = np.ascontiguousarray(csr.indices, dtype=np.int32)
        d = np.ascontiguousarray(csr.data, dtype=np.float32)
        self._c(self._lib.ddx_upload_raw(self._h, csr.shape[0], csr.shape[1], _p(ip, c_i64_p), _p(ix, c_i32_p), _p(d, c_f32_p)))
        self._rawG = csr.shape[1]
        self._rawN = csr.shape[0]

    def gene_variances(self):
        out = np.empty(self._rawG, dtype=np.float32)
        self._c(self._lib.ddx_gene_variances(self._h, _p(out, c_f32_p)))
        return out

    def select_columns(self, cols):
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        self._c(self._lib.ddx_select_columns(self._h, _p(cols, c_i64_p), cols.shape[0]))
        self.N, self.H, self.S = self._rawN, int(cols.shape[0]), 0

    def upload_counts(self, csr):
        ip = np.ascontiguousarray(csr.indptr, dtype=np.int64)
        ix = np.ascontiguousarray(csr.indices, dtype=np.int32)
        d = np.ascontiguousarray(csr.data, dtype=np.float32)
        self._c(self._lib.ddx_upload_counts(self._h, csr.shape[0], csr.shape[1], _p(ip, c_i64_p), _p(ix, c_i32_p), _p(d, c_f32_p)))
        self.N, self.H, self.S = int(csr.shape[0]), int(csr.shape[1]), 0

    def clone_counts_from(self, src: "Context"):
        """Device-to-device copy of the resident counts of another context on the same GPU (ddx_clone_counts)."""
        self._c(self._lib.ddx_clone_counts(self._h, src._h))
        self.N, self.H, self.S = src.N, src.H, 0

    def get_counts(self):
        import scipy.sparse as sp

        nnz = C.c_int64(0)
        self._c(self._lib.ddx_get_counts_nnz(self._h, C.byref(nnz)))
        ip = np.empty(self.N + 1, dtype=np.int64)
        ix = np.empty(nnz.value, dtype=np.int32)
        d = np.empty(nnz.value, dtype=np.float32)
        self._c(self._lib.ddx_get_counts(self._h, _p(ip, c_i64_p), _p(ix, c_i32_p), _p(d, c_f32_p)))
        return sp.csr_matrix((d, ix, ip), shape=(self.N, self.H))

    def lib_size(self):
        out = np.empty(self.N, dtype=np.float32)
        self._c(self._lib.ddx_get_lib_size(self._h, _p(out, c_f32_p)))
        return out

    def normed(self):
        nnz = C.c_int64(0)
        self._c(self._lib.ddx_get_counts_nnz(self._h, C.byref(nnz)))
        out = np.empty(nnz.value, dtype=np.float32)
        self._c(self._lib.ddx_get_normed(self._h, _p(out, c_f32_p)))
        return out

    # doublets
    def create_doublets(self, parents):
        parents = np.ascontiguousarray(parents, dtype=np.int64)
        assert parents.ndim == 2 and parents.shape[1] == 2
        self._c(self._lib.ddx_create_doublets(self._h, parents.shape[0], _p(parents, c_i64_p)))
        self.S = int(parents.shape[0])

    def get_synth(self):
        import scipy.sparse as sp

        nnz = C.c_int64(0)
        self._c(self._lib.ddx_get_synth_nnz(self._h, C.byref(nnz)))
        ip = np.empty(self.S + 1, dtype=np.int64)
        ix = np.empty(nnz.value, dtype=np.int32)
        d = np.empty(nnz.value, dtype=np.float32)
        self._c(self._lib.ddx_get_synth(self._h, _p(ip, c_i64_p), _p(ix, c_i32_p), _p(d, c_f32_p)))
        return sp.csr_matrix((d, ix, ip), shape=(self.S, self.H))

    # normalisation
    def lognormalise(self, pseudocount: float):
        self._c(self._lib.ddx_lognormalise(self._h, float(pseudocount)))

    def aug_lib(self):
        out = np.empty(self.M, dtype=np.float32)
        med = C.c_float(0.0)
        self._c(self._lib.ddx_get_aug_lib(self._h, _p(out, c_f32_p), C.byref(med)))
        return out, np.float32(med.value)

    def aug_nnz(self) -> int:
        nnz = C.c_int64(0)
        self._c(self._lib.ddx_get_aug_nnz(self._h, C.byref(nnz)))
        return nnz.value

    def aug_values(self):
        nnz = C.c_int64(0)
        self._c(self._lib.ddx_get_aug_nnz(self._h, C.byref(nnz)))
        v = np.empty(nnz.value, dtype=np.float32)
        z = np.empty(self.H, dtype=np.float32)
        self._c(self._lib.ddx_get_aug_values(self._h, _p(v, c_f32_p), _p(z, c_f32_p)))
        return v, z

    def aug_dense_rows(self, row0: int, nrows: int):
        out = np.empty((nrows, self.H), dtype=np.float32)
        self._c(self._lib.ddx_get_aug_dense_rows(self._h, int(row0), int(nrows), _p(out, c_f32_p)))
        return out

    def scale(self, max_value: float):
        self._c(self._lib.ddx_scale(self._h, float(max_value if max_value is not None else 0.0)))

    # PCA
    def pca(self, n_components: int, q0, n_oversamples: int = 10, n_iter: int = -1, q0_rows: int = 0):
        """q0 = None reuses the start matrix the previous call left on the device (pass q0_rows then)."""
        if q0 is None:
            self._c(self._lib.ddx_pca(self._h, int(n_components), int(n_oversamples), int(n_iter), None, int(q0_rows)))
        else:
            q0 = np.ascontiguousarray(q0, dtype=np.float64)
            self._c(self._lib.ddx_pca(self._h, int(n_components), int(n_oversamples), int(n_iter), _p(q0, c_f64_p), q0.shape[0]))
        self._C = int(n_components)
        self._embM = self.M

    def pca_exact_sparse(self, n_components: int, start, tol: float = 1e-7, max_steps: int = 12, n_oversamples: int = 10) -> int:
        """Block Lanczos PCA of the sparse operator (ddx_pca_exact_sparse); returns the number of steps taken.  The small
        projected eigenproblem is solved by LAPACK (scipy.linalg.eigh, wanted pairs only), handed to the library as a callback."""
        start = np.ascontiguousarray(start, dtype=np.float64)
        steps = C.c_int32(0)
        failure = []

        def eigh(n, n_largest, a, w, _user):
            try:
                mat = np.ctypeslib.as_array(a, shape=(n, n))
                vals_out = np.ctypeslib.as_array(w, shape=(n,))
                # the wanted pairs only: dense tridiagonalisation + MRRR (dsyevr).  The matrix is block tridiagonal, but LAPACK's
                # banded driver accumulates an n x n orthogonal factor rotation by rotation and is three times slower at n = 800
                if n > n_largest:
                    from scipy.linalg import eigh as _eigh

                    vals, vecs = _eigh(mat, subset_by_index=[n - n_largest, n - 1], driver="evr", check_finite=False)
                else:
                    vals, vecs = np.linalg.eigh(mat)
                mat[:, n - n_largest:] = vecs
                vals_out[n - n_largest:] = vals
                return 0
            except Exception as err:          # (never let an exception cross the C boundary)
                failure.append(err)
                return 1

        cb = _EIGH_FN(eigh)
        with single_threaded_blas():
            rc = self._lib.ddx_pca_exact_sparse(self._h, int(n_components), int(n_oversamples), float(tol), int(max_steps),
                                                _p(start, c_f64_p), C.byref(steps), C.cast(cb, C.c_void_p), None)
        if failure:
            raise failure[0]
        self.lanczos_converged = rc != W_UNCONVERGED          # (a positive code: the best Ritz pairs found are in place; the caller decides)
        if rc != W_UNCONVERGED:
            self._c(rc)
        self._embM, self._C = self.M, int(n_components)
        return int(steps.value)

    def operator_apply(self, X, mode: int):
        """Products with the centred matrix A held by the context (include/ddx.h: ddx_operator_apply):
        mode 0: A @ X, 1: A.T @ X, 2: A.T @ (A @ X), 3: A @ (A.T @ X)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        n = X.shape[1]
        rows_out = self.M if mode in (0, 3) else self.H
        out = np.empty((rows_out, n), dtype=np.float64)
        self._c(self._lib.ddx_operator_apply(self._h, int(mode), _p(X, c_f64_p), n, _p(out, c_f64_p)))
        return out

    def embedding(self):
        out = np.empty((self._embM, self._C), dtype=np.float32)
        self._c(self._lib.ddx_get_embedding(self._h, _p(out, c_f32_p)))
        return out

    def embedding_f64(self):
        out = np.empty((self._embM, self._C), dtype=np.float64)
        s = np.empty(self._C, dtype=np.float64)
        self._c(self._lib.ddx_get_embedding_f64(self._h, _p(out, c_f64_p), _p(s, c_f64_p)))
        return out, s

    def set_embedding(self, emb):
        emb = np.ascontiguousarray(emb, dtype=np.float32)
        self._c(self._lib.ddx_set_embedding(self._h, _p(emb, c_f32_p), emb.shape[0], emb.shape[1]))
        self._embM, self._C = int(emb.shape[0]), int(emb.shape[1])

    # kNN / graph
    METRICS = {"euclidean": 0, "minkowski": 0, "manhattan": 1, "cosine": 2, "correlation": 3}

    def knn(self, k: int, include_self: bool, metric: str = "euclidean"):
        if metric in ("euclidean", "minkowski"):
            self._c(self._lib.ddx_knn(self._h, int(k), 1 if include_self else 0))
        else:
            self._c(self._lib.ddx_knn_metric(self._h, int(k), 1 if include_self else 0, self.METRICS[metric]))
        self._K = int(k)

    def get_knn(self, with_dist: bool = True):
        idx = np.empty((self._embM, self._K), dtype=np.int32)
        dist = np.empty((self._embM, self._K), dtype=np.float64) if with_dist else None
        self._c(self._lib.ddx_get_knn(self._h, _p(idx, c_i32_p), _p(dist, c_f64_p) if with_dist else None))
        return idx, (np.sqrt(dist) if with_dist else None)   # the C-ABI returns squared distances

    def knn_window_fraction(self) -> float:
        f = C.c_double(0.0)
        self._c(self._lib.ddx_get_knn_window_fraction(self._h, C.byref(f)))
        return f.value

    def knn_overflow_count(self) -> int:
        n = C.c_int64(0)
        self._c(self._lib.ddx_get_knn_overflow_count(self._h, C.byref(n)))
        return int(n.value)

    def upload_form(self) -> int:
        """How the last upload_raw crossed the link: 0 plain arrays, 1 packed by this context, 2 another context's packed image."""
        v = C.c_int32(0)
        self._c(self._lib.ddx_get_upload_form(self._h, C.byref(v)))
        return int(v.value)

    def bitplane_stats(self) -> dict:
        """Did the last PCA's operator products take the bit-plane route, and what is left to the sparse products."""
        out = np.zeros(8, dtype=np.int64)
        self._c(self._lib.ddx_get_bitplane_stats(self._h, _p(out, c_i64_p)))
        return {"active": bool(out[0]), "rest_original": int(out[1]), "rest_synthetic": int(out[2]), "digits": int(out[3]),
                "scaled": bool(out[4]), "demoted_columns": int(out[5]), "format": "mx6" if out[6] else "int8", "digits_early": int(out[7])}

    def knn_candidate_counts(self) -> np.ndarray:
        out = np.empty(self._embM, dtype=np.int32)
        self._c(self._lib.ddx_get_knn_candidate_counts(self._h, _p(out, c_i32_p)))
        return out

    def build_graph(self, mode: int, fetch: bool = True):
        self._c(self._lib.ddx_build_graph(self._h, int(mode)))
        if not fetch:
            return None
        n = C.c_int64(0)
        e = C.c_int64(0)
        self._c(self._lib.ddx_get_graph_size(self._h, C.byref(n), C.byref(e)))
        ip = np.empty(n.value + 1, dtype=np.int64)
        ix = np.empty(e.value, dtype=np.int32)
        w = np.empty(e.value, dtype=np.float64)
        self._c(self._lib.ddx_get_graph(self._h, _p(ip, c_i64_p), _p(ix, c_i32_p), _p(w, c_f64_p)))
        return ip, ix, w

    def fetch_graph(self):
        """The symmetric CSR left on the device by build_graph."""
        n = C.c_int64(0)
        e = C.c_int64(0)
        self._c(self._lib.ddx_get_graph_size(self._h, C.byref(n), C.byref(e)))
        ip = np.empty(n.value + 1, dtype=np.int64)
        ix = np.empty(e.value, dtype=np.int32)
        w = np.empty(e.value, dtype=np.float64)
        self._c(self._lib.ddx_get_graph(self._h, _p(ip, c_i64_p), _p(ix, c_i32_p), _p(w, c_f64_p)))
        return ip, ix, w

    def coarsen_graph(self, gamma: float, sweeps: int = PRESWEEPS, levels: int = PRESWEEP_LEVELS):
        """Part A of the community detection on the device graph.  Returns (member, indptr, indices, weights)."""
        self._c(self._lib.ddx_coarsen_graph(self._h, float(gamma), int(sweeps), int(levels)))
        n = C.c_int64(0)
        e = C.c_int64(0)
        self._c(self._lib.ddx_get_coarse_size(self._h, C.byref(n), C.byref(e)))
        member = np.empty(self._embM, dtype=np.int32)
        ip = np.empty(n.value + 1, dtype=np.int64)
        ix = np.empty(e.value, dtype=np.int32)
        w = np.empty(e.value, dtype=np.float64)
        self._c(self._lib.ddx_get_coarse_graph(self._h, _p(member, c_i32_p), _p(ip, c_i64_p), _p(ix, c_i32_p), _p(w, c_f64_p)))
        return member, ip, ix, w

    def refine_communities(self, coarse_labels, gamma: float, sweeps: int = REFINE_SWEEPS):
        """Part C on the device: labels of the coarse nodes (part B's result) -> final labels of the original nodes."""
        coarse_labels = np.ascontiguousarray(coarse_labels, dtype=np.int32)
        out = np.empty(self._embM, dtype=np.int32)
        self._c(self._lib.ddx_refine_communities(self._h, _p(coarse_labels, c_i32_p), float(gamma), int(sweeps), _p(out, c_i32_p)))
        return out

    def graph_relations(self, mode: int):
        idx = np.empty((self._embM, self._K), dtype=np.int32)
        w = np.empty((self._embM, self._K), dtype=np.float64)
        self._c(self._lib.ddx_graph_relations(self._h, int(mode), _p(idx, c_i32_p), _p(w, c_f64_p)))
        return idx, w

    # timing
    def timing_enable(self, on: bool = True):
        self._c(self._lib.ddx_timing_enable(self._h, 1 if on else 0))

    def timing_reset(self):
        self._c(self._lib.ddx_timing_reset(self._h))

    def timing_reference(self, share_with: "Context" = None):
        """Clock origin of the scope intervals: now on this context's stream, or the origin of `share_with`."""
        self._c(self._lib.ddx_timing_reference(self._h, share_with._h if share_with is not None else None))

    def timing_intervals(self):
        """[(begin_ms, end_ms)] of every timed scope since the last reset, relative to the clock origin."""
        n = C.c_int64(0)
        self._c(self._lib.ddx_timing_intervals(self._h, 0, None, C.byref(n)))
        out = np.empty((n.value, 2), dtype=np.float64)
        if n.value:
            self._c(self._lib.ddx_timing_intervals(self._h, n.value, _p(out, c_f64_p), C.byref(n)))
        return out

    def timings(self):
        n = C.c_int32(0)
        self._c(self._lib.ddx_timing_count(self._h, C.byref(n)))
        out = {}
        buf = C.create_string_buffer(64)
        for i in range(n.value):
            launches = C.c_int64(0)
            ms = C.c_double(0.0)
            self._c(self._lib.ddx_timing_get(self._h, i, buf, C.byref(launches), C.byref(ms)))
            out[buf.value.decode()] = (launches.value, ms.value)
        return out
