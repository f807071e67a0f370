"""Host-side native code of libddx (no GPU needed): symbol export, deterministic Louvain against the
pure-Python specification, hypergeometric scoring against scipy / the reference's golden vectors."""
import ctypes
import re

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT, load_golden
from doubletdetection_amd import _lib
from oracle import dd_oracle as orc
from oracle import louvain_ref


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(f"{ROOT}/include/ddx.h").read()
    declared = set(re.findall(r"\b(ddx_[a-z0-9_]+)\s*\(", header))
    declared.discard("ddx_ctx")
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ddx_abi_version() == _lib.ABI_VERSION


def test_context_creation_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DdxError):
        _lib.Context(0)


def _random_graph(n, k, seed, weighted):
    rng = np.random.default_rng(seed)
    blocks = rng.integers(0, max(2, n // 40), size=n)
    rows, cols = [], []
    for i in range(n):
        same = np.flatnonzero(blocks == blocks[i])
        cand = np.r_[rng.choice(same, size=min(k, len(same)), replace=False), rng.integers(0, n, size=2)]
        for j in cand:
            if j != i:
                rows.append(i); cols.append(int(j))
    A = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n)).tocsr()
    A = ((A + A.T) > 0).astype(np.float64)
    A = sp.csr_matrix(A)
    if weighted:
        W = sp.triu(A, 1).tocoo()
        w = rng.random(W.nnz) + 0.05
        A = sp.coo_matrix((w, (W.row, W.col)), shape=(n, n)).tocsr()
        A = A + A.T
    A.sort_indices()
    return A


@pytest.mark.parametrize("n,k,seed,weighted,gamma", [
    (60, 4, 1, False, 1.0), (300, 6, 2, True, 1.0), (300, 6, 3, False, 4.0), (800, 8, 4, True, 1.0),
    (800, 5, 123, False, 4.0), (50, 3, 5, True, 0.5)])
def test_louvain_matches_python_specification_bit_for_bit(n, k, seed, weighted, gamma):
    A = _random_graph(n, k, seed, weighted)
    ref = louvain_ref.louvain(A.indptr, A.indices, A.data, gamma, seed)
    got, q = _lib.louvain(A.indptr, A.indices, A.data, gamma, seed)
    np.testing.assert_array_equal(got.astype(np.int64), ref)
    # q is what part B reports on the aggregated graph (weights are multiples of 2^-20); part C only adds to it
    assert louvain_ref.modularity(A.indptr, A.indices, A.data, ref, gamma) > q - 1e-5
    # sanity: it finds structure, and the labels are numbered by ascending smallest member
    assert len(np.unique(got)) < n
    np.testing.assert_array_equal(got, louvain_ref.canonical_labels(got))


@pytest.mark.parametrize("n,k,seed,weighted,gamma", [
    (60, 4, 1, False, 1.0), (300, 6, 2, True, 1.0), (300, 6, 3, False, 4.0), (800, 8, 4, True, 4.0),
    (800, 5, 123, False, 4.0), (50, 3, 5, True, 0.5), (1500, 10, 8, True, 2.0)])
def test_leiden_matches_python_specification_bit_for_bit(n, k, seed, weighted, gamma):
    """Part B' (Leiden: local moving with the empty-community option, refinement, aggregation on the refined groups,
    iterated until stable) on its own and behind the pre-sweeps; and the property Leiden exists for: every community
    is connected (on the graph part B' ran on)."""
    from scipy.sparse.csgraph import connected_components

    A = _random_graph(n, k, seed, weighted)
    ref = louvain_ref._leiden_sequential(A.indptr, A.indices, A.data, gamma, seed)
    got = _lib.leiden_sequential(A.indptr, A.indices, A.data, gamma, seed)
    np.testing.assert_array_equal(got.astype(np.int64), ref)
    assert 1 < len(np.unique(got)) < n
    for c in np.unique(got):
        idx = np.flatnonzero(got == c)
        assert connected_components(A[idx][:, idx], directed=False)[0] == 1
    # not worse than the Louvain levels on the same graph (refinement only adds ways out of a local optimum)
    q_leiden = louvain_ref.modularity(A.indptr, A.indices, A.data, got, gamma)
    q_louvain = louvain_ref.modularity(A.indptr, A.indices, A.data,
                                       _lib.louvain_sequential(A.indptr, A.indices, A.data, gamma, seed)[0], gamma)
    assert q_leiden > q_louvain - 0.01
    # whole = pre-sweeps, then part B' on the aggregated graph, then the refinement sweeps on the original one
    whole = _lib.leiden(A.indptr, A.indices, A.data, gamma, seed)
    np.testing.assert_array_equal(whole.astype(np.int64), louvain_ref.leiden(A.indptr, A.indices, A.data, gamma, seed))
    graphs, members = _lib.presweep_levels(A.indptr, A.indices, A.data, gamma)
    np.testing.assert_array_equal(whole, _lib.refine_down(graphs, members, _lib.leiden_sequential(*graphs[-1], gamma, seed), gamma))


def test_leiden_degenerate_graphs():
    """No edges, a single node, isolated nodes beside a clique."""
    lab = _lib.leiden_sequential(np.zeros(6, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0), 1.0, 0)
    np.testing.assert_array_equal(lab, np.arange(5))
    lab = _lib.leiden(np.zeros(2, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0), 1.0, 0)
    np.testing.assert_array_equal(lab, [0])
    A = sp.lil_matrix((7, 7))
    for i in range(4):
        for j in range(4):
            if i != j:
                A[i + 2, j + 2] = 1.0
    A = A.tocsr()
    for fn, rf in ((_lib.leiden_sequential, louvain_ref._leiden_sequential), (_lib.leiden, louvain_ref.leiden)):
        lab = fn(A.indptr, A.indices, A.data, 1.0, 3)
        np.testing.assert_array_equal(lab.astype(np.int64), rf(A.indptr, A.indices, A.data, 1.0, 3))
        assert len(set(lab[2:6])) == 1 and len(set(lab)) == 4


@pytest.mark.parametrize("n,k,seed,weighted,gamma", [(300, 6, 2, True, 1.0), (800, 5, 123, False, 4.0), (2000, 10, 9, True, 1.0)])
def test_presweep_and_sequential_parts_match_specification(n, k, seed, weighted, gamma):
    """Part A (synchronous sub-round sweeps + exact aggregation), part B (sequential levels) and part C (refinement sweeps)
    separately, and C o B o A = whole."""
    A = _random_graph(n, k, seed, weighted)
    m_ref, ip_ref, ix_ref, w_ref = louvain_ref.presweep(A.indptr, A.indices, A.data, gamma)
    m, ip, ix, w = _lib.presweep(A.indptr, A.indices, A.data, gamma)
    np.testing.assert_array_equal(m, m_ref)
    np.testing.assert_array_equal(ip, ip_ref)
    np.testing.assert_array_equal(ix, ix_ref)
    np.testing.assert_array_equal(w, w_ref)
    assert len(ip) - 1 < n                                   # it does coarsen
    for sweeps, subrounds in ((0, 4), (1, 4), (3, 4), (3, 1), (2, 3), (6, 2)):
        ms, ips, ixs, ws = _lib.presweep(A.indptr, A.indices, A.data, gamma, sweeps, subrounds)
        mr, ipr, ixr, wr = louvain_ref.presweep(A.indptr, A.indices, A.data, gamma, sweeps, subrounds)
        np.testing.assert_array_equal(ms, mr)
        np.testing.assert_array_equal(ws, wr)
    seq, _ = _lib.louvain_sequential(ip, ix, w, gamma, seed)
    np.testing.assert_array_equal(seq.astype(np.int64), louvain_ref._louvain_sequential(ip_ref, ix_ref, w_ref, gamma, seed))
    # the whole = PRESWEEP_LEVELS applications of part A, then part B, then part C on every level on the way back down
    graphs, members = _lib.presweep_levels(A.indptr, A.indices, A.data, gamma)
    whole, _ = _lib.louvain(A.indptr, A.indices, A.data, gamma, seed)
    lab = _lib.louvain_sequential(*graphs[-1], gamma, seed)[0]
    after_b = lab
    for mm in reversed(members):
        after_b = after_b[mm]
    np.testing.assert_array_equal(whole, _lib.refine_down(graphs, members, lab, gamma))
    # part C against its Python statement, from B's partition and from arbitrary labellings (any non-negative ids)
    rng = np.random.default_rng(seed)
    for labels in (after_b, rng.integers(0, 7, size=n) * 3 + 1, np.arange(n)[::-1].copy()):
        for sweeps, subrounds in ((3, 4), (1, 1), (2, 2)):
            for canonical in (True, False):
                np.testing.assert_array_equal(_lib.refine(A.indptr, A.indices, A.data, labels, gamma, sweeps, subrounds, canonical).astype(np.int64),
                                              louvain_ref.refine(A.indptr, A.indices, A.data, labels, gamma, sweeps, subrounds, canonical))
    np.testing.assert_array_equal(_lib.refine(A.indptr, A.indices, A.data, after_b, gamma, 0), louvain_ref.canonical_labels(after_b))
    # refinement never loses modularity worth mentioning, and the whole is no worse than the purely sequential optimisation
    q_whole = louvain_ref.modularity(A.indptr, A.indices, A.data, whole, gamma)
    q_b = louvain_ref.modularity(A.indptr, A.indices, A.data, after_b, gamma)
    q_seq = louvain_ref.modularity(A.indptr, A.indices, A.data, _lib.louvain_sequential(A.indptr, A.indices, A.data, gamma, seed)[0], gamma)
    assert q_whole >= q_b - 1e-9
    assert q_whole > q_seq - 0.02


@pytest.mark.parametrize("n,k,seed,weighted,gamma,q_tol", [(400, 6, 1, True, 1.0, 1e-3), (900, 5, 77, False, 4.0, 1e-3),
                                                           (1500, 8, 5, True, 1.0, 1e-5), (300, 4, 2, True, 1.0, 0.5)])
def test_best_of_restarts_matches_python_specification(n, k, seed, weighted, gamma, q_tol):
    """PhenoGraph's restart rule (oracle/louvain_ref.py:louvain_best_of): labels, kept modularity and the number of runs
    are those of the Python statement, whatever number of host threads evaluates the batches."""
    A = _random_graph(n, k, seed, weighted)
    lab, q, runs = louvain_ref.louvain_best_of(A.indptr, A.indices, A.data, gamma, seed, q_tol)
    assert runs >= 20                                              # never fewer than `stall` runs
    for threads in (1, 7):
        got, gq, gruns = _lib.louvain_best_of(A.indptr, A.indices, A.data, gamma, seed, q_tol, threads=threads)
        np.testing.assert_array_equal(got, lab)
        assert gq == q and gruns == runs
    # the kept run is at least as good as the single deterministic run; its Q is the modularity part B reported for it
    # (part A quantises the weights to multiples of 2**-20 before part B sees them), which part C can only raise
    single, q_single = _lib.louvain(A.indptr, A.indices, A.data, gamma, seed)
    assert q >= q_single
    mq = louvain_ref.modularity(A.indptr, A.indices, A.data, lab, gamma)
    assert mq > q - 1e-4
    # without part A the rule applies to the graph as given (how the classifier finishes a graph coarsened on the GPU)
    lab2, q2, runs2 = louvain_ref.louvain_best_of(A.indptr, A.indices, A.data, gamma, seed, q_tol, presweeps=0, refine_sweeps=0)
    got2, gq2, gruns2 = _lib.louvain_best_of(A.indptr, A.indices, A.data, gamma, seed, q_tol, threads=3, presweeps=False)
    np.testing.assert_array_equal(got2, lab2)
    assert gq2 == q2 and gruns2 == runs2


def test_louvain_edge_cases():
    # no edges at all
    ip = np.zeros(6, dtype=np.int64)
    got, _ = _lib.louvain(ip, np.zeros(0, np.int32), np.zeros(0), 1.0, 0)
    np.testing.assert_array_equal(got, np.arange(5))
    np.testing.assert_array_equal(louvain_ref.louvain(ip, [], [], 1.0, 0), np.arange(5))
    # isolated nodes next to a triangle
    A = sp.csr_matrix(np.array([[0, 1, 1, 0, 0], [1, 0, 1, 0, 0], [1, 1, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]], float))
    got, _ = _lib.louvain(A.indptr, A.indices, A.data, 1.0, 7)
    np.testing.assert_array_equal(got.astype(np.int64), louvain_ref.louvain(A.indptr, A.indices, A.data, 1.0, 7))
    assert got[0] == got[1] == got[2] and len({got[0], got[3], got[4]}) == 3


def test_relabel_by_size_matches_oracle():
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 12, size=500).astype(np.int32)
    lab[lab == 3] = 4           # a gap in the label set
    lab[:7] = 11
    for mcs in (None, 10, 45):
        np.testing.assert_array_equal(_lib.relabel_by_size(lab, mcs), orc.relabel_by_size(lab, mcs))


def test_f10_hypergeom_logsf_against_scipy():
    g = load_golden("f10_hypergeom")
    got = np.array([_lib.hypergeom_logsf(*row) for row in g["query"]])
    want = g["logsf"]
    inf = np.isinf(want)
    np.testing.assert_array_equal(got[inf], want[inf])
    # float64 special functions: lgamma-based restatement vs cephes betaln; tolerance, not bit equality
    np.testing.assert_allclose(got[~inf], want[~inf], rtol=1e-9, atol=1e-9)


def test_f8_score_communities_against_reference_lines():
    g = load_golden("f8_scoring")
    n = int(g["num_cells"])
    for key in ("plain", "with_minus1", "zero_synth_comm", "synth_only_comm", "single"):
        s, lp = _lib.score_communities(g[key + "_full"], n)
        np.testing.assert_array_equal(np.isnan(s), np.isnan(g[key + "_scores"]))
        m = ~np.isnan(s)
        np.testing.assert_array_equal(s[m], g[key + "_scores"][m])          # integer ratio: bit exact
        np.testing.assert_allclose(lp[m], g[key + "_logp"][m], rtol=1e-9, atol=1e-9)


def _expand_codes16(indptr, codes, pos, col, val):
    """Independent statement of the 2-byte transfer form (include/ddx.h: ddx_pack_rows16): the expansion the device performs."""
    listed = {int(p): (int(c), np.float32(v)) for p, c, v in zip(pos, col, val)}
    nnz = int(indptr[-1])
    idx = np.empty(nnz, dtype=np.int32)
    data = np.empty(nnz, dtype=np.float32)
    for r in range(len(indptr) - 1):
        prev = -1
        for i in range(int(indptr[r]), int(indptr[r + 1])):
            step, count = int(codes[i]) & 255, int(codes[i]) >> 8
            if step == 0:
                assert count == 0 and i in listed
                idx[i], data[i] = listed[i]
            else:
                assert i not in listed
                idx[i], data[i] = prev + step, np.float32(count)
            prev = int(idx[i])
    return idx, data


def test_two_byte_transfer_form_round_trips_any_matrix():
    """dd.py:149-160 (the matrix handed to fit()): whatever the values and the column order, codes + listed entries expand to the
    caller's arrays bit for bit; counts and steps that fit are not listed."""
    import scipy.sparse as sp
    from doubletdetection_amd import _lib

    rng = np.random.default_rng(5)
    # a typical count matrix: nothing is listed except first columns >= 255 / steps > 255 / counts > 255
    A = sp.random(300, 4000, density=0.05, format="csr", random_state=3, data_rvs=lambda n: rng.integers(1, 40, n)).astype(np.float32)
    A.sort_indices()
    codes, pos, col, val, n = _lib.pack_rows16(A.indptr, A.indices, A.data)
    assert n == len(pos) < 0.02 * A.nnz
    idx, data = _expand_codes16(A.indptr, codes, pos, col, val)
    assert np.array_equal(idx, A.indices) and np.array_equal(data.view(np.uint32), A.data.view(np.uint32))
    assert np.all(np.diff(pos) > 0)
    steps = np.diff(np.concatenate([[-1], A.indices]).astype(np.int64))
    first = np.zeros(A.nnz, dtype=bool)
    first[A.indptr[:-1][np.diff(A.indptr) > 0]] = True
    steps[first] = A.indices[first] + 1
    fits = (steps >= 1) & (steps <= 255) & (A.data <= 255)
    assert np.array_equal(np.flatnonzero(~fits), pos)
    assert np.array_equal(codes[fits], (steps[fits] + (A.data[fits].astype(np.int64) << 8)).astype(np.uint16))
    # hostile content: unsorted and repeated columns, negative and huge columns, fractions, negatives, NaN, infinities, -0.0,
    # 255 / 256, empty rows, a single enormous row
    lens = np.array([0, 5, 0, 0, 700, 1, 3, 0], dtype=np.int64)
    indptr = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(indptr[-1])
    indices = rng.integers(-5, 70000, nnz).astype(np.int32)
    indices[10:300] = np.sort(rng.choice(3000, 290, replace=False))          # a well-formed stretch inside the long row
    data = rng.integers(0, 300, nnz).astype(np.float32)
    special = np.array([2.5, -1.0, np.nan, np.inf, -np.inf, -0.0, 255.0, 256.0, 1e9, 0.0], dtype=np.float32)
    data[: 7 * len(special) : 7] = special
    codes, pos, col, val, n = _lib.pack_rows16(indptr, indices, data)
    idx, out = _expand_codes16(indptr, codes, pos, col, val)
    assert np.array_equal(idx, indices) and np.array_equal(out.view(np.uint32), data.view(np.uint32))
    # a list that is too short reports the full number (the upload then falls back to the plain copies)
    codes2, pos2, _, _, n2 = _lib.pack_rows16(indptr, indices, data, capacity=3)
    assert n2 == n and len(pos2) == 3 and np.array_equal(codes2, codes)
    with pytest.raises(_lib.DdxError):
        _lib.pack_rows16(np.array([0, 3, 2]), indices[:3], data[:3])           # row pointer not monotone
