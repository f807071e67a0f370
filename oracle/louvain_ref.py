"""TEST INFRASTRUCTURE -- pure-Python specification of the build's deterministic Louvain.

Why this exists: the reference delegates community detection to third-party native code that is
absent from /root/reference and from this image -- PhenoGraph's bundled Louvain executables
(Blondel, Guillaume, Lambiotte, Lefebvre 2008; reached from dd.py:320-322) and Traag's
``louvain`` / ``leidenalg`` C++ (reached through scanpy from dd.py:337-342).  No version is pinned
(pyproject.toml:27-35,47-48 give lower bounds only), PhenoGraph seeds from time/pid, and the
reference's tests pin no clustering output.  **Parity with upstream clustering is therefore
unpinned.**  What *is* pinned is this specification: the product's host C++ implementation
(``doubletdetection_amd/csrc/louvain.cpp``) must reproduce it bit-for-bit (same visiting order,
same float64 operation order, no FMA contraction), which ``tests/test_louvain.py`` checks.

Algorithm (the published multi-level modularity optimisation with a resolution parameter), in two parts.

**Part A -- synchronous pre-sweeps** (``presweep``, applied ``PRESWEEP_LEVELS`` times, each time to the graph the
previous application aggregated; what the GPU runs, cf. the parallel Louvain variants of
Lu, Halappanavar, Kalyanaraman 2015 and Naim et al. 2017).  Weights are quantised to integers
``wq = rint(w * 2**20)`` so that every sum below is exact and independent of summation order.  A sweep consists of
``SUBROUNDS`` sub-rounds; in sub-round r of sweep s the nodes with ``(v + s) mod SUBROUNDS == r`` decide *at once*
from the same state, everybody else stays (node numbers carry no structure -- cells come in arbitrary order, aggregated
nodes are numbered by community id -- and if they did, neighbours i, i+1 deciding in different sub-rounds is what one wants):
``score(v, c) = W(v,c) * 2m  -  gamma * (tot_c - [c == own] k_v) * k_v`` (float64, this operation order), the
best community among the neighbours' communities (ties: smaller id) is taken if its score is strictly larger than
staying, except that a node that is alone in its community does not move to another lone node with a larger id
(the minimum-label rule that prevents two singletons from swapping for ever).  Community totals are recomputed
between sub-rounds.  ``PRESWEEPS`` sweeps are made (stopping early when a whole sweep moves nothing), then the
communities are aggregated exactly (integer sums, renumbered by ascending id).  Why sub-rounds: with every node
deciding at once (the round-2 text) neighbours merge in arbitrary pairs at the first sweep and the sequential levels
cannot undo those groups -- measured against networkx's Louvain (tests/test_clustering_independent.py) that cost
0.03 of modularity on the resolution-4 neighbour graphs; half of the nodes at a time already behaves like the
sequential sweep (2, 3 and 4 sub-rounds, 4 to 6 sweeps, 2 or 3 refinement sweeps all end within 0.001 of each other and
of networkx on the fixture graphs; the cheapest setting with the widest margin was taken).

**Part C -- refinement on the way back down** (``refine``; the uncoarsening refinement of multi-level Louvain,
Rotta & Noack 2011): the partition part B (or B') produced is projected onto the nodes of the graph the LAST
application of part A started from (a community keeps the id part B gave it, all the way down), and ``REFINE_SWEEPS``
sweeps of the very same sub-round moves are made on that (quantised) graph; the result is projected one level further
down and refined there, and so on until the original graph: groups (then single nodes) that part A put on the wrong side
of a community border change sides.  The result is numbered by ascending smallest member.  With parts A + B + C the build reaches the modularity of networkx's
sequential Louvain on the reference's graphs to within 0.001 and the same number of communities
(tests/golden/clustering_networkx.npz).

**Part B -- sequential multi-level optimisation** of the aggregated graph:

* quality  Q = sum_c [ in_c / 2m  -  gamma * (tot_c / 2m)^2 ]   (RB-configuration form; gamma=1 is
  Newman-Girvan modularity as in PhenoGraph, gamma=4 is what dd.py:417-420 passes to scanpy);
* one level: visit nodes in index order starting at a seeded offset (one splitmix64 stream for all
  levels); take the node out of its community; evaluate gain(c) = w(v,c) - gamma*tot_c*k_v/2m for
  its own community first and then each neighbouring community in adjacency order; move to the
  strictly best; repeat passes while nodes moved and the pass improved Q by more than 1e-6 (the
  Blondel executable's default epsilon).  After the first pass of a level only *active* nodes are
  revisited: a move of v activates, for the next pass, every neighbour of v that is not in v's new
  community (the pruning of Ozaki et al. 2016 / Traag's fast local move);
* aggregate communities into super-nodes (self-loop = total internal weight, both directions),
  repeat until a level moves nothing.

**Part B' -- sequential Leiden** (``leiden``; Traag, Waltman, van Eck 2019, "From Louvain to Leiden", Algorithm 1,
with the randomised merge of its refinement phase taken in the limit theta -> 0, i.e. the best admissible merge,
which makes the result a function of the seed only).  Used instead of part B for ``clustering_algorithm="leiden"``
(dd.py:329,337-342 -> sc.tl.leiden -> leidenalg, absent here):

* local moving as in part B, but started from a given partition (singletons on the first level, the unrefined
  communities afterwards) and with the extra option of leaving for an empty community when every gain is negative
  (which happens for gamma > 1);
* refinement inside every community C: every node starts alone; nodes are visited in index order from a seeded
  offset; a node that is still alone and well connected, E(v, C-v) >= gamma k_v (K_C - k_v) / 2m, joins the
  well-connected (E(R, C-R) >= gamma K_R (K_C - K_R) / 2m) refined group R of C with the strictly largest positive
  gain w(v,R) - gamma K_R k_v / 2m (ties: first in adjacency order);
* aggregate on the *refined* groups, the unrefined communities become the starting partition of the next level;
  stop when every community is a single node or the refinement merged nothing;
* repeat the whole procedure from the resulting partition until an iteration moves no node (leidenalg's
  ``n_iterations=-1``, scanpy's default), at most ``LEIDEN_MAX_ITERATIONS`` times.
"""
from __future__ import annotations

import numpy as np

_MASK = 0xFFFFFFFFFFFFFFFF
MIN_GAIN = 1e-6
PRESWEEPS = 6
PRESWEEP_LEVELS = 2
SUBROUNDS = 2
REFINE_SWEEPS = 3
WEIGHT_SCALE = float(1 << 20)
LEIDEN_MAX_ITERATIONS = 16


def _sync_sweeps(indptr, indices, wq, comm, gamma: float, sweeps: int, subrounds: int):
    """``sweeps`` sweeps of ``subrounds`` synchronous sub-rounds from the partition ``comm`` (community ids: any
    integers in [0, n)) on integer weights ``wq``.  Returns the new ``comm``."""
    n = len(indptr) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    K = np.zeros(n, dtype=np.int64)
    np.add.at(K, rows, wq)
    m2 = int(K.sum())
    comm = np.asarray(comm, dtype=np.int64).copy()
    noself = rows != indices
    r_ns, u_ns, w_ns = rows[noself], indices[noself], wq[noself]
    gamma = float(gamma)
    h = np.arange(n, dtype=np.int64)
    subrounds = max(1, int(subrounds))
    if m2 <= 0 or len(r_ns) == 0:
        return comm
    kvf = K.astype(np.float64)
    for s in range(int(sweeps)):
        moved = False
        for r in range(subrounds):
            tot = np.zeros(n, dtype=np.int64)
            np.add.at(tot, comm, K)
            size = np.bincount(comm, minlength=n)
            key = r_ns * n + comm[u_ns]                       # (node, neighbour community)
            order = np.argsort(key, kind="stable")
            ks, ws = key[order], w_ns[order]
            first = np.concatenate([[True], ks[1:] != ks[:-1]])
            W = np.add.reduceat(ws, np.flatnonzero(first))    # exact integer sums
            kv, kc = ks[first] // n, ks[first] % n
            own = kc == comm[kv]
            own_w = np.zeros(n, dtype=np.int64)
            own_w[kv[own]] = W[own]
            own_score = own_w.astype(np.float64) * float(m2) - (gamma * (tot[comm] - K).astype(np.float64)) * kvf
            cand = ~own
            cv, cc, cw = kv[cand], kc[cand], W[cand]
            score = cw.astype(np.float64) * float(m2) - (gamma * tot[cc].astype(np.float64)) * kvf[cv]
            o2 = np.lexsort((cc, -score, cv))                 # per node: best score first, ties by smaller community
            cv2 = cv[o2]
            f2 = np.concatenate([[True], cv2[1:] != cv2[:-1]]) if len(cv2) else np.zeros(0, dtype=bool)
            bv, bc, bs = cv2[f2], cc[o2][f2], score[o2][f2]
            move = bs > own_score[bv]
            lone = (size[comm[bv]] == 1) & (size[bc] == 1) & (bc > comm[bv])
            move &= ~lone
            move &= ((h[bv] + s) % subrounds) == r            # only this sub-round's class decides
            if move.any():
                moved = True
                new_comm = comm.copy()
                new_comm[bv[move]] = bc[move]
                comm = new_comm
        if not moved:
            break
    return comm


def presweep(indptr, indices, weights, gamma: float = 1.0, sweeps: int = PRESWEEPS, subrounds: int = SUBROUNDS):
    """Part A.  Returns (member, c_indptr, c_indices, c_weights): member[v] = coarse node of v (numbered by
    ascending community id) and the aggregated graph with float64 weights (integer sums / 2**20)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    n = len(indptr) - 1
    wq = np.rint(np.asarray(weights, dtype=np.float64) * WEIGHT_SCALE).astype(np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    comm = _sync_sweeps(indptr, indices, wq, np.arange(n, dtype=np.int64), gamma, sweeps, subrounds)
    used, member = np.unique(comm, return_inverse=True)
    nc = len(used)
    ckey = member[rows] * nc + member[indices]
    order = np.argsort(ckey, kind="stable")
    ks, ws = ckey[order], wq[order]
    if len(ks):
        first = np.concatenate([[True], ks[1:] != ks[:-1]])
        W = np.add.reduceat(ws, np.flatnonzero(first))
        cr, ccol = ks[first] // nc, ks[first] % nc
    else:
        W = np.zeros(0, dtype=np.int64)
        cr = ccol = np.zeros(0, dtype=np.int64)
    c_indptr = np.zeros(nc + 1, dtype=np.int64)
    np.add.at(c_indptr, cr + 1, 1)
    c_indptr = np.cumsum(c_indptr)
    return member.astype(np.int64), c_indptr, ccol.astype(np.int64), W.astype(np.float64) / WEIGHT_SCALE


def canonical_labels(labels) -> np.ndarray:
    """Labels 0..K-1 numbered by ascending smallest member (independent of how the communities were named)."""
    labels = np.asarray(labels)
    n = len(labels)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    _, inv = np.unique(labels, return_inverse=True)
    first = np.full(inv.max() + 1, n, dtype=np.int64)
    np.minimum.at(first, inv, np.arange(n, dtype=np.int64))
    rank = np.empty(len(first), dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(len(first))
    return rank[inv]


def refine(indptr, indices, weights, labels, gamma: float = 1.0, sweeps: int = REFINE_SWEEPS, subrounds: int = SUBROUNDS,
           canonical: bool = True) -> np.ndarray:
    """One level of part C: the refinement sweeps on a graph from the labelling ``labels`` of its nodes.  The labels are
    the community ids (ties between equally good moves go to the smaller id); labels outside [0, n) are first replaced
    by their canonical numbering.  Returns the canonical labels of the result (``canonical=False``: the raw ids)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int64)
    labels = np.asarray(labels, dtype=np.int64)
    n = len(indptr) - 1
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    wq = np.rint(np.asarray(weights, dtype=np.float64) * WEIGHT_SCALE).astype(np.int64)
    if labels.min() < 0 or labels.max() >= n:
        labels = canonical_labels(labels)
    comm = _sync_sweeps(indptr, indices, wq, labels, gamma, sweeps, subrounds)
    return canonical_labels(comm) if canonical else comm


class SplitMix64:
    def __init__(self, seed: int):
        self.state = int(seed) & _MASK

    def next(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & _MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)


def _visit_order(n: int, rng: SplitMix64):
    """Index order starting at a seeded offset."""
    start = rng.next() % n
    return [(start + i) % n for i in range(n)]


def _quality(in_, tot, m2, gamma):
    q = 0.0
    for c in range(len(tot)):
        if tot[c] > 0.0:
            b = tot[c] / m2
            q += in_[c] / m2 - gamma * b * b
    return q


def _one_level(indptr, indices, weights, gamma, rng):
    """Returns (comm list, moved_any)."""
    n = len(indptr) - 1
    deg = [0.0] * n
    loops = [0.0] * n
    for v in range(n):
        s = 0.0
        for e in range(indptr[v], indptr[v + 1]):
            s += weights[e]
            if indices[e] == v:
                loops[v] += weights[e]
        deg[v] = s
    m2 = 0.0
    for v in range(n):
        m2 += deg[v]
    comm = list(range(n))
    if m2 == 0.0:
        return comm, False, 0.0
    tot = deg[:]
    in_ = loops[:]
    order = _visit_order(n, rng)
    neigh_w = [-1.0] * n
    improved = False
    new_q = _quality(in_, tot, m2, gamma)
    active = [True] * n
    while True:
        cur_q = new_q
        moves = 0
        next_active = [False] * n
        for v in order:
            if not active[v]:
                continue
            c_old = comm[v]
            kv = deg[v]
            seen = [c_old]
            neigh_w[c_old] = 0.0
            for e in range(indptr[v], indptr[v + 1]):
                u = indices[e]
                if u == v:
                    continue
                c = comm[u]
                if neigh_w[c] == -1.0:
                    neigh_w[c] = 0.0
                    seen.append(c)
                neigh_w[c] += weights[e]
            # take v out of its community
            tot[c_old] -= kv
            in_[c_old] -= 2.0 * neigh_w[c_old] + loops[v]
            best = c_old
            best_gain = neigh_w[c_old] - gamma * tot[c_old] * kv / m2
            for c in seen[1:]:
                g = neigh_w[c] - gamma * tot[c] * kv / m2
                if g > best_gain:
                    best_gain = g
                    best = c
            tot[best] += kv
            in_[best] += 2.0 * neigh_w[best] + loops[v]
            comm[v] = best
            if best != c_old:
                moves += 1
                for e in range(indptr[v], indptr[v + 1]):
                    u = indices[e]
                    if u != v and comm[u] != best:
                        next_active[u] = True
            for c in seen:
                neigh_w[c] = -1.0
        new_q = _quality(in_, tot, m2, gamma)
        if moves > 0:
            improved = True
        if not (moves > 0 and new_q - cur_q > MIN_GAIN):
            break
        active = next_active
    return comm, improved, new_q


def _aggregate(indptr, indices, weights, comm):
    """Super-node graph; communities renumbered by ascending id. Returns (indptr, indices, weights, renum)."""
    n = len(indptr) - 1
    used = sorted(set(comm))
    renum = {c: i for i, c in enumerate(used)}
    members = [[] for _ in used]
    for v in range(n):
        members[renum[comm[v]]].append(v)
    new_indptr = [0]
    new_indices = []
    new_weights = []
    for cn, mem in enumerate(members):
        acc = {}
        for v in mem:
            for e in range(indptr[v], indptr[v + 1]):
                t = renum[comm[indices[e]]]
                if t in acc:
                    acc[t] += weights[e]
                else:
                    acc[t] = weights[e]
        for t in sorted(acc):
            new_indices.append(t)
            new_weights.append(acc[t])
        new_indptr.append(len(new_indices))
    return new_indptr, new_indices, new_weights, renum


def louvain(indptr, indices, weights, gamma: float = 1.0, seed: int = 0, presweeps: int = PRESWEEPS,
            levels: int = PRESWEEP_LEVELS, refine_sweeps: int = REFINE_SWEEPS) -> np.ndarray:
    """Parts A + B + C.  Community label per node (0..K-1, numbered by ascending smallest member; without pre-sweeps
    there is nothing to refine and the labels are part B's, numbered by ascending representative id)."""
    graphs, members = _presweep_levels(indptr, indices, weights, gamma, presweeps, levels)
    lab = _louvain_sequential(*graphs[-1], gamma, seed)
    return _refine_down(graphs, members, lab, gamma, refine_sweeps)


def _presweep_levels(indptr, indices, weights, gamma, presweeps, levels):
    """Part A applied ``levels`` times: ([graph_0, ..., graph_levels], [member_0, ..., member_{levels-1}])."""
    graphs, members = [(indptr, indices, weights)], []
    for _ in range(levels if presweeps > 0 else 0):
        m, ip, ix, w = presweep(*graphs[-1], gamma, presweeps)
        members.append(m)
        graphs.append((ip, ix, w))
    return graphs, members


def _refine_down(graphs, members, lab, gamma, refine_sweeps):
    """Part C: ``lab`` labels the nodes of graphs[-1]; refine on graphs[-2], ..., graphs[0]."""
    lab = np.asarray(lab)
    for level in range(len(members) - 1, -1, -1):
        lab = refine(*graphs[level], lab[members[level]], gamma, refine_sweeps, canonical=False)     # ids of part B throughout
    return canonical_labels(lab) if members else lab


def _louvain_sequential(indptr, indices, weights, gamma: float = 1.0, seed: int = 0, with_quality: bool = False):
    """Part B.  ``with_quality``: also return Q of the final partition (the quality the last level evaluates)."""
    indptr = [int(x) for x in np.asarray(indptr)]
    indices = [int(x) for x in np.asarray(indices)]
    weights = [float(x) for x in np.asarray(weights, dtype=np.float64)]
    n = len(indptr) - 1
    rng = SplitMix64(seed)
    membership = list(range(n))
    gamma = float(gamma)
    q = 0.0
    while True:
        comm, improved, q = _one_level(indptr, indices, weights, gamma, rng)
        indptr, indices, weights, renum = _aggregate(indptr, indices, weights, comm)
        membership = [renum[comm[c]] for c in membership]
        if not improved:
            break
    membership = np.asarray(membership, dtype=np.int64)
    return (membership, q) if with_quality else membership


def louvain_best_of(indptr, indices, weights, gamma: float = 1.0, seed: int = 0, q_tol: float = 1e-3, stall: int = 20,
                    max_runs: int = 1000, presweeps: int = PRESWEEPS, presweep_levels: int = PRESWEEP_LEVELS,
                    refine_sweeps: int = REFINE_SWEEPS):
    """PhenoGraph's restart rule around part B (upstream ``phenograph.core.runlouvain``, reached from dd.py:320-322 --
    restated, the package is absent): the Louvain executable is run again and again from another random node order; a
    run replaces the best result when its modularity exceeds the best by more than ``q_tol`` (1e-3 upstream); the loop
    ends after ``stall`` (20) consecutive runs without such a gain.  Upstream seeds every run from time / pid; here run r
    uses ``seed + r``, so the outcome is a function of (graph, gamma, seed, q_tol, stall).  Part A (the pre-sweeps) is
    deterministic and runs once, and so does part C, applied to the run that was kept (``refine_sweeps=0``: the kept
    run as it is, numbered as part B numbers it).  Returns (labels, quality part B reported for the kept run, number of runs)."""
    graphs, members = _presweep_levels(indptr, indices, weights, gamma, presweeps, presweep_levels)
    indptr, indices, weights = graphs[-1]
    best, best_q, run, updated = None, 0.0, 0, 0
    while run - updated < stall and run < max_runs:
        lab, q = _louvain_sequential(indptr, indices, weights, gamma, (seed + run) & 0xFFFFFFFFFFFFFFFF, with_quality=True)
        if best is None or q - best_q > q_tol:
            best, best_q, updated = lab, q, run
        run += 1
    if refine_sweeps > 0:
        best = _refine_down(graphs, members, best, gamma, refine_sweeps)
    else:
        for m in reversed(members):
            best = best[m]
    return best, best_q, run


# ----------------------------------------------------------------------------------------------------------------------
# Part B': Leiden
# ----------------------------------------------------------------------------------------------------------------------
def _degrees(indptr, indices, weights):
    n = len(indptr) - 1
    deg = [0.0] * n
    loops = [0.0] * n
    for v in range(n):
        s = 0.0
        for e in range(indptr[v], indptr[v + 1]):
            s += weights[e]
            if indices[e] == v:
                loops[v] += weights[e]
        deg[v] = s
    m2 = 0.0
    for v in range(n):
        m2 += deg[v]
    return deg, loops, m2


def _leiden_move(indptr, indices, weights, gamma, rng, init):
    """Local moving from the partition ``init`` (ids < n).  Returns (comm list, moved_any)."""
    n = len(indptr) - 1
    deg, loops, m2 = _degrees(indptr, indices, weights)
    comm = list(init)
    if m2 == 0.0:
        return comm, False
    tot = [0.0] * n
    in_ = [0.0] * n
    size = [0] * n
    for v in range(n):
        tot[comm[v]] += deg[v]
        size[comm[v]] += 1
    for v in range(n):
        for e in range(indptr[v], indptr[v + 1]):
            if comm[indices[e]] == comm[v]:
                in_[comm[v]] += weights[e]
    empty = [c for c in range(n - 1, -1, -1) if size[c] == 0]      # pop() gives the smallest empty id first
    order = _visit_order(n, rng)
    neigh_w = [-1.0] * n
    improved = False
    new_q = _quality(in_, tot, m2, gamma)
    active = [True] * n
    while True:
        cur_q = new_q
        moves = 0
        next_active = [False] * n
        for v in order:
            if not active[v]:
                continue
            c_old = comm[v]
            kv = deg[v]
            seen = [c_old]
            neigh_w[c_old] = 0.0
            for e in range(indptr[v], indptr[v + 1]):
                u = indices[e]
                if u == v:
                    continue
                c = comm[u]
                if neigh_w[c] == -1.0:
                    neigh_w[c] = 0.0
                    seen.append(c)
                neigh_w[c] += weights[e]
            tot[c_old] -= kv
            in_[c_old] -= 2.0 * neigh_w[c_old] + loops[v]
            size[c_old] -= 1
            best = c_old
            best_gain = neigh_w[c_old] - gamma * tot[c_old] * kv / m2
            for c in seen[1:]:
                g = neigh_w[c] - gamma * tot[c] * kv / m2
                if g > best_gain:
                    best_gain = g
                    best = c
            w_best = neigh_w[best]
            if best_gain < 0.0 and size[c_old] > 0:
                best = empty.pop()                                 # alone is better than any neighbour
                w_best = 0.0
            tot[best] += kv
            in_[best] += 2.0 * w_best + loops[v]
            size[best] += 1
            comm[v] = best
            if best != c_old:
                moves += 1
                if size[c_old] == 0:
                    empty.append(c_old)
                for e in range(indptr[v], indptr[v + 1]):
                    u = indices[e]
                    if u != v and comm[u] != best:
                        next_active[u] = True
            for c in seen:
                neigh_w[c] = -1.0
        new_q = _quality(in_, tot, m2, gamma)
        if moves > 0:
            improved = True
        if not (moves > 0 and new_q - cur_q > MIN_GAIN):
            break
        active = next_active
    return comm, improved


def _leiden_refine(indptr, indices, weights, gamma, rng, comm):
    """Refined partition (list, ids = a member's index) of the partition ``comm``."""
    n = len(indptr) - 1
    deg, _, m2 = _degrees(indptr, indices, weights)
    ref = list(range(n))
    if m2 == 0.0:
        return ref
    ktot = [0.0] * n                     # K_C of the unrefined communities
    for v in range(n):
        ktot[comm[v]] += deg[v]
    ext = [0.0] * n                      # E(R, C - R) per refined group (initially per node)
    for v in range(n):
        s = 0.0
        for e in range(indptr[v], indptr[v + 1]):
            u = indices[e]
            if u != v and comm[u] == comm[v]:
                s += weights[e]
        ext[v] = s
    rtot = deg[:]
    rsize = [1] * n
    neigh_w = [-1.0] * n
    for v in _visit_order(n, rng):
        if rsize[ref[v]] != 1:
            continue
        kc = ktot[comm[v]]
        kv = deg[v]
        if not (ext[v] >= gamma * kv * (kc - kv) / m2):
            continue
        seen = []
        for e in range(indptr[v], indptr[v + 1]):
            u = indices[e]
            if u == v or comm[u] != comm[v]:
                continue
            r = ref[u]
            if neigh_w[r] == -1.0:
                neigh_w[r] = 0.0
                seen.append(r)
            neigh_w[r] += weights[e]
        best = -1
        best_gain = 0.0
        for r in seen:
            if not (ext[r] >= gamma * rtot[r] * (kc - rtot[r]) / m2):
                continue
            g = neigh_w[r] - gamma * rtot[r] * kv / m2
            if g > best_gain:
                best_gain = g
                best = r
        if best >= 0:
            ext[best] = ext[best] + ext[v] - 2.0 * neigh_w[best]
            rtot[best] += kv
            rsize[best] += 1
            rsize[v] = 0
            ref[v] = best
        for r in seen:
            neigh_w[r] = -1.0
    return ref


def _leiden_sequential(indptr, indices, weights, gamma: float = 1.0, seed: int = 0) -> np.ndarray:
    """Part B'."""
    indptr0 = [int(x) for x in np.asarray(indptr)]
    indices0 = [int(x) for x in np.asarray(indices)]
    weights0 = [float(x) for x in np.asarray(weights, dtype=np.float64)]
    n = len(indptr0) - 1
    rng = SplitMix64(seed)
    gamma = float(gamma)
    partition = list(range(n))                     # community of every original node
    for _ in range(LEIDEN_MAX_ITERATIONS):
        g_indptr, g_indices, g_weights = indptr0, indices0, weights0
        node_of = list(range(n))                   # aggregate node of every original node
        init = partition[:]
        any_move = False
        while True:
            comm, moved = _leiden_move(g_indptr, g_indices, g_weights, gamma, rng, init)
            any_move = any_move or moved
            ng = len(g_indptr) - 1
            if len(set(comm)) == ng:
                partition = [comm[node_of[v]] for v in range(n)]
                break
            ref = _leiden_refine(g_indptr, g_indices, g_weights, gamma, rng, comm)
            g_indptr, g_indices, g_weights, renum = _aggregate(g_indptr, g_indices, g_weights, ref)
            # unrefined communities, renumbered by ascending id, become the next level's starting partition
            cren = {c: i for i, c in enumerate(sorted(set(comm)))}
            init = [0] * len(renum)
            for v in range(ng):
                init[renum[ref[v]]] = cren[comm[v]]
            node_of = [renum[ref[node_of[v]]] for v in range(n)]
            if len(renum) == ng:                   # refinement merged nothing: the partition is final
                partition = [init[node_of[v]] for v in range(n)]
                break
        # ids < n for the next iteration: renumber by ascending id
        pren = {c: i for i, c in enumerate(sorted(set(partition)))}
        partition = [pren[c] for c in partition]
        if not any_move:
            break
    return np.asarray(partition, dtype=np.int64)


def leiden(indptr, indices, weights, gamma: float = 1.0, seed: int = 0, presweeps: int = PRESWEEPS,
           levels: int = PRESWEEP_LEVELS, refine_sweeps: int = REFINE_SWEEPS) -> np.ndarray:
    """Part A, part B' on the aggregated graph, part C on the way back down."""
    graphs, members = _presweep_levels(indptr, indices, weights, gamma, presweeps, levels)
    lab = _leiden_sequential(*graphs[-1], gamma, seed)
    return _refine_down(graphs, members, lab, gamma, refine_sweeps)


def modularity(indptr, indices, weights, labels, gamma: float = 1.0) -> float:
    """Q of a labelling on a symmetric CSR graph (diagnostic)."""
    indptr = np.asarray(indptr)
    indices = np.asarray(indices)
    weights = np.asarray(weights, dtype=np.float64)
    labels = np.asarray(labels)
    n = len(indptr) - 1
    deg = np.add.reduceat(np.concatenate([weights, [0.0]]), indptr[:-1]) * (np.diff(indptr) > 0)
    m2 = deg.sum()
    rows = np.repeat(np.arange(n), np.diff(indptr))
    same = labels[rows] == labels[indices]
    k = labels.max() + 1
    in_c = np.bincount(labels[rows][same], weights=weights[same], minlength=k)
    tot_c = np.bincount(labels, weights=deg, minlength=k)
    return float((in_c / m2 - gamma * (tot_c / m2) ** 2).sum())
