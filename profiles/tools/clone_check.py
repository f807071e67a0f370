"""Does a cloned context reproduce its source?  (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from doubletdetection_amd import _lib
from doubletdetection_amd._synthetic import make_counts

counts = make_counts(900, 700, density=0.15, n_types=5, seed=21)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 45
rng = np.random.default_rng(2)
parents = [rng.choice(900, size=(225, 2), replace=False) for _ in range(2)]
a = _lib.Context(0)
a.upload_raw(counts)
var = a.gene_variances()
top = np.argsort(var)[-600:]
a.select_columns(top)
b = _lib.Context(0)
b.clone_counts_from(a)
q0 = np.random.RandomState(2).normal(size=(600, C + 10)).astype(np.float32).astype(np.float64)

def run(c, p):
    c.create_doublets(p)
    c.lognormalise(0.1)
    c.pca(C, q0)
    e, s = c.embedding_f64()
    c.knn(10, True)
    idx, _ = c.get_knn()
    g1 = c.build_graph(2)
    g2 = c.build_graph(2)
    print("   graph sizes", len(g1[1]), len(g2[1]), "rebuild equal", all(np.array_equal(u, v) for u, v in zip(g1, g2)), "max col", g1[1].max(), "indptr ok", bool(np.all(np.diff(g1[0]) >= 0)), g1[0][-1])
    try:
        co = c.coarsen_graph(4.0)
    except Exception as ex:
        print("   coarsen failed:", ex)
        co = g1
    return e, s, idx, c.aug_values()[0], co

for it in range(2):
    ea, sa, ia, xa, ca = run(a, parents[it])
    eb, sb, ib, xb, cb = run(b, parents[it])
    print("coarse equal", [bool(np.array_equal(u, v)) for u, v in zip(ca, cb)], len(ca[1]), len(cb[1]))
    print("iter", it, "x equal", np.array_equal(xa, xb), "emb maxdiff", np.abs(ea - eb).max(), "sing", np.abs(sa - sb).max(), "knn equal", np.array_equal(ia, ib))
sa_ = a.get_counts(); sb_ = b.get_counts()
print("counts equal", (sa_ != sb_).nnz == 0, np.array_equal(a.lib_size(), b.lib_size()))

import threading
res = {}
def work(c, name, p):
    res[name] = run(c, p)
for it in range(2):
    ta = threading.Thread(target=work, args=(a, "a", parents[it])); tb = threading.Thread(target=work, args=(b, "b", parents[1 - it]))
    ta.start(); tb.start(); ta.join(); tb.join()
    ra = run(a, parents[1 - it])
    print("threaded: b vs sequential a on same parents: emb", np.abs(res["b"][0] - ra[0]).max(), "knn", np.array_equal(res["b"][2], ra[2]),
          "coarse", [bool(np.array_equal(u, v)) for u, v in zip(res["b"][4], ra[4])])
