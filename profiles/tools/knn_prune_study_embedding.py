import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from doubletdetection_amd._synthetic import make_counts
from oracle import dd_oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
t = time.time()
X = make_counts(N, 30000, density=0.03)
print("counts", X.shape, X.nnz, time.time() - t, flush=True)
raw = orc.coerce_counts(X)
top, raw = orc.select_hvg(raw, 10000)
lib = orc.library_sizes(raw)
normed = orc.l1_normalise_rows(raw)
rng = np.random.default_rng(0)
par = orc.draw_parents(rng, N, 0.25, False)
syn = orc.create_doublets(raw, par)
aug, _, _ = orc.lognormalise(normed, lib, syn, 0.1)
print("aug", aug.shape, time.time() - t, flush=True)
emb = orc.pca_sklearn(aug, 30, 0).astype(np.float32)
print("emb", emb.shape, emb.std(axis=0), time.time() - t, flush=True)
np.save(f"/tmp/emb_{N}.npy", emb)
