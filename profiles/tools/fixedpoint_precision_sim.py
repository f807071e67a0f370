"""CPU simulation (round 5): scikit-learn's randomized PCA (7 power iterations, width 40) on a log-normalised augmented matrix
with the operand of every operator product quantised column by column to an n-bit fixed point (what the bit-plane products
feed the int8 matrix cores as 8-bit digits: 3 digits = 23 bits below the column maximum, 4 digits = 30); error of the 30
score columns against the all-float64 run.  Pessimistic: here EVERY stored entry multiplies the quantised operand (on the
device only the entries equal to 1 do, the others multiply the float32 copy).

    python profiles/tools/fixedpoint_precision_sim.py [cells genes hvg]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.linalg as sla
from doubletdetection_amd._synthetic import make_counts
from oracle import dd_oracle as orc

N, G, HV = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (24000, 8000, 3000)))
X = make_counts(N, G, density=0.04, seed=5)
var = orc.gene_variances(X)
top = np.sort(np.argsort(var)[-HV:])
raw = X.tocsc()[:, top].tocsr()
parents = np.random.default_rng(0).choice(N, size=(N // 4, 2), replace=False)
synth = orc.create_doublets(raw, parents)
aug, _, _ = orc.lognormalise(orc.l1_normalise_rows(raw), orc.library_sizes(raw), synth, 0.1)
aug = np.asarray(aug, dtype=np.float32)
z = np.float32(np.log(np.float32(0.1)))
D = (aug - z).astype(np.float64)
D[np.abs(D) < 1e-12] = 0.0
M, H = D.shape
print("matrix", D.shape, "density", round(float((D != 0).mean()), 4), flush=True)
m = D.mean(axis=0)
# the bit-plane split: ones of the ORIGINAL count matrix carry a row-only value s_i
ones = np.zeros(D.shape, dtype=bool)
import scipy.sparse as sp
full_counts = sp.vstack([raw, synth]).toarray()
ones = full_counts == 1.0
srow = np.where(ones.any(axis=1), np.max(np.where(ones, D, 0.0), axis=1), 0.0)
B = ones.astype(np.float64)
R = np.where(ones, 0.0, D)
print("share of entries equal to 1:", round(float(ones.sum() / (D != 0).sum()), 4), flush=True)


def quant(V, bits):
    if bits is None:
        return V.astype(np.float32).astype(np.float64)
    mx = np.abs(V).max(axis=0, keepdims=True)
    mx[mx == 0] = 1.0
    e = np.floor(np.log2(mx)) + 1            # mx = f 2^e, 0.5 <= f < 1
    sc = 2.0 ** (bits - e)
    return np.rint(V * sc) / sc


class Product:
    def __init__(self, bits):
        self.bits = bits

    def DQ(self, Q):
        if self.bits == "f64":
            return D @ Q
        if self.bits == "f32op":
            return D @ Q.astype(np.float32).astype(np.float64)
        return srow[:, None] * (B @ quant(Q, self.bits)) + R @ Q.astype(np.float32).astype(np.float64)

    def DtY(self, Y):
        if self.bits == "f64":
            return D.T @ Y
        if self.bits == "f32op":
            return D.T @ Y.astype(np.float32).astype(np.float64)
        return B.T @ quant(srow[:, None] * Y, self.bits) + R.T @ Y.astype(np.float32).astype(np.float64)


def scores(P, n_comps=30, size=40, n_iter=7):
    Q = orc.pca_start_matrix(0, H, size)
    A_Q = lambda Q: P.DQ(Q) - np.outer(np.ones(M), m @ Q)
    At_Y = lambda Y: P.DtY(Y) - np.outer(m, Y.sum(axis=0))
    norm = lambda Y: sla.qr(Y, mode="economic", check_finite=False)[0]
    for _ in range(n_iter):
        Q = norm(A_Q(Q))
        Q = norm(At_Y(Q))
    Q = norm(A_Q(Q))
    Bm = At_Y(Q).T
    Uhat, s, Vt = sla.svd(Bm, full_matrices=False)
    U = (Q @ Uhat)[:, :n_comps]
    Vt = Vt[:n_comps]
    sg = np.sign(Vt[np.arange(n_comps), np.argmax(np.abs(Vt), axis=1)])
    return U * sg * s[:n_comps]


ref = scores(Product("f64"))
for bits in ("f32op", 30, 23, 22, 20, 16):
    e = scores(Product(bits))
    rel = np.linalg.norm(e - ref, axis=0) / np.linalg.norm(ref, axis=0)
    print(f"operand {str(bits):6s}: max rel err of a score column {rel.max():.2e} (column {rel.argmax()}), first 12 columns {rel[:12].max():.2e}", flush=True)
