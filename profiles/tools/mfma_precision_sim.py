"""CPU simulation: scikit-learn's randomized PCA (7 power iterations, width 40) on a log-normalised augmented matrix,
with the two operator products D Q / D^T Y evaluated in reduced-precision schemes; error of the 30 score columns
against the all-float64 run.  (Would dense MFMA tiles with split half-precision operands meet the 1e-5 bar?)

    python profiles/tools/mfma_precision_sim.py [cells genes hvg]
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
D = (aug - z).astype(np.float64)                  # stored entries x - z, zeros elsewhere (what the kernels multiply)
D[np.abs(D) < 1e-12] = 0.0
M, H = D.shape
print("matrix", D.shape, "density of stored entries", round(float((D != 0).mean()), 4), flush=True)
m = D.mean(axis=0)


def pow2_scale(v, target=2.0 ** 13):
    mx = np.abs(v).max(axis=0, keepdims=True)
    mx[mx == 0] = 1.0
    return 2.0 ** np.floor(np.log2(target / mx))


def split(v, kind):
    if kind == "f16":
        hi = v.astype(np.float16).astype(np.float64)
        lo = (v - hi).astype(np.float16).astype(np.float64)
    else:                                           # bfloat16: keep the upper 16 bits of the float32 (round to nearest even)
        def bf(x):
            u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
            u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
            return u.astype(np.uint32).view(np.float32).astype(np.float64)
        hi = bf(v)
        lo = bf(v - hi)
    return hi, lo


class Product:
    def __init__(self, scheme, chunk=128):
        self.scheme, self.chunk = scheme, chunk
        if scheme in ("f16x3", "bf16x3"):
            kind = scheme[:-2]
            self.sD = float(pow2_scale(D.reshape(-1, 1))[0, 0]) if kind == "f16" else 1.0
            self.Dh, self.Dl = (a.astype(np.float32) for a in split(D * self.sD, kind))
            self.kind = kind

    def _mm(self, A32h, A32l, B, trans):
        """(A or A^T) @ B with split operands, float32 accumulation inside chunks of the contraction, float64 across."""
        sB = pow2_scale(B) if self.kind == "f16" else np.ones((1, B.shape[1]))
        Bh, Bl = (a.astype(np.float32) for a in split(B * sB, self.kind))
        K = A32h.shape[0] if trans else A32h.shape[1]
        out = np.zeros(((A32h.shape[1] if trans else A32h.shape[0]), B.shape[1]))
        for k0 in range(0, K, self.chunk):
            k1 = min(K, k0 + self.chunk)
            ah = A32h[k0:k1].T if trans else A32h[:, k0:k1]
            al = A32l[k0:k1].T if trans else A32l[:, k0:k1]
            part = ah @ Bh[k0:k1] + (ah @ Bl[k0:k1] + al @ Bh[k0:k1])        # float32 sgemm
            out += part.astype(np.float64)
        return out / (self.sD * sB)

    def DQ(self, Q):
        if self.scheme == "f64":
            return D @ Q
        if self.scheme == "f32op":                  # shipped: float32 copy of the operand, float64 sums
            return D @ Q.astype(np.float32).astype(np.float64)
        if self.scheme == "f32gemm":
            return (D.astype(np.float32) @ Q.astype(np.float32)).astype(np.float64)
        return self._mm(self.Dh, self.Dl, Q, False)

    def DtY(self, Y):
        if self.scheme == "f64":
            return D.T @ Y
        if self.scheme == "f32op":
            return D.T @ Y.astype(np.float32).astype(np.float64)
        if self.scheme == "f32gemm":
            return (D.T.astype(np.float32) @ Y.astype(np.float32)).astype(np.float64)
        return self._mm(self.Dh, self.Dl, Y, True)


def scores(P, n_comps=30, size=40, n_iter=7):
    Q = orc.pca_start_matrix(0, H, size)
    A_Q = lambda Q: P.DQ(Q) - np.outer(np.ones(M), m @ Q)
    At_Y = lambda Y: P.DtY(Y) - np.outer(m, Y.sum(axis=0))
    norm = lambda Y: sla.qr(Y, mode="economic", check_finite=False)[0]
    for _ in range(n_iter):
        Q = norm(A_Q(Q))
        Q = norm(At_Y(Q))
    Q = norm(A_Q(Q))
    B = At_Y(Q).T
    Uhat, s, Vt = sla.svd(B, full_matrices=False)
    U = (Q @ Uhat)[:, :n_comps]
    Vt = Vt[:n_comps]
    sg = np.sign(Vt[np.arange(n_comps), np.argmax(np.abs(Vt), axis=1)])
    return U * sg * s[:n_comps]


ref = scores(Product("f64"))
for scheme in ("f32op", "f32gemm", "f16x3", "bf16x3"):
    for chunk in ((128, 1024) if scheme.endswith("x3") else (128,)):
        e = scores(Product(scheme, chunk))
        rel = np.linalg.norm(e - ref, axis=0) / np.linalg.norm(ref, axis=0)
        print(f"{scheme:8s} chunk {chunk:5d}: max rel err of a score column {rel.max():.2e} (column {rel.argmax()}), first 12 columns {rel[:12].max():.2e}", flush=True)
