import sys, time, numpy as np, torch
emb = np.load(sys.argv[1]).astype(np.float64)
M, C = emb.shape
k = 30
E = torch.from_numpy(emb)
n2 = (E * E).sum(1)

def lloyd(Kc, dims, iters):
    o = np.argsort(emb[:, 0], kind="stable")
    cen = emb[o[(np.arange(Kc) * M) // Kc + M // (2 * Kc)], :dims].copy()
    X = emb[:, :dims]
    for it in range(iters + 1):
        d = (X * X).sum(1)[:, None] - 2 * X @ cen.T + (cen * cen).sum(1)[None, :]
        lab = d.argmin(1)
        if it == iters: break
        for c in range(Kc):
            m = lab == c
            if m.any(): cen[c] = X[m].mean(0)
    return lab

def bound_T(order, W=8192):
    # kth smallest exact d2 within the W positions around each query in `order`
    Eo = E[order]; n2o = n2[order]
    T = np.empty(M)
    B = 512
    for s in range(0, M, B):
        lo = min(max(s + B // 2 - W // 2, 0), M - W)
        d = n2o[s:s+B, None] + n2o[None, lo:lo+W] - 2 * Eo[s:s+B] @ Eo[lo:lo+W].T
        r = torch.arange(d.shape[0]); d[r, r + (s - lo)] = float("inf")
        T[s:s+B] = torch.topk(d, k, dim=1, largest=False).values[:, -1].numpy()
    Tfull = np.empty(M); Tfull[order] = T
    return Tfull

def ncand(T):
    tot = 0
    Tt = torch.from_numpy(T)
    for s in range(0, M, 4000):
        d = n2[s:s+4000, None] + n2[None, :] - 2 * E[s:s+4000] @ E.T
        tot += (d <= Tt[s:s+4000, None]).sum().item()
    return tot / M - 1

def study(lab, T, gq, gc, tag):
    Kc = lab.max() + 1
    cen = np.zeros((Kc, C))
    for c in range(Kc): cen[c] = emb[lab == c].mean(0)
    cord = np.argsort(cen[:, 0]); rank = np.empty(Kc, int); rank[cord] = np.arange(Kc)
    order = np.lexsort((emb[:, 0], rank[lab]))
    groups_q, groups_c = [], []
    pos = 0
    for c in cord:
        n = (lab == c).sum(); idx = order[pos:pos + n]; pos += n
        for s in range(0, n, gq): groups_q.append((c, idx[s:s + gq]))
        for s in range(0, n, gc): groups_c.append((c, idx[s:s + gc]))
    nq, nc = len(groups_q), len(groups_c)
    U = cen[None, :, :] - cen[:, None, :]
    nr = np.linalg.norm(U, axis=2, keepdims=True); nr[nr == 0] = 1
    U = U / nr
    qcell = np.array([g[0] for g in groups_q]); ccell = np.array([g[0] for g in groups_c])
    qlo = np.empty((nq, Kc)); qhi = np.empty((nq, Kc)); qT = np.empty(nq)
    for i, (A, idx) in enumerate(groups_q):
        p = emb[idx] @ U[A].T
        qlo[i] = p.min(0); qhi[i] = p.max(0); qT[i] = T[idx].max()
    clo = np.empty((nc, Kc)); chi = np.empty((nc, Kc)); c1l = np.empty(nc); c1h = np.empty(nc)
    for i, (B, idx) in enumerate(groups_c):
        p = emb[idx] @ U[:, B].T
        clo[i] = p.min(0); chi[i] = p.max(0); c1l[i] = emb[idx, 0].min(); c1h[i] = emb[idx, 0].max()
    w = np.array([len(g[1]) for g in groups_c], dtype=float)
    keep = 0.0; tot = 0.0
    for i in range(nq):
        A = qcell[i]; idx = groups_q[i][1]
        gap = np.maximum(np.maximum(clo[:, A] - qhi[i, ccell], qlo[i, ccell] - chi[:, A]), 0)
        # same cell: PC1 gap
        g1 = np.maximum(np.maximum(c1l - emb[idx, 0].max(), emb[idx, 0].min() - c1h), 0)
        gap = np.where(ccell == A, g1, np.maximum(gap, g1))
        keep += len(idx) * (w * (gap * gap <= qT[i])).sum(); tot += len(idx) * w.sum()
    print("  %s gq=%3d gc=%3d: kept %.3f" % (tag, gq, gc, keep / tot), flush=True)
    return order

o1 = np.argsort(emb[:, 0], kind="stable")
T1 = bound_T(o1)
print("PC1 order: bound T mean %.1f, candidates/query %.1f" % (T1.mean(), ncand(T1)))
for Kc, dims, iters in ((128, 30, 2), (256, 30, 2), (512, 30, 2), (256, 30, 0), (256, 30, 6), (256, 12, 2), (1024, 30, 2)):
    t0 = time.time()
    lab = lloyd(Kc, dims, iters)
    sizes = np.bincount(lab, minlength=Kc)
    print("Kc=%d dims=%d iters=%d: cell sizes min %d med %d max %d" % (Kc, dims, iters, sizes.min(), np.median(sizes), sizes.max()))
    order = study(lab, T1, 32, 16, "T(pc1 order)")
    T2 = bound_T(order)
    print("  cell order: bound T mean %.1f, candidates/query %.1f" % (T2.mean(), ncand(T2)))
    for gq, gc in ((32, 16), (128, 16), (128, 128), (256, 128)):
        study(lab, T2, gq, gc, "T(cell order)")
