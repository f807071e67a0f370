"""Stress: many consecutive fits on 2-3 device contexts must all give the single-context result (race detector)."""
import sys, os, warnings, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from doubletdetection_amd import BoostClassifier
from doubletdetection_amd._synthetic import make_counts

os.environ.setdefault("DDX_ARENA_GUARD", "1")
warnings.simplefilter("ignore")
bad = 0
for (n, g, d, algo, scal) in ((50_000, 20_000, 0.05, "phenograph", False), (30_000, 15_000, 0.05, "louvain", True), (20_000, 12_000, 0.06, "leiden", False)):
    X = make_counts(n, g, density=d, device="cuda:0", seed=n)
    kw = dict(n_iters=7, clustering_algorithm=algo, standard_scaling=scal, random_state=1, n_jobs=-1)
    base = BoostClassifier(streams_per_device=1, **kw).fit(X)
    t0 = time.perf_counter()
    for rep in range(12):
        for s in (2, 3):
            clf = BoostClassifier(streams_per_device=s, **kw).fit(X)
            ok = (np.array_equal(clf.all_log_p_values_, base.all_log_p_values_, equal_nan=True) and np.array_equal(clf.communities_, base.communities_)
                  and np.array_equal(clf.synth_communities_, base.synth_communities_))
            if not ok:
                bad += 1
                print("MISMATCH", n, algo, "rep", rep, "streams", s, flush=True)
    print(f"{n} x {g} {algo} scaling={scal}: 24 fits on 2/3 contexts in {time.perf_counter() - t0:.1f} s, mismatches so far {bad}", flush=True)
print("stress OK" if bad == 0 else f"stress FAILED: {bad}")
