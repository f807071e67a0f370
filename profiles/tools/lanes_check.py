"""streams=2 vs streams=1 on the failing configuration (diagnostic)"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from doubletdetection_amd import BoostClassifier
from doubletdetection_amd._synthetic import make_counts

counts = make_counts(900, 700, density=0.15, n_types=5, seed=21)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 45
algo = sys.argv[2] if len(sys.argv) > 2 else "louvain"
kw = dict(n_iters=4, n_top_var_genes=600, n_components=C, clustering_algorithm=algo, random_state=2)
warnings.simplefilter("ignore")
base = BoostClassifier(streams_per_device=1, **kw).fit(counts)
for rep in range(4):
    for s in (2, 3):
        clf = BoostClassifier(streams_per_device=s, **kw).fit(counts)
        print("rep", rep, "streams", s, "equal per iteration:", [bool(np.array_equal(clf.communities_[i], base.communities_[i])) for i in range(4)], flush=True)
