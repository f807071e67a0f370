import numpy as np, time
from scipy.linalg import eigh
from threadpoolctl import threadpool_limits
rng = np.random.default_rng(0)
for n in (480, 880):
    A = rng.normal(size=(n, n)); A = A + A.T
    for th in (1, 2, 4, 8, 16):
        with threadpool_limits(limits=th):
            eigh(A, subset_by_index=[n - 40, n - 1], driver="evr", check_finite=False)
            t0 = time.perf_counter()
            for _ in range(3):
                eigh(A, subset_by_index=[n - 40, n - 1], driver="evr", check_finite=False)
            print(n, th, "threads:", round((time.perf_counter() - t0) / 3 * 1e3, 1), "ms", flush=True)
