"""CPU time a fit burns against the container's CFS quota (cgroup v2 cpu.max / cpu.stat): python profiles/tools/cpu_quota_check.py [fits]
A pod that shows 256 CPUs may be allowed 16 CPUs' worth of time per 100 ms; host threads beyond that are throttled, GPU-launching threads included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from doubletdetection_amd import BoostClassifier
from doubletdetection_amd._synthetic import make_counts


def stat():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return {k: int(v) for k, v in d.items()}
    except Exception:
        return {}


try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), " visible CPUs", os.cpu_count())
except Exception:
    print("no cgroup v2 cpu.max")
X = make_counts(100_000, 30_000, density=0.03, device="cuda:0", seed=20250227)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for i in range(n):
    a = stat(); t0 = time.perf_counter()
    BoostClassifier(random_state=0).fit(X)
    dt = time.perf_counter() - t0; b = stat()
    if a:
        print(f"fit {i}: {dt * 1e3:7.1f} ms   cpu time {(b['usage_usec'] - a['usage_usec']) / 1e3:8.1f} ms (user {(b['user_usec'] - a['user_usec']) / 1e3:.0f}, system {(b['system_usec'] - a['system_usec']) / 1e3:.0f})"
              f" = {(b['usage_usec'] - a['usage_usec']) / 1e6 / dt:5.1f} CPUs   throttled periods +{b['nr_throttled'] - a['nr_throttled']}, {(b['throttled_usec'] - a['throttled_usec']) / 1e3:.0f} ms", flush=True)
    else:
        print(f"fit {i}: {dt * 1e3:7.1f} ms", flush=True)
