"""Per-fit table of a rocprofv3 kernel_stats CSV (bench.py --steps 1 --warmup 1 = two fits), library kernels only:
    python profiles/tools/kernel_table.py gpurun_out/<tag>_kernel_stats_1stream.csv [fits]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
fits = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
skip = ("at::", "poisson", "partition_kernel", "block_reduce_kernel")
tot = 0.0
out = []
for r in rows:
    n = r["Name"]
    if any(s in n for s in skip):
        continue
    ms = float(r["TotalDurationNs"]) / 1e6 / fits
    tot += ms
    out.append((ms, n, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
print(f"library kernels: {tot:.1f} ms per fit")
for ms, n, calls, avg in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print(f"{n[:84]:84s} {calls:6d} calls {ms:8.2f} ms/fit {avg:8.1f} us")
