# How full is the GPU during the iterations of a fit?  Kernel trace of a short default bench run; for the last fit, from the first
# doublet kernel to the last kernel: share of the time with 0 / 1 / 2 / 3+ kernels in flight, and share of the time in which no kernel
# with at least 256 workgroups (one per CU) is in flight.     bash profiles/tools/concurrency_profile.sh <tag>
set -u
tag=${1:-r05x}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_conc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_conc -- python $repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline --resident-steps 0 --instrumented-steps 0 --no-exclusive > $out/${tag}_conc.log 2>&1
t=$(find /tmp/prof_conc -name "*kernel_trace.csv" | head -1)
python - "$t" > $out/${tag}_concurrency.txt <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    wg = max(1, int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1))
    grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid // wg if wg else 0))
rows.sort()
starts = [a for a, b, n, g in rows if "k_validate_csr" in n]
t_fit = starts[-1]
it0 = min(a for a, b, n, g in rows if a > t_fit and ("k_bp_synth" in n or "k_doublet_fill" in n))
sel = [(a, b, n, g) for a, b, n, g in rows if a >= it0]
t1 = max(b for a, b, n, g in sel)
ev = []
for a, b, n, g in sel:
    big = 1 if g >= 256 else 0
    ev.append((a, 1, big)); ev.append((b, -1, -big))
ev.sort()
hist = {}
nobig = 0
cur = 0; curbig = 0; last = it0
for t, d, db in ev:
    dt = t - last
    if dt > 0:
        hist[min(cur, 6)] = hist.get(min(cur, 6), 0) + dt
        if curbig == 0: nobig += dt
    cur += d; curbig += db; last = t
tot = t1 - it0
print(f"iterations window {(tot) / 1e6:.1f} ms of the last fit (under the tracer)")
for k in sorted(hist):
    print(f"  {k}{'+' if k == 6 else ' '} kernels in flight: {hist[k] / tot:6.3f}")
print(f"  no kernel with >= 256 workgroups in flight: {nobig / tot:6.3f}")
avg = sum(k * v for k, v in hist.items()) / tot
print(f"  mean kernels in flight {avg:.2f}")
PY
cd $repo; cat $out/${tag}_concurrency.txt
