# union of kernel intervals over one timed fit of the default bench (all contexts), in 5 ms windows: where is the GPU idle?
#   bash profiles/tools/busy_timeline.sh <tag>
set -u
tag=${1:-r03x}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_busy
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_busy -- python $repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline --resident-steps 0 --instrumented-steps 0 --no-exclusive > $out/${tag}_busy.log 2>&1
t=$(find /tmp/prof_busy -name "*kernel_trace.csv" | head -1)
python - "$t" > $out/${tag}_busy_timeline.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last fit: from the last k_expand_packed16 (or k_validate_csr) launch to the end
starts = [a for a, b, n in rows if "k_validate_csr" in n]
t0 = starts[-1] - 8_000_000 if starts else rows[0][0]
rows = [(a, b, n) for a, b, n in rows if a >= t0]
t0 = rows[0][0]; t1 = max(b for a, b, n in rows)
W = 5_000_000
nb = (t1 - t0) // W + 1
busy = [0] * nb
cur_a, cur_b = rows[0][0], rows[0][1]
spans = []
for a, b, n in rows[1:]:
    if a > cur_b:
        spans.append((cur_a, cur_b)); cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
spans.append((cur_a, cur_b))
for a, b in spans:
    k = (a - t0) // W
    while a < b:
        e = min(b, t0 + (k + 1) * W)
        busy[k] += e - a
        a = e; k += 1
tot = sum(busy)
print(f"fit window {(t1 - t0) / 1e6:.1f} ms, busy {tot / 1e6:.1f} ms = {tot / (t1 - t0):.3f}")
for k, v in enumerate(busy):
    print(f"{k * 5:5d} ms  {v / W:5.2f}  " + "#" * int(40 * v / W))
gaps = sorted(((spans[i + 1][0] - spans[i][1]) / 1e3, (spans[i][1] - t0) / 1e6) for i in range(len(spans) - 1))[-12:]
print("largest gaps (us, at ms):", [(round(g), round(at, 1)) for g, at in gaps])
PY
cd $repo; cat $out/${tag}_busy_timeline.txt
