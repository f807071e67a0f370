# kernels of the LAST fit of a short bench run from its first kernel to the first doublet kernel (the per-fit fixed part), with start
# times relative to the first, and the largest gaps:  bash profiles/tools/prologue_timeline.sh <tag>
set -u
tag=${1:-r05x}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_pro
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_pro -- python $repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline --resident-steps 0 --instrumented-steps 0 --no-exclusive > $out/${tag}_pro.log 2>&1
t=$(find /tmp/prof_pro -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/prof_pro -name "*memory_copy_trace.csv" | head -1)
python - "$t" "$m" > $out/${tag}_prologue_timeline.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]) for r in csv.DictReader(open(sys.argv[1]))]
try:
    for r in csv.DictReader(open(sys.argv[2])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + str(r.get("Bytes", ""))))
except Exception as e:
    print("no copy trace", e)
rows.sort()
starts = [a for a, b, n in rows if "k_validate_csr" in n]
t0 = starts[-1]
fills = [a for a, b, n in rows if ("k_bp_synth" in n or "k_doublet_fill" in n) and a > t0]
t1 = fills[0] if fills else rows[-1][1]
sel = [(a, b, n) for a, b, n in rows if t0 - 12_000_000 <= a <= t1 + 1_000_000]
print(f"validate_csr -> first doublet kernel: {(t1 - t0) / 1e6:.2f} ms")
prev_end = None
for a, b, n in sel:
    gap = (a - prev_end) / 1e3 if prev_end else 0.0
    flag = "   <-- gap %.0f us" % gap if gap > 150 else ""
    if (b - a) > 20_000 or gap > 150:
        print(f"{(a - t0) / 1e6:9.3f} ms  {(b - a) / 1e3:9.1f} us  {n}{flag}")
    prev_end = max(prev_end or b, b)
PY
cat $out/${tag}_prologue_timeline.txt | head -120
