#!/bin/bash
# kernel-trace only: per-kernel totals and the launch sequence of the last iterations -> gpurun_out/<tag>_*
tag=${1:-r01x}; repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_tl
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tl -- python $repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > "$out/${tag}_tl.log" 2>&1
f=$(find /tmp/prof_tl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv"
t=$(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python - "$t" > "$out/${tag}_launch_sequence.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-1300:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:10.3f} ms  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  {r["Kernel_Name"][:100]}')
PY
