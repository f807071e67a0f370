# single-stream kernel profile of the bench + the community-detection / parity GPU tests:  bash profiles/tools/prof_1stream.sh <tag>
set -u
tag=${1:-r03x}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_stats1
DDX_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats1 -- python $repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --resident-steps 0 --instrumented-steps 0 --no-exclusive > $out/${tag}_1stream.log 2>&1
f1=$(find /tmp/prof_stats1 -name "*kernel_stats.csv" | head -1)
[ -n "$f1" ] && cp "$f1" $out/${tag}_kernel_stats_1stream.csv
t=$(find /tmp/prof_stats1 -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python - "$t" > $out/${tag}_launch_sequence_1stream.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-900:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:10.3f} ms  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  {r["Kernel_Name"][:90]}')
PY
cd $repo
python - "$tag" <<'PY'
import csv, sys
rows=list(csv.DictReader(open("gpurun_out/%s_kernel_stats_1stream.csv" % sys.argv[1])))
for r in rows[:28]:
    if 'at::' in r['Name'] or 'rocprim' in r['Name']: continue
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
