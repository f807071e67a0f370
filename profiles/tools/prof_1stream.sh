set -u
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_stats1
DDX_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats1 -- python $repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --resident-steps 0 --no-exclusive > $out/r03b_1stream.log 2>&1
f1=$(find /tmp/prof_stats1 -name "*kernel_stats.csv" | head -1)
[ -n "$f1" ] && cp "$f1" $out/r03b_kernel_stats_1stream.csv
t=$(find /tmp/prof_stats1 -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python - "$t" > $out/r03b_launch_sequence_1stream.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-900:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:10.3f} ms  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  {r["Kernel_Name"][:90]}')
PY
cd $repo
head -30 $out/r03b_kernel_stats_1stream.csv
python -m pytest tests/test_clustering_independent.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
