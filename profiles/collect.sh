#!/bin/bash
# Collects the per-round profile evidence on a GPU box:  bash profiles/collect.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of a short bench run        -> gpurun_out/<tag>_kernel_stats.csv
#   2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE / L2 hit-miss) -> gpurun_out/<tag>_pmc_counters.txt
# Counter passes never enable the sys/runtime/hip/hsa trace domains.  Copy the outputs into profiles/.
set -u
tag=${1:-r01x}
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
bench="python $repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --resident-steps 0 --instrumented-steps 0 --no-exclusive"
short="python $repo/bench.py --steps 1 --warmup 0 --iters 2 --no-cpu-baseline --resident-steps 0 --instrumented-steps 0 --no-exclusive"

cd /tmp
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $bench > "$out/${tag}_stats_bench.log" 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv"
# the same command on a single device context: every kernel alone on the GPU (what bench.py's `roofline` reports)
rm -rf /tmp/prof_stats1
DDX_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats1 -- $bench > "$out/${tag}_stats_bench_1stream.log" 2>&1
f1=$(find /tmp/prof_stats1 -name "*kernel_stats.csv" | head -1)
[ -n "$f1" ] && cp "$f1" "$out/${tag}_kernel_stats_1stream.csv"
t=$(find /tmp/prof_stats -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python - "$t" > "$out/${tag}_launch_sequence.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-1300:]:          # the last boosting iterations of the timed fit
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:10.3f} ms  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  {r["Kernel_Name"][:100]}')
PY

for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    rm -rf /tmp/prof_$name
    timeout 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/prof_$name -- $short > "$out/${tag}_pmc_$name.log" 2>&1
done
cd "$repo"
{
    echo "# rocprofv3 --kernel-trace --pmc <counters> (one pass per tag) -- $short"
    echo "# per-dispatch averages; FETCH_SIZE/WRITE_SIZE in KB; fetch_x2_MB = FETCH_SIZE doubled (gfx950 wide-read correction)"
    python profiles/summarise_pmc.py fetch=/tmp/prof_fetch write=/tmp/prof_write l2=/tmp/prof_l2
} > "$out/${tag}_pmc_counters.txt"
# the table bench.py's `roofline.traffic` is read from (profiles/pmc_traffic.json: copy it there together with the summaries)
python profiles/summarise_pmc.py --traffic-json "$out/pmc_traffic.json" fetch=/tmp/prof_fetch write=/tmp/prof_write source="profiles/${tag}_pmc_counters.txt (separate rocprofv3 --pmc passes of: $short)"
tail -3 "$out/${tag}_stats_bench.log"
