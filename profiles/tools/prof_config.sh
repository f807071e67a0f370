# kernel statistics (single device context) + FETCH/WRITE counters of one BASELINE configuration:
#   bash profiles/tools/prof_config.sh <tag> <cells> <genes> <density> <iters>
set -u
tag=$1; cells=$2; genes=$3; dens=$4; iters=$5
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cmd="python $repo/bench.py --cells $cells --genes $genes --density $dens --iters $iters --steps 1 --warmup 1 --no-cpu-baseline --resident-steps 0 --instrumented-steps 0 --no-exclusive"
cd /tmp
rm -rf /tmp/pc_stats
DDX_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_stats -- $cmd > $out/${tag}_stats.log 2>&1
f=$(find /tmp/pc_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats_1stream.csv
for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    rm -rf /tmp/pc_$name
    DDX_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pc_$name -- $cmd > $out/${tag}_pmc_$name.log 2>&1
done
cd $repo
python profiles/summarise_pmc.py fetch=/tmp/pc_fetch write=/tmp/pc_write 2>/dev/null | grep -E "spmm_lds|knn_emit|knn_bound|knn_select|lv_sweep|doublet|mirror|lognorm" > $out/${tag}_pmc_counters.txt
python - "$tag" <<'PY'
import csv, sys
rows=list(csv.DictReader(open("gpurun_out/%s_kernel_stats_1stream.csv" % sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows if 'at::' not in r['Name'])
print(sys.argv[1], "total kernel ms (ddx + prims):", round(tot/1e6,1))
n=0
for r in rows:
    if 'at::' in r['Name']: continue
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f}  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
    n+=1
    if n>=14: break
PY
cat $out/${tag}_pmc_counters.txt | cut -c1-220
