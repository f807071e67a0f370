# where the GPU idles during a fit with the default number of device contexts:  bash profiles/tools/gap_analysis.sh <tag>
set -u
tag=${1:-r04x}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_gap
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gap -- python $repo/bench.py --steps 1 --warmup 2 --no-cpu-baseline --resident-steps 0 --instrumented-steps 0 --no-exclusive > $out/${tag}_gap.log 2>&1
t=$(find /tmp/prof_gap -name "*kernel_trace.csv" | head -1)
python - "$t" > $out/${tag}_gaps.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last fit starts with the last upload: its first expansion kernel after a pause
starts = [i for i, r in enumerate(rows) if "k_expand_packed16" in r[2] or "k_validate" in r[2]]
# group into fits: a new fit when the previous such kernel is > 50 ms earlier
fits = []
for i in starts:
    if not fits or rows[i][0] - rows[fits[-1][-1]][0] > 50e6: fits.append([i])
    else: fits[-1].append(i)
i0 = fits[-1][0]
sel = rows[i0:]
t0 = sel[0][0]
print(f"last fit: {len(sel)} kernels over {(sel[-1][1] - t0) / 1e6:.1f} ms")
cur_s, cur_e = sel[0][0], sel[0][1]
busy = 0; gaps = []
prev = sel[0][2]
for s, e, n in sel[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, cur_e - t0, prev, n)); busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev = n
busy += cur_e - cur_s
span = cur_e - t0
print(f"busy {busy / 1e6:.1f} ms of {span / 1e6:.1f} ({busy / span:.3f})")
W = 10e6
nb = int(span // W) + 1
idle = [0.0] * nb
for g, at, _, _ in gaps:
    idle[int(at // W)] += g
print("idle ms per 10-ms window:", " ".join(f"{x / 1e6:.1f}" for x in idle))
print("largest gaps (ms, at ms, after kernel -> before kernel):")
for g, at, a, b in sorted(gaps, reverse=True)[:40]:
    print(f"  {g / 1e6:6.2f} at {at / 1e6:7.1f}  {a[:48]:48s} -> {b[:48]}")
small = sum(g for g, *_ in gaps if g < 50e3)
print(f"gaps below 50 us: {small / 1e6:.1f} ms in {sum(1 for g, *_ in gaps if g < 50e3)} gaps; above: {sum(g for g, *_ in gaps if g >= 50e3) / 1e6:.1f} ms")
PY
cat $out/${tag}_gaps.txt
