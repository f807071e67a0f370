#!/bin/bash
out=gpurun_out; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q > $out/r06p_gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -3 $out/r06p_gpu_tests.log
python bench.py --steps 10 --warmup 5 > $out/r06p_bench_line.json 2> $out/r06p_bench_err.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06p_bench_line.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["value_resident"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["cpu_baseline"]["value"])
PY
