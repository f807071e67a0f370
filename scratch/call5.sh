#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q > $out/r06h_gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -4 $out/r06h_gpu_tests.log
python bench.py --steps 10 --warmup 5 > $out/r06h_bench_line.json 2> $out/r06h_bench_err.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06h_bench_line.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["value_resident"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["operator_product"])
PY
