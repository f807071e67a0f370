#!/bin/bash
# round 6, re-entry: early-digit schedule measured (error + time), sanity of the shipped-layout tests, share times for DESIGN section 6
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -k "operator_product_variants" -x -q -s > $out/r06e_digits_test.log 2>&1
echo "digits test rc=$?"
grep -E "digits|passed|failed|Error" $out/r06e_digits_test.log | tail -8
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --instrumented-steps 0 --resident-steps 0"
for v in A B C A B C; do
  case $v in A) o="";; B) o="--option bp_digits_early=3";; C) o="--option bp_digits_early=2";; esac
  $B $o 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', '$o', d['ms_per_step'], 'ms', d['value'], 'cells/s', d['operator_product']['A Q']['kernels_ms'], d['operator_product']['A^T Y']['kernels_ms'])"
done 2>&1 | tee $out/r06e_digits_ab.txt
timeout 900 python -m pytest tests/test_gpu_shipped_layout.py tests/test_gpu_scaled_bitplane.py -x -q > $out/r06e_layout_tests.log 2>&1
echo "layout tests rc=$?"; tail -3 $out/r06e_layout_tests.log
bash profiles/tools/share_times.sh 2>&1 | tee $out/r06_share_times.txt
