#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -k "operator_product_variants" -x -q -s > $out/r06f_mx_test.log 2>&1
echo "variants test rc=$?"; grep -E "against|digits|passed|failed|Error|assert" $out/r06f_mx_test.log | tail -12
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --instrumented-steps 0 --resident-steps 0"
for v in A B A B; do
  case $v in A) o="--option bp_format=int8";; B) o="";; esac
  $B $o 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', '$o', d['ms_per_step'], 'ms', d['value'], 'cells/s')"
done 2>&1 | tee $out/r06f_mx_ab.txt
python bench.py --steps 3 --warmup 2 --no-cpu-baseline --resident-steps 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(json.dumps(d['operator_product']));print(json.dumps(d['roofline_top_kernels'][:4]))" | tee $out/r06f_mx_products.txt
timeout 900 python -m pytest tests/test_gpu_shipped_layout.py tests/test_gpu_scaled_bitplane.py tests/test_gpu_parity.py -x -q > $out/r06f_tests.log 2>&1
echo "tests rc=$?"; tail -5 $out/r06f_tests.log
