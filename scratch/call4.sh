#!/bin/bash
out=gpurun_out; mkdir -p $out
python profiles/tools/mx_ablation.py 2>&1 | grep mode | tee $out/r06g_mx_ablation.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -k "operator_product_variants" -x -q -s > $out/r06g_mx_test.log 2>&1
echo "variants test rc=$?"; grep -E "against|passed|failed|Error|assert" $out/r06g_mx_test.log | tail -6
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --instrumented-steps 0 --resident-steps 0"
for v in A B A B; do
  case $v in A) o="--option bp_format=int8";; B) o="";; esac
  $B $o 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', '$o', d['ms_per_step'], 'ms', d['value'], 'cells/s')"
done 2>&1 | tee $out/r06g_mx_ab.txt
