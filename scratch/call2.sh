#!/bin/bash
out=gpurun_out; mkdir -p $out
./scratch/mfma_fp6_probe.bin 2>&1 | tee $out/r06e_fp6_probe.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --instrumented-steps 0 --resident-steps 0"
for v in A B A B A B; do
  case $v in A) o="";; B) o="--option bp_digits_early=3";; esac
  $B $o 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', '$o', d['ms_per_step'], 'ms', d['value'], 'cells/s')"
done 2>&1 | tee $out/r06e_digits_ab.txt
