#!/bin/bash
out=gpurun_out; mkdir -p $out
DDX_AB_ORDER="A B A B" bash profiles/tools/ab.sh 2>&1 | tee $out/r06j_expand_ab_kernels.txt
repo=$(pwd)
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --instrumented-steps 0 --resident-steps 0"
cp doubletdetection_amd/libddx.so /tmp/keep.so
for v in A B A B A B; do
  cp doubletdetection_amd/_ab/libddx_$v.so doubletdetection_amd/libddx.so
  $B 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', d['ms_per_step'], 'ms', d['value'], 'cells/s')"
done 2>&1 | tee $out/r06j_expand_ab_fits.txt
cp /tmp/keep.so doubletdetection_amd/libddx.so
timeout 900 python -m pytest tests/test_gpu_parity.py -k "pca_scores or fit_matches" -x -q 2>&1 | tail -3
