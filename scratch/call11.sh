#!/bin/bash
out=gpurun_out; mkdir -p $out
export DDX_OPTIONS=bp_format=mx6
DDX_AB_ORDER="A B A B" bash profiles/tools/ab.sh 2>&1 | python -c "
import sys,ast
for l in sys.stdin:
    l=l.strip()
    if '{' in l:
        i=l.index('{'); d=ast.literal_eval(l[i:]); print(l[:i], {k:v for k,v in d.items() if k.startswith('bitplane')})" | tee $out/r06r_mx_copies_ab.txt
unset DDX_OPTIONS
python profiles/tools/spmm_time.py int8 2>&1 | tail -1 | python -c "
import sys,ast
l=sys.stdin.read().strip(); i=l.index('{'); d=ast.literal_eval(l[i:]); print('int8', {k:v for k,v in d.items() if k.startswith('bitplane')})" | tee -a $out/r06r_mx_copies_ab.txt
DDX_OPTIONS=bp_format=mx6 timeout 600 python -m pytest tests/test_gpu_parity.py -k "pca_scores" -x -q 2>&1 | tail -2
