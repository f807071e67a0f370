#!/bin/bash
out=gpurun_out; mkdir -p $out
cp doubletdetection_amd/libddx.so /tmp/keep.so
for v in A B A B; do
  cp doubletdetection_amd/_ab/libddx_$v.so doubletdetection_amd/libddx.so
  python profiles/tools/graph_time.py $v 2>&1 | tail -1
done | tee $out/r06m_graph_weights_ab.txt
cp /tmp/keep.so doubletdetection_amd/libddx.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_flavours_fullsize.py -x -q 2>&1 | tail -3
