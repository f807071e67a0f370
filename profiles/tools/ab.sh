#!/bin/bash
# A/B of two builds of libddx.so on one GPU box: doubletdetection_amd/_ab/libddx_{A,B}.so, alternating.
repo=$(cd "$(dirname "$0")/../.." && pwd)
cp "$repo/doubletdetection_amd/libddx.so" /tmp/libddx_keep.so
for v in ${DDX_AB_ORDER:-A B A B}; do
    cp "$repo/doubletdetection_amd/_ab/libddx_$v.so" "$repo/doubletdetection_amd/libddx.so"
    python "$repo/profiles/tools/spmm_time.py" $v 2>&1 | tail -1
done
cp /tmp/libddx_keep.so "$repo/doubletdetection_amd/libddx.so"
