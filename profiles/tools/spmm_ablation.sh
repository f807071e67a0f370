#!/bin/bash
# Ablation builds of the LDS operator-product kernel (DDX_SPMM_DBG bits: 1 no operand reads, 2 no entry fetches,
# 4 no staged reads).  Results are wrong by construction; only the kernel times matter.
#   build (here):  bash profiles/tools/spmm_ablation.sh build     -> profiles/tools/variants/libddx_dbg<N>.so
#   run (GPU box): bash profiles/tools/spmm_ablation.sh run
set -u
repo=$(cd "$(dirname "$0")/../.." && pwd)
var=$repo/profiles/tools/variants
mkdir -p "$var"
if [ "${1:-run}" = build ]; then
    cp "$repo/doubletdetection_amd/libddx.so" "$var/libddx_dbg0.so"
    for n in ${DDX_ABLATIONS:-1 2 3 4 7}; do
        touch "$repo/doubletdetection_amd/csrc/k_pca.hip"
        DDX_EXTRA_HIPCC_FLAGS="-DDDX_SPMM_DBG=$n" python -c "from doubletdetection_amd import _build; _build.build(verbose=False)"
        cp "$repo/doubletdetection_amd/libddx.so" "$var/libddx_dbg$n.so"
    done
    touch "$repo/doubletdetection_amd/csrc/k_pca.hip"
    python -c "from doubletdetection_amd import _build; _build.build(verbose=False)"
    exit 0
fi
cp "$repo/doubletdetection_amd/libddx.so" /tmp/libddx_keep.so
for n in 0 ${DDX_ABLATIONS:-1 2 3 4 7}; do
    cp "$var/libddx_dbg$n.so" "$repo/doubletdetection_amd/libddx.so"
    python "$repo/profiles/tools/spmm_time.py" dbg$n 2>&1 | tail -2
done
cp /tmp/libddx_keep.so "$repo/doubletdetection_amd/libddx.so"
