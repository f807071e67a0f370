#!/bin/bash
# PMC passes over the kNN kernels of the benchmark embedding:  bash profiles/tools/knn_pmc.sh <tag> [cells]
# (counter passes use --kernel-trace only; one pass per counter group)
tag=${1:-r04}; cells=${2:-0}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
cmd="python $repo/profiles/tools/knn_cells_check.py 100000 30000 0.03 $cells 0"
for grp in "fetch:FETCH_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum" "sq:SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "sq2:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "sq3:SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    rm -rf /tmp/kp_$name
    timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/kp_$name -- $cmd > $out/${tag}_knnpmc_$name.log 2>&1
done
cd $repo
python profiles/summarise_pmc.py fetch=/tmp/kp_fetch l2=/tmp/kp_l2 sq=/tmp/kp_sq sq2=/tmp/kp_sq2 sq3=/tmp/kp_sq3 | grep -i "knn\|cells" > $out/${tag}_knn_pmc.txt
cat $out/${tag}_knn_pmc.txt
