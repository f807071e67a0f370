#!/bin/bash
# SQ / LDS / MFMA counters of the operator products and the kNN passes (single device context, one iteration).
# Separate passes per counter group; no sys/hip/hsa tracing together with --pmc.     bash profiles/pmc_sq.sh <tag>
tag=${1:-r02x}
repo=$(pwd); export TMPDIR=/tmp; cd /tmp
short="python $repo/bench.py --steps 1 --warmup 0 --iters 1 --no-cpu-baseline --resident-steps 0 --instrumented-steps 0 --no-exclusive"
export DDX_STREAMS=1
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" "SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/p$i
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/p$i -- $short > $repo/gpurun_out/${tag}_pmc_sq_$i.log 2>&1
done
cd $repo
python profiles/summarise_pmc.py a=/tmp/p1 b=/tmp/p2 c=/tmp/p3 d=/tmp/p4 e=/tmp/p5 f=/tmp/p6 g=/tmp/p7 | grep -E "spmm_lds|k_bp_|knn_emit|knn_bound|knn_select|knn_rescan|mirror_tiles|lv_sweep|lv_apply" > gpurun_out/${tag}_pmc_sq.txt
cat gpurun_out/${tag}_pmc_sq.txt
