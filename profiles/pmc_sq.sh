#!/bin/bash
repo=$(pwd); export TMPDIR=/tmp; cd /tmp
short="python $repo/bench.py --steps 1 --warmup 0 --iters 1 --no-cpu-baseline"
i=0
for ctrs in "FETCH_SIZE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/p$i
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/p$i -- $short > $repo/gpurun_out/pmc_sq_$i.log 2>&1
done
cd $repo
python profiles/summarise_pmc.py a=/tmp/p1 b=/tmp/p2 c=/tmp/p3 d=/tmp/p4 e=/tmp/p5 | grep -E "spmm|knn_emit" > gpurun_out/pmc_sq.txt
cat gpurun_out/pmc_sq.txt
