#!/bin/bash
# Counters of the mirror kernels (profiles/tools/mirror_time.py under rocprofv3; separate passes per counter group).
repo=$(pwd); export TMPDIR=/tmp; cd /tmp
cmd="python $repo/profiles/tools/mirror_time.py"
rm -rf /tmp/mp_*
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp_stats -- $cmd > /dev/null 2>&1
grep -h "mirror" $(find /tmp/mp_stats -name "*kernel_stats.csv") | cut -c1-60,150-400
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/mp_$i -- $cmd > /dev/null 2>&1
done
cd $repo
python profiles/summarise_pmc.py a=/tmp/mp_1 b=/tmp/mp_2 c=/tmp/mp_3 d=/tmp/mp_4 e=/tmp/mp_5 f=/tmp/mp_6 g=/tmp/mp_7 h=/tmp/mp_8 | grep -E "mirror|kernel|^#" 
